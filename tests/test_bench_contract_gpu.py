"""bench.py's output contract on a GPU box, at a rehearsal size that finishes in seconds: exactly one JSON
line on stdout with the keys the driver reads, the roofline and cpu_baseline objects, and the same
(alpha*, lambda*, cv.error) whether the per-cell errors are kept local or go through the RCCL all-gather
(PAREBEN_BENCH_FORCE_DIST=1: a one-rank process group on the one GPU, the N > 1 code path)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--n", "240", "--p", "700", "--nfolds", "3", "--nalpha", "4", "--nlambda", "6", "--steps", "2", "--warmup", "1"]


def _run(extra_args, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra_args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                                   # ONE JSON line, nothing else on stdout
    return json.loads(lines[0])


def test_bench_json_contract_and_collective_path():
    r = _run(["--cpu-budget-s", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["metric"] == "cv_fits_per_sec" and r["unit"] == "fits/s" and r["n_gpus"] == 1 and r["steps"] == 2 and r["warmup"] == 1
    assert r["higher_is_better"] is True and r["vs_baseline"] is None and r["dtype"] == "f64" and r["data"] == "synthetic"
    assert "workload" in r["config"] and "rehearsal" in r["config"]["workload"]     # not the BASELINE size: labelled as such
    assert r["config"]["fits_per_step"] == 4 * 6 * 3
    assert abs(r["value"] - r["config"]["fits_per_step"] / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    c = r["config"]
    assert c["wall_from_entry_s"] >= c["setup_s"]["ctx_create"] + c["setup_s"]["first_run"] > 0
    assert c["aborted_fits"] == 0
    ro = r["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["kernel"].startswith(("gm_cv_kernel", "gram_kernel"))
    assert (ro["unit"], ro["peak"]) in (("GB/s", 8000.0), ("TFLOP/s", 78.6))
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12 and ro["achieved"] > 0
    assert ro["traffic"] is None                                    # PMC traffic is attached to the profiled workload only
    fk = ro if "mfma" in ro else ro["fit_kernel"]
    assert fk["mfma"]["executed_mfma_flops_per_launch"] > 0 and fk["hbm"]["algorithmic_bytes_per_launch"] > 0
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "fits/s" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
    d = _run(["--cpu-baseline", "0"], {"PAREBEN_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533",
                                        "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    for k in ("alpha_opt", "lambda_opt", "cv_error", "aborted_fits"):
        assert d["config"][k] == r["config"][k], k                  # bit-identical through the all-gather
    assert "cpu_baseline" not in d


def test_bench_gpus_flag_is_honoured():
    """--gpus must agree with the launcher's WORLD_SIZE (a mismatch is an error, not a silent one-GPU run)."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 2 and b"WORLD_SIZE" in p.stderr


def test_bench_binomial_workload():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "config3", "--steps", "1", "--warmup", "0",
                        "--cpu-baseline", "0"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    r = json.loads([l for l in p.stdout.decode().splitlines() if l.strip()][-1])
    assert r["config"]["fits_per_step"] == 2000 and "configs[2]" in r["config"]["workload"]
    ro = r["roofline"]
    assert ro["bound"] == "hbm" and ro["kernel"] == "bm_cv_kernel" and ro["achieved"] > 0
    assert abs(r["config"]["alpha_opt"] - 0.2) < 1e-12 and abs(r["config"]["lambda_opt"] - 0.014589761289879953) < 1e-15
