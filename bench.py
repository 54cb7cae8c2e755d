#!/usr/bin/env python3
"""bench.py -- CV fits/sec of the nFolds x alpha x lambda grid on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload config2|config3|config4|config5]

Workloads (BASELINE.json `configs`):
  config2 (default, the configuration the metric is quoted on): synthetic Gaussian n=1000, p=10000, nFolds=5,
          20 alpha x 100 lambda = 10 000 EBelasticNet.Gaussian fits (gm_cv_kernel)
  config3 yBinomial / BASISbinomial (500 x 481), binomial prior, nFolds=5, 20 x 20 = 2000 fits (bm_cv_kernel)
  config4 yeast genotypes of the authors' Epis timing table (n=200, k=300 -> 45 150 columns), Gaussian,
          Epis="yes", nFolds=5, 20 x 20 = 2000 fits (expand_kernel + gram_kernel + gm_cv_kernel)
  config5 synthetic Gaussian n=2000, p=50000, nFolds=10, 20 alpha x 200 lambda = 40 000 fits

One "step" = one complete pass of the hot path over the grid: per-fold preparation kernels (row split, column
statistics, Gram matrices), the persistent fit kernel, result copy, -- for N > 1 -- the all-gather of the
per-cell fold errors, and the host-side summary / arg-min that yields (alpha*, lambda*).  BASIS / Target / fold
ids are staged in HBM before the timed region; `config.wall_from_entry_s` is the cold first call including
BuildGrid, AssignToFolds, context creation (H2D, hipMalloc) and the first run.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); the cost-sorted cell list is dealt
round-robin to the ranks (total work fixed -> "strong" scaling), one all-gather per step.  Started without a
launcher (`python bench.py --gpus N`, WORLD_SIZE unset) it starts the N ranks itself.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP64_MFMA_PEAK_TFLOPS = 78.6  # FP64 matrix = FP64 vector rate on MI355X: 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz
                              # (v_mfma_f64_16x16x4_f64 measured at 64.5 cycles, 77.9 TFLOP/s: tools/ubench/mfma_f64_rate.hip)

WORKLOADS = {
    "config2": dict(kind="synthetic", n=1000, p=10000, nfolds=5, nalpha=20, nlambda=100, prior="gaussian", epis=False,
                    tag="BASELINE configs[1]"),
    "config5": dict(kind="synthetic", n=2000, p=50000, nfolds=10, nalpha=20, nlambda=200, prior="gaussian", epis=False,
                    tag="BASELINE configs[4]"),
    "config3": dict(kind="binomial", nfolds=5, nalpha=20, nlambda=20, prior="binomial", epis=False, tag="BASELINE configs[2]"),
    "config4": dict(kind="yeast_epis", k=300, nfolds=5, nalpha=20, nlambda=20, prior="gaussian", epis=True, tag="BASELINE configs[3]"),
}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (before anything in this
    process touches the GPU), relay rank 0's JSON line, exit with the worst return code."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = rc or p.wait()
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc


def load_workload(name, args, np):
    """-> (X, y, prior, epis, nfolds, nalpha, nlambda, description)"""
    w = dict(WORKLOADS[name])
    for k in ("n", "p", "nfolds", "nalpha", "nlambda"):
        v = getattr(args, k)
        if v:
            w[k] = v
    g = os.path.join(ROOT, "tests", "golden")
    if w["kind"] == "synthetic":
        from pareben_amd.synth import synthetic_gaussian
        X, y, _, _ = synthetic_gaussian(w["n"], w["p"])
        full = all(w[k] == WORKLOADS[name][k] for k in ("n", "p", "nfolds", "nalpha", "nlambda"))
        desc = "synthetic gaussian n=%d p=%d nFolds=%d grid=%dalpha x %dlambda Epis=no%s" % (
            w["n"], w["p"], w["nfolds"], w["nalpha"], w["nlambda"],
            " (%s)" % w["tag"] if full else " (reduced rehearsal size, not the BASELINE workload)")
    elif w["kind"] == "binomial":
        X = np.asfortranarray(np.load(os.path.join(g, "BASISbinomial.npy")).astype(np.float64))
        y = np.load(os.path.join(g, "yBinomial.npy")).astype(np.float64).reshape(-1)
        desc = "yBinomial/BASISbinomial %dx%d binomial nFolds=%d grid=%dalpha x %dlambda Epis=no (%s)" % (
            X.shape[0], X.shape[1], w["nfolds"], w["nalpha"], w["nlambda"], w["tag"])
    else:
        d = np.load(os.path.join(g, "yeast_timing_200x600.npz"))
        n = int(d["n"])
        B = np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64) * 2.0 - 1.0
        X = np.asfortranarray(B[:, :w["k"]])
        y = d["y"].astype(np.float64)
        desc = ("yeast genotypes n=%d k=%d (%d columns with pairs) gaussian Epis=yes nFolds=%d grid=%dalpha x %dlambda (%s at the "
                "paper's own Epis timing size; the bundled yeastFull.rda is a missing blob)" % (
                    n, w["k"], w["k"] * (w["k"] + 1) // 2, w["nfolds"], w["nalpha"], w["nlambda"], w["tag"]))
    return X, y, w["prior"], w["epis"], w["nfolds"], w["nalpha"], w["nlambda"], desc


def gm_algorithmic(cnt, K, np):
    """Gaussian fit kernel, per launch, from its own per-fit event counters (DESIGN.md 'Roofline accounting'):
    HBM bytes this algorithm has to move (Gram-row sweeps per action / full-stat pass + the K-vector traffic of every
    inner iteration), nominal FP64 flops, and the flops actually executed on the matrix cores."""
    c = cnt.reshape(-1, cnt.shape[-1]).astype(np.float64).sum(axis=0)
    n_outer, n_inner, n_add, n_del, n_reest, n_full, sm_act, sm_full, sm2_full = c[:9]
    n_act = n_add + n_del + n_reest
    sm_swept = c[13] if len(c) > 13 else sm_act        # rows the action sweeps really read (a sweep that a full-stat pass overtakes is never run)
    bytes_ = 8.0 * K * (sm_swept + n_add + sm_full + 6.0 * n_inner + 4.0 * n_act + 3.0 * n_outer)
    flops_nominal = 2.0 * K * sm2_full + 2.0 * K * sm_act + 30.0 * K * n_inner
    mfma_flops = 8192.0 * c[12]
    return bytes_, flops_nominal, mfma_flops


def bm_algorithmic(cnt, K, N, np):
    """Binomial fit kernel: no cacheable BASIS_PHI, one sweep of the N x K design per action and per
    full-stat pass, the model panel re-read, the K-vector traffic of every inner iteration."""
    c = cnt.reshape(-1, cnt.shape[-1]).astype(np.float64).sum(axis=0)
    n_outer, n_inner, n_add, n_del, n_reest, n_full, sm_act, sm_full, sm2_full = c[:9]
    bytes_ = 8.0 * (N * K * (n_add + n_del + n_reest + n_full) + N * (sm_act + sm_full) + 6.0 * K * n_inner)
    flops = 2.0 * N * K * (sm_act + sm_full) + 2.0 * K * sm2_full
    return bytes_, flops


def cpu_leg(np, X, y, folds, nF, alpha, lam, prior, epis, budget_s, heavy):
    """Bounded CPU leg: the oracle ("port" of the reference algorithm, one fit per thread) on a stratified sample of
    the same grid -- 3 alpha (first / middle / last of the alpha set) x the lambda deciles, one held-out fold after the
    other until the time budget is used up (at least one fold).  Throughput = threads x fits / CPU-seconds consumed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    A = np.unique(alpha)[::-1]
    L = np.unique(lam)[::-1]
    a_pick = A[[0, len(A) // 2, len(A) - 1]] if not heavy else A[[len(A) // 2]]
    l_idx = np.unique(np.round(np.linspace(0, len(L) - 1, 10 if not heavy else 5)).astype(int))
    a_s = np.repeat(a_pick, len(l_idx)); l_s = np.tile(L[l_idx], len(a_pick))
    thr = max(1, min(avail, 16, len(a_s)))               # a one-GPU box's CPU share
    fits, done, cpu, wall = 0, [], 0.0, 0.0
    for f in range(1, nF + 1):
        fid1 = np.where(folds == f, 1, 2).astype(np.int32)  # 2 pseudo-folds: only the one holding out fold f is evaluated
        w0, c0 = time.perf_counter(), time.process_time()
        oracle_lib.cv_grid(X, y, fid1, 1, a_s, l_s, prior=prior, epis=epis, n_threads=thr)
        wall += time.perf_counter() - w0; cpu += time.process_time() - c0
        fits += len(a_s); done.append(f)
        if wall > budget_s or heavy:
            break
    return {"value": thr * fits / cpu, "unit": "fits/s", "cores": thr, "kind": "port",
            "sample": "%d fits: alpha in %s x lambda indices %s of %d x held-out fold(s) %s of %d; %.1f CPU-s on %d threads "
                      "(%.1f s wall); value = threads x fits / CPU-seconds (perfect load balance over those threads)"
                      % (fits, [round(float(v), 4) for v in a_pick], l_idx.tolist(), len(L), done, nF, cpu, thr, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="config2")
    ap.add_argument("--n", type=int, default=0, help="synthetic workloads: override the size (rehearsal)")
    ap.add_argument("--p", type=int, default=0)
    ap.add_argument("--nfolds", type=int, default=0)
    ap.add_argument("--nalpha", type=int, default=0)
    ap.add_argument("--nlambda", type=int, default=0)
    ap.add_argument("--cpu-baseline", type=int, default=1, help="0 disables the CPU leg")
    ap.add_argument("--cpu-budget-s", type=float, default=45.0, help="wall-clock budget of the CPU leg (checked between folds)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        sys.exit(2)
    import numpy as np
    import torch
    # PAREBEN_BENCH_BACKEND=gloo + PAREBEN_BENCH_ONE_DEVICE=1 rehearse the N > 1 path on a one-GPU box
    backend = os.environ.get("PAREBEN_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("PAREBEN_BENCH_ONE_DEVICE") else local_rank
    # PAREBEN_BENCH_FORCE_DIST=1 initialises the process group (and takes the all-gather path) even at
    # world size 1, so the RCCL code path can be exercised on a one-GPU box
    use_dist = world > 1 or bool(os.environ.get("PAREBEN_BENCH_FORCE_DIST"))
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(dev_index)
        # RCCL prints a version banner on stdout when the first communicator comes up; stdout must
        # carry exactly one JSON line, so the banner is sent to stderr (fd-level, it is printed from C)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    else:
        dist = None
        torch.cuda.set_device(0)

    import pareben_amd
    from pareben_amd.grid import BuildGrid, AssignToFolds, summarise_cv
    from pareben_amd.dist import shard_cells, all_gather_cells

    X, y, prior, epis, nF, n_alpha, n_lambda, desc = load_workload(args.workload, args, np)
    n, p = X.shape
    K = p * (p + 1) // 2 if epis else p
    binom = prior == "binomial"

    # ---- the cold first call, from CrossValidate's entry: grid, folds, context (H2D + device buffers), first run
    t_entry = time.perf_counter()
    alpha, lam = BuildGrid(X, y, nF, "yes" if epis else "no", nAlpha=n_alpha, nLambda=n_lambda, device=dev_index if world > 1 else 0)
    folds = AssignToFolds(X, nF)
    t_grid = time.perf_counter() - t_entry
    n_cells = len(alpha)
    mine = shard_cells(alpha, lam, rank, world)
    t0 = time.perf_counter()
    ctx = pareben_amd.Context(X, y, folds, nF, prior=prior, epis=epis, device=dev_index if world > 1 else 0)
    t_ctx = time.perf_counter() - t0
    state = {}

    def step():
        err, st, cnt = ctx.run(alpha[mine], lam[mine])
        if use_dist:
            fold_err, status = all_gather_cells(mine, err, st, n_cells, nF)
        else:
            fold_err = np.empty((n_cells, nF)); status = np.empty((n_cells, nF), dtype=np.int32)
            fold_err[mine] = err; status[mine] = st
        a_s, l_s, se, cv, idx = summarise_cv(alpha, lam, fold_err, nF, prior)
        state.update(cnt=cnt, status=status, best=(float(a_s[idx]), float(l_s[idx]), float(cv[idx])),
                     timing=ctx.last_timing(), launch=ctx.launch_info())

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # a step of the large workloads runs for minutes without a word: a line on stderr every minute says the run is alive
    # (rank 0 only; a sleeping Python thread, nothing in the timed path)
    import threading
    alive = threading.Event()
    if rank == 0:
        def heartbeat(t_start=time.perf_counter()):
            while not alive.wait(60.0):
                sys.stderr.write("bench.py: %s running, %.0f s\n" % (desc, time.perf_counter() - t_start)); sys.stderr.flush()
        threading.Thread(target=heartbeat, daemon=True).start()

    t_first = None
    for w in range(args.warmup):
        t0 = time.perf_counter()
        step()
        if w == 0:
            t_first = time.perf_counter() - t0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    alive.set()
    if t_first is None:                               # no warm-up: the first timed step was the cold one
        t_first = elapsed / args.steps
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_fits = n_cells * nF
    result = None
    if rank == 0:
        tim, launch = state["timing"], state["launch"]
        fit_ms, prep_ms = tim["fit_ms"], tim["prep_ms"]
        # PMC figures of the dominant kernel come from separate rocprofv3 passes of this same command
        # (tools/refresh_profiles.sh -> profiles/<round>/pmc_<workload>.json), attached when they are for this workload
        pmc = None
        try:
            import glob
            for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_%s.json" % args.workload)))[::-1]:
                tj = json.load(open(tf))
                if tj.get("workload") == desc and world == 1:
                    pmc = dict(tj, file=os.path.relpath(tf, ROOT))
                    break
        except Exception:
            pmc = None
        traffic = pmc["hbm_bytes_per_launch"] if pmc else None
        ntr = float(np.mean([(folds != f + 1).sum() for f in range(nF)]))
        if binom:
            bytes_, flops = bm_algorithmic(state["cnt"], K, ntr, np)
            ach = bytes_ / (fit_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "bm_cv_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": bytes_, "launch_ms": fit_ms,
                    "formula": "SURVEY 8(d) binomial: 8*[N*K*(A+D+R+F) + N*sum M_t + 6*K*I] from the kernel's event counters",
                    "fp64_flops_per_launch": flops}
        else:
            bytes_, flops_nom, mfma_flops = gm_algorithmic(state["cnt"], K, np)
            hbm = {"achieved": bytes_ / (fit_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": bytes_ / (fit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_,
                   "note": "bytes this (Gram-space) algorithm has to move per launch, DESIGN.md section 5; runs of adds share one sweep, "
                           "so the measured traffic can be below it"}
            mf = {"achieved": mfma_flops / (fit_ms * 1e-3) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": mfma_flops / (fit_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, "executed_mfma_flops_per_launch": mfma_flops,
                  "nominal_fp64_flops_per_launch": flops_nom,
                  "note": "flops executed on the FP64 matrix cores (mfma_tiles counter x 8192: symmetric full-stat schedule + blocked "
                          "inverse, padding to 16-row blocks included) over the FP64 matrix peak = matrix-pipe busy fraction"}
            gram_flops = float(sum(2.0 * float((folds != f + 1).sum()) * K * K for f in range(nF)))   # executed: gram_kernel computes all K^2 entries (the symmetric variant was dropped, tools/gram_rate.py)
            if fit_ms >= prep_ms:
                bound = "mfma" if mf["frac"] >= 0.05 else "hbm"
                main_obj = mf if bound == "mfma" else hbm
                roof = {"bound": bound, "kernel": "gm_cv_kernel", "achieved": main_obj["achieved"], "peak": main_obj["peak"],
                        "unit": main_obj["unit"], "frac": main_obj["frac"], "traffic": traffic, "launch_ms": fit_ms,
                        "mfma": mf, "hbm": hbm}
            else:                                      # the per-fold preparation (Gram matrices) outweighs the fits
                ach = gram_flops / (prep_ms * 1e-3) / 1e12
                roof = {"bound": "mfma", "kernel": "gram_kernel (+ split / expand / column statistics: HIP events bracket the whole preparation)",
                        "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                        "traffic": traffic, "launch_ms": prep_ms, "executed_flops": gram_flops,
                        "fit_kernel": {"launch_ms": fit_ms, "mfma": mf, "hbm": hbm}}
        if pmc:
            roof["pmc"] = pmc
        st = state["status"]
        result = {
            "metric": "cv_fits_per_sec",
            "value": total_fits * args.steps / elapsed,
            "unit": "fits/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if args.workload in ("config2", "config5") else "bundled (tests/golden, derived from the reference's data files)",
            "config": {"workload": desc, "fits_per_step": total_fits, "parallelism": "cells sharded over %d GPU(s)" % world,
                       "wall_to_optimum_s": elapsed / args.steps,
                       "wall_from_entry_s": t_grid + t_ctx + t_first,
                       "setup_s": {"build_grid_and_folds": t_grid, "ctx_create": t_ctx, "first_run": t_first},
                       "alpha_opt": state["best"][0], "lambda_opt": state["best"][1], "cv_error": state["best"][2],
                       "aborted_fits": int(np.sum(st & 8 != 0)), "fits_past_reference_basisMax": int(np.sum(st & 1 != 0)),
                       "status_histogram": {str(int(k)): int(v) for k, v in zip(*np.unique(st, return_counts=True))},
                       "active_set_max": int(state["cnt"][..., 10].max()),
                       "event_totals": {k: int(v) for k, v in zip(pareben_amd._lib.COUNTER_NAMES, state["cnt"].reshape(-1, state["cnt"].shape[-1]).sum(axis=0))
                                        if k not in ("m_final", "m_max", "status")},
                       "launch": launch, "kernel_ms": tim},
            "roofline": roof,
        }
    ctx.close()

    if rank == 0 and world == 1 and args.cpu_baseline:
        result["cpu_baseline"] = cpu_leg(np, X, y, folds, nF, alpha, lam, prior, epis, args.cpu_budget_s,
                                         heavy=(n * K > 4e7))
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
