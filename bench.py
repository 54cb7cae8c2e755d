#!/usr/bin/env python3
"""bench.py -- CV fits/sec of the nFolds x alpha x lambda grid on MI355X (BASELINE.json metric).

Workload (default = BASELINE.json configs[1]): synthetic Gaussian n=1000, p=10000, nFolds=5,
20 alpha x 100 lambda = 10 000 EBelasticNet.Gaussian fits.  One "step" = one complete pass of the
hot path over that grid: per-fold preparation kernels (row split, column statistics, Gram
matrices), the persistent fit kernel, result copy, and -- for N > 1 -- the all-gather of the
per-cell fold errors, followed by the host-side summary / arg-min that yields (alpha*, lambda*).
BASIS / Target / fold ids are staged in HBM before the timed region.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); the cost-sorted cell list is
dealt round-robin to the ranks (total work fixed -> "strong" scaling), one all-gather per step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md, HBM3E spec
FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector peak (spec)


def algorithmic_bytes_flops(cnt, K):
    """Algorithmic HBM bytes / FP64 flops of the fit kernel from its own per-fit event counters
    (DESIGN.md 'Roofline accounting'): Gram-row sweeps per action and per full-stat pass plus the
    K-vector traffic of every inner iteration.  cnt: [..., 12] int64."""
    c = cnt.reshape(-1, cnt.shape[-1]).astype(np.float64).sum(axis=0)
    n_outer, n_inner, n_add, n_del, n_reest, n_full, sm_act, sm_full, sm2_full = c[:9]
    n_act = n_add + n_del + n_reest
    bytes_ = 8.0 * K * (sm_act          # M Gram rows read per add / delete / re-estimate
                        + n_add          # the new feature's own Gram row
                        + sm_full        # M Gram rows per full-stat pass
                        + 6.0 * n_inner  # S_out,Q_out read; dML, root written; S/Q refresh
                        + 4.0 * n_act    # S_in,Q_in read+write per action
                        + 3.0 * n_outer)  # bt0, cs read, bt written
    flops = 2.0 * K * sm2_full + 2.0 * K * sm_act + 30.0 * K * n_inner
    return bytes_, flops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--p", type=int, default=10000)
    ap.add_argument("--nfolds", type=int, default=5)
    ap.add_argument("--nalpha", type=int, default=20)
    ap.add_argument("--nlambda", type=int, default=100)
    ap.add_argument("--cpu-baseline", type=int, default=1, help="0 disables the CPU leg")
    ap.add_argument("--cpu-sample", type=int, default=12, help="number of sampled fits for the CPU leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    from pareben_amd.synth import synthetic_gaussian
    from pareben_amd.dist import shard_cells, all_gather_cells

    X, y, _, _ = synthetic_gaussian(args.n, args.p)
    alpha, lam = BuildGrid(X, y, args.nfolds, "no", nAlpha=args.nalpha, nLambda=args.nlambda)
    folds = AssignToFolds(X, args.nfolds)
    n_cells, nF = len(alpha), args.nfolds
    mine = shard_cells(alpha, lam, rank, world)

    ctx = pareben_amd.Context(X, y, folds, nF, device=dev_index if world > 1 else 0)   # H2D staging, untimed
    state = {}

    def step():
        err, st, cnt = ctx.run(alpha[mine], lam[mine])
        if use_dist:
            fold_err, status = all_gather_cells(mine, err, st, n_cells, nF)
        else:
            fold_err = np.empty((n_cells, nF)); status = np.empty((n_cells, nF), dtype=np.int32)
            fold_err[mine] = err; status[mine] = st
        a_s, l_s, se, cv, idx = summarise_cv(alpha, lam, fold_err, nF)
        state.update(cnt=cnt, status=status, best=(float(a_s[idx]), float(l_s[idx]), float(cv[idx])),
                     timing=ctx.last_timing(), launch=ctx.launch_info())

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_fits = n_cells * nF
    result = None
    # HBM traffic of the dominant kernel comes from a separate rocprofv3 PMC run of this same command
    # (profiles/<round>/traffic.json); it is attached only when the workload is the one profiled.
    traffic, traffic_note = None, None
    try:
        import glob
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))[::-1]:
            tj = json.load(open(tf))
            if tj.get("workload", "").startswith("synthetic gaussian n=%d p=%d nFolds=%d grid=%dalpha x %dlambda" %
                                                 (args.n, args.p, args.nfolds, args.nalpha, args.nlambda)) and world == 1:
                traffic = tj["corrected_bytes"] / tj.get("launches", 1)
                traffic_note = "PMC (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch from %s; raw %.3g B" % (os.path.relpath(tf, ROOT), tj["raw_bytes"])
                break
    except Exception:
        pass
    if rank == 0:
        bytes_, flops = algorithmic_bytes_flops(state["cnt"], args.p)
        fit_ms = state["timing"]["fit_ms"]
        ach = bytes_ / (fit_ms * 1e-3) / 1e9
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
            "data": "synthetic",
            "config": {"workload": "synthetic gaussian n=%d p=%d nFolds=%d grid=%dalpha x %dlambda Epis=no%s"
                                   % (args.n, args.p, nF, args.nalpha, args.nlambda,
                                      " (BASELINE configs[1])" if (args.n, args.p, nF, args.nalpha, args.nlambda) == (1000, 10000, 5, 20, 100)
                                      else " (reduced rehearsal size, not the BASELINE workload)"),
                       "fits_per_step": total_fits, "parallelism": "cells sharded over %d GPU(s)" % world,
                       "wall_to_optimum_s": elapsed / args.steps,
                       "alpha_opt": state["best"][0], "lambda_opt": state["best"][1], "cv_error": state["best"][2],
                       "aborted_fits": int(np.sum(state["status"] & 8 != 0)),
                       "launch": state["launch"], "kernel_ms": state["timing"]},
            "roofline": {"bound": "hbm", "kernel": "gm_cv_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": bytes_, "launch_ms": fit_ms,
                         "fp64_vector": {"achieved_tflops": flops / (fit_ms * 1e-3) / 1e12, "peak_tflops": FP64_PEAK_TFLOPS,
                                         "frac": flops / (fit_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}},
        }
    ctx.close()

    if rank == 0 and world == 1 and args.cpu_baseline:
        # bounded CPU leg: the oracle ("port" of the reference algorithm) on a stratified sample of
        # (cell, fold) fits -- every (nlambda/sample)-th lambda at alpha = 0.5, fold 1 -- on all host
        # cores, one fit per thread.  Throughput = cores x fits / CPU-seconds consumed.
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        try:
            avail = len(os.sched_getaffinity(0))
        except Exception:
            avail = os.cpu_count() or 1
        cores = min(avail, 16)                       # a one-GPU box's CPU share
        L = np.unique(lam)[::-1]
        pick = np.linspace(0, len(L) - 1, args.cpu_sample).round().astype(int)
        a_s = np.full(len(pick), 0.5); l_s = L[pick]
        tr = folds != 1
        fid1 = np.where(tr, 2, 1).astype(np.int32)          # 2 pseudo-folds: only fold 1 is evaluated
        thr = min(cores, len(pick))
        cores = thr                                  # report the threads actually used
        w0, c0 = time.perf_counter(), time.process_time()
        # evaluate only fold 1 of each sampled cell
        Eo, cnt_o, rc = oracle_lib.cv_grid(X, y, fid1, 1, a_s, l_s, n_threads=thr)
        wall, cpu = time.perf_counter() - w0, time.process_time() - c0
        result["cpu_baseline"] = {
            "value": cores * len(pick) / cpu, "unit": "fits/s", "cores": cores, "kind": "port",
            "sample": "%d fits: alpha=0.5, fold 1, lambda indices %s of %d; %.1f CPU-s on %d threads (%.1f s wall); "
                      "value = threads x fits / CPU-seconds (perfect load balance over those threads)"
                      % (len(pick), pick.tolist(), len(L), cpu, thr, wall),
        }
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
