"""BASELINE config 5 shape (synthetic n=2000, p=50000, nFolds=10, 20 x 20 grid) on the GPU through the
on-demand Gram-row pool.  usage: config5_probe.py [n_alpha_sub] [n_lambda_sub] [oracle_cells]
Runs a stratified sub-grid (or everything with 20 20), reports timing / counters and checks
`oracle_cells` of the largest-lambda cells against the oracle.  Writes gpurun_out/config5_probe.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pareben_amd
from pareben_amd.grid import BuildGrid, AssignToFolds, summarise_cv
from pareben_amd.synth import synthetic_gaussian

na = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_or = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n, p, nf = 2000, 50000, 10
t0 = time.time()
X, y, _, _ = synthetic_gaussian(n, p)
alpha, lam = BuildGrid(X, y, nf, nAlpha=20, nLambda=20)
fid = AssignToFolds(X, nf)
print("data + grid %.1f s" % (time.time() - t0), flush=True)
A = alpha.reshape(20, 20); L = lam.reshape(20, 20)          # [alpha index, lambda index]
ai = np.unique(np.round(np.linspace(0, 19, na)).astype(int))
li = np.unique(np.round(np.linspace(0, 19, nl)).astype(int))
sel = (ai[:, None] * 20 + li[None, :]).ravel()
t0 = time.time()
with pareben_amd.Context(X, y, fid, nf) as ctx:
    print("context %.1f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    E, st, cnt = ctx.run(alpha[sel], lam[sel])
    wall = time.time() - t0
    tim, info = ctx.last_timing(), ctx.launch_info()
rep = {"n": n, "p": p, "n_folds": nf, "cells": int(len(sel)), "fits": int(E.size), "wall_s": wall,
       "timing_ms": dict(tim), "launch": {k: int(v) for k, v in info.items()},
       "aborted": int(((st & 8) != 0).sum()), "status_hist": {str(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))},
       "m_max": int(cnt[..., 10].max()), "m_final_mean": float(cnt[..., 9].mean()),
       "n_add_total": int(cnt[..., 2].sum()), "n_inner_total": int(cnt[..., 1].sum()),
       "fits_per_s": E.size / (tim["total_ms"] / 1e3)}
cv = E.mean(axis=1)                                      # Results.Summary$MSE = mean of the fold SSEs; NaN where a fit was stopped
rep["cells_with_stopped_fits"] = int(np.isnan(cv).sum())
if np.all(np.isnan(cv)):
    rep["best_cell"] = None
else:
    best = int(np.nanargmin(cv))
    rep["best_cell"] = {"alpha": float(alpha[sel][best]), "lambda": float(lam[sel][best]), "cv_error": float(cv[best])}
rep["flagged_past_basisMax"] = int(((st & 1) != 0).sum())
rep["m_max_by_lambda_index"] = {int(l): int(cnt[..., 10].reshape(len(ai), len(li), nf)[:, j, :].max()) for j, l in enumerate(li)}
ab = np.argwhere((st & 8) != 0)
rep["stopped"] = [{"alpha": float(alpha[sel][c]), "lambda_index": int(np.searchsorted(-np.unique(lam)[::-1], -lam[sel][c])), "fold": int(f) + 1,
                   "status": int(st[c, f]), "m_max": int(cnt[c, f, 10]), "m_final": int(cnt[c, f, 9]), "n_inner": int(cnt[c, f, 1]),
                   "n_outer": int(cnt[c, f, 0])} for c, f in ab[:60]]
print(json.dumps(rep), flush=True)
if n_or > 0:
    import oracle_lib
    order = np.argsort(-lam[sel])[:n_or]
    t0 = time.time()
    Eo, _, rc = oracle_lib.cv_grid(X, y, fid, nf, alpha[sel][order], lam[sel][order], n_threads=10)
    rel = np.abs(E[order] - Eo) / np.abs(Eo)
    rep["oracle"] = {"cells": int(n_or), "rc": int(rc), "seconds": time.time() - t0, "max_rel_fold_sse": float(rel.max())}
    print(json.dumps(rep["oracle"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "config5_probe.json"), "w") as f:
    json.dump(rep, f, indent=1)
