"""How much of the K-space work of the actions is dead: an inner iteration whose block of actions is followed by a
full-stat pass (noise precision moved by more than 1e-6 in log) has its S_in / Q_in recomputed from scratch, so the
Gram-row sweeps of that block's actions feed nothing but the adds of the same block.  From the decision trace of
config-2 fits (pareben_set_trace): share of action rows (sum of M over actions) that sit in such iterations.

    python tools/deferral_potential.py [cells ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
from trace_divergence import NSLOT, MAXREC  # noqa: E402

if __name__ == "__main__":
    from pareben_amd import _lib
    from pareben_amd.grid import BuildGrid, AssignToFolds
    from pareben_amd.synth import synthetic_gaussian
    X, y, _, _ = synthetic_gaussian(1000, 10000)
    alpha, lam = BuildGrid(X, y, 5, nAlpha=20, nLambda=100)
    fid = AssignToFolds(X, 5)
    cells = [int(c) for c in sys.argv[1:]] or [100, 600, 410, 910, 1410, 1019, 1999, 1205, 710, 739, 200, 755, 502]
    L = _lib.load()
    L.pareben_set_trace.argtypes = [C.POINTER(C.c_uint64), C.c_int64]
    tr = fid != 1
    Xt, yt = np.asfortranarray(X[tr]), np.ascontiguousarray(y[tr])
    tot_rows = tot_dead = tot_act = tot_dead_act = 0
    for c in cells:
        buf = np.zeros((MAXREC + 1) * NSLOT, dtype=np.uint64)
        L.pareben_set_trace(buf.ctypes.data_as(C.POINTER(C.c_uint64)), MAXREC)
        r = _lib.fit_gaussian(Xt, yt, lam[c], alpha[c])
        L.pareben_set_trace(None, 0)
        n = int(buf[0])
        t = buf[NSLOT:(n + 1) * NSLOT].reshape(n, NSLOT)
        ints = t[:, :8].astype(np.int64)
        beta = t[:, 12].copy().view(np.float64)
        prev = np.concatenate([[np.nan], beta[:-1]])
        # a new outer iteration restarts beta bookkeeping only at iter 1; afterwards beta carries over
        moved = np.abs(np.log(beta) - np.log(prev)) > 1e-6
        moved[0] = True
        ntodo, Mb, Ma, sel = ints[:, 5], ints[:, 2], ints[:, 7], ints[:, 6]
        rows = ntodo * (Mb + Ma) / 2.0
        dead = moved & (sel != 10)
        tot_rows += rows.sum(); tot_dead += rows[dead].sum(); tot_act += ntodo.sum(); tot_dead_act += ntodo[dead].sum()
        print("cell %4d alpha %.2f lambda %.4g: inner %5d actions %6d (in iterations followed by a full-stat pass: %5.1f %%), action rows %.3g (%5.1f %% dead), m_max %d"
              % (c, alpha[c], lam[c], n, ntodo.sum(), 100.0 * ntodo[dead].sum() / max(ntodo.sum(), 1), rows.sum(),
                 100.0 * rows[dead].sum() / max(rows.sum(), 1), r["counters"]["m_max"]))
    print("all: actions %d, %.1f %% in iterations followed by a full-stat pass; action rows %.3g, %.1f %% dead"
          % (tot_act, 100.0 * tot_dead_act / tot_act, tot_rows, 100.0 * tot_dead / tot_rows))
