"""Oracle fixture for BASELINE config 4 at bench size: the paper's Epis timing data (yeast genotypes, n = 200, k = 300
markers -> 45 150 columns with the pairs; paper_materials/Timing Tests/test_time_Gaus.R:13-19, 36-48), nFolds = 5, cells
around the sparse-to-dense transition of the reference's own 20 x 20 grid plus the two corners below it.

The cells, the grid values and the fold ids are those of the HIP run saved by tools/config4_table.py (so that the test
compares the same (alpha, lambda) bit for bit); the oracle (oracle/eben_gm.c, Gf rule set -- PARITY UNPINNED, see
eben_oracle.h) computes the fold SSEs and event counters.  About 30 CPU-minutes on 6 cores.

    python tools/make_config4_golden.py gpurun_out/r03/c4_k300.npz tests/golden/config4_k300_cells.npz"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import oracle_lib  # noqa: E402
from config4_table import load  # noqa: E402

CELLS_BY_K = {300: [40, 52, 55, 56, 59, 60, 69, 79, 100, 399],      # k = 300: 45 150 columns, resident Gram matrices
              600: [41, 50, 51, 56, 60, 64]}                          # k = 600: 180 300 columns, Gram rows on demand

if __name__ == "__main__":
    src, out = sys.argv[1], sys.argv[2]
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    d = np.load(src)
    k = int(d["k"])
    CELLS = CELLS_BY_K[k]
    X, y = load(k)
    alpha, lam, fid = d["alpha"][CELLS], d["lam"][CELLS], d["fold_id"]
    oracle_lib.build()
    oracle_lib.set_capacity_policy(True, 0)          # the HIP build's flag-and-continue policy
    E = np.zeros((len(CELLS), 5))
    t0 = time.time()
    # one (cell, fold) at a time inside cv_grid's OpenMP loop; per-fold calls so that the counters come back per fit
    for f in range(1, 6):
        fid1 = np.where(fid == f, 1, 2).astype(np.int32)
        for ci in range(0, len(CELLS), threads):
            sl = slice(ci, min(ci + threads, len(CELLS)))
            Ef, _, rc = oracle_lib.cv_grid(X, y, fid1, 1, alpha[sl], lam[sl], epis=True, n_threads=threads)
            assert rc == 0
            E[sl, f - 1] = Ef[:, 0]
            print("fold %d cells %s done, %.0f s" % (f, CELLS[sl], time.time() - t0), flush=True)
    np.savez_compressed(out, k=k, cells=np.array(CELLS), alpha=alpha, lam=lam, fold_id=fid, fold_err=E,
                        gpu_fold_err=d["E"][CELLS], gpu_status=d["status"][CELLS], gpu_counters=d["counters"][CELLS])
    rel = np.abs(E - d["E"][CELLS]) / np.abs(E)
    print("max relative difference oracle vs the saved HIP table: %.3g" % rel.max())
    print(rel)
