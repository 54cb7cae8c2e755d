"""tests/golden/looser19871_oracle_cell1.json: the oracle's fold SSEs of cell 1 (alpha = 0.95, lambda = lambda_max) of the
3-fold grid on the 19871-column Full_Test design (tests/golden/fulltest_looser19871.npz), flag-and-continue capacity
policy with a 1024-column workspace (EBEN_ORACLE_WS_CAP / PAREBEN_WS_CAP = 1024: fold 2 peaks above that and is cut short
there on both sides; with the default 2048 columns the oracle would need hours for it).  19 CPU-minutes (three threads,
one per fold); the GPU test test_looser19871_cell_vs_oracle compares with it."""
import json, os, sys, time
os.environ["EBEN_ORACLE_WS_CAP"] = "1024"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from pareben_amd.grid import AssignToFolds, BuildGrid

d = np.load(os.path.join(ROOT, "tests", "golden", "fulltest_looser19871.npz")); n = int(d["n"])
X = np.asfortranarray(np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64)[1:] * 2 - 1); y = d["pheno"][1:].astype(np.float64)
fid = AssignToFolds(X, 3, sample_kind="Rounding"); a, l = BuildGrid(X, y, 3)
O.set_capacity_policy(1, 0)
t = time.time()
E, cnt, rc = O.cv_grid(X, y, fid, 3, a[1:2], l[1:2], n_threads=3)
out = {"cell": 1, "alpha": float(a[1]), "lambda": float(l[1]), "fold_sse": E[0].tolist(), "rc": int(rc),
       "note": "fold 2 runs into the 1024-column workspace (status bit 0 of the aggregate): its value is not a completed fit",
       "counters_all_folds": {k: int(np.asarray(v).sum()) if k not in ("status", "m_max", "m_final") else int(np.asarray(v).max()) for k, v in cnt.items()},
       "cpu_s": time.time() - t}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "looser19871_oracle_cell1.json"), "w"), indent=1)
print(out)
