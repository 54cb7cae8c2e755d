"""How far does the REFERENCE algorithm itself move when its BLAS sums in a different order?

Runs single (cell, fold) fits of the stored real-R table (tests/golden/rds_10000.npz, yeast 3803 x 10000)
with the oracle in netlib accumulation order (liboracle.so: reproduces real R to ~1e-15) and with the
ddot stand-in summing in four partial sums like an optimised BLAS (liboracle_blas4.so).  On the long
alpha = 1 add/delete trajectories the second one lands on a different model -- the same size of
difference the HIP path shows against real R on exactly those fits ("Parity on chaotic fits").

usage: blas_order_sensitivity.py CELL FOLD [netlib|blas4]      (one fit = 5-15 min of one CPU core)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
variant = sys.argv[3] if len(sys.argv) > 3 else "netlib"
if variant == "blas4":
    os.environ["EBEN_ORACLE_LIB"] = os.path.join(ROOT, "oracle", "liboracle_blas4.so")
import numpy as np
import oracle_lib as O
from pareben_amd.grid import AssignToFolds

cell, fold = int(sys.argv[1]), int(sys.argv[2])
d = np.load(os.path.join(ROOT, "tests", "golden", "yeast_looser10000.npz"))
n = int(d["n"])
G = np.asfortranarray(np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64) * 2.0 - 1.0)
y = d["pheno"].astype(np.float64)
gold = np.load(os.path.join(ROOT, "tests", "golden", "rds_10000.npz"))
row = cell * 3 + (fold - 1)
al, lm, r_sse = float(gold["detail_alpha"][row]), float(gold["detail_lambda"][row]), float(gold["detail_MSE"][row])
fid = AssignToFolds(G, 3, sample_kind="Rounding")
tr, te = fid != fold, fid == fold
t0 = time.time()
r = O.fit_gaussian(np.asfortranarray(G[tr]), y[tr], lm, al)
nz = np.nonzero(r["Beta"][:, 2])[0]
pred = r["intercept"] + G[te][:, nz] @ r["Beta"][nz, 2]
sse = float(np.sum((y[te] - pred) ** 2))
print(json.dumps({"variant": variant, "cell": cell, "fold": fold, "alpha": al, "lambda": lm, "sse": sse, "real_r": r_sse,
                  "rel_vs_real_r": abs(sse - r_sse) / r_sse, "seconds": time.time() - t0,
                  "n_inner": r["counters"]["n_inner"], "m_final": r["counters"]["m_final"], "m_max": r["counters"]["m_max"]}))
