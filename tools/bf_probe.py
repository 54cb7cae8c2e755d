"""Bf (binomial + epistasis) at larger sizes than the test suite uses: yeast genotypes n = 200, k = 60 / 150 markers (1830 / 11 325
columns), target = phenotype above its median; whole 20 x 20 x 5 grid timed, k = 60 spot-checked against the oracle."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, pareben_amd
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "yeast_timing_200x600.npz"))
n = int(d["n"]); B = np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64) * 2 - 1
y = (d["y"] > np.median(d["y"])).astype(np.float64)
for k in (60, 150):
    X = np.asfortranarray(B[:, :k])
    t0 = time.time()
    out = pareben_amd.CrossValidate(X, y, nFolds=5, Epis="yes", prior="binomial", return_stats=True)
    st = out["stats"]
    print(json.dumps({"k": k, "columns": k * (k + 1) // 2, "wall_s": time.time() - t0, "timing": st["timing"], "launch": st["launch"],
                      "alpha_opt": out["alpha.optimal"], "lambda_opt": out["lambda.optimal"], "stopped": st["stopped_fits"],
                      "m_max": int(st["counters"][..., 10].max())}), flush=True)
import oracle_lib as O
from pareben_amd.grid import BuildGrid, AssignToFolds
X = np.asfortranarray(B[:, :60]); fid = AssignToFolds(X, 5); a, l = BuildGrid(X, y, 5, "yes")
sel = np.arange(5, 400, 41)
with pareben_amd.Context(X, y, fid, 5, prior="binomial", epis=True) as ctx:
    E, s, c = ctx.run(a[sel], l[sel])
t0 = time.time(); Eo, co, rc = O.cv_grid(X, y, fid, 5, a[sel], l[sel], prior="binomial", epis=True, n_threads=16)
print("k=60 vs oracle: max |dlogL| %.2e, rc %d, oracle %.1f s, adds %d/%d" % (np.abs(E - Eo).max(), rc, time.time() - t0, c[..., 2].sum(), co["n_add"]))
