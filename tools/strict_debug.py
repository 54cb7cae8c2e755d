"""Strict-order mode, one cell of the Subset_Test table three ways: the CV launch, the per-fit entry + a host-side score, real R."""
import os, sys
os.environ["PAREBEN_STRICT_ORDER"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import numpy as np
import pareben_amd
import trace_divergence as td
from pareben_amd.grid import AssignToFolds, BuildGrid
cell = int(sys.argv[1]) if len(sys.argv) > 1 else 20
X, y = td.load_table("subset5356")
d = np.load(os.path.join(ROOT, "tests", "golden", "subset5356.npz"))
fid = AssignToFolds(X, 3, sample_kind="Rounding")
alpha, lam = BuildGrid(X, y, 3)
key = {(round(float(a_), 6), "%.6e" % l_, int(f_)): m_ for f_, a_, l_, m_ in zip(d["detail_foldId"], d["detail_alpha"], d["detail_lambda"], d["detail_MSE"])}
want = np.array([key[(round(float(alpha[cell]), 6), "%.6e" % lam[cell], f + 1)] for f in range(3)])
with pareben_amd.Context(X, y, fid, 3) as ctx:
    E, st, cnt = ctx.run(alpha[[cell]], lam[[cell]])
print("CV strict   ", E[0], "rel vs R", np.abs(E[0] - want) / want, "counters", cnt[0, :, :6].tolist(), flush=True)
for f in (1, 2, 3):
    tr = fid != f
    r = pareben_amd.fit_gaussian(np.asfortranarray(X[tr]), y[tr], lam[cell], alpha[cell])
    B = r["Beta"]
    pred = np.zeros(int((~tr).sum()))
    Xte = X[~tr]
    for i in np.nonzero(B[:, 2])[0]:
        pred = pred + Xte[:, i] * B[i, 2]
    res = y[~tr] - (r["intercept"] + pred)
    sse = 0.0
    for v in res:
        sse = sse + v * v
    print("per-fit fold", f, sse, "rel vs R", abs(sse - want[f - 1]) / want[f - 1], "rel vs CV", abs(sse - E[0, f - 1]) / sse,
          [r["counters"][k] for k in ("n_outer", "n_inner", "n_add", "n_del", "n_reest", "n_fullstat")], flush=True)
