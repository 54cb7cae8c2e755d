"""How large do the active sets of the fold-2 fits of the 19871-column design get when the workspace allows it?
PAREBEN_WS_CAP=<cols> python tools/ws_cap_probe.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pareben_amd
from pareben_amd.grid import AssignToFolds, BuildGrid
d = np.load(os.path.join(ROOT, "tests", "golden", "fulltest_looser19871.npz")); n = int(d["n"])
X = np.asfortranarray(np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64)[1:] * 2 - 1); y = d["pheno"].astype(np.float64)[1:]
fid = AssignToFolds(X, 3, sample_kind="Rounding"); a, l = BuildGrid(X, y, 3)
sel = np.array([0, 10, 19, 200, 210, 219, 380, 390, 399])
t = time.time()
with pareben_amd.Context(X, y, fid, 3) as ctx:
    E, st, cnt = ctx.run(a[sel], l[sel])
    print(ctx.last_timing())
print("wall", time.time() - t)
for i, c in enumerate(sel):
    print(c, "a=%.2f l=%.4g" % (a[c], l[c]), "sse", E[i].round(3).tolist(), "status", st[i].tolist(), "m_max", cnt[i, :, 10].tolist(), "m_final", cnt[i, :, 9].tolist(), "inner", cnt[i, :, 1].tolist())
