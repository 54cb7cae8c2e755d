"""BASELINE config 3 (yBinomial / BASISbinomial 500 x 481, binomial prior, nFolds = 5, 20 x 20 grid = 2000 fits): kernel time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pareben_amd
from pareben_amd.grid import BuildGrid, AssignToFolds
g = os.path.join(ROOT, "tests", "golden")
X = np.load(os.path.join(g, "BASISbinomial.npy")).astype(np.float64); y = np.load(os.path.join(g, "yBinomial.npy")).astype(np.float64)
alpha, lam = BuildGrid(X, y, 5)
fid = AssignToFolds(X, 5)
with pareben_amd.Context(X, y, fid, 5, prior="binomial") as ctx:
    for rep in range(2):
        E, st, cnt = ctx.run(alpha, lam)
        print(ctx.last_timing(), ctx.launch_info(), "fits/s %.0f" % (E.size / (ctx.last_timing()["total_ms"] / 1e3)), "m_max", int(cnt[..., 10].max()), flush=True)
