"""A/B check of the shared full-stat passes: the share of the config-2 grid that one of 8 GPUs would get
(1250 fits), run with PAREBEN_SHARE=0 and =1 -> must be bit-identical; prints both kernel times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pareben_amd
from pareben_amd.grid import BuildGrid, AssignToFolds
from pareben_amd.synth import synthetic_gaussian
from pareben_amd.dist import shard_cells
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
X, y, _, _ = synthetic_gaussian(1000, 10000)
alpha, lam = BuildGrid(X, y, 5, nAlpha=20, nLambda=100)
fid = AssignToFolds(X, 5)
mine = shard_cells(alpha, lam, 0, world)
res = {}
for share in ("0", "1", "2"):
    os.environ["PAREBEN_SHARE"] = share
    with pareben_amd.Context(X, y, fid, 5) as ctx:
        E, st, cnt = ctx.run(alpha[mine], lam[mine])
        res[share] = (E, st, cnt, ctx.last_timing())
    print("share", share, "fits", E.size, res[share][3], flush=True)
same = all(np.array_equal(res["0"][0], res[k][0], equal_nan=True) and np.array_equal(res["0"][1], res[k][1]) and np.array_equal(res["0"][2][..., :11], res[k][2][..., :11]) for k in ("1", "2"))
print("bit-identical:", same, "aborted:", int(((res["2"][1] & 8) != 0).sum()))
sys.exit(0 if same else 1)
