"""Kernel time of every rank's share of config 2 at world size N (one MI355X stands in for each rank in turn):
the step time of an N-GPU run is the maximum.  usage: share_ranks.py N"""
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
ms = []
with pareben_amd.Context(X, y, fid, 5) as ctx:
    for r in range(world):
        mine = shard_cells(alpha, lam, r, world)
        ctx.run(alpha[mine], lam[mine])
        ms.append(ctx.last_timing()["total_ms"])
print("world", world, "per-rank ms", [round(v) for v in ms], "max", round(max(ms)), "mean", round(float(np.mean(ms))), flush=True)
