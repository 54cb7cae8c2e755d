"""1000-fit probe (synthetic n=1000 p=10000, 20 alpha x 10 lambda, 5 folds) for profiling runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pareben_amd
from pareben_amd.grid import BuildGrid, AssignToFolds
from pareben_amd.synth import synthetic_gaussian
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 10
X, y, _, _ = synthetic_gaussian(1000, 10000)
alpha, lam = BuildGrid(X, y, 5, nAlpha=20, nLambda=nl)
fid = AssignToFolds(X, 5)
with pareben_amd.Context(X, y, fid, 5) as ctx:
    E, st, cnt = ctx.run(alpha, lam)
    print(ctx.last_timing(), ctx.launch_info())
