"""Rate of gram_kernel at a given shape (default: BASELINE config 5, n=2000 p=50000 nFolds=10): one run with
a single large-lambda cell, so the launch is the preparation kernels + nFolds near-empty fits.
Nominal flops = nFolds * 2 * N_train * K^2 (the symmetric schedule executes half).  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel figure.  usage: gram_rate.py [n p nfolds]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pareben_amd
from pareben_amd.grid import AssignToFolds
from pareben_amd.synth import synthetic_gaussian

n, p, nf = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2000, 50000, 10)
X, y, _, _ = synthetic_gaussian(n, p)
fid = AssignToFolds(X, nf)
with pareben_amd.Context(X, y, fid, nf) as ctx:
    for rep in range(2):
        E, st, cnt = ctx.run(np.array([1.0]), np.array([1e6]))
        t = ctx.last_timing()
ntr = np.array([(fid != f + 1).sum() for f in range(nf)], dtype=np.float64)
flops = float(np.sum(2.0 * ntr * p * p))
print(json.dumps({"n": n, "p": p, "n_folds": nf, "prep_ms": t["prep_ms"], "nominal_tflops_over_prep": flops / (t["prep_ms"] * 1e-3) / 1e12,
                  "nominal_flops": flops, "launch": ctx.launch_info() if False else None}))
