"""Search (on the GPU) for small problems whose active set passes 1024 columns, cheap enough for the oracle to follow:
prints m_max and the oracle-cost proxy K * sum M^2 over full-stat passes for a list of synthetic single fits."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pareben_amd
from pareben_amd.synth import synthetic_gaussian
from pareben_amd.grid import GetLambdaMax


def ld_design(N, nblk, bsz, nnoise, flip, seed=7):
    rng = np.random.default_rng(seed)
    cols = []; base = []
    for b in range(nblk):
        x = rng.integers(0, 2, N) * 2.0 - 1
        base.append(x)
        for j in range(bsz):
            c = x.copy(); k = rng.choice(N, size=max(1, int(flip * N)), replace=False); c[k] = -c[k]; cols.append(c)
    for j in range(nnoise):
        cols.append(rng.integers(0, 2, N) * 2.0 - 1)
    X = np.asfortranarray(np.stack(cols, axis=1))
    y = np.stack(base, axis=1) @ np.linspace(1.0, 0.5, nblk) + 0.5 * rng.standard_normal(N)
    return X, y


cases = []
for n, p in ((2600, 3200), (3000, 3600), (3000, 4500), (3400, 4000)):
    X, y, _, _ = synthetic_gaussian(n, p)
    lm = GetLambdaMax(X, y)
    for frac, al in ((1e-4, 0.05), (1e-5, 0.05), (1e-6, 0.05), (1e-5, 1.0)):
        cases.append(("syn n=%d p=%d frac=%g alpha=%g" % (n, p, frac, al), X, y, lm * frac, al))
names = None
for name, X, y, lam, al in cases:
    t = time.time()
    try:
        r = pareben_amd.fit_gaussian(X, y, lam, al)
        c = r["counters"]
        print(name, "| m_max", c["m_max"], "m_final", c["m_final"], "inner", c["n_inner"], "fullstat", c["n_fullstat"],
              "K*sumM2 %.2e" % (X.shape[1] * c["sum_m2_full"]), "status", c["status"], "%.1fs" % (time.time() - t), flush=True)
    except pareben_amd.ParebenError as e:
        print(name, "| error", e, flush=True)
