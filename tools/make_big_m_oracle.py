"""tests/golden/big_m_oracle.npz: oracle outputs of two single fits whose active sets pass 1040 columns -- beyond what
the device's blocked inverse keeps in LDS -- on a synthetic design small enough for the oracle (n = 3000, p = 4500;
the reference's own basisMax there is 1e7/4500 = 2222, so these are fits the reference handles too).
~20 CPU-minutes each; the GPU test test_active_sets_beyond_1024_columns compares with it."""
import os, sys, time
import numpy as np
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
N, P = 3000, 4500
CASES = ((1e-5, 1.0), (1e-5, 0.05))


def run(case):
    import oracle_lib as O
    from pareben_amd.synth import synthetic_gaussian
    from pareben_amd.grid import GetLambdaMax
    frac, al = case
    X, y, _, _ = synthetic_gaussian(N, P)
    lam = GetLambdaMax(X, y) * frac
    t = time.time()
    o = O.fit_gaussian(X, y, lam, al)
    return dict(lam=lam, alpha=al, beta=o["Beta"][:, 2].copy(), var=o["Beta"][:, 3].copy(), wald=o["wald"], intercept=o["intercept"],
                residual=o["residual"], counters=o["counters"], rc=o["rc"], cpu_s=time.time() - t)


if __name__ == "__main__":
    with ProcessPoolExecutor(2) as ex:
        res = list(ex.map(run, CASES))
    out = {"n": N, "p": P}
    for i, r in enumerate(res):
        print(i, r["lam"], r["alpha"], r["rc"], r["counters"], r["cpu_s"], flush=True)
        pre = "c%d_" % i
        out.update({pre + "lambda": r["lam"], pre + "alpha": r["alpha"], pre + "beta": r["beta"], pre + "var": r["var"], pre + "wald": r["wald"],
                    pre + "intercept": r["intercept"], pre + "residual": r["residual"],
                    pre + "counter_names": np.array(sorted(r["counters"])), pre + "counters": np.array([r["counters"][k] for k in sorted(r["counters"])])})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "big_m_oracle.npz"), **out)
