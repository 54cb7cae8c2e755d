import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a CPU test of a minute or more (still part of the default CPU suite)")


@pytest.fixture(scope="session")
def golden():
    class G:
        BASIS = np.load(os.path.join(GOLDEN, "BASIS.npy")).astype(np.float64)
        y = np.load(os.path.join(GOLDEN, "y.npy"))
        BASISbinomial = np.load(os.path.join(GOLDEN, "BASISbinomial.npy")).astype(np.float64)
        yBinomial = np.load(os.path.join(GOLDEN, "yBinomial.npy")).astype(np.float64)
        config1 = np.load(os.path.join(GOLDEN, "config1_gm.npz"))
        basis481 = np.load(os.path.join(GOLDEN, "basis481_gm.npz"))
        config3 = np.load(os.path.join(GOLDEN, "config3_bm.npz"))
        config4 = np.load(os.path.join(GOLDEN, "config4_gf.npz"))
        rds = np.load(os.path.join(GOLDEN, "rds_10000.npz"))
        import json
        known = json.load(open(os.path.join(GOLDEN, "survey_known_answers.json")))
    return G


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


def synthetic_gaussian(n, p, n_causal=20, seed=20251004):
    """SURVEY.md 8(d) config-2 style data: X ~ N(0,1) iid, n_causal effects ~ N(0,1), unit noise.
    (numpy Generator stream; the benchmark only needs the shape and distribution.)"""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p))
    idx = rng.choice(p, size=n_causal, replace=False)
    beta = rng.standard_normal(n_causal)
    y = X[:, idx] @ beta + rng.standard_normal(n)
    return np.asfortranarray(X), y


@pytest.fixture(scope="session")
def yeast():
    """Design of the stored real-R CrossValidate() run (paper_materials/.../10000_Features):
    3803 x 10000 (+-1) genotypes and the phenotype; folds from R 3.5's sampler (the run used R 3.5.0)."""
    d = np.load(os.path.join(GOLDEN, "yeast_looser10000.npz"))
    n, p = int(d["n"]), int(d["p"])
    G = np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64) * 2.0 - 1.0
    return np.asfortranarray(G), d["pheno"].astype(np.float64)


@pytest.fixture(scope="session")
def fulltest():
    """Inputs and stored real-R outputs of paper_materials/Real Data Analysis/Full_Test and Subset_Test ("subset5356":
    read with col_names, all rows) (tools/make_golden_fulltest.py).
    fulltest(name) -> (X, y, fixture); the rows the authors' runs saw (their read.delim() took the first sample as a
    header line, Full_Test/dataprep.R:3-4), +-1 genotypes as doubles, column-major."""
    def load(name):
        d = np.load(os.path.join(GOLDEN, ("%s.npz" if name.startswith("subset") else "fulltest_%s.npz") % name))
        n, k = int(d["n"]), int(d["drop_first_row"])
        X = np.unpackbits(d["bits"], axis=0)[:n].astype(np.float64)[k:] * 2.0 - 1.0
        return np.asfortranarray(X), d["pheno"].astype(np.float64)[k:], d
    return load
