"""BASELINE config 4 (Gaussian, Epis = "yes") at the sizes `bench.py --workload config4` and the paper's Epis timing table
run (paper_materials/Timing Tests/test_time_Gaus.R:13-19, 36-48: yeast genotypes, n = 200, nFolds = 5; the bundled
yeastFull.rda is a missing blob): k = 300 markers -> 45 150 columns with the pairs (Gram matrices resident)
and k = 600 -> 180 300 columns (Gram rows computed on demand, elasticNetLinearNeFull2.c:67-80 capacity 4K clipped to the
2048-column workspace).

Oracle fixtures: tools/make_config4_golden.py (oracle/eben_gm.c with the Gf rule set -- PARITY UNPINNED: the reference
tree holds no output of an epistasis fit); grid, folds and expected status words: tools/config4_table.py."""
import os

import numpy as np
import pytest

import pareben_amd

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _design(k):
    d = np.load(os.path.join(GOLDEN, "yeast_timing_200x600.npz"))
    B = np.unpackbits(d["bits"], axis=0)[:int(d["n"])].astype(np.float64) * 2.0 - 1.0
    return np.asfortranarray(B[:, :k]), d["y"].astype(np.float64)


def _rel(a, b):
    return np.abs(a - b) / np.abs(b)


def test_config4_k300_cells_vs_oracle_fixture():
    """Ten cells of the reference's own 20 x 20 grid at k = 300 -- six on the sparse-to-dense transition (active sets 4 ...
    202 of 160 training rows), the two neighbours below it, a mid-grid and the smallest-lambda corner -- x 5 folds against
    the oracle: fold SSE to 1e-6 (observed <= 5e-8) and exactly the status words the fixture records (0, or 4 = the
    reference's stale-slot delete).  One listed exception: cell 399 (alpha 0.05, smallest lambda) fold 4, where the noise
    precision sits at its clamp 1e6 / var(y) (residual variance 5e-7: 108+ columns interpolate the 160 training rows) and
    S_in = beta - beta^2 b' Sigma b cancels to noise -- the two builds part at inner iteration 2.99 (a delete with dML 0.097
    against an add with dML 4.70; tools/trace_divergence.py run config4_300 399 4) and end 1.2e-3 apart."""
    d = np.load(os.path.join(GOLDEN, "config4_k300_cells.npz"))
    X, y = _design(300)
    with pareben_amd.Context(X, y, d["fold_id"], 5, epis=True) as ctx:
        E, st, cnt = ctx.run(d["alpha"], d["lam"])
        info = ctx.launch_info()
    assert info["capacity"] == 1200 and info["reference_capacity"] == 1200        # N_train = 160 < 200: basisMax = 4K
    assert np.array_equal(st, d["gpu_status"]) and set(np.unique(st)) <= {0, 4}
    rel = _rel(E, d["fold_err"])
    listed = np.zeros(E.shape, dtype=bool)
    listed[list(d["cells"]).index(399), 3] = True
    assert rel[~listed].max() < 1e-6, rel
    assert rel[listed].max() < 5e-3
    assert cnt[..., 10].max() >= 200 and cnt[list(d["cells"]).index(56), :, 10].max() >= 190     # the optimum cell's active sets pass N_train


def test_config4_k300_full_grid_status_and_properties():
    """The whole 2000-fit grid at k = 300 (the bench workload): exactly the expected status word of every fit (no fit
    stopped; 203 take the reference's stale-slot path), the same arg-min cell, finite positive scores, no cell worse than
    5 x the intercept-only model (the optimum just beats it: the phenotype carries little signal at n = 200), bit-identical rerun in another order, and target-shift invariance on three cells."""
    g = np.load(os.path.join(GOLDEN, "config4_grid_status.npz"))
    X, y = _design(300)
    fid = g["k300_fold_id"].astype(np.int32)
    alpha, lam = g["k300_alpha"], g["k300_lam"]
    with pareben_amd.Context(X, y, fid, 5, epis=True) as ctx:
        E, st, cnt = ctx.run(alpha, lam)
        sel = np.array([56, 60, 131, 399, 40])
        E2, st2, _ = ctx.run(alpha[sel][::-1], lam[sel][::-1])
    assert np.array_equal(st, g["k300_status"]) and (st & 8).sum() == 0 and (st == 4).sum() == 203
    assert np.all(np.isfinite(E)) and np.all(E > 0)
    assert int(np.argmin(E.mean(axis=1))) == int(np.argmin(g["k300_cv_mean"])) == 56
    assert np.array_equal(E2[::-1], E[sel]) and np.array_equal(st2[::-1], st[sel])
    null = np.array([np.sum((y[fid == f + 1] - y[fid != f + 1].mean()) ** 2) for f in range(5)])
    assert np.all(E < 5.0 * null[None, :]) and E[56].sum() < null.sum()              # weak signal: the optimum just beats the intercept-only model
    with pareben_amd.Context(X, y + 2.5, fid, 5, epis=True) as ctx:
        Es, _, _ = ctx.run(alpha[sel[:3]], lam[sel[:3]])
    assert _rel(Es, E[sel[:3]]).max() < 1e-6


def test_config4_k600_gram_rows_on_demand_vs_oracle():
    """k = 600 -> 180 300 columns: five 260 GB Gram matrices cannot be resident, so the fit kernel fills a pool of Gram rows
    on demand (gm_row / gm_rows_prefetch).  Six cells around the transition (active sets 4 ... 217) x 5 folds against the
    oracle to 1e-6 (one listed near-interpolating fit: 7.8e-4), the expected status words, and the whole grid's status words
    (2000 fits, ~25 s)."""
    d = np.load(os.path.join(GOLDEN, "config4_k600_cells.npz"))
    g = np.load(os.path.join(GOLDEN, "config4_grid_status.npz"))
    X, y = _design(600)
    fid = d["fold_id"]
    assert np.array_equal(fid, g["k600_fold_id"])
    with pareben_amd.Context(X, y, fid, 5, epis=True) as ctx:
        E, st, cnt = ctx.run(d["alpha"], d["lam"])
        info = ctx.launch_info()
        Eg, stg, _ = ctx.run(g["k600_alpha"], g["k600_lam"], want_counters=False)
    assert info["capacity"] == 2048                                               # 4K = 2400 clipped to the workspace bound
    assert np.array_equal(st, d["gpu_status"])
    rel = _rel(E, d["fold_err"])
    listed = np.zeros(E.shape, dtype=bool)
    listed[list(d["cells"]).index(60), 3] = True        # near-interpolating fit (104 columns on 160 rows, residual variance 1e-5): see the k = 300 test
    assert rel[~listed].max() < 1e-6, rel               # observed <= 4.2e-7
    assert rel[listed].max() < 5e-3                     # observed 7.8e-4
    assert cnt[..., 10].max() >= 200
    assert np.array_equal(stg, g["k600_status"]) and (stg & 8).sum() == 0
    assert int(np.argmin(Eg.mean(axis=1))) == int(np.argmin(g["k600_cv_mean"]))
    cells = list(d["cells"])
    assert np.array_equal(Eg[cells], E)                                            # same fits inside the full launch: same bits
