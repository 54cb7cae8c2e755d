"""The device source of the Gaussian fit (pareben_amd/csrc/gm_fit.h), compiled for the CPU with one
thread per workgroup, against the oracle: same action sequence (all event counters equal) and
fold SSE within 1e-10 relative.  This checks the Gram-space reformulation and every control-flow
quirk without a GPU; the parallel execution itself is covered by the -m gpu tests."""
import numpy as np

import emul_lib
from pareben_amd.grid import BuildGrid, AssignToFolds

NAMES = ("n_outer", "n_inner", "n_add", "n_del", "n_reest", "n_fullstat", "sum_m_action",
         "sum_m_full", "sum_m2_full", "m_final")


def test_config1_all_cells(golden):
    g = golden.config1
    X, y = golden.BASIS[:50, :100], golden.y[:50]
    E, st, cnt = emul_lib.cv_grid(X, y, g["fold_id"], 3, g["alpha"], g["lam"])
    rel = np.abs(E - g["fold_err"]) / np.abs(g["fold_err"])
    assert rel.max() < 1e-10
    assert set(np.unique(st)) <= {0, 4}              # 4 = reference's stale-slot delete path
    assert (st == 4).sum() == 11
    tot = cnt.sum(axis=(0, 1))
    want = dict(zip(g["counter_names"], g["counters"]))
    for i, n in enumerate(NAMES):
        assert tot[i] == want[n], n


def test_basis481_subgrid(golden):
    g = golden.basis481
    sel = [0, 2, 4, 7, 12]                           # keep the CPU suite short
    E, st, cnt = emul_lib.cv_grid(golden.BASIS, golden.y, g["fold_id"], 5, g["alpha"][sel], g["lam"][sel])
    ref = g["fold_err"][sel]
    assert (np.abs(E - ref) / np.abs(ref)).max() < 1e-10
    assert cnt[..., 10].max() > 100                  # exercises the M > 100 delete-priority regime


def test_binomial_subgrid(golden):
    """Binomial device source (bm_fit.h) on the CPU vs the oracle table: 5 cells x 5 folds."""
    g = golden.config3
    sel = [0, 101, 222, 317, 399]
    E, st, cnt = emul_lib.cv_grid(golden.BASISbinomial, golden.yBinomial, g["fold_id"], 5, g["alpha"][sel], g["lam"][sel],
                                  prior="binomial")
    assert np.abs(E - g["fold_err"][sel]).max() < 1e-12
    assert np.all(st == 0)


def test_binomial_epistasis_subgrid(golden, oracle):
    """Bf: the binomial device source with the NeFull.c rule set on the expanded design (CPU build) against the
    oracle, which keeps the reference's implicit pair columns and its association of the products -- two
    restatements that share no code.  BASISbinomial is {-1, 0, 1}-coded, so the products are exact either way."""
    X, y = golden.BASISbinomial[::2, :30][:200], golden.yBinomial[::2][:200]
    fid = AssignToFolds(X, 3)
    alpha, lam = BuildGrid(X, y, 3)
    sel = np.arange(0, 400, 27)
    Eo, co, rc = oracle.cv_grid(X, y, fid, 3, alpha[sel], lam[sel], prior="binomial", epis=True, n_threads=8)
    E, st, cnt = emul_lib.cv_grid(X, y, fid, 3, alpha[sel], lam[sel], prior="binomial", epis=True)
    assert rc == 0 and np.all(st == 0)
    assert np.abs(E - Eo).max() < 1e-8
    assert cnt[..., 2].sum() == co["n_add"] and cnt[..., 3].sum() == co["n_del"] and cnt[..., 4].sum() == co["n_reest"]
    assert cnt[..., 10].max() == co["m_max"] and co["m_max"] > 20


def test_epistasis_subgrid(golden):
    """Epistasis rule set (GmVariant epis=1) of the device source on the CPU vs the oracle."""
    g = golden.config4
    X = golden.BASIS[:200, :60]
    sel = [0, 7, 13, 20]
    E, st, cnt = emul_lib.cv_grid(X, g["y_scaled"], g["fold_id"], 5, g["alpha_scaled"][sel], g["lam_scaled"][sel], epis=True)
    ref = g["fold_err_scaled"][sel]
    ok = (st & 8) == 0
    assert ok.sum() >= 8                              # the rest hit the 2K = 120 capacity (oracle too)
    assert (np.abs(E - ref) / ref)[ok].max() < 1e-10
    sel2 = [0, 399]
    E2, st2, _ = emul_lib.cv_grid(X, golden.y[:200], g["fold_id"], 5, g["alpha"][sel2], g["lam"][sel2], epis=True)
    assert (np.abs(E2 - g["fold_err"][sel2]) / g["fold_err"][sel2]).max() < 1e-10


def test_lazy_gram_rows_match_full_matrix(golden, monkeypatch):
    """On-demand Gram-row pool (gm_row) vs the precomputed matrix: identical bits and counters; a
    pool smaller than a fit's active set flags the fit instead of returning a wrong score."""
    g = golden.config1
    X, y = golden.BASIS[:50, :100], golden.y[:50]
    sel = np.arange(0, 400, 25)
    E0, st0, c0 = emul_lib.cv_grid(X, y, g["fold_id"], 3, g["alpha"][sel], g["lam"][sel])
    monkeypatch.setenv("PAREBEN_EMUL_LAZY", "60")
    E1, st1, c1 = emul_lib.cv_grid(X, y, g["fold_id"], 3, g["alpha"][sel], g["lam"][sel])
    assert np.array_equal(E0, E1) and np.array_equal(st0, st1) and np.array_equal(c0, c1)
    monkeypatch.setenv("PAREBEN_EMUL_LAZY", "3")       # pool runs out: rows go to the fit's private rows
    E2, st2, c2 = emul_lib.cv_grid(X, y, g["fold_id"], 3, g["alpha"][sel], g["lam"][sel])
    assert np.array_equal(E0, E2) and np.array_equal(st0, st2) and np.array_equal(c0, c2)
    monkeypatch.setenv("PAREBEN_EMUL_LAZY", "3,2")     # ... and with too few of those the fit is flagged
    E3, st3, _ = emul_lib.cv_grid(X, y, g["fold_id"], 3, g["alpha"][sel], g["lam"][sel])
    starved = (st3 & 9) == 9                          # overflow + abort
    assert starved.any() and not starved.all()
    assert np.array_equal(E3[~starved], E0[~starved])
