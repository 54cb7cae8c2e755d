"""PAREBEN_STRICT_ORDER=1 (pareben_amd/csrc/gm_strict.h): the Gaussian main-effect fit in the reference's own formulation and
operation order -- separate multiply and add, netlib ddot / dgemv loop order, dpotf2 + dtrti2 + dlauu2, the per-fit BASIS_PHI
cache, the reference's visiting order for arg-max ties, a correctly rounded logarithm -- parallel only across independent outputs.

What it shows: on the long add/delete trajectories of the stored real-R tables (alpha = 1, duplicated genotype columns) the
action taken is decided again and again by dML margins of 0 ... 4e-15 (tools/first_divergence.py), i.e. by the summation order of
everything upstream.  In strict mode the GPU follows the netlib-order CPU oracle bit for bit through thousands of such
decisions (fold SSEs equal to the last bit on fits with up to 5082 inner iterations and 1217 active columns): the
deviations of the production path are order-only.  Real R's own build followed a THIRD order: of the 36 fits of the Subset_Test
cells that hold a listed deviating pair, netlib order lands on R's value (1e-9) in 21 -- among them all three folds of the
optimum cell, so cv.error at the optimum meets the north-star 1e-6 in strict mode (1e-13; default mode 3.3e-5) -- the
production order in 20, and in 4 of them the production order follows R where netlib order does not
(tests/golden/subset5356_strict_cells_oracle.json, tools/make_strict_cells_oracle.py)."""
import json
import os

import numpy as np
import pytest

import pareben_amd
from pareben_amd.grid import AssignToFolds, BuildGrid, summarise_cv

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_strict_mode_is_bit_identical_to_the_oracle(golden, oracle, monkeypatch):
    """Fold SSEs, status words and event counts of a CV sub-grid in strict mode against the netlib-order CPU oracle:
    equal to the last bit (the device's FP64 division and square root round like the host's: tools/ubench/libm_bits.hip)."""
    monkeypatch.setenv("PAREBEN_STRICT_ORDER", "1")
    X, y = golden.BASIS[:300, :200], golden.y[:300]
    y = (y - y.mean()) / y.std()
    fid = AssignToFolds(X, 3)
    alpha, lam = BuildGrid(X, y, 3)
    sel = np.array([0, 140, 205, 260, 330, 399])
    with pareben_amd.Context(X, y, fid, 3) as ctx:
        E, st, cnt = ctx.run(alpha[sel], lam[sel])
    Eo, co, rc = oracle.cv_grid(X, y, fid, 3, alpha[sel], lam[sel], n_threads=6)
    assert rc == 0 and np.all(st & 8 == 0)
    assert np.array_equal(E, Eo), np.abs(E - Eo).max()
    for k, name in ((2, "n_add"), (3, "n_del"), (4, "n_reest"), (5, "n_fullstat"), (1, "n_inner")):
        assert cnt[..., k].sum() == co[name], name
    assert cnt[..., 10].max() >= 20


def test_strict_mode_follows_the_oracle_bit_for_bit_on_the_subset_table(fulltest, monkeypatch):
    """Subset_Test table (R 3.5.0 + CRAN EBEN, K = 5356; tests/golden/subset5356.npz), the optimum cell (120) and cell 20, six
    fits of 1419 ... 5082 inner iterations with up to 1217 active columns, in strict mode:
    * every fold SSE equals the netlib-order oracle's (committed fixture, 3 CPU-hours) to the last bit;
    * the optimum cell: all three folds within 1e-9 of R's Results.Detail$MSE and cv.error within the north-star 1e-6 of R's
      (the default mode misses it by 3.3e-5: fold 2 is a listed pair);
    * cell 20: two of the three fits end where the oracle ends and NOT where R ends (9.9e-7, 3.3e-5) -- R's build summed in a
      third order; what strict mode demonstrates is order-only divergence, not R's order."""
    X, y, d = fulltest("subset5356")
    fx = json.load(open(os.path.join(GOLDEN, "subset5356_strict_cells_oracle.json")))
    ora = {(r["cell"], r["fold"]): r for r in fx["fits"]}
    fid = AssignToFolds(X, 3, sample_kind="Rounding")
    alpha, lam = BuildGrid(X, y, 3)
    key = {(round(float(a_), 6), "%.6e" % l_, int(f_)): m_
           for f_, a_, l_, m_ in zip(d["detail_foldId"], d["detail_alpha"], d["detail_lambda"], d["detail_MSE"])}
    want = np.array([[key[(round(float(a_), 6), "%.6e" % l_, f + 1)] for f in range(3)] for a_, l_ in zip(alpha, lam)])
    a_s, l_s, se, cv, idx = summarise_cv(alpha, lam, want, 3)
    opt = int(np.nonzero((alpha == a_s[idx]) & (lam == l_s[idx]))[0][0])
    assert a_s[idx] == float(d["alpha_optimal"]) and opt == 120
    cells = [20, opt]
    monkeypatch.setenv("PAREBEN_STRICT_ORDER", "1")
    with pareben_amd.Context(X, y, fid, 3) as ctx:
        E, st, cnt = ctx.run(alpha[cells], lam[cells])
    assert np.all(st & 9 == 0)
    Eo = np.array([[ora[(c, f + 1)]["oracle_sse"] for f in range(3)] for c in cells])
    assert np.array_equal(E, Eo), (E - Eo).tolist()
    for k, c in enumerate(cells):
        for f in range(3):
            assert cnt[k, f, 1] == ora[(c, f + 1)]["counters"]["n_inner"] and cnt[k, f, 10] == ora[(c, f + 1)]["counters"]["m_max"]
    rel = np.abs(E - want[cells]) / want[cells]
    assert rel[1].max() < 1e-9, rel[1]
    assert abs(E[1].mean() - float(d["summary_MSE"][idx])) <= 1e-6 * float(d["summary_MSE"][idx])
    assert rel[0, 0] > 5e-7 and rel[0, 1] > 1e-5 and rel[0, 2] < 1e-9, rel[0]
