"""EBelasticNet.Gaussian / EBelasticNet.Binomial: the refit users run at
(alpha*, lambda*) after CrossValidate, with the reference's M x 6 weight table.
CPU part: the host-side table logic and Student-t CDF.  GPU part: against the oracle's per-fit
outputs through the C ABI (pareben_fit_gaussian / _gaussian_epis / _binomial)."""
import numpy as np
import pytest
from scipy import stats

import pareben_amd
from pareben_amd import eben
from pareben_amd.grid import BuildGrid


def test_pt_matches_scipy():
    for df in (1, 2, 5, 32, 99, 100, 101, 399, 799, 3802, 100000):
        for t in (0.0, 1e-3, 0.5, 1.0, 2.5, 7.0, 20.0, 60.0, -3.0, -0.25):
            assert abs(eben.pt(t, df) - stats.t.cdf(t, df)) < 2e-14, (t, df)
    assert np.isnan(eben.pt(float("nan"), 5))


def test_weight_table_layout():
    """main effects first ordered by locus, then pairs ordered by first locus; 5th column dropped;
    an empty model gives one all-zero row (EBelasticNet.Gaussian.R:55-83)."""
    Beta = np.zeros((10, 5))
    Beta[:, 0] = [1, 2, 3, 4, 1, 1, 1, 2, 2, 3]
    Beta[:, 1] = [1, 2, 3, 4, 2, 3, 4, 3, 4, 4]
    Beta[[2, 0, 8, 5], 2] = [0.5, -1.5, 0.25, 2.0]
    Beta[[2, 0, 8, 5], 3] = [0.04, 0.09, 0.01, 0.16]
    Beta[[2, 0, 8, 5], 4] = [3, 1, 9, 6]
    W = eben._weight_table(Beta, 4, 50, True)
    assert W.shape == (4, 6)
    assert W[:, 0].tolist() == [1, 3, 1, 2] and W[:, 1].tolist() == [1, 3, 3, 4]
    assert np.allclose(W[:, 4], np.abs(W[:, 2]) / np.sqrt(W[:, 3]))
    assert np.allclose(W[:, 5], 2 * stats.t.sf(W[:, 4], 49), rtol=1e-10, atol=1e-15)
    E = eben._weight_table(np.zeros((6, 4)), 2, 20, False)
    assert E.shape == (1, 6) and np.all(E[0, :5] == 0) and E[0, 5] == 1.0


def _check_weight(W, oBeta, keep_col, n):
    keep = np.nonzero(oBeta[:, keep_col] != 0)[0]
    assert W.shape == (max(len(keep), 1), 6)
    if len(keep) == 0:
        return
    key = lambda M: np.lexsort((M[:, 1], M[:, 0]))
    Wk, Ok = W[key(W)], oBeta[keep][key(oBeta[keep])]
    assert np.array_equal(Wk[:, :2], Ok[:, :2])
    assert np.allclose(Wk[:, 2], Ok[:, 2], rtol=1e-8, atol=0) and np.allclose(Wk[:, 3], Ok[:, 3], rtol=1e-8, atol=0)
    t = np.abs(Ok[:, 2]) / (np.sqrt(Ok[:, 3]) + 1e-20)
    assert np.allclose(Wk[:, 4], t, rtol=1e-8)
    assert np.allclose(Wk[:, 5], 2 * (1 - stats.t.cdf(t, n - 1)), rtol=1e-6, atol=1e-13)


@pytest.mark.gpu
def test_gaussian_refit_vs_oracle(golden, oracle):
    X, y = golden.BASIS[:200, :150], golden.y[:200]
    alpha, lam = BuildGrid(X, y, 5)
    for c in (140, 399):
        r = pareben_amd.EBelasticNet.Gaussian(X, y, lam[c], alpha[c])
        o = oracle.fit_gaussian(X, y, lam[c], alpha[c])
        _check_weight(r["weight"], o["Beta"], 2, 200)
        assert abs(r["WaldScore"] - o["wald"]) < 1e-8 * abs(o["wald"])
        assert abs(r["Intercept"] - o["intercept"]) < 1e-9 * abs(o["intercept"])
        assert abs(r["residVar"] - o["residual"]) < 1e-9 * o["residual"]
        assert r["lambda"] == lam[c] and r["alpha"] == alpha[c]
    r = pareben_amd.EBelasticNet.Gaussian(X, y, lam[0], alpha[0])        # 10 * lambda_max: a <= 2-feature model
    assert r["weight"].shape[0] <= 2 and r["weight"].shape[1] == 6


@pytest.mark.gpu
def test_gaussian_epistasis_refit_vs_oracle(golden, oracle):
    g = golden.config4
    X, y = golden.BASIS[:200, :60], g["y_scaled"]
    for c in (12, 20):
        lam, al = g["lam_scaled"][c], g["alpha_scaled"][c]
        r = pareben_amd.EBelasticNet.Gaussian(X, y, lam, al, Epis="yes")
        o = oracle.fit_gaussian(X, y, lam, al, epis=True)
        assert o["rc"] == 0
        _check_weight(r["weight"], o["Beta"], 4, 200)
        W = r["weight"]
        nmain = int((W[:, 0] == W[:, 1]).sum())
        assert np.all(W[:nmain, 0] == W[:nmain, 1]) and np.all(W[nmain:, 0] != W[nmain:, 1])
        assert np.all(np.diff(W[:nmain, 0]) > 0) and np.all(np.diff(W[nmain:, 0]) >= 0)
        assert abs(r["WaldScore"] - o["wald"]) < 1e-8 * abs(o["wald"])
        assert abs(r["Intercept"] - o["intercept"]) < 1e-9 * max(abs(o["intercept"]), 1e-3)
        assert abs(r["residVar"] - o["residual"]) < 1e-9 * o["residual"]
    raw = pareben_amd.fit_gaussian(X, y, g["lam_scaled"][12], g["alpha_scaled"][12], epis=True)
    o = oracle.fit_gaussian(X, y, g["lam_scaled"][12], g["alpha_scaled"][12], epis=True)
    assert np.array_equal(raw["Beta"][:, :2], o["Beta"][:, :2]) and np.array_equal(raw["Beta"][:, 4], o["Beta"][:, 4])


@pytest.mark.gpu
def test_binomial_refit_vs_oracle(golden, oracle):
    X, y = golden.BASISbinomial[:300, :200], golden.yBinomial[:300]
    alpha, lam = BuildGrid(X, y, 5)
    for c in (150, 330):
        r = pareben_amd.EBelasticNet.Binomial(X, y, lam[c], alpha[c])
        o = oracle.fit_binomial(X, y, lam[c], alpha[c])
        assert o["rc"] == 0
        _check_weight(r["weight"], o["Beta"], 2, 300)
        assert abs(r["logLikelihood"] - o["loglik"]) < 1e-8 * abs(o["loglik"])
        assert abs(r["WaldScore"] - o["wald"]) < 1e-7 * abs(o["wald"])
        assert np.allclose(r["Intercept"], o["intercept"], rtol=1e-8)


@pytest.mark.gpu
def test_binomial_epistasis_refit_vs_oracle(golden, oracle):
    """EBelasticNet.Binomial(Epis = "yes"): pareben_fit_binomial_epis' raw table (2K x 4, used bases in model order,
    ElasticNetBinaryNeFull.c:154-211) and the weight table of EBelasticNet.Binomial.R:47-78 (mains then pairs, each
    ordered by locus1, t and p columns) against the oracle."""
    X, y = golden.BASISbinomial[::2, :30][:200], golden.yBinomial[::2][:200]
    alpha, lam = BuildGrid(X, y, 3)
    for c in (150, 399):
        raw = pareben_amd.fit_binomial(X, y, lam[c], alpha[c], epis=True)
        o = oracle.fit_binomial(X, y, lam[c], alpha[c], epis=True)
        m = o["counters"]["m_final"]
        assert o["rc"] == 0 and raw["counters"]["m_final"] == m
        assert np.array_equal(raw["Beta"][:, :2], o["Beta"][:, :2]) and np.all(raw["Beta"][m:] == 0)
        assert np.allclose(raw["Beta"][:m, 2:], o["Beta"][:m, 2:], rtol=1e-7, atol=0)
        assert abs(raw["logLikelihood"] - o["loglik"]) < 1e-8 * abs(o["loglik"])
        assert abs(raw["wald"] - o["wald"]) < 1e-7 * abs(o["wald"])
        assert np.allclose(raw["intercept"], o["intercept"], rtol=1e-8)
        r = pareben_amd.EBelasticNet.Binomial(X, y, lam[c], alpha[c], Epis="yes")
        w = r["weight"]
        assert w.shape == (m, 6)
        main = w[w[:, 0] == w[:, 1]]; pair = w[w[:, 0] != w[:, 1]]
        assert np.array_equal(w, np.vstack([main, pair])) and np.all(np.diff(main[:, 0]) >= 0) and np.all(np.diff(pair[:, 0]) >= 0)
        assert np.allclose(w[:, 4], np.abs(w[:, 2]) / (np.sqrt(w[:, 3]) + 1e-20))


@pytest.mark.gpu
def test_refit_helper_workgroups_bit_identical(fulltest, monkeypatch):
    """A single Gaussian fit is one workgroup; the other workgroups of `gm_fit_kernel`'s launch take chunks of its
    full-stat passes and action sweeps through the CV kernel's job board (3843 x 11 597: 2.3 s alone, 0.9 s with
    help; 3802 x 19 871: 18.5 -> 4.2 s, profiles/r02/refit_time.txt).  Same bits either way."""
    X, y, d = fulltest("zeo_main_epi")
    lam, al = float(d["lambda"]), float(d["alpha"])
    helped = pareben_amd.fit_gaussian(X, y, lam, al)
    monkeypatch.setenv("PAREBEN_SHARE", "0")
    alone = pareben_amd.fit_gaussian(X, y, lam, al)
    assert np.array_equal(helped["Beta"], alone["Beta"])
    assert helped["wald"] == alone["wald"] and helped["intercept"] == alone["intercept"] and helped["residual"] == alone["residual"]
    assert helped["counters"] == alone["counters"] and helped["counters"]["n_inner"] > 1000


@pytest.mark.gpu
def test_reference_dot_c_symbols(golden, capfd):
    """The reference's own .C entry points (elasticNetLinearNeMainEff.c:55-57, elasticNetLinearNeFull2.c:57-58,
    ElasticNetBinaryNEmainEff.c:236-238, ElasticNetBinaryNeFull.c:52-55) called the way R's .C does -- every argument by
    pointer, outputs in place, no return value -- give bit for bit what the pareben_fit_* entries give, and `verbose`
    prints the reference's lines (:70-71, :196, :205)."""
    from pareben_amd import _lib
    X, y = golden.BASIS[:200, :150], golden.y[:200]
    alpha, lam = BuildGrid(X, y, 5)
    ys = (y - y.mean()) / y.std()                      # unit-scale target: models of tens of features (Q9: the raw one keeps <= 1)
    alpha, lam = BuildGrid(X, ys, 5)
    y = ys
    a = _lib.dot_c("elasticNetLinearNeMainEff", X, y, lam[140], alpha[140])
    b = _lib.fit_gaussian(X, y, lam[140], alpha[140])
    assert np.array_equal(a["Beta"], b["Beta"]) and (a["wald"], a["intercept"], a["residual"]) == (b["wald"], b["intercept"], b["residual"])
    assert np.count_nonzero(a["Beta"][:, 2]) >= 1
    Xe, ye = golden.BASIS[:200, :30], golden.y[:200]
    ae, le = BuildGrid(Xe, ye, 5, Epis="yes")
    a = _lib.dot_c("elasticNetLinearNeEpisEff", Xe, ye, le[150], ae[150])
    b = _lib.fit_gaussian(Xe, ye, le[150], ae[150], epis=True)
    assert a["Beta"].shape == (465, 5) and np.array_equal(a["Beta"], b["Beta"]) and a["wald"] == b["wald"]
    Xb, yb = golden.BASISbinomial[:, :120], golden.yBinomial
    ab, lb = BuildGrid(Xb, yb, 5)
    a = _lib.dot_c("ElasticNetBinaryNEmainEff", Xb, yb, lb[250], ab[250])
    b = _lib.fit_binomial(Xb, yb, lb[250], ab[250])
    assert np.array_equal(a["Beta"], b["Beta"]) and a["logLikelihood"] == b["logLikelihood"] and np.array_equal(a["intercept"], b["intercept"])
    a = _lib.dot_c("ElasticNetBinaryNEfull", Xb[:, :20], yb, lb[250], ab[250])
    b = _lib.fit_binomial(Xb[:, :20], yb, lb[250], ab[250], epis=True)
    assert a["Beta"].shape == (40, 4) and np.array_equal(a["Beta"], b["Beta"]) and a["logLikelihood"] == b["logLikelihood"]
    capfd.readouterr()
    r = _lib.dot_c("elasticNetLinearNeMainEff", X, y, lam[140], alpha[140], verbose=5)
    out = capfd.readouterr().out
    n_eff = int(np.count_nonzero(r["Beta"][:, 2]))
    assert "basisMax: 150" in out and "start EB-elasticNet with alpha:" in out and "outer loop starts" in out
    assert "Iteration number: 1, err:" in out and "sigma0:" in out and ("EBEN Finished, number of basis: %d" % n_eff) in out
    assert "\t inner loop 1; number of basis:" in out
    _lib.dot_c("ElasticNetBinaryNEmainEff", Xb, yb, lb[250], ab[250], verbose=3)
    out = capfd.readouterr().out
    assert "Empirical Bayesian Elastic Net outer loop starts" in out and "Iteration number: 1, err:" in out and "EBEN Finished" in out
