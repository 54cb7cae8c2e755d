"""CPU tests of the host-side restatements of the R-level pieces (no GPU, no oracle)."""
import math
import numpy as np

from pareben_amd.rlang import RRandom, r_seq_by, r_sd
from pareben_amd.grid import BuildGrid, GetLambdaMax, AssignToFolds, summarise_cv


def test_r_rng_known_answers(golden):
    k = golden.known["rng"]
    r = RRandom(1)
    assert [round(r.unif_rand(), 7) for _ in range(3)] == k["runif3"]
    assert RRandom(1).sample(range(1, 11)) == k["sample10"]
    assert RRandom(1, "Rounding").sample(range(1, 11)) == k["sample10_rounding"]


def test_assign_to_folds_config1(golden):
    X = golden.BASIS[:50, :100]
    k = golden.known["config1"]
    assert "".join(map(str, AssignToFolds(X, 3))) == k["folds"]
    assert "".join(map(str, AssignToFolds(X, 3, sample_kind="Rounding"))) == k["folds_rounding"]
    f = AssignToFolds(X, 3)
    assert sorted(np.bincount(f)[1:].tolist()) == [16, 17, 17]
    # a full-length foldId is passed through (R/AssignToFolds.R:10)
    given = np.arange(50) % 3 + 1
    assert np.array_equal(AssignToFolds(X, 3, given), given)
    # divisible case: equal fold sizes
    assert np.bincount(AssignToFolds(golden.BASIS[:60], 3))[1:].tolist() == [20, 20, 20]


def test_seq_semantics():
    a = r_seq_by(1.0, 0.05, -0.05)
    assert len(a) == 20
    assert a[18] == 1.0 + 18 * (-0.05)          # 0.09999999999999998, not 0.1
    assert a[18] != 0.1


def test_build_grid_config1(golden):
    X, y = golden.BASIS[:50, :100], golden.y[:50]
    k = golden.known["config1"]
    alpha, lam = BuildGrid(X, y, 3)
    assert len(alpha) == 400 and len(lam) == 400
    assert lam[0] == k["lambda_first"] and abs(lam[-1] - k["lambda_last"]) < 1e-17
    # expand.grid: alpha fastest
    assert np.array_equal(alpha[:20], r_seq_by(1.0, 0.05, -0.05))
    assert np.all(lam[:20] == lam[0]) and lam[20] < lam[0]
    assert GetLambdaMax(X, y) * 10 == lam[0]


def test_build_grid_matches_real_r_grid_shape(golden):
    """The stored real-R output (rds_10000) pins grid order and the alpha values bit for bit."""
    r = golden.rds
    a = r["detail_alpha"][::3][:20]
    assert np.array_equal(a, r_seq_by(1.0, 0.05, -0.05))
    lam = r["detail_lambda"][::60]
    assert len(lam) == 20
    step = (math.log(lam[0]) - math.log(0.001 * lam[0])) / 19
    ours = np.exp(r_seq_by(math.log(lam[0]), math.log(0.001 * lam[0]), -step))
    assert np.allclose(ours, lam, rtol=1e-14, atol=0)
    assert abs(lam[0] - golden.known["yeast10000"]["lambda_max_x10"]) < 1e-13


def test_lambda_max_floor_and_epis():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((40, 6))
    y = rng.standard_normal(40) * 1e-3 + 5          # nearly constant, uncorrelated target
    lm = GetLambdaMax(X, y)
    assert lm >= math.log(1.1)
    # Epis pass correlates with the UN-normalised centred target (R/BuildGrid.R:26)
    y2 = X[:, 0] * X[:, 1] * 10 + rng.standard_normal(40)
    assert GetLambdaMax(X, y2, "yes") > GetLambdaMax(X, y2, "no")


def test_summary_matches_real_r_tables(golden):
    """mean / sd/sqrt(nFolds) / sort order / first-minimum arg-min, checked on the stored real-R
    Results.Detail -> Results.Summary pair."""
    r = golden.rds
    nF = 3
    alpha = r["detail_alpha"][::nF]
    lam = r["detail_lambda"][::nF]
    E = r["detail_MSE"].reshape(-1, nF)
    a_s, l_s, se, err, idx = summarise_cv(alpha, lam, E, nF)
    assert np.array_equal(a_s, r["summary_alpha"]) and np.array_equal(l_s, r["summary_lambda"])
    assert np.allclose(err, r["summary_MSE"], rtol=1e-14, atol=0)
    assert np.allclose(se, r["summary_SE"], rtol=1e-12, atol=0)
    assert l_s[idx] == r["lambda_optimal"][0] and a_s[idx] == r["alpha_optimal"][0]


def test_grid_folds_summary_on_the_other_real_r_tables(fulltest):
    """The same R-level pieces on the two other stored real-R tables (August 2018: first 13 248 columns of the 19 871-column
    design, rows 2..n; April/May 2018: the reassembled 5356-column Subset_Test design, all rows): BuildGrid's alpha /
    lambda values, and Results.Detail -> Results.Summary -> (lambda.optimal, alpha.optimal).  Both runs stored their
    Detail rows in another order than the May 2018 one (fold fastest within (alpha, lambda) blocks that are not sorted):
    the summary does not depend on it."""
    for name, cols in (("looser19871", 13248), ("subset5356", None)):
        X, y, d = fulltest(name)
        if cols:
            X = X[:, :cols]
        alpha, lam = BuildGrid(X, y, 3)
        assert np.allclose(np.unique(alpha), np.unique(d["detail_alpha"]), rtol=0, atol=1e-15)
        assert np.allclose(np.unique(lam), np.unique(d["detail_lambda"]), rtol=1e-13, atol=0)
        key = {(round(float(a_), 6), "%.6e" % l_, int(f_)): m_
               for f_, a_, l_, m_ in zip(d["detail_foldId"], d["detail_alpha"], d["detail_lambda"], d["detail_MSE"])}
        assert len(key) == 1200
        E = np.array([[key[(round(float(a_), 6), "%.6e" % l_, f + 1)] for f in range(3)] for a_, l_ in zip(alpha, lam)])
        a_s, l_s, se, err, idx = summarise_cv(alpha, lam, E, 3)
        assert np.allclose(a_s, d["summary_alpha"], rtol=0, atol=1e-15) and np.allclose(l_s, d["summary_lambda"], rtol=1e-13, atol=0)
        assert np.allclose(err, d["summary_MSE"], rtol=1e-14, atol=0) and np.allclose(se, d["summary_SE"], rtol=1e-11, atol=0)
        assert abs(l_s[idx] - float(d["lambda_optimal"])) <= 1e-14 * l_s[idx] and abs(a_s[idx] - float(d["alpha_optimal"])) < 1e-15
        fid = AssignToFolds(X, 3, sample_kind="Rounding")
        assert sorted(np.bincount(fid)[1:].tolist()) == sorted([X.shape[0] // 3 + (1 if k < X.shape[0] % 3 else 0) for k in range(3)])


def test_summary_binomial_intent():
    alpha = np.array([1.0, 0.5]); lam = np.array([0.3, 0.3])
    E = np.array([[-0.6, -0.7], [-0.4, -0.5]])
    a_s, l_s, se, err, idx = summarise_cv(alpha, lam, E, 2, prior="binomial")
    assert np.allclose(err, [0.45, 0.65]) and a_s[idx] == 0.5
    assert abs(r_sd([1.0, 2.0, 4.0]) - 1.5275252316519468) < 1e-15


def test_local_search_policy_matches_real_r_output(fulltest):
    """The host policy of search = "local" (pareben_amd/local.py, R/LocalSearch.R:52-134) against a real-R output:
    Full_Test/parEBENoutput_epi0.08_residual_cv3local*.RDS = CrossValidate(search = "local", nFolds = 3) on
    filter_matrix_epi0.08 + pheno_Zeo_residual.  Its fold errors cannot be recomputed (that run drew its folds with an
    unseeded sample(), R/LocalSearch.R:13-20), but the policy can be replayed over R's own per-cell (mean, SE): same
    lambda grid, same walk (all 400 cells: no early stop fires with SEs of ~10 on differences of ~1), same fullCV rows in
    the same order, same per-alpha minima table and the same (alpha.optimal, lambda.optimal) = (0.05, 0.6625895...)."""
    from pareben_amd.local import replay_local_search
    X, y, d = fulltest("epi008")
    C, F = d["local_CrossValidation"], d["local_fullCV"]
    alpha, lam = BuildGrid(X, y, 3)
    a_desc, l_desc = np.unique(alpha)[::-1], np.unique(lam)[::-1]
    assert np.allclose(a_desc, F[::20, 0], rtol=0, atol=1e-15) and np.allclose(l_desc, F[:20, 1], rtol=1e-13, atol=0)
    mean = F[:, 2].reshape(20, 20); se = F[:, 3].reshape(20, 20)

    def folds_of(ia, il):                      # three fold errors with exactly this mean and sd/sqrt(3)
        c = se[ia, il] * math.sqrt(3.0)
        return [mean[ia, il] - c, mean[ia, il], mean[ia, il] + c]
    each, a_opt, l_opt, msecv, visited = replay_local_search(a_desc, l_desc, folds_of)
    assert len(visited) == 400
    assert np.allclose(msecv[:, 2:], F[:, 2:], rtol=1e-12, atol=0) and np.allclose(msecv[:, :2], F[:, :2], rtol=1e-13, atol=1e-15)
    assert np.allclose(each, C, rtol=1e-12, atol=1e-15)
    assert abs(a_opt - float(d["local_alpha_optimal"])) < 1e-15 and abs(l_opt - float(d["local_lambda_optimal"])) <= 1e-13 * l_opt
    # and an early stop, which this table never takes: raise one cell of alpha = 1 by more than its predecessor's SE
    mean2 = mean.copy(); mean2[0, 3] = mean[0, :3].min() + se[0, int(np.argmin(mean[0, :3]))] + 1.0

    def folds2(ia, il):
        c = se[ia, il] * math.sqrt(3.0)
        return [mean2[ia, il] - c, mean2[ia, il], mean2[ia, il] + c]
    each2, _, _, msecv2, visited2 = replay_local_search(a_desc, l_desc, folds2)
    assert [v for v in visited2 if v[0] == 0] == [(0, 0), (0, 1), (0, 2), (0, 3)] and len(visited2) == 400 - 16
    assert np.all(msecv2[len(visited2):] == 0) and each2[0, 1] == l_desc[int(np.argmin(mean2[0, :4]))]
