"""GPU parity tests: the HIP path, called through the C ABI (pareben_amd._lib -> libpareben_hip.so),
against the oracle on the same inputs and against the committed golden fixtures.

Bar (BASELINE.json north_star): cv.error within 1e-6 (relative), selected (alpha, lambda) exact.
Tolerances used here are tighter where the data allow: fold SSE 1e-9 relative, identical event
counters (= identical add/delete/re-estimate sequence) on the bundled data."""
import os

import numpy as np
import pytest

import pareben_amd
from pareben_amd.grid import BuildGrid, AssignToFolds, summarise_cv
from conftest import synthetic_gaussian

pytestmark = pytest.mark.gpu

REL_FOLD = 1e-9
REL_CV = 1e-6


def _rel(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)


def test_library_loaded_and_gpu_visible():
    L = pareben_amd.load_library()
    assert L.pareben_device_count() >= 1


def test_config1_full_grid_vs_golden(golden):
    """Reference's own test case (tests/CrossValidate-test.R): BASIS[1:50,1:100], nFolds=3."""
    X, y = golden.BASIS[:50, :100], golden.y[:50]
    out = pareben_amd.CrossValidate(X, y, nFolds=3, Epis="no", prior="gaussian", search="global", return_stats=True)
    g, k = golden.config1, golden.known["config1"]
    D = out["Results.Detail"]
    E = np.asarray(D["MSE"]).reshape(400, 3)
    assert np.array_equal(np.asarray(D["alpha"])[::3], g["alpha"]) and np.array_equal(np.asarray(D["lambda"])[::3], g["lam"])
    assert np.array_equal(np.asarray(D["foldId"])[:3], [1, 2, 3])
    assert _rel(E, g["fold_err"]).max() < REL_FOLD
    assert out["alpha.optimal"] == k["alpha_opt"] and out["lambda.optimal"] == k["lambda_opt"]
    S = out["Results.Summary"]
    i = int(np.argmin(np.asarray(S["MSE"])))
    assert abs(np.asarray(S["MSE"])[i] - k["cv_error"]) < REL_CV * k["cv_error"]
    assert abs(np.asarray(S["SE"])[i] - k["SE"]) < REL_CV * k["SE"]
    st = out["stats"]["status"]
    assert set(np.unique(st)) <= {0, 4} and (st == 4).sum() == 11
    cnt = out["stats"]["counters"].sum(axis=(0, 1))
    want = dict(zip(g["counter_names"], g["counters"]))
    for j, n in enumerate(("n_outer", "n_inner", "n_add", "n_del", "n_reest", "n_fullstat")):
        assert cnt[j] == want[n], n


def test_basis481_subgrid_vs_golden(golden):
    g = golden.basis481
    with pareben_amd.Context(golden.BASIS, golden.y, g["fold_id"], 5) as ctx:
        E, st, cnt = ctx.run(g["alpha"], g["lam"])
    assert _rel(E, g["fold_err"]).max() < REL_FOLD
    want = dict(zip(g["counter_names"], g["counters"]))
    tot = cnt.sum(axis=(0, 1))
    for j, n in enumerate(("n_outer", "n_inner", "n_add", "n_del", "n_reest", "n_fullstat")):
        assert tot[j] == want[n], n


def test_per_fit_entry_vs_oracle(golden, oracle):
    X, y = golden.BASIS[:200, :150], golden.y[:200]
    alpha, lam = BuildGrid(X, y, 5)
    for c in (0, 210, 399):
        r = pareben_amd.fit_gaussian(X, y, lam[c], alpha[c])
        o = oracle.fit_gaussian(X, y, lam[c], alpha[c])
        nz = np.nonzero(o["Beta"][:, 2])[0]
        assert np.array_equal(np.nonzero(r["Beta"][:, 2])[0], nz)
        assert np.array_equal(r["Beta"][:, :2], o["Beta"][:, :2])
        assert _rel(r["Beta"][nz, 2], o["Beta"][nz, 2]).max() < 1e-8
        assert _rel(r["Beta"][nz, 3], o["Beta"][nz, 3]).max() < 1e-8
        assert abs(r["intercept"] - o["intercept"]) < 1e-9 * abs(o["intercept"])
        assert abs(r["residual"] - o["residual"]) < 1e-9 * o["residual"]


def test_synthetic_vs_oracle(oracle):
    """Ragged sizes (n not a multiple of nFolds, p not a multiple of 64) and N(0,1) designs."""
    X, y = synthetic_gaussian(157, 333, n_causal=8, seed=5)
    fid = AssignToFolds(X, 4)
    alpha, lam = BuildGrid(X, y, 4)
    sel = np.arange(0, 400, 7)
    with pareben_amd.Context(X, y, fid, 4) as ctx:
        E, st, cnt = ctx.run(alpha[sel], lam[sel])
    Eo, co, rc = oracle.cv_grid(X, y, fid, 4, alpha[sel], lam[sel])
    assert rc == 0 and np.all(st & 8 == 0)          # 4 = the reference's stale-slot path, legitimate
    assert _rel(E, Eo).max() < 1e-8
    assert cnt[..., 2].sum() == co["n_add"] and cnt[..., 3].sum() == co["n_del"]


def test_rerun_is_bit_identical_and_order_free(golden):
    """Deterministic reductions: the same cells give bit-identical scores whatever else is in
    the launch and in whatever order they are submitted (what makes 1/2/4/8-GPU runs agree)."""
    X, y = golden.BASIS[:120, :90], golden.y[:120]
    fid = AssignToFolds(X, 3)
    alpha, lam = BuildGrid(X, y, 3)
    with pareben_amd.Context(X, y, fid, 3) as ctx:
        E1, _, _ = ctx.run(alpha, lam)
        E2, _, _ = ctx.run(alpha, lam)
        perm = np.random.default_rng(0).permutation(400)[:57]
        E3, _, _ = ctx.run(alpha[perm], lam[perm])
    assert np.array_equal(E1, E2)
    assert np.array_equal(E3, E1[perm])


def test_gram_kernel_vs_numpy(golden):
    """gram_kernel (FP64 matrix cores) against the reference's own operation order, restated in numpy:
    BASIS_PHI[u][i] = (sum over samples, in order, of x_i[h] * (x_u[h] / |x_u|)) / |x_i|
    (elasticNetLinearNeMainEff.c:1171-1177, :1608-1630).  For integer-coded designs (BASIS is {-1, 0, 1}) every
    product is exact, so the fma chain of the matrix cores must reproduce that sequential sum bit for bit.
    Gaussian design with ragged sizes (K not a multiple of the 128-block, N not a multiple of the 16-sample
    slab): to rounding."""
    X = golden.BASIS[:, :481]
    g = golden.basis481
    fid = g["fold_id"]
    with pareben_amd.Context(X, golden.y, fid, 5) as ctx:
        for f in (0, 4):
            Gd = ctx.gram(f)
            Xt = X[fid != f + 1]
            q = np.sum(Xt * Xt, axis=0); q[q == 0] = 1.0
            sc = np.sqrt(q)
            phi = Xt / sc[None, :]
            acc = np.zeros((481, 481))
            for h in range(Xt.shape[0]):                             # sequential over samples, like the reference's ddot
                acc += phi[h][:, None] * Xt[h][None, :]
            want = acc / sc[None, :]
            assert np.array_equal(Gd, want)
    rng = np.random.default_rng(2)
    n, p = 203, 333
    Xg = np.asfortranarray(rng.standard_normal((n, p)))
    Xg[:, 7] = 0.0
    fid = AssignToFolds(Xg, 3)
    with pareben_amd.Context(Xg, rng.standard_normal(n), fid, 3) as ctx:
        Gd = ctx.gram(1)
    Xt = Xg[fid != 2]
    q = np.sum(Xt * Xt, axis=0); q[q == 0] = 1.0
    sc = np.sqrt(q)
    want = (Xt.T @ Xt) / sc[:, None] / sc[None, :]
    assert np.abs(Gd - want).max() < 1e-13
    assert np.all(Gd[7] == 0) and np.all(Gd[:, 7] == 0)


def test_poisoned_workspace_is_bit_identical(golden, monkeypatch):
    """Every fit is self-contained: nothing a previous fit of the same workgroup (or hipMalloc) left in
    the workspace may reach a result.  PAREBEN_WS_POISON=1 fills the fit workspaces with 0xFF bytes (NaN
    doubles, -1 ints) instead of zeros; a NaN just outside the active block would turn every S_in / Q_in of
    the matrix-core full-stat pass into NaN if the ragged 16-blocks were not masked.  Gaussian (active sets
    up to ~240, i.e. many ragged block shapes), epistasis and binomial workspaces."""
    g = golden.basis481
    g4 = golden.config4
    Xb, yb = golden.BASISbinomial[:300, :120], golden.yBinomial[:300]
    fb = AssignToFolds(Xb, 3)
    ab, lb = BuildGrid(Xb, yb, 3)
    selb = np.arange(0, 400, 13)

    def run_all():
        with pareben_amd.Context(golden.BASIS, golden.y, g["fold_id"], 5) as ctx:
            r1 = ctx.run(g["alpha"], g["lam"])
            r1b = ctx.run(g["alpha"][::-1].copy(), g["lam"][::-1].copy())      # second launch on the used workspace
        with pareben_amd.Context(golden.BASIS[:200, :60], g4["y_scaled"], g4["fold_id"], 5, epis=True) as ctx:
            r2 = ctx.run(g4["alpha_scaled"], g4["lam_scaled"])
        with pareben_amd.Context(Xb, yb, fb, 3, prior="binomial") as ctx:
            r3 = ctx.run(ab[selb], lb[selb])
        return r1, r1b, r2, r3

    clean = run_all()
    monkeypatch.setenv("PAREBEN_WS_POISON", "1")
    dirty = run_all()
    for (Ec, sc, cc), (Ed, sd, cd) in zip(clean, dirty):
        assert np.array_equal(sc, sd)
        assert np.array_equal(Ec, Ed, equal_nan=True)
        assert np.array_equal(cc, cd)
    assert np.array_equal(clean[0][0], clean[1][0][::-1])
    assert _rel(clean[0][0], g["fold_err"]).max() < REL_FOLD


def test_edge_cases(oracle):
    rng = np.random.default_rng(11)
    X = np.asfortranarray(rng.standard_normal((41, 17)))
    y = X[:, 3] * 1.5 + rng.standard_normal(41) * 0.2
    X[:, 9] = 0                      # all-zero column: scale 1, never selected
    X[:, 12] = X[:, 3]               # exact duplicate column
    fid = AssignToFolds(X, 2)
    alpha = np.array([1.0, 0.5, 0.05, 1.0]); lam = np.array([0.2, 0.05, 0.01, 1e6])
    with pareben_amd.Context(X, y, fid, 2) as ctx:
        E, st, cnt = ctx.run(alpha, lam)
    Eo, co, rc = oracle.cv_grid(X, y, fid, 2, alpha, lam)
    assert np.all(st & 8 == 0)
    assert _rel(E, Eo).max() < 1e-8
    assert cnt[3, :, 2].sum() == 0   # lambda = 1e6: nothing added
    # argument errors surface as exceptions, not crashes
    with pytest.raises(pareben_amd.ParebenError):
        pareben_amd.Context(X, y, np.zeros(41, dtype=np.int32), 2)
    with pytest.raises(ValueError):
        pareben_amd.Context(X, y, fid, 2, prior="poisson")


@pytest.mark.parametrize("n,p,nf", [(1000, 2000, 5)])
def test_larger_synthetic_properties(n, p, nf, oracle):
    """At a size the oracle cannot sweep in seconds: size-independent properties, plus a spot
    check of three cells against the oracle."""
    X, y = synthetic_gaussian(n, p, n_causal=20, seed=20251004)
    fid = AssignToFolds(X, nf)
    alpha, lam = BuildGrid(X, y, nf)
    sel = np.concatenate([np.arange(0, 400, 40), [399]])
    with pareben_amd.Context(X, y, fid, nf) as ctx:
        E, st, cnt = ctx.run(alpha[sel], lam[sel])
        # shifting the target shifts only the intercept: fold SSEs are invariant
        pass
    assert np.all(st & 8 == 0) and np.all(np.isfinite(E)) and np.all(E > 0)
    # lambda = 10*lambda_max: a one- or two-feature model -> SSE at most the null model's (+1%)
    null = np.array([np.sum((y[fid == f + 1] - y[fid != f + 1].mean()) ** 2) for f in range(nf)])
    assert np.all(E[0] < 1.01 * null) and np.all(E[0] > 0.5 * null)
    # informative cells beat the null model
    assert E[5:].mean(axis=1).min() < 0.5 * null.mean()
    spot = [0, 5, 9]
    Eo, co, rc = oracle.cv_grid(X, y, fid, nf, alpha[sel][spot], lam[sel][spot], n_threads=8)
    assert _rel(E[spot], Eo).max() < 1e-7
    with pareben_amd.Context(X, y + 3.0, fid, nf) as ctx:
        Es, _, _ = ctx.run(alpha[sel][spot], lam[sel][spot])
    assert _rel(Es, E[spot]).max() < 1e-6


def test_binomial_config3_full_grid_vs_golden(golden):
    """BASELINE config 3: yBinomial / BASISbinomial, binomial prior, nFolds=5, 20 x 20 grid =
    2000 fits, against the oracle table (itself equal to the survey's compiled-reference numbers)."""
    g, k = golden.config3, golden.known["config3"]
    out = pareben_amd.CrossValidate(golden.BASISbinomial, golden.yBinomial, nFolds=5, Epis="no", prior="binomial",
                                    search="global", return_stats=True)
    D = out["Results.Detail"]
    E = np.asarray(D["logL"]).reshape(400, 5)
    assert np.all(out["stats"]["status"] & 8 == 0)
    assert np.abs(E - g["fold_err"]).max() < 1e-8                      # mean log-likelihoods, O(0.3)
    assert out["alpha.optimal"] == k["alpha_opt"] and out["lambda.optimal"] == g["summary_lambda"][int(g["idx"])]
    S = out["Results.Summary"]
    i = int(np.argmin(np.asarray(S["Likelihood"])))
    assert abs(np.asarray(S["Likelihood"])[i] - k["likelihood"]) < 1e-6 * k["likelihood"]
    assert abs(np.asarray(S["SE"])[i] - k["SE"]) < 1e-6 * k["SE"]


def test_binomial_launch_shapes(golden, monkeypatch):
    """The binomial CV launch runs two 256-thread fits per CU when it has fits enough (config 3: 2000 fits), else one
    512-thread fit per CU; PAREBEN_BM_THREADS forces a shape.  A fit's reductions run over its own threads, so the two
    shapes may differ in the last bits -- and in nothing else: each shape repeats itself bit for bit, the two agree to 1e-10
    (the oracle table is 1e-8 away from either: test_binomial_config3_full_grid_vs_golden runs the default shape) and pick
    the same optimum."""
    X, y = golden.BASISbinomial, golden.yBinomial
    fid = AssignToFolds(X, 5)
    alpha, lam = BuildGrid(X, y, 5)
    runs = {}
    with pareben_amd.Context(X, y, fid, 5, prior="binomial") as ctx:
        for shape in ("256", "256", "512", "512"):
            monkeypatch.setenv("PAREBEN_BM_THREADS", shape)
            E, st, _ = ctx.run(alpha, lam)
            assert ctx.launch_info()["threads"] == int(shape) and np.all(st & 8 == 0)
            if shape in runs:
                assert np.array_equal(runs[shape], E)
            runs[shape] = E
        monkeypatch.delenv("PAREBEN_BM_THREADS")
        E, st, _ = ctx.run(alpha, lam)
        assert ctx.launch_info()["threads"] == 256 and np.array_equal(E, runs["256"])      # the default for a launch of this size
    assert np.abs(runs["256"] - runs["512"]).max() < 1e-10
    assert int(np.argmin(runs["256"].mean(axis=1))) == int(np.argmin(runs["512"].mean(axis=1)))


def test_binomial_synthetic_vs_oracle(oracle):
    rng = np.random.default_rng(9)
    X = np.asfortranarray(rng.standard_normal((123, 77)))
    y = (X[:, 5] - 0.7 * X[:, 40] + 0.5 * rng.standard_normal(123) > 0).astype(float)
    fid = AssignToFolds(X, 3)
    alpha, lam = BuildGrid(X, y, 3)
    sel = np.arange(0, 400, 9)
    with pareben_amd.Context(X, y, fid, 3, prior="binomial") as ctx:
        E, st, cnt = ctx.run(alpha[sel], lam[sel])
    Eo, co, rc = oracle.cv_grid(X, y, fid, 3, alpha[sel], lam[sel], prior="binomial")
    assert rc == 0 and np.all(st & 8 == 0)
    assert np.abs(E - Eo).max() < 1e-8
    assert cnt[..., 2].sum() == co["n_add"] and cnt[..., 3].sum() == co["n_del"]


def test_binomial_epistasis_vs_oracle(golden, oracle):
    """Bf through the C ABI: CrossValidate(prior = "binomial", Epis = "yes") -- the grid with the pairwise pass of
    GetLambdaMax, the NeFull.c rule set on the expanded design, held-out log-likelihood with pair columns -- against
    the oracle (implicit pair columns, the reference's association).  Oracle parity for Bf is unpinned (no
    reference-held output); the two sides share no code."""
    X, y = golden.BASISbinomial[::2, :30][:200], golden.yBinomial[::2][:200]
    fid = AssignToFolds(X, 3)
    alpha, lam = BuildGrid(X, y, 3)                              # a grid that reaches active sets of ~25 bases
    sel = np.arange(0, 400, 9)
    with pareben_amd.Context(X, y, fid, 3, prior="binomial", epis=True) as ctx:
        E, st, cnt = ctx.run(alpha[sel], lam[sel])
    Eo, co, rc = oracle.cv_grid(X, y, fid, 3, alpha[sel], lam[sel], prior="binomial", epis=True, n_threads=8)
    assert rc == 0 and np.all(st & 8 == 0)
    assert np.abs(E - Eo).max() < 1e-8
    assert cnt[..., 2].sum() == co["n_add"] and cnt[..., 3].sum() == co["n_del"] and cnt[..., 4].sum() == co["n_reest"]
    assert cnt[..., 10].max() == co["m_max"] and co["m_max"] > 20
    out = pareben_amd.CrossValidate(X, y, nFolds=3, Epis="yes", prior="binomial", search="global", return_stats=True)
    D = out["Results.Detail"]
    a2, l2 = np.asarray(D["alpha"])[::3], np.asarray(D["lambda"])[::3]
    E2 = np.asarray(D["logL"]).reshape(400, 3)
    pick = np.arange(0, 400, 37)
    Eo2, _, rc2 = oracle.cv_grid(X, y, fid, 3, a2[pick], l2[pick], prior="binomial", epis=True, n_threads=8)
    assert rc2 == 0 and np.abs(E2[pick] - Eo2).max() < 1e-8
    S = out["Results.Summary"]
    i = int(np.argmin(np.asarray(S["Likelihood"])))
    assert out["alpha.optimal"] == np.asarray(S["alpha"])[i] and out["lambda.optimal"] == np.asarray(S["lambda"])[i]


def test_epistasis_vs_golden(golden):
    """BASELINE config-4 shape at the size the oracle sweeps in seconds: Epis="yes" on
    BASIS[1:200,1:60] (1830 implicit columns), the reference's own grid (2000 fits, Q9: all <= 1
    feature) and a sub-grid on the normalised target with active sets up to ~100."""
    X, y = golden.BASIS[:200, :60], golden.y[:200]
    g, k = golden.config4, golden.known["gf_basis200x60"]
    out = pareben_amd.CrossValidate(X, y, nFolds=5, Epis="yes", prior="gaussian", search="global", return_stats=True)
    E = np.asarray(out["Results.Detail"]["MSE"]).reshape(400, 5)
    assert _rel(E, g["fold_err"]).max() < REL_FOLD
    assert out["alpha.optimal"] == k["alpha_opt"] and out["lambda.optimal"] == k["lambda_opt"]
    S = out["Results.Summary"]
    assert abs(np.min(np.asarray(S["MSE"])) - k["cv_error"]) < REL_CV * k["cv_error"]
    with pareben_amd.Context(X, g["y_scaled"], g["fold_id"], 5, epis=True) as ctx:
        E2, st, cnt = ctx.run(g["alpha_scaled"], g["lam_scaled"])
    # the reference's basisMax here is 2K = 120: the fixture (oracle with the reference's policy) has no score for
    # the small-lambda fits that need more; the HIP path flags those (bit 0) and lets them continue to
    # min(N_train, 2048) = 160 columns -- compared in test_flag_and_continue_past_reference_capacity
    ref = g["fold_err_scaled"]
    want = np.load(os.path.join(os.path.dirname(__file__), "golden", "config4_grid_status.npz"))["basis60_scaled_status"]
    assert np.array_equal(st, want) and (st == 0).sum() == 57 and (st == 1).sum() == 48      # exact status words: 57 fits inside basisMax, 48 flagged, none stopped
    ok = st == 0
    assert _rel(E2[ok], ref[ok]).max() < 1e-8
    assert np.all(np.isfinite(E2))


def test_flag_and_continue_past_reference_capacity(golden, oracle, monkeypatch):
    """Capacity policy (include/pareben_hip.h, pareben_ctx_create): a fit that outgrows the reference's basisMax
    is flagged (status bit 0) and continues in the larger workspace instead of being dropped; it is stopped (bit 3)
    only at the workspace capacity.  PAREBEN_REF_CAP lowers basisMax so that small problems take the path; the oracle
    runs the same policy.  Same scores, same flags, same event counts."""
    X, y = golden.BASIS[:150, :200], golden.y[:150]
    fid = AssignToFolds(X, 3)
    alpha, lam = BuildGrid(X, y, 3)
    sel = np.arange(2, 400, 7)
    monkeypatch.setenv("PAREBEN_REF_CAP", "12")
    with pareben_amd.Context(X, y, fid, 3) as ctx:
        E, st, cnt = ctx.run(alpha[sel], lam[sel])
        info = ctx.launch_info()
    assert info["reference_capacity"] == 12 and info["capacity"] == 200        # max(basisMax = K = 200, min(N_train, 2048))
    oracle.set_capacity_policy(True, 12)
    try:
        Eo, co, rc = oracle.cv_grid(X, y, fid, 3, alpha[sel], lam[sel])
    finally:
        oracle.set_capacity_policy(False, 0)
    flagged = (st & 1) != 0
    assert flagged.sum() > 20 and np.all((st & 8) == 0)                        # many fits outgrow 12 columns, none is stopped
    assert cnt[..., 10].max() > 12
    assert _rel(E, Eo).max() < 1e-8
    assert cnt[..., 2].sum() == co["n_add"] and cnt[..., 3].sum() == co["n_del"] and cnt[..., 4].sum() == co["n_reest"]
    assert co["status"] & 1
    # a hard limit below what the fits need: stopped fits score NaN and are reported, the rest is unchanged
    with pareben_amd.Context(X, y, fid, 3, max_active=12) as ctx:
        E2, st2, _ = ctx.run(alpha[sel], lam[sel])
    stopped = (st2 & 8) != 0
    assert np.array_equal(stopped, flagged) and np.all(np.isnan(E2[stopped])) and np.array_equal(E2[~stopped], E[~stopped])


def test_lazy_gram_rows_vs_full_matrix_and_golden(golden, monkeypatch):
    """Large-p mode (per-fold K x K Gram matrices do not fit): Gram rows are computed on first use
    into a per-fold pool shared by all workgroups (gm_row).  Forced here at small sizes:
      * 1200 fits on 3 folds -> every workgroup races for the same few rows of 3 pools;
      * same scores as the golden table, same counters as the full-matrix mode;
      * a second run (pools emptied and refilled in a different order) is bit-identical."""
    X, y = golden.BASIS[:50, :100], golden.y[:50]
    g = golden.config1
    with pareben_amd.Context(X, y, g["fold_id"], 3) as ctx:
        Ef, stf, cf = ctx.run(g["alpha"], g["lam"])
    monkeypatch.setenv("PAREBEN_GRAM_ROWS", "100")
    with pareben_amd.Context(X, y, g["fold_id"], 3) as ctx:
        El, stl, cl = ctx.run(g["alpha"], g["lam"])
        order = np.random.default_rng(3).permutation(400)
        El2, stl2, _ = ctx.run(g["alpha"][order], g["lam"][order])
    assert _rel(El, g["fold_err"]).max() < REL_FOLD
    assert np.array_equal(stl, stf) and np.array_equal(cl[..., :10], cf[..., :10])
    assert np.array_equal(El2, El[order]) and np.array_equal(stl2, stl[order])


def test_lazy_gram_rows_larger_synthetic(oracle, monkeypatch):
    """p = 3000 with a pool of 600 rows per fold: the grid touches far more than 600 features per
    fold, so most rows end up in the workgroups' private rows (recomputed per fit, as the reference
    does) while the pool serves the popular ones.  Compared with the full-matrix mode and the oracle."""
    X, y = synthetic_gaussian(400, 3000, n_causal=15, seed=77)
    fid = AssignToFolds(X, 4)
    alpha, lam = BuildGrid(X, y, 4)
    sel = np.arange(3, 400, 4)
    with pareben_amd.Context(X, y, fid, 4) as ctx:
        Ef, stf, cf = ctx.run(alpha[sel], lam[sel])
    for rows in ("3000", "600"):
        monkeypatch.setenv("PAREBEN_GRAM_ROWS", rows)
        with pareben_amd.Context(X, y, fid, 4) as ctx:
            El, stl, cl = ctx.run(alpha[sel], lam[sel])
        assert np.array_equal(stl, stf) and np.all(stl & 8 == 0)
        assert _rel(El, Ef).max() < 1e-8
        assert np.array_equal(cl[..., :6], cf[..., :6])
    spot = [0, 50, 99]
    Eo, _, rc = oracle.cv_grid(X, y, fid, 4, alpha[sel][spot], lam[sel][spot], n_threads=8)
    assert rc == 0 and _rel(El[spot], Eo).max() < 1e-8


def test_lazy_gram_rows_epistasis(golden, monkeypatch):
    """Epistasis on the expanded design (1830 columns) through the row pool."""
    X = golden.BASIS[:200, :60]
    g = golden.config4
    with pareben_amd.Context(X, g["y_scaled"], g["fold_id"], 5, epis=True) as ctx:
        Ef, stf, _ = ctx.run(g["alpha_scaled"], g["lam_scaled"])
    monkeypatch.setenv("PAREBEN_GRAM_ROWS", "700")
    with pareben_amd.Context(X, g["y_scaled"], g["fold_id"], 5, epis=True) as ctx:
        El, stl, _ = ctx.run(g["alpha_scaled"], g["lam_scaled"])
    assert np.array_equal(stl, stf)
    ok = (stl & 9) == 0
    assert _rel(El[ok], Ef[ok]).max() < 1e-8
    # fits past the reference's capacity (120 columns) run on to as many columns as training rows (160): nearly singular
    # systems, on which the two modes' different summation order of a Gram row shows (observed 1e-6)
    big = (stl & 9) == 1
    assert big.sum() > 0 and _rel(El[big], Ef[big]).max() < 1e-4


def test_shared_phases_bit_identical(oracle, monkeypatch):
    """Fewer fits than workgroups: the queue is drained at once, so every full-stat pass, action
    mat-vec and batched-add sweep of the running fits is opened to ~200 idle workgroups (job board, CAS
    claims, agent-scope hand-offs).  Scores, status and counters must be bit-identical to the run with
    sharing off."""
    X, y = synthetic_gaussian(500, 4500, n_causal=25, seed=4242)
    fid = AssignToFolds(X, 3)
    alpha, lam = BuildGrid(X, y, 3)
    sel = np.array([45, 130, 135, 190, 250, 255, 310, 375, 399])        # around and below the sparse-to-dense transition
    out = {}
    for share in ("0", "1"):
        monkeypatch.setenv("PAREBEN_SHARE", share)
        with pareben_amd.Context(X, y, fid, 3) as ctx:
            out[share] = ctx.run(alpha[sel], lam[sel])
            if share == "1":
                again = ctx.run(alpha[sel][::-1], lam[sel][::-1])
    E0, st0, c0 = out["0"]
    E1, st1, c1 = out["1"]
    assert np.all(st1 & 8 == 0) and c1[..., 10].max() >= 96             # active sets large enough to open both job kinds
    assert np.array_equal(E0, E1) and np.array_equal(st0, st1) and np.array_equal(c0[..., :11], c1[..., :11])
    assert np.array_equal(again[0][::-1], E1)
    Eo, _, rc = oracle.cv_grid(X, y, fid, 3, alpha[sel][:2], lam[sel][:2], n_threads=6)
    assert rc == 0 and _rel(E1[:2], Eo).max() < 1e-8


def test_odd_feature_count_with_shared_phases(oracle, monkeypatch):
    """p odd and not a multiple of any tile size (4357 = 17 full-stat tiles + 5 features, 34 mat-vec tiles + 5,
    one unpaired last feature), active sets in the hundreds, every shareable phase opened from the start of
    the launch: bit-identical to the unshared run, and both against the oracle."""
    X, y = synthetic_gaussian(320, 4357, n_causal=30, seed=77)
    fid = AssignToFolds(X, 3)
    alpha, lam = BuildGrid(X, y, 3)
    sel = np.array([130, 190, 250, 255, 310, 330])
    out = {}
    for share in ("0", "2"):
        monkeypatch.setenv("PAREBEN_SHARE", share)
        with pareben_amd.Context(X, y, fid, 3) as ctx:
            out[share] = ctx.run(alpha[sel], lam[sel])
    E0, st0, c0 = out["0"]
    E2, st2, c2 = out["2"]
    assert np.all(st2 & 8 == 0) and c2[..., 10].max() >= 96
    assert np.array_equal(E0, E2) and np.array_equal(st0, st2) and np.array_equal(c0[..., :11], c2[..., :11])
    Eo, _, rc = oracle.cv_grid(X, y, fid, 3, alpha[sel][:3], lam[sel][:3], n_threads=6)
    assert rc == 0 and _rel(E2[:3], Eo).max() < 1e-8


def test_paper_epistasis_dataset_vs_oracle(oracle):
    """The Epis data set of the authors' timing script (yeast genotypes, n = 200; tools/make_golden.py):
    its first 90 markers -> 4095 implicit columns, five cells around the sparse-to-dense transition
    against the oracle, plus search = "local" reporting a cell of the global table (the full 300 / 600
    marker jobs: profiles/r01/paper_timing_jobs.json)."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "yeast_timing_200x600.npz"))
    B = np.asfortranarray(np.unpackbits(d["bits"], axis=0)[:200].astype(np.float64) * 2.0 - 1.0)
    X, y = np.asfortranarray(B[:, :90]), d["y"].astype(np.float64)
    glo = pareben_amd.CrossValidate(X, y, nFolds=5, Epis="yes", prior="gaussian", search="global", return_stats=True)
    loc = pareben_amd.CrossValidate(X, y, nFolds=5, Epis="yes", prior="gaussian", search="local")
    # exact status words of all 2000 fits (tools/config4_table.py 90): 1755 plain, 244 through the reference's stale-slot
    # delete (bit 2), and ONE stopped: cell 79 (alpha 0.05, lambda 8.45) fold 4, status 9 = its active set reached the
    # workspace = the reference's own basisMax 2K = 180 columns (Full2.c:67-80, N_train = 160 > K = 90) -- more columns than
    # training rows, which the Gf rule set allows (no M >= N delete priority, Full2.c:1257) and where the reference itself
    # prints "out of Memory" and runs off its arrays (MainEff.c:605-611 has the same code)
    st_all = glo["stats"]["status"]
    want = np.load(os.path.join(os.path.dirname(__file__), "golden", "config4_grid_status.npz"))["k90_status"]
    assert np.array_equal(st_all, want)
    assert np.argwhere(st_all & 8).tolist() == [[79, 3]] and st_all[79, 3] == 9 and glo["stats"]["counters"][79, 3, 10] == 180
    ok = (st_all & 8 == 0).all(axis=1)
    alpha, lam = BuildGrid(X, y, 5, "yes")
    # the early-stopping walk need not find the global optimum; what it reports is a cell of the same table
    S = glo["Results.Summary"]
    j = np.nonzero((np.asarray(S["alpha"]) == loc["alpha.optimal"]) & (np.asarray(S["lambda"]) == loc["lambda.optimal"]))[0]
    k = int(np.argmin(loc["CrossValidation"][:, 2]))
    assert len(j) == 1 and abs(np.asarray(S["MSE"])[j[0]] - loc["CrossValidation"][k, 2]) < 1e-9 * loc["CrossValidation"][k, 2]
    fid = AssignToFolds(X, 5)
    E = np.asarray(glo["Results.Detail"]["MSE"]).reshape(400, 5)
    m = np.where(ok, glo["stats"]["counters"][..., 10].max(axis=1), -1)
    sel = np.argsort(-m)[:5]                                   # the five complete cells with the largest active sets
    Eo, _, rc = oracle.cv_grid(X, y, fid, 5, alpha[sel], lam[sel], epis=True, n_threads=8)
    assert rc == 0 and m[sel].max() >= 20
    assert _rel(E[sel], Eo).max() < 1e-6                       # active sets close to N = 160: ill-conditioned, observed <= 6e-8


def test_config2_cells_vs_oracle_fixture():
    """BASELINE config 2 at full size (synthetic n=1000, p=10000, nFolds=5): eight cells spread over the
    grid against the oracle's committed table (tools/make_config2_golden.py: tens of CPU-minutes, so a
    fixture): fold SSE <= 1e-8 relative and the same add / delete / re-estimate / full-stat counts."""
    import os
    from pareben_amd.synth import synthetic_gaussian as synth
    path = os.path.join(os.path.dirname(__file__), "golden", "config2_cells.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    g = np.load(path)
    X, y, _, _ = synth(1000, 10000)
    with pareben_amd.Context(X, y, g["fold_id"], 5) as ctx:
        E, st, cnt = ctx.run(g["alpha"], g["lam"])
    assert np.all(st & 8 == 0)
    assert _rel(E, g["fold_err"]).max() < 1e-8
    want = dict(zip([str(s) for s in g["counter_names"]], g["counters"]))
    tot = cnt.sum(axis=(0, 1))
    for j, n in enumerate(("n_outer", "n_inner", "n_add", "n_del", "n_reest", "n_fullstat")):
        assert tot[j] == want[n], n


def test_small_and_ragged_shapes(oracle):
    """The smallest designs the entry points accept, against the oracle: one column; two columns with their pair
    (Epis: 3 columns, fewer than a matrix-core tile); five training rows; more folds than a multiple of anything; a
    fold whose held-out set is one row; K = 129 (one column past a 128-block of gram_kernel) with N = 17 (one sample
    past a 16-sample slab)."""
    rng = np.random.default_rng(21)
    cases = []
    X1 = np.asfortranarray(rng.standard_normal((12, 1))); cases.append((X1, 2.0 * X1[:, 0] + 0.1 * rng.standard_normal(12), 3, False))
    X2 = np.asfortranarray(rng.standard_normal((15, 2))); cases.append((X2, X2[:, 0] * X2[:, 1] + 0.1 * rng.standard_normal(15), 3, True))
    X3 = np.asfortranarray(rng.standard_normal((7, 4))); cases.append((X3, X3[:, 2] + 0.05 * rng.standard_normal(7), 7, False))     # leave-one-out
    X4 = np.asfortranarray(rng.standard_normal((26, 129))); cases.append((X4, X4[:, 128] - X4[:, 0] + 0.1 * rng.standard_normal(26), 3, False))
    for X, y, nf, epis in cases:
        fid = AssignToFolds(X, nf)
        alpha, lam = BuildGrid(X, y, nf, "yes" if epis else "no")
        sel = np.arange(0, 400, 23)
        with pareben_amd.Context(X, y, fid, nf, epis=epis) as ctx:
            E, st, cnt = ctx.run(alpha[sel], lam[sel])
        Eo, co, rc = oracle.cv_grid(X, y, fid, nf, alpha[sel], lam[sel], epis=epis)
        ok = (st & 9) == 0
        assert ok.mean() > 0.9
        assert _rel(E[ok], Eo[ok]).max() < 1e-7, (X.shape, epis)
    # argument errors of the epistasis entries
    with pytest.raises(pareben_amd.ParebenError):
        pareben_amd.fit_gaussian(X1, np.zeros(12), 0.1, 0.5, epis=True)          # one column has no pair
    with pytest.raises(pareben_amd.ParebenError):
        pareben_amd.fit_binomial(X1, np.zeros(12), 0.1, 0.5, epis=True)


def test_lambda_max_pair_pass_on_device(golden, yeast):
    """pareben_lambda_max_pairs (R/BuildGrid.R:21-30 on the GPU) against the numpy restatement of the same R
    expressions: bundled BASIS (with an all-zero pair column: 0/0 never wins), a Gaussian design with ragged n, and the
    paper's Epis size (yeast n = 200, k = 300: 44 850 pairs).  BuildGrid(device=...) then yields the same grid."""
    from pareben_amd.grid import _pairs_host, GetLambdaMax
    cases = [(golden.BASIS[:200, :60], golden.y[:200])]
    rng = np.random.default_rng(4)
    Xg = rng.standard_normal((157, 41)); Xg[:, 3] = 0.0
    cases.append((Xg, rng.standard_normal(157) * 3 + 1))
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "yeast_timing_200x600.npz"))
    B = np.unpackbits(d["bits"], axis=0)[:int(d["n"])].astype(np.float64) * 2.0 - 1.0
    cases.append((B[:, :300], d["y"].astype(np.float64)))
    for X, y in cases:
        host = _pairs_host(np.asarray(X, dtype=np.float64), y - y.mean())
        dev = pareben_amd._lib.lambda_max_pairs(X, y)
        assert abs(dev - host) <= 1e-12 * abs(host), (dev, host)
        a1, l1 = BuildGrid(X, y, 5, "yes", device=None)
        a2, l2 = BuildGrid(X, y, 5, "yes", device=0)
        assert np.array_equal(a1, a2) and np.allclose(l1, l2, rtol=1e-12, atol=0)
    assert pareben_amd._lib.lambda_max_pairs(golden.BASIS[:50, :1], golden.y[:50]) == -np.inf


def test_multi_gpu_entry(golden, monkeypatch):
    """pareben_cv_grid_multi (one process, a host thread + context per device, every device's fit kernel pulling units
    from ONE queue head in pinned host memory, one grouped ncclAllGather of the per-device tables): bit-identical to the
    single-context run for Gaussian and binomial grids, with n_gpu given and defaulted.  With >= 2 devices visible the
    n_gpu = 2 branch runs through RCCL and must report a 2-rank communicator; on a one-GPU box the N > 1 deal and merge
    are exercised by two ranks on device 0 (PAREBEN_MULTI_DEVICES, host merge: RCCL refuses duplicate devices)."""
    X, y = golden.BASIS[:120, :90], golden.y[:120]
    fid = AssignToFolds(X, 3)
    alpha, lam = BuildGrid(X, y, 3)
    sel = np.arange(0, 400, 3)
    with pareben_amd.Context(X, y, fid, 3) as ctx:
        E1, s1, c1 = ctx.run(alpha[sel], lam[sel])
    n_dev = pareben_amd.load_library().pareben_device_count()
    for n_gpu in sorted({1, 0, min(n_dev, 2)}):
        E2, s2, c2 = pareben_amd.cv_grid_multi(X, y, fid, 3, alpha[sel], lam[sel], n_gpu=n_gpu)
        assert np.array_equal(E1, E2) and np.array_equal(s1, s2) and np.array_equal(c1, c2), n_gpu
        ranks, most, least, gpus = pareben_amd.multi_last_stats()
        want = n_dev if n_gpu == 0 else n_gpu
        assert gpus == want and ranks == (want if want > 1 else 1) and most + least * (want - 1) <= 3 * len(sel) <= most * want
    if n_dev >= 2:
        pareben_amd.cv_grid_multi(X, y, fid, 3, alpha[sel], lam[sel], n_gpu=2)
        assert pareben_amd.multi_last_stats()[0] == 2                # RCCL saw two ranks
    monkeypatch.setenv("PAREBEN_MULTI_DEVICES", "0,0,0")             # three ranks on device 0: shared queue + merge
    E5, s5, c5 = pareben_amd.cv_grid_multi(X, y, fid, 3, alpha[sel], lam[sel], n_gpu=1)
    monkeypatch.delenv("PAREBEN_MULTI_DEVICES")
    assert np.array_equal(E1, E5) and np.array_equal(s1, s5) and np.array_equal(c1, c5)
    ranks, most, least, gpus = pareben_amd.multi_last_stats()
    assert gpus == 3 and most + least <= 3 * len(sel) and most >= len(sel)
    out = pareben_amd.CrossValidate(X, y, nFolds=3, nGPU=0)
    ref = pareben_amd.CrossValidate(X, y, nFolds=3)
    assert out["lambda.optimal"] == ref["lambda.optimal"] and out["alpha.optimal"] == ref["alpha.optimal"]
    assert np.array_equal(np.asarray(out["Results.Detail"]["MSE"]), np.asarray(ref["Results.Detail"]["MSE"]))
    Xb, yb = golden.BASISbinomial[::2, :80], golden.yBinomial[::2]      # both classes present
    fb = AssignToFolds(Xb, 2)
    ab, lb = BuildGrid(Xb, yb, 2)
    with pareben_amd.Context(Xb, yb, fb, 2, prior="binomial") as ctx:
        E3, s3, _ = ctx.run(ab[::7], lb[::7])
    E4, s4, _ = pareben_amd.cv_grid_multi(Xb, yb, fb, 2, ab[::7], lb[::7], prior="binomial", n_gpu=1)
    assert np.array_equal(E3, E4) and np.array_equal(s3, s4)
    with pytest.raises(pareben_amd.ParebenError):
        pareben_amd.cv_grid_multi(X, y, fid, 3, alpha[sel], lam[sel], n_gpu=n_dev + 1)


def test_config5_full_size_properties():
    """BASELINE config 5 at its stated size (synthetic n = 2000, p = 50000, nFolds = 10, 20 alpha x 200 lambda;
    ten 20 GB Gram matrices resident in HBM), on twelve cells spread over the lambda range (120 fits; the whole
    grid is `bench.py --workload config5`, 40 000 fits, profiles/r02/).  The oracle cannot follow at this size, so
    size-independent properties: every score finite (the reference's basisMax = 1e7/p = 200 only flags a fit
    here; the workspace holds min(N_train, 2048) = 1800 columns), null-model bound at the largest lambda, informative cells beat the
    null model, target-shift invariance, bit-identical rerun in another order."""
    from pareben_amd.synth import synthetic_gaussian as bench_design
    n, p, nf = 2000, 50000, 10
    X, y, _, _ = bench_design(n, p)
    fid = AssignToFolds(X, nf)
    alpha, lam = BuildGrid(X, y, nf, nAlpha=20, nLambda=200)
    A = alpha.reshape(200, 20); L = lam.reshape(200, 20)             # [lambda index, alpha index], alpha fastest
    li = np.array([0, 40, 80, 100, 120, 160]); ai = np.array([0, 10])
    sel = (li[:, None] * 20 + ai[None, :]).ravel()
    assert np.all(A.ravel()[sel].reshape(6, 2)[:, 0] == 1.0)
    with pareben_amd.Context(X, y, fid, nf) as ctx:
        E, st, cnt = ctx.run(alpha[sel], lam[sel])
        info = ctx.launch_info()
        perm = np.random.default_rng(5).permutation(len(sel))
        E2, st2, _ = ctx.run(alpha[sel][perm], lam[sel][perm])
    assert info["reference_capacity"] == 200 and info["capacity"] == 1800
    assert np.all((st & 8) == 0) and np.all(np.isfinite(E)) and np.all(E > 0)
    assert ((st & 1) != 0).sum() > 0 and cnt[..., 10].max() > 200       # fits the reference would have run off its arrays with
    assert np.array_equal(E2, E[perm]) and np.array_equal(st2, st[perm])
    null = np.array([np.sum((y[fid == f + 1] - y[fid != f + 1].mean()) ** 2) for f in range(nf)])
    assert np.all(E < 1.01 * null[None, :])                          # no cell does worse than the intercept-only model
    assert E.mean(axis=1).min() < 0.1 * null.mean()                  # 20 strong causal columns: the good cells explain > 90 %
    spot = [0, 4, 9]
    with pareben_amd.Context(X, y + 3.0, fid, nf) as ctx:
        Es, _, _ = ctx.run(alpha[sel][spot], lam[sel][spot])
    assert _rel(Es, E[spot]).max() < 1e-6


def test_active_sets_beyond_1024_columns(monkeypatch):
    """Two single fits whose active sets peak at 1087 and 1239 columns (synthetic n = 3000, p = 4500, lambda =
    1e-5 lambda_max, alpha = 0.05 and 1; the reference's own basisMax there is 1e7/4500 = 2222, so these are fits the
    reference handles, not flag-and-continue cases) against the oracle (tools/make_big_m_oracle.py, ~20 CPU-minutes each).
    Past 1040 columns the blocked inverse's pivot-column panel lives in the fit's HBM scratch instead of LDS
    (gm_spd_inverse_blocked<global>), and the full-stat pass runs its 16-row blocks up to FS_MAX_M = 2048: the same
    add / delete / re-estimate sequence, the same model, effects to 1e-7."""
    import os
    from pareben_amd.synth import synthetic_gaussian
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "big_m_oracle.npz"))
    X, y, _, _ = synthetic_gaussian(int(g["n"]), int(g["p"]))
    for i in (0, 1):
        pre = "c%d_" % i
        want = dict(zip([str(k) for k in g[pre + "counter_names"]], [int(v) for v in g[pre + "counters"]]))
        r = pareben_amd.fit_gaussian(X, y, float(g[pre + "lambda"]), float(g[pre + "alpha"]))
        c = r["counters"]
        assert c["status"] == 0 and c["m_max"] > 1040
        for k in ("n_outer", "n_inner", "n_add", "n_del", "n_reest", "n_fullstat", "m_final", "m_max", "sum_m_action"):
            assert c[k] == want[k], (k, c[k], want[k])
        b, bo = r["Beta"][:, 2], g[pre + "beta"]
        assert np.array_equal(b != 0, bo != 0)
        nz = bo != 0
        assert _rel(b[nz], bo[nz]).max() < 1e-7 and _rel(r["Beta"][nz, 3], g[pre + "var"][nz]).max() < 1e-7
        assert abs(r["wald"] - float(g[pre + "wald"])) < 1e-9 * abs(r["wald"])
        assert abs(r["residual"] - float(g[pre + "residual"])) < 1e-9 * r["residual"]
        assert abs(r["intercept"] - float(g[pre + "intercept"])) < 1e-7 * max(abs(r["intercept"]), 1e-3)
    # the HBM panel of the inverse and the 16-row pad bands up to 1088 columns do not depend on what the workspace held
    monkeypatch.setenv("PAREBEN_WS_POISON", "1")
    r2 = pareben_amd.fit_gaussian(X, y, float(g["c1_lambda"]), float(g["c1_alpha"]))
    assert np.array_equal(r2["Beta"], r["Beta"]) and r2["wald"] == r["wald"] and r2["counters"] == r["counters"]
