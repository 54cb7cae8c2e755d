"""ctypes binding of libpareben_hip.so (include/pareben_hip.h).  There is no CPU fallback: if the
HIP library is missing or no GPU is visible every compute entry point raises."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PAREBEN_LIB=<path> loads another build of the same library (A/B timing of kernel variants, diagnostic builds)
LIB_PATH = os.environ.get("PAREBEN_LIB") or os.path.join(_HERE, "lib", "libpareben_hip.so")
NCOUNTERS = 14
COUNTER_NAMES = ("n_outer", "n_inner", "n_add", "n_del", "n_reest", "n_fullstat", "sum_m_action",
                 "sum_m_full", "sum_m2_full", "m_final", "m_max", "status", "mfma_tiles", "sum_m_swept")
ST_OVERFLOW, ST_CHOLESKY, ST_STALE, ST_ABORT = 1, 2, 4, 8

_lib = None


class ParebenError(RuntimeError):
    pass


def load():
    """Load the shared library (does not touch the GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ParebenError(
            "libpareben_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pareben_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    L.pareben_version.restype = C.c_char_p
    L.pareben_last_error.restype = C.c_char_p
    L.pareben_device_count.restype = C.c_int
    L.pareben_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, dp, C.c_int, C.c_int, dp, ip, C.c_int, C.c_int, C.c_int, C.c_int]
    L.pareben_ctx_run.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, ip, lp]
    L.pareben_ctx_last_timing.argtypes = [C.c_void_p, dp]
    L.pareben_ctx_launch_info.argtypes = [C.c_void_p, lp]
    L.pareben_ctx_destroy.argtypes = [C.c_void_p]
    L.pareben_ctx_gram.argtypes = [C.c_void_p, C.c_int, dp]
    L.pareben_cv_grid.argtypes = [dp, C.c_int, C.c_int, dp, ip, C.c_int, dp, dp, C.c_int, C.c_int, C.c_int, C.c_int, dp, ip, lp]
    L.pareben_cv_grid_multi.argtypes = L.pareben_cv_grid.argtypes
    L.pareben_lambda_max_pairs.argtypes = [dp, C.c_int, C.c_int, dp, C.c_int, dp]
    L.pareben_fit_gaussian.argtypes = [dp, dp, C.c_double, C.c_double, dp, dp, dp, C.c_int, C.c_int, C.c_int, dp, C.c_int, lp]
    L.pareben_fit_gaussian_epis.argtypes = L.pareben_fit_gaussian.argtypes
    L.pareben_fit_binomial.argtypes = [dp, dp, C.c_double, C.c_double, dp, dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, lp]
    L.pareben_fit_binomial_epis.argtypes = L.pareben_fit_binomial.argtypes
    _lib = L
    return L


def _chk(rc, what):
    if rc != 0:
        raise ParebenError("%s failed (code %d): %s" % (what, rc, load().pareben_last_error().decode()))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class Context:
    """One CV problem staged in HBM (pareben_ctx_create / _run / _destroy)."""

    def __init__(self, BASIS, Target, fold_id, n_folds, prior="gaussian", epis=False, device=0, max_active=0):
        L = load()
        X = np.asfortranarray(BASIS, dtype=np.float64)
        y = np.ascontiguousarray(Target, dtype=np.float64).reshape(-1)
        fid = np.ascontiguousarray(fold_id, dtype=np.int32).reshape(-1)
        if X.ndim != 2 or y.shape[0] != X.shape[0] or fid.shape[0] != X.shape[0]:
            raise ValueError("BASIS must be n x p and Target / fold_id must have n entries")
        if prior not in ("gaussian", "binomial"):
            raise ValueError('prior must be "gaussian" or "binomial"')
        self.n, self.p = X.shape
        self.n_folds = int(n_folds)
        self.epis = bool(epis)
        self._h = C.c_void_p()
        _chk(L.pareben_ctx_create(C.byref(self._h), int(device), _dp(X), self.n, self.p, _dp(y), _ip(fid), self.n_folds,
                                  0 if prior == "gaussian" else 1, 1 if epis else 0, int(max_active)), "pareben_ctx_create")

    def run(self, alpha, lam, want_counters=True):
        """-> (fold_err [n_cells, n_folds], status [n_cells, n_folds], counters [n_cells, n_folds, 13] | None)"""
        L = load()
        alpha = np.ascontiguousarray(alpha, dtype=np.float64).reshape(-1)
        lam = np.ascontiguousarray(lam, dtype=np.float64).reshape(-1)
        nc = alpha.shape[0]
        err = np.empty((nc, self.n_folds))
        st = np.empty((nc, self.n_folds), dtype=np.int32)
        cnt = np.empty((nc, self.n_folds, NCOUNTERS), dtype=np.int64) if want_counters else None
        _chk(L.pareben_ctx_run(self._h, nc, _dp(alpha), _dp(lam), _dp(err), _ip(st),
                               _lp(cnt) if cnt is not None else None), "pareben_ctx_run")
        return err, st, cnt

    def gram(self, fold):
        """Test hook (pareben_ctx_gram): normalised Gram matrix of 0-based fold `fold`, [K, K], row u = Gram row of basis u."""
        k = self.p * (self.p + 1) // 2 if self.epis else self.p
        out = np.empty((k, k))
        _chk(load().pareben_ctx_gram(self._h, int(fold), _dp(out)), "pareben_ctx_gram")
        return out

    def last_timing(self):
        ms = np.zeros(3)
        _chk(load().pareben_ctx_last_timing(self._h, _dp(ms)), "pareben_ctx_last_timing")
        return {"prep_ms": float(ms[0]), "fit_ms": float(ms[1]), "total_ms": float(ms[2])}

    def launch_info(self):
        info = np.zeros(5, dtype=np.int64)
        _chk(load().pareben_ctx_launch_info(self._h, _lp(info)), "pareben_ctx_launch_info")
        return {"workgroups": int(info[0]), "threads": int(info[1]), "capacity": int(info[2]), "ws_kib": int(info[3]),
                "reference_capacity": int(info[4])}

    def close(self):
        if self._h:
            load().pareben_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def lambda_max_pairs(BASIS, Target, device=0):
    """pareben_lambda_max_pairs: the pairwise pass of GetLambdaMax (R/BuildGrid.R:21-30) on the GPU -> float (-inf if no pair counts)."""
    X = np.asfortranarray(BASIS, dtype=np.float64)
    y = np.ascontiguousarray(Target, dtype=np.float64).reshape(-1)
    out = C.c_double(0)
    _chk(load().pareben_lambda_max_pairs(_dp(X), X.shape[0], X.shape[1], _dp(y), int(device), C.byref(out)), "pareben_lambda_max_pairs")
    return out.value


def cv_grid_multi(BASIS, Target, fold_id, n_folds, alpha, lam, prior="gaussian", epis=False, n_gpu=0, want_counters=True):
    """pareben_cv_grid_multi: the grid on n_gpu devices from this one process (one host thread + context per device,
    one RCCL all-gather of the per-cell results); n_gpu <= 0 = every visible device.
    -> (fold_err [n_cells, n_folds], status, counters | None)"""
    L = load()
    X = np.asfortranarray(BASIS, dtype=np.float64)
    y = np.ascontiguousarray(Target, dtype=np.float64).reshape(-1)
    fid = np.ascontiguousarray(fold_id, dtype=np.int32).reshape(-1)
    alpha = np.ascontiguousarray(alpha, dtype=np.float64).reshape(-1)
    lam = np.ascontiguousarray(lam, dtype=np.float64).reshape(-1)
    n, p = X.shape
    nc = alpha.shape[0]
    err = np.empty((nc, n_folds))
    st = np.empty((nc, n_folds), dtype=np.int32)
    cnt = np.zeros((nc, n_folds, NCOUNTERS), dtype=np.int64) if want_counters else None
    _chk(L.pareben_cv_grid_multi(_dp(X), n, p, _dp(y), _ip(fid), int(n_folds), _dp(alpha), _dp(lam), nc,
                                 1 if epis else 0, 0 if prior == "gaussian" else 1, int(n_gpu),
                                 _dp(err), _ip(st), _lp(cnt) if cnt is not None else None), "pareben_cv_grid_multi")
    return err, st, cnt


def multi_last_stats():
    """(ranks in the RCCL communicator | 1 when nothing was exchanged, units pulled by the busiest GPU, by the idlest, GPUs)
    of the last cv_grid_multi call."""
    out = np.zeros(4, dtype=np.int64)
    _chk(load().pareben_multi_last_stats(_lp(out)), "pareben_multi_last_stats")
    return tuple(int(v) for v in out)


def fit_gaussian(BASIS, Target, lam, alpha, device=0, epis=False):
    """Mirror of the reference's .C("elasticNetLinearNeMainEff") / .C("elasticNetLinearNeEpisEff") tuples
    (EBEN_orig/R/EBelasticNet.Gaussian.R:16-51) -> dict(Beta K x 4 | K(K+1)/2 x 5, wald, intercept,
    residual, counters)."""
    L = load()
    X = np.asfortranarray(BASIS, dtype=np.float64)
    y = np.ascontiguousarray(Target, dtype=np.float64).reshape(-1)
    n, k = X.shape
    Beta = np.zeros((k * (k + 1) // 2, 5), order="F") if epis else np.zeros((k, 4), order="F")
    wald, icpt, resid = C.c_double(0), C.c_double(0), C.c_double(0)
    cnt = np.zeros(NCOUNTERS, dtype=np.int64)
    fn, name = (L.pareben_fit_gaussian_epis, "pareben_fit_gaussian_epis") if epis else (L.pareben_fit_gaussian, "pareben_fit_gaussian")
    _chk(fn(_dp(X), _dp(y), float(lam), float(alpha), _dp(Beta), C.byref(wald), C.byref(icpt),
            n, k, 0, C.byref(resid), int(device), _lp(cnt)), name)
    return dict(Beta=Beta, wald=wald.value, intercept=icpt.value, residual=resid.value,
                counters=dict(zip(COUNTER_NAMES, (int(v) for v in cnt))))


def fit_binomial(BASIS, Target, lam, alpha, device=0, epis=False):
    """Mirror of the reference's .C("ElasticNetBinaryNEmainEff") / .C("ElasticNetBinaryNEfull") tuples
    (EBEN_orig/R/EBelasticNet.Binomial.R:6-46) -> dict(Beta, logLikelihood, wald, intercept[2] = (mu0, Sigma00), counters).
    Beta: K x 4 indexed by column (main effects), or 2K x 4 listing the used bases in model order (Epis = "yes")."""
    L = load()
    X = np.asfortranarray(BASIS, dtype=np.float64)
    y = np.ascontiguousarray(Target, dtype=np.float64).reshape(-1)
    n, k = X.shape
    Beta = np.zeros((2 * k if epis else k, 4), order="F")
    ll, wald = C.c_double(0), C.c_double(0)
    icpt = np.zeros(2)
    cnt = np.zeros(NCOUNTERS, dtype=np.int64)
    fn, name = (L.pareben_fit_binomial_epis, "pareben_fit_binomial_epis") if epis else (L.pareben_fit_binomial, "pareben_fit_binomial")
    _chk(fn(_dp(X), _dp(y), float(lam), float(alpha), C.byref(ll), _dp(Beta), C.byref(wald), _dp(icpt),
            n, k, 0, 2 * k if epis else k, int(device), _lp(cnt)), name)
    return dict(Beta=Beta, logLikelihood=ll.value, wald=wald.value, intercept=icpt,
                counters=dict(zip(COUNTER_NAMES, (int(v) for v in cnt))))


def dot_c(name, BASIS, Target, lam, alpha, verbose=0):
    """R's .C(name, BASIS, Target, lamda, alpha, ..., PACKAGE = "pareben_hip") spelled in ctypes: the reference's own
    symbol (elasticNetLinearNeMainEff | elasticNetLinearNeEpisEff | ElasticNetBinaryNEmainEff | ElasticNetBinaryNEfull),
    every argument a pointer, outputs written in place, no return value -- the tuples of
    EBEN_orig/R/EBelasticNet.Gaussian.R:16-51 and EBEN_orig/R/EBelasticNet.Binomial.R:10-46 -> the same dicts as
    fit_gaussian / fit_binomial (without counters)."""
    L = load()
    X = np.asfortranarray(BASIS, dtype=np.float64)
    y = np.ascontiguousarray(Target, dtype=np.float64).reshape(-1)
    n, k = X.shape
    N, K, VB = C.c_int(n), C.c_int(k), C.c_int(int(verbose))
    lam_, alpha_ = C.c_double(float(lam)), C.c_double(float(alpha))
    fn = getattr(L, name)
    fn.restype = None
    wald = C.c_double(0)
    if name in ("elasticNetLinearNeMainEff", "elasticNetLinearNeEpisEff"):
        epis = name.endswith("EpisEff")
        Beta = np.zeros((k * (k + 1) // 2, 5), order="F") if epis else np.zeros((k, 4), order="F")
        icpt, resid = C.c_double(0), C.c_double(0)
        fn(_dp(X), _dp(y), C.byref(lam_), C.byref(alpha_), _dp(Beta), C.byref(wald), C.byref(icpt), C.byref(N), C.byref(K),
           C.byref(VB), C.byref(resid))
        return dict(Beta=Beta, wald=wald.value, intercept=icpt.value, residual=resid.value)
    if name in ("ElasticNetBinaryNEmainEff", "ElasticNetBinaryNEfull"):
        epis = name.endswith("full")
        bmax = C.c_int(2 * k if epis else k)
        Beta = np.zeros((bmax.value, 4), order="F")
        ll = C.c_double(0)
        icpt = np.zeros(2)
        fn(_dp(X), _dp(y), C.byref(lam_), C.byref(alpha_), C.byref(ll), _dp(Beta), C.byref(wald), _dp(icpt), C.byref(N), C.byref(K),
           C.byref(VB), C.byref(bmax))
        return dict(Beta=Beta, logLikelihood=ll.value, wald=wald.value, intercept=icpt)
    raise ValueError(name)
