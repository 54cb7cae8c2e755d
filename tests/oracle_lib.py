"""ctypes front-end to oracle/liboracle.so (the CPU checker).  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
pareben_amd/."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_outer", "n_inner", "n_add", "n_del", "n_reest", "n_fullstat", "sum_m_action",
        "sum_m_full", "sum_m2_full", "m_final", "m_max", "status")]

    def asdict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("EBEN_ORACLE_LIB") or os.path.join(_ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        _LIB.eben_gm_fit.argtypes = [dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, dp, dp, C.POINTER(Counters)]
        _LIB.eben_gf_fit.argtypes = _LIB.eben_gm_fit.argtypes
        _LIB.eben_bm_fit.argtypes = [dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, dp, dp, C.POINTER(Counters)]
        _LIB.eben_bf_fit.argtypes = [dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, C.c_int, dp, dp, C.POINTER(Counters)]
        _LIB.eben_cv_grid.argtypes = [dp, C.c_int, C.c_int, dp, ip, C.c_int, dp, dp, C.c_int, C.c_int, C.c_int, C.c_int, dp, C.POINTER(Counters)]
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def set_capacity_policy(continue_past_basismax=False, ref_cap_override=0):
    """(False, 0): the reference's behaviour (a fit needing more than basisMax columns is stopped);
    (True, r): the HIP build's flag-and-continue policy, basisMax lowered to r when r > 0."""
    lib().eben_set_capacity_policy(int(bool(continue_past_basismax)), int(ref_cap_override))


def fit_gaussian(X, y, lam, alpha, epis=False):
    """-> dict(Beta (n_eff x 4|5), wald, intercept, residual, counters, rc)"""
    X = np.asfortranarray(X, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
    N, K = X.shape
    n_eff = K * (K + 1) // 2 if epis else K
    cols = 5 if epis else 4
    Beta = np.zeros((n_eff, cols), order="F")
    wald, icpt, resid = C.c_double(0), C.c_double(0), C.c_double(0)
    cnt = Counters()
    fn = lib().eben_gf_fit if epis else lib().eben_gm_fit
    rc = fn(_dp(X), _dp(y), N, K, float(lam), float(alpha), _dp(Beta), C.byref(wald), C.byref(icpt), C.byref(resid), C.byref(cnt))
    return dict(Beta=Beta, wald=wald.value, intercept=icpt.value, residual=resid.value, counters=cnt.asdict(), rc=rc)


def fit_binomial(X, y, lam, alpha, epis=False):
    """epis=False: Beta K x 4 (ElasticNetBinaryNEmainEff); epis=True: Beta 2K x 4 = the used bases in model order
    (ElasticNetBinaryNEfull with the R wrapper's bMax = 2K)."""
    X = np.asfortranarray(X, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
    N, K = X.shape
    Beta = np.zeros((2 * K if epis else K, 4), order="F")
    ll, wald = C.c_double(0), C.c_double(0)
    icpt = np.zeros(2)
    cnt = Counters()
    if epis:
        rc = lib().eben_bf_fit(_dp(X), _dp(y), N, K, float(lam), float(alpha), C.byref(ll), _dp(Beta), 2 * K, C.byref(wald), _dp(icpt), C.byref(cnt))
    else:
        rc = lib().eben_bm_fit(_dp(X), _dp(y), N, K, float(lam), float(alpha), C.byref(ll), _dp(Beta), C.byref(wald), _dp(icpt), C.byref(cnt))
    return dict(Beta=Beta, loglik=ll.value, wald=wald.value, intercept=icpt, counters=cnt.asdict(), rc=rc)


def cv_grid(BASIS, y, fold_id, n_folds, alpha, lam, prior="gaussian", epis=False, n_threads=0):
    """-> (fold_err [n_cells x n_folds], counters dict, rc)"""
    X = np.asfortranarray(BASIS, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
    fid = np.ascontiguousarray(fold_id, dtype=np.int32)
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    n, p = X.shape
    out = np.zeros((len(alpha), n_folds))
    cnt = Counters()
    rc = lib().eben_cv_grid(_dp(X), n, p, _dp(y), fid.ctypes.data_as(C.POINTER(C.c_int32)), n_folds,
                            _dp(alpha), _dp(lam), len(alpha), 0 if prior == "gaussian" else 1,
                            1 if epis else 0, n_threads, _dp(out), C.byref(cnt))
    return out, cnt.asdict(), rc
