"""``EBelasticNet.Gaussian`` / ``EBelasticNet.Binomial`` -- counterparts of the EBEN functions that
parEBEN users call right after ``CrossValidate`` to refit at (alpha*, lambda*)
(tests/CrossValidate-test.R:23, README.md:88-94 of the reference).  Same arguments, same returned
list (EBEN_orig/R/EBelasticNet.Gaussian.R:1-101, EBelasticNet.Binomial.R:1-93); the fit itself runs on
the GPU through the per-fit C-ABI entries, the M x 6 ``weight`` table (t and p columns) is host work."""
import math

import numpy as np

from . import _lib


def _betacf(a, b, x):
    """Continued fraction of the incomplete beta function (modified Lentz)."""
    tiny = 1e-300
    qab, qap, qam = a + b, a + 1.0, a - 1.0
    c, d = 1.0, 1.0 - qab * x / qap
    if abs(d) < tiny:
        d = tiny
    d = 1.0 / d
    h = d
    for m in range(1, 1000):
        m2 = 2 * m
        aa = m * (b - m) * x / ((qam + m2) * (a + m2))
        d = 1.0 + aa * d
        if abs(d) < tiny:
            d = tiny
        c = 1.0 + aa / c
        if abs(c) < tiny:
            c = tiny
        d = 1.0 / d
        h *= d * c
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2))
        d = 1.0 + aa * d
        if abs(d) < tiny:
            d = tiny
        c = 1.0 + aa / c
        if abs(c) < tiny:
            c = tiny
        d = 1.0 / d
        delta = d * c
        h *= delta
        if abs(delta - 1.0) < 1e-16:
            break
    return h


def _betainc(a, b, x, xc):
    """Regularised incomplete beta I_x(a, b); xc = 1 - x supplied by the caller without cancellation."""
    if x <= 0.0:
        return 0.0
    if xc <= 0.0:
        return 1.0
    if b == 0.5 and a >= 100.0:      # Gamma(a+1/2)/Gamma(a) by its asymptotic series: no lgamma cancellation
        r = 1.0 / a
        ratio = math.sqrt(a) * (1.0 + r * (-1.0 / 8 + r * (1.0 / 128 + r * (5.0 / 1024 + r * (-21.0 / 32768 + r * (-399.0 / 262144 + r * (869.0 / 4194304)))))))
        lbeta_inv = math.log(ratio / math.sqrt(math.pi))
    else:
        lbeta_inv = math.lgamma(a + b) - math.lgamma(a) - math.lgamma(b)
    lx = math.log1p(-xc) if xc < 0.5 else math.log(x)
    lxc = math.log1p(-x) if x < 0.5 else math.log(xc)
    front = math.exp(lbeta_inv + a * lx + b * lxc)
    if x < (a + 1.0) / (a + b + 2.0):
        return front * _betacf(a, b, x) / a
    return 1.0 - front * _betacf(b, a, xc) / b


def pt(t, df):
    """Student-t distribution function, R's pt(t, df)."""
    t = float(t)
    if math.isnan(t):
        return float("nan")
    den = df + t * t
    tail = 0.5 * _betainc(0.5 * df, 0.5, df / den, t * t / den)          # P(T > |t|)
    return 1.0 - tail if t >= 0 else tail


def _weight_table(Beta, keep_col, N, epis):
    """rows kept / ordered as EBelasticNet.Gaussian.R:55-83, then t = |b|/(sqrt(v)+1e-20) and
    p = 2(1 - pt(t, N-1)) (:84-98)."""
    keep = np.nonzero(Beta[:, keep_col] != 0)[0]
    ncol = Beta.shape[1]
    Blup = np.zeros((1, ncol)) if len(keep) == 0 else Beta[keep, :].copy()
    if epis:
        main = Blup[Blup[:, 0] == Blup[:, 1]]
        pair = Blup[Blup[:, 0] != Blup[:, 1]]
        main = main[np.argsort(main[:, 0], kind="stable")]
        pair = pair[np.argsort(pair[:, 0], kind="stable")]
        Blup = np.vstack([main, pair])
    Blup = Blup[:, :4]
    t = np.abs(Blup[:, 2]) / (np.sqrt(Blup[:, 3]) + 1e-20)
    p = np.array([2.0 * (1.0 - pt(v, N - 1)) for v in t])
    return np.column_stack([Blup, t, p])


def Gaussian(BASIS, Target, lambda_, alpha, Epis="no", verbose=0, device=0):
    """EBelasticNet.Gaussian(BASIS, Target, lambda, alpha, Epis = "no", verbose = 0) ->
    dict(weight M x 6 [locus1, locus2, beta, posterior variance, t-value, p-value], WaldScore,
    Intercept, residVar, lambda, alpha)."""
    X = np.asarray(BASIS, dtype=np.float64)
    epis = Epis == "yes"
    r = _lib.fit_gaussian(X, Target, lambda_, alpha, device=device, epis=epis)
    weight = _weight_table(r["Beta"], 4 if epis else 2, X.shape[0], epis)
    return {"weight": weight, "WaldScore": r["wald"], "Intercept": r["intercept"], "residVar": r["residual"],
            "lambda": lambda_, "alpha": alpha}


def Binomial(BASIS, Target, lambda_, alpha, Epis="no", verbose=0, device=0):
    """EBelasticNet.Binomial(BASIS, Target, lambda, alpha, Epis = "no", verbose = 0) ->
    dict(weight M x 6, logLikelihood, WaldScore, Intercept[2], lambda, alpha)."""
    X = np.asarray(BASIS, dtype=np.float64)
    epis = Epis == "yes"
    r = _lib.fit_binomial(X, Target, lambda_, alpha, device=device, epis=epis)
    weight = _weight_table(r["Beta"], 2, X.shape[0], epis)      # :47-67: rows with a non-zero effect; Epis: mains then pairs, each by locus1
    return {"weight": weight, "logLikelihood": r["logLikelihood"], "WaldScore": r["wald"], "Intercept": r["intercept"],
            "lambda": lambda_, "alpha": alpha}


class _Namespace:
    """so that calls read like the reference's: EBelasticNet.Gaussian(...), EBelasticNet.Binomial(...)"""
    Gaussian = staticmethod(Gaussian)
    Binomial = staticmethod(Binomial)


EBelasticNet = _Namespace()
