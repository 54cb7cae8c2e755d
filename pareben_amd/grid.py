"""Host-side counterparts of parEBEN's R helpers around the fit grid.

Same names, argument meaning and results as the R functions they mirror:

* ``GetLambdaMax`` / ``BuildGrid``  -- R/BuildGrid.R:5-52
* ``AssignToFolds``                 -- R/AssignToFolds.R:6-19 (``set.seed(1)`` + ``sample``)
* ``summarise_cv``                  -- the dplyr summary and first-minimum arg-min of
                                        R/CrossValidate.R:72-85 / :93-101

These are O(grid) or one O(n*p) pass and stay on the host, exactly as in the reference where
they run in the R master process; the fits they parameterise run on the GPU.
"""
import math
import numpy as np

from .rlang import RRandom, r_seq_by, r_sd


def _as_matrix(BASIS):
    X = np.asarray(BASIS, dtype=np.float64)
    if X.ndim != 2:
        raise ValueError("BASIS must be a 2-d matrix (rows = samples, columns = features)")
    return X


def _pairs_host(X, centred):
    """The pairwise pass of GetLambdaMax on the host (numpy): the checker of the device pass, and what
    ``GetLambdaMax(..., device=None)`` uses."""
    K = X.shape[1]
    best = -math.inf
    for i in range(K - 1):
        prod = X[:, i:i + 1] * X[:, i + 1:]
        nrm = np.sqrt(np.sum(prod * prod, axis=0))
        with np.errstate(divide="ignore", invalid="ignore"):
            c2 = (prod / nrm).T @ centred
        for c in c2:
            if c > best:
                best = float(c)
    return best


def _resolve_device(device):
    """"auto": GPU 0 when the library sees a device, else the numpy path (a box without a GPU can build a grid, not fit it)."""
    if device != "auto":
        return device
    try:
        from . import _lib
        return 0 if _lib.load().pareben_device_count() > 0 else None
    except Exception:
        return None


def GetLambdaMax(BASIS, Target, Epis="no", device="auto"):
    """R/BuildGrid.R:5-32.  max(log 1.1, max_j x_j.response/|x_j|); with Epis="yes" also all
    pairs x_i*x_j, correlated with the centred but *un-normalised* target.
    The O(n K^2) pairwise pass runs on GPU `device` (pareben_lambda_max_pairs); device = "auto" (default) takes GPU 0
    when there is one -- so a direct BuildGrid() and the grid inside CrossValidate() are the same numbers on a GPU box --
    and the numpy pass otherwise; None forces numpy (the checker of the device pass).  The two passes sum in different
    orders and can differ in the last bits: match grid cells by index, never by comparing lambda values."""
    device = _resolve_device(device)
    X = _as_matrix(BASIS)
    yv = np.asarray(Target, dtype=np.float64).reshape(-1)
    lam = math.log(1.1)
    centred = yv - yv.mean()
    response = centred / math.sqrt(float(np.sum(centred * centred)))
    norms = np.sqrt(np.sum(X * X, axis=0))
    with np.errstate(divide="ignore", invalid="ignore"):
        cor = (X / norms).T @ response
    for c in cor:                       # NaN (all-zero column) never compares greater, as in R
        if c > lam:
            lam = float(c)
    if Epis == "yes":
        if device is None:
            pair = _pairs_host(X, centred)
        else:
            from . import _lib
            pair = _lib.lambda_max_pairs(X, yv, device=device)
        if pair > lam:
            lam = float(pair)
    return lam


def BuildGrid(BASIS, Target, nFolds, Epis="no", nAlpha=20, nLambda=20, device="auto"):
    """R/BuildGrid.R:34-52.  Returns (alpha, lambda) arrays of the expanded grid, alpha fastest
    (``expand.grid(alpha = Alpha, lambda = Lambda)``).  nAlpha/nLambda default to the reference's
    hard-wired 20 x 20; other sizes are an extension: lambda keeps the
    10*lambda_max ... 0.001*10*lambda_max log range split in nLambda-1 steps, alpha is
    seq(1, by = -1/nAlpha)."""
    lambda_max = GetLambdaMax(BASIS, Target, Epis, device=device) * 10
    lambda_min = math.log(0.001 * lambda_max)
    step = (math.log(lambda_max) - lambda_min) / (nLambda - 1)
    Lambda = np.exp(r_seq_by(math.log(lambda_max), lambda_min, -step))
    if nAlpha == 20:
        Alpha = r_seq_by(1.0, 0.05, -0.05)
    else:
        Alpha = r_seq_by(1.0, 1.0 / nAlpha, -1.0 / nAlpha)
    alpha = np.tile(Alpha, len(Lambda))
    lam = np.repeat(Lambda, len(Alpha))
    return alpha, lam


def AssignToFolds(BASIS, nFolds=0, foldId=0, sample_kind="Rejection"):
    """R/AssignToFolds.R:6-19: ``set.seed(1)`` then a permutation of rep(1:nFolds, ...).
    ``sample_kind`` picks R's sampler generation ("Rejection": R >= 3.6 default; "Rounding":
    what R < 3.6 used, e.g. the authors' R 3.5.0 runs)."""
    N = _as_matrix(BASIS).shape[0]
    fid = np.atleast_1d(np.asarray(foldId))
    if fid.size == N:
        return fid.astype(np.int32)
    rng = RRandom(1, sample_kind)
    base = list(range(1, nFolds + 1)) * (N // nFolds)
    if N % nFolds != 0:
        base = base + list(range(1, N % nFolds + 1))
    return np.asarray(rng.sample(base), dtype=np.int32)


def summarise_cv(alpha, lam, fold_err, nFolds, prior="gaussian"):
    """group_by(alpha, lambda) %>% summarise(SE = sd/sqrt(nFolds), MSE = mean) with rows sorted
    alpha ascending then lambda ascending, then which.min -> first minimum in that order
    (R/CrossValidate.R:72-80).  For the binomial prior the summary column is
    Likelihood = -mean(logL) and the arg-min is taken on it (the evident intent; the reference
    indexes a non-existent MSE column there).
    Returns (alpha_sorted, lambda_sorted, SE, err, index_of_optimum)."""
    alpha = np.asarray(alpha, dtype=np.float64)
    lam = np.asarray(lam, dtype=np.float64)
    E = np.asarray(fold_err, dtype=np.float64).reshape(len(alpha), nFolds)
    order = np.lexsort((lam, alpha))
    se = np.array([r_sd(E[c]) / math.sqrt(nFolds) for c in order])
    mean = np.array([float(np.mean(E[c])) for c in order])
    err = mean if prior == "gaussian" else -mean
    idx = -1
    best = math.inf
    for i, v in enumerate(err):         # which.min: first minimum, NaN skipped
        if v < best:
            best = v
            idx = i
    if idx < 0:
        # R's which.min() would return integer(0) here and CrossValidate() would hand back empty optima; a table
        # without a single complete cell (every cell has a stopped fit, status bit 8) is an error, not an answer
        raise ValueError("no (alpha, lambda) cell has a complete set of fold scores: every cell holds a stopped fit")
    return alpha[order], lam[order], se, err, idx
