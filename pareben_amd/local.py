"""``LocalSearch`` -- counterpart of parEBEN's R/LocalSearch.R:6-140, what ``CrossValidate(search =
"local")`` returns: for every alpha (1, 0.95 ... 0.05) walk lambda from the largest value down and stop
as soon as the CV error exceeds the best one seen for that alpha plus its standard error.

The walk is sequential in (alpha, lambda) by construction, but every cell it can visit belongs to the
20 x 20 grid of the global search and a cell's result does not depend on the order of evaluation.  So the
whole grid is evaluated in one launch on the GPU (seconds) and the reference's early-stopping policy is
replayed over the table on the host: same visited cells, same returned object
(``CrossValidation`` nAlpha x 4, ``alpha.optimal``, ``lambda.optimal``, ``fullCV`` with one row per
visited cell and zero rows after them).

Differences, both on the reference's side: it draws the folds with an UNSEEDED ``sample()`` unless a
full-length ``foldId`` is passed (R/LocalSearch.R:13-20), so its result is not reproducible; here the
folds come from ``AssignToFolds`` (``set.seed(1)``, as in the global search) unless ``foldId`` is given.
The binomial prior is refused with the reference's own message (:137)."""
import math
import warnings

import numpy as np

from . import _lib
from .grid import BuildGrid, AssignToFolds
from .rlang import r_sd


def replay_local_search(alpha_desc, lam_desc, fold_err_of):
    """The early-stopping walk of R/LocalSearch.R:52-131 over a table.

    alpha_desc: alphas in visiting order (1 ... 0.05); lam_desc: lambdas, largest first;
    fold_err_of(ia, il) -> the nFolds held-out SSEs of cell (alpha_desc[ia], lam_desc[il]).
    Returns (MSEeachAlpha [nAlpha x 4], alpha_opt, lambda_opt, MSEcv [nAlpha*nLambda x 4], visited)."""
    n_a, n_l = len(alpha_desc), len(lam_desc)
    msecv = np.zeros((n_a * n_l, 4))
    each = np.zeros((n_a, 4))
    visited = []
    step = 0
    for ia, a in enumerate(alpha_desc):
        sse = np.full((n_l, 2), 1e10)                           # SSE1Alpha <- matrix(1e10, N_step, 2)
        for il, lam in enumerate(lam_desc):
            # which.min(SSE1Alpha[1:(i_s-1), 1]): for the first step R's 1:0 = c(1, 0) selects row 1 (still 1e10)
            upto = il if il >= 1 else 1
            mi = int(np.nanargmin(sse[:upto, 0]))              # which.min skips NA (a flagged fit scores NaN)
            previous = sse[mi, 0] + sse[mi, 1]
            e = np.asarray(fold_err_of(ia, il), dtype=np.float64)
            mean, se = float(np.mean(e)), r_sd(e) / math.sqrt(len(e))
            if not math.isfinite(mean):                           # a stopped fit (NaN score): the cell counts as "no improvement"
                mean, se = math.inf, 0.0                          # (R's which.min / if() would stop with an error on the NA)
            sse[il] = (mean, se)
            msecv[step] = (a, lam, mean, se)
            visited.append((ia, il))
            step += 1
            if mean - previous > 0:                              # early stop for this alpha
                break
        idx = int(np.nanargmin(sse[:, 0]))
        each[ia] = (a, lam_desc[idx], sse[idx, 0], sse[idx, 1])
    best = int(np.nanargmin(each[:, 2]))
    return each, float(each[best, 0]), float(each[best, 1]), msecv, visited


def LocalSearch(BASIS, Target, nFolds, Epis="no", foldId=0, prior="gaussian", device=0, sample_kind="Rejection"):
    """LocalSearch(BASIS, Target, nFolds, Epis = "no", foldId = 0, prior = "gaussian") ->
    dict(CrossValidation, alpha.optimal, lambda.optimal, fullCV) as in R/LocalSearch.R:132-134."""
    if prior != "gaussian":
        raise ValueError("For the binomial prior, please use the global search.")
    X = np.asarray(BASIS, dtype=np.float64)
    y = np.asarray(Target, dtype=np.float64).reshape(-1)
    alpha, lam = BuildGrid(X, y, nFolds, Epis, device=device)    # same lambda / alpha values as :21-50
    folds = AssignToFolds(X, nFolds, foldId, sample_kind=sample_kind)
    with _lib.Context(X, y, folds, nFolds, prior="gaussian", epis=(Epis == "yes"), device=device) as ctx:
        fold_err, status, _ = ctx.run(alpha, lam)
    if np.any((status & _lib.ST_ABORT) != 0):
        warnings.warn("%d fits were stopped early (status bit 8) and score NaN; their cells end the walk of their alpha"
                      % int(np.sum((status & _lib.ST_ABORT) != 0)), RuntimeWarning)
    a_desc = np.unique(alpha)[::-1]
    l_desc = np.unique(lam)[::-1]
    cell = {(float(a), float(l)): i for i, (a, l) in enumerate(zip(alpha, lam))}
    each, a_opt, l_opt, msecv, _ = replay_local_search(
        a_desc, l_desc, lambda ia, il: fold_err[cell[(float(a_desc[ia]), float(l_desc[il]))]])
    return {"CrossValidation": each, "alpha.optimal": a_opt, "lambda.optimal": l_opt, "fullCV": msecv}
