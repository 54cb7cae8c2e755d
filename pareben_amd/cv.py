"""``CrossValidate`` -- drop-in counterpart of parEBEN's exported R function
(R/CrossValidate.R:61-117) for the global grid search: same arguments, same returned object
(``Results.Detail``, ``Results.Summary``, ``lambda.optimal``, ``alpha.optimal``), with the
nFolds x alpha x lambda fits evaluated by the HIP library instead of a foreach backend."""
import warnings

import numpy as np

from . import _lib
from .grid import BuildGrid, AssignToFolds, summarise_cv

try:                                    # data.frame stand-in when pandas is importable
    import pandas as _pd
except Exception:                       # pragma: no cover
    _pd = None


def _frame(cols):
    if _pd is not None:
        return _pd.DataFrame(cols)
    return {k: np.asarray(v) for k, v in cols.items()}


def CrossValidate(BASIS, Target, nFolds, foldId=0, Epis="no", prior="gaussian", search="global",
                  nAlpha=20, nLambda=20, nGPU=1, device=0, sample_kind="Rejection", rank=0, world_size=1,
                  gather=None, return_stats=False):
    """Hyper-parameter sweep + cross-validation of the empirical Bayesian elastic net.

    BASIS, Target, nFolds, foldId, Epis, prior, search: as in R/CrossValidate.R:61.  Like the
    reference's global search, folds always come from ``AssignToFolds`` (``set.seed(1)``); the
    ``foldId`` argument is accepted and ignored there (R/TestModel.R:9) -- here a
    full-length ``foldId`` is honoured, anything else falls back to ``AssignToFolds``.

    Extensions (trailing, optional): nAlpha/nLambda grid sizes (reference: fixed 20 x 20); nGPU: devices this ONE
    process spreads the grid over through the C ABI (pareben_cv_grid_multi: a host thread per device and one RCCL
    all-gather; 0 = all visible; the R drop-in's route); device, sample_kind (R's sampler generation for the folds);
    and rank/world_size/gather for the one-process-per-GPU split (pareben_amd.dist, what bench.py / torchrun use).
    """
    if search != "global":                  # R/CrossValidate.R:112-115
        from .local import LocalSearch
        return LocalSearch(BASIS, Target, nFolds, Epis, foldId, prior, device=device, sample_kind=sample_kind)
    if prior not in ("gaussian", "binomial"):
        raise ValueError('prior must be "gaussian" or "binomial"')
    X = np.asarray(BASIS, dtype=np.float64)
    y = np.asarray(Target, dtype=np.float64).reshape(-1)
    alpha, lam = BuildGrid(X, y, nFolds, Epis, nAlpha=nAlpha, nLambda=nLambda, device=device)
    folds = AssignToFolds(X, nFolds, foldId, sample_kind=sample_kind)
    n_cells = len(alpha)

    # cells of this rank: interleaved over the cost-sorted order so every GPU gets the same mix
    order = np.lexsort((alpha, lam))
    mine = order[rank::world_size] if world_size > 1 else np.arange(n_cells)
    stats = {}
    if nGPU != 1 and world_size == 1:
        err_local, st_local, cnt_local = _lib.cv_grid_multi(X, y, folds, nFolds, alpha, lam, prior=prior, epis=(Epis == "yes"), n_gpu=nGPU)
    else:
        with _lib.Context(X, y, folds, nFolds, prior=prior, epis=(Epis == "yes"), device=device) as ctx:
            err_local, st_local, cnt_local = ctx.run(alpha[mine], lam[mine])
            stats["timing"] = ctx.last_timing()
            stats["launch"] = ctx.launch_info()
    if world_size > 1:
        if gather is None:
            raise ValueError("world_size > 1 needs a gather callable (see pareben_amd.dist.all_gather_cells)")
        fold_err, status = gather(mine, err_local, st_local, n_cells, nFolds)
    else:
        fold_err, status = err_local, st_local
    stats["status"] = status
    stats["counters"] = cnt_local
    n_stopped = int(np.sum((status & _lib.ST_ABORT) != 0))
    n_flagged = int(np.sum((status & _lib.ST_OVERFLOW) != 0))
    stats["stopped_fits"], stats["fits_past_reference_capacity"] = n_stopped, n_flagged
    if n_stopped:
        # a stopped fit (workspace capacity, non-SPD Hessian: states the reference leaves undefined) scores NaN; its cell
        # is left out of the arg-min like an NA in R's which.min -- say so instead of doing it silently
        warnings.warn("%d of %d fits were stopped early (status bit 8) and score NaN; %d cell(s) are excluded from the arg-min"
                      % (n_stopped, status.size, int(np.sum(np.any((status & _lib.ST_ABORT) != 0, axis=1)))), RuntimeWarning)

    col = "MSE" if prior == "gaussian" else "logL"
    detail = _frame({
        "foldId": np.tile(np.arange(1, nFolds + 1), n_cells),
        "alpha": np.repeat(alpha, nFolds),
        "lambda": np.repeat(lam, nFolds),
        col: fold_err.reshape(-1),
    })
    a_s, l_s, se, err, idx = summarise_cv(alpha, lam, fold_err, nFolds, prior)
    summary = _frame({"alpha": a_s, "lambda": l_s, "SE": se,
                      ("MSE" if prior == "gaussian" else "Likelihood"): err})
    out = {
        "Results.Detail": detail,
        "Results.Summary": summary,
        "lambda.optimal": float(l_s[idx]),
        "alpha.optimal": float(a_s[idx]),
    }
    if return_stats:
        out["stats"] = stats
    return out
