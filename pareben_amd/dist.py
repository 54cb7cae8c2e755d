"""One process per GPU: split of the (alpha, lambda) grid across ranks and the single collective of
the path -- an all-gather of each rank's fold-error slice (RCCL over xGMI when the backend is
"nccl"; "gloo" on CPU for tests).  Cells are independent, so this is the only exchange
(SURVEY.md 8(e)); results are bit-identical for any number of ranks."""
import numpy as np


def shard_cells(alpha, lam, rank, world_size):
    """Cell indices of `rank`: cost-sorted (small lambda first), then dealt round-robin so every
    GPU receives the same mix of cheap and expensive cells."""
    order = np.lexsort((np.asarray(alpha), np.asarray(lam)))
    return order[rank::world_size]


def all_gather_cells(mine, err_local, st_local, n_cells, n_folds, device=None):
    """torch.distributed all-gather of (cell index, fold errors, status) slices -> full tables on
    every rank.  Slices are padded to the same length (ranks may differ by one cell)."""
    import torch
    import torch.distributed as dist
    ws = dist.get_world_size()
    per = (n_cells + ws - 1) // ws
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    buf = torch.full((per, n_folds + 1 + n_folds), float("nan"), dtype=torch.float64)
    k = len(mine)
    buf[:k, 0] = torch.from_numpy(np.asarray(mine, dtype=np.float64))
    buf[:k, 1:1 + n_folds] = torch.from_numpy(np.ascontiguousarray(err_local))
    buf[:k, 1 + n_folds:] = torch.from_numpy(np.ascontiguousarray(st_local).astype(np.float64))
    buf = buf.to(dev)
    out = torch.empty((ws * per, buf.shape[1]), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy()
    fold_err = np.full((n_cells, n_folds), np.nan)
    status = np.full((n_cells, n_folds), -1, dtype=np.int32)
    valid = ~np.isnan(out[:, 0])
    idx = out[valid, 0].astype(np.int64)
    fold_err[idx] = out[valid, 1:1 + n_folds]
    status[idx] = out[valid, 1 + n_folds:].astype(np.int32)
    return fold_err, status
