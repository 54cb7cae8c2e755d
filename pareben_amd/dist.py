"""One process per GPU: split of the (alpha, lambda) grid across ranks and the single collective of
the path -- an all-gather of each rank's fold-error slice (RCCL over xGMI when the backend is
"nccl"; "gloo" on CPU for tests).  Cells are independent, so this is the only exchange
(SURVEY.md 8(e)); results are bit-identical for any number of ranks."""
import numpy as np


def shard_cells(alpha, lam, rank, world_size):
    """Cell indices of `rank`: cells sorted by (lambda, alpha) and dealt round-robin, so every GPU receives the same mix of
    lambdas -- and, on a rectangular grid (the same number of alphas at every lambda), of alphas too: a plain
    `order[rank::world_size]` deal hands a rank the same alpha columns at every lambda whenever the shift per lambda row,
    nAlpha mod world_size, shares a factor with the world size (20 alphas over 4 GPUs: the alpha = 1 ridge, where the heavy
    fits are, went to one rank; measured rank shares of BASELINE configs[1] 1.91 / 2.51 / 1.92 / 2.42 s).  So the deal is
    rotated by t further ranks per lambda row, t the smallest number that makes the row shift coprime to the world size.
    Which rank runs a cell never changes its result."""
    from math import gcd
    lam = np.asarray(lam)
    order = np.lexsort((np.asarray(alpha), lam))
    pos = np.arange(len(order))
    row = np.zeros(len(order), dtype=np.int64)
    t = 0
    if len(order):
        ls = lam[order]
        row = np.concatenate(([0], np.cumsum(ls[1:] != ls[:-1])))
        sizes = np.bincount(row)
        if world_size > 1 and len(sizes) > 1 and np.all(sizes == sizes[0]):
            while gcd(int(sizes[0] + t) % world_size, world_size) != 1:
                t += 1
    owner = (pos + t * row) % world_size
    return order[owner == rank]


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
