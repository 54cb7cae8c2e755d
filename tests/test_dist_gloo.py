"""N > 1 path on CPU: two gloo ranks shard the cell list, each evaluates its own cells (with the
oracle standing in for the GPU), one all-gather reassembles the table; the result must equal the
single-process table bit for bit and every rank must end up with the full table."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib
    from pareben_amd.grid import BuildGrid, AssignToFolds
    from pareben_amd.dist import shard_cells, all_gather_cells
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X = np.load(os.path.join(ROOT, "tests", "golden", "BASIS.npy")).astype(np.float64)[:40, :30]
    y = np.load(os.path.join(ROOT, "tests", "golden", "y.npy"))[:40]
    fid = AssignToFolds(X, 3)
    alpha, lam = BuildGrid(X, y, 3)
    alpha, lam = alpha[:37], lam[:37]                      # odd count: ranks get 19 / 18 cells
    mine = shard_cells(alpha, lam, rank, world)
    E, _, rc = oracle_lib.cv_grid(X, y, fid, 3, alpha[mine], lam[mine], n_threads=1)
    st = np.full(E.shape, rank, dtype=np.int32)
    full, status = all_gather_cells(mine, E, st, len(alpha), 3)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), full=full, status=status, mine=mine)
    dist.destroy_process_group()


def test_two_rank_gloo_all_gather(tmp_path, oracle):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npz"); r1 = np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["full"], r1["full"]) and np.array_equal(r0["status"], r1["status"])
    assert len(r0["mine"]) + len(r1["mine"]) == 37 and not set(r0["mine"]) & set(r1["mine"])
    from pareben_amd.grid import BuildGrid, AssignToFolds
    X = np.load(os.path.join(ROOT, "tests", "golden", "BASIS.npy")).astype(np.float64)[:40, :30]
    y = np.load(os.path.join(ROOT, "tests", "golden", "y.npy"))[:40]
    fid = AssignToFolds(X, 3)
    alpha, lam = BuildGrid(X, y, 3)
    E, _, rc = oracle.cv_grid(X, y, fid, 3, alpha[:37], lam[:37], n_threads=1)
    assert np.array_equal(r0["full"], E)                   # sharding does not change a single bit
    assert np.array_equal(r0["status"][r0["mine"]], np.zeros((len(r0["mine"]), 3), dtype=np.int32))
    assert np.all(r0["status"][r1["mine"]] == 1)


def test_shard_cells_is_a_partition_and_cost_interleaved():
    from pareben_amd.dist import shard_cells
    rng = np.random.default_rng(0)
    alpha = rng.random(101); lam = rng.random(101)
    parts = [shard_cells(alpha, lam, r, 4) for r in range(4)]
    allc = np.concatenate(parts)
    assert sorted(allc.tolist()) == list(range(101))
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    # every rank sees the whole lambda range (cost mix), not a contiguous slab
    for p in parts:
        assert lam[p].min() < 0.1 and lam[p].max() > 0.9


def test_shard_cells_spreads_every_alpha_column_on_a_rectangular_grid():
    """20 alphas x 100 lambdas over 2, 4, 5, 8 ranks: every rank gets its share of every alpha column (the heavy fits sit on one
    of them), within one cell; counts per rank within two cells of each other."""
    from pareben_amd.dist import shard_cells
    alpha = np.tile(np.linspace(1, 0.05, 20), 100)
    lam = np.repeat(np.exp(np.linspace(2, -5, 100)), 20)
    for w in (2, 4, 5, 8):
        parts = [shard_cells(alpha, lam, r, w) for r in range(w)]
        assert sorted(np.concatenate(parts).tolist()) == list(range(2000))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 2
        for a in np.unique(alpha):
            per = [int((alpha[p] == a).sum()) for p in parts]
            assert max(per) - min(per) <= 1, (w, a, per)
