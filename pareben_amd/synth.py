"""Synthetic Gaussian design of BASELINE.json configs 2 and 5: X_ij ~ N(0,1) iid,
20 causal columns with N(0,1) effects, unit-variance noise.  Deterministic and platform
independent: splitmix64 -> Box-Muller, column-major fill order, so any host language can regenerate
the same matrix from (n, p, seed)."""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(n, seed):
    """first n outputs of splitmix64 seeded with `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _normals(n, seed):
    m = (n + 1) // 2
    u = _splitmix64(2 * m, seed)
    u1 = ((u[0::2] >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)   # (0,1]
    u2 = (u[1::2] >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)           # [0,1)
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.empty(2 * m)
    z[0::2] = r * np.cos(2.0 * np.pi * u2)
    z[1::2] = r * np.sin(2.0 * np.pi * u2)
    return z[:n]


def synthetic_gaussian(n, p, n_causal=20, seed=20251004):
    """-> (X [n x p, Fortran order], y [n], causal_idx, causal_beta)"""
    X = _normals(n * p, seed).reshape((n, p), order="F")
    aux = _splitmix64(n_causal, seed + 1)
    idx = np.unique((aux % np.uint64(p)).astype(np.int64))
    beta = _normals(len(idx), seed + 2)
    y = X[:, idx] @ beta + _normals(n, seed + 3)
    return np.asfortranarray(X), y, idx, beta
