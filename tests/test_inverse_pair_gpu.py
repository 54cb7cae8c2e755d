"""The forms of the blocked inverse (gm_dev.h) against each other: the register form (the whole triangle in the waves' registers,
up to 288 columns), two pivot blocks per trip through memory (up to 512 columns) and one block per trip (PAREBEN_INV_PAIR = 3
default | 1 | 0) run the same chain of operations per element, so every fold SSE, status word and event counter of a grid must
be equal to the last bit.  Synthetic Gaussian design of BASELINE configs[1]'s shape (n = 1000, p = 10000, 5 folds, 20 alpha x
20 lambda = 2000 fits; active sets up to ~650 columns: inversions on every path, odd and even numbers of pivot blocks)."""
import numpy as np
import pytest

import pareben_amd
from pareben_amd.grid import AssignToFolds, BuildGrid
from pareben_amd.synth import synthetic_gaussian

pytestmark = pytest.mark.gpu


def test_forms_of_the_inverse_are_bit_identical(monkeypatch):
    X, y, _, _ = synthetic_gaussian(1000, 10000)
    alpha, lam = BuildGrid(X, y, 5, nAlpha=20, nLambda=20)
    fid = AssignToFolds(X, 5)
    out = {}
    with pareben_amd.Context(X, y, fid, 5) as ctx:
        for mode in ("3", "1", "0"):
            monkeypatch.setenv("PAREBEN_INV_PAIR", mode)
            out[mode] = ctx.run(alpha, lam)
            print("PAREBEN_INV_PAIR=%s" % mode, ctx.last_timing())
    E3, st3, c3 = out["3"]
    assert np.all(st3 & 8 == 0)
    assert c3[..., 10].max() > 512 and (c3[..., 10] > 48).sum() > 100      # m_max: every form was exercised
    for mode in ("1", "0"):
        E, st, c = out[mode]
        assert np.array_equal(st3, st) and np.array_equal(c3, c), mode
        assert np.array_equal(E3, E), (mode, float(np.nanmax(np.abs(E3 - E) / np.abs(E))))
