"""search = "local" (R/LocalSearch.R): the early-stopping walk replayed over the grid table."""
import numpy as np
import pytest

import pareben_amd
from pareben_amd.local import replay_local_search


def test_replay_policy_on_a_table():
    """Hand-checkable table: 2 alphas x 5 lambdas, 2 folds (SE = |a - b| / 2)."""
    lam = np.array([16.0, 8.0, 4.0, 2.0, 1.0])
    alph = np.array([1.0, 0.5])
    table = {
        (0, 0): (10.0, 10.0), (0, 1): (8.0, 8.0), (0, 2): (9.0, 7.0),      # mean 8, SE 1: not > 8 + 0 -> go on
        (0, 3): (9.5, 8.7),                                                # mean 9.1 > min(8) + its SE (0) -> stop
        (0, 4): (0.0, 0.0),                                                # never visited
        (1, 0): (5.0, 7.0),                                                # mean 6, SE 1
        (1, 1): (6.5, 7.1),                                                # 6.8 <= 6 + 1 -> go on
        (1, 2): (7.2, 7.0),                                                # 7.1 > 7 -> stop
        (1, 3): (0.0, 0.0), (1, 4): (0.0, 0.0),
    }
    each, a_opt, l_opt, msecv, visited = replay_local_search(alph, lam, lambda ia, il: table[(ia, il)])
    assert visited == [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1), (1, 2)]
    assert np.allclose(each[0], [1.0, 8.0, 8.0, 0.0]) and np.allclose(each[1], [0.5, 16.0, 6.0, 1.0])
    assert (a_opt, l_opt) == (0.5, 16.0)
    assert np.allclose(msecv[3], [1.0, 2.0, 9.1, 0.4]) and np.all(msecv[7:] == 0)
    # first step of an alpha is never a stop (R's 1:0 indexing leaves 1e10 + 1e10 as the bar)
    each2, _, _, _, v2 = replay_local_search(alph[:1], lam[:2], lambda ia, il: (1e9, 1e9))
    assert v2 == [(0, 0), (0, 1)]


def test_binomial_is_refused():
    with pytest.raises(ValueError, match="global search"):
        pareben_amd.LocalSearch(np.zeros((4, 2)), np.zeros(4), 2, prior="binomial")


@pytest.mark.gpu
def test_local_search_vs_replay_over_golden_table(golden):
    """CrossValidate(search = "local") on the reference's own test case = the policy replayed over the
    oracle's table of that case (fold SSEs agree to 1e-9, so every stop decision is the same)."""
    X, y = golden.BASIS[:50, :100], golden.y[:50]
    g = golden.config1
    out = pareben_amd.CrossValidate(X, y, nFolds=3, Epis="no", prior="gaussian", search="local")
    a_desc = np.unique(g["alpha"])[::-1]
    l_desc = np.unique(g["lam"])[::-1]
    cell = {(float(a), float(l)): i for i, (a, l) in enumerate(zip(g["alpha"], g["lam"]))}
    each, a_opt, l_opt, msecv, visited = replay_local_search(
        a_desc, l_desc, lambda ia, il: g["fold_err"][cell[(float(a_desc[ia]), float(l_desc[il]))]])
    assert out["alpha.optimal"] == a_opt and out["lambda.optimal"] == l_opt
    assert out["CrossValidation"].shape == (20, 4) and out["fullCV"].shape == (400, 4)
    assert np.array_equal(out["fullCV"][:, :2], msecv[:, :2])                    # same visited cells in the same order
    assert np.allclose(out["fullCV"][:, 2:], msecv[:, 2:], rtol=1e-8, atol=0)
    assert np.allclose(out["CrossValidation"], each, rtol=1e-8, atol=0)
    assert 20 <= len(visited) < 400                                              # the walk does stop early
