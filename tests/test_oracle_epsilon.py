"""The oracle's KDTreeMatcher{epsilon} search (orc_kdtree_knn_eps: libnabo's `new_rd * (1 + epsilon)^2 < heap.headValue()` on the oracle's
tree) against brute force: the (1 + epsilon) guarantee rank by rank, real points at their real distances; epsilon 0 is the exact search."""
import numpy as np
import pytest


@pytest.mark.parametrize("k", [1, 6])
@pytest.mark.parametrize("eps", [0.0, 0.5, 1.0, 4.0])
def test_oracle_epsilon_search_keeps_the_guarantee(oracle, k, eps):
    rng = np.random.default_rng(11)
    m = np.ones((20000, 4), dtype=np.float32); m[:, :3] = rng.uniform(-10, 10, (20000, 3)).astype(np.float32)
    m[:5000, 2] = 0.0                                             # a plane: uneven density
    q = np.ones((1500, 4), dtype=np.float32); q[:, :3] = rng.uniform(-12, 12, (1500, 3)).astype(np.float32)
    ex_ids, ex_d2 = oracle.knn(m, q, k=k, brute=True)
    ids, d2 = oracle.knn(m, q, k=k, nthreads=4, epsilon=eps) if eps > 0 else oracle.knn(m, q, k=k, nthreads=4)
    if eps == 0.0:
        assert np.array_equal(ids, ex_ids) and np.array_equal(d2, ex_d2)
        return
    assert (ids >= 0).all()
    dd = ((q[:, None, :3].astype(np.float64) - m[ids, :3].astype(np.float64)) ** 2).sum(-1)
    assert np.allclose(d2, dd, rtol=4e-6, atol=1e-12)
    assert (d2[:, 1:] >= d2[:, :-1]).all()
    assert (d2 >= ex_d2).all() and (d2 <= ex_d2.astype(np.float64) * (1 + eps) ** 2 * (1 + 1e-5) + 1e-12).all()
    assert (d2 > ex_d2).any() or eps < 1.0                         # (the pruning really bites at large epsilon)
