"""KDTreeMatcher{epsilon} as an APPROXIMATE search (icpmi_config::epsilon_approx; the shipped examples/config.yaml:56-60 asks for `knn: 6,
epsilon: 1`): libnabo prunes a branch when `new_rd * (1 + epsilon)^2 >= heap.headValue()`; WHICH answer comes back depends on the kd-tree's
traversal order, so the common ground with libnabo (and with the oracle's own epsilon search) is the guarantee, which is what is tested:
every returned distance d_j <= (1 + epsilon) x the exact j-th distance, every returned pair is a real map point at its real distance, rows
ascending without repeats.  epsilon_approx = 0 keeps the exact search bit for bit whatever `epsilon` says."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


def centred(c, mean):
    o = c.copy(); o[:, :3] = o[:, :3] - mean[:3]; return o


def check_epsilon_answer(mapc, q, ids, d2, ex_d2, eps):
    n, k = ids.shape
    fin = np.isfinite(d2)
    assert ((ids >= 0) == fin).all()
    # real points at their real distances (float32 fmaf chain vs float64: a few ulp)
    ii = np.where(fin, ids, 0)
    dd = ((q[:, None, :3].astype(np.float64) - mapc[ii, :3].astype(np.float64)) ** 2).sum(-1)
    assert np.allclose(np.where(fin, d2, 0), np.where(fin, dd, 0), rtol=4e-6, atol=1e-12)
    # ascending, no point twice in a row
    d2i = np.where(fin, d2, np.inf)
    assert (d2i[:, 1:] >= d2i[:, :-1]).all()
    srt = np.sort(np.where(fin, ids, -np.arange(1, k + 1)[None, :]), axis=1)
    assert (np.diff(srt, axis=1) != 0).all()
    # the guarantee, rank by rank (squared distances: (1 + eps)^2); an approximate row never holds fewer points than the exact one under
    # the same radius unless the missing ones are beyond the radius / (1 + eps) ... libnabo makes no promise there: only compare where both are finite
    both = fin & np.isfinite(ex_d2)
    lim = ex_d2.astype(np.float64) * (1.0 + eps) ** 2 * (1 + 1e-5) + 1e-12
    assert (d2[both] <= lim[both]).all()
    assert (d2[both] >= ex_d2[both]).all()          # and never below the exact j-th distance


@pytest.mark.parametrize("k", [1, 6, 10])
@pytest.mark.parametrize("eps", [0.5, 1.0, 3.0])
def test_epsilon_knn_keeps_libnabos_guarantee(amd, oracle, small_scene, k, eps):
    m = small_scene["map"]
    exact = amd.ICPSequence(minimizer=0, knn=k, epsilon=eps)                      # epsilon alone: the exact search
    approx = amd.ICPSequence(minimizer=0, knn=k, epsilon=eps, epsilon_approx=1)
    assert exact.setMap(m) and approx.setMap(m)
    mean = exact.getMapMean()
    mapc, q = centred(m, mean), centred(small_scene["scan"], mean)
    rng = np.random.default_rng(3)
    far = q[:500].copy(); far[:, :3] += rng.normal(0, 1.5, (500, 3)).astype(np.float32)     # queries off the surfaces: wide searches
    q = np.concatenate([q, far])
    for md in (math.inf, 2.0):
        ex_ids, ex_d2 = exact.knn(q, k=k, max_dist=md)
        r_ids, r_d2 = oracle.knn(mapc, q, k=k, max_dist=md, nthreads=8)
        assert np.array_equal(ex_ids, r_ids) and np.array_equal(ex_d2, r_d2)   # (epsilon without epsilon_approx changes nothing)
        ids, d2 = approx.knn(q, k=k, max_dist=md)
        check_epsilon_answer(mapc, q, ids, d2, ex_d2, eps)
        # the oracle's epsilon search (libnabo's rule on the oracle's tree) is another valid answer: held to the same guarantee
        o_ids, o_d2 = oracle.knn(mapc, q, k=k, max_dist=md, nthreads=8, epsilon=eps)
        check_epsilon_answer(mapc, q, o_ids, o_d2, ex_d2, eps)


def test_epsilon_registration_stays_on_the_exact_pose(amd, mid_scene):
    """the shipped matcher (knn 6, epsilon 1, maxDist 2) with the approximate search against the exact one: same pose within the
    north-star tolerance on the benchmark scene, and the approximate run is what a second run reproduces (deterministic)"""
    sc = mid_scene
    kw = dict(minimizer=2, knn=6, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20, epsilon=1.0)
    poses = {}
    for name, extra in (("exact", {}), ("approx", {"epsilon_approx": 1}), ("approx2", {"epsilon_approx": 1})):
        icp = amd.ICPSequence(**kw, **extra)
        assert icp.setMap(sc["map"], sc["normals"])
        poses[name] = icp(sc["scan"])
    dt, dr = amd.synth.pose_error(poses["approx"], poses["exact"])
    gt_t, gt_r = amd.synth.pose_error(poses["approx"], sc["T_gt"])
    ex_t, ex_r = amd.synth.pose_error(poses["exact"], sc["T_gt"])
    assert np.array_equal(poses["approx"], poses["approx2"])
    assert dt <= 1e-3 and dr <= 1e-3, (dt, dr)
    assert gt_t <= ex_t + 1e-3 and gt_r <= ex_r + 1e-3
