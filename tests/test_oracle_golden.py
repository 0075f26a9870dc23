"""CPU tests (no GPU): the oracle (oracle/liboracle.so, a C restatement of the libpointmatcher /
libnabo path) against the committed numpy / scipy golden vectors and against ground truth by
construction.  This is what pins the oracle in the absence of the reference binary (parity unpinned,
SURVEY.md 8c)."""
import math
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "numpy_scipy_vectors.npz"))


def test_transform_matches_numpy(oracle, gold):
    out = oracle.transform(gold["xf_T"], gold["knn_qry"])
    np.testing.assert_allclose(out[:, :3], gold["xf_out"], rtol=0, atol=2e-5)
    assert np.array_equal(out[:, 3], np.ones(out.shape[0], dtype=np.float32))
    # identity is exact
    assert np.array_equal(oracle.transform(np.eye(4), gold["knn_qry"]), gold["knn_qry"])


@pytest.mark.parametrize("k", [1, 6])
@pytest.mark.parametrize("radius", [math.inf, 2.0])
def test_knn_matches_scipy_and_bruteforce(oracle, gold, k, radius):
    ref, qry = gold["knn_ref"], gold["knn_qry"]
    ids, d2 = oracle.knn(ref, qry, k=k, max_dist=radius)
    bids, bd2 = oracle.knn(ref, qry, k=k, max_dist=radius, brute=True)
    assert np.array_equal(ids, bids) and np.array_equal(d2, bd2)  # kd-tree == brute force, bit for bit
    tag = f"knn_k{k}" + ("_r2" if radius == 2.0 else "")
    gids, gd = gold[tag + "_ids"], gold[tag + "_d"]
    finite = np.isfinite(gd)
    # scipy uses float64 distances: compare values, and ids wherever the neighbour is unambiguous
    np.testing.assert_allclose(np.sqrt(d2[finite].astype(np.float64)), gd[finite], rtol=1e-5, atol=1e-6)
    assert np.array_equal(np.isfinite(d2), finite)
    assert (ids[~finite] == -1).all()
    same = ids[finite] == gids[finite]
    assert same.mean() > 0.999
    # ascending order, squared distances
    assert (np.diff(np.where(np.isfinite(d2), d2, np.float32(3e38)), axis=1) >= 0).all()


def test_knn_contract_details(oracle):
    pts = np.array([[0, 0, 0, 1], [1, 0, 0, 1], [0, 2, 0, 1], [0, 0, 0, 1]], dtype=np.float32)
    q = np.array([[0, 0, 0, 1]], dtype=np.float32)
    ids, d2 = oracle.knn(pts, q, k=3)
    assert ids.tolist() == [[0, 3, 1]] and d2.tolist() == [[0.0, 0.0, 1.0]]  # tie -> smallest index first
    ids, d2 = oracle.knn(pts, q, k=3, allow_self=False)  # optionFlags = 0: d2 <= eps is rejected
    assert ids.tolist() == [[1, 2, -1]] and d2[0, :2].tolist() == [1.0, 4.0] and np.isinf(d2[0, 2])
    ids, d2 = oracle.knn(pts, q, k=2, max_dist=1.0)  # accepts d2 <= r^2 (inclusive)
    assert ids.tolist() == [[0, 3]]
    ids, d2 = oracle.knn(pts[1:3], q, k=2, max_dist=1.0)
    assert ids.tolist() == [[0, -1]] and d2[0, 0] == 1.0 and np.isinf(d2[0, 1])
    ids, d2 = oracle.knn(np.zeros((0, 4), np.float32), q, k=1)
    assert ids.tolist() == [[-1]] and np.isinf(d2[0, 0])


def test_quantile_matches_numpy(oracle, gold):
    d2 = gold["q_d2"]
    assert oracle.dists_quantile(d2, 0.85) == gold["q85"]
    assert oracle.dists_quantile(d2, 0.5) == gold["q50"]
    assert oracle.dists_quantile(d2, 0.1) == gold["q10"]
    assert oracle.dists_quantile(d2, 1.0) == gold["q100"]
    assert oracle.dists_quantile(np.array([np.inf, 0.0], dtype=np.float32), 0.5) < 0  # "no outlier to filter"


def test_outlier_chain(oracle, gold):
    d2 = gold["q_d2"].reshape(-1, 1)
    ids = np.zeros_like(d2, dtype=np.int32)
    err, w, lim = oracle.outlier_weights(oracle.make_config(outliers=[(4, 0.85)]), d2, ids)
    assert err == 0 and lim == gold["q85"]
    assert np.array_equal(w[:, 0], (d2[:, 0] <= lim).astype(np.float32))
    err, w, lim = oracle.outlier_weights(oracle.make_config(outliers=[(3, 3.0)]), d2, ids)
    assert err == 0 and lim == np.float32(3.0) * gold["q50"]
    err, w, _ = oracle.outlier_weights(oracle.make_config(outliers=[(1, 0.2), (2, 0.1)]), d2, ids)
    assert np.array_equal(w[:, 0], ((d2[:, 0] <= np.float32(0.2) ** 2) & (d2[:, 0] >= np.float32(0.1) ** 2)).astype(np.float32))
    err, w, _ = oracle.outlier_weights(oracle.make_config(outliers=[]), d2, ids)
    assert (w == 1).all()  # empty chain: all ones, even for invalid matches
    err, _, _ = oracle.outlier_weights(oracle.make_config(outliers=[(4, 0.5)]), np.full((4, 1), np.inf, np.float32), ids[:4])
    assert err == 2


def _pairs(n):
    ids = np.arange(n, dtype=np.int32).reshape(n, 1)
    d2 = np.full((n, 1), 0.01, dtype=np.float32)
    return ids, d2


def h(p):
    out = np.ones((p.shape[0], 4), dtype=np.float32)
    out[:, :3] = p
    return out


def test_point_to_point_matches_kabsch(oracle, gold):
    from norlab_icp_mapper_amd import synth
    P, Q, w = gold["p2p_P"], gold["p2p_Q"], gold["p2p_w"].reshape(-1, 1)
    ids, d2 = _pairs(P.shape[0])
    err, T, _, _, _, st = oracle.minimize(1, h(P), h(Q), None, ids, d2, w)
    assert err == 0 and st.pairs == int(w.sum())
    dt, dr = synth.pose_error(T, gold["p2p_T"])
    assert dt < 2e-5 and dr < 2e-6
    np.testing.assert_allclose(T[3], [0, 0, 0, 1])
    assert abs(np.linalg.det(T[:3, :3].astype(np.float64)) - 1) < 1e-5
    # reflection fix
    R = oracle.rotation_from_H(gold["refl_H"])
    np.testing.assert_allclose(R, gold["refl_R"], atol=2e-6)
    assert np.linalg.det(R.astype(np.float64)) > 0.999


def test_point_to_plane_matches_numpy(oracle, gold):
    from norlab_icp_mapper_amd import synth
    P, Q, N = gold["p2l_P"], gold["p2l_Q"], gold["p2l_N"]
    w = gold["p2p_w"].reshape(-1, 1)
    ids, d2 = _pairs(P.shape[0])
    err, T, A, b, x, st = oracle.minimize(2, h(P), h(Q), N, ids, d2, w)
    assert err == 0
    np.testing.assert_allclose(A, gold["p2l_A"], rtol=2e-5, atol=1e-3)
    np.testing.assert_allclose(b, gold["p2l_b"], rtol=2e-4, atol=1e-3)
    np.testing.assert_allclose(x, gold["p2l_x"], rtol=2e-3, atol=2e-6)
    dt, dr = synth.pose_error(T, gold["p2l_T"])
    assert dt < 2e-5 and dr < 2e-6
    # rank-deficient system -> minimum-norm solution
    xs = oracle.solve6(gold["sing_A"], gold["sing_b"])
    np.testing.assert_allclose(xs, gold["sing_x"], atol=2e-5)
    # no pairs -> "no point to minimize"
    err, *_ = oracle.minimize(2, h(P), h(Q), N, ids, d2, np.zeros_like(w))
    assert err == 1


def test_surface_normals_and_cells(oracle, gold):
    n = oracle.surface_normals(gold["sn_pts"], knn=10)
    dots = np.abs(n.astype(np.float64) @ gold["sn_normal"])
    assert dots.min() > 1 - 1e-4
    assert np.array_equal(oracle.cell_ids(gold["cell_pts"], 20.0), gold["cell_ijk"])
    ijk = oracle.cell_ids(gold["cell_pts"][:6], 20.0)[:, 0]
    assert ijk.tolist() == [-1, 1, 0, 0, 0, -2]  # floor semantics at the cell faces (Map.cpp:232-235)


def test_point_distance_keep(oracle):
    rng = np.random.default_rng(1)
    m = h(rng.uniform(-5, 5, (3000, 3)).astype(np.float32))
    i = h(rng.uniform(-6, 6, (500, 3)).astype(np.float32))
    keep = oracle.point_distance_keep(m, i, 0.4)
    d = np.sqrt(((i[:, None, :3].astype(np.float64) - m[None, :, :3]) ** 2).sum(-1)).min(1)
    sure = np.abs(d - 0.4) > 1e-4
    assert np.array_equal(keep[sure], (d >= 0.4)[sure])
    # an exact duplicate of a map point skips its twin and is judged on the next neighbour
    d2 = np.sqrt(((m[:50, None, :3].astype(np.float64) - m[None, :, :3]) ** 2).sum(-1))
    d2[np.arange(50), np.arange(50)] = np.inf
    assert np.array_equal(oracle.point_distance_keep(m, m[:50], 0.3), d2.min(1) >= 0.3)


CH = dict(max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1, nthreads=4)


def test_icp_recovers_ground_truth(oracle, small_scene):
    from norlab_icp_mapper_amd import synth
    sc = small_scene
    icp = oracle.OracleICP(oracle.make_config(minimizer=2, **CH))
    assert icp.setMap(sc["map"], sc["normals"])
    err, T = icp(sc["scan"])
    assert err == 0 and icp.stats.stop_reason == 2 and 3 <= icp.stats.iterations < 40
    dt, dr = synth.pose_error(T, sc["T_gt"])
    assert dt < 0.01 and dr < 1e-3
    assert abs(icp.stats.weighted_point_used_ratio - 0.85) < 0.01
    # Counter only: exactly max_iterations passes of the loop
    icp2 = oracle.OracleICP(oracle.make_config(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=7))
    icp2.setMap(sc["map"], sc["normals"])
    err, _ = icp2(sc["scan"])
    assert err == 0 and icp2.stats.iterations == 7 and icp2.stats.stop_reason == 1


def test_icp_sequence_semantics(oracle, small_scene):
    sc = small_scene
    icp = oracle.OracleICP(oracle.make_config(minimizer=0, knn=6, max_dist=2.0, max_iterations=10))
    err, T = icp(sc["scan"])
    assert err == 0 and np.array_equal(T, np.eye(4, dtype=np.float32))  # no map: identity
    assert icp.setMap(np.zeros((0, 4), np.float32)) is False
    assert icp.setMap(sc["map"])
    mean = icp.getMapMean()
    np.testing.assert_allclose(mean, sc["map"][:, :3].astype(np.float64).mean(0), atol=1e-5)
    err, T = icp(sc["scan"])  # the bundled example's chain: 10 NN passes, identity correction
    assert err == 0 and icp.stats.iterations == 10 and np.array_equal(T, np.eye(4, dtype=np.float32))
    far = sc["scan"].copy(); far[:, :3] += 1000
    err, _ = icp(far)
    assert err == 1
    icp_b = oracle.OracleICP(oracle.make_config(minimizer=2, max_dist=2.0, use_bound=1, max_rot_norm=1e-4, max_trans_norm=1e-4))
    icp_b.setMap(sc["map"], sc["normals"])
    err, _ = icp_b(sc["scan"])
    assert err == 3


def test_voxel_keep_first(oracle):
    rng = np.random.default_rng(11)
    c = np.ones((3000, 4), dtype=np.float32)
    c[:, :3] = rng.uniform(-5, 5, (3000, 3)).astype(np.float32)
    for edge in (0.5, 2.0):
        keep = oracle.voxel_keep_first(c, edge)
        lo = c[:, :3].min(0)
        ijk = np.floor((c[:, :3] - lo) / np.float32(edge)).astype(np.int64)
        key = (ijk[:, 0] * 2097152 + ijk[:, 1]) * 2097152 + ijk[:, 2]
        _, first = np.unique(key, return_index=True)
        assert np.array_equal(np.flatnonzero(keep), np.sort(first))


def test_voxel_keep_pseudo_random_representative(oracle):
    """samplingMethod 1 made reproducible: the point whose index has the smallest fmix32 represents its voxel."""
    rng = np.random.default_rng(12)
    c = np.ones((4000, 4), dtype=np.float32)
    c[:, :3] = rng.uniform(-3, 3, (4000, 3)).astype(np.float32)

    def fmix32(h):
        h = np.asarray(h, dtype=np.uint64)
        h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
        h ^= h >> np.uint64(16)
        return h
    keep = oracle.voxel_keep(c, 0.75, 1)
    lo = c[:, :3].min(0)
    ijk = np.floor((c[:, :3] - lo) / np.float32(0.75)).astype(np.int64)
    key = (ijk[:, 0] * 2097152 + ijk[:, 1]) * 2097152 + ijk[:, 2]
    h = fmix32(np.arange(4000))
    order = np.lexsort((h, key))
    first = order[np.r_[True, key[order][1:] != key[order][:-1]]]
    assert np.array_equal(np.flatnonzero(keep), np.sort(first))
    assert keep.sum() == oracle.voxel_keep(c, 0.75, 0).sum()        # same voxels, other representatives


def test_filter_points_against_numpy(oracle):
    """DistanceLimit / BoundingBox predicates (Mapper.cpp:27-31, examples/config.yaml:2-18)."""
    rng = np.random.default_rng(13)
    c = np.ones((5000, 4), dtype=np.float32)
    c[:, :3] = rng.uniform(-8, 8, (5000, 3)).astype(np.float32)
    c[0, :3] = (0.5, 0.0, 0.0); c[1, :3] = (-1.5, 0.2, 0.1)          # exactly on a box face: strict comparisons keep them outside
    filters = [("distance_limit", -1, 7.5, False), ("bounding_box", (-1.5, -1, -1), (0.5, 1, 0.5), True),
               ("bounding_box", (-6, -2.5, -1), (-1.5, 2.5, 1), True), ("distance_limit", 2, -6.0, False)]
    keep = oracle.filter_points(c, filters)
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    r = np.sqrt(x * x + y * y + z * z, dtype=np.float32)
    in1 = (x > -1.5) & (x < 0.5) & (y > -1) & (y < 1) & (z > -1) & (z < 0.5)
    in2 = (x > -6) & (x < -1.5) & (y > -2.5) & (y < 2.5) & (z > -1) & (z < 1)
    ref = (r < 7.5) & ~in1 & ~in2 & (np.abs(z) < 6.0)
    assert np.array_equal(keep, ref) and keep[0] and keep[1]
    assert np.array_equal(oracle.filter_points(c, [("distance_limit", 0, 3.0, True)]), np.abs(x) > 3.0)
    assert oracle.filter_points(c, []).all()
