"""GPU parity tests: the HIP path (through the C ABI of libicpmi.so) against the CPU oracle on the
same seeded inputs.  Bars: bit-exact for indices, squared distances, weights and the quantile limit
(integer / comparison work on identically specified float arithmetic); poses within 1e-4 m / 1e-4 rad
(the tolerance BASELINE.json's north_star states)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


def centred(cloud, mean):
    out = cloud.copy()
    out[:, :3] = cloud[:, :3] - mean[None, :]
    return out


def test_transform_bit_exact(amd, oracle, small_scene):
    icp = amd.ICPSequence(minimizer=0)
    T = amd.synth.make_T((0.3, -0.2, 0.5), (1.5, -2.0, 0.25))
    out, outn = icp.transform(T, small_scene["scan"], small_scene["scan_normals"])
    ref = oracle.transform(T, small_scene["scan"])
    refn = oracle.rotate3(T, small_scene["scan_normals"])
    assert np.array_equal(out, ref)
    assert np.array_equal(outn, refn)
    bad = np.eye(4); bad[0, 0] = 1.5
    with pytest.raises(amd.TransformationError):
        icp.transform(bad, small_scene["scan"])


@pytest.mark.parametrize("k,max_dist", [(1, 2.0), (1, math.inf), (1, 0.05), (6, 2.0), (10, math.inf), (12, 2.0), (16, math.inf), (16, 0.1), (20, 2.0)])
def test_knn_matches_oracle_exactly(amd, oracle, small_scene, k, max_dist):
    icp = amd.ICPSequence(minimizer=0, knn=min(k, 32))
    assert icp.setMap(small_scene["map"])
    mean = icp.getMapMean()
    ocfg = oracle.make_config()
    oicp = oracle.OracleICP(ocfg); oicp.setMap(small_scene["map"])
    assert np.array_equal(mean, oicp.getMapMean())
    mapc = centred(small_scene["map"], mean)
    q = centred(small_scene["scan"], mean)
    ids, d2 = icp.knn(q, k=k, max_dist=max_dist)
    rids, rd2 = oracle.knn(mapc, q, k=k, max_dist=max_dist, nthreads=8)
    assert np.array_equal(d2, rd2)
    assert np.array_equal(ids, rids)


def test_knn_far_queries_and_self_match(amd, oracle, small_scene):
    icp = amd.ICPSequence(minimizer=0)
    m = small_scene["map"][:20000]
    icp.setMap(m)
    mean = icp.getMapMean()
    mapc = centred(m, mean)
    # queries far outside the bounding box, with and without a radius
    rng = np.random.default_rng(5)
    q = np.ones((300, 4), dtype=np.float32)
    q[:, :3] = rng.uniform(-400, 400, size=(300, 3)).astype(np.float32)
    for md in (math.inf, 30.0):
        ids, d2 = icp.knn(q, k=1, max_dist=md)
        rids, rd2 = oracle.knn(mapc, q, k=1, max_dist=md)
        assert np.array_equal(d2, rd2) and np.array_equal(ids, rids)
    # self match excluded (PointDistanceMapperModule.cpp:36 passes optionFlags = 0)
    ids, d2 = icp.knn(mapc[:2000], k=1, allow_self=False)
    rids, rd2 = oracle.knn(mapc, mapc[:2000], k=1, allow_self=False)
    assert np.array_equal(d2, rd2) and np.array_equal(ids, rids)
    assert (ids[:, 0] != np.arange(2000)).all()


@pytest.mark.parametrize("shape", ["single", "identical", "planar", "collinear", "duplicates", "far_offset", "two_clusters"])
def test_knn_degenerate_maps(amd, oracle, shape):
    """Degenerate geometry through the grid pyramid: zero extents, ties, k larger than the map, big offsets."""
    rng = np.random.default_rng(7)
    def cloud(xyz):
        c = np.ones((xyz.shape[0], 4), dtype=np.float32); c[:, :3] = xyz.astype(np.float32); return c
    if shape == "single":
        m = cloud(np.array([[1.0, 2.0, 3.0]]))
    elif shape == "identical":
        m = cloud(np.tile([[0.5, -0.25, 4.0]], (7, 1)))
    elif shape == "planar":
        xy = rng.uniform(-3, 3, (500, 2)); m = cloud(np.c_[xy, np.full(500, 1.25)])
    elif shape == "collinear":
        t = rng.uniform(-5, 5, 300); m = cloud(np.c_[t, 2 * t, np.zeros(300)])
    elif shape == "duplicates":
        base = rng.uniform(-2, 2, (200, 3)); m = cloud(np.r_[base, base, base[:50]])
    elif shape == "far_offset":
        m = cloud(rng.uniform(-1, 1, (400, 3)) + np.array([8000.0, -6000.0, 120.0]))
    else:
        m = cloud(np.r_[rng.normal(0, 0.05, (150, 3)), rng.normal(0, 0.05, (150, 3)) + 40.0])
    q = cloud(m[rng.integers(0, m.shape[0], 64), :3] + rng.normal(0, 0.3, (64, 3)))
    q = np.concatenate([q, m[: min(8, m.shape[0])]])  # exact hits (d2 = 0)
    for k in (1, 3, 6, 10, 16):               # (9..16: the cooperative kernels since r3; k may exceed the map)
        for md in (math.inf, 0.5):
            icp = amd.ICPSequence(minimizer=0, knn=k, max_dist=md if math.isfinite(md) else 2.0)
            assert icp.setMap(m)
            mean = icp.getMapMean()
            ids, d2 = icp.knn(centred(q, mean), k=k, max_dist=md)
            rids, rd2 = oracle.knn(centred(m, mean), centred(q, mean), k=k, max_dist=md)
            assert np.array_equal(d2, rd2), (shape, k, md)
            assert np.array_equal(ids, rids), (shape, k, md)


def test_trimmed_limit_and_weights_exact(amd, oracle, small_scene):
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)])
    icp.setMap(small_scene["map"])
    mean = icp.getMapMean()
    q = centred(small_scene["scan"], mean)
    ids, d2 = icp.knn(q, k=1, max_dist=2.0)
    w, lim = icp.outlierWeights(d2, ids)
    ocfg = oracle.make_config(max_dist=2.0, outliers=[(4, 0.85)])
    err, rw, rlim = oracle.outlier_weights(ocfg, d2, ids)
    assert err == 0
    assert lim == rlim
    assert np.array_equal(w, rw)
    for chain in ([(3, 3.0)], [(1, 0.5), (4, 0.7)], [(2, 0.05)], [(4, 1.0)], [(4, 0.0)]):
        icp2 = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=chain)
        w, lim = icp2.outlierWeights(d2, ids)
        err, rw, rlim = oracle.outlier_weights(oracle.make_config(outliers=chain), d2, ids)
        assert err == 0 and np.array_equal(w, rw), chain
        if any(t in (3, 4) for t, _ in chain):
            assert lim == rlim


@pytest.mark.parametrize("minimizer", [1, 2])
def test_single_step_matches_oracle(amd, oracle, small_scene, minimizer):
    sc = small_scene
    icp = amd.ICPSequence(minimizer=minimizer, max_dist=2.0, outliers=[(4, 0.85)])
    icp.setMap(sc["map"], sc["normals"])
    mean = icp.getMapMean()
    mapc = centred(sc["map"], mean)
    q = centred(sc["scan"], mean)
    T_iter = amd.synth.make_T((0.002, 0.001, -0.003), (0.02, 0.01, -0.01)).astype(np.float32)
    T_step, sums = icp.minimizeStep(q, T_iter)
    step = oracle.transform(T_iter, q)
    ids, d2 = oracle.knn(mapc, step, k=1, max_dist=2.0)
    ocfg = oracle.make_config(max_dist=2.0, outliers=[(4, 0.85)])
    err, w, lim = oracle.outlier_weights(ocfg, d2, ids)
    err, T_ref, A, b, x, st = oracle.minimize(minimizer, step, mapc, sc["normals"], ids, d2, w)
    assert err == 0
    assert icp.stats.pairs == st.pairs
    assert icp.stats.trimmed_limit == lim
    if minimizer == 2:
        iu = np.triu_indices(6)
        np.testing.assert_allclose(sums[:21], A[iu], rtol=1e-10, atol=1e-9)
        np.testing.assert_allclose(sums[21:27], b, rtol=1e-10, atol=1e-9)
    dt, dr = amd.synth.pose_error(T_step, T_ref)
    assert dt < 1e-5 and dr < 1e-5


CHAINS = {
    "p2p": dict(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1),
    "p2plane": dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1),
    "p2plane_knn6": dict(minimizer=2, knn=6, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=15),
    "p2plane_nofilter_counter": dict(minimizer=2, max_dist=2.0, outliers=[], max_iterations=12),
    "p2p_median_maxdist": dict(minimizer=1, max_dist=math.inf, outliers=[(1, 1.0), (3, 3.0)], max_iterations=10),
    "identity": dict(minimizer=0, knn=6, max_dist=2.0, outliers=[], max_iterations=10),
}


@pytest.mark.parametrize("name", list(CHAINS))
def test_full_registration_matches_oracle(amd, oracle, mid_scene, name):
    sc = mid_scene
    kw = dict(CHAINS[name])
    icp = amd.ICPSequence(**kw)
    assert icp.setMap(sc["map"], sc["normals"])
    T = icp(sc["scan"])
    okw = dict(kw); okw["nthreads"] = 8
    oicp = oracle.OracleICP(oracle.make_config(**okw))
    oicp.setMap(sc["map"], sc["normals"])
    err, T_ref = oicp(sc["scan"])
    assert err == 0
    assert icp.stats.iterations == oicp.stats.iterations
    assert icp.stats.stop_reason == oicp.stats.stop_reason
    assert icp.stats.pairs == oicp.stats.pairs
    dt, dr = amd.synth.pose_error(T, T_ref)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)
    assert abs(icp.errorMinimizer.getOverlap() - oicp.stats.weighted_point_used_ratio) < 1e-6
    if name == "p2plane":
        # ground truth by construction
        dt, dr = amd.synth.pose_error(T, sc["T_gt"])
        assert dt < 5e-3 and dr < 5e-4
    if name == "identity":
        assert np.array_equal(T, np.eye(4, dtype=np.float32))


def test_caller_stream_and_config_swap(amd, mid_scene):
    """icpmi_set_stream: the handle's work runs on the caller's stream (here a torch stream); icpmi_set_config swaps the
    chain of a live handle and keeps its map."""
    import torch
    sc = mid_scene
    kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=10, use_differential=0)
    ref = amd.ICPSequence(**kw); ref.setMap(sc["map"], sc["normals"]); T_ref = ref(sc["scan"])
    icp = amd.ICPSequence(**kw)
    stream = torch.cuda.Stream()
    icp.setStream(stream.cuda_stream)
    icp.setMap(sc["map"], sc["normals"])
    d = torch.from_numpy(sc["scan"]).cuda()
    with torch.cuda.stream(stream):
        d2 = d * 1.0                                            # produced on that stream, consumed by the registration
    T = icp.registerDev(d2.data_ptr(), d2.shape[0])
    assert np.array_equal(T, T_ref)
    # same handle, other chain: point-to-point, median filter; the map stays
    kw2 = dict(minimizer=1, max_dist=2.0, outliers=[(3, 3.0)], max_iterations=6, use_differential=0)
    icp.setConfig(**kw2)
    ref2 = amd.ICPSequence(**kw2); ref2.setMap(sc["map"], sc["normals"])
    assert np.array_equal(icp(sc["scan"]), ref2(sc["scan"]))


@pytest.mark.parametrize("knn", [1, 3])
def test_surface_normal_outlier_filter_chain(amd, oracle, mid_scene, knn):
    """SurfaceNormalOutlierFilter needs the reading's normals (rotated with the cloud, B.7) next to each match: with the
    loop state kept in tile-sorted query order the descriptor lookup goes through the sort's index."""
    sc = mid_scene
    kw = dict(minimizer=2, knn=knn, max_dist=2.0, outliers=[(5, 0.5), (4, 0.9)], max_iterations=12, use_differential=0)
    icp = amd.ICPSequence(**kw)
    icp.setMap(sc["map"], sc["normals"])
    rng = np.random.default_rng(4)
    sn = sc["scan_normals"].copy()
    flip = rng.random(sn.shape[0]) < 0.3                     # a third of the reading disagrees with the map's normals
    sn[flip] = np.roll(sn[flip], 1, axis=1)
    T = icp(sc["scan"], sn)
    oicp = oracle.OracleICP(oracle.make_config(nthreads=8, **kw))
    oicp.setMap(sc["map"], sc["normals"])
    err, T_ref = oicp(sc["scan"], sn)
    assert err == 0
    assert icp.stats.iterations == oicp.stats.iterations and icp.stats.pairs == oicp.stats.pairs
    assert icp.stats.pairs < 0.8 * sc["scan"].shape[0] * knn   # the filter did reject the flipped normals
    dt, dr = amd.synth.pose_error(T, T_ref)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)
    with pytest.raises(amd.InvalidField):
        icp(sc["scan"])                                       # reading without normals


def test_graph_and_eager_agree(amd, mid_scene):
    import ctypes as C
    sc = mid_scene
    out = {}
    for use_graph in (0, 1):
        icp = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], use_graph=use_graph)
        icp.setMap(sc["map"], sc["normals"])
        import torch
        d = torch.from_numpy(sc["scan"]).cuda()
        T = icp.registerDev(d.data_ptr(), d.shape[0], fixed_iterations=8)
        T2 = icp.registerDev(d.data_ptr(), d.shape[0], fixed_iterations=8)  # graph replay
        assert np.array_equal(T, T2)
        assert icp.stats.iterations == 8
        out[use_graph] = T
    assert np.array_equal(out[0], out[1])


def test_checked_loop_head_graph_follows_the_previous_iteration_count(amd, oracle, mid_scene):
    """r5: the head graph of a checked registration (Counter + Differential: what Mapper::processInput runs) is cut to the iteration count of
    the handle's previous registrations; a scan that needs more iterations than predicted goes on with ordinary segments, one that needs
    fewer runs dead iterations.  Scans of different difficulty in turn on ONE handle: every result must be the oracle's (iterations, stop
    reason, pose bit for bit with the eager loop of a fresh handle)."""
    sc = mid_scene
    kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
    rng = np.random.default_rng(3)
    scans = [sc["scan"]]
    for shift, yaw in ((0.02, 0.0), (1.2, 0.0), (0.0, 0.12), (0.5, -0.06)):   # closer to / farther from the map: fewer / more iterations
        cy, sy = np.float32(np.cos(yaw)), np.float32(np.sin(yaw))
        s2 = sc["scan"].copy()
        s2[:, 0], s2[:, 1] = cy * s2[:, 0] - sy * s2[:, 1] + np.float32(shift), sy * sc["scan"][:, 0] + cy * s2[:, 1] - np.float32(shift / 2)
        scans.append(s2)
    scans.append(sc["scan"][rng.permutation(sc["scan"].shape[0])[:7000]].copy())   # another size: the graphs are rebuilt
    icp = amd.ICPSequence(**kw)
    icp.setMap(sc["map"], sc["normals"])
    ref = amd.ICPSequence(use_graph=0, **kw)
    ref.setMap(sc["map"], sc["normals"])
    counts = []
    for rep in range(2):
        for s in scans + scans[::-1]:
            T = icp(s); it, why = icp.stats.iterations, icp.stats.stop_reason
            Tr = ref(s)
            assert (it, why) == (ref.stats.iterations, ref.stats.stop_reason)
            assert np.array_equal(T, Tr)
            counts.append(it)
    assert len(set(counts)) >= 3, counts              # the sequence really alternates between different loop lengths
    o = oracle.OracleICP(oracle.make_config(nthreads=8, **kw)); o.setMap(sc["map"], sc["normals"])
    for s in scans[:3]:
        _, To = o(s); T = icp(s)
        assert icp.stats.iterations == o.stats.iterations
        dt, dr = amd.synth.pose_error(T, To)
        assert dt <= 1e-4 and dr <= 1e-4


@pytest.mark.parametrize("minimizer", [1, 2])
def test_registration_is_bitwise_reproducible(amd, mid_scene, minimizer):
    """The loop keeps its state and runs its pair sums in tile-sorted query order; that order comes
    from a stable radix sort, so repeated runs and fresh handles give the same bits."""
    sc = mid_scene
    seen = []
    for _ in range(3):
        icp = amd.ICPSequence(minimizer=minimizer, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=12, use_differential=0)
        icp.setMap(sc["map"], sc["normals"])
        for _ in range(2):
            seen.append(np.asarray(icp(sc["scan"])).copy())
    for T in seen[1:]:
        assert np.array_equal(T.view(np.uint32), seen[0].view(np.uint32))


def test_error_paths(amd, small_scene):
    sc = small_scene
    icp = amd.ICPSequence(minimizer=2, max_dist=2.0)
    # no map: identity, like upstream
    assert not icp.hasMap()
    assert np.array_equal(icp(sc["scan"]), np.eye(4, dtype=np.float32))
    # empty map rejected, state unchanged
    assert icp.setMap(np.zeros((0, 4), dtype=np.float32)) is False
    assert not icp.hasMap()
    # point-to-plane without normals
    icp.setMap(sc["map"])
    with pytest.raises(amd.InvalidField):
        icp(sc["scan"])
    icp.setMap(sc["map"], sc["normals"])
    # empty reading
    with pytest.raises(amd.ConvergenceError):
        icp(np.zeros((0, 4), dtype=np.float32))
    # reading entirely out of range: no pairs
    far = sc["scan"].copy(); far[:, :3] += 1000.0
    with pytest.raises(amd.ConvergenceError):
        icp(far)
    icp_t = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)])
    icp_t.setMap(sc["map"], sc["normals"])
    with pytest.raises(amd.ConvergenceError):
        icp_t(far)
    # bound checker
    icp_b = amd.ICPSequence(minimizer=2, max_dist=2.0, use_bound=1, max_rot_norm=1e-4, max_trans_norm=1e-4)
    icp_b.setMap(sc["map"], sc["normals"])
    with pytest.raises(amd.ConvergenceError):
        icp_b(sc["scan"])
    with pytest.raises(amd.InvalidParameter):
        amd.ICPSequence(knn=0)


@pytest.mark.parametrize("minimizer", [1, 2])
def test_small_and_exact_readings_match_oracle(amd, oracle, small_scene, minimizer):
    """Ragged / tiny readings and readings that hit map points exactly (d2 = 0 is excluded from the
    quantile, libpointmatcher's getDistsQuantile): same poses, same iteration counts, same errors."""
    sc = small_scene
    rng = np.random.default_rng(3)
    kw = dict(minimizer=minimizer, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=15, use_differential=0)
    icp = amd.ICPSequence(**kw)
    icp.setMap(sc["map"], sc["normals"])
    oicp = oracle.OracleICP(oracle.make_config(nthreads=4, **kw))
    oicp.setMap(sc["map"], sc["normals"])
    exact = sc["map"][rng.integers(0, sc["map"].shape[0], 400)].copy()
    readings = {
        "n=7": sc["scan"][:7],
        "n=64": sc["scan"][:64],
        "n=257": sc["scan"][100:357],
        "half_exact": np.concatenate([exact[:200], sc["scan"][:200]]),
    }
    for name, rd in readings.items():
        err, T_ref = oicp(rd)
        if err != 0:
            with pytest.raises(amd.ConvergenceError):
                icp(rd)
            continue
        T = icp(rd)
        assert icp.stats.iterations == oicp.stats.iterations, name
        assert icp.stats.pairs == oicp.stats.pairs, name
        dt, dr = amd.synth.pose_error(T, T_ref)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (name, dt, dr)
    # every match exact: nothing for the quantile to rank => "no outlier to filter" on both sides
    err, _ = oicp(exact)
    assert err != 0
    with pytest.raises(amd.ConvergenceError):
        icp(exact)


def test_surface_normals_match_oracle(amd, oracle, small_scene):
    pts = small_scene["map"][:30000]
    icp = amd.ICPSequence(minimizer=0)
    n = icp.surfaceNormals(pts, knn=10)
    rn = oracle.surface_normals(pts, knn=10, nthreads=8)
    # the neighbour sets are exact (ids and d2 bit-equal to the oracle's: test_knn_matches_oracle_exactly), so both sides
    # diagonalise the same 3 x 3 matrices: no point is allowed to be wrong
    ids, _ = oracle.knn(pts, pts, k=10, nthreads=8)
    assert (ids >= 0).all()
    ok = oracle.check_normals_are_smallest_eigenvectors(pts, ids, n, "gpu")
    oracle.check_normals_are_smallest_eigenvectors(pts, ids, rn, "oracle")
    assert ok.mean() > 0.999
    # and where the smallest eigenvalue is isolated the two sides agree up to sign to float precision
    P = pts[:, :3].astype(np.float64); nb = P[ids]; d = nb - nb.mean(axis=1, keepdims=True)
    lam = np.linalg.eigvalsh(np.einsum("nki,nkj->nij", d, d))
    isolated = ok & ((lam[:, 1] - lam[:, 0]) > 1e-2 * lam[:, 2])
    dots = np.abs(np.sum(n.astype(np.float64) * rn.astype(np.float64), axis=1))
    assert isolated.mean() > 0.9 and dots[isolated].min() > 1 - 1e-6, float(dots[isolated].min())


def test_point_distance_keep_and_bins_exact(amd, oracle, small_scene):
    sc = small_scene
    icp = amd.ICPSequence(minimizer=0)
    m, i = sc["map"][:40000], sc["scan"]
    keep = icp.pointDistanceKeep(m, i, 0.15)
    rkeep = oracle.point_distance_keep(m, i, 0.15, nthreads=8)
    assert np.array_equal(keep, rkeep)
    # duplicates of map points are KEPT (self-match exclusion quirk)
    keep2 = icp.pointDistanceKeep(m, m[:100], 1e-3)
    rkeep2 = oracle.point_distance_keep(m, m[:100], 1e-3)
    assert np.array_equal(keep2, rkeep2)
    pts = sc["map"].copy(); pts[:, :3] *= 3.0
    assert np.array_equal(icp.binCells(pts, 20.0), oracle.cell_ids(pts, 20.0))


# ---- bundled example data of the reference (tests/golden/bundled_scans.npz, see make_golden.py) ----
def _quat_T(row):
    x, y, z, qx, qy, qz, qw = row[2:9]
    R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                  [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                  [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = [x, y, z]
    return T.astype(np.float32)


def _bundled_input_filters(xyz):
    """examples/config.yaml:1-17 + Mapper.cpp:27-31: radius 200 m, two BoundingBox{removeInside:1}"""
    keep = np.linalg.norm(xyz, axis=1) < 200.0
    for lo, hi in (((-1.5, -1, -1), (0.5, 1, 0.5)), ((-6, -2.5, -1), (-1.5, 2.5, 1))):
        inside = np.all((xyz > np.array(lo)) & (xyz < np.array(hi)), axis=1)
        keep &= ~inside
    out = np.ones((int(keep.sum()), 4), dtype=np.float32)
    out[:, :3] = xyz[keep]
    return out


def test_voxel_keep_first_exact(amd, oracle, small_scene):
    icp = amd.ICPSequence(minimizer=0)
    rng = np.random.default_rng(5)
    cloud = small_scene["map"].copy()
    cloud = np.concatenate([cloud, cloud[:500]])          # exact duplicates share a voxel
    cloud[:, :3] += rng.normal(0, 1e-3, (cloud.shape[0], 3)).astype(np.float32) * (np.arange(cloud.shape[0]) % 2)[:, None]
    for edge in (0.15, 1.0, 7.5, 1e6):
        keep = icp.voxelKeepFirst(cloud, edge)
        ref = oracle.voxel_keep_first(cloud, edge)
        assert np.array_equal(keep, ref), edge
        # properties: one survivor per occupied voxel, and it is the first of that voxel
        lo = cloud[:, :3].min(0)
        ijk = np.minimum(np.floor((cloud[:, :3] - lo) / np.float32(edge)), 2097151).astype(np.int64)
        key = (ijk[:, 0] * 2097152 + ijk[:, 1]) * 2097152 + ijk[:, 2]
        _, first = np.unique(key, return_index=True)
        assert np.array_equal(np.flatnonzero(keep), np.sort(first))
    assert icp.voxelKeepFirst(cloud[:0], 0.5).shape == (0,)
    with pytest.raises(amd.InvalidParameter):
        icp.voxelKeepFirst(cloud, 0.0)


def test_dynamic_points_update_matches_oracle(amd, oracle, small_scene):
    """DynamicPointsMapperModule::inPlaceUpdateMap: the device's angular bucket grid against the
    oracle's brute-force beam search, bit for bit (same float arithmetic, asin/atan2 rounded from double)."""
    sc = small_scene
    icp = amd.ICPSequence(minimizer=0)
    rng = np.random.default_rng(21)
    pose = amd.synth.make_T((0.02, -0.01, 0.3), (3.0, -2.0, 1.5)).astype(np.float32)
    to_sensor = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
    mp = sc["map"][:15000]
    nrm = sc["normals"][:15000]
    scan = sc["scan"]
    # a moved object: some scan points pulled towards the sensor along their beam => map points behind them turn dynamic
    sensor = pose[:3, 3]
    scan = scan.copy()
    pull = rng.random(scan.shape[0]) < 0.2
    scan[pull, :3] = sensor + (scan[pull, :3] - sensor) * 0.6
    prob0 = rng.uniform(0.0, 1.0, mp.shape[0]).astype(np.float32)
    for kw in (dict(), dict(beam_half_angle=0.03, sensor_max_range=40.0), dict(threshold_dynamic=0.5, alpha=0.6, beta=0.9, epsilon_a=0.05, epsilon_d=0.1)):
        got = icp.dynamicPointsUpdate(to_sensor, scan, mp, nrm, prob0, **kw)
        ref = oracle.dynamic_points_update(to_sensor, scan, mp, nrm, prob0, nthreads=8, **kw)
        changed = ref != prob0
        assert changed.sum() > 100
        assert np.array_equal(got, ref), (np.flatnonzero(got != ref)[:10], kw)
        assert np.isfinite(got).all() and (got >= 0).all() and (got <= 1).all()
    # no beams / no map: nothing changes
    assert np.array_equal(icp.dynamicPointsUpdate(to_sensor, scan[:0], mp, nrm, prob0), prob0)


def test_dynamic_points_update_planar_matches_oracle(amd, oracle):
    """The is3D == false branch (DynamicPointsMapperModule.cpp:156-172): elevation 0 for every point, azimuth atan2(y, x), radii
    over the two axes.  Planar clouds keep z == 0 in the 4 x N layout, where the 3-D formulas give exactly that -- asin(0 / r) = 0,
    sqrt(x^2 + y^2 + 0) -- so the device kernel and the oracle's restatement serve both cases; a dynamic object (scan points
    pulled towards the sensor) must raise the probability of the wall points behind it, bit for bit as the oracle does."""
    from test_oracle_ext import _planar_scene
    mp, scan, T = _planar_scene(n_map=12000, n_scan=3000, seed=4)
    icp = amd.ICPSequence(minimizer=0, is_2d=1)
    nrm = icp.surfaceNormals(mp, knn=8)
    assert np.all(nrm[:, 2] == 0)
    pose = T.astype(np.float32)                                    # planar pose: the scan is given in this sensor frame
    to_sensor = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
    scan_map = icp.transform(pose, scan)
    assert np.all(scan_map[:, 2] == 0)
    rng = np.random.default_rng(3)
    sensor = pose[:3, 3]
    pull = rng.random(scan_map.shape[0]) < 0.25
    scan_map[pull, :3] = sensor + (scan_map[pull, :3] - sensor) * 0.5
    prob0 = rng.uniform(0.0, 1.0, mp.shape[0]).astype(np.float32)
    for kw in (dict(beam_half_angle=0.01), dict(beam_half_angle=0.03, sensor_max_range=5.0), dict(threshold_dynamic=0.5, alpha=0.6, beta=0.9, epsilon_a=0.05, epsilon_d=0.1)):
        got = icp.dynamicPointsUpdate(to_sensor, scan_map, mp, nrm, prob0, **kw)
        ref = oracle.dynamic_points_update(to_sensor, scan_map, mp, nrm, prob0, nthreads=8, **kw)
        assert (ref != prob0).sum() > 100
        assert np.array_equal(got, ref), (np.flatnonzero(got != ref)[:10], kw)
    # and through the resident chain in planar mode: the same update as one operator of icpmi_map_update_chain
    icp.setMap(mp, nrm)
    icp.setMapScalar(prob0)
    dyn = ("dynamic_points", 0.6, 0.8, 0.99, 0.01, 0.01, 0.01, 200.0)
    icp.mapUpdateChain(scan_map, [dyn], [], scan_scalar=np.full(scan_map.shape[0], 0.6, np.float32), scan_normals=icp.surfaceNormals(scan_map, knn=8),
                       to_sensor=to_sensor)
    ref = oracle.dynamic_points_update(to_sensor, scan_map, mp, nrm, prob0, nthreads=8)
    got = icp.getMapScalar()
    assert np.array_equal(got, ref)                                # (the module alone updates the map's descriptor: it adds no point)


@pytest.mark.parametrize("normals_knn", [0, 7])
def test_resident_map_update_equals_composed_path(amd, mid_scene, normals_knn):
    """icpmi_map_update_point_distance (keep mask vs the resident map, append, normals, rebuild -- all on the
    device) against the same chain composed from the host-pointer operators: identical map, identical normals,
    identical registrations afterwards."""
    sc = mid_scene
    base = sc["map"][::2]
    rng = np.random.default_rng(9)
    scan_map = sc["map"][1::2][rng.permutation(sc["map"].shape[0] // 2)[:20000]].copy()   # unseen half of the surface
    scan_map[:, :3] += rng.normal(0, 0.02, (scan_map.shape[0], 3)).astype(np.float32)
    kw = dict(minimizer=2 if normals_knn else 1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=10, use_differential=0)

    a = amd.ICPSequence(**kw)                                   # composed, host pointers
    keep = a.pointDistanceKeep(base, scan_map, 0.25)
    grown = np.concatenate([base, scan_map[keep]])
    nrm = a.surfaceNormals(grown, knn=normals_knn) if normals_knn else None
    a.setMap(grown, nrm)

    b = amd.ICPSequence(**kw)                                   # resident
    b.setMap(base, b.surfaceNormals(base, knn=normals_knn) if normals_knn else None)
    appended, m, keep_b = b.mapUpdatePointDistance(scan_map, 0.25, normals_knn=normals_knn, return_keep=True)
    assert appended == int(keep.sum()) and m == grown.shape[0] and np.array_equal(keep_b, keep)
    if normals_knn:
        got, got_n = b.getMap(with_normals=True)
        assert np.array_equal(got_n, nrm)
    else:
        got = b.getMap()
    assert np.array_equal(got, grown)
    Ta, Tb = a(sc["scan"]), b(sc["scan"])
    assert np.array_equal(Ta, Tb)
    # a second, shifted scan: same decision as the composed path (exact duplicates count as self matches and are
    # ignored by the search, PointDistanceMapperModule.cpp:36 passes optionFlags = 0)
    scan2 = scan_map.copy(); scan2[:, :3] += np.float32(0.1)
    keep2 = a.pointDistanceKeep(grown, scan2, 0.25)
    appended2, m2 = b.mapUpdatePointDistance(scan2, 0.25, normals_knn=normals_knn)
    assert appended2 == int(keep2.sum()) and m2 == m + appended2
    assert np.array_equal(b.getMap()[m:], scan2[keep2])
    # a handle without a map: the scan becomes the map (PointDistanceMapperModule::createMap)
    c = amd.ICPSequence(**kw)
    app3, m3 = c.mapUpdatePointDistance(scan_map, 0.25, normals_knn=normals_knn)
    assert app3 == m3 == scan_map.shape[0] and np.array_equal(c.getMap(), scan_map)


def test_staged_scan_process_input_equals_composed_path(amd, mid_scene):
    """icpmi_register_prior + icpmi_map_update_staged (one upload per processInput) against transform / register /
    transform / map update composed from the host-pointer entry points: same correction, same grown map."""
    sc = mid_scene
    rng = np.random.default_rng(2)
    base = sc["map"][::2]
    prior = amd.synth.make_T((0.004, -0.003, 0.006), (0.05, -0.04, 0.02)).astype(np.float32)
    # the scan as the sensor sees it: undo the prior on the (already displaced) synthetic scan
    scan_sensor = sc["scan"].copy()
    inv = np.linalg.inv(prior.astype(np.float64))
    scan_sensor[:, :3] = (sc["scan"][:, :3].astype(np.float64) @ inv[:3, :3].T + inv[:3, 3]).astype(np.float32)
    kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=15, use_differential=1)

    a = amd.ICPSequence(**kw); a.setMap(base, a.surfaceNormals(base, knn=10))
    in_map = a.transform(prior, scan_sensor)
    corr_a = a(in_map)
    moved = a.transform(corr_a, in_map)
    app_a, m_a, keep_a = a.mapUpdatePointDistance(moved, 0.2, normals_knn=10, return_keep=True)

    b = amd.ICPSequence(**kw); b.setMap(base, b.surfaceNormals(base, knn=10))
    corr_b = b.registerWithPrior(scan_sensor, prior)
    assert np.array_equal(corr_a, corr_b) and a.stats.iterations == b.stats.iterations
    app_b, m_b, keep_b = b.mapUpdateStaged(corr_b, 0.2, normals_knn=10, return_keep=True)
    assert (app_a, m_a) == (app_b, m_b) and np.array_equal(keep_a, keep_b)
    ma, na = a.getMap(with_normals=True); mb, nb = b.getMap(with_normals=True)
    assert np.array_equal(ma, mb) and np.array_equal(na, nb)
    with pytest.raises(amd.InvalidParameter):
        amd.ICPSequence(**kw).mapUpdateStaged(np.eye(4), 0.2)      # nothing staged


def test_sharded_mapper_single_rank(amd, mid_scene):
    """The map-growth epoch of SURVEY 8(e) with the GPU operators (world size 1: no process group)."""
    sc = mid_scene
    icp = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=30, use_differential=1)
    from norlab_icp_mapper_amd.dist import ShardedMapper
    mapper = ShardedMapper(ShardedMapper.gpu_backend(icp), min_dist_new_point=0.3, normals_knn=10)
    half = sc["map"][::2]
    mapper.set_map(half)
    m0 = mapper.map.shape[0]
    pose, mine, appended = mapper.epoch(sc["scan"], np.eye(4))
    dt, dr = amd.synth.pose_error(pose, sc["T_gt"])
    assert dt < 2e-2 and dr < 2e-3, (dt, dr)
    assert mapper.map.shape[0] == m0 + appended and 0 < appended == mine   # one rank: nothing to merge against
    # accepted points are at least min_dist from the old map
    new_pts = mapper.map[m0:]
    assert icp.pointDistanceKeep(half, new_pts, 0.3).all()
    # the same scan again contributes (almost) nothing
    _, mine2, appended2 = mapper.epoch(sc["scan"], np.eye(4))
    assert appended2 < 0.1 * appended + 5


def test_sharded_mapper_resident_backend_matches_host_backend(amd, mid_scene):
    """The same epochs with the map resident in HBM (register_prior / staged keep / device-side append): same poses, same
    accepted sets as the host-pointer backend up to the rounding of how the scan is placed (one 4x4 in numpy there, prior
    then correction on the device here), and the replica downloaded from the device is the concatenation the host keeps."""
    from norlab_icp_mapper_amd.dist import ShardedMapper
    sc = mid_scene
    kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=30, use_differential=1)
    half = sc["map"][::2]
    host = ShardedMapper(ShardedMapper.gpu_backend(amd.ICPSequence(**kw)), min_dist_new_point=0.3, normals_knn=10)
    ricp = amd.ICPSequence(**kw)
    res = ShardedMapper(ShardedMapper.resident_backend(ricp), min_dist_new_point=0.3, normals_knn=10)
    host.set_map(half); res.set_map(half)
    scans = [sc["scan"], amd.synth.make_scene(m=8, n=20000, seed_scan=77)["scan"]]
    for scan in scans:
        ph, mine_h, app_h = host.epoch(scan, np.eye(4))
        pr, mine_r, app_r = res.epoch(scan, np.eye(4))
        dt, dr = amd.synth.pose_error(ph, pr)
        assert dt < 1e-6 and dr < 1e-6, (dt, dr)
        # both backends place the scan by the same two device transforms (prior, then correction: Mapper.cpp:197, :221) and
        # decide against the same map: the accepted sets are identical -- no allowance
        assert mine_h == mine_r and app_h == app_r, (mine_h, mine_r, app_h, app_r)
    got = res.get_map()
    assert got.shape[0] == res._resident_points and np.array_equal(got[: half.shape[0]], half)
    assert np.array_equal(got, host.map)
    # the keep decision itself, against the composed operator on the same placed cloud: identical masks
    corr = ricp.registerWithPrior(scans[0], np.eye(4))
    mask, placed = ricp.stagedPointDistanceKeep(corr, 0.3)
    ref = ricp.pointDistanceKeep(got, placed, 0.3)
    assert (mask != ref).sum() <= 2          # centred index here, raw coordinates there: a tie on the threshold may flip
    assert np.array_equal(ricp.getMap(), got)                                   # and the map was not touched


@pytest.fixture(scope="module")
def bundled():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bundled_scans.npz"))
    s0, s1 = _bundled_input_filters(g["scan0_xyz"]), _bundled_input_filters(g["scan1_xyz"])
    return {"s0": s0, "s1": s1, "T0": _quat_T(g["trajectory"][0]), "T1": _quat_T(g["trajectory"][1])}


def test_bundled_example_chain_identity_known_answer(amd, oracle, bundled):
    """The shipped config (KDTreeMatcher knn 6 maxDist 2, IdentityErrorMinimizer, Counter 10): ICP runs
    10 matching passes and returns identity, so the corrected pose equals the prior (Mapper.cpp:215)."""
    b = bundled
    icp = amd.ICPSequence(minimizer=0, knn=6, max_dist=2.0, epsilon=1.0, outliers=[], max_iterations=10)
    assert 36000 < b["s0"].shape[0] < 38000                       # ~36.9 k points survive the input filters
    map0 = icp.transform(b["T0"], b["s0"])                         # first scan becomes the map (Mapper.cpp:200-207)
    assert icp.setMap(map0)
    inp = icp.transform(b["T1"], b["s1"])                          # Mapper.cpp:197
    T = icp(inp)
    assert np.array_equal(T, np.eye(4, dtype=np.float32)) and icp.stats.iterations == 10
    oicp = oracle.OracleICP(oracle.make_config(minimizer=0, knn=6, max_dist=2.0, max_iterations=10, nthreads=8))
    oicp.setMap(map0)
    err, T_ref = oicp(inp)
    assert err == 0 and icp.stats.pairs == oicp.stats.pairs
    assert abs(icp.errorMinimizer.getOverlap() - oicp.stats.weighted_point_used_ratio) < 1e-6


def test_bundled_scans_point_to_plane_parity(amd, oracle, bundled):
    """Real lidar geometry (config 4 flavour): point-to-plane, epsilon 0, normals from the knn-10
    surface-normal operator; HIP pose vs oracle pose on the same inputs."""
    b = bundled
    kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
    icp = amd.ICPSequence(**kw)
    map0 = icp.transform(b["T0"], b["s0"])
    normals = oracle.surface_normals(map0, knn=10, nthreads=8)     # same normals for both sides
    assert icp.setMap(map0, normals)
    inp = icp.transform(b["T1"], b["s1"])
    T = icp(inp)
    oicp = oracle.OracleICP(oracle.make_config(nthreads=8, **kw))
    oicp.setMap(map0, normals)
    err, T_ref = oicp(inp)
    assert err == 0 and icp.stats.iterations == oicp.stats.iterations and icp.stats.pairs == oicp.stats.pairs
    dt, dr = amd.synth.pose_error(T, T_ref)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)
    # the robot is quasi-static: the correction is small
    dt0, dr0 = amd.synth.pose_error(T, np.eye(4))
    assert dt0 < 0.2 and dr0 < 0.05
    # GPU surface normals agree with the oracle's on real data (unoriented)
    n_gpu = icp.surfaceNormals(map0, knn=10)
    dots = np.abs(np.sum(n_gpu.astype(np.float64) * normals.astype(np.float64), axis=1))
    assert np.quantile(dots, 0.02) > 1 - 1e-5


def test_full_size_properties(amd):
    """BASELINE sizes (100 k vs 1 M) through size-independent properties: the squared distances of a
    registration are consistent with an independent kNN call, matches are symmetric under a rigid
    motion of both clouds, and the trimmed ratio is met exactly."""
    sc = amd.synth.make_scene()
    icp = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=6)
    assert icp.setMap(sc["map"], sc["normals"])
    T = icp(sc["scan"])
    assert icp.stats.iterations == 6
    n_valid = icp.stats.pairs / 0.85
    assert abs(icp.stats.pairs - int(np.float32(round(n_valid)) * np.float32(0.85)) - 1) <= 2
    mean = icp.getMapMean()
    q = sc["scan"].copy(); q[:, :3] -= mean
    ids, d2 = icp.knn(q, k=1, max_dist=2.0)
    mapc = sc["map"].copy(); mapc[:, :3] -= mean
    sel = ids[:, 0] >= 0
    chk = ((q[sel, :3].astype(np.float64) - mapc[ids[sel, 0], :3].astype(np.float64)) ** 2).sum(1)
    np.testing.assert_allclose(d2[sel, 0], chk, rtol=2e-4, atol=1e-9)
    # moving map and scan by the same rigid motion moves the answer by conjugation
    M = amd.synth.make_T((0.2, 0.1, -0.3), (5.0, -3.0, 1.0)).astype(np.float32)
    icp2 = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=6)
    map2, n2 = icp2.transform(M, sc["map"], sc["normals"])
    icp2.setMap(map2, n2)
    T2 = icp2(icp2.transform(M, sc["scan"]))
    Tc = M.astype(np.float64) @ T.astype(np.float64) @ np.linalg.inv(M.astype(np.float64))
    dt, dr = amd.synth.pose_error(T2, Tc)
    assert dt < 2e-3 and dr < 2e-4, (dt, dr)
    # and the registration recovers the ground truth of the scene
    dt, dr = amd.synth.pose_error(T, sc["T_gt"])
    assert dt < 5e-3 and dr < 5e-4


def test_sharded_mapper_device_backend_equals_resident_backend(amd, mid_scene):
    """The epoch inside the library (icpmi_staged_merge_allgather: compaction, RCCL all-gather on the handle's stream -- here a
    one-rank communicator, the only kind one GPU allows --, rank-ordered merge, append, normals, index) against the resident
    backend that moves the accepted points through the host: same poses, same accepted counts, the same map bit for bit."""
    from norlab_icp_mapper_amd.dist import ShardedMapper
    sc = mid_scene
    kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=30, use_differential=1)
    half = sc["map"][::2]
    ricp, dicp = amd.ICPSequence(**kw), amd.ICPSequence(**kw)
    dicp.commInit(amd.ICPSequence.commUniqueId(), 1, 0)
    res = ShardedMapper(ShardedMapper.resident_backend(ricp), min_dist_new_point=0.3, normals_knn=10)
    dev = ShardedMapper(ShardedMapper.device_backend(dicp), min_dist_new_point=0.3, normals_knn=10)
    res.set_map(half); dev.set_map(half)
    for scan in (sc["scan"], amd.synth.make_scene(m=8, n=20000, seed_scan=77)["scan"]):
        pr, mine_r, app_r = res.epoch(scan, np.eye(4))
        pd, mine_d, app_d = dev.epoch(scan, np.eye(4))
        assert np.array_equal(pr, pd) and (mine_r, app_r) == (mine_d, app_d)
    assert np.array_equal(res.get_map(), dev.get_map())
    m_before = dev.get_map().shape[0]
    acc, app, m1, merged = dicp.stagedMergeAllGather(np.eye(4, dtype=np.float32), 0.3, normals_knn=0, return_merged=True)
    assert acc == app == merged.shape[0] and m1 == m_before + app       # the staged scan offered once more, placed by another transform
    assert np.array_equal(dev.get_map()[m_before:], merged)
    dicp.commDestroy()


def test_sharded_mapping_example_over_rccl(tmp_path):
    """examples/sharded_mapping.py under torch.distributed.run with one rank: process group over RCCL (backend nccl),
    device-tensor all-gather of the accepted points, identical rebuild -- the N > 1 code path on the GPU that is here."""
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "examples", "sharded_mapping.py"), "--map-points", "100000",
                          "--scan-points", "8000", "--epochs", "2"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "epoch 1:" in out.stdout and "scans/s" in out.stdout


def test_fused_input_filters_bit_exact(amd, oracle, mid_scene):
    """icpmi_filter_points (Mapper::applyInputFilters as one pass) against the oracle's predicates, same keep mask."""
    rng = np.random.default_rng(21)
    c = mid_scene["scan"].copy()
    c[:200, :3] = rng.uniform(-2, 2, (200, 3)).astype(np.float32)           # points inside the robot-body boxes
    c[200, :3] = (0.5, 0.0, 0.0)                                            # on a face
    c[201, :3] = np.nan                                                     # NaN fails every comparison alike
    icp = amd.ICPSequence(minimizer=0)
    shipped = [("distance_limit", -1, 40.0, False), ("bounding_box", (-1.5, -1, -1), (0.5, 1, 0.5), True),
               ("bounding_box", (-6, -2.5, -1), (-1.5, 2.5, 1), True)]
    for filters in (shipped, [("distance_limit", 1, 12.5, True)], [("bounding_box", (-30, -30, 0.5), (30, 30, 9), False)] * 16, []):
        got = icp.filterPoints(c, filters)
        assert np.array_equal(got, oracle.filter_points(c, filters))
    assert 0 < icp.filterPoints(c, shipped).sum() < c.shape[0]
    assert icp.filterPoints(c[:0], shipped).shape == (0,)
    with pytest.raises(amd.InvalidParameter):
        icp.filterPoints(c, [("distance_limit", 3, 1.0, False)])
    with pytest.raises(amd.InvalidParameter):
        icp.filterPoints(c, [("distance_limit", -1, 1.0, False)] * 17)


def test_inspector_style_json_stats(amd, mid_scene, tmp_path):
    """ICPMI_STATS_JSON: one line per registration with libpointmatcher's inspector names (read once per process: subprocess)"""
    import json, os, subprocess, sys
    path = tmp_path / "stats.jsonl"
    code = ("import norlab_icp_mapper_amd as pkg\n"
            "sc = pkg.synth.make_scene(m=50000, n=5000)\n"
            "icp = pkg.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)\n"
            "icp.setMap(sc['map'], sc['normals'])\n"
            "for _ in range(3): icp(sc['scan'])\n"
            "print(icp.stats.iterations, icp.stats.pairs)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, ICPMI_STATS_JSON=str(path), PYTHONPATH=root), text=True)
    its, pairs = (int(v) for v in out.split()[-2:])
    lines = [json.loads(l) for l in open(path)]
    assert len(lines) == 3
    for rec in lines:
        assert rec["IterationsCount"] == its and rec["PairsUsed"] == pairs and rec["ReadingPoints"] == 5000 and rec["Status"] == 0
        assert 0 < rec["OverlapRatio"] <= 1 and rec["ConvergenceDuration"] > 0 and rec["StopReason"] in (1, 2)
