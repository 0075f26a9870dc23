"""GenericDescriptorOutlierFilter, RobustOutlierFilter and PointToPlaneErrorMinimizer{force4DOF} on the HIP path against the
oracle (the chain elements upstream's registrar offers beyond the bundled configurations).  Weights are
compared bit for bit (identically specified float arithmetic; exp / pow go through double on both sides), poses within the
1e-4 m / 1e-4 rad of BASELINE.json's north_star."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4

GEN, ROB = 6, 7
SOFT, LARGER = 2, 4
FCT = {"cauchy": 0, "welsch": 1, "sc": 2, "gm": 3, "tukey": 4, "huber": 5, "L1": 6, "student": 7}


def rob(fct, tuning, scale="none", nb=0, dist="point2point", approximation=0.0):
    return (ROB, float(tuning), FCT[fct] | ({"none": 0, "mad": 1, "berg": 2, "std": 3}[scale] << 4) | ({"point2point": 0, "point2plane": 1}[dist] << 8),
            float(nb), float(approximation))


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


def centred(cloud, mean):
    out = cloud.copy()
    out[:, :3] = cloud[:, :3] - mean[None, :]
    return out


def scalar_of(sc):
    # a per-point descriptor with structure and ties at the thresholds used below
    rng = np.random.default_rng(5)
    s = rng.random(sc["map"].shape[0]).astype(np.float32)
    s[::7] = 0.5
    return s


@pytest.mark.parametrize("flags,thr", [(LARGER, 0.5), (0, 0.5), (SOFT, 0.0), (SOFT | LARGER, 0.3)])
def test_generic_descriptor_weights_exact(amd, oracle, small_scene, flags, thr):
    sc = small_scene
    chain = [(GEN, thr, flags, 0.0), (4, 0.9)]
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=chain)
    icp.setMap(sc["map"])
    s = scalar_of(sc)
    icp.setMapScalar(s)
    mean = icp.getMapMean()
    ids, d2 = icp.knn(centred(sc["scan"], mean), k=3, max_dist=0.6)  # leaves unmatched slots (id -1)
    assert (ids < 0).any()
    w, lim = icp.outlierWeights(d2, ids)
    err, rw, rlim = oracle.outlier_weights(oracle.make_config(outliers=chain), d2, ids, ref_scalar=s)
    assert err == 0 and lim == rlim
    assert np.array_equal(w, rw)
    assert 0 < np.count_nonzero(w) < w.size


def test_generic_descriptor_needs_the_scalar(amd, small_scene):
    sc = small_scene
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=[(GEN, 0.5, LARGER, 0.0)])
    icp.setMap(sc["map"])
    with pytest.raises(Exception, match="InvalidField|scalar"):
        icp(sc["scan"])


@pytest.mark.parametrize("fct", list(FCT))
@pytest.mark.parametrize("scale", ["none", "mad"])
def test_robust_weights_exact(amd, oracle, small_scene, fct, scale):
    sc = small_scene
    tuning = {"sc": 0.02, "gm": 0.05, "student": 1.5}.get(fct, 0.1 if scale == "none" else 1.2)
    chain = [rob(fct, tuning, scale)]
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=chain)
    icp.setMap(sc["map"])
    mean = icp.getMapMean()
    ids, d2 = icp.knn(centred(sc["scan"], mean), k=2, max_dist=0.8)
    w, lim = icp.outlierWeights(d2, ids)
    err, rw, rlim = oracle.outlier_weights(oracle.make_config(outliers=chain), d2, ids)
    assert err == 0
    if scale == "mad":
        assert lim == rlim and lim > 0  # the scale estimate itself
    assert np.array_equal(w, rw), (fct, scale, np.abs(w - rw).max())
    assert np.isfinite(w).all() and w.max() > 0
    # every M-estimator down-weights the far matches (L1 / Huber / ... are monotone in e2)
    fin = np.isfinite(d2)
    near, far = w[fin & (d2 < np.quantile(d2[fin], 0.2))].mean(), w[fin & (d2 > np.quantile(d2[fin], 0.8))].mean()
    assert near >= far
    assert np.all(w[~fin] == 0)


@pytest.mark.parametrize("fct", ["cauchy", "tukey", "huber", "welsch"])
@pytest.mark.parametrize("scale,approximation", [("berg", 0.0), ("std", 0.0), ("mad", 1.5), ("berg", 0.8), ("none", 0.3)])
def test_robust_berg_std_approximation_weights_exact(amd, oracle, small_scene, fct, scale, approximation):
    """r5: the scale estimators berg / std and `approximation` (rejected until r4).  Stage call = iteration 1: berg's scale is
    1.9 sqrt(median d2), std's the square root of the standard deviation of every entry (all finite here: the search radius holds
    every query's k neighbours)."""
    sc = small_scene
    tuning = 0.05 if scale == "berg" else (0.1 if scale == "none" else 1.2)      # berg: the scale the estimate converges to
    chain = [rob(fct, tuning, scale, approximation=approximation)]
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=chain)
    icp.setMap(sc["map"])
    mean = icp.getMapMean()
    ids, d2 = icp.knn(centred(sc["scan"], mean), k=2, max_dist=50.0)
    assert np.isfinite(d2).all()
    w, lim = icp.outlierWeights(d2, ids)
    err, rw, rlim = oracle.outlier_weights(oracle.make_config(outliers=chain), d2, ids)
    assert err == 0
    if scale == "berg":
        pos = np.sort(d2[d2 > 0]); med = pos[int(np.float32(pos.size) * np.float32(0.5))]
        assert lim == np.float32(1.9 * float(np.sqrt(np.float32(med))))
    if scale == "std":
        assert lim == pytest.approx(np.sqrt(np.std(d2.astype(np.float64), ddof=1)), rel=2e-6)
    if scale != "none":
        assert lim == rlim and lim > 0
    assert np.array_equal(w, rw), (fct, scale, np.abs(w - rw).max())
    assert np.isfinite(w).all() and w.max() > 0
    if approximation:
        e2 = d2 / np.float32(lim if scale != "none" else 1.0) ** 2
        cut = e2 >= np.float32(approximation) ** 2
        assert 0 < cut.sum() < cut.size and np.all(w[cut] == 0) and np.any(w[~cut] > 0)   # (tukey / welsch reach 0 on their own too)


def test_robust_std_with_an_unmatched_entry_is_not_a_number(amd, oracle, small_scene):
    """Matches::getStandardDeviation runs over EVERY entry of the distance matrix: one infinite entry (maxDist) makes the scale NaN
    upstream; both sides of this repository end the registration with "not a number" then."""
    sc = small_scene
    kw = dict(minimizer=1, max_dist=0.3, knn=3, outliers=[rob("cauchy", 1.0, "std")], max_iterations=5)
    icp = amd.ICPSequence(**kw)
    icp.setMap(sc["map"], sc["normals"])
    with pytest.raises(amd.ConvergenceError, match="not a number|NaN|nan"):
        icp(sc["scan"])
    o = oracle.OracleICP(oracle.make_config(**kw)); o.setMap(sc["map"], sc["normals"])
    err, _ = o(sc["scan"])
    assert err == 4   # ORC_ERR_NAN


VT = 8


@pytest.mark.parametrize("params", [(0.05, 0.99, 0.95), (0.3, 0.6, 0.95), (0.05, 0.99, 2.5), (0.7, 0.99, 0.2)])
@pytest.mark.parametrize("k,max_dist", [(1, 2.0), (3, 0.6)])
def test_var_trimmed_weights_exact(amd, oracle, small_scene, params, k, max_dist):
    sc = small_scene
    chain = [(VT, params[0], 0, params[1], params[2])]
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=chain)
    icp.setMap(sc["map"])
    mean = icp.getMapMean()
    # a reading with outliers: a fifth of the points pushed off the surfaces
    q = centred(sc["scan"], mean)
    q[::5, :3] += np.random.default_rng(8).normal(0, 0.4, (q[::5].shape[0], 3)).astype(np.float32)
    ids, d2 = icp.knn(q, k=k, max_dist=max_dist)
    w, lim = icp.outlierWeights(d2, ids)
    err, rw, rlim = oracle.outlier_weights(oracle.make_config(outliers=chain), d2, ids)
    assert err == 0
    assert lim == rlim, (lim, rlim, oracle.var_trimmed_ratio(d2, *params))
    assert np.array_equal(w, rw)
    assert 0 < np.count_nonzero(w) < w.size


def test_var_trimmed_no_valid_match_raises(amd, small_scene):
    sc = small_scene
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=[(VT, 0.05, 0, 0.99, 0.95)])
    icp.setMap(sc["map"])
    far = sc["scan"].copy(); far[:, :3] += 500.0
    with pytest.raises(amd.ConvergenceError):
        icp(far)
    with pytest.raises(Exception):
        amd.ICPSequence(minimizer=1, outliers=[(VT, 0.9, 0, 0.5, 0.95)])  # minRatio > maxRatio


CHAINS = {
    "p2plane_cauchy_mad": dict(minimizer=2, max_dist=2.0, outliers=[rob("cauchy", 1.0, "mad")], max_iterations=25, use_differential=1),
    "p2plane_huber_mad_nb3_plane": dict(minimizer=2, max_dist=2.0, outliers=[rob("huber", 1.5, "mad", 3, "point2plane")], max_iterations=12),
    "p2p_welsch_none_trim": dict(minimizer=1, max_dist=2.0, outliers=[(4, 0.9), rob("welsch", 0.3)], max_iterations=15),
    "p2plane_tukey_plane": dict(minimizer=2, max_dist=1.0, outliers=[rob("tukey", 0.5, "none", 0, "point2plane")], max_iterations=12),
    "p2plane_cauchy_berg": dict(minimizer=2, max_dist=50.0, outliers=[rob("cauchy", 0.05, "berg")], max_iterations=25, use_differential=1),
    "p2p_tukey_berg_nb4_apx": dict(minimizer=1, max_dist=50.0, outliers=[rob("tukey", 0.1, "berg", 4, approximation=5.0)], max_iterations=12),
    "p2plane_huber_std_plane": dict(minimizer=2, max_dist=50.0, outliers=[rob("huber", 1.0, "std", 0, "point2plane")], max_iterations=12),
    "p2plane_welsch_std_nb2_trim": dict(minimizer=2, max_dist=50.0, knn=2, outliers=[(4, 0.9), rob("welsch", 2.0, "std", 2)], max_iterations=10),
    "p2plane_cauchy_mad_apx": dict(minimizer=2, max_dist=2.0, outliers=[rob("cauchy", 1.0, "mad", approximation=2.0)], max_iterations=25, use_differential=1),
    "p2plane_generic_trim": dict(minimizer=2, max_dist=2.0, outliers=[(GEN, 0.25, LARGER, 0.0), (4, 0.85)], max_iterations=20, use_differential=1),
    "p2plane_generic_soft": dict(minimizer=2, max_dist=2.0, outliers=[(GEN, 0.0, SOFT, 0.0)], max_iterations=10),
    "p2plane_vartrimmed": dict(minimizer=2, max_dist=2.0, outliers=[(VT, 0.05, 0, 0.99, 0.95)], max_iterations=25, use_differential=1),
    "p2p_vartrimmed_knn3_maxdist": dict(minimizer=1, knn=3, max_dist=1.0, outliers=[(1, 0.8), (VT, 0.2, 0, 0.9, 1.5)], max_iterations=8),
    "p2plane_2d": dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=30, use_differential=1, force_2d=1),
    "p2plane_4dof": dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=30, use_differential=1, force_4dof=1),
}


@pytest.mark.parametrize("name", list(CHAINS))
def test_registration_matches_oracle(amd, oracle, mid_scene, name):
    sc = mid_scene
    kw = dict(CHAINS[name])
    icp = amd.ICPSequence(**kw)
    assert icp.setMap(sc["map"], sc["normals"])
    s = scalar_of(sc)
    if "generic" in name:
        icp.setMapScalar(s)
    T = icp(sc["scan"])
    okw = dict(kw); okw["nthreads"] = 8
    oicp = oracle.OracleICP(oracle.make_config(**okw))
    oicp.setMap(sc["map"], sc["normals"])
    if "generic" in name:
        oicp.setMapScalar(s)
    err, T_ref = oicp(sc["scan"])
    assert err == 0
    assert icp.stats.iterations == oicp.stats.iterations
    assert icp.stats.stop_reason == oicp.stats.stop_reason
    assert icp.stats.pairs == oicp.stats.pairs
    dt, dr = amd.synth.pose_error(T, T_ref)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)
    assert abs(icp.errorMinimizer.getOverlap() - oicp.stats.weighted_point_used_ratio) < 2e-6
    if "vartrimmed" in name:
        assert icp.stats.trimmed_limit == oicp.stats.trimmed_limit
    if name == "p2plane_2d":
        # yaw + (tx, ty): no motion along z, z axis kept
        assert T[2, 3] == 0 and np.allclose(T[2, :3], [0, 0, 1], atol=1e-6) and np.allclose(T[:3, 2], [0, 0, 1], atol=1e-6)
    if name in ("p2plane_4dof", "p2plane_vartrimmed", "p2plane_2d"):
        # fixed launch sequence (hipGraph) == checked loop
        import torch
        d = torch.from_numpy(np.ascontiguousarray(sc["scan"], dtype=np.float32)).cuda()
        its = icp.stats.iterations
        assert np.array_equal(icp.registerDev(d.data_ptr(), d.shape[0], fixed_iterations=its), T)
    if name == "p2plane_4dof":
        # yaw + translation only: the z axis stays the z axis in every step, hence in the product
        assert np.allclose(T[2, :3], [0, 0, 1], atol=1e-6) and np.allclose(T[:3, 2], [0, 0, 1], atol=1e-6)


def test_force_4dof_single_step_matches_oracle(amd, oracle, small_scene):
    sc = small_scene
    icp = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], force_4dof=1)
    icp.setMap(sc["map"], sc["normals"])
    mean = icp.getMapMean()
    mapc = centred(sc["map"], mean)
    q = centred(sc["scan"], mean)
    T_iter = amd.synth.make_T((0.0, 0.0, 0.01), (0.02, 0.01, -0.01)).astype(np.float32)
    T_step, sums = icp.minimizeStep(q, T_iter)
    step = oracle.transform(T_iter, q)
    ids, d2 = oracle.knn(mapc, step, k=1, max_dist=2.0)
    err, w, lim = oracle.outlier_weights(oracle.make_config(max_dist=2.0, outliers=[(4, 0.85)]), d2, ids)
    err, T_ref, A, b, x, st = oracle.minimize(2, step, mapc, sc["normals"], ids, d2, w, force_4dof=1)
    assert err == 0 and x[0] == 0 and x[1] == 0
    # the 4 x 4 system is the {2..5} block of the 6-DOF sums
    x4 = oracle.solve_n(A[2:, 2:].astype(np.float32), b[2:].astype(np.float32))
    assert np.array_equal(x4, x[2:])
    np.testing.assert_allclose(np.linalg.solve(A[2:, 2:], b[2:]), x4, rtol=2e-3, atol=1e-6)
    dt, dr = amd.synth.pose_error(T_step, T_ref)
    assert dt < 1e-5 and dr < 1e-5
    assert np.allclose(T_step[2, :3], [0, 0, 1], atol=1e-7)


def test_force_2d_single_step_matches_oracle(amd, oracle, small_scene):
    sc = small_scene
    icp = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], force_2d=1)
    icp.setMap(sc["map"], sc["normals"])
    mean = icp.getMapMean()
    mapc = centred(sc["map"], mean)
    q = centred(sc["scan"], mean)
    T_iter = amd.synth.make_T((0.0, 0.0, 0.01), (0.02, 0.01, 0.0)).astype(np.float32)
    T_step, sums = icp.minimizeStep(q, T_iter)
    step = oracle.transform(T_iter, q)
    ids, d2 = oracle.knn(mapc, step, k=1, max_dist=2.0)
    err, w, lim = oracle.outlier_weights(oracle.make_config(max_dist=2.0, outliers=[(4, 0.85)]), d2, ids)
    err, T_ref, A, b, x, st = oracle.minimize(2, step, mapc, sc["normals"], ids, d2, w, force_4dof=2)
    assert err == 0
    dt, dr = amd.synth.pose_error(T_step, T_ref)
    assert dt < 1e-5 and dr < 1e-5
    assert T_step[2, 3] == 0 and np.allclose(T_step[2, :3], [0, 0, 1], atol=1e-7)
    with pytest.raises(Exception):
        amd.ICPSequence(minimizer=2, force_2d=1, force_4dof=1)


def test_batch_with_ext_chain_equals_single_registrations(amd, mid_scene):
    # a batch whose chain holds a Robust filter runs reading by reading: same results as the calls one after the other
    sc = mid_scene
    kw = dict(CHAINS["p2plane_cauchy_mad"])
    icp = amd.ICPSequence(**kw)
    icp.setMap(sc["map"], sc["normals"])
    import torch
    scans = [sc["scan"], sc["scan"][::2].copy(), sc["scan"][1::3].copy()]
    dev = [torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda() for x in scans]
    singles = [icp.registerDev(d.data_ptr(), d.shape[0]).copy() for d in dev]
    Ts, stats, status = icp.registerBatchDev([d.data_ptr() for d in dev], [d.shape[0] for d in dev])
    for b in range(3):
        assert status[b] == 0
        assert np.array_equal(Ts[b], singles[b])


@pytest.mark.parametrize("minimizer", [1, 2])
@pytest.mark.parametrize("knn", [1, 3])
def test_sensor_noise_overlap_matches_oracle(amd, oracle, mid_scene, minimizer, knn):
    """ErrorMinimizer::getOverlap() for a reading that carries `simpleSensorNoise` and `normals` (read at Mapper.cpp:219, drives the
    `overlap` update condition, Mapper.cpp:257-260): the share of the last iteration's pairs within the sensor noise, not the
    weighted ratio (VERDICT r2 missing 3).  Device pass (icpmi_set_reading_sensor_noise) against the oracle's restatement."""
    sc = mid_scene
    n = sc["scan"].shape[0]
    rng = np.random.default_rng(11)
    noise = rng.uniform(0.002, 0.03, n).astype(np.float32)               # e.g. what SimpleSensorNoiseDataPointsFilter writes per beam
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    kw = dict(minimizer=minimizer, knn=knn, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=12, use_differential=1)
    icp = amd.ICPSequence(**kw)
    assert icp.setMap(sc["map"], sc["normals"])
    icp.setReadingSensorNoise(noise)
    T = icp(sc["scan"], nrm)
    o = oracle.OracleICP(oracle.make_config(nthreads=8, **kw))
    o.setMap(sc["map"], sc["normals"])
    o.setReadingNoise(noise)
    err, T_ref = o(sc["scan"], nrm)
    assert err == 0 and icp.stats.iterations == o.stats.iterations and icp.stats.pairs == o.stats.pairs
    assert 0.0 < o.stats.sensor_noise_overlap < 1.0
    # counts of pairs: identical unless a pair sits within rounding of its threshold (the two sides sum the mean in a different order)
    assert abs(icp.stats.sensor_noise_overlap - o.stats.sensor_noise_overlap) <= 2.0 / max(o.stats.pairs, 1)
    assert icp.errorMinimizer.getOverlap() == pytest.approx(icp.stats.sensor_noise_overlap)
    assert abs(icp.errorMinimizer.getOverlap() - icp.stats.weighted_point_used_ratio) > 1e-3   # it is NOT the weighted ratio
    # one shot: the next registration (no noise handed over) answers with the weighted ratio again
    icp(sc["scan"], nrm)
    assert icp.stats.sensor_noise_overlap == -1.0
    assert icp.errorMinimizer.getOverlap() == pytest.approx(icp.stats.weighted_point_used_ratio)
    # and the fixed-count graph path keeps the pose it would have had without the overlap pass
    dt, dr = amd.synth.pose_error(T, T_ref)
    assert dt <= 1e-4 and dr <= 1e-4


def test_sensor_noise_overlap_p2p_needs_no_reading_normals(amd, oracle, mid_scene):
    """PointToPointErrorMinimizer::getOverlap() needs `simpleSensorNoise` alone: a reading without normals still gets the
    sensor-noise count; the point-to-plane variant without reading normals falls back to the weighted ratio (-1)."""
    sc = mid_scene
    n = sc["scan"].shape[0]
    noise = np.random.default_rng(12).uniform(0.002, 0.03, n).astype(np.float32)
    kw = dict(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=12, use_differential=1)
    icp = amd.ICPSequence(**kw)
    assert icp.setMap(sc["map"], sc["normals"])
    icp.setReadingSensorNoise(noise)
    icp(sc["scan"])
    o = oracle.OracleICP(oracle.make_config(nthreads=8, **kw))
    o.setMap(sc["map"], sc["normals"])
    o.setReadingNoise(noise)
    err, _ = o(sc["scan"])
    assert err == 0 and icp.stats.iterations == o.stats.iterations and icp.stats.pairs == o.stats.pairs
    assert 0.0 < o.stats.sensor_noise_overlap < 1.0
    assert abs(icp.stats.sensor_noise_overlap - o.stats.sensor_noise_overlap) <= 2.0 / max(o.stats.pairs, 1)
    kw2 = dict(kw, minimizer=2)
    icp2 = amd.ICPSequence(**kw2)
    assert icp2.setMap(sc["map"], sc["normals"])
    icp2.setReadingSensorNoise(noise)
    icp2(sc["scan"])
    assert icp2.stats.sensor_noise_overlap == -1.0
    assert icp2.errorMinimizer.getOverlap() == pytest.approx(icp2.stats.weighted_point_used_ratio)


def test_sensor_noise_sentinel_on_every_stats_path(amd, mid_scene):
    """stats.sensor_noise_overlap is -1 ("not computed") on every path that fills stats -- minimizeStep, the batch entry, a handle
    without a map -- so that getOverlap() falls back to the weighted ratio and never reads a zeroed field."""
    import torch
    sc = mid_scene
    icp = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=6)
    icp(sc["scan"])                                        # no map: identity
    assert icp.stats.sensor_noise_overlap == -1.0
    assert icp.setMap(sc["map"], sc["normals"])
    icp.minimizeStep(sc["scan"] - np.r_[icp.getMapMean(), 0].astype(np.float32))
    assert icp.stats.sensor_noise_overlap == -1.0
    assert icp.errorMinimizer.getOverlap() == pytest.approx(icp.stats.weighted_point_used_ratio) and icp.errorMinimizer.getOverlap() > 0.1
    dev = [torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda() for x in (sc["scan"], sc["scan"][::2].copy())]
    _, stats, status = icp.registerBatchDev([d.data_ptr() for d in dev], [d.shape[0] for d in dev])
    assert not any(status) and all(s.sensor_noise_overlap == -1.0 for s in stats)


def test_set_map_with_then_without_normals_drops_segment_graphs(amd, oracle, mid_scene):
    """A checked loop replays segment graphs; a rebuild of the index -- same cloud, same grid, but normals gone, or another cloud of the
    same size -- must not replay a graph captured for the previous map (only the fixed-count graph was dropped)."""
    sc = mid_scene
    kw = dict(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
    icp = amd.ICPSequence(**kw)
    o = oracle.OracleICP(oracle.make_config(nthreads=8, **kw))
    assert icp.setMap(sc["map"], sc["normals"])
    T0 = icp(sc["scan"])
    assert icp.setMap(sc["map"], None)                     # same points, same grid: normals gone
    T1 = icp(sc["scan"])
    assert np.array_equal(T0, T1)
    shifted = sc["map"].copy(); shifted[:, 0] += np.float32(0.25)   # same size, same extent up to a shift: the mean moves
    assert icp.setMap(shifted, None)
    T2 = icp(sc["scan"])
    o.setMap(shifted, None)
    err, T2_ref = o(sc["scan"])
    assert err == 0 and icp.stats.iterations == o.stats.iterations
    dt, dr = amd.synth.pose_error(T2, T2_ref)
    assert dt <= 1e-4 and dr <= 1e-4


def test_generic_descriptor_source_reading(amd, oracle, mid_scene):
    """GenericDescriptorOutlierFilter{source: reading} (r4, icpmi_set_reading_scalar): the descriptor of the READING point decides.  Known
    answer by numpy (Identity minimiser, one iteration: the pairs are the matched reading points whose descriptor passes), then a full
    point-to-plane registration against the oracle, hard and soft."""
    sc = mid_scene
    n = sc["scan"].shape[0]
    rng = np.random.default_rng(21)
    desc = rng.random(n).astype(np.float32)
    desc[::9] = 0.5
    READ = 1
    icp = amd.ICPSequence(minimizer=0, max_dist=50.0, outliers=[(GEN, 0.5, READ | LARGER, 0.0)], max_iterations=1)
    assert icp.setMap(sc["map"], sc["normals"])
    icp.setReadingScalar(desc)
    icp(sc["scan"])
    assert icp.stats.pairs == int((desc > np.float32(0.5)).sum())
    with pytest.raises(Exception, match="InvalidField|descriptor"):      # one shot: the next reading came without its row
        icp(sc["scan"])
    with pytest.raises(Exception):                                       # the stage entry points carry no reading descriptor
        icp.outlierWeights(np.ones((4, 1), np.float32), np.zeros((4, 1), np.int32))
    for flags in (READ | LARGER, READ, READ | SOFT):
        kw = dict(minimizer=2, max_dist=2.0, outliers=[(GEN, 0.4, flags, 0.0), (4, 0.9)], max_iterations=25, use_differential=1)
        icp = amd.ICPSequence(**kw)
        assert icp.setMap(sc["map"], sc["normals"])
        icp.setReadingScalar(desc)
        T = icp(sc["scan"])
        o = oracle.OracleICP(oracle.make_config(nthreads=8, **kw))
        o.setMap(sc["map"], sc["normals"])
        o.setReadingScalar(desc)
        err, T_ref = o(sc["scan"])
        assert err == 0 and icp.stats.iterations == o.stats.iterations and icp.stats.pairs == o.stats.pairs, (flags, icp.stats.pairs, o.stats.pairs)
        assert icp.stats.weighted_point_used_ratio == pytest.approx(o.stats.weighted_point_used_ratio, rel=1e-6)
        dt, dr = amd.synth.pose_error(T, T_ref)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (flags, dt, dr)


def test_surface_normal_keep_matched_ids_and_mean_dist(amd, oracle):
    """icpmi_surface_normals_ex2: keepMatchedIds / keepMeanDist against the oracle (ids bit-equal, distances to float rounding: the
    index works in a centred frame) and against scipy directly."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(5)
    pts = np.ones((20000, 4), dtype=np.float32)
    pts[:, :3] = rng.uniform(-30, 30, (20000, 3)).astype(np.float32)
    pts[:, 2] *= 0.1
    k = 9
    icp = amd.ICPSequence(minimizer=0)
    n, dens, ids, md = icp.surfaceNormals(pts, knn=k, with_densities=True, with_matched_ids=True, with_mean_dist=True)
    rn, rids, rmd = oracle.surface_normals_extras(pts, knn=k, nthreads=8)
    assert ids.dtype == np.int32 and ids.shape == (20000, k)
    assert np.array_equal(ids, rids)
    np.testing.assert_allclose(md, rmd, rtol=2e-4, atol=2e-5)
    P = pts[:, :3].astype(np.float64)
    _, ref = cKDTree(P).query(P, k=k)
    assert np.array_equal(np.sort(ids, axis=1), np.sort(ref, axis=1))
    # the plain entries give the same normals / densities
    n0, d0 = icp.surfaceNormals(pts, knn=k, with_densities=True)
    assert np.array_equal(n, n0) and np.array_equal(dens, d0)
    # either output alone
    (n1, ids1) = icp.surfaceNormals(pts, knn=k, with_matched_ids=True)
    (n2, md2) = icp.surfaceNormals(pts, knn=k, with_mean_dist=True)
    assert np.array_equal(ids1, ids) and np.array_equal(md2, md) and np.array_equal(n1, n) and np.array_equal(n2, n)
