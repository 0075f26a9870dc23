"""Batched registration (icpmi_register_batch_dev) and the run-ahead eager loop.

Bars: a reading registered in a batch ends on the SAME BITS as the same reading registered alone (the kernels are
shared, blockIdx.y = reading); a batch against the CPU oracle stays within the north-star tolerance like any single
registration; the eager loop that runs ahead of the solve kernel's progress word stops where the periodic read-back
loop stops."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


def _readings(amd, sc, sizes, seed0=700):
    """independent readings of the same scene: different sub-samples of the scan, different extra misalignments"""
    out = []
    for j, n in enumerate(sizes):
        rng = np.random.default_rng(seed0 + j)
        idx = rng.permutation(sc["scan"].shape[0])[:n]
        scan = sc["scan"][idx].copy()
        T = amd.synth.make_T((0.002 * j, -0.001 * j, 0.0015 * j), (0.01 * j, -0.02 * j, 0.005 * j))
        scan[:, :3] = scan[:, :3] @ T[:3, :3].T + T[:3, 3]
        out.append(np.ascontiguousarray(scan, dtype=np.float32))
    return out


CHAINS = {
    "p2p_trimmed": dict(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)]),
    "p2plane_trimmed": dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)]),
    "p2plane_maxdist_median": dict(minimizer=2, max_dist=2.0, outliers=[(1, 1.0), (3, 3.0)]),
    "p2plane_no_filter": dict(minimizer=2, max_dist=1.5, outliers=[]),
}


@pytest.mark.parametrize("name", list(CHAINS))
@pytest.mark.parametrize("fixed", [0, 7])
def test_batch_is_bitwise_the_single_registration(amd, mid_scene, name, fixed):
    import torch
    sc = mid_scene
    kw = dict(CHAINS[name], max_iterations=30, use_differential=1)
    icp = amd.ICPSequence(**kw)
    icp.setMap(sc["map"], sc["normals"])
    n0 = sc["scan"].shape[0]
    scans = _readings(amd, sc, [n0, n0 - 1234, n0 // 3, 777, n0 - 1])
    dev = [torch.from_numpy(s).cuda() for s in scans]
    single = []
    for d in dev:
        T = icp.registerDev(d.data_ptr(), d.shape[0], fixed_iterations=fixed)
        single.append((T, icp.stats.iterations, icp.stats.stop_reason, icp.stats.pairs, icp.stats.trimmed_limit))
    for rep in range(2):                                   # second round: graph replay / warm buffers
        Ts, stats, status = icp.registerBatchDev([d.data_ptr() for d in dev], [d.shape[0] for d in dev], fixed_iterations=fixed)
        assert status == [0] * len(dev)
        for b, (T, it, why, pairs, lim) in enumerate(single):
            assert np.array_equal(Ts[b], T), (name, fixed, b, np.abs(Ts[b] - T).max())
            assert (stats[b].iterations, stats[b].stop_reason, stats[b].pairs) == (it, why, pairs)
            assert stats[b].trimmed_limit == lim or (np.isnan(lim) and np.isnan(stats[b].trimmed_limit))


def test_batch_matches_oracle(amd, oracle, mid_scene):
    import torch
    sc = mid_scene
    kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
    icp = amd.ICPSequence(**kw)
    icp.setMap(sc["map"], sc["normals"])
    n0 = sc["scan"].shape[0]
    scans = _readings(amd, sc, [n0, n0 // 2, n0 - 99])
    dev = [torch.from_numpy(s).cuda() for s in scans]
    Ts, stats, status = icp.registerBatchDev([d.data_ptr() for d in dev], [d.shape[0] for d in dev])
    assert status == [0, 0, 0]
    oicp = oracle.OracleICP(oracle.make_config(nthreads=8, **kw))
    oicp.setMap(sc["map"], sc["normals"])
    for b, s in enumerate(scans):
        err, T_ref = oicp(s)
        assert err == 0
        assert stats[b].iterations == oicp.stats.iterations and stats[b].stop_reason == oicp.stats.stop_reason
        assert stats[b].pairs == oicp.stats.pairs
        dt, dr = amd.synth.pose_error(Ts[b], T_ref)
        assert dt <= 1e-4 and dr <= 1e-4, (b, dt, dr)


def test_batch_error_and_fallback_paths(amd, mid_scene):
    import torch
    sc = mid_scene
    icp = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20, use_differential=1)
    icp.setMap(sc["map"], sc["normals"])
    good = torch.from_numpy(sc["scan"]).cuda()
    far = sc["scan"].copy(); far[:, :3] += 1000.0            # no match within maxDist: "no outlier to filter"
    bad = torch.from_numpy(far).cuda()
    T_alone = icp.registerDev(good.data_ptr(), good.shape[0])
    Ts, stats, status = icp.registerBatchDev([good.data_ptr(), bad.data_ptr(), good.data_ptr()], [good.shape[0]] * 3)
    assert status[0] == 0 and status[2] == 0 and status[1] == 4          # ICPMI_ERR_NO_OUTLIER_TO_FILTER
    assert np.array_equal(Ts[0], T_alone) and np.array_equal(Ts[2], T_alone)
    assert np.array_equal(Ts[1], np.eye(4, dtype=np.float32))
    # a chain the batched kernels do not serve (knn 3) runs the readings one after the other: same results
    icp3 = amd.ICPSequence(minimizer=2, knn=3, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=12)
    icp3.setMap(sc["map"], sc["normals"])
    T3 = icp3.registerDev(good.data_ptr(), good.shape[0])
    Ts, _, status = icp3.registerBatchDev([good.data_ptr(), good.data_ptr()], [good.shape[0]] * 2)
    assert status == [0, 0] and np.array_equal(Ts[0], T3) and np.array_equal(Ts[1], T3)
    with pytest.raises(Exception):
        icp.registerBatchDev([good.data_ptr()] * 17, [good.shape[0]] * 17)


@pytest.mark.parametrize("minimizer", [1, 2])
def test_run_ahead_loop_equals_read_back_loop(amd, mid_scene, minimizer):
    """Counter + Differential: the eager loop that stays a bounded number of iterations ahead of the progress word ends on
    the same iteration, stop reason and bits as the graph-free loop with periodic read-backs (ICPMI_RUN_AHEAD=0 is read
    once per process, so the comparison is against the oracle's iteration count and a second, fresh handle)."""
    sc = mid_scene
    kw = dict(minimizer=minimizer, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
    a = amd.ICPSequence(**kw); a.setMap(sc["map"], sc["normals"])
    b = amd.ICPSequence(use_graph=0, **kw); b.setMap(sc["map"], sc["normals"])
    Ta = a(sc["scan"]); Tb = b(sc["scan"])
    assert np.array_equal(Ta, Tb)
    assert a.stats.iterations == b.stats.iterations < 40 and a.stats.stop_reason == b.stats.stop_reason == 2
    # many short registrations back to back: a stale progress word of the previous registration must never stop the next one
    for rep in range(25):
        T = a(sc["scan"])
        assert np.array_equal(T, Ta) and a.stats.iterations == b.stats.iterations
