"""DataPointsFilters of the C++ host shell that the ICP chains use (PM::ICPSequence::setDefault: RandomSampling on the reading,
SamplingSurfaceNormal on the reference; MaxDensity) against the oracle's restatements -- same std::minstd_rand streams, so the
kept sets are identical.  CPU part: the host-only filters.  GPU part: densities from the device, the default chain end to end."""
import os

import numpy as np
import pytest


def _cloud(n, seed):
    rng = np.random.default_rng(seed)
    c = np.ones((n, 4), np.float32)
    c[:, :3] = (rng.uniform(-1, 1, (n, 3)) * [20.0, 12.0, 0.4]).astype(np.float32)
    c[: n // 3, 2] = 0.0                      # a flat patch: rank-2 boxes
    return c


@pytest.fixture(scope="module")
def host():
    import subprocess
    import host_bindings as hb
    if not os.path.exists(hb.LIB):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "norlab_icp_mapper_amd", "csrc"), "-j8"])
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "norlab_icp_mapper_amd", "host"), "-j8"])
    return hb


def test_observation_direction_orient_normals_and_the_old_distance_names(host):
    """ObservationDirection / OrientNormals (they sit beside SurfaceNormal in most mapper configurations) and the MinDist / MaxDist
    names of DistanceLimit, against numpy"""
    c = _cloud(5000, 3)
    rng = np.random.default_rng(4)
    nrm = rng.normal(size=(5000, 3)).astype(np.float32); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    sensor = np.float32([1.5, -2.0, 0.75])
    out, _, od = host.filter_chain("[{ObservationDirectionDataPointsFilter: {x: 1.5, y: -2.0, z: 0.75}}]", c,
                                   desc_name="observationDirections", desc=np.zeros((5000, 3), np.float32))
    assert np.array_equal(out, c) and np.array_equal(od, sensor[None, :] - c[:, :3])
    for toward in (1, 0):
        chain = "[{ObservationDirectionDataPointsFilter: {x: 1.5, y: -2.0, z: 0.75}}, {OrientNormalsDataPointsFilter: {towardCenter: %d}}]" % toward
        _, got, _ = host.filter_chain(chain, c, desc_name="normals", desc=nrm)
        dots = (nrm * (sensor[None, :] - c[:, :3])).sum(1, dtype=np.float32)
        flip = dots < 0 if toward else dots > 0
        want = np.where(flip[:, None], -nrm, nrm)
        assert np.array_equal(got, want) and 0.3 < flip.mean() < 0.7
    with pytest.raises(RuntimeError, match="observation directions"):
        host.filter_chain("[OrientNormalsDataPointsFilter]", c, desc_name="normals", desc=nrm)
    r = np.sqrt((c[:, :3].astype(np.float32) ** 2).sum(1, dtype=np.float32))
    far, _, _ = host.filter_chain("[{MinDistDataPointsFilter: {minDist: 8.0}}]", c)
    near, _, _ = host.filter_chain("[{MaxDistDataPointsFilter: {maxDist: 8.0}}]", c)
    assert far.shape[0] + near.shape[0] <= 5000 and far.shape[0] > 0 and near.shape[0] > 0
    assert np.array_equal(far, c[r > 8.0]) and np.array_equal(near, c[r < 8.0])
    ax, _, _ = host.filter_chain("[{MaxDistDataPointsFilter: {dim: 1, maxDist: 3.0}}]", c)
    assert np.array_equal(ax, c[np.abs(c[:, 1]) < 3.0])


@pytest.mark.parametrize("method", [0, 1])
@pytest.mark.parametrize("prob,seed", [(0.75, 1), (0.2, 12345), (1.0, 3), (0.0, 9)])
def test_random_sampling_equals_oracle(host, oracle, prob, seed, method):
    c = _cloud(5000, 1)
    out, _, _ = host.filter_chain(f"- RandomSamplingDataPointsFilter: {{prob: {prob}, randomSamplingMethod: {method}, seed: {seed}}}", c)
    keep = oracle.random_sampling_keep(c.shape[0], prob, method, seed)
    assert np.array_equal(out, c[keep])
    assert out.shape[0] <= int(np.float32(c.shape[0]) * np.float32(prob)) + 1


def test_sampling_surface_normal_equals_oracle(host, oracle):
    for n, knn, ratio, seed in ((6000, 7, 0.5, 1), (3001, 12, 0.9, 5), (50, 7, 1.0, 2), (5, 7, 0.5, 1)):
        c = _cloud(n, n)
        out, nrm, _ = host.filter_chain(f"- SamplingSurfaceNormalDataPointsFilter: {{ratio: {ratio}, knn: {knn}, seed: {seed}}}", c)
        order, onrm = oracle.sampling_surface_normal(c, ratio, knn, seed=seed)
        assert np.array_equal(out, c[order]), (n, knn)
        if order.shape[0]:
            assert nrm is not None
            dots = np.abs(np.einsum("ij,ij->i", nrm, onrm))
            assert (dots > 1 - 1e-5).all()
            assert np.allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-5)


def test_max_density_equals_oracle(host, oracle):
    c = _cloud(4000, 7)
    rng = np.random.default_rng(2)
    dens = rng.uniform(0.0, 40.0, c.shape[0]).astype(np.float32)
    out, _, d = host.filter_chain("- MaxDensityDataPointsFilter: {maxDensity: 10, seed: 4}", c, desc_name="densities", desc=dens)
    keep = oracle.max_density_keep(dens, 10.0, 4)
    assert np.array_equal(out, c[keep]) and np.array_equal(d, dens[keep])
    with pytest.raises(RuntimeError):
        host.filter_chain("- MaxDensityDataPointsFilter: {maxDensity: 10}", c)          # no `densities`: InvalidField


@pytest.mark.gpu
def test_densities_and_max_density_chain_on_gpu(host, oracle):
    import norlab_icp_mapper_amd as amd
    sc = amd.synth.make_scene(m=60_000, n=10)
    icp = amd.ICPSequence()
    n_gpu, d_gpu = icp.surfaceNormals(sc["map"], knn=10, with_densities=True)
    n_cpu, d_cpu = oracle.surface_normals(sc["map"], knn=10, nthreads=8, with_densities=True)
    assert np.array_equal(d_gpu, d_cpu)                                                   # same neighbours, same double formula
    out, nrm, _ = host.filter_chain("- SurfaceNormalDataPointsFilter: {knn: 10, keepDensities: 1}\n- MaxDensityDataPointsFilter: {maxDensity: 30, seed: 2}",
                                    sc["map"], handle=icp._h.value if hasattr(icp._h, "value") else icp._h)
    keep = oracle.max_density_keep(d_cpu, 30.0, 2)
    assert np.array_equal(out, sc["map"][keep]) and 0 < keep.sum() < keep.shape[0]


@pytest.mark.gpu
def test_default_chain_reading_and_reference_filters(host, oracle, mid_scene, tmp_path):
    """A configuration without an `icp:` key (Mapper.cpp:74-78): PM::ICPSequence::setDefault -- RandomSampling(0.75) on the reading,
    SamplingSurfaceNormal on the map.  The C++ host shell registers what the oracle registers when it is given the oracle-filtered
    clouds (the host's filters draw from std::random_device for seed -1, so the reading filter is pinned with a seed here)."""
    import subprocess
    import norlab_icp_mapper_amd as amd
    sc = mid_scene
    # the reference filter of the default chain on the map: host == oracle (kept set and normals)
    icp = amd.ICPSequence()
    href = icp._h.value if hasattr(icp._h, "value") else icp._h
    m_out, m_nrm, _ = host.filter_chain("- SamplingSurfaceNormalDataPointsFilter", sc["map"], handle=href)
    order, onrm = oracle.sampling_surface_normal(sc["map"], 0.5, 7, seed=1)
    assert np.array_equal(m_out, sc["map"][order])
    r_out, _, _ = host.filter_chain("- RandomSamplingDataPointsFilter: {prob: 0.75, seed: 11}", sc["scan"])
    keep = oracle.random_sampling_keep(sc["scan"].shape[0], 0.75, 0, 11)
    assert np.array_equal(r_out, sc["scan"][keep])
    # the default chain on those clouds: GPU vs oracle
    kw = dict(minimizer=2, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)        # knn 1, maxDist inf
    g = amd.ICPSequence(**kw); g.setMap(m_out, m_nrm)
    T = g(r_out)
    o = oracle.OracleICP(oracle.make_config(nthreads=8, **kw)); o.setMap(m_out, onrm)
    err, T_ref = o(r_out)
    assert err == 0 and g.stats.iterations == o.stats.iterations
    dt, dr = amd.synth.pose_error(T, T_ref)
    assert dt <= 1e-4 and dr <= 1e-4, (dt, dr)
    gt, gr = amd.synth.pose_error(T, sc["T_gt"])
    assert gt < 1e-2 and gr < 1e-3, (gt, gr)


@pytest.mark.gpu
def test_surface_normal_keep_matched_ids_and_mean_dist_descriptors(host, oracle):
    """SurfaceNormalDataPointsFilter{keepMatchedIds: 1, keepMeanDist: 1} of the host shell: descriptors `matchedIds` (knn rows, ids as
    the cloud's scalar type) and `meanDist` (1 row), equal to the oracle's."""
    import norlab_icp_mapper_amd as amd
    sc = amd.synth.make_scene(m=20_000, n=10)
    icp = amd.ICPSequence()
    h = icp._h.value if hasattr(icp._h, "value") else icp._h
    k = 6
    _, ids, md = oracle.surface_normals_extras(sc["map"], knn=k, nthreads=8)
    n = sc["map"].shape[0]
    yaml = f"- SurfaceNormalDataPointsFilter: {{knn: {k}, keepMatchedIds: 1, keepMeanDist: 1}}"
    # (the test hook reads back the descriptor it was given by name: the filter replaces it)
    out, nrm, got_ids = host.filter_chain(yaml, sc["map"], handle=h, desc_name="matchedIds", desc=np.zeros((n, k), np.float32))
    assert nrm is not None and np.array_equal(out, sc["map"])
    assert np.array_equal(got_ids, ids.astype(np.float32))
    _, _, got_md = host.filter_chain(yaml, sc["map"], handle=h, desc_name="meanDist", desc=np.zeros(n, np.float32))
    np.testing.assert_allclose(got_md.reshape(-1), md, rtol=2e-4, atol=2e-5)


def _ssn_boxes_from_method0(oracle, c, knn):
    """samplingMethod 1 restated on top of samplingMethod 0 at ratio 1 (every point of a surviving box is kept, in box order, with its box's
    normal): a box is a maximal run of kept points with the same normal whose length fits the partition; rather than guess the cuts, take
    the member runs from the oracle's method-1 output and check them against this stream -- then means and firsts follow in numpy."""
    order0, nrm0 = oracle.sampling_surface_normal(c, 1.0, knn, seed=1)
    first, nrm, mean, ms, mc, mem = oracle.sampling_surface_normal_boxes(c, knn)
    assert np.array_equal(mem, order0)                                   # same boxes, same depth-first order, index order inside
    assert ms[0] == 0 and np.array_equal(ms[1:], np.cumsum(mc)[:-1]) and ms[-1] + mc[-1] == mem.shape[0]
    assert (mc >= 1).all() and (mc <= knn).all()
    for j in range(first.shape[0]):
        run = mem[ms[j]:ms[j] + mc[j]]
        assert (np.diff(run) > 0).all() and first[j] == run[0]
        assert np.array_equal(nrm0[ms[j]:ms[j] + mc[j]], np.repeat(nrm[j][None], mc[j], 0))
        acc = np.zeros(3, np.float64)
        for i in run:                                                    # the same left-to-right double sum
            acc += c[i, :3].astype(np.float64)
        assert np.array_equal((acc / float(mc[j])).astype(np.float32), mean[j])
    return first, nrm, mean, ms, mc, mem


def test_sampling_surface_normal_method1_oracle_and_host(host, oracle):
    """samplingMethod 1 (one point per box, at the mean; descriptors averaged): the oracle against its own method 0 + numpy, and the host
    shell WITHOUT a GPU context (the recursion a CPU-only unit test runs) against the oracle -- positions, normals and the averaged row."""
    for n, knn in ((3000, 7), (1201, 12), (40, 7)):
        c = _cloud(n, 100 + n)
        first, nrm, mean, ms, mc, mem = _ssn_boxes_from_method0(oracle, c, knn)
        rng = np.random.default_rng(n)
        d = rng.uniform(0, 1, n).astype(np.float32)
        out, hn, dout = host.filter_chain(f"- SamplingSurfaceNormalDataPointsFilter: {{samplingMethod: 1, knn: {knn}}}", c, desc_name="intensity", desc=d)
        assert out.shape[0] == first.shape[0]
        assert np.array_equal(out[:, :3], mean) and np.array_equal(out[:, 3], c[first, 3])
        dots = np.abs(np.einsum("ij,ij->i", hn, nrm))
        assert (dots > 1 - 1e-5).all()
        want = np.array([np.float32(sum(float(d[i]) for i in mem[ms[j]:ms[j] + mc[j]]) / float(mc[j])) for j in range(first.shape[0])], np.float32)
        assert np.array_equal(dout, want)


@pytest.mark.gpu
def test_sampling_surface_normal_method1_on_the_device(host, oracle):
    """icpmi_sampling_surface_normal_ex(method 1) against the oracle -- boxes, firsts, member lists and means bit for bit, normals to the
    rounding of the two Jacobi sweeps -- and the host filter WITH a GPU context against the host filter without one."""
    import norlab_icp_mapper_amd as amd
    icp = amd.ICPSequence()
    href = icp._h.value if hasattr(icp._h, "value") else icp._h
    sc = amd.synth.make_scene(m=150_000, n=10)
    for c, knn, box in ((sc["map"], 7, np.inf), (_cloud(5003, 9), 12, np.inf), (_cloud(800, 3), 7, 0.9), (_cloud(5, 1), 7, np.inf)):
        got = icp.samplingSurfaceNormalBoxes(c, knn=knn, max_box_dim=box)
        want = oracle.sampling_surface_normal_boxes(c, knn=knn, max_box_dim=box)
        assert got[0].shape == want[0].shape
        for k in (0, 3, 4, 5):
            assert np.array_equal(got[k], want[k]), k
        assert np.array_equal(got[2].view(np.uint32), want[2].view(np.uint32))
        if want[0].shape[0]:
            assert (np.abs(np.einsum("ij,ij->i", got[1], want[1])) > 1 - 1e-5).all()
    c = sc["map"][:40_000]
    d = np.random.default_rng(4).uniform(0, 1, c.shape[0]).astype(np.float32)
    y = "- SamplingSurfaceNormalDataPointsFilter: {samplingMethod: 1, knn: 9}"
    a = host.filter_chain(y, c, handle=href, desc_name="intensity", desc=d)
    b = host.filter_chain(y, c, desc_name="intensity", desc=d)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    assert (np.abs(np.einsum("ij,ij->i", a[1], b[1])) > 1 - 1e-5).all()
