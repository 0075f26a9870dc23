"""SurfaceNormalDataPointsFilter over the resident map of an append-only update (Map.cpp:524 applies the `post:` filters to the WHOLE local map on
every update; examples/config.yaml:25-27 `SurfaceNormalDataPointsFilter: knn: 10`) served incrementally (csrc/ops.hip: surface_normals_dev,
csrc/selfgrid.hip: the subset search): after an append only the appended points and the old points an appended point can have entered the
neighbourhood of are searched and solved again.  The result must be the whole-map pass bit for bit -- checked against the filter run from
scratch on the downloaded map (icpmi_surface_normals on another handle), update after update, and the counters must say the subset path ran."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


def cloud(xyz):
    c = np.ones((xyz.shape[0], 4), dtype=np.float32); c[:, :3] = xyz.astype(np.float32); return c


def check_against_a_fresh_pass(amd, icp, knn):
    pts, nrm = icp.getMap(with_normals=True)
    ref = amd.ICPSequence(minimizer=1).surfaceNormals(pts, knn=knn)
    assert np.array_equal(nrm, ref), f"{int((nrm != ref).any(axis=1).sum())} of {pts.shape[0]} normals differ from a pass over the whole map"
    return pts.shape[0]


@pytest.mark.parametrize("knn", [10, 5])
def test_appends_recompute_only_what_changed_dense_scene(amd, mid_scene, knn):
    sc = mid_scene
    icp = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=10)
    assert icp.setMap(sc["map"][::2].copy(), sc["normals"][::2].copy())           # caller's normals: the first pass replaces them all
    rng = np.random.default_rng(1)
    searched = []
    for step in range(4):
        scan = sc["scan"].copy()
        scan[:, :3] += np.float32(0.6 * step) * np.array([1.0, 0.3, 0.0], dtype=np.float32) + rng.normal(0, 0.02, (scan.shape[0], 3)).astype(np.float32)
        app, m = icp.mapUpdatePointDistance(scan, 0.15, normals_knn=knn)
        assert app > 100
        assert check_against_a_fresh_pass(amd, icp, knn) == m
        c = icp.debugCounters()
        searched.append((int(c[20]), int(c[21]), int(c[22]), m))
    # the registration index carries the same normals (changed ones patched into the sorted copy, the rest moved along with their points): a
    # point-to-plane registration against it and against a fresh handle built from the downloaded map + normals are the same bits
    pts, nrm = icp.getMap(with_normals=True)
    fresh = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=10)
    assert fresh.setMap(pts, nrm)
    assert np.array_equal(icp(sc["scan"]), fresh(sc["scan"]))
    assert [s[1] for s in searched] == [1, 1, 1, 1]          # one pass over the whole map (the first), ...
    assert [s[0] for s in searched] == [0, 1, 2, 3]          # ... then the subset path
    assert all(s[2] < s[3] for s in searched[1:]), searched      # and it searched a part of the map only (this scan covers most of the room)


def test_sparse_periphery_and_tiny_maps(amd):
    """a heavy-tailed cloud (k-th neighbours metres away: the coarse levels of the test) growing by points far out and close in;
    then a map that starts with fewer than k points"""
    rng = np.random.default_rng(4)
    def heavy(n, scale):
        d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        r = scale * rng.pareto(1.5, size=(n, 1))
        return cloud(d * r * np.array([1.0, 1.0, 0.15]))
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, max_iterations=5)
    assert icp.setMap(heavy(30000, 2.0))
    for step, (n, scale) in enumerate([(2000, 2.0), (300, 40.0), (5000, 0.5), (50, 400.0)]):
        app, m = icp.mapUpdatePointDistance(heavy(n, scale), 0.05, normals_knn=10)
        assert app > 0
        check_against_a_fresh_pass(amd, icp, 10)
    c = icp.debugCounters()
    assert int(c[20]) == 3 and int(c[21]) == 1
    tiny = amd.ICPSequence(minimizer=1, max_dist=2.0, max_iterations=5)
    assert tiny.setMap(cloud(rng.uniform(-1, 1, (4, 3))))
    for n in (3, 2, 40, 500):
        tiny.mapUpdatePointDistance(cloud(rng.uniform(-1, 1, (n, 3))), 0.0, normals_knn=10)
        check_against_a_fresh_pass(amd, tiny, 10)


def test_epoch_with_normals_and_a_rewritten_map_start_over(amd, mid_scene, monkeypatch):
    """the map-growth epoch of the scan-sharded mapper (icpmi_staged_merge_allgather with normals_knn) takes the same path; a setMap in
    between (the resident copy replaced) sends the next pass over the whole map again"""
    sc = mid_scene
    monkeypatch.setenv("ICPMI_COMM_LOOPBACK", "3")
    monkeypatch.setenv("ICPMI_COMM_LOOPBACK_SHIFT", "0.5")
    icp = amd.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20, use_differential=1)
    assert icp.setMap(sc["map"][::2].copy(), sc["normals"][::2].copy())
    icp.commInit(icp.commUniqueId(), 1, 0)
    eye = np.eye(4, dtype=np.float32)
    for step in range(3):
        scan = sc["scan"].copy(); scan[:, 0] += np.float32(0.4 * step)
        corr = icp.registerWithPrior(scan, eye)
        mine, app, m = icp.stagedMergeAllGather(corr, 0.3, normals_knn=10)
        assert check_against_a_fresh_pass(amd, icp, 10) == m
    c = icp.debugCounters()
    assert int(c[20]) == 2 and int(c[21]) == 1
    pts, nrm = icp.getMap(with_normals=True)
    assert icp.setMap(pts[::3].copy(), nrm[::3].copy())
    corr = icp.registerWithPrior(sc["scan"], eye)
    icp.stagedMergeAllGather(corr, 0.3, normals_knn=10)
    check_against_a_fresh_pass(amd, icp, 10)
    c = icp.debugCounters()
    assert int(c[20]) == 2 and int(c[21]) == 2


def test_planar_map_appends(amd):
    """the mapper's is3D == false (z == 0 everywhere, 2 x 2 normals): the same subset path, the same bits as the whole pass"""
    rng = np.random.default_rng(9)
    def ring(n, r, jitter):
        a = rng.uniform(0, 2 * np.pi, n)
        xy = np.c_[r * np.cos(a), r * np.sin(a)] + rng.normal(0, jitter, (n, 2))
        return cloud(np.c_[xy, np.zeros(n)])
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, max_iterations=5, is_2d=1)
    assert icp.setMap(ring(20000, 10.0, 0.02))
    for r in (10.5, 6.0, 14.0):
        app, m = icp.mapUpdatePointDistance(ring(3000, r, 0.05), 0.02, normals_knn=8)
        assert app > 0
        pts, nrm = icp.getMap(with_normals=True)
        ref = amd.ICPSequence(minimizer=1, is_2d=1).surfaceNormals(pts, knn=8)
        assert np.array_equal(nrm, ref) and (nrm[:, 2] == 0).all()
    c = icp.debugCounters()
    assert int(c[20]) == 2 and int(c[21]) == 1
