"""OctreeGridDataPointsFilter on the device (icpmi_octree_sample, ICPMI_MOP_OCTREE) against the oracle's recursive restatement
(oracle/icp_oracle.c: orc_octree_sample).  Bar: the same points, in the same (leaf-visiting) order -- integer work."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


def _clouds(amd):
    rng = np.random.default_rng(11)
    out = {}
    sc = amd.synth.make_scene(m=120_000, n=10)
    out["scene"] = sc["map"]
    u = np.ones((50_000, 4), np.float32); u[:, :3] = rng.uniform(-7, 9, (50_000, 3)) * [1.0, 0.6, 0.05]
    out["slab"] = u
    d = np.ones((30_000, 4), np.float32); d[:, :3] = rng.normal(0, 0.02, (30_000, 3)); d[::7, :3] += 3.0   # two dense blobs: deep, crowded leaves
    out["blobs"] = d
    dup = np.ones((2_000, 4), np.float32); dup[:, :3] = rng.integers(0, 4, (2_000, 3)).astype(np.float32) * 0.5    # many exact duplicates
    out["duplicates"] = dup
    out["one"] = np.array([[1.0, 2.0, 3.0, 1.0]], np.float32)
    out["line"] = np.ones((500, 4), np.float32); out["line"][:, 0] = np.linspace(0, 10, 500, dtype=np.float32); out["line"][:, 1:3] = 0
    lattice = np.ones((4096, 4), np.float32)                                                                         # points ON node boundaries
    g = np.stack(np.meshgrid(*[np.arange(16, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)
    lattice[:, :3] = g * 0.25
    out["lattice"] = lattice
    return out


@pytest.mark.parametrize("method", [0, 1])
@pytest.mark.parametrize("max_pts", [1, 2, 5, 64])
@pytest.mark.parametrize("max_size", [0.0, 0.15, 0.5, 3.0])
def test_octree_sample_equals_oracle_recursion(amd, oracle, method, max_pts, max_size):
    icp = amd.ICPSequence()
    for name, cloud in _clouds(amd).items():
        got, leaf = icp.octreeSample(cloud, max_size, max_pts, method, with_leaves=True)
        want = oracle.octree_sample(cloud, max_size, max_pts, method)
        assert got.shape == want.shape, (name, got.shape, want.shape)
        assert np.array_equal(got, want), (name, int(np.argmax(got != want)))
        # every kept point represents its own leaf, leaves are numbered in output order
        assert np.array_equal(leaf[got], np.arange(got.shape[0]))
        assert leaf.min() == 0 and leaf.max() == got.shape[0] - 1
        if max_pts == 1 and max_size > 0 and cloud.shape[0] > 1:
            # one point per cube-aligned cell of the depth the edge criterion stops at: no two kept points share a cell
            lo, hi = cloud[:, :3].min(0), cloud[:, :3].max(0)
            c = (lo + hi) / 2; r = float((hi - c).max())
            if r > 0:
                D = 0
                while D < 21 and not (r / 2 ** D * 2 <= max_size):
                    D += 1
                cell = 2 * r / 2 ** D
                assert cell <= max_size * (1 + 1e-6) or D == 21


def test_octree_depth_speculation_recovers_from_a_wrong_guess(amd, oracle):
    """r5: the sort passes are enqueued with the tree depth of the handle's PREVIOUS call, the real depth is checked when the leaf count
    arrives.  A shallow cloud first, then a deep one on the same handle: the second call must notice, sort again and still agree with the
    oracle; a deep-then-shallow pair must not re-sort (sorting on more bits than the paths have is still their order)."""
    clouds = _clouds(amd)
    icp = amd.ICPSequence()
    before = icp.debugCounters()[16]
    for name in ("line", "blobs", "scene", "one", "scene", "slab"):
        got = icp.octreeSample(clouds[name], 0.05, 1, 0)
        assert np.array_equal(got, oracle.octree_sample(clouds[name], 0.05, 1, 0)), name
    redone = icp.debugCounters()[16] - before
    assert 1 <= redone <= 3, redone          # line (first call: waits) -> blobs deeper? -> scene ... at least the one -> scene step re-sorts


def test_octree_chain_equals_oracle_composition(amd, oracle):
    """OctreeMapperModule on the resident map (concatenate, then the octree, map left in leaf order) inside the shipped chain"""
    import oracle_mapper as om
    sc = amd.synth.make_scene(m=40_000, n=8_000)
    rng = np.random.default_rng(3)
    scan = sc["map"][rng.permutation(40_000)[:8_000]].copy(); scan[:, :3] += rng.normal(0, 0.04, (8_000, 3)).astype(np.float32)
    icp = amd.ICPSequence(minimizer=2, max_dist=2.0)
    n0 = oracle.surface_normals(sc["map"], knn=10, nthreads=8)
    icp.setMap(sc["map"], n0)
    icp.setMapScalar(np.full(sc["map"].shape[0], 0.6, np.float32))
    pose = amd.synth.make_T((0.01, -0.02, 0.03), (1.0, -2.0, 0.5))
    to_sensor = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
    dyn7 = (0.9, 0.8, 0.99, 0.01, 0.01, 0.01, 200.0)
    src, m = icp.mapUpdateChain(scan, [("dynamic_points",) + dyn7, ("octree", 0.3, 0, 2)], [("surface_normals", 10), ("cut_scalar", 0.65, 1)],
                                scan_scalar=np.full(scan.shape[0], 0.6, np.float32), to_sensor=to_sensor, from_sensor=pose)
    got = icp.getMap()
    ref = om.OracleMapper(dict(minimizer=2, max_dist=2.0), [("dynamic_points", dict(threshold_dynamic=0.9, alpha=0.8, beta=0.99, beam_half_angle=0.01,
                          epsilon_a=0.01, epsilon_d=0.01)), ("octree", 0.3, 0, 2)], post=[("surface_normals", 10), ("cut", "probabilityDynamic", 1, 0.65)], nthreads=8)
    ref.map = {"xyz1": sc["map"].copy(), "normals": n0.copy(), "probabilityDynamic": np.full((sc["map"].shape[0], 1), 0.6, np.float32)}
    ref.update_local_point_cloud({"xyz1": scan, "probabilityDynamic": np.full((scan.shape[0], 1), 0.6, np.float32)}, pose)
    assert got.shape == ref.map["xyz1"].shape
    assert np.array_equal(got, ref.map["xyz1"])
    assert np.array_equal(icp.getMapScalar(), ref.map["probabilityDynamic"][:, 0])
    both = np.concatenate([sc["map"], scan])
    # provenance: new map point j is point src[j] of [old map ; scan] -- before the sensor-frame round trip moved it by rounding
    assert np.abs(both[src, :3] - got[:, :3]).max() < 5e-5


@pytest.mark.parametrize("n,knn,ratio,seed,max_box", [(6000, 7, 0.5, 1, np.inf), (3001, 12, 0.9, 5, np.inf), (50, 7, 1.0, 2, np.inf), (5, 7, 0.5, 1, np.inf),
                                                      (40000, 7, 0.5, 1, np.inf), (20000, 9, 0.7, 3, 1.5), (9, 3, 0.5, 1, np.inf)])
def test_sampling_surface_normal_on_device_equals_oracle(amd, oracle, n, knn, ratio, seed, max_box):
    """icpmi_sampling_surface_normal (csrc/ssn.hip: one radix sort per tree level, PCA per box, minstd by skip-ahead) against the oracle's
    recursion (orc_sampling_surface_normal): the kept indices in box order must be IDENTICAL, the normals agree to rounding.
    SamplingSurfaceNormalDataPointsFilter is the reference filter of PM::ICPSequence::setDefault (Mapper.cpp:74-78)."""
    rng = np.random.default_rng(n + knn)
    c = np.ones((n, 4), np.float32)
    c[:, :3] = (rng.uniform(-1, 1, (n, 3)) * [20.0, 12.0, 0.4]).astype(np.float32)
    c[: n // 3, 2] = 0.0                                    # a flat patch: rank-2 boxes, and many equal z coordinates (ties by index)
    if n >= 3000:
        c[100:160] = c[100]                                 # duplicates: rank-0 boxes are dropped, ties everywhere
        c[200:400, 0] = np.float32(-0.0); c[400:600, 0] = np.float32(0.0)   # signed zeros compare equal
    icp = amd.ICPSequence()
    order, nrm = icp.samplingSurfaceNormal(c, ratio=ratio, knn=knn, max_box_dim=max_box, seed=seed)
    o_order, o_nrm = oracle.sampling_surface_normal(c, ratio, knn, seed=seed, max_box_dim=max_box)
    assert np.array_equal(order, o_order), (n, knn, order.shape, o_order.shape)
    if order.shape[0]:
        dots = np.abs(np.einsum("ij,ij->i", nrm, o_nrm))
        assert (dots > 1 - 1e-5).all()
        assert np.allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-5)
