"""GPU parity of the resident map-update chain (icpmi_map_update_chain: Map::updateLocalPointCloud, Map.cpp:502-534,
for a whole mapper-module chain + post filters on the device copy of the map) against the same chain composed on the
host from the CPU oracle's single operators.  Bar: identical provenance (integer), identical points / scalar
descriptor (copies and the oracle's float formulas), normals equal to the oracle's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DYN = (0.9, 0.8, 0.99, 0.01, 0.01, 0.01, 200.0)   # examples/config.yaml:40-46 + sensorMaxRange


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


def host_chain(ob, map_pts, map_n, map_s, scan, scan_s, to_sensor, modules, post):
    """The reference's updateLocalPointCloud, one oracle operator after the other.  Returns (pts, normals, scalar, src)."""
    m0, n = map_pts.shape[0], scan.shape[0]
    pts, nrm, sc = map_pts.copy(), map_n.copy(), map_s.copy()
    src = np.arange(m0, dtype=np.int64)
    scan_src = m0 + np.arange(n, dtype=np.int64)
    zeros3 = np.zeros((n, 3), np.float32)

    def append(mask=None):
        nonlocal pts, nrm, sc, src
        sel = slice(None) if mask is None else mask
        pts = np.concatenate([pts, scan[sel]]); nrm = np.concatenate([nrm, zeros3[sel]])
        sc = np.concatenate([sc, scan_s[sel]]); src = np.concatenate([src, scan_src[sel]])

    created = m0 > 0
    for op in modules:
        name = op[0]
        if name == "point_distance":
            if not created or pts.shape[0] == 0:
                append()
            else:
                append(ob.point_distance_keep(pts, scan, op[1], nthreads=8))
        elif name == "dynamic_points":
            if not created:
                append()
            elif pts.shape[0]:
                prm = dict(zip(("threshold_dynamic", "alpha", "beta", "beam_half_angle", "epsilon_a", "epsilon_d", "sensor_max_range"), op[1:]))
                sc = ob.dynamic_points_update(to_sensor, scan, pts, nrm, sc, nthreads=8, **prm)
        elif name == "voxel":
            append()
            keep = ob.voxel_keep(pts, op[1], op[2])
            pts, nrm, sc, src = pts[keep], nrm[keep], sc[keep], src[keep]
        elif name == "octree":      # ("octree", maxSizeByNode, samplingMethod, maxPointByNode): the cloud comes out in leaf order
            append()
            if pts.shape[0]:
                order = ob.octree_sample(pts, op[1], op[3] if len(op) > 3 else 1, op[2] if len(op) > 2 else 0)
                pts, nrm, sc, src = pts[order], nrm[order], sc[order], src[order]
        created = True
    for op in post:
        if op[0] == "surface_normals":
            nrm = ob.surface_normals(pts, knn=op[1], nthreads=8)
        elif op[0] == "cut_scalar":
            keep = ~(sc > op[1]) if op[2] else ~(sc < op[1])
            pts, nrm, sc, src = pts[keep], nrm[keep], sc[keep], src[keep]
    return pts, nrm, sc, src


def make_clouds(amd, seed, m=30000, n=6000):
    sc = amd.synth.make_scene(m=m, n=n, seed_map=42 + 10 * seed)
    rng = np.random.default_rng(seed)
    scan = sc["map"][rng.permutation(m)[:n]].copy()
    scan[:, :3] += rng.normal(0, 0.05, (n, 3)).astype(np.float32)
    return sc["map"].copy(), scan


def normals_close(a, b):
    # PCA normals: the sign is not defined by the filter; compare up to sign, tolerance for the eigen solver
    dots = np.abs(np.einsum("ij,ij->i", a.astype(np.float64), b.astype(np.float64)))
    return float(np.mean(dots > 1 - 1e-4))


CHAINS = {
    "three_appends": ([("voxel", 0.05, 0), ("point_distance", 0.0), ("voxel", 0.02, 1)], []),   # the working map outgrows m + 2 n
    "point_distance": ([("point_distance", 0.25)], []),
    "point_distance+normals": ([("point_distance", 0.25)], [("surface_normals", 8)]),
    "shipped": ([("dynamic_points",) + DYN, ("voxel", 0.3, 1)], [("surface_normals", 10), ("cut_scalar", 0.65, 1)]),
    "voxel_first": ([("voxel", 0.4, 0), ("point_distance", 0.2)], [("surface_normals", 6)]),
    "pd_then_voxel": ([("point_distance", 0.1), ("voxel", 0.5, 0)], [("cut_scalar", 0.3, 0)]),
}


@pytest.mark.parametrize("chain", sorted(CHAINS))
def test_chain_equals_host_composition(amd, oracle, chain):
    modules, post = CHAINS[chain]
    base, scan = make_clouds(amd, 5)
    rng = np.random.default_rng(11)
    pose = amd.synth.make_T((0.02, -0.01, 0.4), (3.0, -2.0, 1.5)).astype(np.float32)
    to_sensor = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
    base_n = oracle.surface_normals(base, knn=8, nthreads=8)
    base_s = rng.uniform(0.0, 1.0, base.shape[0]).astype(np.float32)
    scan_s = np.full(scan.shape[0], 0.6, np.float32)

    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=5, use_differential=0)
    icp.setMap(base, base_n)
    icp.setMapScalar(base_s)
    src, m, head = icp.mapUpdateChain(scan, modules, post, scan_scalar=scan_s, to_sensor=to_sensor, with_prefix=True)
    pts, nrm, sc, ref_src = host_chain(oracle, base, base_n, base_s, scan, scan_s, to_sensor, modules, post)
    moved = np.nonzero(ref_src != np.arange(ref_src.shape[0]))[0]
    assert head == (moved[0] if moved.size else ref_src.shape[0])
    if chain.startswith("point_distance"):
        assert head == base.shape[0]            # append-only chain: only the tail is reported

    assert m == pts.shape[0]
    assert np.array_equal(src, ref_src)
    got, got_n = icp.getMap(with_normals=True)
    assert np.array_equal(got, pts)
    assert np.array_equal(icp.getMapScalar(), sc)
    if any(p[0] == "surface_normals" for p in post):
        assert normals_close(got_n, nrm) > 0.999
    else:
        assert np.array_equal(got_n, nrm)
    # the rebuilt index answers like a fresh handle on the same cloud
    fresh = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=5, use_differential=0)
    fresh.setMap(got, got_n)
    assert np.array_equal(icp(scan), fresh(scan))


@pytest.mark.parametrize("chain", ["shipped", "voxel_first", "point_distance+normals"])
def test_chain_creates_the_map_from_the_first_scan(amd, oracle, chain):
    """Map.cpp:508-516: the first module creates the map from the scan, the others update it with the same scan."""
    modules, post = CHAINS[chain]
    _, scan = make_clouds(amd, 6, n=8000)
    scan_s = np.full(scan.shape[0], 0.6, np.float32)
    to_sensor = np.eye(4, dtype=np.float32)
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, max_iterations=5)
    # "shipped": DynamicPoints as the creating module needs no normals; the Octree module then sees the scan twice
    src, m = icp.mapUpdateChain(scan, modules, post, scan_scalar=scan_s, to_sensor=to_sensor)
    empty = np.zeros((0, 4), np.float32)
    pts, nrm, sc, ref_src = host_chain(oracle, empty, np.zeros((0, 3), np.float32), np.zeros(0, np.float32), scan, scan_s, to_sensor, modules, post)
    assert m == pts.shape[0] and np.array_equal(src, ref_src)
    got, got_n = icp.getMap(with_normals=True)
    assert np.array_equal(got, pts) and np.array_equal(icp.getMapScalar(), sc)
    assert normals_close(got_n, nrm) > 0.999
    # a second scan through the same chain keeps agreeing (the resident arrays are now the device's own product)
    _, scan2 = make_clouds(amd, 7, n=8000)
    scan2[:, :3] += np.float32(0.03)
    src2, m2 = icp.mapUpdateChain(scan2, modules, post, scan_scalar=scan_s, to_sensor=to_sensor)
    pts2, nrm2, sc2, ref_src2 = host_chain(oracle, got, got_n, sc.copy(), scan2, scan_s, to_sensor, modules, post)
    assert m2 == pts2.shape[0] and np.array_equal(src2, ref_src2)
    assert np.array_equal(icp.getMap(), pts2) and np.array_equal(icp.getMapScalar(), sc2)


def test_chain_staged_scan_equals_host_scan(amd):
    """icpmi_map_update_chain_staged (the scan staged by icpmi_register_prior, moved by the correction on the device)
    against icpmi_map_update_chain on the same cloud transformed through icpmi_transform."""
    base, scan = make_clouds(amd, 8)
    modules, post = CHAINS["point_distance+normals"]
    kw = dict(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=8, use_differential=0)
    prior = amd.synth.make_T((0.004, -0.003, 0.006), (0.05, -0.04, 0.02)).astype(np.float32)
    a = amd.ICPSequence(**kw); a.setMap(base)
    b = amd.ICPSequence(**kw); b.setMap(base)
    corr = a.registerWithPrior(scan, prior)
    src_a, m_a = a.mapUpdateChain(None, modules, post, staged_correction=corr)
    in_map = b.transform(prior, scan)
    corr_b = b(in_map)
    assert np.array_equal(corr, corr_b)
    moved = b.transform(corr_b, in_map)
    src_b, m_b = b.mapUpdateChain(moved, modules, post)
    assert m_a == m_b and np.array_equal(src_a, src_b)
    ga, na = a.getMap(with_normals=True); gb, nb = b.getMap(with_normals=True)
    assert np.array_equal(ga, gb) and np.array_equal(na, nb)


def test_chain_errors(amd):
    base, scan = make_clouds(amd, 9, m=5000, n=1000)
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, max_iterations=5)
    icp.setMap(base)
    with pytest.raises(amd.InvalidParameter):   # scalar-tracking chain without the scalar on the input
        icp.mapUpdateChain(scan, [("voxel", 0.3, 0)], [("cut_scalar", 0.5, 1)])
    with pytest.raises(amd.InvalidParameter):   # ... or on the map
        icp.mapUpdateChain(scan, [("voxel", 0.3, 0)], [("cut_scalar", 0.5, 1)], scan_scalar=np.zeros(scan.shape[0], np.float32))
    icp.setMapScalar(np.zeros(base.shape[0], np.float32))
    with pytest.raises(amd.InvalidField):     # DynamicPoints on a map without normals (DynamicPointsMapperModule.cpp:38-41)
        icp.mapUpdateChain(scan, [("dynamic_points",) + DYN], [], scan_scalar=np.zeros(scan.shape[0], np.float32), to_sensor=np.eye(4))
    with pytest.raises(amd.InvalidParameter):
        icp.mapUpdateChain(scan, [], [("surface_normals", 5)])
    # the failures above left the map untouched
    assert np.array_equal(icp.getMap(), base)
    src, m = icp.mapUpdateChain(scan, [("voxel", 0.3, 0)], [])
    assert m == src.shape[0] and (np.diff(src) > 0).all()


def test_chain_with_an_empty_scan_still_runs_the_program(amd, oracle):
    """n = 0: the modules have nothing to add, the decimation and the post filters still run over the map."""
    base, _ = make_clouds(amd, 10, m=8000, n=10)
    sc0 = np.linspace(0.0, 1.0, base.shape[0]).astype(np.float32)
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, max_iterations=5)
    icp.setMap(base); icp.setMapScalar(sc0)
    empty = np.zeros((0, 4), np.float32)
    modules, post = [("point_distance", 0.1), ("voxel", 0.5, 0)], [("cut_scalar", 0.8, 1)]
    src, m = icp.mapUpdateChain(empty, modules, post, scan_scalar=np.zeros(0, np.float32))
    pts, _, sc, ref_src = host_chain(oracle, base, np.zeros((base.shape[0], 3), np.float32), sc0, empty, np.zeros(0, np.float32),
                                     np.eye(4, dtype=np.float32), modules, post)
    assert m == pts.shape[0] and np.array_equal(src, ref_src)
    assert np.array_equal(icp.getMap(), pts) and np.array_equal(icp.getMapScalar(), sc)
    # nothing at all: no map, no scan
    fresh = amd.ICPSequence(minimizer=1, max_dist=2.0, max_iterations=5)
    src0, m0 = fresh.mapUpdateChain(empty, modules, [])
    assert m0 == 0 and src0.shape == (0,)


def test_chain_that_removes_every_point_leaves_an_empty_map(amd):
    """CutAtDescriptorThreshold cutting the whole map: the reference goes on with an empty local cloud (`icp.setMap` ignores an
    empty cloud and keeps its previous map, Map.cpp:528) and the next scan creates the map anew (Map.cpp:505-515)."""
    base, scan = make_clouds(amd, 12, m=6000, n=1500)
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, max_iterations=5)
    icp.setMap(base); icp.setMapScalar(np.full(base.shape[0], 0.9, np.float32))
    src, m = icp.mapUpdateChain(scan, [("point_distance", 0.2)], [("cut_scalar", 0.5, 1)], scan_scalar=np.full(scan.shape[0], 0.9, np.float32))
    assert m == 0 and src.shape == (0,) and icp.getMap().shape[0] == 0
    assert icp.hasMap()                                   # the registration index is the previous map
    T = icp(scan)                                         # and still registers
    assert np.isfinite(T).all()
    # the next scan creates the map: first-scan semantics of the chain
    src, m = icp.mapUpdateChain(scan, [("point_distance", 0.2)], [("cut_scalar", 0.95, 1)], scan_scalar=np.full(scan.shape[0], 0.9, np.float32))
    assert m == scan.shape[0] and np.array_equal(icp.getMap(), scan) and np.array_equal(src, np.arange(scan.shape[0]))
