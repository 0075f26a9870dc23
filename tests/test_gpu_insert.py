"""Incremental index insert (r4, csrc/map_build.hip: map_insert): a map that grows by an append keeps its grid, the delta is merged into
the cell-sorted arrays of every pyramid level and every point is recentred on the NEW centroid -- the index a full build would give,
up to the order inside a cell.  Checked against a fresh handle that indexes the concatenated cloud from scratch: kNN ids / d^2 bit for
bit (k = 1 and 6, radius and unbounded), registrations bit for bit, after one and after several appends, with and without normals,
and for a delta that leaves the bounding box (the insert must step aside)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


def builds(icp):
    c = icp.debugCounters()
    return {"ins": c[18] & 0xffffffff, "full": c[19] & 0xffffffff, "raw_ins": c[18] >> 32, "raw_full": c[19] >> 32, "raw_view": c[17]}


def queries(rng, cloud, n):
    q = cloud[rng.integers(0, cloud.shape[0], n)].copy()
    q[:, :3] += rng.normal(0, 0.2, (n, 3)).astype(np.float32)
    return q


@pytest.mark.parametrize("with_normals", [False, True])
def test_appends_give_the_index_of_the_concatenated_cloud(amd, mid_scene, with_normals):
    sc = mid_scene
    rng = np.random.default_rng(4)
    m = sc["map"].shape[0]
    base, nb = sc["map"][: m * 6 // 10].copy(), sc["normals"][: m * 6 // 10].copy()
    kw = dict(minimizer=2 if with_normals else 1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=12)
    grow = amd.ICPSequence(**kw)
    assert grow.setMap(base, nb if with_normals else None)
    cur, curn = base, nb
    lo = m * 6 // 10
    for step in range(3):
        hi = lo + m // 10
        delta = sc["map"][lo:hi].copy()
        dn = sc["normals"][lo:hi].copy()
        lo = hi
        # the delta lies inside the box of the base cloud (same surfaces): an append through the resident-map update, nothing rejected
        app, new_m = grow.mapUpdatePointDistance(delta, 0.0, normals_knn=0, scan_normals=dn if with_normals else None)
        assert app == delta.shape[0]
        cur = np.concatenate([cur, delta]); curn = np.concatenate([curn, dn])
        assert new_m == cur.shape[0]
        assert builds(grow)["ins"] == step + 1, builds(grow)
        fresh = amd.ICPSequence(**kw)
        assert fresh.setMap(cur, curn if with_normals else None)
        assert np.array_equal(grow.getMapMean(), fresh.getMapMean())
        got = grow.getMap(with_normals=with_normals)
        assert np.array_equal(got[0] if with_normals else got, cur)
        q = queries(rng, cur, 4000)
        q[:, :3] -= fresh.getMapMean()
        for k, r in ((1, 2.0), (6, np.inf), (6, 0.7)):
            ia, da = grow.knn(q, k=k, max_dist=r)
            ib, db = fresh.knn(q, k=k, max_dist=r)
            assert np.array_equal(ia, ib) and np.array_equal(da, db), (step, k, r)
        Ta = grow(sc["scan"]); Tb = fresh(sc["scan"])
        assert np.array_equal(Ta, Tb) and grow.stats.pairs == fresh.stats.pairs


def test_delta_outside_the_box_rebuilds(amd, small_scene):
    sc = small_scene
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=8)
    assert icp.setMap(sc["map"])
    out = sc["map"][:500].copy(); out[:, 0] += 300.0                      # far outside the indexed box
    app, new_m = icp.mapUpdatePointDistance(out, 0.0)
    assert app == 500 and builds(icp)["ins"] == 0 and builds(icp)["full"] >= 2
    fresh = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=8)
    assert fresh.setMap(np.concatenate([sc["map"], out]))
    assert np.array_equal(icp(sc["scan"]), fresh(sc["scan"]))


def test_point_distance_updates_use_the_grown_raw_index(amd, oracle, mid_scene):
    """Map::updateLocalPointCloud with PointDistanceMapperModule, three times: the keep decisions are taken in the map's own frame
    (PointDistanceMapperModule.cpp:33-42 searches the map as it is) -- on the raw twin of the grown registration index -- and must be
    the oracle's search of the grown raw map, mask for mask."""
    sc = mid_scene
    half = sc["map"][::2].copy()
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=8)
    assert icp.setMap(half)
    rng = np.random.default_rng(9)
    cur = half
    for step in range(3):
        scan = sc["map"][1::2][rng.integers(0, half.shape[0], 15000)].copy()
        scan[:, :3] += rng.normal(0, 0.05, (15000, 3)).astype(np.float32)
        app, new_m, keep = icp.mapUpdatePointDistance(scan, 0.15, return_keep=True)
        want = oracle.point_distance_keep(cur, scan, 0.15, nthreads=8)
        assert np.array_equal(keep, want), (step, int((keep != want).sum()))
        cur = np.concatenate([cur, scan[keep]])
        assert new_m == cur.shape[0]
    b = builds(icp)
    # the registration index grew by inserts; the raw-frame searches ran on its raw twin (a view: no second index was built at all)
    assert b["ins"] >= 2 and b["raw_view"] >= 3 and b["raw_full"] == 0, b


def test_two_hundred_appends_stay_bit_equal_to_a_fresh_build(amd, mid_scene):
    """VERDICT r4 (weak 11): the incremental centroid is `(sum_raw + sum(delta)) / m1` carried in double from insert to insert, a fresh build
    sums per-block partials in another order, and the level-0 cell edge stays tuned to the density of the FIRST build.  200 small appends in a
    row, all served by the insert; after appends 1, 10, 50 and 200 the index must still answer like a fresh build of the concatenated cloud:
    same centroid float, kNN ids / d^2 bit for bit (k = 1 and 6, radius and unbounded), same registration.  The occupied-cell count of
    icpmi_get_grid_info follows the growth and agrees with a recount of the resident cloud on the handle's own grid."""
    sc = mid_scene
    rng = np.random.default_rng(21)
    m = sc["map"].shape[0]
    nbase = m * 6 // 10
    kw = dict(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=10)
    grow = amd.ICPSequence(**kw)
    assert grow.setMap(sc["map"][:nbase].copy())
    occ0 = grow.gridInfo()["n_occupied"]
    rest = sc["map"][nbase:]
    per = rest.shape[0] // 200
    cur = [sc["map"][:nbase]]
    last_occ = occ0
    for step in range(1, 201):
        delta = rest[(step - 1) * per: step * per].copy()
        app, new_m = grow.mapUpdatePointDistance(delta, 0.0, normals_knn=0)
        cur.append(delta)
        assert app == per and new_m == nbase + step * per
        b = builds(grow)
        assert b["ins"] == step and b["full"] == 1, (step, b)          # every append went through the insert
        gi = grow.gridInfo()
        assert last_occ <= gi["n_occupied"] <= gi["n_cells"]
        last_occ = gi["n_occupied"]
        if step in (1, 10, 50, 200):
            cloud = np.concatenate(cur)
            fresh = amd.ICPSequence(**kw)
            assert fresh.setMap(cloud)
            assert np.array_equal(grow.getMapMean(), fresh.getMapMean()), step
            q = queries(rng, cloud, 3000)
            q[:, :3] -= fresh.getMapMean()
            for k, r in ((1, 2.0), (1, np.inf), (6, np.inf), (6, 0.7)):
                ia, da = grow.knn(q, k=k, max_dist=r)
                ib, db = fresh.knn(q, k=k, max_dist=r)
                assert np.array_equal(ia, ib) and np.array_equal(da, db), (step, k, r)
            assert np.array_equal(grow(sc["scan"]), fresh(sc["scan"])), step
    assert last_occ > occ0                                             # 80 k new points do occupy new cells
    # recount on the handle's own grid (box of the first build, moved with the centroid; overhanging points clamp into border cells); the
    # arithmetic at cell walls is the kernel's, not numpy's: 1 % tolerance
    gi = grow.gridInfo()
    mean = grow.getMapMean().astype(np.float32)
    cloud = np.concatenate(cur)[:, :3]
    o = sc["map"][:nbase, :3].min(0) - mean
    idx = np.floor((cloud - mean - o) / np.float32(gi["cell"])).astype(np.int64)
    idx = np.clip(idx, 0, np.array(gi["dims"]) - 1)
    recount = np.unique(idx[:, 0] + gi["dims"][0] * (idx[:, 1] + gi["dims"][1] * idx[:, 2])).shape[0]
    assert abs(recount - gi["n_occupied"]) <= 0.01 * recount, (recount, gi)
