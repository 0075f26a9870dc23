"""Randomised differential test: chain configurations drawn from a seeded generator (minimiser x knn x maxDist x outlier-filter stack x
checkers x scene size / motion), the HIP path against the CPU oracle on the same inputs.  What must agree: the error class, the number of
iterations, the stop reason, the number of pairs, the pose within 1e-4 m / 1e-4 rad (BASELINE.json's tolerance).  The targeted tests pick
their configurations by hand; this one walks combinations nobody thought of.  ICPMI_FUZZ_N widens the walk (default 24 draws, ~1 s each);
ICPMI_FUZZ_SEED moves it."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4
MAXD, MIND, MED, TRIM, SNO, GEN, ROB, VT = 1, 2, 3, 4, 5, 6, 7, 8
N_DRAWS = int(os.environ.get("ICPMI_FUZZ_N", "24"))
SEED0 = int(os.environ.get("ICPMI_FUZZ_SEED", "0"))


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


def draw(rng):
    """one configuration: (scene kwargs, chain kwargs, needs reading normals, generic-descriptor source or None)"""
    scene = dict(m=int(rng.integers(15_000, 60_000)), n=int(rng.integers(1_500, 9_000)), scale=0.25,
                 seed_map=int(rng.integers(1, 1 << 20)), seed_scan=int(rng.integers(1, 1 << 20)), seed_noise=int(rng.integers(1, 1 << 20)),
                 rotvec=tuple(rng.uniform(-0.02, 0.02, 3)), trans=tuple(rng.uniform(-0.12, 0.12, 3)))
    minimizer = int(rng.choice([1, 2, 2]))
    knn = int(rng.choice([1, 1, 1, 2, 3, 6]))
    max_dist = float(rng.choice([0.6, 1.0, 2.0, math.inf]))
    outliers, need_rn, gen = [], False, None
    pool = [MAXD, MIND, MED, TRIM, SNO, ROB, VT, GEN]
    for t in rng.choice(pool, size=int(rng.integers(0, 4)), replace=False):
        t = int(t)
        if t == MAXD: outliers.append((MAXD, float(rng.uniform(0.4, 1.5))))
        elif t == MIND: outliers.append((MIND, float(rng.uniform(0.0, 0.01))))
        elif t == MED: outliers.append((MED, float(rng.uniform(1.5, 4.0))))
        elif t == TRIM: outliers.append((TRIM, float(rng.uniform(0.6, 0.98))))
        elif t == SNO: outliers.append((SNO, float(rng.uniform(0.6, 1.4)))); need_rn = True
        elif t == ROB:
            fct = int(rng.integers(0, 8)); scale = int(rng.integers(0, 2)); dist = int(rng.integers(0, 2)) if minimizer == 2 else 0
            tuning, nb = float(rng.uniform(0.3, 2.0)), float(rng.choice([0, 0, 3]))
            # r5: berg / std / approximation, derived from the tuning's digits so that the draws of the earlier rounds keep their streams
            x = int(tuning * 1e6) % 10
            if x in (0, 1): scale, tuning = 2, tuning * 0.1      # berg: `tuning` is the scale the estimate converges to
            elif x == 2: scale = 3                                # std (a finite maxDist makes it "not a number" on both sides)
            apx = [0.0, 0.0, 0.0, 1.5, 3.0][int(tuning * 1e5) % 5]
            outliers.append((ROB, tuning, fct | (scale << 4) | (dist << 8), nb, apx))
        elif t == GEN:
            gen = str(rng.choice(["reference", "reading"]))
            flags = int(rng.choice([0, 4, 2])) | (1 if gen == "reading" else 0)   # include/icpmi.h: 1 source reading, 2 soft, 4 useLargerThan
            outliers.append((GEN, float(rng.uniform(0.2, 0.8)), flags, 0.0))
        elif t == VT: outliers.append((VT, float(rng.uniform(0.05, 0.4)), 0, float(rng.uniform(0.7, 0.99)), float(rng.uniform(0.8, 2.0))))
    chain = dict(minimizer=minimizer, knn=knn, max_dist=max_dist, outliers=outliers, max_iterations=int(rng.integers(4, 30)),
                 use_differential=int(rng.integers(0, 2)), smooth_length=int(rng.integers(2, 5)),
                 min_diff_rot=float(rng.choice([1e-3, 1e-4])), min_diff_trans=float(rng.choice([1e-3, 1e-4])))
    if minimizer == 2 and rng.integers(0, 8) == 0:
        chain.update(force_4dof=1)
    if rng.integers(0, 4) == 0:
        chain.update(use_bound=1, max_rot_norm=float(rng.choice([0.5, 0.01])), max_trans_norm=float(rng.choice([1.0, 0.05])))
    return scene, chain, need_rn, gen


@pytest.mark.parametrize("i", range(N_DRAWS))
def test_random_chain_matches_oracle(amd, oracle, i):
    rng = np.random.default_rng(1000 * SEED0 + i)
    scene, chain, need_rn, gen = draw(rng)
    sc = amd.synth.make_scene(**scene)
    rn = sc["scan_normals"] if need_rn else None
    oicp = oracle.OracleICP(oracle.make_config(nthreads=8, **chain))
    assert oicp.setMap(sc["map"], sc["normals"])
    icp = amd.ICPSequence(**chain)
    assert icp.setMap(sc["map"], sc["normals"])
    # the 1-row descriptor a GenericDescriptorOutlierFilter reads: of the map (set once) or of the reading (one shot per call)
    srng = np.random.default_rng(7 + i)
    map_s = srng.uniform(0.0, 1.0, sc["map"].shape[0]).astype(np.float32)
    read_s = srng.uniform(0.0, 1.0, sc["scan"].shape[0]).astype(np.float32)

    def arm():
        if gen == "reference":
            icp.setMapScalar(map_s); oicp.setMapScalar(map_s)
        elif gen == "reading":
            icp.setReadingScalar(read_s)

    arm()
    if gen == "reading":
        oicp.setReadingScalar(read_s)
    err, T_ref = oicp(sc["scan"], rn)
    what = (i, scene, chain, gen)
    if err != 0:
        # the same error class: nothing to filter / no match / too few points -> ConvergenceError; a Bound checker past its limit too
        with pytest.raises(amd.ConvergenceError):
            icp(sc["scan"], rn)
        assert icp.stats.iterations == oicp.stats.iterations, what
        return
    T = icp(sc["scan"], rn)
    assert icp.stats.iterations == oicp.stats.iterations, what
    assert icp.stats.stop_reason == oicp.stats.stop_reason, what
    assert icp.stats.pairs == oicp.stats.pairs, what
    dt, dr = amd.synth.pose_error(T, T_ref)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr, what)
    assert abs(icp.errorMinimizer.getOverlap() - oicp.stats.weighted_point_used_ratio) < 5e-6, what
    # and the fixed launch sequence of as many iterations lands on the checked loop's pose
    import torch
    d = torch.from_numpy(np.ascontiguousarray(sc["scan"], dtype=np.float32)).cuda()
    dn = torch.from_numpy(np.ascontiguousarray(rn, dtype=np.float32)).cuda() if need_rn else None
    if gen == "reading":
        icp.setReadingScalar(read_s)
    Tf = icp.registerDev(d.data_ptr(), d.shape[0], fixed_iterations=icp.stats.iterations, d_normals_ptr=dn.data_ptr() if need_rn else None)
    dt, dr = amd.synth.pose_error(Tf, T)
    assert dt <= 1e-6 and dr <= 1e-6, (dt, dr, what)


def draw_map_chain(rng):
    """a mapper-module chain + post filters (Map::updateLocalPointCloud, Map.cpp:502-534) and the clouds of three successive updates"""
    modules = []
    for t in rng.choice(["point_distance", "dynamic_points", "voxel", "octree"], size=int(rng.integers(1, 4)), replace=True):
        if t == "point_distance": modules.append(("point_distance", float(rng.choice([0.0, 0.05, 0.15, 0.3]))))
        elif t == "dynamic_points":
            modules.append(("dynamic_points", float(rng.uniform(0.6, 0.95)), float(rng.uniform(0.5, 0.9)), float(rng.uniform(0.9, 0.999)),
                            float(rng.uniform(0.005, 0.03)), float(rng.uniform(0.005, 0.03)), float(rng.uniform(0.005, 0.03)), float(rng.choice([30.0, 200.0]))))
        elif t == "voxel": modules.append(("voxel", float(rng.uniform(0.03, 0.5)), int(rng.integers(0, 2))))
        else: modules.append(("octree", float(rng.uniform(0.1, 1.0)), 0, int(rng.choice([1, 2, 6]))))
    post = []
    if rng.integers(0, 2): post.append(("surface_normals", int(rng.integers(5, 13))))
    if rng.integers(0, 3) == 0: post.append(("cut_scalar", float(rng.uniform(0.3, 0.9)), int(rng.integers(0, 2))))
    return modules, post


@pytest.mark.parametrize("i", range(max(N_DRAWS // 2, 1)))
def test_random_map_update_chain_matches_oracle(amd, oracle, i):
    """Three successive updates of a resident map by a random module chain (the second and third run on the state the first left:
    the incremental index insert, the raw-frame view, regrown buffers) against the same chain composed from the oracle's operators."""
    from test_gpu_map_chain import host_chain, normals_close
    rng = np.random.default_rng(50_000 + 1000 * SEED0 + i)
    modules, post = draw_map_chain(rng)
    m, n = int(rng.integers(8_000, 40_000)), int(rng.integers(1_000, 7_000))
    sc = amd.synth.make_scene(m=m, n=8, scale=0.25, seed_map=int(rng.integers(1, 1 << 20)))
    base = sc["map"].copy()
    base_n = oracle.surface_normals(base, knn=8, nthreads=8)
    base_s = rng.uniform(0.0, 1.0, m).astype(np.float32)
    icp = amd.ICPSequence(minimizer=1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=5, use_differential=0)
    icp.setMap(base, base_n); icp.setMapScalar(base_s)
    pts, nrm, scal = base, base_n, base_s
    what = (i, modules, post, m, n)
    for u in range(3):
        pick = rng.permutation(pts.shape[0])[:min(n, pts.shape[0])]
        scan = pts[pick].copy()
        scan[:, :3] += rng.normal(0, float(rng.choice([0.02, 0.1, 0.4])), (scan.shape[0], 3)).astype(np.float32)
        scan_s = np.full(scan.shape[0], float(rng.uniform(0.3, 0.8)), np.float32)
        pose = amd.synth.make_T(tuple(rng.uniform(-0.05, 0.05, 3)), tuple(rng.uniform(-3.0, 3.0, 3))).astype(np.float32)
        to_sensor = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
        src, m_new, _ = icp.mapUpdateChain(scan, modules, post, scan_scalar=scan_s, to_sensor=to_sensor, with_prefix=True)
        pts, nrm, scal, ref_src = host_chain(oracle, pts, nrm, scal, scan, scan_s, to_sensor, modules, post)
        assert m_new == pts.shape[0], (u, what)
        assert np.array_equal(src, ref_src), (u, what)
        got, got_n = icp.getMap(with_normals=True)
        assert np.array_equal(got, pts), (u, what)
        assert np.array_equal(icp.getMapScalar(), scal), (u, what)
        if any(p[0] == "surface_normals" for p in post):
            if pts.shape[0] > 12:
                assert normals_close(got_n, nrm) > 0.995, (u, what)
            nrm = got_n            # both sides go on from the same normals (PCA signs / eigen-solver round-off must not compound)
        else:
            assert np.array_equal(got_n, nrm), (u, what)
        if pts.shape[0] == 0:
            break
