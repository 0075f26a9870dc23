"""BASELINE.json configurations 3, 4 and 5 under `-m gpu`, against the CPU oracle.

config 3  point-to-plane against SurfaceNormalDataPointsFilter normals, COMPOSED: the GPU computes the normals and
          registers against them; the oracle computes its own normals and registers against those.
config 4  the full examples/ trajectory replay: all 14 bundled scans (tests/golden/bundled_scans_all.npz) through the C++
          host shell (`build_map_from_scans_and_trajectory`, the reference's harness and its lexicographic scan / trajectory
          pairing, examples/build_map_from_scans_and_trajectory.cpp:191-232) with the shipped configuration switched to
          PointToPlane / epsilon 0 / samplingMethod 0, against an oracle-side replay of
          Mapper::processInput / Map::updateLocalPointCloud (tests/oracle_mapper.py) -- pose per scan.
config 5  the 10 M-point map on one GPU: kNN ids / d^2 and one full point-to-plane registration against the oracle on a
          query subset the oracle can afford, plus size-independent properties at the full 100 k-point reading.
"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "norlab_icp_mapper_amd")
POSE_TOL_M, POSE_TOL_RAD = 1e-4, 1e-4


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


# ------------------------------------------------------------------------------------------------ config 3
def test_config3_gpu_normals_then_point_to_plane_against_oracle_normals_then_oracle(amd, oracle, mid_scene):
    sc = mid_scene
    kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
    icp = amd.ICPSequence(**kw)
    n_gpu = icp.surfaceNormals(sc["map"], knn=10)
    n_cpu = oracle.surface_normals(sc["map"], knn=10, nthreads=8)
    # the two normal fields agree point by point up to sign (unoriented PCA normals) and rounding of the eigen-solve
    dots = np.abs(np.einsum("ij,ij->i", n_gpu, n_cpu))
    deg = np.linalg.norm(n_cpu, axis=1) < 0.5                     # rank-deficient neighbourhoods: zero normal on both sides
    assert np.array_equal(deg, np.linalg.norm(n_gpu, axis=1) < 0.5)
    assert (dots[~deg] > 1 - 1e-4).all(), float(dots[~deg].min())
    assert icp.setMap(sc["map"], n_gpu)
    T = icp(sc["scan"])
    oicp = oracle.OracleICP(oracle.make_config(nthreads=8, **kw))
    oicp.setMap(sc["map"], n_cpu)
    err, T_ref = oicp(sc["scan"])
    assert err == 0
    assert icp.stats.iterations == oicp.stats.iterations and icp.stats.stop_reason == oicp.stats.stop_reason
    assert icp.stats.pairs == oicp.stats.pairs
    dt, dr = amd.synth.pose_error(T, T_ref)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)
    gt, gr = amd.synth.pose_error(T, sc["T_gt"])
    assert gt < 5e-3 and gr < 5e-4, (gt, gr)


# ------------------------------------------------------------------------------------------------ configs 2 / 3 at the BASELINE size
@pytest.fixture(scope="module")
def full_scene():
    from norlab_icp_mapper_amd import synth
    return synth.make_scene(m=1_000_000, n=100_000)       # BASELINE.json configs 2 / 3: 100 k-point scan vs 1 M-point map


@pytest.mark.parametrize("name,minimizer,normals", [("config2_p2p", 1, "analytic"), ("config3_p2plane", 2, "analytic"), ("config3_p2plane_filter", 2, "filter")])
def test_full_size_checked_registration_matches_oracle(amd, oracle, full_scene, name, minimizer, normals):
    """The production-shape chain (Counter 40 + Differential, SURVEY.md 8d chains A / B) at the FULL BASELINE size against the oracle:
    iterations, stop reason, pairs of the last iteration, trimmed limit and pose (VERDICT r3 missing 4: until r4 this comparison
    existed at full size only inside bench.py).  The oracle registers in ~1 s on 16 threads; its kd-tree build is the long part."""
    sc = full_scene
    kw = dict(minimizer=minimizer, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
    icp = amd.ICPSequence(**kw)
    nrm = sc["normals"]
    if normals == "filter":                                   # config 3 (ii): SurfaceNormalDataPointsFilter{knn: 10} run on the map
        nrm = icp.surfaceNormals(sc["map"], knn=10)
    assert icp.setMap(sc["map"], nrm)
    T = icp(sc["scan"])
    o = oracle.OracleICP(oracle.make_config(nthreads=min(16, len(os.sched_getaffinity(0))), **kw))
    o.setMap(sc["map"], nrm)
    err, T_ref = o(sc["scan"])
    assert err == 0
    assert icp.stats.iterations == o.stats.iterations and icp.stats.stop_reason == o.stats.stop_reason, (icp.stats.iterations, o.stats.iterations)
    assert icp.stats.pairs == o.stats.pairs
    assert icp.stats.trimmed_limit == o.stats.trimmed_limit
    dt, dr = amd.synth.pose_error(T, T_ref)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (name, dt, dr)
    if minimizer == 2:       # (point-to-point slides along the walls: the Differential checker stops it ~0.1 m short, on both sides alike)
        gt, gr = amd.synth.pose_error(T, sc["T_gt"])
        assert gt < 5e-3 and gr < 5e-4, (gt, gr)
    # ... and throughput mode (what bench.py times: Counter only, fixed 20 iterations) lands where the oracle lands
    import torch
    d = torch.from_numpy(sc["scan"]).cuda()
    T20 = icp.registerDev(d.data_ptr(), d.shape[0], fixed_iterations=20)
    o20 = oracle.OracleICP(oracle.make_config(nthreads=min(16, len(os.sched_getaffinity(0))), **dict(kw, max_iterations=20, use_differential=0)))
    o20.setMap(sc["map"], nrm)
    err, T20_ref = o20(sc["scan"])
    assert err == 0 and icp.stats.iterations == 20 == o20.stats.iterations
    dt, dr = amd.synth.pose_error(T20, T20_ref)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (name, "fixed 20", dt, dr)


# ------------------------------------------------------------------------------------------------ config 4
from config4_data import CONFIG4_YAML, quat_T as _quat_T, write_bundled_dataset as _write_bundled_dataset, oracle_mapper_args  # noqa: E402


def test_config4_full_bundled_trajectory_replay_against_oracle_replay(amd, oracle, tmp_path):
    from test_host_cpp import _build_host, _read_vtk
    import oracle_mapper as om
    _build_host()
    z = np.load(os.path.join(ROOT, "tests", "golden", "bundled_scans_all.npz"))
    tmp = str(tmp_path)
    names, traj = _write_bundled_dataset(tmp, z)
    # the example pairs trajectory row i with the i-th file in LEXICOGRAPHIC order (cpp:191): the fixture keeps that order,
    # and it is NOT chronological -- cloud_1690309710_85582848 (t = 710.0856 s) sorts last
    assert names == sorted(names) and len(names) == 14 and names[-1].startswith("cloud_1690309710_85582848")
    stamps_from_names = [int(n.split("_")[1]) * 10**9 + int(n.split("_")[2].split(".")[0]) for n in names]
    assert stamps_from_names != sorted(stamps_from_names)
    cfg = os.path.join(tmp, "config.yaml")
    open(cfg, "w").write(CONFIG4_YAML)
    traj_out = os.path.join(tmp, "traj.vtk")
    out = subprocess.run([os.path.join(PKG, "build_map_from_scans_and_trajectory"), tmp, cfg, traj_out], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr + out.stdout
    assert out.stdout.count("iterations 10") == 13, out.stdout          # scan 1 creates the map, 13 registrations of 10 passes each
    pos, desc = _read_vtk(traj_out)
    assert pos.shape[0] == 14
    cpp = []
    for i in range(14):
        T = np.eye(4, dtype=np.float32)
        T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = desc["orientationX"][i], desc["orientationY"][i], desc["orientationZ"][i], pos[i]
        cpp.append(T)

    # ---- the oracle's replay of the same trajectory ----
    nt = min(16, len(os.sched_getaffinity(0)))
    a_icp, a_mod, a_kw = oracle_mapper_args(nt)
    mapper = om.OracleMapper(a_icp, a_mod, **a_kw)
    worst = (0.0, 0.0)
    moved = 0.0
    for i in range(14):
        cloud = mapper.apply_input_filters(z[f"scan{i}_xyz"])
        prior = _quat_T(traj[i, 2:])
        T_ref = mapper.process_input(cloud, prior, traj[i, 0] + traj[i, 1] * 1e-9)
        dt, dr = amd.synth.pose_error(cpp[i], T_ref)
        assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (i, names[i], dt, dr)
        worst = (max(worst[0], dt), max(worst[1], dr))
        moved = max(moved, amd.synth.pose_error(T_ref, prior)[0])
    assert mapper.iterations == [0] + [10] * 13 and all(mapper.updated)          # delay 0.05 s: every scan updates the map
    assert moved > 1e-3                                                          # the registrations do correct the priors
    # the maps agree in size (the decimation and the probability cut are index / threshold decisions on near-identical clouds)
    mp, mdesc = _read_vtk(os.path.join(tmp, "map.vtk"))
    m_ref = mapper.map["xyz1"].shape[0]
    assert abs(mp.shape[0] - m_ref) <= max(5, m_ref // 200), (mp.shape[0], m_ref)
    assert {"normals", "probabilityDynamic"} <= set(mdesc)
    print(f"config 4: 14 scans, worst pose difference to the oracle replay {worst[0]:.2e} m / {worst[1]:.2e} rad, map {mp.shape[0]} vs {m_ref} points")


# ------------------------------------------------------------------------------------------------ config 5
@pytest.fixture(scope="module")
def scene_10m(amd):
    return amd.synth.make_scene(m=10_000_000, n=100_000, scale=3.16)


def test_config5_knn_on_the_10M_map_matches_oracle(amd, oracle, scene_10m):
    sc = scene_10m
    icp = amd.ICPSequence(minimizer=2, max_dist=2.0)
    assert icp.setMap(sc["map"], sc["normals"])
    mean = icp.getMapMean()
    rng = np.random.default_rng(5)
    q = sc["scan"][rng.permutation(sc["scan"].shape[0])[:5000]].copy()
    q[:, :3] -= mean[None, :]
    ref = sc["map"].copy(); ref[:, :3] -= mean[None, :]
    for k, md in ((1, 2.0), (6, 2.0)):
        ids, d2 = icp.knn(q, k=k, max_dist=md)
        oid, od2 = oracle.knn(ref, q, k=k, max_dist=md, nthreads=16)
        assert np.array_equal(d2, od2)
        assert np.array_equal(ids, oid)


def test_config5_registration_on_the_10M_map(amd, oracle, scene_10m):
    sc = scene_10m
    kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
    icp = amd.ICPSequence(**kw)
    assert icp.setMap(sc["map"], sc["normals"])
    # (a) a 5 k-point reading the oracle can afford: iterations, stop reason, pairs, pose
    rng = np.random.default_rng(6)
    sub = np.ascontiguousarray(sc["scan"][np.sort(rng.permutation(sc["scan"].shape[0])[:5000])])
    T = icp(sub)
    oicp = oracle.OracleICP(oracle.make_config(nthreads=16, **kw))
    oicp.setMap(sc["map"], sc["normals"])
    err, T_ref = oicp(sub)
    assert err == 0
    assert (icp.stats.iterations, icp.stats.stop_reason, icp.stats.pairs) == (oicp.stats.iterations, oicp.stats.stop_reason, oicp.stats.pairs)
    dt, dr = amd.synth.pose_error(T, T_ref)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)
    # (b) the full 100 k-point reading: size-independent properties.  (Not "recovers T_gt": at this scale the 60 m scan sees
    # little but floor and ceiling, x / y / yaw are barely constrained and the minimum-norm solve leaves them where they are.)
    T_full = icp(sc["scan"])
    it_full = icp.stats.iterations
    assert np.array_equal(icp(sc["scan"]), T_full) and icp.stats.iterations == it_full   # bitwise reproducible
    assert 0.84 < icp.stats.point_used_ratio <= 0.8501                          # TrimmedDist 0.85 keeps rank floor(0.85 n) + ties
    assert icp.stats.hard_queries == 0                                          # the pyramid decides every bounded-radius query
    gt = sc["T_gt"]
    assert abs(T_full[2, 3] - gt[2, 3]) < 5e-3                                  # the constrained directions are recovered: height,
    assert np.abs(T_full[2, :2] - gt[2, :2]).max() < 5e-4                       # roll and pitch (third row of R)
    # the registration does not end on a larger objective than the ground-truth pose has: trimmed mean of the squared
    # point-to-plane residuals (x / y may slide along the planes, which point-to-point distances would punish)
    mean = icp.getMapMean()

    def objective(T):
        moved = icp.transform(T, sc["scan"])
        q = moved.copy(); q[:, :3] -= mean[None, :]
        ids, d2 = icp.knn(q, k=1, max_dist=2.0)
        ok = ids[:, 0] >= 0
        m = sc["map"][ids[ok, 0], :3]; nrm = sc["normals"][ids[ok, 0]]
        r2 = np.sort(np.einsum("ij,ij->i", moved[ok, :3] - m, nrm) ** 2)
        return float(r2[: int(0.85 * r2.shape[0])].mean())
    assert objective(T_full) <= 1.05 * objective(gt)
    # idempotence: registering the reading moved by the correction moves it by less than what the Differential checker stops
    # at (1e-3 m / 1e-3 rad per iteration, mean over 3 iterations)
    T_again = icp(icp.transform(T_full, sc["scan"]))
    mt, mr = amd.synth.pose_error(T_again, np.eye(4, dtype=np.float32))
    assert mt < 3e-3 and mr < 3e-3, (mt, mr)


# ------------------------------------------------------------------------------------------------ cell paging (a12 / f4)
PAGING_CONFIG = """
post:
  - SurfaceNormalDataPointsFilter:
      knn: 10
mapper:
  updateCondition:
    type: distance
    value: 2.0
  mapperModule:
    - PointDistanceMapperModule:
        minDistNewPoint: 0.3
  sensorMaxRange: 25
icp:
  matcher:
    KDTreeMatcher:
      knn: 1
      maxDist: 2.0
      epsilon: 0
  outlierFilters:
    - TrimmedDistOutlierFilter:
        ratio: 0.85
  errorMinimizer:
    PointToPlaneErrorMinimizer:
  transformationCheckers:
    - CounterTransformationChecker:
        maxIterationCount: 30
    - DifferentialTransformationChecker:
        minDiffRotErr: 0.001
        minDiffTransErr: 0.001
        smoothLength: 3
  inspector: NullInspector
"""


def _make_corridor_dataset(tmp, amd, out_steps=26, back_steps=18, step=5.0, n_pts=5000, scale=2.4):
    """a sensor driving 125 m along x through the (scaled) synthetic hall and 90 m back, seeing 25 m: with 20 m cells, a window of
    sensorMaxRange + 2 buffer cells and a hysteresis of 2 cells the map pages cells out behind it on the way out and loads
    them again -- points included -- on the way back (Map.cpp:246-460)"""
    from test_host_cpp import _write_vtk, _quat
    os.makedirs(os.path.join(tmp, "scans"))
    rows, scans, priors = [], [], []
    x0 = -82.0
    along = list(range(out_steps)) + list(range(out_steps - 2, out_steps - 2 - back_steps, -1))
    for s, k in enumerate(along):
        T_true = amd.synth.make_T((0.0, 0.0, 0.01 * k), (x0 + step * k, 1.0 + 0.05 * k + (0.4 if s >= out_steps else 0.0), 0.0))
        pts, _ = amd.synth.sample_surfaces(60 * n_pts, seed=700 + s, scale=scale)
        sensor = T_true[:3, 3] + np.array([0.0, 0.0, 1.5])
        pts = pts[np.linalg.norm(pts - sensor, axis=1) < 24.0][:n_pts]
        pts = pts + np.stack([amd.synth.gaussian(1100 + s, 2 * r, len(pts)) for r in range(3)], 1) * 0.01
        Ti = np.linalg.inv(T_true)
        local = (pts @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
        T_prior = amd.synth.make_T((0.002, -0.002, 0.001), (0.02, -0.015, 0.01)) @ T_true if s else T_true
        q = _quat(T_prior[:3, :3])
        rows.append([1700000000, 100000000 * s, *T_prior[:3, 3], *q])
        _write_vtk(os.path.join(tmp, "scans", f"cloud_{s:03d}.vtk"), local)
        scans.append(local)
    with open(os.path.join(tmp, "trajectory.csv"), "w") as f:
        f.write("header.stamp.sec,header.stamp.nanosec,header.frame_id,child_frame_id,pose.pose.position.x,pose.pose.position.y,"
                "pose.pose.position.z,pose.pose.orientation.x,pose.pose.orientation.y,pose.pose.orientation.z,pose.pose.orientation.w,pose.covariance\n")
        for r in rows:
            f.write(f"{r[0]},{r[1]},map,base_link," + ",".join(repr(float(v)) for v in r[2:]) + ",[0. 0. 0.]\n")
    for r in rows:
        priors.append(_quat_T(np.array(r[2:], dtype=np.float64)))
    return scans, priors, [r[0] + r[1] * 1e-9 for r in rows]


def test_cell_paging_replay_against_oracle_replay(amd, oracle, tmp_path):
    """Map::updatePose / loadCells / unloadCells / RAMCellManager under the GPU-backed host shell: a drive long enough to page
    cells out (and, on the way back in the second half of the list, in again), scan by scan against the oracle-side replay with
    the reference's paging restated (tests/oracle_mapper.py: update_pose) -- poses, and the global map (local cloud + saved
    cells, Mapper::getMap) as a point set."""
    from test_host_cpp import _build_host, _read_vtk
    import oracle_mapper as om
    _build_host()
    tmp = str(tmp_path)
    scans, priors, stamps = _make_corridor_dataset(tmp, amd)
    cfg = os.path.join(tmp, "config.yaml")
    open(cfg, "w").write(PAGING_CONFIG)
    traj_out = os.path.join(tmp, "traj.vtk")
    out = subprocess.run([os.path.join(PKG, "build_map_from_scans_and_trajectory"), tmp, cfg, traj_out], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr + out.stdout
    pos, desc = _read_vtk(traj_out)
    mapper = om.OracleMapper(dict(knn=1, max_dist=2.0, minimizer=2, outliers=[(4, 0.85)], max_iterations=30, use_differential=1),
                             [("point_distance", 0.3)], post=[("surface_normals", 10)], update=("distance", 2.0), sensor_max_range=25.0,
                             nthreads=min(16, len(os.sched_getaffinity(0))), paging=True)
    worst = 0.0
    sizes = []
    for i, (scan, prior, stamp) in enumerate(zip(scans, priors, stamps)):
        cloud = mapper.apply_input_filters(scan)
        T_ref = mapper.process_input(cloud, prior, stamp)
        sizes.append(mapper.map["xyz1"].shape[0])
        T = np.eye(4, dtype=np.float32)
        T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = desc["orientationX"][i], desc["orientationY"][i], desc["orientationZ"][i], pos[i]
        dt, dr = amd.synth.pose_error(T, T_ref)
        if not (dt <= POSE_TOL_M and dr <= POSE_TOL_RAD):
            import re
            its = re.findall(r"iterations (\d+)  overlap \S+  local map (\d+)", out.stdout)
            raise AssertionError((i, dt, dr, "harness (iterations, local map)", its[max(0, i - 8):i + 1], "oracle local map", sizes[max(0, i - 8):i + 1],
                                  "oracle iterations", mapper.iterations[max(0, i - 8):i + 1], "page events", mapper.page_events))
        worst = max(worst, dt)
    unloads = [e for e in mapper.page_events if e[0] == "unload"]
    loads = [e for e in mapper.page_events if e[0] == "load"]
    assert len(unloads) >= 4 and len(loads) >= 4, mapper.page_events            # the window moved out and back
    assert mapper.reloaded_points > 0                                            # and points saved on the way out came back into the local map
    assert any(c["xyz1"].shape[0] > 0 for cid, c in mapper.cells.items() if cid not in mapper.loaded)   # and cells hold points that left the local map
    # Mapper::getMap(): local cloud + saved cells not loaded -- the same point set on both sides
    mp, mdesc = _read_vtk(os.path.join(tmp, "map.vtk"))
    ref = mapper.get_map()["xyz1"][:, :3]
    assert mp.shape[0] == ref.shape[0], (mp.shape, ref.shape)
    a = np.sort(np.ascontiguousarray(mp).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel())
    b = np.sort(np.ascontiguousarray(ref).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel())
    same = np.mean(a == b)
    assert same > 0.999, same        # (ASCII VTK keeps 9 significant digits: exact round trip of float32)
    local_n = mapper.map["xyz1"].shape[0]
    assert local_n < ref.shape[0]                                                # part of the map lives in the cell manager
    print(f"paging: {len(scans)} scans, {len(unloads)} unloads / {len(loads)} loads, local {local_n} of {ref.shape[0]} points, worst pose difference {worst:.2e} m")


def test_set_map_repages_the_global_cloud(amd, oracle, tmp_path):
    """Mapper::setMap -> Map::setGlobalPointCloud (Mapper.cpp:295-301, Map.cpp:575-588; SURVEY section 5 'checkpoint / resume', 8f rank 4):
    in the middle of the paging drive the whole map (local cloud + saved cells, Mapper::getMap) is taken out and handed back; the next
    updatePose must page it into cells again and the drive must go on as the oracle-side replay of the same sequence does -- poses
    scan by scan, cell traffic, and the final global map as a point set (VERDICT r2 missing 4: this path had no test)."""
    from test_host_cpp import _build_host, _read_vtk
    import oracle_mapper as om
    _build_host()
    tmp = str(tmp_path)
    scans, priors, stamps = _make_corridor_dataset(tmp, amd, out_steps=22, back_steps=12)
    cfg = os.path.join(tmp, "config.yaml")
    open(cfg, "w").write(PAGING_CONFIG)
    traj_out = os.path.join(tmp, "traj.vtk")
    AT = 17
    env = dict(os.environ, NIM_SETMAP_AT=str(AT))
    out = subprocess.run([os.path.join(PKG, "build_map_from_scans_and_trajectory"), tmp, cfg, traj_out], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr + out.stdout
    assert f"handed back after scan {AT + 1}" in out.stdout
    pos, desc = _read_vtk(traj_out)
    assert pos.shape[0] == len(scans) - (AT + 1)                              # Mapper::setMap clears the trajectory
    mapper = om.OracleMapper(dict(knn=1, max_dist=2.0, minimizer=2, outliers=[(4, 0.85)], max_iterations=30, use_differential=1),
                             [("point_distance", 0.3)], post=[("surface_normals", 10)], update=("distance", 2.0), sensor_max_range=25.0,
                             nthreads=min(16, len(os.sched_getaffinity(0))), paging=True)
    worst, handed = 0.0, None
    for i, (scan, prior, stamp) in enumerate(zip(scans, priors, stamps)):
        T_ref = mapper.process_input(mapper.apply_input_filters(scan), prior, stamp)
        if i == AT:
            handed = mapper.get_map()
            events_before = len(mapper.page_events)
            mapper.set_map(handed)
            assert mapper.first_pose_update
        if i > AT:
            r = i - (AT + 1)
            T = np.eye(4, dtype=np.float32)
            T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = desc["orientationX"][r], desc["orientationY"][r], desc["orientationZ"][r], pos[r]
            dt, dr = amd.synth.pose_error(T, T_ref)
            assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (i, dt, dr)
            worst = max(worst, dt)
    assert f"setMap: {handed['xyz1'].shape[0]} points" in out.stdout            # the same number of points went through getMap / setMap
    assert not mapper.first_pose_update                                        # the first updatePose after setMap re-paged the cloud
    assert len(mapper.page_events) > events_before                             # ... and the window kept moving afterwards
    assert any(c["xyz1"].shape[0] > 0 for cid, c in mapper.cells.items() if cid not in mapper.loaded)
    mp, _ = _read_vtk(os.path.join(tmp, "map.vtk"))
    ref = mapper.get_map()["xyz1"][:, :3]
    assert mp.shape[0] == ref.shape[0], (mp.shape, ref.shape)
    a = np.sort(np.ascontiguousarray(mp).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel())
    b = np.sort(np.ascontiguousarray(ref).view([("x", "f4"), ("y", "f4"), ("z", "f4")]).ravel())
    assert np.mean(a == b) > 0.999
    print(f"setMap re-paging: {len(scans)} scans, handed back {handed['xyz1'].shape[0]} points after scan {AT + 1}, worst pose difference {worst:.2e} m")
