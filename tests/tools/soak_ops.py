#!/usr/bin/env python
"""Soak test of the stage / map-side operators against the oracle (bit-exact where the spec says so)."""
import sys, os, time, math
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import norlab_icp_mapper_amd as pkg
import oracle_bindings as ob

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
base = pkg.synth.make_scene(m=200_000, n=30_000)
t0 = time.time(); fails = 0
def cloud(n, src):
    c = src[rng.permutation(src.shape[0])[:n]].copy()
    if rng.random() < 0.3: c[:, :3] *= np.float32(rng.choice([0.01, 1.0, 30.0]))
    if rng.random() < 0.3: c[:, :3] += rng.uniform(-2000, 2000, 3).astype(np.float32)
    if rng.random() < 0.2 and n > 4: c[: n // 4] = c[n // 4: 2 * (n // 4)]          # duplicates
    return c
for case in range(cases):
    try:
        m = int(rng.choice([1, 2, 9, 200, 3000, 40_000, 200_000])); n = int(rng.choice([1, 3, 100, 2500, 30_000]))
        mp = cloud(m, base["map"]); q = cloud(n, base["scan"])
        if rng.random() < 0.5: q[:, :3] = mp[rng.integers(0, m, n), :3] + rng.normal(0, rng.choice([0, 1e-3, 0.2]), (n, 3)).astype(np.float32)
        k = int(rng.choice([1, 1, 2, 3, 6, 8, 9, 12])); md = float(rng.choice([0.05, 0.7, math.inf]))
        icp = pkg.ICPSequence(minimizer=0, knn=k, max_dist=md if math.isfinite(md) else 2.0)
        icp.setMap(mp); mean = icp.getMapMean()
        cm = mp.copy(); cm[:, :3] -= mean[None, :3]; cq = q.copy(); cq[:, :3] -= mean[None, :3]
        allow = bool(rng.integers(0, 2))
        ids, d2 = icp.knn(cq, k=k, max_dist=md, allow_self=allow)
        if m * n <= 40_000 * 30_000:
            rids, rd2 = ob.knn(cm, cq, k=k, max_dist=md, allow_self=allow, nthreads=16)
            assert np.array_equal(d2, rd2) and np.array_equal(ids, rids), ("knn", m, n, k, md, allow)
        if case % 3 == 0:
            edge = float(rng.choice([0.05, 0.4, 3.0]))
            assert np.array_equal(icp.voxelKeepFirst(mp, edge), ob.voxel_keep_first(mp, edge)), ("voxel", m, edge)
        if case % 3 == 1 and m * n <= 40_000 * 30_000:
            dmin = float(rng.choice([0.0, 0.05, 0.5]))
            assert np.array_equal(icp.pointDistanceKeep(mp, q, dmin), ob.point_distance_keep(mp, q, dmin, nthreads=16)), ("keep", m, n, dmin)
        if case % 3 == 2 and m <= 40_000 and m >= 12:
            kk = int(rng.choice([3, 7, 10, 16]))
            a = icp.surfaceNormals(mp, knn=kk); b = ob.surface_normals(mp, kk)
            dots = np.abs((a * b).sum(1))
            assert (dots > 0.999).mean() > 0.995 or m < 50, ("normals", m, kk, float((dots > 0.999).mean()))
        if case % 4 == 3 and m >= 9 and m * n <= 40_000 * 30_000:
            pose = pkg.synth.make_T(tuple(rng.uniform(-1.5, 1.5, 3)), tuple(mp[rng.integers(0, m), :3] + rng.uniform(-5, 5, 3))).astype(np.float32)
            to_sensor = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
            nr = rng.normal(0, 1, (m, 3)).astype(np.float32); nr /= np.linalg.norm(nr, axis=1, keepdims=True)
            prob0 = rng.uniform(0, 1, m).astype(np.float32)
            prm = dict(beam_half_angle=float(rng.choice([0.002, 0.01, 0.05, 0.4])), sensor_max_range=float(rng.choice([5.0, 80.0, 1e4])),
                       threshold_dynamic=float(rng.choice([0.3, 0.6, 0.95])), epsilon_a=float(rng.choice([0.001, 0.01, 0.1])),
                       epsilon_d=float(rng.choice([0.001, 0.01, 0.3])))
            got = icp.dynamicPointsUpdate(to_sensor, q, mp, nr, prob0, **prm)
            ref = ob.dynamic_points_update(to_sensor, q, mp, nr, prob0, nthreads=16, **prm)
            same = got == ref
            both_nan = np.isnan(got) & np.isnan(ref)
            assert (same | both_nan).all(), ("dynpts", m, n, prm, int((~(same | both_nan)).sum()))
        icp.close()
    except AssertionError as e:
        fails += 1; print("CASE", case, "FAILED:", e)
print(f"soak_ops: {cases} cases, {fails} failures, {time.time() - t0:.1f} s")
sys.exit(1 if fails else 0)
