#!/usr/bin/env python
"""Soak test of icpmi_map_update_chain: random programs of map operators, random cloud sizes, several consecutive
updates per case, against the composition of the CPU oracle's single operators (tests/test_gpu_map_chain.py: host_chain).
Bit-exact bar for provenance, points and the scalar descriptor; normals only where no SURFACE_NORMALS step ran."""
import sys, os, time
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import norlab_icp_mapper_amd as pkg
import oracle_bindings as ob
from test_gpu_map_chain import host_chain

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
base = pkg.synth.make_scene(m=60_000, n=20_000)
t0 = time.time(); fails = 0


def cloud(n, src):
    c = src[rng.permutation(src.shape[0])[:n]].copy()
    c[:, :3] += rng.normal(0, rng.choice([0.0, 0.01, 0.2]), (n, 3)).astype(np.float32)
    if rng.random() < 0.2 and n > 4:
        c[: n // 4] = c[n // 4: 2 * (n // 4)]          # duplicates
    return c


def program():
    mods = []
    for _ in range(int(rng.integers(1, 4))):
        t = rng.choice(["pd", "dyn", "vox", "oct"])
        if t == "pd":
            mods.append(("point_distance", float(rng.choice([0.0, 0.05, 0.3, 1.0]))))
        elif t == "dyn":
            mods.append(("dynamic_points", float(rng.choice([0.3, 0.6, 0.9])), 0.8, 0.99, float(rng.choice([0.005, 0.01, 0.05])),
                         float(rng.choice([0.001, 0.01, 0.1])), float(rng.choice([0.001, 0.01, 0.3])), float(rng.choice([30.0, 200.0]))))
        elif t == "oct":   # the real OctreeGridDataPointsFilter: (maxSizeByNode, samplingMethod 0 = first point, maxPointByNode)
            mods.append(("octree", float(rng.choice([0.05, 0.3, 2.0, 0.0])), 0, int(rng.choice([1, 1, 4, 16]))))
        else:
            mods.append(("voxel", float(rng.choice([0.05, 0.3, 2.0, 500.0])), int(rng.integers(0, 2))))
    post = []
    if rng.random() < 0.6:
        post.append(("surface_normals", int(rng.choice([3, 6, 10, 16]))))
    if rng.random() < 0.5:
        post.append(("cut_scalar", float(rng.choice([0.2, 0.5, 0.65, 0.9])), int(rng.integers(0, 2))))
    if rng.random() < 0.15:
        post.reverse()
    return mods, post


for case in range(cases):
    mods, post = program()
    try:
        m = int(rng.choice([0, 0, 12, 300, 5000, 60_000])); n = int(rng.choice([1, 20, 700, 8000, 20_000]))
        needs_normals = any(o[0] == "dynamic_points" for o in mods)
        icp = pkg.ICPSequence(minimizer=1, max_dist=2.0, max_iterations=3)
        pts = cloud(m, base["map"]) if m else np.zeros((0, 4), np.float32)
        nrm = np.zeros((m, 3), np.float32); sc = rng.uniform(0, 1, m).astype(np.float32)
        if m:
            if needs_normals:
                nrm = ob.surface_normals(pts, knn=min(8, m - 1), nthreads=8) if m > 8 else np.tile(np.float32([0, 0, 1]), (m, 1))
                icp.setMap(pts, nrm)
            else:
                icp.setMap(pts)
            icp.setMapScalar(sc)
        has_n = needs_normals and m > 0
        for step in range(int(rng.integers(1, 4))):
            scan = cloud(n, base["scan"] if rng.random() < 0.5 else base["map"])
            scan_s = rng.uniform(0, 1, n).astype(np.float32)
            pose = pkg.synth.make_T(tuple(rng.uniform(-0.5, 0.5, 3)), tuple(rng.uniform(-10, 10, 3))).astype(np.float32)
            to_sensor = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
            # DynamicPoints on a map without normals is an error in both worlds: skip those programs
            first = pts.shape[0] == 0
            dyn_at = [i for i, o in enumerate(mods) if o[0] == "dynamic_points"]
            if dyn_at and ((not first and not has_n) or (first and dyn_at[-1] > 0)):
                break
            try:
                src, mm = icp.mapUpdateChain(scan, mods, post, scan_scalar=scan_s, to_sensor=to_sensor)
            except pkg.InvalidParameter as e:
                rp, rn, rs, rsrc = host_chain(ob, pts, nrm, sc, scan, scan_s, to_sensor, mods, post)
                assert rp.shape[0] == 0 and "removed every point" in str(e), ("unexpected error", str(e), rp.shape)
                break
            rp, rn, rs, rsrc = host_chain(ob, pts, nrm, sc, scan, scan_s, to_sensor, mods, post)
            if mm == 0:
                # the chain removed every point: an empty resident map, like the reference's empty local cloud (r2)
                assert rp.shape[0] == 0 and icp.getMap().shape[0] == 0 and icp.getMapScalar().shape[0] == 0, ("empty map", rp.shape)
                break
            assert mm == rp.shape[0] and np.array_equal(src, rsrc), ("src", mm, rp.shape[0])
            got, got_n = icp.getMap(with_normals=True) if (has_n or any(o[0] == "surface_normals" for o in post)) else (icp.getMap(), None)
            assert np.array_equal(got, rp), "points"
            gs = icp.getMapScalar()
            assert np.array_equal(gs, rs) or (np.isnan(gs) == np.isnan(rs)).all() and np.array_equal(gs[~np.isnan(gs)], rs[~np.isnan(rs)]), "scalar"
            ran_normals = any(o[0] == "surface_normals" for o in post)
            if ran_normals:
                has_n = True
                if rp.shape[0] > 40:
                    dots = np.abs((got_n.astype(np.float64) * rn).sum(1))
                    assert (dots > 0.999).mean() > 0.99, ("normals", float((dots > 0.999).mean()))
                # carry the DEVICE normals forward so that both worlds keep identical inputs
                nrm = got_n
            else:
                if got_n is not None:
                    assert np.array_equal(got_n, rn), "normals copy"
                nrm = rn
            pts, sc = rp, rs
        icp.close()
    except AssertionError as e:
        fails += 1
        print("FAIL case", case, mods, post, e.args, flush=True)
print(f"{cases} cases, {fails} failures, {time.time() - t0:.1f} s")
