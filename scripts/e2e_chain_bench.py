#!/usr/bin/env python
"""End-to-end scans/s of the mapping loop with the SHIPPED map-update chain (examples/config.yaml:26-50: DynamicPoints +
Octree 0.15 m modules, SurfaceNormal knn 10 + probabilityDynamic cut as post filters), point-to-plane registration in
front.  resident: icpmi_register_prior + icpmi_map_update_chain_staged (the map never leaves HBM, one scan upload);
composed: the same chain from the host-pointer operators (what host/Map.cpp's host path does per update)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import norlab_icp_mapper_amd as pkg

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 10
base = pkg.synth.make_scene(m=m, n=n)
scans = [pkg.synth.make_scene(m=8, n=n, seed_scan=500 + s)["scan"] for s in range(S)]
prior = np.eye(4, dtype=np.float32)
DYN = (0.9, 0.8, 0.99, 0.01, 0.01, 0.01, 200.0)
modules = [("dynamic_points",) + DYN, ("voxel", 0.15, 1)]
post = [("surface_normals", 10), ("cut_scalar", 0.65, 1)]
kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
map0, nrm0 = base["map"][::2], base["normals"][::2]
prob0 = np.full(map0.shape[0], 0.6, np.float32)
for mode in ("composed", "resident"):
    icp = pkg.ICPSequence(**kw)
    icp.setMap(map0, nrm0)
    if mode == "resident":
        icp.setMapScalar(prob0)
    pts, nrm, prob = map0, nrm0, prob0
    t0 = time.perf_counter(); its = 0
    for sc in scans:
        s_prob = np.full(sc.shape[0], 0.6, np.float32)
        if mode == "resident":
            corr = icp.registerWithPrior(sc, prior)
            pose = (corr @ prior).astype(np.float32)
            src, msize, head = icp.mapUpdateChain(None, modules, post, scan_scalar=s_prob, to_sensor=np.linalg.inv(pose), staged_correction=corr,
                                                  with_prefix=True)
        else:
            in_map = icp.transform(prior, sc)
            corr = icp(in_map)
            pose = (corr @ prior).astype(np.float32)
            moved = icp.transform(corr, in_map)
            prob = icp.dynamicPointsUpdate(np.linalg.inv(pose), moved, pts, nrm, prob, *DYN)
            pts = np.concatenate([pts, moved]); prob = np.concatenate([prob, s_prob])
            keep = icp.voxelKeep(pts, 0.15, 1)
            pts, prob = pts[keep], prob[keep]
            nrm = icp.surfaceNormals(pts, knn=10)
            keep = ~(prob > 0.65)
            pts, nrm, prob = pts[keep], nrm[keep], prob[keep]
            icp.setMap(pts, nrm)
            msize = pts.shape[0]
        its += icp.stats.iterations
    dt = time.perf_counter() - t0
    print(f"{mode:9s}: {S / dt:7.1f} scans/s ({dt / S * 1e3:.2f} ms per scan, {its / S:.1f} ICP iterations per scan), map {msize} points")
