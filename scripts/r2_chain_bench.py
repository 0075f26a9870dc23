#!/usr/bin/env python
"""Resident map-update chain, per-call wall time: registration (icpmi_register_prior) and update (icpmi_map_update_chain_staged)
separately, for the decimation operator (voxel lattice | octree) and with / without the sensor-frame round trip.
    python scripts/r2_chain_bench.py [map points] [scan points] [scans]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import norlab_icp_mapper_amd as pkg

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 12
only = sys.argv[4] if len(sys.argv) > 4 else None
base = pkg.synth.make_scene(m=m, n=n)
scans = [pkg.synth.make_scene(m=8, n=n, seed_scan=500 + s)["scan"] for s in range(S)]
prior = np.eye(4, dtype=np.float32)
DYN = (0.9, 0.8, 0.99, 0.01, 0.01, 0.01, 200.0)
post = [("surface_normals", 10), ("cut_scalar", 0.65, 1)]
kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1, use_graph=int(os.environ.get("CHAIN_USE_GRAPH", "1")))
map0, nrm0 = base["map"][::2], base["normals"][::2]
prob0 = np.full(map0.shape[0], 0.6, np.float32)
for name, dec, trip in (("voxel, map frame", ("voxel", 0.15, 1), False), ("octree, map frame", ("octree", 0.15, 1, 1), False),
                        ("octree, sensor-frame round trip", ("octree", 0.15, 1, 1), True), ("point_distance + normals, round trip", None, True)):
    if only and only not in name:
        continue
    icp = pkg.ICPSequence(**kw)
    icp.setMap(map0, nrm0)
    icp.setMapScalar(prob0)
    modules = [("dynamic_points",) + DYN, dec] if dec else [("point_distance", 0.15)]
    pst = post if dec else [("surface_normals", 10)]
    t_reg = t_upd = 0.0
    for k, sc in enumerate(scans):
        s_prob = np.full(sc.shape[0], 0.6, np.float32)
        t0 = time.perf_counter()
        corr = icp.registerWithPrior(sc, prior)
        t1 = time.perf_counter()
        pose = (corr @ prior).astype(np.float32)
        _, msize = icp.mapUpdateChain(None, modules, pst, scan_scalar=s_prob if dec else None, to_sensor=np.linalg.inv(pose), from_sensor=pose if trip else None,
                                      staged_correction=corr, want_src=False)
        t2 = time.perf_counter()
        if k >= 2:
            t_reg += t1 - t0; t_upd += t2 - t1
    print(f"{name:38s}: register {t_reg / (S - 2) * 1e3:6.2f} ms, update {t_upd / (S - 2) * 1e3:6.2f} ms per scan, map {msize} points")
