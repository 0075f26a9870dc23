#!/usr/bin/env python
"""End-to-end scans/s of the mapping loop (reported separately from the ICP rate): every scan is registered
(point-to-plane, Trimmed 0.85, Differential checker) and merged (PointDistance 0.15 m, SurfaceNormal knn 10 post filter).
staged: icpmi_register_prior + icpmi_map_update_staged (one upload per scan); composed: the host-pointer entry points."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import norlab_icp_mapper_amd as pkg

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
S = int(sys.argv[3]) if len(sys.argv) > 3 else 12
base = pkg.synth.make_scene(m=m, n=n)
scans = [pkg.synth.make_scene(m=8, n=n, seed_scan=500 + s)["scan"] for s in range(S)]
prior = np.eye(4, dtype=np.float32)
kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
for mode in ("composed", "staged"):
    icp = pkg.ICPSequence(**kw)
    icp.setMap(base["map"][::2], base["normals"][::2])
    t0 = time.perf_counter(); its = 0; t_warm = None
    for k, sc in enumerate(scans):
        if k == 2: t_warm = time.perf_counter()   # (the first scans of the first handle of a process carry one-time costs: kernel loading, first captures, buffer growth)
        if mode == "staged":
            corr = icp.registerWithPrior(sc, prior)
            icp.mapUpdateStaged(corr, 0.15, normals_knn=10)
        else:
            in_map = icp.transform(prior, sc)
            corr = icp(in_map)
            icp.mapUpdatePointDistance(icp.transform(corr, in_map), 0.15, normals_knn=10)
        its += icp.stats.iterations
    dt = time.perf_counter() - t0
    steady = (time.perf_counter() - t_warm) / max(1, S - 2) if t_warm else dt / S
    print(f"{mode:9s}: {S / dt:7.1f} scans/s ({dt / S * 1e3:.2f} ms per scan, {its / S:.1f} ICP iterations per scan), map {icp.getMap().shape[0]} points; "
          f"scans 3..{S}: {1.0 / steady:7.1f} scans/s ({steady * 1e3:.2f} ms per scan)")
