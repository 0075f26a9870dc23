#!/usr/bin/env python
"""Throughput of the documented chain (docs/MapperConfiguration.md:174-189: knn 6, point-to-plane, TrimmedDist 0.85),
fixed 20 iterations, 100k vs 1M, scan resident in HBM."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import norlab_icp_mapper_amd as pkg
knn = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sc = pkg.synth.make_scene(m=1_000_000, n=100_000)
icp = pkg.ICPSequence(minimizer=2, knn=knn, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20)
d_map = torch.from_numpy(sc["map"]).cuda(); d_nrm = torch.from_numpy(sc["normals"]).cuda(); d_scan = torch.from_numpy(sc["scan"]).cuda()
icp.setMapDev(d_map.data_ptr(), d_map.shape[0], d_nrm.data_ptr())
for _ in range(3): icp.registerDev(d_scan.data_ptr(), d_scan.shape[0], fixed_iterations=20)
torch.cuda.synchronize(); t0 = time.perf_counter()
R = 10
for _ in range(R): T = icp.registerDev(d_scan.data_ptr(), d_scan.shape[0], fixed_iterations=20)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"knn={knn}: {R * 20 / dt:.0f} it/s, {dt / R * 1e3:.2f} ms per registration; pose err vs gt {pkg.synth.pose_error(T, sc['T_gt'])}")
