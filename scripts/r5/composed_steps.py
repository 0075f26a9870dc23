#!/usr/bin/env python
"""the composed (host-pointer) mapping loop of scripts/e2e_bench.py, call by call: where a scan's 3 ms go"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import norlab_icp_mapper_amd as pkg
base = pkg.synth.make_scene(m=1_000_000, n=100_000)
scans = [pkg.synth.make_scene(m=8, n=100_000, seed_scan=500 + s)["scan"] for s in range(12)]
prior = np.eye(4, dtype=np.float32)
icp = pkg.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
icp.setMap(base["map"][::2], base["normals"][::2])
acc = np.zeros(4)
for k, sc in enumerate(scans):
    t0 = time.perf_counter(); in_map = icp.transform(prior, sc)
    t1 = time.perf_counter(); corr = icp(in_map)
    t2 = time.perf_counter(); moved = icp.transform(corr, in_map)
    t3 = time.perf_counter(); icp.mapUpdatePointDistance(moved, 0.15, normals_knn=10)
    t4 = time.perf_counter()
    if k >= 2: acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3]
print("transform %.3f  register %.3f  transform %.3f  mapUpdatePointDistance %.3f ms  (sum %.3f)" % (*(acc / 10 * 1e3), acc.sum() / 10 * 1e3))
