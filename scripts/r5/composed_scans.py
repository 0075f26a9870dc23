#!/usr/bin/env python
"""the composed loop of scripts/e2e_bench.py scan by scan (ms), handle creation and first setMap included"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import norlab_icp_mapper_amd as pkg
base = pkg.synth.make_scene(m=1_000_000, n=100_000)
scans = [pkg.synth.make_scene(m=8, n=100_000, seed_scan=500 + s)["scan"] for s in range(12)]
prior = np.eye(4, dtype=np.float32)
for rep in range(2):
    t0 = time.perf_counter()
    icp = pkg.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
    icp.setMap(base["map"][::2], base["normals"][::2])
    ts = [(time.perf_counter() - t0) * 1e3]
    for sc in scans:
        t0 = time.perf_counter()
        in_map = icp.transform(prior, sc); corr = icp(in_map); icp.mapUpdatePointDistance(icp.transform(corr, in_map), 0.15, normals_knn=10)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("create+setMap %.1f | scans " % ts[0] + " ".join("%.2f" % t for t in ts[1:]) + " | sum %.1f" % sum(ts[1:]))
