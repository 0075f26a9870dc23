#!/usr/bin/env python
"""Wall time of the two calls of the resident shipped chain: registerWithPrior vs mapUpdateChain (staged)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import norlab_icp_mapper_amd as pkg
base = pkg.synth.make_scene(m=1_000_000, n=100_000)
scans = [pkg.synth.make_scene(m=8, n=100_000, seed_scan=500 + s)["scan"] for s in range(12)]
prior = np.eye(4, dtype=np.float32)
DYN = (0.9, 0.8, 0.99, 0.01, 0.01, 0.01, 200.0)
modules = [("dynamic_points",) + DYN, ("voxel", 0.15, 1)]
post = [("surface_normals", 10), ("cut_scalar", 0.65, 1)]
icp = pkg.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
icp.setMap(base["map"][::2], base["normals"][::2]); icp.setMapScalar(np.full(500_000, 0.6, np.float32))
tr = tu = tp = 0.0
for i, sc in enumerate(scans):
    s_prob = np.full(sc.shape[0], 0.6, np.float32)
    t0 = time.perf_counter(); corr = icp.registerWithPrior(sc, prior); t1 = time.perf_counter()
    inv = np.linalg.inv((corr @ prior).astype(np.float32)); t2 = time.perf_counter()
    src, m = icp.mapUpdateChain(None, modules, post, scan_scalar=s_prob, to_sensor=inv, staged_correction=corr, want_src=(os.environ.get('SRC','1')=='1')); t3 = time.perf_counter()
    if i >= 2: tr += t1 - t0; tp += t2 - t1; tu += t3 - t2
print(f"register {tr/10*1e3:.3f} ms  numpy {tp/10*1e3:.3f} ms  chain update {tu/10*1e3:.3f} ms (map {m})")
