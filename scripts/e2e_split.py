import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import norlab_icp_mapper_amd as pkg
base = pkg.synth.make_scene(m=1_000_000, n=100_000)
scans = [pkg.synth.make_scene(m=8, n=100_000, seed_scan=500 + s)["scan"] for s in range(12)]
prior = np.eye(4, dtype=np.float32)
icp = pkg.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1)
icp.setMap(base["map"][::2], base["normals"][::2])
tr = tu = 0.0; dev = 0.0
for i, sc in enumerate(scans):
    t0 = time.perf_counter(); corr = icp.registerWithPrior(sc, prior); t1 = time.perf_counter()
    dev += icp.stats.loop_ms
    icp.mapUpdateStaged(corr, 0.15, normals_knn=10); t2 = time.perf_counter()
    if i >= 2: tr += t1 - t0; tu += t2 - t1
print(f"register {tr/10*1e3:.3f} ms (device loop {dev/12:.3f} ms)  update {tu/10*1e3:.3f} ms")
