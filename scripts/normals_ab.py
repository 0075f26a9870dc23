#!/usr/bin/env python
"""A/B of the self-kNN behind SurfaceNormalDataPointsFilter: tiled (LDS-staged neighbourhoods) vs one lane per query.
Run twice with ICPMI_SELF_KNN_TILED=0/1 and compare the saved normals (must be identical)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import norlab_icp_mapper_amd as pkg
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
out = sys.argv[2]
sc = pkg.synth.make_scene(m=m, n=1000)
icp = pkg.ICPSequence(minimizer=0, max_dist=2.0)
res = {}
for knn in (5, 10, 20):
    icp.surfaceNormals(sc["map"], knn=knn)
    t0 = time.perf_counter(); res[str(knn)] = icp.surfaceNormals(sc["map"], knn=knn); dt = time.perf_counter() - t0
    print(f"knn={knn}: {dt*1e3:.2f} ms")
# a sparse cloud with isolated points (redo paths)
rng = np.random.default_rng(0)
sp = np.ones((5000, 4), dtype=np.float32); sp[:, :3] = rng.uniform(-200, 200, (5000, 3)).astype(np.float32)
res["sparse"] = icp.surfaceNormals(np.concatenate([sc["map"][:20000], sp]), knn=10)
np.savez(out, **res)
