#!/usr/bin/env python
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import norlab_icp_mapper_amd as pkg
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = pkg.synth.make_scene(m=m, n=1000)
icp = pkg.ICPSequence(minimizer=0)
icp.surfaceNormals(sc["map"], knn=10)
t0 = time.perf_counter(); nr = icp.surfaceNormals(sc["map"], knn=10); dt = time.perf_counter() - t0
dots = np.abs((nr * sc["normals"]).sum(1))
print(f"surfaceNormals M={m} knn=10: {dt*1e3:.2f} ms; |n.n_true| median {np.median(dots):.4f}")
