#!/usr/bin/env python
"""Registration time against the matcher's knn (point-to-plane, 20 iterations, 100 k x 1 M): where the one-lane kernels take over."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import norlab_icp_mapper_amd as pkg
sc = pkg.synth.make_scene(m=1_000_000, n=100_000)
for k in [int(a) for a in sys.argv[1:]] or [1, 6, 8, 10, 16]:
    icp = pkg.ICPSequence(minimizer=2, knn=k, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20, use_differential=0)
    icp.setMap(sc["map"], sc["normals"])
    icp(sc["scan"])
    t0 = time.perf_counter()
    for _ in range(3):
        T = icp(sc["scan"])
    dt = (time.perf_counter() - t0) / 3
    dtr, drr = pkg.synth.pose_error(T, sc["T_gt"])
    print("knn %2d: %.2f ms per registration (host scan upload included), loop %.3f ms, nn avg %.1f us, err %.2e m" % (k, dt * 1e3, icp.stats.loop_ms, icp.stats.nn_ms_avg * 1e3, dtr))
