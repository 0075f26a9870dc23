#!/usr/bin/env python
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import norlab_icp_mapper_amd as pkg
sc = pkg.synth.make_scene(m=200_000, n=20_000)
for mn in (1, 2):
    icp = pkg.ICPSequence(minimizer=mn, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20, use_differential=0)
    icp.setMap(sc["map"], sc["normals"]); icp(sc["scan"])
    d = icp.debugCounters()
    print("minimizer", mn, "serial solve cycles per iteration:", d[20] / max(d[21], 1), "pre", d[15] / max(d[21], 1), "solve6", d[16] / max(d[21], 1), "angle_axis", d[17] / max(d[21], 1), "mat4_mul", d[18] / max(d[21], 1))
