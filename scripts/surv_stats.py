import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import norlab_icp_mapper_amd as pkg
sc = pkg.synth.make_scene(m=1_000_000, n=100_000)
for mn in (2, 1):
    icp = pkg.ICPSequence(minimizer=mn, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20, use_differential=0)
    icp.setMap(sc["map"], sc["normals"]); icp(sc["scan"])
    d = icp.debugCounters()
    print("minimizer", mn, "iterations >= 10: queries survived", d[22], "of", d[23], "=", d[22] / max(d[23], 1), "waves fully survived", d[19], "of", d[18], "=", d[19] / max(d[18], 1))
