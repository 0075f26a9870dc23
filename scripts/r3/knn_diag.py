#!/usr/bin/env python
"""Per-pass counters of one nnk_wg_kernel launch (library built with -DICPMI_NNK_DIAG=<iteration>): queries that ran the pass,
candidates offered to their lists, mean level, list overflows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import norlab_icp_mapper_amd as pkg
sc = pkg.synth.make_scene(m=1_000_000, n=100_000)
icp = pkg.ICPSequence(minimizer=2, knn=6, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20, use_differential=0, use_graph=0)
icp.setMap(sc["map"], sc["normals"])
icp(sc["scan"])
d = np.array(icp.debugCounters(), dtype=np.float64)
for ps in range(6):
    if d[ps] == 0: continue
    print("pass %d%s: queries %d  offered/query %.1f  mean level %.2f  overflows %d" % (ps, "+" if ps == 5 else "", d[ps], d[6 + ps] / d[ps], d[12 + ps] / d[ps], d[18 + ps]))
