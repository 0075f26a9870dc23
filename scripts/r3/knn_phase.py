#!/usr/bin/env python
"""Per-phase cycle counts of nnk_wg_kernel (timing build, see scripts/r3/wq_phase.sh) on the knn-6 point-to-plane chain."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import norlab_icp_mapper_amd as pkg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sc = pkg.synth.make_scene(m=1_000_000, n=n)
icp = pkg.ICPSequence(minimizer=2, knn=6, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20, use_differential=0, use_graph=0)
icp.setMap(sc["map"], sc["normals"])
for _ in range(2):
    icp(sc["scan"])
d = np.array(icp.debugCounters(), dtype=np.float64)
names = ["load+seed", "1a", "rows+emit", "pieces", "merge+decide", "tail"]
for base, tag in ((0, "first two"), (8, "steady")):
    w = d[base + 7]
    if w == 0: continue
    print(tag, "wgs sampled", int(w), "passes/wg", round(d[base + 6] / w, 2), "pieces/wg", round(d[16 + (1 if base else 0)] / w, 1),
          {nm: round(d[base + i] / w) for i, nm in enumerate(names)}, "sum", round(d[base:base + 6].sum() / w))
print("steady: offered candidates per query-pass-lane sum/wg", round(d[19] / max(d[15], 1), 1), "overflows/wg", round(d[18] / max(d[15], 1), 3))
print("heaviest steady workgroup: life %d cycles, %d pieces" % (d[20], d[21]))
