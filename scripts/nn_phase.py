#!/usr/bin/env python
"""Per-phase cycle counts of nn1_ml_kernel (library built with -DICPMI_NN_TIMING): serialised
(s_waitcnt 0 before every tick) wall time of each phase, averaged per wave."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import norlab_icp_mapper_amd as pkg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sc = pkg.synth.make_scene(m=1_000_000, n=n)
icp = pkg.ICPSequence(minimizer=int(sys.argv[2]) if len(sys.argv) > 2 else 1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20, use_differential=0, use_graph=0)
icp.setMap(sc["map"], sc["normals"])
for _ in range(2):
    icp(sc["scan"])
d = np.array(icp.debugCounters(), dtype=np.float64)
names = ["query+T", "seed", "rows", "scan", "fold", "tail", "levels", "waves"]
for base, tag in ((0, "first"), (8, "seeded")):
    w = d[base + 7]
    if w == 0: continue
    print(tag, "candidates per query", d[16 + (1 if base else 0)] / w)
    print(tag, "waves", int(w), {nm: round(d[base + i] / w, 1) for i, nm in enumerate(names[:7])}, "sum", round(d[base:base + 6].sum() / w))
print("seeded candidates/query histogram (<=16, <=32, <=64, <=128, <=256, >256):", [int(x) for x in d[18:24]])
