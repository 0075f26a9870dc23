#!/usr/bin/env python
"""Per-phase cycle counts of nn1_wq_kernel (library built with -DICPMI_NN_TIMING, see scripts/r3/wq_phase.sh): serialised wall
time of each phase per wave (s_waitcnt 0 before every tick), averaged over a sample of waves."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import norlab_icp_mapper_amd as pkg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sc = pkg.synth.make_scene(m=1_000_000, n=n)
icp = pkg.ICPSequence(minimizer=int(sys.argv[2]) if len(sys.argv) > 2 else 1, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20, use_differential=0, use_graph=0)
icp.setMap(sc["map"], sc["normals"])
for _ in range(2):
    icp(sc["scan"])
d = np.array(icp.debugCounters(), dtype=np.float64)
names = ["load+seed", "rows", "alloc+emit", "pieces", "decide", "tail"]
for base, tag in ((0, "first two"), (8, "steady")):
    w = d[base + 7]
    if w == 0: continue
    print(tag, "waves sampled", int(w), "passes/wave", round(d[base + 6] / w, 2), "pieces/wave", round(d[16 + (1 if base else 0)] / w, 1),
          {nm: round(d[base + i] / w) for i, nm in enumerate(names)}, "sum", round(d[base:base + 6].sum() / w))
if d[23] > 0:
    print("shader clock during the NN kernel: %.0f MHz (clock64 / wall_clock64 @ 100 MHz)" % (d[22] / d[23] * 100.0))
print("heaviest steady workgroup: life %d cycles, %d pieces" % (d[20], d[21]))
