#!/usr/bin/env python
"""NN kernel micro-benchmark: identity minimiser keeps T_iter = I so every iteration repeats the same
NN problem.  usage: nn_bench.py [aligned|misaligned|p2plane] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import norlab_icp_mapper_amd as pkg
mode = sys.argv[1] if len(sys.argv) > 1 else "aligned"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sc = pkg.synth.make_scene()
scan = sc["scan"]
if mode == "aligned":
    T = sc["T_gt"].astype(np.float32)
    scan = scan.copy(); scan[:, :3] = scan[:, :3] @ T[:3, :3].T + T[:3, 3]
minimizer = 2 if mode == "p2plane" else 0
kw = dict(minimizer=minimizer, max_dist=2.0, outliers=[] if minimizer == 0 else [(4, 0.85)], max_iterations=20, profile=1)
for k, v in (a.split("=") for a in sys.argv[3:]):
    kw[k] = float(v) if "." in v else int(v)
icp = pkg.ICPSequence(**kw)
dm, dn, ds = (torch.from_numpy(sc[k] if k != "scan" else scan).cuda() for k in ("map", "normals", "scan"))
icp.setMapDev(dm.data_ptr(), dm.shape[0], dn.data_ptr())
print("grid", icp.gridInfo())
for r in range(reps):
    icp.registerDev(ds.data_ptr(), ds.shape[0], fixed_iterations=20)
    dbg = icp.debugCounters()
    print("   phase cycles per ring pass:", [round(x / max(dbg[0], 1)) for x in dbg[8:15]], "passes", dbg[0], "staged", dbg[1], "to_global", dbg[2], "solve serial cycles", round(dbg[20] / max(dbg[21], 1)))
    print(f"{mode}: nn avg {icp.stats.nn_ms_avg*1e3:.1f} us over {icp.stats.nn_launches} launches, loop {icp.stats.loop_ms:.3f} ms, pairs {icp.stats.pairs}, "
          f"dbg ring_passes/staged_pts/items_to_global = {[ (icp.stats.reserved[2*i] & 0xffffffff) | (icp.stats.reserved[2*i+1] << 32) for i in range(3)]}")
