#!/usr/bin/env python
"""One GPU, R simulated ranks (ICPMI_COMM_LOOPBACK): the map-growth epoch of the scan-sharded mapper on the 1 M-point map.
    python scripts/r4/loopback_bench.py R [scans]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
import norlab_icp_mapper_amd as pkg
R = int(sys.argv[1]); S = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sc = pkg.synth.make_scene()
d_map = torch.from_numpy(sc["map"]).cuda()
scans = [torch.from_numpy(x).cuda() for x in bench.circle_scans(pkg, S, 100000, 1.0, 0)]
r = bench.config5_stream(np, torch, pkg, 0, d_map, None, scans, dict(bench.CHAINS["p2p"]), 0.15, None if R == 1 else ("loopback", R, 0.4), torch.cuda.synchronize)
print("R", R, "epoch ms", r["merge_epoch_ms"], "appended", r["appended_per_epoch_all_ranks"][:3])
