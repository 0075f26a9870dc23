#!/usr/bin/env python
"""One GPU, R simulated ranks (ICPMI_COMM_LOOPBACK): the map-growth epoch with ONE collective (ICPMI_MERGE_BLOCK=32768, default) against the
three-collective epoch of r4 (ICPMI_MERGE_BLOCK=0), same process, alternating.   python scripts/r5/ab_epoch.py [scans]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
import norlab_icp_mapper_amd as pkg
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sc = pkg.synth.make_scene()
d_map = torch.from_numpy(sc["map"]).cuda()
scans = [torch.from_numpy(x).cuda() for x in bench.circle_scans(pkg, S, 100000, 1.0, 0)]
for rep in range(2):
    for R in (1, 2, 4, 8):
        for blk in ("32768", "0"):
            os.environ["ICPMI_MERGE_BLOCK"] = blk
            r = bench.config5_stream(np, torch, pkg, 0, d_map, None, scans, dict(bench.CHAINS["p2p"]), 0.15, None if R == 1 else ("loopback", R, 0.4), torch.cuda.synchronize)
            e = r["merge_epoch_ms"]
            print(f"R {R} block {blk:>5}: epoch median {e['median']:.3f} min {e['min']:.3f} ms  one/three-collective epochs {r['epochs_one_collective']}/{r['epochs_three_collectives']}  appended {r['appended_per_epoch_all_ranks'][:2]}")
