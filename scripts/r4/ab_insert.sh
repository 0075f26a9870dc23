#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for mn in 0 3000000; do
ICPMI_INSERT_MIN=$mn python - <<'PY'
import os, sys, json, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, bench
import norlab_icp_mapper_amd as pkg
sc = pkg.synth.make_scene()
d_map = torch.from_numpy(sc["map"]).cuda()
def barrier(): torch.cuda.synchronize()
scans = [torch.from_numpy(x).cuda() for x in bench.circle_scans(pkg, 6, 100000, 1.0, 0)]
for R in (1, 8):
    r = bench.config5_stream(np, torch, pkg, 0, d_map, None, scans, dict(bench.CHAINS["p2p"]), 0.15, None if R == 1 else ("loopback", R, 0.4), barrier)
    print("INSERT_MIN", os.environ["ICPMI_INSERT_MIN"], "R", R, "epoch ms", r["merge_epoch_ms"], "register", r["register_ms"]["median"])
PY
done
python scripts/e2e_bench.py 2>&1 | grep scans
ICPMI_INSERT_MIN=0 python scripts/e2e_bench.py 2>&1 | grep scans | sed 's/^/INSERT_MIN=0 /'
