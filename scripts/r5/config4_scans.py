#!/usr/bin/env python
"""per-scan clocks of the BASELINE config 4 replay (the harness's `timing:` lines of the last pass)"""
import sys, os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import config4_data as c4
exe = os.path.join(ROOT, "norlab_icp_mapper_amd", "build_map_from_scans_and_trajectory")
z = np.load(os.path.join(ROOT, "tests", "golden", "bundled_scans_all.npz"))
with tempfile.TemporaryDirectory() as tmp:
    names, traj = c4.write_bundled_dataset(tmp, z)
    cfg = os.path.join(tmp, "config.yaml"); open(cfg, "w").write(c4.CONFIG4_YAML)
    run = subprocess.run([exe, tmp, cfg], capture_output=True, text=True, timeout=600, env=dict(os.environ, NIM_TIMING="3"))
sys.stderr.write("\n".join(run.stderr.splitlines()[-40:]) + "\n")
lines = [l for l in run.stdout.splitlines() if l.startswith("timing: pass 2") or l.startswith("replay:")]
print("\n".join(lines))
