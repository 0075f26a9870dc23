#!/usr/bin/env python
"""BASELINE config 4 alone (bench.py's config4_replay without the rest of the driver line): scans/s, ms per scan."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import norlab_icp_mapper_amd as pkg
import bench
r = bench.config4_replay(np, pkg, False)
print(json.dumps({k: r[k] for k in ("value", "ms_per_scan", "icp_iterations", "map_points_final") if k in r}))
