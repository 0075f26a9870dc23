#!/usr/bin/env python
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import norlab_icp_mapper_amd as pkg
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = pkg.synth.make_scene(m=m, n=1000)
icp = pkg.ICPSequence(minimizer=2, max_dist=2.0)
d_map = torch.from_numpy(sc["map"]).cuda(); d_nrm = torch.from_numpy(sc["normals"]).cuda()
for _ in range(3): icp.setMapDev(d_map.data_ptr(), d_map.shape[0], d_nrm.data_ptr())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): icp.setMapDev(d_map.data_ptr(), d_map.shape[0], d_nrm.data_ptr())
torch.cuda.synchronize()
print(f"set_map (device pointers) M={m}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms", icp.gridInfo())
