#!/usr/bin/env python
"""The production registration (Counter 40 + Differential: what Mapper::processInput runs) on a resident scan: ms per registration
and iterations, for the benchmark chains.  ICPMI_SEG=0: eager run-ahead loop (r2); default: segment graphs (r3)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import norlab_icp_mapper_amd as pkg
sc = pkg.synth.make_scene()
dm, dn, ds = (torch.from_numpy(sc[k]).cuda() for k in ("map", "normals", "scan"))
for name, kw in (("p2p", dict(minimizer=1)), ("p2plane", dict(minimizer=2))):
    icp = pkg.ICPSequence(max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=1, **kw)
    icp.setMapDev(dm.data_ptr(), dm.shape[0], dn.data_ptr())
    for _ in range(5):
        T = icp.registerDev(ds.data_ptr(), ds.shape[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); R = 50
    for _ in range(R):
        T = icp.registerDev(ds.data_ptr(), ds.shape[0])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    gt = pkg.synth.pose_error(T, sc["T_gt"])
    print(f"{name}: {dt * 1e3:.3f} ms per checked registration, {icp.stats.iterations} iterations ({dt * 1e6 / icp.stats.iterations:.1f} us per iteration), "
          f"device loop {icp.stats.loop_ms:.3f} ms, err vs ground truth {gt[0]:.2e} m, T checksum {float(np.abs(T).sum()):.9f}")
