#!/usr/bin/env python
"""PM::ICPSequence::setDefault()'s matcher has maxDist = inf: the chain may need the brute pass.  r5 keeps its loop state in query order
(ICPMI_HARD_SORTED=1, default) instead of dropping to the caller's order (0).  Run once per setting; prints ms per registration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import norlab_icp_mapper_amd as pkg
sc = pkg.synth.make_scene()
dm, dn, ds = (torch.from_numpy(sc[k]).cuda() for k in ("map", "normals", "scan"))
for name, kw in (("p2plane maxDist inf checked", dict(minimizer=2, max_dist=float("inf"), outliers=[(4, 0.85)], max_iterations=40, use_differential=1)),
                 ("p2p maxDist inf fixed 20", dict(minimizer=1, max_dist=float("inf"), outliers=[(4, 0.85)], max_iterations=20))):
    icp = pkg.ICPSequence(**kw)
    icp.setMapDev(dm.data_ptr(), dm.shape[0], dn.data_ptr())
    fixed = 20 if "fixed" in name else 0
    reg = (lambda: icp.registerDev(ds.data_ptr(), ds.shape[0], fixed_iterations=fixed)) if fixed else (lambda: icp.registerDev(ds.data_ptr(), ds.shape[0]))
    for _ in range(5):
        T = reg()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); R = 50; its = 0
    for _ in range(R):
        T = reg(); its += icp.stats.iterations
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print(f"ICPMI_HARD_SORTED={os.environ.get('ICPMI_HARD_SORTED', '1')} {name}: {dt * 1e3:.3f} ms per registration, {its / R:.1f} iterations ({dt * 1e6 * R / its:.1f} us per iteration), "
          f"T checksum {float(np.abs(T).sum()):.9f}")
