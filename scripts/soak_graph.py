#!/usr/bin/env python
"""Soak test of the cached-graph path: one long-lived handle, maps and scans of changing sizes, config swaps;
every fixed-iteration registration is compared bit for bit with an eager handle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import norlab_icp_mapper_amd as pkg

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
base = pkg.synth.make_scene(m=300_000, n=50_000)
d_map_all = torch.from_numpy(base["map"]).cuda(); d_nrm_all = torch.from_numpy(base["normals"]).cuda(); d_scan_all = torch.from_numpy(base["scan"]).cuda()
g = pkg.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], use_graph=1)
e = pkg.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], use_graph=0)
fails = 0; t0 = time.time(); m = 0
for case in range(cases):
    r = rng.random()
    if m == 0 or r < 0.15:
        m = int(rng.choice([500, 20_000, 300_000])); off = int(rng.integers(0, 300_000 - m + 1))
        for h in (g, e): h.setMapDev(d_map_all[off:off + m].data_ptr(), m, d_nrm_all[off:off + m].contiguous().data_ptr())
        cur_n = d_nrm_all[off:off + m].contiguous()  # keep alive
    n = int(rng.choice([300, 5_000, 50_000])); so = int(rng.integers(0, 50_000 - n + 1)); iters = int(rng.choice([1, 3, 8, 20]))
    scan = d_scan_all[so:so + n]
    reps = int(rng.choice([1, 1, 3]))
    for _ in range(reps):
        Tg = g.registerDev(scan.data_ptr(), n, fixed_iterations=iters)
        Te = e.registerDev(scan.data_ptr(), n, fixed_iterations=iters)
        if not np.array_equal(Tg.view(np.uint32), Te.view(np.uint32)) or not np.isfinite(Tg).all():
            fails += 1; print("CASE", case, "graph != eager", m, n, iters); break
print(f"soak_graph: {cases} cases, {fails} failures, {time.time() - t0:.1f} s")
sys.exit(1 if fails else 0)
