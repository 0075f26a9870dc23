#!/usr/bin/env python
"""Aggregate ICP iterations/s of S independent scan streams sharing ONE GPU (one handle = one HIP stream per scan stream,
one host thread each; BASELINE config 5 runs 8 such streams over 8 GPUs).  A single registration is latency-bound (four
small kernels per iteration), so concurrent streams fill the idle part of the chip."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import norlab_icp_mapper_amd as pkg

chain = sys.argv[1] if len(sys.argv) > 1 else "p2p"
kw = dict(minimizer=1 if chain == "p2p" else 2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20)
STEPS = 20
for S in (1, 2, 4, 8):
    scenes = [pkg.synth.make_scene(m=1_000_000, n=100_000, seed_scan=43 + 1000 * s) for s in range(S)]
    d_map = torch.from_numpy(scenes[0]["map"]).cuda(); d_nrm = torch.from_numpy(scenes[0]["normals"]).cuda()
    hs, scans = [], []
    for s in range(S):
        icp = pkg.ICPSequence(**kw)
        assert icp.setMapDev(d_map.data_ptr(), d_map.shape[0], d_nrm.data_ptr())
        sc = torch.from_numpy(scenes[s]["scan"]).cuda()
        for _ in range(3):
            icp.registerDev(sc.data_ptr(), sc.shape[0], fixed_iterations=20)
        hs.append(icp); scans.append(sc)
    torch.cuda.synchronize()
    bar = threading.Barrier(S + 1)
    def work(i):
        bar.wait()
        for _ in range(STEPS):
            hs[i].registerDev(scans[i].data_ptr(), scans[i].shape[0], fixed_iterations=20)
        bar.wait()
    th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
    for t in th: t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    for t in th: t.join()
    print(f"{chain}: {S} streams: {S * STEPS * 20 / dt:9.0f} iterations/s aggregate ({dt / STEPS * 1e3:.2f} ms per registration round)")
    for h in hs: h.close()
