"""Where the single-lane part of solve_kernel spends its time (clock64 stamps in IcpState::dbg[20..23])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import norlab_icp_mapper_amd as pkg

sc = pkg.synth.make_scene(m=1000000, n=100000)
for name, kw in (("p2p", dict(minimizer=1)), ("p2plane", dict(minimizer=2))):
    for diff in (0, 1):
        icp = pkg.ICPSequence(max_dist=2.0, outliers=[(4, 0.85)], max_iterations=40, use_differential=diff, use_graph=0, **kw)
        icp.setMap(sc["map"], sc["normals"])
        d = torch.from_numpy(sc["scan"]).cuda()
        for _ in range(2):
            icp.registerDev(d.data_ptr(), d.shape[0], fixed_iterations=0 if diff else 20)
        c = icp.debugCounters()
        it = max(c[21], 1)
        print(f"{name} differential={diff}: iterations {c[21]}  serial {c[20]/it:.0f} clk  minimiser {c[22]/it:.0f}  compose {(c[20]-c[22]-c[23])/it:.0f}  checkers {c[23]/it:.0f}")
