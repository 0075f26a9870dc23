#!/usr/bin/env python
"""Wall time of the map-side operators at the BASELINE sizes (host-pointer entry points: the times
include the PCIe copies of the call)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import norlab_icp_mapper_amd as pkg

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
sc = pkg.synth.make_scene(m=m, n=n)
icp = pkg.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=20)

def timeit(name, f, reps=3):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); t.append(time.perf_counter() - t0)
    print(f"{name:44s} {min(t) * 1e3:9.2f} ms")
    return r

timeit(f"setMap M={m}", lambda: icp.setMap(sc["map"], sc["normals"]))
timeit(f"surfaceNormals M={m} knn=10", lambda: icp.surfaceNormals(sc["map"], knn=10), reps=2)
timeit(f"pointDistanceKeep N={n} vs M={m}", lambda: icp.pointDistanceKeep(sc["map"], sc["scan"], 0.15))
timeit(f"transform M={m}", lambda: icp.transform(sc["T_gt"], sc["map"], sc["normals"]))
mean = np.array(icp.getMapMean(), dtype=np.float32)
q = sc["scan"].copy(); q[:, :3] -= mean[:3]
timeit(f"knn k=1 N={n} (stage call)", lambda: icp.knn(q, k=1, max_dist=2.0))
timeit(f"knn k=6 N={n} (stage call)", lambda: icp.knn(q, k=6, max_dist=2.0))
timeit(f"binCells M={m}", lambda: icp.binCells(sc["map"]))
timeit(f"register N={n} (20 it, p2plane)", lambda: icp(sc["scan"]))
# resident map update: only the scan crosses PCIe
base = sc["map"][::2]; icp2 = pkg.ICPSequence(minimizer=2, max_dist=2.0)
for knn in (0, 10):
    def upd():
        icp2.setMap(base, sc["normals"][::2])
        t0 = time.perf_counter(); icp2.mapUpdatePointDistance(sc["scan"], 0.15, normals_knn=knn); return time.perf_counter() - t0
    upd(); ts = [upd() for _ in range(3)]
    print(f"{'mapUpdatePointDistance N=%d M=%d normals_knn=%d' % (n, base.shape[0], knn):44s} {min(ts) * 1e3:9.2f} ms")
