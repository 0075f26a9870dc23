#!/usr/bin/env python
"""Device-memory leak check: handles created, used (map, registration, map-side operators, resident update) and destroyed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import norlab_icp_mapper_amd as pkg
sc = pkg.synth.make_scene(m=200_000, n=20_000)
def free_mb():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return f / 2**20
base = None
for rep in range(6):
    for i in range(40):
        icp = pkg.ICPSequence(minimizer=2, knn=1 if i % 2 else 6, max_dist=2.0, outliers=[(4, 0.85)], max_iterations=5)
        icp.setMap(sc["map"], sc["normals"]); icp(sc["scan"])
        if i % 4 == 0:
            icp.surfaceNormals(sc["map"][:50_000], knn=10); icp.pointDistanceKeep(sc["map"], sc["scan"], 0.1); icp.voxelKeepFirst(sc["map"], 0.2)
            icp.mapUpdatePointDistance(sc["scan"], 0.1, normals_knn=5); icp.registerWithPrior(sc["scan"], np.eye(4)); icp.mapUpdateStaged(np.eye(4), 0.1)
        if i % 8 == 0:   # the resident chain (scratch arena, ping-pong set, scalar channel) and the fused input filters
            prob = np.full(sc["scan"].shape[0], 0.6, np.float32)
            icp.setMapScalar(np.full(icp.getMap().shape[0], 0.6, np.float32))
            icp.mapUpdateChain(sc["scan"], [("dynamic_points", 0.9, 0.8, 0.99, 0.01, 0.01, 0.01, 200.0), ("voxel", 0.15, 1)],
                               [("surface_normals", 10), ("cut_scalar", 0.65, 1)], scan_scalar=prob, to_sensor=np.eye(4))
            icp.filterPoints(sc["scan"], [("distance_limit", -1, 40.0, False), ("bounding_box", (-1, -1, -1), (1, 1, 1), True)])
            # r2: octree operator, batch, the library's RCCL epoch (1-rank communicator), the raw-frame index, VarTrimmed / Robust scratch
            icp.octreeSample(sc["map"][:80_000], 0.2, 4)
            icp.mapUpdateChain(sc["scan"], [("octree", 0.15, 0, 1)], [("surface_normals", 10)], scan_scalar=prob, to_sensor=np.eye(4), from_sensor=np.eye(4))
            d = [torch.from_numpy(sc["scan"]).cuda(), torch.from_numpy(sc["scan"][::2].copy()).cuda()]
            icp.registerBatchDev([t.data_ptr() for t in d], [t.shape[0] for t in d])
            icp.commInit(icp.commUniqueId(), 1, 0)
            corr = icp.registerWithPrior(sc["scan"], np.eye(4)); icp.stagedMergeAllGather(corr, 0.1, normals_knn=5); icp.commDestroy()
            e = pkg.ICPSequence(minimizer=2, max_dist=2.0, outliers=[(8, 0.05, 0, 0.99, 0.95), (7, 1.0, 0 | (1 << 4), 0.0)], max_iterations=4)
            e.setMap(sc["map"], sc["normals"]); e(sc["scan"]); e.close()
        icp.close()
    f = free_mb()
    if base is None: base = f
    print(f"round {rep}: free {f:.0f} MiB (delta vs first round {f - base:+.1f})")
assert abs(free_mb() - base) < 64, "device memory drifts"
print("leak check ok")
