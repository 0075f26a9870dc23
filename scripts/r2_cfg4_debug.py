#!/usr/bin/env python
"""Where does the config-4 replay leave the oracle replay?  Per scan: pose difference and map set difference of the resident
GPU replay against the oracle replay with the reference's sensor-frame round trip and without it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import norlab_icp_mapper_amd as amd
import oracle_mapper as om
from test_gpu_configs import _quat_T

z = np.load(os.path.join(ROOT, "tests", "golden", "bundled_scans_all.npz"))
traj = z["trajectory"]
DYN = dict(threshold_dynamic=0.9, alpha=0.8, beta=0.99, beam_half_angle=0.01, epsilon_a=0.01, epsilon_d=0.01)
ICP = dict(knn=6, max_dist=2.0, minimizer=2, outliers=[], max_iterations=10)


def make(frame):
    return om.OracleMapper(ICP, [("dynamic_points", DYN), ("octree", 0.15, 0)], post=[("surface_normals", 10), ("cut", "probabilityDynamic", 1, 0.65)],
                           update=("delay", 0.05), sensor_max_range=200.0,
                           input_filters=[("bounding_box", (-1.5, -1, -1), (0.5, 1, 0.5), 1), ("bounding_box", (-6, -2.5, -1), (-1.5, 2.5, 1), 1)],
                           add_descriptors=[("probabilityDynamic", 0.6)], nthreads=16, post_in_map_frame=frame)


ref, alt = make(False), make(True)
icp = amd.ICPSequence(**ICP)
dyn7 = (0.9, 0.8, 0.99, 0.01, 0.01, 0.01, 200.0)
gpu_map = None
for i in range(14):
    prior = _quat_T(traj[i, 2:]); stamp = traj[i, 0] + traj[i, 1] * 1e-9
    c = ref.apply_input_filters(z[f"scan{i}_xyz"])
    Tr = ref.process_input(c, prior, stamp)
    Ta = alt.process_input(c, prior, stamp)
    # resident GPU replay through the C ABI
    inp = icp.transform(prior, c["xyz1"])
    if not icp.hasMap():
        corr = np.eye(4, dtype=np.float32); Tg = prior
    else:
        corr = icp(inp); Tg = om.mat4_mul_f32(corr, prior)
    moved = icp.transform(corr, inp)
    to_sensor = np.linalg.inv(Tg.astype(np.float64)).astype(np.float32)
    icp.mapUpdateChain(moved, [("dynamic_points",) + dyn7, ("octree", 0.15, 0, 1)], [("surface_normals", 10), ("cut_scalar", 0.65, 1)],
                       scan_scalar=np.full(moved.shape[0], 0.6, np.float32), to_sensor=to_sensor, from_sensor=Tg, want_src=False)
    gm = icp.getMap()
    def setdiff(a, b):
        sa = set(map(bytes, np.ascontiguousarray(a[:, :3]))); sb = set(map(bytes, np.ascontiguousarray(b[:, :3])))
        return len(sa - sb), len(sb - sa)
    print(f"scan {i:2d}: |gpu-ref| {amd.synth.pose_error(Tg, Tr)[0]:.2e} m  |gpu-alt| {amd.synth.pose_error(Tg, Ta)[0]:.2e} m  |alt-ref| {amd.synth.pose_error(Ta, Tr)[0]:.2e} m"
          f"  maps gpu {gm.shape[0]} ref {ref.map['xyz1'].shape[0]} alt {alt.map['xyz1'].shape[0]}  setdiff(gpu,ref) {setdiff(gm, ref.map['xyz1'])}  it {icp.stats.iterations if i else 0}")
