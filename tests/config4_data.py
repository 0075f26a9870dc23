"""BASELINE config 4 (examples/build_map_from_scans_and_trajectory.cpp:196-239 on the bundled examples/data): the shipped configuration
(examples/config.yaml with `epsilon: 0` and PointToPlane) and the fixture tests/golden/bundled_scans_all.npz written back into the reference's
on-disk layout.  Shared by tests/test_gpu_configs.py (parity of the replay against the oracle's) and bench.py (chains.config4_replay: the same
replay on the clock, the oracle's beside it)."""
import os

import numpy as np

CONFIG4_YAML = """
input:
  - BoundingBoxDataPointsFilter:
      xMin: -1.5
      xMax: 0.5
      yMin: -1
      yMax: 1
      zMin: -1
      zMax: 0.5
      removeInside: 1
  - BoundingBoxDataPointsFilter:
      xMin: -6
      xMax: -1.5
      yMin: -2.5
      yMax: 2.5
      zMin: -1
      zMax: 1
      removeInside: 1
  - AddDescriptorDataPointsFilter:
      descriptorName: probabilityDynamic
      descriptorDimension: 1
      descriptorValues: [0.6]
post:
    - SurfaceNormalDataPointsFilter:
        knn: 10
    - CutAtDescriptorThresholdDataPointsFilter:
        descName: probabilityDynamic
        useLargerThan: 1
        threshold: 0.65
mapper:
  updateCondition:
    type: delay
    value: 0.05
  mapperModule:
    - DynamicPointsMapperModule:
        thresholdDynamic: 0.9
        alpha: 0.8
        beta: 0.99
        beamHalfAngle: 0.01
        epsilonA: 0.01
        epsilonD: 0.01
    - OctreeMapperModule:
        buildParallel: 1
        maxSizeByNode: 0.15
        samplingMethod: 0
  sensorMaxRange: 200
icp:
  matcher:
    KDTreeMatcher:
      knn: 6
      maxDist: 2.0
      epsilon: 0
  errorMinimizer:
    PointToPlaneErrorMinimizer:
  transformationCheckers:
    - CounterTransformationChecker:
        maxIterationCount: 10
  inspector: NullInspector
"""


def quat_T(row):
    x, y, z, qx, qy, qz, qw = row
    R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                  [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                  [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = [x, y, z]
    return T.astype(np.float32)


def write_bundled_dataset(tmp, z):
    """the fixture back into the reference's on-disk layout: scans/<original names>.vtk (ASCII, libpointmatcher's
    dialect) + trajectory.csv (the ROS odometry dump's columns)"""
    os.makedirs(os.path.join(tmp, "scans"))
    names = [str(s) for s in z["scan_names"]]
    for k, name in enumerate(names):
        pts = z[f"scan{k}_xyz"]
        n = pts.shape[0]
        with open(os.path.join(tmp, "scans", name), "w") as f:
            f.write("# vtk DataFile Version 3.0\nFile created by libpointmatcher\nASCII\nDATASET POLYDATA\n")
            f.write(f"POINTS {n} float\n")
            np.savetxt(f, pts, fmt="%.9g")
            f.write(f"VERTICES {n} {2 * n}\n")
            np.savetxt(f, np.stack([np.ones(n, int), np.arange(n)], 1), fmt="%d")
    traj = z["trajectory"]
    with open(os.path.join(tmp, "trajectory.csv"), "w") as f:
        f.write("header.stamp.sec,header.stamp.nanosec,header.frame_id,child_frame_id,pose.pose.position.x,pose.pose.position.y,"
                "pose.pose.position.z,pose.pose.orientation.x,pose.pose.orientation.y,pose.pose.orientation.z,pose.pose.orientation.w,pose.covariance\n")
        for r in traj:
            f.write(f"{int(r[0])},{int(r[1])},map,base_link," + ",".join(repr(float(v)) for v in r[2:]) + ",[0. 0. 0.]\n")
    return names, traj



def oracle_mapper_args(nthreads):
    """tests/oracle_mapper.OracleMapper arguments of the same configuration"""
    return (dict(knn=6, max_dist=2.0, minimizer=2, outliers=[], max_iterations=10),
            [("dynamic_points", dict(threshold_dynamic=0.9, alpha=0.8, beta=0.99, beam_half_angle=0.01, epsilon_a=0.01, epsilon_d=0.01)),
             ("octree", 0.15, 0)],
            dict(post=[("surface_normals", 10), ("cut", "probabilityDynamic", 1, 0.65)], update=("delay", 0.05), sensor_max_range=200.0,
                 input_filters=[("bounding_box", (-1.5, -1, -1), (0.5, 1, 0.5), 1), ("bounding_box", (-6, -2.5, -1), (-1.5, 2.5, 1), 1)],
                 add_descriptors=[("probabilityDynamic", 0.6)], nthreads=nthreads))
