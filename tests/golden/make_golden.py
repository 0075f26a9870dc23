#!/usr/bin/env python
"""Generates the committed fixtures under tests/golden/.

Two kinds of vectors, neither produced by this repo's own code:

1. ``numpy_scipy_vectors.npz`` -- small seeded inputs with expected outputs computed by numpy /
   scipy (float64 linear algebra, cKDTree, brute force): the independent cross-check that pins the
   CPU oracle (SURVEY.md 8c "What pins results instead").  libpointmatcher / libnabo are not
   installed and not vendored, so the reference itself cannot produce vectors (parity unpinned).

2. ``bundled_scans.npz`` -- data files the reference ships for its example
   (/root/reference/examples/data/scans/*.vtk, trajectory.csv; BSD-3): the first two scans in the
   example's lexicographic order, as float32 arrays, plus all 14 trajectory rows.  With the bundled
   config the minimiser is IdentityErrorMinimizer (examples/config.yaml:62-63), so the expected
   trajectory equals the input trajectory -- a known answer that needs no reference binary.

Run from the repository root inside the authoring container (needs /root/reference for part 2):
    python tests/golden/make_golden.py
"""
import csv
import glob
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from norlab_icp_mapper_amd import synth  # noqa: E402  (scene generator only: no ICP code)


def rot(rv):
    return synth.rotvec_to_R(rv)


def numpy_vectors():
    rng = np.random.default_rng(20240928)
    out = {}
    # ---- clouds: a slab with structure in all directions ----
    m, n = 4000, 600
    ref = np.ones((m, 4), dtype=np.float32)
    ref[:, :3] = (rng.uniform(-1, 1, (m, 3)) * np.array([20.0, 12.0, 3.0])).astype(np.float32)
    qry = np.ones((n, 4), dtype=np.float32)
    qry[:, :3] = (rng.uniform(-1.1, 1.1, (n, 3)) * np.array([20.0, 12.0, 3.0])).astype(np.float32)
    out["knn_ref"], out["knn_qry"] = ref, qry
    # exact kNN, float64 distances of the float32 coordinates, k = 1 / 6, radius 2.0 and unbounded
    tree = cKDTree(ref[:, :3].astype(np.float64))
    for k in (1, 6):
        d, i = tree.query(qry[:, :3].astype(np.float64), k=k)
        out[f"knn_k{k}_ids"] = np.asarray(i, dtype=np.int64).reshape(n, k)
        out[f"knn_k{k}_d"] = np.asarray(d, dtype=np.float64).reshape(n, k)
        d, i = tree.query(qry[:, :3].astype(np.float64), k=k, distance_upper_bound=2.0)
        i = np.asarray(i, dtype=np.int64).reshape(n, k)
        d = np.asarray(d, dtype=np.float64).reshape(n, k)
        i[~np.isfinite(d)] = -1
        out[f"knn_k{k}_r2_ids"], out[f"knn_k{k}_r2_d"] = i, d

    # ---- rigid transform ----
    T = synth.make_T((0.4, -0.3, 0.2), (1.0, -2.0, 0.5))
    out["xf_T"] = T
    out["xf_out"] = (qry[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3])

    # ---- quantile of squared distances (Matches::getDistsQuantile semantics) ----
    d2 = (rng.gamma(2.0, 0.02, 5000)).astype(np.float32)
    d2[rng.integers(0, 5000, 200)] = np.inf
    d2[rng.integers(0, 5000, 50)] = 0.0
    out["q_d2"] = d2
    valid = np.sort(d2[np.isfinite(d2) & (d2 > 0)])
    for name, q in (("q85", np.float32(0.85)), ("q50", np.float32(0.5)), ("q10", np.float32(0.1))):
        idx = int(np.float32(valid.size) * q)
        out[name] = valid[min(idx, valid.size - 1)]
    out["q100"] = valid[-1]

    # ---- point-to-point: Kabsch with weights (float64 SVD) ----
    P = rng.normal(size=(800, 3)) * np.array([10.0, 6.0, 1.5])
    Rt = rot((0.05, -0.02, 0.08)); tt = np.array([0.3, -0.1, 0.05])
    Q = P @ Rt.T + tt + rng.normal(size=P.shape) * 0.01
    w = (rng.uniform(size=800) > 0.2).astype(np.float64)
    mp = (P * w[:, None]).sum(0) / w.sum(); mq = (Q * w[:, None]).sum(0) / w.sum()
    H = ((Q - mq) * w[:, None]).T @ (P - mp)
    U, S, Vt = np.linalg.svd(H)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        Vt[-1] *= -1; R = U @ Vt
    out["p2p_P"], out["p2p_Q"], out["p2p_w"] = P.astype(np.float32), Q.astype(np.float32), w.astype(np.float32)
    Tk = np.eye(4); Tk[:3, :3] = R; Tk[:3, 3] = mq - R @ mp
    out["p2p_T"] = Tk
    # reflection case: planar data, H with negative determinant
    Hn = np.diag([5.0, 2.0, 1e-9]) @ rot((0.3, 0.1, -0.2)); Hn[:, 2] *= -1
    U, S, Vt = np.linalg.svd(Hn); Rn = U @ Vt
    if np.linalg.det(Rn) < 0:
        Vt[-1] *= -1; Rn = U @ Vt
    out["refl_H"], out["refl_R"] = Hn, Rn

    # ---- point-to-plane: normal equations and solution (float64) ----
    nrm = rng.normal(size=(800, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    Pp = (P @ rot((0.002, -0.001, 0.003)).T + np.array([0.02, -0.01, 0.015]))
    F = np.concatenate([np.cross(Pp, nrm), nrm], axis=1)
    dot = ((Pp - P) * nrm).sum(1)
    A = (F * w[:, None]).T @ F
    b = -(F * w[:, None]).T @ dot
    x = np.linalg.solve(A, b)
    out["p2l_P"], out["p2l_Q"], out["p2l_N"] = Pp.astype(np.float32), P.astype(np.float32), nrm.astype(np.float32)
    out["p2l_A"], out["p2l_b"], out["p2l_x"] = A, b, x
    th = np.linalg.norm(x[:3]); k = x[:3] / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    Tp = np.eye(4); Tp[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K; Tp[:3, 3] = x[3:]
    out["p2l_T"] = Tp
    # rank-deficient system: all normals along z => only z translation and x/y rotations observable
    nz = np.tile(np.array([[0.0, 0.0, 1.0]]), (800, 1))
    Fz = np.concatenate([np.cross(Pp, nz), nz], axis=1)
    dz = ((Pp - P) * nz).sum(1)
    Az = Fz.T @ Fz; bz = -Fz.T @ dz
    out["sing_A"], out["sing_b"], out["sing_x"] = Az, bz, np.linalg.pinv(Az, rcond=1e-6) @ bz

    # ---- surface normals: plane patches with known normal ----
    g = np.stack(np.meshgrid(np.linspace(-2, 2, 30), np.linspace(-2, 2, 30)), -1).reshape(-1, 2)
    nn = np.array([0.2, -0.3, 0.93]); nn /= np.linalg.norm(nn)
    e1 = np.cross(nn, [1, 0, 0]); e1 /= np.linalg.norm(e1); e2 = np.cross(nn, e1)
    plane = g[:, :1] * e1 + g[:, 1:] * e2 + rng.normal(size=(g.shape[0], 3)) * 1e-4
    pl = np.ones((plane.shape[0], 4), dtype=np.float32); pl[:, :3] = plane.astype(np.float32)
    out["sn_pts"], out["sn_normal"] = pl, nn

    # ---- 20 m cell binning ----
    cpts = (rng.uniform(-130, 130, (500, 3))).astype(np.float32)
    cpts[:6, 0] = [-20.0, 20.0, 0.0, -0.0, 19.999998, -20.000002]
    c4 = np.ones((500, 4), dtype=np.float32); c4[:, :3] = cpts
    out["cell_pts"] = c4
    out["cell_ijk"] = np.floor(cpts.astype(np.float32) / np.float32(20.0)).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "numpy_scipy_vectors.npz"), **out)
    print("wrote numpy_scipy_vectors.npz", {k: v.shape if hasattr(v, "shape") else v for k, v in list(out.items())[:5]})


def read_vtk_points(path):
    """ASCII VTK POLYDATA of libpointmatcher: POINTS n float, then SCALARS blocks."""
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    while not lines[i].startswith("POINTS"):
        i += 1
    n = int(lines[i].split()[1])
    pts = np.array([[float(v) for v in lines[i + 1 + r].split()] for r in range(n)], dtype=np.float32)
    desc = {}
    j = i + 1 + n
    while j < len(lines):
        if lines[j].startswith("SCALARS"):
            name = lines[j].split()[1]
            vals = np.array([float(lines[j + 2 + r]) for r in range(n)], dtype=np.float32)
            desc[name] = vals
            j += 2 + n
        else:
            j += 1
    return pts, desc


def bundled():
    ref = "/root/reference/examples/data"
    if not os.path.isdir(ref):
        print("reference not mounted: skipping bundled_scans.npz")
        return
    scans = sorted(glob.glob(os.path.join(ref, "scans", "*.vtk")))  # lexicographic, like the example (cpp:191)
    out = {"scan_names": np.array([os.path.basename(s) for s in scans])}
    for k in (0, 1):
        pts, desc = read_vtk_points(scans[k])
        out[f"scan{k}_xyz"] = pts
        out[f"scan{k}_intensity"] = desc.get("intensity", np.zeros(len(pts), np.float32))
    rows = []
    with open(os.path.join(ref, "trajectory.csv")) as f:
        for r in csv.DictReader(f):
            rows.append([float(r["header.stamp.sec"]), float(r["header.stamp.nanosec"]),
                         float(r["pose.pose.position.x"]), float(r["pose.pose.position.y"]), float(r["pose.pose.position.z"]),
                         float(r["pose.pose.orientation.x"]), float(r["pose.pose.orientation.y"]),
                         float(r["pose.pose.orientation.z"]), float(r["pose.pose.orientation.w"])])
    out["trajectory"] = np.array(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "bundled_scans.npz"), **out)
    print("wrote bundled_scans.npz", out["scan0_xyz"].shape, out["scan1_xyz"].shape, out["trajectory"].shape)


def bundled_all():
    """All 14 scans of the reference's example (xyz as float32, the example's lexicographic file order, cpp:191) and the
    14 trajectory rows: the input of BASELINE config 4 (full trajectory replay).  ~5 MB compressed."""
    ref = "/root/reference/examples/data"
    if not os.path.isdir(ref):
        print("reference not mounted: skipping bundled_scans_all.npz")
        return
    scans = sorted(glob.glob(os.path.join(ref, "scans", "*.vtk")))
    out = {"scan_names": np.array([os.path.basename(s) for s in scans])}
    for k, path in enumerate(scans):
        pts, _ = read_vtk_points(path)
        out[f"scan{k}_xyz"] = pts
    out["trajectory"] = np.load(os.path.join(HERE, "bundled_scans.npz"))["trajectory"]
    np.savez_compressed(os.path.join(HERE, "bundled_scans_all.npz"), **out)
    print("wrote bundled_scans_all.npz", [out[f"scan{k}_xyz"].shape[0] for k in range(len(scans))])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bundled_all":
        bundled_all()
    else:
        numpy_vectors()
        bundled()
        bundled_all()
