"""Deterministic synthetic workload of SURVEY.md section 8(d) ("box + pillars").

Room x,y in [-50 s, 50 s], z in [0, 10] (six faces) plus 16 vertical square pillars (1 m side, full
height) on the 4 x 4 lattice (+-10 s, +-30 s).  Points are area-uniform on all faces with isotropic
Gaussian noise sigma; the analytic face normal is returned as the ``normals`` descriptor.  The scan is
an independent sample of the same surfaces restricted to ``scan_range`` metres around the sensor,
then moved by T_gt^-1 so that ICP with an identity prior must recover T_gt.

The generator is counter based (splitmix64 of seed and index) so that any language can reproduce the
clouds bit for bit: u_i = splitmix64(seed * 0x9E3779B97F4A7C15 + i) >> 11, scaled by 2^-53.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed, stream, n, offset=0):
    """n doubles in [0, 1) from (seed, stream); streams are independent counters."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03)
        idx = base + np.arange(offset, offset + n, dtype=np.uint64)
    return (splitmix64(idx) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def gaussian(seed, stream, n):
    u1 = uniform(seed, stream, n)
    u2 = uniform(seed, stream + 1, n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def _faces(scale=1.0):
    """list of (origin, edge_u, edge_v, normal); a face is origin + a*edge_u + b*edge_v, a,b in [0,1)"""
    L = 50.0 * scale
    H = 10.0
    f = []
    f.append(((-L, -L, 0.0), (2 * L, 0, 0), (0, 2 * L, 0), (0, 0, 1)))     # floor
    f.append(((-L, -L, H), (2 * L, 0, 0), (0, 2 * L, 0), (0, 0, -1)))      # ceiling
    f.append(((-L, -L, 0.0), (0, 2 * L, 0), (0, 0, H), (1, 0, 0)))         # wall x = -L
    f.append(((L, -L, 0.0), (0, 2 * L, 0), (0, 0, H), (-1, 0, 0)))         # wall x = +L
    f.append(((-L, -L, 0.0), (2 * L, 0, 0), (0, 0, H), (0, 1, 0)))         # wall y = -L
    f.append(((-L, L, 0.0), (2 * L, 0, 0), (0, 0, H), (0, -1, 0)))         # wall y = +L
    for px in (-30.0, -10.0, 10.0, 30.0):
        for py in (-30.0, -10.0, 10.0, 30.0):
            cx, cy = px * scale, py * scale
            f.append(((cx - 0.5, cy - 0.5, 0.0), (0, 1.0, 0), (0, 0, H), (-1, 0, 0)))
            f.append(((cx + 0.5, cy - 0.5, 0.0), (0, 1.0, 0), (0, 0, H), (1, 0, 0)))
            f.append(((cx - 0.5, cy - 0.5, 0.0), (1.0, 0, 0), (0, 0, H), (0, -1, 0)))
            f.append(((cx - 0.5, cy + 0.5, 0.0), (1.0, 0, 0), (0, 0, H), (0, 1, 0)))
    org = np.array([x[0] for x in f], dtype=np.float64)
    eu = np.array([x[1] for x in f], dtype=np.float64)
    ev = np.array([x[2] for x in f], dtype=np.float64)
    nrm = np.array([x[3] for x in f], dtype=np.float64)
    area = np.linalg.norm(np.cross(eu, ev), axis=1)
    return org, eu, ev, nrm, area


def sample_surfaces(n, seed, scale=1.0, offset=0):
    """n area-uniform surface samples (float64 xyz, normals)."""
    org, eu, ev, nrm, area = _faces(scale)
    cdf = np.cumsum(area) / np.sum(area)
    u0 = uniform(seed, 0, n, offset)
    a = uniform(seed, 1, n, offset)
    b = uniform(seed, 2, n, offset)
    face = np.minimum(np.searchsorted(cdf, u0, side="right"), len(area) - 1)
    pts = org[face] + a[:, None] * eu[face] + b[:, None] * ev[face]
    return pts, nrm[face]


def rotvec_to_R(rv):
    rv = np.asarray(rv, dtype=np.float64)
    th = np.linalg.norm(rv)
    if th == 0:
        return np.eye(3)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def make_T(rotvec, t):
    T = np.eye(4)
    T[:3, :3] = rotvec_to_R(rotvec)
    T[:3, 3] = t
    return T


T_GT_ROTVEC = (0.010, -0.015, 0.020)
T_GT_TRANS = (0.10, -0.08, 0.05)


def make_scene(m=1_000_000, n=100_000, sigma=0.01, scale=1.0, seed_map=42, seed_scan=43, seed_noise=44,
               sensor=(3.0, -2.0, 1.5), scan_range=60.0, rotvec=T_GT_ROTVEC, trans=T_GT_TRANS):
    """Returns dict(map (M,4) f32, normals (M,3) f32, scan (N,4) f32, scan_normals (N,3) f32, T_gt (4,4) f64)."""
    mp, mn = sample_surfaces(m, seed_map, scale)
    noise = np.stack([gaussian(seed_noise, 10 + 2 * r, m) for r in range(3)], axis=1) * sigma
    mp = mp + noise

    # scan: rejection on range from the sensor, first n accepted candidates of the stream
    sensor = np.asarray(sensor, dtype=np.float64)
    acc_p, acc_n, got, off = [], [], 0, 0
    while got < n:
        batch = max(4 * (n - got), 4096)
        p, nn = sample_surfaces(batch, seed_scan, scale, offset=off)
        off += batch
        keep = np.linalg.norm(p - sensor, axis=1) <= scan_range
        acc_p.append(p[keep]); acc_n.append(nn[keep]); got += int(keep.sum())
    sp = np.concatenate(acc_p)[:n]
    sn = np.concatenate(acc_n)[:n]
    snoise = np.stack([gaussian(seed_noise, 100 + 2 * r, n) for r in range(3)], axis=1) * sigma
    sp = sp + snoise
    T_gt = make_T(rotvec, trans)
    Ti = np.linalg.inv(T_gt)
    sp = sp @ Ti[:3, :3].T + Ti[:3, 3]
    sn = sn @ Ti[:3, :3].T

    def h(p):
        out = np.ones((p.shape[0], 4), dtype=np.float32)
        out[:, :3] = p.astype(np.float32)
        return out

    return {"map": h(mp), "normals": np.ascontiguousarray(mn, dtype=np.float32), "scan": h(sp),
            "scan_normals": np.ascontiguousarray(sn, dtype=np.float32), "T_gt": T_gt}


def pose_error(T_a, T_b):
    """(translation error in m, rotation error in rad = angle of R_a^T R_b)"""
    T_a = np.asarray(T_a, dtype=np.float64); T_b = np.asarray(T_b, dtype=np.float64)
    dt = float(np.linalg.norm(T_a[:3, 3] - T_b[:3, 3]))
    R = T_a[:3, :3].T @ T_b[:3, :3]
    c = (np.trace(R) - 1.0) / 2.0
    s = 0.5 * np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return dt, float(np.arctan2(s, c))
