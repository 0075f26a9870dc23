#!/usr/bin/env python
"""Independent vectors for the part of the path the oracle restates "as recalled" (VERDICT r3 missing 1): Robust weights, VarTrimmedDist,
SurfaceNormalOutlierFilter, OctreeGridDataPointsFilter, SamplingSurfaceNormalDataPointsFilter and DynamicPointsMapperModule.

Nothing here imports, links or calls oracle/ or the HIP library: every expected value comes from numpy / scipy (float64) or from a
pure-Python recursion written from the specification (for DynamicPoints from the reference's own source,
/root/reference/norlab_icp_mapper/MapperModules/DynamicPointsMapperModule.cpp:34-172 -- every block below cites the lines it follows).
The output, tests/golden/numpy_recalled_vectors.npz, is committed; tests/test_oracle_recalled.py (CPU) holds the oracle to it and
tests/test_gpu_recalled.py (-m gpu) the HIP path, directly.

    python tests/golden/make_recalled.py
"""
import os

import numpy as np
from scipy.spatial import cKDTree

HERE = os.path.dirname(os.path.abspath(__file__))
F32 = np.float32


# ---------------------------------------------------------------------------------------------------------------------------------
# RobustOutlierFilter: the eight M-estimator weight functions of e2 = d2 / scale^2 with tuning k (closed forms), scale = 1 or
# sqrt(MAD) with MAD = median(|d2 - median(d2)|) over the finite entries (odd counts: numpy.median IS the middle element)
# ---------------------------------------------------------------------------------------------------------------------------------
def robust_weight(name, e2, k):
    e2 = np.asarray(e2, dtype=np.float64)
    if name == "cauchy":
        return 1.0 / (1.0 + e2 / k ** 2)
    if name == "welsch":
        return np.exp(-e2 / k ** 2)
    if name == "sc":
        return np.where(e2 >= k, 4.0 * k ** 2 / (k + e2) ** 2, 1.0)
    if name == "gm":
        return k ** 2 / (k + e2) ** 2
    if name == "tukey":
        return np.where(e2 >= k ** 2, 0.0, (1.0 - e2 / k ** 2) ** 2)
    if name == "huber":
        return np.where(e2 >= k ** 2, k / np.sqrt(np.where(e2 > 0, e2, 1.0)), 1.0)
    if name == "L1":
        return 1.0 / np.sqrt(e2)
    if name == "student":
        return (k + 3.0) * (1.0 + e2 / k) ** (-(k + 3.0) / 2.0) / (k + e2)
    raise ValueError(name)


def robust_vectors(rng, out):
    n = 1501
    d2 = rng.gamma(2.0, 0.02, n).astype(F32)
    d2[rng.choice(n, 100, replace=False)] = np.inf            # 1401 finite entries: odd
    out["rob_d2"] = d2
    fin = np.isfinite(d2)
    med = np.median(d2[fin].astype(np.float64))
    mad = np.median(np.abs(d2[fin].astype(np.float64) - med))
    out["rob_mad_scale"] = np.sqrt(mad)
    names = ["cauchy", "welsch", "sc", "gm", "tukey", "huber", "L1", "student"]
    out["rob_names"] = np.array(names)
    out["rob_tuning"] = np.array([1.5, 0.8, 0.3, 0.5, 2.0, 1.2, 1.0, 2.5])
    for name, k in zip(names, out["rob_tuning"]):
        for tag, scale in (("none", 1.0), ("mad", np.sqrt(mad))):
            w = np.zeros(n)
            w[fin] = robust_weight(name, d2[fin].astype(np.float64) / scale ** 2, float(k))
            out[f"rob_w_{name}_{tag}"] = w


# ---------------------------------------------------------------------------------------------------------------------------------
# VarTrimmedDistOutlierFilter (Phillips et al. 2007): FRMS(i) = (sum of the i + 1 smallest valid d2) / ((i + 1) ((i + 1) / N)^(2 lambda))
# minimised by brute force over floor(minRatio N) <= i < min(floor(maxRatio N), V); ratio = i / N; the limit is the element a sort puts
# at rank (size_t)(float(V) ratio) of the V valid (finite, > 0) entries
# ---------------------------------------------------------------------------------------------------------------------------------
def var_trimmed(d2, min_ratio, max_ratio, lam):
    N = d2.size
    valid = np.sort(d2[np.isfinite(d2) & (d2 > 0)].astype(np.float64))
    V = valid.size
    lo, hi = int(np.floor(F32(min_ratio) * F32(N))), min(int(np.floor(F32(max_ratio) * F32(N))), V)
    cs = np.cumsum(valid)
    best, best_i = None, lo
    for i in range(lo, hi):
        frms = cs[i] / ((i + 1) * ((i + 1) / N) ** (2.0 * lam))
        if best is None or frms < best:
            best, best_i = frms, i
    ratio = F32(best_i) / F32(N)
    rank = min(int(F32(V) * ratio), V - 1)
    return float(ratio), valid[rank], best_i


def var_trimmed_vectors(rng, out):
    cases = []
    for c, (n, frac_out, minr, maxr, lam) in enumerate([(3000, 0.2, 0.05, 0.99, 0.95), (2500, 0.4, 0.3, 0.9, 0.8), (1800, 0.05, 0.05, 0.99, 2.0),
                                                         (2200, 0.3, 0.6, 0.7, 0.95)]):
        inl = rng.gamma(2.0, 0.002, n)
        outl = rng.uniform(0.5, 4.0, n)
        d2 = np.where(rng.uniform(size=n) < frac_out, outl, inl).astype(F32)
        d2[rng.choice(n, n // 50, replace=False)] = np.inf
        ratio, lim, bi = var_trimmed(d2, minr, maxr, lam)
        out[f"vt{c}_d2"] = d2
        out[f"vt{c}_prm"] = np.array([minr, maxr, lam])
        out[f"vt{c}_ratio"], out[f"vt{c}_limit"], out[f"vt{c}_rank"] = ratio, F32(lim), bi
        cases.append(c)
    out["vt_cases"] = np.array(cases)


# ---------------------------------------------------------------------------------------------------------------------------------
# SurfaceNormalOutlierFilter{maxAngle}: w = (n_read . n_ref[id]) > cos(maxAngle); pairs within 1e-4 of the threshold are left out of
# the comparison (mask) -- a float dot and a double dot may fall on different sides there
# ---------------------------------------------------------------------------------------------------------------------------------
def surface_normal_vectors(rng, out):
    m, n, k = 900, 400, 3
    ref_n = rng.normal(size=(m, 3)); ref_n /= np.linalg.norm(ref_n, axis=1, keepdims=True)
    read_n = rng.normal(size=(n, 3)); read_n /= np.linalg.norm(read_n, axis=1, keepdims=True)
    ids = rng.integers(0, m, (n, k)).astype(np.int32)
    ids[rng.integers(0, n, 20), k - 1] = -1
    ref_n, read_n = ref_n.astype(F32), read_n.astype(F32)
    out["sno_ref_n"], out["sno_read_n"], out["sno_ids"] = ref_n, read_n, ids
    for tag, ang in (("a", 0.7), ("b", 1.57)):
        dots = np.einsum("nkc,nc->nk", ref_n.astype(np.float64)[np.maximum(ids, 0)], read_n.astype(np.float64))
        w = (dots > np.cos(ang)) & (ids >= 0)
        out[f"sno_{tag}_angle"] = ang
        out[f"sno_{tag}_w"] = w.astype(F32)
        out[f"sno_{tag}_sure"] = (np.abs(dots - np.cos(ang)) > 1e-4) | (ids < 0)


# ---------------------------------------------------------------------------------------------------------------------------------
# OctreeGridDataPointsFilter{maxSizeByNode, maxPointByNode, samplingMethod 0}: bounding-cube root (centre = min +
# (max - min) / 2, radius = max extent / 2, float32), a node is a leaf iff 2 radius <= maxSize or it holds <= maxPoint points,
# children by `p > centre` per axis (bit 0 x, bit 1 y, bit 2 z), child centre +- radius / 2, lists keep the parent's order, leaves
# visited depth first in child order; one point per leaf: the first of its list
# ---------------------------------------------------------------------------------------------------------------------------------
def octree_sample(pts, max_size, max_pts):
    p = pts[:, :3].astype(F32)
    lo, hi = p.min(axis=0), p.max(axis=0)
    ext = (hi - lo).astype(F32)
    centre = (lo + ext * F32(0.5)).astype(F32)
    radius = F32((ext * F32(0.5)).max())
    out = []

    def build(idx, c, r, depth):
        if len(idx) == 0:
            return
        if F32(r * F32(2.0)) <= F32(max_size) or len(idx) <= max(max_pts, 1) or depth >= 21:
            out.append(idx[0])
            return
        kids = [[] for _ in range(8)]
        for i in idx:
            o = (1 if p[i, 0] > c[0] else 0) | (2 if p[i, 1] > c[1] else 0) | (4 if p[i, 2] > c[2] else 0)
            kids[o].append(i)
        half = F32(r * F32(0.5))
        for o in range(8):
            cc = np.array([c[0] + (half if o & 1 else -half), c[1] + (half if o & 2 else -half), c[2] + (half if o & 4 else -half)], dtype=F32)
            build(kids[o], cc, half, depth + 1)

    build(list(range(p.shape[0])), centre, radius, 0)
    return np.array(out, dtype=np.int32)


def octree_vectors(rng, out):
    n = 1800
    pts = np.ones((n, 4), dtype=F32)
    # a wall, a floor patch and a cluster with duplicates
    pts[:700, :3] = np.c_[rng.uniform(-4, 4, 700), np.full(700, 3.0) + rng.normal(0, 0.01, 700), rng.uniform(0, 2.5, 700)]
    pts[700:1500, :3] = np.c_[rng.uniform(-4, 4, 800), rng.uniform(-3, 3, 800), rng.normal(0, 0.01, 800)]
    pts[1500:, :3] = rng.normal(0, 0.05, (300, 3)) + np.array([1.0, -1.0, 1.0])
    pts[1790:, :3] = pts[1500:1510, :3]
    out["oct_pts"] = pts
    for tag, (ms, mp) in (("a", (0.15, 1)), ("b", (0.5, 4)), ("c", (0.0, 16))):
        out[f"oct_{tag}_prm"] = np.array([ms, mp])
        out[f"oct_{tag}_order"] = octree_sample(pts, ms, mp)


# ---------------------------------------------------------------------------------------------------------------------------------
# SamplingSurfaceNormalDataPointsFilter{ratio, knn, samplingMethod 0, maxBoxDim, seed}: median split of the widest box dimension
# (ties of the coordinate by index) until a box holds <= knn points; the right half takes cnt / 2 points; the box edges follow the
# cut; a box wider than maxBoxDim or of rank < 2 is dropped; its normal is the eigenvector of the smallest eigenvalue of the
# covariance (numpy.linalg.eigh, float64); the points of a surviving box, in index order, are kept with probability `ratio`:
# std::minstd_rand (x <- 48271 x mod 2^31 - 1, seed % (2^31 - 1), 0 -> 1) drawn as float(x) / 2147483645.0f, one number per point
# ---------------------------------------------------------------------------------------------------------------------------------
def ssn_sample(pts, ratio, knn, max_box, seed):
    p = pts[:, :3].astype(F32)
    state = [seed % 2147483647 or 1]
    order, normals = [], []

    def draw():
        state[0] = (state[0] * 48271) % 2147483647
        return F32(state[0]) / F32(2147483645.0)

    def fuse(idx):
        idx = sorted(idx)
        q = p[idx].astype(np.float64)
        if float((p[idx].max(axis=0) - p[idx].min(axis=0)).max()) > max_box:
            return
        C = np.cov(q.T, bias=True) * len(idx) if len(idx) > 1 else np.zeros((3, 3))
        w, Q = np.linalg.eigh(C)
        wmax = np.abs(w).max()
        if not (wmax > 0 and (np.abs(w) > 3.0 * np.finfo(F32).eps * wmax).sum() >= 2):
            return
        nrm = Q[:, int(np.argmin(w))]
        for i in idx:
            if draw() < F32(ratio):
                order.append(i); normals.append(nrm)

    def build(idx, lo, hi):
        if not idx:
            return
        if len(idx) <= knn:
            fuse(idx)
            return
        dim = 0
        for r in (1, 2):
            if F32(hi[r] - lo[r]) > F32(hi[dim] - lo[dim]):
                dim = r
        idx = sorted(idx, key=lambda i: (p[i, dim], i))
        right = len(idx) // 2
        left = len(idx) - right
        cut = p[idx[left], dim]
        lhi, rlo = hi.copy(), lo.copy()
        lhi[dim] = cut; rlo[dim] = cut
        build(idx[:left], lo, lhi)
        build(idx[left:], rlo, hi)

    build(list(range(p.shape[0])), p.min(axis=0), p.max(axis=0))
    return np.array(order, dtype=np.int32), np.array(normals, dtype=np.float64).reshape(-1, 3)


def ssn_vectors(rng, out):
    n = 1500
    pts = np.ones((n, 4), dtype=F32)
    pts[:900, :3] = np.c_[rng.uniform(-5, 5, 900), rng.uniform(-4, 4, 900), rng.normal(0, 0.02, 900)]
    pts[900:1400, :3] = np.c_[np.full(500, 5.0) + rng.normal(0, 0.02, 500), rng.uniform(-4, 4, 500), rng.uniform(0, 3, 500)]
    pts[1400:, :3] = np.c_[np.linspace(-1, 1, 100), np.zeros(100), np.full(100, 1.0)]       # a line: rank-1 boxes are dropped
    pts[50:60, 0] = pts[40:50, 0]                                                            # coordinate ties
    out["ssn_pts"] = pts
    for tag, (ratio, knn, mb, seed) in (("a", (0.5, 7, np.inf, 1)), ("b", (1.0, 12, np.inf, 5)), ("c", (0.8, 5, 0.6, 123))):
        o, nr = ssn_sample(pts, ratio, knn, mb, seed)
        out[f"ssn_{tag}_prm"] = np.array([ratio, knn, mb, seed], dtype=np.float64)
        out[f"ssn_{tag}_order"], out[f"ssn_{tag}_normals"] = o, nr


# ---------------------------------------------------------------------------------------------------------------------------------
# DynamicPointsMapperModule::inPlaceUpdateMap, numpy float64 transliteration of
# /root/reference/norlab_icp_mapper/MapperModules/DynamicPointsMapperModule.cpp (line numbers in the comments)
# ---------------------------------------------------------------------------------------------------------------------------------
def spherical(p):                                                                      # :156-172 convertToSphericalCoordinates (3-D)
    radii = np.linalg.norm(p, axis=1)
    return radii, np.c_[np.arcsin(p[:, 2] / radii), np.arctan2(p[:, 1], p[:, 0])]


def dynamic_points_update(pose, inp, mp, normals, prob, thresholdDynamic, alpha, beta, beamHalfAngle, epsilonA, epsilonD, sensorMaxRange):
    eps = 0.0001                                                                       # :49
    Tinv = np.linalg.inv(pose.astype(np.float64))
    inS = inp[:, :3].astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3]                 # :51
    _, inAngles = spherical(inS)                                                       # :53-55
    mS = mp[:, :3].astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3]                   # :57
    nS = normals.astype(np.float64) @ Tinv[:3, :3].T                                   # (Transformation::compute rotates `normals`)
    within = np.linalg.norm(mS, axis=1) < sensorMaxRange                               # :60-69
    gid = np.nonzero(within)[0]
    mS, nS = mS[gid], nS[gid]
    _, mAngles = spherical(mS)                                                         # :71-73
    tree = cKDTree(inAngles)
    d, ids = tree.query(mAngles, k=1, distance_upper_bound=2 * beamHalfAngle)          # :75-78 (libnabo returns SQUARED distances)
    d2nd, _ = tree.query(mAngles, k=2)                                                 # (unbounded: for the margins only)
    out = prob.astype(np.float64).copy()
    margin = np.full(prob.shape[0], np.inf)      # how far the point is from every decision that a float32 evaluation could take otherwise
    margin[~within] = np.abs(np.linalg.norm(mp[~within, :3].astype(np.float64) @ Tinv[:3, :3].T + Tinv[:3, 3], axis=1) - sensorMaxRange)
    for i in range(mS.shape[0]):
        if not np.isfinite(d[i]):                                                      # :83
            margin[gid[i]] = abs(d2nd[i, 0] - 2 * beamHalfAngle) * 100
            continue
        ip, lp = inS[ids[i]], mS[i]                                                    # :85-90
        delta = np.linalg.norm(ip - lp)                                                # :91
        d_max = epsilonA * np.linalg.norm(ip)                                          # :92
        w_v = eps + (1. - eps) * abs(nS[i] @ (lp / np.linalg.norm(lp)))                # :96
        w_d1 = eps + (1. - eps) * (1. - d[i] / (2 * beamHalfAngle))                    # :97 (sqrt of the squared distance = d)
        offset = delta - epsilonD                                                      # :99
        w_d2 = 1.                                                                      # :100-111
        if delta < epsilonD or np.linalg.norm(lp) > np.linalg.norm(ip):
            w_d2 = eps
        elif offset < d_max:
            w_d2 = eps + (1 - eps) * offset / d_max
        w_p2 = eps                                                                     # :113-124
        if delta < epsilonD:
            w_p2 = 1
        elif offset < d_max:
            w_p2 = eps + (1. - eps) * (1. - offset / d_max)
        g = gid[i]
        margin[g] = min(abs(d2nd[i, 0] - 2 * beamHalfAngle) * 100, abs(np.linalg.norm(lp) - sensorMaxRange), abs(delta - epsilonD), abs(np.linalg.norm(lp) - np.linalg.norm(ip)), abs(offset - d_max),
                        abs(np.linalg.norm(ip) + epsilonD + d_max - np.linalg.norm(lp)), abs(prob[g] - thresholdDynamic) * 10)
        if (np.linalg.norm(ip) + epsilonD + d_max) >= np.linalg.norm(lp):              # :126
            lastDyn = float(prob[g])                                                   # :128
            c1 = 1 - (w_v * w_d1)                                                      # :130-131
            c2 = w_v * w_d1
            if lastDyn < thresholdDynamic:                                             # :135-139
                pd = c1 * lastDyn + c2 * w_d2 * ((1 - alpha) * (1 - lastDyn) + beta * lastDyn)
                ps = c1 * (1 - lastDyn) + c2 * w_p2 * (alpha * (1 - lastDyn) + (1 - beta) * lastDyn)
            else:                                                                      # :141-144
                pd, ps = 1 - eps, eps
            out[g] = pd / (pd + ps)                                                    # :147
    # second-nearest beams at (almost) the same angular distance make the choice of the beam rounding dependent: mark them unsure
    amb = np.zeros(prob.shape[0], dtype=bool)
    amb[gid] = np.abs(d2nd[:, 1] - d2nd[:, 0]) < 1e-6
    return out, margin, amb


def dynamic_vectors(rng, out):
    # a sensor in a room; the map is the room with a few objects the new scan sees through (moved away) or in front of
    pose = np.eye(4)
    th = 0.3
    pose[:3, :3] = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    pose[:3, 3] = [1.0, -0.5, 0.8]
    n, m = 3000, 2500
    dirs = rng.normal(size=(n, 3)); dirs[:, 2] *= 0.3; dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    rng_in = rng.uniform(3.0, 12.0, n)
    in_sensor = dirs * rng_in[:, None]
    inp = np.ones((n, 4), dtype=F32)
    inp[:, :3] = (in_sensor @ pose[:3, :3].T + pose[:3, 3]).astype(F32)
    # map points: along input beams (slightly off in angle), at a range ratio in {closer, same, farther}, plus some beyond sensorMaxRange
    pick = rng.integers(0, n, m)
    ratio = rng.choice([0.5, 0.8, 0.97, 0.995, 1.005, 1.03, 1.3], m)
    md = dirs[pick] + rng.normal(0, 0.004, (m, 3)); md /= np.linalg.norm(md, axis=1, keepdims=True)
    map_sensor = md * (rng_in[pick] * ratio)[:, None]
    map_sensor[:100] *= 40.0                                                            # out of range
    mp = np.ones((m, 4), dtype=F32)
    mp[:, :3] = (map_sensor @ pose[:3, :3].T + pose[:3, 3]).astype(F32)
    nrm = rng.normal(size=(m, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    prob = rng.choice([0.1, 0.3, 0.55, 0.7, 0.9], m).astype(F32)
    prm = dict(thresholdDynamic=0.6, alpha=0.8, beta=0.99, beamHalfAngle=0.01, epsilonA=0.01, epsilonD=0.01, sensorMaxRange=100.0)
    expected, margin, amb = dynamic_points_update(pose.astype(F32), inp, mp, nrm.astype(F32), prob, **prm)
    out["dyn_pose"], out["dyn_input"], out["dyn_map"], out["dyn_normals"], out["dyn_prob"] = pose.astype(F32), inp, mp, nrm.astype(F32), prob
    out["dyn_prm"] = np.array([prm[k] for k in ("thresholdDynamic", "alpha", "beta", "beamHalfAngle", "epsilonA", "epsilonD", "sensorMaxRange")])
    out["dyn_expected"] = expected
    out["dyn_sure"] = (margin > 1e-4) & ~amb
    print("dynamic points: updated", int((expected != prob).sum()), "of", m, "sure", int(out["dyn_sure"].sum()))


def main():
    rng = np.random.default_rng(20240929)
    out = {}
    robust_vectors(rng, out)
    var_trimmed_vectors(rng, out)
    surface_normal_vectors(rng, out)
    octree_vectors(rng, out)
    ssn_vectors(rng, out)
    dynamic_vectors(rng, out)
    np.savez_compressed(os.path.join(HERE, "numpy_recalled_vectors.npz"), **out)
    print("wrote numpy_recalled_vectors.npz:", len(out), "arrays;", {k: out[k].shape for k in ("oct_a_order", "oct_b_order", "oct_c_order", "ssn_a_order", "ssn_b_order", "ssn_c_order")})


if __name__ == "__main__":
    main()
