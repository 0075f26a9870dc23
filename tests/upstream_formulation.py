"""libpointmatcher's OWN formulations of the registration loop, written independently of oracle/icp_oracle.c in numpy float32 (TEST INFRASTRUCTURE).

The reference delegates all arithmetic to libpointmatcher 1.4.x / libnabo / Eigen, none of which exist in this image, so the C oracle is a
restatement -- and in a handful of places it knowingly leaves the way upstream computes a quantity (oracle/DEVIATIONS.md lists them).  This
module states those quantities the way upstream does, as recalled from its sources, so that tests/test_oracle_deviations.py
can put a NUMBER on every deviation: same inputs through both formulations, difference bounded at the level that matters (the pose).

What is deliberately the same on both sides: the correspondences (exact k-nearest neighbours with ties on the smaller index -- libnabo's result
set for epsilon 0; taken from the oracle's kd-tree, which is itself pinned against brute force and scipy's cKDTree in tests/test_oracle_golden.py)
and the quantile rule of the Trimmed / Median filters (Matches::getDistsQuantile, pinned separately in tests/test_oracle_recalled.py).

Upstream formulations restated here (all in float32, as `PointMatcher<float>`):
  * ICPSequence::setMap / operator():   the reference is centred on its ROW-WISE MEAN formed in float (Eigen `rowwise().sum() / cols`);
  * PointToPointErrorMinimizer:         weighted means, centred clouds, H = ref_c diag(w) read_c^T as float products, JacobiSVD -> R = U V^T with
                                        the reflection repair on V's last row, t = mean_ref - R mean_read  (numpy.linalg.svd on the float32 H);
  * PointToPlaneErrorMinimizer:         F = [cross; n], A = wF F^T and b = -(wF dot^T) as float matrix products; solvePossiblyUnderdeterminedLinearSystem:
                                        fullPivHouseholderQr().isInvertible() (pivots above 6 eps of the largest) -> LLT, else the minimum-norm solution;
                                        the step is AngleAxis(|x_rot|, x_rot / |x_rot|) + translation;
  * DifferentialTransformationChecker:  quaternion angular distance / translation distance of the last smoothLength + 1 poses, means compared
                                        with the limits -- in float;
  * VarTrimmedDistOutlierFilter:        running sum of the sorted distances and FRMS in float.
"""
import numpy as np

import oracle_bindings as ob

F32 = np.float32   # storage and per-element arithmetic: `PointMatcher<float>` -- always float32
ACC = np.float32   # the width of the SUMS over pairs (means, H, A, b): float32 = what Eigen's products do; `precision(np.float64)`: exact sums


class precision:
    """with precision(np.float64): ... -- upstream's formulas with their sums over pairs carried exactly (everything else stays float32):
    the value every float32 summation order of those sums scatters around"""
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global ACC
        self.prev, ACC = ACC, self.dtype

    def __exit__(self, *a):
        global ACC
        ACC = self.prev



def rowwise_mean_f32(xyz):
    """Eigen's `features.rowwise().sum() / cols` in float: numpy's float32 add-reduce (pairwise, like Eigen's packet reduction it is a
    float32 summation in SOME order; the ledger bounds the effect of the order, not one particular order)"""
    return (np.add.reduce(xyz.astype(F32), axis=0, dtype=F32) / F32(xyz.shape[0])).astype(F32)


def rowwise_mean_f32_sequential(xyz):
    """the worst float32 order: one accumulator, left to right"""
    return (np.cumsum(xyz.astype(F32), axis=0, dtype=F32)[-1] / F32(xyz.shape[0])).astype(F32)


def rotation_jacobi_svd(H):
    """PointToPoint.cpp: JacobiSVD(H, ComputeThinU | ComputeThinV); R = U V^T; det < 0 -> V^T's last row negated"""
    U, _, Vt = np.linalg.svd(np.asarray(H, dtype=F32))
    R = (U @ Vt).astype(F32)
    if np.linalg.det(R.astype(np.float64)) < 0:
        Vt = Vt.copy(); Vt[-1, :] *= -1
        R = (U @ Vt).astype(F32)
    return R


def qr_is_invertible(A):
    """FullPivHouseholderQR::isInvertible(): every pivot above threshold * |largest pivot|, threshold = epsilon * diagonalSize (Eigen's default)"""
    from scipy.linalg import qr
    _, R, _ = qr(np.asarray(A, dtype=F32), pivoting=True)
    d = np.abs(np.diag(R))
    return bool(np.all(d > F32(np.finfo(F32).eps * A.shape[0]) * d.max())), d


def solve_possibly_underdetermined(A, b):
    """ErrorMinimizersImpl.cpp solvePossiblyUnderdeterminedLinearSystem: LLT when invertible, else the minimum-norm solution"""
    A = np.asarray(A, dtype=F32); b = np.asarray(b, dtype=F32)
    ok, _ = qr_is_invertible(A)
    if ok:
        L = np.linalg.cholesky(A)
        y = np.linalg.solve(L, b)
        return np.linalg.solve(L.T, y).astype(F32), True
    return (np.linalg.pinv(A.astype(np.float64)) @ b.astype(np.float64)).astype(F32), False


def angle_axis_T(x):
    """Eigen::AngleAxis<float>(x.head(3).norm(), x.head(3).normalized()) + translation x.segment(3, 3)"""
    x = np.asarray(x, dtype=F32)
    T = np.eye(4, dtype=F32)
    th = F32(np.sqrt(np.dot(x[:3], x[:3])))
    if th > 0:
        k = (x[:3] / th).astype(F32)
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=F32)
        T[:3, :3] = (np.eye(3, dtype=F32) + F32(np.sin(th)) * K + F32(1 - np.cos(th)) * (K @ K)).astype(F32)
    T[:3, 3] = x[3:6]
    return T


def minimize_point_to_point(read, ref, ids, w):
    """read: (n, 3) float32 step reading; ref: (m, 3); ids / w: (n, k) with w = 0 for rejected pairs"""
    sel = w > 0
    rows = np.nonzero(sel)[0]
    p = read[rows].astype(F32); q = ref[ids[sel]].astype(F32); ww = w[sel].astype(F32)
    wsi = ACC(1.0) / np.add.reduce(ww.astype(ACC), dtype=ACC)
    mp = (np.add.reduce((p * ww[:, None]).astype(ACC), axis=0, dtype=ACC) * wsi).astype(F32)
    mq = (np.add.reduce((q * ww[:, None]).astype(ACC), axis=0, dtype=ACC) * wsi).astype(F32)
    pc = (p - mp).astype(F32); qc = (q - mq).astype(F32)
    H = ((qc.T * ww).astype(ACC) @ pc.astype(ACC)).astype(F32)   # ref_c diag(w) read_c^T
    R = rotation_jacobi_svd(H)
    T = np.eye(4, dtype=F32)
    T[:3, :3] = R
    T[:3, 3] = (mq - R @ mp).astype(F32)
    return T, H


def minimize_point_to_plane(read, ref, ref_normals, ids, w):
    sel = w > 0
    rows = np.nonzero(sel)[0]
    p = read[rows].astype(F32); q = ref[ids[sel]].astype(F32); nn = ref_normals[ids[sel]].astype(F32); ww = w[sel].astype(F32)
    cross = np.cross(p, nn).astype(F32)
    F = np.concatenate([cross, nn], axis=1).T.astype(F32)            # 6 x P
    wF = (F * ww).astype(F32)
    A = (wF.astype(ACC) @ F.T.astype(ACC)).astype(F32)
    dot = np.add.reduce(((p - q).astype(F32) * nn), axis=1, dtype=F32)
    b = (-(wF.astype(ACC) @ dot.astype(ACC))).astype(F32)
    x, invertible = solve_possibly_underdetermined(A, b)
    return angle_axis_T(x), A, b, x, invertible


def quat_from_R(R):
    """Eigen::Quaternion<float>(Matrix3f): the branch on the trace"""
    R = R.astype(np.float64)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = np.sqrt(t + 1.0); w = 0.5 * s; s = 0.5 / s
        return np.array([w, (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s], dtype=F32)
    i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]])); j = (i + 1) % 3; k = (j + 1) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    v = np.zeros(3); v[i] = 0.5 * s; s = 0.5 / s
    w = (R[k, j] - R[j, k]) * s; v[j] = (R[j, i] + R[i, j]) * s; v[k] = (R[k, i] + R[i, k]) * s
    return np.array([w, v[0], v[1], v[2]], dtype=F32)


class DifferentialCheckerF32:
    """TransformationCheckersImpl.cpp DifferentialTransformationChecker in float: rotations as quaternions, `angularDistance`, translations;
    once smoothLength + 1 poses are in, the means of the last smoothLength differences against the two limits"""
    def __init__(self, min_rot=1e-3, min_trans=1e-3, smooth=3):
        self.min_rot, self.min_trans, self.smooth = F32(min_rot), F32(min_trans), smooth
        self.q, self.t = [], []

    def init(self, T):
        self.q, self.t = [quat_from_R(T[:3, :3])], [T[:3, 3].astype(F32)]

    def check(self, T):
        """returns True while the loop should go on"""
        self.q.append(quat_from_R(T[:3, :3])); self.t.append(T[:3, 3].astype(F32))
        if len(self.q) <= self.smooth:
            return True
        rot = F32(0); tr = F32(0)
        for i in range(len(self.q) - self.smooth, len(self.q)):
            d = F32(abs(float(np.dot(self.q[i].astype(np.float64), self.q[i - 1].astype(np.float64)))))
            rot += F32(2.0) * F32(np.arccos(min(F32(1.0), d)))      # Quaternion::angularDistance
            tr += F32(np.linalg.norm((self.t[i] - self.t[i - 1]).astype(F32)))
        rot /= F32(self.smooth); tr /= F32(self.smooth)
        return not (rot < self.min_rot and tr < self.min_trans)


def dists_quantile(d2, q):
    """Matches::getDistsQuantile (pinned separately): the finite, positive entries, rank int(n * q), the maximum for q == 1"""
    v = d2[np.isfinite(d2) & (d2 > 0)].astype(F32)
    if v.size == 0:
        return None
    if q >= 1.0:
        return F32(v.max())
    return F32(np.partition(v, int(F32(v.size) * F32(q)))[int(F32(v.size) * F32(q))])


def var_trimmed_ratio_f32(d2, min_ratio=0.05, max_ratio=0.99, lam=0.95):
    """VarTrimmedDistOutlierFilter::optimizeInlierRatio with upstream's float running sum and float FRMS"""
    count = d2.size
    v = np.sort(d2[np.isfinite(d2) & (d2 > 0)].astype(F32))
    if v.size == 0:
        return -1.0
    min_el, max_el = int(np.floor(F32(min_ratio) * F32(count))), int(np.floor(F32(max_ratio) * F32(count)))
    hi = min(max_el, v.size)
    cum = np.cumsum(v[:hi], dtype=F32)
    ids = np.arange(1, hi + 1, dtype=F32)
    frms = cum / (ids * np.power(ids / F32(count), F32(2.0) * F32(lam), dtype=F32))
    if hi <= min_el:
        return float(F32(min_el) / F32(count))
    i = int(np.argmin(frms[min_el:hi])) + min_el
    return float(F32(i) / F32(count))


def icp(map4, normals, scan4, minimizer, knn=1, max_dist=2.0, trimmed=0.85, max_iterations=40, differential=True, nthreads=8,
        mean_fn=rowwise_mean_f32, pair_order=None):
    """PM::ICPSequence::operator() (upstream ICP.cpp computeWithTransformedReference), float32 throughout.
    Returns (T 4 x 4 float32, iterations, per-iteration records)."""
    xyz = map4[:, :3].astype(F32)
    mean = mean_fn(xyz)
    ref = (xyz - mean).astype(F32)
    T_ref = np.eye(4, dtype=F32); T_ref[:3, 3] = mean
    read0 = (scan4[:, :3].astype(F32) - mean).astype(F32)        # reading through T_refIn_refMean^-1
    ref4 = np.ones((ref.shape[0], 4), F32); ref4[:, :3] = ref
    T = np.eye(4, dtype=F32)
    chk = DifferentialCheckerF32()
    chk.init(T)
    rec = []
    it = 0
    while it < max_iterations:
        step = (read0 @ T[:3, :3].T + T[:3, 3]).astype(F32)
        step4 = np.ones((step.shape[0], 4), np.float32); step4[:, :3] = step
        ids, d2 = ob.knn(ref4.astype(np.float32), step4, k=knn, max_dist=max_dist, nthreads=nthreads)
        w = (np.isfinite(d2)).astype(F32)
        limit = None
        if trimmed is not None:
            limit = dists_quantile(d2, trimmed)
            if limit is not None:
                w = w * (d2 <= limit).astype(F32)
        ids_safe = np.where(ids < 0, 0, ids)
        if pair_order is not None:   # the pairs in another order: another float32 summation order of the same sums
            perm = np.random.default_rng(pair_order).permutation(step.shape[0])
            step_m, ids_m, w_m = step[perm], ids_safe[perm], w[perm]
        else:
            step_m, ids_m, w_m = step, ids_safe, w
        if minimizer == 1:
            Ts, H = minimize_point_to_point(step_m, ref, ids_m, w_m)
            rec.append(dict(H=H, limit=limit, pairs=int((w > 0).sum())))
        else:
            Ts, A, b, x, inv = minimize_point_to_plane(step_m, ref, normals.astype(F32), ids_m, w_m)
            rec.append(dict(A=A, b=b, x=x, invertible=inv, limit=limit, pairs=int((w > 0).sum())))
        T = (Ts @ T).astype(F32)
        it += 1
        if differential and not chk.check(T):
            break
    Tinv = np.eye(4, dtype=F32); Tinv[:3, 3] = -mean
    return (T_ref @ T @ Tinv).astype(F32), it, rec
