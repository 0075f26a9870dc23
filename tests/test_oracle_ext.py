"""CPU checks of the oracle's restatement of RobustOutlierFilter / GenericDescriptorOutlierFilter / force4DOF against hand-computed
values (the formulas of upstream's OutlierFiltersImpl.cpp robustFiltering and PointToPlane.cpp, as recalled: libpointmatcher is
absent, parity unpinned)."""
import math

import numpy as np
import pytest

GEN, ROB = 6, 7
FCT = {"cauchy": 0, "welsch": 1, "sc": 2, "gm": 3, "tukey": 4, "huber": 5, "L1": 6, "student": 7}


def rob(fct, tuning, scale="none", nb=0, dist="point2point", approximation=0.0):
    return (ROB, float(tuning), FCT[fct] | ({"none": 0, "mad": 1, "berg": 2, "std": 3}[scale] << 4) | ({"point2point": 0, "point2plane": 1}[dist] << 8),
            float(nb), float(approximation))


def test_robust_known_answers(oracle):
    # hand-computed weights (the formulas of upstream's robustFiltering) on three residuals, scale none
    d2 = np.array([[0.01], [0.04], [1.0]], dtype=np.float32)
    ids = np.zeros((3, 1), dtype=np.int32)
    k = np.float32(0.2); k2 = k * k
    want = {
        "cauchy": 1 / (1 + d2 / k2),
        "welsch": np.exp(-(d2 / k2).astype(np.float64)).astype(np.float32),
        "sc": np.where(d2 >= k, 4 * k2 / (k + d2) ** 2, 1),
        "gm": k2 / (k + d2) ** 2,
        "tukey": np.where(d2 >= k2, 0, (1 - d2 / k2) ** 2),
        "huber": np.where(d2 >= k2, k / np.sqrt(d2), 1),
        "L1": 1 / np.sqrt(d2),
        "student": (1 + d2 / k) ** (-(k + 3) / 2) * (k + 3) / (k + d2),
    }
    for fct, ref in want.items():
        err, w, _ = oracle.outlier_weights(oracle.make_config(outliers=[rob(fct, 0.2)]), d2, ids)
        assert err == 0
        np.testing.assert_allclose(w, ref.astype(np.float32), rtol=3e-6, err_msg=fct)
    # mad: median of {0.01, 0.04, 1.0} at rank 3 / 2 = 1 -> 0.04; |d2 - 0.04| = {0.03, 0, 0.96} -> rank 1 -> 0.03
    err, w, scale = oracle.outlier_weights(oracle.make_config(outliers=[rob("cauchy", 1.0, "mad")]), d2, ids)
    assert err == 0
    assert scale == pytest.approx(math.sqrt(0.03), rel=1e-6)
    np.testing.assert_allclose(w[:, 0], 1 / (1 + d2[:, 0] / np.float32(0.03)), rtol=1e-5)


def test_robust_scale_is_kept_after_nb_iterations(oracle):
    d2 = np.array([[0.01], [0.04], [1.0]], dtype=np.float32)
    ids = np.zeros((3, 1), dtype=np.int32)
    cfg = oracle.make_config(outliers=[rob("cauchy", 1.0, "mad", nb=2)])
    # iteration 3 > nbIterationForScale: the scale handed in stays
    err, w, scale = oracle.outlier_weights(cfg, d2, ids, iteration=3, scale=0.5)
    assert err == 0 and scale == 0.5
    np.testing.assert_allclose(w[:, 0], 1 / (1 + d2[:, 0] / np.float32(0.25)), rtol=1e-6)
    err, w, scale = oracle.outlier_weights(cfg, d2, ids, iteration=2, scale=0.5)
    assert scale == pytest.approx(math.sqrt(0.03), rel=1e-6)
    # no finite match at all: "no outlier to filter"
    err, _, _ = oracle.outlier_weights(cfg, np.full((3, 1), np.inf, dtype=np.float32), -np.ones((3, 1), dtype=np.int32))
    assert err != 0


def test_robust_berg_std_approximation_known_answers(oracle):
    """r5: hand-computed scales of the berg / std estimators and the `approximation` cut (upstream's robustFiltering as recalled)."""
    d2 = np.array([[0.01], [0.04], [0.09], [1.0], [0.0]], dtype=np.float32)
    ids = np.zeros((5, 1), dtype=np.int32)
    # berg, iteration 1: getDistsQuantile(0.5) over the positive finite {0.01, 0.04, 0.09, 1.0} -> index 4 * 0.5 = 2 -> 0.09; scale = 1.9 * 0.3
    cfg = oracle.make_config(outliers=[rob("cauchy", 0.05, "berg")])
    err, w, scale = oracle.outlier_weights(cfg, d2, ids)
    assert err == 0 and scale == pytest.approx(0.57, rel=1e-6)
    k2 = np.float32(4.3040) ** 2                                   # Bergstrom's constant, not `tuning`
    np.testing.assert_allclose(w[:, 0], 1 / (1 + (d2[:, 0] / np.float32(scale) ** 2) / k2), rtol=1e-6)
    # later iterations: scale -> 0.85 (scale - tuning) + tuning, no matter what the distances are
    err, w, s2 = oracle.outlier_weights(cfg, d2, ids, iteration=2, scale=scale)
    assert s2 == pytest.approx(0.85 * (0.57 - 0.05) + 0.05, rel=1e-6)
    err, w, s3 = oracle.outlier_weights(oracle.make_config(outliers=[rob("cauchy", 0.05, "berg", nb=2)]), d2, ids, iteration=3, scale=s2)
    assert s3 == s2                                                 # iteration > nbIterationForScale: kept
    # tukey / huber take their own constants, welsch keeps `tuning` for both roles
    for fct, kk in (("tukey", 7.0589), ("huber", 2.0138), ("welsch", 0.05)):
        err, w, scale = oracle.outlier_weights(oracle.make_config(outliers=[rob(fct, 0.05, "berg")]), d2, ids)
        e2 = d2[:, 0].astype(np.float64) / 0.57 ** 2
        ref = {"tukey": np.where(e2 >= kk * kk, 0, (1 - e2 / kk ** 2) ** 2), "huber": np.where(e2 >= kk * kk, kk / np.sqrt(np.maximum(e2, 1e-30)), 1),
               "welsch": np.exp(-e2 / kk ** 2)}[fct]
        np.testing.assert_allclose(w[:, 0], ref, rtol=2e-5, atol=1e-30, err_msg=fct)
    # std: sqrt of the sample standard deviation (ddof 1) of every entry, zeros included
    err, w, scale = oracle.outlier_weights(oracle.make_config(outliers=[rob("cauchy", 1.0, "std")]), d2, ids)
    sd = np.std(d2.astype(np.float64), ddof=1)
    assert err == 0 and scale == pytest.approx(math.sqrt(sd), rel=1e-6)
    np.testing.assert_allclose(w[:, 0], 1 / (1 + d2[:, 0] / sd), rtol=1e-5)
    # ... and an infinite entry poisons it, as it does upstream
    d2i = d2.copy(); d2i[3, 0] = np.inf
    err, w, scale = oracle.outlier_weights(oracle.make_config(outliers=[rob("cauchy", 1.0, "std")]), d2i, ids)
    assert math.isnan(scale)
    # approximation: e2 >= approximation^2 -> 0 (e2 = d2 here: scale none)
    err, w, _ = oracle.outlier_weights(oracle.make_config(outliers=[rob("cauchy", 1.0, approximation=0.3)]), d2, ids)
    assert list(w[:, 0] == 0) == [False, False, True, True, False]
    np.testing.assert_allclose(w[:2, 0], 1 / (1 + d2[:2, 0]), rtol=1e-6)


def test_robust_point2plane_residual(oracle):
    # one pair: p = (0, 0, 1), q = origin, n = (0, 0.6, 0.8): plane distance 0.8, squared 0.64 (the match distance is 1)
    step = np.array([[0, 0, 1, 1]], dtype=np.float32); ref = np.array([[0, 0, 0, 1]], dtype=np.float32)
    nn = np.array([[0, 0.6, 0.8]], dtype=np.float32)
    d2 = np.array([[1.0]], dtype=np.float32); ids = np.zeros((1, 1), dtype=np.int32)
    err, w, _ = oracle.outlier_weights(oracle.make_config(outliers=[rob("cauchy", 1.0, dist="point2plane")]), d2, ids, ref_normals=nn, step=step, ref=ref)
    assert err == 0
    assert w[0, 0] == pytest.approx(1 / (1 + 0.64), rel=1e-6)
    err, w, _ = oracle.outlier_weights(oracle.make_config(outliers=[rob("cauchy", 1.0)]), d2, ids)
    assert w[0, 0] == pytest.approx(0.5, rel=1e-6)


def test_generic_descriptor_known_answers(oracle):
    d2 = np.full((4, 1), 0.1, dtype=np.float32)
    ids = np.array([[0], [1], [2], [-1]], dtype=np.int32)
    s = np.array([0.2, 0.5, 0.9], dtype=np.float32)
    for flags, want in ((4, [0, 0, 1, 0]), (0, [1, 0, 0, 0]), (2, [0.2, 0.5, 0.9, 0]), (6, [0.2, 0.5, 0.9, 0])):
        err, w, _ = oracle.outlier_weights(oracle.make_config(outliers=[(GEN, 0.5, flags, 0.0)]), d2, ids, ref_scalar=s)
        assert err == 0
        np.testing.assert_array_equal(w[:, 0], np.array(want, dtype=np.float32))
    err, _, _ = oracle.outlier_weights(oracle.make_config(outliers=[(GEN, 0.5, 1, 0.0)]), d2, ids, ref_scalar=s)
    assert err != 0  # source: reading without the reading's row
    rs = np.array([0.9, 0.1, 0.6, 0.7], dtype=np.float32)   # ... with it: the READING point's descriptor decides (entry 3 is unmatched)
    for flags, want in ((1 | 4, [1, 0, 1, 0]), (1, [0, 1, 0, 0]), (1 | 2, [0.9, 0.1, 0.6, 0])):
        err, w, _ = oracle.outlier_weights(oracle.make_config(outliers=[(GEN, 0.5, flags, 0.0)]), d2, ids, read_scalar=rs)
        assert err == 0
        np.testing.assert_array_equal(w[:, 0], np.array(want, dtype=np.float32))


def test_force_4dof_recovers_a_yaw_and_ignores_tilt(oracle):
    rng = np.random.default_rng(3)
    n = 4000
    ref = np.c_[rng.uniform(-5, 5, (n, 3)), np.ones(n)].astype(np.float32)
    nn = rng.normal(size=(n, 3)); nn /= np.linalg.norm(nn, axis=1, keepdims=True); nn = nn.astype(np.float32)
    yaw, t = 0.01, np.array([0.02, -0.01, 0.03])
    c, s = math.cos(-yaw), math.sin(-yaw)
    Rinv = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    reading = ref.copy(); reading[:, :3] = ((ref[:, :3] - t) @ Rinv.T).astype(np.float32)
    ids = np.arange(n, dtype=np.int32)[:, None]; d2 = np.ones((n, 1), dtype=np.float32); w = np.ones((n, 1), dtype=np.float32)
    err, T, A, b, x, st = oracle.minimize(2, reading, ref, nn, ids, d2, w, force_4dof=1)
    assert err == 0 and x[0] == 0 and x[1] == 0
    assert x[2] == pytest.approx(yaw, abs=2e-4)
    np.testing.assert_allclose(T[:3, 3], t, atol=3e-4)
    np.testing.assert_allclose(T[2, :3], [0, 0, 1], atol=1e-7)
    # the 6-DOF solve of a tilted reading finds the tilt, the 4-DOF one cannot (x0 = x1 = 0 by construction)
    tilt = np.array([[1, 0, 0], [0, math.cos(0.01), -math.sin(0.01)], [0, math.sin(0.01), math.cos(0.01)]])
    reading2 = ref.copy(); reading2[:, :3] = (ref[:, :3] @ tilt).astype(np.float32)
    _, _, _, _, x6, _ = oracle.minimize(2, reading2, ref, nn, ids, d2, w)
    _, T4, _, _, x4, _ = oracle.minimize(2, reading2, ref, nn, ids, d2, w, force_4dof=1)
    assert abs(x6[0]) > 5e-3 and x4[0] == 0
    np.testing.assert_allclose(T4[2, :3], [0, 0, 1], atol=1e-7)


def test_solve_n_matches_numpy_and_handles_rank_deficiency(oracle):
    rng = np.random.default_rng(0)
    for n in (4, 6):
        M = rng.normal(size=(n + 3, n)); A = (M.T @ M).astype(np.float32); b = rng.normal(size=n).astype(np.float32)
        np.testing.assert_allclose(oracle.solve_n(A, b), np.linalg.solve(A.astype(np.float64), b), rtol=2e-3, atol=1e-5)
        v = rng.normal(size=(n, 2)); A = (v @ v.T).astype(np.float32)   # rank 2: the minimum-norm solution
        b = (A @ rng.normal(size=n)).astype(np.float32)
        np.testing.assert_allclose(oracle.solve_n(A, b), np.linalg.pinv(A.astype(np.float64), rcond=1e-6) @ b, rtol=1e-3, atol=1e-4)


def test_rotation_from_H_polar_route_equals_the_svd_route(oracle):
    """the Newton polar iteration and the Jacobi SVD reach the same U V^T; singular and reflecting H take the SVD route"""
    rng = np.random.default_rng(11)
    def rot():
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(Q) < 0: Q[:, 0] *= -1
        return Q
    for kappa, tol in ((1, 3e-7), (30, 6e-7), (1e3, 2e-6), (1e4, 5e-6)):
        for _ in range(100):
            U, V = rot(), rot()
            S = np.array([1.0, rng.uniform(1 / kappa, 1), 1 / kappa]) * 10.0 ** rng.uniform(-3, 9)
            H = ((U * S) @ V.T).astype(np.float32)
            Ud, _, Vtd = np.linalg.svd(H.astype(np.float64))
            want = Ud @ Vtd
            R = oracle.rotation_from_H(H)
            assert np.abs(R - want).max() < tol, (kappa, np.abs(R - want).max())
            assert np.abs(oracle.rotation_from_H(H, svd=True) - want).max() < 4 * tol
            assert np.abs(R.astype(np.float64) @ R.T - np.eye(3)).max() < 1e-6
    # reflection (det H < 0), rank 2, rank 1, zero: the SVD route, bit for bit
    U, V = rot(), rot()
    for S in ([3, 2, -1], [3, 2, 0], [3, 0, 0], [0, 0, 0], [1, 1, 1e-8]):
        H = ((U * np.array(S, dtype=np.float64)) @ V.T).astype(np.float32)
        assert np.array_equal(oracle.rotation_from_H(H), oracle.rotation_from_H(H, svd=True))
        R = oracle.rotation_from_H(H)
        assert np.linalg.det(R.astype(np.float64)) > 0.999


def test_step_angle_sincos_within_an_ulp(oracle):
    """the operation-by-operation sin / cos of the point-to-plane step angle (shared bit for bit with the device)"""
    xs = np.concatenate([np.linspace(0, 0.5, 20001)[:-1], [1e-8, 1e-5, 0.49999997, 0.5, 0.7, 1.5, 3.0]]).astype(np.float32)
    for x in xs:
        s, c = oracle.sincos_f(x)
        ts, tc = math.sin(float(x)), math.cos(float(x))
        assert abs(s - ts) <= 0.75 * np.spacing(np.float32(abs(ts))) + 1e-45
        assert abs(c - tc) <= 0.75 * np.spacing(np.float32(abs(tc)))
    assert oracle.sincos_f(0.0) == (0.0, 1.0)


def test_oracle_normals_are_smallest_eigenvectors_everywhere(oracle):
    from norlab_icp_mapper_amd import synth
    pts = synth.make_scene(m=20000, n=16)["map"]
    rn = oracle.surface_normals(pts, knn=10, nthreads=4)
    ids, _ = oracle.knn(pts, pts, k=10, nthreads=4)
    ok = oracle.check_normals_are_smallest_eigenvectors(pts, ids, rn, "oracle")
    assert ok.mean() > 0.999


def test_var_trimmed_ratio_known_answers(oracle):
    """optimizeInlierRatio against a direct numpy evaluation of FRMS over the candidate ranks"""
    rng = np.random.default_rng(21)
    def ref(d2, lo, hi, lam):
        N = d2.size
        v = np.sort(d2[np.isfinite(d2) & (d2 > 0)].astype(np.float64))
        cum = np.cumsum(v)
        a, b = int(math.floor(np.float32(lo) * np.float32(N))), min(int(math.floor(np.float32(hi) * np.float32(N))), v.size)
        if b <= a:
            return np.float32(a) / np.float32(N)
        i = np.arange(a, b)
        frms = cum[a:b] / ((i + 1) * ((i + 1) / N) ** (2 * lam))
        return np.float32(a + int(np.argmin(frms))) / np.float32(N)
    # inliers ~ small residuals, 30 % outliers far away: with lambda above 1 the FRMS minimum sits at the inlier fraction (the
    # mean of the smallest fraction f of Gaussian squared residuals grows like f^2, so lambda <= 1 favours the smallest f)
    d2 = np.concatenate([rng.normal(0, 0.02, 7000) ** 2, rng.uniform(0.5, 4.0, 3000)]).astype(np.float32)
    rng.shuffle(d2)
    r = oracle.var_trimmed_ratio(d2, 0.05, 0.99, 0.95)
    assert r == ref(d2, 0.05, 0.99, 0.95)
    r13 = oracle.var_trimmed_ratio(d2, 0.05, 0.99, 1.3)
    assert r13 == ref(d2, 0.05, 0.99, 1.3) and 0.4 < r13 <= 0.7001      # never past the inlier fraction
    r3 = oracle.var_trimmed_ratio(d2, 0.05, 0.99, 3.0)
    assert r3 == ref(d2, 0.05, 0.99, 3.0) and 0.69 < r3 <= 0.7001
    # unmatched entries count in N but are no candidates; all-invalid -> -1
    d2b = d2.copy(); d2b[::5] = np.inf; d2b[1::50] = 0.0
    assert oracle.var_trimmed_ratio(d2b, 0.05, 0.99, 0.95) == ref(d2b, 0.05, 0.99, 0.95)
    assert oracle.var_trimmed_ratio(np.full(10, np.inf, np.float32)) < 0
    for lo, hi, lam in ((0.3, 0.5, 0.95), (0.9, 0.99, 0.1), (0.05, 0.99, 3.0), (0.999, 0.9995, 1.0)):
        assert oracle.var_trimmed_ratio(d2b, lo, hi, lam) == ref(d2b, lo, hi, lam), (lo, hi, lam)
    # the filter = TrimmedDist at that ratio
    ids = np.zeros((d2.size, 1), dtype=np.int32)
    err, w, lim = oracle.outlier_weights(oracle.make_config(outliers=[(8, 0.05, 0, 0.99, 0.95)]), d2.reshape(-1, 1), ids)
    assert err == 0 and lim == oracle.dists_quantile(d2, r)
    np.testing.assert_array_equal(w[:, 0], (d2 <= lim).astype(np.float32))


def test_force_2d_is_the_planar_system_of_upstream(oracle):
    """force2D on 3-D clouds: F = [x ny - y nx; nx; ny], residual (dx nx + dy ny), x = (yaw, tx, ty) -- against numpy"""
    rng = np.random.default_rng(4)
    n = 3000
    ref = np.c_[rng.uniform(-5, 5, (n, 3)), np.ones(n)].astype(np.float32)
    nn = rng.normal(size=(n, 3)); nn /= np.linalg.norm(nn, axis=1, keepdims=True); nn = nn.astype(np.float32)
    yaw, t = 0.012, np.array([0.03, -0.02, 0.5])          # a z offset the planar solve must not see
    c, s = math.cos(-yaw), math.sin(-yaw)
    Rinv = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    reading = ref.copy(); reading[:, :3] = ((ref[:, :3] - t) @ Rinv.T).astype(np.float32)
    ids = np.arange(n, dtype=np.int32)[:, None]; d2 = np.ones((n, 1), dtype=np.float32)
    w = rng.uniform(0.2, 1.0, (n, 1)).astype(np.float32)
    err, T, A, b, x, st = oracle.minimize(2, reading, ref, nn, ids, d2, w, force_4dof=2)
    assert err == 0 and x[0] == 0 and x[1] == 0 and x[5] == 0
    p, q, m = reading[:, :3].astype(np.float64), ref[:, :3].astype(np.float64), nn.astype(np.float64)
    F = np.c_[p[:, 0] * m[:, 1] - p[:, 1] * m[:, 0], m[:, 0], m[:, 1]]
    dot2 = (p[:, 0] - q[:, 0]) * m[:, 0] + (p[:, 1] - q[:, 1]) * m[:, 1]
    A3 = (F * w.astype(np.float64)).T @ F
    b3 = -(F * w.astype(np.float64)).T @ dot2
    np.testing.assert_allclose(x[2:5], np.linalg.solve(A3, b3), rtol=2e-4, atol=2e-6)
    assert x[2] == pytest.approx(yaw, abs=3e-4)
    np.testing.assert_allclose(T[:2, 3], t[:2], atol=5e-4)
    assert T[2, 3] == 0 and np.allclose(T[2, :3], [0, 0, 1], atol=1e-7) and np.allclose(T[:3, 2], [0, 0, 1], atol=1e-7)


def _planar_scene(n_map=4000, n_scan=1500, seed=9):
    """a room outline with two round pillars, z = 0; the scan is the same outline seen from a moved pose, with noise"""
    rng = np.random.default_rng(seed)
    def outline(n):
        t = rng.random(n)
        side = rng.integers(0, 6, n)
        x = np.where(side == 0, -6 + 12 * t, np.where(side == 1, 6.0, np.where(side == 2, 6 - 12 * t, np.where(side == 3, -6.0, 0.0))))
        y = np.where(side == 0, -4.0, np.where(side == 1, -4 + 8 * t, np.where(side == 2, 4.0, np.where(side == 3, 4 - 8 * t, 0.0))))
        a = 2 * np.pi * t
        x = np.where(side == 4, 2.0 + 0.4 * np.cos(a), np.where(side == 5, -3.0 + 0.6 * np.cos(a), x))
        y = np.where(side == 4, 1.0 + 0.4 * np.sin(a), np.where(side == 5, -1.5 + 0.6 * np.sin(a), y))
        return np.c_[x, y]
    mp = outline(n_map) + rng.normal(0, 0.005, (n_map, 2))
    yaw, t = 0.03, np.array([0.08, -0.05])
    c, s = math.cos(yaw), math.sin(yaw)
    R = np.array([[c, -s], [s, c]])
    sc = (outline(n_scan) + rng.normal(0, 0.005, (n_scan, 2)) - t) @ R          # = R^T (p - t): the scan in the moved sensor frame
    to4 = lambda p: np.c_[p, np.zeros(len(p)), np.ones(len(p))].astype(np.float32)
    T = np.eye(4); T[:2, :2] = R; T[:2, 3] = t
    return to4(mp), to4(sc), T


def test_planar_mode_normals_and_registration(oracle):
    from norlab_icp_mapper_amd import synth
    mp, sc, T_gt = _planar_scene()
    nrm = oracle.surface_normals(mp, knn=8, nthreads=4, planar=True)
    assert np.all(nrm[:, 2] == 0) and np.allclose(np.linalg.norm(nrm, axis=1), 1, atol=1e-6)
    wall = (np.abs(mp[:, 1] + 4) < 0.02) & (np.abs(mp[:, 0]) < 5)                      # the y = -4 wall: normal along y
    assert np.median(np.abs(nrm[wall, 1])) > 0.995 and np.quantile(np.abs(nrm[wall, 1]), 0.05) > 0.9     # 5 mm noise over ~4 cm neighbourhoods
    # the 3-D filter on the same cloud can only answer z (the zero eigenvalue): that is why planar clouds need their own
    assert np.abs(oracle.surface_normals(mp, knn=8, nthreads=4)[wall, 2]).min() > 0.99
    for minimizer in (1, 2):
        o = oracle.OracleICP(oracle.make_config(minimizer=minimizer, max_dist=1.0, outliers=[(4, 0.9)], max_iterations=40, use_differential=1,
                                                nthreads=4, is_2d=1))
        o.setMap(mp, nrm)
        err, T = o(sc)
        assert err == 0
        dt, dr = synth.pose_error(T, T_gt)
        assert dt < (1e-2 if minimizer == 1 else 5e-3) and dr < (6e-3 if minimizer == 1 else 2e-3), (minimizer, dt, dr)
        assert T[2, 3] == 0 and np.allclose(T[2, :3], [0, 0, 1], atol=1e-7) and np.allclose(T[:3, 2], [0, 0, 1], atol=1e-7)


def test_sensor_noise_overlap_known_answer(oracle):
    """getOverlap() with `simpleSensorNoise` + `normals` on the reading, by hand: an identity registration (the reading IS a subset of
    the map, IdentityErrorMinimizer is not involved -- point-to-plane on exact matches solves to identity) of points lifted off a plane."""
    rng = np.random.default_rng(3)
    g = np.stack(np.meshgrid(np.arange(20.0), np.arange(20.0)), -1).reshape(-1, 2)
    m = np.ones((g.shape[0], 4), np.float32); m[:, :2] = g; m[:, 2] = 0
    mn = np.tile(np.array([[0, 0, 1]], np.float32), (m.shape[0], 1))
    scan = m[::2].copy()
    lift = rng.uniform(0.0, 0.02, scan.shape[0]).astype(np.float32)
    scan[:, 2] += lift                                               # distance to the match = lift, along the normal
    noise = np.full(scan.shape[0], 0.01, np.float32)
    nrm = np.tile(np.array([[0, 0, 2]], np.float32), (scan.shape[0], 1))  # un-normalised on purpose: upstream normalises
    o = oracle.OracleICP(oracle.make_config(minimizer=2, max_dist=1.0, outliers=[], max_iterations=1))
    o.setMap(m, mn)
    o.setReadingNoise(noise)
    err, _ = o(scan, nrm)
    assert err == 0 and o.stats.pairs == scan.shape[0]
    assert o.stats.sensor_noise_overlap == np.float32((lift < 0.01).sum() / scan.shape[0])
    # point-to-point: dist < mean(dist) + noise
    o = oracle.OracleICP(oracle.make_config(minimizer=1, max_dist=1.0, outliers=[], max_iterations=1))
    o.setMap(m, mn)
    o.setReadingNoise(noise)
    err, _ = o(scan, nrm)
    assert err == 0
    assert o.stats.sensor_noise_overlap == np.float32((lift < np.float32(lift.astype(np.float64).mean()) + noise).sum() / scan.shape[0])
    # no noise handed over -> -1 (getOverlap() falls back to the weighted ratio)
    err, _ = o(scan, nrm)
    assert o.stats.sensor_noise_overlap == -1.0


def test_surface_normal_matched_ids_and_mean_dist_match_ckdtree(oracle):
    """keepMatchedIds / keepMeanDist (SurfaceNormalDataPointsFilter as recalled): the ids are the point's own kNN set (self first),
    the mean distance is |p - mean(neighbours)| -- pinned against scipy's cKDTree, independent of the oracle's own k-d tree."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(77)
    pts = np.ones((4000, 4), dtype=np.float32)
    pts[:, :3] = rng.uniform(-10, 10, (4000, 3)).astype(np.float32)
    k = 7
    n, ids, md = oracle.surface_normals_extras(pts, knn=k)
    P = pts[:, :3].astype(np.float64)
    _, ref = cKDTree(P).query(P, k=k)
    assert np.array_equal(ids[:, 0], np.arange(4000))
    assert np.array_equal(np.sort(ids, axis=1), np.sort(ref, axis=1))
    want = np.linalg.norm(P - P[ref].mean(axis=1), axis=1)
    np.testing.assert_allclose(md, want, rtol=1e-5, atol=1e-6)
    # the extras do not disturb the normals, and they are one-shot (the next call runs without them)
    n2 = oracle.surface_normals(pts, knn=k)
    assert np.array_equal(n, n2)
