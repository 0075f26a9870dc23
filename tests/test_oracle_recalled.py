"""The oracle against tests/golden/numpy_recalled_vectors.npz (numpy / scipy / pure-Python, tests/golden/make_recalled.py): the part of
the restatement that follows upstream "as recalled" -- Robust, VarTrimmedDist, SurfaceNormalOutlierFilter, the octree, SamplingSurfaceNormal,
the Differential / Bound checkers -- and DynamicPointsMapperModule, whose vectors transliterate the reference's own source
(DynamicPointsMapperModule.cpp:34-172) line by line.  VERDICT r3 missing 1.  tests/test_gpu_recalled.py holds the HIP path to the same file."""
import math
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROB, VT, SNO = 7, 8, 5
FCT = {"cauchy": 0, "welsch": 1, "sc": 2, "gm": 3, "tukey": 4, "huber": 5, "L1": 6, "student": 7}


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "numpy_recalled_vectors.npz"))


def robust_chain(name, tuning, scale):
    return [(ROB, float(tuning), FCT[name] | ({"none": 0, "mad": 1}[scale] << 4), 0.0)]


@pytest.mark.parametrize("scale", ["none", "mad"])
def test_robust_weights_match_closed_forms(oracle, gold, scale):
    d2 = gold["rob_d2"].reshape(-1, 1)
    ids = np.zeros(d2.shape, dtype=np.int32)
    for name, k in zip(gold["rob_names"], gold["rob_tuning"]):
        err, w, lim = oracle.outlier_weights(oracle.make_config(outliers=robust_chain(str(name), k, scale)), d2, ids)
        assert err == 0
        if scale == "mad":
            assert lim == pytest.approx(float(gold["rob_mad_scale"]), rel=1e-6)
        np.testing.assert_allclose(w[:, 0], gold[f"rob_w_{name}_{scale}"], rtol=3e-5, atol=1e-7, err_msg=str(name))


def test_var_trimmed_matches_brute_force(oracle, gold):
    for c in gold["vt_cases"]:
        d2 = gold[f"vt{c}_d2"].reshape(-1, 1)
        minr, maxr, lam = gold[f"vt{c}_prm"]
        assert oracle.var_trimmed_ratio(d2, minr, maxr, lam) == np.float32(gold[f"vt{c}_ratio"])
        err, w, lim = oracle.outlier_weights(oracle.make_config(outliers=[(VT, float(minr), 0, float(maxr), float(lam))]), d2, np.zeros(d2.shape, np.int32))
        assert err == 0 and np.float32(lim) == gold[f"vt{c}_limit"]
        assert np.array_equal(w[:, 0], (d2[:, 0] <= gold[f"vt{c}_limit"]).astype(np.float32))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_surface_normal_outlier_matches_numpy(oracle, gold, tag):
    ids = gold["sno_ids"]
    d2 = np.where(ids >= 0, np.float32(0.01), np.float32(np.inf)).astype(np.float32)
    err, w, _ = oracle.outlier_weights(oracle.make_config(outliers=[(SNO, float(gold[f"sno_{tag}_angle"]))]), d2, ids,
                                       read_normals=gold["sno_read_n"], ref_normals=gold["sno_ref_n"])
    assert err == 0
    sure = gold[f"sno_{tag}_sure"]
    assert sure.mean() > 0.99 and np.array_equal(w[sure], gold[f"sno_{tag}_w"][sure])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_octree_matches_python_recursion(oracle, gold, tag):
    ms, mp = gold[f"oct_{tag}_prm"]
    assert np.array_equal(oracle.octree_sample(gold["oct_pts"], float(ms), int(mp), 0), gold[f"oct_{tag}_order"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_sampling_surface_normal_matches_python_recursion(oracle, gold, tag):
    ratio, knn, mb, seed = gold[f"ssn_{tag}_prm"]
    order, nrm = oracle.sampling_surface_normal(gold["ssn_pts"], float(ratio), int(knn), float(mb), int(seed))
    assert np.array_equal(order, gold[f"ssn_{tag}_order"])
    dots = np.abs(np.einsum("ij,ij->i", nrm.astype(np.float64), gold[f"ssn_{tag}_normals"]))   # eigenvectors: sign free
    assert dots.min() > 1 - 1e-5


def test_dynamic_points_match_reference_transliteration(oracle, gold):
    pose = gold["dyn_pose"].astype(np.float64)
    prm = dict(zip(("threshold_dynamic", "alpha", "beta", "beam_half_angle", "epsilon_a", "epsilon_d", "sensor_max_range"), map(float, gold["dyn_prm"])))
    got = oracle.dynamic_points_update(np.linalg.inv(pose), gold["dyn_input"], gold["dyn_map"], gold["dyn_normals"], gold["dyn_prob"], **prm)
    sure = gold["dyn_sure"]
    assert sure.mean() > 0.8
    np.testing.assert_allclose(got[sure], gold["dyn_expected"][sure], rtol=0, atol=2e-4)
    changed = gold["dyn_expected"] != gold["dyn_prob"]
    assert (changed & sure).sum() > 1000                      # the update did something on most points
    assert (np.abs(got - gold["dyn_expected"]) < 2e-4).mean() > 0.995   # and even the borderline ones almost always agree


# ---- Differential / Bound checkers against scipy.spatial.transform.Rotation --------------------------------------

def checker_series(oracle, sc, kw, iters):
    """pose after 1 .. iters iterations (Counter only) in the matcher's centred frame, by re-running the registration"""
    series = [np.eye(4)]
    mean = None
    for it in range(1, iters + 1):
        o = oracle.OracleICP(oracle.make_config(nthreads=8, max_iterations=it, **kw))
        o.setMap(sc["map"], sc["normals"])
        mean = o.getMapMean().astype(np.float64)
        err, T = o(sc["scan"])
        assert err == 0 and o.stats.iterations == it
        M = np.eye(4); M[:3, 3] = mean
        Mi = np.eye(4); Mi[:3, 3] = -mean
        series.append(Mi @ T.astype(np.float64) @ M)
    return series


def differential_stop(series, smooth, min_rot, min_trans):
    """first iteration after which mean |angular distance| and mean translation step over the last `smooth` pose pairs are both below
    their limits (needs more than `smooth` poses), with the margin of that decision"""
    rot = [Rotation.from_matrix(T[:3, :3]) for T in series]
    for it in range(1, len(series)):
        if it + 1 <= smooth:
            continue
        r = np.mean([(rot[j].inv() * rot[j - 1]).magnitude() for j in range(it - smooth + 1, it + 1)])
        t = np.mean([np.linalg.norm(series[j][:3, 3] - series[j - 1][:3, 3]) for j in range(it - smooth + 1, it + 1)])
        if r < min_rot and t < min_trans:
            return it
    return None


def test_differential_and_bound_checkers_match_scipy(oracle, small_scene):
    sc = small_scene
    kw = dict(minimizer=2, max_dist=2.0, outliers=[(4, 0.85)])
    series = checker_series(oracle, sc, kw, 9)
    rot = [Rotation.from_matrix(T[:3, :3]) for T in series]
    steps_r = [(rot[j].inv() * rot[j - 1]).magnitude() for j in range(1, len(series))]
    steps_t = [np.linalg.norm(series[j][:3, 3] - series[j - 1][:3, 3]) for j in range(1, len(series))]
    for smooth in (2, 3):
        # limits placed between two well separated window means, so that the stop iteration does not hinge on rounding
        means_r = [np.mean(steps_r[i - smooth:i]) for i in range(smooth, len(steps_r) + 1)]
        means_t = [np.mean(steps_t[i - smooth:i]) for i in range(smooth, len(steps_t) + 1)]
        j = 2
        min_rot, min_trans = math.sqrt(means_r[j] * means_r[j + 1]), math.sqrt(means_t[j] * means_t[j + 1]) * 50
        want = differential_stop(series, smooth, min_rot, min_trans)
        assert want is not None
        o = oracle.OracleICP(oracle.make_config(nthreads=8, max_iterations=40, use_differential=1, min_diff_rot=min_rot, min_diff_trans=min_trans,
                                                smooth_length=smooth, **kw))
        o.setMap(sc["map"], sc["normals"])
        err, _ = o(sc["scan"])
        assert err == 0 and o.stats.stop_reason == 2 and o.stats.iterations == want, (smooth, o.stats.iterations, want)
    # Bound: accumulated rotation / translation from the initial pose; a limit between iterations 2 and 3 must raise, one above all passes
    acc_r = [(rot[j].inv() * rot[0]).magnitude() for j in range(len(series))]
    acc_t = [np.linalg.norm(series[j][:3, 3] - series[0][:3, 3]) for j in range(len(series))]
    for use_rot in (True, False):
        acc = acc_r if use_rot else acc_t
        lim_fail = 0.5 * (acc[1] + acc[2]) if acc[2] > acc[1] else None
        big = 10.0
        if lim_fail is not None:
            o = oracle.OracleICP(oracle.make_config(nthreads=8, max_iterations=9, use_bound=1, max_rot_norm=lim_fail if use_rot else big,
                                                    max_trans_norm=big if use_rot else lim_fail, **kw))
            o.setMap(sc["map"], sc["normals"])
            err, _ = o(sc["scan"])
            assert err == 3, (use_rot, err)                      # ORC_ERR_BOUND
        o = oracle.OracleICP(oracle.make_config(nthreads=8, max_iterations=9, use_bound=1, max_rot_norm=max(acc_r) * 1.5, max_trans_norm=max(acc_t) * 1.5, **kw))
        o.setMap(sc["map"], sc["normals"])
        err, _ = o(sc["scan"])
        assert err == 0 and o.stats.iterations == 9
