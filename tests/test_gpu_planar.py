"""Planar clouds (the mapper's is3D == false, Mapper.h:53): z == 0 everywhere, icpmi_config::is_2d.  The registration core keeps
its 3-D search (identical to a 2-D one on such data) and switches the minimisers -- point-to-point: the in-plane rotation in
closed form; point-to-plane: upstream's 2-D system [x ny - y nx; nx; ny] -- and the SurfaceNormal filter to the 2 x 2 problem.
GPU against the oracle's restatement; the scene is a room outline with two round pillars (tests/test_oracle_ext.py)."""
import numpy as np
import pytest

from test_oracle_ext import _planar_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import norlab_icp_mapper_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def planar():
    mp, sc, T = _planar_scene(n_map=60000, n_scan=8000, seed=21)
    return {"map": mp, "scan": sc, "T_gt": T}


def test_planar_normals_match_oracle(amd, oracle, planar):
    mp = planar["map"]
    icp = amd.ICPSequence(minimizer=0, is_2d=1)
    n = icp.surfaceNormals(mp, knn=10)
    rn = oracle.surface_normals(mp, knn=10, nthreads=8, planar=True)
    assert np.all(n[:, 2] == 0) and np.allclose(np.linalg.norm(n, axis=1), 1, atol=2e-6)
    ids, _ = oracle.knn(mp, mp, k=10, nthreads=8)
    P = mp[:, :2].astype(np.float64); nb = P[ids]; d = nb - nb.mean(axis=1, keepdims=True)
    C = np.einsum("nki,nkj->nij", d, d)
    lam = np.linalg.eigvalsh(C)
    nn = n[:, :2].astype(np.float64)
    ray = np.einsum("ni,nij,nj->n", nn, C, nn)
    assert np.all((ray - lam[:, 0]) <= 2e-5 * np.maximum(lam[:, 1], 1e-300))        # EVERY normal is a smallest-eigenvalue direction
    iso = (lam[:, 1] - lam[:, 0]) > 1e-2 * lam[:, 1]
    dots = np.abs((n.astype(np.float64) * rn.astype(np.float64)).sum(1))
    assert iso.mean() > 0.9 and dots[iso].min() > 1 - 1e-6
    # a 3-D handle on the same cloud answers the zero eigenvalue's direction: z
    n3 = amd.ICPSequence(minimizer=0).surfaceNormals(mp, knn=10)
    assert np.abs(n3[:, 2]).min() > 0.99


@pytest.mark.parametrize("minimizer", [1, 2])
def test_planar_registration_matches_oracle(amd, oracle, planar, minimizer):
    mp, sc = planar["map"], planar["scan"]
    nrm = oracle.surface_normals(mp, knn=10, nthreads=8, planar=True)
    kw = dict(minimizer=minimizer, max_dist=1.0, outliers=[(4, 0.9)], max_iterations=40, use_differential=1, is_2d=1)
    icp = amd.ICPSequence(**kw)
    assert icp.setMap(mp, nrm)
    T = icp(sc)
    o = oracle.OracleICP(oracle.make_config(nthreads=8, **kw)); o.setMap(mp, nrm)
    err, T_ref = o(sc)
    assert err == 0
    assert icp.stats.iterations == o.stats.iterations and icp.stats.pairs == o.stats.pairs and icp.stats.stop_reason == o.stats.stop_reason
    dt, dr = amd.synth.pose_error(T, T_ref)
    assert dt <= 1e-4 and dr <= 1e-4, (dt, dr)
    gt, gr = amd.synth.pose_error(T, planar["T_gt"])
    assert gt < 1e-2 and gr < 6e-3, (gt, gr)       # 5 mm noise, a tenth of the outline trimmed
    assert T[2, 3] == 0 and np.allclose(T[2, :3], [0, 0, 1], atol=1e-7) and np.allclose(T[:3, 2], [0, 0, 1], atol=1e-7)


def test_planar_map_update_with_normals(amd, oracle, planar):
    """PointDistance accept + SurfaceNormal post filter on the resident planar map: the normals that come back are the 2-D ones"""
    mp, sc = planar["map"], planar["scan"]
    icp = amd.ICPSequence(minimizer=2, max_dist=1.0, outliers=[(4, 0.9)], max_iterations=30, use_differential=1, is_2d=1)
    half = mp[::2].copy()
    icp.setMap(half, oracle.surface_normals(half, knn=10, nthreads=8, planar=True))
    corr = icp.registerWithPrior(sc, np.eye(4, dtype=np.float32))
    appended, m, keep = icp.mapUpdateStaged(corr, 0.05, normals_knn=10, return_keep=True)
    placed = oracle.transform(corr, sc)
    assert np.array_equal(keep, oracle.point_distance_keep(half, placed, 0.05, nthreads=8)) and appended == int(keep.sum())
    got, got_n = icp.getMap(with_normals=True)
    assert np.array_equal(got, np.concatenate([half, placed[keep]])) and np.all(got[:, 2] == 0)
    rn = oracle.surface_normals(got, knn=10, nthreads=8, planar=True)
    assert np.all(got_n[:, 2] == 0)
    dots = np.abs((got_n.astype(np.float64) * rn.astype(np.float64)).sum(1))
    assert np.quantile(dots, 0.02) > 1 - 1e-6
