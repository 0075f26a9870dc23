"""Pins that need no reference build:
* std::minstd_rand is fixed by the C++ standard -- [rand.predef]: "the 10000th consecutive invocation of a default-constructed object of
  type minstd_rand produces the value 399268537".  The three restatements of that stream (the oracle's, the C++ host filters', the device's
  skip-ahead in csrc/ssn.hip) are held to it, and to each other on other seeds / positions through Python integers.
* MaxDensityDataPointsFilter (libpointmatcher, as recalled in SURVEY B.9 / host/IcpSequence.cpp): a numpy restatement that shares no code with the
  oracle or the host shell, held to both."""
import ctypes as C
import os

import numpy as np
import pytest

MINSTD_10000TH = 399268537       # [rand.predef]
M31 = 2147483647


def _minstd_py(seed, n):
    x = seed % M31 or 1
    for _ in range(n):
        x = x * 48271 % M31
    return x


def test_python_minstd_is_the_standards_stream():
    assert _minstd_py(1, 10000) == MINSTD_10000TH
    assert _minstd_py(1, 1) == 48271 and _minstd_py(0, 1) == 48271      # seed 0 is mapped to 1 (linear_congruential_engine, c == 0)
    assert _minstd_py(M31, 1) == 48271                                   # seed mod m


def test_oracle_minstd_10000th_value(oracle):
    lib = oracle.load()
    lib.orc_minstd_nth.restype = C.c_uint32
    lib.orc_minstd_nth.argtypes = [C.c_uint32, C.c_uint32]
    assert lib.orc_minstd_nth(1, 10000) == MINSTD_10000TH
    for seed, n in [(1, 1), (7, 123), (12345, 99999), (0, 5), (M31, 3), (4294967295, 17)]:
        assert lib.orc_minstd_nth(seed, n) == _minstd_py(seed, n), (seed, n)


def test_host_filters_minstd_10000th_value():
    import host_bindings as hb
    if not os.path.exists(hb.LIB):
        pytest.skip("host shell not built")
    lib = hb.load()
    lib.nim_test_minstd_nth.restype = C.c_uint32
    lib.nim_test_minstd_nth.argtypes = [C.c_uint32, C.c_uint32]
    assert lib.nim_test_minstd_nth(1, 10000) == MINSTD_10000TH
    for seed, n in [(1, 1), (7, 123), (12345, 99999), (0, 5), (M31, 3), (4294967295, 17)]:
        assert lib.nim_test_minstd_nth(seed, n) == _minstd_py(seed, n), (seed, n)


@pytest.mark.gpu
def test_device_skip_ahead_minstd_10000th_value():
    """csrc/ssn.hip: minstd_nth (48271^n x0 mod 2^31 - 1 by square-and-multiply), the function ssn_draw_kernel draws with"""
    import norlab_icp_mapper_amd as pkg
    from norlab_icp_mapper_amd import _capi
    icp = pkg.ICPSequence()
    lib = _capi.load()
    out = C.c_uint32(0)
    assert lib.icpmi_debug_minstd_nth(icp._h, 1, 10000, C.byref(out)) == 0 and out.value == MINSTD_10000TH
    for seed, n in [(1, 1), (7, 123), (12345, 99999), (0, 5), (M31, 3), (4294967295, 17), (3, 2 ** 31)]:
        assert lib.icpmi_debug_minstd_nth(icp._h, seed, n, C.byref(out)) == 0
        assert out.value == _minstd_py(seed, n) if n < 10 ** 6 else out.value == pow(48271, n, M31) * (seed % M31 or 1) % M31, (seed, n)


def _max_density_numpy(dens, max_density, seed):
    """MaxDensityDataPointsFilter: a point whose density exceeds maxDensity survives with probability maxDensity / density; the generator
    (std::minstd_rand(seed), 'direct' unit = x / float(max - min)) advances ONLY for such points.  float32 throughout, as upstream's T = float."""
    x = seed % M31 or 1
    keep = np.ones(dens.shape[0], bool)
    md = np.float32(max_density)
    for i, d in enumerate(dens.astype(np.float32)):
        if d > md:
            x = x * 48271 % M31
            keep[i] = np.float32(x) / np.float32(2147483645.0) < md / d
    return keep


@pytest.mark.parametrize("max_density,seed", [(10.0, 1), (3.5, 77), (1e9, 5), (0.0, 2)])
def test_max_density_numpy_vector(oracle, max_density, seed):
    rng = np.random.default_rng(11)
    dens = np.exp(rng.normal(2.0, 1.5, 4000)).astype(np.float32)
    dens[::97] = np.float32(max_density)                       # equality: not above the limit -> kept without a draw
    want = _max_density_numpy(dens, max_density, seed)
    keep = np.zeros(dens.shape[0], np.uint8)
    oracle.load().orc_max_density_keep(dens.ctypes.data, dens.shape[0], C.c_float(max_density), seed, keep.ctypes.data)
    assert np.array_equal(keep.astype(bool), want)
    if 0 < max_density < 1e8:
        assert 0.2 < want.mean() < 0.999                       # the vector exercises both branches
    import host_bindings as hb
    if os.path.exists(hb.LIB) and max_density > 0:      # (the host shell rejects maxDensity <= 0, as the parameter's documented range does)
        c = np.ones((dens.shape[0], 4), np.float32); c[:, 0] = np.arange(dens.shape[0], dtype=np.float32)
        out, _, _ = hb.filter_chain("[{MaxDensityDataPointsFilter: {maxDensity: %r, seed: %d}}]" % (max_density, seed), c, desc_name="densities", desc=dens)
        assert np.array_equal(out[:, 0].astype(np.int64), np.nonzero(want)[0])


# ---- SurfaceNormalDataPointsFilter{keepEigenValues, keepEigenVectors, sortEigen: 1} (r5) -------------------------------------------------
def _eigh_reference(cloud, knn):
    """numpy / scipy restatement: scatter matrix NN NN^T of the knn neighbours (self included) about their mean, numpy.linalg.eigh (ascending)"""
    from scipy.spatial import cKDTree
    xyz = cloud[:, :3].astype(np.float64)
    _, ids = cKDTree(xyz).query(xyz, k=knn)
    nb = xyz[ids]                                            # (m, knn, 3)
    c = nb - nb.mean(axis=1, keepdims=True)
    S = np.einsum("mki,mkj->mij", c, c)
    w, v = np.linalg.eigh(S)                                 # ascending, columns = eigenvectors
    return w, v


def _check_eigen_against_numpy(ev, evec, normals, cloud, knn):
    w, v = _eigh_reference(cloud, knn)
    scale = np.abs(w).max(axis=1, keepdims=True) + 1e-30
    assert np.all(np.diff(ev, axis=1) >= 0)                                         # ascending
    assert np.max(np.abs(ev - w) / scale) < 2e-5
    V = evec.reshape(-1, 3, 3)                                                      # V[i, k, j] = component k of eigenvector j
    gap_ok = (np.diff(w, axis=1).min(axis=1) > 1e-3 * scale[:, 0])                  # an eigenvector is defined up to sign where its value is simple
    dots = np.abs(np.einsum("mkj,mkj->mj", V.astype(np.float64), v))
    assert gap_ok.mean() > 0.9 and np.all(dots[gap_ok] > 1 - 1e-4)
    assert np.array_equal(V[gap_ok][:, :, 0], normals[gap_ok])                      # the normal IS the first (smallest) eigenvector


def test_oracle_surface_normal_eigen_outputs_match_numpy_eigh(oracle):
    rng = np.random.default_rng(5)
    cloud = np.ones((3000, 4), np.float32)
    cloud[:, :3] = (rng.uniform(-1, 1, (3000, 3)) * [6.0, 4.0, 0.3]).astype(np.float32)
    nrm, ev, evec = oracle.surface_normals_eigen(cloud, knn=9, nthreads=4)
    assert np.array_equal(nrm, oracle.surface_normals(cloud, knn=9, nthreads=4))   # asking for the eigen outputs changes nothing else
    _check_eigen_against_numpy(ev, evec, nrm, cloud, 9)
    # a rank-1 neighbourhood (points on a line): upstream's degenerate answer -- eigenvalues 0, eigenvectors identity, normal (1, 0, 0)
    line = np.ones((50, 4), np.float32); line[:, 0] = np.arange(50, dtype=np.float32) * 0.1; line[:, 1:3] = 0
    nrm, ev, evec = oracle.surface_normals_eigen(line, knn=5)
    assert np.all(ev == 0) and np.array_equal(evec, np.tile(np.eye(3, dtype=np.float32).reshape(-1), (50, 1))) and np.all(nrm == [1, 0, 0])


@pytest.mark.gpu
def test_device_surface_normal_eigen_outputs_match_oracle_and_numpy(oracle):
    import norlab_icp_mapper_amd as pkg
    rng = np.random.default_rng(6)
    cloud = np.ones((20000, 4), np.float32)
    cloud[:, :3] = (rng.uniform(-1, 1, (20000, 3)) * [20.0, 12.0, 0.5]).astype(np.float32)
    cloud[:5000, 2] = 0.0                                    # a flat patch: the smallest eigenvalue is exactly 0 there
    icp = pkg.ICPSequence()
    nrm, ev, evec = icp.surfaceNormalsEigen(cloud, knn=10)
    o_nrm, o_ev, o_evec = oracle.surface_normals_eigen(cloud, knn=10, nthreads=8)
    assert np.array_equal(nrm, icp.surfaceNormals(cloud, knn=10))
    flip = np.sign((nrm * o_nrm).sum(1)); flip[flip == 0] = 1   # (the suite compares normals up to sign)
    assert np.allclose(nrm * flip[:, None], o_nrm, atol=1e-6)
    assert np.allclose(ev, o_ev, rtol=1e-6, atol=1e-9)
    assert np.allclose(np.abs(evec), np.abs(o_evec), atol=2e-5) or np.mean(np.abs(np.abs(evec) - np.abs(o_evec)) < 2e-5) > 0.999
    _check_eigen_against_numpy(ev, evec, nrm, cloud, 10)
    line = np.ones((64, 4), np.float32); line[:, 0] = np.arange(64, dtype=np.float32) * 0.1; line[:, 1:3] = 0
    nrm, ev, evec = icp.surfaceNormalsEigen(line, knn=5)
    assert np.all(ev == 0) and np.array_equal(evec, np.tile(np.eye(3, dtype=np.float32).reshape(-1), (64, 1)))


def test_host_filter_serves_eigen_descriptors_only_sorted():
    import host_bindings as hb
    if not os.path.exists(hb.LIB):
        pytest.skip("host shell not built")
    c = np.ones((10, 4), np.float32)
    with pytest.raises(RuntimeError, match="sortEigen"):
        hb.filter_chain("[{SurfaceNormalDataPointsFilter: {knn: 5, keepEigenValues: 1}}]", c)
    with pytest.raises(RuntimeError, match="smoothNormals"):
        hb.filter_chain("[{SurfaceNormalDataPointsFilter: {knn: 5, smoothNormals: 1}}]", c)
