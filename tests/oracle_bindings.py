"""ctypes binding of oracle/liboracle.so (the CPU restatement; test infrastructure only)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.environ.get("ICP_ORACLE_LIB") or os.path.join(_ROOT, "oracle", "liboracle.so")  # (override: the sanitizer build, tests/tools/oracle_sanitize.sh)

MIN_IDENTITY, MIN_POINT_TO_POINT, MIN_POINT_TO_PLANE = 0, 1, 2
OUT_MAXDIST, OUT_MINDIST, OUT_MEDIANDIST, OUT_TRIMMEDDIST, OUT_SURFACENORMAL = 1, 2, 3, 4, 5
OUT_GENERICDESCRIPTOR, OUT_ROBUST, OUT_VARTRIMMEDDIST = 6, 7, 8
STOP_COUNTER, STOP_DIFFERENTIAL = 1, 2


class Outlier(C.Structure):
    _fields_ = [("type", C.c_int), ("param", C.c_float), ("iparam", C.c_int), ("param2", C.c_float), ("param3", C.c_float)]


class Config(C.Structure):
    _fields_ = [("knn", C.c_int), ("max_dist", C.c_float), ("minimizer", C.c_int), ("n_outlier", C.c_int),
                ("outlier", Outlier * 8), ("max_iterations", C.c_int), ("use_differential", C.c_int),
                ("min_diff_rot", C.c_float), ("min_diff_trans", C.c_float), ("smooth_length", C.c_int),
                ("use_bound", C.c_int), ("max_rot_norm", C.c_float), ("max_trans_norm", C.c_float),
                ("nthreads", C.c_int), ("force_4dof", C.c_int), ("force_2d", C.c_int), ("is_2d", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("stop_reason", C.c_int), ("error", C.c_int), ("pairs", C.c_int64),
                ("point_used_ratio", C.c_float), ("weighted_point_used_ratio", C.c_float),
                ("trimmed_limit", C.c_float), ("seconds_knn", C.c_double), ("seconds_total", C.c_double),
                ("sensor_noise_overlap", C.c_float)]


_lib = None
_P = C.c_void_p


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        build()
    lib = C.CDLL(_LIB)
    lib.orc_kdtree_build.restype = _P
    lib.orc_kdtree_build.argtypes = [_P, C.c_int64, C.c_int, C.c_int]
    lib.orc_kdtree_free.argtypes = [_P]
    lib.orc_kdtree_knn.argtypes = [_P, _P, C.c_int64, C.c_int, C.c_float, C.c_int, _P, _P, C.c_int]
    lib.orc_kdtree_knn_eps.argtypes = [_P, _P, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_int, _P, _P, C.c_int]
    lib.orc_bruteforce_knn.argtypes = [_P, C.c_int64, C.c_int, _P, C.c_int64, C.c_int, C.c_float, C.c_int, _P, _P]
    lib.orc_transform.argtypes = [_P, _P, _P, C.c_int64]
    lib.orc_rotate3.argtypes = [_P, _P, _P, C.c_int64]
    lib.orc_dists_quantile.restype = C.c_float
    lib.orc_dists_quantile.argtypes = [_P, C.c_int64, C.c_float]
    lib.orc_outlier_weights.argtypes = [C.POINTER(Config), _P, _P, C.c_int, C.c_int64, _P, _P, _P, C.POINTER(C.c_float)]
    lib.orc_outlier_weights_ex.argtypes = [C.POINTER(Config), _P, _P, C.c_int, C.c_int64, _P, _P, _P, _P, _P, C.c_int, C.POINTER(C.c_float),
                                           _P, C.POINTER(C.c_float)]
    lib.orc_var_trimmed_ratio.argtypes = [_P, C.c_int64, C.c_float, C.c_float, C.c_float]
    lib.orc_var_trimmed_ratio.restype = C.c_float
    lib.orc_icp_set_map_scalar.argtypes = [_P, _P]
    lib.orc_icp_set_map_scalar.restype = None
    lib.orc_solve_n.argtypes = [C.c_int, _P, _P, _P]
    lib.orc_solve_n.restype = None
    lib.orc_minimize.argtypes = [C.c_int, _P, C.c_int64, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, C.POINTER(Stats)]
    lib.orc_minimize_ex.argtypes = [C.c_int, C.c_int, _P, C.c_int64, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, C.POINTER(Stats)]
    lib.orc_rotation_from_H.argtypes = [_P, _P]
    lib.orc_rotation_from_H_svd.argtypes = [_P, _P]
    lib.orc_sincos_f.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.orc_sincos_f.restype = None
    lib.orc_solve6.argtypes = [_P, _P, _P]
    lib.orc_icp_create.restype = _P
    lib.orc_icp_create.argtypes = [C.POINTER(Config)]
    lib.orc_icp_destroy.argtypes = [_P]
    lib.orc_icp_set_map.argtypes = [_P, _P, C.c_int64, _P]
    lib.orc_icp_has_map.argtypes = [_P]
    lib.orc_icp_get_mean.argtypes = [_P, _P]
    lib.orc_icp_register.argtypes = [_P, _P, C.c_int64, _P, _P, C.POINTER(Stats)]
    lib.orc_icp_set_reading_noise.argtypes = [_P, _P, C.c_int64]
    lib.orc_icp_set_reading_noise.restype = None
    lib.orc_icp_set_reading_scalar.argtypes = [_P, _P, C.c_int64]
    lib.orc_icp_set_reading_scalar.restype = None
    lib.orc_set_reading_scalar.argtypes = [_P]
    lib.orc_set_reading_scalar.restype = None
    lib.orc_surface_normals.argtypes = [_P, C.c_int64, C.c_int, _P, C.c_int]
    lib.orc_surface_normals_ex.argtypes = [_P, C.c_int64, C.c_int, _P, _P, C.c_int]
    lib.orc_surface_normals_2d.argtypes = [_P, C.c_int64, C.c_int, _P, C.c_int]
    lib.orc_point_distance_keep.argtypes = [_P, C.c_int64, _P, C.c_int64, C.c_float, _P, C.c_int]
    lib.orc_cell_ids.argtypes = [_P, C.c_int64, C.c_float, _P]
    lib.orc_voxel_keep_first.argtypes = [_P, C.c_int64, C.c_float, _P]
    lib.orc_filter_points.argtypes = [_P, C.c_int64, _P, C.c_int, _P]
    lib.orc_voxel_keep.argtypes = [_P, C.c_int64, C.c_float, C.c_int, _P]
    lib.orc_random_sampling_keep.argtypes = [C.c_int64, C.c_float, C.c_int, C.c_int, _P]
    lib.orc_max_density_keep.argtypes = [_P, C.c_int64, C.c_float, C.c_int, _P]
    lib.orc_sampling_surface_normal.restype = C.c_int64
    lib.orc_sampling_surface_normal.argtypes = [_P, C.c_int64, C.c_float, C.c_int, C.c_float, C.c_int, _P, _P]
    lib.orc_octree_sample.restype = C.c_int64
    lib.orc_octree_sample.argtypes = [_P, C.c_int64, C.c_float, C.c_int64, C.c_int, _P]
    lib.orc_dynamic_points_update.argtypes = [_P, _P, _P, C.c_int64, _P, _P, C.c_int64, _P, C.c_int]
    _lib = lib
    return lib


def make_config(knn=1, max_dist=math.inf, minimizer=MIN_POINT_TO_PLANE, outliers=(), max_iterations=40,
                use_differential=0, min_diff_rot=1e-3, min_diff_trans=1e-3, smooth_length=3, use_bound=0,
                max_rot_norm=1.0, max_trans_norm=1.0, nthreads=1, force_4dof=0, force_2d=0, is_2d=0):
    cfg = Config()
    cfg.knn, cfg.max_dist, cfg.minimizer = knn, max_dist, minimizer
    cfg.n_outlier = len(outliers)
    for i, o in enumerate(outliers):  # (type, param[, iparam[, param2]])
        cfg.outlier[i].type, cfg.outlier[i].param = o[0], o[1]
        cfg.outlier[i].iparam = o[2] if len(o) > 2 else 0
        cfg.outlier[i].param2 = o[3] if len(o) > 3 else 0.0
        cfg.outlier[i].param3 = o[4] if len(o) > 4 else 0.0
    cfg.max_iterations, cfg.use_differential = max_iterations, use_differential
    cfg.min_diff_rot, cfg.min_diff_trans, cfg.smooth_length = min_diff_rot, min_diff_trans, smooth_length
    cfg.use_bound, cfg.max_rot_norm, cfg.max_trans_norm = use_bound, max_rot_norm, max_trans_norm
    cfg.nthreads = nthreads
    cfg.force_4dof = force_4dof
    cfg.force_2d = force_2d
    cfg.is_2d = is_2d
    return cfg


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def T_to_c(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).ravel()


def T_from_c(buf):
    return np.array(buf, dtype=np.float32).reshape(4, 4).T.copy()


def transform(T, cloud):
    lib = load(); cloud = _f32(cloud); out = np.empty_like(cloud); Tc = T_to_c(T)
    lib.orc_transform(Tc.ctypes.data, cloud.ctypes.data, out.ctypes.data, cloud.shape[0])
    return out


def rotate3(T, normals):
    lib = load(); normals = _f32(normals); out = np.empty_like(normals); Tc = T_to_c(T)
    lib.orc_rotate3(Tc.ctypes.data, normals.ctypes.data, out.ctypes.data, normals.shape[0])
    return out


def knn(cloud, queries, k=1, max_dist=math.inf, allow_self=True, bucket=8, nthreads=1, brute=False, epsilon=0.0):
    lib = load(); cloud = _f32(cloud); q = _f32(queries)
    ids = np.empty((q.shape[0], k), dtype=np.int32); d2 = np.empty((q.shape[0], k), dtype=np.float32)
    if epsilon > 0.0:   # KDTreeMatcher{epsilon}: libnabo's approximate rule on the oracle's tree
        t = lib.orc_kdtree_build(cloud.ctypes.data, cloud.shape[0], 3, bucket)
        lib.orc_kdtree_knn_eps(t, q.ctypes.data, q.shape[0], k, max_dist, epsilon, int(allow_self), ids.ctypes.data, d2.ctypes.data, nthreads)
        lib.orc_kdtree_free(t)
        return ids, d2
    if brute:
        lib.orc_bruteforce_knn(cloud.ctypes.data, cloud.shape[0], 3, q.ctypes.data, q.shape[0], k, max_dist, int(allow_self),
                               ids.ctypes.data, d2.ctypes.data)
    else:
        t = lib.orc_kdtree_build(cloud.ctypes.data, cloud.shape[0], 3, bucket)
        lib.orc_kdtree_knn(t, q.ctypes.data, q.shape[0], k, max_dist, int(allow_self), ids.ctypes.data, d2.ctypes.data, nthreads)
        lib.orc_kdtree_free(t)
    return ids, d2


def dists_quantile(d2, q):
    lib = load(); d2 = _f32(d2).ravel()
    return float(lib.orc_dists_quantile(d2.ctypes.data, d2.size, q))


def var_trimmed_ratio(d2, min_ratio=0.05, max_ratio=0.99, lam=0.95):
    lib = load(); d2 = _f32(d2).ravel()
    return float(lib.orc_var_trimmed_ratio(d2.ctypes.data, d2.size, min_ratio, max_ratio, lam))


def outlier_weights(cfg, d2, ids, read_normals=None, ref_normals=None, ref_scalar=None, step=None, ref=None, iteration=1, scale=1.0, read_scalar=None):
    """OutlierFilters::compute of the chain; the keyword extras feed GenericDescriptor (ref_scalar) and Robust (step, ref,
    iteration, the scale kept from the previous iteration).  Returns (err, weights, limit) -- with Robust{mad} limit = scale."""
    lib = load(); d2 = _f32(d2); ids = np.ascontiguousarray(ids, dtype=np.int32)
    n, k = d2.shape
    w = np.empty_like(d2); lim = C.c_float(-1); sc = C.c_float(scale)
    ptr = lambda a: _f32(a).ctypes.data if a is not None else None
    keep = [_f32(a) if a is not None else None for a in (read_normals, ref_normals, ref_scalar, step, ref)]
    args = [a.ctypes.data if a is not None else None for a in keep]
    rs = _f32(read_scalar).ravel() if read_scalar is not None else None
    lib.orc_set_reading_scalar(rs.ctypes.data if rs is not None else None)
    err = lib.orc_outlier_weights_ex(C.byref(cfg), d2.ctypes.data, ids.ctypes.data, k, n, args[0], args[1], args[2], args[3], args[4],
                                     int(iteration), C.byref(sc), w.ctypes.data, C.byref(lim))
    lib.orc_set_reading_scalar(None)
    return err, w, float(lim.value)


def solve_n(A, b):
    """solvePossiblyUnderdeterminedLinearSystem at any size <= 6 (column-major == row-major: A is symmetric)"""
    lib = load(); A = _f32(A); b = _f32(b); n = b.size
    x = np.zeros(n, dtype=np.float32)
    lib.orc_solve_n(n, A.ctypes.data, b.ctypes.data, x.ctypes.data)
    return x


def minimize(minimizer, reading, ref, ref_normals, ids, d2, w, force_4dof=0):
    lib = load(); reading = _f32(reading); ref = _f32(ref)
    ids = np.ascontiguousarray(ids, dtype=np.int32); d2 = _f32(d2); w = _f32(w)
    n, k = d2.shape
    T = np.zeros(16, dtype=np.float32); A = np.zeros(36); b = np.zeros(6); x = np.zeros(6, dtype=np.float32)
    st = Stats()
    nr = _f32(ref_normals) if ref_normals is not None else None
    err = lib.orc_minimize_ex(minimizer, int(force_4dof), reading.ctypes.data, n, ref.ctypes.data, nr.ctypes.data if nr is not None else None,
                           ids.ctypes.data, d2.ctypes.data, w.ctypes.data, k, T.ctypes.data, A.ctypes.data, b.ctypes.data,
                           x.ctypes.data, C.byref(st))
    return err, T_from_c(T), A.reshape(6, 6).T.copy(), b, x, st


def sincos_f(x):
    lib = load(); s = C.c_float(); c = C.c_float()
    lib.orc_sincos_f(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def rotation_from_H(H, svd=False):
    """R = U V^T of H (reflection repaired); svd=True: the route through the SVD itself instead of the polar iteration"""
    lib = load(); Hc = np.ascontiguousarray(np.asarray(H, dtype=np.float32).T).ravel(); R = np.zeros(9, dtype=np.float32)
    (lib.orc_rotation_from_H_svd if svd else lib.orc_rotation_from_H)(Hc.ctypes.data, R.ctypes.data)
    return R.reshape(3, 3).T.copy()


def solve6(A, b):
    lib = load(); Ac = np.ascontiguousarray(np.asarray(A, dtype=np.float32).T).ravel(); bc = _f32(b); x = np.zeros(6, dtype=np.float32)
    lib.orc_solve6(Ac.ctypes.data, bc.ctypes.data, x.ctypes.data)
    return x


class OracleICP:
    def __init__(self, cfg):
        self.lib = load(); self.cfg = cfg
        self.h = self.lib.orc_icp_create(C.byref(cfg)); self.stats = Stats()

    def __del__(self):
        try:
            if self.h:
                self.lib.orc_icp_destroy(self.h); self.h = None
        except Exception:
            pass

    def setMap(self, cloud, normals=None):
        cloud = _f32(cloud); n = _f32(normals) if normals is not None else None
        return bool(self.lib.orc_icp_set_map(self.h, cloud.ctypes.data, cloud.shape[0], n.ctypes.data if n is not None else None))

    def setMapScalar(self, scalar):
        """the 1-row descriptor GenericDescriptorOutlierFilter{source: reference} reads"""
        s = _f32(scalar).ravel(); self.lib.orc_icp_set_map_scalar(self.h, s.ctypes.data)

    def getMapMean(self):
        m = np.zeros(3, dtype=np.float32); self.lib.orc_icp_get_mean(self.h, m.ctypes.data); return m

    def setReadingNoise(self, noise):
        """`simpleSensorNoise` row of the next reading (one shot): stats.sensor_noise_overlap is then getOverlap()"""
        nz = np.ascontiguousarray(noise, dtype=np.float32).ravel()
        self.lib.orc_icp_set_reading_noise(self.h, nz.ctypes.data, nz.shape[0])

    def setReadingScalar(self, scalar):
        """GenericDescriptorOutlierFilter{source: reading}: that 1-row descriptor of the next reading (one shot)"""
        s = np.ascontiguousarray(scalar, dtype=np.float32).ravel()
        self.lib.orc_icp_set_reading_scalar(self.h, s.ctypes.data, s.shape[0])

    def __call__(self, scan, scan_normals=None):
        scan = _f32(scan); sn = _f32(scan_normals) if scan_normals is not None else None
        T = np.zeros(16, dtype=np.float32)
        err = self.lib.orc_icp_register(self.h, scan.ctypes.data, scan.shape[0], sn.ctypes.data if sn is not None else None,
                                        T.ctypes.data, C.byref(self.stats))
        return err, T_from_c(T)


def surface_normals_extras(cloud, knn=5, nthreads=1):
    """(normals, matched ids (m, knn), mean distance (m,)): keepMatchedIds / keepMeanDist of the filter"""
    lib = load(); cloud = _f32(cloud); m = cloud.shape[0]
    out = np.empty((m, 3), dtype=np.float32); ids = np.empty((m, knn), dtype=np.int32); md = np.empty(m, dtype=np.float32)
    lib.orc_surface_normals_extras.argtypes = [_P, _P]; lib.orc_surface_normals_extras.restype = None
    lib.orc_surface_normals_extras(ids.ctypes.data, md.ctypes.data)
    lib.orc_surface_normals(cloud.ctypes.data, m, knn, out.ctypes.data, nthreads)
    return out, ids, md


def surface_normals_eigen(cloud, knn=5, nthreads=1):
    """(normals, eigenvalues (m, 3) ascending, serialised eigenvectors (m, 9)): keepEigenValues / keepEigenVectors with sortEigen 1"""
    lib = load(); cloud = _f32(cloud); m = cloud.shape[0]
    out = np.empty((m, 3), dtype=np.float32); ev = np.empty((m, 3), dtype=np.float32); evec = np.empty((m, 9), dtype=np.float32)
    lib.orc_surface_normals_eigen.argtypes = [_P, _P]; lib.orc_surface_normals_eigen.restype = None
    lib.orc_surface_normals_eigen(ev.ctypes.data, evec.ctypes.data)
    lib.orc_surface_normals(cloud.ctypes.data, m, knn, out.ctypes.data, nthreads)
    return out, ev, evec


def surface_normals(cloud, knn=5, nthreads=1, with_densities=False, planar=False):
    lib = load(); cloud = _f32(cloud); out = np.empty((cloud.shape[0], 3), dtype=np.float32)
    if planar:  # 2-D clouds (z == 0): the smaller eigenvector of the 2 x 2 covariance
        lib.orc_surface_normals_2d(cloud.ctypes.data, cloud.shape[0], knn, out.ctypes.data, nthreads)
        return out
    if not with_densities:
        lib.orc_surface_normals(cloud.ctypes.data, cloud.shape[0], knn, out.ctypes.data, nthreads)
        return out
    dens = np.empty(cloud.shape[0], dtype=np.float32)
    lib.orc_surface_normals_ex(cloud.ctypes.data, cloud.shape[0], knn, out.ctypes.data, dens.ctypes.data, nthreads)
    return out, dens


def point_distance_keep(map_cloud, in_cloud, min_dist, nthreads=1):
    lib = load(); m = _f32(map_cloud); i = _f32(in_cloud); keep = np.empty(i.shape[0], dtype=np.uint8)
    lib.orc_point_distance_keep(m.ctypes.data, m.shape[0], i.ctypes.data, i.shape[0], min_dist, keep.ctypes.data, nthreads)
    return keep.astype(bool)


def cell_ids(cloud, cell_size=20.0):
    lib = load(); c = _f32(cloud); out = np.empty((c.shape[0], 3), dtype=np.int32)
    lib.orc_cell_ids(c.ctypes.data, c.shape[0], cell_size, out.ctypes.data)
    return out


def voxel_keep_first(cloud, edge):
    lib = load(); c = _f32(cloud); keep = np.zeros(c.shape[0], dtype=np.uint8)
    lib.orc_voxel_keep_first(c.ctypes.data, c.shape[0], edge, keep.ctypes.data)
    return keep.astype(bool)


def filter_points(cloud, filters):
    """filters as ICPSequence.filterPoints takes them"""
    lib = load(); c = _f32(cloud); keep = np.zeros(c.shape[0], dtype=np.uint8)
    rows = np.zeros((max(1, len(filters)), 8), dtype=np.float32)
    for k, f in enumerate(filters):
        if f[0] == "distance_limit":
            rows[k] = [0, f[1], f[2], 1.0 if f[3] else 0.0, 0, 0, 0, 0]
        else:
            rows[k] = [1, 1.0 if f[3] else 0.0, *f[1], *f[2]]
    lib.orc_filter_points(c.ctypes.data, c.shape[0], rows.ctypes.data, len(filters), keep.ctypes.data)
    return keep.astype(bool)


def voxel_keep(cloud, edge, method):
    lib = load(); c = _f32(cloud); keep = np.zeros(c.shape[0], dtype=np.uint8)
    lib.orc_voxel_keep(c.ctypes.data, c.shape[0], edge, method, keep.ctypes.data)
    return keep.astype(bool)


def random_sampling_keep(n, prob=0.75, method=0, seed=1):
    lib = load(); keep = np.zeros(n, dtype=np.uint8)
    lib.orc_random_sampling_keep(n, prob, method, seed, keep.ctypes.data)
    return keep.astype(bool)


def max_density_keep(densities, max_density=10.0, seed=1):
    lib = load(); d = _f32(densities); keep = np.zeros(d.shape[0], dtype=np.uint8)
    lib.orc_max_density_keep(d.ctypes.data, d.shape[0], max_density, seed, keep.ctypes.data)
    return keep.astype(bool)


def sampling_surface_normal(cloud, ratio=0.5, knn=7, max_box_dim=math.inf, seed=1):
    """(kept indices in box order, their normals)"""
    lib = load(); c = _f32(cloud); order = np.empty(c.shape[0], dtype=np.int32); nrm = np.empty((c.shape[0], 3), dtype=np.float32)
    m = lib.orc_sampling_surface_normal(c.ctypes.data, c.shape[0], ratio, knn, max_box_dim, seed, order.ctypes.data, nrm.ctypes.data)
    return order[:m].copy(), nrm[:m].copy()


def sampling_surface_normal_boxes(cloud, knn=7, max_box_dim=math.inf):
    """samplingMethod 1: (first member index, normal, mean, member start, member count, members) per surviving box"""
    lib = load(); c = _f32(cloud); n = c.shape[0]
    order = np.empty(n, dtype=np.int32); nrm = np.empty((n, 3), dtype=np.float32); mean = np.empty((n, 3), dtype=np.float32)
    ms = np.empty(n, dtype=np.int32); mc = np.empty(n, dtype=np.int32); mem = np.empty(n, dtype=np.int32)
    lib.orc_sampling_surface_normal_ex.restype = C.c_int64
    lib.orc_sampling_surface_normal_ex.argtypes = [_P, C.c_int64, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]
    k = lib.orc_sampling_surface_normal_ex(c.ctypes.data, n, 1.0, knn, max_box_dim, 1, 1, order.ctypes.data, nrm.ctypes.data, mean.ctypes.data,
                                           ms.ctypes.data, mc.ctypes.data, mem.ctypes.data)
    tot = int(ms[k - 1] + mc[k - 1]) if k else 0
    return order[:k].copy(), nrm[:k].copy(), mean[:k].copy(), ms[:k].copy(), mc[:k].copy(), mem[:tot].copy()


def octree_sample(cloud, max_size, max_pts=1, method=0):
    """OctreeGridDataPointsFilter: original indices of the kept points, in leaf-visiting order"""
    lib = load(); c = _f32(cloud); order = np.empty(c.shape[0], dtype=np.int32)
    m = lib.orc_octree_sample(c.ctypes.data, c.shape[0], max_size, max_pts, method, order.ctypes.data)
    return order[:m].copy()


DYNPTS_DEFAULTS = dict(threshold_dynamic=0.6, alpha=0.8, beta=0.99, beam_half_angle=0.01, epsilon_a=0.01, epsilon_d=0.01,
                       sensor_max_range=200.0)


def dynamic_points_update(to_sensor, input_cloud, map_cloud, map_normals, prob, nthreads=1, **params):
    lib = load()
    prm = dict(DYNPTS_DEFAULTS); prm.update(params)
    pv = np.array([prm[k] for k in DYNPTS_DEFAULTS], dtype=np.float32)
    T = np.ascontiguousarray(np.asarray(to_sensor, dtype=np.float32).T)  # col-major
    i = _f32(input_cloud); m = _f32(map_cloud); nn = np.ascontiguousarray(map_normals, dtype=np.float32)
    out = np.ascontiguousarray(prob, dtype=np.float32).copy()
    lib.orc_dynamic_points_update(pv.ctypes.data, T.ctypes.data, i.ctypes.data, i.shape[0], m.ctypes.data, nn.ctypes.data, m.shape[0],
                                  out.ctypes.data, nthreads)
    return out


def check_normals_are_smallest_eigenvectors(pts, ids, normals, what):
    """EVERY normal must be a smallest-eigenvalue direction of the covariance of its exact neighbour set, to float
    precision: Rayleigh quotient within 2e-5 of the largest eigenvalue above the smallest one (float64 reference).  Where the
    two smallest eigenvalues coincide the direction is free inside that eigenspace -- and only there."""
    P = pts[:, :3].astype(np.float64)
    nb = P[ids]                                   # n x k x 3
    d = nb - nb.mean(axis=1, keepdims=True)
    C = np.einsum("nki,nkj->nij", d, d)
    lam = np.linalg.eigvalsh(C)                   # ascending
    nn = normals.astype(np.float64)
    np.testing.assert_allclose(np.linalg.norm(nn, axis=1), 1.0, atol=2e-6, err_msg=what)
    ray = np.einsum("ni,nij,nj->n", nn, C, nn)
    rank2 = lam[:, 1] > 3 * np.finfo(np.float32).eps * lam[:, 2]   # upstream's rank test; below it the normal is the fallback
    excess = (ray - lam[:, 0]) / np.maximum(lam[:, 2], 1e-300)
    assert np.all(excess[rank2] <= 2e-5), (what, float(excess[rank2].max()), int(np.argmax(excess)))
    return rank2
