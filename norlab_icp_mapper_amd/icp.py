"""Python-side mirror of ``PM::ICPSequence`` (the object behind ``Mapper::processInput``,
/root/reference norlab_icp_mapper/Mapper.h:23, Mapper.cpp:72,77,213,219; Map.cpp:111,178,528,581)
over the C ABI of libicpmi.so.  Same method names and error behaviour as the reference object so
that the parity tests read like tests of the reference would:

    icp = ICPSequence.loadFromYaml(chain)     # icp.loadFromYamlNode(node["icp"])
    icp.setMap(cloud, normals)                # bool, False on an empty cloud
    T = icp(scan)                             # 4x4 correction in the map frame
    icp.errorMinimizer.getOverlap()

Clouds are numpy float32 arrays of shape (N, 4) (== a 4 x N column-major ``features`` block),
normals (N, 3); transforms are ordinary 4x4 numpy matrices.  All compute happens in the HIP library.
"""
import ctypes as C
import math

import numpy as np

from . import _capi


class ConvergenceError(RuntimeError):
    """PM::ConvergenceError"""


class InvalidField(RuntimeError):
    """PM::DataPoints::InvalidField"""


class InvalidParameter(ValueError):
    """PM::Parametrizable::InvalidParameter"""


class TransformationError(ValueError):
    """PM::TransformationError"""


class HipError(RuntimeError):
    pass


_STATUS_EXC = {
    _capi.ERR_INVALID_ARG: InvalidParameter,
    _capi.ERR_HIP: HipError,
    _capi.ERR_NO_POINT_TO_MINIMIZE: ConvergenceError,
    _capi.ERR_NO_OUTLIER_TO_FILTER: ConvergenceError,
    _capi.ERR_BOUND: ConvergenceError,
    _capi.ERR_NAN: ConvergenceError,
    _capi.ERR_MISSING_NORMALS: InvalidField,
    _capi.ERR_UNSUPPORTED: NotImplementedError,
}

_OUTLIER_NAMES = {
    "MaxDistOutlierFilter": (_capi.OUT_MAXDIST, "maxDist", 1.0),
    "MinDistOutlierFilter": (_capi.OUT_MINDIST, "minDist", 1.0),
    "MedianDistOutlierFilter": (_capi.OUT_MEDIANDIST, "factor", 3.0),
    "TrimmedDistOutlierFilter": (_capi.OUT_TRIMMEDDIST, "ratio", 0.85),
    "SurfaceNormalOutlierFilter": (_capi.OUT_SURFACENORMAL, "maxAngle", 1.57),
}
_MINIMIZERS = {
    "IdentityErrorMinimizer": _capi.MIN_IDENTITY,
    "PointToPointErrorMinimizer": _capi.MIN_POINT_TO_POINT,
    "PointToPlaneErrorMinimizer": _capi.MIN_POINT_TO_PLANE,
}


def _f32c(a, cols):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != cols:
        raise InvalidParameter(f"expected an (N, {cols}) float32 array, got {a.shape}")
    return a


def _T_to_c(T):
    """4x4 numpy (row-major) -> column-major float[16]"""
    T = np.asarray(T, dtype=np.float32)
    return np.ascontiguousarray(T.T).ravel()


def _T_from_c(buf):
    return np.array(buf, dtype=np.float32).reshape(4, 4).T.copy()


def default_config(**kw):
    lib = _capi.load()
    cfg = _capi.Config()
    lib.icpmi_config_default(C.byref(cfg))
    outliers = kw.pop("outliers", None)
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise InvalidParameter(f"unknown config field {k}")
        setattr(cfg, k, v)
    if outliers is not None:
        cfg.n_outlier = len(outliers)
        for i, o in enumerate(outliers):  # (type, param[, iparam[, param2]])
            cfg.outlier[i].type = o[0]
            cfg.outlier[i].param = o[1]
            cfg.outlier[i].iparam = o[2] if len(o) > 2 else 0
            cfg.outlier[i].param2 = o[3] if len(o) > 3 else 0.0
            cfg.outlier[i].param3 = o[4] if len(o) > 4 else 0.0
    return cfg


def config_from_yaml_chain(chain, **engine):
    """Translate the ``icp:`` sub-tree of a mapper YAML (Mapper.cpp:72) into an icpmi_config.

    Only the modules on the hot path are accepted; anything else raises like libpointmatcher's
    registrar would for an unknown module.
    """
    chain = chain or {}
    kw = {}
    valid = {"matcher", "outlierFilters", "errorMinimizer", "transformationCheckers", "inspector", "logger",
             "readingDataPointsFilters", "referenceDataPointsFilters", "readingStepDataPointsFilters"}
    for key in chain:
        if key not in valid:
            raise InvalidParameter(f"unknown ICP chain key: {key}")
    for key in ("readingDataPointsFilters", "referenceDataPointsFilters", "readingStepDataPointsFilters"):
        if chain.get(key):
            raise NotImplementedError(f"{key} inside the ICP chain are not on the accelerated path; apply them as input filters")

    def single(node, what):
        if isinstance(node, str):
            return node, {}
        if isinstance(node, dict) and len(node) == 1:
            (name, params), = node.items()
            return name, (params or {})
        raise InvalidParameter(f"malformed {what} node: {node!r}")

    matcher = chain.get("matcher", {"KDTreeMatcher": {}})
    name, p = single(matcher, "matcher")
    if name != "KDTreeMatcher":
        raise InvalidParameter(f"unknown matcher {name}")
    for key in p:
        if key not in ("knn", "epsilon", "searchType", "maxDist", "maxDistField"):
            raise InvalidParameter(f"KDTreeMatcher: unknown parameter {key}")
    kw["knn"] = int(p.get("knn", 1))
    kw["epsilon"] = float(p.get("epsilon", 0))
    md = p.get("maxDist", math.inf)
    kw["max_dist"] = math.inf if str(md) in ("inf", ".inf") else float(md)

    outs = []
    for node in chain.get("outlierFilters", []) or []:
        name, p = single(node, "outlier filter")
        if name == "GenericDescriptorOutlierFilter":
            # upstream defaults: source reference, descName none, useSoftThreshold 0, useLargerThan 1, threshold 0.1
            for key in p:
                if key not in ("source", "descName", "useSoftThreshold", "useLargerThan", "threshold"):
                    raise InvalidParameter(f"{name}: unknown parameter {key}")
            source = p.get("source", "reference")
            if source not in ("reference", "reading"):
                raise InvalidParameter(f"{name}: source must be reference or reading")
            # (source: reading -- r4: the reading's row goes to the device with ICPSequence.setReadingScalar before every registration)
            kw["generic_read_desc_name" if source == "reading" else "generic_desc_name"] = str(p.get("descName", "none"))
            flags = (_capi.GEN_SOFT if int(p.get("useSoftThreshold", 0)) else 0) | (_capi.GEN_LARGER if int(p.get("useLargerThan", 1)) else 0) | \
                    (_capi.GEN_SOURCE_READING if source == "reading" else 0)
            outs.append((_capi.OUT_GENERICDESCRIPTOR, float(p.get("threshold", 0.1)), flags, 0.0))
            continue
        if name == "VarTrimmedDistOutlierFilter":
            # upstream defaults: minRatio 0.05, maxRatio 0.99, lambda 0.95
            for key in p:
                if key not in ("minRatio", "maxRatio", "lambda"):
                    raise InvalidParameter(f"{name}: unknown parameter {key}")
            outs.append((_capi.OUT_VARTRIMMEDDIST, float(p.get("minRatio", 0.05)), 0, float(p.get("maxRatio", 0.99)), float(p.get("lambda", 0.95))))
            continue
        if name == "RobustOutlierFilter":
            # upstream defaults: robustFct cauchy, tuning 1, scaleEstimator mad, nbIterationForScale 0, distanceType point2point
            for key in p:
                if key not in ("robustFct", "tuning", "scaleEstimator", "nbIterationForScale", "distanceType", "approximation"):
                    raise InvalidParameter(f"{name}: unknown parameter {key}")
            fct, sc, dt = str(p.get("robustFct", "cauchy")), str(p.get("scaleEstimator", "mad")), str(p.get("distanceType", "point2point"))
            if fct not in _capi.ROBUST_FCT or dt not in _capi.ROBUST_DIST or sc not in ("none", "mad", "berg", "std"):
                raise InvalidParameter(f"{name}: unknown robustFct / scaleEstimator / distanceType")
            apx = float(str(p.get("approximation", "inf")).replace(".inf", "inf"))
            if not apx > 0.0:
                raise InvalidParameter(f"{name}: approximation must be > 0")
            outs.append((_capi.OUT_ROBUST, float(p.get("tuning", 1.0)),
                         _capi.ROBUST_FCT[fct] | (_capi.ROBUST_SCALE[sc] << 4) | (_capi.ROBUST_DIST[dt] << 8), float(int(p.get("nbIterationForScale", 0))), apx))
            continue
        if name not in _OUTLIER_NAMES:
            raise InvalidParameter(f"unknown outlier filter {name}")
        t, pname, default = _OUTLIER_NAMES[name]
        for key in p:
            if key != pname:
                raise InvalidParameter(f"{name}: unknown parameter {key}")
        outs.append((t, float(p.get(pname, default))))
    kw["outliers"] = outs

    name, p = single(chain.get("errorMinimizer", "PointToPlaneErrorMinimizer"), "errorMinimizer")
    if name not in _MINIMIZERS:
        raise InvalidParameter(f"unknown error minimizer {name}")
    kw["minimizer"] = _MINIMIZERS[name]
    kw["force_4dof"] = 1 if (name == "PointToPlaneErrorMinimizer" and int(p.get("force4DOF", 0))) else 0
    kw["force_2d"] = 1 if (name == "PointToPlaneErrorMinimizer" and int(p.get("force2D", 0))) else 0
    if kw["force_4dof"] and kw["force_2d"]:
        raise InvalidParameter("PointToPlaneErrorMinimizer: force2D and force4DOF exclude each other")

    kw["max_iterations"] = 40
    for node in chain.get("transformationCheckers", [{"CounterTransformationChecker": {}}]) or []:
        name, p = single(node, "transformation checker")
        if name == "CounterTransformationChecker":
            kw["max_iterations"] = int(p.get("maxIterationCount", 40))
        elif name == "DifferentialTransformationChecker":
            kw["use_differential"] = 1
            kw["min_diff_rot"] = float(p.get("minDiffRotErr", 0.001))
            kw["min_diff_trans"] = float(p.get("minDiffTransErr", 0.001))
            kw["smooth_length"] = int(p.get("smoothLength", 3))
        elif name == "BoundTransformationChecker":
            kw["use_bound"] = 1
            kw["max_rot_norm"] = float(p.get("maxRotationNorm", 1))
            kw["max_trans_norm"] = float(p.get("maxTranslationNorm", 1))
        else:
            raise InvalidParameter(f"unknown transformation checker {name}")
    kw.update(engine)
    kw.pop("generic_desc_name", None)  # the caller hands the descriptor over with ICPSequence.setMapScalar
    kw.pop("generic_read_desc_name", None)  # ... and the reading's with ICPSequence.setReadingScalar
    return default_config(**kw)


class _ErrorMinimizerView:
    def __init__(self, owner):
        self._o = owner

    def getOverlap(self):
        """upstream: the sensor-noise count when the reading carried `simpleSensorNoise` and `normals` (ICPSequence.setReadingSensorNoise),
        the weighted ratio of used points otherwise"""
        sn = float(self._o.stats.sensor_noise_overlap)
        return sn if sn >= 0.0 else float(self._o.stats.weighted_point_used_ratio)

    def getPointUsedRatio(self):
        return float(self._o.stats.point_used_ratio)

    def getWeightedPointUsedRatio(self):
        return float(self._o.stats.weighted_point_used_ratio)


class ICPSequence:
    """``PM::ICPSequence`` backed by one icpmi handle (one GPU, one stream)."""

    def __init__(self, cfg=None, **kw):
        self._lib = _capi.load()
        self.cfg = cfg if cfg is not None else default_config(**kw)
        h = C.c_void_p()
        st = self._lib.icpmi_create(C.byref(self.cfg), C.byref(h))
        if st != _capi.ICPMI_OK:
            msg = self._lib.icpmi_last_error(None).decode()
            raise _STATUS_EXC.get(st, RuntimeError)(msg)
        self._h = h
        self.stats = _capi.Stats()
        self.errorMinimizer = _ErrorMinimizerView(self)

    @classmethod
    def loadFromYaml(cls, chain, **engine):
        return cls(config_from_yaml_chain(chain, **engine))

    def setConfig(self, cfg=None, **kw):
        """Swap the chain of the live handle (icpmi_set_config): the map and the handle stay valid."""
        new = cfg if cfg is not None else default_config(device=self.cfg.device, **kw)
        self._check(self._lib.icpmi_set_config(self._h, C.byref(new)))
        self.cfg = new

    def close(self):
        if getattr(self, "_h", None):
            self._lib.icpmi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != _capi.ICPMI_OK:
            msg = self._lib.icpmi_last_error(self._h).decode()
            raise _STATUS_EXC.get(st, RuntimeError)(msg)

    # ---- PM::ICPSequence surface ----
    def setMap(self, cloud, normals=None):
        cloud = _f32c(cloud, 4)
        acc = C.c_int32(0)
        nptr = None
        if normals is not None:
            normals = _f32c(normals, 3)
            if normals.shape[0] != cloud.shape[0]:
                raise InvalidParameter("normals / cloud size mismatch")
            nptr = normals.ctypes.data
        self._check(self._lib.icpmi_set_map(self._h, cloud.ctypes.data, cloud.shape[0], nptr, C.byref(acc)))
        return bool(acc.value)

    def setMapDev(self, d_cloud_ptr, m, d_normals_ptr=None):
        acc = C.c_int32(0)
        self._check(self._lib.icpmi_set_map_dev(self._h, d_cloud_ptr, m, d_normals_ptr, C.byref(acc)))
        return bool(acc.value)

    def hasMap(self):
        return bool(self._lib.icpmi_has_map(self._h))

    def getMapMean(self):
        out = (C.c_float * 3)()
        self._check(self._lib.icpmi_get_map_mean(self._h, out))
        return np.array(out[:], dtype=np.float32)

    def __call__(self, scan, scan_normals=None):
        scan = _f32c(scan, 4)
        nptr = None
        if scan_normals is not None:
            scan_normals = _f32c(scan_normals, 3)
            nptr = scan_normals.ctypes.data
        T = (C.c_float * 16)()
        st = self._lib.icpmi_register(self._h, scan.ctypes.data, scan.shape[0], nptr, T, C.byref(self.stats))
        self._check(st)
        return _T_from_c(T[:])

    def setReadingSensorNoise(self, noise):
        """icpmi_set_reading_sensor_noise: the `simpleSensorNoise` row of the NEXT reading (one shot; that call must bring scan_normals)."""
        nz = None if noise is None else np.ascontiguousarray(noise, dtype=np.float32).ravel()
        self._check(self._lib.icpmi_set_reading_sensor_noise(self._h, None if nz is None else nz.ctypes.data, 0 if nz is None else nz.shape[0]))

    def setReadingScalar(self, scalar):
        """icpmi_set_reading_scalar: the descriptor row GenericDescriptorOutlierFilter{source: reading} reads, for the NEXT reading (one shot)."""
        sc = None if scalar is None else np.ascontiguousarray(scalar, dtype=np.float32).ravel()
        self._check(self._lib.icpmi_set_reading_scalar(self._h, None if sc is None else sc.ctypes.data, 0 if sc is None else sc.shape[0]))

    def registerDev(self, d_scan_ptr, n, fixed_iterations=0, d_normals_ptr=None):
        T = (C.c_float * 16)()
        if fixed_iterations > 0:
            st = self._lib.icpmi_register_fixed_dev(self._h, d_scan_ptr, n, d_normals_ptr, fixed_iterations, T, C.byref(self.stats))
        else:
            st = self._lib.icpmi_register_dev(self._h, d_scan_ptr, n, d_normals_ptr, T, C.byref(self.stats))
        self._check(st)
        return _T_from_c(T[:])

    def registerBatchDev(self, d_scan_ptrs, ns, fixed_iterations=0):
        """icpmi_register_batch_dev: B readings (device pointers, sizes) against the map in one launch sequence.
        Returns (list of 4x4 corrections, list of Stats, list of status codes)."""
        B = len(d_scan_ptrs)
        ptrs = (C.c_void_p * B)(*[int(p) for p in d_scan_ptrs])
        nn = (C.c_int64 * B)(*[int(n) for n in ns])
        T = (C.c_float * (16 * B))()
        stats = (_capi.Stats * B)()
        status = (C.c_int * B)()
        self._check(self._lib.icpmi_register_batch_dev(self._h, B, ptrs, nn, fixed_iterations, T, stats, status))
        self.batch_stats = stats
        return [_T_from_c(T[16 * b:16 * b + 16]) for b in range(B)], [stats[b] for b in range(B)], [int(status[b]) for b in range(B)]

    # ---- stage-level entry points ----
    def transform(self, T, cloud, normals=None):
        cloud = _f32c(cloud, 4)
        out = np.empty_like(cloud)
        Tc = _T_to_c(T)
        nin = nout = None
        outn = None
        if normals is not None:
            normals = _f32c(normals, 3)
            outn = np.empty_like(normals)
            nin, nout = normals.ctypes.data, outn.ctypes.data
        st = self._lib.icpmi_transform(self._h, Tc.ctypes.data_as(C.POINTER(C.c_float)), cloud.ctypes.data, cloud.shape[0],
                                       out.ctypes.data, nin, nout)
        if st == _capi.ERR_INVALID_ARG:
            raise TransformationError(self._lib.icpmi_last_error(self._h).decode())
        self._check(st)
        return (out, outn) if normals is not None else out

    def knn(self, queries_centred, k=1, max_dist=math.inf, allow_self=True):
        q = _f32c(queries_centred, 4)
        ids = np.empty((q.shape[0], k), dtype=np.int32)
        d2 = np.empty((q.shape[0], k), dtype=np.float32)
        self._check(self._lib.icpmi_knn(self._h, q.ctypes.data, q.shape[0], k, max_dist, int(allow_self), ids.ctypes.data, d2.ctypes.data))
        return ids, d2

    def outlierWeights(self, d2, ids=None, read_normals=None):
        d2 = np.ascontiguousarray(d2, dtype=np.float32)
        n, k = d2.shape
        w = np.empty_like(d2)
        lim = C.c_float(-1)
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.int32)
            idp = ids.ctypes.data
        rnp = None
        if read_normals is not None:
            read_normals = _f32c(read_normals, 3)
            rnp = read_normals.ctypes.data
        self._check(self._lib.icpmi_outlier_weights(self._h, d2.ctypes.data, idp, k, n, rnp, w.ctypes.data, C.byref(lim)))
        return w, float(lim.value)

    def minimizeStep(self, reading_centred, T_iter=None):
        r = _f32c(reading_centred, 4)
        T = (C.c_float * 16)()
        sums = (C.c_double * 32)()
        tptr = None
        if T_iter is not None:
            Tc = _T_to_c(T_iter)
            tptr = Tc.ctypes.data
        self._check(self._lib.icpmi_minimize_step(self._h, r.ctypes.data, r.shape[0], tptr, T, sums, C.byref(self.stats)))
        return _T_from_c(T[:]), np.array(sums[:])

    def surfaceNormals(self, cloud, knn=5, with_densities=False, with_matched_ids=False, with_mean_dist=False):
        """normals [, densities] [, matched ids (m, knn) int32] [, mean distance (m,)] -- SurfaceNormalDataPointsFilter with
        keepDensities / keepMatchedIds / keepMeanDist."""
        c = _f32c(cloud, 4)
        out = np.empty((c.shape[0], 3), dtype=np.float32)
        if with_matched_ids or with_mean_dist:
            m = c.shape[0]
            dens = np.empty(m, dtype=np.float32) if with_densities else None
            ids = np.empty((m, knn), dtype=np.int32) if with_matched_ids else None
            md = np.empty(m, dtype=np.float32) if with_mean_dist else None
            ptr = lambda a: a.ctypes.data if a is not None else None
            self._check(self._lib.icpmi_surface_normals_ex2(self._h, c.ctypes.data, m, knn, out.ctypes.data, ptr(dens), ptr(ids), ptr(md)))
            return tuple(x for x in (out, dens, ids, md) if x is not None)
        if not with_densities:
            self._check(self._lib.icpmi_surface_normals(self._h, c.ctypes.data, c.shape[0], knn, out.ctypes.data))
            return out
        dens = np.empty(c.shape[0], dtype=np.float32)
        self._check(self._lib.icpmi_surface_normals_ex(self._h, c.ctypes.data, c.shape[0], knn, out.ctypes.data, dens.ctypes.data))
        return out, dens

    def surfaceNormalsEigen(self, cloud, knn=5):
        """(normals, eigenvalues (m, 3) ascending, eigenvectors (m, 9): entry 3 k + j = component k of eigenvector j) --
        SurfaceNormalDataPointsFilter with keepEigenValues / keepEigenVectors and sortEigen: 1 (icpmi_surface_normals_ex3)."""
        c = _f32c(cloud, 4); m = c.shape[0]
        out = np.empty((m, 3), dtype=np.float32); ev = np.empty((m, 3), dtype=np.float32); evec = np.empty((m, 9), dtype=np.float32)
        self._check(self._lib.icpmi_surface_normals_ex3(self._h, c.ctypes.data, m, knn, out.ctypes.data, None, None, None, ev.ctypes.data, evec.ctypes.data))
        return out, ev, evec

    def pointDistanceKeep(self, map_cloud, input_cloud, min_dist):
        m = _f32c(map_cloud, 4)
        i = _f32c(input_cloud, 4)
        keep = np.empty(i.shape[0], dtype=np.uint8)
        self._check(self._lib.icpmi_point_distance_keep(self._h, m.ctypes.data, m.shape[0], i.ctypes.data, i.shape[0], min_dist, keep.ctypes.data))
        return keep.astype(bool)

    def voxelKeepFirst(self, cloud, edge):
        """Lattice stand-in of OctreeGridDataPointsFilter{maxSizeByNode: edge, samplingMethod: 0}
        (OctreeMapperModule.cpp:35-39): mask of the first point of every occupied voxel."""
        c = _f32c(cloud, 4)
        keep = np.empty(c.shape[0], dtype=np.uint8)
        self._check(self._lib.icpmi_voxel_keep_first(self._h, c.ctypes.data, c.shape[0], edge, keep.ctypes.data))
        return keep.astype(bool)

    def filterPoints(self, cloud, filters):
        """Mapper::applyInputFilters as one pass: filters = [("distance_limit", dim, dist, remove_inside) |
        ("bounding_box", (xmin, ymin, zmin), (xmax, ymax, zmax), remove_inside)]; returns the keep mask."""
        from . import _capi
        c = _f32c(cloud, 4)
        arr = (_capi.PointFilter * max(1, len(filters)))()
        for k, flt in enumerate(filters):
            if flt[0] == "distance_limit":
                arr[k].type = 0; arr[k].i = int(flt[1]); arr[k].f[0] = flt[2]; arr[k].f[1] = 1.0 if flt[3] else 0.0
            elif flt[0] == "bounding_box":
                arr[k].type = 1; arr[k].i = 1 if flt[3] else 0
                for r in range(3):
                    arr[k].f[r] = flt[1][r]; arr[k].f[3 + r] = flt[2][r]
            else:
                raise InvalidParameter("unknown point filter " + str(flt[0]))
        keep = np.empty(c.shape[0], dtype=np.uint8)
        self._check(self._lib.icpmi_filter_points(self._h, c.ctypes.data, c.shape[0], arr, len(filters), keep.ctypes.data))
        return keep.astype(bool)

    def samplingSurfaceNormal(self, cloud, ratio=0.5, knn=7, max_box_dim=math.inf, seed=1):
        """icpmi_sampling_surface_normal: (kept indices in box order, their normals) -- SamplingSurfaceNormalDataPointsFilter, samplingMethod 0."""
        c = _f32c(cloud, 4)
        order = np.empty(c.shape[0], dtype=np.int32)
        nrm = np.empty((c.shape[0], 3), dtype=np.float32)
        n_out = C.c_int64(0)
        self._check(self._lib.icpmi_sampling_surface_normal(self._h, c.ctypes.data, c.shape[0], ratio, knn, max_box_dim, seed, order.ctypes.data,
                                                            nrm.ctypes.data, C.byref(n_out)))
        return order[:n_out.value].copy(), nrm[:n_out.value].copy()

    def samplingSurfaceNormalBoxes(self, cloud, knn=7, max_box_dim=math.inf):
        """icpmi_sampling_surface_normal_ex with samplingMethod 1: one point per surviving box -- (first member index, normal, mean of the box,
        member start, member count, members in index order)."""
        c = _f32c(cloud, 4); n = c.shape[0]
        order = np.empty(n, dtype=np.int32); nrm = np.empty((n, 3), dtype=np.float32); mean = np.empty((n, 3), dtype=np.float32)
        ms = np.empty(n, dtype=np.int32); mc = np.empty(n, dtype=np.int32); mem = np.empty(n, dtype=np.int32)
        n_out = C.c_int64(0)
        self._check(self._lib.icpmi_sampling_surface_normal_ex(self._h, c.ctypes.data, n, 1.0, knn, max_box_dim, 1, 1, order.ctypes.data, nrm.ctypes.data,
                                                               C.byref(n_out), mean.ctypes.data, ms.ctypes.data, mc.ctypes.data, mem.ctypes.data))
        k = n_out.value
        tot = int(ms[k - 1] + mc[k - 1]) if k else 0
        return order[:k].copy(), nrm[:k].copy(), mean[:k].copy(), ms[:k].copy(), mc[:k].copy(), mem[:tot].copy()

    def octreeSample(self, cloud, max_size, max_points=1, method=0, with_leaves=False):
        """OctreeGridDataPointsFilter: indices of the kept points in leaf-visiting order (+ the leaf ordinal of every point)"""
        c = _f32c(cloud, 4)
        order = np.empty(c.shape[0], dtype=np.int32)
        leaf = np.empty(c.shape[0], dtype=np.int32) if with_leaves else None
        m = C.c_int64(0)
        self._check(self._lib.icpmi_octree_sample(self._h, c.ctypes.data, c.shape[0], max_size, max_points, method, order.ctypes.data,
                                                  None if leaf is None else leaf.ctypes.data, C.byref(m)))
        return (order[:m.value].copy(), leaf) if with_leaves else order[:m.value].copy()

    def voxelKeep(self, cloud, edge, method=0):
        """Same lattice, representative by `samplingMethod`: 0 first point, 1 pseudo-random point (smallest fmix32 of the index)."""
        c = _f32c(cloud, 4)
        keep = np.empty(c.shape[0], dtype=np.uint8)
        self._check(self._lib.icpmi_voxel_keep(self._h, c.ctypes.data, c.shape[0], edge, method, keep.ctypes.data))
        return keep.astype(bool)

    @staticmethod
    def _mapOps(modules, post):
        """[(name, params...)] -> MapOp array.  Names: 'point_distance' (min_dist), 'dynamic_points' (7 parameters in
        icpmi_dynpts_params order), 'voxel' (edge, method), 'surface_normals' (knn), 'cut_scalar' (threshold, use_larger_than)."""
        from . import _capi
        ops = (_capi.MapOp * (len(modules) + len(post)))()
        for j, item in enumerate(list(modules) + list(post)):
            name, args = item[0], item[1:]
            op = ops[j]
            if name == "point_distance":
                op.type = _capi.MOP_POINT_DISTANCE; op.f[0] = args[0]
            elif name == "dynamic_points":
                op.type = _capi.MOP_DYNAMIC_POINTS
                for r in range(7):
                    op.f[r] = args[r]
            elif name == "voxel":
                op.type = _capi.MOP_VOXEL; op.f[0] = args[0]; op.i = int(args[1]) if len(args) > 1 else 0
            elif name == "octree":      # (maxSizeByNode, samplingMethod = 0, maxPointByNode = 1)
                op.type = _capi.MOP_OCTREE; op.f[0] = args[0]; op.i = int(args[1]) if len(args) > 1 else 0
                op.f[1] = float(args[2]) if len(args) > 2 else 1.0
            elif name == "surface_normals":
                op.type = _capi.MOP_SURFACE_NORMALS; op.i = int(args[0])
            elif name == "cut_scalar":
                op.type = _capi.MOP_CUT_SCALAR; op.f[0] = args[0]; op.i = int(args[1]) if len(args) > 1 else 1
            else:
                raise InvalidParameter("unknown map operator " + name)
        return ops

    def mapUpdateChain(self, scan_in_map_frame, modules, post=(), scan_scalar=None, scan_normals=None, to_sensor=None, staged_correction=None,
                       with_prefix=False, want_src=True, from_sensor=None):
        """Map::updateLocalPointCloud (Map.cpp:502-534) for a whole module chain + post filters on the resident map.
        Returns (src, m): new map point j was point src[j] of [old map ; scan].  With staged_correction the scan is the
        one staged by registerWithPrior (scan_in_map_frame is ignored)."""
        ops = self._mapOps(modules, post)
        m_old = C.c_int64(0)
        self._check(self._lib.icpmi_get_map(self._h, None, None, 0, C.byref(m_old)))
        ss = None if scan_scalar is None else np.ascontiguousarray(scan_scalar, dtype=np.float32)
        Ts = None if to_sensor is None else _T_to_c(to_sensor)
        Tf = None if from_sensor is None else _T_to_c(from_sensor)   # pose: the post filters then run in the sensor frame (Map.cpp:523-525)
        Tfp = None if Tf is None else Tf.ctypes.data
        new_m = C.c_int64(0)
        head = C.c_int64(0)
        hp = C.byref(head) if with_prefix else None
        if not want_src:   # a caller without further descriptors does not need the provenance vector (3 MB per update at 800 k points)
            Tc = None if staged_correction is None else _T_to_c(staged_correction)
            if staged_correction is not None:
                self._check(self._lib.icpmi_map_update_chain_staged(self._h, Tc.ctypes.data, None if ss is None else ss.ctypes.data,
                                                                    None if Ts is None else Ts.ctypes.data, Tfp, ops, len(ops), len(modules),
                                                                    None, 0, None, C.byref(new_m)))
            else:
                sc = _f32c(scan_in_map_frame, 4)
                sn = None if scan_normals is None else _f32c(scan_normals, 3)
                self._check(self._lib.icpmi_map_update_chain(self._h, sc.ctypes.data, sc.shape[0], None if sn is None else sn.ctypes.data,
                                                             None if ss is None else ss.ctypes.data, None if Ts is None else Ts.ctypes.data, Tfp,
                                                             ops, len(ops), len(modules), None, 0, None, C.byref(new_m)))
            return None, int(new_m.value)
        if staged_correction is not None:
            n = self._staged_n
            src = np.empty(m_old.value + max(1, len(modules)) * n + 1, dtype=np.int32)
            Tc = _T_to_c(staged_correction)
            self._check(self._lib.icpmi_map_update_chain_staged(self._h, Tc.ctypes.data, None if ss is None else ss.ctypes.data,
                                                                None if Ts is None else Ts.ctypes.data, Tfp, ops, len(ops), len(modules),
                                                                src.ctypes.data, src.shape[0], hp, C.byref(new_m)))
        else:
            sc = _f32c(scan_in_map_frame, 4)
            sn = None if scan_normals is None else _f32c(scan_normals, 3)
            n = sc.shape[0]
            src = np.empty(m_old.value + max(1, len(modules)) * n + 1, dtype=np.int32)
            self._check(self._lib.icpmi_map_update_chain(self._h, sc.ctypes.data, n, None if sn is None else sn.ctypes.data,
                                                         None if ss is None else ss.ctypes.data, None if Ts is None else Ts.ctypes.data, Tfp,
                                                         ops, len(ops), len(modules), src.ctypes.data, src.shape[0], hp, C.byref(new_m)))
        if with_prefix:  # the head was not written by the library: it is the identity
            src[:head.value] = np.arange(head.value, dtype=np.int32)
            return src[:new_m.value].copy(), int(new_m.value), int(head.value)
        return src[:new_m.value].copy(), int(new_m.value)

    def setMapScalar(self, scalar):
        s = np.ascontiguousarray(scalar, dtype=np.float32)
        self._check(self._lib.icpmi_set_map_scalar(self._h, s.ctypes.data, s.shape[0]))

    def getMapScalar(self):
        m = C.c_int64(0)
        self._check(self._lib.icpmi_get_map(self._h, None, None, 0, C.byref(m)))
        out = np.empty(m.value, dtype=np.float32)
        self._check(self._lib.icpmi_get_map_scalar(self._h, out.ctypes.data, out.shape[0]))
        return out

    def dynamicPointsUpdate(self, to_sensor, input_cloud, map_cloud, map_normals, prob_dynamic, threshold_dynamic=0.6, alpha=0.8,
                            beta=0.99, beam_half_angle=0.01, epsilon_a=0.01, epsilon_d=0.01, sensor_max_range=200.0):
        """DynamicPointsMapperModule::inPlaceUpdateMap (DynamicPointsMapperModule.cpp:34-172): returns the
        updated `probabilityDynamic` of the map.  `to_sensor` = pose^-1 (row-major 4x4 numpy)."""
        prm = np.array([threshold_dynamic, alpha, beta, beam_half_angle, epsilon_a, epsilon_d, sensor_max_range], dtype=np.float32)
        T = _T_to_c(to_sensor)
        i = _f32c(input_cloud, 4); m = _f32c(map_cloud, 4); nn = _f32c(map_normals, 3)
        out = np.ascontiguousarray(prob_dynamic, dtype=np.float32).copy()
        if out.shape[0] != m.shape[0] or nn.shape[0] != m.shape[0]:
            raise InvalidParameter("dynamicPointsUpdate: map, normals and probabilities must have the same length")
        self._check(self._lib.icpmi_dynamic_points_update(self._h, prm.ctypes.data, T.ctypes.data,
                                                          i.ctypes.data, i.shape[0], m.ctypes.data, nn.ctypes.data, m.shape[0], out.ctypes.data))
        return out

    def mapUpdatePointDistance(self, scan_in_map_frame, min_dist, normals_knn=0, scan_normals=None, return_keep=False):
        """Map::updateLocalPointCloud for the PointDistance chain on the resident map (Map.cpp:502-534): returns
        (points appended, points in the map afterwards[, keep mask]).  Only the scan is uploaded."""
        sc = _f32c(scan_in_map_frame, 4)
        sn = None if scan_normals is None else _f32c(scan_normals, 3)
        app = C.c_int64(0); m = C.c_int64(0)
        keep = np.zeros(sc.shape[0], dtype=np.uint8) if return_keep else None
        self._check(self._lib.icpmi_map_update_point_distance(self._h, sc.ctypes.data, sc.shape[0], None if sn is None else sn.ctypes.data,
                                                              min_dist, normals_knn, None if keep is None else keep.ctypes.data,
                                                              C.byref(app), C.byref(m)))
        return (int(app.value), int(m.value), keep.astype(bool)) if return_keep else (int(app.value), int(m.value))

    def registerWithPrior(self, scan_sensor_frame, prior):
        """Mapper::processInput's first half with the scan staged once (Mapper.cpp:197,213): the scan goes into the map
        frame by `prior` on the device, is registered, and stays in HBM for mapUpdateStaged.  Returns the correction."""
        sc = _f32c(scan_sensor_frame, 4)
        P = _T_to_c(prior)
        T = (C.c_float * 16)()
        self._check(self._lib.icpmi_register_prior(self._h, sc.ctypes.data, sc.shape[0], P.ctypes.data, T, C.byref(self.stats)))
        self._staged_n = sc.shape[0]
        return _T_from_c(T[:])

    def registerWithPriorDev(self, d_scan_ptr, n, prior):
        """icpmi_register_prior_dev: registerWithPrior for a sensor-frame scan that is already in HBM."""
        P = _T_to_c(prior)
        T = (C.c_float * 16)()
        self._check(self._lib.icpmi_register_prior_dev(self._h, d_scan_ptr, n, P.ctypes.data, T, C.byref(self.stats)))
        self._staged_n = n
        return _T_from_c(T[:])

    def mapUpdateStaged(self, correction, min_dist, normals_knn=0, return_keep=False):
        """Second half (Mapper.cpp:221 + Map::updateLocalPointCloud): the staged cloud moved by `correction`, then the
        PointDistance update on the resident map."""
        Tc = _T_to_c(correction)
        app = C.c_int64(0); m = C.c_int64(0)
        n = C.c_int64(0)
        keep = None
        if return_keep:
            # the staged scan's size is the size of the last registerWithPrior reading
            keep = np.zeros(self._staged_n, dtype=np.uint8)
        self._check(self._lib.icpmi_map_update_staged(self._h, Tc.ctypes.data, min_dist, normals_knn, None if keep is None else keep.ctypes.data,
                                                      C.byref(app), C.byref(m)))
        return (int(app.value), int(m.value), keep.astype(bool)) if return_keep else (int(app.value), int(m.value))

    def stagedPointDistanceKeep(self, correction, min_dist):
        """Keep mask of the staged scan (moved by `correction`) against the resident map, map untouched; returns
        (mask, moved cloud).  The scan-sharded loop exchanges the accepted points before any replica appends them."""
        Tc = _T_to_c(correction)
        keep = np.zeros(self._staged_n, dtype=np.uint8)
        placed = np.empty((self._staged_n, 4), dtype=np.float32)
        self._check(self._lib.icpmi_staged_point_distance_keep(self._h, Tc.ctypes.data, min_dist, keep.ctypes.data, placed.ctypes.data))
        return keep.astype(bool), placed

    # ---- scan-sharded mapping: RCCL communicator + device-resident map-growth epoch ----
    @staticmethod
    def commUniqueId():
        """rank 0: a fresh communicator id (128 bytes) to hand to the other ranks"""
        lib = _capi.load()
        buf = (C.c_char * 128)()
        st = lib.icpmi_comm_get_unique_id(buf)
        if st != _capi.ICPMI_OK:
            raise HipError(lib.icpmi_last_error(None).decode())
        return bytes(buf)

    def commInit(self, unique_id, n_ranks, rank):
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        self._check(self._lib.icpmi_comm_init(self._h, buf, n_ranks, rank))

    def commInfo(self):
        """icpmi_comm_info: (ranks, rank, kind) as the communicator itself reports them; kind 0 none / 1 RCCL / 2 loopback."""
        a, b, k = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(self._lib.icpmi_comm_info(self._h, C.byref(a), C.byref(b), C.byref(k)))
        return int(a.value), int(b.value), int(k.value)

    def commDestroy(self):
        self._check(self._lib.icpmi_comm_destroy(self._h))

    def stagedMergeAllGather(self, correction, min_dist, normals_knn=0, return_merged=False, merged_capacity=None):
        """icpmi_staged_merge_allgather: (accepted on this rank, appended on every replica, new map size[, merged points]).
        correction None: this rank contributes nothing (empty scan / failed registration) and still takes part in the exchange.
        The merged set is fetched with icpmi_staged_merged_points once its size is known (merged_capacity only sizes the
        optional in-call copy, which may be smaller than the set)."""
        Tc = None if correction is None else _T_to_c(correction)
        acc, app, new_m, mn = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        out = None
        if return_merged and merged_capacity is not None:
            out = np.empty((int(merged_capacity), 4), dtype=np.float32)
        self._check(self._lib.icpmi_staged_merge_allgather(self._h, None if Tc is None else Tc.ctypes.data, min_dist, normals_knn, C.byref(acc),
                                                           C.byref(app), C.byref(new_m), None if out is None else out.ctypes.data,
                                                           0 if out is None else out.shape[0], C.byref(mn)))
        if return_merged:
            if out is None or mn.value > out.shape[0]:
                out = self.stagedMergedPoints()
            return int(acc.value), int(app.value), int(new_m.value), out[:mn.value].copy()
        return int(acc.value), int(app.value), int(new_m.value)

    def stagedMergedPoints(self):
        """icpmi_staged_merged_points: the merged set of the last epoch."""
        n = C.c_int64(0)
        self._check(self._lib.icpmi_staged_merged_points(self._h, None, 0, C.byref(n)))
        out = np.empty((n.value, 4), dtype=np.float32)
        if n.value:
            self._check(self._lib.icpmi_staged_merged_points(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def stagedBinCells(self, cell_size=20.0, capacity=4096):
        """icpmi_staged_bin_cells: the last epoch's merged set binned on the device and appended to the handle's cell log;
        returns (ijk [c, 3] int32, offsets [c] int64, counts [c] int64), cells in the order of their first point."""
        ijk = np.empty((int(capacity), 3), dtype=np.int32)
        off = np.empty(int(capacity), dtype=np.int64)
        cnt = np.empty(int(capacity), dtype=np.int64)
        nc = C.c_int64(0)
        self._check(self._lib.icpmi_staged_bin_cells(self._h, cell_size, ijk.ctypes.data, off.ctypes.data, cnt.ctypes.data, int(capacity), C.byref(nc)))
        return ijk[:nc.value].copy(), off[:nc.value].copy(), cnt[:nc.value].copy()

    def cellLogConfigure(self, cell_size):
        """icpmi_cell_log_configure: cell_size > 0 -- every epoch enqueues the binning of its merged set itself; stagedBinCells(cell_size) collects it."""
        self._check(self._lib.icpmi_cell_log_configure(self._h, float(cell_size)))

    def cellLogSize(self):
        n = C.c_int64(0)
        self._check(self._lib.icpmi_cell_log_read(self._h, 0, 0, None, C.byref(n)))
        return int(n.value)

    def cellLogRead(self, offset, count):
        """icpmi_cell_log_read: `count` points of the device-resident cell log from `offset`."""
        out = np.empty((int(count), 4), dtype=np.float32)
        if count:
            self._check(self._lib.icpmi_cell_log_read(self._h, int(offset), int(count), out.ctypes.data, None))
        return out

    def cellLogClear(self):
        self._check(self._lib.icpmi_cell_log_clear(self._h))

    def stageDiscard(self):
        """icpmi_stage_discard: drop the scan staged by registerWithPrior."""
        self._check(self._lib.icpmi_stage_discard(self._h))
        self._staged_n = 0

    def getMap(self, with_normals=False):
        """The resident map in the caller's order (Map::getLocalPointCloud, Map.cpp:536-540)."""
        m = C.c_int64(0)
        self._check(self._lib.icpmi_get_map(self._h, None, None, 0, C.byref(m)))
        out = np.empty((m.value, 4), dtype=np.float32)
        nrm = np.empty((m.value, 3), dtype=np.float32) if with_normals else None
        if m.value:
            self._check(self._lib.icpmi_get_map(self._h, out.ctypes.data, None if nrm is None else nrm.ctypes.data, m.value, C.byref(m)))
        return (out, nrm) if with_normals else out

    def binCells(self, cloud, cell_size=20.0):
        c = _f32c(cloud, 4)
        out = np.empty((c.shape[0], 3), dtype=np.int32)
        self._check(self._lib.icpmi_bin_cells(self._h, c.ctypes.data, c.shape[0], cell_size, out.ctypes.data))
        return out

    def setStream(self, hip_stream_ptr):
        self._check(self._lib.icpmi_set_stream(self._h, hip_stream_ptr))

    def debugCounters(self):
        out = (C.c_uint64 * 24)()
        self._check(self._lib.icpmi_debug_counters(self._h, out))
        return list(out)

    def gridInfo(self):
        cell = C.c_float()
        dims = (C.c_int32 * 3)()
        nc, no = C.c_int64(), C.c_int64()
        self._check(self._lib.icpmi_get_grid_info(self._h, C.byref(cell), dims, C.byref(nc), C.byref(no)))
        return {"cell": cell.value, "dims": list(dims), "n_cells": nc.value, "n_occupied": no.value}
