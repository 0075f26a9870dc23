"""ctypes binding of libicpmi.so -- the C ABI declared in include/icpmi.h.

This is plumbing for tests and bench.py: it mirrors the header 1:1 and never computes anything
itself.  Loading fails loudly when the shared library has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C norlab_icp_mapper_amd/csrc``); there is no CPU
fallback of any kind.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libicpmi.so")

ICPMI_OK = 0
ERR_INVALID_ARG, ERR_HIP, ERR_NO_POINT_TO_MINIMIZE, ERR_NO_OUTLIER_TO_FILTER = 1, 2, 3, 4
ERR_BOUND, ERR_NAN, ERR_MISSING_NORMALS, ERR_UNSUPPORTED = 5, 6, 7, 8
MIN_IDENTITY, MIN_POINT_TO_POINT, MIN_POINT_TO_PLANE = 0, 1, 2
OUT_MAXDIST, OUT_MINDIST, OUT_MEDIANDIST, OUT_TRIMMEDDIST, OUT_SURFACENORMAL = 1, 2, 3, 4, 5
OUT_GENERICDESCRIPTOR, OUT_ROBUST, OUT_VARTRIMMEDDIST = 6, 7, 8
GEN_SOURCE_READING, GEN_SOFT, GEN_LARGER = 1, 2, 4
ROBUST_FCT = {"cauchy": 0, "welsch": 1, "sc": 2, "gm": 3, "tukey": 4, "huber": 5, "L1": 6, "student": 7}
ROBUST_SCALE = {"none": 0, "mad": 1, "berg": 2, "std": 3}
ROBUST_DIST = {"point2point": 0, "point2plane": 1}
STOP_NONE, STOP_COUNTER, STOP_DIFFERENTIAL = 0, 1, 2


class Outlier(C.Structure):
    _fields_ = [("type", C.c_int32), ("param", C.c_float), ("iparam", C.c_int32), ("param2", C.c_float), ("param3", C.c_float)]


class Config(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("knn", C.c_int32),
        ("max_dist", C.c_float),
        ("epsilon", C.c_float),
        ("n_outlier", C.c_int32),
        ("outlier", Outlier * 8),
        ("minimizer", C.c_int32),
        ("force_4dof", C.c_int32),
        ("force_2d", C.c_int32),
        ("is_2d", C.c_int32),
        ("max_iterations", C.c_int32),
        ("use_differential", C.c_int32),
        ("min_diff_rot", C.c_float),
        ("min_diff_trans", C.c_float),
        ("smooth_length", C.c_int32),
        ("use_bound", C.c_int32),
        ("max_rot_norm", C.c_float),
        ("max_trans_norm", C.c_float),
        ("grid_cell", C.c_float),
        ("use_graph", C.c_int32),
        ("profile", C.c_int32),
        ("fuse_solve", C.c_int32),
        ("knn_wg_from", C.c_int32),
        ("sel_window_off", C.c_int32),
        ("epsilon_approx", C.c_int32),
        ("reserved", C.c_int32 * 4),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("stop_reason", C.c_int32),
        ("pairs", C.c_int64),
        ("point_used_ratio", C.c_float),
        ("weighted_point_used_ratio", C.c_float),
        ("trimmed_limit", C.c_float),
        ("loop_ms", C.c_float),
        ("nn_ms_avg", C.c_float),
        ("nn_launches", C.c_int32),
        ("hard_queries", C.c_int64),
        ("reserved", C.c_int32 * 4),
        ("sensor_noise_overlap", C.c_float),
        ("reserved2", C.c_int32),
    ]


class MapOp(C.Structure):
    """icpmi_map_op: one step of the resident map-update chain."""
    _fields_ = [("type", C.c_int32), ("i", C.c_int32), ("f", C.c_float * 7)]


class PointFilter(C.Structure):
    """icpmi_point_filter: one DistanceLimit / BoundingBox predicate of the fused input-filter pass."""
    _fields_ = [("type", C.c_int32), ("i", C.c_int32), ("f", C.c_float * 6)]


MOP_POINT_DISTANCE, MOP_DYNAMIC_POINTS, MOP_VOXEL, MOP_SURFACE_NORMALS, MOP_CUT_SCALAR, MOP_OCTREE = range(6)


# every symbol include/icpmi.h declares: (name, restype, argtypes)
_P = C.c_void_p
_F = C.POINTER(C.c_float)
SYMBOLS = [
    ("icpmi_version", C.c_int32, []),
    ("icpmi_build_info", C.c_char_p, []),
    ("icpmi_trim_cache", C.c_int, []),
    ("icpmi_config_default", None, [C.POINTER(Config)]),
    ("icpmi_create", C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    ("icpmi_set_config", C.c_int, [_P, C.POINTER(Config)]),
    ("icpmi_destroy", None, [_P]),
    ("icpmi_last_error", C.c_char_p, [_P]),
    ("icpmi_set_map", C.c_int, [_P, _P, C.c_int64, _P, C.POINTER(C.c_int32)]),
    ("icpmi_set_map_dev", C.c_int, [_P, _P, C.c_int64, _P, C.POINTER(C.c_int32)]),
    ("icpmi_has_map", C.c_int32, [_P]),
    ("icpmi_get_map_mean", C.c_int, [_P, _F]),
    ("icpmi_register", C.c_int, [_P, _P, C.c_int64, _P, _F, C.POINTER(Stats)]),
    ("icpmi_set_reading_sensor_noise", C.c_int, [_P, _P, C.c_int64]),
    ("icpmi_register_dev", C.c_int, [_P, _P, C.c_int64, _P, _F, C.POINTER(Stats)]),
    ("icpmi_register_fixed_dev", C.c_int, [_P, _P, C.c_int64, _P, C.c_int32, _F, C.POINTER(Stats)]),
    ("icpmi_register_batch_dev", C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, _P, _P]),
    ("icpmi_transform", C.c_int, [_P, _F, _P, C.c_int64, _P, _P, _P]),
    ("icpmi_knn", C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_float, C.c_int32, _P, _P]),
    ("icpmi_outlier_weights", C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, _P, _P, _F]),
    ("icpmi_minimize_step", C.c_int, [_P, _P, C.c_int64, _P, _F, C.POINTER(C.c_double), C.POINTER(Stats)]),
    ("icpmi_surface_normals", C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    ("icpmi_surface_normals_ex", C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P]),
    ("icpmi_surface_normals_ex2", C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P]),
    ("icpmi_surface_normals_ex3", C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P, _P, _P]),
    ("icpmi_point_distance_keep", C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_float, _P]),
    ("icpmi_voxel_keep_first", C.c_int, [_P, _P, C.c_int64, C.c_float, _P]),
    ("icpmi_filter_points", C.c_int, [_P, _P, C.c_int64, _P, C.c_int32, _P]),
    ("icpmi_voxel_keep", C.c_int, [_P, _P, C.c_int64, C.c_float, C.c_int32, _P]),
    ("icpmi_sampling_surface_normal", C.c_int, [_P, _P, C.c_int64, C.c_float, C.c_int32, C.c_float, C.c_int32, _P, _P, _P]),
    ("icpmi_sampling_surface_normal_ex", C.c_int, [_P, _P, C.c_int64, C.c_float, C.c_int32, C.c_float, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    ("icpmi_octree_sample", C.c_int, [_P, _P, C.c_int64, C.c_float, C.c_int32, C.c_int32, _P, _P, _P]),
    ("icpmi_map_update_chain", C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int64, _P, _P]),
    ("icpmi_map_update_chain_staged", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_int64, _P, _P]),
    ("icpmi_set_map_scalar", C.c_int, [_P, _P, C.c_int64]),
    ("icpmi_get_map_scalar", C.c_int, [_P, _P, C.c_int64]),
    ("icpmi_map_update_point_distance", C.c_int, [_P, _P, C.c_int64, _P, C.c_float, C.c_int32, _P, _P, _P]),
    ("icpmi_staged_point_distance_keep", C.c_int, [_P, _P, C.c_float, _P, _P]),
    ("icpmi_get_map", C.c_int, [_P, _P, _P, C.c_int64, _P]),
    ("icpmi_comm_get_unique_id", C.c_int, [_P]),
    ("icpmi_comm_init", C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    ("icpmi_comm_destroy", C.c_int, [_P]),
    ("icpmi_comm_info", C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("icpmi_staged_merge_allgather", C.c_int, [_P, _P, C.c_float, C.c_int32, _P, _P, _P, _P, C.c_int64, _P]),
    ("icpmi_staged_merged_points", C.c_int, [_P, _P, C.c_int64, _P]),
    ("icpmi_staged_bin_cells", C.c_int, [_P, C.c_float, _P, _P, _P, C.c_int64, _P]),
    ("icpmi_cell_log_configure", C.c_int, [_P, C.c_float]),
    ("icpmi_cell_log_read", C.c_int, [_P, C.c_int64, C.c_int64, _P, _P]),
    ("icpmi_cell_log_clear", C.c_int, [_P]),
    ("icpmi_stage_discard", C.c_int, [_P]),
    ("icpmi_register_prior", C.c_int, [_P, _P, C.c_int64, _P, _F, C.POINTER(Stats)]),
    ("icpmi_set_reading_scalar", C.c_int, [_P, _P, C.c_int64]),
    ("icpmi_register_prior_dev", C.c_int, [_P, _P, C.c_int64, _P, _F, C.POINTER(Stats)]),
    ("icpmi_map_update_staged", C.c_int, [_P, _P, C.c_float, C.c_int32, _P, _P, _P]),
    ("icpmi_dynamic_points_update", C.c_int, [_P, _P, _P, _P, C.c_int64, _P, _P, C.c_int64, _P]),
    ("icpmi_bin_cells", C.c_int, [_P, _P, C.c_int64, C.c_float, _P]),
    ("icpmi_set_stream", C.c_int, [_P, _P]),
    ("icpmi_debug_counters", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("icpmi_debug_minstd_nth", C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("icpmi_get_grid_info", C.c_int, [_P, _F, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
]

_lib = None


def load():
    """Load libicpmi.so and type every exported symbol. Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64 / libhsa-runtime64.
    # If libicpmi.so pulled in /opt/rocm's copy first, a later torch.cuda init would find the GPU
    # already owned ("No HIP GPUs are available").  Importing torch first makes the dynamic linker
    # resolve libicpmi.so's libamdhip64.so.7 dependency to the copy torch already loaded.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing only; the library itself does not need it
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (__graft_entry__.build()). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
