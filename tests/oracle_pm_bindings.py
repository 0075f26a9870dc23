"""ctypes binding of oracle/_ref/liboracle_pm.so -- the REAL libpointmatcher PM::ICPSequence behind a C entry point
(oracle/oracle_pm.cpp).  Test infrastructure only, and only where libpointmatcher is installed (`make -C oracle
oracle_pm` builds the library then; this container and, so far, the GPU box have neither libpointmatcher nor libnabo)."""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(_ROOT, "oracle", "_ref", "liboracle_pm.so")

YAML_CHAINS = {
    "p2p": """
matcher:
  KDTreeMatcher:
    knn: 1
    maxDist: 2.0
    epsilon: 0
outlierFilters:
  - TrimmedDistOutlierFilter:
      ratio: 0.85
errorMinimizer:
  PointToPointErrorMinimizer
transformationCheckers:
  - CounterTransformationChecker:
      maxIterationCount: {iters}
inspector:
  NullInspector
logger:
  NullLogger
""",
    "p2plane": """
matcher:
  KDTreeMatcher:
    knn: 1
    maxDist: 2.0
    epsilon: 0
outlierFilters:
  - TrimmedDistOutlierFilter:
      ratio: 0.85
errorMinimizer:
  PointToPlaneErrorMinimizer
transformationCheckers:
  - CounterTransformationChecker:
      maxIterationCount: {iters}
inspector:
  NullInspector
logger:
  NullLogger
""",
}
YAML_CHAINS["docs_knn6"] = YAML_CHAINS["p2plane"].replace("knn: 1", "knn: 6")


def available():
    return os.path.exists(LIB)


def register(yaml_text, map4, map_normals3, scan4, scan_normals3=None):
    lib = C.CDLL(LIB)
    lib.orc_pm_register.restype = C.c_int
    m = np.ascontiguousarray(map4, dtype=np.float32); s = np.ascontiguousarray(scan4, dtype=np.float32)
    mn = np.ascontiguousarray(map_normals3, dtype=np.float32) if map_normals3 is not None else None
    sn = np.ascontiguousarray(scan_normals3, dtype=np.float32) if scan_normals3 is not None else None
    T = (C.c_float * 16)(); ov = C.c_float(0); err = C.create_string_buffer(512)
    rc = lib.orc_pm_register(yaml_text.encode(), C.c_void_p(m.ctypes.data), C.c_int64(m.shape[0]),
                             C.c_void_p(mn.ctypes.data if mn is not None else None), C.c_void_p(s.ctypes.data), C.c_int64(s.shape[0]),
                             C.c_void_p(sn.ctypes.data if sn is not None else None), T, C.byref(ov), err, 512)
    if rc:
        raise RuntimeError("libpointmatcher: " + err.value.decode(errors="replace"))
    return np.array(T[:], dtype=np.float32).reshape(4, 4).T.copy(), float(ov.value)


def register_default_chain(chain, map4, map_normals3, scan4, iterations):
    return register(YAML_CHAINS[chain].format(iters=iterations), map4, map_normals3, scan4)[0]
