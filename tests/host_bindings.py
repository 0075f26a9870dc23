"""ctypes access to the C++ host shell's test hook (norlab_icp_mapper_amd/host/TestHooks.cpp)."""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(_ROOT, "norlab_icp_mapper_amd", "libnorlab_icp_mapper_host.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
        _lib.nim_test_filter_chain.restype = C.c_int
    return _lib


def filter_chain(yaml_seq, cloud, handle=None, desc_name=None, desc=None):
    """Run a YAML sequence of DataPointsFilters of the host shell on `cloud` ((n, 4) float32).  handle: the icpmi handle of a
    norlab_icp_mapper_amd.ICPSequence (filters that need the GPU), or None.  Returns (cloud_out, normals or None, desc_out or None)."""
    lib = load()
    c = np.ascontiguousarray(cloud, dtype=np.float32); n = c.shape[0]
    out = np.empty_like(c); nrm = np.empty((n, 3), np.float32)
    span = 0 if desc is None else (1 if np.ndim(desc) == 1 else np.shape(desc)[1])
    d = None if desc is None else np.ascontiguousarray(desc, dtype=np.float32)
    dout = None if desc is None else np.empty_like(d)
    m = C.c_int64(0); hn = C.c_int(0); err = C.create_string_buffer(512)
    rc = lib.nim_test_filter_chain(C.c_void_p(handle), yaml_seq.encode(), C.c_void_p(c.ctypes.data), C.c_int64(n),
                                   None if desc_name is None else desc_name.encode(), C.c_int(span),
                                   C.c_void_p(None if d is None else d.ctypes.data), C.c_void_p(out.ctypes.data), C.c_void_p(nrm.ctypes.data),
                                   C.c_void_p(None if dout is None else dout.ctypes.data), C.byref(m), C.byref(hn), err, 512)
    if rc:
        raise RuntimeError(err.value.decode(errors="replace"))
    k = m.value
    return out[:k].copy(), (nrm[:k].copy() if hn.value else None), (None if dout is None else dout[:k].copy())
