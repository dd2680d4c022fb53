"""ctypes wrapper of oracle/mp_oracle.c (TEST INFRASTRUCTURE: checker and cpu_baseline only, see the C file header)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("MP_ORACLE_LIB") or os.path.join(_HERE, "libmp_oracle.so")   # override: `make -C oracle asan`
_lib = None


def available():
    return os.path.exists(_LIB)


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_LIB)
        _lib.mpo_num_threads.restype = ctypes.c_int
        _lib.mpo_schnet_forward.restype = ctypes.c_int
    return _lib


def num_threads():
    return int(lib().mpo_num_threads())


def set_num_threads(n):
    lib().mpo_set_num_threads(ctypes.c_int(int(n)))


def schnet_forward(params, z, xyz, idx, node_splits, edge_splits, depth=3, gauss_args=None):
    """Same contract as ``kgcnn_oracle.schnet_forward`` (graph output ``(G, 1)``), weights in constructor order."""
    ga = gauss_args or {"bins": 20, "distance": 4, "offset": 0.0, "sigma": 0.4}
    ws = [np.ascontiguousarray(v, dtype=np.float32) for v in params.values()]
    arr = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    z = np.ascontiguousarray(z, np.float32)
    xyz = np.ascontiguousarray(xyz, np.float32)
    idx = np.ascontiguousarray(idx, np.int64)
    ns = np.ascontiguousarray(node_splits, np.int64)
    es = np.ascontiguousarray(edge_splits, np.int64)
    g = len(ns) - 1
    out = np.empty((g, 1), np.float32)
    rc = lib().mpo_schnet_forward(
        ctypes.c_void_p(z.ctypes.data), ctypes.c_void_p(xyz.ctypes.data), ctypes.c_void_p(idx.ctypes.data),
        ctypes.c_void_p(ns.ctypes.data), ctypes.c_void_p(es.ctypes.data), ctypes.c_int64(g), ctypes.c_int(depth),
        ctypes.c_int(int(ga["bins"])), ctypes.c_float(float(ga["distance"])), ctypes.c_float(float(ga["sigma"])),
        ctypes.c_float(float(ga["offset"])), arr, ctypes.c_int(int(ws[0].shape[0])), ctypes.c_void_p(out.ctypes.data))
    if rc != 0:
        raise MemoryError("mpo_schnet_forward failed")
    return out
