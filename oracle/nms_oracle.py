"""ctypes binding of oracle/nms_oracle.c (TEST INFRASTRUCTURE; see that file's header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_nms.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_nms.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        i64p = ctypes.POINTER(ctypes.c_int64)
        f32p = ctypes.POINTER(ctypes.c_float)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        _lib.oracle_nms_gpu.restype = ctypes.c_int64
        _lib.oracle_nms_gpu.argtypes = [f32p, ctypes.c_int64, ctypes.c_float, i64p, u64p]
        _lib.oracle_nms_cpu.restype = ctypes.c_int64
        _lib.oracle_nms_cpu.argtypes = [f32p, ctypes.c_int64, ctypes.c_float, i64p]
        _lib.oracle_sort_desc.restype = None
        _lib.oracle_sort_desc.argtypes = [f32p, ctypes.c_int64, i64p]
    return _lib


def nms(dets, thresh, mode="gpu", return_mask=False):
    """dets: [N,5] float32 (x1,y1,x2,y2,score).  Returns int64 keep indices (original order ids)."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    keep = np.zeros(max(n, 1), dtype=np.int64)
    L = lib()
    f32p = ctypes.POINTER(ctypes.c_float)
    i64p = ctypes.POINTER(ctypes.c_int64)
    if mode == "gpu":
        mask = None
        mp = None
        if return_mask:
            mask = np.zeros((n, (n + 63) // 64), dtype=np.uint64)
            mp = mask.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
        k = L.oracle_nms_gpu(dets.ctypes.data_as(f32p), n, ctypes.c_float(thresh),
                             keep.ctypes.data_as(i64p), mp)
        return (keep[:k].copy(), mask) if return_mask else keep[:k].copy()
    k = L.oracle_nms_cpu(dets.ctypes.data_as(f32p), n, ctypes.c_float(thresh), keep.ctypes.data_as(i64p))
    return keep[:k].copy()


def sort_desc(dets):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    order = np.zeros(dets.shape[0], dtype=np.int64)
    lib().oracle_sort_desc(dets.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), dets.shape[0],
                           order.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return order
