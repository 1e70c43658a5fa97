"""ctypes binding of oracle/_ref/libref_nms_kernel*.so — the REFERENCE's own `_nms` / `nms_kernel`
(lib/nms/src/cuda/nms_kernel.cu:26-83), compiled unmodified by `make -C oracle ref`.

TEST INFRASTRUCTURE: only tests/ may import this.  `_nms` launches on the NULL stream with no error
checking (as in the reference, nms_cuda.c:32); callers synchronise around it.  The host scan below is the
restatement of nms_cuda.c:47-58 (that file needs <TH/TH.h> and cannot be compiled here).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = {"default": "libref_nms_kernel.so", "nocontract": "libref_nms_kernel_nocontract.so"}
_libs = {}


def path(variant="default"):
    return os.path.join(_HERE, "_ref", VARIANTS[variant])


def available(variant="default"):
    return os.path.exists(path(variant))


def lib(variant="default"):
    if variant not in _libs:
        L = ctypes.CDLL(path(variant))
        L._nms.restype = None
        L._nms.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float]
        _libs[variant] = L
    return _libs[variant]


def ref_mask(sorted_dets_dev, thresh, variant="default"):
    """sorted_dets_dev: contiguous f32 [n,5] DEVICE tensor, rows in descending score order (nms_cuda.c:18 precondition).
    Returns the reference kernel's full mask, uint64 [n, ceil(n/64)] (numpy)."""
    import torch
    n = int(sorted_dets_dev.shape[0])
    cb = (n + 63) // 64
    assert sorted_dets_dev.is_cuda and sorted_dets_dev.is_contiguous() and sorted_dets_dev.dtype == torch.float32
    mask = torch.zeros((n, cb), dtype=torch.int64, device=sorted_dets_dev.device)
    torch.cuda.synchronize()
    lib(variant)._nms(n, sorted_dets_dev.data_ptr(), mask.data_ptr(), ctypes.c_float(thresh))
    torch.cuda.synchronize()
    return mask.cpu().numpy().view(np.uint64)


def host_scan(mask, n):
    """nms_cuda.c:47-58: serial scan over the mask rows; returns kept positions (into the SORTED order)."""
    cb = (n + 63) // 64
    remv = np.zeros(cb, dtype=np.uint64)
    keep = []
    for i in range(n):
        nb, ib = i // 64, i % 64
        if not (int(remv[nb]) >> ib) & 1:
            keep.append(i)
            remv[nb:] |= mask[i, nb:]
    return np.asarray(keep, dtype=np.int64)
