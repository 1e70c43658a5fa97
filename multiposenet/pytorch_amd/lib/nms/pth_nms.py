"""pth_nms — drop-in for the reference's ``lib/nms/pth_nms.py:5`` backed by the HIP kernels.

    pth_nms(dets: FloatTensor[N,5] (x1,y1,x2,y2,score), thresh: float) -> LongTensor[k]

Returns indices into the UNSORTED input, in descending-score order, exactly like the reference.
Comparison semantics follow the reference's two native paths: device tensors use the GPU kernel's
strict ``IoU > thresh`` (lib/nms/src/cuda/nms_kernel.cu:63), CPU tensors use ``cpu_nms``'s
``IoU >= thresh`` (lib/nms/src/nms.c:59).  Both run on the MI355X (CPU inputs are staged to the
device and the result is returned on the CPU); there is no host fallback.  Unlike the reference no
N x N/64 mask is copied to the host: sort, mask and the greedy scan all stay on the device and only
the kept-count crosses PCIe (one 8-byte read, needed to size the returned tensor).
"""
import torch

from ... import ops


def pth_nms(dets, thresh, mode=None):
    if dets.dim() != 2 or dets.size(1) != 5:
        raise ValueError("dets must be [N,5] (x1,y1,x2,y2,score)")
    if not torch.cuda.is_available():
        raise ops._lib.MpnError("pth_nms needs the MI355X HIP kernels (no CPU fallback)")
    was_cpu = not dets.is_cuda
    if mode is None:
        mode = "cpu" if was_cpu else "gpu"
    d = dets.detach()
    if d.dtype != torch.float32:
        d = d.float()
    d = d.cuda() if was_cpu else d
    keep = ops.nms(d, float(thresh), 0 if mode == "gpu" else 1)
    return keep.cpu() if was_cpu else keep


def nms(dets, thresh):
    """network/posenet.py:19-22 dispatcher."""
    return pth_nms(dets, thresh)
