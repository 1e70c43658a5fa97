"""Heat-map peak extraction on the GPU — the device half of the reference's ``network/joint_utils.py``.

Mirrors ``find_peaks`` (joint_utils.py:19-31), ``NMS`` (:61-138) and ``get_joint_list`` (:141-152) with the same
signatures and return structures, but the heat-maps stay on the device: the 3x3-cross maximum filter, the threshold, the
row-major compaction, the 5x5-patch bicubic refinement and the running peak ids are one kernel launch per batch
(``mpn_heatmap_peaks``) instead of scipy/cv2 calls per joint and per peak on the host.  Only the (few dozen) peaks cross
PCIe.  The drawing helpers of the reference file (cv2) are out of scope.
"""
import numpy as np
import torch

from .._lib import MpnError, call
from .. import ops

NUM_JOINTS = 18                      # joint_utils.py:16
DEFAULT_CAP = 256                    # first-try peaks per joint type (the reference has no limit: the launch is repeated with a
                                     # larger buffer when a plane holds more; an explicit `cap` argument is a hard limit instead)


def _peaks_device(heat_bjhw, thre1, upsamp, refine, cap=DEFAULT_CAP):
    """heat_bjhw: CUDA float32 tensor viewed as [B, J, H, W] (any strides).  Returns (peaks [B,J,cap,4] f64, counts [B,J])."""
    if not heat_bjhw.is_cuda:
        raise MpnError("heat-map peak extraction runs on the MI355X only; there is no CPU path")
    if heat_bjhw.dtype != torch.float32 or heat_bjhw.dim() != 4:
        raise MpnError("heat-maps must be a float32 [B, J, H, W] view")
    B, J, H, W = heat_bjhw.shape
    sB, sJ, sY, sX = heat_bjhw.stride()
    peaks = torch.empty((B, J, cap, 4), dtype=torch.float64, device=heat_bjhw.device)
    counts = torch.empty((B, J), dtype=torch.int32, device=heat_bjhw.device)
    ws = ops.workspace(call("mpn_heatmap_peaks_workspace_bytes", B, J, H, W, cap), heat_bjhw.device, slot=8)
    call("mpn_heatmap_peaks", ops.ptr(heat_bjhw), sB, sJ, sY, sX, B, J, H, W, float(thre1), float(upsamp), 1 if refine else 0,
         ops.ptr(peaks), ops.ptr(counts), cap, ops.ptr(ws), ops.stream_ptr())
    return peaks, counts


def _extract(heat_bjhw, thre1, upsamp, refine, cap):
    """Peaks of every plane as host arrays; a plane with more than `cap` peaks (noise-like maps) reruns the launch with a
    capacity that fits — the reference has no limit."""
    grow = cap is None                      # an explicit capacity is a hard limit (overflow raises)
    cap = DEFAULT_CAP if cap is None else cap
    pk, cnt = _peaks_device(heat_bjhw, thre1, upsamp, refine, cap)
    counts = cnt.cpu().numpy()              # the one host sync of the extraction
    most = int(counts.max(initial=0))
    if most > cap:
        if not grow:
            raise MpnError("more than %d peaks of one joint type in an image; raise `cap`" % cap)
        cap = 1 << (most - 1).bit_length()
        pk, cnt = _peaks_device(heat_bjhw, thre1, upsamp, refine, cap)
        counts = cnt.cpu().numpy()
    return _split(pk, counts, most)


def _split(peaks, counts, most):
    """Only the occupied prefix [:, :, :most] of the peak buffer crosses PCIe."""
    B, J = counts.shape
    if most == 0:
        return [[np.zeros((0, 4)) for _ in range(J)] for _ in range(B)]
    host = peaks[:, :, :most].cpu().numpy()
    return [[host[b, j, :counts[b, j]].copy() for j in range(J)] for b in range(B)]


def find_peaks(param, img):
    """joint_utils.py:19-31.  img: CUDA float32 [H, W].  Returns an int array [[x, y], ...] in row-major order."""
    out = _extract(img[None, None], param['thre1'], 1.0, False, None)[0][0]
    return out[:, :2].astype(np.int64)


def NMS(param, heatmaps, upsampFactor=1., bool_refine_center=True, bool_gaussian_filt=False, cap=None):
    """joint_utils.py:61-138.  heatmaps: CUDA float32 tensor [H, W, J] (any strides, e.g. ``pred[0].permute(1, 2, 0)``).
    Returns the reference's list of J arrays [n, 4] = (x, y, score, id)."""
    if bool_gaussian_filt:
        raise MpnError("bool_gaussian_filt=True (off by default in the reference, joint_utils.py:61) is not built")
    return _extract(heatmaps.permute(2, 0, 1)[None], param['thre1'], upsampFactor, bool_refine_center, cap)[0]


def NMS_batch(param, pred, upsampFactor=1., bool_refine_center=True, cap=None):
    """All images of a ``[B, J, H, W]`` heat-map tensor in one launch; a list (per image) of NMS() results."""
    return _extract(pred, param['thre1'], upsampFactor, bool_refine_center, cap)


def NMS_batch_arrays(param, pred, upsampFactor=1., bool_refine_center=True, cap=None):
    """NMS_batch without the per-image / per-joint Python lists: (peaks float64 [B, J, most, 4] = (x, y, score, id), counts int32 [B, J])
    on the host — entries [b, j, :counts[b, j]] are image b's peaks of joint type j in row-major order."""
    grow = cap is None
    cap = DEFAULT_CAP if cap is None else cap
    pk, cnt = _peaks_device(pred, param['thre1'], upsampFactor, bool_refine_center, cap)
    counts = cnt.cpu().numpy()
    most = int(counts.max(initial=0))
    if most > cap:
        if not grow:
            raise MpnError("more than %d peaks of one joint type in an image; raise `cap`" % cap)
        cap = 1 << (most - 1).bit_length()
        pk, cnt = _peaks_device(pred, param['thre1'], upsampFactor, bool_refine_center, cap)
        counts = cnt.cpu().numpy()
    return pk[:, :, :max(most, 1)].cpu().numpy(), counts


def body_peaks_flat(peaks, counts, scale=1.0, keep=None):
    """The 17 body joint types of tester.py:158-164 (the neck, type 1, dropped; later types shifted down) as the flat arrays
    evaluate.prn_process.prn_assign_arrays takes: (peaks_xy float64 [Np, 2] grouped by image then type, joint_off int32 [B, 18]).
    `keep`: at most that many peaks per type (in detection order).  Vectorised: no Python loop over peaks."""
    B, J, M, _ = peaks.shape
    types = [0] + list(range(2, J))                       # 17 body types in PRN order
    cnt = counts[:, types].astype(np.int64)
    if keep is not None:
        cnt = np.minimum(cnt, keep)
        peaks, M = peaks[:, :, :keep], min(M, keep)
    sel = np.arange(M)[None, None, :] < cnt[:, :, None]                       # [B, 17, M]
    xy = peaks[:, types][..., :2] * scale                                       # [B, 17, M, 2]
    flat = xy[sel]                                                              # row-major: image, type, detection order
    off = np.zeros((B, 18), dtype=np.int32)
    csum = np.cumsum(cnt.reshape(-1))
    off.reshape(-1)[0] = 0
    starts = np.concatenate([[0], csum])                                        # B*17 + 1 running offsets
    off[:, :17] = starts[:-1].reshape(B, 17)
    off[:, 17] = starts[17::17][:B]
    return np.ascontiguousarray(flat, dtype=np.float64), off


def get_joint_list(img_orig, param, heatmaps, scale):
    """joint_utils.py:141-152: rows (x*scale, y*scale, score, id, joint_type)."""
    per_type = NMS(param, heatmaps, img_orig.shape[0] / float(heatmaps.shape[0]))
    for peaks in per_type:
        peaks[:, :2] = peaks[:, :2] * scale
    return np.array([tuple(peak) + (joint_type,) for joint_type, joint_peaks in enumerate(per_type) for peak in joint_peaks])
