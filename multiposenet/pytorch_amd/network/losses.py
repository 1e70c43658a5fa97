"""Loss layer of the hot path — same names and return structure as the reference
(``network/losses.py`` FocalLoss / calc_iou, ``network/posenet.py:367-445`` build_*_loss), computed
by fused HIP kernels (csrc/losses.hip) instead of per-image python loops and ~60 tiny ATen kernels.
"""
import ctypes
import numbers
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import ops
from .._lib import call

_i64x5 = ctypes.c_int64 * 5
_i32x5 = ctypes.c_int32 * 5
_vpx5 = ctypes.c_void_p * 5


def _pixel_major(t):
    """logical [B,C,H,W] f32 -> [B,H,W,C] view with unit channel stride and row-major pixels."""
    g = t.permute(0, 2, 3, 1)
    if g.dtype != torch.float32:
        g = g.float()
    if g.stride(3) != 1 or g.stride(1) != g.shape[2] * g.stride(2) or g.stride(0) != g.shape[1] * g.stride(1):
        g = g.contiguous()
    return g


def mse_forward_raw(pm, heat_nhwc, wgt_nhwc):
    """pm: five pixel-major [B,H,W,C_j] f32 predictions.  Returns the device vector out[8] (five level means, total, max, min)."""
    B, H, W, _ = pm[0].shape
    npix = B * H * W
    dev = pm[0].device
    chunks = call("mpn_mse_chunks", npix)
    part = ops.workspace(chunks * 8 * 4, dev, slot=5)
    out = torch.empty(8, dtype=torch.float32, device=dev)
    call("mpn_mse_heatmap_forward", _vpx5(*[p.data_ptr() for p in pm]), _i64x5(*[p.stride(2) for p in pm]),
         ops.ptr(heat_nhwc), ops.ptr(wgt_nhwc), npix, ops.ptr(part), chunks, ops.ptr(out), ops.stream_ptr())
    return out


def mse_backward_raw(pm, heat, wgt, gs, need):
    """Gradients of the total w.r.t. the five predictions (logical [B,C,H,W] views of fresh NHWC tensors; None where
    need[j] is false); gs: device float[1] upstream gradient."""
    B, H, W, _ = pm[0].shape
    npix = B * H * W
    grads, gptr, gsp, gc = [], [], [], []
    for j, p in enumerate(pm):
        if need[j]:
            g = torch.empty((B, H, W, p.shape[3]), dtype=torch.float32, device=p.device)
            grads.append(g.permute(0, 3, 1, 2))
            gptr.append(g.data_ptr()); gsp.append(g.stride(2)); gc.append(p.shape[3])
        else:
            grads.append(None)
            gptr.append(None); gsp.append(0); gc.append(0)
    call("mpn_mse_heatmap_backward", _vpx5(*[p.data_ptr() for p in pm]), _vpx5(*gptr), _i64x5(*[p.stride(2) for p in pm]),
         _i64x5(*gsp), _i32x5(*gc), ops.ptr(heat), ops.ptr(wgt), npix, ops.ptr(gs), ops.stream_ptr())
    return grads


def mse_train_supported(levels, heat):
    """The one-pass form needs every intermediate level at an exact power-of-two fraction of a heat-map whose sides are multiples
    of 8 (true of the reference's 480x480 / 384x384 training crops), and dense NCHW targets."""
    B, C, H, W = heat.shape
    if C != 18 or H % 8 or W % 8 or not heat.is_contiguous():
        return False
    geo = [(H >> s, W >> s) for s in (0, 1, 2, 3)] + [(H, W)]
    return all(a.B == B and (a.H, a.W) == g and a.Cs == 32 and a.t.dtype == torch.float32 for a, g in zip(levels, geo))


def mse_train_raw(levels, heat, wgt, gs, dtype):
    """Recorded step: heat-map loss and its gradients straight from the network's internal tensors.  levels: the five f32
    activations (k2..k5 at 1, 1/2, 1/4, 1/8 of the heat-map size, then the prediction); heat / wgt: the reference's NCHW f32 targets,
    read in place.  Returns (out[8] as mse_forward_raw, five internal gradient activations of ``dtype``)."""
    B, _, H, W = heat.shape
    dev = heat.device
    blocks = call("mpn_mse_train_blocks", B, H, W)
    part = ops.workspace(blocks * 8 * 4, dev, slot=5)
    out = torch.empty(8, dtype=torch.float32, device=dev)
    grads = [ops.Act(torch.empty(a.t.shape, dtype=dtype, device=dev), a.C) for a in levels]
    call("mpn_mse_heatmap_train", _vpx5(*[a.t.data_ptr() for a in levels]), _vpx5(*[g.t.data_ptr() for g in grads]),
         _i32x5(*[a.Cs for a in levels]), ops.dtype_code(dtype), ops.ptr(heat), ops.ptr(wgt), heat.stride(0), heat.stride(1),
         heat.stride(2), B, H, W, ops.ptr(gs), ops.ptr(part), blocks, ops.ptr(out), ops.stream_ptr())
    return out, grads


class _HeatmapMSE(torch.autograd.Function):
    """sum_j mean(((pred_j[:, :18] * w) - (w * gt))^2)   (posenet.py:376-387)."""

    @staticmethod
    def forward(ctx, heat_nhwc, wgt_nhwc, *preds):
        pm = [_pixel_major(p.detach()) for p in preds]
        out = mse_forward_raw(pm, heat_nhwc, wgt_nhwc)
        ctx.pm = pm
        ctx.gt = (heat_nhwc, wgt_nhwc)
        ctx.mark_non_differentiable(out)
        return out[5].clone(), out

    @staticmethod
    def backward(ctx, gtotal, _gout):
        heat, wgt = ctx.gt
        gs = gtotal.detach().reshape(1).float().contiguous()
        grads = mse_backward_raw(ctx.pm, heat, wgt, gs, [ctx.needs_input_grad[2 + j] for j in range(len(ctx.pm))])
        return (None, None) + tuple(grads)


class _LazyVec(object):
    """Device -> host copy of a small log vector that does not stall the launch stream: the copy is enqueued behind the
    kernels that produce the values (pinned destination, non-blocking) and only waited for when somebody reads one."""

    def __init__(self, t):
        self._host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        self._host.copy_(t.detach(), non_blocking=True)
        self._ev = torch.cuda.Event()
        self._ev.record()
        self._vals = None

    def get(self):
        if self._vals is None:
            self._ev.synchronize()
            self._vals = self._host.tolist()
            self._host = None
        return self._vals


class LazyFloat(numbers.Real):
    """A log value that becomes a Python float the moment it is used (printed, formatted, added to a meter ...).

    The reference's build_*_loss return floats obtained with ``.item()`` (posenet.py:383-401, :417-423), i.e. a host
    sync between forward and backward in every step.  The trainer only looks at them after ``optimizer.step()``
    (trainer.py:251-262), so with ``set_lazy_log(True)`` (or ``MPN_LAZY_LOG=1``) the values are fetched asynchronously
    and materialised on first use.  It is a ``numbers.Real``, not a ``float`` subclass: code that tests
    ``isinstance(v, float)`` (the reference trainer does, trainer.py:39,324) must test ``numbers.Real`` instead, which is
    why this is opt-in."""
    __slots__ = ("_src", "_i")

    def __init__(self, src, i):
        self._src, self._i = src, i

    def __float__(self):
        return float(self._src.get()[self._i])

    item = __float__

    def __repr__(self):
        return repr(float(self))

    __str__ = __repr__

    def __format__(self, spec):
        return format(float(self), spec)

    def __hash__(self):
        return hash(float(self))

    def __bool__(self):
        return bool(float(self))

    def __int__(self):
        return int(float(self))

    def __array__(self, dtype=None, copy=None):
        import numpy as np
        return np.asarray(float(self), dtype=dtype)

    def __abs__(self): return abs(float(self))
    def __neg__(self): return -float(self)
    def __pos__(self): return float(self)
    def __trunc__(self): return float(self).__trunc__()
    def __floor__(self): return float(self).__floor__()
    def __ceil__(self): return float(self).__ceil__()
    def __round__(self, n=None): return round(float(self), n)
    def __add__(self, o): return float(self) + o
    def __radd__(self, o): return o + float(self)
    def __sub__(self, o): return float(self) - o
    def __rsub__(self, o): return o - float(self)
    def __mul__(self, o): return float(self) * o
    def __rmul__(self, o): return o * float(self)
    def __truediv__(self, o): return float(self) / o
    def __rtruediv__(self, o): return o / float(self)
    def __floordiv__(self, o): return float(self) // o
    def __rfloordiv__(self, o): return o // float(self)
    def __mod__(self, o): return float(self) % o
    def __rmod__(self, o): return o % float(self)
    def __pow__(self, o): return float(self) ** o
    def __rpow__(self, o): return o ** float(self)
    def __eq__(self, o): return float(self) == o
    def __ne__(self, o): return float(self) != o
    def __lt__(self, o): return float(self) < o
    def __le__(self, o): return float(self) <= o
    def __gt__(self, o): return float(self) > o
    def __ge__(self, o): return float(self) >= o


LAZY_LOG = os.environ.get("MPN_LAZY_LOG", "0") == "1"


def set_lazy_log(on=True):
    """Opt in to asynchronous log values (LazyFloat).  Off by default: build_*_loss then return plain Python floats
    exactly like the reference (whose trainer tests ``isinstance(v, (int, float))``, trainer.py:39,324) at the price of
    one host sync between forward and backward (about 5 % of the step at 700 images/s).  Returns the previous setting."""
    global LAZY_LOG
    was, LAZY_LOG = LAZY_LOG, bool(on)
    return was


class _Deferred(object):
    """Placeholder for log value `i` of captured vector `slot` (a capturing stepper — tools/archive/r2/hipgraph_step.py — turns it into a float / LazyFloat per replay)."""
    __slots__ = ("slot", "i")

    def __init__(self, slot, i):
        self.slot, self.i = slot, i


CAPTURE_LOG = None      # list of static device vectors while a hipGraph capture of a step is in progress (tools/archive/r2/hipgraph_step.py) (no D2H copy inside a capture)


def _log_values(t):
    """Python-visible values of a small device vector: floats (default) or LazyFloat proxies (set_lazy_log)."""
    if CAPTURE_LOG is not None:
        CAPTURE_LOG.append(t)
        return [_Deferred(len(CAPTURE_LOG) - 1, i) for i in range(t.numel())]
    if not LAZY_LOG or not t.is_cuda:
        return t.detach().cpu().tolist()
    src = _LazyVec(t)
    return [LazyFloat(src, i) for i in range(t.numel())]


def build_names():
    names = []
    for j in range(2, 6):
        names.append('heatmap_loss_k%d' % j)
        names.append('seg_loss_k%d' % j)
    names.append('heatmap_loss')
    names.append('seg_loss')
    return names


def _check_heatmap_targets(preds, heat_temp, heat_weight):
    """The kernels index the targets as pixel * 18 + channel with the pixel grid of the first prediction: anything the reference's
    torch expression would broadcast (a [B,1,H,W] mask) is expanded here, anything it would reject raises before the launch."""
    from .._lib import MpnError
    p0 = preds[0]
    B, _, H, W = p0.shape
    for j, p in enumerate(preds):
        if p.dim() != 4 or p.shape[0] != B or p.shape[2] != H or p.shape[3] != W or p.shape[1] < 18:
            raise MpnError("heat-map loss: prediction %d has shape %s, expected [%d, >=18, %d, %d]" % (j, tuple(p.shape), B, H, W))
        if p.device != p0.device:
            raise MpnError("heat-map loss: predictions live on different devices")
    out = []
    for name, t in (("heat_temp", heat_temp), ("heat_weight", heat_weight)):
        if t.dim() != 4 or t.shape[0] != B or t.shape[2] != H or t.shape[3] != W or t.shape[1] not in (1, 18):
            raise MpnError("heat-map loss: %s has shape %s, expected [%d, 18, %d, %d]" % (name, tuple(t.shape), B, H, W))
        if t.device != p0.device:
            raise MpnError("heat-map loss: %s is on %s, the predictions on %s" % (name, t.device, p0.device))
        out.append(t.expand(B, 18, H, W) if t.shape[1] == 1 else t)
    return out


def build_keypoint_loss(saved_for_loss, heat_temp, heat_weight):
    """posenet.py:367-403: returns (total_loss tensor with grad, OrderedDict of floats)."""
    names = build_names()
    heat_temp, heat_weight = _check_heatmap_targets(saved_for_loss[:5], heat_temp, heat_weight)
    heat = ops.nchw_to_nhwc_f32(heat_temp.detach().float())
    wgt = ops.nchw_to_nhwc_f32(heat_weight.detach().float())
    total, out = _HeatmapMSE.apply(heat, wgt, *saved_for_loss[:5])
    vals = _log_values(out)         # one asynchronous D2H copy instead of the reference's seven .item() syncs
    log = OrderedDict()
    for j in range(5):
        log[names[j * 2]] = vals[j]
    log['max_ht'] = vals[6]
    log['min_ht'] = vals[7]
    return total, log


def focal_forward_raw(cls, reg, anchors, anno):
    """Returns (out[2] = {classification loss, regression loss}, saved operands for focal_backward_raw)."""
    B, A = cls.shape[0], cls.shape[1]
    if cls.shape[2] != 1:
        raise NotImplementedError("the hot path is single-class (posenet.py:189 num_classes=1)")
    dev = cls.device
    c = cls.detach().float().contiguous()
    r = reg.detach().float().contiguous()
    an = anchors.detach().float().reshape(-1, 4).contiguous()
    ann = anno.detach().float().contiguous()
    maxn = ann.shape[1]
    blocks = call("mpn_focal_blocks", A)
    part = ops.workspace(B * blocks * 4 * 4, dev, slot=6)
    per_img = torch.empty((B, 4), dtype=torch.float32, device=dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    call("mpn_focal_forward", ops.ptr(c), ops.ptr(r), ops.ptr(an), ops.ptr(ann), B, A, maxn, ops.ptr(part), ops.ptr(per_img),
         ops.ptr(out), ops.stream_ptr())
    return out, (c, r, an, ann, per_img)


def focal_backward_raw(saved, gs):
    """gs: device float[2] = upstream gradients of {cls loss, reg loss}.  Returns (dcls [B,A,1], dreg [B,A,4])."""
    c, r, an, ann, per_img = saved
    B, A = c.shape[0], c.shape[1]
    dev = c.device
    dcls = torch.empty((B, A, 1), dtype=torch.float32, device=dev)
    dreg = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
    call("mpn_focal_backward", ops.ptr(c), ops.ptr(r), ops.ptr(an), ops.ptr(ann), B, A, ann.shape[1], ops.ptr(per_img),
         ops.ptr(gs), ops.ptr(dcls), ops.ptr(dreg), ops.stream_ptr())
    return dcls, dreg


class _Focal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls, reg, anchors, anno):
        out, ctx.saved = focal_forward_raw(cls, reg, anchors, anno)
        return out[0:1].clone(), out[1:2].clone()

    @staticmethod
    def backward(ctx, gc, gr):
        gs = torch.empty(2, dtype=torch.float32, device=gc.device)      # {d/d cls_loss, d/d reg_loss}
        gs[0:1].copy_(gc.detach().reshape(1))
        gs[1:2].copy_(gr.detach().reshape(1))
        dcls, dreg = focal_backward_raw(ctx.saved, gs)
        return dcls, dreg, None, None


class FocalLoss(nn.Module):
    """network/losses.py:24-137: forward(classifications, regressions, anchors, annotations)
    -> (classification_loss[1], regression_loss[1]) batch means."""

    def forward(self, classifications, regressions, anchors, annotations):
        return _Focal.apply(classifications, regressions, anchors, annotations)


def build_detection_loss(saved_for_loss, anno):
    """posenet.py:405-425."""
    log = OrderedDict()
    closs, rloss = FocalLoss()(*saved_for_loss, anno)
    closs = closs.mean()
    rloss = rloss.mean()
    total = closs + rloss
    vals = _log_values(torch.stack([total.detach(), closs.detach(), rloss.detach()]))
    log['total_loss'], log['classification_loss'], log['regression_loss'] = vals
    return total, log
