"""Tensor-level wrappers over the C-ABI (``include/mpn.h``).

torch is used only as the device-memory allocator and stream provider: every function here passes
raw device pointers, sizes and the current ``hipStream_t`` to ``libmpn_hip.so``.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import BF16, F16, F32, ConvParams, WgradParams, call


def round_up(v, m):
    return (v + m - 1) // m * m


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16
    raise _lib.MpnError("unsupported dtype %s" % dt)


# Launch stream.  Kernels go to torch's current stream unless the engine has pushed an explicit one (the weight-gradient
# side stream): switching torch's current stream with `torch.cuda.stream(...)` costs ~20 us of Python per use, and
# torch.cuda.current_stream() ~8 us per launch; the raw-handle query below is two C calls.
_STREAM_OVERRIDE = []          # stack of (raw handle, torch.cuda.Stream)


def push_stream(stream):
    _STREAM_OVERRIDE.append((stream.cuda_stream, stream))


def pop_stream():
    _STREAM_OVERRIDE.pop()


def stream_handle():
    if _STREAM_OVERRIDE:
        return _STREAM_OVERRIDE[-1][0]
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def check_device(t):
    """Kernels are enqueued on the CURRENT device's stream: refuse tensors that live on another GPU (the reference selects
    the device with params.gpus[0] and never calls set_device; training/batch_processor.py here does)."""
    if not t.is_cuda:
        raise _lib.MpnError("tensor is on %s; the HIP path has no CPU fallback" % t.device)
    cur = torch._C._cuda_getDevice()
    if t.device.index != cur:
        raise _lib.MpnError("tensor is on cuda:%d but the current device is cuda:%d — call torch.cuda.set_device(%d) "
                            "(or run under torch.cuda.device) first" % (t.device.index, cur, t.device.index))


def stream_obj():
    """torch.cuda.Stream to record events on (None = torch's current stream)."""
    return _STREAM_OVERRIDE[-1][1] if _STREAM_OVERRIDE else None


def stream_ptr():
    return ctypes.c_void_p(stream_handle())


def is16(dt):
    """bf16 / f16: 32 elements per 64-byte K chunk, 8 per 16-byte vector."""
    return dt == torch.bfloat16 or dt == torch.float16


def dtype_name(dt):
    """Element-type spelling rocprofv3 prints for a kernel instantiation (tools/rocprof_summary.py)."""
    return "bf16" if dt == torch.bfloat16 else ("_Float16" if dt == torch.float16 else "float")


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class Act(object):
    """Dense pixel-major activation: storage ``t`` is [B, H, W, Cs] contiguous, ``C`` logical channels.

    Cs (channel storage) is a multiple of 32; lanes [C, Cs) hold zeros (see include/mpn.h).
    """
    __slots__ = ("t", "B", "H", "W", "C", "Cs", "needs_grad", "tag", "seg", "cons", "bn_src", "mask", "conv_cons", "relu_out", "other_cons")

    def __init__(self, t, C, needs_grad=False, tag=""):
        self.t = t
        self.B, self.H, self.W, self.Cs = t.shape
        self.C = C
        self.needs_grad = needs_grad
        self.tag = tag
        self.seg = None          # (flat buffer, index) when this activation is one level of a pyramid group (alloc_seg)
        self.cons = 0            # gradient contributions still to come in backward (engine: last-contributor detection)
        self.bn_src = None       # (y, BNState, relu, has_residual) when this is the output of a BatchNorm
        self.relu_out = False    # forward: this tensor is relu(conv(.)) (conv epilogue act 1)
        self.other_cons = 0      # consumers other than plain / pyramid convolutions (residual adds, relu, pooling)
        self.conv_cons = 0       # how many of the pending contributions are input gradients of plain convolutions (engine.conv)
        self.mask = None         # sign bits of this tensor (uint8 [P, Cs / V]) when bn_act produced them for the backward pass

    @staticmethod
    def empty(B, H, W, C, dtype, device, needs_grad=False, tag=""):
        return Act(torch.empty((B, H, W, round_up(C, 32)), dtype=dtype, device=device), C, needs_grad, tag)

    @property
    def P(self):
        return self.B * self.H * self.W

    @property
    def dtype(self):
        return self.t.dtype

    def nchw(self):
        """Logical [B, C, H, W] view (channels_last strides) — the shape the reference API exposes."""
        return self.t[..., : self.C].permute(0, 3, 1, 2)


class _KernelEvents(object):
    """Optional live timing of conv launches with HIP events on the launch stream (bench.py roofline).

    Each instrumented launch is bracketed by two events recorded on the SAME stream the kernel is
    enqueued on (torch's current stream); durations are read after the timed region."""

    def __init__(self):
        self.on = False
        self.detail = False          # name launches by shape (tools/shape_report.py)
        self.rec = []

    def enable(self):
        self.on = True
        self.rec = []

    def disable(self):
        self.on = False

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream_obj())
        return e

    def end(self, name, flops, e0):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(stream_obj())
        self.rec.append((name, flops, e0, e1))

    def summary(self):
        out = {}
        for name, flops, e0, e1 in self.rec:
            d = out.setdefault(name, {"ms": 0.0, "flops": 0.0, "n": 0})
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["n"] += 1
        return out


KERNEL_EVENTS = _KernelEvents()

_ws = {}
WS_EPOCH = 0          # replay.py sets a unique value while it records: the recorded launches then own their scratch buffers


def take_epoch_workspaces(epoch):
    """Remove and return the scratch buffers allocated under capture epoch `epoch` (the graph keeps them alive; nothing
    outside the graph may re-grow or free them)."""
    keys = [k for k in _ws if k[3] == epoch]
    return [_ws.pop(k) for k in keys]


def workspace(nbytes, device, slot=0):
    """Persistent scratch (grown geometrically); one buffer per (device, slot)."""
    # one buffer per stream as well: launches on different streams must not share (or re-grow) a scratch area
    key = (device, slot, stream_handle() if device.type == "cuda" else 0, WS_EPOCH)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        so = stream_obj()
        if buf is not None and so is not None:
            buf.record_stream(so)        # the outgoing buffer may still be read by launches queued on the pushed stream
        buf = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        if so is not None:
            buf.record_stream(so)
        _ws[key] = buf
    return buf


def alloc_seg(like, C, dtype):
    """One buffer holding an activation per pyramid level (geometry of the Acts in `like`, C channels): the levels are views
    of a single flat tensor, so element-wise kernels can treat the whole pyramid as one array."""
    Cs = round_up(C, 32)
    sizes = [a.B * a.H * a.W * Cs for a in like]
    flat = torch.empty(sum(sizes), dtype=dtype, device=like[0].t.device)
    out, off = [], 0
    for i, (a, n) in enumerate(zip(like, sizes)):
        v = Act(flat[off: off + n].view(a.B, a.H, a.W, Cs), C)
        v.seg = (flat, i)
        out.append(v)
        off += n
    return out


def seg_flat(acts):
    """The flat tensor behind a complete pyramid group (None if `acts` is not exactly one group in order)."""
    if not acts or acts[0].seg is None:
        return None
    flat = acts[0].seg[0]
    for i, a in enumerate(acts):
        if a.seg is None or a.seg[0] is not flat or a.seg[1] != i:
            return None
    return flat if sum(a.t.numel() for a in acts) == flat.numel() else None


def conv_forward_seg(xs, w, Cout, R, S, pad, bias=None, act=0, out_f32=False, mode=0, outs=None, accumulate=False, cin=None):
    """The same convolution (stride 1, same-size output) over every level of a pyramid in ONE launch — the shared RetinaNet
    towers of posenet.py:327-328.  xs: Acts with equal B / C / dtype.  Returns the per-level outputs (views of one buffer)."""
    x0 = xs[0]
    dt = x0.t.dtype
    odt = torch.float32 if out_f32 else dt
    if outs is None:
        outs = alloc_seg(xs, Cout, odt)
    p = ConvParams()
    p.w = w.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.B = x0.B
    p.Cin = cin if cin is not None else round_up(x0.C, 32 if is16(dt) else 16)
    p.Cout, p.Cout_store = Cout, outs[0].Cs
    p.x_sW, p.y_sP = x0.Cs, outs[0].Cs
    p.R, p.S, p.stride, p.pad = R, S, 1, pad
    p.mode, p.act, p.accumulate = mode, act, 1 if accumulate else 0
    p.dtype = dtype_code(dt)
    p.out_f32 = 1 if (out_f32 and dt != torch.float32) else 0
    p.nseg = len(xs)
    tile0 = 0
    for l, (x, o) in enumerate(zip(xs, outs)):
        assert (x.B, x.Cs, x.t.dtype) == (x0.B, x0.Cs, dt) and (o.H, o.W, o.Cs, o.t.dtype) == (x.H, x.W, outs[0].Cs, odt)
        p.seg_x[l], p.seg_y[l] = x.t.data_ptr(), o.t.data_ptr()
        p.seg_H[l], p.seg_W[l] = x.H, x.W
        p.seg_tile0[l] = tile0
        tile0 += (x.B * x.H * x.W + 127) // 128
    p.seg_tile0[len(xs)] = tile0
    if KERNEL_EVENTS.on:
        e0 = KERNEL_EVENTS.begin()
        call("mpn_conv_forward", ctypes.byref(p), stream_ptr())
        tc = call("mpn_conv_tile_rows", ctypes.byref(p))
        general = (bias is not None or accumulate or act != 0 or Cout % tc != 0)
        name = ("conv_igemm_s3_kernel" if call("mpn_conv_shared_tile", ctypes.byref(p)) == 1 else "conv_igemm_kernel") + "<%s, %d, 128, %s, %s>" % (dtype_name(dt), tc, "true" if p.out_f32 else "false", "true" if general else "false")
        if KERNEL_EVENTS.detail:
            name = "%s %dx%d %d->%d pyramid(%s)|0" % ("dgrad" if mode == 1 else "fwd", R, S, p.Cin, Cout, ",".join(str(x.H) for x in xs))
        KERNEL_EVENTS.end(name, sum(2.0 * x.B * x.H * x.W * Cout * R * S * min(p.Cin, x.C) for x in xs), e0)
    else:
        call("mpn_conv_forward", ctypes.byref(p), stream_ptr())
    return outs


def conv_wgrad_seg(xs, dys, dw, Cout, R, S, pad, db=None):
    """dw += sum over the pyramid levels of wgrad(x_l, dy_l) in ONE launch (LDS-DMA kernel).  Returns (handled, fused_db)."""
    x0 = xs[0]
    dev, dt = x0.t.device, x0.t.dtype
    p = WgradParams()
    p.dw = dw.data_ptr()
    p.x_sW, p.dy_sP = x0.Cs, dys[0].Cs
    p.B, p.Cin, p.Cout = x0.B, x0.C, Cout
    p.R, p.S, p.stride, p.pad = R, S, 1, pad
    p.dtype = dtype_code(dt)
    p.nseg = len(xs)
    for l, (x, d) in enumerate(zip(xs, dys)):
        p.seg_x[l], p.seg_dy[l] = x.t.data_ptr(), d.t.data_ptr()
        p.seg_H[l], p.seg_W[l] = x.H, x.W
    p.chunks = 1
    if not (call("mpn_conv_wgrad_kernel_id", ctypes.byref(p)) & 1):
        return False, False
    chunks = call("mpn_conv_wgrad_seg_plan", ctypes.byref(p))
    if chunks < 1:
        raise _lib.MpnError("mpn_conv_wgrad_seg_plan failed with status %d" % chunks)
    if chunks > 1:
        p.ws = workspace(chunks * Cout * R * S * x0.C * 4, dev, slot=1).data_ptr()
    fused_db = db is not None
    if fused_db:
        p.db = db.data_ptr()
        if chunks > 1:
            p.db_ws = workspace(chunks * Cout * 4, dev, slot=4).data_ptr()
    if KERNEL_EVENTS.on:
        kid = call("mpn_conv_wgrad_kernel_id", ctypes.byref(p))
        e0 = KERNEL_EVENTS.begin()
        call("mpn_conv_wgrad_partials" if chunks > 1 else "mpn_conv_wgrad", ctypes.byref(p), stream_ptr())
        name = "conv_wgrad_dma_seg%s_kernel<%d, %d>" % ("_f16" if dt == torch.float16 else "", kid >> 16, (kid >> 4) & 0xfff)
        if KERNEL_EVENTS.detail:
            name = "wgrad %dx%d %d->%d pyramid(%s) chunks=%d|0" % (R, S, x0.C, Cout, ",".join(str(x.H) for x in xs), chunks)
        KERNEL_EVENTS.end(name, sum(2.0 * x.B * x.H * x.W * Cout * R * S * x.C for x in xs), e0)
        if chunks > 1:
            call("mpn_reduce_partials", p.ws, chunks, Cout * R * S * x0.C, p.dw, 1, stream_ptr())
            if fused_db:
                call("mpn_reduce_partials", p.db_ws, chunks, Cout, p.db, 1, stream_ptr())
    else:
        call("mpn_conv_wgrad", ctypes.byref(p), stream_ptr())
    return True, fused_db


def _kseg_fill(p, srcs, H, W):
    """Describe the virtual concatenation cat_s(nearest_upsample(srcs[s])) (MpnConvParams / MpnWgradParams kseg_*)."""
    c = srcs[0].Cs
    p.kseg_n, p.kseg_c = len(srcs), c
    for i, a in enumerate(srcs):
        sh = (H // a.H).bit_length() - 1
        if a.Cs != c or a.C != c or (a.H << sh) != H or (a.W << sh) != W or a.B != srcs[0].B or a.t.dtype != srcs[0].t.dtype:
            raise _lib.MpnError("virtual concatenation: member %d does not up-sample to %dx%d by a power of two" % (i, H, W))
        p.kseg_shift[i] = sh
        p.kseg_x[i] = a.t.data_ptr()


def cat_supported(srcs, H, W):
    """True when conv_forward_cat / conv_wgrad_cat serve cat(up(srcs)) at H x W: 16-bit members of 128 channels, power-of-two ratios."""
    if not srcs or len(srcs) > 4 or not is16(srcs[0].t.dtype):
        return False
    for a in srcs:
        sh = (H // max(a.H, 1)).bit_length() - 1
        if a.C != 128 or a.Cs != 128 or (a.H << sh) != H or (a.W << sh) != W or a.t.dtype != srcs[0].t.dtype:
            return False
    return True


def conv_forward_cat(srcs, H, W, w, Cout, bias=None, act=0, tag="", res=None):
    """3x3 / stride 1 / pad 1 convolution over torch.cat([nearest_upsample(s) for s in srcs], 1) WITHOUT materialising it
    (posenet.py:311-315: the 512-channel input of conv2).  w: [Cout][3][3][len(srcs) * 128].  res: same-size tensor added in the
    epilogue (act 3 = ReLU after it): the expanded class maps of the members conv2cls_* serve."""
    x0 = srcs[0]
    dt, dev = x0.t.dtype, x0.t.device
    out = Act.empty(x0.B, H, W, Cout, dt, dev, False, tag)
    p = ConvParams()
    _kseg_fill(p, srcs, H, W)
    p.w, p.y = w.data_ptr(), out.t.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.B, p.H, p.W, p.Ho, p.Wo = x0.B, H, W, H, W
    p.Cin, p.Cout, p.Cout_store = p.kseg_n * p.kseg_c, Cout, out.Cs
    p.x_sW, p.x_sH, p.x_sB = p.Cin, W * p.Cin, H * W * p.Cin          # geometry of the virtual tensor (not dereferenced)
    p.y_sB, p.y_sP = H * W * out.Cs, out.Cs
    p.R, p.S, p.stride, p.pad = 3, 3, 1, 1
    p.act, p.dtype = act, dtype_code(dt)
    if res is not None:
        assert res.t.dtype == dt and (res.B, res.H, res.W, res.Cs) == (out.B, H, W, out.Cs)
        p.res, p.res_mode = res.t.data_ptr(), 1
        p.res_sB, p.res_sP = H * W * res.Cs, res.Cs
        p.res_H, p.res_W = H, W
    if KERNEL_EVENTS.on:
        e0 = KERNEL_EVENTS.begin()
        call("mpn_conv_forward", ctypes.byref(p), stream_ptr())
        tc = call("mpn_conv_tile_rows", ctypes.byref(p))
        name = "conv_igemm_s3_kernel<%s, %d, 128, false, true>" % (dtype_name(dt), tc)
        if KERNEL_EVENTS.detail:
            name = "fwd 3x3 %d->%d @%dx%d s1 virtual-cat%s%s|%d" % (p.Cin, Cout, H, W, " bias" if bias is not None else "", " act%d" % act if act else "",
                                                                    sum(a.t.numel() for a in srcs) * 2 + out.t.numel() * 2)
        KERNEL_EVENTS.end(name, 2.0 * x0.B * H * W * Cout * 9 * p.Cin, e0)
    else:
        call("mpn_conv_forward", ctypes.byref(p), stream_ptr())
    return out


def conv_wgrad_cat(srcs, H, W, dy, dw, Cout, db=None):
    """dw += wgrad of the convolution of conv_forward_cat (x = the virtual concatenation); db fused as in conv_wgrad."""
    x0 = srcs[0]
    dev, dt = x0.t.device, x0.t.dtype
    p = WgradParams()
    _kseg_fill(p, srcs, H, W)
    Cin = p.kseg_n * p.kseg_c
    p.dy, p.dw = dy.t.data_ptr(), dw.data_ptr()
    p.dy_sP = dy.Cs
    p.x_sW, p.x_sH, p.x_sB = Cin, W * Cin, H * W * Cin
    p.B, p.H, p.W, p.Cin = x0.B, H, W, Cin
    p.Ho, p.Wo, p.Cout = H, W, Cout
    p.R, p.S, p.stride, p.pad = 3, 3, 1, 1
    p.dtype = dtype_code(dt)
    p.chunks = 1
    chunks = call("mpn_conv_wgrad_chunks", ctypes.byref(p))
    p.chunks = chunks
    if chunks > 1:
        p.ws = workspace(chunks * Cout * 9 * Cin * 4, dev, slot=1).data_ptr()
    fused_db = db is not None
    if fused_db:
        p.db = db.data_ptr()
        if chunks > 1:
            p.db_ws = workspace(chunks * Cout * 4, dev, slot=4).data_ptr()
    if KERNEL_EVENTS.on:
        kid = call("mpn_conv_wgrad_kernel_id", ctypes.byref(p))
        e0 = KERNEL_EVENTS.begin()
        call("mpn_conv_wgrad_partials" if chunks > 1 else "mpn_conv_wgrad", ctypes.byref(p), stream_ptr())
        name = "conv_wgrad_dma%s%s_kernel<%d, %d>" % ("_lin" if (kid & 2) else "", "_f16" if dt == torch.float16 else "", kid >> 16, (kid >> 4) & 0xfff)
        if KERNEL_EVENTS.detail:
            name = "wgrad 3x3 %d->%d @%dx%d s1 virtual-cat chunks=%d|%d" % (Cin, Cout, H, W, chunks, sum(a.t.numel() for a in srcs) * 2 + dy.t.numel() * 2)
        KERNEL_EVENTS.end(name, 2.0 * x0.B * H * W * Cout * 9 * Cin, e0)
        if chunks > 1:
            call("mpn_reduce_partials", p.ws, chunks, Cout * 9 * Cin, p.dw, 1, stream_ptr())
            if fused_db:
                call("mpn_reduce_partials", p.db_ws, chunks, Cout, p.db, 1, stream_ptr())
    else:
        call("mpn_conv_wgrad", ctypes.byref(p), stream_ptr())
    return fused_db


# ---- conv2 of the keypoint head by position classes (csrc/conv2cls.hip) -------------------------------------------------------------
def conv2cls_supported(srcs, H, W):
    """The class formulation serves exactly posenet.py:311-315's geometry: four dense 128-channel members of one element type at 1/8,
    1/4, 1/2, 1/1 of an H x W (multiples of 8) output."""
    if len(srcs) != 4 or H % 8 or W % 8:
        return False
    if any(a.C != 128 or a.Cs != 128 or a.t.dtype != srcs[0].t.dtype or a.B != srcs[0].B for a in srcs):
        return False
    return [(a.H * f, a.W * f) for a, f in zip(srcs, (8, 4, 2, 1))] == [(H, W)] * 4


class Conv2ClsOperands(object):
    """Per-forward operands of the class formulation, all derived from the f32 master filter [O][3][3][4 C] in four launches: the
    combined filters (f32), their compute-dtype copies, and the transposed (input-gradient) copies when a backward pass will follow."""

    def __init__(self, w_master, O, C, dtype, want_t):
        dev = w_master.device
        n = call("mpn_conv2cls_comb_elems", O, C)
        self.O, self.C = O, C
        self.nm, self.nc, self.nt = O * 9 * 2 * C, 9 * O * 9 * C, 9 * O * C
        comb32 = torch.empty(n, dtype=torch.float32, device=dev)
        call("mpn_conv2cls_combine", ptr(w_master), ptr(comb32), O, C, stream_ptr())
        nf = self.nm + 2 * self.nc                                              # main filter + the two frame-filter sets
        if is16(dtype):
            self.comb = torch.empty(n, dtype=dtype, device=dev)
            cast_lowp(comb32, self.comb)
        else:
            self.comb = comb32                                                  # f32 kernels read the f32 filters in place
        self.wm = self.comb[: self.nm]                                         # [O][3][3][2C]
        self.wc = [self.comb[self.nm + m * self.nc: self.nm + (m + 1) * self.nc] for m in range(2)]      # [9 O][3][3][C], members q5 / q4
        self.wtap = [self.comb[nf + m * self.nt: nf + (m + 1) * self.nt] for m in range(2)]              # [9 O][C]: nine 1x1 filters
        self.wm_t = self.wtap_t = None
        if want_t:
            self.wm_t = torch.empty((2 * C, 3, 3, O), dtype=dtype, device=dev)
            weight_transpose(comb32[: self.nm], self.wm_t, O, 9, 2 * C, O)
            self.wtap_t = []                                                    # [C][1][1][9 O]: the member as nine 1x1 convolutions, transposed
            for m in range(2):
                t = torch.empty((C, 1, 1, 9 * O), dtype=dtype, device=dev)
                weight_transpose(comb32[nf + m * self.nt: nf + (m + 1) * self.nt], t, 9 * O, 1, C, 9 * O)
                self.wtap_t.append(t)
        self.keep = comb32


def conv2cls_expand(m8, m4, B, H, W, O, dtype):
    e = Act(torch.empty((B, H, W, O), dtype=dtype, device=m8.t.device), O)
    call("mpn_conv2cls_expand", ptr(m8.t), ptr(m4.t), ptr(e.t), B, H, W, O, dtype_code(dtype), stream_ptr())
    return e


def conv2cls_pool(dy):
    """(P8 [B, H/8, W/8, 9 O], P4 [B, H/4, W/4, 9 O]): per-class sums of dy over 8 x 8 / 4 x 4 blocks (one pass over dy)."""
    B, H, W, O = dy.B, dy.H, dy.W, dy.Cs
    p8 = Act(torch.empty((B, H // 8, W // 8, 9 * O), dtype=dy.t.dtype, device=dy.t.device), 9 * O)
    p4 = Act(torch.empty((B, H // 4, W // 4, 9 * O), dtype=dy.t.dtype, device=dy.t.device), 9 * O)
    call("mpn_conv2cls_pool", ptr(dy.t), ptr(p8.t), ptr(p4.t), B, H, W, O, dtype_code(dy.t.dtype), stream_ptr())
    return p8, p4


def conv2cls_classsum(t):
    """Per-tap products (f32 [B, h, w, 9 O]) -> class maps of the same shape (csrc/conv2cls.hip: conv2cls_classsum_kernel)."""
    m = Act(torch.empty_like(t.t), t.C)
    call("mpn_conv2cls_classsum", ptr(t.t), ptr(m.t), t.B, t.H, t.W, t.Cs // 9, stream_ptr())
    return m


def conv2cls_tapsum(pc):
    """Class sums [B, h, w, 9 O] -> per-tap sums of the same shape (csrc/conv2cls.hip: conv2cls_tapsum_kernel)."""
    g = Act(torch.empty_like(pc.t), pc.C)
    call("mpn_conv2cls_tapsum", ptr(pc.t), ptr(g.t), pc.B, pc.H, pc.W, pc.Cs // 9, dtype_code(pc.t.dtype), stream_ptr())
    return g


def conv_out_hw(H, W, R, S, stride, pad):
    return (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1


def conv_forward(x, w, Cout, R, S, stride, pad, bias=None, scale=None, act=0, res=None, res_mode=0,
                 want_stats=False, out_f32=False, out=None, accumulate=False, mode=0, out_hw=None,
                 cin=None, x_geom=None, y_geom=None, cout_store=None, needs_grad=False, tag="", bnb=None, bn_fin=None, res_mask=None,
                 _cls=None):
    """Implicit-GEMM convolution.  x: Act.  w: compute-dtype tensor laid out [Cout][R][S][Cin].

    mode 0 = forward gather, mode 1 = dgrad gather (then ``out_hw`` is the input-gradient size and
    ``w`` is the transposed copy).  x_geom=(H, W, sB, sH, sW) overrides the source geometry (stem).
    y_geom=(ptr_tensor, sB, sP) writes into a slice of a larger buffer.  Returns (Act|None, stats).
    """
    dev = x.t.device
    dt = x.t.dtype
    if (DGRAD_S2_CLASSES and _cls is None and mode == 1 and stride == 2 and not out_f32
            and res is None and not want_stats and bias is None and scale is None and act == 0 and x_geom is None and y_geom is None
            and out_hw is not None and cout_store is None):
        if R == 3 and S == 3 and pad == 1:
            return _conv_dgrad_s2_classes(x, w, Cout, out_hw, cin, out, accumulate, bnb, needs_grad, tag)
        if R == 1 and S == 1 and pad == 0 and accumulate and bnb is None and out is not None:
            # 1x1 / stride 2 (the ResNet down-sampling shortcut): only the pixels (2 i, 2 j) of dx receive anything — one class, one tap.  Needs an
            # existing dx to add to (the other three quarters keep what they hold) and no epilogue statistics (they would need every pixel).
            conv_forward(x, w, Cout, 1, 1, 1, 0, out=out, accumulate=True, mode=0, out_hw=((out_hw[0] + 1) // 2, (out_hw[1] + 1) // 2), cin=cin,
                         _cls=(0, 0, None, 0, 1))
            return out, None
    Cin = cin if cin is not None else round_up(x.C, 32 if is16(dt) else 16)
    if x_geom is None:
        H, W = x.H, x.W
        sB, sH, sW = x.H * x.W * x.Cs, x.W * x.Cs, x.Cs
    else:
        H, W, sB, sH, sW = x_geom
    if out_hw is None:
        Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    else:
        Ho, Wo = out_hw
    odt = torch.float32 if out_f32 else dt
    p = ConvParams()
    if y_geom is None:
        if out is None:
            out = Act.empty(x.B, Ho, Wo, Cout, odt, dev, needs_grad, tag)
        if _cls is None:
            assert out.t.dtype == odt and (out.B, out.H, out.W) == (x.B, Ho, Wo), "conv output geometry mismatch"
            p.y_sB, p.y_sP = Ho * Wo * out.Cs, out.Cs
        else:       # parity class (a, c) of a stride-2 input gradient (mpn.h: y_step): output pixel (i, j) -> (2 i + a, 2 j + c) of `out`
            ca, cc_, p.y_H, p.y_W = _cls[0], _cls[1], out.H, out.W
            assert out.t.dtype == odt and out.B == x.B and out.Cs == round_up(Cout, 32) and cout_store is None, "class launch: output tensor mismatch"
            assert 2 * (Ho - 1) + ca < out.H and 2 * (Wo - 1) + cc_ < out.W, "class launch: class grid exceeds the output"
            p.y_step, p.y_oh, p.y_ow = 2, ca, cc_
            if len(_cls) > 4 and _cls[4] == 1:          # 1x1 filter: its only tap
                p.w_taps, p.wtap0, p.wtap_dr, p.wtap_ds = 1, 0, 0, 0
            else:                                       # 3x3, pad 1: tap (t_r, t_s) of the class is filter tap (a + 1 - 2 t_r, c + 1 - 2 t_s)
                p.w_taps, p.wtap0, p.wtap_dr, p.wtap_ds = 9, (ca + 1) * 3 + (cc_ + 1), -6, -2
            p.y_sB, p.y_sP = out.H * out.W * out.Cs, out.Cs
        p.y = out.t.data_ptr()
        p.Cout_store = out.Cs if cout_store is None else cout_store
    else:
        yt, ysB, ysP = y_geom
        p.y = yt.data_ptr()
        p.y_sB, p.y_sP = ysB, ysP
        p.Cout_store = cout_store if cout_store is not None else round_up(Cout, 4)
    p.x = x.t.data_ptr()
    p.w = w.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.scale = scale.data_ptr() if scale is not None else None
    p.x_sB, p.x_sH, p.x_sW = sB, sH, sW
    p.B, p.H, p.W, p.Cin = x.B, H, W, Cin
    p.Ho, p.Wo, p.Cout = Ho, Wo, Cout
    p.R, p.S, p.stride, p.pad = R, S, stride, pad
    p.mode, p.act, p.accumulate = mode, act, 1 if accumulate else 0
    p.dtype = dtype_code(dt)
    p.out_f32 = 1 if (out_f32 and dt != torch.float32) else 0
    if res is not None:
        assert res.t.dtype == odt
        p.res = res.t.data_ptr()
        p.res_mode = res_mode
        p.res_sB, p.res_sP = res.H * res.W * res.Cs, res.Cs
        p.res_H, p.res_W = res.H, res.W
        if res_mask is not None:          # residual * mask bits (bn_act(want_mask=True) layout): res_mode 1 only
            assert res_mode == 1
            p.res_mask = res_mask.data_ptr()
    stats = None
    keep = None
    if want_stats:
        tiles = call("mpn_conv_stats_tiles", ctypes.byref(p))
        stats = torch.empty((tiles, Cout, 2), dtype=torch.float32, device=dev)
        p.stats = stats.data_ptr()
        if bn_fin is not None and fin_in_launch(tiles, Cout):
            # the last workgroup of every channel tile turns the tile partials into the BatchNorm coefficients (no finalize launch):
            # the BNState comes back in the stats slot
            gamma, beta, rm, rv, momentum, eps = bn_fin
            keep, stats = stats, BNState(Cout, dev)
            fin_attach(p, tiles, Cout, dev)
            p.fin_gamma, p.fin_beta = gamma.data_ptr(), beta.data_ptr()
            p.fin_rm = rm.data_ptr() if rm is not None else None
            p.fin_rv = rv.data_ptr() if rv is not None else None
            p.fin_out = stats.mean.data_ptr()
            p.fin_count, p.fin_momentum, p.fin_eps = float(x.B * Ho * Wo), momentum, eps
    if bnb is not None:
        # this launch completes dz of a BatchNorm: its backward statistics ride in the epilogue (returned in the stats slot)
        by, bz, st, relu = bnb[:4]
        assert y_geom is None and not out_f32 and by.t.shape == out.t.shape and by.t.dtype == out.t.dtype
        tiles = call("mpn_conv_stats_tiles", ctypes.byref(p))
        if _cls is None:
            stats = torch.empty((tiles, Cout, 2), dtype=torch.float32, device=dev)
            p.bnb_partial = stats.data_ptr()
        else:       # rows [tile0, tile0 + tiles) of the table the four class launches share
            stats, tile0 = _cls[2], _cls[3]
            assert tile0 + tiles <= stats.shape[0]
            p.bnb_partial = stats.data_ptr() + tile0 * Cout * 2 * 4
        p.bnb_y = by.t.data_ptr()
        p.bnb_z = bz.t.data_ptr() if bz is not None else None
        if bz is not None and bz.mask is not None:          # the ReLU mask as bits (bn_act(want_mask=True)): z itself is not read
            p.bnb_mask, p.bnb_z = bz.mask.data_ptr(), None
        p.bnb_mean, p.bnb_invstd = st.mean.data_ptr(), st.invstd.data_ptr()
        p.bnb_scale, p.bnb_shift = st.scale.data_ptr(), st.shift.data_ptr()
        p.bnb_relu = 1 if relu else 0
        if _cls is None and len(bnb) > 4 and bnb[4] is not None and fin_in_launch(tiles, Cout):
            # ... and the last workgroup of every channel tile finishes the reduction: dgamma / dbeta and the (k1, k2, k3) of
            # dy = k1*g + k2*y + k3 (mpn_bn_bwd_finalize's work); the coefficient tensor comes back in the stats slot
            gamma, train, dgamma, dbeta = bnb[4]
            keep, stats = stats, (torch.empty((3, Cout), dtype=torch.float32, device=dev) if train else None)
            fin_attach(p, tiles, Cout, dev)
            p.fin_gamma = gamma.data_ptr()
            p.fin_out = stats.data_ptr() if stats is not None else None
            p.fin_dgamma = dgamma.data_ptr() if dgamma is not None else None
            p.fin_dbeta = dbeta.data_ptr() if dbeta is not None else None
            p.fin_count, p.fin_train = float(x.B * Ho * Wo), 1 if train else 0
            if stats is None:
                stats = "frozen"
    if KERNEL_EVENTS.on:
        e0 = KERNEL_EVENTS.begin()
        call("mpn_conv_forward", ctypes.byref(p), stream_ptr())
        tc = call("mpn_conv_tile_rows", ctypes.byref(p))
        # algorithmic FLOPs: 2 * output pixels * Cout * taps * Cin; a stride-s dgrad only has 1/s^2 live taps
        live = (R * S) / float(stride * stride) if mode == 1 else R * S
        flops = 2.0 * x.B * Ho * Wo * Cout * live * min(Cin, x.C if x_geom is None else Cin)
        # the name rocprofv3 prints for this instantiation (tools/rocprof_summary.py spelling)
        general = (scale is not None or bias is not None or res is not None or accumulate or act != 0 or Cout % tc != 0 or bnb is not None)
        kind = call("mpn_conv_shared_tile", ctypes.byref(p))
        name = ("conv_igemm_s3_kernel" if kind == 1 else "conv_igemm_kernel") + "<%s, %d, 128, %s, %s>" % (dtype_name(dt), tc,
                                                          "true" if p.out_f32 else "false", "true" if general else "false")
        if KERNEL_EVENTS.detail:
            es = 2 if is16(dt) else 4
            byts = x.B * H * W * Cin * es + x.B * Ho * Wo * p.Cout_store * (4 if out_f32 else es) * (2 if accumulate else 1) \
                + (x.B * Ho * Wo * p.Cout_store * es if res is not None and res_mode == 1 else 0)
            name = "%s %dx%d %d->%d @%dx%d s%d%s%s%s%s%s|%d" % (
                ("dgrad s2 class(%d,%d)" % (_cls[0], _cls[1])) if _cls is not None else ("dgrad" if mode == 1 else "fwd"), R, S, Cin, Cout, Ho, Wo, stride,
                " stats" if want_stats else "",
                " bias" if bias is not None else "", " act%d" % act if act else "", " res%d" % res_mode if res is not None else "",
                " acc" if accumulate else "", byts)
        KERNEL_EVENTS.end(name, flops, e0)
    else:
        call("mpn_conv_forward", ctypes.byref(p), stream_ptr())
    return out, stats


DGRAD_S2_CLASSES = os.environ.get("MPN_DGRAD_S2_CLASSES", "1") != "0"


def _conv_dgrad_s2_classes(dy, wt, Cout, out_hw, cin, out, accumulate, bnb, needs_grad, tag):
    """Input gradient of a 3x3 / stride 2 / pad 1 convolution as four parity-class launches (mpn.h: y_step): the pixel (h, w) of dx only
    receives the taps with r = (h + 1) mod 2 (+ 2), s likewise — 1, 2, 2 or 4 of the nine — so each class is a small stride-1 gather over
    dy written to every second row / column of dx; together 2.25 taps per pixel instead of the 9 (6.75 of them against zeros) of the
    strided gather.  The BatchNorm-backward statistics of the epilogue (bnb) fill consecutive row ranges of ONE partial table."""
    H, W = out_hw
    dev, dt = dy.t.device, dy.t.dtype
    if out is None:
        out = Act.empty(dy.B, H, W, Cout, dt, dev, needs_grad, tag)
    plan = dgrad_s2_class_plan(dy.B, H, W)
    rows = sum(t for _, _, _, _, t, _ in plan)
    part = torch.empty((rows, Cout, 2), dtype=torch.float32, device=dev) if bnb is not None else None
    for a, c, ho, wo, t, tile0 in plan:
        if t > 0:
            conv_forward(dy, wt, Cout, 1 + a, 1 + c, 1, 0, out=out, accumulate=accumulate, mode=0, out_hw=(ho, wo), cin=cin, bnb=bnb,
                         _cls=(a, c, part, tile0))
    return out, part


def dgrad_s2_class_plan(B, H, W, tile_pixels=128):
    """The four parity classes of a 3x3 / stride 2 / pad 1 input gradient over a [B][H][W] dx: (a, c, class rows, class columns, pixel tiles,
    first row of the class in the shared statistics table).  Class (a, c) owns the pixels (2 i + a, 2 j + c); its launch has (1 + a) x (1 + c)
    taps, tap (t_r, t_s) being filter tap (a + 1 - 2 t_r, c + 1 - 2 t_s) applied to dy[i + t_r][j + t_s]."""
    plan, tile0 = [], 0
    for a in (0, 1):
        for c in (0, 1):
            ho, wo = (H - a + 1) // 2, (W - c + 1) // 2
            t = (B * ho * wo + tile_pixels - 1) // tile_pixels
            plan.append((a, c, ho, wo, t, tile0))
            tile0 += t
    return plan


def conv_wgrad(x, dy, dw, Cout, R, S, stride, pad, cin=None, x_geom=None, db=None):
    """dw (f32 view laid out [Cout][R][S][Cin], contiguous) += wgrad(x, dy).  db (f32 [Cout], optional) += column sums of
    dy, fused into the same kernel where the LDS-DMA path serves the launch; returns True if db was taken care of."""
    dev = x.t.device
    dt = x.t.dtype
    p = WgradParams()
    if x_geom is None:
        H, W = x.H, x.W
        sB, sH, sW = x.H * x.W * x.Cs, x.W * x.Cs, x.Cs
        Cin = cin if cin is not None else x.C
    else:
        H, W, sB, sH, sW = x_geom
        Cin = cin
    p.x, p.dy, p.dw = x.t.data_ptr(), dy.t.data_ptr(), dw.data_ptr()
    p.x_sB, p.x_sH, p.x_sW = sB, sH, sW
    p.dy_sP = dy.Cs
    p.B, p.H, p.W, p.Cin = x.B, H, W, Cin
    p.Ho, p.Wo, p.Cout = dy.H, dy.W, Cout
    p.R, p.S, p.stride, p.pad = R, S, stride, pad
    p.dtype = dtype_code(dt)
    p.chunks = 1
    chunks = call("mpn_conv_wgrad_chunks", ctypes.byref(p))
    p.chunks = chunks
    if chunks > 1:
        ws = workspace(chunks * Cout * R * S * Cin * 4, dev, slot=1)
        p.ws = ws.data_ptr()
    fused_db = False
    if db is not None and (call("mpn_conv_wgrad_kernel_id", ctypes.byref(p)) & 1):
        fused_db = True
        p.db = db.data_ptr()
        if chunks > 1:
            p.db_ws = workspace(chunks * Cout * 4, dev, slot=4).data_ptr()
    if KERNEL_EVENTS.on:
        # bracket the MFMA kernel alone (the partial-sum reduction is launched separately) so the class time
        # matches the kernel's own row in a rocprofv3 trace
        kid = call("mpn_conv_wgrad_kernel_id", ctypes.byref(p))
        e0 = KERNEL_EVENTS.begin()
        call("mpn_conv_wgrad_partials" if chunks > 1 else "mpn_conv_wgrad", ctypes.byref(p), stream_ptr())
        dts = dtype_name(dt)
        name = ("conv_wgrad_dma%s%s_kernel<%d, %d>" % ("_lin" if (kid & 2) else "", {torch.float16: "_f16", torch.float32: "_f32"}.get(dt, ""), kid >> 16, (kid >> 4) & 0xfff) if (kid & 1)
                else "conv_wgrad_kernel<%s, %d, %d>" % (dts, kid >> 16, (kid >> 4) & 0xfff))
        if KERNEL_EVENTS.detail:
            es = 2 if is16(dt) else 4
            name = "wgrad %dx%d %d->%d @%dx%d s%d chunks=%d|%d" % (R, S, Cin, Cout, dy.H, dy.W, stride, chunks,
                                                                 (x.B * H * W * Cin + x.B * dy.H * dy.W * dy.Cs) * es)
        KERNEL_EVENTS.end(name, 2.0 * x.B * dy.H * dy.W * Cout * R * S * Cin, e0)
        if chunks > 1:
            call("mpn_reduce_partials", p.ws, chunks, Cout * R * S * Cin, p.dw, 1, stream_ptr())
            if fused_db:
                call("mpn_reduce_partials", p.db_ws, chunks, Cout, p.db, 1, stream_ptr())
    else:
        call("mpn_conv_wgrad", ctypes.byref(p), stream_ptr())
    return fused_db


def bias_grad(dy, db, C):
    """db[C] (f32) += column sums of dy."""
    v = 4 if dy.t.dtype == torch.float32 else 8
    g = dy.Cs // v
    if dy.P <= 256 or dy.Cs % v != 0 or (g & (g - 1)) != 0:
        call("mpn_colsum_rows", ptr(dy.t), dtype_code(dy.t.dtype), dy.P, C, dy.Cs, ptr(db), stream_ptr())
        return
    chunks = call("mpn_channel_sum_chunks", dy.P, dy.Cs, dtype_code(dy.t.dtype))
    ws = workspace(chunks * C * 4, dy.t.device, slot=2)
    call("mpn_channel_sum", ptr(dy.t), dtype_code(dy.t.dtype), dy.P, C, dy.Cs, ptr(ws), chunks, stream_ptr())
    call("mpn_reduce_partials", ptr(ws), chunks, C, ptr(db), 1, stream_ptr())


def weight_transpose(w_master, wt, Cout, RS, Cin, Cout_pad):
    call("mpn_weight_transpose", ptr(w_master), ptr(wt), Cout, RS, Cin, Cout_pad, dtype_code(wt.dtype), stream_ptr())


def cast_lowp(src, dst):
    """f32 -> bf16 / f16 copy (dst's dtype decides)."""
    call("mpn_cast_f32", ptr(src), ptr(dst), src.numel(), dtype_code(dst.dtype), stream_ptr())


cast_bf16 = cast_lowp


_fin_counters = {}
# In-launch BatchNorm finalize (mpn.h: fin_*).  Up to FIN_MAX_TILES pixel tiles one workgroup per channel tile does the whole
# reduction; beyond that a single reader behind an acquire is slower than the separate (wide) finalize launch (round 2: +0.85 ms at
# <= 256 tiles), and so were the two-level form and the finalize inside the bn_act launch (round 3: +0.26 / +0.2 ms; both removed).
FIN_MAX_TILES = int(os.environ.get("MPN_BN_FIN_MAX_TILES", "64"))
FIN_COUNTERS = 64


def fin_in_launch(tiles, Cout=2):
    """The in-launch finalize reads the partial table as float4 pairs of channels: even widths only (every BatchNorm of the network is;
    an odd width takes the separate finalize launch instead of failing with BADARG — ADVICE r5)."""
    return tiles <= FIN_MAX_TILES and Cout % 2 == 0


def fin_attach(p, tiles, Cout, device):
    p.fin_counters = fin_counters(device).data_ptr()


def fin_counters(device):
    """Ticket counters of the in-launch BatchNorm finalize (mpn.h: fin_counters): zero between launches.  One buffer per
    (device, launch stream): launches of one stream run one after another, launches of different streams (a second model or a
    Tester beside training) must not draw tickets from the same counters."""
    key = (device, stream_handle() if device.type == "cuda" else 0)
    t = _fin_counters.get(key)
    if t is None:
        t = _fin_counters[key] = torch.zeros(FIN_COUNTERS, dtype=torch.int32, device=device)
    return t


class BNState(object):
    __slots__ = ("mean", "invstd", "scale", "shift")

    def __init__(self, C, device):
        buf = torch.empty((4, C), dtype=torch.float32, device=device)
        self.mean, self.invstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]


def bn_finalize_train(stats, count, gamma, beta, rm, rv, momentum=0.1, eps=1e-5):
    C = stats.shape[1]
    st = BNState(C, stats.device)
    call("mpn_bn_finalize_train", ptr(stats), stats.shape[0], C, count, ptr(gamma), ptr(beta), ptr(rm), ptr(rv),
         momentum, eps, ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), stream_ptr())
    return st


def bn_finalize_eval(gamma, beta, rm, rv, eps=1e-5):
    C = gamma.numel()
    st = BNState(C, gamma.device)
    call("mpn_bn_finalize_eval", C, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), eps, ptr(st.mean), ptr(st.invstd),
         ptr(st.scale), ptr(st.shift), stream_ptr())
    return st


def bn_act(y, st, relu, res=None, needs_grad=False, tag="", want_mask=False):
    """z = act(y * scale + shift [+ res]).  want_mask (with relu): also the sign bits of z, one byte per 16-byte chunk — all the
    backward of relu(bn(.) + shortcut) needs of z (the ReLU mask), at 1/16 of z's bytes (z.mask)."""
    z = Act(torch.empty_like(y.t), y.C, needs_grad, tag)
    if want_mask and relu:
        z.mask = torch.empty((y.P, y.Cs // (4 if y.t.dtype == torch.float32 else 8)), dtype=torch.uint8, device=y.t.device)
    call("mpn_bn_act_forward", ptr(y.t), ptr(res.t) if res is not None else None, ptr(z.t), ptr(st.scale), ptr(st.shift),
         y.P, y.C, y.Cs, 1 if relu else 0, dtype_code(y.t.dtype), ptr(z.mask), stream_ptr())
    return z


def masked_copy(dz, mask_bits, out):
    """out = dz * mask (mask bits in bn_act's layout): the shortcut gradient of relu(bn(.) + shortcut), materialised."""
    call("mpn_bn_bwd_apply", ptr(dz.t), None, None, None, None, None, None, None, None, ptr(out.t), 0, dz.P, dz.C, dz.Cs, 1,
         dtype_code(dz.t.dtype), ptr(mask_bits), stream_ptr())
    return out


def bn_backward(dz, z, y, st, gamma, relu, train, dgamma=None, dbeta=None, want_dy=True, dres=None, dres_acc=False, remask=False,
                partial=None, coef=None):
    """Returns dy (Act or None).  dres (Act) receives/accumulates g = dz*(z>0).  remask=True (forward without a
    residual input): the ReLU mask is recomputed from y and the forward's scale/shift instead of reading z.
    partial: per-tile (sum g, sum g*xhat) already produced by the launch that wrote dz (conv_forward(bnb=...)); the
    reduction pass over dz / y / z is skipped."""
    dev = y.t.device
    P, C, Cs = y.P, y.C, y.Cs
    dc = dtype_code(y.t.dtype)
    k1, k2, k3 = st.scale, None, None        # frozen BN, no parameter gradients: dy = g * gamma * invstd
    if coef is not None:
        # reduction AND finalize already happened in the launch that completed dz (conv_forward(bnb=(..., fin))): dgamma / dbeta are
        # in place, coef = [3][C] k1, k2, k3 (or "frozen": k1 = gamma * invstd = the forward scale)
        if not isinstance(coef, str):
            k1, k2, k3 = coef[0], coef[1], coef[2]
    elif train or dgamma is not None or dbeta is not None:
        if partial is not None:
            part, chunks = partial, partial.shape[0]
        else:
            chunks = call("mpn_bn_bwd_chunks", P, Cs, dc)
            part = workspace(chunks * C * 2 * 4, dev, slot=3)
            call("mpn_bn_bwd_reduce", ptr(dz.t), ptr(z.t) if (relu and not remask) else None, ptr(y.t), ptr(st.mean), ptr(st.invstd),
                 ptr(st.scale) if remask else None, ptr(st.shift) if remask else None, ptr(part),
                 chunks, P, C, Cs, 1 if relu else 0, dc, stream_ptr())
        coef = torch.empty((3, C), dtype=torch.float32, device=dev) if train else None
        call("mpn_bn_bwd_finalize", ptr(part), chunks, C, P, ptr(gamma), ptr(st.mean), ptr(st.invstd), 1 if train else 0,
             ptr(dgamma), ptr(dbeta), ptr(coef), stream_ptr())
        if train:
            k1, k2, k3 = coef[0], coef[1], coef[2]
    dy = None
    if want_dy or dres is not None:
        if want_dy:
            dy = Act(torch.empty_like(y.t), C)
        bits = z.mask if (relu and not remask) else None
        call("mpn_bn_bwd_apply", ptr(dz.t), ptr(z.t) if (relu and not remask and bits is None) else None, ptr(y.t), ptr(k1), ptr(k2), ptr(k3),
             ptr(st.scale) if remask else None, ptr(st.shift) if remask else None,
             ptr(dy.t) if dy is not None else None, ptr(dres.t) if dres is not None else None,
             1 if dres_acc else 0, P, C, Cs, 1 if relu else 0, dc, ptr(bits), stream_ptr())
    return dy


def maxpool_forward(x, needs_grad=False):
    Ho, Wo = (x.H + 2 - 3) // 2 + 1, (x.W + 2 - 3) // 2 + 1
    y = Act(torch.empty((x.B, Ho, Wo, x.Cs), dtype=x.t.dtype, device=x.t.device), x.C, needs_grad)
    idx = torch.empty((x.B, Ho, Wo, x.Cs), dtype=torch.uint8, device=x.t.device) if needs_grad else None
    call("mpn_maxpool3x3s2_forward", ptr(x.t), ptr(y.t), ptr(idx), x.B, x.H, x.W, x.Cs, Ho, Wo, dtype_code(x.t.dtype), stream_ptr())
    return y, idx


def maxpool_backward(dy, idx, x_like):
    dx = Act(torch.empty_like(x_like.t), x_like.C)
    call("mpn_maxpool3x3s2_backward", ptr(dy.t), ptr(idx), ptr(dx.t), x_like.B, x_like.H, x_like.W, x_like.Cs, dy.H, dy.W,
         dtype_code(dy.t.dtype), stream_ptr())
    return dx


def upsample_backward(dfine, dcoarse, accumulate):
    call("mpn_upsample_nearest_backward", ptr(dfine.t), ptr(dcoarse.t), dfine.B, dfine.H, dfine.W, dcoarse.H, dcoarse.W,
         dfine.Cs, 1 if accumulate else 0, dtype_code(dfine.t.dtype), stream_ptr())


def upsample_slice(src, dst, c_off):
    call("mpn_upsample_nearest_slice", ptr(src.t), ptr(dst.t), src.B, src.H, src.W, src.Cs, dst.H, dst.W, dst.Cs, c_off,
         dtype_code(src.t.dtype), stream_ptr())


def upsample_slice_backward(ddst, dsrc, c_off):
    call("mpn_upsample_nearest_slice_backward", ptr(ddst.t), ptr(dsrc.t), dsrc.B, dsrc.H, dsrc.W, dsrc.Cs, ddst.H, ddst.W,
         ddst.Cs, c_off, dtype_code(dsrc.t.dtype), stream_ptr())


def export_f32(src, C, Ho, Wo):
    """Internal padded [B,h,w,Cs] -> exact f32 tensor of logical shape [B,C,Ho,Wo] (channels_last)."""
    out = torch.empty((src.B, Ho, Wo, C), dtype=torch.float32, device=src.t.device)
    call("mpn_export_f32", ptr(src.t), dtype_code(src.t.dtype), ptr(out), src.B, src.H, src.W, src.Cs, C, Ho, Wo,
         Ho * Wo * C, C, stream_ptr())
    return out.permute(0, 3, 1, 2)


def import_grad(g_nchw, like, dtype):
    """f32 gradient w.r.t. an exported tensor (logical [B,C,Ho,Wo], any strides w/ channel stride 1
    after permute) -> internal padded gradient Act shaped like ``like`` (sums nearest children)."""
    B, C, Ho, Wo = g_nchw.shape
    g = g_nchw.permute(0, 2, 3, 1)
    if g.stride(3) != 1 or g.stride(1) != Wo * g.stride(2):
        g = g.contiguous()
    d = Act(torch.empty((like.B, like.H, like.W, like.Cs), dtype=dtype, device=like.t.device), like.C)
    call("mpn_import_grad", ptr(g), g.stride(0), g.stride(2), ptr(d.t), dtype_code(dtype), like.B, like.H, like.W, like.Cs,
         like.C, Ho, Wo, stream_ptr())
    return d


def relu_forward(x, needs_grad=False):
    y = Act(torch.empty_like(x.t), x.C, needs_grad)
    call("mpn_relu_forward", ptr(x.t), ptr(y.t), x.t.numel(), dtype_code(x.t.dtype), stream_ptr())
    return y


def relu_backward(dz, z, dx=None, accumulate=False):
    if dx is None:
        dx = Act(torch.empty_like(z.t), z.C)
        accumulate = False
    call("mpn_relu_backward", ptr(dz.t), ptr(z.t), ptr(dx.t), z.t.numel(), 1 if accumulate else 0, dtype_code(z.t.dtype), stream_ptr())
    return dx


def copy_act(dst, src):
    """dst <- src for two activations of identical storage geometry."""
    call("mpn_copy_bytes", ptr(dst.t), ptr(src.t), src.t.numel() * src.t.element_size(), stream_ptr())


def add_inplace(dst, src):
    call("mpn_add_inplace", ptr(dst.t), ptr(src.t), dst.t.numel(), dtype_code(dst.t.dtype), stream_ptr())


def nchw_to_nhwc_f32(t):
    """Arbitrary-strided f32 [B,C,H,W] -> dense [B,H,W,C] f32."""
    B, C, H, W = t.shape
    out = torch.empty((B, H, W, C), dtype=torch.float32, device=t.device)
    call("mpn_nchw_to_nhwc_f32", ptr(t), t.stride(0), t.stride(1), t.stride(2), t.stride(3), ptr(out), B, C, H, W, stream_ptr())
    return out


def nms(dets, thresh, mode=0):
    """dets: f32 [N,5] device tensor.  Returns int64 device tensor of kept ORIGINAL indices (one D2H of the count)."""
    n = dets.shape[0]
    dev = dets.device
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=dev)
    dets = dets.contiguous()
    keep = torch.empty((n,), dtype=torch.int64, device=dev)
    num = torch.empty((1,), dtype=torch.int64, device=dev)
    ws = workspace(call("mpn_nms_workspace_bytes", n), dev, slot=4)
    call("mpn_nms", ptr(dets), n, float(thresh), mode, ptr(keep), ptr(num), ptr(ws), stream_ptr())
    k = int(num.item())
    return keep[:k]


def box_decode_clip(anchors, deltas, img_w, img_h, clip=True, mean_std=None):
    """mean_std: None = the reference defaults (mean 0, std .1 .1 .2 .2), or the eight floats (mean[0..3], std[0..3]) of
    BBoxTransform(mean, std) (network/utils.py:8-17)."""
    B, A, _ = deltas.shape
    boxes = torch.empty((B, A, 4), dtype=torch.float32, device=deltas.device)
    w, h = (float(img_w), float(img_h)) if clip else (-1.0, -1.0)
    if mean_std is None:
        call("mpn_box_decode_clip", ptr(anchors), ptr(deltas), ptr(boxes), B, A, w, h, stream_ptr())
    else:
        ms = (ctypes.c_float * 8)(*[float(v) for v in mean_std])
        call("mpn_box_decode_clip_ms", ptr(anchors), ptr(deltas), ptr(boxes), B, A, w, h, ms, stream_ptr())
    return boxes


def clip_boxes_(boxes, img_w, img_h):
    if not boxes.is_contiguous():
        raise _lib.MpnError("clip_boxes_ needs a contiguous [B,A,4] tensor")
    call("mpn_clip_boxes", ptr(boxes), boxes.numel() // 4, float(img_w), float(img_h), stream_ptr())
    return boxes


def gather_dets(dets, keep):
    k = keep.numel()
    boxes = torch.empty((k, 4), dtype=torch.float32, device=dets.device)
    scores = torch.empty((k,), dtype=torch.float32, device=dets.device)
    if k > 0:
        call("mpn_gather_dets", ptr(dets), ptr(keep), k, ptr(boxes), ptr(scores), stream_ptr())
    return boxes, scores


_NMS_WARNED = []


def nms_pre_topn_env():
    """MPN_NMS_PRE_TOPN, read at CALL time: cap of candidates per image entering NMS when the caller passes none.  Unset / 0 = no cap —
    the reference's behaviour (posenet.py:269-285 hands every candidate above the score threshold to nms); there is NO default cap."""
    try:
        return max(0, int(os.environ.get("MPN_NMS_PRE_TOPN", "0")))
    except ValueError:
        return 0


def detect_batched(boxes, scores, score_thresh, iou_thresh, mode=0, ws_limit=4 << 30, padded=False, pre_nms_top_n=None):
    """Score filter + per-image NMS + gather for EVERY image of a batch with two host round trips per batch (the candidate
    counts size the NMS launches, the kept counts size the returned tensors) instead of two per image.
    boxes [B,A,4], scores [B,A] (f32, contiguous).  Returns per image (boxes[k,4], scores[k]) device tensors (views); with
    padded=True the batch-wide tensors themselves: (boxes [B,nmax,4], scores [B,nmax] in descending order, kept counts list).
    pre_nms_top_n (not in the reference; None = every candidate, as posenet.py:269-285): only that many best-scored candidates of
    an image enter the suppression, which bounds its N x N/64 mask."""
    B, A = scores.shape[0], scores.shape[1]
    dev = scores.device
    dets = torch.empty((B, A, 5), dtype=torch.float32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    call("mpn_score_filter_batched", ptr(boxes), ptr(scores), B, A, float(score_thresh), ptr(dets), None, ptr(counts), stream_ptr())
    cnt = counts.tolist()                              # host round trip 1
    nmax = max(cnt)
    if nmax == 0:
        if padded:
            return torch.zeros((B, 0, 4), dtype=torch.float32, device=dev), torch.zeros((B, 0), dtype=torch.float32, device=dev), [0] * B
        return [(None, None)] * B
    ncand = nmax
    if pre_nms_top_n is None and nms_pre_topn_env() > 0:
        pre_nms_top_n = nms_pre_topn_env()
    top = int(pre_nms_top_n) if pre_nms_top_n else 0
    if top == 0 and nmax > 32768 and not _NMS_WARNED:
        # the reference's behaviour (every candidate above the score threshold enters the suppression) costs the upper triangle of an
        # N x N / 64 mask per image: 67 MB at 32 768 candidates, 0.6 GB at 100 000 — say so once instead of silently taking seconds
        import warnings
        _NMS_WARNED.append(True)
        warnings.warn("NMS over %d candidates in one image (upper-triangle N x N / 128 mask = %.0f MB per image); pass pre_nms_top_n or set "
                      "MPN_NMS_PRE_TOPN to bound it (not in the reference)" % (nmax, nmax * float(nmax) / 128 * 8 / 1e6))
    if top > 0:
        nmax = min(nmax, top)                          # rows of the sort / mask scratch and of the outputs
    keep = torch.empty((B, nmax), dtype=torch.int64, device=dev)
    num = torch.empty((B,), dtype=torch.int64, device=dev)
    per_img = call("mpn_nms_batched_workspace_bytes", 1, nmax)
    group = max(1, min(B, int(ws_limit // per_img)))   # bound the N x N/64 mask scratch; groups of images per launch set
    for b0 in range(0, B, group):
        nb = min(group, B - b0)
        ws = workspace(per_img * nb, dev, slot=4)
        head = (ctypes.c_void_p(dets.data_ptr() + b0 * A * 5 * 4), A * 5, ctypes.c_void_p(counts.data_ptr() + b0 * 4), nb, ncand)
        tail = (float(iou_thresh), mode, ctypes.c_void_p(keep.data_ptr() + b0 * nmax * 8), nmax, ctypes.c_void_p(num.data_ptr() + b0 * 8),
                ptr(ws), stream_ptr())
        if top > 0:
            call("mpn_nms_batched_topk", *(head + (top,) + tail))
        else:
            call("mpn_nms_batched", *(head + tail))
    out_boxes = torch.empty((B, nmax, 4), dtype=torch.float32, device=dev)
    out_scores = torch.empty((B, nmax), dtype=torch.float32, device=dev)
    call("mpn_gather_dets_batched", ptr(dets), A * 5, ptr(keep), nmax, ptr(num), B, nmax, ptr(out_boxes), ptr(out_scores), nmax, stream_ptr())
    kept = num.tolist()                                # host round trip 2
    if padded:
        return out_boxes, out_scores, [k if cnt[b] > 0 else 0 for b, k in enumerate(kept)]
    return [(out_boxes[b, :k], out_scores[b, :k]) if cnt[b] > 0 else (None, None) for b, k in enumerate(kept)]


def score_filter(boxes0, scores0, thresh):
    """Image-0 candidates with score > thresh: returns (dets[n,5], src_idx[n]) (one D2H of n)."""
    A = scores0.numel()
    dev = scores0.device
    dets = torch.empty((A, 5), dtype=torch.float32, device=dev)
    src = torch.empty((A,), dtype=torch.int32, device=dev)
    cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    call("mpn_score_filter", ptr(boxes0), ptr(scores0), A, float(thresh), ptr(dets), ptr(src), ptr(cnt), stream_ptr())
    n = int(cnt.item())
    return dets[:n], src[:n]
