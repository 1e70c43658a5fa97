"""Forward/backward runtime of the hot path.

The reference lets torch autograd drive ~400 separate ATen ops per step (network/fpn.py,
network/posenet.py) and accumulates gradients with extra elementwise kernels.  Here the whole
poseNet forward is recorded on an explicit tape of fused HIP launches (``ops``); ``loss.backward()``
enters through ONE autograd node (``_NetFn``) and the tape is replayed in reverse with
caller-controlled gradient buffers: every kernel either overwrites or accumulates (dgrad /
bn_bwd / upsample_bwd take an ``accumulate`` flag), so no torch compute kernel runs in either pass.
Weight gradients are accumulated straight into the flat gradient arena; when a data-parallel reducer
is attached it is told as each parameter's gradient completes, so RCCL all-reduces of finished
arena slices overlap the rest of backward.
"""
import ctypes
import os

import torch

from . import ops
from ._lib import call, gpu_op
from .ops import Act, round_up


class Ctx(object):
    """State of one forward pass (tape + gradient buffers)."""

    def __init__(self, train):
        self.train = train
        self.tape = []
        self.grads = {}          # id(Act) -> Act
        self.keep = []           # keeps Acts alive while their id() is used as a key
        self.out_grads = {}      # export slot -> torch grad tensor (filled by _NetFn.backward)
        self.wt = {}             # id(weight param) -> transposed operand
        self.wt_buf = None       # one buffer holding every layer's transposed operand for this pass
        self.uses = {}           # id(param) -> number of pending gradient contributions
        self.bn_train_ran = False
        self.side_keep = []      # operands of side-stream launches, kept alive until the streams join
        self.side_pending = []   # closures waiting for the next fork point (Engine.fork_every > 1)
        self.wt_ready = None     # event: the transposed weight copies (made on the side stream) are complete
        self.bnb = {}            # id(BN output Act) -> per-tile backward statistics produced by the launch that completed its gradient
        self.fwd_side_join = False   # forward work is in flight on the side stream (Engine.det_pyramid): join before it is consumed
        self.cls_ev = None       # event: the class input gradients of conv2 (side stream) are complete (Engine.wait_class_grads)
        self.schedule = None     # bucket schedule of this backward pass: the data-parallel GradReducer, or its process-group-free
                                 # twin when one GPU runs per-bucket optimizer updates (Engine.run_backward)
        self.lazy_res = {}       # id(Act) -> (dz, mask bits): shortcut gradient dz * (z > 0) NOT materialised; the convolution whose
                                 # input gradient completes that Act adds it in its epilogue (Engine.defer_shortcut_grad)

    def gbuf(self, act, dtype=None):
        """Gradient buffer for ``act``: returns (Act, existed)."""
        g = self.grads.get(id(act))
        if g is not None:
            return g, True
        g = Act(torch.empty(act.t.shape, dtype=dtype or act.t.dtype, device=act.t.device), act.C)
        self.grads[id(act)] = g
        self.keep.append(act)
        return g, False

    def grad_of(self, act):
        return self.grads.get(id(act))

    def set_grad(self, act, g):
        self.grads[id(act)] = g
        self.keep.append(act)

    def pop_grad(self, act):
        return self.grads.pop(id(act), None)


def _geom(layer):
    """(Cout, Cin, R, S, stride, pad) of an nn.Conv2d, or of an nn.Linear seen as a 1x1 conv on a 1x1 image."""
    w = layer.weight
    if w.dim() == 2:
        return w.shape[0], w.shape[1], 1, 1, 1, 0
    return w.shape[0], w.shape[1], w.shape[2], w.shape[3], layer.stride[0], layer.padding[0]


class Engine(object):
    def __init__(self, model):
        self.m = model
        self._drop_calls = 0
        self._side = None
        self._wt_plan_cache = None
        self.overlap_wgrad = os.environ.get("MPN_SIDE_STREAM", "1") != "0"
        # side-stream work is handed over in groups of `fork_every` layers (one event record / wait per group): a captured
        # hipGraph pays for every cross-stream edge, the eager tape does not care much
        self.fork_every = max(1, int(os.environ.get("MPN_SIDE_FORK_EVERY", "1")))
        # the RetinaNet towers share their weights over p3..p7: one launch per layer over the whole pyramid instead of one per level
        self.pyramid_towers = os.environ.get("MPN_PYRAMID_TOWERS", "1") != "0"
        # BatchNorm-backward statistics ride in the epilogue of the dgrad launch that completes dz (no separate reduction pass)
        self.fuse_bn_stats = os.environ.get("MPN_BN_FUSED_STATS", "1") != "0"
        # inference with frozen statistics: BatchNorm (+ReLU, + the bottleneck's residual add) folds into the conv epilogue
        self.fold_bn = os.environ.get("MPN_FOLD_BN", "1") != "0"
        # the BatchNorm finalize steps (tile partials -> coefficients) run inside the producing conv launch (last-arriving workgroup)
        self.fuse_bn_finalize = os.environ.get("MPN_BN_FUSED_FINALIZE", "1") != "0"
        # forward: the detection pyramid runs on the side stream (0 = off, 1 = only its two tiny-grid stride-2 convolutions P6 / P7)
        self.det_pyramid_side = int(os.environ.get("MPN_DET_PYRAMID_SIDE", "2"))
        # relu(bn3(.) + shortcut): the forward also writes the sign bits of z (1/16 of its bytes); both backward passes that need
        # the ReLU mask (statistics in the dgrad epilogue, bn_bwd_apply) read those instead of z
        self.bn_mask_bits = os.environ.get("MPN_BN_MASK_BITS", "1") != "0"
        # ... and the shortcut gradient of an identity block, dz * (z > 0), is not written by bn_bwd_apply at all: conv1's input-gradient
        # launch (the only other contribution to that tensor) reads dz and the bits in its epilogue (MpnConvParams.res_mask)
        self.defer_shortcut_grad = os.environ.get("MPN_DEFER_SHORTCUT_GRAD", "1") != "0"
        # torch.cat((up8(q5), up4(q4), up2(q3), q2), 1) -> conv2 (posenet.py:311-315): the 512-channel tensor is never written; conv2's
        # forward and weight-gradient launches gather from the four members, its input gradient lands in q2's gradient directly
        self.virtual_concat = os.environ.get("MPN_VIRTUAL_CONCAT", "1") != "0"
        # ... and its x8 / x4 members do not go through conv2 at full resolution at all: nine position-class maps each at their own
        # resolution (csrc/conv2cls.hip), expanded into conv2's epilogue; conv2 itself contracts over the x2 / x1 members only
        self.conv2_classes = os.environ.get("MPN_CONV2_CLASSES", "1") != "0"

    def side_stream(self, device):
        """Second HIP stream for weight/bias gradients.  They are off the backward critical path (only the
        optimizer / all-reduce needs them), while dgrad -> BN backward -> dgrad is a serial chain of launches that
        individually leave CUs idle (layer3/4 grids of 450-900 workgroups, HBM-bound BN passes): running the
        wgrad of layer L beside the dgrad/BN work of layers < L fills those holes."""
        if not self.overlap_wgrad or device.type != "cuda":
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def _on_side(self, ctx, device, keep, fn, torch_ops=False):
        """Run fn() on the side stream, ordered after everything enqueued so far on the current stream.  Our own
        launches take the stream explicitly (ops.push_stream); only closures that also run torch ops (torch_ops=True)
        pay for switching torch's current stream."""
        side = self.side_stream(device)
        if side is None:
            fn()
            return
        ctx.side_keep.append(keep)
        ctx.side_pending.append((fn, torch_ops))
        if len(ctx.side_pending) >= self.fork_every:
            self.flush_side(ctx, device)

    def flush_side(self, ctx, device):
        """Fork point: everything enqueued so far on the current stream happens-before the pending side-stream closures."""
        if not ctx.side_pending:
            return
        side = self.side_stream(device)
        pending, ctx.side_pending = ctx.side_pending, []
        ev = torch.cuda.Event()
        gpu_op(ev.record, torch.cuda.current_stream(device))
        gpu_op(side.wait_event, ev)
        for fn, torch_ops in pending:
            if torch_ops:
                with torch.cuda.stream(side):
                    fn()
                continue
            ops.push_stream(side)
            try:
                fn()
            finally:
                ops.pop_stream()

    # ------------------------------------------------------------------ weights
    @property
    def cdt(self):
        return self.m.compute_dtype

    def w_fwd(self, layer):
        """Forward operand [Cout][R][S][Cin] in the compute dtype (a view, never a copy in f32)."""
        ar = self.m._arena
        w = layer.weight
        kc = 32 if ops.is16(self.cdt) else 16
        if w.dim() == 2 and w.shape[1] % kc != 0:
            # Linear layer whose fan-in is not a whole number of 64-byte K chunks (PRN with prn_coeff 1 or 3): zero-padded rows
            K, Kp = w.shape[1], round_up(w.shape[1], kc)
            dst = torch.empty((w.shape[0], Kp), dtype=self.cdt, device=w.device)
            call("mpn_weight_pad_k", ops.ptr(ar.data_seg(w)), ops.ptr(dst), w.shape[0], K, Kp, ops.dtype_code(self.cdt), ops.stream_ptr())
            return dst
        if self.cdt == torch.float32:
            return ar.data_seg(layer.weight)
        return ar.data_seg(layer.weight, ar.lowp[self.cdt])

    def _wt_plan(self):
        """Layout of every conv / linear layer's dgrad operand Wt[Cin][R][S][Cout_pad] inside one buffer, plus the device
        table mpn_weight_transpose_batched walks (built once per arena)."""
        ar = self.m._arena
        plan = self._wt_plan_cache
        if plan is not None and plan["arena"] is ar and plan["dtype"] == self.cdt:
            return plan
        kc = 32 if ops.is16(self.cdt) else 16
        rows, views, off, blk = [], {}, 0, 0
        stem = self.m.fpn.conv1.weight
        prn_w = {id(q) for q in self.m.prn.parameters()}      # the PRN's Linear layers (71 M parameters) are not part of the
        for mod in self.m.modules():                            # backbone / head passes: single transposes when the PRN trains
            w = getattr(mod, "weight", None)
            if not isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear)) or w is None or id(w) not in ar.index or w is stem or id(w) in prn_w:
                continue
            if id(w) in views:
                continue
            O, I, R, S, _, _ = _geom(mod)
            opad = round_up(O, kc)
            gx, gy = (I + 31) // 32, (opad + 31) // 32
            rows.append([ar.offsets[ar.index[id(w)]], off, O, R * S, I, opad, blk, gx])
            views[id(w)] = (off, (I, R, S, opad))
            off += round_up(I * R * S * opad, 64)
            blk += gx * gy * R * S
        plan = {"arena": ar, "dtype": self.cdt, "views": views, "total": off, "blocks": blk, "n": len(rows),
                "table": torch.tensor(rows, dtype=torch.int64, device=ar.flat.device)}
        self._wt_plan_cache = plan
        return plan

    def w_t(self, ctx, layer):
        """dgrad operand Wt[Cin][R][S][Cout_pad]: all layers are transposed by ONE launch the first time a forward
        pass asks for one (a buffer per pass, so a later optimizer step cannot change what this tape will read)."""
        key = id(layer.weight)
        wt = ctx.wt.get(key)
        if wt is not None:
            return wt
        plan = self._wt_plan()
        if key not in plan["views"]:          # not an arena conv/linear weight: single transpose
            O, I, R, S, _, _ = _geom(layer)
            kc = 32 if ops.is16(self.cdt) else 16
            opad = round_up(O, kc)
            wt = torch.empty((I, R, S, opad), dtype=self.cdt, device=layer.weight.device)
            ops.weight_transpose(self.m._arena.data_seg(layer.weight), wt, O, R * S, I, opad)
            ctx.wt[key] = wt
            return wt
        if ctx.wt_buf is None:
            dev = plan["table"].device
            ctx.wt_buf = torch.empty(plan["total"], dtype=self.cdt, device=dev)
            side = self.side_stream(dev)
            if side is not None:
                ctx.wt_buf.record_stream(side)

            def transpose_all():
                call("mpn_weight_transpose_batched", ops.ptr(self.m._arena.flat), ops.ptr(ctx.wt_buf), ops.ptr(plan["table"]),
                     plan["n"], plan["blocks"], ops.dtype_code(self.cdt), ops.stream_ptr())
            if side is None:
                transpose_all()
            else:
                # only backward reads these copies: make them beside the forward pass (they depend on nothing but the last
                # optimizer step, which the fork below orders) and let run_backward wait for the event
                ev = torch.cuda.Event()
                gpu_op(ev.record, torch.cuda.current_stream(dev))
                gpu_op(side.wait_event, ev)
                ops.push_stream(side)
                try:
                    transpose_all()
                finally:
                    ops.pop_stream()
                ctx.wt_ready = torch.cuda.Event()
                gpu_op(ctx.wt_ready.record, side)
        off, shape = plan["views"][key]
        n = shape[0] * shape[1] * shape[2] * shape[3]
        wt = ctx.wt_buf[off: off + n].view(shape)
        ctx.wt[key] = wt
        return wt

    def _note_use(self, ctx, p):
        if p is not None and p.requires_grad:
            ctx.uses[id(p)] = ctx.uses.get(id(p), 0) + 1

    def _grad_done(self, ctx, p):
        n = ctx.uses.get(id(p), 0) - 1
        ctx.uses[id(p)] = n
        if n == 0 and ctx.schedule is not None:
            ctx.schedule.param_ready(p)

    # ------------------------------------------------------------------ ops
    def _bn_fin(self, bn):
        """Arguments of the in-launch forward finalize for BatchNorm layer `bn` (train mode)."""
        if bn is None or not self.fuse_bn_finalize:
            return None
        return (bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, bn.momentum if bn.momentum is not None else 0.1, bn.eps)

    def conv(self, ctx, x, layer, act=0, res=None, res_mode=0, stats=False, out_f32=False, tag="", bn=None):
        O, I, R, S, stride, pad = _geom(layer)
        bias = layer.bias
        y, st = ops.conv_forward(x, self.w_fwd(layer), O, R, S, stride, pad, bias=bias.data if bias is not None else None,
                                 act=act, res=res, res_mode=res_mode, want_stats=stats, out_f32=out_f32, tag=tag,
                                 bn_fin=self._bn_fin(bn) if stats else None)
        y.relu_out = act == 1
        if ctx.train:
            y.needs_grad = bool(x.needs_grad or layer.weight.requires_grad or (bias is not None and bias.requires_grad)
                                or (res is not None and res.needs_grad))
            if y.needs_grad:
                self._note_use(ctx, layer.weight)
                self._note_use(ctx, bias)
                if x.needs_grad:
                    self.w_t(ctx, layer)       # make the transposed operand now (weights may change before backward)
                    x.cons += 1
                    x.conv_cons += 1
                if res is not None and res.needs_grad:
                    res.cons += 1
                    res.other_cons += 1
                ctx.tape.append(lambda: self._conv_bwd(ctx, x, layer, y, act, res, res_mode))
        return y, st

    def _conv_bwd(self, ctx, x, layer, y, act, res, res_mode):
        dy = ctx.pop_grad(y)
        O, I, R, S, stride, pad = _geom(layer)
        bias = layer.bias
        if dy is None:       # nothing flowed back through this output (e.g. unused head)
            if layer.weight.requires_grad:
                self._grad_done(ctx, layer.weight)
            if bias is not None and bias.requires_grad:
                self._grad_done(ctx, bias)
            if x.needs_grad:
                x.cons -= 1
                x.conv_cons -= 1
                lazy = ctx.lazy_res.pop(id(x), None)
                if lazy is not None:            # the deferred shortcut gradient has no convolution to ride on: materialise it
                    g, existed = ctx.gbuf(x)
                    assert not existed
                    ops.masked_copy(lazy[0], lazy[1], g)
            if res is not None and res.needs_grad:
                res.cons -= 1
            return
        if dy.t.dtype != self.cdt:
            raise ops._lib.MpnError("gradient dtype mismatch for %s" % y.tag)
        if act == 1:
            dy = ops.relu_backward(dy, y)
        elif act == 2:
            raise ops._lib.MpnError("sigmoid backward is handled at the detection edge")
        if res is not None and res.needs_grad:
            res.cons -= 1
            g, existed = ctx.gbuf(res)
            if res_mode == 2:
                ops.upsample_backward(dy, g, existed)
            elif existed:
                ops.add_inplace(g, dy)
            else:
                ops.copy_act(g, dy)
        ar = self.m._arena
        wg = layer.weight.requires_grad
        bg = bias is not None and bias.requires_grad
        if wg or bg:
            def param_grads():
                done = False
                if wg:           # the bias gradient rides along in the wgrad kernel where the LDS-DMA path serves it
                    done = ops.conv_wgrad(x, dy, ar.grad_seg(layer.weight), O, R, S, stride, pad,
                                          db=ar.grad_seg(bias) if bg else None)
                if bg and not done:
                    ops.bias_grad(dy, ar.grad_seg(bias), O)
            self._on_side(ctx, dy.t.device, (x, dy), param_grads)
            if wg:
                self._grad_done(ctx, layer.weight)
            if bg:
                self._grad_done(ctx, bias)
        if x.needs_grad:
            x.cons -= 1
            x.conv_cons -= 1
            g, existed = ctx.gbuf(x)
            lazy = ctx.lazy_res.pop(id(x), None)      # (dz, bits) of the identity shortcut: added in this launch's epilogue
            wt = self.w_t(ctx, layer)
            bnb = None
            if x.cons == 0 and x.bn_src is not None and self.fuse_bn_stats:
                # this launch completes dz of the BatchNorm that produced x: its backward statistics ride in the epilogue
                by, st, relu, has_res, wants_stats, bn_layer, train_stats = x.bn_src
                if wants_stats and by.t.dtype == g.t.dtype:
                    fin = None
                    if self.fuse_bn_finalize:
                        ar = self.m._arena
                        fin = (bn_layer.weight.data, train_stats,
                               ar.grad_seg(bn_layer.weight) if bn_layer.weight.requires_grad else None,
                               ar.grad_seg(bn_layer.bias) if bn_layer.bias.requires_grad else None)
                    bnb = (by, x if (relu and has_res) else None, st, relu, fin)
            if lazy is not None:
                assert not existed
                _, part = ops.conv_forward(dy, wt, I, R, S, stride, pad, mode=1, out_hw=(x.H, x.W), cin=wt.shape[3], out=g, bnb=bnb,
                                           res=lazy[0], res_mode=1, res_mask=lazy[1])
            else:
                _, part = ops.conv_forward(dy, wt, I, R, S, stride, pad, mode=1, out_hw=(x.H, x.W), cin=wt.shape[3], out=g, accumulate=existed,
                                           bnb=bnb)
            if bnb is not None:
                ctx.bnb[id(x)] = part

    # ------------------------------------------------------------------ shared-weight convolution over a pyramid
    def conv_seg(self, ctx, xs, layer, act=0, out_f32=False):
        """layer(x_l) for every level l in ONE launch (stride-1 same-size convolution; posenet.py:327-328 loops over levels)."""
        O, I, R, S, stride, pad = _geom(layer)
        bias = layer.bias
        ys = ops.conv_forward_seg(xs, self.w_fwd(layer), O, R, S, pad, bias=bias.data if bias is not None else None, act=act, out_f32=out_f32)
        for y in ys:
            y.relu_out = act == 1
        if ctx.train:
            need = bool(any(x.needs_grad for x in xs) or layer.weight.requires_grad or (bias is not None and bias.requires_grad))
            for y in ys:
                y.needs_grad = need
            if need:
                self._note_use(ctx, layer.weight)
                self._note_use(ctx, bias)
                if any(x.needs_grad for x in xs):
                    self.w_t(ctx, layer)
                ctx.tape.append(lambda: self._conv_seg_bwd(ctx, xs, layer, ys, act))
        return ys

    def _conv_seg_bwd(self, ctx, xs, layer, ys, act):
        dys = [ctx.pop_grad(y) for y in ys]
        O, I, R, S, stride, pad = _geom(layer)
        bias = layer.bias
        wg = layer.weight.requires_grad
        bg = bias is not None and bias.requires_grad
        if all(d is None for d in dys):
            if wg:
                self._grad_done(ctx, layer.weight)
            if bg:
                self._grad_done(ctx, bias)
            return
        if any(d is None for d in dys):
            raise ops._lib.MpnError("pyramid convolution: gradient missing for some levels")
        if act == 1:          # ReLU mask over the whole pyramid in one launch when both sides are single buffers
            fd, fy = ops.seg_flat(dys), ops.seg_flat(ys)
            if fd is not None and fy is not None and fd.numel() == fy.numel():
                out = ops.alloc_seg(ys, ys[0].C, dys[0].t.dtype)
                call("mpn_relu_backward", ops.ptr(fd), ops.ptr(fy), ops.ptr(out[0].seg[0]), fy.numel(), 0, ops.dtype_code(fy.dtype), ops.stream_ptr())
                dys = out
            else:
                dys = [ops.relu_backward(d, y) for d, y in zip(dys, ys)]
        ar = self.m._arena
        if wg or bg:
            def param_grads():
                handled, fused = (False, False)
                if wg:
                    handled, fused = ops.conv_wgrad_seg(xs, dys, ar.grad_seg(layer.weight), O, R, S, pad, db=ar.grad_seg(bias) if bg else None)
                for x, d in zip(xs, dys):          # per-level path (exact-fp32 kernels) / separate bias gradient
                    done = fused
                    if wg and not handled:
                        done = ops.conv_wgrad(x, d, ar.grad_seg(layer.weight), O, R, S, 1, pad, db=ar.grad_seg(bias) if bg else None)
                    if bg and not done:
                        ops.bias_grad(d, ar.grad_seg(bias), O)
            self._on_side(ctx, dys[0].t.device, (xs, dys), param_grads)
            if wg:
                self._grad_done(ctx, layer.weight)
            if bg:
                self._grad_done(ctx, bias)
        if any(x.needs_grad for x in xs):
            wt = self.w_t(ctx, layer)
            grouped = ops.seg_flat(xs) is not None
            if grouped and ctx.grad_of(xs[0]) is None:
                gs = ops.alloc_seg(xs, xs[0].C, xs[0].t.dtype)          # gradients of a pyramid group live in one buffer too
                for x, g in zip(xs, gs):
                    ctx.set_grad(x, g)
                existed = [False] * len(xs)
            else:
                pairs = [ctx.gbuf(x) for x in xs]
                gs, existed = [g for g, _ in pairs], [e for _, e in pairs]
            if all(existed) or not any(existed):
                ops.conv_forward_seg(dys, wt, I, R, S, pad, mode=1, cin=wt.shape[3], outs=gs, accumulate=existed[0])
            else:
                for d, x, g, e in zip(dys, xs, gs, existed):
                    ops.conv_forward(d, wt, I, R, S, 1, pad, mode=1, out_hw=(x.H, x.W), cin=wt.shape[3], out=g, accumulate=e)

    def bn(self, ctx, y, stats, layer, relu, res=None, tag=""):
        train_stats = layer.training
        want_mask = bool(self.bn_mask_bits and ctx.train and relu and res is not None)
        if train_stats:
            momentum = layer.momentum if layer.momentum is not None else 0.1
            if isinstance(stats, ops.BNState):         # the conv launch finalized in place
                st = stats
            else:
                st = ops.bn_finalize_train(stats, y.P, layer.weight.data, layer.bias.data, layer.running_mean, layer.running_var,
                                           momentum, layer.eps)
            ctx.bn_train_ran = True
        else:
            st = ops.bn_finalize_eval(layer.weight.data, layer.bias.data, layer.running_mean, layer.running_var, layer.eps)
        z = ops.bn_act(y, st, relu, res=res, tag=tag, want_mask=want_mask)
        if ctx.train:
            z.needs_grad = bool(y.needs_grad or layer.weight.requires_grad or layer.bias.requires_grad
                                or (res is not None and res.needs_grad))
            if z.needs_grad:
                self._note_use(ctx, layer.weight)
                self._note_use(ctx, layer.bias)
                if res is not None and res.needs_grad:
                    res.cons += 1
                    res.other_cons += 1
                wants_stats = bool(train_stats or layer.weight.requires_grad or layer.bias.requires_grad)
                z.bn_src = (y, st, relu, res is not None, wants_stats, layer, train_stats)
                ctx.tape.append(lambda: self._bn_bwd(ctx, y, z, st, layer, relu, res, train_stats))
        return z

    @staticmethod
    def _bnb_args(fused):
        """What the launch that completed dz left behind: nothing, per-tile partial statistics, or finished coefficients."""
        if fused is None:
            return {}
        if isinstance(fused, str) or (fused.dim() == 2 and fused.shape[0] == 3):
            return {"coef": fused}
        return {"partial": fused}

    def _bn_bwd(self, ctx, y, z, st, layer, relu, res, train_stats):
        dz = ctx.pop_grad(z)
        wg, bg = layer.weight.requires_grad, layer.bias.requires_grad
        if dz is None:
            if wg:
                self._grad_done(ctx, layer.weight)
            if bg:
                self._grad_done(ctx, layer.bias)
            if res is not None and res.needs_grad:
                res.cons -= 1
            return
        ar = self.m._arena
        dres, dres_acc = None, False
        if res is not None and res.needs_grad:
            res.cons -= 1
            if (self.defer_shortcut_grad and relu and z.mask is not None and res.cons == 1 and res.conv_cons == 1
                    and ctx.grad_of(res) is None and dz.t.dtype == res.t.dtype and dz.t.shape == res.t.shape):
                # identity shortcut whose only other gradient contribution is one convolution's input gradient (conv1 of this block):
                # g = dz * (z > 0) is not written here; that launch reads dz and the mask bits (kept alive by the entry)
                ctx.lazy_res[id(res)] = (dz, z.mask)
                ctx.keep.append(res)
            else:
                dres, dres_acc = ctx.gbuf(res)
        want_dy = y.needs_grad
        dy = ops.bn_backward(dz, z, y, st, layer.weight.data, relu, train_stats,
                             dgamma=ar.grad_seg(layer.weight) if wg else None,
                             dbeta=ar.grad_seg(layer.bias) if bg else None,
                             want_dy=want_dy, dres=dres, dres_acc=dres_acc, remask=bool(relu and res is None),
                             **self._bnb_args(ctx.bnb.pop(id(z), None)))
        if wg:
            self._grad_done(ctx, layer.weight)
        if bg:
            self._grad_done(ctx, layer.bias)
        if want_dy:
            ctx.set_grad(y, dy)

    def dropout(self, ctx, x, p):
        """nn.Dropout in training mode (posenet.py:139,341-342): counter-based mask, regenerated in backward."""
        self._drop_calls += 1
        seed = (torch.initial_seed() * 1000003 + self._drop_calls) & 0xFFFFFFFFFFFFFFFF
        y = Act(torch.empty_like(x.t), x.C)
        call("mpn_dropout", ops.ptr(x.t), ops.ptr(y.t), x.t.numel(), seed, float(p), ops.dtype_code(x.t.dtype), ops.stream_ptr())
        if ctx.train and x.needs_grad:
            y.needs_grad = True

            def bwd():
                dy = ctx.pop_grad(y)
                if dy is None:
                    return
                g = Act(torch.empty_like(dy.t), x.C)
                call("mpn_dropout", ops.ptr(dy.t), ops.ptr(g.t), dy.t.numel(), seed, float(p), ops.dtype_code(dy.t.dtype), ops.stream_ptr())
                ctx.set_grad(x, g)
            ctx.tape.append(bwd)
        return y

    def relu(self, ctx, x):
        z = ops.relu_forward(x)
        if ctx.train and x.needs_grad:
            z.needs_grad = True
            x.cons += 1
            x.other_cons += 1

            def bwd():
                x.cons -= 1
                dz = ctx.pop_grad(z)
                if dz is None:
                    return
                g, existed = ctx.gbuf(x)
                ops.relu_backward(dz, z, g, existed)
            ctx.tape.append(bwd)
        return z

    def maxpool(self, ctx, x):
        need = ctx.train and x.needs_grad
        y, idx = ops.maxpool_forward(x, needs_grad=need)
        if need:
            x.cons += 1
            x.other_cons += 1

            def bwd():
                x.cons -= 1
                dy = ctx.pop_grad(y)
                if dy is None:
                    return
                ctx.set_grad(x, ops.maxpool_backward(dy, idx, x))
            ctx.tape.append(bwd)
        return y

    def concat_up(self, ctx, srcs, Ho, Wo):
        """torch.cat((up8(p5), up4(p4), up2(p3), p2), 1)  — posenet.py:311-315 — as channel-slice writes."""
        B = srcs[0].B
        Ctot = sum(s.C for s in srcs)
        dst = Act(torch.empty((B, Ho, Wo, Ctot), dtype=srcs[0].t.dtype, device=srcs[0].t.device), Ctot)
        off = 0
        offs = []
        for s in srcs:
            ops.upsample_slice(s, dst, off)
            offs.append(off)
            off += s.C
        if ctx.train and any(s.needs_grad for s in srcs):
            dst.needs_grad = True

            def bwd():
                d = ctx.pop_grad(dst)
                if d is None:
                    return
                for s, o in zip(srcs, offs):
                    if s.needs_grad:
                        g = Act(torch.empty_like(s.t), s.C)
                        ops.upsample_slice_backward(d, g, o)
                        ctx.set_grad(s, g)
            ctx.tape.append(bwd)
        return dst

    def conv_cat(self, ctx, srcs, H, W, layer, act=0):
        """layer(torch.cat([nearest_upsample(s) to H x W for s in srcs], 1)) for a 3x3 / stride-1 layer, the concatenation virtual
        (ops.conv_forward_cat).  Backward: weight gradient gathers the same way; the input gradient of the full-resolution LAST
        member is written by the dgrad launch itself (split output), the up-sampled members' slices go through one buffer of
        their channels only and are summed down to their own resolution."""
        O, I, R, S, stride, pad = _geom(layer)
        bias = layer.bias
        y = ops.conv_forward_cat(srcs, H, W, self.w_fwd(layer), O, bias=bias.data if bias is not None else None, act=act)
        y.relu_out = act == 1
        if ctx.train:
            need_x = any(s.needs_grad for s in srcs)
            y.needs_grad = bool(need_x or layer.weight.requires_grad or (bias is not None and bias.requires_grad))
            if y.needs_grad:
                self._note_use(ctx, layer.weight)
                self._note_use(ctx, bias)
                if need_x:
                    self.w_t(ctx, layer)
                ctx.tape.append(lambda: self._conv_cat_bwd(ctx, srcs, H, W, layer, y, act))
        return y

    def _conv_cat_bwd(self, ctx, srcs, H, W, layer, y, act):
        dy = ctx.pop_grad(y)
        O, I, R, S, stride, pad = _geom(layer)
        bias = layer.bias
        wg = layer.weight.requires_grad
        bg = bias is not None and bias.requires_grad
        if dy is None:
            if wg:
                self._grad_done(ctx, layer.weight)
            if bg:
                self._grad_done(ctx, bias)
            return
        if act == 1:
            dy = ops.relu_backward(dy, y)
        ar = self.m._arena
        if wg or bg:
            def param_grads():
                done = False
                if wg:
                    done = ops.conv_wgrad_cat(srcs, H, W, dy, ar.grad_seg(layer.weight), O, db=ar.grad_seg(bias) if bg else None)
                if bg and not done:
                    ops.bias_grad(dy, ar.grad_seg(bias), O)
            self._on_side(ctx, dy.t.device, (srcs, dy), param_grads)
            if wg:
                self._grad_done(ctx, layer.weight)
            if bg:
                self._grad_done(ctx, bias)
        if any(s.needs_grad for s in srcs):
            wt = self.w_t(ctx, layer)
            # one buffer of all members' channels, then each member's slice summed down to its own resolution (writing the full-resolution
            # member's slice straight from the dgrad launch — a split output — measured slower: 1 121 vs 876 + 50 us, round 3; removed)
            c0 = sum(s.Cs for s in srcs)
            d_all = Act(torch.empty((dy.B, H, W, c0), dtype=dy.t.dtype, device=dy.t.device), c0)
            ops.conv_forward(dy, wt, I, R, S, 1, pad, mode=1, out_hw=(H, W), cin=wt.shape[3], out=d_all)
            off = 0
            for s_ in srcs:
                if s_.needs_grad:
                    g = Act(torch.empty_like(s_.t), s_.C)
                    ops.upsample_slice_backward(d_all, g, off)
                    ctx.set_grad(s_, g)
                off += s_.Cs

    # ------------------------------------------------------------------ conv2 by position classes
    def conv2cls_begin(self, ctx, q5, q4, H, W, layer):
        """First half of conv_cat_cls, issued as soon as the x8 / x4 members exist (before the x2 / x1 branches of the head are
        enqueued, so that it runs under them): combined filters, the two class convolutions, the expansion — on the weight-gradient
        side stream.  Returns the state conv_cat_cls consumes, or None when the geometry is not the class formulation's."""
        O, I, R, S, stride, pad = _geom(layer)
        if not (I == 512 and (R, S, stride, pad) == (3, 3, 1, 1) and H % 8 == 0 and W % 8 == 0 and q5.C == 128 and q4.C == 128
                and q5.Cs == 128 and q4.Cs == 128 and (q5.H * 8, q5.W * 8, q4.H * 4, q4.W * 4) == (H, W, H, W) and q5.t.dtype == self.cdt):
            return None
        C = I // 4
        dev = q5.t.device
        ar = self.m._arena
        st = {"need_x": bool(ctx.train and (q5.needs_grad or q4.needs_grad))}

        def classes():
            # (the transposed operands are made whenever a tape is being recorded: whether q3 / q2 need gradients is not known yet)
            st["ops"] = wo = ops.Conv2ClsOperands(ar.data_seg(layer.weight), O, C, self.cdt, ctx.train)
            if not ops.is16(self.cdt):
                # per TAP: nine 1x1 convolutions + a low-resolution class sum — 9x fewer FLOPs than the frame filters, which matters where
                # the matrix pipe is the limit (f32); in 16 bits the extra f32 pass over the maps costs more than the frames waste
                # (headline step 36.57 -> 36.78 ms, profiles/r06_cfg2_conv2_classes_ab.txt)
                t8, _ = ops.conv_forward(q5, wo.wtap[0], 9 * O, 1, 1, 1, 0, out_f32=True)
                t4, _ = ops.conv_forward(q4, wo.wtap[1], 9 * O, 1, 1, 1, 0, out_f32=True)
                m8, m4 = ops.conv2cls_classsum(t8), ops.conv2cls_classsum(t4)
                st["keep"] = (t8, t4, m8, m4)
            else:
                m8, _ = ops.conv_forward(q5, wo.wc[0], 9 * O, 3, 3, 1, 1, out_f32=True)
                m4, _ = ops.conv_forward(q4, wo.wc[1], 9 * O, 3, 3, 1, 1, out_f32=True)
                st["keep"] = (m8, m4)
            st["e"] = ops.conv2cls_expand(m8, m4, q5.B, H, W, O, self.cdt)
        # The weight-gradient side stream carries it (behind the detection pyramid, forked earlier): a stream of its own measured
        # 0.2 ms SLOWER (profiles/r06_conv2_classes_ab.txt) — a third stream only adds contention.
        side = self.side_stream(dev)
        if side is not None:
            self._on_side(ctx, dev, (q5, q4, st), classes)
            self.flush_side(ctx, dev)
        else:
            classes()
        if side is not None:
            st["ev"] = torch.cuda.Event()
            gpu_op(st["ev"].record, side)
            ctx.side_keep.append(st)
        return st

    def conv_cat_cls(self, ctx, srcs, H, W, layer, act, st):
        """relu(layer(cat(up8(q5), up4(q4), up2(q3), q2))) with the x8 / x4 members as position-class maps (csrc/conv2cls.hip; ops.conv2cls_*):
        the 3x3 convolution over the virtual concatenation of (q3, q2) with the expanded class maps of conv2cls_begin as its residual.
        Backward mirrors it (_conv_cat_cls_bwd)."""
        q5, q4, q3, q2 = srcs
        O, I, R, S, stride, pad = _geom(layer)
        bias = layer.bias
        dev = q2.t.device
        need_x = ctx.train and any(s_.needs_grad for s_ in srcs)
        if "ev" in st:
            gpu_op(torch.cuda.current_stream(dev).wait_event, st["ev"])
        wo = st["ops"]
        cat2 = None
        if ops.is16(self.cdt):
            y = ops.conv_forward_cat([q3, q2], H, W, wo.wm, O, bias=bias.data if bias is not None else None, act=3 if act == 1 else act, res=st["e"])
        else:       # f32 has no virtual concatenation: the two remaining members are materialised (half of what concat_up writes)
            C = I // 4
            cat2 = Act(torch.empty((q2.B, H, W, 2 * C), dtype=q2.t.dtype, device=dev), 2 * C)
            ops.upsample_slice(q3, cat2, 0)
            ops.upsample_slice(q2, cat2, C)
            y, _ = ops.conv_forward(cat2, wo.wm, O, 3, 3, 1, 1, bias=bias.data if bias is not None else None, act=3 if act == 1 else act,
                                    res=st["e"], res_mode=1)
        y.relu_out = act == 1
        if ctx.train:
            y.needs_grad = bool(need_x or layer.weight.requires_grad or (bias is not None and bias.requires_grad))
            if y.needs_grad:
                self._note_use(ctx, layer.weight)
                self._note_use(ctx, bias)
                ctx.tape.append(lambda: self._conv_cat_cls_bwd(ctx, srcs, H, W, layer, y, act, wo, cat2))
        ctx.side_keep.append((st, y))
        return y

    def _conv_cat_cls_bwd(self, ctx, srcs, H, W, layer, y, act, wo, cat2=None):
        q5, q4, q3, q2 = srcs
        dy = ctx.pop_grad(y)
        O, I, R, S, stride, pad = _geom(layer)
        C = I // 4
        bias = layer.bias
        wg = layer.weight.requires_grad
        bg = bias is not None and bias.requires_grad
        if dy is None:
            if wg:
                self._grad_done(ctx, layer.weight)
            if bg:
                self._grad_done(ctx, bias)
            return
        if act == 1:
            dy = ops.relu_backward(dy, y)
        ar = self.m._arena
        dev = dy.t.device
        need_x = any(s_.needs_grad for s_ in srcs)
        g5 = g4 = None
        if need_x:
            g5 = Act(torch.empty_like(q5.t), q5.C)
            g4 = Act(torch.empty_like(q4.t), q4.C)

        def class_side():
            # everything of the x8 / x4 members: class pooling of dy and the per-tap sums, their input gradients (1x1 over nine taps;
            # consumed by the main stream a dozen launches later: ctx.cls_ev), the per-tap filter gradients, the main part's weight
            # gradient and the fold into dW
            p8, p4 = ops.conv2cls_pool(dy)
            g8, g4t = ops.conv2cls_tapsum(p8), ops.conv2cls_tapsum(p4)
            if need_x:
                ops.conv_forward(g8, wo.wtap_t[0], C, 1, 1, 1, 0, mode=1, out_hw=(q5.H, q5.W), cin=9 * O, out=g5)
                ops.conv_forward(g4t, wo.wtap_t[1], C, 1, 1, 1, 0, mode=1, out_hw=(q4.H, q4.W), cin=9 * O, out=g4)
                ev = torch.cuda.Event()
                gpu_op(ev.record, ops.stream_obj() if ops.stream_obj() is not None else torch.cuda.current_stream(dev))
                ctx.cls_ev = ev
            if wg or bg:
                n = wo.nm + 2 * wo.nt
                dcomb = torch.empty(n, dtype=torch.float32, device=dev)
                call("mpn_fill_f32", ops.ptr(dcomb), 0.0, n, ops.stream_ptr())
                done = False
                if wg:
                    if cat2 is None:
                        done = ops.conv_wgrad_cat([q3, q2], H, W, dy, dcomb[: wo.nm], O, db=ar.grad_seg(bias) if bg else None)
                    else:
                        done = ops.conv_wgrad(cat2, dy, dcomb[: wo.nm], O, 3, 3, 1, 1, db=ar.grad_seg(bias) if bg else None)
                    ops.conv_wgrad(q5, g8, dcomb[wo.nm: wo.nm + wo.nt], 9 * O, 1, 1, 1, 0)
                    ops.conv_wgrad(q4, g4t, dcomb[wo.nm + wo.nt:], 9 * O, 1, 1, 1, 0)
                    call("mpn_conv2cls_fold", ops.ptr(dcomb), ops.ptr(ar.grad_seg(layer.weight)), O, C, ops.stream_ptr())
                if bg and not done:
                    ops.bias_grad(dy, ar.grad_seg(bias), O)
                ctx.side_keep.append((dcomb,))
            ctx.side_keep.append((p8, p4, g8, g4t))
        self._on_side(ctx, dev, (srcs, dy, g5, g4, wo, cat2), class_side)
        if self.side_stream(dev) is not None:
            self.flush_side(ctx, dev)
        if wg:
            self._grad_done(ctx, layer.weight)
        if bg:
            self._grad_done(ctx, bias)
        if need_x:
            ctx.set_grad(q5, g5)
            ctx.set_grad(q4, g4)
            d_all = Act(torch.empty((dy.B, H, W, 2 * C), dtype=dy.t.dtype, device=dev), 2 * C)
            ops.conv_forward(dy, wo.wm_t, 2 * C, 3, 3, 1, 1, mode=1, out_hw=(H, W), cin=O, out=d_all)
            for s_, off in ((q3, 0), (q2, C)):
                g = Act(torch.empty_like(s_.t), s_.C)
                ops.upsample_slice_backward(d_all, g, off)
                ctx.set_grad(s_, g)

    def wait_class_grads(self, ctx, device):
        """Tape marker in front of the consumers of the class input gradients (keypoint_head): they were produced on the side stream."""
        ev = getattr(ctx, "cls_ev", None)
        if ev is not None:
            gpu_op(torch.cuda.current_stream(device).wait_event, ev)
            ctx.cls_ev = None

    def export(self, ctx, src, C, Ho, Wo, slot):
        """Internal padded tensor -> exact f32 API tensor (nearest up-sampled to Ho x Wo)."""
        out = ops.export_f32(src, C, Ho, Wo)
        if ctx.train and src.needs_grad:
            def bwd():
                g = ctx.out_grads.get(slot)
                if g is None:
                    return
                ctx.set_grad(src, ops.import_grad(g, src, self.cdt))
            ctx.tape.append(bwd)
        return out

    def export_internal(self, ctx, src, slot):
        """The recorded step's form of export(): no API tensor.  Its loss kernel reads ``src`` itself and puts an internal gradient
        activation under ``slot`` (losses.mse_train_raw)."""
        if ctx.train and src.needs_grad:
            def bwd():
                g = ctx.out_grads.get(slot)
                if g is not None:
                    ctx.set_grad(src, g)
            ctx.tape.append(bwd)
        return src

    # ------------------------------------------------------------------ network pieces
    def stem(self, ctx, img):
        """conv1 7x7/s2 + bn1 + relu + maxpool (fpn.py:99-100) via the NHWC4 packed row-conv."""
        f = self.m.fpn
        B, _, H, W = img.shape
        Hp, Wp = H + 6, W + 8
        img = img.detach()
        if img.dtype != torch.float32:
            img = img.float()
        packed = torch.empty((B, Hp, Wp, 4), dtype=self.cdt, device=img.device)
        call("mpn_stem_pack_image", ops.ptr(img), img.stride(0), img.stride(1), img.stride(2), img.stride(3), ops.ptr(packed),
             B, H, W, ops.dtype_code(self.cdt), ops.stream_ptr())
        xa = Act(packed, 4)
        w = f.conv1.weight
        wp = torch.empty((64, 7, 32), dtype=self.cdt, device=img.device)
        call("mpn_stem_pack_weight", ops.ptr(self.m._arena.data_seg(w)), ops.ptr(wp), 64, ops.dtype_code(self.cdt), ops.stream_ptr())
        Ho, Wo = ops.conv_out_hw(H, W, 7, 7, 2, 3)
        geom = (Hp, Wp, Hp * Wp * 4, Wp * 4, 4)
        bn_train = f.bn1.training
        if self.fold_bn and not ctx.train and not bn_train:        # frozen statistics, no tape: bn1 + ReLU in the stem's epilogue
            bst = ops.bn_finalize_eval(f.bn1.weight.data, f.bn1.bias.data, f.bn1.running_mean, f.bn1.running_var, f.bn1.eps)
            z, _ = ops.conv_forward(xa, wp, 64, 7, 1, 2, 0, cin=32, x_geom=geom, out_hw=(Ho, Wo), scale=bst.scale, bias=bst.shift, act=1)
            return self.maxpool(ctx, z)
        y, st = ops.conv_forward(xa, wp, 64, 7, 1, 2, 0, cin=32, x_geom=geom, out_hw=(Ho, Wo), want_stats=bn_train,
                                 bn_fin=self._bn_fin(f.bn1) if bn_train else None)
        if ctx.train and w.requires_grad:
            y.needs_grad = True
            self._note_use(ctx, w)

            def bwd():
                dy = ctx.pop_grad(y)
                if dy is not None:
                    dwp = torch.empty((64, 7, 32), dtype=torch.float32, device=img.device)

                    def stem_wgrad():
                        call("mpn_fill_f32", ops.ptr(dwp), 0.0, dwp.numel(), ops.stream_ptr())
                        ops.conv_wgrad(xa, dy, dwp, 64, 7, 1, 2, 0, cin=32, x_geom=geom)
                        call("mpn_stem_unpack_wgrad", ops.ptr(dwp), ops.ptr(self.m._arena.grad_seg(w)), 64, ops.stream_ptr())
                    self._on_side(ctx, dy.t.device, (xa, dy, dwp), stem_wgrad)
                self._grad_done(ctx, w)
            ctx.tape.append(bwd)
        z = self.bn(ctx, y, st, f.bn1, True)
        return self.maxpool(ctx, z)

    def conv_bn(self, ctx, x, conv, bn, relu, res=None):
        """act(bn(conv(x)) [+ res]).  Without a tape and with frozen statistics this is ONE launch: the BatchNorm becomes the
        epilogue's per-channel scale / shift (f32, before any rounding), ReLU and the residual add follow in registers; otherwise
        conv (+ statistics) and the BatchNorm passes."""
        if self.fold_bn and not ctx.train and not bn.training:
            st = ops.bn_finalize_eval(bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, bn.eps)
            O, I, R, S, stride, pad = _geom(conv)
            act = (3 if res is not None else 1) if relu else 0
            y, _ = ops.conv_forward(x, self.w_fwd(conv), O, R, S, stride, pad, scale=st.scale, bias=st.shift, act=act,
                                    res=res, res_mode=1 if res is not None else 0)
            return y
        y, s = self.conv(ctx, x, conv, stats=bn.training, bn=bn)
        return self.bn(ctx, y, s, bn, relu, res=res)

    def bottleneck(self, ctx, x, blk):
        """fpn.py:28-34."""
        z1 = self.conv_bn(ctx, x, blk.conv1, blk.bn1, True)
        z2 = self.conv_bn(ctx, z1, blk.conv2, blk.bn2, True)
        sc = self.conv_bn(ctx, x, blk.downsample[0], blk.downsample[1], False) if len(blk.downsample) > 0 else x
        return self.conv_bn(ctx, z2, blk.conv3, blk.bn3, True, res=sc)

    def backbone(self, ctx, img):
        f = self.m.fpn
        c = self.stem(ctx, img)
        feats = []
        for layer in (f.layer1, f.layer2, f.layer3, f.layer4):
            for blk in layer:
                c = self.bottleneck(ctx, c, blk)
            feats.append(c)
        return feats      # c2..c5

    def det_pyramid(self, ctx, c3, c4, c5):
        """fpn.py:107-114 (p4 is built from the UN-smoothed p5)."""
        f = self.m.fpn

        def coarse(h):                                           # P6 / P7: 3x3 stride 2 on the 15x15 / 8x8 maps
            h["p6"], _ = self.conv(ctx, c5, f.conv6)
            h["r6"] = self.relu(ctx, h["p6"])
            h["p7"], _ = self.conv(ctx, h["r6"], f.conv7)

        def rest(h):
            h["p5"], _ = self.conv(ctx, c5, f.latlayer1)
            h["p4"], _ = self.conv(ctx, c4, f.latlayer2, res=h["p5"], res_mode=2)
            h["p3"], _ = self.conv(ctx, c3, f.latlayer3, res=h["p4"], res_mode=2)
            h["p5s"], _ = self.conv(ctx, h["p5"], f.toplayer0)
            h["p4s"], _ = self.conv(ctx, h["p4"], f.toplayer1)
            h["p3s"], _ = self.conv(ctx, h["p3"], f.toplayer2)
        h = {}
        dev = c5.t.device
        mode = self.det_pyramid_side if self.side_stream(dev) is not None else 0
        if mode:
            # Forward fork: this pyramid feeds only the detection head, which runs after the keypoint head — so it runs on the side
            # stream under the keypoint head's launches and the detection head joins (join_forward_side).  P6 / P7 alone are 32 - 64
            # workgroups walking 576 / 72 k-steps (190 + 31 us on a nearly empty chip): mode 1 moves just those (-0.13 ms/step),
            # mode 2 the whole pyramid (another -0.06; profiles/r03_forward_side_fork_ab.txt).  `h` keeps every tensor of the branch
            # alive until the join; the tape order (hence the backward pass) is the same as without the fork.
            def branch():
                coarse(h)
                if mode == 2:
                    rest(h)
            self._on_side(ctx, dev, (c3, c4, c5, h), branch)
            self.flush_side(ctx, dev)
            ctx.fwd_side_join = True
            if mode != 2:
                rest(h)
        else:
            coarse(h)
            rest(h)
        return [h["p3s"], h["p4s"], h["p5s"], h["p6"], h["p7"]]

    def kp_pyramid(self, ctx, c2, c3, c4, c5):
        """fpn.py:117-124 (fp5 is not smoothed)."""
        f = self.m.fpn
        fp5, _ = self.conv(ctx, c5, f.toplayer)
        fp4, _ = self.conv(ctx, c4, f.flatlayer1, res=fp5, res_mode=2)
        fp3, _ = self.conv(ctx, c3, f.flatlayer2, res=fp4, res_mode=2)
        fp2, _ = self.conv(ctx, c2, f.flatlayer3, res=fp3, res_mode=2)
        fp4s, _ = self.conv(ctx, fp4, f.smooth1)
        fp3s, _ = self.conv(ctx, fp3, f.smooth2)
        fp2s, _ = self.conv(ctx, fp2, f.smooth3)
        return [fp2s, fp3s, fp4s, fp5]

    def keypoint_head(self, ctx, feats, intermediate, internal=False):
        """posenet.py:288-318 / :243-257.  Returns (pred, [k2,k3,k4,k5]) as f32 API tensors, or with ``internal`` as the
        network's own activations at their native resolutions (the recorded step's loss kernel up-samples by indexing)."""
        m = self.m
        p2, p3, p4, p5 = feats
        Ho, Wo = p2.H, p2.W
        saved = []
        if intermediate:
            for i, (src, layer) in enumerate(((p2, m.convfin_k2), (p3, m.convfin_k3), (p4, m.convfin_k4), (p5, m.convfin_k5))):
                k, _ = self.conv(ctx, src, layer, out_f32=True)
                saved.append(self.export_internal(ctx, k, "k%d" % i) if internal else self.export(ctx, k, 19, Ho, Wo, "k%d" % i))
        q5, _ = self.conv(ctx, self.conv(ctx, p5, m.convt1)[0], m.convs1)
        q4, _ = self.conv(ctx, self.conv(ctx, p4, m.convt2)[0], m.convs2)
        cls = self.conv2cls_begin(ctx, q5, q4, Ho, Wo, m.conv2) if (self.conv2_classes and (self.virtual_concat or not ops.is16(self.cdt))) else None
        if ctx.train:
            # backward: the gradients of q5 / q4 may come from the side stream (conv2 by position classes); this marker runs right
            # before the first launch that reads them (the tape is walked backwards)
            ctx.tape.append(lambda: self.wait_class_grads(ctx, p2.t.device))
        q3, _ = self.conv(ctx, self.conv(ctx, p3, m.convt3)[0], m.convs3)
        q2, _ = self.conv(ctx, self.conv(ctx, p2, m.convt4)[0], m.convs4)
        if cls is not None and ops.conv2cls_supported([q5, q4, q3, q2], Ho, Wo):
            h = self.conv_cat_cls(ctx, [q5, q4, q3, q2], Ho, Wo, m.conv2, 1, cls)
        elif self.virtual_concat and ops.cat_supported([q5, q4, q3, q2], Ho, Wo):
            h = self.conv_cat(ctx, [q5, q4, q3, q2], Ho, Wo, m.conv2, act=1)
        else:
            cat = self.concat_up(ctx, [q5, q4, q3, q2], Ho, Wo)
            h, _ = self.conv(ctx, cat, m.conv2, act=1)
        pr, _ = self.conv(ctx, h, m.convfin, out_f32=True)
        pred = self.export_internal(ctx, pr, "pred") if internal else self.export(ctx, pr, 18, Ho, Wo, "pred")
        return pred, saved

    def join_forward_side(self, ctx, device):
        """The main stream waits for forward work that was forked to the side stream (det_pyramid's P6 / P7 branch)."""
        if ctx.fwd_side_join:
            gpu_op(torch.cuda.current_stream(device).wait_stream, self.side_stream(device))
            ctx.fwd_side_join = False

    def detection_head(self, ctx, feats):
        """posenet.py:327-328: shared towers over p3..p7, outputs written straight into [B,A,4] / [B,A,1]."""
        self.join_forward_side(ctx, feats[0].t.device)
        m = self.m
        B = feats[0].B
        dev = feats[0].t.device
        cells = [f.H * f.W for f in feats]
        A = 9 * sum(cells)
        reg_all = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
        cls_all = torch.empty((B, A, 1), dtype=torch.float32, device=dev)
        outs = []
        off = 0
        rm, cm = m.regressionModel, m.classificationModel
        if self.pyramid_towers and len(feats) > 1:
            r = c = feats
            for layer in (rm.conv1, rm.conv2, rm.conv3, rm.conv4):
                r = self.conv_seg(ctx, r, layer, act=1)
            ros = self.conv_seg(ctx, r, rm.output, out_f32=True)
            for layer in (cm.conv1, cm.conv2, cm.conv3, cm.conv4):
                c = self.conv_seg(ctx, c, layer, act=1)
            cos = self.conv_seg(ctx, c, cm.output, out_f32=True)
            for ro, co, n in zip(ros, cos, cells):
                outs.append((ro, co, off, n))
                off += n * 9
        else:
            for f, n in zip(feats, cells):
                r = f
                for layer in (rm.conv1, rm.conv2, rm.conv3, rm.conv4):
                    r, _ = self.conv(ctx, r, layer, act=1)
                ro, _ = self.conv(ctx, r, rm.output, out_f32=True)
                c = f
                for layer in (cm.conv1, cm.conv2, cm.conv3, cm.conv4):
                    c, _ = self.conv(ctx, c, layer, act=1)
                co, _ = self.conv(ctx, c, cm.output, out_f32=True)
                outs.append((ro, co, off, n))
                off += n * 9
        # pack the per-level outputs, then one sigmoid over [B,A,1] (its backward needs only p)
        for ro, co, o, n in outs:
            call("mpn_det_pack", ops.ptr(ro.t), 0, ctypes.c_void_p(reg_all.data_ptr() + o * 4 * 4), B, n, ro.Cs, 36, A * 4, ops.stream_ptr())
            call("mpn_det_pack", ops.ptr(co.t), 0, ctypes.c_void_p(cls_all.data_ptr() + o * 4), B, n, co.Cs, 9, A, ops.stream_ptr())
        call("mpn_sigmoid_forward", ops.ptr(cls_all), ops.ptr(cls_all), cls_all.numel(), ops.stream_ptr())
        if ctx.train:
            if any(x[0].needs_grad or x[1].needs_grad for x in outs):
                def bwd():
                    gr, gc = ctx.out_grads.get("reg"), ctx.out_grads.get("cls")
                    if gc is not None:
                        gc = gc.contiguous()
                        dlogit = torch.empty_like(cls_all)
                        call("mpn_sigmoid_backward", ops.ptr(gc), ops.ptr(cls_all), ops.ptr(dlogit), cls_all.numel(), ops.stream_ptr())
                    if gr is not None:
                        gr = gr.contiguous()
                    # output gradients of all levels in one buffer per tower (the pyramid backward consumes them in one launch)
                    dros = ops.alloc_seg([x[0] for x in outs], 36, self.cdt)
                    dcos = ops.alloc_seg([x[1] for x in outs], 9, self.cdt)
                    for (ro, co, o, n), dr, dc in zip(outs, dros, dcos):
                        if gr is not None and ro.needs_grad:
                            call("mpn_det_unpack", ctypes.c_void_p(gr.data_ptr() + o * 4 * 4), ops.ptr(dr.t), ops.dtype_code(self.cdt),
                                 B, n, dr.Cs, 36, A * 4, ops.stream_ptr())
                            ctx.set_grad(ro, dr)
                        if gc is not None and co.needs_grad:
                            call("mpn_det_unpack", ctypes.c_void_p(dlogit.data_ptr() + o * 4), ops.ptr(dc.t), ops.dtype_code(self.cdt),
                                 B, n, dc.Cs, 9, A, ops.stream_ptr())
                            ctx.set_grad(co, dc)
                ctx.tape.append(bwd)
        return cls_all, reg_all

    # ------------------------------------------------------------------ backward
    def run_backward(self, ctx, out_grads, schedule=None, on_bucket=None):
        """``schedule``: bucket schedule to notify as parameter gradients complete (default: the model's data-parallel reducer);
        ``on_bucket``: the optimizer's slice update, run behind each bucket (ddp.GradReducer) — only a step that owns its optimizer
        passes one (replay.py); ``loss.backward()`` leaves the update to ``optimizer.step()``."""
        m = self.m
        ctx.out_grads = out_grads
        ops.check_device(m._arena.flat)
        m._arena.ensure_grads()
        dev = m._arena.flat.device
        side = self.side_stream(dev)
        sched = ctx.schedule = schedule if schedule is not None else m._reducer
        if sched is not None:
            sched.launch_stream = side
            sched.pre_launch = (lambda: self.flush_side(ctx, dev)) if side is not None else None
            gpu_op(sched.begin, on_bucket)
        if ctx.wt_ready is not None:
            gpu_op(torch.cuda.current_stream(dev).wait_event, ctx.wt_ready)
            ctx.wt_ready = None
        tape = ctx.tape
        while tape:
            tape.pop()()
        if side is not None:
            self.flush_side(ctx, dev)
            gpu_op(torch.cuda.current_stream(dev).wait_stream, side)      # join: parameter gradients are complete
        ctx.grads.clear()
        ctx.lazy_res.clear()
        ctx.keep = []
        ctx.side_keep = []
        ctx.wt.clear()
        ctx.wt_buf = None
        if sched is not None:
            sched.pre_launch = None
            gpu_op(sched.finish)
        ctx.schedule = None
