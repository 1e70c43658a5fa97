"""poseNet — keypoint subnet + detection subnet (RetinaNet) + PRN, drop-in for the reference's
``network/posenet.py`` (class poseNet :154-364, build_*_loss :367-445), running on hand-written HIP
kernels for MI355X.

Boundary kept identical to the reference:
  * ``poseNet(layers in {50,101}, prn_node_count=1024, prn_coeff=2)``; same child-module names and
    registration order, hence the same ``state_dict()`` keys / shapes (402 for R50, 708 for R101);
  * ``model([img_batch f32 NCHW, subnet_name])`` with the four return structures of posenet.py:226-285;
  * ``poseNet.build_loss(saved_for_loss, subnet_name, *gts)`` -> (0-dim loss with grad, OrderedDict);
  * ``freeze_bn()``, ``.train()/.eval()``, ``requires_grad=False`` on frozen groups honoured.
One addition: ``subnet_name='train_both'`` runs ONE shared backbone pass with both
heads for the combined keypoint+detection training step the benchmark is quoted on.

Internals differ completely: parameters are views into one flat f32 arena (conv weights stored
[Cout][R][S][Cin]), activations are NHWC in ``compute_dtype`` (bf16 default, f32 for parity), and the
forward/backward is a tape of fused kernel launches (engine.py) entered by autograd through a single
node.  If ``libmpn_hip.so`` is missing every forward raises ``MpnError`` — there is no fallback.
"""
import math
import weakref
from collections import OrderedDict

import torch
import torch.nn as nn
from torch.nn import init

from .. import ops
from ..arena import ParamArena
from ..engine import Ctx, Engine
from .._lib import MpnError, call, gpu_op
from ..lib.nms.pth_nms import pth_nms
from . import losses
from .anchors import Anchors
from .fpn import FPN50, FPN101
from .losses import build_detection_loss, build_keypoint_loss, build_names, _log_values  # noqa: F401  (reference exports)
from .utils import BBoxTransform, ClipBoxes, decode_and_clip


def nms(dets, thresh):
    """posenet.py:19-22."""
    return pth_nms(dets, thresh)


def _conv(cin, cout, k, padding=0):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding=padding, device="meta")


class Concat(nn.Module):
    """posenet.py:25-30 (parameter-free; the engine writes channel slices instead of torch.cat)."""

    def forward(self, up1, up2, up3, up4):
        raise MpnError("Concat is fused into the keypoint head (engine.concat_up)")


class RegressionModel(nn.Module):
    """posenet.py:33-69: 4 x (3x3 conv + ReLU) + 3x3 conv -> 9*4."""

    def __init__(self, num_features_in, num_anchors=9, feature_size=256):
        super(RegressionModel, self).__init__()
        self.conv1 = _conv(num_features_in, feature_size, 3, 1)
        self.act1 = nn.ReLU()
        self.conv2 = _conv(feature_size, feature_size, 3, 1)
        self.act2 = nn.ReLU()
        self.conv3 = _conv(feature_size, feature_size, 3, 1)
        self.act3 = nn.ReLU()
        self.conv4 = _conv(feature_size, feature_size, 3, 1)
        self.act4 = nn.ReLU()
        self.output = _conv(feature_size, num_anchors * 4, 3, 1)


class ClassificationModel(nn.Module):
    """posenet.py:72-117: 4 x (3x3 conv + ReLU) + 3x3 conv -> 9*num_classes + sigmoid."""

    def __init__(self, num_features_in, num_anchors=9, num_classes=80, prior=0.01, feature_size=256):
        super(ClassificationModel, self).__init__()
        self.num_classes = num_classes
        self.num_anchors = num_anchors
        self.conv1 = _conv(num_features_in, feature_size, 3, 1)
        self.act1 = nn.ReLU()
        self.conv2 = _conv(feature_size, feature_size, 3, 1)
        self.act2 = nn.ReLU()
        self.conv3 = _conv(feature_size, feature_size, 3, 1)
        self.act3 = nn.ReLU()
        self.conv4 = _conv(feature_size, feature_size, 3, 1)
        self.act4 = nn.ReLU()
        self.output = _conv(feature_size, num_anchors * num_classes, 3, 1)
        self.output_act = nn.Sigmoid()


class Flatten(nn.Module):
    def forward(self, input):
        return input.view(input.size(0), -1)


class Add(nn.Module):
    def forward(self, input1, input2):
        raise MpnError("Add is fused into the PRN softmax kernel")


class PRN(nn.Module):
    """posenet.py:130-152."""

    def __init__(self, node_count, coeff):
        super(PRN, self).__init__()
        self.flatten = Flatten()
        self.height = coeff * 28
        self.width = coeff * 18
        n = self.height * self.width * 17
        self.dens1 = nn.Linear(n, node_count, device="meta")
        self.bneck = nn.Linear(node_count, node_count, device="meta")
        self.dens2 = nn.Linear(node_count, n, device="meta")
        self.drop = nn.Dropout()
        self.add = Add()
        self.softmax = nn.Softmax(dim=1)


class _NetFn(torch.autograd.Function):
    """The single autograd node through which ``loss.backward()`` enters the engine."""

    @staticmethod
    def forward(fctx, anchor, holder, *raw):
        fctx.holder = holder
        return tuple(t.view_as(t) for t in raw)

    @staticmethod
    def backward(fctx, *gouts):
        model, ectx, slots = fctx.holder
        fctx.holder = None
        grads = {}
        for s, g in zip(slots, gouts):
            if g is not None:
                grads[s] = g
        model._engine.run_backward(ectx, grads)
        return (None, None) + (None,) * len(gouts)


class poseNet(nn.Module):
    def __init__(self, layers, prn_node_count=1024, prn_coeff=2, compute_dtype=torch.bfloat16, device=None):
        super(poseNet, self).__init__()
        if layers == 101:
            self.fpn = FPN101()
        elif layers == 50:
            self.fpn = FPN50()
        else:
            raise ValueError("layers must be 50 or 101")
        # keypoint subnet (posenet.py:161-186)
        self.convfin_k2 = _conv(256, 19, 1)
        self.convfin_k3 = _conv(256, 19, 1)
        self.convfin_k4 = _conv(256, 19, 1)
        self.convfin_k5 = _conv(256, 19, 1)
        self.convt1 = _conv(256, 128, 3, 1)
        self.convt2 = _conv(256, 128, 3, 1)
        self.convt3 = _conv(256, 128, 3, 1)
        self.convt4 = _conv(256, 128, 3, 1)
        self.convs1 = _conv(128, 128, 3, 1)
        self.convs2 = _conv(128, 128, 3, 1)
        self.convs3 = _conv(128, 128, 3, 1)
        self.convs4 = _conv(128, 128, 3, 1)
        self.upsample1 = nn.Upsample(scale_factor=8, mode='nearest', align_corners=None)
        self.upsample2 = nn.Upsample(scale_factor=4, mode='nearest', align_corners=None)
        self.upsample3 = nn.Upsample(scale_factor=2, mode='nearest', align_corners=None)
        self.concat = Concat()
        self.conv2 = _conv(512, 256, 3, 1)
        self.convfin = _conv(256, 18, 1)
        # detection subnet (posenet.py:188-194)
        self.regressionModel = RegressionModel(256)
        self.classificationModel = ClassificationModel(256, num_classes=1)
        self.anchors = Anchors()
        self.regressBoxes = BBoxTransform()
        self.clipBoxes = ClipBoxes()
        self.focalLoss = losses.FocalLoss()
        # prn subnet (posenet.py:198)
        self.prn = PRN(prn_node_count, prn_coeff)

        self.layers = layers
        self.compute_dtype = compute_dtype
        self._arena = None
        self._reducer = None
        self._engine = Engine(self)
        self._anchor = torch.zeros(1, requires_grad=True)
        self.fpn._owner = weakref.ref(self)
        dev = torch.device(device) if device is not None else torch.device("cpu")
        self._materialize(dev)
        self.freeze_bn()      # from retinanet (posenet.py:211); undone by any later .train()

    # ------------------------------------------------------------------ parameters
    def _materialize(self, device):
        """Allocate real storage for the meta-constructed parameters/buffers and initialise them as the
        reference does (posenet.py:203-218), then move everything into the flat arena."""
        for mod in self.modules():
            for name, p in list(mod._parameters.items()):
                if p is not None and p.is_meta:
                    mod._parameters[name] = nn.Parameter(torch.empty(p.shape, dtype=torch.float32, device=device),
                                                         requires_grad=p.requires_grad)
            for name, b in list(mod._buffers.items()):
                if b is not None and b.is_meta:
                    mod._buffers[name] = torch.zeros(b.shape, dtype=b.dtype, device=device)
        for mod in self.modules():
            if isinstance(mod, nn.BatchNorm2d):
                init.ones_(mod.weight); init.zeros_(mod.bias)
                mod.running_mean.zero_(); mod.running_var.fill_(1); mod.num_batches_tracked.zero_()
            elif isinstance(mod, nn.Linear):
                mod.reset_parameters()
        self._initialize_weights_norm()
        prior = 0.01
        self.classificationModel.output.weight.data.fill_(0)
        self.classificationModel.output.bias.data.fill_(-math.log((1.0 - prior) / prior))
        self.regressionModel.output.weight.data.fill_(0)
        self.regressionModel.output.bias.data.fill_(0)
        self._build_arena(device)

    def _initialize_weights_norm(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                init.normal_(m.weight, std=0.01)
                if m.bias is not None:
                    init.constant_(m.bias, 0.0)

    def _build_arena(self, device):
        named = [(n, p) for n, p in self.named_parameters()]
        self._arena = ParamArena(named, device)
        # all num_batches_tracked counters share one int64 tensor: one add per training forward
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        nbt = torch.zeros(len(bns), dtype=torch.int64, device=device)
        for i, m in enumerate(bns):
            nbt[i] = int(m.num_batches_tracked.item()) if m.num_batches_tracked is not None else 0
            m._buffers["num_batches_tracked"] = nbt[i]
        self._nbt = nbt
        self._bns = bns

    def _apply(self, fn, *args, **kwargs):
        out = super(poseNet, self)._apply(fn, *args, **kwargs)
        p0 = next(self.parameters())
        if p0.dtype != torch.float32:
            raise MpnError("master parameters stay f32; choose the arithmetic type with compute_dtype")
        if self._arena is None or not self._arena.consistent():
            self._build_arena(p0.device)
        return out

    def freeze_bn(self):
        '''Freeze BatchNorm layers (posenet.py:220-224).'''
        for layer in self.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()

    # ------------------------------------------------------------------ forward
    def _prn_split(self):
        """Arena offset where the PRN's parameters start when they form the tail of the arena (they are registered last,
        posenet.py:209), else None.  The PRN holds 71 of the 132 M parameters and is not part of the backbone / head passes."""
        ar = self._arena
        cached = getattr(ar, "_prn_split", False)
        if cached is not False:
            return cached
        ids = [ar.index[id(q)] for q in self.prn.parameters() if id(q) in ar.index]
        split = None
        if ids and sorted(ids) == list(range(min(ids), len(ar.params))):
            split = ar.offsets[min(ids)]
        ar._prn_split = split
        return split

    def _prepare(self, img, prn=False):
        if not img.is_cuda:
            raise MpnError("poseNet runs on the MI355X only (input is on %s); there is no CPU path" % img.device)
        ops.check_device(img)
        if self._arena is None or not self._arena.consistent() or self._arena.device != img.device:
            if next(self.parameters()).device != img.device:
                raise MpnError("model parameters are on %s but the input is on %s" % (next(self.parameters()).device, img.device))
            self._build_arena(img.device)
        ar = self._arena
        cdt = self.compute_dtype
        if ops.is16(cdt):
            buf = ar.lowp.get(cdt)
            if buf is None:
                buf = ar.lowp[cdt] = torch.empty(ar.total, dtype=cdt, device=ar.device)
            split = self._prn_split()
            if split is None:
                ops.cast_lowp(ar.flat, buf)          # one launch refreshes every forward operand
            elif prn:
                ops.cast_lowp(ar.flat[split:], buf[split:])
            elif split > 0:
                ops.cast_lowp(ar.flat[:split], buf[:split])
        elif cdt != torch.float32:
            raise MpnError("compute_dtype must be torch.bfloat16, torch.float16 or torch.float32")

    def _want_tape(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self._arena.params)

    def _wrap(self, ctx, slots, raw):
        """Make the raw output tensors differentiable through the engine tape."""
        if not ctx.train or not ctx.tape:
            return list(raw)
        if self._anchor.device != raw[0].device:
            self._anchor = torch.zeros(1, requires_grad=True, device=raw[0].device)
        self._arena.ensure_grads()
        return list(_NetFn.apply(self._anchor, (self, ctx, slots), *raw))

    def _finish_forward(self, ctx):
        self._engine.join_forward_side(ctx, self._arena.flat.device)      # forward work forked to the side stream (Engine.det_pyramid)
        if ctx.bn_train_ran:
            if all(m.training for m in self._bns) and all(m.num_batches_tracked.data_ptr() == self._nbt[i].data_ptr()
                                                          for i, m in enumerate(self._bns)):
                gpu_op(self._nbt.add_, 1)
            else:
                for m in self._bns:
                    if m.training:
                        m.num_batches_tracked += 1

    def forward(self, x):
        img_batch, subnet_name = x
        if subnet_name == 'keypoint_subnet':
            return self.keypoint_forward(img_batch)
        elif subnet_name == 'detection_subnet':
            return self.detection_forward(img_batch)
        elif subnet_name == 'prn_subnet':
            return self.prn_forward(img_batch)
        elif subnet_name == 'train_both':
            return self.train_both_forward(img_batch)
        # entire net (posenet.py:236-285): inference; detections for image 0 only, exactly like the reference
        predict_keypoint, dets = self._entire_net(img_batch, all_images=False)
        return predict_keypoint, dets[0]

    def forward_all_images(self, img_batch):
        """Entire-net inference with detections for EVERY image of the batch (the reference thresholds and
        NMSes image 0 only, posenet.py:271,281).  Returns (heat-maps [B,18,H/4,W/4], [per-image
        [nms_scores, nms_class, boxes]]); entry b equals what the reference returns for image b run alone."""
        return self._entire_net(img_batch, all_images=True)

    def forward_all_images_padded(self, img_batch, pre_nms_top_n=None):
        """forward_all_images for batched post-processing: (heat-maps [B,18,H/4,W/4], boxes [B,nmax,4], scores [B,nmax], kept) with
        image b's detections in rows [:kept[b]] (descending score; single class) — no per-image tensors or Python lists.
        pre_nms_top_n: optional cap on the candidates that enter the suppression (ops.detect_batched; not in the reference)."""
        predict_keypoint, transformed_anchors, classification, _keep = self.forward_padded_begin(img_batch)
        boxes, scores, kept = self.detect_padded(transformed_anchors, classification, pre_nms_top_n)
        return predict_keypoint, boxes, scores, kept

    def forward_padded_begin(self, img_batch):
        """First half of forward_all_images_padded: the network and the box decode are ENQUEUED, nothing is read back — the host returns
        while the GPU still runs, so a serving loop can post-process the previous batch meanwhile (evaluate/tester.py:
        infer_images_batched).  Returns (heat-maps, decoded boxes [B,A,4], scores [B,A,1], keep-alive): hold on to the last item until
        this batch's results have been read (it owns tensors that side-stream launches of this forward still use)."""
        self._prepare(img_batch)
        eng = self._engine
        ctx = Ctx(False)
        c2, c3, c4, c5 = eng.backbone(ctx, img_batch)
        kp = eng.kp_pyramid(ctx, c2, c3, c4, c5)
        det = eng.det_pyramid(ctx, c3, c4, c5)
        predict_keypoint, _ = eng.keypoint_head(ctx, kp, False)
        classification, regression = eng.detection_head(ctx, det)
        self._finish_forward(ctx)
        if classification.shape[-1] != 1:
            # the padded batch path has no class column: with several classes the arg-max class (posenet.py:283) would have to travel
            # with every detection — use forward_all_images, which carries it
            raise MpnError("forward_all_images_padded serves the single-class detector (classificationModel num_classes = 1), "
                                "got %d classes" % classification.shape[-1])
        transformed_anchors = decode_and_clip(self.anchors(img_batch), regression, img_batch)
        return predict_keypoint, transformed_anchors, classification, (ctx, regression)

    @staticmethod
    def detect_padded(transformed_anchors, classification, pre_nms_top_n=None):
        """Second half: score filter + per-image NMS + gather on the CURRENT stream (two host reads)."""
        return ops.detect_batched(transformed_anchors, classification.reshape(classification.shape[0], -1), 0.05, 0.5, padded=True,
                                  pre_nms_top_n=pre_nms_top_n)

    def _entire_net(self, img_batch, all_images):
        self._prepare(img_batch)
        eng = self._engine
        ctx = Ctx(False)
        c2, c3, c4, c5 = eng.backbone(ctx, img_batch)
        kp = eng.kp_pyramid(ctx, c2, c3, c4, c5)
        det = eng.det_pyramid(ctx, c3, c4, c5)
        predict_keypoint, _ = eng.keypoint_head(ctx, kp, False)
        classification, regression = eng.detection_head(ctx, det)
        self._finish_forward(ctx)
        anchors = self.anchors(img_batch)
        transformed_anchors = decode_and_clip(anchors, regression, img_batch)
        results = []
        if all_images:
            # every image thresholded (posenet.py:269-271), NMS'd (:281) and gathered (:283-285) by batch-wide launches;
            # sizes stay on the device between the stages (ops.detect_batched: two host reads per BATCH)
            cls2 = classification.reshape(classification.shape[0], -1)
            dets = ops.detect_batched(transformed_anchors, cls2, 0.05, 0.5)
            # single class: the arg-max class of every detection (posenet.py:283) is 0 — ONE zero vector for the batch, a view per image
            most = max([b.shape[0] for b, _ in dets if b is not None], default=0)
            zeros = torch.zeros(most, dtype=torch.int64, device=img_batch.device) if most else None
            for boxes, nms_scores in dets:
                if boxes is None:
                    results.append([torch.zeros(0), torch.zeros(0), torch.zeros(0, 4)])      # posenet.py:273-275 (CPU empties)
                else:
                    results.append([nms_scores, zeros[: nms_scores.shape[0]], boxes])
            return predict_keypoint, results
        for b in range(1):
            # posenet.py:269-275: score > 0.05, early-out with the CPU empty triple
            dets, _src = ops.score_filter(transformed_anchors[b], classification[b, :, 0], 0.05)
            if dets.shape[0] == 0:
                results.append([torch.zeros(0), torch.zeros(0), torch.zeros(0, 4)])
                continue
            anchors_nms_idx = nms(dets, 0.5)
            boxes, nms_scores = ops.gather_dets(dets, anchors_nms_idx)
            nms_class = torch.zeros(nms_scores.shape[0], dtype=torch.int64, device=nms_scores.device)   # single class
            results.append([nms_scores, nms_class, boxes])
        return predict_keypoint, results

    def keypoint_forward(self, img_batch):
        """posenet.py:288-318 -> (pred [B,18,H/4,W/4], [k2,k3,k4,k5 ([B,19,...]), pred])."""
        self._prepare(img_batch)
        eng = self._engine
        ctx = Ctx(self._want_tape())
        c2, c3, c4, c5 = eng.backbone(ctx, img_batch)
        kp = eng.kp_pyramid(ctx, c2, c3, c4, c5)       # the unused detection pyramid (fpn.py:107-114) is skipped
        pred, saved = eng.keypoint_head(ctx, kp, True)
        self._finish_forward(ctx)
        outs = self._wrap(ctx, ["k0", "k1", "k2", "k3", "pred"], saved + [pred])
        return outs[4], outs

    def detection_forward(self, img_batch):
        """posenet.py:320-335 -> ([], [classification [B,A,1], regression [B,A,4], anchors [1,A,4]])."""
        self._prepare(img_batch)
        eng = self._engine
        ctx = Ctx(self._want_tape())
        c2, c3, c4, c5 = eng.backbone(ctx, img_batch)
        det = eng.det_pyramid(ctx, c3, c4, c5)
        cls, reg = eng.detection_head(ctx, det)
        self._finish_forward(ctx)
        cls, reg = self._wrap(ctx, ["cls", "reg"], [cls, reg])
        return [], [cls, reg, self.anchors(img_batch)]

    def train_both_forward(self, img_batch):
        """SURVEY.md 8d combined step: shared backbone, both pyramids + heads.
        -> (pred, (saved_for_keypoint_loss, saved_for_detection_loss))."""
        self._prepare(img_batch)
        eng = self._engine
        ctx = Ctx(self._want_tape())
        c2, c3, c4, c5 = eng.backbone(ctx, img_batch)
        kp = eng.kp_pyramid(ctx, c2, c3, c4, c5)
        det = eng.det_pyramid(ctx, c3, c4, c5)
        pred, saved = eng.keypoint_head(ctx, kp, True)
        cls, reg = eng.detection_head(ctx, det)
        self._finish_forward(ctx)
        outs = self._wrap(ctx, ["k0", "k1", "k2", "k3", "pred", "cls", "reg"], saved + [pred, cls, reg])
        return outs[4], (outs[:5], [outs[5], outs[6], self.anchors(img_batch)])

    def _fpn_features(self, img_batch):
        self._prepare(img_batch)
        eng = self._engine
        ctx = Ctx(False)
        c2, c3, c4, c5 = eng.backbone(ctx, img_batch)
        kp = eng.kp_pyramid(ctx, c2, c3, c4, c5)
        det = eng.det_pyramid(ctx, c3, c4, c5)
        self._finish_forward(ctx)
        return [[ops.export_f32(a, a.C, a.H, a.W) for a in kp], [ops.export_f32(a, a.C, a.H, a.W) for a in det]]

    def prn_forward(self, img_batch):
        """posenet.py:337-350: flatten -> relu(fc) -> drop -> relu(fc) -> drop -> relu(fc) + residual -> softmax over
        all 34272 entries -> [B, h, w, 17].  The three Linear layers run on the conv kernels (1x1 image)."""
        x = img_batch
        if not x.is_cuda:
            raise MpnError("poseNet runs on the MI355X only; there is no CPU path")
        self._prepare(x, prn=True)
        eng = self._engine
        prn = self.prn
        B = x.shape[0]
        n = prn.height * prn.width * 17
        res = x.detach().float().reshape(B, n).contiguous()
        npad = ops.round_up(n, 32)        # activation rows are stored padded to 32 channels (prn_coeff 1 / 3: n % 32 != 0)
        if npad != n:
            xin = torch.zeros((B, 1, 1, npad), dtype=self.compute_dtype, device=x.device)
            xin[:, 0, 0, :n].copy_(res)
        elif ops.is16(self.compute_dtype):
            xin = torch.empty((B, 1, 1, n), dtype=self.compute_dtype, device=x.device)
            ops.cast_lowp(res, xin)
        else:
            xin = res.view(B, 1, 1, n)
        train = torch.is_grad_enabled() and any(p.requires_grad for p in prn.parameters())
        ctx = Ctx(train)
        h, _ = eng.conv(ctx, ops.Act(xin, n), prn.dens1, act=1)
        if prn.drop.training:
            h = eng.dropout(ctx, h, prn.drop.p)
        h, _ = eng.conv(ctx, h, prn.bneck, act=1)
        if prn.drop.training:
            h = eng.dropout(ctx, h, prn.drop.p)
        o, _ = eng.conv(ctx, h, prn.dens2, out_f32=True)            # its ReLU is folded into the softmax kernel
        out = torch.empty((B, n), dtype=torch.float32, device=x.device)
        call("mpn_add_softmax_rows", ops.ptr(o.t), o.Cs, ops.ptr(res), ops.ptr(out), B, n, 1, ops.stream_ptr())
        if ctx.train and o.needs_grad:
            def bwd():
                g = ctx.out_grads.get("prn")
                if g is None:
                    return
                g = g.reshape(B, n).float().contiguous()
                dl = torch.empty((B, n), dtype=torch.float32, device=x.device)
                call("mpn_softmax_rows_backward", ops.ptr(out), ops.ptr(g), ops.ptr(o.t), o.Cs, ops.ptr(dl), B, n, ops.stream_ptr())
                if npad != n:
                    d = torch.zeros((B, 1, 1, npad), dtype=self.compute_dtype, device=x.device)
                    d[:, 0, 0, :n].copy_(dl)
                elif ops.is16(self.compute_dtype):
                    d = torch.empty((B, 1, 1, n), dtype=self.compute_dtype, device=x.device)
                    ops.cast_lowp(dl, d)
                else:
                    d = dl.view(B, 1, 1, n)
                ctx.set_grad(o, ops.Act(d, n))
            ctx.tape.append(bwd)
        out4 = out.view(B, prn.height, prn.width, 17)
        out4 = self._wrap(ctx, ["prn"], [out4])[0]
        return out4, [out4]

    # ------------------------------------------------------------------ losses
    @staticmethod
    def build_loss(saved_for_loss, *args):
        """posenet.py:352-364."""
        subnet_name = args[0]
        if subnet_name == 'keypoint_subnet':
            return build_keypoint_loss(saved_for_loss, args[1], args[2])
        elif subnet_name == 'detection_subnet':
            return build_detection_loss(saved_for_loss, args[1])
        elif subnet_name == 'prn_subnet':
            return build_prn_loss(saved_for_loss, args[1])
        elif subnet_name == 'train_both':
            kp_loss, kp_log = build_keypoint_loss(saved_for_loss[0], args[1], args[2])
            det_loss, det_log = build_detection_loss(saved_for_loss[1], args[3])
            log = OrderedDict(kp_log)
            log.update(det_log)
            return kp_loss + det_loss, log
        else:
            return 0


class _BCEMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, y):
        pc = p.detach().float().contiguous()
        yc = y.detach().float().contiguous()
        n = pc.numel()
        chunks = call("mpn_bce_chunks", n)
        part = ops.workspace(chunks * 4, pc.device, slot=7)
        res = torch.empty(1, dtype=torch.float32, device=pc.device)
        call("mpn_bce_mean_forward", ops.ptr(pc), ops.ptr(yc), n, ops.ptr(part), chunks, ops.ptr(res), ops.stream_ptr())
        ctx.saved = (pc, yc, p.shape)
        return res[0].clone()

    @staticmethod
    def backward(ctx, g):
        pc, yc, shape = ctx.saved
        dp = torch.empty_like(pc)
        gs = g.detach().reshape(1).float().contiguous()
        call("mpn_bce_mean_backward", ops.ptr(pc), ops.ptr(yc), ops.ptr(dp), pc.numel(), ops.ptr(gs), ops.stream_ptr())
        return dp.view(shape), None


def build_prn_loss(saved_for_loss, label):
    """posenet.py:427-445: BCELoss(size_average=True)(out, label)."""
    saved_for_log = OrderedDict()
    total_loss = _BCEMean.apply(saved_for_loss[0], label)
    saved_for_log['PRN loss'] = _log_values(total_loss.detach().reshape(1))[0]
    return total_loss, saved_for_log
