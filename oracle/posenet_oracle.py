"""CPU restatement of the reference ``poseNet`` hot path (TEST INFRASTRUCTURE — the oracle).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (``multiposenet.pytorch_amd``) never does: it runs hand-written HIP
kernels through the C-ABI library and fails loudly when that library is missing.

What this is: a *functional* torch-CPU fp32 restatement of the arithmetic the reference performs,
written against a flat ``state_dict`` (name -> tensor).  The convolution / batch-norm / pooling /
upsample arithmetic itself lives in PyTorch (reference pins pytorch=0.4.0,
multipose_environment.yaml:6); here it is torch 2.10 CPU kernels, whose semantics for these ops are
unchanged (BN eps 1e-5, momentum 0.1, biased variance for normalisation, nearest upsample floor
index).  Parity pin: ``tests/test_oracle_pin.py`` checks every function below against golden vectors
produced by importing the real reference in the build container
(``tests/golden/make_golden.py``), so the restatement is pinned to the reference itself.

Each function cites the reference file:line it follows (paths relative to the reference root).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

# ----------------------------------------------------------------------------------------------
# Optional storage-precision model.  The HIP path in bf16 / f16 keeps every activation and every activation gradient in the 16-bit
# type and accumulates in fp32: operands are rounded when they are STORED (conv outputs after bias / ReLU, BatchNorm outputs after
# the residual add and ReLU, FPN top-down sums, input-gradient tensors), never inside a contraction.  `rounding(dtype)` makes this
# restatement round at those same points — forward value and, straight through, the gradient arriving at that tensor — so a bf16
# run can be judged against fp32 arithmetic that carries the SAME rounding noise instead of against noise-free fp32 (where
# batch-statistics BatchNorm amplifies the difference by ~6 % per bottleneck and hides real errors).  Outputs the HIP path produces
# in f32 straight from its accumulators (convfin, convfin_k2..5, the tower outputs) are not rounded.
# ----------------------------------------------------------------------------------------------
_QUANT = None


class _RoundST(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.to(dt).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).float(), None


class _RoundGrad(torch.autograd.Function):
    """f32 head outputs: the value stays f32, the gradient coming back from the loss is stored in the 16-bit type (import_grad)."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).float(), None


def _q(x):
    return x if _QUANT is None else _RoundST.apply(x, _QUANT)


class rounding(object):
    """with rounding(torch.bfloat16): ... — see above."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _QUANT
        self.prev, _QUANT = _QUANT, self.dtype

    def __exit__(self, *exc):
        global _QUANT
        _QUANT = self.prev


# ----------------------------------------------------------------------------------------------
# backbone + dual FPN                                            network/fpn.py
# ----------------------------------------------------------------------------------------------
def _bn(sd, prefix, x, training, momentum=0.1):
    # nn.BatchNorm2d defaults (fpn.py:15,17,19,25,43)
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"],
                        training=training, momentum=momentum, eps=1e-5)


def _conv(sd, name, x, stride=1, padding=0, f32_out=False):
    w = sd[name + ".weight"]
    if _QUANT is not None:
        w = w + (w.detach().to(_QUANT).float() - w.detach())      # operand copy in the 16-bit type, gradient straight to the f32 master
    y = F.conv2d(x, w, sd.get(name + ".bias"), stride=stride, padding=padding)
    if f32_out:
        return y if _QUANT is None else _RoundGrad.apply(y, _QUANT)
    return _q(y)


def bottleneck(sd, p, x, stride, has_down, bn_training):
    """fpn.py:28-34 (Bottleneck.forward); stride sits on the 3x3 (fpn.py:16)."""
    out = _q(F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x), bn_training)))
    out = _q(F.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, stride=stride, padding=1), bn_training)))
    out = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", out), bn_training)
    if has_down:      # fpn.py:22-26
        sc = _q(_bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride=stride), bn_training))
    else:             # fpn.py:21 (empty Sequential == identity)
        sc = x
    return _q(F.relu(out + sc))


def upsample_add(x, y):
    """fpn.py:84-95: nearest upsample of x to y's exact (H, W), then add."""
    return _q(F.interpolate(x, size=y.shape[2:], mode="nearest") + y)


def fpn_forward(sd, x, layers, bn_training, pre="fpn."):
    """fpn.py:97-126.  Returns ([fp2,fp3,fp4,fp5], [p3,p4,p5,p6,p7], [c2..c5])."""
    c1 = _q(F.relu(_bn(sd, pre + "bn1", _conv(sd, pre + "conv1", _q(x), stride=2, padding=3), bn_training)))
    c1 = F.max_pool2d(c1, kernel_size=3, stride=2, padding=1)
    feats = []
    cur = c1
    in_planes = 64
    for li, (planes, nb, stride) in enumerate(zip((64, 128, 256, 512), BLOCKS[layers], (1, 2, 2, 2))):
        for bi in range(nb):
            s = stride if bi == 0 else 1
            has_down = (s != 1) or (in_planes != planes * 4)
            cur = bottleneck(sd, "%slayer%d.%d" % (pre, li + 1, bi), cur, s, has_down, bn_training)
            in_planes = planes * 4
        feats.append(cur)
    c2, c3, c4, c5 = feats
    # detection pyramid (fpn.py:107-114)
    p6 = _conv(sd, pre + "conv6", c5, stride=2, padding=1)
    p7 = _conv(sd, pre + "conv7", F.relu(p6), stride=2, padding=1)
    p5 = _conv(sd, pre + "latlayer1", c5)
    p4 = upsample_add(p5, _conv(sd, pre + "latlayer2", c4))
    p3 = upsample_add(p4, _conv(sd, pre + "latlayer3", c3))
    p5 = _conv(sd, pre + "toplayer0", p5, padding=1)
    p4 = _conv(sd, pre + "toplayer1", p4, padding=1)
    p3 = _conv(sd, pre + "toplayer2", p3, padding=1)
    # keypoint pyramid (fpn.py:117-124); fp5 is NOT smoothed
    fp5 = _conv(sd, pre + "toplayer", c5)
    fp4 = upsample_add(fp5, _conv(sd, pre + "flatlayer1", c4))
    fp3 = upsample_add(fp4, _conv(sd, pre + "flatlayer2", c3))
    fp2 = upsample_add(fp3, _conv(sd, pre + "flatlayer3", c2))
    fp4 = _conv(sd, pre + "smooth1", fp4, padding=1)
    fp3 = _conv(sd, pre + "smooth2", fp3, padding=1)
    fp2 = _conv(sd, pre + "smooth3", fp2, padding=1)
    return [fp2, fp3, fp4, fp5], [p3, p4, p5, p6, p7], [c2, c3, c4, c5]


# ----------------------------------------------------------------------------------------------
# heads                                                          network/posenet.py
# ----------------------------------------------------------------------------------------------
def _up(x, f):
    return F.interpolate(x, scale_factor=f, mode="nearest")


def keypoint_head(sd, kp_feats, with_intermediate):
    """posenet.py:288-318 (keypoint_forward) / :243-257 (entire-net branch)."""
    p2, p3, p4, p5 = kp_feats
    saved = []
    if with_intermediate:   # posenet.py:296-299
        saved.append(_conv(sd, "convfin_k2", p2, f32_out=True))
        saved.append(_up(_conv(sd, "convfin_k3", p3, f32_out=True), 2))
        saved.append(_up(_conv(sd, "convfin_k4", p4, f32_out=True), 4))
        saved.append(_up(_conv(sd, "convfin_k5", p5, f32_out=True), 8))
    # posenet.py:302-309 — no activation between convt and convs
    q5 = _conv(sd, "convs1", _conv(sd, "convt1", p5, padding=1), padding=1)
    q4 = _conv(sd, "convs2", _conv(sd, "convt2", p4, padding=1), padding=1)
    q3 = _conv(sd, "convs3", _conv(sd, "convt3", p3, padding=1), padding=1)
    q2 = _conv(sd, "convs4", _conv(sd, "convt4", p2, padding=1), padding=1)
    if _QUANT is not None and CONV2_CLASSES and q2.shape[2] % 8 == 0 and q2.shape[3] % 8 == 0:
        h = _conv2_position_classes(sd, q5, q4, q3, q2)                     # the ROUNDING MODEL of csrc/conv2cls.hip; same mathematics
    else:
        cat = torch.cat((_up(q5, 8), _up(q4, 4), _up(q3, 2), q2), 1)        # posenet.py:311-315
        h = F.relu(_conv(sd, "conv2", cat, padding=1))
    pred = _conv(sd, "convfin", h, f32_out=True)
    saved.append(pred)
    return pred, saved


# Rounding model only (never used without `rounding(...)`): the product evaluates conv2 over cat(up8(q5), up4(q4), up2(q3), q2) with the
# x8 / x4 members as nine position-class maps each (csrc/conv2cls.hip).  Mathematically identical to posenet.py:311-315 (checked to
# 3e-7 in fp32 by tests/test_round6_cpu.py); what differs is WHERE 16-bit rounding happens, and an oracle "that rounds where the kernels
# round" has to follow: the frame filters are rounded AFTER the taps were summed in f32, the expanded class maps are rounded once, the
# main part (x2 / x1 members + bias) is rounded when its tile is staged, and the sum is rounded after the ReLU.
CONV2_CLASSES = False
_CLS_ROWS = {0: ((0,), (1, 2), ()), 1: ((), (0, 1, 2), ()), 2: ((), (0, 1), (2,))}      # class -> original taps collected by frame tap u


def _conv2_position_classes(sd, q5, q4, q3, q2):
    w, b = sd["conv2.weight"], sd["conv2.bias"]
    H, W = q2.shape[2], q2.shape[3]

    def rq(t):          # operand copy in the 16-bit type, gradient straight through (as _conv does for weights)
        return t if _QUANT is None else t + (t.detach().to(_QUANT).float() - t.detach())
    e = 0
    for q, s, c0 in ((q5, 8, 0), (q4, 4, 128)):
        wm = w[:, c0: c0 + 128]
        full = torch.zeros(q.shape[0], w.shape[0], H, W, dtype=q.dtype)
        for a in range(s):
            ca = 0 if a == 0 else (2 if a == s - 1 else 1)
            for c in range(s):
                cc = 0 if c == 0 else (2 if c == s - 1 else 1)
                parts = []
                for u in range(3):
                    for v in range(3):
                        taps = [wm[:, :, r, t_] for r in _CLS_ROWS[ca][u] for t_ in _CLS_ROWS[cc][v]]
                        parts.append(sum(taps) if taps else torch.zeros_like(wm[:, :, 0, 0]))
                fr = torch.stack(parts, -1).reshape(wm.shape[0], wm.shape[1], 3, 3)
                full[:, :, a::s, c::s] = F.conv2d(q, rq(fr), None, padding=1)        # class map (f32), scattered to its pixels
        e = e + full
    main = F.conv2d(torch.cat((_up(q3, 2), q2), 1), rq(w[:, 256:]), b, padding=1)
    return _q(F.relu(_q(main) + _q(e)))


def _tower(sd, pre, x):
    out = x
    for i in (1, 2, 3, 4):
        out = F.relu(_conv(sd, "%s.conv%d" % (pre, i), out, padding=1))
    return _conv(sd, pre + ".output", out, padding=1, f32_out=True)


def regression_model(sd, x):
    """posenet.py:52-69: NCHW -> NHWC, view [B, H*W*9, 4]."""
    out = _tower(sd, "regressionModel", x).permute(0, 2, 3, 1)
    return out.contiguous().view(out.shape[0], -1, 4)


def classification_model(sd, x, num_classes=1):
    """posenet.py:95-117: sigmoid, NCHW -> NHWC, view [B, H*W*9, num_classes]."""
    out = torch.sigmoid(_tower(sd, "classificationModel", x)).permute(0, 2, 3, 1)
    return out.contiguous().view(x.shape[0], -1, num_classes)


def detection_head(sd, det_feats):
    """posenet.py:327-328: shared-weight towers over p3..p7, concatenated on dim 1."""
    reg = torch.cat([regression_model(sd, f) for f in det_feats], dim=1)
    cls = torch.cat([classification_model(sd, f) for f in det_feats], dim=1)
    return cls, reg


# ----------------------------------------------------------------------------------------------
# anchors / decode / clip                     network/anchors.py, network/utils.py
# ----------------------------------------------------------------------------------------------
def generate_anchors(base_size, ratios, scales):
    """anchors.py:39-70.  float64; order = ratio-major, scale-minor."""
    n = len(ratios) * len(scales)
    a = np.zeros((n, 4))
    a[:, 2:] = base_size * np.tile(scales, (2, len(ratios))).T
    areas = a[:, 2] * a[:, 3]
    a[:, 2] = np.sqrt(areas / np.repeat(ratios, len(scales)))
    a[:, 3] = a[:, 2] * np.repeat(ratios, len(scales))
    a[:, 0::2] -= np.tile(a[:, 2] * 0.5, (2, 1)).T
    a[:, 1::2] -= np.tile(a[:, 3] * 0.5, (2, 1)).T
    return a


def anchors_for_image(h, w):
    """anchors.py:21-37 + shift :106-126.  Returns float32 [1, A, 4]; order (level, y, x, anchor)."""
    levels = [3, 4, 5, 6, 7]
    ratios = np.array([0.5, 1, 2])
    scales = np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)])
    shape = np.array([h, w])
    out = np.zeros((0, 4)).astype(np.float32)
    for lv in levels:
        fs = (shape + 2 ** lv - 1) // (2 ** lv)
        base = generate_anchors(2 ** (lv + 2), ratios, scales)
        sx = (np.arange(0, fs[1]) + 0.5) * 2 ** lv
        sy = (np.arange(0, fs[0]) + 0.5) * 2 ** lv
        sx, sy = np.meshgrid(sx, sy)
        shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
        k = shifts.shape[0]
        allp = (base.reshape((1, 9, 4)) + shifts.reshape((1, k, 4)).transpose((1, 0, 2))).reshape((k * 9, 4))
        out = np.append(out, allp, axis=0)
    return np.expand_dims(out, 0).astype(np.float32)


def bbox_transform(anchors, deltas):
    """utils.py:19-43 (mean 0, std [.1,.1,.2,.2])."""
    std = torch.tensor([0.1, 0.1, 0.2, 0.2], dtype=torch.float32)
    w = anchors[:, :, 2] - anchors[:, :, 0]
    h = anchors[:, :, 3] - anchors[:, :, 1]
    cx = anchors[:, :, 0] + 0.5 * w
    cy = anchors[:, :, 1] + 0.5 * h
    dx = deltas[:, :, 0] * std[0]
    dy = deltas[:, :, 1] * std[1]
    dw = deltas[:, :, 2] * std[2]
    dh = deltas[:, :, 3] * std[3]
    pcx = cx + dx * w
    pcy = cy + dy * h
    pw = torch.exp(dw) * w
    ph = torch.exp(dh) * h
    return torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], dim=2)


def clip_boxes(boxes, height, width):
    """utils.py:51-61."""
    b = boxes.clone()
    b[:, :, 0] = torch.clamp(b[:, :, 0], min=0)
    b[:, :, 1] = torch.clamp(b[:, :, 1], min=0)
    b[:, :, 2] = torch.clamp(b[:, :, 2], max=width)
    b[:, :, 3] = torch.clamp(b[:, :, 3], max=height)
    return b


# ----------------------------------------------------------------------------------------------
# losses                                         network/posenet.py:367-445, network/losses.py
# ----------------------------------------------------------------------------------------------
def keypoint_loss(saved_for_loss, heat_temp, heat_weight):
    """posenet.py:367-403.  MSELoss(size_average=True) == mean over all B*18*h*w elements."""
    names = ["heatmap_loss_k2", "heatmap_loss_k3", "heatmap_loss_k4", "heatmap_loss_k5", "heatmap_loss"]
    log = OrderedDict()
    total = 0
    for j in range(5):
        pred = saved_for_loss[j][:, :18] * heat_weight
        gt = heat_weight * heat_temp
        l = F.mse_loss(pred, gt, reduction="mean")
        total = total + l
        log[names[j]] = l.item()
    log["max_ht"] = saved_for_loss[-1][:, :18].max().item()
    log["min_ht"] = saved_for_loss[-1][:, :18].min().item()
    return total, log


def calc_iou(a, b):
    """losses.py:5-22 (no +1 convention; union clamped at 1e-8)."""
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iw = torch.min(a[:, 2].unsqueeze(1), b[:, 2]) - torch.max(a[:, 0].unsqueeze(1), b[:, 0])
    ih = torch.min(a[:, 3].unsqueeze(1), b[:, 3]) - torch.max(a[:, 1].unsqueeze(1), b[:, 1])
    iw = torch.clamp(iw, min=0)
    ih = torch.clamp(ih, min=0)
    ua = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])).unsqueeze(1) + area - iw * ih
    ua = torch.clamp(ua, min=1e-8)
    return iw * ih / ua


def focal_loss(classifications, regressions, anchors, annotations):
    """losses.py:27-137.  Returns (cls_loss[1], reg_loss[1]) batch means.

    Restated rule for images with zero valid annotations (losses.py:49-53 cannot run under
    torch>=1: ``torch.tensor(0, requires_grad=True)`` on an int): such an image contributes 0 to
    both batch means and no gradient.
    """
    alpha, gamma = 0.25, 2.0
    B = classifications.shape[0]
    cls_losses, reg_losses = [], []
    anchor = anchors[0]
    aw = anchor[:, 2] - anchor[:, 0]
    ah = anchor[:, 3] - anchor[:, 1]
    acx = anchor[:, 0] + 0.5 * aw
    acy = anchor[:, 1] + 0.5 * ah
    for j in range(B):
        cls = classifications[j]
        reg = regressions[j]
        ann = annotations[j]
        ann = ann[ann[:, 4] != -1]
        if ann.shape[0] == 0:
            cls_losses.append(torch.zeros(()))
            reg_losses.append(torch.zeros(()))
            continue
        cls = torch.clamp(cls, 1e-4, 1.0 - 1e-4)
        iou = calc_iou(anchor, ann[:, :4])
        iou_max, iou_arg = torch.max(iou, dim=1)
        targets = -torch.ones_like(cls)
        targets[iou_max < 0.4, :] = 0
        pos = iou_max >= 0.5
        npos = pos.sum()
        assigned = ann[iou_arg]
        targets[pos, :] = 0
        targets[pos, assigned[pos, 4].long()] = 1
        af = torch.where(targets == 1., torch.full_like(targets, alpha), torch.full_like(targets, 1. - alpha))
        fw = torch.where(targets == 1., 1. - cls, cls)
        fw = af * torch.pow(fw, gamma)
        bce = -(targets * torch.log(cls) + (1.0 - targets) * torch.log(1.0 - cls))
        cl = fw * bce
        cl = torch.where(targets != -1.0, cl, torch.zeros_like(cl))
        cls_losses.append(cl.sum() / torch.clamp(npos.float(), min=1.0))
        if npos > 0:
            asg = assigned[pos]
            gw = asg[:, 2] - asg[:, 0]
            gh = asg[:, 3] - asg[:, 1]
            gcx = asg[:, 0] + 0.5 * gw
            gcy = asg[:, 1] + 0.5 * gh
            gw = torch.clamp(gw, min=1)
            gh = torch.clamp(gh, min=1)
            t = torch.stack(((gcx - acx[pos]) / aw[pos], (gcy - acy[pos]) / ah[pos],
                             torch.log(gw / aw[pos]), torch.log(gh / ah[pos]))).t()
            t = t / torch.tensor([[0.1, 0.1, 0.2, 0.2]])
            d = torch.abs(t - reg[pos])
            rl = torch.where(d <= 1.0 / 9.0, 0.5 * 9.0 * d * d, d - 0.5 / 9.0)
            reg_losses.append(rl.mean())
        else:
            reg_losses.append(torch.zeros(()))
    return (torch.stack(cls_losses).mean(dim=0, keepdim=True),
            torch.stack(reg_losses).mean(dim=0, keepdim=True))


def detection_loss(saved_for_loss, anno):
    """posenet.py:405-425."""
    c, r = focal_loss(saved_for_loss[0], saved_for_loss[1], saved_for_loss[2], anno)
    c = c.mean()
    r = r.mean()
    total = c + r
    log = OrderedDict(total_loss=total.item(), classification_loss=c.item(), regression_loss=r.item())
    return total, log


# ----------------------------------------------------------------------------------------------
# PRN                                                  network/posenet.py:130-152, 337-350, 427-445
# ----------------------------------------------------------------------------------------------
def prn_forward(sd, x):
    """posenet.py:337-350 in eval mode (dropout == identity)."""
    res = x.reshape(x.shape[0], -1)
    out = F.relu(F.linear(res, sd["prn.dens1.weight"], sd["prn.dens1.bias"]))
    out = F.relu(F.linear(out, sd["prn.bneck.weight"], sd["prn.bneck.bias"]))
    out = F.relu(F.linear(out, sd["prn.dens2.weight"], sd["prn.dens2.bias"]))
    out = torch.softmax(out + res, dim=1)
    return out.view(x.shape[0], x.shape[1], x.shape[2], 17)


def prn_loss(out, label):
    """posenet.py:427-445: BCELoss mean."""
    return F.binary_cross_entropy(out, label, reduction="mean")


# ----------------------------------------------------------------------------------------------
# NMS (python mirror of oracle/nms_oracle.c — kept tiny; the C file is the timed/port oracle)
# ----------------------------------------------------------------------------------------------
def nms_numpy(dets, thresh, mode="gpu"):
    """Independent O(N^2) greedy NMS used to cross-check oracle/nms_oracle.c.

    mode 'gpu': lib/nms/src/cuda/nms_kernel.cu:16-24,53-68 + nms_cuda.c:47-58 (IoU +1, strict >);
    mode 'cpu': lib/nms/src/nms.c:35-63 (IoU +1, >=).  Sort = descending score, ties -> lower index
    (pth_nms.py:17/33 uses an unstable sort; goldens avoid ties).  Returns original indices.
    """
    dets = np.asarray(dets, dtype=np.float32)
    n = dets.shape[0]
    order = np.argsort(-dets[:, 4], kind="stable")
    f = np.float32
    x1, y1, x2, y2 = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3]
    areas = (x2 - x1 + f(1)) * (y2 - y1 + f(1))
    sup = np.zeros(n, bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if sup[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(f(0), xx2 - xx1 + f(1)); h = np.maximum(f(0), yy2 - yy1 + f(1))
        inter = (w * h).astype(np.float32)
        ovr = inter / ((areas[i] + areas[rest]).astype(np.float32) - inter)
        if mode == "gpu":
            sup[rest[ovr > f(thresh)]] = True
        else:
            sup[rest[ovr >= f(thresh)]] = True
    return np.asarray(keep, dtype=np.int64)


# ----------------------------------------------------------------------------------------------
# whole-path drivers used by tests / cpu_baseline
# ----------------------------------------------------------------------------------------------
def posenet_forward(sd, img, subnet, layers, bn_training):
    """posenet.py:226-285 dispatch.  'both' returns (heat, cls, boxes, scores) *before* thresholding."""
    if subnet == "keypoint_subnet":
        kp, _, _ = fpn_forward(sd, img, layers, bn_training)
        return keypoint_head(sd, kp, True)
    if subnet == "detection_subnet":
        _, det, _ = fpn_forward(sd, img, layers, bn_training)
        cls, reg = detection_head(sd, det)
        anc = torch.from_numpy(anchors_for_image(img.shape[2], img.shape[3]))
        return [], [cls, reg, anc]
    if subnet == "train_both":   # one shared backbone pass, both heads, both losses
        kp, det, _ = fpn_forward(sd, img, layers, bn_training)
        pred, saved = keypoint_head(sd, kp, True)
        cls, reg = detection_head(sd, det)
        anc = torch.from_numpy(anchors_for_image(img.shape[2], img.shape[3]))
        return pred, (saved, [cls, reg, anc])
    kp, det, _ = fpn_forward(sd, img, layers, bn_training)
    heat, _ = keypoint_head(sd, kp, False)
    cls, reg = detection_head(sd, det)
    anc = torch.from_numpy(anchors_for_image(img.shape[2], img.shape[3]))
    boxes = clip_boxes(bbox_transform(anc, reg), img.shape[2], img.shape[3])
    scores = torch.max(cls, dim=2, keepdim=True)[0]
    return heat, cls, boxes, scores


def entire_net_postprocess(cls, boxes, scores, nms_mode="gpu"):
    """posenet.py:271-285 — score>0.05 on image 0 only, NMS 0.5, then max over classes."""
    over = (scores > 0.05)[0, :, 0]
    if over.sum() == 0:
        return [torch.zeros(0), torch.zeros(0), torch.zeros(0, 4)]
    c = cls[:, over, :]
    b = boxes[:, over, :]
    s = scores[:, over, :]
    keep = nms_numpy(torch.cat([b, s], dim=2)[0].numpy(), 0.5, nms_mode)
    keep = torch.from_numpy(keep)
    sc, ci = c[0, keep, :].max(dim=1)
    return [sc, ci, b[0, keep, :]]
