#!/usr/bin/env python3
"""Generate the committed golden vectors by importing the REAL reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Needs /root/reference (read-only mount).  Never runs on the GPU box; only the .npz files travel.

Import recipe: the reference targets torch 0.4 / CUDA, so
  * lib.nms.pth_nms is pre-seeded (torch.utils.ffi no longer exists) with the C oracle's gpu-mode NMS,
  * Tensor.cuda / Module.cuda become no-ops (this container has no GPU),
  * bool.__rsub__ is patched for the dead statement at network/losses.py:124.
Weights/inputs come from multiposenet/pytorch_amd/synthetic.py so the GPU box can regenerate the identical tensors.
"""
import hashlib
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn

from oracle import nms_oracle
from multiposenet.pytorch_amd import synthetic as weightgen

torch.set_num_threads(8)

# ---- shims -----------------------------------------------------------------------------------
stub = types.ModuleType("lib.nms.pth_nms")


def _pth_nms(dets, thresh):
    keep = nms_oracle.nms(dets.detach().cpu().numpy(), float(thresh), "gpu")
    return torch.from_numpy(keep)


stub.pth_nms = _pth_nms
sys.modules["lib.nms.pth_nms"] = stub
torch.Tensor.cuda = lambda self, *a, **k: self
nn.Module.cuda = lambda self, *a, **k: self
_orig_rsub = torch.Tensor.__rsub__


def _rsub(self, other):
    if self.dtype == torch.bool:
        return torch.logical_not(self)
    return _orig_rsub(self, other)


torch.Tensor.__rsub__ = _rsub

from network.posenet import poseNet, build_keypoint_loss, build_detection_loss, build_prn_loss  # noqa: E402
from network.anchors import Anchors  # noqa: E402
from network.utils import BBoxTransform, ClipBoxes  # noqa: E402
from network.losses import FocalLoss  # noqa: E402


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def load_weights(model, seed, flavour, skip_prn=True):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = weightgen.gen_state_dict(shapes, seed=seed, flavour=flavour,
                                  skip_prefixes=("prn.",) if skip_prn else ())
    model.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    return shapes


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote %s  (%.1f KB)" % (name, os.path.getsize(path) / 1024.0))


def stats(x):
    x = x.detach().double()
    return np.array([x.mean().item(), x.abs().max().item(), x.norm().item()], dtype=np.float64)


# ---- G0: state_dict key/shape contract -----------------------------------------------------
def g0_keys():
    out = {}
    for layers in (50, 101):
        m = poseNet(layers)
        keys = list(m.state_dict().keys())
        shapes = [list(v.shape) for v in m.state_dict().values()]
        out["keys_%d" % layers] = np.array(keys)
        out["shapes_%d" % layers] = np.array([",".join(map(str, s)) for s in shapes])
        out["children_%d" % layers] = np.array([n for n, _ in m.named_children()])
        out["fpn_children_%d" % layers] = np.array([n for n, _ in m.fpn.named_children()])
        del m
    save("g0_keys.npz", **out)


# ---- G1: anchors -----------------------------------------------------------------------------
def g1_anchors():
    out = {}
    an = Anchors()
    for (h, w) in ((256, 256), (480, 480), (608, 608), (640, 640), (800, 800), (128, 96), (100, 70)):
        a = an(torch.zeros(1, 3, h, w)).numpy()
        tag = "%dx%d" % (h, w)
        out["shape_" + tag] = np.array(a.shape)
        out["sha_" + tag] = np.array(hashlib.sha256(a.tobytes()).hexdigest())
        out["head_" + tag] = a[0, :16]
        out["tail_" + tag] = a[0, -16:]
        if h <= 128:
            out["full_" + tag] = a
    save("g1_anchors.npz", **out)


# ---- G2: forward ------------------------------------------------------------------------------
def g2_forward():
    for layers in (50, 101):
        model = poseNet(layers)
        load_weights(model, seed=0, flavour="he")
        out = {}
        for mode in ("eval", "train"):
            for (b, h, w) in ((2, 64, 64), (1, 128, 96), (2, 128, 128)):
                if mode == "train" and (h, w) != (128, 128):
                    continue
                if layers == 101 and (h, w) == (128, 96):
                    continue
                tag = "%s_%dx%dx%d" % (mode, b, h, w)
                img = t(weightgen.gen_images(1, b, h, w))
                load_weights(model, seed=0, flavour="he")       # reset running stats
                model.train() if mode == "train" else model.eval()
                with torch.no_grad():
                    feats = model.fpn(img)
                    for n, f in zip(("fp2", "fp3", "fp4", "fp5"), feats[0]):
                        out["stat_%s_%s" % (n, tag)] = stats(f)
                    for n, f in zip(("p3", "p4", "p5", "p6", "p7"), feats[1]):
                        out["stat_%s_%s" % (n, tag)] = stats(f)
                    if mode == "train":
                        out["bn1_rm_" + tag] = model.fpn.bn1.running_mean.numpy().copy()
                        out["bn1_rv_" + tag] = model.fpn.bn1.running_var.numpy().copy()
                        out["l4bn3_rv_" + tag] = model.fpn.layer4[2].bn3.running_var.numpy().copy()
                        load_weights(model, seed=0, flavour="he")
                        model.train()
                    pred, saved = model([img, "keypoint_subnet"])
                    out["kp_pred_" + tag] = pred.numpy()
                    for j in range(4):
                        out["kp_saved%d_%s" % (j, tag)] = saved[j].numpy()
                    if mode == "train":
                        load_weights(model, seed=0, flavour="he")
                        model.train()
                    _, dsaved = model([img, "detection_subnet"])
                    out["det_cls_" + tag] = dsaved[0].numpy()
                    out["det_reg_" + tag] = dsaved[1].numpy()
                    if mode == "eval":
                        heat, det = model([img, "both"])
                        out["both_heat_" + tag] = heat.numpy()
                        out["both_scores_" + tag] = det[0].numpy()
                        out["both_class_" + tag] = det[1].numpy()
                        out["both_boxes_" + tag] = det[2].numpy()
        save("g2_forward_r%d.npz" % layers, **out)
        del model


# ---- G3: keypoint / detection / combined loss + grads (R50) --------------------------------
def _grad_pack(model, out, tag, full=()):
    names, norms = [], []
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(n)
        norms.append(p.grad.double().norm().item())
    out["gnames_" + tag] = np.array(names)
    out["gnorms_" + tag] = np.array(norms, dtype=np.float64)
    pd = dict(model.named_parameters())
    for n in full:
        g = pd[n].grad.numpy().reshape(-1)
        stride = max(1, g.size // 2048)          # big tensors: strided sample (index i*stride)
        out["g_%s_%s" % (n, tag)] = g[::stride].copy()
        out["gstride_%s_%s" % (n, tag)] = np.array([stride])


FULL_GRADS = ("convfin.bias", "convfin.weight", "fpn.bn1.weight", "fpn.bn1.bias", "fpn.conv1.weight",
              "fpn.layer1.0.conv2.weight", "fpn.layer4.2.bn3.weight", "fpn.flatlayer3.weight",
              "convs4.weight", "fpn.layer2.0.downsample.0.weight")
FULL_GRADS_DET = ("classificationModel.output.bias", "regressionModel.output.bias",
                  "regressionModel.output.weight", "classificationModel.conv1.bias",
                  "fpn.conv7.weight", "fpn.latlayer3.weight", "fpn.layer1.0.conv2.weight")


def g3_losses():
    out = {}
    b, s = 2, 128
    img = t(weightgen.gen_images(2, b, s, s))
    heat, wgt = weightgen.gen_keypoint_gt(2, b, s // 4, s // 4)
    anno = weightgen.gen_boxes_gt(2, b, s, max_n=6)
    anno[1, 3:] = -1
    out["anno"] = anno
    for layers in (50,):
        model = poseNet(layers)
        # keypoint, train-mode BN (trainer.py:172)
        load_weights(model, 0, "he")
        model.train()
        pred, saved = model([img, "keypoint_subnet"])
        loss, log = build_keypoint_loss(saved, t(heat), t(wgt))
        model.zero_grad()
        loss.backward()
        out["kp_loss"] = np.array([loss.item()] + [log[k] for k in log], dtype=np.float64)
        out["kp_lognames"] = np.array(list(log.keys()))
        _grad_pack(model, out, "kp", FULL_GRADS)
        # detection, frozen BN (trainer.py:173-174)
        load_weights(model, 0, "he")
        model.train()
        model.freeze_bn()
        _, dsaved = model([img, "detection_subnet"])
        dloss, dlog = build_detection_loss(dsaved, t(anno))
        model.zero_grad()
        dloss.backward()
        out["det_loss"] = np.array([dlog["total_loss"], dlog["classification_loss"], dlog["regression_loss"]])
        _grad_pack(model, out, "det", FULL_GRADS_DET)
        # combined step: train-mode BN, loss = kp + det, one set of batch statistics
        load_weights(model, 0, "he")
        model.train()
        pred, saved = model([img, "keypoint_subnet"])
        _, dsaved = model([img, "detection_subnet"])   # train-mode BN: outputs use batch stats only
        l1, _ = build_keypoint_loss(saved, t(heat), t(wgt))
        l2, dlog2 = build_detection_loss(dsaved, t(anno))
        model.zero_grad()
        (l1 + l2).backward()
        out["both_loss"] = np.array([l1.item(), l2.item(), dlog2["classification_loss"], dlog2["regression_loss"]])
        _grad_pack(model, out, "both", FULL_GRADS + FULL_GRADS_DET)
    save("g3_losses_r50.npz", **out)


# ---- G4: focal loss on synthetic tensors ---------------------------------------------------
def g4_focal():
    out = {}
    an = Anchors()(torch.zeros(1, 3, 128, 128))
    A = an.shape[1]
    cls = t(weightgen.uniform(4, "cls", (3, A, 1), 0.0, 1.0)).requires_grad_(True)
    cls.data[0, :50, 0] = 0.0        # exercises the clamp (zero-gradient) branch
    cls.data[0, 50:100, 0] = 1.0
    reg = t(weightgen.normal(4, "reg", (3, A, 4), std=0.5)).requires_grad_(True)
    anno = weightgen.gen_boxes_gt(4, 3, 128, max_n=5)
    anno[2, :, :] = -1               # image with no annotation at all -> restated rule (0 loss)
    # reference cannot run the empty branch under torch 2 (losses.py:49-53): run images 0,1 through the
    # reference and apply the restated rule (sum over valid images / batch size) for image 2.
    c2, r2 = FocalLoss()(cls[:2], reg[:2], an, t(anno[:2]))
    closs = c2.mean() * 2.0 / 3.0
    rloss = r2.mean() * 2.0 / 3.0
    (closs + rloss).backward()
    out["anchors"] = an.numpy()
    out["cls"] = cls.detach().numpy()
    out["reg"] = reg.detach().numpy()
    out["anno"] = anno
    out["loss"] = np.array([closs.item(), rloss.item()], dtype=np.float64)
    out["dcls"] = cls.grad.numpy()
    out["dreg"] = reg.grad.numpy()
    save("g4_focal.npz", **out)


# ---- G5: NMS (reference native code is unbuildable; C oracle x independent numpy greedy) ----
def g5_nms():
    from oracle.posenet_oracle import nms_numpy
    out = {}
    for n in (1, 2, 63, 64, 65, 128, 1000, 4097):
        g = np.random.Generator(np.random.Philox(key=1000 + n))
        xy = g.uniform(0, 440, (n, 2))
        wh = g.uniform(8, 160, (n, 2))
        sc = (g.permutation(n) + 1.0) / (n + 1.0)          # no ties
        d = np.concatenate([xy, xy + wh, sc[:, None]], 1).astype(np.float32)
        if n >= 64:                                          # near-duplicates straddling the threshold
            d[1::7, :4] = d[0::7, :4][: len(d[1::7])] + g.uniform(-6, 6, (len(d[1::7]), 4)).astype(np.float32)
        out["dets_%d" % n] = d
        for mode in ("gpu", "cpu"):
            k = nms_oracle.nms(d, 0.5, mode)
            assert np.array_equal(k, nms_numpy(d, 0.5, mode)), (n, mode)
            out["keep_%s_%d" % (mode, n)] = k
    save("g5_nms.npz", **out)


# ---- G6: decode / clip ------------------------------------------------------------------------
def g6_decode():
    an = Anchors()(torch.zeros(1, 3, 128, 96))
    A = an.shape[1]
    deltas = t(weightgen.normal(6, "deltas", (2, A, 4), std=1.5))
    boxes = BBoxTransform()(an, deltas)
    boxes = ClipBoxes()(boxes, torch.zeros(2, 3, 128, 96))
    save("g6_decode.npz", deltas=deltas.numpy(), boxes=boxes.numpy(), anchors=an.numpy())


# ---- G7: PRN eval forward + loss ------------------------------------------------------------
def g7_prn():
    model = poseNet(50)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if k.startswith("prn.")}
    sd = weightgen.gen_state_dict(shapes, seed=7, flavour="he")
    model.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    model.eval()
    x = t(weightgen.uniform(7, "prn_in", (3, 56, 36, 17), 0.0, 1.0))
    label = t((weightgen.uniform(7, "prn_label", (3, 56, 36, 17)) < 0.01).astype(np.float32))
    with torch.no_grad():
        out, saved = model([x, "prn_subnet"])
        loss, _ = build_prn_loss(saved, label)
    o = out.numpy()
    save("g7_prn.npz", out_sample=o[:, ::7, ::6, :], out_sum=o.reshape(3, -1).sum(1),
         argmax=o.reshape(3, -1).argmax(1), loss=np.array([loss.item()]),
         out_stats=stats(out))


# ---- G8: three Adam steps on cfg-1 shapes (R50 keypoint 256^2 B2) --------------------------
def g8_steps():
    model = poseNet(50)
    load_weights(model, 0, "he")
    model.train()
    for name, module in model.fpn.named_children():
        if name in ("conv6", "conv7", "latlayer1", "latlayer2", "latlayer3", "toplayer0", "toplayer1", "toplayer2"):
            for p in module.parameters():
                p.requires_grad = False
    for name, module in model.named_children():
        if name in ("regressionModel", "classificationModel", "prn"):
            for p in module.parameters():
                p.requires_grad = False
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.0)
    img = t(weightgen.gen_images(8, 2, 256, 256))
    heat, wgt = weightgen.gen_keypoint_gt(8, 2, 64, 64)
    losses = []
    for step in range(3):
        pred, saved = model([img, "keypoint_subnet"])
        loss, log = build_keypoint_loss(saved, t(heat), t(wgt))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append([loss.item()] + [log[k] for k in log])
        print("  step", step, losses[-1][0])
    sd = model.state_dict()
    save("g8_steps_r50.npz", losses=np.array(losses, dtype=np.float64),
         convfin_bias=sd["convfin.bias"].numpy(), bn1_weight=sd["fpn.bn1.weight"].numpy(),
         bn1_rm=sd["fpn.bn1.running_mean"].numpy(), bn1_rv=sd["fpn.bn1.running_var"].numpy(),
         conv1_w_stat=stats(sd["fpn.conv1.weight"]), convs4_w_stat=stats(sd["convs4.weight"]),
         nbt=np.array([sd["fpn.bn1.num_batches_tracked"].item()]))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g0", "g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8"]
    fns = dict(g0=g0_keys, g1=g1_anchors, g2=g2_forward, g3=g3_losses, g4=g4_focal, g5=g5_nms,
               g6=g6_decode, g7=g7_prn, g8=g8_steps)
    for w in which:
        print("==", w)
        fns[w]()
