"""Pins the CPU oracle (oracle/*.py, oracle/nms_oracle.c) to the REAL reference.

Every golden vector here was produced by importing the reference itself in the build container
(tests/golden/make_golden.py; reference @ /root/reference, never shipped).  If these pass, the
restatement the GPU tests compare against IS the reference's arithmetic.  CPU only.
"""
import hashlib

import numpy as np
import pytest
import torch

from helpers import gold
from oracle import nms_oracle, posenet_oracle as po
from multiposenet.pytorch_amd import synthetic as weightgen

torch.set_num_threads(8)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def shapes_for(layers):
    g = gold("g0_keys.npz")
    return {str(k): tuple(int(v) for v in str(s).split(",")) if str(s) else ()
            for k, s in zip(g["keys_%d" % layers], g["shapes_%d" % layers])}


def he_sd(layers, grad=False):
    sd = weightgen.gen_state_dict(shapes_for(layers), seed=0, flavour="he", skip_prefixes=("prn.",))
    out = {}
    for k, v in sd.items():
        if v.dtype == np.int64:
            continue
        out[k] = t(v).clone()
        if grad and not k.endswith(("running_mean", "running_var")):
            out[k].requires_grad_(True)
    return out


def close(got, ref, atol=2e-5, rtol=2e-5):
    got, ref = got.detach().double(), ref.double()
    err = (got - ref).abs().max().item()
    assert got.shape == ref.shape
    assert err <= atol + rtol * ref.abs().max().item(), "max err %.3e (ref max %.3e)" % (err, ref.abs().max().item())


def test_anchors_bit_exact_vs_reference():
    g = gold("g1_anchors.npz")
    for (h, w) in ((256, 256), (480, 480), (608, 608), (640, 640), (800, 800), (128, 96), (100, 70)):
        tag = "%dx%d" % (h, w)
        a = po.anchors_for_image(h, w)
        assert tuple(a.shape) == tuple(g["shape_" + tag])
        assert hashlib.sha256(a.tobytes()).hexdigest() == str(g["sha_" + tag])
        assert np.array_equal(a[0, :16], g["head_" + tag]) and np.array_equal(a[0, -16:], g["tail_" + tag])
    assert po.anchors_for_image(480, 480).shape[1] == 43245          # A at 480^2


@pytest.mark.parametrize("layers", [50, 101])
def test_forward_restatement_vs_reference(layers):
    g = gold("g2_forward_r%d.npz" % layers)
    cases = [("eval", 2, 64, 64), ("train", 2, 128, 128)]
    if layers == 50:
        cases.append(("eval", 1, 128, 96))
    for mode, b, h, w in cases:
        tag = "%s_%dx%dx%d" % (mode, b, h, w)
        img = t(weightgen.gen_images(1, b, h, w))
        with torch.no_grad():
            sd = he_sd(layers)
            pred, saved = po.posenet_forward(sd, img, "keypoint_subnet", layers, mode == "train")
            close(pred, t(g["kp_pred_" + tag]))
            for j in range(4):
                close(saved[j], t(g["kp_saved%d_%s" % (j, tag)]))
            if mode == "train":
                close(sd["fpn.bn1.running_mean"], t(g["bn1_rm_" + tag]), 1e-6, 1e-6)
                close(sd["fpn.bn1.running_var"], t(g["bn1_rv_" + tag]), 1e-6, 1e-6)
            sd = he_sd(layers)
            _, ds = po.posenet_forward(sd, img, "detection_subnet", layers, mode == "train")
            close(ds[0], t(g["det_cls_" + tag]))
            close(ds[1], t(g["det_reg_" + tag]))
            if mode == "eval":
                heat, cls, boxes, scores = po.posenet_forward(he_sd(layers), img, "both", layers, False)
                close(heat, t(g["both_heat_" + tag]))
                sc, ci, bx = po.entire_net_postprocess(cls, boxes, scores, "gpu")
                assert sc.shape[0] == g["both_scores_" + tag].shape[0]
                close(sc, t(g["both_scores_" + tag]))
                close(bx, t(g["both_boxes_" + tag]), 1e-3, 1e-5)


def test_losses_and_grads_restatement_vs_reference():
    g = gold("g3_losses_r50.npz")
    b, s = 2, 128
    img = t(weightgen.gen_images(2, b, s, s))
    heat, wgt = weightgen.gen_keypoint_gt(2, b, s // 4, s // 4)
    anno = t(g["anno"])
    sd = he_sd(50, grad=True)
    pred, (ks, ds) = po.posenet_forward(sd, img, "train_both", 50, True)
    l1, log = po.keypoint_loss(ks, t(heat), t(wgt))
    l2, dlog = po.detection_loss(ds, anno)
    (l1 + l2).backward()
    ref = g["both_loss"]
    assert abs(l1.item() - ref[0]) <= 1e-5 * abs(ref[0]) and abs(l2.item() - ref[1]) <= 1e-5 * abs(ref[1])
    assert abs(dlog["classification_loss"] - ref[2]) <= 1e-5 and abs(dlog["regression_loss"] - ref[3]) <= 1e-5
    for k, r in zip(g["kp_lognames"], g["kp_loss"][1:]):
        assert abs(log[str(k)] - r) <= 1e-5 * max(1.0, abs(r))
    ref = dict(zip(g["gnames_both"], g["gnorms_both"]))
    scale = max(ref.values())
    for n, r in ref.items():
        got = sd[str(n)].grad.double().norm().item()
        assert abs(got - r) <= 2e-4 * max(r, 1e-6 * scale), (n, got, r)


def test_focal_restatement_vs_reference_including_empty_image_rule():
    g = gold("g4_focal.npz")
    cls = t(g["cls"]).requires_grad_(True)
    reg = t(g["reg"]).requires_grad_(True)
    c, r = po.focal_loss(cls, reg, t(g["anchors"]), t(g["anno"]))
    (c.mean() + r.mean()).backward()
    assert abs(c.item() - g["loss"][0]) <= 1e-5 * max(1.0, g["loss"][0])
    assert abs(r.item() - g["loss"][1]) <= 1e-5 * max(1.0, g["loss"][1])
    close(cls.grad, t(g["dcls"]), 1e-7, 1e-4)
    close(reg.grad, t(g["dreg"]), 1e-7, 1e-4)
    assert float(cls.grad[2].abs().sum()) == 0.0      # image with no annotation: no gradient


def test_decode_clip_restatement_vs_reference():
    g = gold("g6_decode.npz")
    boxes = po.clip_boxes(po.bbox_transform(t(g["anchors"]), t(g["deltas"])), 128, 96)
    close(boxes, t(g["boxes"]), 1e-4, 1e-6)


def test_prn_restatement_vs_reference():
    g = gold("g7_prn.npz")
    shapes = {k: v for k, v in shapes_for(50).items() if k.startswith("prn.")}
    sd = {k: t(v) for k, v in weightgen.gen_state_dict(shapes, seed=7, flavour="he").items()}
    x = t(weightgen.uniform(7, "prn_in", (3, 56, 36, 17), 0.0, 1.0))
    label = t((weightgen.uniform(7, "prn_label", (3, 56, 36, 17)) < 0.01).astype(np.float32))
    with torch.no_grad():
        out = po.prn_forward(sd, x)
        loss = po.prn_loss(out, label)
    close(out[:, ::7, ::6, :], t(g["out_sample"]), 1e-9, 1e-4)
    assert np.array_equal(out.reshape(3, -1).argmax(1).numpy(), g["argmax"])
    assert abs(loss.item() - g["loss"][0]) <= 1e-5 * g["loss"][0]


def test_nms_c_oracle_vs_goldens_and_independent_numpy():
    g = gold("g5_nms.npz")
    for n in (1, 2, 63, 64, 65, 128, 1000, 4097):
        d = g["dets_%d" % n]
        for mode in ("gpu", "cpu"):
            k = nms_oracle.nms(d, 0.5, mode)
            assert np.array_equal(k, g["keep_%s_%d" % (mode, n)])
            if n <= 1000:
                assert np.array_equal(k, po.nms_numpy(d, 0.5, mode))
    # mask words follow nms_kernel.cu (diagonal tile starts at t+1; strict >)
    d = g["dets_128"]
    keep, mask = nms_oracle.nms(d, 0.5, "gpu", return_mask=True)
    assert mask.shape == (128, 2) and all(((int(mask[i, i // 64]) >> (i % 64)) & 1) == 0 for i in range(128))
    # a box exactly at the threshold separates the two comparison modes
    d2 = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 19, 0.8]], np.float32)      # IoU = 100/200 = 0.5 exactly
    assert list(nms_oracle.nms(d2, 0.5, "gpu")) == [0, 1] and list(nms_oracle.nms(d2, 0.5, "cpu")) == [0]
    assert list(nms_oracle.nms(np.zeros((0, 5), np.float32), 0.5, "gpu")) == []


def test_three_adam_steps_restatement_vs_reference():
    g = gold("g8_steps_r50.npz")
    sd = he_sd(50, grad=True)
    frozen = ("fpn.conv6", "fpn.conv7", "fpn.latlayer", "fpn.toplayer0", "fpn.toplayer1", "fpn.toplayer2",
              "regressionModel", "classificationModel")
    leaves = [v for k, v in sd.items() if v.requires_grad and not k.startswith(frozen)]
    opt = torch.optim.Adam(leaves, lr=1e-4, weight_decay=0.0)
    img = t(weightgen.gen_images(8, 2, 256, 256))
    heat, wgt = weightgen.gen_keypoint_gt(8, 2, 64, 64)
    for step in range(2):
        pred, saved = po.posenet_forward(sd, img, "keypoint_subnet", 50, True)
        loss, _ = po.keypoint_loss(saved, t(heat), t(wgt))
        opt.zero_grad()
        loss.backward()
        opt.step()
        assert abs(loss.item() - g["losses"][step, 0]) <= 2e-4 * g["losses"][step, 0], (step, loss.item())


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_gt_heatmap_restatement_vs_reference(case):
    """oracle/heatmap_oracle.py vs maps rendered by the reference's own putGaussianMaps (heatmap.py:20-41): bit-exact,
    including the 1.0 clamp, keypoints outside the crop, skipped (visibility 2) keypoints and an image without people."""
    from oracle import heatmap_oracle as ho
    g = gold("g9_gt_heatmaps.npz")
    crop, stride, sigma = (float(v) for v in g["cfg_" + case])
    out = ho.gt_heatmaps(g["joints_" + case], g["num_" + case], crop, crop, stride, sigma)
    ref = g["out_" + case]
    assert out.dtype == np.float32 and out.shape == ref.shape
    assert np.array_equal(out, ref)
    assert float(ref.max()) == 1.0 and int(g["num_" + case][1]) == 0 and not ref[1].any()


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_peak_extraction_restatement_vs_reference(case):
    """oracle/joint_oracle.py vs the reference's own find_peaks / NMS(bool_refine_center=False) (joint_utils.py:19-31,
    61-138): exact.  The cv2-bicubic refinement branch has a reference golden only if g10 was made on a box with cv2."""
    from oracle import joint_oracle as jo
    g = gold("g10_peaks.npz")
    heat, up = g["heat_" + case], float(g["up_" + case])
    fp = [jo.find_peaks(0.1, heat[:, :, j]) for j in range(18)]
    assert np.array_equal(np.array([len(p) for p in fp]), g["fp_counts_" + case])
    assert np.array_equal(np.concatenate([p.reshape(-1, 2) for p in fp]), g["fp_xy_" + case])
    plain = np.concatenate([p.reshape(-1, 4) for p in jo.nms_peaks(0.1, heat, up, refine=False)])
    assert np.array_equal(plain, g["nms_plain_" + case])
    if int(g["refined"]) == 1:
        refd = np.concatenate([p.reshape(-1, 4) for p in jo.nms_peaks(0.1, heat, up, refine=True)])
        assert np.array_equal(refd, g["nms_refined_" + case])


def test_reference_nms_kernel_builds_unmodified_into_oracle_ref():
    """oracle/_ref: /root/reference/lib/nms/src/cuda/nms_kernel.cu compiled in place with hipcc (recipe: oracle/Makefile `ref`).
    Here (no GPU) only: the recipe runs where the reference tree exists, both variants export `_nms`, and the files stay out of
    history (.gitignore) but travel to the GPU box (.gpurunignore does not list them).  The comparison itself is
    tests/test_nms_ref_gpu.py (-m gpu)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/root/reference/lib/nms/src/cuda/nms_kernel.cu"):
        pytest.skip("reference tree not present (GPU box): the prebuilt oracle/_ref files are used as they are")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"])
    for so in ("libref_nms_kernel.so", "libref_nms_kernel_nocontract.so"):
        p = os.path.join(root, "oracle", "_ref", so)
        assert os.path.exists(p), p
        syms = subprocess.check_output(["nm", "-D", "--defined-only", p]).decode()
        assert " T _nms" in syms and "nms_kernel" in syms
    assert "oracle/_ref/" in open(os.path.join(root, ".gitignore")).read()
    gi = os.path.join(root, ".gpurunignore")
    assert not os.path.exists(gi) or "oracle/_ref" not in open(gi).read()
    # the recipe must never write into the reference tree (make's built-in `%: %.o` rule once tried to "link" nms_kernel.cu
    # from the nms_kernel.cu.o lying next to it; built-in rules are off in oracle/Makefile)
    mk = open(os.path.join(root, "oracle", "Makefile")).read()
    assert ".SUFFIXES:" in mk and "MAKEFLAGS += -r" in mk
