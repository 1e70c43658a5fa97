"""Round-3 GPU tests: BASELINE config 2 at its exact size, the operand-resident 1x1 kernels, the virtual 512-channel
concatenation, the in-kernel split-K reduction, a bf16 drift gate against an oracle that rounds where the HIP path rounds,
the Trainer / Tester.val drivers against fixtures recorded from the real reference, config 5 end to end."""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT, gold, report
from test_model_gpu import close, get_model, load_he, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests selected but no GPU is visible"
    from multiposenet.pytorch_amd import _lib
    _lib.lib()


# ------------------------------------------------------------------------------------------------ BASELINE config 2
def test_cfg2_r50_keypoint_480_batch16_fp32_full_size():
    """BASELINE config 2 at its exact workload (multipose_keypoint_train.py:49-64 shapes): ResNet-50 `keypoint_subnet`,
    480x480, batch 16, fp32 arithmetic, batch-statistics BatchNorm, every non-PRN parameter trainable.  At full size: loss and
    the whole gradient arena finite, bit-reproducible, detection-pyramid parameters untouched (the keypoint tape skips them);
    on a batch-2 slice of the same inputs the fp32 loss and every logged per-level loss equal the CPU oracle's within 2e-4
    relative, the heat-maps within the north-star 1e-3 abs."""
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from oracle import posenet_oracle as po
    from multiposenet.pytorch_amd import synthetic as weightgen
    B, S = 16, 480
    m = get_model(50, torch.float32)
    for p in m.prn.parameters():
        p.requires_grad = False
    m.train()
    img = t(weightgen.gen_images(310, B, S, S)).cuda()
    heat, wgt = (t(a).cuda() for a in weightgen.gen_keypoint_gt(311, B, S // 4, S // 4))
    bn_state = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}

    def run(sl):
        m.load_state_dict(bn_state, strict=False)
        m._arena.ensure_grads()
        m._arena.grad_flat.zero_()
        pred, saved = m([img[sl].contiguous(), "keypoint_subnet"])
        loss, log = poseNet.build_loss(saved, "keypoint_subnet", heat[sl].contiguous(), wgt[sl].contiguous())
        loss.backward()
        torch.cuda.synchronize()
        return pred.detach().clone(), loss.detach().clone(), m._arena.grad_flat.clone(), {k: float(v) for k, v in log.items()}
    full = slice(0, B)
    p0, l0, g0, log0 = run(full)
    p1, l1, g1, _ = run(full)
    assert p0.shape == (B, 18, S // 4, S // 4) and p0.dtype == torch.float32
    assert torch.isfinite(l0) and torch.isfinite(g0).all() and float(g0.abs().max()) > 0
    assert torch.equal(p0, p1) and torch.equal(l0, l1) and torch.equal(g0, g1), "cfg2 step is not bit-reproducible"
    ar = m._arena
    for name, prm in m.named_parameters():
        i = ar.index.get(id(prm))
        if i is None:
            continue
        gn = float(g0[ar.offsets[i]: ar.offsets[i] + ar.sizes[i]].abs().max())
        det_only = name.startswith(("regressionModel.", "classificationModel.", "fpn.conv6", "fpn.conv7", "fpn.latlayer", "fpn.toplayer0",
                                    "fpn.toplayer1", "fpn.toplayer2", "prn."))
        if det_only:
            assert gn == 0.0, "%s received a gradient in the keypoint-only step" % name
        else:
            assert gn > 0.0, "%s has no gradient in the keypoint step" % name
    # batch-2 slice against the oracle (BatchNorm uses the slice's own statistics on both sides)
    sl = slice(0, 2)
    ps, ls, gs, logs = run(sl)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if v.dtype != torch.int64 and not k.startswith("prn.")}
    sd.update({k: v.cpu().clone() for k, v in bn_state.items() if v.dtype != torch.int64})
    with torch.no_grad():
        opred, oks = po.posenet_forward(sd, img[sl].cpu(), "keypoint_subnet", 50, True)
        ol, olog = po.keypoint_loss(oks, heat[sl].cpu(), wgt[sl].cpu())
    rel = abs(float(ls) - float(ol)) / abs(float(ol))
    close("cfg2 slice (R50 kp 480x480 B=2 fp32) heat-map vs oracle", ps, opred, 1e-3, 1e-4)
    for k, v in olog.items():
        if k in logs and isinstance(v, float) and abs(v) > 1e-12:
            assert abs(logs[k] - v) <= 2e-4 * abs(v) + 1e-7, "log value %s: %.7g vs oracle %.7g" % (k, logs[k], v)
    report("cfg2 full size (R50 keypoint_subnet 480x480 B=16 fp32): finite, bit-reproducible, loss %.6f; B=2 slice loss %.7f vs oracle %.7f (rel %.2e)"
           % (float(l0), float(ls), float(ol), rel))
    assert rel <= 2e-4


# ------------------------------------------------------------------------------------------------ drivers
def test_trainer_follows_the_real_reference_trainer_on_device(tmp_path):
    """The g14 fixture (recorded from the real reference Trainer) with model, optimizer state and batches on the MI355X:
    same learning rates, modes, files, resume facts; restored Adam moments live on the device."""
    from test_round3_cpu import check_trainer_against_reference
    assert check_trainer_against_reference(tmp_path, device=0) == 10
    report("Trainer on the device: 10 scenarios identical to the real reference Trainer (g14_trainer.json)")


def test_trainer_uses_the_recorded_step_and_tester_val_matches_a_hand_loop(tmp_path):
    """poseNet + FusedAdam through Trainer: the step is the recorded launch list (replays counted), the parameters equal the
    hand-written eager loop bit for bit, log values arrive as numbers without a per-step sync; Tester.val (tester.py:515-543)
    over the same validation batches returns the mean of the per-batch eval-mode losses computed by hand."""
    from multiposenet.pytorch_amd.evaluate.tester import Tester, TestParams
    from multiposenet.pytorch_amd.optim import FusedAdam
    from multiposenet.pytorch_amd.network import losses
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from multiposenet.pytorch_amd.training.batch_processor import batch_processor, train_step
    from multiposenet.pytorch_amd.training.trainer import Trainer, TrainParams
    from multiposenet.pytorch_amd import synthetic as weightgen
    B, S = 2, 64

    def loader(n, seed):
        out = []
        for i in range(n):
            heat, wgt = (t(a) for a in weightgen.gen_keypoint_gt(seed + 10 + i, B, S // 4, S // 4))
            out.append((t(weightgen.gen_images(seed + i, B, S, S)), heat, wgt))
        return out
    train_data, val_data = loader(5, 400), loader(3, 500)
    model = get_model(50, torch.bfloat16)
    for p in model.prn.parameters():
        p.requires_grad = False
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    model.train()
    opt = FusedAdam(model, lr=1e-3)

    class S_(object):
        pass
    st = S_(); st.model = model; st.params = S_(); st.params.subnet_name = 'keypoint_subnet'; st.params.gpus = [0]
    for batch in train_data:
        inputs, gts, _ = batch_processor(st, batch)
        train_step(model, opt, inputs, gts)
    want = model._arena.flat.clone()
    model.load_state_dict(state0)
    P = TrainParams(exp_name='unit3', subnet_name='keypoint_subnet', batch_size=B, max_epoch=1, save_dir=str(tmp_path / "run"),
                    print_freq=2, val_nbatch_end_epoch=0)
    P.optimizer = FusedAdam(model, lr=1e-3)
    was_lazy = losses.LAZY_LOG
    try:
        tr = Trainer(model, P, batch_processor, train_data, None)
        tr.train()
        assert losses.LAZY_LOG == was_lazy, "Trainer.train() must restore the process-wide lazy-log setting"
        torch.cuda.synchronize()
        assert tr._step.fast is not None and tr._step.fast.replays == 3       # 1 eager + 1 recording + 3 replays
        assert torch.equal(model._arena.flat, want), "Trainer's recorded steps differ from the hand-written eager loop"
        assert "heatmap_loss" in tr.log_values and np.isfinite(tr.log_values["heatmap_loss"].value()[0]) or tr.log_values["heatmap_loss"].count == 0
    finally:
        losses.set_lazy_log(was_lazy)
    # Tester.val over val_data == the hand loop (eval mode, frozen statistics)
    tp = TestParams()
    tp.ckpt, tp.subnet_name, tp.batch_size, tp.print_freq = os.path.join(P.save_dir, "ckpt_1.h5"), 'keypoint_subnet', B, 2
    fresh = poseNet(50, compute_dtype=torch.bfloat16)
    tester = Tester(fresh, tp, batch_processor=batch_processor, val_data=val_data)
    assert torch.equal(fresh._arena.flat, want)
    mean, std = tester.val()
    hand = []
    fresh.eval()
    with torch.no_grad():
        for batch in val_data:
            inputs, gts, _ = batch_processor(tester, batch)
            _, saved = fresh(*inputs)
            loss, _ = fresh.build_loss(saved, *gts)
            hand.append(float(loss))
    assert abs(mean - float(np.mean(hand))) <= 1e-6 * abs(np.mean(hand)) and abs(std - float(np.std(hand, ddof=1))) <= 1e-5 * max(np.std(hand, ddof=1), 1e-9) + 1e-9
    assert not fresh.training and not any(m.training for m in fresh._bns)
    report("Trainer(poseNet, FusedAdam): 5 steps through the recorded list == eager loop bit for bit; Tester.val mean %.6f == hand loop" % mean)


# ------------------------------------------------------------------------------------------------ bf16 drift with teeth
def _oracle_leaves(sd_np):
    sd = {k: torch.from_numpy(v).clone() for k, v in sd_np.items() if v.dtype != np.int64}
    leaves = {}
    for k, v in sd.items():
        if not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
            leaves[k] = v
    return sd, leaves


def test_bf16_bottlenecks_against_an_oracle_that_rounds_where_the_kernels_round():
    """VERDICT r2 weak 4.  With batch-statistics BatchNorm and He-random weights the network amplifies ANY perturbation (~6 % per
    bottleneck of what is injected), so a whole-network bf16 comparison cannot be tight whatever the reference — see the next test.
    The teeth are here: every Bottleneck of R50 (fpn.py:9-34) is run ALONE in bf16, forward and backward, on inputs taken from the
    CPU oracle, against the oracle of the same block rounding to bf16 at the same points as the kernels (operand copies of the
    weights, every stored activation, every stored activation gradient; fp32 accumulation).  What is left is fp32 summation order
    and the rare rounding flips it causes: block output within 2e-3 rel-L2 (measured <= 3e-4: flip level), input gradient within
    3e-2 and every parameter gradient within 5e-2 (three batch-statistics BatchNorm backward passes per block re-amplify the flips;
    measured about a third of the gates) — a wrong epilogue, mask, statistic or rounding point in any bf16 train-mode kernel moves
    these by far more."""
    from multiposenet.pytorch_amd import ops
    from multiposenet.pytorch_amd.engine import Ctx
    from oracle import posenet_oracle as po
    from multiposenet.pytorch_amd import synthetic as weightgen
    import torch.nn.functional as F
    B, S = 4, 128
    m = get_model(50, torch.bfloat16)
    sd_np = load_he(m)
    for p in m.prn.parameters():
        p.requires_grad = False
    m.train()
    eng = m._engine
    img = t(weightgen.gen_images(700, B, S, S))
    m._prepare(img.cuda())                               # refresh the bf16 operand copies of the weights
    Q = torch.bfloat16
    overlap = eng.overlap_wgrad
    eng.overlap_wgrad = False                            # weight gradients inline: the test pops the tape itself
    worst = {"out": 0.0, "dx": 0.0, "dw": 0.0}
    try:
        sd, leaves = _oracle_leaves(sd_np)
        with po.rounding(Q), torch.no_grad():
            cur = po._q(F.relu(po._bn(sd, "fpn.bn1", po._conv(sd, "fpn.conv1", po._q(img), stride=2, padding=3), True)))
            cur = F.max_pool2d(cur, kernel_size=3, stride=2, padding=1)
        in_planes, nblk = 64, 0
        for li, (planes, nb, stride) in enumerate(zip((64, 128, 256, 512), po.BLOCKS[50], (1, 2, 2, 2))):
            for bi in range(nb):
                s_ = stride if bi == 0 else 1
                has_down = (s_ != 1) or (in_planes != planes * 4)
                prefix = "fpn.layer%d.%d" % (li + 1, bi)
                blk = getattr(m.fpn, "layer%d" % (li + 1))[bi]
                # oracle block, forward + backward
                for v in leaves.values():
                    v.grad = None
                x_leaf = cur.clone().requires_grad_(True)
                with po.rounding(Q):
                    out_o = po.bottleneck(sd, prefix, po._q(x_leaf), s_, has_down, True)
                g = torch.Generator().manual_seed(900 + nblk)
                d_out = (torch.randn(out_o.shape, generator=g) * 0.05).to(Q).float()
                out_o.backward(d_out)
                # HIP block on the same input / output gradient
                m._arena.ensure_grads()
                m._arena.grad_flat.zero_()
                ctx = Ctx(True)
                x = ops.Act(cur.permute(0, 2, 3, 1).contiguous().to(Q).cuda(), cur.shape[1], needs_grad=True)
                out_h = eng.bottleneck(ctx, x, blk)
                ctx.set_grad(out_h, ops.Act(d_out.permute(0, 2, 3, 1).contiguous().to(Q).cuda(), out_o.shape[1]))
                while ctx.tape:
                    ctx.tape.pop()()
                torch.cuda.synchronize()

                def rel(a, b_):
                    return float((a.double() - b_.double()).norm() / max(float(b_.double().norm()), 1e-12))
                r_out = rel(out_h.t.float().cpu().permute(0, 3, 1, 2), out_o.detach())
                r_dx = rel(ctx.grad_of(x).t.float().cpu().permute(0, 3, 1, 2), x_leaf.grad)
                r_dw, r_dw_name = 0.0, ""
                for name, prm in blk.named_parameters():
                    go = leaves[prefix + "." + name].grad
                    r = rel(prm.grad.detach().float().cpu(), go)
                    if r > r_dw:
                        r_dw, r_dw_name = r, name
                worst = {"out": max(worst["out"], r_out), "dx": max(worst["dx"], r_dx), "dw": max(worst["dw"], r_dw)}
                report("    %-18s out %.2e  dx %.2e  worst dparam %.2e (%s)" % (prefix, r_out, r_dx, r_dw, r_dw_name))
                assert r_out <= 2e-3 and r_dx <= 3e-2 and r_dw <= 5e-2, "%s: out %.2e dx %.2e dparam %.2e (%s)" % (prefix, r_out, r_dx, r_dw, r_dw_name)
                cur = out_o.detach()
                in_planes = planes * 4
                nblk += 1
    finally:
        eng.overlap_wgrad = overlap
    report("bf16 bottlenecks vs same-rounding oracle (R50, 16 blocks, teacher-forced, batch-stat BN): worst rel-L2 output %.2e, "
           "input gradient %.2e, parameter gradient %.2e" % (worst["out"], worst["dx"], worst["dw"]))


def test_bf16_training_step_sits_closer_to_the_rounding_oracle_than_to_fp32():
    """Whole network, R50 `train_both` 128x128 batch 4, batch statistics: the bf16 HIP step against (a) the fp32 oracle and (b)
    the oracle that rounds where the kernels round.  Both oracles run the same chaotic map, so the distances are large either way,
    but the HIP result must be CLOSER to (b) than to (a) — heat-maps and the typical parameter gradient — and the loss must agree
    with (b) to 1e-3."""
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from oracle import posenet_oracle as po
    from multiposenet.pytorch_amd import synthetic as weightgen
    B, S = 4, 128
    m = get_model(50, torch.bfloat16)
    sd_np = load_he(m)
    for p in m.prn.parameters():
        p.requires_grad = False
    m.train()
    img = t(weightgen.gen_images(700, B, S, S))
    heat, wgt = (t(a) for a in weightgen.gen_keypoint_gt(701, B, S // 4, S // 4))
    anno = t(weightgen.gen_boxes_gt(702, B, S))
    m._arena.ensure_grads()
    m._arena.grad_flat.zero_()
    pred, saved = m([img.cuda(), "train_both"])
    loss, log = poseNet.build_loss(saved, "train_both", heat.cuda(), wgt.cuda(), anno.cuda())
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if p.requires_grad and p.grad is not None}

    def oracle(quant):
        sd, leaves = _oracle_leaves(sd_np)
        with po.rounding(torch.bfloat16 if quant else None):
            opred, (oks, ods) = po.posenet_forward(sd, img, "train_both", 50, True)
            l1, _ = po.keypoint_loss(oks, heat, wgt)
            l2, _ = po.detection_loss(ods, anno)
            (l1 + l2).backward()
        return opred.detach(), float(l1 + l2), {k: v.grad for k, v in leaves.items() if v.grad is not None}
    p32, l32, g32 = oracle(False)
    pq, lq, gq = oracle(True)
    got = pred.detach().float().cpu()
    rl2_q = float((got - pq).norm() / pq.norm())
    rl2_32 = float((got - p32).norm() / p32.norm())

    def med(ref):
        rels = []
        for name, g in ref.items():
            if name in grads and float(g.norm()) > 1e-7:
                rels.append(abs(float(grads[name].norm()) - float(g.norm())) / float(g.norm()))
        return float(np.median(rels)), len(rels)
    mq, n = med(gq)
    m32, _ = med(g32)
    report("bf16 R50 train_both 128x128 B=4, batch-stat BN: heat-map rel-L2 vs rounding oracle %.2e, vs fp32 oracle %.2e; loss %.5f vs %.5f / %.5f; "
           "median of %d per-parameter gradient-norm errors %.2e vs %.2e" % (rl2_q, rl2_32, float(loss), lq, l32, n, mq, m32))
    assert rl2_q < rl2_32 and rl2_q <= 5e-2
    assert abs(float(loss) - lq) <= 1e-3 * abs(lq)
    assert n >= 150 and mq <= 2e-2


# ------------------------------------------------------------------------------------------------ ReLU mask bits
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_relu_mask_bits_replace_z_in_both_backward_passes(dtype):
    """relu(bn3(.) + shortcut) (fpn.py:30-33): the forward writes the sign bits of z (one byte per 16-byte chunk) and the two
    backward passes that need the ReLU mask — the statistics in the epilogue of the dgrad that completes dz, and bn_bwd_apply —
    read them instead of z.  Same predicate, so the whole training step (loss, every gradient) is bit-identical with the bits on
    and off; the bits themselves equal (z > 0) element for element."""
    from multiposenet.pytorch_amd import ops
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from multiposenet.pytorch_amd import synthetic as weightgen
    # kernel level
    B, H, W, C = 2, 9, 7, 64
    g = torch.Generator().manual_seed(5)
    y = ops.Act(torch.randn(B, H, W, C, generator=g).to(dtype).cuda(), C)
    res = ops.Act(torch.randn(B, H, W, C, generator=g).to(dtype).cuda(), C)
    st = ops.BNState(C, torch.device("cuda"))
    st.scale.copy_(torch.rand(C, generator=g) + 0.5); st.shift.copy_(torch.randn(C, generator=g) * 0.3)
    z = ops.bn_act(y, st, True, res=res, want_mask=True)
    V = 4 if dtype == torch.float32 else 8
    bits = z.mask.cpu().numpy()
    want = (z.t.float().cpu().numpy().reshape(B * H * W, C // V, V) > 0)
    got = ((bits[:, :, None] >> np.arange(V)[None, None, :]) & 1).astype(bool)
    assert np.array_equal(got, want)
    # whole step
    m = get_model(50, dtype)
    for p in m.prn.parameters():
        p.requires_grad = False
    m.train()
    Bn, S = 2, 96
    img = t(weightgen.gen_images(800, Bn, S, S)).cuda()
    heat, wgt = (t(a).cuda() for a in weightgen.gen_keypoint_gt(801, Bn, S // 4, S // 4))
    anno = t(weightgen.gen_boxes_gt(802, Bn, S)).cuda()
    bn_state = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}
    outs = []
    for on, defer in ((False, False), (True, False), (True, True)):
        m._engine.bn_mask_bits, m._engine.defer_shortcut_grad = on, defer
        m.load_state_dict(bn_state, strict=False)
        m._arena.ensure_grads()
        m._arena.grad_flat.zero_()
        pred, saved = m([img, "train_both"])
        loss, _ = poseNet.build_loss(saved, "train_both", heat, wgt, anno)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((loss.detach().clone(), m._arena.grad_flat.clone()))
    m._engine.bn_mask_bits = m._engine.defer_shortcut_grad = True
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "mask bits changed the gradients"
    # ... and with the identity-shortcut gradient dz * (z > 0) never materialised (conv1's dgrad epilogue adds it from dz and the bits)
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1]), "deferred shortcut gradient changed the gradients"
    report("ReLU mask bits (%s): bits == (z > 0); train step bit-identical with z replaced by its sign bits in backward and with the "
           "shortcut gradient folded into conv1's input-gradient launch" % str(dtype))


# ------------------------------------------------------------------------------------------------ virtual concatenation
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_virtual_concat_equals_the_materialised_concatenation(dtype):
    """posenet.py:311-315: conv2 over torch.cat((up8(q5), up4(q4), up2(q3), q2), 1).  The 512-channel tensor is never written:
    conv2's forward (shared-tile 3x3 kernel) and weight-gradient (LDS-DMA kernel) launches gather from the four members with the
    nearest-neighbour index in the DMA address.  Same k-order, same slices: outputs and weight / bias gradients are bit-identical to
    the materialised path —
    at kernel level on odd sizes, and for a whole training step (loss, every gradient)."""
    from multiposenet.pytorch_amd import ops
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from multiposenet.pytorch_amd import synthetic as weightgen
    dev = "cuda"
    g = torch.Generator().manual_seed(11)
    for (B, H, W) in ((2, 24, 40), (1, 8, 8), (3, 16, 24)):
        srcs = [ops.Act((torch.randn(B, H >> sh, W >> sh, 128, generator=g) * 0.5).to(dtype).to(dev), 128) for sh in (3, 2, 1, 0)]
        w = (torch.randn(256, 3, 3, 512, generator=g) * 0.02).to(dtype).to(dev)
        bias = torch.randn(256, generator=g).to(dev)
        cat = ops.Act(torch.empty(B, H, W, 512, dtype=dtype, device=dev), 512)
        for i, a in enumerate(srcs):
            ops.upsample_slice(a, cat, i * 128)
        y_ref, _ = ops.conv_forward(cat, w, 256, 3, 3, 1, 1, bias=bias, act=1)
        y_vir = ops.conv_forward_cat(srcs, H, W, w, 256, bias=bias, act=1)
        assert torch.equal(y_ref.t.view(torch.int16), y_vir.t.view(torch.int16)), "virtual-concat forward differs at %dx%dx%d" % (B, H, W)
        dy = ops.Act((torch.randn(B, H, W, 256, generator=g) * 0.1).to(dtype).to(dev), 256)
        dw_ref = torch.zeros(256 * 9 * 512, device=dev); db_ref = torch.zeros(256, device=dev)
        dw_vir = torch.zeros_like(dw_ref); db_vir = torch.zeros_like(db_ref)
        assert ops.conv_wgrad(cat, dy, dw_ref, 256, 3, 3, 1, 1, db=db_ref) and ops.conv_wgrad_cat(srcs, H, W, dy, dw_vir, 256, db=db_vir)
        assert torch.equal(dw_ref, dw_vir) and torch.equal(db_ref, db_vir), "virtual-concat weight gradient differs"
    if dtype != torch.bfloat16:
        return
    m = get_model(50, dtype)
    for p in m.prn.parameters():
        p.requires_grad = False
    m.train()
    Bn, S = 2, 128
    img = t(weightgen.gen_images(810, Bn, S, S)).cuda()
    heat, wgt = (t(a).cuda() for a in weightgen.gen_keypoint_gt(811, Bn, S // 4, S // 4))
    bn_state = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}
    outs = []
    classes = m._engine.conv2_classes
    m._engine.conv2_classes = False          # (round 6: the position-class form of conv2 has its own tests, tests/test_round6_gpu.py)
    for virt in (False, True):
        m._engine.virtual_concat = virt
        m.load_state_dict(bn_state, strict=False)
        m._arena.ensure_grads()
        m._arena.grad_flat.zero_()
        pred, saved = m([img, "keypoint_subnet"])
        loss, _ = poseNet.build_loss(saved, "keypoint_subnet", heat, wgt)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((pred.detach().clone(), loss.detach().clone(), m._arena.grad_flat.clone()))
    m._engine.virtual_concat, m._engine.conv2_classes = True, classes
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), "virtual concatenation changed the training step"
    report("virtual concatenation (conv2 of the keypoint head): forward / weight gradient / input gradient bit-identical to the "
           "materialised 512-channel tensor; whole keypoint step bit-identical")


# ------------------------------------------------------------------------------------------------ BASELINE config 5 end to end
def test_cfg5_chain_batch64_f16_equals_per_image_inference():
    """BASELINE config 5, the whole chain at its batch size: R101 `both` inference 640x640, 64 images, fp16, decode + NMS for every
    image, heat-map peaks, ONE batched PRN forward, candidate compaction on the device and the C++ matching
    (Tester.infer_images_batched).  The detector's output bias of the random-weight network is shifted so that people are
    "detected" (a few boxes per image above 0.5), the PRN has seeded weights.  Size-independent check at FULL size: the batched
    result dicts of six sampled images equal `Tester.infer_image` run on each alone (the reference semantics, tester.py:194-245) —
    boxes, scores and all 51 keypoint numbers."""
    from multiposenet.pytorch_amd.evaluate.tester import Tester, TestParams
    from multiposenet.pytorch_amd import synthetic as weightgen
    m = get_model(101, torch.float16)
    sd = weightgen.gen_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items() if k.startswith("prn.")}, seed=3, flavour="he")
    m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    m.eval()
    B, S = 64, 640
    rs = np.random.RandomState(5)
    sizes = [(S, S)] * 60 + [(480, 640), (640, 400), (333, 500), (512, 512)]          # mixed sizes share the 640x640 forward
    images = [rs.uniform(0, 255, (h, w, 3)).astype(np.float32) for h, w in sizes]
    tp = TestParams()
    tp.ckpt, tp.inp_size = None, S
    tester = Tester(m, tp)
    # shift the classification bias so that roughly 4 anchors per image survive NMS above 0.5
    with torch.no_grad():
        _, (cls, _, _) = m([torch.zeros(2, 3, S, S, device="cuda").normal_(), "detection_subnet"])
        s_ = cls.float().flatten().clamp(1e-6, 1 - 1e-6)
        q = torch.quantile(s_[torch.randperm(s_.numel(), device=s_.device)[:500000]], 1.0 - 40.0 / float(cls.shape[1]))
        old_bias = m.classificationModel.output.bias.data.clone()
        m.classificationModel.output.bias.data += float(-torch.log(q / (1 - q)))
    try:
        res = tester.infer_images_batched(images, ["f%d.jpg" % i for i in range(B)], list(range(B)), batch=64)
        assert len(res) == B
        nboxes = sum(len(r) for r in res)
        assert nboxes >= 32, "the calibrated detector should find people (%d boxes in 64 images)" % nboxes
        for i in (0, 7, 31, 60, 61, 62):
            single = tester.infer_image(images[i], "f%d.jpg" % i, i)
            assert len(single) == len(res[i]), "image %d: %d vs %d people" % (i, len(single), len(res[i]))
            for a, b in zip(single, res[i]):
                assert a["bbox"] == b["bbox"] and a["score"] == b["score"] and a["keypoints"] == b["keypoints"] and a["image_id"] == b["image_id"], i
    finally:
        m.classificationModel.output.bias.data.copy_(old_bias)
    report("cfg5 chain (R101 640x640 B=64 f16): %d people in 64 images; batched Tester results identical to per-image inference on 6 sampled images" % nboxes)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_one_pass_heatmap_loss_equals_the_export_loss_import_chain(dtype):
    """losses.mse_train_raw (the recorded step's loss: one launch over the internal tensors, up-sampling by indexing, coarse-level
    gradients summed in the launch) against the API path's chain on the same tensors — export_f32 (nearest up-sampling,
    posenet.py:243-257) -> mse_forward_raw / mse_backward_raw (posenet.py:376-387) -> import_grad: gradients bit-identical in every
    level and padding lane, loss values within f32 summation order, max / min exact."""
    from multiposenet.pytorch_amd import ops
    from multiposenet.pytorch_amd.network import losses
    B, H, W = 3, 24, 40
    g = torch.Generator().manual_seed(77)
    levels = []
    for s, C in ((0, 19), (1, 19), (2, 19), (3, 19), (0, 18)):
        x = torch.randn(B, H >> s, W >> s, 32, generator=g)
        x[..., C:] = 0
        levels.append(ops.Act(x.cuda(), C))
    heat = torch.rand(B, 18, H, W, generator=g).cuda()
    wgt = (torch.rand(B, 18, H, W, generator=g) > 0.2).float().cuda() * torch.rand(B, 18, H, W, generator=g).cuda()
    ones = torch.ones(2, device="cuda")
    assert losses.mse_train_supported(levels, heat)
    out1, g1 = losses.mse_train_raw(levels, heat, wgt, ones, dtype)
    exported = [ops.export_f32(a, a.C, H, W) for a in levels]
    pm = [losses._pixel_major(p) for p in exported]
    hn, wn = ops.nchw_to_nhwc_f32(heat), ops.nchw_to_nhwc_f32(wgt)
    out0 = losses.mse_forward_raw(pm, hn, wn)
    g0 = [ops.import_grad(gr, a, dtype) for gr, a in zip(losses.mse_backward_raw(pm, hn, wn, ones, [True] * 5), levels)]
    torch.cuda.synchronize()
    for j, (a, b) in enumerate(zip(g0, g1)):
        assert a.t.shape == b.t.shape and a.t.dtype == b.t.dtype
        assert torch.equal(a.t.view(torch.uint8), b.t.view(torch.uint8)), "level %d gradient differs" % j
        assert float(b.t.float().abs().max()) > 0
    o0, o1 = out0.cpu().numpy(), out1.cpu().numpy()
    assert np.allclose(o0[:6], o1[:6], rtol=2e-6, atol=0) and o0[6] == o1[6] and o0[7] == o1[7]
    # ragged geometry is refused by the support test (the recorded step then keeps the chain)
    odd = [ops.Act(torch.zeros(B, 25, 25, 32).cuda(), 19)] + levels[1:]
    assert not losses.mse_train_supported(odd, torch.zeros(B, 18, 25, 25).cuda())
    report("one-pass heat-map loss (%s, %dx%dx%d): 5 gradient tensors bit-identical to export -> loss -> import, losses %s vs %s"
           % (str(dtype).split(".")[1], B, H, W, np.round(o1[:5], 6).tolist(), np.round(o0[:5], 6).tolist()))


def test_pre_nms_top_k_cap_equals_the_oracle_on_the_best_k_candidates():
    """mpn_nms_batched_topk / ops.detect_batched(pre_nms_top_n=K): per image the result equals the oracle's NMS over the K
    best-scored candidates (stable: score ties by index) and the plain call when an image has fewer than K; the scratch is
    sized by K, not by the candidate count."""
    from multiposenet.pytorch_amd import ops
    from multiposenet.pytorch_amd._lib import call
    from oracle import nms_oracle
    rs = np.random.RandomState(5)
    counts = [0, 1, 40, 100, 101, 900, 2500]
    B, cap, K = len(counts), 2600, 100
    dets = np.zeros((B, cap, 5), np.float32)
    for b, n in enumerate(counts):
        xy = rs.uniform(0, 300, (n, 2)); wh = rs.uniform(8, 150, (n, 2))
        dets[b, :n] = np.concatenate([xy, xy + wh, np.round(rs.uniform(0.05, 1.0, (n, 1)), 2)], 1)      # rounded: many score ties
    d = torch.from_numpy(dets).cuda()
    cnt = torch.tensor(counts, dtype=torch.int32, device="cuda")
    nmax = max(counts)
    assert call("mpn_nms_batched_workspace_bytes", B, K) * 100 < call("mpn_nms_batched_workspace_bytes", B, nmax)
    for mode, name in ((0, "gpu"), (1, "cpu")):
        keep = torch.full((B, K), -1, dtype=torch.int64, device="cuda")
        num = torch.full((B,), -1, dtype=torch.int64, device="cuda")
        ws = torch.empty(call("mpn_nms_batched_workspace_bytes", B, K), dtype=torch.uint8, device="cuda")
        call("mpn_nms_batched_topk", ops.ptr(d), cap * 5, ops.ptr(cnt), B, nmax, K, 0.5, mode, ops.ptr(keep), K, ops.ptr(num), ops.ptr(ws), ops.stream_ptr())
        torch.cuda.synchronize()
        for b, n in enumerate(counts):
            got = keep[b, :int(num[b])].cpu().numpy()
            if n == 0:
                assert got.size == 0
                continue
            best = np.argsort(-dets[b, :n, 4], kind="stable")[:K]          # the sort's order: score descending, ties by index
            want = best[nms_oracle.nms(dets[b, best], 0.5, name)]
            assert np.array_equal(got, want), "image %d (n=%d, mode %s)" % (b, n, name)
            if n <= K:
                assert np.array_equal(got, nms_oracle.nms(dets[b, :n], 0.5, name))
    # through detect_batched: boxes + scores, candidates in their own [B, A] layout
    A = 3000
    boxes = torch.zeros(B, A, 4, device="cuda"); scores = torch.zeros(B, A, device="cuda")
    for b, n in enumerate(counts):
        boxes[b, :n] = d[b, :n, :4]; scores[b, :n] = d[b, :n, 4]
    ob, os_, kept = ops.detect_batched(boxes, scores, 0.04, 0.5, padded=True, pre_nms_top_n=K)
    assert ob.shape[1] <= K
    for b, n in enumerate(counts):
        if n == 0:
            assert kept[b] == 0
            continue
        best = np.argsort(-dets[b, :n, 4], kind="stable")[:K]
        want = best[nms_oracle.nms(dets[b, best], 0.5, "gpu")]
        assert kept[b] == len(want) and np.array_equal(ob[b, :kept[b]].cpu().numpy(), dets[b, want, :4])
        assert np.array_equal(os_[b, :kept[b]].cpu().numpy(), dets[b, want, 4])
    report("pre-NMS top-%d cap: %d images with %s candidates == oracle NMS over the best %d, both modes; scratch %d x smaller"
           % (K, B, counts, K, call("mpn_nms_batched_workspace_bytes", B, nmax) // call("mpn_nms_batched_workspace_bytes", B, K)))


