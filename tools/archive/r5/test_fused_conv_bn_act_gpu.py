"""Round 5: training-mode Conv -> BatchNorm (+ shortcut) -> ReLU as ONE launch (MpnConvParams.fz; network/fpn.py:28-34 with the module
in train mode, training/trainer.py:172-174).

The gates, in order of teeth:
  * kernel level, bit-exact: the fused launch's z / sign bits equal mpn_bn_act_forward applied to the fused launch's own y and
    coefficients (same arithmetic on the same values); y equals the plain conv's y; the coefficients equal mpn_bn_finalize_train's
    within double-precision summation order (<= 1 float32 ulp);
  * hand-off discipline: hundreds of back-to-back launches over mixed shapes with a bandwidth hog on a second stream (uneven
    load), every result compared, error latch zero, generation counters advance by one per launch and channel tile;
  * the tile cap: a layer above MPN_FZ_MAX_TILES is refused by mpn_conv_bn_fusable and runs the three-launch path;
  * whole step: fused on / off — losses and gradient arena agree like the in-launch finalize test's, fused runs are
    bit-reproducible, the recorded launch list replays them, and the BatchNorm launches are gone from the tape.
"""
import numpy as np
import pytest
import torch

from helpers import report, rng_normal, to_act, w_krsc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests selected but no GPU is visible"
    from multiposenet.pytorch_amd import _lib
    _lib.lib()


def _ulp_diff(a, b):
    ai = a.contiguous().view(torch.int32).long(); bi = b.contiguous().view(torch.int32).long()
    return int((ai - bi).abs().max())


def _bn_params(C, seed):
    g = torch.Generator().manual_seed(seed)
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    beta = (torch.randn(C, generator=g) * 0.2).cuda()
    return gamma, beta


def _run(ops, x, w, Cout, k, stride, pad, gamma, beta, relu, res, want_mask, fused):
    rm = torch.zeros(Cout, device="cuda"); rv = torch.ones(Cout, device="cuda")
    old = ops.FUSE_BN_ACT
    ops.FUSE_BN_ACT = fused
    try:
        y, st = ops.conv_forward(x, w, Cout, k, k, stride, pad, want_stats=True,
                                 bn_fin=(gamma, beta, rm, rv, 0.1, 1e-5), bn_tail=(relu, res, want_mask))
    finally:
        ops.FUSE_BN_ACT = old
    return y, st, rm, rv


# B, H, W, Cin, Cout, k, stride  (pixel tiles = ceil(B*Ho*Wo / 128))
FUSE_CASES = [
    (32, 30, 30, 1024, 256, 1, 1),      # layer3 conv1: 225 tiles, two 128-row channel tiles
    (32, 30, 30, 256, 256, 3, 1),       # layer3 conv2: shared-tile 3x3 kernel
    (32, 30, 30, 256, 1024, 1, 1),      # layer3 conv3: eight channel tiles, 1 800 workgroups (groups wait while later ones run)
    (32, 15, 15, 512, 512, 3, 1),       # layer4 conv2: 57 tiles
    (8, 30, 30, 512, 1024, 1, 2),       # down-sampling shortcut: stride 2, no ReLU
    (2, 9, 7, 64, 64, 3, 1),            # one tile, 64-row channel tile, ragged last tile
    (4, 30, 30, 64, 128, 1, 1),         # 29 tiles of a 64-row tile (few workgroups -> pick_tc 64)
    (32, 32, 32, 128, 128, 3, 1),       # exactly MPN_FZ_MAX_TILES = 256 pixel tiles
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_conv_bn_act_equals_the_three_launch_path(dtype):
    from multiposenet.pytorch_amd import ops
    worst = 0
    for ci, (B, H, W, Cin, Cout, k, stride) in enumerate(FUSE_CASES):
        pad = k // 2
        x = to_act(rng_normal(300 + ci, B, Cin, H, W), dtype)
        w = w_krsc(rng_normal(400 + ci, Cout, Cin, k, k) * (2.0 / (Cin * k * k)) ** 0.5, dtype)
        gamma, beta = _bn_params(Cout, 500 + ci)
        Ho, Wo = ops.conv_out_hw(H, W, k, k, stride, pad)
        for relu, with_res in ((True, False), (True, True), (False, False)):
            res = to_act(rng_normal(600 + ci, B, Cout, Ho, Wo), dtype) if with_res else None
            want_mask = relu and with_res
            y1, st1, rm1, rv1 = _run(ops, x, w, Cout, k, stride, pad, gamma, beta, relu, res, want_mask, True)
            assert isinstance(st1, ops.BNState) and st1.z is not None, "case %d was not fused" % ci
            z1 = st1.z
            # the plain conv + finalize launch + bn_act
            y0, st0, rm0, rv0 = _run(ops, x, w, Cout, k, stride, pad, gamma, beta, relu, res, want_mask, False)
            if not isinstance(st0, ops.BNState):
                st0 = ops.bn_finalize_train(st0, y0.P, gamma, beta, rm0, rv0, 0.1, 1e-5)
            assert st0.z is None
            torch.cuda.synchronize()
            assert not ops.fz_error()
            assert torch.equal(y1.t, y0.t), "case %d: y differs" % ci
            for name in ("mean", "invstd", "scale", "shift"):
                u = _ulp_diff(getattr(st1, name), getattr(st0, name))
                worst = max(worst, u)
                assert u <= 1 or torch.allclose(getattr(st1, name), getattr(st0, name), rtol=1e-6, atol=1e-7), (ci, name, u)
            assert torch.allclose(rm1, rm0, rtol=1e-6, atol=1e-8) and torch.allclose(rv1, rv0, rtol=1e-6, atol=1e-8)
            # normalise pass: bn_act on the fused launch's own y and coefficients must give the same bits
            z2 = ops.bn_act(y1, st1, relu, res=res, want_mask=want_mask)
            torch.cuda.synchronize()
            assert torch.equal(z1.t, z2.t), "case %d relu=%s res=%s: z differs from bn_act(y, coefficients)" % (ci, relu, with_res)
            if want_mask:
                assert z1.mask is not None and torch.equal(z1.mask, z2.mask), "case %d: sign bits differ" % ci
            else:
                assert z1.mask is None
    report("fused conv+BN+act (%s): %d shapes x 3 tails: y, z, sign bits bit-identical to conv + bn_act; coefficients within %d ulp of the finalize launch"
           % (str(dtype), len(FUSE_CASES), worst))


def test_layers_above_the_tile_cap_are_refused_and_take_the_three_launch_path():
    from multiposenet.pytorch_amd import ops
    dtype = torch.bfloat16
    B, H, W, Cin, Cout = 3, 128, 96, 64, 128           # 288 tiles > 256
    x = to_act(rng_normal(1, B, Cin, H, W), dtype)
    w = w_krsc(rng_normal(2, Cout, Cin, 1, 1) * 0.1, dtype)
    gamma, beta = _bn_params(Cout, 3)
    y, st, rm, rv = _run(ops, x, w, Cout, 1, 1, 0, gamma, beta, True, None, False, True)
    assert not (isinstance(st, ops.BNState) and st.z is not None), "a 288-tile layer must not take the co-resident group path"
    # f32 operands are not covered either
    xf = to_act(rng_normal(1, 2, 64, 16, 16), torch.float32)
    wf = w_krsc(rng_normal(2, 64, 64, 1, 1) * 0.1, torch.float32)
    g2, b2 = _bn_params(64, 4)
    y, st, rm, rv = _run(ops, xf, wf, 64, 1, 1, 0, g2, b2, True, None, False, True)
    assert not (isinstance(st, ops.BNState) and st.z is not None)
    torch.cuda.synchronize()


def test_hand_off_under_uneven_load_and_generation_counters():
    """300 fused launches over mixed shapes while a second stream streams 1 GB copies (uneven load, warm caches, recycled scratch
    addresses): every z compared with bn_act of the launch's own y / coefficients, latch clear, generations advance exactly."""
    from multiposenet.pytorch_amd import ops
    dtype = torch.bfloat16
    shapes = [(32, 30, 30, 256, 1024, 1), (32, 30, 30, 256, 256, 3), (32, 15, 15, 512, 2048, 1), (4, 30, 30, 64, 128, 1), (16, 30, 30, 1024, 256, 1)]
    data = []
    for i, (B, H, W, Cin, Cout, k) in enumerate(shapes):
        x = to_act(rng_normal(700 + i, B, Cin, H, W), dtype)
        w = w_krsc(rng_normal(710 + i, Cout, Cin, k, k) * (2.0 / (Cin * k * k)) ** 0.5, dtype)
        res = to_act(rng_normal(720 + i, B, Cout, H, W), dtype)
        data.append((x, w, res, _bn_params(Cout, 730 + i), Cout, k))
    sync, _ = ops.fz_state(torch.device("cuda", torch.cuda.current_device()), 2048)
    torch.cuda.synchronize()
    gen0 = sync[:64].clone()
    hog_a = torch.empty(256 << 20, dtype=torch.float32, device="cuda"); hog_b = torch.empty_like(hog_a)
    side = torch.cuda.Stream()
    expect = torch.zeros(64, dtype=torch.int64)
    bad = 0
    for it in range(300):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                hog_b.copy_(hog_a)
        x, w, res, (gamma, beta), Cout, k = data[it % len(data)]
        y, st, rm, rv = _run(ops, x, w, Cout, k, 1, k // 2, gamma, beta, True, res, True, True)
        assert st.z is not None
        tc = 128 if Cout >= 128 and (x.P + 127) // 128 * (Cout // 128) >= 200 else 64
        expect[: Cout // tc] += 1
        z2 = ops.bn_act(y, st, True, res=res, want_mask=True)
        bad += int(not torch.equal(st.z.t, z2.t)) + int(not torch.equal(st.z.mask, z2.mask))
    torch.cuda.synchronize()
    assert bad == 0, "%d of 300 launches produced a z that differs from bn_act(y, coefficients)" % bad
    assert not ops.fz_error(), "a fused BatchNorm launch gave up waiting"
    adv = (sync[:64].cpu().long() - gen0.cpu().long())
    assert int(adv.sum()) == int(expect.sum()) and int(adv.max()) <= 300, (adv.tolist(), expect.tolist())
    report("fused conv+BN+act hand-off: 300 launches beside a copy stream, 0 mismatches, %d generation releases, latch clear" % int(adv.sum()))


def _train_setup(layers, dtype, B, S, seed=50):
    from test_round2_gpu import _train_setup as ts
    return ts(layers, dtype, B, S, seed)


@pytest.mark.parametrize("mode", ["train", "frozen_affine"])
def test_training_step_fused_bn_on_off(mode):
    """R101 train_both step at 240x240 B=8 (layer3 = 15x15 -> 15 tiles, layer2 = 30x30 -> 57 tiles, layer1 = 60x60 -> 225 tiles: every
    Bottleneck BatchNorm is eligible).  Fused on / off: loss and gradient arena agree to bf16 chaos (the coefficients differ in the
    last ulp of a double-precision sum now and then; the per-kernel test above is the bit-exact gate), two fused runs are
    bit-identical, the BatchNorm forward launches are gone."""
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from multiposenet.pytorch_amd import _lib
    import multiposenet.pytorch_amd.ops as ops_mod
    dtype = torch.bfloat16
    m, inputs, gts = _train_setup(101, dtype, 8, 240, seed=171)
    if mode == "frozen_affine":
        m.freeze_bn()
    orig = _lib.call
    calls = []

    def counting(name, *a):
        if name in ("mpn_bn_finalize_train", "mpn_bn_act_forward"):
            calls[-1][name] = calls[-1].get(name, 0) + 1
        return orig(name, *a)
    ops_mod.call = counting
    res = []
    bn0 = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}
    old = ops_mod.FUSE_BN_ACT
    try:
        for fused in (False, True, True):
            calls.append({})
            ops_mod.FUSE_BN_ACT = fused
            m.load_state_dict(bn0, strict=False)
            m._arena.ensure_grads()
            m._arena.grad_flat.zero_()
            pred, saved = m(*inputs)
            loss, log = poseNet.build_loss(saved, *gts)
            loss.backward()
            torch.cuda.synchronize()
            res.append((loss.detach().clone(), m._arena.grad_flat.clone(), pred.detach().clone(),
                        {k: v.clone() for k, v in m.state_dict().items() if "running_" in k}))
    finally:
        ops_mod.call = orig
        ops_mod.FUSE_BN_ACT = old
        m.train()
    assert not ops_mod.fz_error()
    (l0, g0, p0, r0), (l1, g1, p1, r1), (l2, g2, p2, r2) = res
    assert torch.equal(g1, g2) and torch.equal(p1, p2) and float(l1) == float(l2), "fused step is not bit-reproducible"
    rel_l = abs(float(l0) - float(l1)) / abs(float(l0))
    rel = float((g0 - g1).norm() / g0.norm())
    relp = float((p0.float() - p1.float()).norm() / p0.float().norm())
    rs = max(float((r0[k].float() - r1[k].float()).abs().max() / r0[k].float().abs().max().clamp_min(1e-12)) for k in r0)
    report("fused conv+BN+act, whole step (%s): bn_act launches %d -> %d, finalize launches %d -> %d; loss rel %.1e, heat-map rel-L2 %.2e, "
           "gradient arena rel-L2 %.2e, running statistics max rel %.1e; fused twice: bit-identical"
           % (mode, calls[0].get("mpn_bn_act_forward", 0), calls[1].get("mpn_bn_act_forward", 0),
              calls[0].get("mpn_bn_finalize_train", 0), calls[1].get("mpn_bn_finalize_train", 0), rel_l, relp, rel, rs))
    if mode == "train":
        assert calls[1].get("mpn_bn_act_forward", 0) <= 5 and calls[1].get("mpn_bn_finalize_train", 0) <= 2, calls[1]
        assert rel_l <= 1e-3 and rs <= 1e-5 and rel <= 1e-2 and relp <= 1e-2
    else:       # frozen statistics: nothing to fuse, nothing may change
        assert calls[0] == calls[1] and torch.equal(g0, g1)


def test_recorded_step_replays_the_fused_launches():
    """Three optimizer steps through the recorded launch list == three eager steps, bit for bit, with the fused launches in the list
    (the generation words make the replayed launches self-synchronising: nothing is reset between replays)."""
    from multiposenet.pytorch_amd.optim import FusedAdam
    from multiposenet.pytorch_amd.replay import ReplayedTrainStep
    from multiposenet.pytorch_amd.training.batch_processor import train_step
    import multiposenet.pytorch_amd.ops as ops_mod
    assert ops_mod.FUSE_BN_ACT
    m, inputs, gts = _train_setup(50, torch.bfloat16, 4, 128, seed=181)
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    outs = []
    for recorded in (False, True):
        m.load_state_dict(state0)
        opt = FusedAdam(m, lr=1e-3)
        stepper = ReplayedTrainStep(m, opt) if recorded else None
        losses = []
        for i in range(4):
            loss, log = stepper(inputs, gts) if recorded else train_step(m, opt, inputs, gts)
            losses.append(float(loss))
        torch.cuda.synchronize()
        outs.append((losses, m._arena.flat.clone()))
    assert not ops_mod.fz_error()
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1]), (outs[0][0], outs[1][0])
