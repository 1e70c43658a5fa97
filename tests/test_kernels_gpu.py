"""Kernel-level parity: every HIP kernel, called through the C-ABI, vs the CPU oracle arithmetic
(torch CPU fp32 ops — the same library arithmetic the reference itself calls — and oracle/*.py|c).

Tolerances: f32 kernels 2e-4 of the reference tensor's max-abs (exact-fp32 MFMA, summation order
differs from oneDNN); bf16 kernels 2e-2 of max-abs against a CPU reference fed the bf16-rounded
operands.  Integer results (NMS indices, max-pool routing) are compared exactly.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import check_close, from_act, gold, rnd, rng_normal, report, to_act, w_krsc

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests selected but no GPU is visible"
    from multiposenet.pytorch_amd import _lib
    _lib.lib()   # fail loudly if the HIP library is missing


def _ops():
    from multiposenet.pytorch_amd import ops
    return ops


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 17, 13, 64, 64, 3, 1, 1),
    (2, 16, 16, 128, 256, 1, 1, 0),
    (1, 15, 15, 256, 19, 1, 1, 0),
    (2, 9, 11, 64, 128, 3, 2, 1),
    (2, 8, 8, 512, 36, 3, 1, 1),
    (1, 32, 32, 32, 64, 1, 1, 0),
    (2, 16, 16, 256, 512, 1, 2, 0),
    (3, 4, 4, 256, 9, 3, 1, 1),
    (1, 30, 30, 128, 128, 3, 1, 1),
    (2, 15, 15, 2048, 256, 3, 2, 1),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward(case, dtype):
    ops = _ops()
    B, H, W, Cin, Cout, k, stride, pad = case
    x = rnd(dtype, rng_normal(1, B, Cin, H, W))
    w = rnd(dtype, rng_normal(2, Cout, Cin, k, k) / math.sqrt(Cin * k * k))
    bias = rng_normal(3, Cout)
    ref = F.conv2d(x, w, bias, stride=stride, padding=pad)
    out, _ = ops.conv_forward(to_act(x, dtype), w_krsc(w, dtype), Cout, k, k, stride, pad, bias=bias.cuda())
    torch.cuda.synchronize()
    check_close("conv_fwd %s %s" % (case, dtype), from_act(out), ref, dtype)
    # pad lanes must be zero
    assert out.t[..., Cout:].float().abs().max().item() == 0.0 if out.Cs > Cout else True
    # relu epilogue + f32 output from a bf16 kernel
    out2, _ = ops.conv_forward(to_act(x, dtype), w_krsc(w, dtype), Cout, k, k, stride, pad, bias=bias.cuda(), act=1, out_f32=True)
    assert out2.t.dtype == torch.float32
    check_close("conv_fwd+relu+f32out %s %s" % (case, dtype), from_act(out2), F.relu(ref), dtype,
                factor=1.0 if dtype == torch.float32 else 0.5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_epilogue_residual_and_stats(dtype):
    ops = _ops()
    B, H, W, Cin, Cout = 2, 12, 10, 64, 256
    x = rnd(dtype, rng_normal(4, B, Cin, H, W))
    w = rnd(dtype, rng_normal(5, Cout, Cin, 1, 1) / 8.0)
    bias = rng_normal(6, Cout)
    res_same = rnd(dtype, rng_normal(7, B, Cout, H, W))
    res_up = rnd(dtype, rng_normal(8, B, Cout, H // 2, W // 2))
    ref = F.conv2d(x, w, bias)
    o1, _ = ops.conv_forward(to_act(x, dtype), w_krsc(w, dtype), Cout, 1, 1, 1, 0, bias=bias.cuda(), res=to_act(res_same, dtype), res_mode=1)
    check_close("conv+res_same %s" % dtype, from_act(o1), ref + res_same, dtype)
    o2, _ = ops.conv_forward(to_act(x, dtype), w_krsc(w, dtype), Cout, 1, 1, 1, 0, bias=bias.cuda(), res=to_act(res_up, dtype), res_mode=2)
    check_close("conv+res_up %s" % dtype, from_act(o2), ref + F.interpolate(res_up, size=(H, W), mode="nearest"), dtype)
    # stats (no bias): per-channel sum / sumsq of the stored output
    o3, st = ops.conv_forward(to_act(x, dtype), w_krsc(w, dtype), Cout, 1, 1, 1, 0, want_stats=True)
    y = from_act(o3)
    s = st.sum(0).cpu()
    check_close("conv stats sum %s" % dtype, s[:, 0], y.sum((0, 2, 3)), torch.float32, scale=y.abs().sum((0, 2, 3)).max().item())
    check_close("conv stats sumsq %s" % dtype, s[:, 1], (y * y).sum((0, 2, 3)), torch.float32)
    # accumulate
    o4 = to_act(res_same, dtype)
    ops.conv_forward(to_act(x, dtype), w_krsc(w, dtype), Cout, 1, 1, 1, 0, out=o4, accumulate=True)
    check_close("conv accumulate %s" % dtype, from_act(o4), F.conv2d(x, w) + res_same, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_stem_conv_packed(dtype):
    """7x7/s2/p3 3->64 stem (fpn.py:42,99) through the NHWC4 zero-bordered packing."""
    ops = _ops()
    from multiposenet.pytorch_amd import _lib
    B, H, W = 2, 32, 48
    x = rnd(dtype, rng_normal(9, B, 3, H, W))
    w = rnd(dtype, rng_normal(10, 64, 3, 7, 7) / 12.0)
    ref = F.conv2d(x, w, None, stride=2, padding=3)
    Hp, Wp = H + 6, W + 8
    xg = x.cuda()
    packed = torch.empty((B, Hp, Wp, 4), dtype=dtype, device="cuda")
    _lib.call("mpn_stem_pack_image", ops.ptr(xg), xg.stride(0), xg.stride(1), xg.stride(2), xg.stride(3), ops.ptr(packed),
              B, H, W, ops.dtype_code(dtype), ops.stream_ptr())
    wk = w.permute(0, 2, 3, 1).contiguous().cuda()          # master layout [64][7][7][3] f32
    wp = torch.empty((64, 7, 32), dtype=dtype, device="cuda")
    _lib.call("mpn_stem_pack_weight", ops.ptr(wk), ops.ptr(wp), 64, ops.dtype_code(dtype), ops.stream_ptr())
    xa = ops.Act(packed, 4)
    Ho, Wo = H // 2, W // 2
    out, _ = ops.conv_forward(xa, wp, 64, 7, 1, 2, 0, cin=32, x_geom=(Hp, Wp, Hp * Wp * 4, Wp * 4, 4), out_hw=(Ho, Wo))
    check_close("stem conv %s" % dtype, from_act(out), ref, dtype)
    # wgrad through the same packing
    dy = rnd(dtype, rng_normal(11, B, 64, Ho, Wo))
    xr = x.clone().requires_grad_(False)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, stride=2, padding=3).backward(dy)
    dwp = torch.zeros((64, 7, 32), dtype=torch.float32, device="cuda")
    ops.conv_wgrad(xa, to_act(dy, dtype), dwp, 64, 7, 1, 2, 0, cin=32, x_geom=(Hp, Wp, Hp * Wp * 4, Wp * 4, 4))
    dw = torch.zeros((64, 7, 7, 3), dtype=torch.float32, device="cuda")
    _lib.call("mpn_stem_unpack_wgrad", ops.ptr(dwp), ops.ptr(dw), 64, ops.stream_ptr())
    check_close("stem wgrad %s" % dtype, dw.cpu().permute(0, 3, 1, 2), wr.grad, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES[:9])
def test_conv_dgrad_wgrad(case, dtype):
    ops = _ops()
    B, H, W, Cin, Cout, k, stride, pad = case
    x = rnd(dtype, rng_normal(21, B, Cin, H, W)).requires_grad_(True)
    w = rnd(dtype, rng_normal(22, Cout, Cin, k, k) / math.sqrt(Cin * k * k)).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=stride, padding=pad)
    dy = rnd(dtype, rng_normal(23, *y.shape))
    y.backward(dy)
    Ho, Wo = y.shape[2:]
    kc = 32 if dtype == torch.bfloat16 else 16
    cout_pad = (Cout + kc - 1) // kc * kc
    wm = w.detach().permute(0, 2, 3, 1).contiguous().cuda()      # [Cout][R][S][Cin] f32 master
    wt = torch.empty((Cin, k, k, cout_pad), dtype=dtype, device="cuda")
    ops.weight_transpose(wm, wt, Cout, k * k, Cin, cout_pad)
    dya = to_act(dy, dtype)
    dx, _ = ops.conv_forward(dya, wt, Cin, k, k, stride, pad, mode=1, out_hw=(H, W), cin=cout_pad)
    check_close("dgrad %s %s" % (case, dtype), from_act(dx), x.grad, dtype)
    dw = torch.zeros((Cout, k, k, Cin), dtype=torch.float32, device="cuda")
    ops.conv_wgrad(to_act(x.detach(), dtype), dya, dw, Cout, k, k, stride, pad)
    check_close("wgrad %s %s" % (case, dtype), dw.cpu().permute(0, 3, 1, 2), w.grad, dtype)
    # accumulate semantics: a second call doubles the result
    ops.conv_wgrad(to_act(x.detach(), dtype), dya, dw, Cout, k, k, stride, pad)
    check_close("wgrad x2 %s %s" % (case, dtype), dw.cpu().permute(0, 3, 1, 2), 2 * w.grad, dtype)
    db = torch.zeros((Cout,), dtype=torch.float32, device="cuda")
    ops.bias_grad(dya, db, Cout)
    check_close("bias grad %s %s" % (case, dtype), db.cpu(), dy.sum((0, 2, 3)), torch.float32, factor=5)


def test_wgrad_large_split():
    """Many pixels -> chunks > 1 path (workspace partials + deterministic reduce), bf16 and f32."""
    ops = _ops()
    for dtype in DTYPES:
        B, H, W, Cin, Cout = 4, 60, 60, 64, 64
        x = rnd(dtype, rng_normal(31, B, Cin, H, W)).requires_grad_(False)
        w = rnd(dtype, rng_normal(32, Cout, Cin, 3, 3) / 24.0).requires_grad_(True)
        y = F.conv2d(x, w, None, padding=1)
        dy = rnd(dtype, rng_normal(33, *y.shape))
        y.backward(dy)
        dw = torch.zeros((Cout, 3, 3, Cin), dtype=torch.float32, device="cuda")
        ops.conv_wgrad(to_act(x, dtype), to_act(dy, dtype), dw, Cout, 3, 3, 1, 1)
        check_close("wgrad split %s" % dtype, dw.cpu().permute(0, 3, 1, 2), w.grad, dtype)
        dw2 = torch.zeros_like(dw)
        ops.conv_wgrad(to_act(x, dtype), to_act(dy, dtype), dw2, Cout, 3, 3, 1, 1)
        assert torch.equal(dw, dw2), "wgrad must be run-to-run deterministic"


def test_wgrad_partials_entry_point_matches_fused_call():
    """mpn_conv_wgrad_partials + mpn_reduce_partials (the profiler-bracketed path ops.conv_wgrad takes when kernel
    events are on) must produce exactly what the single mpn_conv_wgrad call does; every LDS-DMA tile variant."""
    ops = _ops()
    dtype = torch.bfloat16
    for (Cin, Cout) in ((128, 256), (64, 64), (256, 36), (32, 128)):
        B, H, W = 4, 40, 40
        x = rnd(dtype, rng_normal(34, B, Cin, H, W))
        dy = rnd(dtype, rng_normal(35, B, Cout, H, W))
        dw = torch.zeros((Cout, 1, 1, Cin), dtype=torch.float32, device="cuda")
        ops.conv_wgrad(to_act(x, dtype), to_act(dy, dtype), dw, Cout, 1, 1, 1, 0)
        dw2 = torch.zeros_like(dw)
        ops.KERNEL_EVENTS.enable()
        try:
            ops.conv_wgrad(to_act(x, dtype), to_act(dy, dtype), dw2, Cout, 1, 1, 1, 0)
            names = [r[0] for r in ops.KERNEL_EVENTS.rec]
        finally:
            ops.KERNEL_EVENTS.disable()
        assert torch.equal(dw, dw2)
        assert names and names[0].startswith("conv_wgrad_"), names
        # bias gradient fused into the same kernel (ones-vector MFMA): sliced and bracketed paths, vs the column sums
        for events in (False, True):
            dw3 = torch.zeros_like(dw)
            db = torch.full((Cout,), 0.5, dtype=torch.float32, device="cuda")
            if events:
                ops.KERNEL_EVENTS.enable()
            try:
                assert ops.conv_wgrad(to_act(x, dtype), to_act(dy, dtype), dw3, Cout, 1, 1, 1, 0, db=db) is True
            finally:
                ops.KERNEL_EVENTS.disable()
            assert torch.equal(dw3, dw)
            check_close("fused bias grad %dx%d" % (Cin, Cout), db.cpu() - 0.5, dy.float().cpu().sum((0, 2, 3)), torch.float32, factor=5)
        ref = torch.einsum("bohw,bihw->oi", dy.float().cpu(), x.float().cpu())
        check_close("wgrad dma tile %dx%d" % (Cin, Cout), dw.cpu().view(Cout, Cin), ref, dtype)


def test_weight_transpose_batched_matches_per_layer():
    """One launch over a table of layers == mpn_weight_transpose per layer (bf16 and f32 operands)."""
    ops = _ops()
    from multiposenet.pytorch_amd import _lib
    geoms = [(64, 9, 64), (19, 1, 256), (256, 1, 1024), (36, 9, 256), (128, 49, 3 * 8)]       # (Cout, RS, Cin)
    for dtype in DTYPES:
        kc = 32 if dtype == torch.bfloat16 else 16
        arena, rows, off_src, off_dst, blk = [], [], 0, 0, 0
        for O, RS, I in geoms:
            w = rng_normal(50 + O, O * RS * I)
            arena.append(w)
            opad = (O + kc - 1) // kc * kc
            gx, gy = (I + 31) // 32, (opad + 31) // 32
            rows.append([off_src, off_dst, O, RS, I, opad, blk, gx])
            off_src += O * RS * I
            off_dst += (I * RS * opad + 63) // 64 * 64
            blk += gx * gy * RS
        flat = torch.cat(arena).cuda()
        table = torch.tensor(rows, dtype=torch.int64, device="cuda")
        dst = torch.full((off_dst,), 7.0, dtype=dtype, device="cuda")
        _lib.call("mpn_weight_transpose_batched", ops.ptr(flat), ops.ptr(dst), ops.ptr(table), len(rows), blk,
                  ops.dtype_code(dtype), ops.stream_ptr())
        for (O, RS, I), row in zip(geoms, rows):
            opad = row[5]
            one = torch.empty((I, RS, opad), dtype=dtype, device="cuda")
            ops.weight_transpose(flat[row[0]: row[0] + O * RS * I], one, O, RS, I, opad)
            got = dst[row[1]: row[1] + I * RS * opad].view(I, RS, opad)
            assert torch.equal(got, one), "batched transpose differs for layer %s" % ((O, RS, I),)
            ref = flat[row[0]: row[0] + O * RS * I].view(O, RS, I).permute(2, 1, 0).to(dtype)
            assert torch.equal(got[:, :, :O], ref) and bool((got[:, :, O:] == 0).all())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [64, 256, 2048])
def test_batchnorm_train_and_eval(dtype, C):
    ops = _ops()
    B, H, W = 2, 9, 7
    y = rnd(dtype, rng_normal(41, B, C, H, W) * 1.5 + 0.3)
    res = rnd(dtype, rng_normal(42, B, C, H, W))
    gamma = torch.rand(C, generator=torch.Generator().manual_seed(43)) + 0.5
    beta = rng_normal(44, C) * 0.1
    rm0, rv0 = rng_normal(45, C) * 0.1, torch.rand(C, generator=torch.Generator().manual_seed(46)) + 0.5
    for relu, use_res in ((True, False), (True, True), (False, False)):
        # ---- train mode: stats from a 1x1 identity-free path: use conv epilogue stats on y itself
        yl = y.clone().requires_grad_(True)
        g_ = gamma.clone().requires_grad_(True)
        b_ = beta.clone().requires_grad_(True)
        rm, rv = rm0.clone(), rv0.clone()
        z = F.batch_norm(yl, rm, rv, g_, b_, training=True, momentum=0.1, eps=1e-5)
        if use_res:
            rl = res.clone().requires_grad_(True)
            z = z + rl
        if relu:
            z = F.relu(z)
        dz = rnd(dtype, rng_normal(47, B, C, H, W))
        z.backward(dz)
        ya = to_act(y, dtype)
        yv = from_act(ya)
        stats = torch.stack([yv.sum((0, 2, 3)), (yv * yv).sum((0, 2, 3))], 1).unsqueeze(0).contiguous().cuda()
        rmg, rvg = rm0.clone().cuda(), rv0.clone().cuda()
        st = ops.bn_finalize_train(stats, B * H * W, gamma.cuda(), beta.cuda(), rmg, rvg)
        za = ops.bn_act(ya, st, relu, res=to_act(res, dtype) if use_res else None)
        check_close("bn train fwd C=%d relu=%s res=%s %s" % (C, relu, use_res, dtype), from_act(za), z.detach(), dtype)
        check_close("bn running_mean", rmg.cpu(), rm, torch.float32)
        check_close("bn running_var", rvg.cpu(), rv, torch.float32)
        dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        dres = ops.Act(torch.empty_like(ya.t), C) if use_res else None
        dy = ops.bn_backward(to_act(dz, dtype), za, ya, st, gamma.cuda(), relu, True, dgamma=dg, dbeta=db, dres=dres)
        f = 3.0 if dtype == torch.bfloat16 else 5.0
        check_close("bn train dy C=%d relu=%s res=%s %s" % (C, relu, use_res, dtype), from_act(dy), yl.grad, dtype, factor=f)
        check_close("bn dgamma", dg.cpu(), g_.grad, dtype, factor=f)
        check_close("bn dbeta", db.cpu(), b_.grad, dtype, factor=f)
        if use_res:
            check_close("bn dres", from_act(dres), rl.grad, dtype)
        if relu and not use_res:
            # mask recomputed from y (no z read) must reproduce the z-based result bit for bit
            dg2 = torch.zeros(C, device="cuda"); db2 = torch.zeros(C, device="cuda")
            dy2 = ops.bn_backward(to_act(dz, dtype), za, ya, st, gamma.cuda(), relu, True, dgamma=dg2, dbeta=db2, remask=True)
            assert torch.equal(dy2.t, dy.t) and torch.equal(dg2, dg) and torch.equal(db2, db), "remask path differs from z-mask path"
    # ---- eval / frozen
    yl = y.clone().requires_grad_(True)
    g_ = gamma.clone().requires_grad_(True); b_ = beta.clone().requires_grad_(True)
    z = F.relu(F.batch_norm(yl, rm0.clone(), rv0.clone(), g_, b_, training=False, eps=1e-5))
    dz = rnd(dtype, rng_normal(48, B, C, H, W))
    z.backward(dz)
    st = ops.bn_finalize_eval(gamma.cuda(), beta.cuda(), rm0.cuda(), rv0.cuda())
    ya = to_act(y, dtype)
    za = ops.bn_act(ya, st, True)
    check_close("bn eval fwd C=%d %s" % (C, dtype), from_act(za), z.detach(), dtype)
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    dy = ops.bn_backward(to_act(dz, dtype), za, ya, st, gamma.cuda(), True, False, dgamma=dg, dbeta=db)
    check_close("bn eval dy", from_act(dy), yl.grad, dtype)
    check_close("bn eval dgamma", dg.cpu(), g_.grad, dtype, factor=3)
    check_close("bn eval dbeta", db.cpu(), b_.grad, dtype, factor=3)
    dg2 = torch.zeros(C, device="cuda"); db2 = torch.zeros(C, device="cuda")
    dy2 = ops.bn_backward(to_act(dz, dtype), za, ya, st, gamma.cuda(), True, False, dgamma=dg2, dbeta=db2, remask=True)
    assert torch.equal(dy2.t, dy.t) and torch.equal(dg2, dg) and torch.equal(db2, db), "frozen-BN remask path differs"


@pytest.mark.parametrize("dtype", DTYPES)
def test_maxpool_and_resample(dtype):
    ops = _ops()
    B, C, H, W = 2, 64, 13, 18
    x = rnd(dtype, F.relu(rng_normal(51, B, C, H, W))).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    dy = rnd(dtype, rng_normal(52, *y.shape))
    y.backward(dy)
    xa = to_act(x.detach(), dtype)
    ya, idx = ops.maxpool_forward(xa, needs_grad=True)
    assert torch.equal(from_act(ya), y.detach()), "max-pool forward must be exact"
    dx = ops.maxpool_backward(to_act(dy, dtype), idx, xa)
    # positive maxima are unique with probability 1; all-zero windows route to the first tap in both
    check_close("maxpool bwd %s" % dtype, from_act(dx), x.grad, dtype)
    # nearest upsample backward (2x and a non-integer ratio)
    for (hc, wc, hf, wf) in ((5, 6, 10, 12), (4, 3, 7, 5)):
        c = rnd(dtype, rng_normal(53, B, C, hc, wc)).requires_grad_(True)
        up = F.interpolate(c, size=(hf, wf), mode="nearest")
        g = rnd(dtype, rng_normal(54, *up.shape))
        up.backward(g)
        dc = ops.Act(torch.zeros((B, hc, wc, C), dtype=dtype, device="cuda"), C)
        ops.upsample_backward(to_act(g, dtype), dc, False)
        check_close("upsample bwd %s %s" % ((hc, wc, hf, wf), dtype), from_act(dc), c.grad, dtype)
    # concat slice forward/backward (posenet.py:311-315)
    s = rnd(dtype, rng_normal(55, B, 128, 4, 5))
    dst = ops.Act(torch.zeros((B, 16, 20, 512), dtype=dtype, device="cuda"), 512)
    ops.upsample_slice(to_act(s, dtype), dst, 128)
    assert torch.equal(from_act(dst)[:, 128:256], F.interpolate(s, scale_factor=4, mode="nearest"))
    gd = rnd(dtype, rng_normal(56, B, 512, 16, 20))
    ds = ops.Act(torch.empty((B, 4, 5, 128), dtype=dtype, device="cuda"), 128)
    ops.upsample_slice_backward(to_act(gd, dtype), ds, 128)
    check_close("slice bwd %s" % dtype, from_act(ds), F.avg_pool2d(gd[:, 128:256], 4) * 16, dtype)
    # API edge export/import
    src = rnd(dtype, rng_normal(57, B, 19, 6, 7))
    e = ops.export_f32(to_act(src, dtype), 19, 12, 14)
    assert e.shape == (B, 19, 12, 14)
    assert torch.equal(e.cpu(), F.interpolate(src, scale_factor=2, mode="nearest"))
    gi = rng_normal(58, B, 19, 12, 14)
    d = ops.import_grad(gi.cuda().contiguous(memory_format=torch.channels_last), to_act(src, dtype), dtype)
    check_close("import grad %s" % dtype, from_act(d), F.avg_pool2d(gi, 2) * 4, dtype)
    assert d.t[..., 19:].float().abs().max().item() == 0.0


def test_mse_heatmap_loss():
    from multiposenet.pytorch_amd.network import losses as L
    from oracle import posenet_oracle as po
    B, h, w = 2, 12, 10
    preds = [rng_normal(60 + j, B, 19 if j < 4 else 18, h, w).requires_grad_(True) for j in range(5)]
    heat = torch.rand(B, 18, h, w, generator=torch.Generator().manual_seed(66))
    wgt = (torch.rand(B, 18, h, w, generator=torch.Generator().manual_seed(67)) < 0.9).float()
    total, log = po.keypoint_loss(preds, heat, wgt)
    (total * 1.7).backward()
    gp = [p.detach().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) for p in preds]
    loss, glog = L.build_keypoint_loss(gp, heat.cuda(), wgt.cuda())
    (loss * 1.7).backward()
    check_close("mse total", loss.detach().cpu(), total.detach(), torch.float32)
    for k in log:
        assert abs(glog[k] - log[k]) <= 2e-4 * max(1.0, abs(log[k])), (k, glog[k], log[k])
    for j in range(5):
        check_close("mse dpred%d" % j, gp[j].grad.cpu(), preds[j].grad, torch.float32)


def test_focal_loss_vs_reference_golden():
    from multiposenet.pytorch_amd.network import losses as L
    g = gold("g4_focal.npz")
    cls = torch.from_numpy(g["cls"]).cuda().requires_grad_(True)
    reg = torch.from_numpy(g["reg"]).cuda().requires_grad_(True)
    anchors = torch.from_numpy(g["anchors"]).cuda()
    anno = torch.from_numpy(g["anno"]).cuda()
    closs, rloss = L.FocalLoss()(cls, reg, anchors, anno)
    (closs.mean() + rloss.mean()).backward()
    assert abs(closs.item() - g["loss"][0]) <= 2e-4 * max(1.0, g["loss"][0]), (closs.item(), g["loss"][0])
    assert abs(rloss.item() - g["loss"][1]) <= 2e-4 * max(1.0, g["loss"][1]), (rloss.item(), g["loss"][1])
    check_close("focal dcls", cls.grad.cpu(), torch.from_numpy(g["dcls"]), torch.float32)
    check_close("focal dreg", reg.grad.cpu(), torch.from_numpy(g["dreg"]), torch.float32)


def test_box_decode_clip_vs_reference_golden():
    ops = _ops()
    g = gold("g6_decode.npz")
    boxes = ops.box_decode_clip(torch.from_numpy(g["anchors"]).cuda(), torch.from_numpy(g["deltas"]).cuda(), 96.0, 128.0)
    check_close("decode+clip", boxes.cpu(), torch.from_numpy(g["boxes"]), torch.float32, factor=0.05)


def test_nms_bit_exact_goldens_and_oracle():
    from multiposenet.pytorch_amd.lib.nms.pth_nms import pth_nms
    from oracle import nms_oracle
    g = gold("g5_nms.npz")
    for n in (1, 2, 63, 64, 65, 128, 1000, 4097):
        d = torch.from_numpy(g["dets_%d" % n])
        keep = pth_nms(d.cuda(), 0.5)
        assert keep.dtype == torch.int64
        assert np.array_equal(keep.cpu().numpy(), g["keep_gpu_%d" % n]), "NMS (gpu mode) differs at n=%d" % n
        keep_c = pth_nms(d.cuda(), 0.5, mode="cpu")
        assert np.array_equal(keep_c.cpu().numpy(), g["keep_cpu_%d" % n]), "NMS (cpu mode) differs at n=%d" % n
    # random + ties + empty
    rs = np.random.RandomState(5)
    for n in (0, 7, 300, 2500, 9000):
        xy = rs.uniform(0, 500, (n, 2)); wh = rs.uniform(4, 200, (n, 2))
        sc = np.round(rs.uniform(0, 1, (n, 1)), 2)            # many exact score ties
        d = np.concatenate([xy, xy + wh, sc], 1).astype(np.float32)
        for thr in (0.3, 0.5, 0.7):
            k = pth_nms(torch.from_numpy(d).cuda(), thr).cpu().numpy()
            ref = nms_oracle.nms(d, thr, "gpu") if n else np.zeros(0, np.int64)
            assert np.array_equal(k, ref), "NMS differs n=%d thr=%s" % (n, thr)
    report("nms: all index lists bit-exact")


def test_adam_matches_torch():
    ops = _ops()
    from multiposenet.pytorch_amd import _lib
    n = 10007
    p0 = rng_normal(70, n); steps = 4
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-3, weight_decay=0.0)
    grads = [rng_normal(71 + i, n) for i in range(steps)]
    pg = p0.clone().cuda(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    for i in range(steps):
        p.grad = grads[i].clone()
        opt.step()
        t = i + 1
        bc1 = 1 - 0.9 ** t; bc2 = 1 - 0.999 ** t
        _lib.call("mpn_adam_step", ops.ptr(pg), ops.ptr(grads[i].cuda()), ops.ptr(m), ops.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 0.0,
                  bc1, math.sqrt(bc2), 1.0, ops.stream_ptr())
    check_close("adam", pg.cpu(), p.detach(), torch.float32, factor=0.05)
