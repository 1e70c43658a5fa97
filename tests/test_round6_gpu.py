"""Round 6 GPU tests: the N > 1 step is tail-free and fail-safe before it is first measured.

  * per-bucket Adam behind each bucket's gradients / all-reduce on the finishing stream == the single-launch Adam, bit for bit
    (one GPU, local schedule; two ranks over gloo in tests/test_round2_gpu.py's replay worker, which now asserts the count);
  * `bench.py --gpus 8 --shared-device-test`: eight ranks through spawn -> rendezvous -> recorded data-parallel step -> ONE line;
  * `--rccl-channels` / environment echoed in the dist block; a rank that cannot rendezvous exits 3 with a diagnosis.
"""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

from helpers import ROOT, report

pytestmark = pytest.mark.gpu


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _train_setup(layers=50, B=4, S=128, dtype=torch.bfloat16):
    from multiposenet.pytorch_amd import synthetic as weightgen
    from multiposenet.pytorch_amd.network.posenet import poseNet
    torch.manual_seed(0)
    m = poseNet(layers, compute_dtype=dtype).cuda()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = weightgen.gen_state_dict(shapes, seed=0, flavour="he", skip_prefixes=("prn.",))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    for p in m.prn.parameters():
        p.requires_grad = False
    m.train()
    img = torch.from_numpy(weightgen.gen_images(7, B, S, S)).cuda()
    heat, wgt = (torch.from_numpy(a).cuda() for a in weightgen.gen_keypoint_gt(8, B, S // 4, S // 4))
    anno = torch.from_numpy(weightgen.gen_boxes_gt(9, B, S)).cuda()
    return m, [[img, "train_both"]], ["train_both", heat, wgt, anno]


@pytest.mark.parametrize("subnet", ["train_both", "keypoint_subnet"])
def test_per_bucket_adam_equals_the_single_launch_adam_bit_for_bit(subnet, monkeypatch):
    """Recorded step with MPN_BUCKET_ADAM=1 (Adam per ~8 MB bucket on the finishing stream while backward still runs) vs =0 (one
    launch over the arena after backward), five optimizer steps with batch-statistics BatchNorm: parameters, both moments and the
    step count identical bit for bit; every bucket updated exactly once per step, the unused detection head's buckets (keypoint
    step) by finish().  Replaces trainer.py:259's optimizer.step() for the step that owns its optimizer."""
    import copy
    from multiposenet.pytorch_amd.optim import FusedAdam
    from multiposenet.pytorch_amd.replay import ReplayedTrainStep
    monkeypatch.setenv("MPN_BUCKET_MB", "8")
    m, inputs, gts = _train_setup()
    if subnet == "keypoint_subnet":
        inputs = [[inputs[0][0], subnet]]
        gts = [subnet, gts[1], gts[2]]
    start = copy.deepcopy(m.state_dict())
    start_flat = m._arena.flat.detach().cpu().numpy().copy()
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MPN_BUCKET_ADAM", mode)
        m.load_state_dict(start)
        opt = FusedAdam(m, lr=1e-3, weight_decay=1e-4)
        stepper = ReplayedTrainStep(m, opt)
        assert stepper.bucketed_update == (mode == "1")
        losses = [float(stepper(inputs, gts)[0]) for _ in range(5)]
        torch.cuda.synchronize()
        res[mode] = (m._arena.flat.detach().cpu().numpy().copy(), opt._m.cpu().numpy().copy(), opt._v.cpu().numpy().copy(), opt.step_count(), losses)
        if mode == "1":
            sc = stepper._local
            assert sc.local and len(sc.buckets) >= 4 and sc.updated == len(sc.buckets) == sc.launched, (sc.updated, len(sc.buckets))
            assert stepper.replays >= 3
            nb = len(sc.buckets)
    a, b = res["0"], res["1"]
    assert a[3] == b[3] == 5
    assert np.array_equal(a[0], b[0]), "parameters differ between per-bucket and single-launch Adam"
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), "Adam moments differ"
    assert a[4] == b[4] and all(np.isfinite(a[4]))
    assert np.abs(a[0] - start_flat).max() > 1e-4, "parameters never moved"
    report("per-bucket Adam (%s, %d buckets of <= 8 MB, 5 recorded steps) == single-launch Adam: parameters / moments bit-identical; loss %.4f -> %.4f"
           % (subnet, nb, a[4][0], a[4][-1]))


def test_eager_backward_after_a_recorded_step_does_not_update_in_backward():
    """The per-bucket update is installed by the recorded step's begin(on_bucket) and cleared by the next plain backward: an eager
    loss.backward() + optimizer.step() in between applies exactly ONE update (bench.py runs instrumented eager steps on the same
    model and reducer after the timed region)."""
    from multiposenet.pytorch_amd.optim import FusedAdam
    from multiposenet.pytorch_amd.replay import ReplayedTrainStep
    from multiposenet.pytorch_amd.training.batch_processor import train_step
    m, inputs, gts = _train_setup(B=2, S=96, dtype=torch.float32)
    m.freeze_bn()
    opt = FusedAdam(m, lr=1e-3)
    stepper = ReplayedTrainStep(m, opt)
    for _ in range(3):
        stepper(inputs, gts)
    torch.cuda.synchronize()
    t0 = opt.step_count()
    before = m._arena.flat.detach().clone()
    train_step(m, opt, inputs, gts)
    torch.cuda.synchronize()
    assert opt.step_count() == t0 + 1
    # Adam's first-order bound: one update moves a parameter by at most ~lr (bias-corrected |m/sqrt(v)| <= ~1 after a few steps, generous 3x)
    moved = (m._arena.flat.detach() - before).abs().max().item()
    assert 0 < moved <= 3e-3, moved
    stepper(inputs, gts)
    torch.cuda.synchronize()
    assert opt.step_count() == t0 + 2


def test_bench_spawn_path_runs_eight_ranks_to_one_json_line():
    """The driver's 8-GPU command, on one device: `bench.py --gpus 8 --shared-device-test` -> spawn_ranks -> torch.distributed.run ->
    rendezvous of EIGHT ranks on 127.0.0.1 (with the init timeout and the probe collective) -> ddp.attach -> recorded data-parallel
    step with per-bucket updates -> barrier / max over ranks -> ONE JSON line with eight per_rank_ms.  gloo, every rank on device 0,
    labelled not_a_measurement (RCCL refuses two ranks on one GPU).  Replaces datasets/data_parallel.py:16-87 / trainer.py:170."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--shared-device-test",
           "--layers", "50", "--size", "256", "--batch", "2", "--no-kernel-events"]
    t0 = time.time()
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=_clean_env(), cwd=ROOT)
    assert res.returncode == 0, "bench.py --gpus 8 failed (rc %d):\n%s" % (res.returncode, res.stderr[-3000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line, got %d:\n%s" % (len(lines), res.stdout[-2000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 2 and out["not_a_measurement"] is True and out["config"]["global_batch"] == 16
    d = out["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 8 and len(d["per_rank_ms"]) == 8 and all(v > 0 for v in d["per_rank_ms"])
    assert d["collectives_per_step"] == d["buckets"] == d["optimizer_updates_behind_buckets"] >= 1
    assert d["rccl_channels"] == "library default" and d["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "cpu_baseline" not in out and np.isfinite(out["last_step_log"]["total_loss"])
    report("bench.py --gpus 8 --shared-device-test: 8 ranks -> one JSON line in %.0f s (per-rank ms %s; not a measurement)" % (time.time() - t0, d["per_rank_ms"]))


def test_force_dist_line_echoes_channels_and_runs_updates_behind_buckets():
    """One rank over RCCL with --rccl-channels 8: NCCL_MIN/MAX_NCHANNELS set before init and echoed; the per-bucket updates ran."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--rccl-channels", "8", "--steps", "4", "--warmup", "2", "--layers", "50",
           "--size", "256", "--batch", "8", "--no-kernel-events", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_clean_env(), cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')][0])
    d = out["dist"]
    assert d["rccl_channels"] == 8 and d["env"]["NCCL_MIN_NCHANNELS"] == "8" and d["env"]["NCCL_MAX_NCHANNELS"] == "8"
    assert d["backend"] == "nccl" and d["optimizer_updates_behind_buckets"] == d["buckets"] == d["collectives_per_step"]
    assert d["allreduce_ms_exposed"][0] is not None and 0.0 <= d["allreduce_ms_exposed"][0] < out["ms_per_step"]
    report("bench.py --force-dist --rccl-channels 8: %d buckets updated behind their collectives, exposed tail %.3f ms of %.2f ms/step"
           % (d["buckets"], d["allreduce_ms_exposed"][0], out["ms_per_step"]))


def test_rank_that_cannot_rendezvous_exits_3_with_a_diagnosis():
    """A lone rank of a 2-rank job (its peer never starts) must not hold the GPU lease for torch's default ten minutes: with
    --init-timeout 15 it exits with status 3 and ONE diagnostic line naming the rendezvous, and prints no JSON."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(_clean_env(), RANK="1", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-device-test", "--init-timeout", "15", "--layers", "50",
           "--size", "128", "--batch", "2", "--steps", "1", "--warmup", "0"]
    t0 = time.time()
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    dt = time.time() - t0
    assert res.returncode == 3, (res.returncode, res.stderr[-1500:])
    assert "could not join the process group" in res.stderr and "MASTER_PORT=%d" % port in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert dt < 200, dt
    report("lone rank of a 2-rank job: exit 3 after %.0f s with: %s" % (dt, [ln for ln in res.stderr.splitlines() if "could not join" in ln][0][:160]))


# ------------------------------------------------------------------------------------------------ conv2 by position classes
def _head_run(m, feats, gs_pred, classes, overlap):
    """keypoint_head forward + backward on fixed pyramid inputs / output gradient; returns (pred, input grads, conv2 dW, conv2 db, every head dparam)."""
    from multiposenet.pytorch_amd import ops
    from multiposenet.pytorch_amd.engine import Ctx
    eng = m._engine
    eng.conv2_classes, eng.overlap_wgrad = classes, overlap
    m._arena.ensure_grads()
    m._arena.grad_flat.zero_()
    ctx = Ctx(True)
    dt = m.compute_dtype
    xs = [ops.Act(f.permute(0, 2, 3, 1).contiguous().to(dt).cuda(), f.shape[1], needs_grad=True) for f in feats]
    pred, _ = eng.keypoint_head(ctx, xs, False, internal=True)
    g = ops.Act(torch.zeros(pred.t.shape, dtype=dt, device=pred.t.device), pred.C)          # the loss kernel hands gradients over in the compute dtype
    g.t[..., : pred.C] = gs_pred.permute(0, 2, 3, 1).to(dt).cuda()
    ctx.out_grads = {"pred": g}
    eng.run_backward(ctx, ctx.out_grads)
    torch.cuda.synchronize()
    out = pred.t[..., : pred.C].float().cpu()
    return out, m.conv2.weight.grad.detach().float().cpu().clone(), m.conv2.bias.grad.detach().float().cpu().clone(), \
        {n: p.grad.detach().float().cpu().clone() for n, p in m.named_parameters() if n.startswith(("convt", "convs", "convfin.")) and p.grad is not None}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv2_by_position_classes_equals_the_virtual_concatenation(dtype):
    """conv2 of the keypoint head (posenet.py:311-315) with its x8 / x4 members as nine position-class maps each (csrc/conv2cls.hip:
    combined filters, low-resolution class convolutions, expansion into conv2's epilogue; backward: class pooling of dy, low-resolution
    input / filter gradients, fold) against (a) the same head through the plain virtual concatenation in the same 16-bit type and (b)
    the fp32 kernels (materialised concatenation) as the truth: prediction, conv2's filter and bias gradients and every gradient that
    flows on through q5 .. q2 (the convt / convs parameters).  The class path rounds one partial sum more (the expanded maps): its
    error against fp32 may exceed the plain path's by at most 1.6x; serial and two-stream schedules give identical bits."""
    from test_model_gpu import get_model
    from test_round4_gpu import _rel, _bf16_randn
    B, S = 2, 64                                               # pyramid levels 64, 32, 16, 8 -> conv2 at 64 x 64
    feats = [_bf16_randn(900 + i, B, 256, S >> i, S >> i, relu=True) for i in range(4)]
    g_pred = _bf16_randn(910, B, 18, S, S, scale=0.05)
    ref_m = get_model(50, torch.float32)
    ref_m.train()
    saved = (ref_m._engine.conv2_classes, ref_m._engine.overlap_wgrad)
    ref_m._prepare(torch.zeros((B, 3, 4 * S, 4 * S), device="cuda"))
    ref = _head_run(ref_m, feats, g_pred, False, False)
    ref_m._engine.conv2_classes, ref_m._engine.overlap_wgrad = saved
    m = get_model(50, dtype)
    m.train()
    saved = (m._engine.conv2_classes, m._engine.overlap_wgrad)
    m._prepare(torch.zeros((B, 3, 4 * S, 4 * S), device="cuda"))
    try:
        plain = _head_run(m, feats, g_pred, False, False)
        cls = _head_run(m, feats, g_pred, True, False)
        cls2 = _head_run(m, feats, g_pred, True, True)
    finally:
        m._engine.conv2_classes, m._engine.overlap_wgrad = saved
    assert torch.equal(cls[0], cls2[0]) and torch.equal(cls[1], cls2[1]) and torch.equal(cls[2], cls2[2]), "schedules differ"
    assert all(torch.equal(cls[3][k], cls2[3][k]) for k in cls[3])
    rows = []
    for name, i in (("prediction", 0), ("conv2 dW", 1), ("conv2 db", 2)):
        ep, ec, d = _rel(plain[i], ref[i]), _rel(cls[i], ref[i]), _rel(cls[i], plain[i])
        rows.append((name, ep, ec, d))
    worst = max(((k, _rel(plain[3][k], ref[3][k]), _rel(cls[3][k], ref[3][k]), _rel(cls[3][k], plain[3][k])) for k in cls[3]), key=lambda r: r[2])
    rows.append(("worst convt/convs/convfin dparam (%s)" % worst[0],) + worst[1:])
    for name, ep, ec, d in rows:
        report("conv2 by position classes (%s) %-52s vs fp32: plain %.2e, classes %.2e; classes vs plain %.2e" % (str(dtype)[6:], name, ep, ec, d))
        assert ec <= 1.6 * ep + 2e-4, (name, ep, ec)
        assert d <= 1.5 * ep + 1e-3, (name, d)                 # two 16-bit evaluations of the same quantity: no further apart than either is from fp32
    # the four members really take different routes: the class filters' gradient slices are non-zero and differ from each other
    dw = cls[1].reshape(256, 3, 3, 512)
    assert all(float(dw[..., k * 128:(k + 1) * 128].abs().max()) > 0 for k in range(4))


def test_conv2_by_position_classes_in_fp32_is_the_same_arithmetic():
    """fp32 (BASELINE config 2's arithmetic): the class formulation — forward per TAP (nine 1x1 convolutions of the low-resolution member
    + class sums), expansion, the 3x3 convolution over the materialised (q3, q2) half, per-tap backward — against the plain fp32 path
    (materialised 512-channel concatenation): identical mathematics, so the prediction agrees to fp32 summation order (<= 2e-5 rel-L2;
    the 1e-3 abs heat-map gate of the goldens keeps three decades of headroom).  Gradients pass through conv2's ReLU: of its 2 M outputs
    about two lie within 1e-6 of zero and take the other side of the kink in the other summation order; one flipped element is 1 / 64 of
    its channel's bias gradient (a sum of ~4 000 random-sign terms), i.e. ~1.4e-3 rel-L2 over the 256 channels — measured 1.3e-3 / 1.4e-3
    for dW / db at 64 x 64, which is that and nothing else (a wrong tap or class is 1e-1 or more): gate 5e-3.  Also a non-square 40 x 24 head."""
    from test_model_gpu import get_model
    from test_round4_gpu import _rel, _bf16_randn
    m = get_model(50, torch.float32)
    m.train()
    saved = (m._engine.conv2_classes, m._engine.overlap_wgrad)
    try:
        for B, (Hh, Ww) in ((2, (64, 64)), (1, (40, 24))):
            feats = [_bf16_randn(920 + i, B, 256, Hh >> i, Ww >> i, relu=True) for i in range(4)]
            g_pred = _bf16_randn(930, B, 18, Hh, Ww, scale=0.05)
            m._prepare(torch.zeros((B, 3, 4 * Hh, 4 * Ww), device="cuda"))
            plain = _head_run(m, feats, g_pred, False, False)
            cls = _head_run(m, feats, g_pred, True, True)
            worst = max([_rel(cls[i], plain[i]) for i in range(1, 3)] + [_rel(cls[3][k], plain[3][k]) for k in cls[3]])
            report("conv2 by position classes (fp32, %dx%d): prediction %.1e, conv2 dW %.1e, db %.1e, worst gradient %.1e vs the plain fp32 path"
                   % (Hh, Ww, _rel(cls[0], plain[0]), _rel(cls[1], plain[1]), _rel(cls[2], plain[2]), worst))
            assert _rel(cls[0], plain[0]) <= 2e-5 and worst <= 5e-3, (_rel(cls[0], plain[0]), worst)
    finally:
        m._engine.conv2_classes, m._engine.overlap_wgrad = saved


# ------------------------------------------------------------------------------------------------ does it train, bf16 beside fp32
def test_bf16_and_fp32_training_curves_fall_together():
    """VERDICT r5 item 9 (tools/train_sanity.py as a gate): R50 full posenet, 128 x 128, 4 images, two fixed synthetic batches (Gaussian
    heat-map targets rendered by datasets/heatmap.py, random person boxes), 32 recorded steps with FusedAdam at the reference's 1e-4
    (training/multipose_keypoint_train.py:106-110), batch-statistics BatchNorm — once in bf16, once in fp32, same initial weights.
    A SYSTEMATIC bf16 error (a dropped gradient term, a wrong rounding point, statistics from the wrong tensor) bends the bf16 curve away
    from the fp32 one; rounding noise does not: the heat-map losses (last-8-step means) must agree within 5 % (measured 0.8 - 0.9 %),
    the total loss — dominated at step 32 by the focal / smooth-L1 terms, which are still falling 20x per 30 steps and lag by a step
    or two under bf16 noise — within 12 % (measured 6.3 % with conv2 by position classes, 6.9 % without: the same curve either way;
    after 200 steps at 256 x 256 the totals are 0.196 / 0.203, profiles/r06_train_sanity.txt), and both curves must have fallen.  (The full-size bf16 test accepts rel-L2 0.3 on the heat-maps of ONE
    forward under batch statistics — test_round2_gpu.py:245 — which cannot see such a drift.)"""
    from multiposenet.pytorch_amd.datasets.heatmap import put_gaussian_maps
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from multiposenet.pytorch_amd.optim import FusedAdam
    from multiposenet.pytorch_amd.replay import ReplayedTrainStep
    from multiposenet.pytorch_amd import synthetic as weightgen
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(7)
    S, B, steps = 128, 4, 32
    batches = []
    for _ in range(2):
        img = torch.from_numpy(rs.uniform(-2, 2, (B, 3, S, S)).astype(np.float32)).to(dev)
        joints = np.zeros((B, 2, 18, 3), np.float64)
        joints[..., 0] = rs.uniform(8, S - 8, (B, 2, 18)); joints[..., 1] = rs.uniform(8, S - 8, (B, 2, 18)); joints[..., 2] = 1
        heat = put_gaussian_maps(torch.from_numpy(joints).to(dev), torch.full((B,), 2, dtype=torch.int32, device=dev), S, S, stride=4, sigma=7.0)
        anno = np.full((B, 8, 5), -1, np.float32)
        for b in range(B):
            for k in range(2):
                x, y = rs.uniform(4, S - 70, 2); w, h = rs.uniform(30, 60, 2)
                anno[b, k] = [x, y, x + w, y + h, 0]
        batches.append((img, heat.contiguous(), torch.ones_like(heat), torch.from_numpy(anno).to(dev)))
    curves = {}
    for dt in (torch.bfloat16, torch.float32):
        m = poseNet(50, compute_dtype=dt).to(dev)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        sd = weightgen.gen_state_dict(shapes, seed=0, flavour="he", skip_prefixes=("prn.",))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        for p in m.prn.parameters():
            p.requires_grad = False
        m.train()
        step = ReplayedTrainStep(m, FusedAdam(m, lr=1e-4))
        rows = []
        for i in range(steps):
            img, heat, wgt, anno = batches[i % 2]
            loss, log = step([[img, "train_both"]], ["train_both", heat, wgt, anno])
            rows.append((float(loss), float(log["heatmap_loss"])))
        torch.cuda.synchronize()
        assert step.replays >= steps - 4
        curves[dt] = np.array(rows)
        del step, m
        torch.cuda.empty_cache()
    b, f = curves[torch.bfloat16], curves[torch.float32]
    assert np.isfinite(b).all() and np.isfinite(f).all()
    tb, tf = b[-8:, 0].mean(), f[-8:, 0].mean()
    hb, hf = b[-8:, 1].mean(), f[-8:, 1].mean()
    report("does it train (R50 128x128 B=4, 32 recorded steps, lr 1e-4): total loss bf16 %.4f -> %.4f, fp32 %.4f -> %.4f (last-8 means %.1f %% apart); "
           "heat-map loss %.5f / %.5f (%.1f %% apart)" % (b[0, 0], tb, f[0, 0], tf, 100 * abs(tb - tf) / tf, hb, hf, 100 * abs(hb - hf) / hf))
    assert tb < 0.8 * b[:2, 0].mean() and tf < 0.8 * f[:2, 0].mean(), "the loss did not fall"
    assert abs(hb - hf) <= 0.05 * hf and abs(tb - tf) <= 0.12 * tf


def test_bbox_transform_with_given_mean_and_std():
    """network/utils.py:8-48 with non-default coefficients (a reference option the hot path never takes; round 5 raised NotImplementedError):
    deltas * std + mean -> centre / size decode, against the reference's formula evaluated in torch on the CPU; the default-coefficient
    call is bit-identical to what it was (g6_decode.npz keeps gating that)."""
    from multiposenet.pytorch_amd.network.utils import BBoxTransform
    g = torch.Generator().manual_seed(5)
    A, B = 777, 3
    xy = torch.rand(A, 2, generator=g) * 300
    boxes = torch.cat([xy, xy + 5 + torch.rand(A, 2, generator=g) * 120], 1)[None]
    deltas = torch.randn(B, A, 4, generator=g)
    mean, std = torch.tensor([0.05, -0.02, 0.1, 0.0]), torch.tensor([0.2, 0.15, 0.25, 0.3])
    got = BBoxTransform(mean, std)(boxes.cuda(), deltas.cuda()).cpu()
    w, h = boxes[:, :, 2] - boxes[:, :, 0], boxes[:, :, 3] - boxes[:, :, 1]
    cx, cy = boxes[:, :, 0] + 0.5 * w, boxes[:, :, 1] + 0.5 * h
    d = deltas * std + mean
    pcx, pcy, pw, ph = cx + d[..., 0] * w, cy + d[..., 1] * h, torch.exp(d[..., 2]) * w, torch.exp(d[..., 3]) * h
    ref = torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph], 2)
    assert float((got - ref).abs().max()) <= 2e-3 and float(((got - ref).abs() / ref.abs().clamp_min(1.0)).max()) <= 2e-5      # expf + the centre / size cancellation: a few float32 ulps
    dflt = BBoxTransform()(boxes.cuda(), deltas.cuda()).cpu()
    expl = BBoxTransform([0, 0, 0, 0], [0.1, 0.1, 0.2, 0.2])(boxes.cuda(), deltas.cuda()).cpu()
    assert torch.equal(dflt, expl)


def test_pipelined_batched_inference_equals_the_serial_one():
    """Tester.infer_images_batched(pipeline=True) — the network of batch k + 1 enqueued before batch k's detections / peaks / PRN
    assignment are read back, post-processing on a second stream — against pipeline=False on 40 images of mixed sizes in batches of 16
    (three batches, the last one short): identical result dicts (same launches on the same data; only their interleaving differs), and
    both equal per-image inference (tester.py:194-245) on sampled images.  R50, fp16, 256 x 256 network input."""
    from multiposenet.pytorch_amd.evaluate.tester import Tester, TestParams
    from multiposenet.pytorch_amd import synthetic as weightgen
    from test_model_gpu import get_model, t
    m = get_model(50, torch.float16)
    sd = weightgen.gen_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items() if k.startswith("prn.")}, seed=3, flavour="he")
    m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    m.eval()
    S, n = 256, 40
    rs = np.random.RandomState(11)
    sizes = [(S, S)] * 30 + [(200, 256), (256, 180), (131, 222), (256, 256)] * 2 + [(240, 256), (256, 250)]
    images = [rs.uniform(0, 255, (h, w, 3)).astype(np.float32) for h, w in sizes]
    tp = TestParams()
    tp.ckpt, tp.inp_size = None, S
    tester = Tester(m, tp)
    with torch.no_grad():
        _, (cls, _, _) = m([torch.zeros(2, 3, S, S, device="cuda").normal_(), "detection_subnet"])
        s_ = cls.float().flatten().clamp(1e-6, 1 - 1e-6)
        q = torch.quantile(s_, 1.0 - 30.0 / float(cls.shape[1]))
        old_bias = m.classificationModel.output.bias.data.clone()
        m.classificationModel.output.bias.data += float(-torch.log(q / (1 - q)))
    try:
        names, ids = ["f%d.jpg" % i for i in range(n)], list(range(n))
        serial = tester.infer_images_batched(images, names, ids, batch=16, pipeline=False)
        piped = tester.infer_images_batched(images, names, ids, batch=16, pipeline=True)
        again = tester.infer_images_batched(images, names, ids, batch=16, pipeline=True)
        assert len(serial) == len(piped) == n and all(r is not None for r in piped)
        assert serial == piped == again, "the pipelined serving loop changed the results"
        people = sum(len(r) for r in piped)
        assert people >= 10, people
        for i in (0, 17, 33, 39):
            single = tester.infer_image(images[i], names[i], i)
            assert single == piped[i], i
    finally:
        m.classificationModel.output.bias.data.copy_(old_bias)
    report("pipelined batched inference (R50 f16 256x256, 40 images in batches of 16): %d people, result dicts identical to the serial loop and to per-image inference" % people)
