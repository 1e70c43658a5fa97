"""Round 6 CPU tests: host logic and oracle pieces added this round (no GPU)."""
import numpy as np
import torch
import torch.nn.functional as F


def test_position_class_model_of_conv2_is_the_same_mathematics_in_fp32():
    """oracle/posenet_oracle.py: _conv2_position_classes (the rounding model of csrc/conv2cls.hip) evaluated WITHOUT rounding equals
    posenet.py:311-315 — relu(conv2(cat(up8(q5), up4(q4), up2(q3), q2))) — to fp32 summation order: the frame filters, the class of
    every pixel offset and the zero-border argument are right (incl. image borders: sizes 8 and 24 put first / last classes there)."""
    from oracle import posenet_oracle as po
    g = torch.Generator().manual_seed(3)
    sd = {"conv2.weight": torch.randn(256, 512, 3, 3, generator=g) * 0.02, "conv2.bias": torch.randn(256, generator=g) * 0.1}
    for H, W in ((8, 8), (24, 16)):
        qs = [torch.randn(2, 128, H >> s, W >> s, generator=g) for s in (3, 2, 1, 0)]
        cat = torch.cat([F.interpolate(q, size=(H, W), mode="nearest") for q in qs], 1)
        ref = F.relu(F.conv2d(cat, sd["conv2.weight"], sd["conv2.bias"], padding=1))
        got = po._conv2_position_classes(sd, *qs)
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (H, W, err)


def test_bucket_schedule_without_a_process_group():
    """ddp.GradReducer(local=True): the readiness schedule alone (one GPU, per-bucket optimizer updates): buckets cover every trainable
    element once, updates fire in readiness order, finish() sweeps the buckets no gradient reached, begin() clears the callback."""
    from multiposenet.pytorch_amd import ddp
    from multiposenet.pytorch_amd.network.posenet import poseNet
    m = poseNet(50)
    for p in m.prn.parameters():
        p.requires_grad = False
    m._build_arena(torch.device("cpu")) if m._arena is None else None
    ar = m._arena
    ar.ensure_grads()
    red = ddp.GradReducer(ar, local=True, bucket_mb=8.0)
    assert red.world == 1 and red.backend == "local" and len(red.buckets) >= 10
    runs = ar.trainable_runs()
    assert sum(b["end"] - b["start"] for b in red.buckets) == sum(e - s for s, e in runs)
    seen = []
    red.begin(lambda s, e, st: seen.append((s, e)))
    trainable = [p for p in reversed(ar.params) if p.requires_grad]
    for p in trainable[: len(trainable) // 3]:
        red.param_ready(p)
    mid = len(seen)
    assert 0 < mid < len(red.buckets)
    red.finish()
    assert sorted(seen) == sorted((b["start"], b["end"]) for b in red.buckets) and red.updated == len(red.buckets)
    red.begin()
    assert red.on_bucket is None
    red.finish()
    assert red.updated == 0


def test_bench_config_tags_and_gflop_table():
    """bench.py: the flags of BASELINE configs 2 / 3 / 4 map to their PMC-profile tags and GFLOP/img entries, so
    that the cfg2 / cfg4 lines carry model_tflops_per_gpu and look up their own roofline.traffic file."""
    import argparse
    import bench
    mk = lambda **kw: argparse.Namespace(**dict(dict(layers=101, size=480, batch=32, dtype="bf16", subnet="train_both"), **kw))
    assert bench.config_tag(mk()) == ""
    assert bench.config_tag(mk(layers=50, batch=16, dtype="f32", subnet="keypoint_subnet")) == "cfg2"
    assert bench.config_tag(mk(size=800, batch=8)) == "cfg4"
    assert bench.config_tag(mk(size=512)) is None
    assert bench.GFLOP_PER_IMG_TRAIN[(101, 800, "train_both")] == 1690.2 and bench.GFLOP_PER_IMG_TRAIN[(101, 480, "train_both")] == 608.6
    assert bench.pmc_traffic("no_such_kernel", None) == (None, None)
