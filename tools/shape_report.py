#!/usr/bin/env python3
"""Per-shape table of the conv launches of one train step (HIP events around every launch): calls, time, TFLOP/s and algorithmic GB/s per
distinct (kind, geometry).  Default: R101 train_both 480x480 B=32 bf16; SR_LAYERS / SR_SIZE / SR_BATCH / SR_DTYPE / SR_SUBNET select another
configuration (cfg2: SR_LAYERS=50 SR_BATCH=16 SR_DTYPE=f32 SR_SUBNET=keypoint_subnet).  Run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from multiposenet.pytorch_amd import ops
from multiposenet.pytorch_amd.network.posenet import poseNet
from multiposenet.pytorch_amd.optim import FusedAdam


def main():
    dev = torch.device("cuda:0")
    layers, size, batch = int(os.environ.get("SR_LAYERS", "101")), int(os.environ.get("SR_SIZE", "480")), int(os.environ.get("SR_BATCH", "32"))
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[os.environ.get("SR_DTYPE", "bf16")]
    subnet = os.environ.get("SR_SUBNET", "train_both")
    model = poseNet(layers, compute_dtype=dtype).to(dev)
    bench.he_weights(model)
    for p in model.prn.parameters():
        p.requires_grad = False
    model.train()
    opt = FusedAdam(model, lr=1e-4, weight_decay=0.0)
    img, heat, wgt, anno = bench.synth(batch, size, dev, seed=100)
    gts = {"train_both": ["train_both", heat, wgt, anno], "keypoint_subnet": ["keypoint_subnet", heat, wgt], "detection_subnet": ["detection_subnet", anno]}[subnet]

    def step():
        pred, saved = model([img, subnet])
        loss, log = poseNet.build_loss(saved, *gts)
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    ops.KERNEL_EVENTS.enable()
    ops.KERNEL_EVENTS.detail = True
    n = 3
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ops.KERNEL_EVENTS.disable()
    rows = []
    for name, d in ops.KERNEL_EVENTS.summary().items():
        key, byts = name.rsplit("|", 1)
        ms = d["ms"] / n
        calls = d["n"] / n
        us = d["ms"] * 1000.0 / d["n"]
        rows.append((ms, key, calls, us, d["flops"] / d["n"] / us / 1e6, float(byts) / us / 1e3))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("# conv launches by shape, R%d %s %dx%d B=%d %s; total %.2f ms/step (event-bracketed)" % (layers, subnet, size, size, batch, os.environ.get("SR_DTYPE", "bf16"), tot))
    print("%-58s %6s %9s %9s %8s %9s" % ("shape", "calls", "ms/step", "avg_us", "TF/s", "GB/s(alg)"))
    for ms, key, calls, us, tf, gb in rows:
        print("%-58s %6.1f %9.3f %9.1f %8.1f %9.1f" % (key, calls, ms, us, tf, gb))


if __name__ == "__main__":
    main()
