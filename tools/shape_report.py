#!/usr/bin/env python3
"""Per-shape table of the conv launches of one R101 480x480 B=32 bf16 train step (HIP events around every launch):
calls, time, TFLOP/s and algorithmic GB/s per distinct (kind, geometry).  Run on the GPU box."""
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
    model = poseNet(101, compute_dtype=torch.bfloat16).to(dev)
    bench.he_weights(model)
    for p in model.prn.parameters():
        p.requires_grad = False
    model.train()
    opt = FusedAdam(model, lr=1e-4, weight_decay=0.0)
    img, heat, wgt, anno = bench.synth(32, 480, dev, seed=100)

    def step():
        pred, (ks, ds) = model([img, "train_both"])
        loss, log = poseNet.build_loss((ks, ds), "train_both", heat, wgt, anno)
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
    print("# conv launches by shape, R101 train_both 480x480 B=32 bf16; total %.2f ms/step (event-bracketed)" % tot)
    print("%-58s %6s %9s %9s %8s %9s" % ("shape", "calls", "ms/step", "avg_us", "TF/s", "GB/s(alg)"))
    for ms, key, calls, us, tf, gb in rows:
        print("%-58s %6.1f %9.3f %9.1f %8.1f %9.1f" % (key, calls, ms, us, tf, gb))


if __name__ == "__main__":
    main()
