#!/usr/bin/env python3
"""what would the 1x1 launches of layer3 cost if their partial last round of workgroups were free?

The 30x30 1x1 launches run 450 - 1 800 workgroups of 128 x 128 on 768 slots (3 per CU): 0.59 or 2.34 rounds.  For every distinct launch
shape of a layer3 bottleneck this script times the launch at B = 32 (the real grid) and at batch sizes whose grids are whole numbers of
rounds, takes the best time PER WORKGROUP among the whole-round grids as the tail-free rate, and prints
    ceiling = calls/step x (t(B = 32) - rate x workgroups(B = 32)).
The sum is an UPPER bound on what fuller launches (co-scheduled independent 1x1 convolutions) could recover."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from multiposenet.pytorch_amd import ops

SLOTS = 768
# (name, Cin, Cout, mode, calls per step in R101 layer3 (23 blocks), epilogue)
SHAPES = [("fwd conv1 1x1 1024->256 stats", 1024, 256, 0, 22, "stats"),
          ("fwd conv3 1x1 256->1024 stats", 256, 1024, 0, 23, "stats"),
          ("dgrad conv3 1x1 1024->256 (out 256)", 1024, 256, 1, 23, "plain"),
          ("dgrad conv1 1x1 256->1024 res1 (out 1024)", 256, 1024, 1, 22, "res")]


def time_launch(B, Cin, Cout, mode, epi, iters=30, nsets=6):
    dt, dev = torch.bfloat16, "cuda"
    xs = [ops.Act(torch.randn(B, 30, 30, Cin, device=dev).to(dt), Cin) for _ in range(nsets)]      # cycle > L2 worth of operands
    outs = [ops.Act.empty(B, 30, 30, Cout, dt, dev) for _ in range(nsets)]
    ress = [ops.Act(torch.randn(B, 30, 30, Cout, device=dev).to(dt), Cout) for _ in range(nsets)] if epi == "res" else None
    w = (torch.randn(Cout, 1, 1, Cin, device=dev) / Cin ** 0.5).to(dt)

    def go(i):
        kw = dict(out=outs[i % nsets])
        if mode == 1:
            kw.update(mode=1, out_hw=(30, 30), cin=Cin)
        if epi == "stats":
            kw["want_stats"] = True
        if epi == "res":
            kw.update(res=ress[i % nsets], res_mode=1)
        ops.conv_forward(xs[i % nsets], w, Cout, 1, 1, 1, 0, **kw)
    for i in range(4):
        go(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        go(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters


def main():
    total = 0.0
    print("# one MI355X, bf16, 30x30, conv_igemm_kernel<bf16,128,128>; workgroups = ceil(B*900/128) * Cout/128; %d slots" % SLOTS)
    for name, Cin, Cout, mode, calls, epi in SHAPES:
        rows = []
        for B in (27, 32, 41, 54, 55, 82, 109):
            wgs = ((B * 900 + 127) // 128) * (Cout // 128)
            rows.append((B, wgs, wgs / SLOTS, time_launch(B, Cin, Cout, mode, epi)))
        t32 = [r for r in rows if r[0] == 32][0]
        whole = [r for r in rows if abs(r[2] - round(r[2])) <= 0.06 and r[2] >= 0.9]
        rate = min(r[3] / r[1] for r in whole) if whole else t32[3] / t32[1]
        free = rate * t32[1]
        ceil_ms = calls * max(0.0, t32[3] - free) / 1000.0
        total += ceil_ms
        print("%-44s B=32: %4d wgs (%.2f rounds) %6.1f us | tail-free rate %.4f us/wg -> %6.1f us | x%d calls: ceiling %.3f ms/step"
              % (name, t32[1], t32[2], t32[3], rate, free, calls, ceil_ms))
        print("    " + "  ".join("B=%d:%d wgs(%.2f r) %.1fus" % r for r in rows))
    print("# sum over layer3's 1x1 launches: %.3f ms/step (upper bound; threshold to act on: 0.5 ms)" % total)


if __name__ == "__main__":
    main()
