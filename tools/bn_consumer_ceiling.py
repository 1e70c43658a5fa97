#!/usr/bin/env python3
"""VERDICT r2 item 6 — "normalise in the consumer, for 1x1 consumers only": what could it save?

The only BatchNorm of a bottleneck whose sole consumer is a 1x1 convolution is bn2 (conv3 reads relu(bn2(conv2 out)),
fpn.py:28-34); bn1 feeds the 3x3 (halo), bn3 feeds the residual add.  Dropping bn2's bn_act launch from the training step is the
ceiling of any consumer-side scheme — the consumer would have to run no slower, and conv3's weight gradient (whose operand is that
same normalised tensor) would have to apply the transform as well.  This tool times exactly those launches at the step's shapes
(R101, 480x480, B=32, bf16; mask bits on, as in training) and prints the ceiling next to the whole bn_act class."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


class St(object):
    pass


def main():
    from multiposenet.pytorch_amd import ops
    torch.cuda.set_device(0)
    B = 32
    dev = torch.device("cuda:0")
    # (blocks, channels of bn2, H=W) per ResNet-101 stage at 480x480; bn1 has the same shape except in the first block of a
    # stage, bn3 has 4x the channels
    stages = [(3, 64, 120), (4, 128, 60), (23, 256, 30), (3, 512, 15)]
    tot = {"bn2": 0.0, "bn1": 0.0, "bn3": 0.0}
    print("# bn_act launch times, R101 480x480 B=%d bf16 (z + ReLU sign bits written, as in the training step)" % B)
    print("%-28s %6s %10s %10s %10s" % ("site", "calls", "us/launch", "GB/s", "ms/step"))
    for blocks, c, hw in stages:
        for name, ch, res in (("bn1", c, False), ("bn2", c, False), ("bn3", 4 * c, True)):
            y = ops.Act(torch.randn(B, hw, hw, ch, device=dev).to(torch.bfloat16), ch)
            r = ops.Act(torch.randn(B, hw, hw, ch, device=dev).to(torch.bfloat16), ch) if res else None
            st = St()
            st.scale = torch.rand(ch, device=dev) + 0.5
            st.shift = torch.randn(ch, device=dev)
            for _ in range(5):
                ops.bn_act(y, st, True, res=r, want_mask=True)
            n = 50
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.bn_act(y, st, True, res=r, want_mask=True)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            byt = y.t.numel() * 2 * (3 if res else 2) + y.t.numel() // 8
            ms = us * blocks / 1e3
            tot[name] += ms
            print("%-28s %6d %10.1f %10.0f %10.3f" % ("%s %dch @%dx%d%s" % (name, ch, hw, hw, " +res" if res else ""), blocks, us, byt / us / 1e3, ms))
    print("bn_act per step by site (back-to-back launches of one shape: includes the ~8 us host enqueue interval where the kernel is shorter):")
    for k in ("bn1", "bn2", "bn3"):
        print("  %s  %.3f ms" % (k, tot[k]))
    print("ceiling of consumer-side normalisation for 1x1 consumers (all 33 bn2 launches gone, conv3 fwd and wgrad unchanged): %.3f ms of a ~38.4 ms step" % tot["bn2"])


if __name__ == "__main__":
    main()
