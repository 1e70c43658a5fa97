#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the shapes that dominate the R101 480x480 B=32 step.
Run on the GPU box: python tools/conv_microbench.py [MPN_DEBUG_FLAGS via env]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multiposenet.pytorch_amd import ops
from multiposenet.pytorch_amd import _lib as _mpn_lib
_mpn_lib.use_experiments_build()      # ablation bits / PROF instantiations live in the experiments build (csrc/Makefile)

SHAPES_ALL = [
    # name, B, H, W, Cin, Cout, k, stride, pad, stats
    ("1x1 64->256 @120 stats", 32, 120, 120, 64, 256, 1, 1, 0, True),
    ("1x1 256->64 @120 stats", 32, 120, 120, 256, 64, 1, 1, 0, True),
    ("1x1 256->1024 @30 stats", 32, 30, 30, 256, 1024, 1, 1, 0, True),
    ("1x1 1024->256 @30 stats", 32, 30, 30, 1024, 256, 1, 1, 0, True),
    ("3x3 256->256 @30 stats", 32, 30, 30, 256, 256, 3, 1, 1, True),
    ("3x3 256->256 @60 bias", 32, 60, 60, 256, 256, 3, 1, 1, False),
    ("3x3 512->256 @120 bias", 32, 120, 120, 512, 256, 3, 1, 1, False),
    ("3x3 64->64 @120 stats", 32, 120, 120, 64, 64, 3, 1, 1, True),
]


SHAPES = SHAPES_ALL if os.environ.get('MB_ALL', '1') == '1' else SHAPES_ALL[:3]
if os.environ.get('MB_ONLY'):
    SHAPES = [SHAPES_ALL[int(i)] for i in os.environ['MB_ONLY'].split(',')]
WGRAD = os.environ.get('MB_WGRAD', '1') == '1'


def main():
    dt = torch.bfloat16
    dev = "cuda"
    cold = os.environ.get('MB_COLD', '0') == '1'
    for name, B, H, W, Cin, Cout, k, stride, pad, stats in SHAPES:
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        set_bytes = 2 * B * (H * W * Cin + Ho * Wo * Cout)
        cold_bytes = float(os.environ.get('MB_COLD_BYTES', '800e6'))
        nsets = max(2, min(4096, int(cold_bytes // set_bytes) + 1)) if cold else 1   # cycle > MALL (256 MB) worth of operands
        xs = [ops.Act(torch.randn(B, H, W, Cin, device=dev).to(dt), Cin) for _ in range(nsets)]
        outs = [ops.Act.empty(B, Ho, Wo, Cout, dt, dev) for _ in range(nsets)]
        x, out = xs[0], outs[0]
        w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).to(dt)
        bias = None if stats else torch.randn(Cout, device=dev)
        for _ in range(3):
            ops.conv_forward(x, w, Cout, k, k, stride, pad, bias=bias, want_stats=stats, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        n = int(os.environ.get('MB_ITERS', '20'))
        n = n if not cold else max(n, nsets)
        e0.record()
        for i in range(n):
            ops.conv_forward(xs[i % nsets], w, Cout, k, k, stride, pad, bias=bias, want_stats=stats, out=outs[i % nsets])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / n
        flops = 2.0 * out.P * Cout * k * k * Cin
        byts = (x.t.numel() + out.t.numel() + w.numel()) * 2
        print("%-28s %9.1f us  %7.1f TF/s  %7.1f GB/s(alg)" % (name, us, flops / us / 1e6, byts / us / 1e3), flush=True)
        if not WGRAD:
            continue
        # wgrad of the same layer
        dys = outs
        for o in dys:
            o.t.normal_()
        dw = torch.zeros(Cout, k, k, Cin, device=dev)
        for _ in range(2):
            ops.conv_wgrad(x, dys[0], dw, Cout, k, k, stride, pad)
        torch.cuda.synchronize()
        e0.record()
        for i in range(n):
            ops.conv_wgrad(xs[i % nsets], dys[i % nsets], dw, Cout, k, k, stride, pad)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / n
        print("%-28s %9.1f us  %7.1f TF/s   (wgrad + reduce)" % ("", us, flops / us / 1e6), flush=True)


if __name__ == "__main__":
    print("MPN_DEBUG_FLAGS =", os.environ.get("MPN_DEBUG_FLAGS", "0"))
    main()
