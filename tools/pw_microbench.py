#!/usr/bin/env python3
"""conv_pw_kernel (pixel tile resident in LDS) vs the generic implicit-GEMM kernel on the short-K wide-output 1x1 layers of the
R101 480x480 B=32 step, cold operands (sets cycled past the 256 MB Infinity Cache).  Run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multiposenet.pytorch_amd import ops
from multiposenet.pytorch_amd._lib import call

SHAPES = [  # name, B, H, Cin, Cout
    ("256->1024 @30", 32, 30, 256, 1024),
    ("128->512 @60", 32, 60, 128, 512),
    ("64->256 @120", 32, 120, 64, 256),
    ("256->512 @60", 32, 60, 256, 512),
    ("256->1024 @40 (cfg5 640/16)", 64, 40, 256, 1024),
]
dt = torch.bfloat16
dev = "cuda"


def bench(fn, nsets, n=40):
    for i in range(3):
        fn(i % nsets)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i % nsets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


def main():
    shapes = SHAPES if not os.environ.get("PW_ONLY") else [SHAPES[int(i)] for i in os.environ["PW_ONLY"].split(",")]
    for name, B, H, Cin, Cout in shapes:
        P = B * H * H
        set_bytes = 2 * P * (Cin + 4 * Cout)
        nsets = max(2, int(900e6 // set_bytes) + 1)
        xs = [ops.Act(torch.randn(B, H, H, Cin, device=dev).to(dt), Cin) for _ in range(nsets)]
        ys = [ops.Act(torch.randn(B, H, H, Cout, device=dev).to(dt), Cout) for _ in range(nsets)]
        by = [ops.Act(torch.randn(B, H, H, Cout, device=dev).to(dt), Cout) for _ in range(nsets)]
        bz = [ops.Act(torch.randn(B, H, H, Cout, device=dev).to(dt), Cout) for _ in range(nsets)]
        w = (torch.randn(Cout, 1, 1, Cin, device=dev) / Cin ** 0.5).to(dt)
        scale, bias = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
        st = ops.BNState(Cout, torch.device(dev))
        st.mean.normal_(); st.invstd.uniform_(0.5, 1.5); st.scale.fill_(1.0); st.shift.zero_()
        variants = [
            ("fwd + BN stats", lambda i: ops.conv_forward(xs[i], w, Cout, 1, 1, 1, 0, want_stats=True, out=ys[i]), 2 * P * (Cin + Cout)),
            ("dgrad acc + bnb(relu,z)", lambda i: ops.conv_forward(xs[i], w, Cout, 1, 1, 1, 0, out=ys[i], accumulate=True, bnb=(by[i], bz[i], st, True)), 2 * P * (Cin + 4 * Cout)),
            ("dgrad acc", lambda i: ops.conv_forward(xs[i], w, Cout, 1, 1, 1, 0, out=ys[i], accumulate=True), 2 * P * (Cin + 2 * Cout)),
            ("folded BN + res + relu", lambda i: ops.conv_forward(xs[i], w, Cout, 1, 1, 1, 0, out=ys[i], scale=scale, bias=bias, act=3, res=by[i], res_mode=1), 2 * P * (Cin + 2 * Cout)),
        ]
        if os.environ.get("PW_VARIANTS"):
            variants = [variants[int(i)] for i in os.environ["PW_VARIANTS"].split(",")]
        for vname, fn, byts in variants:
            res = []
            for thr in (1 << 30, 0):
                call("mpn_conv_pw_set_min_tiles", thr)
                res.append(bench(fn, nsets))
            flops = 2.0 * P * Cin * Cout
            print("%-30s %-26s generic %7.1f us (%5.2f TB/s)   resident %7.1f us (%5.2f TB/s, %6.1f TF/s)   x%.2f"
                  % (name, vname, res[0], byts / res[0] / 1e6, res[1], byts / res[1] / 1e6, flops / res[1] / 1e6, res[0] / res[1]), flush=True)
        del xs, ys, by, bz
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
