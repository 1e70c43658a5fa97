#!/usr/bin/env python3
"""Layer3 of R101 at 480x480 B=32 (23 Bottlenecks at 30x30: 1x1 1024->256, 3x3 256->256, 1x1 256->1024 + shortcut, BatchNorm in train
mode) as a recorded launch list, replayed back to back: conv + finalize + bn_act launches vs the one-launch form (MpnConvParams.fz).
Run on the GPU box; MPN_DEBUG_FLAGS selects the timing-only ablations of the fused tail (512 no sync, 1024 no normalise pass,
2048 nobody waits, 4096 long poll sleep).  usage: python tools/fuse_bn_microbench.py [blocks] [replays]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multiposenet.pytorch_amd import _lib, ops


def block(x, ws, bns, fused):
    ops.FUSE_BN_ACT = fused
    h = x
    for i, (w, Cout, k) in enumerate(ws):
        gamma, beta, rm, rv = bns[i]
        relu, res = True, (x if i == 2 else None)
        y, st = ops.conv_forward(h, w, Cout, k, k, 1, k // 2, want_stats=True, bn_fin=(gamma, beta, rm, rv, 0.1, 1e-5),
                                 bn_tail=(relu, res, res is not None))
        if isinstance(st, ops.BNState) and st.z is not None:
            h = st.z
        else:
            if not isinstance(st, ops.BNState):
                st = ops.bn_finalize_train(st, y.P, gamma, beta, rm, rv, 0.1, 1e-5)
            h = ops.bn_act(y, st, relu, res=res, want_mask=res is not None)
    return h


def main():
    nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 23
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dt, dev = torch.bfloat16, "cuda"
    B, H, W = int(os.environ.get("MB_B", "32")), int(os.environ.get("MB_HW", "30")), int(os.environ.get("MB_HW", "30"))
    C4, C1 = int(os.environ.get("MB_C", "1024")), int(os.environ.get("MB_C", "1024")) // 4
    x0 = ops.Act(torch.randn(B, H, W, C4, device=dev).to(dt), C4)
    ws = [((torch.randn(C1, 1, 1, C4, device=dev) / C4 ** 0.5).to(dt), C1, 1),
          ((torch.randn(C1, 3, 3, C1, device=dev) / (9 * C1) ** 0.5).to(dt), C1, 3),
          ((torch.randn(C4, 1, 1, C1, device=dev) / C1 ** 0.5).to(dt), C4, 1)]
    bns = [(torch.ones(c, device=dev), torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.ones(c, device=dev)) for c in (C1, C1, C4)]
    print("MPN_DEBUG_FLAGS =", os.environ.get("MPN_DEBUG_FLAGS", "0"), " %d blocks @%dx%d B=%d C=%d" % (nblocks, H, W, B, C4))
    for fused in (False, True, False, True):
        keep = []
        _lib.TAPE = tape = []
        h = x0
        for b in range(nblocks):
            h = block(h, ws, bns, fused)
            keep.append(h)
        _lib.TAPE = None
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        for r in range(3):
            for fn, args, _ in tape:
                fn(*args)
        torch.cuda.synchronize()
        e0.record()
        for r in range(reps):
            for fn, args, _ in tape:
                fn(*args)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / reps
        print("fused=%d: %4d launches, %8.1f us per pass, %6.1f us per block" % (fused, len(tape), us, us / nblocks), flush=True)
    print("latch:", ops.fz_error())


if __name__ == "__main__":
    main()
