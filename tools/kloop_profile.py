#!/usr/bin/env python3
"""Where does a k-step go?  Per-wave s_memtime cycle sums of the four phases of the conv k-loops (PROF instantiations of
conv_igemm_kernel / conv_igemm_s3_kernel / conv_wgrad_dma_kernel, 128 x 128 tiles, bf16):

    wait   s_waitcnt vmcnt: the wave's own LDS-DMA of this k-step has not landed yet  (memory latency not covered by the ring)
    barr   s_barrier: waiting for the slowest of the four waves' DMA
    issue  queueing the DMA of the k-step two ahead (address arithmetic + buffer_load ... lds)
    mma    fragment reads (ds_read) + MFMA issue — 16 MFMAs of 16 cycles each = 256 cycles of matrix pipe per k-step and wave

Printed per shape: cycles per k-step of each phase (mean over all waves of all workgroups), the loop and epilogue totals per
workgroup, and the launch's event-timed duration.  Run on the GPU box:  python tools/kloop_profile.py [cold]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from multiposenet.pytorch_amd import ops
from multiposenet.pytorch_amd import _lib as _mpn_lib
_mpn_lib.use_experiments_build()      # ablation bits / PROF instantiations live in the experiments build (csrc/Makefile)
from multiposenet.pytorch_amd._lib import call

dt, dev = torch.bfloat16, "cuda"
SHAPES = [
    # name, B, H, Cin, Cout, k, stats
    ("1x1 256->1024 @30", 32, 30, 256, 1024, 1, True),
    ("1x1 1024->256 @30", 32, 30, 1024, 256, 1, True),
    ("3x3 256->256 @30", 32, 30, 256, 256, 3, True),
    ("3x3 256->256 @60", 32, 60, 256, 256, 3, False),
    ("3x3 128->128 @60", 32, 60, 128, 128, 3, True),
]
CLK = 2.4e3          # cycles per us at the nominal shader clock (s_memtime ticks)


def report(tag, buf, nwg, us):
    t = buf.cpu().numpy().reshape(-1, 4, 8).astype(np.float64)
    live = t[:, :, 6] > 0
    steps = t[:, :, 6][live]
    ph = [t[:, :, k][live] / steps for k in range(4)]
    loop, epi = t[:, :, 4][live], t[:, :, 5][live]
    span = (t[:, :, 7][live] + loop + epi).max() - t[:, :, 7][live].min()
    print("  %-14s %7.1f us | k-steps %3d | per k-step: wait %5.0f barr %5.0f issue %5.0f mma %5.0f = %5.0f cyc | loop %6.0f (+prologue) epilogue %6.0f cyc | "
          "span %5.1f us, %d workgroups" % (tag, us, steps.mean(), ph[0].mean(), ph[1].mean(), ph[2].mean(), ph[3].mean(), sum(p.mean() for p in ph),
                                            loop.mean(), epi.mean(), span / CLK, nwg), flush=True)


def timed(fn, flush):
    if flush is not None:
        flush.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000


def main():
    cold = len(sys.argv) > 1 and sys.argv[1] == "cold"
    flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev) if cold else None
    print("k-loop phase profile (%s caches), cycles = s_memtime ticks" % ("cold" if cold else "warm"))
    for name, B, H, Cin, Cout, k, stats in SHAPES:
        pad = k // 2
        x = ops.Act(torch.randn(B, H, H, Cin, device=dev).to(dt), Cin)
        y = ops.Act.empty(B, H, H, Cout, dt, dev)
        dy = ops.Act(torch.randn(B, H, H, Cout, device=dev).to(dt), Cout)
        w = (torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5).to(dt)
        wt = w.permute(3, 1, 2, 0).contiguous()                 # dgrad operand [Cin][R][S][Cout]
        dw = torch.zeros(Cout, k, k, Cin, device=dev)
        bias = None if stats else torch.randn(Cout, device=dev)
        dx = ops.Act.empty(B, H, H, Cin, dt, dev)
        runs = {
            "fwd": lambda: ops.conv_forward(x, w, Cout, k, k, 1, pad, bias=bias, want_stats=stats, out=y),
            "dgrad": lambda: ops.conv_forward(dy, wt, Cin, k, k, 1, pad, mode=1, out_hw=(H, H), cin=Cout, out=dx),
            "wgrad": lambda: ops.conv_wgrad(x, dy, dw, Cout, k, k, 1, pad),
        }
        if k == 1 and Cin == 1024:
            # the most expensive shape of the step (22 launches x 77.6 us, r03): input gradient of a Bottleneck's conv1 with the deferred
            # shortcut gradient (dz through the ReLU sign bits) and the BatchNorm-backward statistics of the block below in the epilogue
            P, V = B * H * H, 8
            dz = ops.Act(torch.randn(B, H, H, Cin, device=dev).to(dt), Cin)
            by = ops.Act(torch.randn(B, H, H, Cin, device=dev).to(dt), Cin)
            bz = ops.Act(torch.randn(B, H, H, Cin, device=dev).to(dt), Cin)
            bz.mask = torch.randint(0, 256, (P, Cin // V), dtype=torch.uint8, device=dev)
            st = ops.BNState(Cin, torch.device(dev))
            st.mean.normal_(); st.invstd.uniform_(0.5, 1.5); st.scale.fill_(1.0); st.shift.zero_()
            runs["dgrad+res+bnb"] = lambda: ops.conv_forward(dy, wt, Cin, k, k, 1, pad, mode=1, out_hw=(H, H), cin=Cout, out=dx,
                                                              bnb=(by, bz, st, True), res=dz, res_mode=1, res_mask=bz.mask)
        print(name, flush=True)
        for tag, fn in runs.items():
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            base = min(timed(fn, flush) for _ in range(3))
            buf = torch.zeros(8192 * 4 * 8, dtype=torch.int64, device=dev)
            call("mpn_debug_wgrad_prof" if tag == "wgrad" else "mpn_debug_igemm_prof", ops.ptr(buf))
            fn()
            us = min(timed(fn, flush) for _ in range(2))
            call("mpn_debug_wgrad_prof" if tag == "wgrad" else "mpn_debug_igemm_prof", None)
            torch.cuda.synchronize()
            nwg = int((buf.view(-1, 4, 8)[:, 0, 6] > 0).sum())
            if nwg == 0:
                print("  %-6s %7.1f us (not a 128 x 128 launch: no profile)" % (tag, base))
                continue
            report(tag, buf, nwg, us)
            print("         production kernel %7.1f us" % base, flush=True)


if __name__ == "__main__":
    main()
