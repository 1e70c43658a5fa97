#!/usr/bin/env python3
"""Latency of the device heat-map peak extraction (run on the GPU box).

Shapes: the golden 120x120 case (round 2-5 figure) and BASELINE config 5's own: 64 x 18 x 160 x 160 channels-last (what
poseNet.forward_all_images hands over at 640x640), with two contents — "people": 4 Gaussian blobs per joint plane (what a trained
model leaves above 0.1), "noise": uniform noise thresholded so that hundreds of cells per plane are peaks (what random weights give).
Per-kernel times: rocprofv3 --kernel-trace --stats -- python tools/peaks_microbench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multiposenet.pytorch_amd.network import joint_utils as ju


def timeit(pred, thre, refine, cap=ju.DEFAULT_CAP, iters=20):
    for _ in range(3):
        pk, cnt = ju._peaks_device(pred, thre, 4.0, refine, cap)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        pk, cnt = ju._peaks_device(pred, thre, 4.0, refine, cap)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters, cnt


def main():
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g10_peaks.npz"))
    heat = torch.from_numpy(g["heat_c"]).cuda().permute(2, 0, 1)           # [18, 120, 120]
    for B in (1, 32):
        pred = heat[None].repeat(B, 1, 1, 1).contiguous(memory_format=torch.channels_last)
        for refine in (False, True):
            us, cnt = timeit(pred, 0.1, refine)
            print("heatmap_peaks B=%d 18x120x120 refine=%d: %.1f us/call (%d peaks/image), heat-maps read at %.1f GB/s"
                  % (B, refine, us, int(cnt[0].sum()), pred.numel() * 4 / us / 1e3))
    B, J, H, W = 64, 18, 160, 160
    yy, xx = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
    gen = torch.Generator(device="cuda").manual_seed(5)
    people = torch.zeros((B, J, H, W), device="cuda")
    for _ in range(4):
        cx = torch.rand((B, J, 1, 1), device="cuda", generator=gen) * (W - 1)
        cy = torch.rand((B, J, 1, 1), device="cuda", generator=gen) * (H - 1)
        people = torch.maximum(people, torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 2.5 ** 2)))
    people += 0.02 * torch.rand((B, J, H, W), device="cuda", generator=gen)
    noise = torch.rand((B, J, H, W), device="cuda", generator=gen)
    for name, t, thre in (("people (4 blobs / plane)", people, 0.1), ("noise, ~300 peaks / plane", noise, 0.988), ("noise, ~5000 peaks / plane", noise, 0.1)):
        for layout in ("channels_last", "planar"):
            pred = t.contiguous(memory_format=torch.channels_last) if layout == "channels_last" else t.contiguous()
            cap = 256
            _, cnt = timeit(pred, thre, True, cap, iters=1)
            most = int(cnt.max())
            if most > cap:
                cap = 1 << (most - 1).bit_length()
            for refine in (False, True):
                us, cnt = timeit(pred, thre, refine, cap)
                print("heatmap_peaks 64x18x160x160 %s %s refine=%d cap=%d: %.1f us/call (%.1f peaks/plane, max %d), %.1f GB/s of heat-map"
                      % (layout, name, refine, cap, us, float(cnt.float().mean()), most, pred.numel() * 4 / us / 1e3))


if __name__ == "__main__":
    main()
