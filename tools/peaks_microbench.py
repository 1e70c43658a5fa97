#!/usr/bin/env python3
"""Latency of the device heat-map peak extraction at the inference shape (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multiposenet.pytorch_amd.network import joint_utils as ju

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g10_peaks.npz"))
heat = torch.from_numpy(g["heat_c"]).cuda().permute(2, 0, 1)           # [18, 120, 120]
param = {"thre1": 0.1}
for B in (1, 32):
    pred = heat[None].repeat(B, 1, 1, 1).contiguous(memory_format=torch.channels_last)
    for refine in (False, True):
        for _ in range(3):
            ju._peaks_device(pred, 0.1, 4.0, refine)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            pk, cnt = ju._peaks_device(pred, 0.1, 4.0, refine)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20
        print("heatmap_peaks B=%d 18x120x120 refine=%d: %.1f us/launch (%d peaks/image), heat-maps read at %.1f GB/s"
              % (B, refine, us, int(cnt[0].sum()), pred.numel() * 4 / us / 1e3))
