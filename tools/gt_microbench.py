#!/usr/bin/env python3
"""Throughput of the device ground-truth heat-map generator at the bench batch shape (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multiposenet.pytorch_amd.datasets.heatmap import put_gaussian_maps

rng = np.random.RandomState(1)
B, maxP = 32, 12
j = np.zeros((B, maxP, 18, 3))
j[..., :2] = rng.uniform(0, 480, size=(B, maxP, 18, 2))
j[..., 2] = rng.choice([0., 1., 2.], size=(B, maxP, 18))
jt = torch.from_numpy(j).cuda()
n = torch.full((B,), maxP, dtype=torch.int32, device="cuda")
for _ in range(3):
    put_gaussian_maps(jt, n, 480, 480)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    o = put_gaussian_maps(jt, n, 480, 480)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / 50
print("gt_heatmaps B=32 480x480 stride 4, 12 people/image: %.1f us/launch, %.2f TB/s written (HBM-write roofline ~4 us), %.0f images/s"
      % (us, o.numel() * 4 / us / 1e6, B / us * 1e6))
