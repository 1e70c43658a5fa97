#!/usr/bin/env python3
"""Latency of the device NMS (pth_nms) at N = 1k / 5k / 20k boxes and at the worst case of BASELINE config 5 — every one of the
76 725 anchors of a 640x640 image above the score threshold, which is what a random-weight detector produces and what the reference's
uncapped `scores > 0.05` (posenet.py:271) would hand to nms — run on the GPU box.  With `pre_nms_top_n` the suppression sees that many."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multiposenet.pytorch_amd.lib.nms.pth_nms import pth_nms

rng = np.random.RandomState(0)
for n in (1000, 5000, 20000, 76725):
    c = rng.uniform(0, 640, size=(n, 2)); wh = rng.uniform(16, 200, size=(n, 2))
    dets = np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(0.05, 1, size=(n, 1))], 1).astype(np.float32)
    d = torch.from_numpy(dets).cuda()
    for _ in range(2 if n > 50000 else 3):
        keep = pth_nms(d, 0.5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 3 if n > 50000 else 10
    for _ in range(reps):
        keep = pth_nms(d, 0.5)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    print("pth_nms N=%5d: %8.1f us (%d kept; %.1f M IoU pairs -> %.1f G pairs/s)" % (n, us, keep.numel(), n * (n - 1) / 2e6, n * (n - 1) / 2 / us / 1e3))
