"""Deterministic stand-in for ``poseNet([img, 'both'])`` used to pin the multi-scale + flip test-time-augmentation driver
(evaluate/tester.py:131-193,256-331) — shared by tests/golden/make_golden_tta.py (which drives the REAL reference ``Tester`` with it
on the CPU) and tests/test_round4_gpu.py (which drives the product ``Tester`` with it on the device).

The maps are smooth functions of the 4x4-pooled input and of the cell coordinates (a handful of well separated peaks per
channel); the boxes depend on the padded input size in a way that makes every scale's list different, so a driver that keeps
the wrong scale's boxes, pads with the wrong value or scales by the wrong side is visible in the fixture."""
import math

import torch
import torch.nn.functional as F

SCORES = (0.9, 0.6, 0.45, 0.7, 0.51)
CLASSES = (0, 0, 0, 1, 0)


def standin_outputs(im_data):
    """im_data: float32 [1, 3, H, W] (any device).  Returns (heat [1, 18, H/4, W/4], [scores [k], classes [k], boxes [k, 4]])."""
    x = im_data.float()
    dev = x.device
    H, W = int(x.shape[2]), int(x.shape[3])
    pool = F.avg_pool2d(x, 4)
    h, w = int(pool.shape[2]), int(pool.shape[3])
    yy = (torch.arange(h, device=dev, dtype=torch.float32) / float(h))[:, None]
    xx = (torch.arange(w, device=dev, dtype=torch.float32) / float(w))[None, :]
    maps = []
    for c in range(18):
        a = pool[0, c % 3]
        ph = 2.0 * math.pi * ((1 + c % 4) * xx + (1 + c % 3) * yy) + 0.37 * c
        maps.append(0.35 + 0.3 * torch.sin(ph) * torch.tanh(0.5 + 0.25 * a))
    heat = torch.stack(maps)[None]
    k = float((H * W) % 13)
    boxes = torch.tensor([[0.10 * W + k, 0.15 * H, 0.45 * W, 0.90 * H],
                          [0.50 * W, 0.20 * H + k, 0.90 * W, 0.80 * H],
                          [0.05 * W, 0.05 * H, 0.25 * W, 0.35 * H],
                          [0.30 * W, 0.30 * H, 0.60 * W, 0.70 * H],
                          [0.62 * W, 0.10 * H, 0.97 * W - k, 0.55 * H]], dtype=torch.float32, device=dev)
    scores = torch.tensor(SCORES, dtype=torch.float32, device=dev)
    classes = torch.tensor(CLASSES, dtype=torch.float32, device=dev)
    return heat, [scores, classes, boxes]


def synth_image(seed, H, W):
    """A decoded-image stand-in: float32 [H, W, 3] BGR in 0..255 — smooth gradients plus a little seeded noise (numpy array)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.stack([120 + 100 * np.sin(xx / 17.0 + c) * np.cos(yy / 23.0 - c) for c in range(3)], 2)
    img += rs.uniform(-1, 1, size=img.shape)
    return np.clip(img, 0, 255).astype(np.float32)


def fake_prn_results(kps, boxes, file_name, image_id):
    """What the recorder standing in for ``Tester.prn_process`` returns: one result per box with keypoint values that encode their
    position, so the COCO re-ordering (tester.py:167-175) is visible in the output."""
    out = []
    for i, b in enumerate(boxes):
        out.append({'image_id': image_id, 'file_name': file_name, 'category_id': 1, 'bbox': [float(v) for v in b],
                    'score': float(i) + 0.5, 'keypoints': [float(1000 * i + j) for j in range(51)]})
    return out
