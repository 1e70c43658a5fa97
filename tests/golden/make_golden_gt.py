#!/usr/bin/env python3
"""Golden vectors for ground-truth heat-map generation, produced by the REAL reference function
datasets/coco_data/heatmap.py:putGaussianMaps (build container only; /root/reference is not on the GPU box).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gt.py
heatmap.py imports cv2 (absent here) at module scope without using it in putGaussianMaps: an empty stub module is
pre-seeded.  The per-channel / per-person loop is the one of COCO_data_pipeline.py:218-236 (that module cannot be
imported: pycocotools, cv2 image I/O); the arithmetic is the reference's own function.
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("cv2", types.ModuleType("cv2"))

import numpy as np

from datasets.coco_data.heatmap import putGaussianMaps  # noqa: E402


def make_case(seed, B, maxP, crop, stride, sigma):
    rng = np.random.RandomState(seed)
    joints = np.zeros((B, maxP, 18, 3), dtype=np.float64)
    joints[..., 0] = rng.uniform(-30.0, crop + 30.0, size=(B, maxP, 18))      # some centres fall outside the crop
    joints[..., 1] = rng.uniform(-30.0, crop + 30.0, size=(B, maxP, 18))
    joints[..., 2] = rng.choice([0.0, 1.0, 2.0], size=(B, maxP, 18), p=[0.35, 0.45, 0.2])   # 2 = not annotated -> skipped
    num = rng.randint(0, maxP + 1, size=(B,)).astype(np.int32)
    num[0] = maxP
    if B > 1:
        num[1] = 0                                                             # an image without people
    # a crowd on one spot so the 1.0 clamp is exercised
    joints[0, :, 3, 0] = crop * 0.5 + rng.uniform(-2, 2, size=maxP)
    joints[0, :, 3, 1] = crop * 0.4 + rng.uniform(-2, 2, size=maxP)
    joints[0, :, 3, 2] = 1.0
    pt = {"crop_size_y": crop, "crop_size_x": crop, "stride": stride, "sigma": sigma}
    gh = gw = int(crop / stride)
    out = np.zeros((B, 18, gh, gw), dtype=np.float32)
    for b in range(B):
        heat = np.zeros((gh, gw, 18))
        for i in range(18):                      # COCO_data_pipeline.py:224-236
            for j in range(num[b]):
                if joints[b, j, i, 2] <= 1:
                    heat[:, :, i] = putGaussianMaps(joints[b, j, i, :2], heat[:, :, i], params_transform=pt)
        out[b] = heat.transpose((2, 0, 1)).astype(np.float32)      # COCO_data_pipeline.py:283-284
    return joints, num, out


def main():
    data = {}
    for name, args in (("a", (11, 3, 6, 480, 4, 7.0)), ("b", (12, 2, 3, 64, 4, 7.0)), ("c", (13, 2, 4, 96, 8, 5.0))):
        joints, num, out = make_case(*args)
        data["joints_" + name], data["num_" + name], data["out_" + name] = joints, num, out
        data["cfg_" + name] = np.array(args[3:], dtype=np.float64)
    path = os.path.join(HERE, "g9_gt_heatmaps.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
