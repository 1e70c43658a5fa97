#!/usr/bin/env python3
"""Golden vectors for heat-map peak extraction, produced by the REAL reference functions network/joint_utils.py:find_peaks
and NMS (build container only).  cv2 is absent here: an empty stub is pre-seeded so the module imports, which limits the
goldens to the branches that never touch cv2 — find_peaks and NMS(bool_refine_center=False).  If cv2 does import (another
box), the refined branch is recorded as well (`refined = 1`).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_peaks.py
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
try:
    import cv2  # noqa: F401
    HAVE_CV2 = True
except Exception:
    sys.modules["cv2"] = types.ModuleType("cv2")
    HAVE_CV2 = False

import numpy as np

from network.joint_utils import NMS, find_peaks  # noqa: E402


def synth_heat(seed, H, W, J=18):
    """Smooth blobs (a few people) + low noise + exact plateaus/ties + a peak on every border."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    heat = np.zeros((H, W, J), dtype=np.float32)
    for j in range(J):
        for _ in range(rng.randint(0, 5)):
            cx, cy = rng.uniform(-2, W + 2), rng.uniform(-2, H + 2)
            s = rng.uniform(1.2, 2.5)
            heat[:, :, j] += (rng.uniform(0.15, 1.0) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))).astype(np.float32)
        heat[:, :, j] += rng.uniform(0, 0.02, size=(H, W)).astype(np.float32)
    heat[3:5, 7:9, 0] = 0.5                     # 2x2 plateau: every cell equals its neighbourhood maximum
    heat[0, 0, 1] = 0.9
    heat[H - 1, W - 1, 1] = 0.8
    heat[0, W // 2, 2] = 0.7
    heat[H // 2, 0, 2] = 0.10000001             # just above thre1
    heat[H // 2, W - 1, 3] = 0.1                # exactly thre1: not a peak (strict >)
    return heat


def main():
    data = {"refined": np.array(int(HAVE_CV2))}
    param = {"thre1": 0.1, "thre2": 0.05, "thre3": 0.5}
    for name, (seed, H, W, up) in (("a", (21, 30, 30, 4.0)), ("b", (22, 46, 62, 8.0)), ("c", (23, 120, 120, 4.0))):
        heat = synth_heat(seed, H, W)
        data["heat_" + name] = heat
        data["up_" + name] = np.array(up)
        fp = [find_peaks(param, heat[:, :, j]) for j in range(18)]
        data["fp_counts_" + name] = np.array([len(p) for p in fp], dtype=np.int64)
        data["fp_xy_" + name] = np.concatenate([p.reshape(-1, 2) for p in fp]).astype(np.int64)
        plain = NMS(param, heat, up, bool_refine_center=False)
        data["nms_plain_" + name] = np.concatenate([p.reshape(-1, 4) for p in plain])
        if HAVE_CV2:
            ref = NMS(param, heat, up, bool_refine_center=True)
            data["nms_refined_" + name] = np.concatenate([p.reshape(-1, 4) for p in ref])
    path = os.path.join(HERE, "g10_peaks.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes; cv2 available:", HAVE_CV2)


if __name__ == "__main__":
    main()
