"""Anchor generation — same interface as the reference's ``network/anchors.py`` (Anchors :6-37,
generate_anchors :39-70, shift :106-126): ``Anchors()(image) -> FloatTensor[1, A, 4]`` on the image's
device, order (level 3..7, y, x, ratio-major/scale-minor anchor).

The reference rebuilds every anchor in numpy and re-uploads it on EVERY forward (anchors.py:21-37).
The anchors depend only on (H, W), so they are built once per input size (float64 host arithmetic,
exactly as the reference, then rounded to float32) and kept resident on the device.
"""
import numpy as np
import torch
import torch.nn as nn

_PYRAMID_LEVELS = (3, 4, 5, 6, 7)
_RATIOS = np.array([0.5, 1, 2])
_SCALES = np.array([2 ** 0, 2 ** (1.0 / 3.0), 2 ** (2.0 / 3.0)])


def generate_anchors(base_size=16, ratios=None, scales=None):
    ratios = _RATIOS if ratios is None else ratios
    scales = _SCALES if scales is None else scales
    nr, ns = len(ratios), len(scales)
    wh = base_size * np.tile(scales, nr)                    # side of the square anchor, scale-minor
    areas = wh * wh
    r = np.repeat(ratios, ns)                               # ratio-major
    w = np.sqrt(areas / r)
    h = w * r
    return np.stack([-0.5 * w, -0.5 * h, 0.5 * w, 0.5 * h], axis=1)


def shift(shape, stride, anchors):
    cx = (np.arange(0, shape[1]) + 0.5) * stride
    cy = (np.arange(0, shape[0]) + 0.5) * stride
    gx, gy = np.meshgrid(cx, cy)
    centres = np.stack([gx.ravel(), gy.ravel(), gx.ravel(), gy.ravel()], axis=1)     # (K, 4), y-major
    return (centres[:, None, :] + anchors[None, :, :]).reshape(-1, 4)


def anchors_for_hw(h, w):
    shape = np.array([h, w])
    per_level = []
    for lv in _PYRAMID_LEVELS:
        fs = (shape + 2 ** lv - 1) // (2 ** lv)
        per_level.append(shift(fs, 2 ** lv, generate_anchors(base_size=2 ** (lv + 2))))
    return np.concatenate(per_level, axis=0)[None].astype(np.float32)


class Anchors(nn.Module):
    def __init__(self):
        super(Anchors, self).__init__()
        self.pyramid_levels = list(_PYRAMID_LEVELS)
        self.strides = [2 ** x for x in self.pyramid_levels]
        self.sizes = [2 ** (x + 2) for x in self.pyramid_levels]
        self.ratios = _RATIOS
        self.scales = _SCALES
        self._cache = {}

    def forward(self, image):
        h, w = int(image.shape[2]), int(image.shape[3])
        key = (h, w, str(image.device))
        t = self._cache.get(key)
        if t is None:
            t = torch.from_numpy(anchors_for_hw(h, w)).to(image.device)
            self._cache[key] = t
        return t
