"""Deterministic synthetic inputs and weights: COCO-shaped batches for bench.py / the trainer smoke paths, and the shared starting
point of every parity test.

A data generator, no network arithmetic: the same numbers in the build container (where tests/golden/make_golden*.py feed them to
the REAL reference to make the fixtures) and on the GPU box (where only the fixtures travel), so both sides of a parity test start
from identical parameters without committing 100+ MB of weights.  (Lived with the test infrastructure until round 3; bench.py and
the product's own smoke paths must stand on the package alone, so it moved here: the checkers import it, never the other way round.)

Generator: numpy ``Philox`` keyed by SHA-256(seed, tensor name).  Philox is a
counter-based bit generator whose stream is specified (Random123) and stable
across numpy versions/platforms; we never use torch RNG.

Reference facts this mirrors (for the ``reference_init`` flavour):
  * every Conv2d weight ~ N(0, 0.01), bias 0      (network/posenet.py:213-218)
  * classificationModel.output: w = 0, b = -log(99) (network/posenet.py:205-207)
  * regressionModel.output: w = b = 0              (network/posenet.py:208-209)
  * BatchNorm2d defaults gamma=1, beta=0, rm=0, rv=1 (torch.nn)
The ``he`` flavour gives O(1) activations so a 1e-3 absolute gate is a real test.
"""
import hashlib
import math

import numpy as np


def _rng(seed, name):
    h = hashlib.sha256(("%d:%s" % (seed, name)).encode()).digest()
    key = int.from_bytes(h[:16], "little")
    return np.random.Generator(np.random.Philox(key=key))


def normal(seed, name, shape, std=1.0, mean=0.0):
    g = _rng(seed, name)
    return (g.standard_normal(size=shape, dtype=np.float64) * std + mean).astype(np.float32)


def uniform(seed, name, shape, lo=0.0, hi=1.0):
    g = _rng(seed, name)
    return (g.random(size=shape, dtype=np.float64) * (hi - lo) + lo).astype(np.float32)


def gen_param(seed, name, shape, flavour="he"):
    """Return a float32 (int64 for num_batches_tracked) array for one state_dict entry."""
    shape = tuple(int(s) for s in shape)
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    is_bn = (".bn" in name or name.startswith("bn") or "downsample.1" in name)
    if flavour == "reference_init":
        if is_bn:
            if leaf in ("weight", "running_var"):
                return np.ones(shape, np.float32)
            return np.zeros(shape, np.float32)
        if name.startswith("classificationModel.output"):
            if leaf == "weight":
                return np.zeros(shape, np.float32)
            return np.full(shape, -math.log((1.0 - 0.01) / 0.01), np.float32)
        if name.startswith("regressionModel.output"):
            return np.zeros(shape, np.float32)
        if leaf == "bias":
            if name.startswith("prn."):
                return uniform(seed, name, shape, -0.01, 0.01)
            return np.zeros(shape, np.float32)
        if name.startswith("prn."):
            fan_in = shape[1]
            b = 1.0 / math.sqrt(fan_in)
            return uniform(seed, name, shape, -b, b)
        return normal(seed, name, shape, std=0.01)
    # ---- 'he' flavour: everything non-trivial, activations kept O(1) in eval- and train-BN ------
    if is_bn:
        small = name.endswith(("bn3.weight",))           # residual-branch gain: keeps the trunk O(1)
        if leaf == "weight":
            return uniform(seed, name, shape, 0.15, 0.35) if small else uniform(seed, name, shape, 0.5, 1.0)
        if leaf == "bias":
            return normal(seed, name, shape, std=0.1)
        if leaf == "running_mean":
            return normal(seed, name, shape, std=0.1)
        if leaf == "running_var":
            return uniform(seed, name, shape, 0.8, 1.6)
    if len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        relu_after = (name.startswith("fpn.layer") and (".conv1." in name or ".conv2." in name)) \
            or name == "fpn.conv1.weight" or name == "conv2.weight" \
            or (name.startswith(("regressionModel.conv", "classificationModel.conv")))
        std = math.sqrt((2.0 if relu_after else 1.0) / fan_in)
        if name.startswith(("fpn.latlayer", "fpn.toplayer", "fpn.flatlayer", "fpn.conv6")) and shape[1] >= 512:
            std *= 0.7           # c3..c5 are post-ReLU sums with E[x^2] > 1
        return normal(seed, name, shape, std=std)
    if len(shape) == 2:          # PRN linear
        std = math.sqrt(1.0 / shape[1])
        return normal(seed, name, shape, std=std)
    if leaf == "bias":
        if name.startswith("classificationModel.output"):
            return normal(seed, name, shape, std=0.5, mean=-1.0)
        return normal(seed, name, shape, std=0.05)
    return normal(seed, name, shape, std=0.05)


def gen_state_dict(shapes, seed=0, flavour="he", skip_prefixes=()):
    """shapes: mapping name -> shape (e.g. from ``module.state_dict()``).  Returns name -> ndarray."""
    out = {}
    for name, shape in shapes.items():
        if any(name.startswith(p) for p in skip_prefixes):
            continue
        out[name] = gen_param(seed, name, shape, flavour)
    return out


def gen_images(seed, batch, size_h, size_w):
    """Synthetic post-``resnet_preprocess`` images, N(0,1), NCHW float32."""
    return normal(seed, "images", (batch, 3, size_h, size_w))


def gen_keypoint_gt(seed, batch, h, w):
    """heat_temp ~ U(0,1)*mask and heat_weight in {0,1} Bernoulli(0.95), [B,18,h,w]."""
    heat = uniform(seed, "heat_temp", (batch, 18, h, w))
    wgt = (uniform(seed, "heat_weight", (batch, 18, h, w)) < 0.95).astype(np.float32)
    return heat * wgt, wgt


def gen_boxes_gt(seed, batch, size, max_n=8):
    """[B,max_n,5] x1,y1,x2,y2,class; 1-6 valid boxes/image, side U(32,S/2) (clamped), rest -1."""
    g = _rng(seed, "boxes")
    anno = -np.ones((batch, max_n, 5), np.float32)
    for b in range(batch):
        n = int(g.integers(1, min(6, max_n) + 1))
        for i in range(n):
            lo = min(32.0, size / 4.0)
            bw = float(g.uniform(lo, size / 2.0))
            bh = float(g.uniform(lo, size / 2.0))
            x1 = float(g.uniform(0, size - bw))
            y1 = float(g.uniform(0, size - bh))
            anno[b, i] = (x1, y1, x1 + bw, y1 + bh, 0.0)
    return anno
