#!/usr/bin/env python3
"""g13_prn_process.npz: outputs of the REAL ``Tester.prn_process`` (evaluate/tester.py:333-513) on seeded peak / box sets,
with the REAL reference ``poseNet.prn`` (seeded weights from multiposenet/pytorch_amd/synthetic.py) as the model.

Build container only (needs /root/reference).  Shims: cv2 / pycocotools are never touched by prn_process and are stubbed so
that evaluate/tester.py imports; ``skimage.filters.gaussian`` (absent here) is the scipy restatement that
tests/golden/g12_prn_gaussian.npz pins to real skimage; ``.cuda()`` is a no-op; lib.nms is stubbed as in make_golden.py."""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
import torch.nn as nn

from oracle import prn_assign_oracle
from multiposenet.pytorch_amd import synthetic as weightgen


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


stub("cv2")
stub("skimage")
stub("skimage.filters", gaussian=lambda img, *a, **k: prn_assign_oracle.gaussian(img))
stub("pycocotools")
stub("pycocotools.coco", COCO=object)
stub("pycocotools.cocoeval", COCOeval=object)
stub("lib.nms.pth_nms", pth_nms=None)
torch.Tensor.cuda = lambda self, *a, **k: self
nn.Module.cuda = lambda self, *a, **k: self

from evaluate.tester import Tester  # noqa: E402
from network.posenet import poseNet  # noqa: E402


def cases():
    """(kps rows [x, y, score, id, joint_type 0..16], boxes [x1, y1, x2, y2]) — people standing in a 640x480 image."""
    rs = np.random.RandomState(13)
    out = []
    for case in range(6):
        nper = [1, 2, 3, 5, 2, 4][case]
        boxes, kps = [], []
        for p in range(nper):
            cx, cy = rs.uniform(80, 560), rs.uniform(120, 360)
            bw, bh = rs.uniform(40, 160), rs.uniform(120, 320)
            if case == 4:                                  # heavily overlapping people
                cx, cy = 300 + 25 * p, 240
            boxes.append([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2])
            for j in range(17):
                if case == 5 and j in (3, 9):              # joint types nobody has -> the arg-max fallback (tester.py:471-483)
                    continue
                if rs.rand() < 0.8:
                    # some peaks fall in the margin outside the box (in_thres) -> the clamp chain incl. negative indices
                    x = cx + rs.uniform(-0.7, 0.7) * bw
                    y = cy + rs.uniform(-0.7, 0.7) * bh
                    kps.append([float(np.round(x)), float(np.round(y)), float(rs.uniform(0.1, 1.0)), 0.0, float(j)])
        rs.shuffle(kps)
        for i, k in enumerate(kps):
            k[3] = float(i)
        out.append((kps, boxes))
    out.append(([], [[10.0, 10.0, 100.0, 200.0]]))       # no peaks at all
    out.append(([[50.0, 60.0, 0.9, 0.0, 2.0]], []))      # no boxes
    return out


def main():
    torch.manual_seed(0)
    model = poseNet(50)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if k.startswith("prn.")}
    sd = weightgen.gen_state_dict(shapes, seed=3, flavour="he", skip_prefixes=())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model.eval()

    class P(object):
        coeff, in_thres = 2, 0.21

    class Self(object):
        params = P()

    me = Self()
    me.model = model
    out = {}
    with torch.no_grad():
        for ci, (kps, boxes) in enumerate(cases()):
            res = Tester.prn_process(me, kps, boxes, "img%d.jpg" % ci, ci)
            out["kps_%d" % ci] = np.array(kps, dtype=np.float64).reshape(-1, 5)
            out["boxes_%d" % ci] = np.array(boxes, dtype=np.float64).reshape(-1, 4)
            out["n_%d" % ci] = np.array(len(res))
            out["keypoints_%d" % ci] = np.array([r["keypoints"] for r in res], dtype=np.float64).reshape(-1, 51)
            out["score_%d" % ci] = np.array([r["score"] for r in res], dtype=np.float64)
            out["bbox_%d" % ci] = np.array([r["bbox"] for r in res], dtype=np.float64).reshape(-1, 4)
            print("case %d: %d peaks, %d boxes -> %d results, assigned joints per person %s"
                  % (ci, len(kps), len(boxes), len(res), [int(np.count_nonzero(np.array(r["keypoints"])[2::3])) for r in res]))
    out["ncases"] = np.array(len(cases()))
    np.savez_compressed(os.path.join(HERE, "g13_prn_process.npz"), **out)


if __name__ == "__main__":
    main()
