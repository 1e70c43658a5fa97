#!/usr/bin/env python3
"""g15_tta.npz / g15_tta_results.json: the multi-scale + flip test-time-augmentation driver of the REAL reference —
``Tester.coco_eval`` / ``_get_multiplier`` / ``_get_outputs`` / ``_handle_heat`` (evaluate/tester.py:131-193,256-331),
``crop_with_factor`` (:38-82), ``resnet_preprocess`` (datasets/coco_data/preprocessing.py:14-25) and ``get_joint_list`` /
``NMS`` (network/joint_utils.py:61-152) — driven on two synthetic images with a deterministic stand-in model
(tests/tta_standin.py) and a recorder in place of ``prn_process`` (pinned separately: g13).

Build container only (needs /root/reference).  What is REAL here: every line of control flow of the files above — the scale
list, which side the scaling is based on, the pad value, the crop of the padded maps, the float64 averaging, the flip + channel
swap, which scale's boxes are kept (index 1), the neck removal, the COCO keypoint order, the result file.  What is NOT: cv2 is
absent from this image, so ``cv2.resize`` is ``oracle/joint_oracle.cv_resize`` / ``cv_resize_cubic`` (the restatement of OpenCV's
published rules; PARITY UNPINNED) and ``cv2.imread`` hands back the synthetic arrays; pycocotools is a
recording fake.  The fixture therefore pins the DRIVER, not the resize arithmetic."""
import json
import os
import sys
import tempfile
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
import torch.nn as nn

from oracle import joint_oracle
import tta_standin

IMAGES = {}            # file name -> decoded image (what cv2.imread would return)
CALLS = []             # model input shapes, in call order
PRN_ARGS = []          # (kps, boxes, file_name, image_id) handed to prn_process
EVAL_LOG = []


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


INTER_LINEAR, INTER_CUBIC = 1, 2


def cv2_resize(src, dsize, fx=0, fy=0, interpolation=INTER_LINEAR):
    """cv2.resize through the oracle's restatement; 2-D arrays ride as one channel."""
    a = np.asarray(src)
    two_d = a.ndim == 2
    if two_d and dsize is None and interpolation == INTER_CUBIC and fx == fy:
        return joint_oracle.cv_resize_cubic(a, fx)                      # the 5x5 patch of NMS (joint_utils.py:108-109)
    img = a[:, :, None] if two_d else a
    if dsize is None:
        out_hw = (int(np.rint(img.shape[0] * fy)), int(np.rint(img.shape[1] * fx)))
        out = joint_oracle.cv_resize(img, out_hw, interpolation == INTER_CUBIC, inv_scale=(1.0 / fy, 1.0 / fx))
    else:
        out = joint_oracle.cv_resize(img, (dsize[1], dsize[0]), interpolation == INTER_CUBIC)
    return out[:, :, 0] if two_d else out


def cv2_imread(path):
    return IMAGES[os.path.basename(path)].copy()


stub("cv2", resize=cv2_resize, imread=cv2_imread, INTER_LINEAR=INTER_LINEAR, INTER_CUBIC=INTER_CUBIC)
stub("skimage")
stub("skimage.filters", gaussian=None)
stub("lib.nms.pth_nms", pth_nms=None)


class FakeCOCO(object):
    def __init__(self, ann_file=None):
        EVAL_LOG.append(("COCO", os.path.basename(str(ann_file))))

    def getImgIds(self, catIds=None):
        EVAL_LOG.append(("getImgIds", list(catIds)))
        return [11, 22]

    def loadImgs(self, img_id):
        return [{"file_name": "img%d.jpg" % img_id}]

    def loadRes(self, fn):
        EVAL_LOG.append(("loadRes", json.load(open(fn))))
        return "pred"


class FakeEval(object):
    def __init__(self, coco, pred, kind):
        self.params = types.SimpleNamespace(imgIds=None)
        EVAL_LOG.append(("COCOeval", pred, kind))

    def evaluate(self):
        EVAL_LOG.append(("evaluate", list(self.params.imgIds)))

    def accumulate(self):
        EVAL_LOG.append(("accumulate",))

    def summarize(self):
        EVAL_LOG.append(("summarize",))


stub("pycocotools")
stub("pycocotools.coco", COCO=FakeCOCO)
stub("pycocotools.cocoeval", COCOeval=FakeEval)
torch.Tensor.cuda = lambda self, *a, **k: self
nn.Module.cuda = lambda self, *a, **k: self
if not hasattr(np, "float"):
    np.float = float

from evaluate.tester import Tester, TestParams  # noqa: E402


def model(inputs):
    im_data, subnet = inputs
    CALLS.append(tuple(int(v) for v in im_data.shape))
    return tta_standin.standin_outputs(im_data)


def main():
    IMAGES["img11.jpg"] = tta_standin.synth_image(11, 64, 80)
    IMAGES["img22.jpg"] = tta_standin.synth_image(22, 90, 66)
    params = TestParams()
    params.inp_size = 48
    params.coco_root = "coco_root/"
    tmp = tempfile.mkdtemp()
    params.coco_result_filename = os.path.join(tmp, "results.json")
    params.testresult_write_json = True            # keep the file: it is part of the fixture
    me = Tester.__new__(Tester)
    me.params = params
    me.model = model
    me.prn_process = lambda kps, boxes, name, image_id=0: (PRN_ARGS.append((kps, boxes, name, image_id)) or
                                                          tta_standin.fake_prn_results(kps, boxes, name, image_id))
    out = {}
    # ---- the pieces, image by image
    for tag, name in (("a", "img11.jpg"), ("b", "img22.jpg")):
        img = IMAGES[name]
        mult = me._get_multiplier(img)
        del CALLS[:]
        heat, bbox_all = me._get_outputs(mult, img)
        shapes_o = list(CALLS)
        del CALLS[:]
        fheat, fbbox_all = me._get_outputs(mult, img[:, ::-1, :])
        avg = me._handle_heat(heat, fheat)
        out["img_" + tag] = img
        out["multiplier_" + tag] = np.array(mult, dtype=np.float64)
        out["shapes_" + tag] = np.array(shapes_o, dtype=np.int64)
        out["shapes_flip_" + tag] = np.array(list(CALLS), dtype=np.int64)
        out["heat_" + tag] = heat.astype(np.float32)              # float64 accumulators in the reference; float32 is ample for the gates
        out["heat_flip_sum_" + tag] = fheat.sum((0, 1))           # the flipped pass is pinned through its channel sums and the average
        out["heat_avg_" + tag] = avg.astype(np.float32)
        out["bbox_counts_" + tag] = np.array([len(b) for b in bbox_all], dtype=np.int64)
        out["bbox_" + tag] = np.array([v for b in bbox_all for v in b], dtype=np.float64).reshape(-1, 4)
        out["bbox_flip_" + tag] = np.array([v for b in fbbox_all for v in b], dtype=np.float64).reshape(-1, 4)
        print(name, img.shape, "multiplier", ["%.4f" % m for m in mult], "model inputs", shapes_o, "boxes per scale", [len(b) for b in bbox_all])
    # ---- the whole coco_eval loop
    del CALLS[:], PRN_ARGS[:]
    me.coco_eval()
    for i, (kps, boxes, name, image_id) in enumerate(PRN_ARGS):
        out["prn_kps_%d" % i] = np.array(kps, dtype=np.float64).reshape(-1, 5)
        out["prn_boxes_%d" % i] = np.array(boxes, dtype=np.float64).reshape(-1, 4)
        out["prn_id_%d" % i] = np.array(image_id)
        print("coco_eval image %s: %d joints (neck removed), %d boxes" % (name, len(kps), len(boxes)))
    out["n_images"] = np.array(len(PRN_ARGS))
    np.savez_compressed(os.path.join(HERE, "g15_tta.npz"), **out)
    results = json.load(open(params.coco_result_filename))
    log = [list(e) if e[0] != "loadRes" else ["loadRes", len(e[1])] for e in EVAL_LOG]
    with open(os.path.join(HERE, "g15_tta_results.json"), "w") as f:
        json.dump({"results": results, "eval_calls": log, "file_kept": os.path.exists(params.coco_result_filename)}, f, indent=1)
    print("eval calls:", log)


if __name__ == "__main__":
    main()
