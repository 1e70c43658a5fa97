"""BBoxTransform / ClipBoxes with the reference's interface (``network/utils.py:6-61``).

The reference runs ~20 tiny elementwise kernels (decode) plus four in-place clamps (clip); here one
fused HIP kernel (csrc/nms.hip: box_decode_clip_kernel) does both when used through
``decode_and_clip``; the two modules below keep the individual call signatures.
"""
import torch
import torch.nn as nn

from .. import ops


def decode_and_clip(anchors, deltas, img):
    """clipBoxes(regressBoxes(anchors, deltas), img)  — posenet.py:266-267 — in one launch."""
    _, _, h, w = img.shape
    return ops.box_decode_clip(anchors.reshape(-1, 4).contiguous(), deltas.detach().float().contiguous(), float(w), float(h))


class BBoxTransform(nn.Module):
    """forward(boxes[1,A,4], deltas[B,A,4]) -> [B,A,4]; deltas * std + mean with mean 0, std [.1,.1,.2,.2] unless given (utils.py:8-17)."""

    def __init__(self, mean=None, std=None):
        super(BBoxTransform, self).__init__()
        # utils.py:10-17: optional 4-vectors (tensors / arrays); the defaults take the entry point with the constants compiled in
        self.mean_std = None
        if mean is not None or std is not None:
            m = [0.0] * 4 if mean is None else [float(v) for v in torch.as_tensor(mean).flatten().tolist()]
            s_ = [0.1, 0.1, 0.2, 0.2] if std is None else [float(v) for v in torch.as_tensor(std).flatten().tolist()]
            if len(m) != 4 or len(s_) != 4:
                raise ValueError("BBoxTransform: mean and std are 4-vectors")
            self.mean_std = m + s_

    def forward(self, boxes, deltas):
        # no clipping: pass an unbounded image size
        return ops.box_decode_clip(boxes.reshape(-1, 4).contiguous(), deltas.detach().float().contiguous(), 0.0, 0.0, clip=False,
                                   mean_std=self.mean_std)


class ClipBoxes(nn.Module):
    """forward(boxes[B,A,4], img) clamps x1,y1 >= 0, x2 <= W, y2 <= H in place (utils.py:51-61)."""

    def forward(self, boxes, img):
        _, _, h, w = img.shape
        ops.clip_boxes_(boxes, float(w), float(h))
        return boxes
