"""Ground-truth heat-map targets on the GPU.

Device replacement for the CPU target generation in the reference's data loader:
``datasets/coco_data/heatmap.py:20-41`` (``putGaussianMaps``) and the per-keypoint loop of
``datasets/coco_data/COCO_data_pipeline.py:218-236`` (``get_ground_truth``), batched.  The loader keeps doing the
image decoding / augmentation; it hands over the augmented keypoints and the 46 x 18 Gaussians per image are
rendered where the loss will read them.
"""
import torch

from .._lib import MpnError, call
from .. import ops

# datasets/coco_data/COCO_data_pipeline.py:42 and the feat_stride / inp_size the trainers pass (:77-79)
DEFAULT_SIGMA = 7.0
DEFAULT_STRIDE = 4


def put_gaussian_maps(joints, num_people, crop_size_y, crop_size_x, stride=DEFAULT_STRIDE, sigma=DEFAULT_SIGMA):
    """Heat-map targets ``[B, 18, crop_size_y // stride, crop_size_x // stride]`` (float32, on the device of ``joints``).

    joints: float64 tensor ``[B, maxP, 18, 3]`` = (x, y, visibility) in crop pixels, person 0 the annotated one
            (``joint_self``), then ``joint_others`` in order; visibility > 1 means "not annotated" and is skipped
            (COCO_data_pipeline.py:225,230).
    num_people: int32 tensor ``[B]`` = 1 + numOtherPeople (0 for an image without annotations).
    """
    if not joints.is_cuda:
        raise MpnError("put_gaussian_maps runs on the MI355X only; there is no CPU path")
    if joints.dtype != torch.float64 or joints.dim() != 4 or joints.shape[2] != 18 or joints.shape[3] != 3:
        raise MpnError("joints must be a float64 [B, maxP, 18, 3] tensor")
    joints = joints.contiguous()
    num_people = num_people.to(device=joints.device, dtype=torch.int32).contiguous()
    B, maxP = joints.shape[0], joints.shape[1]
    if num_people.shape != (B,):
        raise MpnError("num_people must have shape [B]")
    gh, gw = int(crop_size_y / stride), int(crop_size_x / stride)          # heatmap.py:26-27
    out = torch.empty((B, 18, gh, gw), dtype=torch.float32, device=joints.device)
    call("mpn_gt_heatmaps", ops.ptr(joints), ops.ptr(num_people), B, maxP, ops.ptr(out), gh, gw, float(stride), float(sigma),
         ops.stream_ptr())
    return out
