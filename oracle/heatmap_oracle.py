"""CPU restatement of the reference's ground-truth heat-map generation (TEST INFRASTRUCTURE ONLY — nothing under
multiposenet/ imports this; it is the checker for `mpn_gt_heatmaps`).

Follows datasets/coco_data/heatmap.py:20-41 (`putGaussianMaps`: float64 grid at stride/2 - 0.5 offsets, exponent =
d2/2/sigma/sigma, cut at 4.6052, accumulate, clamp at 1.0) and the per-keypoint loop of
datasets/coco_data/COCO_data_pipeline.py:196-236 (`get_ground_truth`: for each of the 18 keypoint channels the annotated
person first, then the other people, each only if its visibility flag is <= 1; result cast to float32 by
COCO_data_pipeline.py:283-284).  Pinned against tests/golden/g9_gt_heatmaps.npz, which tests/golden/make_golden_gt.py
produced by calling the REAL `putGaussianMaps`.
"""
import numpy as np

CUTOFF = 4.6052          # heatmap.py:36


def put_gaussian_map(center, acc, crop_y, crop_x, stride, sigma):
    """heatmap.py:20-41, one keypoint into one float64 channel (in place, returned)."""
    grid_y, grid_x = int(crop_y / stride), int(crop_x / stride)
    start = stride / 2.0 - 0.5
    xx, yy = np.meshgrid(np.arange(grid_x), np.arange(grid_y))
    xx = xx * stride + start
    yy = yy * stride + start
    d2 = (xx - center[0]) ** 2 + (yy - center[1]) ** 2
    exponent = d2 / 2.0 / sigma / sigma
    mask = exponent <= CUTOFF
    acc += np.multiply(mask, np.exp(-exponent))
    acc[acc > 1.0] = 1.0
    return acc


def gt_heatmaps(joints, num_people, crop_y, crop_x, stride, sigma):
    """joints: float64 [B, maxP, 18, 3] (x, y, visibility), person 0 = the annotated one, then the others in order;
    num_people: int [B].  Returns float32 [B, 18, crop_y/stride, crop_x/stride] (COCO_data_pipeline.py:218-236, :283)."""
    joints = np.asarray(joints, dtype=np.float64)
    B = joints.shape[0]
    gh, gw = int(crop_y / stride), int(crop_x / stride)
    out = np.zeros((B, 18, gh, gw), dtype=np.float64)
    for b in range(B):
        for i in range(18):
            for j in range(int(num_people[b])):
                if joints[b, j, i, 2] <= 1:
                    put_gaussian_map(joints[b, j, i, :2], out[b, i], crop_y, crop_x, stride, sigma)
    return out.astype(np.float32)
