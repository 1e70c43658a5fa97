"""CPU restatement of the reference's PRN person assignment (TEST INFRASTRUCTURE ONLY — product code never imports this).

Follows evaluate/tester.py:333-513 (Tester.prn_process) with its helpers from datasets/coco_data/prn_gaussian.py:
``gaussian`` = skimage.filters.gaussian (:2; sigma 1, mode 'nearest', truncate 4 -> scipy.ndimage.gaussian_filter, pinned to
real skimage output by tests/golden/g12_prn_gaussian.npz) and ``crop`` (:134-158).  ``prn_forward(x[n,56,36,17]) -> [n,56,36,17]``
stands for ``self.model([input, 'prn_subnet'])`` (tester.py:401-404; one call per box there).
Pinned against outputs of the REAL Tester.prn_process (tests/golden/make_golden_prn_process.py -> g13_prn_process.npz).
"""
import math

import numpy as np
from scipy import ndimage as ndi


def gaussian(plane):
    """prn_gaussian.py:2 / tester.py:397 — skimage.filters.gaussian defaults."""
    return ndi.gaussian_filter(np.asarray(plane, dtype=np.float64), 1, mode="nearest", truncate=4.0)


def crop(img, c, N=13):
    """prn_gaussian.py:134-158."""
    H = img.shape[1]
    W = img.shape[0]
    h = (N - 1) / 2
    x1 = int(c[0] - h)
    y1 = int(c[1] - h)
    x2 = int(c[0] + h) + 1
    y2 = int(c[1] + h) + 1
    x1 = max(x1, 0)
    y1 = max(y1, 0)
    if x2 > W - 1:
        x2 = W
    if y2 > H - 1:
        y2 = H
    return img[x1:x2, y1:y2]


def build_maps(kps, bbox_list, coeff=2, in_thres=0.21):
    """tester.py:337-397: returns (peaks, bboxes, old_weights_bbox, blurred input [n,h,w,17] float64)."""
    idx = 0
    peaks = []
    for j in range(17):                                   # :339-349
        tl = []
        for k in kps:
            if k[-1] == j:
                tl.append([k[0], k[1], 1, idx])
                idx += 1
        peaks.append(tl)
    w = int(18 * coeff)
    h = int(28 * coeff)
    bboxes = [[b[0], b[1], b[2] - b[0], b[3] - b[1]] for b in bbox_list]      # :355-357
    weights_bbox = np.zeros((len(bboxes), h, w, 4, 17))
    for joint_id, peak in enumerate(peaks):               # :363-392
        for instance in peak:
            p_x, p_y = instance[0], instance[1]
            for bbox_id, b in enumerate(bboxes):
                inside = (p_x > b[0] - b[2] * in_thres and p_y > b[1] - b[3] * in_thres and
                          p_x < b[0] + b[2] * (1.0 + in_thres) and p_y < b[1] + b[3] * (1.0 + in_thres))
                if not inside:
                    continue
                x_scale = float(w) / math.ceil(b[2])
                y_scale = float(h) / math.ceil(b[3])
                x0 = int((p_x - b[0]) * x_scale)
                y0 = int((p_y - b[1]) * y_scale)
                if x0 >= w and y0 >= h:
                    x0, y0 = w - 1, h - 1
                elif x0 >= w:
                    x0 = w - 1
                elif y0 >= h:
                    y0 = h - 1
                elif x0 < 0 and y0 < 0:
                    x0, y0 = 0, 0
                elif x0 < 0:
                    x0 = 0
                elif y0 < 0:
                    y0 = 0
                weights_bbox[bbox_id, y0, x0, :, joint_id] = [1, instance[2], instance[3], 1e-9]
    old = np.copy(weights_bbox)                            # :393
    inp = np.zeros((len(bboxes), h, w, 17))
    for j in range(len(bboxes)):                           # :395-397
        for t in range(17):
            inp[j, :, :, t] = gaussian(weights_bbox[j, :, :, 0, t])
    return peaks, bboxes, old, inp


def assign(peaks, bboxes, old, output_bbox, coeff=2):
    """tester.py:410-485: window scores, the greedy bbox <-> keypoint matching, and the arg-max fallback.
    Returns bbox_keypoints [n,17,3]."""
    w = int(18 * coeff)
    h = int(28 * coeff)
    n = len(bboxes)
    keypoints_score = []
    for t in range(17):                                    # :412-430
        keypoint = []
        for i in np.argwhere(old[:, :, :, 0, t] == 1):
            cr = crop(output_bbox[i[0], :, :, t], (i[1], i[2]), N=15)
            score = np.sum(cr)
            kp_id = old[i[0], i[1], i[2], 2, t]
            kp_score = old[i[0], i[1], i[2], 1, t]
            keypoint.append([kp_id, i[0], kp_score, kp_score * score])
        keypoints_score.append(keypoint)
    bbox_keypoints = np.zeros((n, 17, 3))
    bbox_ids = list(range(n))
    for i in range(17):                                    # :436-483
        joint_keypoints = keypoints_score[i]
        if len(joint_keypoints) > 0:
            kp_ids = list(set([x[0] for x in joint_keypoints]))
            table = np.zeros((len(bbox_ids), len(kp_ids), 4))
            for bbox in bbox_ids:
                for k_id, kp in enumerate(kp_ids):
                    own = [x for x in joint_keypoints if x[0] == kp and x[1] == bbox]
                    table[bbox, k_id] = own[0] if len(own) > 0 else [0] * 4
            for bbox in bbox_ids:
                row = np.argsort(-table[bbox, :, 3])
                if table[bbox, row[0], 3] > 0:
                    for r in row:
                        if table[bbox, r, 3] > 0:
                            column = np.argsort(-table[:, r, 3])
                            if bbox == column[0]:
                                bbox_keypoints[bbox, i, :] = [x[:3] for x in peaks[i] if x[3] == table[bbox, r, 0]][0]
                                break
                            else:
                                row2 = np.argsort(table[column[0], :, 3])
                                if row2[0] == r:
                                    bbox_keypoints[bbox, i, :] = [x[:3] for x in peaks[i] if x[3] == table[bbox, r, 0]][0]
                                    break
        else:
            for j in range(n):
                b = bboxes[j]
                x_scale = float(w) / math.ceil(b[2])
                y_scale = float(h) / math.ceil(b[3])
                for t in range(17):
                    if len(np.argwhere(old[j, :, :, 0, t] == 1)) == 0:
                        mi = np.argwhere(output_bbox[j, :, :, t] == np.max(output_bbox[j, :, :, t]))
                        bbox_keypoints[j, t, :] = [mi[0][1] / x_scale + b[0], mi[0][0] / y_scale + b[1], 0]
    return bbox_keypoints


def prn_process(prn_forward, kps, bbox_list, file_name="", image_id=0, coeff=2, in_thres=0.21):
    """tester.py:333-513 end to end -> the reference's list of result dicts."""
    prn_result = []
    peaks, bboxes, old, inp = build_maps(kps, bbox_list, coeff, in_thres)
    if len(bboxes) == 0 or len(peaks) == 0:                # :359-360
        return prn_result
    out = np.asarray(prn_forward(inp.astype(np.float32)), dtype=np.float32).reshape(len(bboxes), 56, 36, 17)
    bk = assign(peaks, bboxes, old, out, coeff)
    for i in range(bk.shape[0]):                           # :487-511
        k = np.zeros(51)
        k[0::3], k[1::3], k[2::3] = bk[i, :, 0], bk[i, :, 1], bk[i, :, 2]
        pose_score = 0
        for f in range(17):
            pose_score += bk[i, f, 2]
        pose_score /= 17.0
        prn_result.append({"image_id": image_id, "file_name": file_name, "category_id": 1, "bbox": bboxes[i],
                           "score": pose_score, "keypoints": k.tolist()})
    return prn_result
