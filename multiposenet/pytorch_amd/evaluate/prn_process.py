"""PRN person assignment — the reference's ``Tester.prn_process`` (evaluate/tester.py:333-513) with the per-box work on
the MI355X and ONE batched PRN forward for all boxes (of all images) instead of a batch-1 call per box (:399-406).

    results = prn_process(model, kps, bbox_list, file_name, image_id)                  # same arguments / dict list as the reference
    per_img = prn_process_batch(model, [kps_0, kps_1, ...], [boxes_0, boxes_1, ...])   # a whole batch in the same three launches

``kps`` rows are ``(x, y, score, id, joint_type 0..16)`` as produced at tester.py:158-166 from ``get_joint_list``;
``bbox_list`` rows are ``(x1, y1, x2, y2)``.  Device side (csrc/prn_assign.hip): the one-hot peak maps with the
reference's cell arithmetic, the skimage-gaussian blur, the PRN (``model.prn_forward`` on the conv kernels), the 15x15
window scores and the per-plane arg-max.  Host side (this file): the tiny greedy table matching of tester.py:432-470 on
the (boxes x peaks) score table — integer/argsort logic over a few dozen numbers, kept in numpy so that ties resolve
exactly as in the reference (``list(set(...))`` order, ``np.argsort``'s default sort).
"""
import math

import numpy as np
import torch

from .. import ops
from .._lib import MpnError, call

_X = np.arange(-4, 5)
_W9 = np.exp(-0.5 / 1.0 * _X ** 2)
_W9 = _W9 / _W9.sum()                 # scipy.ndimage._gaussian_kernel1d(sigma=1, order=0, radius=4)


def _peaks_by_joint(kps):
    """tester.py:337-349: per joint type the list [x, y, 1, idx] with a running idx over (joint type, input order)."""
    peaks = [[] for _ in range(17)]
    for k in kps:                         # one pass in input order; the reference scans the list once per joint type
        j = k[-1]
        if j == int(j) and 0 <= j < 17:
            peaks[int(j)].append([k[0], k[1], 1, 0])
    idx = 0
    for tl in peaks:
        for p in tl:
            p[3] = idx
            idx += 1
    return peaks


def prn_process_batch(model, kps_list, bbox_lists, file_names=None, image_ids=None, coeff=2, in_thres=0.21):
    nimg = len(kps_list)
    file_names = file_names if file_names is not None else [""] * nimg
    image_ids = image_ids if image_ids is not None else [0] * nimg
    w, h = int(18 * coeff), int(28 * coeff)
    if (h, w) != (56, 36):
        raise MpnError("the reference reshapes the PRN output to (56, 36, 17) (tester.py:404): coeff must be 2")
    dev = next(model.parameters()).device
    peaks_all = [_peaks_by_joint(k) for k in kps_list]
    boxes_all = [[[b[0], b[1], b[2] - b[0], b[3] - b[1]] for b in bl] for bl in bbox_lists]        # tester.py:355-357
    results = [[] for _ in range(nimg)]
    # flat device inputs: peaks grouped by image then joint type, boxes with their image index
    flat_peaks, joint_off, flat_boxes, box_img, box_slices = [], [], [], [], []
    for i in range(nimg):
        offs = []
        for j in range(17):
            offs.append(len(flat_peaks))
            flat_peaks += [(p[0], p[1]) for p in peaks_all[i][j]]
        offs.append(len(flat_peaks))
        joint_off.append(offs)
        box_slices.append((len(flat_boxes), len(flat_boxes) + len(boxes_all[i])))
        for b in boxes_all[i]:
            if math.ceil(b[2]) == 0 or math.ceil(b[3]) == 0:
                raise ZeroDivisionError("box with zero width/height (the reference divides by ceil(w), tester.py:374)")
            flat_boxes.append(b)
            box_img.append(i)
    nb = len(flat_boxes)
    if nb == 0:
        return results                                     # tester.py:359-360
    peaks_t = torch.tensor(flat_peaks if flat_peaks else [[0.0, 0.0]], dtype=torch.float64).to(dev)
    off_t = torch.tensor(joint_off, dtype=torch.int32).to(dev)
    boxes_t = torch.tensor(flat_boxes, dtype=torch.float64).to(dev)
    bimg_t = torch.tensor(box_img, dtype=torch.int32).to(dev)
    w9_t = torch.from_numpy(_W9).to(dev)
    occ = torch.empty((nb, 17, h, w), dtype=torch.int32, device=dev)
    prn_in = torch.empty((nb, h, w, 17), dtype=torch.float32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    call("mpn_prn_build_maps", ops.ptr(peaks_t), ops.ptr(off_t), ops.ptr(boxes_t), ops.ptr(bimg_t), nb, h, w, float(in_thres), ops.ptr(w9_t),
         ops.ptr(occ), ops.ptr(prn_in), ops.ptr(err), ops.stream_ptr())
    was_training = model.prn.training
    model.prn.eval()
    try:
        with torch.no_grad():
            out, _ = model([prn_in, 'prn_subnet'])           # ONE forward for every box (tester.py:399-406 loops at batch 1)
    finally:
        model.prn.train(was_training)
    out = out.detach().float().contiguous()
    score = torch.zeros((nb, 17, h, w), dtype=torch.float32, device=dev)
    amax = torch.empty((nb, 17), dtype=torch.int32, device=dev)
    call("mpn_prn_scores", ops.ptr(out), ops.ptr(occ), nb, h, w, 15, ops.ptr(score), ops.ptr(amax), ops.stream_ptr())
    occ_h, score_h, amax_h, err_h = occ.cpu().numpy(), score.cpu().numpy(), amax.cpu().numpy(), int(err.item())
    if err_h:
        raise IndexError("a peak falls outside the 56x36 map after the reference's clamp chain (tester.py:376-392 raises here too)")
    for i in range(nimg):
        s, e = box_slices[i]
        if e == s:
            continue
        bk = _assign(peaks_all[i], boxes_all[i], occ_h[s:e], score_h[s:e], amax_h[s:e], w, h)
        for bi in range(e - s):                              # tester.py:487-511
            k = np.zeros(51)
            k[0::3], k[1::3], k[2::3] = bk[bi, :, 0], bk[bi, :, 1], bk[bi, :, 2]
            pose_score = 0
            for f in range(17):
                pose_score += bk[bi, f, 2]
            pose_score /= 17.0
            results[i].append({'image_id': image_ids[i], 'file_name': file_names[i], 'category_id': 1, 'bbox': boxes_all[i][bi],
                               'score': pose_score, 'keypoints': k.tolist()})
    return results


def _assign(peaks, bboxes, occ, score, amax, w, h):
    """tester.py:410-485 on the device-made tables: occ [n,17,h,w] (1 + peak id), score [n,17,h,w], amax [n,17]."""
    n = len(bboxes)
    bbox_keypoints = np.zeros((n, 17, 3))
    by_id = [{p[3]: p for p in peaks[t]} for t in range(17)]
    any_empty_type = False
    cand = []
    for t in range(17):
        rows = []
        for bi, y, x in np.argwhere(occ[:, t] > 0):          # (box, y, x) ascending, like np.argwhere at tester.py:415
            kp_id = float(occ[bi, t, y, x] - 1)
            rows.append((kp_id, int(bi), float(score[bi, t, y, x])))     # kp_score is always 1 (tester.py:345,427)
        cand.append(rows)
        any_empty_type = any_empty_type or len(rows) == 0
    for t in range(17):
        rows = cand[t]
        if not rows:
            continue
        kp_ids = list(set(r[0] for r in rows))               # the reference's column order (tester.py:440)
        col = {kp: c for c, kp in enumerate(kp_ids)}
        table = np.zeros((n, len(kp_ids)))
        ids = np.zeros((n, len(kp_ids)))
        for kp_id, bi, sc in rows:
            table[bi, col[kp_id]] = sc
            ids[bi, col[kp_id]] = kp_id
        for bbox in range(n):                                # tester.py:453-470
            row = np.argsort(-table[bbox])
            if table[bbox, row[0]] > 0:
                for r in row:
                    if table[bbox, r] > 0:
                        column = np.argsort(-table[:, r])
                        if bbox == column[0] or np.argsort(table[column[0]])[0] == r:
                            bbox_keypoints[bbox, t, :] = by_id[t][ids[bbox, r]][:3]
                            break
    if any_empty_type:                                       # tester.py:471-483 (runs for every joint type without candidates; idempotent)
        for j in range(n):
            b = bboxes[j]
            x_scale = float(w) / math.ceil(b[2])
            y_scale = float(h) / math.ceil(b[3])
            for t in range(17):
                if not (occ[j, t] > 0).any():
                    my, mx = divmod(int(amax[j, t]), w)
                    bbox_keypoints[j, t, :] = [mx / x_scale + b[0], my / y_scale + b[1], 0]
    return bbox_keypoints


def prn_process(model, kps, bbox_list, file_name="", image_id=0, coeff=2, in_thres=0.21):
    """tester.py:333: one image."""
    return prn_process_batch(model, [kps], [bbox_list], [file_name], [image_id], coeff, in_thres)[0]
