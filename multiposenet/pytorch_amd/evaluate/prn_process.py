"""PRN person assignment — the reference's ``Tester.prn_process`` (evaluate/tester.py:333-513) with the per-box work on
the MI355X and ONE batched PRN forward for all boxes (of all images) instead of a batch-1 call per box (:399-406).

    results = prn_process(model, kps, bbox_list, file_name, image_id)                  # same arguments / dict list as the reference
    per_img = prn_process_batch(model, [kps_0, kps_1, ...], [boxes_0, boxes_1, ...])   # a whole batch in the same three launches

``kps`` rows are ``(x, y, score, id, joint_type 0..16)`` as produced at tester.py:158-166 from ``get_joint_list``;
``bbox_list`` rows are ``(x1, y1, x2, y2)``.  Device side (csrc/prn_assign.hip): the one-hot peak maps with the
reference's cell arithmetic, the skimage-gaussian blur, the PRN (``model.prn_forward`` on the conv kernels), the 15x15
window scores — compacted on the device to per-plane candidate lists — and the per-plane arg-max.  Host side: the greedy table
matching of tester.py:432-470 in C++ (``mpn_prn_match_host``) on those lists; wherever the result would depend on how numpy orders
EQUAL scores (``np.argsort``'s tie order is a property of the numpy build: its small-array sorts are not stable) the full score
tables come back and the reference's own numpy expressions decide (``_assign``).  ``prn_assign_arrays`` is the batched entry
point on flat arrays; ``prn_process`` / ``prn_process_batch`` keep the reference's list / dict interface on top of it.
"""
import ctypes
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


CAND_CAP = 64          # candidates kept per (box, joint type) plane by the compact score kernel; more -> the full-table path


def prn_assign_arrays(model, peaks_xy, joint_off, boxes_xywh, box_start, in_thres=0.21, fast=True):
    """The assignment for a whole batch on flat arrays — the form the device kernels take and the form a batched caller has.

    peaks_xy  float64 [Np, 2]   heat-map peaks (x, y), grouped by image, then by joint type 0..16, in detection order
    joint_off int32  [nimg, 18] offsets of those groups into peaks_xy
    boxes_xywh float64 [nb, 4]  detected boxes (x, y, w, h), grouped by image
    box_start int32  [nimg + 1] boxes of image i = box_start[i] .. box_start[i + 1]
    Returns keypoints float64 [nb, 17, 3] (x, y, score) — ``bbox_keypoints`` of tester.py:410-485 for every box.

    Device: one-hot maps + blur (mpn_prn_build_maps), ONE PRN forward for all boxes, window scores compacted to per-plane candidate
    lists (mpn_prn_scores_compact: a few bytes per candidate cross PCIe instead of 274 KB of tables per box).  Host: the greedy
    matching in C++ (mpn_prn_match_host).  (image, joint type) pairs whose result would hinge on numpy's ordering of equal scores,
    and planes with more than CAND_CAP candidates, take the full-table numpy path below (``fast=False`` forces it everywhere)."""
    h, w = 56, 36
    dev = next(model.parameters()).device
    nimg = int(box_start.shape[0]) - 1
    nb = int(boxes_xywh.shape[0])
    out_kp = np.zeros((nb, 17, 3))
    if nb == 0:
        return out_kp
    if np.any(np.ceil(boxes_xywh[:, 2]) == 0) or np.any(np.ceil(boxes_xywh[:, 3]) == 0):
        raise ZeroDivisionError("box with zero width/height (the reference divides by ceil(w), tester.py:374)")
    box_img = np.repeat(np.arange(nimg, dtype=np.int32), np.diff(box_start)).astype(np.int32)
    peaks_host = np.ascontiguousarray(peaks_xy, dtype=np.float64) if peaks_xy.shape[0] else np.zeros((1, 2))
    peaks_t = torch.from_numpy(peaks_host).to(dev)
    off_t = torch.from_numpy(np.ascontiguousarray(joint_off, dtype=np.int32)).to(dev)
    boxes_t = torch.from_numpy(np.ascontiguousarray(boxes_xywh, dtype=np.float64)).to(dev)
    bimg_t = torch.from_numpy(box_img).to(dev)
    w9_t = torch.from_numpy(_W9).to(dev)
    occ = torch.empty((nb, 17, h, w), dtype=torch.int32, device=dev)
    prn_in = torch.empty((nb, h, w, 17), dtype=torch.float32, device=dev)
    err = torch.zeros(2, dtype=torch.int32, device=dev)             # [0]: IndexError of the clamp chain, [1]: candidate overflow
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

    def full_tables(b0, b1, img):
        """tester.py:410-485 through the full score / occupancy tables of boxes b0..b1 (numpy; exact in every tie)."""
        n = b1 - b0
        score = torch.zeros((n, 17, h, w), dtype=torch.float32, device=dev)
        amax = torch.empty((n, 17), dtype=torch.int32, device=dev)
        call("mpn_prn_scores", ops.ptr(out[b0:b1]), ops.ptr(occ[b0:b1]), n, h, w, 15, ops.ptr(score), ops.ptr(amax), ops.stream_ptr())
        offs = joint_off[img]
        peaks = [[[peaks_host[q, 0], peaks_host[q, 1], 1, q - offs[0]] for q in range(offs[t], offs[t + 1])] for t in range(17)]
        return _assign(peaks, [list(bx) for bx in boxes_xywh[b0:b1]], occ[b0:b1].cpu().numpy(), score.cpu().numpy(), amax.cpu().numpy(), w, h)

    if not fast:
        if int(err[0].item()):
            raise IndexError("a peak falls outside the 56x36 map after the reference's clamp chain (tester.py:376-392 raises here too)")
        for i in range(nimg):
            if box_start[i + 1] > box_start[i]:
                out_kp[box_start[i]:box_start[i + 1]] = full_tables(int(box_start[i]), int(box_start[i + 1]), i)
        return out_kp
    cap = CAND_CAP
    cand_n = torch.empty((nb, 17), dtype=torch.int32, device=dev)
    cand_id = torch.empty((nb, 17, cap), dtype=torch.int32, device=dev)
    cand_sc = torch.empty((nb, 17, cap), dtype=torch.float32, device=dev)
    amax = torch.empty((nb, 17), dtype=torch.int32, device=dev)
    call("mpn_prn_scores_compact", ops.ptr(out), ops.ptr(occ), nb, h, w, 15, cap, ops.ptr(cand_n), ops.ptr(cand_id), ops.ptr(cand_sc),
         ops.ptr(amax), ctypes.c_void_p(err.data_ptr() + 4), ops.stream_ptr())
    cand_n_h, cand_id_h, cand_sc_h, amax_h, err_h = cand_n.cpu().numpy(), cand_id.cpu().numpy(), cand_sc.cpu().numpy(), amax.cpu().numpy(), err.cpu().numpy()
    if err_h[0]:
        raise IndexError("a peak falls outside the 56x36 map after the reference's clamp chain (tester.py:376-392 raises here too)")
    ties = np.zeros((nimg, 17), dtype=np.uint8)
    bs = np.ascontiguousarray(box_start, dtype=np.int32)
    bx = np.ascontiguousarray(boxes_xywh, dtype=np.float64)
    jo = np.ascontiguousarray(joint_off, dtype=np.int32)

    def hp(a):
        return ctypes.c_void_p(a.ctypes.data)
    call("mpn_prn_match_host", nimg, hp(bs), hp(bx), hp(jo), hp(peaks_host), hp(cand_n_h), hp(cand_id_h), hp(cand_sc_h), cap, hp(amax_h), h, w,
         hp(out_kp), hp(ties))
    redo = set(np.nonzero(ties.any(axis=1))[0].tolist())
    if err_h[1]:                                               # a plane with more candidates than the compact list holds
        over = np.nonzero((cand_n_h > cap).any(axis=1))[0]
        redo.update(int(box_img[b]) for b in over)
    for i in sorted(redo):
        out_kp[bs[i]:bs[i + 1]] = full_tables(int(bs[i]), int(bs[i + 1]), i)
    return out_kp


def prn_process_batch(model, kps_list, bbox_lists, file_names=None, image_ids=None, coeff=2, in_thres=0.21, fast=True):
    nimg = len(kps_list)
    file_names = file_names if file_names is not None else [""] * nimg
    image_ids = image_ids if image_ids is not None else [0] * nimg
    w, h = int(18 * coeff), int(28 * coeff)
    if (h, w) != (56, 36):
        raise MpnError("the reference reshapes the PRN output to (56, 36, 17) (tester.py:404): coeff must be 2")
    boxes_all = [[[b[0], b[1], b[2] - b[0], b[3] - b[1]] for b in bl] for bl in bbox_lists]        # tester.py:355-357
    results = [[] for _ in range(nimg)]
    # tester.py:337-349 without a Python loop per peak (round 5): per image one list -> array conversion, the joint types grouped by a
    # STABLE sort (the reference's per-type scans keep the input order inside a type), offsets by bincount
    flat_peaks, joint_off, box_start = [], np.zeros((nimg, 18), dtype=np.int64), [0]
    base = 0
    for i in range(nimg):
        a = np.asarray(kps_list[i], dtype=np.float64)
        a = a.reshape(-1, a.shape[-1]) if a.size else np.zeros((0, 5))
        jt = a[:, -1]
        a = a[(jt == np.floor(jt)) & (jt >= 0) & (jt < 17)]
        jt = a[:, -1].astype(np.int64)
        flat_peaks.append(a[np.argsort(jt, kind="stable"), :2])
        joint_off[i, 0] = base
        joint_off[i, 1:] = base + np.cumsum(np.bincount(jt, minlength=17))
        base = int(joint_off[i, 17])
        box_start.append(box_start[-1] + len(boxes_all[i]))
    if box_start[-1] == 0:
        return results                                     # tester.py:359-360
    flat_boxes = [b for bl in boxes_all for b in bl]
    bk = prn_assign_arrays(model, np.concatenate(flat_peaks, 0) if flat_peaks else np.zeros((0, 2)), joint_off.astype(np.int32),
                           np.asarray(flat_boxes, dtype=np.float64).reshape(-1, 4), np.asarray(box_start, dtype=np.int32), in_thres, fast)
    # tester.py:487-511; the pose score is the reference's left-to-right float sum over the 17 joints
    kp51 = bk.reshape(bk.shape[0], 51).tolist()
    for i in range(nimg):
        for bi in range(box_start[i], box_start[i + 1]):
            pose_score = 0
            for f in range(17):
                pose_score += bk[bi, f, 2]
            pose_score /= 17.0
            results[i].append({'image_id': image_ids[i], 'file_name': file_names[i], 'category_id': 1, 'bbox': boxes_all[i][bi - box_start[i]],
                               'score': pose_score, 'keypoints': kp51[bi]})
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
