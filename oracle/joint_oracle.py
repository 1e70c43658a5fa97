"""CPU restatement of the reference's heat-map peak extraction (TEST INFRASTRUCTURE ONLY — the checker for
`mpn_heatmap_peaks`; nothing under multiposenet/ imports it).

Follows network/joint_utils.py:19-31 (`find_peaks`: 3x3-cross maximum filter == value and value > thre1),
:34-58 (`compute_resized_coords`) and :61-138 (`NMS`: per joint type, peaks in np.nonzero (row-major) order, optional
refinement on a bicubically up-sampled 5x5 patch, running peak id).

Pinning: `find_peaks` and `NMS(..., bool_refine_center=False)` are pinned bit-exactly to the REAL reference functions
(tests/golden/g10_peaks.npz, made by tests/golden/make_golden_peaks.py; they use only numpy/scipy).  The refinement
branch calls cv2.resize(INTER_CUBIC), and cv2 is not in this image: `cv_resize_cubic` below restates OpenCV's published
algorithm (float32, A = -0.75, source x = (dx + 0.5)/f - 0.5, taps clamped to the patch) — that branch is
PARITY UNPINNED until a box with cv2 regenerates the goldens (make_golden_peaks.py does it automatically when cv2
imports).
"""
import numpy as np

NUM_JOINTS = 18          # joint_utils.py:16


def find_peaks(thre1, img):
    """joint_utils.py:19-31.  scipy's maximum_filter default border mode is 'reflect': the out-of-range neighbour of an
    edge cell is the edge cell itself, so it never changes the maximum."""
    img = np.asarray(img)
    H, W = img.shape
    m = img.copy()
    m[1:, :] = np.maximum(m[1:, :], img[:-1, :])
    m[:-1, :] = np.maximum(m[:-1, :], img[1:, :])
    m[:, 1:] = np.maximum(m[:, 1:], img[:, :-1])
    m[:, :-1] = np.maximum(m[:, :-1], img[:, 1:])
    peaks_binary = (m == img) * (img > thre1)
    return np.array(np.nonzero(peaks_binary)[::-1]).T          # [[x, y], ...] in row-major order


def compute_resized_coords(coords, f):
    """joint_utils.py:34-58."""
    return (np.array(coords, dtype=float) + 0.5) * f - 0.5


def _cubic_coeffs(x):
    """OpenCV interpolateCubic (imgproc/resize.cpp), float32, A = -0.75."""
    A = np.float32(-0.75)
    x = np.float32(x)
    one = np.float32(1.0)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.array([c0, c1, c2, c3], dtype=np.float32)


def cv_resize_cubic(patch, f):
    """cv2.resize(patch, None, fx=f, fy=f, interpolation=cv2.INTER_CUBIC) for a small float32 patch, restated:
    dsize = round(size * f); per destination index d: s = (d + 0.5) / f - 0.5, taps floor(s) - 1 .. floor(s) + 2 clamped
    to the patch, horizontal pass then vertical pass, float32 accumulation in tap order."""
    patch = np.asarray(patch, dtype=np.float32)
    sh, sw = patch.shape
    dh, dw = int(round(sh * f)), int(round(sw * f))
    scale = 1.0 / f

    def taps(n_dst, n_src):
        idx = np.zeros((n_dst, 4), dtype=np.int64)
        co = np.zeros((n_dst, 4), dtype=np.float32)
        for d in range(n_dst):
            fx = np.float32((d + 0.5) * scale - 0.5)
            sx = int(np.floor(fx))
            co[d] = _cubic_coeffs(fx - np.float32(sx))
            for k in range(4):
                idx[d, k] = min(max(sx - 1 + k, 0), n_src - 1)
        return idx, co

    xi, xc = taps(dw, sw)
    yi, yc = taps(dh, sh)
    rows = np.zeros((sh, dw), dtype=np.float32)
    for d in range(dw):
        acc = patch[:, xi[d, 0]] * xc[d, 0]
        for k in range(1, 4):
            acc = acc + patch[:, xi[d, k]] * xc[d, k]
        rows[:, d] = acc
    out = np.zeros((dh, dw), dtype=np.float32)
    for d in range(dh):
        acc = rows[yi[d, 0], :] * yc[d, 0]
        for k in range(1, 4):
            acc = acc + rows[yi[d, k], :] * yc[d, k]
        out[d, :] = acc
    return out


def nms_peaks(thre1, heatmaps, upsamp=1.0, refine=True):
    """joint_utils.py:61-138.  heatmaps: [H, W, J].  Returns a list of J arrays [n, 4] = (x, y, score, id)."""
    heatmaps = np.asarray(heatmaps)
    out, cnt = [], 0
    win = 2
    for joint in range(heatmaps.shape[2]):
        m = heatmaps[:, :, joint]
        coords = find_peaks(thre1, m)
        peaks = np.zeros((len(coords), 4))
        for i, peak in enumerate(coords):
            if refine:
                x_min, y_min = np.maximum(0, peak - win)
                x_max, y_max = np.minimum(np.array(m.T.shape) - 1, peak + win)
                up = cv_resize_cubic(m[y_min:y_max + 1, x_min:x_max + 1], upsamp)
                loc = np.unravel_index(up.argmax(), up.shape)
                centre = compute_resized_coords(peak[::-1] - [y_min, x_min], upsamp)
                refined = (loc - centre)
                score = up[loc]
            else:
                refined = [0, 0]
                score = m[tuple(peak[::-1])]
            peaks[i, :] = tuple([int(round(x)) for x in compute_resized_coords(coords[i], upsamp) + refined[::-1]]) + (score, cnt)
            cnt += 1
        out.append(peaks)
    return out


def cv_resize(img, out_hw, cubic, inv_scale=None):
    """cv2.resize(img, (out_w, out_h), interpolation=INTER_CUBIC | INTER_LINEAR) for a float32 [H, W, C] array, restated
    (evaluate/tester.py:67,213,296-299 call sites).  Source coordinate s = (d + 0.5) * (src/dst) - 0.5; cubic: taps
    floor(s)-1..floor(s)+2 clamped; linear: OpenCV clamps the coordinate (s < 0 -> 0, s >= n-1 -> n-1, weight 0).
    Horizontal then vertical pass, float32 accumulation in tap order.  PARITY UNPINNED (cv2 is not in this image)."""
    img = np.asarray(img, dtype=np.float32)
    Hs, Ws, C = img.shape
    Hd, Wd = int(out_hw[0]), int(out_hw[1])

    def taps(n_dst, n_src, forced=None):
        # dsize form: scale = src / dst; fx / fy form (cv2.resize(img, None, fx=, fy=), tester.py:68): scale = 1 / f
        scale = float(n_src) / float(n_dst) if forced is None else float(forced)
        nt = 4 if cubic else 2
        idx = np.zeros((n_dst, nt), dtype=np.int64)
        co = np.zeros((n_dst, nt), dtype=np.float32)
        for d in range(n_dst):
            fx = np.float32((d + 0.5) * scale - 0.5)
            sx = int(np.floor(fx))
            if cubic:
                co[d] = _cubic_coeffs(fx - np.float32(sx))
                for k in range(4):
                    idx[d, k] = min(max(sx - 1 + k, 0), n_src - 1)
            else:
                fr = np.float32(fx - np.float32(sx))
                if sx < 0:
                    fr, sx = np.float32(0), 0
                if sx >= n_src - 1:
                    fr, sx = np.float32(0), n_src - 1
                idx[d] = (sx, min(sx + 1, n_src - 1))
                co[d] = (np.float32(1) - fr, fr)
        return idx, co

    xi, xc = taps(Wd, Ws, None if inv_scale is None else inv_scale[1])
    yi, yc = taps(Hd, Hs, None if inv_scale is None else inv_scale[0])
    rows = np.zeros((Hs, Wd, C), dtype=np.float32)
    for d in range(Wd):
        acc = img[:, xi[d, 0], :] * xc[d, 0]
        for k in range(1, xi.shape[1]):
            acc = acc + img[:, xi[d, k], :] * xc[d, k]
        rows[:, d, :] = acc
    out = np.zeros((Hd, Wd, C), dtype=np.float32)
    for d in range(Hd):
        acc = rows[yi[d, 0]] * yc[d, 0]
        for k in range(1, yi.shape[1]):
            acc = acc + rows[yi[d, k]] * yc[d, k]
        out[d] = acc
    return out
