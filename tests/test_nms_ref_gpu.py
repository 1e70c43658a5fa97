"""A13, kernel leg pinned to the REFERENCE'S OWN SOURCE: `oracle/_ref/libref_nms_kernel*.so` is
/root/reference/lib/nms/src/cuda/nms_kernel.cu compiled unmodified with hipcc (`make -C oracle ref`).  The product's
`nms_mask_kernel` (nms.hip) and the C restatement (oracle/nms_oracle.c) must produce the same 64-bit mask words as the
reference's `nms_kernel` for every upper-triangle tile (the words gpu_nms's scan reads, nms_cuda.c:47-58), and the keep lists
must agree through that scan.  `nms_cuda.c` / `nms.c` themselves need <TH/TH.h> and stay restated (oracle/ref_nms.host_scan,
oracle/nms_oracle.c).

Also settles DESIGN §4's contraction caveat with a number: the reference kernel is built twice (compiler-default contraction —
the analogue of nvcc's --fmad=true — and -ffp-contract=off) and the differing mask words are counted.
"""
import numpy as np
import pytest
import torch

from helpers import gold, report

pytestmark = pytest.mark.gpu


def _align(v, a=256):
    return (v + a - 1) // a * a


def _product(d_np, thr):
    """product NMS through the C ABI; returns (keep, sorted rows [n,5], order [n], mask [n,cb] u64) read from its workspace."""
    from multiposenet.pytorch_amd import _lib, ops
    n = d_np.shape[0]
    cb = (n + 63) // 64
    d = torch.from_numpy(d_np).cuda()
    keep = ops.nms(d, thr, 0)
    nbytes = _lib.call("mpn_nms_workspace_bytes", n)
    ws = ops.workspace(nbytes, d.device, slot=4)
    torch.cuda.synchronize()
    raw = ws[:nbytes].cpu().numpy()
    off_order = _align(n * 20)
    off_mask = off_order + _align(n * 4) + _align(cb * 8)
    srt = raw[: n * 20].view(np.float32).reshape(n, 5).copy()
    order = raw[off_order: off_order + n * 4].view(np.int32).copy()
    # round 6: the workspace holds the UPPER TRIANGLE only (nms.hip: nms_tri_row) — row r keeps the words of column tiles r // 64 .. cb - 1;
    # unpack into the dense [n, cb] form the reference kernel writes (lower half zero: the product never computes it)
    tri = raw[off_mask: off_mask + 64 * (cb * (cb + 1) // 2) * 8].view(np.uint64)
    assert nbytes == off_mask + _align(64 * (cb * (cb + 1) // 2) * 8), "workspace query does not match the packed layout"
    mask = np.zeros((n, cb), dtype=np.uint64)
    for rb in range(cb):
        r0, r1 = rb * 64, min(n, rb * 64 + 64)
        base = 64 * (rb * cb - rb * (rb - 1) // 2)
        w = cb - rb
        mask[r0:r1, rb:] = tri[base: base + (r1 - r0) * w].reshape(r1 - r0, w)
    return keep.cpu().numpy(), srt, order, mask


def _upper(mask):
    """zero the words below the diagonal tile (never read by the scan; the product does not write them)."""
    n, cb = mask.shape
    rows = (np.arange(n) // 64)[:, None]
    return np.where(np.arange(cb)[None, :] >= rows, mask, np.uint64(0))


def _near_threshold_pairs(rs, npairs, max_exp=10):
    """pairs whose exact IoU (+1 convention) is 1/2 — b is a shifted by a third of the width — translated by a random
    non-integer offset with every coordinate moved by a few float32 steps, so that the computed IoU lands within ulps of 0.5 on
    either side.  The pairs of a set overlap each other freely: every word of the mask is compared, the designated pairs only
    feed the above / not-above count."""
    k = rs.randint(5, 100, npairs)
    W = 3 * k
    H = rs.randint(10, 200, npairs)
    tx = 2.0 ** rs.uniform(0, max_exp, npairs)
    ty = 2.0 ** rs.uniform(0, max_exp, npairs)
    exact = rs.uniform(size=npairs) < 0.1                           # a tenth at integer offsets: IoU == 0.5 exactly, NOT > 0.5
    tx[exact] = np.floor(tx[exact]); ty[exact] = np.floor(ty[exact])
    a = np.stack([tx, ty, tx + W - 1, ty + H - 1], 1)
    b = a.copy(); b[:, 0] += k; b[:, 2] += k
    boxes = np.concatenate([a, b], 0).astype(np.float32)
    # a and b share their fractional parts, so their float32 rounding errors cancel in every difference: move every coordinate
    # of the non-exact pairs by -3..3 float32 steps independently (offsets 2^0..2^max_exp, so a step is 1e-7..6e-5 px)
    j = rs.randint(-3, 4, boxes.shape).astype(np.float32)
    j[np.concatenate([exact, exact])] = 0
    boxes = boxes + j * np.spacing(boxes)
    sc = (rs.permutation(2 * npairs).astype(np.float64) / (2 * npairs)).astype(np.float32)
    return np.concatenate([boxes, sc[:, None]], 1)


def _ulps_from_half(d):
    """float32 steps between the (uncontracted) devIoU of pair (p, p + h) and 0.5 — reporting only."""
    h = d.shape[0] // 2
    a, b, one = d[:h, :4], d[h:, :4], np.float32(1)
    w = np.maximum(np.minimum(a[:, 2], b[:, 2]) - np.maximum(a[:, 0], b[:, 0]) + one, 0)
    hh = np.maximum(np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1]) + one, 0)
    inter = w * hh
    sa = (a[:, 2] - a[:, 0] + one) * (a[:, 3] - a[:, 1] + one)
    sb = (b[:, 2] - b[:, 0] + one) * (b[:, 3] - b[:, 1] + one)
    iou = (inter / (sa + sb - inter)).astype(np.float32)
    return np.abs(iou.view(np.int32).astype(np.int64) - int(np.float32(0.5).view(np.int32)))


def _cases():
    g = gold("g5_nms.npz")
    for n in (1, 2, 63, 64, 65, 128, 1000, 4097):
        yield "g5_%d" % n, g["dets_%d" % n], 0.5
    rs = np.random.RandomState(5)                                   # the tie sets of test_kernels_gpu.py
    for n in (7, 300, 2500, 9000):
        xy = rs.uniform(0, 500, (n, 2)); wh = rs.uniform(4, 200, (n, 2))
        sc = np.round(rs.uniform(0, 1, (n, 1)), 2)
        d = np.concatenate([xy, xy + wh, sc], 1).astype(np.float32)
        for thr in (0.3, 0.5, 0.7):
            yield "ties_%d_%s" % (n, thr), d, thr
    rs = np.random.RandomState(11)                                  # exact duplicates and near-duplicates
    base = rs.uniform(0, 300, (200, 2)); wh = rs.uniform(8, 90, (200, 2))
    d = np.concatenate([base, base + wh, rs.uniform(0, 1, (200, 1))], 1).astype(np.float32)
    dup = np.concatenate([d, d, d + np.float32(1e-4)], 0)
    yield "duplicates", dup, 0.5
    for s in range(50):                                              # 50 x 2 000 = 1e5 pairs at the threshold
        yield "near_%d" % s, _near_threshold_pairs(np.random.RandomState(100 + s), 2000, 4 if s < 40 else 10), 0.5


def test_mask_words_and_keep_lists_equal_the_reference_compiled_kernel():
    from oracle import nms_oracle, ref_nms
    if not ref_nms.available():
        pytest.skip("oracle/_ref/libref_nms_kernel.so missing: run `make -C oracle ref` where /root/reference exists")
    words = pairs_hi = pairs_lo = within4 = 0
    mode_diff = 0
    for name, d, thr in _cases():
        n = d.shape[0]
        keep, srt, order, mask = _product(d, thr)
        # the sort the wrapper does (pth_nms.py:33-36): a descending-score permutation of the input rows
        assert sorted(order.tolist()) == list(range(n)), name
        assert np.array_equal(srt, d[order]), name
        assert np.all(np.diff(srt[:, 4]) <= 0), name
        sd = torch.from_numpy(srt).cuda()
        ref = ref_nms.ref_mask(sd, thr, "default")
        refu = _upper(ref)
        bad = np.nonzero(refu != _upper(mask))
        assert bad[0].size == 0, "%s: %d mask words differ from the reference kernel, first at row %d word %d" % (
            name, bad[0].size, bad[0][0], bad[1][0])
        words += int((np.arange((n + 63) // 64)[None, :] >= (np.arange(n) // 64)[:, None]).sum())
        # keep list through the reference's scan (nms_cuda.c:47-58) and the wrapper's order[keep] (pth_nms.py:44)
        kref = order[ref_nms.host_scan(ref, n)]
        assert np.array_equal(keep, kref), "%s: keep list differs from reference kernel + scan" % name
        # the C restatement against the reference-compiled kernel: full mask incl. the lower triangle, and its keep list
        ko, mo = nms_oracle.nms(d, thr, "gpu", return_mask=True)
        assert np.array_equal(mo, ref), "%s: oracle/nms_oracle.c mask differs from the reference kernel" % name
        assert np.array_equal(ko, kref), name
        if ref_nms.available("nocontract"):
            mode_diff += int((ref_nms.ref_mask(sd, thr, "nocontract") != ref).sum())
        if name.startswith("near_"):
            h = n // 2                                              # pair p = rows p and p + h of d
            pos = np.empty(n, np.int64); pos[order] = np.arange(n)
            lo, hi = np.minimum(pos[:h], pos[h:]), np.maximum(pos[:h], pos[h:])
            bit = (ref[lo, hi // 64] >> (hi % 64).astype(np.uint64)) & np.uint64(1)
            pairs_hi += int(bit.sum()); pairs_lo += int(h - bit.sum())
            within4 += int((_ulps_from_half(d) <= 4).sum())
    assert pairs_hi > 1000 and pairs_lo > 1000, "the near-threshold set must straddle the threshold (%d / %d)" % (pairs_hi, pairs_lo)
    report("nms vs reference-compiled nms_kernel.cu: %d upper-triangle mask words equal, keep lists equal; %d pairs at IoU = 1/2 +- ulps "
           "(%d of them within 4 ulp; %d above / %d not above the threshold) decided identically; words differing between the default-contraction and "
           "-ffp-contract=off builds of the reference kernel: %d" % (words, pairs_hi + pairs_lo, within4, pairs_hi, pairs_lo, mode_diff))
    assert mode_diff == 0, "the two contraction modes of the reference kernel disagree: DESIGN §4 must say which one is followed"
