"""GPU parity for heat-map peak extraction through the C ABI (mpn_heatmap_peaks via the
network/joint_utils.py mirror).

Pinned branches — find_peaks and NMS(bool_refine_center=False) — must equal the goldens produced by the REAL reference
functions exactly (integer coordinates, float32 scores widened to double, ids).  The refinement branch (cv2 bicubic) has
no reference golden in this image (cv2 absent; see oracle/joint_oracle.py): it is compared with the oracle's restatement
of OpenCV's algorithm — coordinates and ids exactly, scores bit-exactly (same float32 operation order), and against the
reference golden too whenever tests/golden/g10_peaks.npz was regenerated on a box with cv2 (`refined == 1`)."""
import numpy as np
import pytest
import torch

from helpers import gold
from oracle import joint_oracle as jo

pytestmark = pytest.mark.gpu
PARAM = {"thre1": 0.1, "thre2": 0.05, "thre3": 0.5}


def _ju():
    from multiposenet.pytorch_amd.network import joint_utils
    return joint_utils


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_find_peaks_and_plain_nms_vs_reference(case):
    g = gold("g10_peaks.npz")
    heat = torch.from_numpy(g["heat_" + case]).cuda()               # [H, W, J] like the reference's numpy input
    up = float(g["up_" + case])
    ju = _ju()
    counts, xy = [], []
    for j in range(18):
        p = ju.find_peaks(PARAM, heat[:, :, j])
        counts.append(len(p)); xy.append(p.reshape(-1, 2))
    assert np.array_equal(np.array(counts), g["fp_counts_" + case])
    assert np.array_equal(np.concatenate(xy), g["fp_xy_" + case])
    plain = np.concatenate([p.reshape(-1, 4) for p in ju.NMS(PARAM, heat, up, bool_refine_center=False)])
    assert np.array_equal(plain, g["nms_plain_" + case])


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_refined_nms_vs_oracle_restatement(case):
    g = gold("g10_peaks.npz")
    heat_np = g["heat_" + case]
    up = float(g["up_" + case])
    got = _ju().NMS(PARAM, torch.from_numpy(heat_np).cuda(), up)
    ref = jo.nms_peaks(PARAM["thre1"], heat_np, up, refine=True)
    assert [len(p) for p in got] == [len(p) for p in ref]
    got, ref = np.concatenate(got), np.concatenate(ref)
    assert np.array_equal(got[:, [0, 1, 3]], ref[:, [0, 1, 3]])
    assert np.array_equal(got[:, 2], ref[:, 2]), float(np.abs(got[:, 2] - ref[:, 2]).max())
    if int(g["refined"]) == 1:
        assert np.array_equal(got, g["nms_refined_" + case])


def test_layouts_batch_and_joint_list():
    """The model's own output layout ([B, J, H, W] channels-last strides), a batch in one launch, get_joint_list rows."""
    g = gold("g10_peaks.npz")
    ju = _ju()
    a = torch.from_numpy(g["heat_a"]).cuda()                       # [H, W, J]
    pred = torch.stack([a.permute(2, 0, 1), a.permute(2, 0, 1).flip(2)]).contiguous(memory_format=torch.channels_last)
    per_img = ju.NMS_batch(PARAM, pred, 4.0)
    one = ju.NMS(PARAM, a, 4.0)
    assert all(np.array_equal(x, y) for x, y in zip(per_img[0], one))
    flipped = jo.nms_peaks(PARAM["thre1"], np.ascontiguousarray(g["heat_a"][:, ::-1, :]), 4.0, refine=True)
    assert all(np.array_equal(x, y) for x, y in zip(per_img[1], flipped))
    img = np.zeros((120, 120, 3), dtype=np.float32)
    jl = ju.get_joint_list(img, PARAM, a, 1.5)
    ref = jo.nms_peaks(PARAM["thre1"], g["heat_a"], 120 / 30.0, refine=True)
    rows = np.array([tuple(p * np.array([1.5, 1.5, 1, 1])) + (t,) for t, ps in enumerate(ref) for p in ps])
    assert np.array_equal(jl, rows)


def test_edge_cases():
    ju = _ju()
    from multiposenet.pytorch_amd._lib import MpnError
    flat = torch.zeros((9, 7, 18), device="cuda")
    assert all(len(p) == 0 for p in ju.NMS(PARAM, flat, 4.0))                     # nothing above the threshold
    flat[:] = 0.3                                                                   # one big plateau: every cell is a peak
    with pytest.raises(MpnError):
        ju.NMS(PARAM, flat, 4.0, cap=32)
    full = ju.NMS(PARAM, flat, 4.0, bool_refine_center=False, cap=64)
    assert [len(p) for p in full] == [63] * 18 and full[17][-1][3] == 63 * 18 - 1
    with pytest.raises(MpnError):
        ju.NMS(PARAM, flat.cpu(), 4.0)


@pytest.mark.parametrize("shape", [(2, 5, 33, 70), (1, 18, 160, 160), (3, 2, 7, 129)])
def test_layouts_and_dense_noise_vs_oracle(shape):
    """Round 6 kernels (flags -> per-plane compaction -> refinement over a device-wide list): the channels-last fast path, the
    planar path and a strided view (generic path) give the same peaks as the oracle's restatement of joint_utils.py:61-138 on
    NOISE maps (~15 % of the cells are peaks: hundreds to thousands per plane, several 256-word chunks per plane, widths that are not
    multiples of 64), with and without refinement — coordinates, ids and float32 scores exactly."""
    ju = _ju()
    B, J, H, W = shape
    rng = np.random.RandomState(B * 1000 + W)
    heat = rng.rand(B, J, H, W).astype(np.float32)
    heat[0, 0, :3, :] = 0.99                                            # a plateau: every cell of it is a peak
    thre = 0.97 if H * W > 10000 else 0.3                               # (the oracle refines peak by peak in Python)
    t = torch.from_numpy(heat).cuda()
    big = torch.zeros((B, J + 1, H + 2, W + 3), device="cuda")
    big[:, 1:, 1:-1, 2:-1] = t
    views = {"planar": t, "channels_last": t.contiguous(memory_format=torch.channels_last), "strided": big[:, 1:, 1:-1, 2:-1]}
    if J > 1:
        assert views["channels_last"].stride(1) == 1 and views["channels_last"].stride(3) == J
    for refine in (False, True):
        ref = [jo.nms_peaks(thre, np.ascontiguousarray(heat[b].transpose(1, 2, 0)), 4.0, refine=refine) for b in range(B)]
        for name, v in views.items():
            got = ju.NMS_batch({"thre1": thre}, v, 4.0, bool_refine_center=refine)
            for b in range(B):
                assert [len(p) for p in got[b]] == [len(p) for p in ref[b]], (name, refine, b)
                g, r = np.concatenate(got[b]), np.concatenate(ref[b])
                assert np.array_equal(g, r), (name, refine, b, float(np.abs(g - r).max()))
    assert sum(len(p) for p in ref[0]) > (0.02 if thre > 0.9 else 0.1) * J * H * W
