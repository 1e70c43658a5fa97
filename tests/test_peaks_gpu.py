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
