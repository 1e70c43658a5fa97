"""GPU parity for the device ground-truth heat-map generator through the C ABI
(mpn_gt_heatmaps via multiposenet.pytorch_amd.datasets.heatmap.put_gaussian_maps).

Reference arithmetic is float64 rounded to float32 at the end; the only operation that may differ from numpy is the
double-precision exp() (device libm vs glibc, both < 1 ulp), so results must match the goldens — rendered by the
reference's own putGaussianMaps — to 1 float32 ulp at most (tolerance 1.2e-7 absolute on values in [0, 1]) and the set
of non-zero cells (the 4.6052 cut-off) and of saturated cells (the 1.0 clamp) must be identical."""
import numpy as np
import pytest
import torch

from helpers import gold
from oracle import heatmap_oracle as ho

pytestmark = pytest.mark.gpu
TOL = 1.2e-7


def _run(joints, num, crop, stride, sigma):
    from multiposenet.pytorch_amd.datasets.heatmap import put_gaussian_maps
    out = put_gaussian_maps(torch.from_numpy(np.ascontiguousarray(joints)).cuda(), torch.from_numpy(np.ascontiguousarray(num)).cuda(),
                            crop, crop, stride, sigma)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _check(out, ref):
    assert out.shape == ref.shape and out.dtype == np.float32
    assert np.array_equal(out != 0, ref != 0), "cut-off mask differs"
    assert np.array_equal(out >= 1.0, ref >= 1.0), "clamp set differs"
    assert float(np.abs(out - ref).max()) <= TOL


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_gt_heatmaps_vs_reference_goldens(case):
    g = gold("g9_gt_heatmaps.npz")
    crop, stride, sigma = (float(v) for v in g["cfg_" + case])
    _check(_run(g["joints_" + case], g["num_" + case], crop, stride, sigma), g["out_" + case])


def test_gt_heatmaps_edge_cases_vs_oracle():
    rng = np.random.RandomState(5)
    # no people at all; everything unannotated; one person far outside; maxP larger than any count
    joints = np.zeros((4, 5, 18, 3))
    joints[..., :2] = rng.uniform(0, 128, size=(4, 5, 18, 2))
    joints[1, ..., 2] = 2.0
    joints[2, 0, :, :2] = -500.0
    num = np.array([0, 5, 1, 3], dtype=np.int32)
    out = _run(joints, num, 128, 4, 7.0)
    _check(out, ho.gt_heatmaps(joints, num, 128, 128, 4, 7.0))
    assert not out[0].any() and not out[1].any() and not out[2].any() and out[3].any()


def test_gt_heatmaps_full_size_properties():
    """Bench-size batch (32 x 480 x 480, up to 12 people): run-to-run determinism, batch independence (bit-exact) and
    the [0, 1] range; a sample of images against the oracle."""
    rng = np.random.RandomState(6)
    B, maxP = 32, 12
    joints = np.zeros((B, maxP, 18, 3))
    joints[..., :2] = rng.uniform(-20, 500, size=(B, maxP, 18, 2))
    joints[..., 2] = rng.choice([0.0, 1.0, 2.0], size=(B, maxP, 18))
    num = rng.randint(0, maxP + 1, size=(B,)).astype(np.int32)
    a = _run(joints, num, 480, 4, 7.0)
    b = _run(joints, num, 480, 4, 7.0)
    assert np.array_equal(a, b)
    assert a.min() >= 0.0 and a.max() <= 1.0
    sub = [3, 17, 31]
    c = _run(joints[sub], num[sub], 480, 4, 7.0)
    assert np.array_equal(c, a[sub])
    _check(c, ho.gt_heatmaps(joints[sub], num[sub], 480, 480, 4, 7.0))


def test_gt_heatmaps_rejects_cpu_and_bad_shapes():
    from multiposenet.pytorch_amd._lib import MpnError
    from multiposenet.pytorch_amd.datasets.heatmap import put_gaussian_maps
    with pytest.raises(MpnError):
        put_gaussian_maps(torch.zeros((1, 1, 18, 3), dtype=torch.float64), torch.ones(1, dtype=torch.int32), 64, 64)
    with pytest.raises(MpnError):
        put_gaussian_maps(torch.zeros((1, 1, 17, 3), dtype=torch.float64, device="cuda"), torch.ones(1, dtype=torch.int32), 64, 64)
