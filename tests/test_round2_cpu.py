"""Host-side logic added in round 2 (no GPU): FusedAdam checkpoint layout, lazy log proxies, reducer re-bucketing."""
import copy
import pickle

import numpy as np
import pytest
import torch


def _model():
    from multiposenet.pytorch_amd.network.posenet import poseNet
    m = poseNet(50)
    for p in m.prn.parameters():
        p.requires_grad = False
    return m


def test_fused_adam_state_dict_uses_torch_adam_layout_and_round_trips():
    """optimizer.state_dict() (pickled by the reference's save_net, network/net_utils.py:37-46, restored at
    training/trainer.py:228) must carry the Adam moments and the step count."""
    from multiposenet.pytorch_amd.optim import FusedAdam
    m = _model()
    opt = FusedAdam(m, lr=3e-4, betas=(0.8, 0.99), eps=1e-7, weight_decay=0.01)
    ar = opt._bind()
    g = torch.Generator().manual_seed(0)
    opt._m.copy_(torch.randn(ar.total, generator=g))
    opt._v.copy_(torch.rand(ar.total, generator=g))
    opt._set_step(7)
    sd = opt.state_dict()
    params = [p for p in m.parameters() if p.requires_grad]
    assert sd["param_groups"][0]["params"] == list(range(len(params)))
    assert sd["param_groups"][0]["lr"] == 3e-4 and sd["param_groups"][0]["betas"] == (0.8, 0.99)
    assert set(sd["state"].keys()) == set(range(len(params)))
    for i, p in enumerate(params):
        st = sd["state"][i]
        assert float(st["step"]) == 7.0
        assert tuple(st["exp_avg"].shape) == tuple(p.shape) and tuple(st["exp_avg_sq"].shape) == tuple(p.shape)
    # what save_net does: deepcopy, .cpu(), pickle
    blob = pickle.dumps([copy.deepcopy(sd)])
    loaded = pickle.loads(blob)[0]
    # torch.optim.Adam over the same parameters accepts it (same layout) ...
    ref = torch.optim.Adam(params, lr=1e-4)
    ref.load_state_dict(copy.deepcopy(loaded))
    assert torch.equal(ref.state[params[3]]["exp_avg"], sd["state"][3]["exp_avg"])
    # ... and a fresh FusedAdam restores moments, step and hyper-parameters; a torch.optim.Adam state dict loads too
    for source in (loaded, ref.state_dict()):
        m2 = _model()
        opt2 = FusedAdam(m2, lr=1e-4)
        opt2.load_state_dict(source)
        assert opt2.step_count() == 7
        assert opt2.param_groups[0]["lr"] == 3e-4 and tuple(opt2.param_groups[0]["betas"]) == (0.8, 0.99)
        sd2 = opt2.state_dict()
        for i in range(len(params)):
            assert torch.equal(sd2["state"][i]["exp_avg"], sd["state"][i]["exp_avg"])
            assert torch.equal(sd2["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
    # conv moments are stored [Cout][R][S][Cin] like the weights: the logical view must still be [Cout,Cin,R,S]
    w = m.fpn.layer1[0].conv2.weight
    i = [k for k, q in enumerate(params) if q is w][0]
    j = ar.index[id(w)]
    seg = opt._m[ar.offsets[j]: ar.offsets[j] + ar.sizes[j]].view(w.shape[0], w.shape[2], w.shape[3], w.shape[1])
    assert torch.equal(sd["state"][i]["exp_avg"], seg.permute(0, 3, 1, 2))


def test_fused_adam_keeps_moments_when_the_arena_is_rebuilt():
    from multiposenet.pytorch_amd.optim import FusedAdam
    m = _model()
    opt = FusedAdam(m)
    ar = opt._bind()
    opt._m.fill_(0.25)
    opt._v.fill_(0.5)
    opt._set_step(3)
    m._build_arena(torch.device("cpu"))          # what .cuda()/.to() triggers through _apply
    ar2 = opt._bind()
    assert ar2 is not ar
    assert opt.step_count() == 3 and float(opt._m.min()) == 0.25 and float(opt._v.max()) == 0.5


def test_lazy_float_compares_like_a_float():
    """`lazy > x`, max(lazy, x) and ReduceLROnPlateau(mode='max') must work."""
    from multiposenet.pytorch_amd.network.losses import LazyFloat

    class Src(object):
        def get(self):
            return [1.5, -2.0]
    a, b = LazyFloat(Src(), 0), LazyFloat(Src(), 1)
    assert a > 1.0 and a >= 1.5 and not (a > 1.5) and a != 2.0 and not (a != 1.5) and b < a and b <= -2.0
    assert max(a, 1.0) == 1.5 and min(b, 0.0) == -2.0 and sorted([a, b, 0.0]) == [-2.0, 0.0, 1.5]
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0), mode="max", patience=0)
    sched.step(a)
    sched.step(b)
    sched.step(b)
    assert sched.optimizer.param_groups[0]["lr"] < 1.0


def test_prn_rows_need_not_be_multiples_of_32():
    """prn_coeff 1 / 3 give 8568 / 77112 inputs (not multiples of 32); the constructor accepts them and the
    parameter shapes follow posenet.py:130-137."""
    from multiposenet.pytorch_amd.network.posenet import poseNet
    m = poseNet(50, prn_node_count=64, prn_coeff=1)
    assert m.prn.dens1.weight.shape == (64, 28 * 18 * 17) and m.prn.dens2.weight.shape == (28 * 18 * 17, 64)


def test_heatmap_loss_rejects_mismatched_targets_before_any_launch():
    """the loss kernels index the targets with the prediction's pixel grid — wrong shapes must raise, a [B,1,H,W]
    mask (which the reference's torch expression broadcasts, posenet.py:376-381) is expanded."""
    import pytest
    from multiposenet.pytorch_amd._lib import MpnError
    from multiposenet.pytorch_amd.network.losses import _check_heatmap_targets
    preds = [torch.zeros(2, 18, 8, 8) for _ in range(5)]
    heat, wgt = torch.zeros(2, 18, 8, 8), torch.ones(2, 1, 8, 8)
    h, w = _check_heatmap_targets(preds, heat, wgt)
    assert tuple(w.shape) == (2, 18, 8, 8) and h is heat
    with pytest.raises(MpnError):
        _check_heatmap_targets(preds, torch.zeros(2, 18, 4, 4), wgt)
    with pytest.raises(MpnError):
        _check_heatmap_targets(preds, heat, torch.ones(2, 3, 8, 8))
    with pytest.raises(MpnError):
        _check_heatmap_targets(preds[:4] + [torch.zeros(2, 17, 8, 8)], heat, wgt)


def test_opencv_resize_restatement_agrees_with_an_independent_implementation():
    """cv2 is absent from this image, so the oracle's restatement of cv2.resize (joint_utils.py:110-113 peak refinement,
    tester.py:264-331 scale pyramid) cannot be pinned to OpenCV itself.  Independent cross-check: torch's CPU interpolate uses the
    same published rules (cubic convolution with a = -0.75, half-pixel centres, replicated border; bilinear likewise), written by
    other people — the two agree to float32 rounding.  (Parity with cv2 proper stays UNPINNED; this rules out a wrong kernel, tap
    offset or border rule.)"""
    import torch.nn.functional as F
    from oracle import joint_oracle as jo
    rs = np.random.RandomState(0)
    for f in (2.0, 4.0, 8.0):
        for _ in range(4):
            p = rs.rand(5, 5).astype(np.float32)
            t = F.interpolate(torch.from_numpy(p)[None, None], scale_factor=f, mode='bicubic', align_corners=False)[0, 0].numpy()
            assert np.abs(jo.cv_resize_cubic(p, f) - t).max() <= 1e-6
    img = rs.rand(23, 31, 3).astype(np.float32)
    chw = torch.from_numpy(img).permute(2, 0, 1)[None]
    for hw in ((46, 62), (92, 124), (37, 55)):
        for cubic, mode in ((True, 'bicubic'), (False, 'bilinear')):
            t = F.interpolate(chw, size=hw, mode=mode, align_corners=False)[0].permute(1, 2, 0).numpy()
            assert np.abs(jo.cv_resize(img, hw, cubic) - t).max() <= 1e-5
