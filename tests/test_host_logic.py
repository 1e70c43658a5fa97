"""Host-side logic of the boundary module (CPU only): state_dict contract vs the reference, arena
layout, anchors, freezing semantics, run/bucket construction."""
import hashlib

import numpy as np
import pytest
import torch

from helpers import gold


@pytest.fixture(scope="module")
def model50():
    from multiposenet.pytorch_amd.network.posenet import poseNet
    return poseNet(50)


def test_state_dict_keys_shapes_children_equal_reference(model50):
    g = gold("g0_keys.npz")
    sd = model50.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys_50"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g["shapes_50"]]
    assert [n for n, _ in model50.named_children()] == [str(n) for n in g["children_50"]]
    assert [n for n, _ in model50.fpn.named_children()] == [str(n) for n in g["fpn_children_50"]]
    assert len(sd) == 402


def test_r101_key_count():
    g = gold("g0_keys.npz")
    assert len(g["keys_101"]) == 708


def test_reference_init_rules(model50):
    m = model50
    assert float(m.classificationModel.output.weight.abs().sum()) == 0.0
    assert abs(float(m.classificationModel.output.bias[0]) + np.log(99.0)) < 1e-6
    assert float(m.regressionModel.output.weight.abs().sum()) == 0.0 and float(m.regressionModel.output.bias.abs().sum()) == 0.0
    w = m.fpn.layer2[0].conv2.weight
    assert abs(float(w.std()) - 0.01) < 5e-4 and m.fpn.layer2[0].conv2.bias is None
    assert float(m.convt1.bias.abs().sum()) == 0.0
    assert not m.fpn.bn1.training            # ctor ends with freeze_bn() (posenet.py:211)
    m.train()
    assert m.fpn.bn1.training                # ...which any later .train() undoes
    m.freeze_bn()
    assert not m.fpn.layer3[2].bn2.training and m.training


def test_arena_layout_and_views(model50):
    ar = model50._arena
    assert ar.consistent()
    w = model50.fpn.layer1[0].conv2.weight
    assert tuple(w.shape) == (64, 64, 3, 3) and w.stride() == (576, 1, 192, 64)      # [Cout][R][S][Cin] storage
    seg = ar.data_seg(w)
    assert seg.data_ptr() == w.data_ptr() and seg.numel() == w.numel()
    assert torch.equal(seg.view(64, 3, 3, 64), w.permute(0, 2, 3, 1))
    for off in ar.offsets:
        assert off % 64 == 0
    # load_state_dict goes through the views and keeps the arena intact
    sd = {k: v.clone() for k, v in model50.state_dict().items()}
    sd["fpn.conv1.weight"] = torch.arange(64 * 3 * 7 * 7, dtype=torch.float32).view(64, 3, 7, 7)
    model50.load_state_dict(sd)
    assert ar.consistent()
    assert torch.equal(model50.fpn.conv1.weight, sd["fpn.conv1.weight"])
    assert float(ar.data_seg(model50.fpn.conv1.weight)[3]) == float(sd["fpn.conv1.weight"][0, 0, 0, 1])   # (r0,s1,c0)


def test_grad_arena_and_trainable_runs(model50):
    m = model50
    for p in m.parameters():
        p.requires_grad = True
    for name, module in m.fpn.named_children():
        if name in ("conv6", "conv7", "latlayer1", "latlayer2", "latlayer3", "toplayer0", "toplayer1", "toplayer2"):
            for p in module.parameters():
                p.requires_grad = False
    for p in m.prn.parameters():
        p.requires_grad = False
    ar = m._arena
    ar.ensure_grads()
    assert m.fpn.conv1.weight.grad is not None and m.fpn.conv6.weight.grad is None
    assert m.fpn.conv1.weight.grad.stride() == m.fpn.conv1.weight.stride()
    m.fpn.conv1.weight.grad.fill_(2.0)
    assert float(ar.grad_seg(m.fpn.conv1.weight).sum()) == 2.0 * 64 * 147
    for p in m.parameters():
        p.grad = None                        # optimizer.zero_grad(set_to_none=True)
    ar.ensure_grads()
    assert float(ar.grad_flat.abs().sum()) == 0.0 and m.fpn.conv1.weight.grad is not None
    runs = ar.trainable_runs()
    covered = sum(e - s for s, e in runs)
    want = sum((p.numel() + 63) // 64 * 64 for p in m.parameters() if p.requires_grad)
    assert covered == want and len(runs) >= 2
    for s, e in runs:                        # no frozen parameter inside a run
        for i, p in enumerate(ar.params):
            if not p.requires_grad:
                assert not (s <= ar.offsets[i] < e)
    for p in m.parameters():
        p.requires_grad = True


def test_product_anchors_bit_exact_vs_reference():
    from multiposenet.pytorch_amd.network.anchors import Anchors
    g = gold("g1_anchors.npz")
    an = Anchors()
    for (h, w) in ((256, 256), (480, 480), (608, 608), (640, 640), (800, 800), (128, 96), (100, 70)):
        tag = "%dx%d" % (h, w)
        a = an(torch.zeros(1, 3, h, w)).numpy()
        assert tuple(a.shape) == tuple(g["shape_" + tag])
        assert hashlib.sha256(a.tobytes()).hexdigest() == str(g["sha_" + tag]), tag
    assert an(torch.zeros(2, 3, 480, 480)) is an(torch.zeros(1, 3, 480, 480))      # cached per (H, W, device)


def test_forward_on_cpu_raises(model50):
    from multiposenet.pytorch_amd._lib import MpnError
    with pytest.raises(MpnError):
        model50([torch.zeros(1, 3, 64, 64), "keypoint_subnet"])
    assert model50.build_loss([], "something_else") == 0         # posenet.py:363-364
