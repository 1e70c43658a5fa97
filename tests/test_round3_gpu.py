"""Round-3 GPU tests: BASELINE config 2 at its exact size, the operand-resident 1x1 kernels, the virtual 512-channel
concatenation, the in-kernel split-K reduction, a bf16 drift gate against an oracle that rounds where the HIP path rounds,
the Trainer / Tester.val drivers against fixtures recorded from the real reference, config 5 end to end."""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT, gold, report
from test_model_gpu import close, get_model, load_he, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests selected but no GPU is visible"
    from multiposenet.pytorch_amd import _lib
    _lib.lib()


# ------------------------------------------------------------------------------------------------ BASELINE config 2
def test_cfg2_r50_keypoint_480_batch16_fp32_full_size():
    """BASELINE config 2 at its exact workload (multipose_keypoint_train.py:49-64 shapes): ResNet-50 `keypoint_subnet`,
    480x480, batch 16, fp32 arithmetic, batch-statistics BatchNorm, every non-PRN parameter trainable.  At full size: loss and
    the whole gradient arena finite, bit-reproducible, detection-pyramid parameters untouched (the keypoint tape skips them);
    on a batch-2 slice of the same inputs the fp32 loss and every logged per-level loss equal the CPU oracle's within 2e-4
    relative, the heat-maps within the north-star 1e-3 abs."""
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from oracle import posenet_oracle as po, weightgen
    B, S = 16, 480
    m = get_model(50, torch.float32)
    for p in m.prn.parameters():
        p.requires_grad = False
    m.train()
    img = t(weightgen.gen_images(310, B, S, S)).cuda()
    heat, wgt = (t(a).cuda() for a in weightgen.gen_keypoint_gt(311, B, S // 4, S // 4))
    bn_state = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}

    def run(sl):
        m.load_state_dict(bn_state, strict=False)
        m._arena.ensure_grads()
        m._arena.grad_flat.zero_()
        pred, saved = m([img[sl].contiguous(), "keypoint_subnet"])
        loss, log = poseNet.build_loss(saved, "keypoint_subnet", heat[sl].contiguous(), wgt[sl].contiguous())
        loss.backward()
        torch.cuda.synchronize()
        return pred.detach().clone(), loss.detach().clone(), m._arena.grad_flat.clone(), {k: float(v) for k, v in log.items()}
    full = slice(0, B)
    p0, l0, g0, log0 = run(full)
    p1, l1, g1, _ = run(full)
    assert p0.shape == (B, 18, S // 4, S // 4) and p0.dtype == torch.float32
    assert torch.isfinite(l0) and torch.isfinite(g0).all() and float(g0.abs().max()) > 0
    assert torch.equal(p0, p1) and torch.equal(l0, l1) and torch.equal(g0, g1), "cfg2 step is not bit-reproducible"
    ar = m._arena
    for name, prm in m.named_parameters():
        i = ar.index.get(id(prm))
        if i is None:
            continue
        gn = float(g0[ar.offsets[i]: ar.offsets[i] + ar.sizes[i]].abs().max())
        det_only = name.startswith(("regressionModel.", "classificationModel.", "fpn.conv6", "fpn.conv7", "fpn.latlayer", "fpn.toplayer0",
                                    "fpn.toplayer1", "fpn.toplayer2", "prn."))
        if det_only:
            assert gn == 0.0, "%s received a gradient in the keypoint-only step" % name
        else:
            assert gn > 0.0, "%s has no gradient in the keypoint step" % name
    # batch-2 slice against the oracle (BatchNorm uses the slice's own statistics on both sides)
    sl = slice(0, 2)
    ps, ls, gs, logs = run(sl)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if v.dtype != torch.int64 and not k.startswith("prn.")}
    sd.update({k: v.cpu().clone() for k, v in bn_state.items() if v.dtype != torch.int64})
    with torch.no_grad():
        opred, oks = po.posenet_forward(sd, img[sl].cpu(), "keypoint_subnet", 50, True)
        ol, olog = po.keypoint_loss(oks, heat[sl].cpu(), wgt[sl].cpu())
    rel = abs(float(ls) - float(ol)) / abs(float(ol))
    close("cfg2 slice (R50 kp 480x480 B=2 fp32) heat-map vs oracle", ps, opred, 1e-3, 1e-4)
    for k, v in olog.items():
        if k in logs and isinstance(v, float) and abs(v) > 1e-12:
            assert abs(logs[k] - v) <= 2e-4 * abs(v) + 1e-7, "log value %s: %.7g vs oracle %.7g" % (k, logs[k], v)
    report("cfg2 full size (R50 keypoint_subnet 480x480 B=16 fp32): finite, bit-reproducible, loss %.6f; B=2 slice loss %.7f vs oracle %.7f (rel %.2e)"
           % (float(l0), float(ls), float(ol), rel))
    assert rel <= 2e-4
