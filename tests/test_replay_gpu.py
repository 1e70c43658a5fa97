"""The recorded training step (replay.py) against the eager autograd step: bit-identical parameters, Adam moments and BN
statistics over several steps with fresh inputs, log values equal up to the f32 summation order of the heat-map loss (the
recorded step sums it inside its one-pass loss kernel; MPN_FUSED_MSE=0 makes the logs bit-identical too); learning-rate
changes; re-recording after a change of the trainable set; the single-subnet bodies."""
import numpy as np
import pytest
import torch

from helpers import report
from test_model_gpu import get_model, t
from test_round2_gpu import _train_setup

pytestmark = pytest.mark.gpu


def _same_logs(a, b, rel=2e-6):
    """[(loss, values, names)] per step: same names, values within the summation-order tolerance of the heat-map loss."""
    assert len(a) == len(b)
    for (la, va, na), (lb, vb, nb) in zip(a, b):
        assert na == nb
        for x, y in zip([la] + va, [lb] + vb):
            assert abs(x - y) <= rel * max(abs(x), abs(y), 1e-30), "log values differ:\n%s\n%s" % (a, b)
    return True


def _run(m, state0, make_step, batches, lr_change_at=None, freeze_at=None):
    from multiposenet.pytorch_amd.optim import FusedAdam
    m.load_state_dict(state0)
    m.train()
    for p in m.parameters():
        p.requires_grad = True
    for p in m.prn.parameters():
        p.requires_grad = False
    opt = FusedAdam(m, lr=1e-3)
    step = make_step(m, opt)
    logs = []
    for i, (inputs, gts) in enumerate(batches):
        if lr_change_at is not None and i == lr_change_at:
            opt.param_groups[0]["lr"] = 2e-5
        if freeze_at is not None and i == freeze_at:
            for p in m.fpn.layer1.parameters():
                p.requires_grad = False
            opt = FusedAdam(m, lr=1e-3)
            step = make_step(m, opt)
        a = [[inputs[0][0].clone(), inputs[0][1]]]
        b = [gts[0]] + [x.clone() for x in gts[1:]]
        loss, log = step(a, b)
        logs.append((float(loss), [float(v) for v in log.values()], list(log.keys())))
    torch.cuda.synchronize()
    bn = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}
    return m._arena.flat.clone(), opt._m.clone(), opt._v.clone(), bn, logs, step


@pytest.mark.parametrize("subnet", ["train_both", "keypoint_subnet", "detection_subnet"])
def test_replayed_step_is_bit_identical_to_the_eager_autograd_step(subnet):
    from multiposenet.pytorch_amd.replay import ReplayedTrainStep
    from multiposenet.pytorch_amd.training.batch_processor import train_step
    m, inputs, gts = _train_setup(50, torch.bfloat16, 4, 128, seed=110)
    _, inputs_b, gts_b = _train_setup(50, torch.bfloat16, 4, 128, seed=120)

    def pick(inp, g):
        if subnet == "train_both":
            return inp, g
        if subnet == "keypoint_subnet":
            return [[inp[0][0], subnet]], [subnet, g[1], g[2]]
        return [[inp[0][0], subnet]], [subnet, g[3]]
    batches = [pick(inputs, gts) if i % 2 == 0 else pick(inputs_b, gts_b) for i in range(6)]
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    eager = _run(m, state0, lambda mm, oo: (lambda a, b: train_step(mm, oo, a, b)), batches, lr_change_at=4)
    rep = _run(m, state0, lambda mm, oo: ReplayedTrainStep(mm, oo), batches, lr_change_at=4)
    assert rep[5].replays == 4                      # one eager pass, one recording, four replays
    assert _same_logs(eager[4], rep[4])
    assert torch.equal(eager[0], rep[0]) and torch.equal(eager[1], rep[1]) and torch.equal(eager[2], rep[2])
    assert all(torch.equal(eager[3][k], rep[3][k]) for k in eager[3])
    assert eager[4][0][0] != eager[4][5][0]
    report("recorded step (%s, R50 128x128 B=4 bf16): 6 steps bit-identical to the eager autograd step; losses %s"
           % (subnet, [round(x[0], 5) for x in rep[4]]))


def test_replayed_step_rerecords_when_the_trainable_set_changes():
    from multiposenet.pytorch_amd.replay import ReplayedTrainStep
    from multiposenet.pytorch_amd.training.batch_processor import train_step
    m, inputs, gts = _train_setup(50, torch.bfloat16, 2, 64, seed=130)
    batches = [(inputs, gts)] * 6
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    eager = _run(m, state0, lambda mm, oo: (lambda a, b: train_step(mm, oo, a, b)), batches, freeze_at=3)
    rep = _run(m, state0, lambda mm, oo: ReplayedTrainStep(mm, oo), batches, freeze_at=3)
    assert torch.equal(eager[0], rep[0]) and _same_logs(eager[4], rep[4])
    for p in m.parameters():
        p.requires_grad = True
