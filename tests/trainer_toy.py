"""Stand-ins shared by tests/golden/make_golden_trainer.py (which drives the REAL reference Trainer with them) and the tests
that drive this repository's Trainer with the same script: a two-parameter network whose loss is dominated by a scripted target
value, loaders that yield those values, and the reference's batch_processor contract for them.  Test infrastructure only."""
from collections import OrderedDict

import torch
import torch.nn as nn


class ToyNet(nn.Module):
    """forward([inp, subnet]) -> (y, [y, inp]); build_loss(saved, subnet, target) -> (loss, log dict).  The loss is
    target.mean() (scripted by the loaders) plus 1e-3 * mean(y^2): gradients flow, the control flow does not depend on them."""

    def __init__(self):
        super(ToyNet, self).__init__()
        self.conv = nn.Conv2d(1, 2, 1, bias=False)
        self.bn = nn.BatchNorm2d(2)
        with torch.no_grad():
            self.conv.weight.copy_(torch.tensor([0.5, -0.25]).view(2, 1, 1, 1))

    def forward(self, x):
        inp, subnet = x
        y = self.bn(self.conv(inp))
        return y, [y, inp]

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    @staticmethod
    def build_loss(saved_for_loss, *gts):
        subnet, target = gts
        y = saved_for_loss[0]
        main = target.mean()
        loss = main + 1e-3 * (y * y).mean()
        log = OrderedDict()
        log['main'] = float(main)
        log['steps_seen'] = int(target.numel())          # an int: metered like a float (trainer.py:324)
        log['note'] = 'subnet=%s' % subnet               # not a number: passed through (trainer.py:328)
        return loss, log


def toy_batch_processor(state, batch):
    """training/batch_processor.py:10-59 shape: (inputs, gts, saved_for_eval)."""
    inp, target = batch
    dev = next(state.model.parameters()).device
    rec = getattr(state, '_toy_record', None)
    if rec is not None:
        rec.append(dict(epoch=state.last_epoch, training=bool(state.model.training),
                        lr=[float(g['lr']) for g in state.optimizer.param_groups]))
    subnet = state.params.subnet_name
    return [[inp.to(dev), subnet]], [subnet, target.to(dev)], []


class ScriptedLoader(object):
    """A loader of `n` batches; pass `p` (0-based count of completed iterations) yields targets filled with values[p % len]."""

    def __init__(self, n, values, seed=0):
        self.n, self.values = n, list(values)
        self.passes = 0
        self.served = []                 # batches handed out per pass
        g = torch.Generator().manual_seed(seed)
        self.inputs = [torch.randn(2, 1, 4, 4, generator=g) for _ in range(n)]

    def __len__(self):
        return self.n

    def __iter__(self):
        v = self.values[self.passes % len(self.values)]
        self.passes += 1
        self.served.append(0)
        for i in range(self.n):
            self.served[-1] += 1
            yield self.inputs[i], torch.full((2, 3), float(v))
