"""FusedAdam — torch.optim.Adam semantics (the reference's optimizer,
training/multipose_keypoint_train.py:106-110, stepped at training/trainer.py:259) as ONE HIP launch
per contiguous run of trainable parameters in the flat arena, instead of ~5 foreach kernels over
~400 tensors.  It is a ``torch.optim.Optimizer`` so the reference's Trainer type checks
(trainer.py:137-145) and ``ReduceLROnPlateau`` keep working (lr is read from param_groups[0]).
"""
import math

import torch

from . import ops
from ._lib import call


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = [p for p in model.parameters() if p.requires_grad]
        super(FusedAdam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self._arena = None
        self._runs = None
        self._m = None
        self._v = None
        self._t = 0
        self.grad_scale = 1.0

    def _bind(self):
        ar = self.model._arena
        if self._arena is not ar:
            self._arena = ar
            self._m = torch.zeros(ar.total, dtype=torch.float32, device=ar.device)
            self._v = torch.zeros(ar.total, dtype=torch.float32, device=ar.device)
        self._runs = ar.trainable_runs()
        return ar

    def zero_grad(self, set_to_none=False):
        ar = self.model._arena
        if ar is not None and ar.grad_flat is not None:
            ar.grad_flat.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        ar = self._bind()
        if ar.grad_flat is None:
            return loss
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        self._t += 1
        bc1 = 1.0 - b1 ** self._t
        bc2s = math.sqrt(1.0 - b2 ** self._t)
        for s, e in self._runs:
            call("mpn_adam_step", ops.ptr(ar.flat[s:e]), ops.ptr(ar.grad_flat[s:e]), ops.ptr(self._m[s:e]), ops.ptr(self._v[s:e]),
                 e - s, float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), bc1, bc2s,
                 float(self.grad_scale), ops.stream_ptr())
        return loss
