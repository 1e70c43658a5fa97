"""FusedAdam — torch.optim.Adam semantics (the reference's optimizer,
training/multipose_keypoint_train.py:106-110, stepped at training/trainer.py:259) as ONE HIP launch
per contiguous run of trainable parameters in the flat arena, instead of ~5 foreach kernels over
~400 tensors.  It is a ``torch.optim.Optimizer`` so the reference's Trainer type checks
(trainer.py:137-145) and ``ReduceLROnPlateau`` keep working (lr is read from param_groups[0]).

Every scalar of the update (lr, betas, eps, weight decay, bias corrections, step count) lives in a
small DEVICE vector (include/mpn.h: mpn_adam_advance / mpn_adam_step_dev), so a training step captured
in a hipGraph (tools/archive/r2/hipgraph_step.py) replays correct Adam steps; the host only rewrites the vector when a
scheduler changes a hyper-parameter.

``state_dict()`` / ``load_state_dict()`` use torch.optim.Adam's own layout (per-parameter ``step``,
``exp_avg``, ``exp_avg_sq``), which is what the reference pickles next to a checkpoint
(network/net_utils.py:37-46) and restores at trainer.py:228 — a checkpoint written by
``torch.optim.Adam`` over the same parameters loads here and vice versa.
"""
import math
import warnings

import torch

from . import ops
import ctypes

from ._lib import call, call_raw

_NH = 16       # floats in the device hyper vector (9 used)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = [p for p in model.parameters() if p.requires_grad]
        super(FusedAdam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self._arena = None
        self._runs = None
        self._m = None
        self._v = None
        self._hyper = None           # device float[_NH]
        self._hyper_sent = None      # the host values last uploaded
        self._pending_step = 0       # step count to install when the device vector is (re)created
        self.grad_scale = 1.0

    # ------------------------------------------------------------------ binding to the arena
    def _bind(self):
        ar = self.model._arena
        if self._arena is not ar:
            m = torch.zeros(ar.total, dtype=torch.float32, device=ar.device)
            v = torch.zeros(ar.total, dtype=torch.float32, device=ar.device)
            if self._m is not None:
                # the arena was rebuilt (e.g. model.cuda() after the optimizer was made): carry the moments over
                if self._m.numel() == ar.total:
                    m.copy_(self._m)
                    v.copy_(self._v)
                    self._pending_step = self.step_count()
                else:
                    warnings.warn("FusedAdam: the parameter arena changed size; Adam moments restart from zero")
                    self._pending_step = 0
            self._arena, self._m, self._v = ar, m, v
            self._hyper = None
        if self._hyper is None:
            self._hyper = torch.zeros(_NH, dtype=torch.float32, device=ar.device)
            self._hyper_sent = None
            self._set_step(self._pending_step)
        self._runs = ar.trainable_runs()
        return ar

    def _host_hyper(self):
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        return (float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), float(self.grad_scale))

    def sync_hyper(self):
        """Upload lr / betas / eps / weight decay / grad scale when they changed on the host (a scheduler step).  Outside
        graph capture only; a captured step reads whatever the vector holds at replay time."""
        self._bind()
        h = self._host_hyper()
        if h != self._hyper_sent:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdam: hyper-parameters changed during graph capture; call sync_hyper() before capturing")
            self._hyper[:6].copy_(torch.tensor(h, dtype=torch.float32))      # rare (scheduler step): a plain staged H2D copy
            self._hyper_sent = h

    def _set_step(self, t):
        t = int(t)
        b1, b2 = self.param_groups[0]["betas"]
        vals = torch.zeros(3, dtype=torch.float32)
        vals[0] = 1.0 - b1 ** t if t > 0 else 1.0
        vals[1] = math.sqrt(1.0 - b2 ** t) if t > 0 else 1.0
        vals[2:3].view(torch.int32)[0] = t
        self._hyper[6:9].copy_(vals)
        self._pending_step = t

    def step_count(self):
        """Number of steps taken (read back from the device: graph replays advance it without the host)."""
        if self._hyper is None:
            return int(self._pending_step)
        return int(self._hyper[8:9].view(torch.int32).item())

    # ------------------------------------------------------------------ Optimizer interface
    def zero_grad(self, set_to_none=False):
        ar = self.model._arena
        if ar is not None and ar.grad_flat is not None:
            if ar.grad_flat.is_cuda:
                # only trainable parameters ever receive gradients (frozen ones stay at the zeros they were allocated with)
                runs = ar.trainable_runs()          # current requires_grad flags: a parameter unfrozen since the last step is included
                for s, e in runs:
                    call("mpn_fill_f32", ops.ptr(ar.grad_flat[s:e]), 0.0, e - s, ops.stream_ptr())
            else:
                ar.grad_flat.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        ar = self._bind()
        if ar.grad_flat is None:
            return loss
        ops.check_device(ar.flat)
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        elif self._hyper_sent is None:
            raise RuntimeError("FusedAdam: call sync_hyper() (or take one eager step) before capturing a graph")
        call("mpn_adam_advance", ops.ptr(self._hyper), ops.stream_ptr())
        for s, e in self._runs:
            call("mpn_adam_step_dev", ops.ptr(ar.flat[s:e]), ops.ptr(ar.grad_flat[s:e]), ops.ptr(self._m[s:e]), ops.ptr(self._v[s:e]),
                 e - s, ops.ptr(self._hyper), ops.stream_ptr())
        return loss

    # ------------------------------------------------------------------ the update, bucket by bucket (ddp.GradReducer.on_bucket)
    @torch.no_grad()
    def begin_bucketed(self):
        """First half of ``step()`` for a step that updates bucket by bucket while backward still runs: bind, upload changed
        hyper-parameters, advance the device step count / bias corrections on the CURRENT stream (every bucket's update is ordered
        after it through the bucket's readiness event).  The second half is ``update_slice`` per bucket."""
        ar = self._bind()
        ops.check_device(ar.flat)
        ar.ensure_grads()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        call("mpn_adam_advance", ops.ptr(self._hyper), ops.stream_ptr())
        base = (ar.flat.data_ptr(), ar.grad_flat.data_ptr(), self._m.data_ptr(), self._v.data_ptr())
        hyper = self._hyper.data_ptr()

        def update_slice(start, end, raw_stream):
            # same kernel, same hyper vector, element-wise: the parameters equal the single-launch step's bit for bit
            o = 4 * start
            call_raw("mpn_adam_step_dev", ctypes.c_void_p(base[0] + o), ctypes.c_void_p(base[1] + o), ctypes.c_void_p(base[2] + o),
                     ctypes.c_void_p(base[3] + o), end - start, ctypes.c_void_p(hyper), ctypes.c_void_p(raw_stream))
        update_slice.keep = (ar.flat, ar.grad_flat, self._m, self._v, self._hyper)
        return update_slice

    # ------------------------------------------------------------------ checkpointing (torch.optim.Adam layout)
    def _group_params(self):
        return [p for g in self.param_groups for p in g["params"]]

    def state_dict(self):
        ar = self.model._arena
        if self._m is not None and self._arena is not ar and ar is not None:
            self._bind()                     # the arena was rebuilt since the last step (model.to(...)): carry the moments over first
        groups, k = [], 0
        for g in self.param_groups:
            d = {key: val for key, val in g.items() if key != "params"}
            d["params"] = list(range(k, k + len(g["params"])))
            k += len(g["params"])
            groups.append(d)
        state = {}
        if self._m is not None and self._arena is ar:
            t = self.step_count()
            if t > 0:
                for i, p in enumerate(self._group_params()):
                    j = ar.index.get(id(p))
                    if j is None:
                        continue
                    # compact copies, not views: pickling a view would serialise the whole arena once per parameter
                    state[i] = {"step": torch.tensor(float(t)),
                                "exp_avg": ar._view(self._m, j, p.shape).clone(),
                                "exp_avg_sq": ar._view(self._v, j, p.shape).clone()}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        ar = self._bind()
        groups = state_dict["param_groups"]
        if len(groups) != len(self.param_groups) or any(len(a["params"]) != len(b["params"]) for a, b in zip(groups, self.param_groups)):
            raise ValueError("loaded state dict has different parameter groups")
        for mine, theirs in zip(self.param_groups, groups):
            for key, val in theirs.items():
                if key != "params":
                    mine[key] = val
        state = state_dict.get("state", {})
        steps = set()
        with torch.no_grad():
            self._m.zero_()
            self._v.zero_()
            for i, p in enumerate(self._group_params()):
                st = state.get(i, state.get(str(i)))
                j = ar.index.get(id(p))
                if st is None or j is None:
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError("optimizer state %d has shape %s, parameter has %s" % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
                ar._view(self._m, j, p.shape).copy_(st["exp_avg"])
                ar._view(self._v, j, p.shape).copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            warnings.warn("FusedAdam keeps ONE step count; the loaded per-parameter counts differ (%s), using the largest" % sorted(steps))
        self._set_step(max(steps) if steps else 0)
        self._hyper_sent = None
