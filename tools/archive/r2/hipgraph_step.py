"""REJECTED PATH, kept out of the package as the record (round 2: a captured hipGraph replays node by node from the host on this
stack and is slower than the recorded launch list, multiposenet/pytorch_amd/replay.py).  Was multiposenet/pytorch_amd/graph.py through round 4.

The training step as ONE hipGraph launch.

The reference's step (training/trainer.py:245-259: forward, build_loss, zero_grad, backward, step) costs the
host ~2 000 kernel launches here (the Python tape of engine.py); enqueueing them takes about as long as the
GPU needs to run them.  ``GraphedTrainStep`` captures the whole step once per input signature —
forward, losses, gradient zeroing, the reverse tape on both HIP streams (weight gradients fork onto the side
stream and join before the optimizer), the RCCL bucket all-reduces when a reducer is attached, and FusedAdam
(whose scalars live in device memory, optim.py) — and replays it with a single ``hipGraphLaunch``.

    step = GraphedTrainStep(model, optimizer)
    loss, saved_for_log = step(inputs, gts)          # same arguments as training.batch_processor.train_step

Rules of the road:
  * the first ``eager_steps`` calls of a signature run eagerly (they populate caches that need host copies:
    anchors, the weight-transpose table, optimizer state); the next call captures and replays;
  * the tensors passed on the capturing call BECOME the graph's static inputs (they are kept, not cloned); later
    calls copy their tensors into them — a no-op when the caller passes the same tensors every step, as bench.py
    does.  ``loss`` is a static 0-dim tensor overwritten by the next replay;
  * log values are read AFTER the replay: plain floats by default (one host sync, the reference's
    behaviour) or ``LazyFloat`` proxies with ``losses.set_lazy_log(True)``;
  * scratch buffers and activations used inside the graph belong to it (torch's graph memory pool); a model
    moved to another device, a changed set of trainable parameters, or a different BN mode triggers a re-capture;
  * PRN training (dropout seeds are host-made) is not captured — use the eager step for 'prn_subnet'.
"""
import itertools
import os
from collections import OrderedDict

import torch

import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiposenet.pytorch_amd import ops
from multiposenet.pytorch_amd.network import losses

_epoch = itertools.count(1)


class _Entry(object):
    __slots__ = ("graph", "img", "gts", "loss", "log", "log_srcs", "keep", "sig")


class GraphedTrainStep(object):
    def __init__(self, model, optimizer, eager_steps=2, fork_every=None):
        self.model = model
        self.opt = optimizer
        self.eager_steps = max(1, int(eager_steps))
        # layers of weight-gradient work handed to the side stream per fork point inside the captured graph (every
        # cross-stream edge of a hipGraph costs a barrier packet and a cache write-back at replay)
        self.fork_every = int(fork_every if fork_every is not None else os.environ.get("HIPGRAPH_FORK_EVERY", "8"))
        self._entries = {}
        self._seen = {}
        self.replays = 0

    # ------------------------------------------------------------------ the step body (also the eager path)
    def _body(self, img, subnet, gts):
        output, saved_for_loss = self.model([img, subnet])
        loss, saved_for_log = self.model.build_loss(saved_for_loss, subnet, *gts)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return loss, saved_for_log

    def _state_sig(self):
        m = self.model
        ar = m._arena
        return (id(ar), id(ar.grad_flat), tuple(p.requires_grad for p in ar.params),
                tuple(b.training for b in m._bns), m.compute_dtype, id(m._reducer), m._engine.overlap_wgrad, self.fork_every)

    def __call__(self, inputs, gts):
        (img, subnet), = inputs
        gts = list(gts)
        if gts[0] != subnet:
            raise ValueError("inputs and gts name different subnets (%r vs %r)" % (subnet, gts[0]))
        tensors = gts[1:]
        if subnet == "prn_subnet":
            return self._body(img, subnet, tensors)
        key = (subnet, tuple(img.shape), img.dtype, tuple((tuple(t.shape), t.dtype) for t in tensors))
        ent = self._entries.get(key)
        if ent is not None and ent.sig != self._state_sig():
            ent = None
            self._entries.pop(key)
        if ent is None:
            n = self._seen.get(key, 0)
            if n < self.eager_steps:
                self._seen[key] = n + 1
                return self._body(img, subnet, tensors)
            ent = self._capture(key, img, subnet, tensors)
        if ent.img.data_ptr() != img.data_ptr():
            ent.img.copy_(img, non_blocking=True)
        for dst, src in zip(ent.gts, tensors):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.opt.sync_hyper()
        ent.graph.replay()
        self.replays += 1
        return ent.loss, self._materialise_log(ent)

    # ------------------------------------------------------------------ capture
    def _capture(self, key, img, subnet, tensors):
        ops.check_device(img)
        self.model._arena.ensure_grads()
        self.opt.sync_hyper()
        ent = _Entry()
        ent.img = img if img.is_contiguous() else img.contiguous()
        ent.gts = list(tensors)
        ent.sig = self._state_sig()
        epoch = next(_epoch)
        graph = torch.cuda.CUDAGraph()
        log_srcs = []
        ops.WS_EPOCH = epoch
        losses.CAPTURE_LOG = log_srcs
        was_on = ops.KERNEL_EVENTS.on
        ops.KERNEL_EVENTS.on = False
        eager_fork = self.model._engine.fork_every
        self.model._engine.fork_every = max(1, self.fork_every)
        try:
            with torch.cuda.graph(graph):
                loss, log = self._body(ent.img, subnet, ent.gts)
        finally:
            ops.WS_EPOCH = 0
            losses.CAPTURE_LOG = None
            ops.KERNEL_EVENTS.on = was_on
            self.model._engine.fork_every = eager_fork
        ent.graph = graph
        ent.loss = loss.detach()
        ent.log = log
        ent.log_srcs = log_srcs
        ent.keep = ops.take_epoch_workspaces(epoch)
        self._entries[key] = ent
        return ent

    def _materialise_log(self, ent):
        vals = [losses._log_values(t) for t in ent.log_srcs]       # D2H copies enqueued behind the replay
        out = OrderedDict()
        for k, v in ent.log.items():
            out[k] = vals[v.slot][v.i] if isinstance(v, losses._Deferred) else v
        return out
