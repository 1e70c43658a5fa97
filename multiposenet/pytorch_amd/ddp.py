"""Data-parallel gradient reduction: one process per GPU, RCCL all-reduce over xGMI.

Replaces the reference's only parallelism, single-process ``nn.DataParallel`` via
``ListDataParallel`` (datasets/data_parallel.py:16-87, wrapped at training/trainer.py:170), which
every step broadcasts all weights, scatters inputs, gathers all five heat-map tensors to GPU 0,
computes the loss there and reduce-adds gradients back.  Here each rank owns a replica and its shard
of the minibatch (images are independent; BatchNorm statistics stay per replica exactly as under
DataParallel), computes its local-mean loss, and only gradients cross xGMI:

  * buckets are CONTIGUOUS slices of the flat gradient arena (arena.py) — no copy-in/copy-out;
  * buckets cover trainable parameters only (frozen groups are never reduced) and are cut at
    ~``bucket_mb`` so an 8-GPU ring/mesh step is link-bandwidth- rather than latency-bound; the first bucket of the arena — the
    last one backward completes, i.e. the un-overlapped tail — at a quarter of that;
  * backward is a tape, so the engine knows the moment a parameter's last gradient contribution has
    been enqueued; when every parameter of a bucket is ready the bucket's all-reduce is issued with
    ``async_op=True`` (torch.distributed runs it on RCCL's own stream, ordered after the producing
    kernels — weight gradients run on the engine's side stream, so the collective is enqueued behind that
    stream after it has been made to wait for the main one) and overlaps the remaining dgrad/wgrad work.
    Buckets complete in reverse-forward order (heads -> FPN -> layer4 ... conv1);
  * ``finish()`` waits for the outstanding handles and averages (``ReduceOp.AVG`` on RCCL; SUM then
    an in-place scale on backends without AVG, e.g. gloo in the CPU tests);
  * **per-bucket optimizer update** (round 6): when the step owns its optimizer (the recorded step, replay.py) the reducer is
    given ``on_bucket`` — FusedAdam's slice update.  Each bucket then goes through a third HIP stream, the *finishing stream*:
    it waits for the two producing streams (events, no stall on either), carries the collective, waits for it, and runs
    ``mpn_adam_step_dev`` over exactly that arena slice.  Backward never waits for a collective, the weight-gradient stream
    never waits for one either, and ``finish()`` only joins the finishing stream: what is exposed at N > 1 is the LAST
    bucket's all-reduce + its update, not the last all-reduce + a whole-arena Adam.  ``GradReducer(arena, local=True)`` is
    the same schedule without a process group (one GPU): the Adam update of a bucket runs under the rest of backward.

Equal shard sizes => the average of local-mean gradients equals the gradient of the global mean
(SURVEY.md 8e).  Parameters and BN buffers are broadcast from rank 0 once at attach time.
"""
import os

import torch
import torch.distributed as dist

from ._lib import gpu_op


class GradReducer(object):
    def __init__(self, arena, process_group=None, bucket_mb=32.0, param_filter=None, local=False):
        self.arena = arena
        self.pg = process_group
        self.local = bool(local)       # no process group: only the readiness schedule (per-bucket optimizer updates on one GPU)
        self.world = 1 if self.local else dist.get_world_size(process_group)
        self.bucket_elems = max(1, int(bucket_mb * 1024 * 1024 / 4))
        self.buckets = []          # dict(start, end, params:set(idx), pending:int, handle)
        self.param_bucket = {}
        self._build()
        self.handles = []
        self.launched = 0
        backend = "local" if self.local else dist.get_backend(process_group)
        self.backend = backend
        self.use_avg = backend == "nccl"
        # gloo (debug / single-GPU tests) reduces host memory: device buckets are staged through pinned buffers
        self.stage_host = backend == "gloo" and arena.device.type == "cuda"
        self.launch_stream = None      # set by the engine when weight gradients are produced on a side stream
        self.pre_launch = None         # engine hook: hand over side-stream work still waiting for a fork point
        self._trainable_sig = tuple(p.requires_grad for p in arena.params)
        self.bucket_mb = float(bucket_mb)
        self.measure = False           # bench.py: bracket finish()'s waits with a pair of timing events (exposed_ms)
        self._ev = None
        # per-bucket optimizer update: callable(start, end, raw stream handle) set for the duration of one backward pass by the
        # step that owns the optimizer (replay.py); None = gradients only (the caller steps the optimizer after backward)
        self.on_bucket = None
        self._finish_stream = None     # third stream: wait for both producers -> collective -> wait -> update, per bucket
        self.updated = 0               # buckets updated through on_bucket in the current / last step

    def finishing_stream(self):
        if self._finish_stream is None:
            self._finish_stream = torch.cuda.Stream(device=self.arena.device)
        return self._finish_stream

    def _build(self):
        ar = self.arena
        align = 64
        cur = None
        for i, p in enumerate(ar.params):
            if not p.requires_grad:
                if cur is not None:
                    self.buckets.append(cur)
                    cur = None
                continue
            s = ar.offsets[i]
            e = s + (ar.sizes[i] + align - 1) // align * align
            # the FIRST bucket of the arena (stem, layer1, ...) is the LAST to complete in backward: its all-reduce (and update) is the
            # tail nothing can hide, so it is cut at a quarter of the bucket size (8 MB of 32: ~0.15 instead of ~0.5 ms of ring time at
            # eight ranks); every other bucket completes with backward still running behind it
            limit = self.bucket_elems if self.buckets else max(1, self.bucket_elems // 4)
            if cur is not None and cur["end"] == s and (cur["end"] - cur["start"]) < limit:
                cur["end"] = e
                cur["params"].append(i)
            else:
                if cur is not None:
                    self.buckets.append(cur)
                cur = dict(start=s, end=e, params=[i])
        if cur is not None:
            self.buckets.append(cur)
        for b, bk in enumerate(self.buckets):
            for i in bk["params"]:
                self.param_bucket[i] = b

    def signature(self):
        return tuple((b["start"], b["end"]) for b in self.buckets)

    # ---- called by the engine -------------------------------------------------------------------
    def begin(self, on_bucket=None):
        """Start of a backward pass.  ``on_bucket`` (callable(start, end, raw stream handle) or None) is the optimizer's slice update
        when the step owns its optimizer; it is an ARGUMENT so that a recorded step re-installs it on every replay and an eager
        ``loss.backward()`` in between (which steps its optimizer afterwards) clears it."""
        self.on_bucket = on_bucket
        if tuple(p.requires_grad for p in self.arena.params) != self._trainable_sig:
            # parameters were (un)frozen after attach(): buckets cover trainable parameters only, so rebuild them — the
            # same change must be made on every rank (the bucket list is the collective schedule)
            self.buckets, self.param_bucket = [], {}
            self._build()
            self._trainable_sig = tuple(p.requires_grad for p in self.arena.params)
        for bk in self.buckets:
            bk["pending"] = len(bk["params"])
        self.handles = []
        self.launched = 0
        self.updated = 0

    def param_ready(self, p):
        i = self.arena.index.get(id(p))
        b = self.param_bucket.get(i)
        if b is None:
            return
        bk = self.buckets[b]
        bk["pending"] -= 1
        if bk["pending"] == 0:
            if self.pre_launch is not None:
                self.pre_launch()              # side-stream work still waiting for a fork point (recorded on its own)
            gpu_op(self._launch, bk)           # (recordable: replay.py re-issues the collective at this point of the step)

    def _launch(self, bk):
        view = self.arena.grad_flat[bk["start"]: bk["end"]]
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        self.launched += 1
        bk["pending"] = -1
        if self.on_bucket is not None:
            if view.is_cuda:
                self._launch_and_update(bk, view, op)
            else:                                   # host arena (the gloo CPU tests): same schedule, nothing to overlap
                if not self.local:
                    dist.all_reduce(view, op=op, group=self.pg)
                    if not self.use_avg:
                        view.mul_(1.0 / self.world)
                self.on_bucket(bk["start"], bk["end"], 0)
                self.updated += 1
            return
        if self.local:
            return
        if self.stage_host:
            if self.launch_stream is not None:
                torch.cuda.current_stream().wait_stream(self.launch_stream)
            host = view.cpu()                       # synchronises with the producing kernels
            h = dist.all_reduce(host, op=op, group=self.pg, async_op=True)
            self.handles.append((h, (view, host)))
            return
        if self.launch_stream is not None:
            # the bucket's gradients come from two streams (wgrad on the side stream, BN/bias pieces on the main one):
            # order the collective after both without stalling the main stream
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.launch_stream):
                self.launch_stream.wait_event(ev)
                h = dist.all_reduce(view, op=op, group=self.pg, async_op=True)
        else:
            h = dist.all_reduce(view, op=op, group=self.pg, async_op=True)
        self.handles.append((h, view))

    def _launch_and_update(self, bk, view, op):
        """Bucket -> finishing stream: wait for both producing streams, all-reduce, wait for it, optimizer update of the slice.
        Neither the main stream nor the weight-gradient stream waits for anything here."""
        fin = self.finishing_stream()
        ev = torch.cuda.Event()
        ev.record()                                 # main stream: BN / bias gradients and everything that READS these parameters
        fin.wait_event(ev)
        if self.launch_stream is not None:
            ev2 = torch.cuda.Event()
            ev2.record(self.launch_stream)          # weight-gradient stream: the bucket's last wgrad launch is already enqueued
            fin.wait_event(ev2)
        if not self.local:
            with torch.cuda.stream(fin):
                if self.stage_host:
                    host = view.cpu()               # gloo: host-staged and host-blocking (tests only)
                    dist.all_reduce(host, op=op, group=self.pg)
                    view.copy_(host.mul_(1.0 / self.world))
                else:
                    h = dist.all_reduce(view, op=op, group=self.pg, async_op=True)
                    h.wait()                        # the FINISHING stream waits for RCCL's stream; the host does not block
                    if not self.use_avg:
                        view.mul_(1.0 / self.world)
        self.on_bucket(bk["start"], bk["end"], fin.cuda_stream)
        self.updated += 1

    def finish(self):
        # parameters that received no gradient this step (unused heads) still have to be reduced so
        # that every rank issues the same collectives in the same order
        for bk in self.buckets:
            if bk["pending"] >= 0:
                self._launch(bk)
        timed = self.measure and torch.cuda.is_available() and self.arena.device.type == "cuda"
        if timed:
            # bench.py: GPU time the launch stream spends in the waits below = all-reduce (and, with per-bucket updates, the last
            # bucket's update) time NOT hidden behind backward
            if self._ev is None:
                self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._ev[0].record()
        for h, view in self.handles:
            h.wait()
            if isinstance(view, tuple):             # host-staged bucket (gloo with device gradients)
                view, host = view
                view.copy_(host.mul_(1.0 / self.world))
            elif not self.use_avg:
                view.mul_(1.0 / self.world)
        if self._finish_stream is not None and self.updated:
            torch.cuda.current_stream().wait_stream(self._finish_stream)      # join: every bucket's update is complete
        if timed:
            self._ev[1].record()
        self.handles = []

    def exposed_ms(self):
        """GPU time between the end of backward on the launch stream and the completion of the last collective (and of the last
        bucket's optimizer update when the step runs them per bucket), for the most recent step (needs ``measure = True``
        before that step and a device synchronisation after it)."""
        if self._ev is None:
            return None
        return float(self._ev[0].elapsed_time(self._ev[1]))


def broadcast_state(model, process_group=None, src=0):
    """One-time sync of parameters and BN buffers from ``src`` (the reference re-broadcasts every step)."""
    ar = model._arena
    staged = dist.get_backend(process_group) != "nccl" and ar.flat.is_cuda      # gloo moves host memory

    def bcast(t):
        if staged:
            h = t.detach().cpu()
            dist.broadcast(h, src=src, group=process_group)
            t.detach().copy_(h)
        else:
            dist.broadcast(t, src=src, group=process_group)
    bcast(ar.flat)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            bcast(m.running_mean)
            bcast(m.running_var)


def attach(model, process_group=None, bucket_mb=None, broadcast=True):
    """Turn ``model`` (a poseNet already on its device) into a data-parallel replica.

    ``bucket_mb`` defaults to ``MPN_BUCKET_MB`` (32): smaller buckets shorten the tail that cannot
    overlap backward and add collective launches."""
    if bucket_mb is None:
        bucket_mb = float(os.environ.get("MPN_BUCKET_MB", "32"))
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    if broadcast:
        broadcast_state(model, process_group)
    model._arena.ensure_grads()
    model._reducer = GradReducer(model._arena, process_group, bucket_mb)
    return model._reducer


def shard_batch(tensor, rank, world):
    """Contiguous equal shard of the global batch for this rank (SURVEY.md 8e partitioning)."""
    n = tensor.shape[0]
    if n % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (n, world))
    k = n // world
    return tensor[rank * k: (rank + 1) * k]
