"""Data-parallel gradient reduction: one process per GPU, RCCL all-reduce over xGMI.

Replaces the reference's only parallelism, single-process ``nn.DataParallel`` via
``ListDataParallel`` (datasets/data_parallel.py:16-87, wrapped at training/trainer.py:170), which
every step broadcasts all weights, scatters inputs, gathers all five heat-map tensors to GPU 0,
computes the loss there and reduce-adds gradients back.  Here each rank owns a replica and its shard
of the minibatch (images are independent; BatchNorm statistics stay per replica exactly as under
DataParallel), computes its local-mean loss, and only gradients cross xGMI:

  * buckets are CONTIGUOUS slices of the flat gradient arena (arena.py) — no copy-in/copy-out;
  * buckets cover trainable parameters only (frozen groups are never reduced) and are cut at
    ~``bucket_mb`` so an 8-GPU ring/mesh step is link-bandwidth- rather than latency-bound;
  * backward is a tape, so the engine knows the moment a parameter's last gradient contribution has
    been enqueued; when every parameter of a bucket is ready the bucket's all-reduce is issued with
    ``async_op=True`` (torch.distributed runs it on RCCL's own stream, ordered after the producing
    kernels — weight gradients run on the engine's side stream, so the collective is enqueued behind that
    stream after it has been made to wait for the main one) and overlaps the remaining dgrad/wgrad work.
    Buckets complete in reverse-forward order (heads -> FPN -> layer4 ... conv1);
  * ``finish()`` waits for the outstanding handles and averages (``ReduceOp.AVG`` on RCCL; SUM then
    an in-place scale on backends without AVG, e.g. gloo in the CPU tests).

Equal shard sizes => the average of local-mean gradients equals the gradient of the global mean
(SURVEY.md 8e).  Parameters and BN buffers are broadcast from rank 0 once at attach time.
"""
import os

import torch
import torch.distributed as dist

from ._lib import gpu_op


class GradReducer(object):
    def __init__(self, arena, process_group=None, bucket_mb=32.0, param_filter=None):
        self.arena = arena
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = max(1, int(bucket_mb * 1024 * 1024 / 4))
        self.buckets = []          # dict(start, end, params:set(idx), pending:int, handle)
        self.param_bucket = {}
        self._build()
        self.handles = []
        self.launched = 0
        backend = dist.get_backend(process_group)
        self.use_avg = backend == "nccl"
        # gloo (debug / single-GPU tests) reduces host memory: device buckets are staged through pinned buffers
        self.stage_host = backend != "nccl" and arena.device.type == "cuda"
        self.launch_stream = None      # set by the engine when weight gradients are produced on a side stream
        self.pre_launch = None         # engine hook: hand over side-stream work still waiting for a fork point
        self._trainable_sig = tuple(p.requires_grad for p in arena.params)
        self.bucket_mb = float(bucket_mb)
        self.measure = False           # bench.py: bracket finish()'s waits with a pair of timing events (exposed_ms)
        self._ev = None

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
            if cur is not None and cur["end"] == s and (cur["end"] - cur["start"]) < self.bucket_elems:
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
    def begin(self):
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

    def param_ready(self, p):
        i = self.arena.index.get(id(p))
        b = self.param_bucket.get(i)
        if b is None:
            return
        bk = self.buckets[b]
        bk["pending"] -= 1
        if bk["pending"] == 0:
            gpu_op(self._launch, bk)           # (recordable: replay.py re-issues the collective at this point of the step)

    def _launch(self, bk):
        if self.pre_launch is not None:
            self.pre_launch()
        view = self.arena.grad_flat[bk["start"]: bk["end"]]
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        if self.stage_host:
            if self.launch_stream is not None:
                torch.cuda.current_stream().wait_stream(self.launch_stream)
            host = view.cpu()                       # synchronises with the producing kernels
            h = dist.all_reduce(host, op=op, group=self.pg, async_op=True)
            self.handles.append((h, (view, host)))
            self.launched += 1
            bk["pending"] = -1
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
        self.launched += 1
        bk["pending"] = -1

    def finish(self):
        # parameters that received no gradient this step (unused heads) still have to be reduced so
        # that every rank issues the same collectives in the same order
        for bk in self.buckets:
            if bk["pending"] >= 0:
                self._launch(bk)
        if self.measure and torch.cuda.is_available():
            # bench.py: GPU time the launch stream spends in the waits below = all-reduce time NOT hidden behind backward
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
        if self.measure and self._ev is not None:
            self._ev[1].record()
        self.handles = []

    def exposed_ms(self):
        """GPU time between the end of backward on the launch stream and the completion of the last collective, for the most
        recent step (needs ``measure = True`` before that step and a device synchronisation after it)."""
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
