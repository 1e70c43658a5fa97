"""The training step as a recorded launch list.

The eager step costs the host ~16 us per kernel launch (the Python tape of engine.py, ctypes marshalling, torch
allocations, autograd): ~37 ms for the ~2 300 launches of an R101 step — as long as the GPU needs to run them.  A
captured hipGraph does not help on this stack (ROCm 7.0 replays a graph node by node from the host: 13 us per node,
30 us with the weight-gradient branch forked — measured), so the step is recorded one level up instead:

  * ``ReplayedTrainStep`` runs ONE step through an autograd-free copy of the step body (forward tape, loss kernels,
    gradient zeroing, the reverse tape on both HIP streams, reducer collectives, FusedAdam) while ``_lib.TAPE`` collects
    every C-ABI launch with its final arguments (device pointers, geometry structs, stream handles) and every
    stream/event/collective operation as ``(callable, args)``;
  * the recording runs inside a private ``torch.cuda.MemPool``: activations, gradients and scratch of the step get
    addresses that no other allocation can take, so the list stays valid; model parameters, optimizer state and the
    input tensors live outside and are referenced in place;
  * every later call re-issues the list — a tight loop of ~2 300 foreign calls (~6 us each) — after copying the new
    batch into the recorded input tensors.

Same interface as ``training.batch_processor.train_step``:

    step = ReplayedTrainStep(model, optimizer)
    loss, saved_for_log = step(inputs, gts)

The parameters after a step are bit-identical to the eager step's (same kernels, same order, same scratch sizes; the heat-map
loss runs as one launch over the internal tensors whose gradients round exactly as the API path's export -> loss -> import chain,
and whose logged loss values agree to f32 summation order: tests/test_replay_gpu.py).
Re-recorded when the model moves, the set of trainable parameters / BN modes / compute dtype changes, or a new input
signature arrives.  'prn_subnet' (host-made dropout seeds) and gradient clipping stay on the eager path.
"""
import itertools
import os
from collections import OrderedDict

import torch

from . import _lib, ops
from .engine import Ctx
from .network import losses

_epoch = itertools.count(1 << 20)


class _Entry(object):
    __slots__ = ("tape", "pool", "img", "gts", "logv", "names", "keep", "sig", "loss")


class ReplayedTrainStep(object):
    def __init__(self, model, optimizer, eager_steps=1, fused_mse=None, max_entries=None, anno_bucket=8):
        self.model = model
        self.opt = optimizer
        # heat-map loss + gradients in one launch over the internal tensors (MPN_FUSED_MSE=0: the API path's kernel chain)
        self.fused_mse = (os.environ.get("MPN_FUSED_MSE", "1") != "0") if fused_mse is None else bool(fused_mse)
        # Adam bucket by bucket behind each bucket's all-reduce (ddp.GradReducer's finishing stream) instead of one launch after
        # backward.  MPN_BUCKET_ADAM: unset = when a data-parallel reducer is attached (it removes the exposed last collective + the
        # whole-arena Adam from the N > 1 critical path); 1 = also on one GPU through the group-free schedule (measured +0.1 ms at the
        # headline size: the update competes with backward for HBM instead of following it, profiles/r06_bucket_adam_ab.txt); 0 = never
        mode = os.environ.get("MPN_BUCKET_ADAM", "")
        self.bucket_mode = mode if mode in ("0", "1") else "auto"
        self._can_bucket = hasattr(optimizer, "begin_bucketed")
        self.eager_steps = max(1, int(eager_steps))       # steps that fill host-side caches (anchors, transpose table, Adam state)
        # One recording per input signature, each owning a MemPool with a whole step's activations (GBs at the headline size).  The
        # reference's bbox_collater pads annotations to the per-batch maximum, so detection training sees a new signature for nearly
        # every distinct max_num_annots: the cache is an LRU of `max_entries` recordings (the evicted one's pool is released), and
        # annotation tensors are padded with -1 rows (FocalLoss drops them, losses.py:47) to a multiple of `anno_bucket` rows before
        # they become part of the key, which folds most of those signatures into a few
        self.max_entries = max(1, int(os.environ.get("MPN_REPLAY_MAX_ENTRIES", "4") if max_entries is None else max_entries))
        self.anno_bucket = max(1, int(anno_bucket))
        self._entries = OrderedDict()
        self._seen = {}
        self._warned = set()
        self.replays = 0
        self.evictions = 0

    # ------------------------------------------------------------------ the autograd-free step body
    def _body(self, img, subnet, tensors):
        m = self.model
        eng = m._engine
        want_kp = subnet in ("keypoint_subnet", "train_both")
        want_det = subnet in ("detection_subnet", "train_both")
        m._prepare(img)
        ctx = Ctx(True)
        c2, c3, c4, c5 = eng.backbone(ctx, img)
        grads = {}
        kp8 = det2 = None
        # same operation order as poseNet.keypoint_forward / detection_forward / train_both_forward (the reverse tape, hence the
        # order in which gradients accumulate into shared feature maps, follows it)
        kp_feats = eng.kp_pyramid(ctx, c2, c3, c4, c5) if want_kp else None
        det_feats = eng.det_pyramid(ctx, c3, c4, c5) if want_det else None
        # heat-map loss: one launch over the network's internal tensors where the geometry allows (losses.mse_train_supported),
        # else the API path's export -> loss -> import chain.  Both produce the same gradient bits.
        fused_mse = False
        if want_kp:
            hp = kp_feats[0]
            fused_mse = self.fused_mse and tensors[0].shape == (hp.B, 18, hp.H, hp.W) and hp.H % 8 == 0 and hp.W % 8 == 0 \
                and tensors[0].shape == tensors[1].shape
            pred, saved = eng.keypoint_head(ctx, kp_feats, True, internal=fused_mse)
            if fused_mse and not losses.mse_train_supported(saved + [pred], tensors[0]):
                raise _lib.MpnError("recorded train step: the keypoint head's internal geometry does not fit the one-pass loss kernel; "
                                    "construct ReplayedTrainStep(..., fused_mse=False)")
        if want_det:
            cls, reg = eng.detection_head(ctx, det_feats)
        m._finish_forward(ctx)
        dev = img.device
        ones = self._ones(dev)
        if want_kp and fused_mse:      # d(total)/d(heat-map total) = 1, known before the loss value is: gradients in the same pass
            kp8, kgrads = losses.mse_train_raw(saved + [pred], tensors[0], tensors[1], ones, eng.cdt)
            grads.update(zip(("k0", "k1", "k2", "k3", "pred"), kgrads))
        elif want_kp:
            heat = ops.nchw_to_nhwc_f32(tensors[0].detach().float())
            wgt = ops.nchw_to_nhwc_f32(tensors[1].detach().float())
            pm = [losses._pixel_major(p) for p in saved + [pred]]
            kp8 = losses.mse_forward_raw(pm, heat, wgt)
        if want_det:
            anno = tensors[2] if want_kp else tensors[0]
            det2, fsaved = losses.focal_forward_raw(cls, reg, m.anchors(img), anno)
        logv = torch.zeros(12, dtype=torch.float32, device=dev)
        _lib.call("mpn_step_log", ops.ptr(kp8), ops.ptr(det2), ops.ptr(logv), ops.stream_ptr())
        self.opt.zero_grad()
        if want_kp and not fused_mse:      # d(total)/d(heat-map total) = 1 (posenet.py:387 sums the level losses; the combined step adds the two totals)
            for slot, g in zip(("k0", "k1", "k2", "k3", "pred"), losses.mse_backward_raw(pm, heat, wgt, ones, [True] * 5)):
                grads[slot] = g
        if want_det:     # d(total)/d(cls loss) = d(total)/d(reg loss) = 1 (posenet.py:417-421)
            grads["cls"], grads["reg"] = losses.focal_backward_raw(fsaved, ones)
        if self.bucketed_update:
            # the optimizer update of a bucket runs as soon as the bucket's gradients (and, data-parallel, its all-reduce) are
            # complete, on the reducer's finishing stream under the rest of backward; one GPU: the same schedule without a group
            sched = m._reducer if m._reducer is not None else self._local_schedule()
            eng.run_backward(ctx, grads, schedule=sched, on_bucket=self.opt.begin_bucketed())
        else:
            eng.run_backward(ctx, grads)
            self.opt.step()
        return logv, want_kp, want_det

    @property
    def bucketed_update(self):
        if not self._can_bucket or self.bucket_mode == "0":
            return False
        return self.bucket_mode == "1" or self.model._reducer is not None

    def _local_schedule(self):
        from .ddp import GradReducer
        ar = self.model._arena
        sc = getattr(self, "_local", None)
        if sc is None or sc.arena is not ar:
            sc = self._local = GradReducer(ar, local=True, bucket_mb=float(os.environ.get("MPN_BUCKET_MB", "32")))
        return sc

    def _ones(self, dev):
        t = getattr(self, "_ones_t", None)
        if t is None or t.device != dev:
            t = self._ones_t = torch.ones(2, dtype=torch.float32, device=dev)
        return t

    @staticmethod
    def _log_names(want_kp, want_det):
        names = []
        if want_kp:
            kn = losses.build_names()
            names += [(kn[j * 2], j) for j in range(5)] + [("max_ht", 6), ("min_ht", 7)]
        if want_det:
            names += [("total_loss", 8), ("classification_loss", 9), ("regression_loss", 10)]
        return names

    def _state_sig(self):
        m = self.model
        ar = m._arena
        return (id(ar), id(ar.grad_flat), tuple(p.requires_grad for p in ar.params), tuple(b.training for b in m._bns),
                m.compute_dtype, id(m._reducer), m._engine.overlap_wgrad, m._engine.fork_every, id(self.opt._m), self.bucketed_update,
                m._engine.conv2_classes, m._engine.virtual_concat)

    # ------------------------------------------------------------------ call
    def __call__(self, inputs, gts):
        (img, subnet), = inputs
        gts = list(gts)
        if gts[0] != subnet:
            raise ValueError("inputs and gts name different subnets (%r vs %r)" % (subnet, gts[0]))
        tensors = gts[1:]
        if subnet not in ("keypoint_subnet", "detection_subnet", "train_both"):
            from .training.batch_processor import train_step
            return train_step(self.model, self.opt, inputs, gts)
        ops.check_device(img)
        # The recorded launches read the input tensors IN PLACE; a conversion inside the body would be a torch op that is not on the
        # list, and replays would keep reading the first batch's converted copy.  So inputs the eager path would convert (float64
        # targets, a permuted image) are converted HERE, outside the recording, with a one-time warning: the copies are what the
        # step reads (a replay copies them into the recorded tensors like any new batch).
        img = self._as_f32(img, "image batch")
        tensors = [self._as_f32(g, "target %d" % i) for i, g in enumerate(tensors)]
        if subnet in ("detection_subnet", "train_both") and self.anno_bucket > 1:
            ai = 2 if subnet == "train_both" else 0
            if ai < len(tensors) and tensors[ai].dim() == 3:
                tensors[ai] = self._bucket_annotations(tensors[ai])
        key = (subnet, tuple(img.shape), img.dtype, tuple((tuple(t.shape), t.dtype) for t in tensors))
        ent = self._entries.get(key)
        if ent is not None and ent.sig != self._state_sig():
            ent = None
            self._drop(key)
        if ent is not None:
            self._entries.move_to_end(key)
        if ent is None:
            n = self._seen.get(key, 0)
            self._seen[key] = n + 1
            if n < self.eager_steps:
                logv, kp, det = self._body(img, subnet, tensors)          # plain eager execution of the same body
                return logv[11], self._log(logv, self._log_names(kp, det))
            ent = self._record(key, img, subnet, tensors)                  # the recording IS this call's step
            return ent.loss, self._log(ent.logv, ent.names)
        if ent.img.data_ptr() != img.data_ptr():
            ent.img.copy_(img, non_blocking=True)
        for dst, src in zip(ent.gts, tensors):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.opt.sync_hyper()
        for fn, args, is_c in ent.tape:
            st = fn(*args)
            if is_c and st:
                raise _lib.MpnError("%s failed during replay" % getattr(fn, "__name__", fn))
        self.replays += 1
        return ent.loss, self._log(ent.logv, ent.names)

    def _as_f32(self, t, what):
        if t.dtype == torch.float32 and t.is_contiguous():
            return t
        if what not in self._warned:
            self._warned.add(what)
            import warnings
            warnings.warn("recorded train step: %s arrives as %s, contiguous=%s — converted to contiguous float32 on every step; "
                          "do it in the data pipeline to save the copy" % (what, t.dtype, t.is_contiguous()))
        return t.detach().to(torch.float32).contiguous()

    def _bucket_annotations(self, anno):
        """[B, n, 5] -> [B, ceil(n / bucket) * bucket, 5], the new rows filled with -1 (= padding: losses.py:47 drops them)."""
        n = anno.shape[1]
        m = max(self.anno_bucket, (n + self.anno_bucket - 1) // self.anno_bucket * self.anno_bucket)
        if m == n:
            return anno
        out = torch.full((anno.shape[0], m, anno.shape[2]), -1.0, dtype=anno.dtype, device=anno.device)
        out[:, :n] = anno
        return out

    def _drop(self, key):
        """Forget a recording and give its pool back.  The launches of its last replay may still be running on either stream, and
        the allocator knows nothing of kernels launched through the C ABI: wait for the device first."""
        ent = self._entries.pop(key, None)
        if ent is None:
            return
        torch.cuda.synchronize()
        ent.tape = ent.keep = ent.img = ent.gts = ent.logv = ent.loss = None
        ent.pool = None

    @staticmethod
    def _log(logv, names):
        vals = losses._log_values(logv)          # one D2H copy behind the step (floats, or LazyFloat with set_lazy_log)
        return OrderedDict((k, vals[i]) for k, i in names)

    # ------------------------------------------------------------------ record
    def _record(self, key, img, subnet, tensors):
        m = self.model
        m._arena.ensure_grads()
        self.opt.sync_hyper()
        self._ones(img.device)
        ent = _Entry()
        ent.img = img if img.is_contiguous() else img.contiguous()      # the tensors of THIS call become the step's inputs
        ent.gts = list(tensors)
        epoch = next(_epoch)
        pool = torch.cuda.MemPool()
        tape = []
        was_on = ops.KERNEL_EVENTS.on
        ops.KERNEL_EVENTS.on = False
        ops.WS_EPOCH = epoch
        _lib.TAPE = tape
        try:
            with torch.cuda.use_mem_pool(pool):
                logv, kp, det = self._body(ent.img, subnet, ent.gts)
        finally:
            _lib.TAPE = None
            ops.WS_EPOCH = 0
            ops.KERNEL_EVENTS.on = was_on
        ent.tape, ent.pool, ent.logv = tape, pool, logv
        ent.loss = logv[11]
        ent.names = self._log_names(kp, det)
        ent.keep = ops.take_epoch_workspaces(epoch)
        ent.sig = self._state_sig()
        self._entries[key] = ent
        while len(self._entries) > self.max_entries:          # least recently used recording out (its pool is released)
            old = next(iter(self._entries))
            self._drop(old)
            self._seen.pop(old, None)                          # ... and forgotten: a signature that returns pays its eager step(s) again
            self.evictions += 1
        if len(self._seen) > 8 * self.max_entries:             # signatures seen once and never again must not accumulate either
            for k in [k for k in self._seen if k not in self._entries][: len(self._seen) - 4 * self.max_entries]:
                del self._seen[k]
        if self.evictions >= 4 * self.max_entries and "thrash" not in self._warned:
            self._warned.add("thrash")
            import warnings
            warnings.warn("recorded train step: %d recordings evicted with %d slots — every eviction re-records a whole step's memory pool; "
                          "raise MPN_REPLAY_MAX_ENTRIES / max_entries, a larger anno_bucket, or fixed input shapes" % (self.evictions, self.max_entries))
        return ent
