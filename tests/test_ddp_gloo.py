"""Data-parallel path on CPU: world_size 2 over gloo (127.0.0.1).  Exercises the real GradReducer
on a real poseNet arena (R50): bucket construction over trainable runs only, readiness-driven
launch order (reverse of registration, as backward produces them), averaging, equal-shard
equivalence (mean of shard gradients == gradient of the global batch), and the one-time broadcast.
The kernels themselves need the GPU; here gradients are written into the arena by the test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1234 + rank)          # different init per rank -> broadcast must fix it
        from multiposenet.pytorch_amd import ddp
        from multiposenet.pytorch_amd.network.posenet import poseNet
        m = poseNet(50)
        for p in m.prn.parameters():
            p.requires_grad = False
        for name, module in m.fpn.named_children():
            if name in ("conv6", "conv7"):
                for p in module.parameters():
                    p.requires_grad = False
        red = ddp.attach(m, bucket_mb=8.0)
        ar = m._arena
        # (1) broadcast: every rank now holds rank 0's parameters
        chk = torch.tensor([float(ar.flat.double().sum())], dtype=torch.float64)
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        assert all(float(g) == float(gathered[0]) for g in gathered), "parameters differ after broadcast"
        # (2) buckets: contiguous, ordered, only trainable ranges, none larger than ~bucket_mb + one param
        runs = ar.trainable_runs()
        assert sum(b["end"] - b["start"] for b in red.buckets) == sum(e - s for s, e in runs)
        for b in red.buckets:
            assert any(s <= b["start"] and b["end"] <= e for s, e in runs)
        frozen_off = ar.offsets[ar.index[id(m.fpn.conv6.weight)]]
        assert not any(b["start"] <= frozen_off < b["end"] for b in red.buckets)
        assert len(red.buckets) >= 10
        # (3) a "backward": per-rank gradients g_r = (rank+1) * pattern, marked ready in reverse order
        pattern = torch.arange(ar.total, dtype=torch.float32) % 97 / 97.0
        ar.grad_flat.copy_(pattern * (rank + 1))
        before_frozen = ar.grad_flat[frozen_off: frozen_off + 8].clone()
        red.begin()
        order = []
        trainable = [p for p in reversed(ar.params) if p.requires_grad]
        for p in trainable:
            n0 = red.launched
            red.param_ready(p)
            if red.launched != n0:
                order.append(red.param_bucket[ar.index[id(p)]])
        red.finish()
        assert order == sorted(order, reverse=True), "buckets must launch in reverse-registration (backward) order"
        assert red.launched == len(red.buckets)
        mean_scale = sum(r + 1 for r in range(world)) / float(world)
        for s, e in runs:
            assert torch.allclose(ar.grad_flat[s:e], pattern[s:e] * mean_scale, rtol=1e-6, atol=1e-7)
        assert torch.equal(ar.grad_flat[frozen_off: frozen_off + 8], before_frozen), "frozen range must not be reduced"
        # (4) unused-parameter step: only some params report ready; finish() still reduces every bucket
        ar.grad_flat.copy_(pattern * (rank + 1))
        red.begin()
        for p in trainable[:5]:
            red.param_ready(p)
        red.finish()
        assert red.launched == len(red.buckets)
        s, e = runs[0]
        assert torch.allclose(ar.grad_flat[s:e], pattern[s:e] * mean_scale, rtol=1e-6, atol=1e-7)
        # (4b) per-bucket optimizer update (round 6): every bucket is updated exactly once, right behind ITS all-reduce, with the
        # averaged gradient; the parameters equal "reduce everything, then one update over the arena"
        ar.grad_flat.copy_(pattern * (rank + 1))
        start = ar.flat.clone()
        seen = []

        def sgd(s_, e_, stream):
            seen.append((s_, e_))
            ar.flat[s_:e_] -= 0.5 * ar.grad_flat[s_:e_]
        red.begin(sgd)
        for p in trainable[: len(trainable) // 2]:
            red.param_ready(p)
        n_mid = len(seen)
        assert 0 < n_mid < len(red.buckets), "updates must start while 'backward' is still producing gradients"
        red.finish()
        assert sorted(seen) == sorted((b["start"], b["end"]) for b in red.buckets) and red.updated == len(red.buckets)
        want = start.clone()
        for s_, e_ in runs:
            want[s_:e_] -= 0.5 * pattern[s_:e_] * mean_scale
        assert torch.allclose(ar.flat, want, rtol=1e-6, atol=1e-7)
        assert torch.equal(ar.flat[frozen_off: frozen_off + 8], start[frozen_off: frozen_off + 8])
        red.begin()                                   # a plain backward afterwards must not update anything
        assert red.on_bucket is None
        red.finish()
        ar.flat.copy_(start)
        # (5) equal-shard equivalence on a real differentiable function of the arena
        x = torch.linspace(-1, 1, 8 * 16).view(8, 16)
        w = ar.flat[:16].clone().requires_grad_(True)
        full = ((x @ w) ** 2).mean()
        gfull, = torch.autograd.grad(full, w)
        shard = ddp.shard_batch(x, rank, world)
        gl, = torch.autograd.grad(((shard @ w) ** 2).mean(), w)
        dist.all_reduce(gl)
        assert torch.allclose(gl / world, gfull, rtol=1e-5, atol=1e-7)
        with pytest.raises(ValueError):
            ddp.shard_batch(torch.zeros(7, 3), rank, world)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1}
