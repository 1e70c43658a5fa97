"""Shared helpers for the parity tests (test-side only)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.txt")


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")


def round_up(v, m):
    return (v + m - 1) // m * m


def to_act(x_nchw, dtype, device="cuda"):
    from multiposenet.pytorch_amd.ops import Act
    B, C, H, W = x_nchw.shape
    t = torch.zeros(B, H, W, round_up(C, 32), dtype=torch.float32)
    t[..., :C] = x_nchw.permute(0, 2, 3, 1)
    return Act(t.to(dtype).to(device), C)


def from_act(a):
    return a.t[..., : a.C].float().cpu().permute(0, 3, 1, 2).contiguous()


def w_krsc(w_oihw, dtype, device="cuda"):
    return w_oihw.permute(0, 2, 3, 1).contiguous().to(dtype).to(device)


def rnd(dtype, x):
    """Round a f32 CPU tensor through `dtype` (so the CPU reference sees the same operand values)."""
    return x.to(dtype).float()


def tol(dtype):
    return 2e-4 if dtype == torch.float32 else 2e-2


def check_close(name, got, ref, dtype, scale=None, factor=1.0):
    got = got.double()
    ref = ref.double()
    s = ref.abs().max().item() if scale is None else scale
    err = (got - ref).abs().max().item()
    lim = tol(dtype) * factor * max(s, 1e-6)
    report("%-60s err=%.3e  lim=%.3e  refmax=%.3e  %s" % (name, err, lim, s, "OK" if err <= lim else "FAIL"))
    assert err <= lim, "%s: max err %.3e > %.3e (ref max %.3e)" % (name, err, lim, s)


def rng_normal(seed, *shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)
