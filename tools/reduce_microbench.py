#!/usr/bin/env python3
"""mpn_reduce_partials on the slice counts / weight sizes of the R101 step (buffers cycled past the MALL)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiposenet.pytorch_amd import ops
from multiposenet.pytorch_amd._lib import call

CASES = [("1x1 1024x256 chunks 32", 262144, 32), ("3x3 256x256 chunks 15", 589824, 15), ("3x3 512x512 chunks 4", 2359296, 4),
         ("1x1 256x64 chunks 256", 16384, 256), ("3x3 512->256 chunks 8", 1179648, 8)]


def main():
    dev = "cuda"
    for name, n, chunks in CASES:
        sets = max(2, int(600e6 // (chunks * n * 4)) + 1)
        ws = [torch.randn(chunks, n, device=dev) for _ in range(sets)]
        dst = torch.zeros(n, device=dev)
        for i in range(3):
            call("mpn_reduce_partials", ops.ptr(ws[i % sets]), chunks, n, ops.ptr(dst), 1, ops.stream_ptr())
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        it = max(40, sets)
        e0.record()
        for i in range(it):
            call("mpn_reduce_partials", ops.ptr(ws[i % sets]), chunks, n, ops.ptr(dst), 1, ops.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / it
        print("%-28s %8.1f us  %7.1f GB/s" % (name, us, (chunks + 2) * n * 4 / us / 1e3), flush=True)
        dst.zero_()
        call("mpn_reduce_partials", ops.ptr(ws[0]), chunks, n, ops.ptr(dst), 0, ops.stream_ptr())
        want = ws[0][0].clone()
        for c in range(1, chunks):
            want += ws[0][c]
        assert torch.equal(dst, want), "sum order changed"


if __name__ == "__main__":
    main()
