#!/usr/bin/env python3
"""Does hipExtStreamCreateWithCUMask confine a stream on this box, and how are the mask bits laid out?  Times a compute-bound torch kernel
on streams whose masks enable the first k of 256 bits / every 2nd bit / etc."""
import ctypes, os, sys, time
import torch
path = [l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l][0]
hip = ctypes.CDLL(path)
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)


torch.cuda.set_device(0)
x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)


def bench(stream, n=10):
    with torch.cuda.stream(stream):
        for _ in range(3):
            y = x @ x
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            y = x @ x
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


full = (1 << 256) - 1
print("plain torch stream        %.3f ms" % bench(torch.cuda.Stream()))
for name, bits in (("all 256 bits", full), ("first 128 bits", (1 << 128) - 1), ("first 64 bits", (1 << 64) - 1), ("every 2nd bit", int("01" * 128, 2)),
                   ("first 192 bits", (1 << 192) - 1), ("bits 0-31 only", (1 << 32) - 1)):
    try:
        print("%-24s %.3f ms" % (name, bench(masked_stream(bits))))
    except Exception as e:
        print(name, "failed:", e)
