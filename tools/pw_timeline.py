#!/usr/bin/env python3
"""Phase timeline of conv_pw_kernel from in-kernel s_memtime stamps (per wave: start, pixel tile landed, then for each strip
main loop done / stores issued).  Prints mean and spread of each phase over all waves, in shader cycles and (at 2.1 GHz) us."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from multiposenet.pytorch_amd import ops
from multiposenet.pytorch_amd._lib import call

dt, dev = torch.bfloat16, "cuda"
B, H, Cin, Cout = 32, 30, 256, 1024
variant = int(os.environ.get("PW_VARIANT", "0"))
P = B * H * H
x = ops.Act(torch.randn(B, H, H, Cin, device=dev).to(dt), Cin)
y = ops.Act(torch.randn(B, H, H, Cout, device=dev).to(dt), Cout)
by = ops.Act(torch.randn(B, H, H, Cout, device=dev).to(dt), Cout)
bz = ops.Act(torch.randn(B, H, H, Cout, device=dev).to(dt), Cout)
w = (torch.randn(Cout, 1, 1, Cin, device=dev) / Cin ** 0.5).to(dt)
st = ops.BNState(Cout, torch.device(dev))
st.mean.normal_(); st.invstd.uniform_(0.5, 1.5); st.scale.fill_(1.0); st.shift.zero_()
call("mpn_conv_pw_set_min_tiles", 0)


def run():
    if variant == 0:
        ops.conv_forward(x, w, Cout, 1, 1, 1, 0, want_stats=True, out=y)
    else:
        ops.conv_forward(x, w, Cout, 1, 1, 1, 0, out=y, accumulate=True, bnb=(by, bz, st, True))


for _ in range(3):
    run()
torch.cuda.synchronize()
nwg, nw = (P + 127) // 128, 8
buf = torch.zeros(nwg * nw * 8, dtype=torch.int64, device=dev)
call("mpn_conv_pw_debug_stamps", ops.ptr(buf))
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
flush.zero_()          # push the operands out of the caches
torch.cuda.synchronize()
run()
torch.cuda.synchronize()
call("mpn_conv_pw_debug_stamps", None)
t = buf.cpu().numpy().reshape(nwg, nw, 8).astype(np.float64)
t0 = t[:, :, 0].min()
names = ["start", "tile landed", "strip1 main done", "strip1 stores issued", "strip2 main done", "strip2 stores issued"]
print("variant", "plain+stats" if variant == 0 else "acc+bnb", " kernel span %.1f us (first start -> last stamp, 2.1 GHz)" % ((t[:, :, 5].max() - t0) / 2100.0))
for k, n in enumerate(names):
    v = t[:, :, k] - t0
    print("%-22s mean %8.0f cyc (%5.2f us)  min %8.0f  max %8.0f" % (n, v.mean(), v.mean() / 2100.0, v.min(), v.max()))
for k in range(1, 6):
    d = t[:, :, k] - t[:, :, k - 1]
    print("phase %-34s mean %8.0f cyc (%5.2f us)  p10 %8.0f  p90 %8.0f" % (names[k - 1] + " -> " + names[k], d.mean(), d.mean() / 2100.0, np.percentile(d, 10), np.percentile(d, 90)))
# waves 0-3 vs 4-7
for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    print(grp, " ".join("%7.0f" % (t[:, sl, k] - t0).mean() for k in range(6)))
