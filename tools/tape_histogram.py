#!/usr/bin/env python3
"""What one recorded training step consists of: histogram of the entries of the launch list (replay.py)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multiposenet.pytorch_amd.network.posenet import poseNet
from multiposenet.pytorch_amd.optim import FusedAdam
from multiposenet.pytorch_amd.replay import ReplayedTrainStep
from multiposenet.pytorch_amd.network import losses
losses.set_lazy_log(True)
dev = torch.device("cuda:0")
m = poseNet(101, compute_dtype=torch.bfloat16).to(dev)
bench.he_weights(m)
for p in m.prn.parameters():
    p.requires_grad = False
m.train()
opt = FusedAdam(m, lr=1e-4)
img, heat, wgt, anno = bench.synth(32, 480, dev, seed=100)
st = ReplayedTrainStep(m, opt)
for _ in range(3):
    st([[img, "train_both"]], ["train_both", heat, wgt, anno])
ent = list(st._entries.values())[0]
c = collections.Counter()
for fn, args, is_c in ent.tape:
    name = getattr(fn, "__name__", None) or repr(fn)
    if not is_c:
        name = "gpu_op:" + (getattr(fn, "__qualname__", None) or name)
    c[name] += 1
print("entries", len(ent.tape))
for k, v in c.most_common():
    print("%6d  %s" % (v, k))
