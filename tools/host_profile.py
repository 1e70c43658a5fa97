#!/usr/bin/env python3
"""cProfile of the host side of one R101 480x480 B=32 bf16 training step (where do the ~37 ms of Python per step go)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from multiposenet.pytorch_amd.network.posenet import poseNet
from multiposenet.pytorch_amd.optim import FusedAdam

dev = torch.device("cuda:0")
model = poseNet(101, compute_dtype=torch.bfloat16).to(dev)
bench.he_weights(model)
for p in model.prn.parameters():
    p.requires_grad = False
model.train()
opt = FusedAdam(model, lr=1e-4, weight_decay=0.0)
img, heat, wgt, anno = bench.synth(32, 480, dev, seed=100)


def step():
    pred, (ks, ds) = model([img, "train_both"])
    loss, log = poseNet.build_loss((ks, ds), "train_both", heat, wgt, anno)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
