#!/usr/bin/env python3
"""Does the step train?  R50 full posenet (train_both), 256x256, 8 images per batch, bf16 and fp32: 4 fixed synthetic batches
(Gaussian heat-map targets rendered by datasets/heatmap.py from random keypoints, random person boxes) are over-fitted for N steps
through the recorded step with FusedAdam (lr 1e-4, the reference's, training/multipose_keypoint_train.py:106-110).  Prints the
loss trajectory; the bf16 and fp32 curves should fall together.  Evidence of end-to-end health, not a parity test."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lr", type=float, default=1e-4)
    args = ap.parse_args()
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from multiposenet.pytorch_amd.optim import FusedAdam
    from multiposenet.pytorch_amd.replay import ReplayedTrainStep
    from multiposenet.pytorch_amd.datasets.heatmap import put_gaussian_maps
    import bench
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(7)
    S, B = args.size, args.batch
    batches = []
    for _ in range(4):
        img = torch.from_numpy(rs.uniform(-2, 2, (B, 3, S, S)).astype(np.float32)).to(dev)
        joints = np.zeros((B, 3, 18, 3), np.float64)
        joints[..., 0] = rs.uniform(16, S - 16, (B, 3, 18)); joints[..., 1] = rs.uniform(16, S - 16, (B, 3, 18)); joints[..., 2] = 1
        heat = put_gaussian_maps(torch.from_numpy(joints).to(dev), torch.full((B,), 3, dtype=torch.int32, device=dev), S, S, stride=4, sigma=7.0)
        wgt = torch.ones_like(heat)
        anno = np.full((B, 4, 5), -1, np.float32)
        for b in range(B):
            for k in range(3):
                x, y = rs.uniform(8, S - 120, 2); w, h = rs.uniform(40, 110, 2)
                anno[b, k] = [x, y, x + w, y + h, 0]
        batches.append((img, heat.contiguous(), wgt, torch.from_numpy(anno).to(dev)))
    for dt in (torch.bfloat16, torch.float32):
        torch.manual_seed(0)
        m = poseNet(50, compute_dtype=dt).to(dev)
        bench.he_weights(m)
        for p in m.prn.parameters():
            p.requires_grad = False
        m.train()
        opt = FusedAdam(m, lr=args.lr)
        step = ReplayedTrainStep(m, opt)
        curve = []
        for i in range(args.steps):
            img, heat, wgt, anno = batches[i % 4]
            loss, log = step([[img, "train_both"]], ["train_both", heat, wgt, anno])
            if i % 4 == 3:
                curve.append((float(loss), float(log["heatmap_loss"]), float(log["classification_loss"]), float(log["regression_loss"])))
        torch.cuda.synchronize()
        assert all(np.isfinite(c).all() for c in curve)
        name = str(dt).split(".")[1]
        pts = [0, 1, 2, 4, 9, 19, len(curve) - 1]
        print("%s, R50 %dx%d B=%d, lr %g, %d steps over 4 fixed batches (every 4th step shown: total | heat-map | cls | reg)" % (name, S, S, B, args.lr, args.steps))
        for k in pts:
            if k < len(curve):
                print("  step %4d   %9.5f | %9.5f | %8.5f | %8.5f" % ((k + 1) * 4, *curve[k]))
        print("  total loss fell by %.1fx, heat-map loss by %.1fx; replays %d" % (curve[0][0] / curve[-1][0], curve[0][1] / curve[-1][1], step.replays))


if __name__ == "__main__":
    main()
