#!/usr/bin/env python3
"""Per-stage difference between two arithmetic types of the forward pass (fault localisation at full size).
usage: python tools/stage_diff.py [--layers 101] [--batch 32] [--size 480] [--mode train|eval] [--a bf16] [--b f32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=101)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=480)
    ap.add_argument("--mode", default="train")
    ap.add_argument("--a", default="bf16")
    ap.add_argument("--b", default="f32")
    args = ap.parse_args()
    from multiposenet.pytorch_amd.engine import Ctx
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from multiposenet.pytorch_amd import synthetic as weightgen
    torch.cuda.set_device(0)
    m = poseNet(args.layers, compute_dtype=torch.float32).cuda()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = weightgen.gen_state_dict(shapes, seed=0, flavour="he", skip_prefixes=("prn.",))
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    img = torch.from_numpy(weightgen.gen_images(90, args.batch, args.size, args.size)).cuda()

    def stages(dt):
        m.load_state_dict(sdt, strict=False)
        m.train() if args.mode == "train" else m.eval()
        m.compute_dtype = dt
        out = {}
        with torch.no_grad():
            m._prepare(img)
            eng, ctx = m._engine, Ctx(False)
            f = m.fpn
            c = eng.stem(ctx, img)
            out["stem"] = c
            for li, layer in enumerate((f.layer1, f.layer2, f.layer3, f.layer4)):
                for bi, blk in enumerate(layer):
                    c = eng.bottleneck(ctx, c, blk)
                    if bi in (0, len(layer) - 1) or (li == 2 and bi % 6 == 0):
                        out["layer%d.%d" % (li + 1, bi)] = c
                out["c%d" % (li + 2)] = c
            kp = eng.kp_pyramid(ctx, out["c2"], out["c3"], out["c4"], out["c5"])
            for n, a in zip(("fp2", "fp3", "fp4", "fp5"), kp):
                out[n] = a
            det = eng.det_pyramid(ctx, out["c3"], out["c4"], out["c5"])
            for n, a in zip(("p3", "p4", "p5", "p6", "p7"), det):
                out[n] = a
            pred, _ = eng.keypoint_head(ctx, kp, False)
            cls, reg = eng.detection_head(ctx, det)
            torch.cuda.synchronize()
        res = {k: v.t[..., :v.C].float() for k, v in out.items()}
        res["pred"] = pred.float()
        res["cls"] = cls.float()
        res["reg"] = reg.float()
        return res
    A = stages(DT[args.a])
    Bv = stages(DT[args.b])
    print("# R%d %s-mode BN, %dx%d batch %d: %s vs %s" % (args.layers, args.mode, args.size, args.size, args.batch, args.a, args.b))
    for k in A:
        a, b = A[k].double(), Bv[k].double()
        rl2 = float((a - b).norm() / max(float(b.norm()), 1e-30))
        # per-image rel-L2 spread: a tile/bounds bug hits some images or positions, rounding noise is uniform
        per = ((a - b).reshape(a.shape[0], -1).norm(dim=1) / b.reshape(b.shape[0], -1).norm(dim=1).clamp_min(1e-30))
        print("%-12s relL2 %.3e   per-image min %.3e max %.3e   |b|max %.3g  nan %d" % (k, rl2, float(per.min()), float(per.max()), float(b.abs().max()), int(torch.isnan(a).sum())))


if __name__ == "__main__":
    main()
