#!/usr/bin/env python3
"""BASELINE config 5 on one MI355X: R101 'both' inference, 640x640, batch 64, fp16, with NMS for every image, heat-map peak
extraction and the PRN assignment.  Prints one JSON line per stage set.
usage: python tools/infer_bench.py [--batch 64] [--size 640] [--dtype f16] [--iters 10] [--no-fold]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-fold", action="store_true")
    ap.add_argument("--people", type=int, default=4, help="boxes per image handed to the PRN assignment / peaks per joint plane the calibrated threshold keeps")
    ap.add_argument("--candidates", type=int, default=1000, help="anchors per image above the 0.05 score threshold (the classification bias "
                    "of the random-weight network is shifted to get there; a trained detector passes a few hundred)")
    args = ap.parse_args()
    from multiposenet.pytorch_amd.evaluate.prn_process import prn_assign_arrays, prn_process_batch
    from multiposenet.pytorch_amd.network.joint_utils import NMS_batch, NMS_batch_arrays, body_peaks_flat
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from multiposenet.pytorch_amd import synthetic as weightgen
    import bench
    torch.cuda.set_device(0)
    dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    m = poseNet(101, compute_dtype=dt).cuda()
    bench.he_weights(m)
    sd = weightgen.gen_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items() if k.startswith("prn.")}, seed=3, flavour="he")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m.eval()
    m._engine.fold_bn = not args.no_fold
    img = torch.from_numpy(weightgen.gen_images(41, args.batch, args.size, args.size)).cuda()
    # calibrate the detector's output bias: a uniform logit shift so that ~args.candidates anchors per image exceed 0.05
    with torch.no_grad():
        _, (cls, _, _) = m([img[:4].contiguous(), "detection_subnet"])
        s = cls.float().flatten().clamp(1e-6, 1 - 1e-6)
        q = torch.quantile(s[torch.randperm(s.numel(), device=s.device)[:1000000]], 1.0 - args.candidates / float(cls.shape[1]))
        shift = float(np.log(0.05 / 0.95) - torch.log(q / (1 - q)))
        m.classificationModel.output.bias.data += shift

    def net_only():
        with torch.no_grad():
            m._prepare(img)
            from multiposenet.pytorch_amd.engine import Ctx
            eng, ctx = m._engine, Ctx(False)
            c2, c3, c4, c5 = eng.backbone(ctx, img)
            heat, _ = eng.keypoint_head(ctx, eng.kp_pyramid(ctx, c2, c3, c4, c5), False)
            return heat, eng.detection_head(ctx, eng.det_pyramid(ctx, c3, c4, c5))

    def net():
        with torch.no_grad():
            return m.forward_all_images(img)

    def full():
        heat, dets = net()
        peaks = NMS_batch({'thre1': 0.1}, heat, 4.0)
        kps, boxes = [], []
        for b in range(args.batch):
            rows = [tuple(p) + (j,) for j, pj in enumerate(peaks[b]) for p in pj[:20]]          # cap the noise peaks of random weights
            kps.append([[r[0], r[1], r[2], r[3], max(0, r[4] - 1)] for r in rows if r[4] != 1])
            bx = dets[b][2][dets[b][0] > 0.5][:8] if dets[b][0].numel() else []
            boxes.append([[float(v) for v in bb] for bb in bx if bb[2] - bb[0] >= 1 and bb[3] - bb[1] >= 1])
        return prn_process_batch(m, kps, boxes)
    # random-weight heat-maps are noise (hundreds of peaks per joint plane, each with its bicubic refinement): the peak threshold is
    # calibrated so that about `--people` peaks per plane survive, which is what the reference's 0.1 leaves on a trained model
    with torch.no_grad():
        h0, _ = m.forward_all_images(img[:4].contiguous())
    hv = h0.float().flatten()
    thre1 = float(torch.quantile(hv[torch.randperm(hv.numel(), device=hv.device)[:2000000]], 1.0 - args.people * 12.0 / (h0.shape[2] * h0.shape[3])))

    def finish_arrays(heat, anchors, cls):
        """decode'd boxes -> NMS, peaks, PRN assignment on flat arrays, on the CURRENT stream."""
        boxes, scores, kept = m.detect_padded(anchors, cls)
        pk, cnt = NMS_batch_arrays({'thre1': thre1}, heat, 4.0)
        peaks_xy, joint_off = body_peaks_flat(pk, cnt, keep=args.people)     # ... and at most `people` peaks per joint type reach the PRN stage
        nb = np.minimum(np.asarray(kept), args.people)
        sel = np.arange(boxes.shape[1])[None, :] < nb[:, None]
        b4 = boxes[:, :4].double().cpu().numpy()[sel[:, :4]]
        b4[:, 2:] -= b4[:, :2]                                                   # (x1, y1, x2, y2) -> (x, y, w, h)
        ok = (b4[:, 2] >= 1) & (b4[:, 3] >= 1)
        start = np.concatenate([[0], np.cumsum(np.add.reduceat(ok, np.concatenate([[0], np.cumsum(nb)[:-1]])) if ok.size else nb * 0)]).astype(np.int32)
        return prn_assign_arrays(m, peaks_xy, joint_off, b4[ok], start)

    def full_arrays():
        """The chain on flat arrays: padded batch detections, peaks as one array, candidates compacted on the device, matching in
        C++ (prn_assign_arrays); the 8 best NMS survivors of every image stand in for its people (no random-weight score exceeds
        0.5).  Returns keypoints [boxes, 17, 3]."""
        with torch.no_grad():
            heat, anchors, cls, _keep = m.forward_padded_begin(img)
        return finish_arrays(heat, anchors, cls)

    post = torch.cuda.Stream()

    def pipelined(nbatches):
        """The same chain as a serving loop (Tester.infer_images_batched(pipeline=True)): batch k + 1's network is enqueued before batch k
        is post-processed, and the post-processing launches go to a second stream that waits for batch k's network only."""
        pending, last = None, None
        for k in range(nbatches + 1):
            nxt = None
            if k < nbatches:
                with torch.no_grad():
                    item = m.forward_padded_begin(img)
                ev = torch.cuda.Event()
                ev.record()
                nxt = (item, ev)
            if pending is not None:
                (heat, anchors, cls, _keep), ev = pending
                post.wait_event(ev)
                with torch.cuda.stream(post):
                    last = finish_arrays(heat, anchors, cls)
            pending = nxt
        return last

    def full_lists():
        """The SAME workload as full_arrays() through the reference's list interface (tester.py:158-168: joint rows as a list of lists,
        boxes as a list per image -> prn_process_batch -> result dicts): what the interface itself costs."""
        with torch.no_grad():
            heat, boxes, scores, kept = m.forward_all_images_padded(img)
        pk, cnt = NMS_batch_arrays({'thre1': thre1}, heat, 4.0)
        b4 = boxes[:, :, :4].double().cpu().numpy() if boxes.dim() == 3 else boxes.double().cpu().numpy()
        kps, bl = [], []
        for b in range(args.batch):
            rows = []
            for j in range(pk.shape[1]):
                if j == 1:
                    continue                                               # the neck is dropped, later types shift down (tester.py:161-165)
                n = min(int(cnt[b, j]), args.people)
                if n:
                    rows.append(np.concatenate([pk[b, j, :n], np.full((n, 1), float(max(0, j - 1)))], 1))
            kps.append(np.concatenate(rows, 0).tolist() if rows else [])
            bb = b4[b, :min(int(kept[b]), args.people)]
            bl.append([r for r in bb.tolist() if r[2] - r[0] >= 1 and r[3] - r[1] >= 1])
        return prn_process_batch(m, kps, bl)
    for name, fn in (("network only (backbone, both pyramids, both heads)", net_only), ("network + decode + NMS for every image", net),
                     ("+ heat-map peaks + PRN assignment, reference list interface, threshold 0.1 (noise: hundreds of peaks per plane)", full),
                     ("+ heat-map peaks + PRN assignment (%d people / image; the SAME workload as the next line through the reference's list interface: "
                      "joint rows / boxes as Python lists in, result dicts out)" % args.people, full_lists),
                     ("+ heat-map peaks + PRN assignment (%d people / image; flat arrays, compact candidates, C++ matching)" % args.people, full_arrays),
                     ("the same chain as a serving loop: next batch's network enqueued before this batch is post-processed, post-processing "
                      "on a second stream (Tester.infer_images_batched(pipeline=True))", None)):
        if fn is None:
            ref = full_arrays()
            got = pipelined(3)
            assert np.array_equal(ref, got), "pipelined chain changed the results"
            torch.cuda.synchronize()
            nb = 3 * args.iters                                  # a serving loop runs for long: the one un-overlapped fill / drain step amortises
            t0 = time.perf_counter()
            pipelined(nb)
            torch.cuda.synchronize()
            dtm = (time.perf_counter() - t0) / nb
        else:
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                fn()
            torch.cuda.synchronize()
            dtm = (time.perf_counter() - t0) / args.iters
        print(json.dumps({"stage": name, "images_per_sec": round(args.batch / dtm, 1), "ms_per_batch": round(dtm * 1e3, 2), "batch": args.batch,
                          "size": args.size, "dtype": args.dtype, "bn_folded": not args.no_fold, "candidates_per_image": args.candidates}), flush=True)


if __name__ == "__main__":
    main()
