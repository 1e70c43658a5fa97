#!/usr/bin/env python3
"""Where does the serving loop's time go?  Host timestamps around the pieces of the pipelined cfg5 chain while the NEXT batch's
network runs on the main stream: a piece that takes ~30 ms is waiting for that network, i.e. synchronises more than its own stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from multiposenet.pytorch_amd.network.posenet import poseNet
from multiposenet.pytorch_amd.network.joint_utils import NMS_batch_arrays, body_peaks_flat
from multiposenet.pytorch_amd.evaluate.prn_process import prn_assign_arrays
from multiposenet.pytorch_amd import synthetic as weightgen
import bench
torch.cuda.set_device(0)
m = poseNet(101, compute_dtype=torch.float16).cuda(); bench.he_weights(m)
sd = weightgen.gen_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items() if k.startswith("prn.")}, seed=3, flavour="he")
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
m.eval()
img = torch.from_numpy(weightgen.gen_images(41, 64, 640, 640)).cuda()
with torch.no_grad():
    _, (cls, _, _) = m([img[:4].contiguous(), "detection_subnet"])
    s = cls.float().flatten().clamp(1e-6, 1 - 1e-6)
    q = torch.quantile(s[torch.randperm(s.numel(), device=s.device)[:1000000]], 1.0 - 1000.0 / float(cls.shape[1]))
    m.classificationModel.output.bias.data += float(np.log(0.05 / 0.95) - torch.log(q / (1 - q)))
    h0, _ = m.forward_all_images(img[:4].contiguous())
hv = h0.float().flatten()
thre1 = float(torch.quantile(hv[torch.randperm(hv.numel(), device=hv.device)[:2000000]], 1.0 - 4 * 12.0 / (h0.shape[2] * h0.shape[3])))
post = torch.cuda.Stream()
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
main_s = torch.cuda.Stream() if mode == "mainstream" else None


def begin():
    with torch.no_grad():
        if main_s is not None:
            with torch.cuda.stream(main_s):
                item = m.forward_padded_begin(img)
                ev = torch.cuda.Event(); ev.record()
        else:
            item = m.forward_padded_begin(img)
            ev = torch.cuda.Event(); ev.record()
    return item, ev


def finish(item, ev, log):
    heat, anchors, cls, _k = item
    post.wait_event(ev)
    with torch.cuda.stream(post):
        t = [time.perf_counter()]
        boxes, scores, kept = m.detect_padded(anchors, cls); t.append(time.perf_counter())
        pk, cnt = NMS_batch_arrays({'thre1': thre1}, heat, 4.0); t.append(time.perf_counter())
        peaks_xy, joint_off = body_peaks_flat(pk, cnt, keep=4); t.append(time.perf_counter())
        nb = np.minimum(np.asarray(kept), 4)
        sel = np.arange(boxes.shape[1])[None, :] < nb[:, None]
        b4 = boxes[:, :4].double().cpu().numpy()[sel[:, :4]]; t.append(time.perf_counter())
        b4[:, 2:] -= b4[:, :2]
        ok = (b4[:, 2] >= 1) & (b4[:, 3] >= 1)
        start = np.concatenate([[0], np.cumsum(np.add.reduceat(ok, np.concatenate([[0], np.cumsum(nb)[:-1]])))]).astype(np.int32)
        kp = prn_assign_arrays(m, peaks_xy, joint_off, b4[ok], start); t.append(time.perf_counter())
    log.append([round((b - a) * 1e3, 2) for a, b in zip(t[:-1], t[1:])])
    return kp


for warm in range(2):
    it, ev = begin(); finish(it, ev, [])
torch.cuda.synchronize()
log, enq = [], []
pending = None
T0 = time.perf_counter()
N = 8
for k in range(N + 1):
    nxt = None
    if k < N:
        t0 = time.perf_counter(); nxt = begin(); enq.append(round((time.perf_counter() - t0) * 1e3, 2))
    if pending is not None:
        finish(pending[0], pending[1], log)
    pending = nxt
torch.cuda.synchronize()
print("mode %s: %.2f ms per batch; enqueue ms %s" % (mode, (time.perf_counter() - T0) / N * 1e3, enq))
print("finish pieces [detect, peaks, flatten, boxes.cpu, prn_assign] ms per batch:")
for r in log:
    print("   ", r)
