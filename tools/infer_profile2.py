#!/usr/bin/env python3
"""Where the time of the cfg5 post-processing goes (stage timers with device syncs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multiposenet.pytorch_amd.evaluate import prn_process as pp
from multiposenet.pytorch_amd.network.joint_utils import NMS_batch_arrays, body_peaks_flat
from multiposenet.pytorch_amd.network.posenet import poseNet
from multiposenet.pytorch_amd import synthetic as weightgen
import bench
torch.cuda.set_device(0)
B, S = 64, 640
m = poseNet(101, compute_dtype=torch.float16).cuda()
bench.he_weights(m)
sd = weightgen.gen_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items() if k.startswith("prn.")}, seed=3, flavour="he")
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
m.eval()
img = torch.from_numpy(weightgen.gen_images(41, B, S, S)).cuda()
with torch.no_grad():
    _, (cls, _, _) = m([img[:4].contiguous(), "detection_subnet"])
    s = cls.float().flatten().clamp(1e-6, 1 - 1e-6)
    q = torch.quantile(s[torch.randperm(s.numel(), device=s.device)[:1000000]], 1.0 - 1000 / float(cls.shape[1]))
    m.classificationModel.output.bias.data += float(np.log(0.05 / 0.95) - torch.log(q / (1 - q)))
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return time.perf_counter()
orig_call = pp.call
def timed_call(name, *a):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = orig_call(name, *a); torch.cuda.synchronize()
    T["  call " + name] = T.get("  call " + name, 0.0) + time.perf_counter() - t0; return r
with torch.no_grad():
    h0, _ = m.forward_all_images(img[:4].contiguous())
hv = h0.float().flatten()
thre1 = float(torch.quantile(hv[torch.randperm(hv.numel(), device=hv.device)[:2000000]], 1.0 - 4 * 12.0 / (h0.shape[2] * h0.shape[3])))
print("calibrated thre1", thre1)
for it in range(4):
    if it == 1:
        T.clear(); pp.call = timed_call
    t0 = time.perf_counter()
    with torch.no_grad():
        heat, boxes, scores, kept = m.forward_all_images_padded(img)
    t0 = tick("network + NMS (padded)", t0)
    pk, cnt = NMS_batch_arrays({'thre1': thre1}, heat, 4.0)
    t0 = tick("peaks kernel + D2H", t0)
    peaks_xy, joint_off = body_peaks_flat(pk, cnt, keep=4)
    t0 = tick("body_peaks_flat", t0)
    nb = np.minimum(np.asarray(kept), 4)
    sel = np.arange(boxes.shape[1])[None, :] < nb[:, None]
    b4 = boxes[:, :4].double().cpu().numpy()[sel[:, :4]]
    b4[:, 2:] -= b4[:, :2]
    start = np.concatenate([[0], np.cumsum(nb)]).astype(np.int32)
    t0 = tick("box selection", t0)
    kp = pp.prn_assign_arrays(m, peaks_xy, joint_off, b4, start)
    t0 = tick("prn_assign_arrays", t0)
print("boxes", b4.shape[0], "peaks", peaks_xy.shape[0], "pk buffer", pk.shape, "max count", int(cnt.max()))
for k, v in T.items():
    print("%-45s %8.2f ms" % (k, v / 3 * 1e3))
