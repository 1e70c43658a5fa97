import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import posenet_oracle as po
from multiposenet.pytorch_amd import synthetic as weightgen
from multiposenet.pytorch_amd.network.posenet import poseNet
t = lambda x: torch.from_numpy(np.ascontiguousarray(x))
b, s = 2, 128
img = t(weightgen.gen_images(2, b, s, s)); heat, wgt = weightgen.gen_keypoint_gt(2, b, s // 4, s // 4)
m = poseNet(50, compute_dtype=torch.float32).cuda()
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
sd = weightgen.gen_state_dict(shapes, 0, "he", skip_prefixes=("prn.",))
def run():
    m.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False); m.train(); m.zero_grad()
    pred, saved = m([img.cuda(), "keypoint_subnet"])
    loss, _ = poseNet.build_loss(saved, "keypoint_subnet", t(heat).cuda(), t(wgt).cuda())
    loss.backward(); torch.cuda.synchronize()
    return {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}, loss.item()
g1, l1 = run(); g2, l2 = run()
print("deterministic:", all(torch.equal(g1[k], g2[k]) for k in g1), l1, l2)
osd = {k: t(v).clone() for k, v in sd.items() if v.dtype != np.int64}
for k, v in osd.items():
    if not k.endswith(("running_mean", "running_var")): v.requires_grad_(True)
pred, saved = po.posenet_forward(osd, img, "keypoint_subnet", 50, True)
lo, _ = po.keypoint_loss(saved, t(heat), t(wgt)); lo.backward()
rows = []
for k in g1:
    if k not in osd: continue
    r = osd[k].grad
    if r is None: continue
    e = ((g1[k].double() - r.double()).norm() / max(r.double().norm().item(), 1e-12)).item()
    rows.append((e, k))
rows.sort(reverse=True)
print("loss", l1, lo.item())
order = {k: i for i, k in enumerate(g1)}
rows2 = sorted(rows, key=lambda r: order[r[1]])
for e, k in rows2:
    if k.endswith("weight") and ("conv" in k or "layer" in k.split(".")[1] or "smooth" in k or "flat" in k or "top" in k) and "bn" not in k: print("%.3e %s" % (e, k))
print("median %.3e" % rows[len(rows)//2][0])
