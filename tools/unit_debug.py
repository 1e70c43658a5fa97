import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from helpers import to_act, from_act, rng_normal
from multiposenet.pytorch_amd import ops
dt = torch.float32
def rl2(a, b): return ((a.double() - b.double()).norm() / b.double().norm()).item()
# dgrad cases at the model's small shapes
for (B, H, W, Cin, Cout, k, s, p) in [(2, 8, 8, 1024, 2048, 1, 2, 0), (2, 8, 8, 512, 512, 3, 2, 1), (2, 8, 8, 1024, 512, 1, 1, 0), (2, 8, 8, 1024, 256, 1, 1, 0), (2, 4, 4, 2048, 512, 1, 1, 0)]:
    x = rng_normal(1, B, Cin, H, W).requires_grad_(True)
    w = (rng_normal(2, Cout, Cin, k, k) / math.sqrt(Cin * k * k)).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=s, padding=p); dy = rng_normal(3, *y.shape); y.backward(dy)
    wm = w.detach().permute(0, 2, 3, 1).contiguous().cuda()
    opad = (Cout + 15) // 16 * 16
    wt = torch.empty((Cin, k, k, opad), dtype=dt, device="cuda"); ops.weight_transpose(wm, wt, Cout, k * k, Cin, opad)
    dya = to_act(dy, dt)
    dx, _ = ops.conv_forward(dya, wt, Cin, k, k, s, p, mode=1, out_hw=(H, W), cin=opad)
    e1 = rl2(from_act(dx), x.grad)
    # accumulate on top of an existing buffer
    base = rng_normal(4, B, Cin, H, W); g = to_act(base, dt)
    ops.conv_forward(dya, wt, Cin, k, k, s, p, mode=1, out_hw=(H, W), cin=opad, out=g, accumulate=True)
    e2 = rl2(from_act(g), x.grad + base)
    dw = torch.zeros((Cout, k, k, Cin), device="cuda"); ops.conv_wgrad(to_act(x.detach(), dt), dya, dw, Cout, k, k, s, p)
    e3 = rl2(dw.cpu().permute(0, 3, 1, 2), w.grad)
    print("dgrad %s relL2 %.2e  accumulate %.2e  wgrad %.2e" % ((B, H, W, Cin, Cout, k, s, p), e1, e2, e3))
# BN train backward with tiny pixel counts
for (B, H, W, C) in [(2, 4, 4, 2048), (2, 4, 4, 512), (2, 8, 8, 1024), (2, 8, 8, 256), (2, 32, 32, 64)]:
    y = rng_normal(5, B, C, H, W) * 0.7 + 0.4; res = rng_normal(6, B, C, H, W)
    gamma = torch.rand(C) + 0.5; beta = rng_normal(7, C) * 0.1
    yl = y.clone().requires_grad_(True); g_ = gamma.clone().requires_grad_(True); b_ = beta.clone().requires_grad_(True); rl = res.clone().requires_grad_(True)
    z = F.relu(F.batch_norm(yl, torch.zeros(C), torch.ones(C), g_, b_, training=True) + rl)
    dz = rng_normal(8, B, C, H, W); z.backward(dz)
    ya = to_act(y, dt); yv = from_act(ya)
    stats = torch.stack([yv.sum((0, 2, 3)), (yv * yv).sum((0, 2, 3))], 1).unsqueeze(0).contiguous().cuda()
    st = ops.bn_finalize_train(stats, B * H * W, gamma.cuda(), beta.cuda(), torch.zeros(C).cuda(), torch.ones(C).cuda())
    za = ops.bn_act(ya, st, True, res=to_act(res, dt))
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda"); dres = ops.Act(torch.empty_like(ya.t), C)
    dy = ops.bn_backward(to_act(dz, dt), za, ya, st, gamma.cuda(), True, True, dgamma=dg, dbeta=db, dres=dres)
    print("bn %s fwd %.2e dy %.2e dgamma %.2e dbeta %.2e dres %.2e" % ((B, H, W, C), rl2(from_act(za), z.detach()), rl2(from_act(dy), yl.grad), rl2(dg.cpu(), g_.grad), rl2(db.cpu(), b_.grad), rl2(from_act(dres), rl.grad)))
