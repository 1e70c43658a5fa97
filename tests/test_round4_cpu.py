"""Round-4 CPU tests: the gradient reducer's schedule for the headline R101 step, and the code-object resource gate (no scratch,
expected occupancy class) for the hot kernels of the built library."""
import os
import re
import subprocess

import pytest
import torch
import torch.distributed as dist

from helpers import ROOT


# ------------------------------------------------------------------------------------------------ reducer schedule (R101)
def _forward_order(m):
    """Parameters in the order the headline step's forward uses them (engine.backbone -> kp_pyramid -> keypoint_head -> det_pyramid ->
    detection_head; the reference: fpn.py:97-126, posenet.py:288-335).  Backward completes them in the reverse order."""
    f = m.fpn
    mods = [f.conv1, f.bn1]
    for layer in (f.layer1, f.layer2, f.layer3, f.layer4):
        for blk in layer:
            mods += [blk.conv1, blk.bn1, blk.conv2, blk.bn2]
            if len(blk.downsample) > 0:
                mods += [blk.downsample[0], blk.downsample[1]]
            mods += [blk.conv3, blk.bn3]
    mods += [f.toplayer, f.flatlayer1, f.flatlayer2, f.flatlayer3, f.smooth1, f.smooth2, f.smooth3]
    mods += [m.convfin_k2, m.convfin_k3, m.convfin_k4, m.convfin_k5, m.convt1, m.convs1, m.convt2, m.convs2, m.convt3, m.convs3, m.convt4, m.convs4,
             m.conv2, m.convfin]
    mods += [f.conv6, f.conv7, f.latlayer1, f.latlayer2, f.latlayer3, f.toplayer0, f.toplayer1, f.toplayer2]
    rm, cm = m.regressionModel, m.classificationModel
    mods += [rm.conv1, rm.conv2, rm.conv3, rm.conv4, rm.output, cm.conv1, cm.conv2, cm.conv3, cm.conv4, cm.output]
    out = []
    for mod in mods:
        out += [p for p in mod.parameters(recurse=False)]
    return out


def test_grad_reducer_schedule_for_r101_covers_every_trainable_element_once_in_reverse_forward_order(tmp_path):
    """datasets/data_parallel.py:16-87 reduce-adds every gradient after backward; here buckets are arena slices launched while
    backward runs.  For the headline model (R101, every non-PRN parameter trainable): (1) the buckets tile the trainable part of
    the gradient arena exactly — every trainable element in exactly one bucket, no frozen (PRN) element in any; (2) fed the
    parameters in the order the backward pass completes them, EVERY bucket launches from ``param_ready`` (``finish`` has nothing
    left to flush), each at the moment its earliest-in-forward parameter completes, so the launch order is the reverse forward
    order; (3) the collectives are spread over backward instead of bunched at its end: when backward reaches layer1 more than
    85 % of the gradient bytes are already in flight — only the last bucket (stem .. layer2, the first 32 MB of the arena) has to
    wait for the end of backward."""
    from multiposenet.pytorch_amd import ddp
    from multiposenet.pytorch_amd.network.posenet import poseNet
    store = dist.FileStore(str(tmp_path / "store"), 1)
    dist.init_process_group("gloo", store=store, rank=0, world_size=1)
    try:
        m = poseNet(101, device="cpu") if "device" in poseNet.__init__.__code__.co_varnames else poseNet(101)
        for p in m.prn.parameters():
            p.requires_grad = False
        red = ddp.attach(m, bucket_mb=32.0, broadcast=False)
        ar = m._arena
        # (1) coverage
        cover = torch.zeros(ar.total, dtype=torch.int16)
        for b in red.buckets:
            cover[b["start"]: b["end"]] += 1
        want = torch.zeros(ar.total, dtype=torch.int16)
        n_train = 0
        for i, p in enumerate(ar.params):
            if p.requires_grad:
                want[ar.offsets[i]: ar.offsets[i] + (ar.sizes[i] + 63) // 64 * 64] = 1
                n_train += ar.sizes[i]
        assert torch.equal(cover, want), "buckets do not tile the trainable gradient ranges exactly once"
        assert abs(n_train - 61.05e6) < 0.05e6, "R101 trainable elements: %d" % n_train
        seen = sorted(i for b in red.buckets for i in b["params"])
        assert seen == [i for i, p in enumerate(ar.params) if p.requires_grad]
        # (2) readiness-driven launch order
        fwd = _forward_order(m)
        assert sorted(ar.index[id(p)] for p in fwd) == seen, "the forward order must list every trainable parameter once"
        pos = {ar.index[id(p)]: k for k, p in enumerate(fwd)}
        launched = []
        red._launch = lambda bk: (launched.append(red.buckets.index(bk)), bk.__setitem__("pending", -1), setattr(red, "launched", red.launched + 1))
        red.begin()
        bytes_at = {}
        for p in reversed(fwd):
            red.param_ready(p)
            bytes_at[ar.index[id(p)]] = sum(red.buckets[b]["end"] - red.buckets[b]["start"] for b in launched)
        assert sorted(launched) == list(range(len(red.buckets))), "a bucket never became ready from param_ready: finish() would flush it at the end"
        first = [min(pos[i] for i in red.buckets[b]["params"]) for b in launched]
        assert first == sorted(first, reverse=True), "buckets must launch in reverse forward order"
        # (3) overlap: bytes in flight when backward reaches layer1 / the stem
        total = sum(b["end"] - b["start"] for b in red.buckets)
        at_layer1 = bytes_at[ar.index[id(m.fpn.layer1[-1].bn3.bias)]] / float(total)
        at_stem = bytes_at[ar.index[id(m.fpn.bn1.bias)]] / float(total)
        assert at_stem > 0.85 and at_layer1 > 0.85, (at_layer1, at_stem)
        assert len(red.buckets) >= 7
        # round 6: the bucket that completes LAST (the first of the arena) is the tail no backward work hides: a quarter-size bucket
        tail = red.buckets[launched[-1]]
        assert launched[-1] == 0 and (tail["end"] - tail["start"]) * 4 <= 8.5 * 1024 * 1024 + 4 * max(ar.sizes[i] for i in tail["params"]), tail["end"] - tail["start"]
        assert 1.0 - (tail["end"] - tail["start"]) / float(total) > 0.96
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ code-object resource gate
DTYPES = ("t", "DF16_", "f")          # bf16 storage, _Float16, float: the three arithmetic types BASELINE's configs run on (cfg3 / cfg5 / cfg2)


def _hot_kernels():
    """(mangled-name fragment, minimum waves per SIMD the schedule was tuned for, LDS bytes or None) — DESIGN.md section 3.
    Round 5: every entry for every element type it is instantiated for (round 4 looked at bf16 only, and two BASELINE configs ran on
    kernels the gate did not see)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from codeobj import mangled
    hot = []
    for dt in DTYPES:
        for gen in (False, True):
            hot.append((mangled("conv_igemm_kernel", dt, 128, 128, False, gen, False, False), 3, 49152))     # 3 workgroups / CU (LDS: 3 x 48 KB)
            if dt != "f":                                             # pick_tc() never takes the 256-row tile for f32: not instantiated
                hot.append((mangled("conv_igemm_kernel", dt, 256, 128, False, gen, False, False), 2, 73728))     # 2 workgroups / CU
            hot.append((mangled("conv_igemm_kernel", dt, 64, 128, False, gen, False, False), 3, 36864))
        # the parity-class launches of the stride-2 input gradients (seven per step) take conv_igemm_kernel's extended instantiation
        hot.append((mangled("conv_igemm_kernel", dt, 128, 128, False, True, True, False), 3, 49152))
        if dt == "f":
            continue                                                  # the shared-tile 3x3 kernel and the f32-output heads are 16-bit paths (f32 has its own
                                                                      # LDS-DMA weight-gradient twins, below)
        hot.append((mangled("conv_igemm_kernel", dt, 256, 128, False, True, True, False), 2, 73728))
        for gen in (False, True):
            hot.append((mangled("conv_igemm_s3_kernel", dt, 128, 128, False, gen, False, False), 3, 49152))
            hot.append((mangled("conv_igemm_s3_kernel", dt, 256, 128, False, gen, False, False), 2, 73728))
            hot.append((mangled("conv_igemm_s3_kernel", dt, 64, 128, False, gen, False, False), 3, 36864))
        # conv2 through the virtual concatenation (extended epilogue; the largest launch of cfg3 and cfg5)
        hot.append((mangled("conv_igemm_s3_kernel", dt, 256, 128, False, True, True, False), 2, 73728))
        # round 6: the class convolutions of conv2 (csrc/conv2cls.hip): 128 -> 9 x 256 channels at 1/8 and 1/4 resolution, f32 class maps
        hot.append((mangled("conv_igemm_s3_kernel", dt, 256, 128, True, False, False, False), 2, 73728))
        # f32 heads of the 16-bit network (convfin*, the detection outputs)
        hot.append((mangled("conv_igemm_kernel", dt, 128, 128, True, True, False, False), 3, 49152))
        hot.append((mangled("conv_igemm_s3_kernel", dt, 128, 128, True, True, False, False), 3, 49152))
    for tm, tn, lds in ((128, 128, 49152), (128, 64, 36864), (64, 128, 36864), (64, 64, 24576)):
        for suffix in ("", "_f16") + (("_f32",) if tm <= 128 else ()):      # round 5: the f32 twins (exact-fp32 MFMA on the same ring)
            hot.append((mangled("conv_wgrad_dma%s_kernel" % suffix, tm, tn), 3, lds))
            hot.append((mangled("conv_wgrad_dma_lin%s_kernel" % suffix, tm, tn), 3, lds))      # the instantiation nearly every launch of the step takes
    hot.append((mangled("conv_wgrad_dma_seg_kernel", 128, 128), 3, 49152))
    hot.append((mangled("conv_wgrad_dma_seg_f16_kernel", 128, 128), 3, 49152))
    for dt in DTYPES:
        for k, w in (("bn_act_kernel", 8), ("bn_bwd_apply_kernel", 5), ("relu_bwd_kernel", 8), ("maxpool_fwd_kernel", 4)):
            hot.append((mangled(k, dt), w, None))
    return hot


# Kernels allowed to keep spilled registers, with the reason; each must still have NO scratch instruction between its first and last
# MFMA (tools/codeobj.py: scratch_vs_mfma) — a spill parked and fetched outside the k-loop costs a few hundred cycles per workgroup,
# one inside multiplies by the k-steps (round 5 found the 256-row extended shared-tile kernel reloading gather state — behind a
# vmcnt(0) that drained its DMA ring — once per tap group; fixed by keeping one fragment offset per tap shift instead of one per fragment).
def _tolerated():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from codeobj import mangled
    tol = {}
    # the general f32 epilogue (cfg2's short-K 1x1 / lateral launches): 10 VGPRs parked across its grouped epilogue loads
    tol[mangled("conv_igemm_kernel", "f", 128, 128, False, True, False, False)] = 12
    for dt in ("t", "DF16_"):
        # f32 heads of the 16-bit network: 8 VGPRs, epilogue only
        tol[mangled("conv_igemm_kernel", dt, 128, 128, True, True, False, False)] = 12
        tol[mangled("conv_igemm_s3_kernel", dt, 128, 128, True, True, False, False)] = 12
        # conv2 through the virtual concatenation: 2 VGPRs + scalar spills in prologue / epilogue
        tol[mangled("conv_igemm_s3_kernel", dt, 256, 128, False, True, True, False)] = 4
        # scalar spills only (to VGPR lanes); the 128-row one reserves 20 bytes of scratch it never touches with a scratch instruction
        tol[mangled("conv_igemm_s3_kernel", dt, 128, 128, False, True, True, False)] = 0
        tol[mangled("conv_igemm_s3_kernel", dt, 64, 128, False, True, True, False)] = 0
        tol[mangled("conv_igemm_s3_kernel", dt, 256, 128, False, True, False, False)] = 0
        tol[mangled("conv_igemm_s3_kernel", dt, 256, 128, True, True, False, False)] = 0
    return tol


def test_hot_kernels_have_no_scratch_and_keep_their_occupancy_class():
    """Round 3 lost 3.3 ms/step to 62 - 112 spilled VGPRs in the 128-row conv tiles before anyone noticed (three epilogue features
    had been compiled into one kernel).  This gate reads the code objects of the BUILT library (tools/codeobj.py: the
    NT_AMDGPU_METADATA note of every gfx950 ELF in .hip_fatbin) and fails when
      * ANY kernel of the library has a scratch segment or a spilled register and is not on the tolerated list, or exceeds its
        allowance there, or touches scratch between its first and last MFMA;
      * a kernel the training / inference steps launch by default — in bf16, f16 or f32 — has fewer register-limited waves per SIMD than
        its schedule was tuned for, or another LDS footprint (= another number of workgroups per CU)."""
    import sys
    from multiposenet.pytorch_amd import _lib
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import codeobj
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    ks = codeobj.kernels(_lib.LIB_PATH)
    assert len(ks) > 100
    bad = []
    tol = _tolerated()
    for k in ks:
        if not (k["scratch"] or k["spill_v"] or k["spill_s"]):
            continue
        frag = [f for f in tol if f in k["name"]]
        if not frag:
            bad.append("%s: scratch %d spill v%d s%d and not on the tolerated list" % (k["name"], k["scratch"], k["spill_v"], k["spill_s"]))
            continue
        if k["spill_v"] > tol[frag[0]]:
            bad.append("%s: %d spilled VGPRs, allowance %d" % (k["name"], k["spill_v"], tol[frag[0]]))
        first, last, sc = codeobj.scratch_vs_mfma(_lib.LIB_PATH, k["name"])
        inside = [i for i in sc if first <= i <= last]
        if inside:
            bad.append("%s: %d scratch instructions inside the MFMA loop (instructions %d..%d): %s" % (k["name"], len(inside), first, last, inside[:8]))
    for frag in tol:
        assert any(frag in k["name"] for k in ks), "tolerated entry %s matches no kernel (renamed? update the list)" % frag
    for entry in _hot_kernels():
        frag, min_waves, lds = entry[:3]
        hits = [k for k in ks if frag in k["name"]]
        assert hits, "no kernel matches %s in %s (renamed? update the gate)" % (frag, _lib.LIB_PATH)
        for k in hits:
            waves = codeobj.waves_per_simd(k["vgpr"] + k["agpr"])
            if waves < min_waves or (lds is not None and k["lds"] != lds):
                bad.append("%s: vgpr %d waves/SIMD %d (want >= %d) lds %d (want %s)" % (k["name"], k["vgpr"], waves, min_waves, k["lds"], lds))
    assert not bad, "kernels lost their register / LDS budget:\n" + "\n".join(bad)
    # nothing in the library may use a stack (dynamic or fixed scratch beyond spills would mean recursion / big local arrays)
    stacky = [k["name"] for k in ks if k["scratch"] > 1024]
    assert not stacky, "kernels with > 1 KB of scratch per lane: %s" % stacky


def test_production_library_carries_no_experiment_code_and_few_knobs():
    """the shipped library must not carry what lost.  (i) no PROF instantiation (s_memtime phase profiles of the
    three DMA kernels: experiments build only, csrc/Makefile `experiments`), no atomic-statistics kernel; (ii) the C sources read the
    environment only through mpn_tune(), which is a constant in the production build; (iii) the package + bench.py read at most 25
    MPN_* environment variables (39 at the end of round 4)."""
    import re
    import sys
    from multiposenet.pytorch_amd import _lib
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import codeobj
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    names = [k["name"] for k in codeobj.kernels(_lib.LIB_PATH)]
    prof = [n for n in names if "prof_kernel" in n or re.search(r"conv_igemm(_s3)?_kernelI\w+Lb1EEEv13MpnConvParamsi$", n)]
    assert not prof, "PROF instantiations in the production library: %s" % prof
    assert not [n for n in names if "bn_act_acc" in n]
    csrc = os.path.join(ROOT, "multiposenet", "pytorch_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            txt = open(os.path.join(csrc, f)).read()
            n_env = len(re.findall(r"\bgetenv\s*\(", txt))
            assert n_env == (1 if f == "common.h" else 0), "%s calls getenv %d times (only mpn_tune in common.h may)" % (f, n_env)
    knobs = set()
    for dp, _, fs in os.walk(os.path.join(ROOT, "multiposenet")):
        for f in fs:
            if f.endswith(".py"):
                knobs.update(re.findall(r"environ(?:\.get\(|\[|\.setdefault\()\s*[\"'](MPN_[A-Z0-9_]+)", open(os.path.join(dp, f)).read()))
    knobs.update(re.findall(r"environ(?:\.get\(|\[|\.setdefault\()\s*[\"'](MPN_[A-Z0-9_]+)", open(os.path.join(ROOT, "bench.py")).read()))
    assert len(knobs) <= 25, "%d MPN_* environment knobs: %s" % (len(knobs), sorted(knobs))


def test_stride2_dgrad_class_plan_covers_every_pixel_and_every_live_tap_once():
    """ops.dgrad_s2_class_plan (host logic of the parity-class input gradient, mpn.h: y_step).  Against a brute-force statement of
    torch.nn.Conv2d(3, stride=2, padding=1) backward: dx[h][w] += dy[i][j] * W[r][s] for every (i, j, r, s) with 2 i - 1 + r = h and 2 j - 1 + s = w.
    The classes must tile dx exactly once, their taps must be exactly the live (dy pixel, filter tap) pairs of each pixel, and the table rows
    must be consecutive — for even, odd and degenerate extents."""
    from multiposenet.pytorch_amd.ops import dgrad_s2_class_plan
    for (B, H, W) in [(2, 16, 16), (1, 15, 15), (3, 9, 7), (1, 1, 1), (2, 1, 8), (1, 120, 30)]:
        Hd, Wd = (H - 1) // 2 + 1, (W - 1) // 2 + 1                      # dy extent of a stride-2 / pad-1 / 3x3 convolution
        want = {}
        for i in range(Hd):
            for j in range(Wd):
                for r in range(3):
                    for s in range(3):
                        h, w = 2 * i - 1 + r, 2 * j - 1 + s
                        if 0 <= h < H and 0 <= w < W:
                            want.setdefault((h, w), set()).add((i, j, r, s))
        got, next_row = {}, 0
        for a, c, ho, wo, tiles, tile0 in dgrad_s2_class_plan(B, H, W):
            assert tile0 == next_row and tiles == (B * ho * wo + 127) // 128
            next_row += tiles
            for i in range(ho):
                for j in range(wo):
                    h, w = 2 * i + a, 2 * j + c
                    assert h < H and w < W and (h, w) not in got, "class (%d,%d) pixel (%d,%d) out of range or owned twice" % (a, c, h, w)
                    taps = set()
                    for tr in range(1 + a):
                        for ts in range(1 + c):
                            r, s = a + 1 - 2 * tr, c + 1 - 2 * ts
                            assert 0 <= r < 3 and 0 <= s < 3 and (a + 1) * 3 + (c + 1) - 6 * tr - 2 * ts == r * 3 + s      # wtap0 / wtap_dr / wtap_ds
                            if i + tr < Hd and j + ts < Wd:                                                                # rows past dy read zeros
                                taps.add((i + tr, j + ts, r, s))
                    got[(h, w)] = taps
        assert set(got) == {(h, w) for h in range(H) for w in range(W)}, (B, H, W)
        for k in got:
            assert got[k] == want.get(k, set()), (B, H, W, k)
