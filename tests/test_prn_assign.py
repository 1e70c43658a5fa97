"""PRN person assignment (second half; evaluate/tester.py:333-513).

CPU: the oracle restatement against outputs of the REAL ``Tester.prn_process`` (g13_prn_process.npz, made by
tests/golden/make_golden_prn_process.py) and its gaussian against real skimage (g12_prn_gaussian.npz).
GPU: the HIP path (maps + blur + batched PRN + window scores + arg-max, host-side greedy matching) against the same real
reference outputs and, stage by stage, against the oracle."""
import numpy as np
import pytest
import torch

from helpers import gold


def _prn_weights(model_or_shapes):
    from multiposenet.pytorch_amd import synthetic as weightgen
    shapes = model_or_shapes if isinstance(model_or_shapes, dict) else {k: tuple(v.shape) for k, v in model_or_shapes.state_dict().items() if k.startswith("prn.")}
    return weightgen.gen_state_dict(shapes, seed=3, flavour="he", skip_prefixes=())


def _cases():
    g = gold("g13_prn_process.npz")
    for ci in range(int(g["ncases"])):
        yield ci, g["kps_%d" % ci].tolist(), g["boxes_%d" % ci].tolist(), g


def test_gaussian_restatement_against_real_skimage():
    from oracle import prn_assign_oracle
    g = gold("g12_prn_gaussian.npz")
    for m, ref in zip(g["maps"].astype(np.float64), g["blurred"]):
        got = prn_assign_oracle.gaussian(m)
        assert np.abs(got - ref).max() <= 1e-15                 # numpy's exp differs by an ulp between the two interpreters
        assert np.array_equal(got.astype(np.float32), ref.astype(np.float32))      # identical once the PRN's float32 input is formed


def test_oracle_against_the_real_prn_process():
    from oracle import posenet_oracle as po, prn_assign_oracle
    n = 56 * 36 * 17
    shapes = {"prn.dens1.weight": (1024, n), "prn.dens1.bias": (1024,), "prn.bneck.weight": (1024, 1024), "prn.bneck.bias": (1024,),
              "prn.dens2.weight": (n, 1024), "prn.dens2.bias": (n,)}
    sd = {k: torch.from_numpy(v) for k, v in _prn_weights(shapes).items()}

    def fwd(x):
        with torch.no_grad():
            return po.prn_forward(sd, torch.from_numpy(x)).numpy()
    for ci, kps, boxes, g in _cases():
        res = prn_assign_oracle.prn_process(fwd, kps, boxes, "img%d.jpg" % ci, ci)
        assert len(res) == int(g["n_%d" % ci])
        for r, kp, sc, bb in zip(res, g["keypoints_%d" % ci], g["score_%d" % ci], g["bbox_%d" % ci]):
            assert np.array_equal(np.array(r["keypoints"]), kp) and r["score"] == sc and np.array_equal(np.array(r["bbox"]), bb)
            assert r["image_id"] == ci and r["file_name"] == "img%d.jpg" % ci and r["category_id"] == 1


@pytest.mark.gpu
def test_hip_prn_process_against_the_real_reference_and_the_oracle():
    from multiposenet.pytorch_amd.evaluate.prn_process import _W9, prn_process, prn_process_batch
    from multiposenet.pytorch_amd import ops
    from multiposenet.pytorch_amd._lib import call
    from multiposenet.pytorch_amd.network.posenet import poseNet
    from oracle import prn_assign_oracle
    assert torch.cuda.is_available()
    model = poseNet(50, compute_dtype=torch.float32).cuda()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in _prn_weights(model).items()}, strict=False)
    model.eval()
    all_kps, all_boxes = [], []
    for ci, kps, boxes, g in _cases():
        all_kps.append(kps); all_boxes.append(boxes)
        res = prn_process(model, kps, boxes, "img%d.jpg" % ci, ci)
        assert len(res) == int(g["n_%d" % ci]), ci
        for r, kp, sc, bb in zip(res, g["keypoints_%d" % ci], g["score_%d" % ci], g["bbox_%d" % ci]):
            assert np.array_equal(np.array(r["keypoints"]), kp), "case %d: keypoints differ from the real reference" % ci
            assert r["score"] == sc and np.array_equal(np.array(r["bbox"]), bb) and r["image_id"] == ci
        # stage by stage against the oracle: cells and blur bit-exact
        if boxes:
            peaks, bboxes, old, inp = prn_assign_oracle.build_maps(kps, boxes)
            nb = len(bboxes)
            flat, off = [], []
            for j in range(17):
                off.append(len(flat)); flat.extend([p[0], p[1]] for p in peaks[j])
            off.append(len(flat))
            dev = "cuda"
            occ = torch.empty((nb, 17, 56, 36), dtype=torch.int32, device=dev)
            pin = torch.empty((nb, 56, 36, 17), dtype=torch.float32, device=dev)
            err = torch.zeros(1, dtype=torch.int32, device=dev)
            # (inputs are named: a temporary handed to ops.ptr() would be freed, and its memory re-used, before the launch)
            d_peaks = torch.tensor(flat if flat else [[0.0, 0.0]], dtype=torch.float64, device=dev)
            d_off = torch.tensor([off], dtype=torch.int32, device=dev)
            d_boxes = torch.tensor(bboxes, dtype=torch.float64, device=dev)
            d_img = torch.zeros(nb, dtype=torch.int32, device=dev)
            d_w9 = torch.from_numpy(_W9).to(dev)
            call("mpn_prn_build_maps", ops.ptr(d_peaks), ops.ptr(d_off), ops.ptr(d_boxes), ops.ptr(d_img), nb, 56, 36, 0.21, ops.ptr(d_w9),
                 ops.ptr(occ), ops.ptr(pin), ops.ptr(err), ops.stream_ptr())
            want_occ = np.where(old[:, :, :, 0, :] == 1, old[:, :, :, 2, :] + 1, 0).transpose(0, 3, 1, 2)
            assert int(err.item()) == 0 and np.array_equal(occ.cpu().numpy(), want_occ.astype(np.int32))
            assert np.array_equal(pin.cpu().numpy(), inp.astype(np.float32)), "blurred PRN input differs from scipy's arithmetic"
            # window sums in np.sum's float32 pairwise order + first arg-max, on a random plane set
            rs = np.random.RandomState(ci)
            outp = rs.rand(nb, 56, 36, 17).astype(np.float32)
            score = torch.zeros((nb, 17, 56, 36), dtype=torch.float32, device=dev)
            amax = torch.empty((nb, 17), dtype=torch.int32, device=dev)
            d_out = torch.from_numpy(outp).to(dev)
            call("mpn_prn_scores", ops.ptr(d_out), ops.ptr(occ), nb, 56, 36, 15, ops.ptr(score), ops.ptr(amax), ops.stream_ptr())
            sc, am, oc = score.cpu().numpy(), amax.cpu().numpy(), occ.cpu().numpy()
            for b, t_, y, x in np.argwhere(oc > 0):
                cr = prn_assign_oracle.crop(outp[b, :, :, t_], (y, x), N=15)
                assert sc[b, t_, y, x] == np.sum(cr), "window sum is not np.sum's float32 result"
            assert np.array_equal(am, outp.reshape(nb, -1, 17).argmax(1).astype(np.int32))
    # every image in the same launches
    batch = prn_process_batch(model, all_kps, all_boxes, ["img%d.jpg" % i for i in range(len(all_kps))], list(range(len(all_kps))))
    for ci, kps, boxes, g in _cases():
        assert len(batch[ci]) == int(g["n_%d" % ci])
        for r, kp in zip(batch[ci], g["keypoints_%d" % ci]):
            assert np.array_equal(np.array(r["keypoints"]), kp)


@pytest.mark.gpu
def test_compact_candidates_and_cpp_matching_equal_the_full_table_numpy_path():
    """The fast path of prn_assign_arrays (candidates compacted on the device, greedy matching in C++) against the full-table
    path that evaluates the reference's own numpy expressions: the real-reference goldens through both, then crowded random
    scenes (16 images, up to 12 overlapping boxes, up to 40 peaks per joint type, joint types nobody has, duplicate peak
    positions that produce EXACT score ties -> those pairs must be detected and handed to the numpy path)."""
    from multiposenet.pytorch_amd.evaluate import prn_process as pp
    from multiposenet.pytorch_amd.network.posenet import poseNet
    model = poseNet(50, compute_dtype=torch.float32).cuda()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in _prn_weights(model).items()}, strict=False)
    model.eval()
    all_kps, all_boxes = [], []
    for ci, kps, boxes, g in _cases():
        all_kps.append(kps); all_boxes.append(boxes)
    for fast in (True, False):
        batch = pp.prn_process_batch(model, all_kps, all_boxes, fast=fast)
        for ci, kps, boxes, g in _cases():
            assert len(batch[ci]) == int(g["n_%d" % ci])
            for r, kp, sc in zip(batch[ci], g["keypoints_%d" % ci], g["score_%d" % ci]):
                assert np.array_equal(np.array(r["keypoints"]), kp) and r["score"] == sc, "fast=%s case %d" % (fast, ci)
    rs = np.random.RandomState(77)
    nimg = 16
    peaks, joint_off, boxes, box_start = [], [], [], [0]
    for i in range(nimg):
        nbx = rs.randint(0, 13)
        for _ in range(nbx):
            cx, cy, bw, bh = rs.uniform(100, 540), rs.uniform(100, 380), rs.uniform(40, 200), rs.uniform(80, 300)
            boxes.append([cx - bw / 2, cy - bh / 2, bw, bh])
        box_start.append(len(boxes))
        offs = []
        for t in range(17):
            offs.append(len(peaks))
            n = 0 if (t in (4, 11) and i % 3 == 0) else rs.randint(1, 41)
            pts = np.stack([rs.uniform(0, 640, n), rs.uniform(0, 480, n)], 1)
            if i % 4 == 1 and n > 4:
                pts[3] = pts[1]                     # a duplicated peak: same cell for every box -> later overwrites earlier
            peaks += pts.tolist()
        offs.append(len(peaks))
        joint_off.append(offs)
    args = (np.asarray(peaks, dtype=np.float64), np.asarray(joint_off, dtype=np.int32), np.asarray(boxes, dtype=np.float64).reshape(-1, 4),
            np.asarray(box_start, dtype=np.int32))
    fast = pp.prn_assign_arrays(model, *args, fast=True)
    full = pp.prn_assign_arrays(model, *args, fast=False)
    assert fast.shape == full.shape == (len(boxes), 17, 3)
    assert np.array_equal(fast, full), "fast path differs from the full-table numpy path in %d boxes" % int((fast != full).any(axis=(1, 2)).sum())
    # a tiny candidate capacity forces the overflow fallback everywhere: still the same answer
    old = pp.CAND_CAP
    pp.CAND_CAP = 2
    try:
        assert np.array_equal(pp.prn_assign_arrays(model, *args, fast=True), full)
    finally:
        pp.CAND_CAP = old
