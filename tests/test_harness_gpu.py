"""Training / inference harness on the device: Trainer (training/trainer.py:105-380) and the Tester drivers
(evaluate/tester.py:105-331) over the HIP path."""
import os

import numpy as np
import pytest
import torch

from helpers import report
from test_model_gpu import get_model, load_he, t

pytestmark = pytest.mark.gpu


def test_resize_follows_the_opencv_rule():
    from multiposenet.pytorch_amd.evaluate.tester import resize
    from oracle import joint_oracle
    rs = np.random.RandomState(0)
    for (hs, ws, c, hd, wd) in ((30, 40, 18, 120, 160), (120, 160, 18, 97, 133), (33, 21, 3, 66, 42), (64, 48, 3, 23, 31), (7, 5, 1, 7, 5)):
        img = rs.rand(hs, ws, c).astype(np.float32)
        for cubic in (True, False):
            got = resize(torch.from_numpy(img).cuda(), (hd, wd), cubic).cpu().numpy()
            want = joint_oracle.cv_resize(img, (hd, wd), cubic)
            assert np.array_equal(got, want), "resize %s %dx%d -> %dx%d differs from the restated OpenCV arithmetic" % ("cubic" if cubic else "linear", hs, ws, hd, wd)
    # the fx / fy call form of crop_with_factor (cv2.resize(im, None, fx=s, fy=s), tester.py:68): the scale is 1/f exactly, which is
    # NOT src/dst when round(src * f) != src * f
    img = rs.rand(37, 53, 3).astype(np.float32)
    for f in (0.73, 1.37, 2.0):
        hd, wd = int(np.rint(37 * f)), int(np.rint(53 * f))
        got = resize(torch.from_numpy(img).cuda(), (hd, wd), False, inv_scale=(1.0 / f, 1.0 / f)).cpu().numpy()
        assert np.array_equal(got, joint_oracle.cv_resize(img, (hd, wd), False, inv_scale=(1.0 / f, 1.0 / f)))
        if f != 2.0:
            assert not np.array_equal(got, joint_oracle.cv_resize(img, (hd, wd), False))
    # strided views (a channel-first heat-map seen as [H, W, C]) take the same path
    chw = torch.from_numpy(rs.rand(18, 20, 24).astype(np.float32)).cuda()
    got = resize(chw.permute(1, 2, 0), (80, 96), True).cpu().numpy()
    assert np.array_equal(got, joint_oracle.cv_resize(chw.permute(1, 2, 0).cpu().numpy(), (80, 96), True))


class _State(object):
    pass


def _loader(n, B, S, seed):
    from multiposenet.pytorch_amd import synthetic as weightgen
    out = []
    for i in range(n):
        img = t(weightgen.gen_images(seed + i, B, S, S))
        heat, wgt = (t(a) for a in weightgen.gen_keypoint_gt(seed + 10 + i, B, S // 4, S // 4))
        out.append((img, heat, wgt))
    return out


def test_trainer_epochs_checkpoints_validation_and_resume(tmp_path):
    """Two epochs of three keypoint batches through Trainer: same parameters as six hand-written steps (trainer.py:245-259),
    ckpt_{epoch}.h5 + optimizer pickle written and pruned, the validation loss drives ReduceLROnPlateau and the .best copy,
    BN is back in train mode after validation; a second Trainer resumes from the newest checkpoint (epoch, Adam state)."""
    from multiposenet.pytorch_amd.optim import FusedAdam
    from multiposenet.pytorch_amd.training.batch_processor import batch_processor, train_step
    from multiposenet.pytorch_amd.training.trainer import Trainer, TrainParams
    B, S = 2, 64
    train_data, val_data = _loader(3, B, S, 200), _loader(2, B, S, 300)
    model = get_model(50, torch.bfloat16)
    for p in model.prn.parameters():
        p.requires_grad = False
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    # reference loop by hand
    model.train()
    opt = FusedAdam(model, lr=1e-3)
    st = _State(); st.model = model; st.params = _State(); st.params.subnet_name = 'keypoint_subnet'; st.params.gpus = [0]
    for epoch in range(2):
        for batch in train_data:
            inputs, gts, _ = batch_processor(st, batch)
            train_step(model, opt, inputs, gts)
    want = model._arena.flat.clone()
    # the same through Trainer
    model.load_state_dict(state0)
    params = TrainParams()
    params.exp_name, params.subnet_name, params.batch_size, params.max_epoch = 'unit', 'keypoint_subnet', B, 2
    params.save_dir = str(tmp_path / "run")
    params.optimizer = FusedAdam(model, lr=1e-3)
    # mode='max' on a falling loss: the first validation sets the best value, the second is "worse" -> the rate halves AFTER
    # the six steps (so the hand-written loop above, which never changes the rate, still applies)
    params.lr_scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(params.optimizer, mode='max', factor=0.5, patience=0, threshold=0.0)
    params.print_freq, params.val_nbatch_end_epoch, params.save_nckpt_max = 2, 2, 5
    tr = Trainer(model, params, batch_processor, train_data, val_data)
    tr.train()
    torch.cuda.synchronize()
    assert torch.equal(model._arena.flat, want), "Trainer's steps differ from the hand-written loop"
    files = sorted(os.listdir(params.save_dir))
    assert "ckpt_1.h5" in files and "ckpt_2.h5" in files and "ckpt_2.h5.optimizer_state.pk" in files and "ckpt_1.h5.optimizer_state.pk" not in files
    assert any(f.startswith("ckpt_1_") and f.endswith(".h5.best") for f in files)
    assert model.training and all(m.training for m in model._bns)           # keypoint subnet: BN stays in train mode after validation
    assert params.lr_scheduler.last_epoch == 2 and params.optimizer.param_groups[0]["lr"] in (1e-3, 5e-4)      # stepped after both validations
    assert tr.last_epoch == 2 and params.optimizer.step_count() == 6
    # resume
    model2 = get_model(50, torch.bfloat16)
    load_he(model2, seed=9)
    for p in model2.prn.parameters():
        p.requires_grad = False
    p2 = TrainParams()
    p2.exp_name, p2.subnet_name, p2.batch_size, p2.max_epoch, p2.save_dir = 'unit', 'keypoint_subnet', B, 3, params.save_dir
    p2.optimizer = FusedAdam(model2, lr=1e-3)
    p2.val_nbatch_end_epoch = 0
    tr2 = Trainer(model2, p2, batch_processor, train_data, None)
    assert tr2.last_epoch == 2 and p2.optimizer.step_count() == 6 and torch.equal(model2._arena.flat, want)
    assert p2.optimizer.param_groups[0]["lr"] == 1e-3           # the rate stored in ckpt_2 (saved before the plateau scheduler stepped, trainer.py:196-217)
    tr2.train()
    assert tr2.last_epoch == 3 and p2.optimizer.step_count() == 9 and "ckpt_3.h5" in os.listdir(params.save_dir)
    report("Trainer: 2 epochs == 6 hand-written steps bit for bit; checkpoints, .best copy, plateau scheduler, resume at epoch 2 -> 3")


def test_tester_single_scale_and_multiscale_flip(tmp_path):
    """Tester drivers end to end on a synthetic image with He-random weights: checkpoint load through HDF5, result dicts in the
    reference's format, deterministic, boxes/keypoints inside the (padded) image; the multi-scale driver's pieces compose as
    tester.py:264-331 prescribes (flip symmetry of _handle_heat, averaging weights)."""
    from multiposenet.pytorch_amd.evaluate.tester import SWAP_HEAT, Tester, TestParams
    from multiposenet.pytorch_amd.network import net_utils
    model = get_model(50, torch.float32)
    ck = str(tmp_path / "ckpt_7.h5")
    net_utils.save_net(ck, model, epoch=7)
    params = TestParams()
    params.ckpt, params.inp_size = ck, 128
    from multiposenet.pytorch_amd.network.posenet import poseNet
    fresh = poseNet(50, compute_dtype=torch.float32)
    tester = Tester(fresh, params)
    assert torch.equal(fresh._arena.flat.cpu(), model._arena.flat.cpu())
    rs = np.random.RandomState(1)
    img = rs.uniform(0, 255, (96, 128, 3)).astype(np.float32)
    r1 = tester.infer_image(img, "a.jpg", 5)
    r2 = tester.infer_image(img, "a.jpg", 5)
    assert r1 == r2
    for r in r1:
        assert set(r.keys()) == {"image_id", "file_name", "category_id", "bbox", "score", "keypoints"} and len(r["keypoints"]) == 51
        assert r["image_id"] == 5 and r["file_name"] == "a.jpg" and 0.0 <= r["score"] <= 1.0
    mult = tester._get_multiplier(torch.zeros(96, 128, 3))
    assert mult == [x * 128 / 96.0 for x in [0.5, 1., 1.5, 2, 2.5]]
    dimg = torch.from_numpy(img).cuda()
    heat, boxes = tester._get_outputs(mult[:2], dimg)
    assert heat.shape == (96, 128, 18) and len(boxes) == 2 and torch.isfinite(heat).all()
    h0, _ = tester._get_outputs(mult[:1], dimg)
    h1, _ = tester._get_outputs(mult[1:2], dimg)
    assert torch.allclose(heat, (h0 + h1) / 2, atol=1e-6)
    flipped = tester._handle_heat(heat, heat.flip(1)[:, :, SWAP_HEAT])
    assert torch.allclose(flipped, heat, atol=1e-7)           # averaging a map with its own flipped+swapped image is the identity
    rm = tester.infer_image_multiscale(img, "a.jpg", 5)
    assert rm == tester.infer_image_multiscale(img, "a.jpg", 5)
    for r in rm:
        assert len(r["keypoints"]) == 51
    # result files: coco_eval (tester.py:176-178, indent 4, COCO keypoint order) and test() (:243-245)
    import json
    params.coco_result_filename = str(tmp_path / "coco_results.json")
    img2 = rs.uniform(0, 255, (80, 64, 3)).astype(np.float32)
    got = tester.coco_eval([(5, "a.jpg", img), (9, "b.jpg", img2)])
    assert got == rm + tester.infer_image_multiscale(img2, "b.jpg", 9)
    with open(params.coco_result_filename) as f:
        text = f.read()
    assert json.loads(text) == got and (not got or text.startswith("[\n    {"))
    params.testresult_write_json, params.testresult_dir = True, str(tmp_path) + "/"
    single = tester.test({"a.jpg": img})
    with open(str(tmp_path / "multipose_results.json")) as f:
        assert json.load(f) == single == tester.infer_image(img, "a.jpg")
    report("Tester: single-scale %d people, multi-scale+flip %d people on a random image (He weights); coco_eval / test result files" % (len(r1), len(rm)))
