"""Tester — the reference's inference drivers (``evaluate/tester.py``) over the MI355X path.

    tester = Tester(model, TestParams())                       # loads params.ckpt (HDF5), eval mode, frozen BN (:105-129)
    results = tester.infer_image(img, 'name.jpg')              # body of Tester.test() for one image (:200-239)
    results = tester.infer_image_multiscale(img, 'name.jpg', image_id)    # body of Tester.coco_eval() for one image (:143-175):
                                                                # 5 scales x {original, flipped} test-time augmentation

    results = tester.coco_eval(images)                         # :131-191 from decoded images on: results file in COCO order
    mean, std = tester.val()                                   # :515-543 validation-loss loop

``img`` is the decoded image the reference gets from ``cv2.imread(...).astype(np.float32)``: [H, W, 3] BGR, 0..255.  Decoding
and the COCO annotation tools (cv2.imread, pycocotools: tester.py:133-139,179-186) stay with the caller — neither library
exists in this image; the result files (:176-178, :243-245) are written here.  Everything between the decoded image and the result dicts runs on the device: OpenCV-rule resizes
(csrc/peaks.hip: mpn_resize), the network, heat-map averaging, peak extraction (network/joint_utils.py) and the PRN
assignment (prn_process.py).  ``cv2.resize`` is restated, not linked: parity with OpenCV is unpinned (no cv2 here), see
DESIGN.md.
"""
import json
import logging
import os

import numpy as np
import torch

from .. import ops
from .._lib import call
from ..network import net_utils
from ..network.joint_utils import get_joint_list
from .prn_process import prn_process

logger = logging.getLogger("multiposenet")

# COCO order of the 17 keypoints (tester.py:140) and the left/right swap of the 18 heat-map channels (tester.py:326-327)
COCO_ORDER = [0, 14, 13, 16, 15, 4, 1, 5, 2, 6, 3, 10, 7, 11, 8, 12, 9]
SWAP_HEAT = [0, 1, 5, 6, 7, 2, 3, 4, 11, 12, 13, 8, 9, 10, 15, 14, 17, 16]


class TestParams(object):
    """tester.py:85-103."""
    trunk = 'resnet101'
    coeff = 2
    in_thres = 0.21
    testdata_dir = './demo/test_images/'
    testresult_dir = './demo/output/'
    testresult_write_image = False
    testresult_write_json = False
    gpus = [0]
    ckpt = './demo/models/ckpt_baseline_resnet101.h5'
    coco_root = 'coco_root/'
    coco_result_filename = './extra/multipose_coco2017_results.json'
    inp_size = 480
    exp_name = 'multipose101'
    subnet_name = 'keypoint_subnet'
    batch_size = 32
    print_freq = 20


def resize(img, out_hw, cubic, inv_scale=None):
    """cv2.resize(img, (out_w, out_h), interpolation=INTER_CUBIC if cubic else INTER_LINEAR) for a CUDA float32 [H, W, C] tensor.
    ``inv_scale=(1/fy, 1/fx)`` selects OpenCV's ``fx/fy`` call form (``cv2.resize(img, None, fx=, fy=)``), which maps destination
    to source coordinates with exactly 1/f instead of src/dst (they differ whenever round(src * f) != src * f)."""
    H, W, C = img.shape
    Hd, Wd = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((Hd, Wd, C), dtype=torch.float32, device=img.device)
    sy, sx = (0.0, 0.0) if inv_scale is None else (float(inv_scale[0]), float(inv_scale[1]))
    call("mpn_resize", ops.ptr(img), img.stride(0), img.stride(1), img.stride(2), H, W, C, ops.ptr(out), Hd, Wd, 1 if cubic else 0, sy, sx, ops.stream_ptr())
    return out


def _round_half_even(v):
    return int(np.rint(v))          # cv::saturate_cast<int> of a double = cvRound


def crop_with_factor(im, dest_size, factor=32, pad_val=0, basedon='min'):
    """tester.py:38-82 for a CUDA [H, W, C] tensor: scale so that the chosen side becomes dest_size (cv2.resize with fx = fy =
    scale, bilinear), pad bottom/right with pad_val to a multiple of `factor`.  Returns (padded, scale, unpadded shape)."""
    h0, w0 = im.shape[0], im.shape[1]
    base = {'min': min(h0, w0), 'max': max(h0, w0), 'w': w0, 'h': h0}.get(basedon, min(h0, w0))
    im_scale = float(dest_size) / base
    h, w = _round_half_even(h0 * im_scale), _round_half_even(w0 * im_scale)       # dsize = round(src * f)
    scaled = resize(im, (h, w), cubic=False, inv_scale=(1.0 / im_scale, 1.0 / im_scale))       # cv2.resize(im, None, fx=s, fy=s), tester.py:68
    new_h, new_w = int(np.ceil(float(h) / factor)) * factor, int(np.ceil(float(w) / factor)) * factor
    padded = torch.full((new_h, new_w, im.shape[2]), float(pad_val), dtype=torch.float32, device=im.device)
    padded[:h, :w] = scaled
    return padded, im_scale, (h, w, im.shape[2])


def resnet_preprocess(image):
    """datasets/coco_data/preprocessing.py:14-25 on the device: BGR 0..255 [H,W,3] -> RGB, /255, ImageNet mean/std, [3,H,W]."""
    img = image.float() / 255.
    img = img.flip(2)
    mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float32, device=img.device)
    std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float32, device=img.device)
    return ((img - mean) / std).permute(2, 0, 1).contiguous()


def _to_device_image(img, dev):
    if isinstance(img, np.ndarray):
        img = torch.from_numpy(np.ascontiguousarray(img))
    return img.to(dev).float()


class Tester(object):
    TestParams = TestParams

    def __init__(self, model, train_params, batch_processor=None, val_data=None):
        assert isinstance(train_params, TestParams)
        self.params = train_params
        self.val_data = val_data
        self.batch_processor = batch_processor
        self.model = model
        if self.params.ckpt is not None:
            self._load_ckpt(self.params.ckpt)
            logger.info('Load ckpt from {}'.format(self.params.ckpt))
        self.dev = torch.device('cuda', self.params.gpus[0])
        torch.cuda.set_device(self.dev)
        self.model = self.model.to(self.dev)
        self.model.eval()
        self.model.freeze_bn()

    def _load_ckpt(self, ckpt):
        _, _ = net_utils.load_net(ckpt, self.model, load_state_dict=True)

    # ------------------------------------------------------------------ validation loss (Tester.val, tester.py:515-543)
    def val(self):
        """Loss of ``params.subnet_name`` over ALL of ``val_data`` in eval mode with frozen statistics: every batch goes through
        ``batch_processor`` -> forward -> ``build_loss``; a log block every ``print_freq`` batches.  The reference only logs the
        result; here ``(mean, std)`` of the per-batch losses is returned as well.  Log values are fetched asynchronously
        (losses.LazyFloat) and read when a block is printed, so the loop never waits for the device in between."""
        from ..network import losses
        from ..training.trainer import LogBook, RunningStat, Stopwatch, _scalar_proxy
        if self.val_data is None or self.batch_processor is None:
            raise ValueError('Tester.val() needs the val_data and batch_processor given to the constructor')
        modes = [(m, m.training) for m in self.model.modules()]      # EVERY module's mode (a PRN Dropout set to eval under a training root …)
        self.model.eval()
        was_lazy = losses.set_lazy_log(True)          # build_loss's log values: asynchronous proxies, read when a block is printed
        try:
            return self._val_loop(LogBook, RunningStat, Stopwatch, _scalar_proxy)
        finally:
            losses.set_lazy_log(was_lazy)
            for m, mode in modes:                     # the modes the caller had — a deliberate divergence: the reference
                m.training = mode                     # leaves the model in eval mode; a Trainer calling val() between epochs must get its modes back

    def _val_loop(self, LogBook, RunningStat, Stopwatch, _scalar_proxy):
        book, seen = LogBook(), RunningStat()
        batch_timer, data_timer = Stopwatch(), Stopwatch()
        logger.info('Val on validation set...')
        n = len(self.val_data)
        batch_timer.start()
        data_timer.start()
        with torch.no_grad():
            for step, batch in enumerate(self.val_data):
                data_timer.lap()
                inputs, gts, _ = self.batch_processor(self, batch)
                _, saved_for_loss = self.model(*inputs)
                batch_timer.lap()
                loss, log = self.model.build_loss(saved_for_loss, *gts)
                seen.add(_scalar_proxy(loss, True))
                book.record(log)
                if step % self.params.print_freq == 0:
                    text = '{}\n{}: epoch {}[{}/{}]'.format(self.params.exp_name, 'Validation', 0, step, n) + ''.join(book.lines())
                    text += '\n\t({:.2f}/{:.2f}s, fps:{:.1f})'.format(data_timer.mean + 1e-6, batch_timer.mean + 1e-6,
                                                                     self.params.batch_size / (batch_timer.mean + 1e-6))
                    logger.info(text)
                    batch_timer.clear()
                    data_timer.clear()
                data_timer.start()
                batch_timer.start()
        mean, std = seen.value()
        logger.info('\n\nValidation loss: mean: {}, std: {}'.format(mean, std))
        self.last_val_log = book
        return mean, std

    # ------------------------------------------------------------------ shared pieces
    def _boxes(self, scores, classification, transformed_anchors, scale):
        """tester.py:228-234 / :304-310: person boxes with score > 0.5, coordinates multiplied by `scale`."""
        if scores.numel() == 0:
            return []
        scores = scores.detach().cpu().numpy()
        classification = classification.detach().cpu().numpy()
        boxes = transformed_anchors.detach().cpu().numpy()
        out = []
        for j in np.where(scores > 0.5)[0]:
            if int(classification[j]) == 0:
                out.append((boxes[j, :] * scale).tolist())
        return out

    @staticmethod
    def _body_joints(joint_list):
        """tester.py:158-164 / :217-223: drop the neck (type 1), shift the later types down by one."""
        joints = []
        for joint in joint_list.tolist():
            if int(joint[-1]) != 1:
                joint[-1] = max(0, int(joint[-1]) - 1)
                joints.append(joint)
        return joints

    def prn_process(self, kps, bbox_list, file_name, image_id=0):
        return prn_process(self.model, kps, bbox_list, file_name, image_id, coeff=self.params.coeff, in_thres=self.params.in_thres)

    # ------------------------------------------------------------------ single scale (Tester.test, :194-245)
    def infer_image(self, img, file_name='', image_id=0):
        img = _to_device_image(img, self.dev)
        shape_dst = max(img.shape[0], img.shape[1])
        scale = float(shape_dst) / self.params.inp_size
        pad = abs(img.shape[1] - img.shape[0])
        sq = torch.zeros((img.shape[0] + pad, img.shape[1] + pad, 3), dtype=torch.float32, device=self.dev)
        sq[:img.shape[0], :img.shape[1]] = img                                    # np.pad(..., 'constant')
        sq = sq[:shape_dst, :shape_dst]
        img_resized = resize(sq, (self.params.inp_size, self.params.inp_size), cubic=False)
        img_input = resnet_preprocess(img_resized)[None]
        with torch.no_grad():
            heatmaps, (scores, classification, transformed_anchors) = self.model([img_input, 'both'])
        param = {'thre1': 0.1, 'thre2': 0.05, 'thre3': 0.5}
        joint_list = get_joint_list(img_resized, param, heatmaps[0].permute(1, 2, 0), scale)
        bboxs = self._boxes(scores, classification, transformed_anchors, scale)
        return self.prn_process(self._body_joints(joint_list), bboxs, file_name, image_id)

    def infer_images_batched(self, images, file_names=None, image_ids=None, batch=64, pipeline=True):
        """``infer_image`` for many images with the network, the peak extraction and the PRN assignment running on whole batches:
        every image is padded to a square and resized to inp_size x inp_size (tester.py:203-213), so images of any size share a
        forward; per-image scales are applied to peaks and boxes afterwards.  Same result dicts, image by image, as
        ``infer_image`` (tests/test_round3_gpu.py).

        ``pipeline`` (round 6): the network of batch k + 1 is enqueued BEFORE batch k is post-processed, and the post-processing
        launches (NMS, peaks, PRN forward, candidate compaction) go to a second stream that waits for batch k's network only — the
        host's numpy work and its device reads run under the next batch's network instead of leaving the GPU idle (2 - 5 ms of a 33 ms
        batch at 640 x 640 x 64).  Results are identical (same launches on the same data)."""
        n = len(images)
        file_names = file_names if file_names is not None else [''] * n
        image_ids = image_ids if image_ids is not None else [0] * n
        S = self.params.inp_size
        out = [None] * n
        post = self._post_stream() if pipeline else None
        pending = None
        for i0 in list(range(0, n, batch)) + [None]:
            nxt = None
            if i0 is not None:
                idx = list(range(i0, min(n, i0 + batch)))
                xs, scales = [], []
                for i in idx:
                    img = _to_device_image(images[i], self.dev)
                    shape_dst = max(img.shape[0], img.shape[1])
                    scales.append(float(shape_dst) / S)
                    pad = abs(img.shape[1] - img.shape[0])
                    sq = torch.zeros((img.shape[0] + pad, img.shape[1] + pad, 3), dtype=torch.float32, device=self.dev)
                    sq[:img.shape[0], :img.shape[1]] = img
                    xs.append(resnet_preprocess(resize(sq[:shape_dst, :shape_dst], (S, S), cubic=False)))
                with torch.no_grad():
                    heat, anchors, cls, keep = self.model.forward_padded_begin(torch.stack(xs))
                ev = None
                if post is not None:
                    ev = torch.cuda.Event()
                    ev.record()
                nxt = (idx, scales, heat, anchors, cls, keep, ev)
            if not pipeline and nxt is not None:
                self._finish_batch(nxt, out, file_names, image_ids, None)
                nxt = None
            if pending is not None:
                self._finish_batch(pending, out, file_names, image_ids, post)
            pending = nxt
        return out

    def _post_stream(self):
        if getattr(self, '_post', None) is None:
            self._post = torch.cuda.Stream(device=self.dev)
        return self._post

    def _finish_batch(self, item, out, file_names, image_ids, post):
        """Detections, peaks and the PRN assignment of one batch whose network has been enqueued (infer_images_batched)."""
        idx, scales, heat, anchors, cls, keep, ev = item
        if post is not None:
            post.wait_event(ev)
            with torch.cuda.stream(post):
                self._finish_batch_on_current_stream(idx, scales, heat, anchors, cls, out, file_names, image_ids)
        else:
            self._finish_batch_on_current_stream(idx, scales, heat, anchors, cls, out, file_names, image_ids)

    def _finish_batch_on_current_stream(self, idx, scales, heat, anchors, cls, out, file_names, image_ids):
        from ..network.joint_utils import NMS_batch_arrays, body_peaks_flat
        from .prn_process import prn_assign_arrays
        S = self.params.inp_size
        boxes, scores, kept = self.model.detect_padded(anchors, cls)
        pk, cnt = NMS_batch_arrays({'thre1': 0.1}, heat, float(S) / heat.shape[2])
        peaks_xy, joint_off = body_peaks_flat(pk, cnt)
        sc = np.asarray(scales)
        per_img = np.diff(np.concatenate([joint_off[:, 0], joint_off[-1:, 17]]))          # peaks per image
        peaks_xy = peaks_xy * np.repeat(sc, per_img)[:, None]                             # get_joint_list: peaks * scale
        nmax = boxes.shape[1]
        if nmax:
            # score > 0.5 among the kept rows (the single class is class 0 = person: forward_all_images_padded refuses anything else,
            # so the reference's `classification == 0` filter, tester.py:232, holds by construction); masks built on the host from the
            # two arrays that travel anyway
            scores_h, boxes_h = scores.cpu().numpy(), boxes.cpu().numpy()
            ok_h = (scores_h > 0.5) & (np.arange(nmax)[None, :] < np.asarray(kept)[:, None])
        else:
            ok_h, boxes_h = np.zeros((len(idx), 0), dtype=bool), np.zeros((len(idx), 0, 4), dtype=np.float32)
        # boxes * scale in float32 like the reference's numpy expression (tester.py:228-234), image-major
        b4 = (boxes_h[ok_h] * np.repeat(sc, ok_h.sum(1))[:, None].astype(np.float32)).astype(np.float64)
        start = np.concatenate([[0], np.cumsum(ok_h.sum(1))]).astype(np.int32)
        b4[:, 2:] -= b4[:, :2]
        kp = prn_assign_arrays(self.model, peaks_xy, joint_off, b4, start, in_thres=self.params.in_thres)
        for li, i in enumerate(idx):
            res = []
            for bi in range(start[li], start[li + 1]):
                k = np.zeros(51)
                k[0::3], k[1::3], k[2::3] = kp[bi, :, 0], kp[bi, :, 1], kp[bi, :, 2]
                pose_score = 0
                for f in range(17):
                    pose_score += kp[bi, f, 2]
                res.append({'image_id': image_ids[i], 'file_name': file_names[i], 'category_id': 1, 'bbox': b4[bi].tolist(),
                            'score': pose_score / 17.0, 'keypoints': k.tolist()})
            out[i] = res

    def test(self, images):
        """tester.py:194-245 over decoded images: ``images`` maps file name -> [H, W, 3] BGR array.  With
        ``params.testresult_write_json`` the result list also goes to <testresult_dir>multipose_results.json (:243-245);
        drawing the canvases (:238-241) needs cv2 and stays with the caller."""
        out = []
        for name, img in images.items():
            out.extend(self.infer_image(img, name))
        if self.params.testresult_write_json:
            with open(self.params.testresult_dir + 'multipose_results.json', "w") as f:
                json.dump(out, f)
        return out

    def coco_eval(self, images, coco=None):
        """tester.py:131-191 from the decoded images on: ``images`` yields (image_id, file_name, [H, W, 3] BGR array) — what the
        reference reads with pycocotools + cv2.imread.  Every image runs the 5-scale, flipped test-time augmentation; the results
        (keypoints in COCO order) are written to ``params.coco_result_filename`` (indent 4, :176-178).  With pycocotools present
        and ``coco`` the loaded annotation object, the keypoint evaluation runs as in :179-186 and the file is removed unless
        ``params.testresult_write_json`` (:188-189); without it (this image has none) the file is the product and is kept.
        Returns the result list."""
        results, img_ids = [], []
        for image_id, file_name, img in images:
            img_ids.append(image_id)
            results.extend(self.infer_image_multiscale(img, file_name, image_id))
        ann_filename = self.params.coco_result_filename
        with open(ann_filename, "w") as f:
            json.dump(results, f, indent=4)
        if coco is not None:
            try:
                from pycocotools.cocoeval import COCOeval
            except ImportError:
                logger.warning("pycocotools is not installed: results left in %s", ann_filename)
                return results
            ev = COCOeval(coco, coco.loadRes(ann_filename), 'keypoints')
            ev.params.imgIds = img_ids
            ev.evaluate()
            ev.accumulate()
            ev.summarize()
            if not self.params.testresult_write_json:
                os.remove(ann_filename)
        return results

    # ------------------------------------------------------------------ multi-scale + flip (Tester.coco_eval, :143-175)
    def _get_multiplier(self, img):
        """tester.py:256-262."""
        return [x * self.params.inp_size / float(img.shape[0]) for x in [0.5, 1., 1.5, 2, 2.5]]

    def _get_outputs(self, multiplier, img):
        """tester.py:264-313: heat-maps of every scale resized back to the image and averaged; boxes per scale."""
        H, W = img.shape[0], img.shape[1]
        heatmap_avg = torch.zeros((H, W, 18), dtype=torch.float64, device=self.dev)       # np.zeros(...) accumulator: float64 (tester.py:266)
        bbox_all = []
        for scale in multiplier:
            inp_size = scale * H
            im_cropped, im_scale, real_shape = crop_with_factor(img, inp_size, factor=32, pad_val=128)
            im_data = resnet_preprocess(im_cropped)[None]
            with torch.no_grad():
                heatmaps, (scores, classification, transformed_anchors) = self.model([im_data, 'both'])
            hm = heatmaps[0].permute(1, 2, 0)[:int(im_cropped.shape[0] / 4), :int(im_cropped.shape[1] / 4), :]
            hm = resize(hm, (hm.shape[0] * 4, hm.shape[1] * 4), cubic=True)
            hm = hm[0:real_shape[0], 0:real_shape[1], :]
            hm = resize(hm, (H, W), cubic=True)
            heatmap_avg = heatmap_avg + hm / len(multiplier)
            bbox_all.append(self._boxes(scores, classification, transformed_anchors, 1.0 / im_scale))
        return heatmap_avg, bbox_all

    @staticmethod
    def _handle_heat(normal_heat, flipped_heat):
        """tester.py:315-331."""
        return (normal_heat + flipped_heat.flip(1)[:, :, SWAP_HEAT]) / 2.

    def infer_image_multiscale(self, img, file_name='', image_id=0):
        img = _to_device_image(img, self.dev)
        multiplier = self._get_multiplier(img)
        orig_heat, orig_bbox_all = self._get_outputs(multiplier, img)
        flipped_heat, _ = self._get_outputs(multiplier, img.flip(1).contiguous())
        heatmaps = self._handle_heat(orig_heat, flipped_heat)
        param = {'thre1': 0.1, 'thre2': 0.05, 'thre3': 0.5}
        # the averaged maps are float64 as in the reference (np.zeros accumulator); the peak kernel takes float32, so they are
        # rounded ONCE here (the reference's find_peaks / cv2 refinement run on the float64 maps: a <= 1 ulp(f32) deviation)
        joint_list = get_joint_list(img, param, heatmaps[:, :, :18].float().contiguous(), 1)
        results = self.prn_process(self._body_joints(joint_list), orig_bbox_all[1], file_name, image_id)
        for result in results:                                    # tester.py:167-175: COCO keypoint order
            kp = result['keypoints']
            result['keypoints'] = [kp[COCO_ORDER[i] * 3 + j] for i in range(17) for j in range(3)]
        return results
