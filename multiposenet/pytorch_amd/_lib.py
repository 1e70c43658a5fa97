"""ctypes binding of libmpn_hip.so — the C-ABI boundary of the product (include/mpn.h).

The shared library holds every kernel of the hot path.  There is NO fallback: if the library is
missing or a call fails, ``MpnError`` is raised.  torch must be imported before the library is
loaded so both resolve the same ``libamdhip64.so.7`` (torch bundles its own copy).
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (must precede CDLL: pins the HIP runtime both sides use)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpn_hip.so")
CSRC = os.path.join(_HERE, "csrc")

F32, BF16, F16 = 0, 1, 2


class MpnError(RuntimeError):
    pass


class ConvParams(ctypes.Structure):
    _fields_ = [
        ("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("y", ctypes.c_void_p),
        ("bias", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("res", ctypes.c_void_p), ("res_mask", ctypes.c_void_p),
        ("stats", ctypes.c_void_p),
        ("x_sB", ctypes.c_int64), ("x_sH", ctypes.c_int64), ("x_sW", ctypes.c_int64),
        ("y_sB", ctypes.c_int64), ("y_sP", ctypes.c_int64),
        ("res_sB", ctypes.c_int64), ("res_sP", ctypes.c_int64),
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("Cin", ctypes.c_int32),
        ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32), ("Cout", ctypes.c_int32), ("Cout_store", ctypes.c_int32),
        ("R", ctypes.c_int32), ("S", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32),
        ("mode", ctypes.c_int32), ("act", ctypes.c_int32), ("res_mode", ctypes.c_int32),
        ("res_H", ctypes.c_int32), ("res_W", ctypes.c_int32), ("accumulate", ctypes.c_int32),
        ("dtype", ctypes.c_int32), ("out_f32", ctypes.c_int32),
        ("nseg", ctypes.c_int32), ("seg_H", ctypes.c_int32 * 5), ("seg_W", ctypes.c_int32 * 5), ("seg_tile0", ctypes.c_int32 * 6),
        ("seg_x", ctypes.c_void_p * 5), ("seg_y", ctypes.c_void_p * 5),
        ("bnb_y", ctypes.c_void_p), ("bnb_z", ctypes.c_void_p), ("bnb_mask", ctypes.c_void_p), ("bnb_mean", ctypes.c_void_p), ("bnb_invstd", ctypes.c_void_p),
        ("bnb_scale", ctypes.c_void_p), ("bnb_shift", ctypes.c_void_p), ("bnb_partial", ctypes.c_void_p), ("bnb_relu", ctypes.c_int32),
        ("kseg_n", ctypes.c_int32), ("kseg_c", ctypes.c_int32), ("kseg_shift", ctypes.c_int32 * 4), ("kseg_x", ctypes.c_void_p * 4),
        ("fin_counters", ctypes.c_void_p), ("fin_gamma", ctypes.c_void_p), ("fin_beta", ctypes.c_void_p), ("fin_rm", ctypes.c_void_p),
        ("fin_rv", ctypes.c_void_p), ("fin_out", ctypes.c_void_p), ("fin_dgamma", ctypes.c_void_p), ("fin_dbeta", ctypes.c_void_p),
        ("fin_count", ctypes.c_double), ("fin_momentum", ctypes.c_float), ("fin_eps", ctypes.c_float), ("fin_train", ctypes.c_int32),
        ("y_step", ctypes.c_int32), ("y_oh", ctypes.c_int32), ("y_ow", ctypes.c_int32), ("y_H", ctypes.c_int32), ("y_W", ctypes.c_int32),
        ("w_taps", ctypes.c_int32), ("wtap0", ctypes.c_int32), ("wtap_dr", ctypes.c_int32), ("wtap_ds", ctypes.c_int32),
    ]


class WgradParams(ctypes.Structure):
    _fields_ = [
        ("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("ws", ctypes.c_void_p),
        ("x_sB", ctypes.c_int64), ("x_sH", ctypes.c_int64), ("x_sW", ctypes.c_int64), ("dy_sP", ctypes.c_int64),
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("Cin", ctypes.c_int32),
        ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32), ("Cout", ctypes.c_int32),
        ("R", ctypes.c_int32), ("S", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32),
        ("dtype", ctypes.c_int32), ("chunks", ctypes.c_int32),
        ("db", ctypes.c_void_p), ("db_ws", ctypes.c_void_p),
        ("nseg", ctypes.c_int32), ("seg_H", ctypes.c_int32 * 5), ("seg_W", ctypes.c_int32 * 5), ("seg_chunk0", ctypes.c_int32 * 6),
        ("seg_chunk_pixels", ctypes.c_int32), ("seg_x", ctypes.c_void_p * 5), ("seg_dy", ctypes.c_void_p * 5),
        ("kseg_n", ctypes.c_int32), ("kseg_c", ctypes.c_int32), ("kseg_shift", ctypes.c_int32 * 4), ("kseg_x", ctypes.c_void_p * 4),
    ]


_vp, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
_PC = ctypes.POINTER(ConvParams)
_PW = ctypes.POINTER(WgradParams)

# name -> (restype, argtypes); mirrors include/mpn.h one to one (tests/test_capi.py checks this table
# against the header and against the exported dynamic symbols).
SIGNATURES = {
    "mpn_gt_heatmaps": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, ctypes.c_double, ctypes.c_double, _vp]),
    "mpn_heatmap_peaks_workspace_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "mpn_heatmap_peaks": (_i, [_vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _f, ctypes.c_double, _i, _vp, _vp, _i, _vp, _vp]),
    "mpn_resize": (_i, [_vp, _i64, _i64, _i64, _i, _i, _i, _vp, _i, _i, _i, ctypes.c_double, ctypes.c_double, _vp]),
    "mpn_prn_build_maps": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, ctypes.c_double, _vp, _vp, _vp, _vp, _vp]),
    "mpn_prn_scores": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mpn_prn_scores_compact": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mpn_prn_match_host": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "mpn_conv_stats_tiles": (_i, [_PC]),
    "mpn_conv_tile_rows": (_i, [_PC]),
    "mpn_conv_shared_tile": (_i, [_PC]),
    "mpn_conv_forward": (_i, [_PC, _vp]),
    "mpn_debug_wgrad_prof": (_i, [_vp]),
    "mpn_debug_igemm_prof": (_i, [_vp]),
    "mpn_conv_wgrad_chunks": (_i, [_PW]),
    "mpn_conv_wgrad_seg_plan": (_i, [_PW]),
    "mpn_conv_wgrad": (_i, [_PW, _vp]),
    "mpn_conv_wgrad_partials": (_i, [_PW, _vp]),
    "mpn_conv_wgrad_kernel_id": (_i, [_PW]),
    "mpn_reduce_partials": (_i, [_vp, _i, _i64, _vp, _i, _vp]),
    "mpn_cast_f32_to_bf16": (_i, [_vp, _vp, _i64, _vp]),
    "mpn_cast_f32": (_i, [_vp, _vp, _i64, _i, _vp]),
    "mpn_weight_transpose": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mpn_weight_transpose_batched": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp]),
    "mpn_weight_pad_k": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "mpn_stem_pack_weight": (_i, [_vp, _vp, _i, _i, _vp]),
    "mpn_stem_unpack_wgrad": (_i, [_vp, _vp, _i, _vp]),
    "mpn_stem_pack_image": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _i, _i, _i, _i, _vp]),
    "mpn_bn_finalize_train": (_i, [_vp, _i, _i, _i64, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "mpn_bn_finalize_eval": (_i, [_i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "mpn_bn_act_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "mpn_bn_bwd_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _vp]),
    "mpn_bn_bwd_finalize": (_i, [_vp, _i, _i, _i64, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "mpn_bn_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _vp, _vp]),
    "mpn_bn_bwd_chunks": (_i, [_i64, _i, _i]),
    "mpn_maxpool3x3s2_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mpn_maxpool3x3s2_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mpn_upsample_nearest_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mpn_upsample_nearest_slice": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mpn_upsample_nearest_slice_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mpn_export_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _vp]),
    "mpn_import_grad": (_i, [_vp, _i64, _i64, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mpn_nchw_to_nhwc_f32": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _i, _i, _i, _i, _vp]),
    "mpn_det_pack": (_i, [_vp, _i, _vp, _i, _i64, _i, _i, _i64, _vp]),
    "mpn_det_unpack": (_i, [_vp, _vp, _i, _i, _i64, _i, _i, _i64, _vp]),
    "mpn_relu_forward": (_i, [_vp, _vp, _i64, _i, _vp]),
    "mpn_relu_backward": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "mpn_add_inplace": (_i, [_vp, _vp, _i64, _i, _vp]),
    "mpn_channel_sum": (_i, [_vp, _i, _i64, _i, _i, _vp, _i, _vp]),
    "mpn_colsum_rows": (_i, [_vp, _i, _i64, _i, _i, _vp, _vp]),
    "mpn_channel_sum_chunks": (_i, [_i64, _i, _i]),
    "mpn_mse_chunks": (_i, [_i64]),
    "mpn_mse_heatmap_forward": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _i, _vp, _vp]),
    "mpn_mse_heatmap_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "mpn_mse_train_blocks": (_i, [_i, _i, _i]),
    "mpn_mse_heatmap_train": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "mpn_focal_blocks": (_i, [_i]),
    "mpn_focal_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mpn_focal_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "mpn_sigmoid_forward": (_i, [_vp, _vp, _i64, _vp]),
    "mpn_gather_dets": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "mpn_sigmoid_backward": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "mpn_add_softmax_rows": (_i, [_vp, _i64, _vp, _vp, _i, _i, _i, _vp]),
    "mpn_softmax_rows_backward": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _i, _vp]),
    "mpn_bce_mean_backward": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "mpn_dropout": (_i, [_vp, _vp, _i64, ctypes.c_uint64, _f, _i, _vp]),
    "mpn_bce_chunks": (_i, [_i64]),
    "mpn_bce_mean_forward": (_i, [_vp, _vp, _i64, _vp, _i, _vp, _vp]),
    "mpn_box_decode_clip": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _vp]),
    "mpn_box_decode_clip_ms": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, ctypes.POINTER(ctypes.c_float), _vp]),
    "mpn_clip_boxes": (_i, [_vp, _i64, _f, _f, _vp]),
    "mpn_score_filter": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    "mpn_score_filter_batched": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "mpn_gather_dets_batched": (_i, [_vp, _i64, _vp, _i64, _vp, _i, _i, _vp, _vp, _i64, _vp]),
    "mpn_nms_batched_workspace_bytes": (_i64, [_i, _i64]),
    "mpn_nms_batched": (_i, [_vp, _i64, _vp, _i, _i64, _f, _i, _vp, _i64, _vp, _vp, _vp]),
    "mpn_nms_batched_topk": (_i, [_vp, _i64, _vp, _i, _i64, _i64, _f, _i, _vp, _i64, _vp, _vp, _vp]),
    "mpn_nms_workspace_bytes": (_i64, [_i64]),
    "mpn_nms": (_i, [_vp, _i64, _f, _i, _vp, _vp, _vp, _vp]),
    "mpn_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _f, _f, _vp]),
    "mpn_adam_advance": (_i, [_vp, _vp]),
    "mpn_adam_step_dev": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "mpn_fill_f32": (_i, [_vp, _f, _i64, _vp]),
    "mpn_copy_bytes": (_i, [_vp, _vp, _i64, _vp]),
    "mpn_step_log": (_i, [_vp, _vp, _vp, _vp]),
    "mpn_conv2cls_comb_elems": (_i64, [_i, _i]),
    "mpn_conv2cls_combine": (_i, [_vp, _vp, _i, _i, _vp]),
    "mpn_conv2cls_expand": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mpn_conv2cls_classsum": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "mpn_conv2cls_pool": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mpn_conv2cls_tapsum": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mpn_conv2cls_fold": (_i, [_vp, _vp, _i, _i, _vp]),
    "mpn_version": (ctypes.c_char_p, []),
}

# entry points that return a count, not a status
_COUNT_FUNCS = {"mpn_conv_stats_tiles", "mpn_conv_tile_rows", "mpn_conv_shared_tile", "mpn_conv_wgrad_chunks", "mpn_conv_wgrad_seg_plan", "mpn_conv_wgrad_kernel_id", "mpn_bn_bwd_chunks", "mpn_channel_sum_chunks",
                "mpn_mse_chunks", "mpn_mse_train_blocks", "mpn_focal_blocks", "mpn_bce_chunks", "mpn_nms_workspace_bytes", "mpn_nms_batched_workspace_bytes", "mpn_heatmap_peaks_workspace_bytes", "mpn_conv2cls_comb_elems", "mpn_version"}

_lib = None


def build(force=False):
    """Compile every HIP source for gfx950 into libmpn_hip.so (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC])
    if not os.path.exists(LIB_PATH):
        raise MpnError("build did not produce %s" % LIB_PATH)
    return LIB_PATH


def use_experiments_build():
    """tools/ only: load libmpn_hip_experiments.so (csrc/Makefile `experiments`: PROF instantiations, MPN_DEBUG_FLAGS / MPN_WGRAD_ABLATE
    ablations, environment overrides of tuned constants) instead of the production library.  Must be called before the first lib()."""
    global LIB_PATH
    assert _lib is None, "call use_experiments_build() before anything loads the library"
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC, "experiments"])
    LIB_PATH = os.path.join(_HERE, "libmpn_hip_experiments.so")
    return LIB_PATH


def lib():
    """Load (once) and return the ctypes handle; raises MpnError when the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MpnError("libmpn_hip.so not found at %s — run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or `make -C %s`).  There is no CPU fallback." % (LIB_PATH, CSRC))
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        raise MpnError("%s failed with status %d" % (what, status))


# While replay.py records a training step every launch is appended here as (callable, args, is_c_abi): C-ABI entry points through
# call(), torch-level stream/event operations through gpu_op().  Replaying the list re-issues exactly the same work.
TAPE = None


def call(name, *args):
    """Invoke a status-returning entry point and raise on failure."""
    fn = getattr(lib(), name)
    st = fn(*args)
    if name in _COUNT_FUNCS:
        return st
    if st != 0:
        raise MpnError("%s failed with status %d" % (name, st))
    if TAPE is not None:
        TAPE.append((fn, args, True))
    return 0


def gpu_op(fn, *args):
    """A device-side operation that is not a C-ABI launch (event record / stream wait, a collective, a tiny torch op on
    static tensors): run it and, while a step is being recorded, remember it for replay."""
    global TAPE
    tape, TAPE = TAPE, None          # launches made INSIDE fn belong to fn: replaying fn re-issues them, so they are not recorded twice
    try:
        out = fn(*args)
    finally:
        TAPE = tape
    if tape is not None:
        tape.append((fn, args, False))
    return out


def call_raw(name, *args):
    """A C-ABI launch issued from inside a gpu_op closure (e.g. the per-bucket Adam update behind a collective): never recorded
    on its own — the enclosing closure is what the launch list holds."""
    st = getattr(lib(), name)(*args)
    if st != 0:
        raise MpnError("%s failed with status %d" % (name, st))
    return 0
