"""C-ABI surface checks (CPU only, no compute calls): the shared library builds/loads, exports every
symbol include/mpn.h declares, the ctypes table matches the header, and the product refuses to run
without the MI355X instead of silently falling back."""
import os
import re
import subprocess

import pytest
import torch

from helpers import ROOT


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mpn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(mpn_[a-z0-9_]+)\s*\(", src))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from multiposenet.pytorch_amd import _lib
    _lib.build()
    L = _lib.lib()
    declared = header_symbols()
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(l.split()[-1] for l in out.splitlines() if " T mpn_" in l)
    assert declared, "no declarations parsed from include/mpn.h"
    assert declared == exported, "header/library mismatch: only in header %s, only in library %s" % (
        sorted(declared - exported), sorted(exported - declared))
    assert set(_lib.SIGNATURES) == declared, "ctypes table mismatch: %s" % sorted(set(_lib.SIGNATURES) ^ declared)
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.mpn_version()


def test_struct_layouts_match_header(tmp_path):
    """Compile include/mpn.h with gcc and compare sizeof/offsetof with the ctypes mirrors."""
    from multiposenet.pytorch_amd._lib import ConvParams, WgradParams
    import ctypes
    src = open(os.path.join(ROOT, "include", "mpn.h")).read()

    def fields(struct):
        body = src[src.index("typedef struct %s" % struct): src.index("} %s;" % struct)]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\d+\])?\s*(?:,|;)", body)      # scalars and fixed arrays

    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "mpn.h"', 'int main(void){']
    for st, cls in (("MpnConvParams", ConvParams), ("MpnWgradParams", WgradParams)):
        names = fields(st)
        assert [n for n, _ in cls._fields_] == names, st
        prog.append('printf("%%zu\\n", sizeof(%s));' % st)
        for n in names:
            prog.append('printf("%%zu\\n", offsetof(%s, %s));' % (st, n))
    prog.append("return 0;}")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).decode().split()]
    k = 0
    for cls in (ConvParams, WgradParams):
        assert ctypes.sizeof(cls) == vals[k]
        k += 1
        for n, _ in cls._fields_:
            assert getattr(cls, n).offset == vals[k], (cls.__name__, n)
            k += 1


def test_product_fails_loudly_without_gpu_or_library(tmp_path, monkeypatch):
    from multiposenet.pytorch_amd import _lib
    from multiposenet.pytorch_amd.lib.nms.pth_nms import pth_nms
    if not torch.cuda.is_available():
        with pytest.raises(_lib.MpnError):
            pth_nms(torch.zeros(3, 5), 0.5)
    # missing library -> MpnError, not a fallback
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MpnError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multiposenet")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("the CPU oracle", "").replace("CPU oracle's", "") or f.endswith(".hip"), \
                    "%s mentions the oracle" % os.path.join(dp, f)
                assert "import oracle" not in txt and "from oracle" not in txt


def test_entry_points_reject_bad_arguments_without_touching_the_gpu():
    """Error behaviour of the C ABI (compare SURVEY.md 8b: the reference's cffi functions return 1 always and never
    check anything): null pointers and nonsensical sizes come back as MPN_E_BADARG before any HIP call is made, so this
    runs on a machine without a GPU."""
    import ctypes
    from multiposenet.pytorch_amd import _lib
    from multiposenet.pytorch_amd._lib import ConvParams, WgradParams
    L = _lib.lib()
    BAD = -1
    src = open(os.path.join(ROOT, "include", "mpn.h")).read()
    m = re.search(r"#define\s+MPN_E_BADARG\s+\(?(-?\d+)\)?", src)
    if m:
        BAD = int(m.group(1))
    nul = ctypes.c_void_p(None)
    one = ctypes.c_void_p(0x1000)            # never dereferenced: validation fails first
    assert L.mpn_conv_forward(None, nul) == BAD
    assert L.mpn_conv_wgrad(None, nul) == BAD
    assert L.mpn_conv_wgrad_partials(None, nul) == BAD
    p = ConvParams()                          # all-zero struct: null tensors
    assert L.mpn_conv_forward(ctypes.byref(p), nul) == BAD
    p.x, p.w, p.y = 0x1000, 0x1000, 0x1000
    p.B, p.H, p.W, p.Ho, p.Wo, p.Cin, p.Cout, p.Cout_store, p.R, p.S, p.stride = 1, 8, 8, 8, 8, 30, 8, 8, 3, 3, 1
    p.dtype = 1
    assert L.mpn_conv_forward(ctypes.byref(p), nul) == BAD            # Cin not a multiple of the 64-byte K chunk
    w = WgradParams()
    assert L.mpn_conv_wgrad(ctypes.byref(w), nul) == BAD
    assert L.mpn_reduce_partials(nul, 4, 16, one, 1, nul) == BAD
    assert L.mpn_reduce_partials(one, 0, 16, one, 1, nul) == BAD
    assert L.mpn_weight_transpose(nul, one, 8, 1, 8, 8, 1, nul) == BAD
    assert L.mpn_weight_transpose(one, one, 8, 1, 8, 4, 1, nul) == BAD    # Cout_pad < Cout
    assert L.mpn_weight_transpose_batched(one, one, nul, 3, 10, 1, nul) == BAD
    assert L.mpn_bn_act_forward(nul, nul, one, one, one, 16, 8, 8, 1, 1, nul, nul) == BAD
    assert L.mpn_bn_bwd_reduce(one, nul, one, one, one, nul, nul, one, 1, 16, 32, 32, 1, 1, nul) == BAD   # relu without z or mask coefficients
    assert L.mpn_gt_heatmaps(nul, one, 1, 1, one, 4, 4, 4.0, 7.0, nul) == BAD
    assert L.mpn_gt_heatmaps(one, one, 1, 1, one, 4, 4, 0.0, 7.0, nul) == BAD
    assert L.mpn_heatmap_peaks(one, 0, 0, 0, 0, 1, 18, 0, 8, 0.1, 4.0, 1, one, one, 16, one, nul) == BAD
    assert L.mpn_heatmap_peaks(one, 0, 0, 0, 0, 1, 18, 8, 8, 0.1, 4.0, 1, one, one, 0, one, nul) == BAD       # cap == 0
    assert L.mpn_heatmap_peaks(one, 0, 0, 0, 0, 1, 18, 8, 8, 0.1, 4.0, 1, one, one, 16, nul, nul) == BAD     # no workspace
    assert L.mpn_heatmap_peaks_workspace_bytes(64, 18, 160, 160, 256) == (64 * 18 * 160 * 3 * 8 + 255) // 256 * 256 + 64 * 18 * 256 * 4 + 256
    # round 6 entry points: conv2 by position classes (H, W multiples of 8; channel counts multiples of 8; known dtype), BBoxTransform coefficients
    assert L.mpn_conv2cls_comb_elems(256, 128) == 256 * 9 * 256 + 2 * 9 * 256 * 9 * 128 + 2 * 9 * 256 * 128 and L.mpn_conv2cls_comb_elems(0, 128) == 0
    assert L.mpn_conv2cls_combine(nul, one, 256, 128, nul) == BAD and L.mpn_conv2cls_fold(one, nul, 256, 128, nul) == BAD
    assert L.mpn_conv2cls_expand(one, one, one, 1, 12, 16, 256, 1, nul) == BAD        # H not a multiple of 8
    assert L.mpn_conv2cls_expand(one, one, one, 1, 16, 16, 256, 7, nul) == BAD        # unknown dtype
    assert L.mpn_conv2cls_pool(one, one, one, 1, 16, 16, 250, 1, nul) == BAD          # channels not a multiple of 8
    assert L.mpn_conv2cls_tapsum(one, nul, 1, 2, 2, 256, 1, nul) == BAD and L.mpn_conv2cls_classsum(nul, one, 1, 2, 2, 256, nul) == BAD
    assert L.mpn_box_decode_clip_ms(one, one, one, 1, 4, 10.0, 10.0, None, nul) == BAD
    assert L.mpn_nms(nul, 4, 0.5, 0, one, one, one, nul) != 0
