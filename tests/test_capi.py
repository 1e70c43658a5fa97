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
        return re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*(?:,|;)", body)

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
