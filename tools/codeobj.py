#!/usr/bin/env python3
"""Resource usage of every kernel in a built HIP library, read from the code objects themselves (no external tool).

    python tools/codeobj.py [path/to/libmpn_hip.so] [name filter]

The shared library carries one clang offload bundle per translation unit in its ``.hip_fatbin`` section; each bundle holds the
gfx950 code object (an ELF) whose NT_AMDGPU_METADATA note is a msgpack map with one entry per kernel: ``.vgpr_count``,
``.sgpr_count``, ``.private_segment_fixed_size`` (scratch bytes per lane — non-zero means spills or a stack), ``.vgpr_spill_count``,
``.sgpr_spill_count``, ``.group_segment_fixed_size`` (LDS).  tests/test_round4_cpu.py gates the hot kernels on these numbers."""
import os
import re
import struct
import sys

import msgpack


def _sections(elf):
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", elf, shoff + i * shentsize)
        secs.append((name, typ, off, size))
    stro = secs[shstrndx][2]
    return [(elf[stro + n: elf.index(b"\0", stro + n)].decode(), typ, off, size) for n, typ, off, size in secs]


def code_objects(so_path, arch="gfx950"):
    """The device ELFs for `arch` inside a fat HIP shared library (one per translation unit)."""
    data = open(so_path, "rb").read()
    fat = [s for s in _sections(data) if s[0] == ".hip_fatbin"]
    if not fat:
        raise ValueError("%s has no .hip_fatbin section" % so_path)
    _, _, off, size = fat[0]
    blob = data[off: off + size]
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):
        b = m.start()
        n, = struct.unpack_from("<Q", blob, b + 24)
        p = b + 32
        for _ in range(n):
            eoff, esize, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24: p + 24 + tl].decode()
            p += 24 + tl
            if arch in triple and esize > 0:
                out.append(blob[b + eoff: b + eoff + esize])
    return out


def kernels(so_path, arch="gfx950"):
    """One dict per kernel: name (mangled), vgpr, agpr, sgpr, scratch, spill_v, spill_s, lds, max_threads."""
    out = []
    for elf in code_objects(so_path, arch):
        out.extend(_kernels_of(elf))
    return out


def _kernels_of(elf):
    out = []
    if True:
        for _name, typ, off, size in _sections(elf):
            if typ != 7:                     # SHT_NOTE
                continue
            p = off
            while p < off + size:
                nsz, dsz, nt = struct.unpack_from("<III", elf, p)
                p += 12
                owner = elf[p: p + nsz]
                p += (nsz + 3) // 4 * 4
                desc = elf[p: p + dsz]
                p += (dsz + 3) // 4 * 4
                if nt == 32 and owner.startswith(b"AMDGPU"):
                    md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                    for k in md.get("amdhsa.kernels", []):
                        out.append({"name": k[".name"], "vgpr": k.get(".vgpr_count", 0), "agpr": k.get(".agpr_count", 0),
                                    "sgpr": k.get(".sgpr_count", 0), "scratch": k.get(".private_segment_fixed_size", 0),
                                    "spill_v": k.get(".vgpr_spill_count", 0), "spill_s": k.get(".sgpr_spill_count", 0),
                                    "lds": k.get(".group_segment_fixed_size", 0), "max_threads": k.get(".max_flat_workgroup_size", 0)})
    return out


def scratch_vs_mfma(so_path, name_fragment, arch="gfx950", objdump="/opt/rocm/lib/llvm/bin/llvm-objdump"):
    """Where a kernel's scratch traffic sits relative to its MFMA loop: disassembles the (first) kernel whose mangled name contains
    `name_fragment` and returns (instruction index of the first v_mfma, of the last v_mfma, [indices of scratch_* instructions]).
    A spill that is stored and reloaded outside [first, last] costs a few hundred cycles per workgroup; one inside multiplies by the
    k-steps (round 3 lost 3.3 ms/step that way).  tests/test_round4_cpu.py uses this for the kernels it tolerates spills in."""
    import subprocess
    import tempfile
    for elf in code_objects(so_path, arch):
        names = [k["name"] for k in _kernels_of(elf)]
        hit = [n for n in names if name_fragment in n]
        if not hit:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf); f.flush()
            txt = subprocess.check_output([objdump, "-d", "--disassemble-symbols=" + hit[0], f.name]).decode()
        ins = [l.split("//")[0].strip() for l in txt.splitlines() if l.startswith("\t")]
        mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
        sc = [i for i, l in enumerate(ins) if l.startswith("scratch_")]
        return (mf[0] if mf else -1), (mf[-1] if mf else -1), sc
    raise KeyError(name_fragment)


def waves_per_simd(vgpr_plus_agpr):
    """Register-limited waves per SIMD on gfx950 (512-entry file, allocation granule 8; MI355X_MICROARCH.md, register files)."""
    alloc = max(8, (vgpr_plus_agpr + 7) // 8 * 8)
    return min(8, 512 // alloc)


def mangled(template, *args):
    """Itanium-mangled fragment of a kernel template instantiation in the anonymous namespace, e.g.
    mangled('conv_igemm_kernel', 't', 128, 128, False, False, False) -> '17conv_igemm_kernelItLi128ELi128ELb0ELb0ELb0EE'
    (types by their mangled spelling: 't' = unsigned short (bf16 storage), 'f' = float, 'DF16_' = _Float16)."""
    parts = []
    for a in args:
        if isinstance(a, bool):
            parts.append("Lb%dE" % int(a))
        elif isinstance(a, int):
            parts.append("Li%dE" % a)
        else:
            parts.append(a)
    return "%d%sI%sE" % (len(template), template, "".join(parts))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    so = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(here, "..", "multiposenet", "pytorch_amd", "libmpn_hip.so")
    flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    for k in sorted(kernels(so), key=lambda k: k["name"]):
        if flt in k["name"]:
            print("%-110s vgpr %3d agpr %3d sgpr %3d waves/SIMD %d scratch %4d spill v%d s%d lds %6d"
                  % (k["name"][:110], k["vgpr"], k["agpr"], k["sgpr"], waves_per_simd(k["vgpr"] + k["agpr"]), k["scratch"], k["spill_v"], k["spill_s"], k["lds"]))
