#!/usr/bin/env python3
"""Per-kernel MFMA / LDS utilisation table from the raw counter dump of tools/gpu_pmc_util.sh (tools/pmc_generic.py lines).
MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES): rocprofv3 reports one value per shader engine (32 per launch on
MI355X: 8 XCDs x 4), SQ_BUSY_CYCLES counts that engine's busy cycles once, the MFMA counter sums over its 8 CUs x 4 SIMDs (check:
the (removed) pixel-resident 1x1 kernel at 256->1024 @30x30 reads 460 800 per engine = 921 600 wave-level MFMAs x 16 cycles / 32 engines); LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import collections
import re
import sys

vals = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"^(.*?)\s+(SQ_\w+|GRBM_\w+|TA_\w+|TCP_\w+|TCC_\w+)\s+launches=\s*(\d+)\s+total=\s*([\d.]+)\s+per_launch=\s*([\d.]+)", line)
    if m:
        k = m.group(1).strip()
        vals[k][m.group(2)] = (int(m.group(3)), float(m.group(4)))
rows = []
for k, v in vals.items():
    if "SQ_BUSY_CYCLES" not in v or v["SQ_BUSY_CYCLES"][1] <= 0:
        continue
    n, busy = v["SQ_BUSY_CYCLES"]
    mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0.0))[1]
    wave_cyc = v.get("SQ_WAVE_CYCLES", (0, 0.0))[1]
    wait = v.get("SQ_WAIT_INST_ANY", (0, 0.0))[1]
    act = v.get("SQ_ACTIVE_INST_ANY", (0, 0.0))[1]
    conf = v.get("SQ_LDS_BANK_CONFLICT", (0, 0.0))[1]
    idx = v.get("SQ_LDS_IDX_ACTIVE", (0, 0.0))[1]
    rows.append((busy, k, n, mf / (32.0 * busy), wait / wave_cyc if wave_cyc else 0.0, act / wave_cyc if wave_cyc else 0.0, conf / idx if idx else 0.0,
                 v.get("SQ_INSTS_MFMA", (0, 0.0))[1], v.get("SQ_INSTS_VALU", (0, 0.0))[1]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("# rocprofv3 --pmc, bench.py --steps 3 --warmup 1, MPN_SIDE_STREAM=0 (one kernel on the GPU at a time), 11 steps incl. set-up; kernels by SQ busy cycles")
print("%-72s %8s %7s %10s %9s %9s %9s %9s" % ("kernel", "launches", "busy %", "MFMA busy", "waiting", "issuing", "LDS conf", "VALU/MFMA"))
for busy, k, n, mfu, wt, ac, cf, im, iv in rows[:40]:
    print("%-72s %8d %6.1f%% %9.1f%% %8.1f%% %8.1f%% %8.1f%% %9.1f" % (k[:72], n, 100 * busy / tot, 100 * mfu, 100 * wt, 100 * ac, 100 * cf, iv / im if im else 0.0))
