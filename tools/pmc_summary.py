#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), collected separately as
MI355X_MICROARCH.md (HBM section) prescribes.  Units: the counters are KiB per dispatch; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, so reads are doubled (the guide's
correction); WRITE_SIZE is taken as is (uncalibrated).

usage: python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/pmc_results.db gpurun_out/pmc_WRITE_SIZE/pmc_results.db STEPS out.json
"""
import json
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n).replace("unsigned short", "bf16")
    return re.sub(r"\(.*$", "", n)


def load(db):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, kb, dur in cur.execute("select name, count(*), sum(counter_value), sum(duration) from pmc_events group by name"):
        out[short(name)] = (n, kb, dur)
    return out


def main():
    f, w, steps, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    F, W = load(f), load(w)
    rows = []
    for k in F:
        n, fkb, dur = F[k]
        wkb = W.get(k, (n, 0.0, 0))[1]
        rd = 2.0 * fkb * 1024.0          # gfx950 correction: FETCH_SIZE counts 64 B per 128-B request
        wr = wkb * 1024.0
        rows.append(dict(kernel=k, launches_per_step=n / steps, read_bytes_per_launch=rd / n, write_bytes_per_launch=wr / n,
                         hbm_bytes_per_launch=(rd + wr) / n, hbm_gb_per_step=(rd + wr) / steps / 1e9))
    rows.sort(key=lambda r: -r["hbm_gb_per_step"])
    tot = sum(r["hbm_gb_per_step"] for r in rows)
    json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; reads doubled per MI355X_MICROARCH.md; "
                        "R101 train_both 480x480 B=32 bf16", total_hbm_gb_per_step=tot, kernels=rows), open(dst, "w"), indent=1)
    print("total HBM traffic %.2f GB/step" % tot)
    for r in rows[:16]:
        print("%-52s %7.1f launches/step  rd %9.2f MB  wr %9.2f MB per launch   %7.2f GB/step" % (
            r["kernel"][:52], r["launches_per_step"], r["read_bytes_per_launch"] / 1e6, r["write_bytes_per_launch"] / 1e6, r["hbm_gb_per_step"]))


if __name__ == "__main__":
    main()
