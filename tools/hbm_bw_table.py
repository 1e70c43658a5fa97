#!/usr/bin/env python3
"""HBM bandwidth per kernel = PMC traffic per launch (tools/pmc_summary.py json) / average launch duration of a serial
rocprofv3 kernel trace (tools/rocprof_summary.py text).  usage: hbm_bw_table.py pmc.json kernel_trace_serial.txt > table.txt"""
import json
import re
import sys


def main():
    pmc = json.load(open(sys.argv[1]))
    dur = {}
    for line in open(sys.argv[2]):
        m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if not m:
            continue
        name = re.sub(r"\(.*$", "", m.group(1)).strip()
        dur[name] = float(m.group(5))
    print("# HBM bandwidth per kernel = PMC traffic per launch (%s: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes,\n"
          "# MPN_SIDE_STREAM=0, library build %s) / average launch duration of the serial kernel trace (%s).\n"
          "# HBM3E peak ~8 TB/s; streaming ceiling measured with torch's elementwise kernels on this box: 5.8-5.9 TB/s."
          % (sys.argv[1], pmc.get("build_id", "?"), sys.argv[2]))
    print("%-52s %9s %12s %9s %8s %9s" % ("kernel", "launch/st", "MB/launch", "avg_us", "TB/s", "GB/step"))
    for k in pmc["kernels"][:24]:
        us = dur.get(k["kernel"])
        if us is None:
            cand = [n for n in dur if n.startswith(k["kernel"].split("<")[0]) and ("<" not in k["kernel"] or k["kernel"] in n)]
            us = dur[cand[0]] if cand else None
        mb = k["hbm_bytes_per_launch"] / 1e6
        print("%-52s %9.1f %12.1f %9s %8s %9.2f" % (k["kernel"][:52], k["launches_per_step"], mb, "%.1f" % us if us else "-",
                                                    "%.2f" % (mb / us) if us else "-", k["hbm_gb_per_step"]))
    print("# total %.1f GB/step" % pmc["total_hbm_gb_per_step"])


if __name__ == "__main__":
    main()
