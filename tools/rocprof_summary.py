#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace rocpd database (…_results.db) into a per-kernel table.

usage: python tools/rocprof_summary.py gpurun_out/prof2/prof2_results.db STEPS "title" > profiles/rNN_….txt
"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = n.replace("unsigned short", "bf16")
    return n[:96]


def main():
    db, steps, title = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# %s" % title)
    print("# source: rocprofv3 --kernel-trace --stats (rocpd sqlite), %d steps; total kernel time %.1f ms = %.2f ms/step" % (steps, tot, tot / steps))
    print("%-98s %7s %10s %9s %9s %9s %6s" % ("kernel", "calls", "total_ms", "ms/step", "avg_us", "max_us", "pct"))
    for r in rows:
        if r[2] / tot < 0.0002:
            continue
        print("%-98s %7d %10.2f %9.3f %9.1f %9.1f %6.2f" % (short(r[0]), r[1], r[2], r[2] / steps, r[3], r[5], 100 * r[2] / tot))
    names = [r[0] for r in rows if ("conv_igemm_kernel" in r[0] or "conv_wgrad" in r[0]) and r[2] / tot > 0.02]
    for kn in names:
        g = cur.execute("select grid_x/workgroup_x, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels where name = ? "
                        "group by grid_x order by 3 desc", (kn,)).fetchall()
        print("\n# %s by launch geometry (workgroups)" % short(kn))
        for r in g[:12]:
            print("  blocks=%7d calls/step=%6.1f  ms/step=%8.3f  avg=%9.1f us" % (r[0], r[1] / steps, r[2] / steps, r[3]))
    foreign_kernels_per_step(cur)


def foreign_kernels_per_step(cur):
    """Per optimizer-step window (adam kernel to adam kernel): launches that are NOT this library's — hipMemcpy kernels
    (__amd_rocclr_copyBuffer / fillBuffer) and torch element-wise kernels — so that set-up steps (eager pass + recording, which do run
    torch ops) can be told from replayed steps."""
    rows = cur.execute("select start, end, name from kernels order by start").fetchall()
    adam = [r for r in rows if "adam" in r[2]]
    if len(adam) < 3:
        return
    print("\n# launches per step window that are not libmpn_hip kernels (copyBuffer / fillBuffer / at::native): window index -> count, kernel time us")
    wins = [(rows[0][0], adam[0][0])] + [(a[1], b[0]) for a, b in zip(adam[:-1], adam[1:])]
    out = []
    for i, (lo, hi) in enumerate(wins):
        f = [r for r in rows if lo <= r[0] and r[1] <= hi and ("rocclr" in r[2] or "at::native" in r[2] or "elementwise" in r[2])]
        out.append((i, len(f), sum(r[1] - r[0] for r in f) / 1e3))
    print("  " + "  ".join("%d:%d(%.0fus)" % o for o in out))
    body = sorted(o[1] for o in out[3:])
    if body:
        print("  steady state (windows 3..): min %d median %d max %d launches per step" % (body[0], body[len(body) // 2], body[-1]))


if __name__ == "__main__":
    main()
