#!/usr/bin/env python3
"""Per-queue view of a rocprofv3 kernel trace: busy time per hardware queue / stream inside the steady-state window and the
tail (time the last queue keeps running after the others went quiet) per step.  usage: prof_streams.py DB"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("stream_id", "queue_id", "queue", "stream") if c in cols), None)
print("columns:", cols)
if qcol is None:
    sys.exit(0)
rows = cur.execute("select start, end, name, %s from kernels order by start" % qcol).fetchall()
adam = [r for r in rows if "adam_dev_kernel" in r[2] or "adam_kernel" in r[2]]
if len(adam) < 5:
    sys.exit(0)
for i in range(2, len(adam) - 1):
    lo, hi = adam[i][1], adam[i + 1][0]            # one step: after optimizer i, up to optimizer i+1
    step = [r for r in rows if r[0] >= lo and r[1] <= hi]
    by = {}
    for s, e, n, q in step:
        d = by.setdefault(q, [0, s, e, 0])
        d[0] += e - s
        d[1] = min(d[1], s)
        d[2] = max(d[2], e)
        d[3] += 1
    line = "step %d: %.2f ms |" % (i, (hi - lo) / 1e6)
    for q, (busy, s, e, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
        line += "  q%s: %d launches busy %.2f ms, first +%.2f last +%.2f ms |" % (q, n, busy / 1e6, (s - lo) / 1e6, (e - lo) / 1e6)
    print(line)

# tail of the side stream: time between the main stream's last kernel before the optimizer and the side stream's last kernel
main_q = max(set(r[3] for r in rows), key=lambda q: sum(1 for r in rows if r[3] == q))
print("tail per step (side stream still running after the main chain's last pre-optimizer kernel):")
for i in range(2, len(adam) - 1):
    lo, hi = adam[i][1], adam[i + 1][0]
    step = [r for r in rows if r[0] >= lo and r[1] <= hi]
    m = [r for r in step if r[3] == main_q and "adam" not in r[2]]
    s_ = [r for r in step if r[3] != main_q]
    if not m or not s_:
        continue
    t_m, t_s = max(r[1] for r in m), max(r[1] for r in s_)
    last = [r for r in s_ if r[1] > t_m]
    print("  step %d: main chain ends +%.2f ms, side ends +%.2f ms, tail %.2f ms (%d side kernels end after the main chain: %s)" % (
        i, (t_m - lo) / 1e6, (t_s - lo) / 1e6, (t_s - t_m) / 1e6, len(last), ", ".join(sorted(set(r[2][:40] for r in last)))[:160]))
