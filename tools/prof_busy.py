#!/usr/bin/env python3
"""GPU busy time (union of kernel intervals) vs wall time from a rocprofv3 rocpd database; lists the largest idle gaps
and the idle time attributed to the kernel that follows each gap.  usage: prof_busy.py DB [skip_fraction]"""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select start, end, name from kernels order by start").fetchall()
# steady-state window: from the end of the 3rd optimizer launch to the end of the last one (whole steps only)
adam = [r for r in rows if "adam_dev_kernel" in r[2] or "adam_kernel" in r[2]]
if len(adam) >= 5:
    lo, hi = adam[2][1], adam[-1][1]
    nsteps = len(adam) - 3
else:
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo, hi, nsteps = t0 + (t1 - t0) * skip, t1, 1
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
wall = hi - lo
print("window: %d steps, %.2f ms/step" % (nsteps, wall / 1e6 / nsteps))
busy, gaps, cur_end = 0, [], rows[0][0]
gap_by = defaultdict(float)
for s, e, n in rows:
    if s > cur_end:
        gaps.append((s - cur_end, n))
        gap_by[n[:70]] += s - cur_end
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
ksum = sum(e - s for s, e, _ in rows)
print("wall %.2f ms  busy(union) %.2f ms (%.1f%%)  idle %.2f ms  sum of kernel durations %.2f ms (overlap factor %.2f)  launches %d" % (
    wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, ksum / 1e6, ksum / busy, len(rows)))
print("gaps: n=%d  >5us: %d  >20us: %d  >100us: %d" % (len(gaps), sum(g > 5e3 for g, _ in gaps), sum(g > 2e4 for g, _ in gaps), sum(g > 1e5 for g, _ in gaps)))
print("idle time by the kernel that ends the gap:")
for n, g in sorted(gap_by.items(), key=lambda kv: -kv[1])[:15]:
    print("  %8.2f ms  %s" % (g / 1e6, n))
