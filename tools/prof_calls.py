#!/usr/bin/env python3
"""List individual launches of one kernel/geometry from a rocprofv3 rocpd database, with the kernel that ran just before.
usage: python tools/prof_calls.py DB name_substring blocks [max_rows]"""
import sqlite3
import sys

db, sub, blocks = sys.argv[1], sys.argv[2], int(sys.argv[3])
lim = int(sys.argv[4]) if len(sys.argv) > 4 else 60
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, grid_x/workgroup_x, start, end from kernels order by start").fetchall()
n = 0
for i, (name, b, s, e) in enumerate(rows):
    if sub in name and b == blocks:
        prev = rows[i - 1]
        print("%8.1f us   gap_before=%6.1f us   prev=%s (%d blocks, %.1f us)" % ((e - s) / 1e3, (s - prev[3]) / 1e3, prev[0][:60], prev[1], (prev[3] - prev[2]) / 1e3))
        n += 1
        if n >= lim:
            break
