#!/usr/bin/env python3
"""Sum arbitrary rocprofv3 --pmc counters per kernel from a rocpd database.  usage: pmc_generic.py DB"""
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
cname = "counter_name" if "counter_name" in cols else ("counter" if "counter" in cols else None)
if cname is None:
    print("pmc_events columns:", cols)
    sys.exit(1)
rows = cur.execute("select name, %s, count(*), sum(counter_value) from pmc_events group by name, %s order by name" % (cname, cname)).fetchall()
for name, c, n, v in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
    print("%-72s %-28s launches=%6d  total=%16.0f  per_launch=%14.1f" % (name, c, n, v, v / n))
