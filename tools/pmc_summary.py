#!/usr/bin/env python
"""Per-kernel averages of every counter in a rocprofv3 --pmc rocpd sqlite.  usage: pmc_summary.py <db> [substr]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute("select kernel_name, grid_size, counter_name, value from counters_collection").fetchall()
agg = {}
for k, g, cn, v in rows:
    if sub not in k:
        continue
    m = re.search(r"([a-z_0-9]+_kernel<[^>]*>)", k)
    key = ((m.group(1) if m else k[:60]), g)
    a = agg.setdefault(key, {})
    t = a.setdefault(cn, [0, 0.0])
    t[0] += 1
    t[1] += v
for (k, g), cs in sorted(agg.items()):
    print(f"{k} grid={g}")
    for cn, (n, tot) in sorted(cs.items()):
        print(f"    {cn:34s} n={n:3d} avg={tot / n:16.1f}")
