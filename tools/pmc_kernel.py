#!/usr/bin/env python
"""Print per-kernel averages of every counter found in one or more rocprofv3 --pmc rocpd databases.
    python tools/pmc_kernel.py <substring of kernel name> db1 [db2 ...]"""
import sqlite3
import sys

pat = sys.argv[1]
for db in sys.argv[2:]:
    c = sqlite3.connect(db)
    agg = {}
    for name, cn, val, dur in c.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        if pat not in name:
            continue
        a = agg.setdefault(cn, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += val
        a[2] += dur
    for k, (n, v, d) in sorted(agg.items()):
        print(f"{k:32s} n={n:4d} avg={v / n:14.1f}  (kernel avg {d / n / 1e3:8.1f} us)")
