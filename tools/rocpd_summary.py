#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration, and -- for the steady-state
part of the run -- the busy fraction of the GPU timeline (sum of kernel durations vs first-start .. last-end).
    python tools/rocpd_summary.py results.db [--last N]      (N = number of trailing dispatches to analyse for gaps)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:110]


def main():
    db = sys.argv[1]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    if last:
        rows = rows[-last:]
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    span = rows[-1][2] - rows[0][1]
    print(f"# {len(rows)} dispatches, kernel time {tot / 1e6:.3f} ms, timeline span {span / 1e6:.3f} ms, busy {100 * tot / span:.1f} %")
    print(f"# {'calls':>7} {'total_ms':>10} {'avg_us':>9} {'%':>6}  kernel")
    for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:7d} {t / 1e6:10.3f} {t / k / 1e3:9.2f} {100 * t / tot:6.2f}  {n}")


if __name__ == "__main__":
    main()
