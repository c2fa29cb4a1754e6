#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration, and -- for the steady-state
part of the run -- the busy fraction of the GPU timeline (sum of kernel durations vs first-start .. last-end).
    python tools/rocpd_summary.py results.db [--last N] [--gaps US]
N = number of trailing dispatches to analyse; --gaps US also lists where the GPU sat idle: every gap >= US microseconds between the end of
one kernel and the start of the next, summed by the pair (kernel before -> kernel after), i.e. the host-bound seams of the run."""
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
    if "--gaps" in sys.argv:
        thr = float(sys.argv[sys.argv.index("--gaps") + 1]) * 1e3
        seams, idle_all, idle_big = {}, 0, 0
        end = rows[0][2]
        for i in range(1, len(rows)):
            gap = rows[i][1] - end
            if gap > 0:
                idle_all += gap
                if gap >= thr:
                    idle_big += gap
                    a = seams.setdefault((short(rows[i - 1][0])[:60], short(rows[i][0])[:60]), [0, 0])
                    a[0] += 1
                    a[1] += gap
            end = max(end, rows[i][2])
        print(f"# idle {idle_all / 1e6:.3f} ms of the span; {idle_big / 1e6:.3f} ms of it in gaps >= {thr / 1e3:.0f} us:")
        print(f"# {'count':>7} {'idle_ms':>10} {'avg_us':>9}  kernel before -> kernel after")
        for (a_, b_), (k, t) in sorted(seams.items(), key=lambda kv: -kv[1][1])[:40]:
            print(f"  {k:7d} {t / 1e6:10.3f} {t / k / 1e3:9.1f}  {a_}  ->  {b_}")


if __name__ == "__main__":
    main()
