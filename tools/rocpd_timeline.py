#!/usr/bin/env python3
"""Ordered dispatch list of the LAST `--window` ms of a rocprofv3 rocpd database (kernel-trace): start offset, duration,
gap to the previous dispatch and the kernel name -- shows what a forward pass is made of, launch by launch, and how much
of it is idle.  Usage: rocpd_timeline.py results.db out.txt [--last N dispatches]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    n_last = int(sys.argv[3]) if len(sys.argv) > 3 else 800
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[-n_last:]
    t0 = rows[0][1]
    out = []
    prev_end = t0
    busy = 0
    agg = {}
    for name, st, en in rows:
        short = name.split("(")[0][-70:]
        out.append("%10.1f %8.1f %7.1f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, short))
        busy += en - st
        a = agg.setdefault(short, [0, 0])
        a[0] += 1
        a[1] += en - st
        prev_end = max(prev_end, en)
    span = prev_end - t0
    head = ["span %.3f ms, busy %.3f ms (%.1f %%), %d dispatches" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(rows))]
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        head.append("%8.3f ms %5d x %7.1f us  %s" % (t / 1e6, c, t / c / 1e3, k))
    head.append("")
    head.append("  start_us   dur_us  gap_us  kernel")
    open(sys.argv[2], "w").write("\n".join(head + out) + "\n")


if __name__ == "__main__":
    main()
