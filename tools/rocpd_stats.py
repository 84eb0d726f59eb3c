#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd database (kernel-trace) into the per-kernel stats table that
`--stats` prints: name, calls, total ms, average us, share.  Usage: rocpd_stats.py results.db [out.md] [header text]
The header (what was run, how many steps the table holds) is written above the table together with counts the table
itself supplies: optimizer steps = adamw launches, backward passes = snr_mse_grad launches, vocoder passes = wav_finalize
launches -- so that a per-step figure can be recomputed from the file alone."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, calls, tot, avg, mn, mx in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.2f |" % (short, calls, tot / 1e6, avg / 1e3, mn / 1e3,
                                                                        mx / 1e3, 100.0 * tot / total))
    text = "\n".join(lines) + "\n\ntotal kernel time %.3f ms over %d dispatches\n" % (total / 1e6, sum(r[1] for r in rows))
    def calls_of(sub):
        return sum(r[1] for r in rows if sub in r[0])
    head = []
    if len(sys.argv) > 3:
        head.append(sys.argv[3])
    marks = [("optimizer steps (adamw launches)", calls_of("adamw")), ("backward passes (snr_mse_grad launches)", calls_of("snr_mse_grad")),
             ("generation steps (wav_finalize / wav_minmax launches)", max(calls_of("wav_finalize"), calls_of("wav_minmax"), calls_of("wav_extrema")))]
    head.append("step marks inside this table: " + "; ".join("%s = %d" % m for m in marks if m[1]))
    text = "\n".join(head) + "\n\n" + text
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
