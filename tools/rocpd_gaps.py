#!/usr/bin/env python3
"""Idle / concurrency accounting of the LAST `window_ms` of a rocprofv3 rocpd database (kernel-trace): how much of the
window no kernel runs at all, how much exactly one / two / three run, which kernels run ALONE for how long (the launches
whose own efficiency bounds the step), and per queue the busy share.
Usage: rocpd_gaps.py results.db out.txt [window_ms | adamw:i:j]   (adamw:i:j = from the start of the i-th adamw_kernel launch to
the start of the j-th: whole optimizer steps of bench.py --mode distill, e.g. the hipGraph-replayed ones)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    win_spec = sys.argv[3] if len(sys.argv) > 3 else "100"
    win = float(win_spec) if not win_spec.startswith("adamw:") else 0.0
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    sel = "name, start, end" + (", %s" % qcol if qcol else ", 0") + (", %s" % gcol if gcol else ", 0")
    rows = db.execute("select %s from kernels order by start" % sel).fetchall()
    t_end = max(r[2] for r in rows)
    t_beg = t_end - int(win * 1e6)
    if isinstance(win_spec, str) and win_spec.startswith("adamw:"):
        _, i, j = win_spec.split(":")
        # one mark per optimizer step: the AdamW launch of the three-launch tail or the one-pass tail (round 6: adamw_ema2_zero_kernel)
        marks = [r[1] for r in rows if "adamw4_kernel" in r[0] or "adamw_ema2_zero_kernel" in r[0] or r[0].startswith("adamw_kernel")]
        print("adamw launches:", len(marks))
        t_beg, t_end = marks[int(i)], marks[int(j)]
        rows = [r for r in rows if r[1] < t_end]
    rows = [r for r in rows if r[2] > t_beg]
    ev = []
    for i, (name, st, en, q, g) in enumerate(rows):
        ev.append((max(st, t_beg), 1, i))
        ev.append((min(en, t_end), -1, i))
    ev.sort()
    active = set()
    conc = {}
    alone = {}
    idle = {}          # (kernel that ended, kernel that started) -> [count, total idle ns]
    last_ended = None
    prev = t_beg
    for t, d, i in ev:
        dt = t - prev
        if dt > 0:
            conc[len(active)] = conc.get(len(active), 0) + dt
            if len(active) == 1:
                k = rows[next(iter(active))][0].split("(")[0][-60:]
                a = alone.setdefault(k, [0, 0])
                a[0] += dt
            if not active and d > 0 and last_ended is not None:
                key = (rows[last_ended][0].split("(")[0][-44:], rows[i][0].split("(")[0][-44:])
                g_ = idle.setdefault(key, [0, 0])
                g_[0] += 1
                g_[1] += dt
        prev = t
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
            last_ended = i
    span = t_end - t_beg
    out = ["window %.1f ms, %d dispatches, columns of `kernels`: %s" % (span / 1e6, len(rows), ",".join(cols)), ""]
    for c in sorted(conc):
        out.append("%d kernels running: %8.3f ms (%5.1f %%)" % (c, conc[c] / 1e6, 100.0 * conc[c] / span))
    out.append("")
    per_q = {}
    for name, st, en, q, g in rows:
        per_q[q] = per_q.get(q, 0) + en - max(st, t_beg)
    for q, t in sorted(per_q.items(), key=lambda kv: -kv[1]):
        out.append("queue %s: sum of kernel time %8.3f ms (%5.1f %% of the window)" % (q, t / 1e6, 100.0 * t / span))
    out.append("")
    out.append("kernels running ALONE (no other kernel on the device), by total time:")
    for k, (t, _) in sorted(alone.items(), key=lambda kv: -kv[1][0])[:40]:
        out.append("%8.3f ms  %s" % (t / 1e6, k))
    out.append("")
    out.append("idle gaps (no kernel on the device) by (kernel that ended -> kernel that started): count, total ms, avg us")
    hist = {}
    for (a_, b_), (n_, t_) in idle.items():
        pass
    for (a_, b_), (n_, t_) in sorted(idle.items(), key=lambda kv: -kv[1][1])[:40]:
        out.append("%6d x %8.3f ms %6.1f us  %s -> %s" % (n_, t_ / 1e6, t_ / n_ / 1e3, a_, b_))
    out.append("idle gaps in total: %d, %.3f ms" % (sum(v[0] for v in idle.values()), sum(v[1] for v in idle.values()) / 1e6))
    out.append("")
    out.append("dispatches in the window by kernel (count, total ms, avg us, median gap to the previous dispatch END on the same queue):")
    agg = {}
    last_end = {}
    for name, st, en, q, g in rows:
        if st < t_beg:
            last_end[q] = max(last_end.get(q, 0), en)
            continue
        k = name.split("(")[0][-60:]
        a = agg.setdefault(k, [0, 0, []])
        a[0] += 1
        a[1] += en - st
        if q in last_end:
            a[2].append(st - last_end[q])
        last_end[q] = max(last_end.get(q, 0), en)
    tot_n = sum(a[0] for a in agg.values())
    out.append("total dispatches %d" % tot_n)
    for k, (n, t, gaps) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
        gaps.sort()
        med = gaps[len(gaps) // 2] / 1e3 if gaps else 0.0
        out.append("%6d x %8.3f ms %8.1f us  gap %6.1f us  %s" % (n, t / 1e6, t / n / 1e3, med, k))
    out.append("")
    out.append("the same by total time:")
    for k, (n, t, gaps) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        out.append("%6d x %8.3f ms %8.1f us  %s" % (n, t / 1e6, t / n / 1e3, k))
    open(sys.argv[2], "w").write("\n".join(out) + "\n")
    print("\n".join(out[:60]))


if __name__ == "__main__":
    main()
