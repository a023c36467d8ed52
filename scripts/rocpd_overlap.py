"""Overlap analysis of a rocprofv3 kernel trace (rocpd sqlite): how much of the timed window has 0 / 1 / >=2 kernels
executing, per-queue busy time, and the per-kernel average duration inside the window (compare with the
one-frame-at-a-time profile to see what sharing the chip costs each kernel).

    python scripts/rocpd_overlap.py <bench_results.db> [tail_fraction=0.5] > profiles/r02_overlap.md
The window is the last `tail_fraction` of the trace by time (the benchmark's timed region + later legs; set-up and
tuning launches are at the front)."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))[:70]


def main(path, frac=0.5):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    w0 = t1 - (t1 - t0) * frac
    rows = [r for r in rows if r[1] >= w0]
    span = (max(r[2] for r in rows) - rows[0][1]) / 1e6
    ev = []
    for _, s, e, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    level, last, hist = 0, ev[0][0], defaultdict(int)
    for t, d in ev:
        hist[min(level, 3)] += t - last
        last = t
        level += d
    print(f"window: last {frac:.0%} of the trace = {span:.2f} ms, {len(rows)} dispatches\n")
    print("| kernels executing | ms | share |\n|---|---:|---:|")
    for k in sorted(hist):
        print(f"| {k if k < 3 else '>=3'} | {hist[k] / 1e6:.2f} | {100 * hist[k] / 1e6 / span:.1f} % |")
    q = defaultdict(int)
    for _, s, e, qid, sid in rows:
        q[(qid, sid)] += e - s
    print("\n| queue, stream | busy ms | share of window |\n|---|---:|---:|")
    for k, v in sorted(q.items(), key=lambda kv: -kv[1]):
        print(f"| {k} | {v / 1e6:.2f} | {100 * v / 1e6 / span:.1f} % |")
    per = defaultdict(list)
    for n, s, e, _, _ in rows:
        per[short(n)].append(e - s)
    tot = sum(sum(v) for v in per.values())
    print("\n| kernel | calls | avg us | total ms | share of kernel time |\n|---|---:|---:|---:|---:|")
    for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:24]:
        print(f"| `{n}` | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {sum(v) / 1e6:.2f} | {100 * sum(v) / tot:.1f} % |")
    print(f"\nsummed kernel time {tot / 1e6:.2f} ms = {tot / 1e6 / span:.2f} x the window")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
