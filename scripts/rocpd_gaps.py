"""Idle-gap analysis of a rocprofv3 kernel trace (rocpd sqlite) of a ONE-STREAM run (bench.py --inflight 1
--frames-per-launch 1: the latency leg's schedule): where does the GPU wait for the host?  Lists the share of the window
with no kernel executing and the idle intervals grouped by (kernel before -> kernel after), largest total first.

    python scripts/rocpd_gaps.py <results.db> [tail_fraction=0.5] [frames_in_window] > profiles/r05_latency_gaps.md"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))
    return re.sub(r"^uoc::", "", name)[:48]


def main(path, frac=0.5, frames=0):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    w0 = t1 - (t1 - t0) * frac
    rows = [r for r in rows if r[1] >= w0]
    span = (max(r[2] for r in rows) - rows[0][1]) / 1e3
    busy_until, prev = rows[0][2], rows[0][0]
    gaps, busy = defaultdict(lambda: [0, 0.0, 0.0]), 0.0
    seg_start = rows[0][1]
    for name, s, e in rows[1:]:
        if s > busy_until:
            busy += (busy_until - seg_start) / 1e3
            g = gaps[(short(prev), short(name))]
            g[0] += 1
            g[1] += (s - busy_until) / 1e3
            g[2] = max(g[2], (s - busy_until) / 1e3)
            seg_start = s
        if e >= busy_until:
            busy_until, prev = e, name
    busy += (busy_until - seg_start) / 1e3
    idle = span - busy
    print(f"window: last {frac:.0%} of the trace = {span / 1e3:.2f} ms, {len(rows)} dispatches; a kernel executes {busy / 1e3:.2f} ms "
          f"({100 * busy / span:.1f} %), idle {idle / 1e3:.2f} ms ({100 * idle / span:.1f} %)")
    if frames:
        print(f"per frame ({frames} frames in the window): {span / frames / 1e3:.3f} ms wall, {busy / frames / 1e3:.3f} ms busy, "
              f"{idle / frames / 1e3:.3f} ms idle, {len(rows) / frames:.0f} dispatches")
    print("\n| idle between (kernel before -> kernel after) | gaps | total us | avg us | max us | share of idle |\n|---|---:|---:|---:|---:|---:|")
    for (a, b), (c, tot, mx) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"| `{a}` -> `{b}` | {c} | {tot:.0f} | {tot / c:.1f} | {mx:.1f} | {100 * tot / max(idle, 1e-9):.1f} % |")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
