"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` reports: calls, total / average / min / max duration, share.

    python scripts/rocpd_stats.py gpurun_out/prof_r01/bench_results.db [tail_fraction] > profiles/r01_kernel_stats.md
tail_fraction (default 1 = whole trace): keep only the dispatches that start in the last fraction of the trace by time —
the benchmark's timed region and profiled pass, without the set-up (tuner launches, single-frame first-use runs)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main(path, frac=1.0):
    db = sqlite3.connect(path)
    t0, t1 = db.execute("select min(start), max(end) from kernels").fetchone()
    w0 = t1 - (t1 - t0) * frac
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels where start >= ? group by name "
                      "order by sum(duration) desc", (w0,)).fetchall()
    if frac < 1.0:
        print(f"window: last {frac:.0%} of the trace by time ({(t1 - w0) / 1e6:.1f} ms)\n")
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % GPU time | vgpr | agpr | lds B |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for n, c, s, a, mn, mx, v, ag, lds in rows:
        print(f"| `{short(n)}` | {c} | {s / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * s / total:.2f} | {v} | {ag} | {lds} |")
    print(f"\ntotal kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
