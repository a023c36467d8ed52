"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` reports: calls, total / average / min / max duration, share.

    python scripts/rocpd_stats.py gpurun_out/prof_r01/bench_results.db > profiles/r01_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name "
                      "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % GPU time | vgpr | agpr | lds B |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for n, c, s, a, mn, mx, v, ag, lds in rows:
        print(f"| `{short(n)}` | {c} | {s / 1e6:.3f} | {a / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * s / total:.2f} | {v} | {ag} | {lds} |")
    print(f"\ntotal kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
