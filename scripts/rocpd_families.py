"""Per-frame GPU time by kernel family from a rocprofv3 kernel trace (rocpd sqlite) of a bench.py run.
    python scripts/rocpd_families.py <db> <frames in the window> [tail_fraction]"""
import re, sqlite3, sys
from collections import defaultdict

def fam(n):
    n = re.sub(r"^void ", "", n)
    for k in ("wino_gemm", "wino_input", "hc_iter", "hc_finalize", "fps_persistent", "fps_step", "assign", "head_kernel", "seed_cc",
              "conv_glds_kernel", "conv_mfma", "maxpool", "nchw3", "roi_crop", "crop_", "label_stats", "paste", "relabel"):
        if k in n:
            if k == "conv_glds_kernel":
                m = re.search(r"conv_glds_kernel<(\d+), (\d+), (\d+), (\d+), (true|false)", n)
                return "conv_stem" if m.group(5) == "true" else f"conv_glds<*,{m.group(2)}>"
            return k
    return "other"

db = sqlite3.connect(sys.argv[1])
frames = float(sys.argv[2])
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.4
rows = db.execute("select name, start, end from kernels order by start").fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
w0 = t1 - (t1 - t0) * frac
tot, cnt = defaultdict(float), defaultdict(int)
for n, s, e in rows:
    if s >= w0:
        tot[fam(n)] += (e - s) / 1e3
        cnt[fam(n)] += 1
# frames in window estimated from head_kernel launches (2 per frame for single-frame launches; report both)
print(f"window {frac:.0%} of trace; head launches {cnt['head_kernel']}")
scale = frames
all_us = sum(tot.values())
print("| family | launches | total ms | share |\n|---|---:|---:|---:|")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"| {k} | {cnt[k]} | {v / 1e3:.2f} | {100 * v / all_us:.1f} % |")
print(f"total kernel time {all_us / 1e3:.2f} ms")
