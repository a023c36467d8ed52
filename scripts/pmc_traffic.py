"""Turn the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh into profiles/<round>_pmc_traffic.md and
profiles/traffic.json (HBM-side bytes per launch per bench kernel class, read by bench.py).

    python scripts/pmc_traffic.py gpurun_out/prof_v7 profiles/r01_pmc_traffic.md [tail_fraction]

tail_fraction (default 1): keep only the last fraction of the dispatches (by dispatch id) — the launch sets of the warm-up
and timed region, without the single-frame first-use runs of the set-up.

FETCH_SIZE is doubled for gfx950 (MI355X_MICROARCH.md, HBM section: 128-B requests are tallied as 64 B);
WRITE_SIZE is used as reported; both counters are in KB and include Infinity-Cache hits."""
import collections, csv, glob, json, os, re, sys


def load(d, counter, frac=1.0):
    acc = collections.defaultdict(float)
    calls = collections.Counter()
    seen = set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        ids = [int(r["Dispatch_Id"]) for r in rows]
        first = max(ids) - (max(ids) - min(ids)) * frac if ids else 0
        for r in rows:
            if r["Counter_Name"] != counter or int(r["Dispatch_Id"]) < first:
                continue
            name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))
            acc[name] += float(r["Counter_Value"])
            key = (name, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                calls[name] += 1
    return acc, calls


def klass(name):
    m = re.search(r"conv_glds_kernel<(\d+), (\d+), (\d+), (\d+), (true|false)", name)
    if m:
        if m.group(5) == "true":
            return "conv_stem"
        return "conv_glds_%sx%s" % ("160" if m.group(3) == "2" else "80", m.group(2))
    for k, v in (("wino4_gemm", "wino4_gemm"), ("wino4_input", "wino4_input"), ("wino4_output", "wino4_output"),
                 ("wino_gemm", "wino_gemm"), ("wino_input", "wino_input"), ("hc_iter", "hc_iter"),
                 ("hc_finalize", "hc_finalize"), ("fps_", "fps_step"), ("assign_kernel", "assign"), ("head_", "head")):
        if k in name:
            return v
    return None


def main(src, out_md, frac=1.0):
    fetch, calls = load(os.path.join(src, "fetch"), "FETCH_SIZE", frac)
    write, _ = load(os.path.join(src, "write"), "WRITE_SIZE", frac)
    rows = []
    for name in fetch:
        c = calls[name]
        rd = 2.0 * fetch[name] * 1024 / c
        wr = write.get(name, 0.0) * 1024 / c
        rows.append((name, c, rd, wr))
    rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
    lines = ["# HBM-side traffic per launch from PMC counters (%s)" % os.path.basename(src.rstrip("/")), "",
             "Commands: `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python bench.py --steps 8 "
             "--warmup 4 --inflight 1 --cpu-frames 0 --profile-steps 0` (one launch set of 4 frames at a time) and the same "
             "with `--pmc WRITE_SIZE` (separate passes; scripts/profile_round.sh); last %d %% of the dispatches." % round(100 * frac), "FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950 "
             "(128-B requests tallied at 64 B); WRITE_SIZE as reported (uncalibrated); counters are in KB. "
             "Infinity-Cache hits are counted, so this is an upper bound on DRAM bytes.", "",
             "| kernel | launches | read MB / launch | written MB / launch | total MB / launch |", "|---|---:|---:|---:|---:|"]
    per_class = collections.defaultdict(lambda: [0.0, 0])
    for name, c, rd, wr in rows:
        if (rd + wr) * c < 1e6:
            continue
        lines.append("| `%s` | %d | %.1f | %.1f | %.1f |" % (name[:90], c, rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
        k = klass(name)
        if k:
            per_class[k][0] += (rd + wr) * c
            per_class[k][1] += c
    traffic = {k: int(v[0] / v[1]) for k, v in sorted(per_class.items())}
    lines += ["", "Per bench kernel class (call-weighted mean over the kernel family), bytes/launch -> `profiles/traffic.json`:", ""]
    lines += ["* %s: %.1f MB" % (k, v / 1e6) for k, v in traffic.items()]
    open(out_md, "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(os.path.join(os.path.dirname(out_md), "traffic.json"), "w"), indent=1)
    print("\n".join(lines[-len(traffic) - 2:]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
