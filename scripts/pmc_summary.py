"""Per-kernel averages of every counter found in the rocprofv3 --pmc passes under a directory.

    python scripts/pmc_summary.py <dir> [kernel-substring ...] > profiles/r02_pmc_<what>.md
Each counter is averaged over the dispatches of a kernel (after the demangled name is cut at the first '(');
`dur us` is the average dispatch duration in that pass (counter passes serialise kernels, so it is the kernel alone)."""
import collections, csv, glob, os, re, sys


def main(d, filters):
    val = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        seen = set()
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:80]
            if filters and not any(s in name for s in filters):
                continue
            val[name][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[name][r["Counter_Name"]] += 1
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name in sorted(val, key=lambda n: -sum(dur[n])):
        print(f"### `{name}` — {len(dur[name])} dispatches, avg {sum(dur[name]) / len(dur[name]):.1f} us\n")
        print("| counter | avg per dispatch |\n|---|---:|")
        for c in sorted(val[name]):
            print(f"| {c} | {val[name][c] / cnt[name][c]:,.0f} |")
        print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
