#!/usr/bin/env python
"""Merges sub-block files of tests/golden/make_bench_margins.py (e.g. 8-frame files made in parallel) into 64-frame files
under tests/golden/bench_margins/.   python tests/golden/merge_bench_margins.py <dir with frames_*.npz> LO HI"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KEYS = ("idx1", "val1", "idxF", "valF", "rois")


def main():
    src, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    frames = {}
    for path in sorted(glob.glob(os.path.join(src, "frames_*.npz"))):
        z = np.load(path)
        first = int(z["first"])
        for i in range(int(z["count"])):
            rec = {k: z[k][z["off_" + k][i]:z["off_" + k][i + 1]] for k in KEYS}
            rec["slack"] = z["slack"][i]
            d = z["differs_from_bench_oracle"][i]
            rec["differs"] = np.asarray(list(d) + [0] * (3 - len(d)), np.int32)
            rec["tau_store"] = z["tau_store"]
            frames[first + i] = rec
    for b in range(lo, hi, 64):
        idx = list(range(b, min(hi, b + 64)))
        missing = [g for g in idx if g not in frames]
        assert not missing, f"frames missing: {missing[:8]}"
        acc = {k: [frames[g][k] for g in idx] for k in KEYS}
        off = {k: np.concatenate([[0], np.cumsum([len(v) for v in acc[k]])]).astype(np.int64) for k in KEYS}
        path = os.path.join(ROOT, "tests", "golden", "bench_margins", f"frames_{idx[0]:04d}_{idx[-1] + 1:04d}.npz")
        np.savez_compressed(path, first=np.int64(idx[0]), count=np.int64(len(idx)), tau_store=frames[idx[0]]["tau_store"],
                            slack=np.stack([frames[g]["slack"] for g in idx]).astype(np.float32),
                            differs_from_bench_oracle=np.stack([frames[g]["differs"] for g in idx]),
                            **{k: np.concatenate(acc[k]) for k in KEYS}, **{"off_" + k: off[k] for k in KEYS})
        print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
