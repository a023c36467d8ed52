#!/usr/bin/env python
"""Does the operand layout of the oracle's clustering call change the committed fixtures?  ("what's weak" 2.)

Until round 4 oracle/glue_oracle.clustering_features clustered `features[j].reshape(C,-1).t().contiguous()`; the reference
multiplies the STRIDED transpose `torch.transpose(features[j].view(C,-1), 0, 1)` (lib/fcn/test_dataset.py:54-55) — same
values, possibly another BLAS kernel / summation order.  The oracle now uses the reference's view.  This script re-runs
the oracle's whole two-stage path (same container, same 4 torch threads as tests/golden/make_bench_oracle.py) on the
given bench frames and compares both label maps with tests/golden/bench_oracle/*.npz, pixel for pixel (identical ids).

    python tests/golden/check_bench_oracle_layout.py OUT.json FRAME [FRAME ...]      (a-b = range)
"""
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    out_path = sys.argv[1]
    frames = []
    for a in sys.argv[2:]:
        if "-" in a:
            lo, hi = a.split("-")
            frames += list(range(int(lo), int(hi)))
        else:
            frames.append(int(a))
    torch.set_num_threads(int(os.environ.get("UOC_ORACLE_THREADS", "4")))
    from oracle import backbone_oracle as BO, glue_oracle as GO
    from unseenobjectclustering_amd import runner, synth
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    net = lambda img, label, depth: BO.segnet_forward(sd, img, depth)
    fixture = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "bench_oracle", "frames_*.npz"))):
        z = np.load(path)
        first, s1, fin = int(z["first"]), z["stage1"], z["final"]      # an NpzFile decompresses the whole array on every access
        for i in range(len(fin)):
            if first + i in frames:
                fixture[first + i] = (s1[i].copy(), fin[i].copy())
    rows, t0 = [], time.time()
    for g in frames:
        s = 10_000 + g
        fr = synth.palette_frame(s, 480, 640, 5 + s % 3)
        out, refined = GO.test_sample(torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"]), net, net,
                                      np.random.RandomState(runner.frame_rng_seed(g)))
        m = (refined if refined is not None else out)[0].numpy().astype(np.uint8)
        d1 = int((out[0].numpy().astype(np.uint8) != fixture[g][0]).sum())
        dF = int((m != fixture[g][1]).sum())
        rows.append({"frame": g, "stage1_pixels_differ": d1, "final_pixels_differ": dF})
        print(rows[-1], f"{time.time() - t0:.0f}s", flush=True)
        json.dump({"threads": torch.get_num_threads(), "frames": len(rows),
                   "frames_identical": sum(r["stage1_pixels_differ"] == 0 and r["final_pixels_differ"] == 0 for r in rows),
                   "rows": rows}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
