#!/usr/bin/env python
"""Decision margins of the oracle on the BENCHMARK frames (the frames of tests/golden/bench_oracle/, same inputs, RNG
seeds and calibrated weights), for the margin-bounded end-to-end parity test (tests/test_headline_parity_gpu.py).

    python tests/golden/make_bench_margins.py LO HI [THREADS]     -> tests/golden/bench_margins/frames_LO_HI.npz

Per frame (oracle/margins.test_sample_with_margins): the pixels of the stage-1 map and of the final map whose
nearest-seed decision is within TAU_STORE = 2e-3 of flipping (sparse: flat index + margin), the padded ROI boxes and the
slack of the cluster-level decisions.  The maps of this run are compared with bench_oracle's (another run of the same
oracle: 4 torch threads instead of 1): `differs_from_bench_oracle` holds, per frame, the pixels that differ in the stage-1
map, in the final map, and how many of them lie beyond TAU (the oracle's last pixels depend on the thread count and the
host, profiles/r03_oracle_thread_sensitivity_gpu_box.json).  ~19 s per frame on one thread.

Provenance of the committed files (round 4): frames 0-191 and 256-383 were computed in the build container (Xeon, one torch
thread per process), frames 192-255 and 384-1023 on the GPU box's host cores (EPYC 9575F, one thread per process, 8-frame
sub-blocks through scripts/margins_on_box.sh, merged by merge_bench_margins.py).  Over all 1 024 frames this 1-thread run
differs from bench_oracle's 4-thread run on 48 frames, on 2 of them beyond TAU (a seed between two modes)."""
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    torch.set_num_threads(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    from oracle import backbone_oracle as BO, margins as M
    from unseenobjectclustering_amd import runner, synth
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    net = lambda img, label, depth: BO.segnet_forward(sd, img, depth)
    fixture = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "bench_oracle", "frames_*.npz"))):
        z = np.load(path)
        if int(z["first"]) < hi and int(z["first"]) + len(z["final"]) > lo:
            first, s1, fin = int(z["first"]), z["stage1"], z["final"]  # an NpzFile decompresses the whole array on every access
            for i in range(len(fin)):
                if lo <= first + i < hi:
                    fixture[first + i] = (s1[i].copy(), fin[i].copy())
    acc = {k: [] for k in ("idx1", "val1", "idxF", "valF", "rois")}
    off = {k: [0] for k in acc}
    slack, differs = [], []
    t0 = time.time()
    for g in range(lo, hi):
        s = 10_000 + g
        fr = synth.palette_frame(s, 480, 640, 5 + s % 3)
        out, refined, info = M.test_sample_with_margins(torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"]), net, net,
                                                        np.random.RandomState(runner.frame_rng_seed(g)))
        final = (refined if refined is not None else out)[0].numpy().astype(np.uint8)
        stage1 = out[0].numpy().astype(np.uint8)
        d1 = np.nonzero(stage1.reshape(-1) != fixture[g][0].reshape(-1))[0]
        dF = np.nonzero(final.reshape(-1) != fixture[g][1].reshape(-1))[0]
        # not an error: at a seed that sits between two modes ten kappa = 20 iterations amplify the last bit of a sum without
        # bound (bench frame 246: one crop seed converges 1.41 away in the 1-thread run from where the 4-thread run of
        # bench_oracle puts it; 12 pixels follow it) — recorded, and handled by the test's perturbation analysis
        beyond = int(np.sum(info["margin1"][d1] > M.TAU)) + int(np.sum(info["marginF"].reshape(-1)[dF] > M.TAU))
        differs.append([len(d1), len(dF), beyond])
        i1, v1 = M.sparse_below(info["margin1"])
        iF, vF = M.sparse_below(info["marginF"])
        for k, v in (("idx1", i1), ("val1", v1), ("idxF", iF), ("valF", vF), ("rois", info["rois"].astype(np.int16).reshape(-1))):
            acc[k].append(v)
            off[k].append(off[k][-1] + len(v))
        sl = info["slack"]
        slack.append([sl.get("seed_cc_stage1", np.inf), sl.get("depth_filter", np.inf), sl.get("seed_cc_crops", np.inf), sl.get("overlap", np.inf)])
        print(f"frame {g}: near-tie pixels {len(i1)} / {len(iF)}, differs from bench_oracle {differs[-1]}, {time.time() - t0:.0f}s", flush=True)
    outdir = os.environ.get("UOC_MARGINS_OUT") or os.path.join(ROOT, "tests", "golden", "bench_margins")
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, f"frames_{lo:04d}_{hi:04d}.npz")
    np.savez_compressed(path, first=np.int64(lo), count=np.int64(hi - lo), tau_store=np.float32(M.TAU_STORE),
                        slack=np.asarray(slack, np.float32), differs_from_bench_oracle=np.asarray(differs, np.int32),
                        **{k: (np.concatenate(v) if v else np.zeros(0)) for k, v in acc.items()},
                        **{"off_" + k: np.asarray(v, np.int64) for k, v in off.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
