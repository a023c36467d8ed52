#!/usr/bin/env python
"""Oracle label maps of the BENCHMARK frames (bench.py's global frames: synth.palette_frame(10000 + g, 480, 640,
5 + (10000 + g) % 3), first-seed RNG runner.frame_rng_seed(g), calibrated weights), computed on the CPU by
oracle/glue_oracle.test_sample — the torch-CPU restatement that tests/golden/*.npz pin to the reference itself.

    python tests/golden/make_bench_oracle.py LO HI [THREADS]     -> tests/golden/bench_oracle/frames_LO_HI.npz

Each file holds, per frame, the stage-1 label map after the depth filter and the refined (final) map as uint8
[n, 480, 640] (compressed: a few KB per frame).  tests/test_headline_parity_gpu.py compares the HIP path with them
(end-to-end mismatch histogram) and rebuilds the oracle's crops from the stage-1 maps, so the GPU box never has to run
the oracle's clustering — only its two network passes per frame.  ~40 s per frame on 8 cores."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def bench_frame(g):
    from unseenobjectclustering_amd import synth
    s = 10_000 + g
    fr = synth.palette_frame(s, 480, 640, 5 + s % 3)
    return torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"])


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    torch.set_num_threads(int(sys.argv[3]) if len(sys.argv) > 3 else max(1, os.cpu_count() or 1))
    from oracle import backbone_oracle as BO, glue_oracle as GO
    from unseenobjectclustering_amd import runner, synth
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    net = lambda img, label, depth: BO.segnet_forward(sd, img, depth)
    stage1, final = [], []
    t0 = time.time()
    for g in range(lo, hi):
        img, dep = bench_frame(g)
        out, refined = GO.test_sample(img, dep, net, net, np.random.RandomState(runner.frame_rng_seed(g)))
        m = (refined if refined is not None else out)[0].numpy()
        assert out.max() < 256 and m.max() < 256
        stage1.append(out[0].numpy().astype(np.uint8))
        final.append(m.astype(np.uint8))
        print(f"frame {g}: {int(out.max())} stage-1 labels, {int(m.max())} objects, {time.time() - t0:.0f}s", flush=True)
    path = os.path.join(ROOT, "tests", "golden", "bench_oracle", f"frames_{lo:04d}_{hi:04d}.npz")
    np.savez_compressed(path, first=np.int64(lo), stage1=np.stack(stage1), final=np.stack(final))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
