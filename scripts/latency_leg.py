"""One frame at a time on one stream (BASELINE configs[3] read literally, bench.py's `latency` leg) in isolation: resident
frames, full two-stage path, host waiting for each uint8 label map.  --graph 1 replays hipGraphs (fcn/graph_replay.py),
0 runs the eager FrameJob.  Run it under `rocprofv3 --kernel-trace` for scripts/rocpd_gaps.py / rocpd_stats.py.
    python scripts/latency_leg.py --frames 12 --reps 4 --graph 1"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--graph", type=int, default=1)
args = ap.parse_args()
from unseenobjectclustering_amd import networks, synth, runner
from unseenobjectclustering_amd.fcn.config import cfg
dev = torch.device("cuda:0")
cfg.device = dev
cfg.TEST.GRAPH_REPLAY = bool(args.graph)
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
net_crop = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
samples = []
for g in range(args.frames):
    fr = synth.palette_frame(10000 + g)
    samples.append(dict(image_color=torch.from_numpy(fr["image_color"]).to(dev), depth=torch.from_numpy(fr["depth"]).to(dev)))
fn = runner.two_stage_frame_fn(samples, net, net_crop)
for rep in range(2):            # first use of every ROI count (eager) + its capture
    for g in range(args.frames):
        np.random.seed(runner.frame_rng_seed(g))
        fn(g).to(torch.uint8).cpu()
torch.cuda.synchronize()
per = []
t0 = time.perf_counter()
for rep in range(args.reps):
    for g in range(args.frames):
        t1 = time.perf_counter()
        np.random.seed(runner.frame_rng_seed(g))
        fn(g).to(torch.uint8).cpu()
        per.append(time.perf_counter() - t1)
torch.cuda.synchronize()
el = time.perf_counter() - t0
n = args.reps * args.frames
per = np.array(per) * 1e3
print(f"graph={args.graph}: {1e3 * el / n:.3f} ms per frame ({n / el:.1f} frames/s) over {n} frames; per frame min {per.min():.3f} "
      f"median {np.median(per):.3f} max {per.max():.3f} ms; ROIs {fn.roi_counts[-args.frames:]}")
