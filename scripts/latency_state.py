"""Why does bench.py's latency leg read 8.05 ms when scripts/latency_leg.py reads 7.76?  One process: the leg cold, then after
a pipelined throughput run (the state the bench leaves behind), then after resetting pieces of that state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from unseenobjectclustering_amd import networks, synth, runner, _native
from unseenobjectclustering_amd.fcn.config import cfg
dev = torch.device("cuda:0")
cfg.device = dev
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
net_crop = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
NF = int(os.environ.get("NF", "20"))
samples = []
for g in range(NF):
    fr = synth.palette_frame(10000 + g)
    samples.append(dict(image_color=torch.from_numpy(fr["image_color"]).to(dev), depth=torch.from_numpy(fr["depth"]).to(dev)))
fn = runner.two_stage_frame_fn(samples, net, net_crop)


def leg(tag, n=12, reps=3):
    for g in range(2):
        np.random.seed(runner.frame_rng_seed(g)); fn(g).cpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for g in range(n):
            np.random.seed(runner.frame_rng_seed(g)); fn(g).to(torch.uint8).cpu()
    torch.cuda.synchronize()
    print(f"{tag}: {1e3 * (time.perf_counter() - t0) / (n * reps):.3f} ms per frame", flush=True)


leg("cold")
leg("cold again")
t0 = time.perf_counter()
for _ in range(6):
    runner.run_sharded(NF, fn, 480, 640, dev, 0, 1, False, inflight=3).cpu()
torch.cuda.synchronize()
print(f"pipelined: {6 * NF / (time.perf_counter() - t0):.1f} frames/s", flush=True)
leg("right after the pipelined run")
_native.lib().uoc_ms_set_stream_ordering(0)
leg("stream ordering off")
time.sleep(3.0)
leg("after 3 s idle")
torch.cuda.empty_cache()
leg("after empty_cache")
