"""Per-kernel-class and per-launch-shape time of the one-frame-at-a-time leg on bench.py's frames (HIP events of csrc/prof.hip,
every launch alone): where do the 7.5 ms of kernels per frame go at batch 1?
    python scripts/latency_by_shape.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from unseenobjectclustering_amd import networks, synth, runner, _native
from unseenobjectclustering_amd.fcn.config import cfg
dev = torch.device("cuda:0")
cfg.device = dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
net_crop = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
samples = []
for g in range(n):
    s = 10000 + g
    fr = synth.palette_frame(s, 480, 640, 5 + s % 3)          # bench.py's frames
    samples.append(dict(image_color=torch.from_numpy(fr["image_color"]).to(dev), depth=torch.from_numpy(fr["depth"]).to(dev)))
fn = runner.two_stage_frame_fn(samples, net, net_crop)
def one_pass():
    for g in range(n):
        np.random.seed(runner.frame_rng_seed(g))
        fn(g).to(torch.uint8).cpu()
for _ in range(2):
    one_pass()
torch.cuda.synchronize()
_native.prof_enable(True)
one_pass()
torch.cuda.synchronize()
rep = _native.prof_report()
_native.prof_enable(False)
tot = sum(r["total_ms"] for r in rep)
print(f"ROIs per frame {fn.roi_counts[-n:]}; kernel time {tot / n:.3f} ms per frame\n")
print("| kernel class | launches per frame | avg us | ms per frame | share |\n|---|---:|---:|---:|---:|")
for r in sorted(rep, key=lambda r: -r["total_ms"]):
    print(f"| {r['kernel']} | {r['launches'] / n:.1f} | {1e3 * r['total_ms'] / r['launches']:.1f} | {r['total_ms'] / n:.3f} | {100 * r['total_ms'] / tot:.1f} % |")
print("\n| kernel | shape tag | launches per frame | avg us | ms per frame | TFLOP/s (algorithmic) |\n|---|---|---:|---:|---:|---:|")
rows = []
for r in rep:
    for sh in r.get("shapes", []):
        rows.append((sh["total_ms"], r["kernel"], sh))
for t, k, sh in sorted(rows, key=lambda x: -x[0])[:45]:
    print(f"| {k} | {sh['tag']} | {sh['launches'] / n:.2f} | {1e3 * sh['total_ms'] / sh['launches']:.1f} | {sh['total_ms'] / n:.3f} | {sh['flops'] / max(sh['total_ms'], 1e-9) / 1e9:.1f} |")
