"""cProfile of the host side of the one-frame-at-a-time leg: where does Python spend its time per frame?"""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from unseenobjectclustering_amd import networks, synth, runner
from unseenobjectclustering_amd.fcn.config import cfg
dev = torch.device("cuda:0")
cfg.device = dev
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
net_crop = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
samples = []
for g in range(12):
    fr = synth.palette_frame(10000 + g)
    samples.append(dict(image_color=torch.from_numpy(fr["image_color"]).to(dev), depth=torch.from_numpy(fr["depth"]).to(dev)))
fn = runner.two_stage_frame_fn(samples, net, net_crop)
for rep in range(2):
    for g in range(12):
        np.random.seed(runner.frame_rng_seed(g)); fn(g).to(torch.uint8).cpu()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for rep in range(3):
    for g in range(12):
        np.random.seed(runner.frame_rng_seed(g)); fn(g).to(torch.uint8).cpu()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()))
