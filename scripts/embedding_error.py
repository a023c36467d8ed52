import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from unseenobjectclustering_amd import networks, synth
from unseenobjectclustering_amd.fcn.config import cfg
dev = torch.device("cuda:0"); cfg.device = dev
g = np.load("tests/golden/backbone.npz")
cases = {"tiny_64x64": (1, [7], 64, 64), "odd_72x104": (2, [8], 72, 104), "crops_224": (2, [4, 5], 224, 224), "full_480x640": (1, [1], 480, 640)}
for name, (ws, frames, H, W) in cases.items():
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synthetic_state_dict(ws).items()}
    net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    fr = [synth.rgbd_frame(s, H, W, 4) for s in frames]
    img = torch.from_numpy(np.concatenate([f["image_color"] for f in fr])).to(dev)
    dep = torch.from_numpy(np.concatenate([f["depth"] for f in fr])).to(dev)
    flat = net(img, None, dep).permute(0, 2, 3, 1).reshape(len(fr), -1, 64).cpu().numpy()
    if name + "/pos" in g: flat = flat[:, g[name + "/pos"]]
    print(name, "max |embedding - reference| = %.2e" % np.abs(flat - g[name + "/embed"]).max())
