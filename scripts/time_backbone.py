"""Dev helper: time the native backbone forward on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from unseenobjectclustering_amd import synth, networks
dev = torch.device("cuda:0")
H, W, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 640, 1)
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synthetic_state_dict(1).items()}
net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
fr = synth.rgbd_frame(1, H, W, 5)
img = torch.from_numpy(fr["image_color"]).to(dev).repeat(B, 1, 1, 1)
dep = torch.from_numpy(fr["depth"]).to(dev).repeat(B, 1, 1, 1)
for _ in range(3): f = net(img, None, dep)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps): f = net(img, None, dep)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
gf = 2 * 212.2 * (H * W) / (480 * 640) * B
print(f"backbone {H}x{W} B={B}: {ms:.3f} ms  -> {gf / ms:.1f} TFLOP/s fp32")
