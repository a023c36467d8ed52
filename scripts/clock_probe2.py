"""Dev helper: sustained shader clock / power (rocm-smi) while looping the hill-climb or the Winograd kernel."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from unseenobjectclustering_amd import _native, synth
from unseenobjectclustering_amd.utils import mean_shift as MS
dev = torch.device("cuda:0"); L = _native.lib(); P = _native.ptr
mode = sys.argv[1] if len(sys.argv) > 1 else "hc"
st = _native.stream_ptr(dev)
if mode == "hc":
    X, _ = synth.embedding_field(1, 480, 640, 64, 7, 0.05)
    Xd = torch.from_numpy(X).to(dev)[None].contiguous(); n, m = X.shape[0], 100
    ws = MS._workspace(dev, L.uoc_ms_workspace_bytes(1, n, m))
    Z = Xd[0, :m].clone().contiguous()
    run = lambda: L.uoc_ms_hill_climb(P(Xd), 1, n, P(Z), m, 20.0, 10, P(ws), ws.numel(), st)
    per = 10
else:
    os.environ["UOC_CONV_WINOGRAD"] = "1"
    G, B, H, W, C, dil = 2, 1, 60, 80, 512, 4
    x = torch.randn(G, B, H, W, C, device=dev); w = torch.randn(G, 9, C, C, device=dev) * 0.02
    b = torch.randn(G, C, device=dev); out = torch.empty(G, B, H, W, C, device=dev)
    run = lambda: L.uoc_conv2d_nhwc(P(x), P(w), P(b), None, P(out), G, B, H, W, C, C, 3, 1, dil, dil, 1, st)
    per = 1
samples = []
def sample():
    for _ in range(6):
        time.sleep(0.4)
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        samples.append([l.strip()[-60:] for l in r.splitlines() if "sclk" in l or "ower" in l])
t = threading.Thread(target=sample); t.start()
t0 = time.time(); k = 0
while time.time() - t0 < 3.0:
    for _ in range(20): run()
    torch.cuda.synchronize(); k += 20 * per
dt = time.time() - t0; t.join()
print(mode, "%.1f us per kernel-iteration" % (dt / k * 1e6))
for s in samples[1:5]: print(s)
