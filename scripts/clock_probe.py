"""Dev helper: sustained shader clock / power while one kernel loops (rocm-smi sampled in-process)."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unseenobjectclustering_amd import _native
dev = torch.device("cuda:0"); L = _native.lib(); P = _native.ptr
G, B, H, W, Cin, Cout, K, dil = 2, 1, 60, 80, 512, 512, 3, 4
x = torch.randn(G, B, H, W, Cin, device=dev); w = torch.randn(G, 9, Cout, Cin, device=dev) * 0.02
b = torch.randn(G, Cout, device=dev); out = torch.empty(G, B, H, W, Cout, device=dev)
st = _native.stream_ptr(dev)
samples = []
def sample():
    for _ in range(6):
        time.sleep(0.4)
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        samples.append([l.strip() for l in r.splitlines() if "sclk" in l or "Power" in l or "power" in l])
t = threading.Thread(target=sample); t.start()
t0 = time.time(); n = 0
while time.time() - t0 < 3.0:
    for _ in range(50):
        L.uoc_conv2d_nhwc(P(x), P(w), P(b), None, P(out), G, B, H, W, Cin, Cout, K, 1, dil, dil, 1, st)
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
t.join()
print("conv layer4 loop: %.1f us/launch, %.1f TFLOP/s" % (dt / n * 1e6, 2.0 * G * B * H * W * Cout * Cin * 9 / (dt / n) / 1e12))
for s in samples[1:5]: print(s)
