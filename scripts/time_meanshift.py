"""Dev helper: per-stage timing of the clustering kernels with torch.cuda events (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from unseenobjectclustering_amd import synth, _native
from unseenobjectclustering_amd.utils import mean_shift as MS

dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
X, lab = synth.embedding_field(1, H, W, 64, 7, 0.05)
Xd = torch.from_numpy(X).to(dev)[None].repeat(B, 1, 1).contiguous()
n, m = X.shape[0], 100
L = _native.lib()
ws = MS._workspace(dev, L.uoc_ms_workspace_bytes(B, n, m))
first = torch.full((B,), 71530 % n, dtype=torch.int32, device=dev)
seeds = torch.empty((B, m, 64), device=dev); idx = torch.empty((B, m), dtype=torch.int32, device=dev)
sl = torch.empty((B, m), dtype=torch.int32, device=dev); nu = torch.empty((B,), dtype=torch.int32, device=dev)
labels = torch.empty((B, n), dtype=torch.int32, device=dev)
st = _native.stream_ptr(dev)
P = _native.ptr

def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def fps(): _native.check(L.uoc_ms_select_seeds(P(Xd), B, n, m, P(first), P(seeds), P(idx), P(ws), ws.numel(), st), "fps")
Z = None
def hc():
    global Z
    Z = seeds.clone()
    _native.check(L.uoc_ms_hill_climb(P(Xd), B, n, P(Z), m, 20.0, 10, P(ws), ws.numel(), st), "hc")
def cc(): _native.check(L.uoc_ms_seed_components(P(Z), B, m, 0.04, P(sl), P(nu), st), "cc")
def asg(): _native.check(L.uoc_ms_assign(P(Xd), B, n, P(Z), P(sl), P(nu), m, P(labels), None, P(ws), ws.numel(), st), "assign")
def full(): MS.cluster_batch(Xd, [71530 % n] * B, 20.0, 100, 10, 0.04)

r = dict(fps_ms=t(fps), hc_ms=t(hc), cc_ms=t(cc), assign_ms=t(asg), full_ms=t(full))
bytes_fps = 99 * (n * 64 * 4 + 2 * n * 4) * B
bytes_hc = 10 * (n * 64 * 4) * B
flops_hc = 10 * (4 * m * n * 64) * B
print({k: round(v, 4) for k, v in r.items()})
print("fps GB/s %.0f | hc GB/s %.0f TFLOP/s %.1f | clusters %d" % (bytes_fps / r["fps_ms"] / 1e6, bytes_hc / r["hc_ms"] / 1e6,
      flops_hc / r["hc_ms"] / 1e9, int(labels.max()) + 1))
