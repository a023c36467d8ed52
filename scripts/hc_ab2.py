"""Interleaved same-process A/B of two library builds on hill-climbing launches: loads both .so files side by side."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unseenobjectclustering_amd import _native
from unseenobjectclustering_amd.utils import mean_shift as MS
dev = torch.device("cuda:0")
libs = {"shipped": _native.lib()}
for p in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(p))
    _native._declare(L) if hasattr(_native, "_declare") else None
    libs[os.path.basename(p)] = L
g = torch.Generator(device=dev).manual_seed(0)
def timeit(L, X, Z0, ws, batch, n, reps=6):
    def once():
        Z = Z0.clone()
        rc = L.uoc_ms_hill_climb(ctypes.c_void_p(X.data_ptr()), batch, n, ctypes.c_void_p(Z.data_ptr()), 100, ctypes.c_float(20.0), 10,
                                 ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    once(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps / 10      # us per iteration (hc_iter + finalize + clone share)
for batch, n in [(1, 2457600), (4, 307200), (8, 307200), (1, 307200), (28, 50176), (7, 50176)]:
    X = torch.nn.functional.normalize(torch.randn(batch, n, 64, device=dev, generator=g), dim=-1)
    Z0 = torch.nn.functional.normalize(torch.randn(batch, 100, 64, device=dev, generator=g), dim=-1)
    ws = MS._workspace(dev, libs["shipped"].uoc_ms_workspace_bytes(batch, n, 100))
    res = {k: [] for k in libs}
    for rnd in range(5):
        for k, L in libs.items():
            res[k].append(timeit(L, X, Z0, ws, batch, n))
    print(f"{batch}x{n}: " + "  ".join(f"{k}: min {min(v):.1f} med {sorted(v)[len(v)//2]:.1f}" for k, v in res.items()), flush=True)
