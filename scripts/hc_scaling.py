"""Where does a hill-climbing launch's time go — per-tile work or a fixed cost per launch?  (instruction-
level accounting of hc_iter.)  Times uoc_ms_hill_climb (10 iterations) through the library's per-kernel HIP events for
fields of growing size n at batch 1 (beyond 307 200 pixels the number of virtual blocks stays 256, so the pixel tiles per wave
grow linearly) and for batches of 480x640 fields / 224x224 crops; a linear fit over the single-field points separates the
cost per pixel tile from the cost per launch.
    python scripts/hc_scaling.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from unseenobjectclustering_amd import _native
from unseenobjectclustering_amd.utils import mean_shift as MS

dev = torch.device("cuda:0")
L = _native.lib()
g = torch.Generator(device=dev).manual_seed(0)


def run(batch, n, iters=10, reps=5):
    X = torch.nn.functional.normalize(torch.randn(batch, n, 64, device=dev, generator=g), dim=-1)
    Z0 = torch.nn.functional.normalize(torch.randn(batch, 100, 64, device=dev, generator=g), dim=-1)
    ws = MS._workspace(dev, L.uoc_ms_workspace_bytes(batch, n, 100))
    def once():
        Z = Z0.clone()
        _native.check(L.uoc_ms_hill_climb(_native.ptr(X), batch, n, _native.ptr(Z), 100, 20.0, iters, _native.ptr(ws), ws.numel(),
                                          _native.stream_ptr(dev)), "hc")
    for _ in range(2):
        once()
    torch.cuda.synchronize()
    _native.prof_enable(True)
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    rep = {r["kernel"]: r for r in _native.prof_report()}
    _native.prof_enable(False)
    hc = rep["hc_iter"]
    return 1e3 * hc["total_ms"] / hc["launches"]


rows = []
print("| batch | n | virtual blocks | pixel tiles per wave | us per hc_iter launch | us per field | TFLOP/s |")
print("|---:|---:|---:|---:|---:|---:|---:|")
for batch, n in [(1, 76800), (1, 153600), (1, 307200), (1, 614400), (1, 1228800), (1, 2457600), (2, 307200), (4, 307200), (8, 307200),
                 (1, 50176), (3, 50176), (5, 50176), (6, 50176), (7, 50176), (10, 50176), (28, 50176)]:
    us = run(batch, n)
    nvb = min(256, (n // 16 + 63) // 64)
    tpw = n / 16 / (nvb * 4)
    tf = 4.0 * batch * 100 * n * 64 / us / 1e6
    print(f"| {batch} | {n} | {nvb} | {tpw:.2f} | {us:.1f} | {us / batch:.1f} | {tf:.1f} |", flush=True)
    if batch == 1 and n >= 307200:
        rows.append((tpw, us))
a = np.array(rows)
slope, icpt = np.polyfit(a[:, 0], a[:, 1], 1)
print(f"\nsingle field, 256 virtual blocks on 256 CUs: {slope:.3f} us per pixel tile per wave ({slope * 2.4e3:.0f} cycles at 2.4 GHz; "
      f"224 MFMAs x 32 = 7168) + {icpt:.1f} us per launch")
