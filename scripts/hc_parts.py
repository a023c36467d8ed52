"""Costs of the seed-tile parts of the hill-climbing kernel's flat item schedule (csrc/meanshift.hip, HcPlan): times
uoc_ms_hill_climb with UOC_HC_PARTS forced (speed-only, bit-identical) for crop batches, through the library's per-kernel
HIP events.  One 224x224 crop = 49 virtual blocks: with P parts on one pass over the grid a launch costs one part + the
fixed launch cost, so P = 1 / 2 / 3 / 6 give the whole item and the 4-, 3- and 2-tile bodies; then the batches of the
one-frame-at-a-time schedule (K = 5..9 crops) and of the throughput schedule (27..31) with the model's own choice (0).
    python scripts/hc_parts.py [lib.so to compare with]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unseenobjectclustering_amd import _native
from unseenobjectclustering_amd.utils import mean_shift as MS

dev = torch.device("cuda:0")
L = _native.lib()
g = torch.Generator(device=dev).manual_seed(0)


def run(batch, n, parts, iters=10, reps=5):
    os.environ["UOC_HC_PARTS"] = str(parts)
    L.uoc_reload_env()
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


print("| batch | n | " + " | ".join(f"parts={p}" for p in (1, 0, 2, 3, 6)) + " |   (us per hc_iter launch; 0 = the makespan model)")
print("|---:|---:|" + "---:|" * 5)
cases = [(1, 50176), (2, 50176), (3, 50176), (5, 50176), (6, 50176), (7, 50176), (8, 50176), (9, 50176), (10, 50176), (12, 50176),
         (27, 50176), (28, 50176), (29, 50176), (30, 50176), (31, 50176), (1, 307200), (2, 307200), (4, 307200)]
for batch, n in cases:
    row = [run(batch, n, p) for p in (1, 0, 2, 3, 6)]
    print(f"| {batch} | {n} | " + " | ".join(f"{u:.1f}" for u in row) + " |", flush=True)
