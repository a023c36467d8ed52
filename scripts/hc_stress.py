"""Run-to-run determinism stress of the hill-climbing launch shapes of the one-frame-at-a-time and throughput schedules
(whole items, seed-tile parts in one and two passes, interleaved and field-affine walks): every repetition must be
bit-identical to the first, with other work queued on a second stream meanwhile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unseenobjectclustering_amd import _native
from unseenobjectclustering_amd.utils import mean_shift as MS
dev = torch.device("cuda:0")
L = _native.lib()
g = torch.Generator(device=dev).manual_seed(0)
side = torch.cuda.Stream()
bad = 0
for batch, n in [(1, 307200), (4, 307200), (3, 307200), (6, 50176), (7, 50176), (8, 50176), (12, 50176), (28, 50176), (29, 50176), (1, 76800), (5, 2000)]:
    X = torch.nn.functional.normalize(torch.randn(batch, n, 64, device=dev, generator=g), dim=-1)
    Z0 = torch.nn.functional.normalize(torch.randn(batch, 100, 64, device=dev, generator=g), dim=-1)
    ws = MS._workspace(dev, L.uoc_ms_workspace_bytes(batch, n, 100))
    junk = torch.randn(4096, 4096, device=dev)
    ref = None
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
        ws.fill_(float("nan") if rep % 2 else 0.0) if ws.dtype.is_floating_point else ws.fill_(255 if rep % 2 else 0)
        Z = Z0.clone()
        with torch.cuda.stream(side):
            junk2 = junk @ junk        # something else on the chip
        _native.check(L.uoc_ms_hill_climb(_native.ptr(X), batch, n, _native.ptr(Z), 100, 20.0, 10, _native.ptr(ws), ws.numel(), _native.stream_ptr(dev)), "hc")
        torch.cuda.synchronize()
        if ref is None:
            ref = Z.clone()
        elif not torch.equal(ref, Z):
            bad += 1
            print(f"MISMATCH batch {batch} n {n} rep {rep}: {(ref != Z).sum().item()} values differ, max {(ref - Z).abs().max().item():.3e}", flush=True)
    print(f"batch {batch} n {n}: done", flush=True)
print("mismatches:", bad)
