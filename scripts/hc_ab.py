"""A/B of hill-climbing launches between library builds (UOC_LIB_PATH): us per hc_iter launch for a few (batch, n)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
        _native.check(L.uoc_ms_hill_climb(_native.ptr(X), batch, n, _native.ptr(Z), 100, 20.0, iters, _native.ptr(ws), ws.numel(), _native.stream_ptr(dev)), "hc")
    for _ in range(2):
        once()
    torch.cuda.synchronize()
    _native.prof_enable(True)
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    rep = {r["kernel"]: r for r in _native.prof_report()}
    _native.prof_enable(False)
    return 1e3 * rep["hc_iter"]["total_ms"] / rep["hc_iter"]["launches"]
cases = [(1, 307200), (2, 307200), (4, 307200), (8, 307200), (3, 307200), (5, 50176), (10, 50176), (28, 50176)]
print(os.environ.get("UOC_LIB_PATH", "shipped"), " ".join(f"{b}x{n}:{run(b, n):.1f}" for b, n in cases), flush=True)
