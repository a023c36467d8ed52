"""Dev helper (GPU box): phase stamps of the register-resident hill-climbing kernel (UOC_HC_ABLATE=9 prints them) and
HIP-event time of single launches, to separate launch / prologue / main loop / epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unseenobjectclustering_amd import synth, _native
from unseenobjectclustering_amd.utils import mean_shift as MS
dev = torch.device("cuda:0")
X, _ = synth.embedding_field(1, 480, 640, 64, 7, 0.05)
Xd = torch.from_numpy(X).to(dev)[None].contiguous()
n, m = X.shape[0], 100
L = _native.lib()
ws = MS._workspace(dev, L.uoc_ms_workspace_bytes(1, n, m))
Z = Xd[0, :m].clone()
st, P = _native.stream_ptr(dev), _native.ptr
for iters in (1, 1, 2, 10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _native.check(L.uoc_ms_hill_climb(P(Xd), 1, n, P(Z), m, 20.0, iters, P(ws), ws.numel(), st), "hc")
    e1.record(); torch.cuda.synchronize()
    print("iters %d: %.1f us total" % (iters, 1e3 * e0.elapsed_time(e1)), flush=True)
