"""Dev helper: which call makes `rocprofv3 --kernel-trace -- python ...` segfault at process exit?

    rocprofv3 --kernel-trace -d /tmp/x -o t -- python scripts/rocprof_exit_bisect.py load|fps|hc|cluster

Finding (round 2, ROCm 7.2): any process that launched the persistent (cooperative or plain) farthest-point sampling
kernel crashes inside the tool's exit handler — also with round 1's library, with or without the stream-ordering
event, with or without dynamic LDS; the streaming sampling kernel (UOC_FPS_PERSISTENT=0) does not.  The trace database
IS written when the interpreter falls off the end of the script (bench.py therefore does not call sys.exit(0)); with
SystemExit -> Py_Exit the crash comes first and the database is lost."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
mode = sys.argv[1]
from unseenobjectclustering_amd import _native, synth
from unseenobjectclustering_amd.utils import mean_shift as MS
dev = torch.device("cuda:0")
L = _native.lib()
if mode == "load":
    torch.zeros(4, device=dev).sum().item()
elif mode in ("hc", "fps", "cluster"):
    X, _ = synth.embedding_field(1, 96, 128, 64, 4, 0.05)
    Xd = torch.from_numpy(X).to(dev)[None].contiguous()
    n, m = X.shape[0], 100
    ws = MS._workspace(dev, L.uoc_ms_workspace_bytes(1, n, m))
    st, P = _native.stream_ptr(dev), _native.ptr
    if mode == "hc":
        Z = Xd[0, :m].clone()
        _native.check(L.uoc_ms_hill_climb(P(Xd), 1, n, P(Z), m, 20.0, 2, P(ws), ws.numel(), st), "hc")
    elif mode == "fps":
        first = torch.zeros(1, dtype=torch.int32, device=dev)
        seeds = torch.empty((1, m, 64), device=dev); idx = torch.empty((1, m), dtype=torch.int32, device=dev)
        _native.check(L.uoc_ms_select_seeds(P(Xd), 1, n, m, P(first), P(seeds), P(idx), P(ws), ws.numel(), st), "fps")
    else:
        MS.cluster_batch(Xd, [5], 20.0, 100, 10, 0.04)
    torch.cuda.synchronize()
print("done", mode, flush=True)
