"""Dev helper for PMC runs: a handful of Winograd launches of one shape (layer4 at 60x80, both branches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UOC_CONV_WINOGRAD"] = "1"
import torch
from unseenobjectclustering_amd import _native
dev = torch.device("cuda:0")
L = _native.lib(); P = _native.ptr
G, B, H, W, C, dil = 2, 1, 60, 80, int(os.environ.get("WINO_C", "512")), int(os.environ.get("WINO_D", "4"))
x = torch.randn(G, B, H, W, C, device=dev); w = torch.randn(G, 9, C, C, device=dev) * 0.02
b = torch.randn(G, C, device=dev); out = torch.empty(G, B, H, W, C, device=dev)
st = _native.stream_ptr(dev)
for _ in range(int(os.environ.get("WINO_N", "6"))):
    _native.check(L.uoc_conv2d_nhwc(P(x), P(w), P(b), None, P(out), G, B, H, W, C, C, 3, 1, dil, dil, 1, st), "conv")
torch.cuda.synchronize()
