"""Dev helper: Winograd GEMM per tile height (UOC_WINO_TMT) and timing ablations (UOC_WINO_VARIANT) on the
layer3 / layer4 shapes of both stages, times from the library's own per-kernel-class HIP events."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UOC_CONV_WINOGRAD"] = "1"
import torch
from unseenobjectclustering_amd import _native
dev = torch.device("cuda:0")
L = _native.lib()
P = _native.ptr
shapes = [  # name, B, H, W, C, dil
    ("layer4 512 d4 60x80", 1, 60, 80, 512, 4),
    ("layer3 256 d2 60x80", 1, 60, 80, 256, 2),
    ("s2 layer4 512 d4 5x28x28", 5, 28, 28, 512, 4),
    ("s2 layer3 256 d2 5x28x28", 5, 28, 28, 256, 2),
    ("s2 layer4 512 d4 2x28x28", 2, 28, 28, 512, 4),
]
G = 2


def measure(B, H, W, C, dil, iters=20):
    x = torch.randn(G, B, H, W, C, device=dev)
    w = torch.randn(G, 9, C, C, device=dev) * 0.02
    b = torch.randn(G, C, device=dev)
    out = torch.empty(G, B, H, W, C, device=dev)
    st = _native.stream_ptr(dev)
    run = lambda: _native.check(L.uoc_conv2d_nhwc(P(x), P(w), P(b), None, P(out), G, B, H, W, C, C, 3, 1, dil, dil, 1, st), "conv")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    _native.prof_enable(True)
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    rep = {r["kernel"]: r for r in _native.prof_report()}
    _native.prof_enable(False)
    g = rep["wino_gemm"]
    i = rep["wino_input"]
    return 1e3 * g["total_ms"] / g["launches"], 1e3 * i["total_ms"] / i["launches"]


for name, B, H, W, C, dil in shapes:
    fl = 2.0 * G * B * H * W * C * C * 9
    line = f"{name:28s}"
    measure(B, H, W, C, dil, 10)   # clock ramp
    for tmt in ("auto", "2", "3", "4", "5", "6", "7"):
        os.environ.pop("UOC_WINO_TMT", None)
        if tmt != "auto":
            os.environ["UOC_WINO_TMT"] = tmt
        us, us_in = measure(B, H, W, C, dil)
        line += f"  T{tmt}:{us:6.1f}"
    os.environ.pop("UOC_WINO_TMT", None)
    line += f"  in:{us_in:5.1f}us  [{fl/1e9:.1f} GF]"
    print(line, flush=True)
if os.environ.get("WINO_ABL") != "1":
    sys.exit(0)
print("ablations (TMT=5): variant 0 full, 1 no DMA, 2 +no barrier, 3 +no frag reads, 4 full but every chunk re-reads chunk 0 (cache hits)")
for name, B, H, W, C, dil in shapes[:2]:
    line = f"{name:28s}"
    os.environ["UOC_WINO_TMT"] = "5"
    for v in ("0", "1", "4"):
        os.environ["UOC_WINO_VARIANT"] = v
        us, _ = measure(B, H, W, C, dil)
        line += f"  V{v}:{us:6.1f}"
    os.environ.pop("UOC_WINO_VARIANT", None)
    print(line, flush=True)
