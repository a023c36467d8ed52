"""Dev helper: one 3x3 layer as Winograd F(2x2,3x3) (csrc/wino.hip) vs F(4x4,3x3) (csrc/wino4.hip: planes as groups of the
direct 1x1 kernel / persistent plane GEMM, tile overrides) on the launch shapes of the pipeline (four frames per launch
set in stage 1, ~28 crops in stage 2).  Times from the library's per-kernel-class HIP events; TF = algorithmic (direct
3x3) flops over the summed time of the layer's kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unseenobjectclustering_amd import _native
dev = torch.device("cuda:0")
L = _native.lib()
P = _native.ptr
G = 2
shapes = [  # name, B, H, W, C, dil
    ("s1 layer4 512 d4 4x60x80", 4, 60, 80, 512, 4),
    ("s1 layer3 256 d2 4x60x80", 4, 60, 80, 256, 2),
    ("s1 layer2 128 d1 4x60x80", 4, 60, 80, 128, 1),
    ("s2 layer4 512 d4 28x28x28", 28, 28, 28, 512, 4),
    ("s2 layer3 256 d2 28x28x28", 28, 28, 28, 256, 2),
    ("s2 layer2 128 d1 28x28x28", 28, 28, 28, 128, 1),
    ("s1 layer4 512 d4 1x60x80", 1, 60, 80, 512, 4),
    # round 4, VERDICT item 5 (F(3x3) remainder tiles for the 7x7 phase images of the crops): the plane GEMM a tile CLASS
    # of that scheme would run — ONE tile per phase image (28 crops x 16 phases = 448 rows per plane) instead of four
    ("s2 layer4 one tile per phase image 28x16x16", 28, 16, 16, 512, 4),
]


def measure(B, H, W, C, dil, env, iters=10):
    for k in ("UOC_CONV_WINOGRAD", "UOC_WINO4_GEMM", "UOC_WINO4_TILE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    L.uoc_reload_env()          # the library caches its knobs
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
    rep = {r["kernel"]: 1e3 * r["total_ms"] / r["launches"] for r in _native.prof_report()}
    _native.prof_enable(False)
    return rep


for name, B, H, W, C, dil in shapes:
    fl = 2.0 * G * B * H * W * C * C * 9
    measure(B, H, W, C, dil, {"UOC_CONV_WINOGRAD": "1"}, 5)   # clock ramp
    print(f"{name}  [{fl / 1e9:.1f} GF]", flush=True)
    for label, env in (("F2", {"UOC_CONV_WINOGRAD": "1"}),
                       ("F4 planes-as-groups", {"UOC_CONV_WINOGRAD": "4", "UOC_WINO4_GEMM": "1"}),
                       ("F4 persistent auto", {"UOC_CONV_WINOGRAD": "4", "UOC_WINO4_GEMM": "2"}),
                       ("F4 persistent 192x128", {"UOC_CONV_WINOGRAD": "4", "UOC_WINO4_GEMM": "2", "UOC_WINO4_TILE": "192x128"}),
                       ("F4 persistent 160x128", {"UOC_CONV_WINOGRAD": "4", "UOC_WINO4_GEMM": "2", "UOC_WINO4_TILE": "160x128"}),
                       ("F4 persistent 128x128", {"UOC_CONV_WINOGRAD": "4", "UOC_WINO4_GEMM": "2", "UOC_WINO4_TILE": "128x128"}),
                       ("F4 persistent 96x128", {"UOC_CONV_WINOGRAD": "4", "UOC_WINO4_GEMM": "2", "UOC_WINO4_TILE": "96x128"}),
                       ("F4 persistent 192x64", {"UOC_CONV_WINOGRAD": "4", "UOC_WINO4_GEMM": "2", "UOC_WINO4_TILE": "192x64"}),
                       ("direct", {})):
        if os.environ.get("WINO4_BENCH_ONLY") and os.environ["WINO4_BENCH_ONLY"] not in label:
            continue
        try:
            rep = measure(B, H, W, C, dil, env)
        except Exception as e:  # noqa: BLE001
            print(f"   {label:24s} failed: {e}")
            continue
        tot = sum(rep.values())
        parts = "  ".join(f"{k}:{v:7.1f}" for k, v in rep.items())
        print(f"   {label:24s} total {tot:7.1f} us = {fl / tot / 1e6:6.1f} TF algorithmic   {parts}", flush=True)
