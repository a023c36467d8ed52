"""EXPERIMENT: one Winograd layer with the fp32 plane GEMM against the split-precision (bf16 x 3) one, on the launch shapes of the
pipeline.  Times from the library's per-kernel-class HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unseenobjectclustering_amd import _native
dev = torch.device("cuda:0")
L = _native.lib()
P = _native.ptr
G = 2
shapes = [("s1 layer4 512 d4 4x60x80", 4, 60, 80, 512, 4), ("s1 layer3 256 d2 4x60x80", 4, 60, 80, 256, 2),
          ("s1 layer2 128 d1 4x60x80", 4, 60, 80, 128, 1), ("s1 layer1 64 d1 4x120x160", 4, 120, 160, 64, 1),
          ("s2 layer4 512 d4 28x28x28", 28, 28, 28, 512, 4), ("s2 layer3 256 d2 28x28x28", 28, 28, 28, 256, 2),
          ("s1 layer4 512 d4 1x60x80", 1, 60, 80, 512, 4)]


def measure(B, H, W, C, dil, algo, iters=10):
    gen = torch.Generator(device=dev).manual_seed(1234 + B + C)
    x = torch.randn(G, B, H, W, C, device=dev, generator=gen)
    w = torch.randn(G, 9, C, C, device=dev, generator=gen) * 0.02
    b = torch.randn(G, C, device=dev, generator=gen)
    out = torch.empty(G, B, H, W, C, device=dev)
    res = torch.randn(G, B, H, W, C, device=dev, generator=gen)
    st = _native.stream_ptr(dev)
    run = lambda: _native.check(L.uoc_conv2d_nhwc_algo(P(x), P(w), P(b), P(res), P(out), G, B, H, W, C, C, 3, 1, dil, dil, 1, algo, st), "conv")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    _native.prof_enable(True)
    for _ in range(iters):
        run()
    torch.cuda.synchronize()
    rep = {r["kernel"]: 1e3 * r["total_ms"] / r["launches"] for r in _native.prof_report()}
    _native.prof_enable(False)
    return rep, out.clone()


for name, B, H, W, C, dil in shapes:
    if os.environ.get("SPLIT_BENCH_SHAPES") and os.environ["SPLIT_BENCH_SHAPES"] not in name:
        continue
    fl = 2.0 * G * B * H * W * C * C * 9
    measure(B, H, W, C, dil, 4, 3)
    r32, o32 = measure(B, H, W, C, dil, 4)
    r3, o3 = measure(B, H, W, C, dil, 5)
    d = (o3 - o32).abs().max().item() / max(1.0, o32.abs().max().item())
    f = lambda r: "  ".join(f"{k}:{v:7.1f}" for k, v in r.items())
    print(f"{name} [{fl / 1e9:.0f} GF]\n   fp32   total {sum(r32.values()):7.1f} us  {f(r32)}\n   bf16x3 total {sum(r3.values()):7.1f} us  {f(r3)}   max rel diff {d:.2e}", flush=True)
