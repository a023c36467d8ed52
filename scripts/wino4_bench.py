"""Micro-benchmark: one 3x3 layer as Winograd F(4x4,3x3) (csrc/wino4.hip) on the launch shapes of the pipeline (four frames
per launch set in stage 1, ~28 crops in stage 2, one frame alone) next to the direct kernel.  Times from the library's
per-kernel-class HIP events; TF = algorithmic (direct 3x3) flops over the summed time of the layer's kernels.
WINO4_BENCH_SHAPES=<substring> restricts the shapes.  (The A/B arms of rounds 3-5 — waves per block, tile overrides, planes
as groups, F(2x2) — needed the development build that round 6 removed; their results are in HISTORY.md.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unseenobjectclustering_amd import _native
dev = torch.device("cuda:0")
L = _native.lib()
P = _native.ptr
G = 2
KNOBS = ()
shapes = [  # name, B, H, W, C, dil
    ("s1 layer4 512 d4 4x60x80", 4, 60, 80, 512, 4),
    ("s1 layer3 256 d2 4x60x80", 4, 60, 80, 256, 2),
    ("s1 layer2 128 d1 4x60x80", 4, 60, 80, 128, 1),
    ("s1 layer1 64 d1 4x120x160", 4, 120, 160, 64, 1),
    ("s2 layer4 512 d4 28x28x28", 28, 28, 28, 512, 4),
    ("s2 layer3 256 d2 28x28x28", 28, 28, 28, 256, 2),
    ("s2 layer2 128 d1 28x28x28", 28, 28, 28, 128, 1),
    ("s1 layer4 512 d4 1x60x80", 1, 60, 80, 512, 4),
]


def measure(B, H, W, C, dil, algo, env, iters=10):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    L.uoc_reload_env()          # the library caches its knobs
    gen = torch.Generator(device=dev).manual_seed(1234 + B + C)       # same operands for every arm: the bit-identity check
    x = torch.randn(G, B, H, W, C, device=dev, generator=gen)
    w = torch.randn(G, 9, C, C, device=dev, generator=gen) * 0.02
    b = torch.randn(G, C, device=dev, generator=gen)
    out = torch.empty(G, B, H, W, C, device=dev)
    res = torch.randn(G, B, H, W, C, device=dev, generator=gen) if os.environ.get("WINO4_BENCH_RESIDUAL", "1") != "0" else None   # conv2 of a BasicBlock
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
    return rep, out


arms = [("F4 shipped", 4, {})]
arms.append(("direct", 0, {}))
for name, B, H, W, C, dil in shapes:
    if os.environ.get("WINO4_BENCH_SHAPES") and os.environ["WINO4_BENCH_SHAPES"] not in name:
        continue
    fl = 2.0 * G * B * H * W * C * C * 9
    measure(B, H, W, C, dil, 4, {}, 5)   # clock ramp
    print(f"{name}  [{fl / 1e9:.1f} GF]", flush=True)
    ref = None
    for label, algo, env in arms:
        if os.environ.get("WINO4_BENCH_ONLY") and os.environ["WINO4_BENCH_ONLY"] not in label:
            continue
        try:
            rep, out = measure(B, H, W, C, dil, algo, env)
        except Exception as e:  # noqa: BLE001
            print(f"   {label:24s} failed: {e}")
            continue
        same = ""
        if algo == 4 and "planes" not in label:        # every plane-GEMM arm must give the same bits
            if ref is None:
                ref = out.clone()
            same = "  bit-identical" if torch.equal(ref, out) else "  DIFFERS FROM THE FIRST F4 ARM"
        tot = sum(rep.values())
        parts = "  ".join(f"{k}:{v:7.1f}" for k, v in rep.items())
        print(f"   {label:24s} total {tot:7.1f} us = {fl / tot / 1e6:6.1f} TF algorithmic   {parts}{same}", flush=True)
