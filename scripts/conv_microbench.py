"""Dev helper: time single conv layers (both branches, G=2) through uoc_conv2d_nhwc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unseenobjectclustering_amd import _native
dev = torch.device("cuda:0")
L = _native.lib()
P = _native.ptr
shapes = [  # name, B, H, W, Cin, Cout, K, stride, dil
    ("layer4 512->512 d4 60x80", 1, 60, 80, 512, 512, 3, 1, 4),
    ("layer3 256->256 d2 60x80", 1, 60, 80, 256, 256, 3, 1, 2),
    ("layer2 128->128 60x80", 1, 60, 80, 128, 128, 3, 1, 1),
    ("layer1 64->64 120x160", 1, 120, 160, 64, 64, 3, 1, 1),
    ("s2 layer4 512->512 d4 7x28x28", 7, 28, 28, 512, 512, 3, 1, 4),
    ("s2 layer3 256->256 d2 7x28x28", 7, 28, 28, 256, 256, 3, 1, 2),
    ("s2 layer2 128->128 7x28x28", 7, 28, 28, 128, 128, 3, 1, 1),
    ("s2 layer1 64->64 7x56x56", 7, 56, 56, 64, 64, 3, 1, 1),
]
G = 2
for name, B, H, W, Cin, Cout, K, stride, dil in shapes:
    x = torch.randn(G, B, H, W, Cin, device=dev)
    w = torch.randn(G, K * K, Cout, Cin, device=dev) * 0.02
    b = torch.randn(G, Cout, device=dev)
    pad = dil if K == 3 else 0
    Ho = (H + 2 * pad - dil * (K - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (K - 1) - 1) // stride + 1
    out = torch.empty(G, B, Ho, Wo, Cout, device=dev)
    st = _native.stream_ptr(dev)
    def run():
        _native.check(L.uoc_conv2d_nhwc(P(x), P(w), P(b), None, P(out), G, B, H, W, Cin, Cout, K, stride, dil, pad, 1, st), "conv")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * G * B * Ho * Wo * Cout * Cin * K * K
    print(f"{name:34s} {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TFLOP/s")
