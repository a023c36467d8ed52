"""Dev helper (historical): wino_gemm_kernel (auto tile height) vs a register-blocked wino_gemm2_kernel (wave tile 32
channels x 16*BT tiles; UOC_WINO_KERNEL=2, BT 2/3/4) that round 2 built, verified and removed again.  Measured on MI355X
(us per launch, two branches):

    shape                  kernel-1 auto |   BT2     BT3     BT4
    layer4 1x60x80             209.4     |  309.5   217.8   277.6
    layer3 1x60x80              70.7     |   83.5   115.4   147.2
    layer4 3x60x80             573.6     |  613.3   643.9   556.0
    layer3 3x60x80             161.6     |  163.5   226.1   149.7
    s2 layer4 7x28x28          253.1     |  304.9   424.1   278.8
    s2 layer3 7x28x28           71.3     |   82.9   114.6   147.1
    s2 layer4 21x28x28         758.7     |  912.8   857.3   832.7
    s2 layer3 21x28x28         198.7     |  242.8   226.5   289.2

The script still times kernel 1; the UOC_WINO_KERNEL switch no longer exists."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UOC_CONV_WINOGRAD"] = "1"
import torch
from unseenobjectclustering_amd import _native
dev = torch.device("cuda:0")
L, P = _native.lib(), _native.ptr
shapes = [("layer4 1x60x80", 1, 60, 80, 512, 4), ("layer3 1x60x80", 1, 60, 80, 256, 2), ("layer4 3x60x80", 3, 60, 80, 512, 4),
          ("layer3 3x60x80", 3, 60, 80, 256, 2), ("s2 layer4 7x28x28", 7, 28, 28, 512, 4), ("s2 layer3 7x28x28", 7, 28, 28, 256, 2),
          ("s2 layer4 21x28x28", 21, 28, 28, 512, 4), ("s2 layer3 21x28x28", 21, 28, 28, 256, 2)]
G = 2

def measure(B, H, W, C, dil, iters=20):
    x = torch.randn(G, B, H, W, C, device=dev); w = torch.randn(G, 9, C, C, device=dev) * 0.02
    b = torch.randn(G, C, device=dev); out = torch.empty(G, B, H, W, C, device=dev)
    st = _native.stream_ptr(dev)
    run = lambda: _native.check(L.uoc_conv2d_nhwc(P(x), P(w), P(b), None, P(out), G, B, H, W, C, C, 3, 1, dil, dil, 1, st), "conv")
    for _ in range(3): run()
    torch.cuda.synchronize(); _native.prof_enable(True)
    for _ in range(iters): run()
    torch.cuda.synchronize()
    rep = {r["kernel"]: r for r in _native.prof_report()}; _native.prof_enable(False)
    return 1e3 * rep["wino_gemm"]["total_ms"] / rep["wino_gemm"]["launches"], out

for name, B, H, W, C, dil in shapes:
    os.environ.pop("UOC_WINO_KERNEL", None)
    measure(B, H, W, C, dil, 5)
    base, ref = measure(B, H, W, C, dil)
    line = f"{name:22s} kernel-1 auto {base:7.1f} us |"
    os.environ["UOC_WINO_KERNEL"] = "2"
    for bt in ("2", "3", "4"):
        os.environ["UOC_WINO_BT"] = bt
        us, _ = measure(B, H, W, C, dil)
        line += f" BT{bt}: {us:7.1f}"
    os.environ.pop("UOC_WINO_KERNEL", None)
    print(line, flush=True)
