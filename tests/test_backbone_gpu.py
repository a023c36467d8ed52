"""GPU parity of the HIP ResNet34-8s RGB-D embedding network.

(a) the fp32-MFMA conv kernel against torch CPU conv2d (plain fp32 reference of the same op),
(b) the whole two-branch network against golden embeddings captured from the reference's own
    SEGNET.forward (tests/golden/backbone.npz) — tolerance 1e-3 absolute as north_star states
    (embeddings are unit-norm, so absolute == relative to the vector scale),
(c) the CPU oracle on a fresh seeded input.
"""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import backbone_oracle as BO
from unseenobjectclustering_amd.fcn.config import cfg
from unseenobjectclustering_amd import _native, networks, synth

pytestmark = pytest.mark.gpu
EMBED_TOL = 1e-3

CONV_CASES = [
    # B, H, W, Cin, Cout, K, stride, dil, residual, relu
    (1, 30, 40, 64, 64, 3, 1, 1, True, True),
    (2, 31, 27, 64, 128, 3, 2, 1, False, True),
    (1, 20, 24, 128, 256, 3, 1, 2, True, True),
    (1, 15, 20, 256, 512, 3, 1, 4, False, True),
    (1, 15, 20, 512, 512, 3, 1, 4, True, True),
    (2, 33, 17, 64, 128, 1, 2, 1, False, False),
    (1, 60, 80, 512, 64, 1, 1, 1, False, False),
    (3, 9, 7, 32, 64, 3, 1, 1, False, False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_kernel_vs_torch_cpu(device, case):
    B, H, W, Cin, Cout, K, stride, dil, use_res, relu = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / np.sqrt(Cin * K * K)
    b = torch.randn(Cout, generator=g)
    pad = dil if K == 3 else 0
    ref = F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dil)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if res is not None:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    xd = x.permute(0, 2, 3, 1).contiguous().to(device)                     # NHWC
    wd = w.permute(2, 3, 0, 1).reshape(K * K, Cout, Cin).contiguous().to(device)   # [tap][cout][cin]
    bd = b.to(device)
    rd = res.permute(0, 2, 3, 1).contiguous().to(device) if res is not None else None
    Ho, Wo = ref.shape[2], ref.shape[3]
    out = torch.empty((B, Ho, Wo, Cout), device=device)
    L = _native.lib()
    rc = L.uoc_conv2d_nhwc(_native.ptr(xd), _native.ptr(wd), _native.ptr(bd), _native.ptr(rd), _native.ptr(out),
                           1, B, H, W, Cin, Cout, K, stride, dil, pad, int(relu), _native.stream_ptr(device))
    _native.check(rc, "uoc_conv2d_nhwc")
    got = out.cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, ref.abs().max().item()), err


def _net(wseed, device):
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synthetic_state_dict(wseed).items()}
    net = networks.seg_resnet34_8s_embedding(2, 64, sd)
    return net.eval(), sd


GOLDEN_CASES = {
    "tiny_64x64": dict(wseed=1, frames=[7], H=64, W=64),
    "odd_72x104": dict(wseed=2, frames=[8], H=72, W=104),
    "crops_224": dict(wseed=2, frames=[4, 5], H=224, W=224),
    "full_480x640": dict(wseed=1, frames=[1], H=480, W=640),
}


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_network_matches_reference_golden(golden_dir, device, name):
    g = np.load(os.path.join(golden_dir, "backbone.npz"))
    c = GOLDEN_CASES[name]
    net, _ = _net(c["wseed"], device)
    frames = [synth.rgbd_frame(s, c["H"], c["W"], 4) for s in c["frames"]]
    img = torch.from_numpy(np.concatenate([f["image_color"] for f in frames])).to(device)
    dep = torch.from_numpy(np.concatenate([f["depth"] for f in frames])).to(device)
    feat = net(img, None, dep)
    assert feat.shape == (len(frames), 64, c["H"], c["W"])
    flat = feat.permute(0, 2, 3, 1).reshape(len(frames), -1, 64).cpu().numpy()
    if name + "/pos" in g:
        flat = flat[:, g[name + "/pos"]]
    err = np.abs(flat - g[name + "/embed"]).max()
    assert err < EMBED_TOL, err
    assert np.abs(np.linalg.norm(flat, axis=2) - 1).max() < 1e-5


def test_network_vs_oracle_fresh_input(device):
    net, sd = _net(5, device)
    fr = synth.rgbd_frame(12, 120, 88, 3)
    img, dep = torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"])
    want = BO.segnet_forward(sd, img, dep)
    got = net(img.to(device), None, dep.to(device)).cpu()
    assert (got - want).abs().max().item() < EMBED_TOL


@pytest.mark.parametrize("hw", [(77, 93), (50, 131)])
def test_network_odd_sizes_vs_oracle(device, hw):
    """Sizes that are not multiples of 8 (the reference's conv arithmetic handles any H x W): odd stem / pool /
    stride-2 outputs, partial Winograd tiles in every dilation phase, non-integer upsampling ratios."""
    net, sd = _net(3, device)
    fr = synth.rgbd_frame(13, hw[0], hw[1], 2)
    img, dep = torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"])
    want = BO.segnet_forward(sd, img, dep)
    got = net(img.to(device), None, dep.to(device)).cpu()
    assert got.shape == want.shape == (1, 64, hw[0], hw[1])
    assert (got - want).abs().max().item() < EMBED_TOL


def test_network_interface(device):
    net, sd = _net(1, device)
    assert set(net.state_dict().keys()) == set(sd.keys())
    with pytest.raises(_native.NativeError):
        net(torch.zeros(1, 3, 64, 64), None, torch.zeros(1, 3, 64, 64))   # CPU tensors: no fallback
    net.train()
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 3, 64, 64, device=device), None, torch.zeros(1, 3, 64, 64, device=device))


WINO_CASES = [
    # G, B, H, W, Cin, Cout, dil, residual, relu   (3x3, stride 1: the layers Winograd F(2x2,3x3) serves)
    (1, 1, 30, 40, 64, 64, 1, True, True),
    (2, 1, 60, 80, 128, 128, 1, False, True),
    (2, 1, 60, 80, 256, 256, 2, True, True),
    (2, 1, 15, 20, 512, 512, 4, True, True),      # odd phase sub-image: partial tiles
    (1, 3, 28, 28, 256, 512, 4, False, False),    # 7x7 phase sub-images, batch 3
    (1, 2, 13, 9, 64, 128, 2, False, True),       # ragged
]


WINO4_EXTRA_CASES = [
    (2, 4, 60, 80, 256, 256, 2, True, True),      # stage-1 layer3 launch set of four frames: several items per block
    (2, 9, 28, 28, 512, 512, 4, True, True),      # stage-2 layer4, nine crops
    (1, 5, 28, 28, 128, 128, 1, False, True),     # stage-2 layer2: 7x7 tiles per crop, 4 cin chunks per item
    (1, 1, 61, 83, 128, 256, 2, False, False),    # odd size, Cin != Cout
]


@pytest.mark.parametrize("case", WINO_CASES + WINO4_EXTRA_CASES)
def test_winograd_conv_vs_torch_cpu(device, case):
    """Winograd F(4x4,3x3) (csrc/wino4.hip) against torch CPU conv2d; fp32 Winograd differs from the direct sum only by
    rounding (bar: 2e-4 of the output scale, measured ~1e-6)."""
    G, B, H, W, Cin, Cout, dil, use_res, relu = case
    g = torch.Generator().manual_seed(abs(hash(case)) % (2 ** 31))
    x = torch.randn(G, B, Cin, H, W, generator=g)
    w = torch.randn(G, Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)
    b = torch.randn(G, Cout, generator=g)
    res = torch.randn(G, B, Cout, H, W, generator=g) if use_res else None
    ref = torch.stack([F.conv2d(x[i], w[i], b[i], padding=dil, dilation=dil) for i in range(G)])
    if res is not None:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    got = _wino4_conv(device, x, w, b, res, dil, relu)
    err = (got - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, ref.abs().max().item()), err


def _wino4_conv(device, x, w, b, res, dil, relu, env=None, algo=_native.CONV_WINOGRAD4, expect_rc=0):
    """uoc_conv2d_nhwc_algo; x [G,B,Cin,H,W], w [G,Cout,Cin,3,3], b [G,Cout] | None -> [G,B,Cout,H,W] (CPU).
    `env`: speed-only / test variables set for the call (uoc_reload_env around it)."""
    G, B, Cin, H, W = x.shape
    Cout = w.shape[1]
    xd = x.permute(0, 1, 3, 4, 2).contiguous().to(device)
    wd = w.permute(0, 3, 4, 1, 2).reshape(G, 9, Cout, Cin).contiguous().to(device)
    bd = b.to(device) if b is not None else None
    rd = res.permute(0, 1, 3, 4, 2).contiguous().to(device) if res is not None else None
    out = torch.empty((G, B, H, W, Cout), device=device)
    L = _native.lib()
    for k, v in (env or {}).items():
        os.environ[k] = v
    L.uoc_reload_env()
    try:
        rc = L.uoc_conv2d_nhwc_algo(_native.ptr(xd), _native.ptr(wd), _native.ptr(bd), _native.ptr(rd), _native.ptr(out),
                                    G, B, H, W, Cin, Cout, 3, 1, dil, dil, int(relu), algo, _native.stream_ptr(device))
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)
        L.uoc_reload_env()
    if expect_rc:
        assert rc == expect_rc, rc
        return None
    _native.check(rc, "uoc_conv2d_nhwc_algo")
    return out.cpu().permute(0, 1, 4, 2, 3)


def test_winograd4_edge_cases(device):
    """(a) Cout = 192 (Cout / 64 not a power of two) is not eligible: the explicit-algorithm entry says so
    (UOC_EINVAL) and the direct kernel — what the network's static rule picks for such a layer — computes it; (b) a null bias
    is zero on the Winograd path as on the direct one; (c) a batch whose frequency planes exceed the 32-bit offset range is
    run as slices of the batch — forced here with UOC_SPLIT_MAX_MB — with bit-identical results; (d) the same for the direct
    kernel's 2 GB input limit (round 5: the batch split replaces the register-staged fallback kernel)."""
    g = torch.Generator().manual_seed(77)
    # (a)
    x = torch.randn(1, 2, 128, 20, 24, generator=g)
    w = torch.randn(1, 192, 128, 3, 3, generator=g) / np.sqrt(128 * 9)
    b = torch.randn(1, 192, generator=g)
    ref = F.relu(F.conv2d(x[0], w[0], b[0], padding=1))
    _wino4_conv(device, x, w, b, None, 1, True, expect_rc=-22)
    got = _wino4_conv(device, x, w, b, None, 1, True, algo=_native.CONV_DIRECT)[0]
    assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    # (b)
    x = torch.randn(2, 1, 128, 20, 24, generator=g)
    w = torch.randn(2, 128, 128, 3, 3, generator=g) / np.sqrt(128 * 9)
    ref = torch.stack([F.conv2d(x[i], w[i], None, padding=2, dilation=2) for i in range(2)])
    got = _wino4_conv(device, x, w, None, None, 2, False)
    assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    # (c) 5 images of 28x28, 256 channels, d = 2: 72 planes x 49 tiles x 256 x 4 B = 3.4 MB per image -> slices of 2, 2, 1
    x = torch.randn(2, 5, 256, 28, 28, generator=g)
    w = torch.randn(2, 256, 256, 3, 3, generator=g) / np.sqrt(256 * 9)
    b = torch.randn(2, 256, generator=g)
    res = torch.randn(2, 5, 256, 28, 28, generator=g)
    whole = _wino4_conv(device, x, w, b, res, 2, True)
    sliced = _wino4_conv(device, x, w, b, res, 2, True, env={"UOC_SPLIT_MAX_MB": "8"})
    assert torch.equal(whole, sliced)
    ref = F.relu(torch.stack([F.conv2d(x[i], w[i], b[i], padding=2, dilation=2) for i in range(2)]) + res)
    assert (whole - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    # (d) direct kernel: 2 groups x 5 images of 28 x 28 x 256 x 4 B = 0.8 MB each; a 2 MB limit -> per group 3 + 2, then 2 + 1 ...
    whole = _wino4_conv(device, x, w, b, res, 2, True, algo=_native.CONV_DIRECT)
    sliced = _wino4_conv(device, x, w, b, res, 2, True, algo=_native.CONV_DIRECT, env={"UOC_SPLIT_MAX_MB": "2"})
    assert torch.equal(whole, sliced)
    assert (whole - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("case", WINO_CASES + WINO4_EXTRA_CASES)
def test_split_precision_plane_gemm_vs_fp32(device, case):
    """EXPERIMENT (round 6, csrc/wino4_split.hip): the Winograd layer with its plane GEMM in split precision — three bf16 terms
    per fp32 operand, six bf16 MFMA products, fp32 accumulation — against torch CPU conv2d (the bar of the fp32 path: 2e-4 of the
    output scale) and against the fp32 Winograd path itself (the dropped terms are <= 2^-24 relative per product: bar 2e-5 of
    the scale, measured ~1e-6)."""
    G, B, H, W, Cin, Cout, dil, use_res, relu = case
    g = torch.Generator().manual_seed(abs(hash(case)) % (2 ** 31))
    x = torch.randn(G, B, Cin, H, W, generator=g)
    w = torch.randn(G, Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)
    b = torch.randn(G, Cout, generator=g)
    res = torch.randn(G, B, Cout, H, W, generator=g) if use_res else None
    ref = torch.stack([F.conv2d(x[i], w[i], b[i], padding=dil, dilation=dil) for i in range(G)])
    if res is not None:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    fp32 = _wino4_conv(device, x, w, b, res, dil, relu)
    got = _wino4_conv(device, x, w, b, res, dil, relu, algo=_native.CONV_WINOGRAD4_BF16X3)
    scale = max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() < 2e-4 * scale
    assert (got - fp32).abs().max().item() < 2e-5 * scale, (got - fp32).abs().max().item()


def test_split_precision_network_vs_oracle(device):
    """The whole two-branch network with split-precision plane GEMMs: embeddings within the north_star's 1e-3 of the oracle's
    (measured next to the fp32 path's error in the assertion message)."""
    net, sd = _net(5, device)
    fr = synth.rgbd_frame(12, 120, 88, 3)
    img, dep = torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"])
    want = BO.segnet_forward(sd, img, dep)
    net.set_split_precision(False)          # (a session run with UOC_SPLIT_GEMM=1 builds its networks in split mode)
    e32 = (net(img.to(device), None, dep.to(device)).cpu() - want).abs().max().item()
    try:
        net.set_split_precision(True)
        e3 = (net(img.to(device), None, dep.to(device)).cpu() - want).abs().max().item()
    finally:
        net.set_split_precision(False)
    back = (net(img.to(device), None, dep.to(device)).cpu() - want).abs().max().item()
    net.set_split_precision(bool(cfg.TEST.SPLIT_PRECISION_GEMM))
    assert e3 < EMBED_TOL and back == e32, (e32, e3, back)
    assert e3 < 10 * max(e32, 1e-6), f"split precision {e3:.2e} vs fp32 {e32:.2e}"
