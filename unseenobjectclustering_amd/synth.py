"""Repo-owned deterministic synthetic inputs.

Nothing here comes from the reference: these generators exist so that the
golden-vector script (tests/golden/make_golden.py, run in the build container
against /root/reference) and the GPU-side parity tests / bench.py can
regenerate *bit-identical* inputs from a seed without shipping 78 MB tensors.

All generators use ``numpy.random.default_rng(seed)`` (PCG64) only.
"""
from __future__ import annotations

import numpy as np

# camera intrinsics of the reference's demo data (data/demo/camera_params.json)
DEMO_CAMERA = dict(fx=612.937, fy=613.173, x_offset=322.549, y_offset=248.158)
# cfg.PIXEL_MEANS of the reference (lib/fcn/config.py:376), BGR order
PIXEL_MEANS = np.array([102.9801, 115.9465, 122.7717], dtype=np.float64)


def tabletop_label_map(seed: int, height: int = 480, width: int = 640,
                       num_objects: int = 5) -> np.ndarray:
    """Background (0) + table plane (1) + ``num_objects`` objects (2..K+1).

    Objects are axis-aligned boxes or ellipses with half-extent 30..90 px
    (scaled with the image size), later objects paint over earlier ones.
    Returns an int32 [H, W] map.
    """
    rng = np.random.default_rng(1000003 * seed + 17)
    lab = np.zeros((height, width), dtype=np.int32)
    # table: lower ~70 % of the image
    lab[int(0.28 * height):, :] = 1
    yy, xx = np.mgrid[0:height, 0:width]
    s = min(height / 480.0, width / 640.0)
    for k in range(num_objects):
        cy = rng.uniform(0.35 * height, 0.85 * height)
        cx = rng.uniform(0.12 * width, 0.88 * width)
        hy = rng.uniform(30, 90) * s
        hx = rng.uniform(30, 90) * s
        if rng.random() < 0.5:
            m = (np.abs(yy - cy) <= hy) & (np.abs(xx - cx) <= hx)
        else:
            m = ((yy - cy) / hy) ** 2 + ((xx - cx) / hx) ** 2 <= 1.0
        lab[m] = k + 2
    return lab


def embedding_field(seed: int, height: int = 480, width: int = 640, channels: int = 64,
                    num_objects: int = 7, noise: float = 0.05):
    """Structured unit-norm embedding field for kernel-level clustering tests.

    X[p] = normalize(centre[label[p]] + noise * N(0, 1)), centres = random unit
    vectors in R^C.  Returns (X [H*W, C] float32 C-contiguous, labels [H, W] int32).
    """
    lab = tabletop_label_map(seed, height, width, num_objects)
    rng = np.random.default_rng(7919 * seed + 3)
    k = int(lab.max()) + 1
    centres = rng.standard_normal((k, channels)).astype(np.float32)
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    x = centres[lab.reshape(-1)]
    x = x + np.float32(noise) * rng.standard_normal(x.shape, dtype=np.float32)
    x /= np.sqrt((x * x).sum(axis=1, keepdims=True, dtype=np.float32))
    return np.ascontiguousarray(x, dtype=np.float32), lab


def rgbd_frame(seed: int, height: int = 480, width: int = 640, num_objects: int = 5,
               hole_fraction: float = 0.05):
    """Tabletop-style RGB-D frame in the reference's network-input convention.

    Returns dict with
      image_color [1,3,H,W] float32  BGR/255 - PIXEL_MEANS/255 (tools/test_images.py:125-129)
      depth       [1,3,H,W] float32  XYZ metres, z = 0 at holes (tools/test_images.py:96-102)
      label       [H,W]      int32   generating label map (not consumed by the path)
    """
    lab = tabletop_label_map(seed, height, width, num_objects)
    rng = np.random.default_rng(104729 * seed + 11)
    k = int(lab.max()) + 1
    base = rng.uniform(0.1, 0.9, size=(k, 3)).astype(np.float32)        # BGR in [0,1]
    img = base[lab] + np.float32(0.05) * rng.standard_normal((height, width, 3), dtype=np.float32)
    img = np.clip(img, 0.0, 1.0).astype(np.float32)
    img = img - (PIXEL_MEANS / 255.0).astype(np.float32)

    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    z = np.float32(0.5) + np.float32(1.0) * (np.float32(height) - yy) / np.float32(height)  # plane 0.5..1.5 m
    off = rng.uniform(0.03, 0.25, size=k).astype(np.float32)
    off[:2] = 0.0
    z = z - off[lab]
    z[lab == 0] = np.float32(1.8)
    z = z + np.float32(0.002) * rng.standard_normal((height, width), dtype=np.float32)
    holes = rng.random((height, width)) < hole_fraction
    z[holes] = 0.0
    z = z.astype(np.float32)
    cam = DEMO_CAMERA
    sx, sy = width / 640.0, height / 480.0
    x = (xx - np.float32(cam["x_offset"] * sx)) * z / np.float32(cam["fx"] * sx)
    y = (yy - np.float32(cam["y_offset"] * sy)) * z / np.float32(cam["fy"] * sy)
    xyz = np.stack([x, y, z], axis=0).astype(np.float32)
    return dict(image_color=np.ascontiguousarray(img.transpose(2, 0, 1))[None],
                depth=np.ascontiguousarray(xyz)[None], label=lab)


# ---------------------------------------------------------------------------------------------
# Synthetic ResNet34-8s weights.  Key names/shapes follow the reference state-dict contract
# (fcn.resnet34_8s.* / fcn_depth.resnet34_8s.*).
# ---------------------------------------------------------------------------------------------

RESNET34_BLOCKS = (3, 4, 6, 3)
RESNET34_PLANES = (64, 128, 256, 512)


def resnet34_8s_param_shapes(num_units: int = 64, in_channels: int = 3):
    """Ordered list of (name, shape) for ONE Resnet34_8s, names relative to 'resnet34_8s.'.

    Mirrors the module tree of lib/networks/resnet.py:116-234 + resnet_dilated.py:287-303:
    conv1/bn1, layer{1..4}.{i}.{conv1,bn1,conv2,bn2[,downsample.0,downsample.1]}, fc (1x1 conv + bias).
    """
    out = []

    def bn(prefix, c):
        out.append((prefix + ".weight", (c,)))
        out.append((prefix + ".bias", (c,)))
        out.append((prefix + ".running_mean", (c,)))
        out.append((prefix + ".running_var", (c,)))
        out.append((prefix + ".num_batches_tracked", ()))

    out.append(("conv1.weight", (64, in_channels, 7, 7)))
    bn("bn1", 64)
    inpl = 64
    for li, (nb, planes) in enumerate(zip(RESNET34_BLOCKS, RESNET34_PLANES), start=1):
        for bi in range(nb):
            p = f"layer{li}.{bi}"
            out.append((p + ".conv1.weight", (planes, inpl if bi == 0 else planes, 3, 3)))
            bn(p + ".bn1", planes)
            out.append((p + ".conv2.weight", (planes, planes, 3, 3)))
            bn(p + ".bn2", planes)
            if bi == 0 and inpl != planes:
                out.append((p + ".downsample.0.weight", (planes, inpl, 1, 1)))
                bn(p + ".downsample.1", planes)
        inpl = planes
    out.append(("fc.weight", (num_units, 512, 1, 1)))
    out.append(("fc.bias", (num_units,)))
    return out


def synthetic_state_dict(seed: int, num_units: int = 64, branches=("fcn", "fcn_depth"), in_channels: int = 3):
    """Deterministic synthetic weights for the two-branch RGB-D net, as numpy arrays
    (branches=("fcn",) for the COLOR / DEPTH / early-fusion nets, in_channels=6 for early fusion).

    Conv weights: N(0, 2/(fan_in+fan_out)) (xavier-normal, what SEG.py:77-85 applies);
    BatchNorm: NON-trivial gamma/beta/running stats so that BN folding is exercised
    (gamma~U(0.5,1.5), beta~N(0,0.1), mean~N(0,0.1), var~U(0.5,1.5));
    fc bias ~ N(0, 0.05).
    """
    sd = {}
    for bidx, br in enumerate(branches):
        rng = np.random.default_rng(15485863 * seed + 101 * bidx + 5)
        for name, shape in resnet34_8s_param_shapes(num_units, in_channels):
            key = f"{br}.resnet34_8s.{name}"
            if name.endswith("num_batches_tracked"):
                sd[key] = np.array(1, dtype=np.int64)
            elif len(shape) == 4:
                fan_in = shape[1] * shape[2] * shape[3]
                fan_out = shape[0] * shape[2] * shape[3]
                std = np.sqrt(2.0 / (fan_in + fan_out))
                sd[key] = (std * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
            elif name.endswith("running_var") or (".bn" in name or "bn1" in name or "downsample.1" in name) and name.endswith(".weight"):
                sd[key] = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
            elif name == "fc.bias":
                sd[key] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
            else:  # bn bias, running_mean
                sd[key] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
    return sd


# ---------------------------------------------------------------------------------------------
# Benchmark workload: palette-coloured tabletop frames + "calibrated" synthetic weights.
#
# There are no trained checkpoints in the build/bench environment and a randomly initialised
# network maps every pixel to (almost) the same embedding, which would leave stage 2 of the
# pipeline without any ROI.  For end-to-end benchmarks the synthetic weights are therefore
# *calibrated* (scripts/make_calibrated_weights.py, CPU oracle): random backbone, BatchNorm
# statistics measured on synthetic frames, and the two 1x1 `fc` layers fitted in closed form
# (ridge regression) so that each palette colour maps to its own embedding direction.  Same
# architecture, same FLOPs, dense non-zero weights; the frames then segment into their objects.
# ---------------------------------------------------------------------------------------------

PALETTE_BGR = np.array([
    [0.15, 0.15, 0.15], [0.55, 0.50, 0.45],                       # background, table
    [0.85, 0.15, 0.15], [0.15, 0.85, 0.15], [0.15, 0.15, 0.85], [0.85, 0.85, 0.15],
    [0.85, 0.15, 0.85], [0.15, 0.85, 0.85], [0.90, 0.90, 0.90], [0.90, 0.50, 0.10],
    [0.10, 0.50, 0.90], [0.50, 0.90, 0.10],
], dtype=np.float32)


def palette_frame(seed: int, height: int = 480, width: int = 640, num_objects: int = 5,
                  hole_fraction: float = 0.05, colour_noise: float = 0.03):
    """Like rgbd_frame but colours come from PALETTE_BGR (background 0, table 1, objects drawn
    without replacement from the rest).  Extra key 'palette' [H,W] int32 = palette index per pixel."""
    assert num_objects <= len(PALETTE_BGR) - 2
    fr = rgbd_frame(seed, height, width, num_objects, hole_fraction)
    lab = fr["label"]
    rng = np.random.default_rng(611953 * seed + 29)
    k = int(lab.max()) + 1
    pal = np.zeros(k, dtype=np.int32)
    pal[:2] = [0, 1][:k]
    if k > 2:
        pal[2:] = 2 + rng.permutation(len(PALETTE_BGR) - 2)[:k - 2]
    img = PALETTE_BGR[pal][lab] + np.float32(colour_noise) * rng.standard_normal((height, width, 3), dtype=np.float32)
    img = np.clip(img, 0.0, 1.0).astype(np.float32) - (PIXEL_MEANS / 255.0).astype(np.float32)
    fr["image_color"] = np.ascontiguousarray(img.transpose(2, 0, 1))[None]
    fr["palette"] = pal[lab]
    return fr


def _gauss7(sigma: float) -> np.ndarray:
    a = np.arange(7, dtype=np.float64) - 3.0
    g = np.exp(-(a[:, None] ** 2 + a[None, :] ** 2) / (2 * sigma * sigma))
    return (g / g.sum()).astype(np.float32)


def uncalibrated_bench_state_dict(seed: int = 0):
    """Random backbone used as the starting point of the calibration: xavier-normal 3x3/1x1 convs,
    stem = random 64x3 colour mix (x) 7x7 Gaussian window (sigma 1.5), BN gamma 1 (0.1 on every
    block's bn2, the usual zero-ish residual init), beta 0, stats 0/1, fc from the generator."""
    sd = synthetic_state_dict(1000 + seed)
    rng = np.random.default_rng(2750159 * seed + 7)
    for br in ("fcn", "fcn_depth"):
        mix = rng.standard_normal((64, 3)).astype(np.float32)
        sd[f"{br}.resnet34_8s.conv1.weight"] = (mix[:, :, None, None] * _gauss7(1.5)[None, None]).astype(np.float32)
    for k in list(sd):
        is_bn = (".bn" in k or "resnet34_8s.bn1" in k or "downsample.1" in k) and sd[k].ndim == 1
        if not is_bn:
            continue
        if k.endswith("running_mean") or k.endswith(".bias"):
            sd[k] = np.zeros_like(sd[k])
        elif k.endswith("running_var"):
            sd[k] = np.ones_like(sd[k])
        elif k.endswith(".weight"):
            sd[k] = np.full_like(sd[k], 0.1 if k.endswith("bn2.weight") else 1.0)
    return sd


def calibrated_state_dict(path: str = None):
    """uncalibrated_bench_state_dict(0) + the committed calibration (BN stats, fitted fc layers)."""
    import os
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "bench_calibration.npz")
    sd = uncalibrated_bench_state_dict(0)
    cal = np.load(path)
    for k in cal.files:
        assert k in sd and sd[k].shape == cal[k].shape, k
        sd[k] = cal[k].astype(np.float32)
    return sd
