"""Fits the calibration of the synthetic benchmark weights on the CPU (torch ops, no reference
code involved) and writes unseenobjectclustering_amd/data/bench_calibration.npz.

    python scripts/make_calibrated_weights.py

1. BN running statistics := per-channel statistics of each BN input over the calibration set
   (full frames and 224x224 object crops of palette frames), layer by layer.
2. fc (both branches) := ridge regression from the concatenated 1/8-resolution features of the
   two branches to a one-hot code of the palette colour under the cell.
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unseenobjectclustering_amd import synth  # noqa: E402

BLOCKS, PLANES = (3, 4, 6, 3), (64, 128, 256, 512)


def features_calibrating(sd, pfx, xs, calibrate):
    """Forward a list of inputs through one branch up to (not including) fc.  With calibrate=True
    every BN's running stats are first set to the pooled statistics of its input."""
    def bn(p, ys):
        if calibrate:
            cat = torch.cat([y.permute(1, 0, 2, 3).reshape(y.shape[1], -1) for y in ys], dim=1)
            sd[p + ".running_mean"] = cat.mean(dim=1)
            sd[p + ".running_var"] = cat.var(dim=1, unbiased=False)
        return [F.batch_norm(y, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                             training=False, eps=1e-5) for y in ys]
    xs = [F.conv2d(x, sd[pfx + "conv1.weight"], stride=2, padding=3) for x in xs]
    xs = [F.max_pool2d(F.relu(y), 3, 2, 1) for y in bn(pfx + "bn1", xs)]
    inpl, cs, cd = 64, 4, 1
    for li, (nb, planes) in enumerate(zip(BLOCKS, PLANES), start=1):
        stride = 1 if li == 1 else 2
        down = stride != 1 or inpl != planes
        if down:
            if cs == 8:
                cd *= stride
                stride = 1
            else:
                cs *= stride
        for bi in range(nb):
            p = f"{pfx}layer{li}.{bi}."
            s = stride if bi == 0 else 1
            o = [F.conv2d(x, sd[p + "conv1.weight"], stride=s, padding=cd, dilation=cd) for x in xs]
            o = [F.relu(y) for y in bn(p + "bn1", o)]
            o = [F.conv2d(y, sd[p + "conv2.weight"], padding=cd, dilation=cd) for y in o]
            o = bn(p + "bn2", o)
            r = xs
            if bi == 0 and down:
                r = bn(p + "downsample.1", [F.conv2d(x, sd[p + "downsample.0.weight"], stride=s) for x in xs])
            xs = [F.relu(a + b) for a, b in zip(o, r)]
        inpl = planes
    return xs


def crop_bilinear(t, box, S=224):
    x0, y0, x1, y1 = box
    return F.interpolate(t[:, :, y0:y1 + 1, x0:x1 + 1], size=(S, S), mode="bilinear", align_corners=True)


def main():
    torch.set_num_threads(os.cpu_count())
    t0 = time.time()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.uncalibrated_bench_state_dict(0).items()}
    imgs, xyzs, pals = [], [], []
    for seed in range(200, 206):                                   # calibration frames (never used by bench/tests)
        fr = synth.palette_frame(seed, 480, 640, 4 + seed % 5)
        img, xyz = torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"])
        pal = torch.from_numpy(fr["palette"].astype(np.float32))[None, None]
        imgs.append(img); xyzs.append(xyz); pals.append(pal)
        lab = fr["label"]
        for obj in range(2, int(lab.max()) + 1):                   # object crops, padded 25 % like the pipeline
            ys, xs = np.nonzero(lab == obj)
            if len(ys) < 200:
                continue
            x0, x1, y0, y1 = xs.min(), xs.max(), ys.min(), ys.max()
            px, py = int(round((x1 - x0) * 0.25)), int(round((y1 - y0) * 0.25))
            box = (max(x0 - px, 0), max(y0 - py, 0), min(x1 + px, 639), min(y1 + py, 479))
            imgs.append(crop_bilinear(img, box)); xyzs.append(crop_bilinear(xyz, box))
            pals.append(F.interpolate(pal[:, :, box[1]:box[3] + 1, box[0]:box[2] + 1], size=(224, 224), mode="nearest"))
    print("calibration inputs:", len(imgs), "tensors", flush=True)
    with torch.no_grad():
        fa = features_calibrating(sd, "fcn.resnet34_8s.", imgs, True)
        fb = features_calibrating(sd, "fcn_depth.resnet34_8s.", xyzs, True)
    print("BN calibrated in %.1fs" % (time.time() - t0), flush=True)
    # ridge regression of the one-hot palette code on [fa | fb | 1]
    rows, targets = [], []
    for a, b, pal in zip(fa, fb, pals):
        h, w = a.shape[2], a.shape[3]
        cell = F.interpolate(pal, size=(h, w), mode="nearest").long().reshape(-1)
        rows.append(torch.cat([a[0].reshape(512, -1).t(), b[0].reshape(512, -1).t(), torch.ones(h * w, 1)], dim=1))
        targets.append(F.one_hot(cell, 64).float())
    A = torch.cat(rows).double()
    T = torch.cat(targets).double()
    lam = 1e-2 * A.shape[0]
    G = A.t() @ A + lam * torch.eye(A.shape[1], dtype=torch.float64)
    Wt = torch.linalg.solve(G, A.t() @ T)                          # [1025, 64]
    sd["fcn.resnet34_8s.fc.weight"] = Wt[:512].t().float().reshape(64, 512, 1, 1).contiguous()
    sd["fcn_depth.resnet34_8s.fc.weight"] = Wt[512:1024].t().float().reshape(64, 512, 1, 1).contiguous()
    sd["fcn.resnet34_8s.fc.bias"] = Wt[1024].float().contiguous()
    sd["fcn_depth.resnet34_8s.fc.bias"] = torch.zeros(64)
    pred = (A @ Wt).float()
    acc = (pred.argmax(1) == T.argmax(1)).float().mean().item()
    print("fc fitted on %d cells, argmax accuracy %.4f, %.1fs" % (A.shape[0], acc, time.time() - t0), flush=True)
    base = synth.uncalibrated_bench_state_dict(0)
    out = {k: v.numpy().astype(np.float32) for k, v in sd.items()
           if not k.endswith("num_batches_tracked") and not np.array_equal(v.numpy(), base[k])}
    path = os.path.join(ROOT, "unseenobjectclustering_amd", "data", "bench_calibration.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%d arrays, %.0f KB" % (len(out), os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
