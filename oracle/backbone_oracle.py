"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the reference's RGB-D
ResNet34-8s embedding forward pass with plain torch CPU ops, driven by a flat state dict.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinned against the reference's own SEGNET.forward by tests/golden/backbone.npz and modes.npz.

Reference: lib/networks/SEG.py:88-119 (RGBD 'add' / 'early', COLOR, DEPTH), lib/networks/resnet_dilated.py:287-327,
lib/networks/resnet.py:116-270.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BLOCKS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
BN_EPS = 1e-5


def _t(sd, key):
    v = sd[key]
    return v if torch.is_tensor(v) else torch.as_tensor(v)


def _bn(sd, pfx, x):
    return F.batch_norm(x, _t(sd, pfx + ".running_mean"), _t(sd, pfx + ".running_var"), _t(sd, pfx + ".weight"),
                        _t(sd, pfx + ".bias"), training=False, eps=BN_EPS)


def resnet34_8s(sd, pfx, x):
    """One Resnet34_8s: returns (features at 1/8 resolution after fc, upsampled features)."""
    size = x.shape[2:]
    x = F.relu(_bn(sd, pfx + "bn1", F.conv2d(x, _t(sd, pfx + "conv1.weight"), stride=2, padding=3)))   # resnet.py:237-239
    x = F.max_pool2d(x, 3, 2, 1)                                                                           # :240
    inpl, cur_stride, cur_dil = 64, 4, 1
    for li, (nb, planes) in enumerate(zip(BLOCKS, PLANES), start=1):
        stride = 1 if li == 1 else 2
        down = stride != 1 or inpl != planes
        if down:
            if cur_stride == 8:          # resnet.py:201-206: output stride reached -> dilate instead
                cur_dil *= stride
                stride = 1
            else:
                cur_stride *= stride
        for bi in range(nb):
            p = f"{pfx}layer{li}.{bi}."
            s = stride if bi == 0 else 1
            out = F.relu(_bn(sd, p + "bn1", F.conv2d(x, _t(sd, p + "conv1.weight"), stride=s, padding=cur_dil,
                                                     dilation=cur_dil)))
            out = _bn(sd, p + "bn2", F.conv2d(out, _t(sd, p + "conv2.weight"), padding=cur_dil, dilation=cur_dil))
            res = x
            if bi == 0 and down:
                res = _bn(sd, p + "downsample.1", F.conv2d(x, _t(sd, p + "downsample.0.weight"), stride=s))
            x = F.relu(out + res)                                                                          # resnet.py:57-73
        inpl = planes
    x = F.conv2d(x, _t(sd, pfx + "fc.weight"), _t(sd, pfx + "fc.bias"))                                    # resnet_dilated.py:303
    up = F.interpolate(x, size=size, mode="bilinear", align_corners=True)                                  # :325 (upsample_bilinear)
    return x, up


def segnet_forward(sd, img, depth, mode="RGBD_ADD"):
    """SEG.py:97-114 -> [B,64,H,W] unit-norm.  mode: 'RGBD_ADD' normalize(fcn(img) + fcn_depth(depth)) (:106-108),
    'COLOR' fcn(img) (:100), 'DEPTH' fcn(depth) (:98), 'RGBD_EARLY' fcn(cat(img, depth)) (:102-103),
    'RGBD_CAT' normalize(cat(fcn(img), fcn_depth(depth))) -> [B,128,H,W] (:109-110)."""
    with torch.no_grad():
        if mode in ("RGBD_ADD", "RGBD_CAT"):
            _, a = resnet34_8s(sd, "fcn.resnet34_8s.", img)
            _, b = resnet34_8s(sd, "fcn_depth.resnet34_8s.", depth)
            a = a + b if mode == "RGBD_ADD" else torch.cat((a, b), 1)            # :107-110
        elif mode == "COLOR":
            _, a = resnet34_8s(sd, "fcn.resnet34_8s.", img)
        elif mode == "DEPTH":
            _, a = resnet34_8s(sd, "fcn.resnet34_8s.", depth)
        elif mode == "RGBD_EARLY":
            _, a = resnet34_8s(sd, "fcn.resnet34_8s.", torch.cat((img, depth), 1))
        else:
            raise ValueError(mode)
        return F.normalize(a, p=2, dim=1)
