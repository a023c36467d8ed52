"""Input preparation of the demo driver — mirror of tools/test_images.py:96-135 in the reference
(`compute_xyz`, `read_sample`): BGR uint8 image -> /255 - PIXEL_MEANS/255 -> [1,3,H,W]; uint16
millimetre depth -> metres -> XYZ point-cloud image with the pinhole model -> [1,3,H,W].
PIL replaces cv2 (not installed here): PIL decodes RGB, cv2.imread returns BGR, hence the flip."""
from __future__ import annotations

import numpy as np
import torch

from .fcn.config import cfg


def compute_xyz(depth_img, fx, fy, px, py, height, width):
    """tools/test_images.py:96-102.  depth_img [H,W] float32 metres -> [H,W,3] (x, y, z)."""
    indices = np.indices((height, width), dtype=np.float32).transpose(1, 2, 0)   # [..., 0] = y, [..., 1] = x
    z_e = depth_img
    x_e = (indices[..., 1] - px) * z_e / fx
    y_e = (indices[..., 0] - py) * z_e / fy
    return np.stack([x_e, y_e, z_e], axis=-1)


def load_images(filename_color, filename_depth):
    from PIL import Image
    im = np.asarray(Image.open(filename_color).convert("RGB"))[:, :, ::-1].copy()     # BGR like cv2.imread (:108)
    depth_img = np.asarray(Image.open(filename_depth)) if filename_depth is not None else None   # uint16 mm (:112)
    return im, depth_img


def make_sample(im_bgr_u8, depth_u16, camera_params):
    """tools/test_images.py:112-133 on decoded arrays."""
    depth = depth_u16.astype(np.float32) / 1000.0 if cfg.INPUT in ("DEPTH", "RGBD") else None
    return make_sample_metric(im_bgr_u8, depth, camera_params["fx"], camera_params["fy"], camera_params["x_offset"],
                              camera_params["y_offset"])


def make_sample_metric(im_bgr_u8, depth_m, fx, fy, px, py):
    """The same from a depth image already in metres (float32) — what the ROS node holds after decoding a 32FC1 /
    16UC1 message (ros/test_images_segmentation.py:139-155 runs the same arithmetic as tools/test_images.py:112-133)."""
    sample = {}
    if cfg.INPUT in ("DEPTH", "RGBD"):
        height, width = depth_m.shape
        xyz_img = compute_xyz(depth_m, fx, fy, px, py, height, width)
        sample["depth"] = torch.from_numpy(xyz_img).permute(2, 0, 1).unsqueeze(0)
    im_tensor = torch.from_numpy(im_bgr_u8) / 255.0
    im_tensor -= torch.tensor(cfg.PIXEL_MEANS / 255.0).float()
    sample["image_color"] = im_tensor.permute(2, 0, 1).unsqueeze(0)
    return sample


def read_sample(filename_color, filename_depth, camera_params):
    """tools/test_images.py:105-135."""
    im, depth_img = load_images(filename_color, filename_depth)
    return make_sample(im, depth_img, camera_params)


def read_sample_raw(filename_color, filename_depth, camera_params):
    """Raw variant of read_sample for the fused device-side input preparation: the
    sample carries the decoded uint8 BGR image and the uint16 millimetre depth (1.5 MB instead of 7.4 MB
    of float tensors); test_sample turns them into the network inputs with one kernel (uoc_prep_rgbd)."""
    im, depth_img = load_images(filename_color, filename_depth)
    return make_sample_raw(im, depth_img, camera_params)


def make_sample_raw(im_bgr_u8, depth_u16, camera_params):
    return {"image_u8": torch.from_numpy(np.ascontiguousarray(im_bgr_u8)),
            "depth_u16": torch.from_numpy(np.ascontiguousarray(depth_u16).astype(np.int32).astype(np.uint16).view(np.int16)),
            "camera": dict(camera_params)}


def prepare_on_device(sample, device, out=None):
    """image_u8 [H,W,3] uint8 + depth_u16 [H,W] (int16 view of uint16) -> float 'image_color' / 'depth'
    [1,3,H,W] on `device`, bit-identical to make_sample().  out = (image [3,H,W], xyz [3,H,W]) contiguous float32 device
    tensors to write into (a launch set's batched input, so that N frames need no torch.cat)."""
    from . import _native
    # non_blocking: asynchronous for pinned host buffers (a capture pipeline's; the caller must not rewrite them before the
    # frame is done), a plain synchronous copy for pageable ones.  Without it every upload also waits for the stream to drain —
    # a host stall per frame that the event-driven runner cannot hide (round 6: the PCIe-inclusive leg's 9 % gap)
    bgr = sample["image_u8"].to(device, non_blocking=True).contiguous()
    dep = sample["depth_u16"].to(device, non_blocking=True).contiguous()
    H, W = dep.shape
    cam = sample["camera"]
    mean = (cfg.PIXEL_MEANS.reshape(-1) / 255.0).astype(np.float32)
    if out is not None:
        image, xyz = out
        assert image.shape[-3:] == (3, H, W) and xyz.shape[-3:] == (3, H, W) and image.is_contiguous() and xyz.is_contiguous()
    else:
        image = torch.empty((1, 3, H, W), dtype=torch.float32, device=device)
        xyz = torch.empty((1, 3, H, W), dtype=torch.float32, device=device)
    f32 = lambda v: float(np.float32(v))
    with torch.cuda.device(device):
        rc = _native.lib().uoc_prep_rgbd(_native.ptr(bgr), _native.ptr(dep), H, W, f32(cam["fx"]), f32(cam["fy"]),
                                         f32(cam["x_offset"]), f32(cam["y_offset"]), float(mean[0]), float(mean[1]),
                                         float(mean[2]), _native.ptr(image), _native.ptr(xyz), _native.stream_ptr(device))
    _native.check(rc, "uoc_prep_rgbd")
    return {"image_color": image, "depth": xyz}
