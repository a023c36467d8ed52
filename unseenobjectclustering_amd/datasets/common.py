"""What the OCID and OSD loaders share (/root/reference/lib/datasets/ocid_object.py:53-110, osd_object.py:45-100):
BGR image -> network input, indexed-PNG labels -> {0..K-1}, organised point cloud -> XYZ image."""
from __future__ import annotations

import numpy as np
import torch

from ..fcn.config import cfg
from .pcd import load_xyz


def imread_bgr(filename):
    """cv2.imread(filename): BGR uint8 [H,W,3] (PIL decodes RGB)."""
    from PIL import Image
    return np.asarray(Image.open(filename).convert("RGB"))[:, :, ::-1].copy()


def imread_indexed(filename):
    """lib/utils/mask.py:152-156: the palette INDICES of an indexed PNG (or the grey values of a plain one)."""
    from PIL import Image
    return np.array(Image.open(filename))


def process_label(foreground_labels):
    """ocid_object.py:53-67 / osd_object.py:45-59: map the label values present to {0, ..., K-1} in ascending order."""
    values = np.unique(foreground_labels)
    mapped = foreground_labels.copy()
    for k in range(values.shape[0]):
        mapped[foreground_labels == values[k]] = k
    return mapped


def image_blobs(im_bgr, pixel_mean):
    """ocid_object.py:78-84: (image_color [3,H,W] = BGR/255 - mean, image_color_bgr [3,H,W] = BGR/255)."""
    im_tensor = torch.from_numpy(im_bgr) / 255.0
    im_tensor_bgr = im_tensor.clone().permute(2, 0, 1)
    im_tensor -= pixel_mean
    return im_tensor.permute(2, 0, 1), im_tensor_bgr


def xyz_blob(pcd_filename, height, width):
    """ocid_object.py:105-108: organised cloud -> [3,H,W] float32, NaN -> 0."""
    pcloud = load_xyz(pcd_filename)
    pcloud[np.isnan(pcloud)] = 0
    return torch.from_numpy(pcloud.reshape((height, width, 3))).permute(2, 0, 1)


def pixel_mean():
    return torch.tensor(cfg.PIXEL_MEANS / 255.0).float()
