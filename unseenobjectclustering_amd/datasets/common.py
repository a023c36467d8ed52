"""What the OCID and OSD loaders share (/root/reference/lib/datasets/ocid_object.py:53-110, osd_object.py:45-100):
BGR image -> network input, indexed-PNG labels -> {0..K-1}, organised point cloud -> XYZ image."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.utils.data as data

from ..fcn.config import cfg
from .imdb import imdb
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


class SceneDataset(data.Dataset, imdb):
    """What OCID and OSD share (ocid_object.py:69-112, osd_object.py:61-100): a list of colour images; per item the
    network-input colour blobs, the compacted foreground labels, the path relative to the dataset root and — for
    DEPTH / RGBD input — the organised cloud as an XYZ image.  Subclasses say where things are (`_label_file`,
    `_cloud_file`), which ids are background (`_drop_background`) and what the root marker in a path is (`_marker`)."""

    _marker = ""
    _width, _height = 640, 480

    def _setup(self, name, root, image_files):
        imdb.__init__(self)
        self._name = name
        self._root = root
        self._classes_all = ("__background__", "foreground")
        self._classes = self._classes_all
        self._pixel_mean = pixel_mean()
        self._files = [str(f) for f in image_files]
        print("%d images for dataset %s" % (len(self._files), self._name))
        self._size = len(self._files)
        assert os.path.exists(root), "{} path does not exist: {}".format(name.rsplit("_", 1)[0], root)

    process_label = staticmethod(process_label)

    def _drop_background(self, labels, labels_filename):
        return labels

    def __getitem__(self, idx):
        if cfg.MODE == "TRAIN":
            raise NotImplementedError("training-time augmentation is out of scope (inference / evaluation only)")
        filename = self._files[idx]
        image_blob, bgr_blob = image_blobs(imread_bgr(filename), self._pixel_mean)
        labels_filename = self._label_file(filename)
        labels = self._drop_background(imread_indexed(labels_filename), labels_filename)
        cut = filename.find(self._marker) + len(self._marker) + 1
        sample = {"image_color": image_blob, "image_color_bgr": bgr_blob,
                  "label": torch.from_numpy(self.process_label(labels)).unsqueeze(0), "filename": filename[cut:]}
        if cfg.INPUT in ("DEPTH", "RGBD"):
            sample["depth"] = xyz_blob(self._cloud_file(filename), self._height, self._width)
        return sample

    def __len__(self):
        return self._size
