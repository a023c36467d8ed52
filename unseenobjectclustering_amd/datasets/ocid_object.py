"""OCID loader — mirror of /root/reference/lib/datasets/ocid_object.py:23-125 for the evaluation path (cfg.MODE
'TEST': the training-time colour augmentations :75-78 are not implemented), with PIL in place of cv2 and the repo's PCD
reader in place of python-pcl.  Directory layout: <root>/**/<...seq...>/{rgb,label,pcd}/<frame>.{png,png,pcd}."""
from __future__ import annotations

import os
from pathlib import Path

import torch
import torch.utils.data as data

from ..fcn.config import cfg
from . import common
from .imdb import imdb


class OCIDObject(data.Dataset, imdb):
    def __init__(self, image_set, ocid_object_path=None):
        imdb.__init__(self)
        self._name = "ocid_object_" + image_set
        self._image_set = image_set
        self._ocid_object_path = self._get_default_path() if ocid_object_path is None else ocid_object_path
        self._classes_all = ("__background__", "foreground")
        self._classes = self._classes_all
        self._pixel_mean = common.pixel_mean()
        self._width = 640
        self._height = 480
        self.image_paths = self.list_dataset()
        print("%d images for dataset %s" % (len(self.image_paths), self._name))
        self._size = len(self.image_paths)
        assert os.path.exists(self._ocid_object_path), "ocid_object path does not exist: {}".format(self._ocid_object_path)

    def list_dataset(self):
        """:43-51 — every directory whose name contains 'seq', in glob order; its rgb/*.png sorted."""
        seqs = list(Path(self._ocid_object_path).glob("**/*seq*"))
        image_paths = []
        for seq in seqs:
            image_paths += sorted(list((seq / "rgb").glob("*.png")))
        return image_paths

    process_label = staticmethod(common.process_label)

    def __getitem__(self, idx):
        if cfg.MODE == "TRAIN":
            raise NotImplementedError("training-time augmentation is out of scope (inference / evaluation only)")
        filename = str(self.image_paths[idx])
        image_blob, im_tensor_bgr = common.image_blobs(common.imread_bgr(filename), self._pixel_mean)
        labels_filename = filename.replace("rgb", "label")
        foreground_labels = common.imread_indexed(labels_filename)
        foreground_labels[foreground_labels == 1] = 0                 # :91 mask table as background
        if "table" in labels_filename:
            foreground_labels[foreground_labels == 2] = 0             # :92-93
        label_blob = torch.from_numpy(self.process_label(foreground_labels)).unsqueeze(0)
        index = filename.find("OCID")
        sample = {"image_color": image_blob, "image_color_bgr": im_tensor_bgr, "label": label_blob,
                  "filename": filename[index + 5:]}
        if cfg.INPUT == "DEPTH" or cfg.INPUT == "RGBD":
            pcd_filename = filename.replace("rgb", "pcd").replace("png", "pcd")       # :103-104
            sample["depth"] = common.xyz_blob(pcd_filename, self._height, self._width)
        return sample

    def __len__(self):
        return self._size

    def _get_default_path(self):
        from . import ROOT_DIR
        return os.path.join(ROOT_DIR, "data", "OCID")
