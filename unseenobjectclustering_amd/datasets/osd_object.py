"""OSD loader — mirror of /root/reference/lib/datasets/osd_object.py:21-113 for the evaluation path, with PIL in
place of cv2 and the repo's PCD reader in place of python-pcl.
Directory layout: <root>/{image_color,annotation,pcd}/<frame>.{png,png,pcd}."""
from __future__ import annotations

import glob
import os

import torch
import torch.utils.data as data

from ..fcn.config import cfg
from . import common
from .imdb import imdb


class OSDObject(data.Dataset, imdb):
    def __init__(self, image_set, osd_object_path=None):
        imdb.__init__(self)
        self._name = "osd_object_" + image_set
        self._image_set = image_set
        self._osd_object_path = self._get_default_path() if osd_object_path is None else osd_object_path
        self._classes_all = ("__background__", "foreground")
        self._classes = self._classes_all
        self._pixel_mean = common.pixel_mean()
        self._width = 640
        self._height = 480
        data_path = os.path.join(self._osd_object_path, "image_color")
        self.image_files = sorted(glob.glob(data_path + "/*.png"))
        print("%d images for dataset %s" % (len(self.image_files), self._name))
        self._size = len(self.image_files)
        assert os.path.exists(self._osd_object_path), "osd_object path does not exist: {}".format(self._osd_object_path)

    process_label = staticmethod(common.process_label)

    def __getitem__(self, idx):
        if cfg.MODE == "TRAIN":
            raise NotImplementedError("training-time augmentation is out of scope (inference / evaluation only)")
        filename = self.image_files[idx]
        image_blob, im_tensor_bgr = common.image_blobs(common.imread_bgr(filename), self._pixel_mean)
        labels_filename = filename.replace("image_color", "annotation")
        foreground_labels = self.process_label(common.imread_indexed(labels_filename))
        label_blob = torch.from_numpy(foreground_labels).unsqueeze(0)
        index = filename.find("OSD")
        sample = {"image_color": image_blob, "image_color_bgr": im_tensor_bgr, "label": label_blob,
                  "filename": filename[index + 4:]}
        if cfg.INPUT == "DEPTH" or cfg.INPUT == "RGBD":
            pcd_filename = filename.replace("image_color", "pcd").replace("png", "pcd")
            sample["depth"] = common.xyz_blob(pcd_filename, self._height, self._width)
        return sample

    def __len__(self):
        return self._size

    def _get_default_path(self):
        from . import ROOT_DIR
        return os.path.join(ROOT_DIR, "data", "OSD")
