"""OCID loader — counterpart of /root/reference/lib/datasets/ocid_object.py:23-125 for the evaluation path (cfg.MODE
'TEST'; the training-time colour augmentations :75-78 are not implemented), PIL in place of cv2, datasets/pcd.py in
place of python-pcl.  Layout: <root>/**/<dir with 'seq' in its name>/{rgb,label,pcd}/<frame>.{png,png,pcd}."""
from __future__ import annotations

import os
from pathlib import Path

from .common import SceneDataset


class OCIDObject(SceneDataset):
    _marker = "OCID"

    def __init__(self, image_set, ocid_object_path=None):
        self._image_set = image_set
        self._ocid_object_path = self._get_default_path() if ocid_object_path is None else ocid_object_path
        self._setup("ocid_object_" + image_set, self._ocid_object_path, self.list_dataset())
        self.image_paths = self._files

    def list_dataset(self):
        """:43-51 — every directory whose name contains 'seq', in glob order; its rgb/*.png sorted."""
        frames = []
        for seq in Path(self._ocid_object_path).glob("**/*seq*"):
            frames += sorted((seq / "rgb").glob("*.png"))
        return frames

    def _label_file(self, filename):
        return filename.replace("rgb", "label")                              # :88

    def _cloud_file(self, filename):
        return filename.replace("rgb", "pcd").replace("png", "pcd")          # :103-104

    def _drop_background(self, labels, labels_filename):
        labels[labels == 1] = 0                                              # :91  the table is background
        if "table" in labels_filename:
            labels[labels == 2] = 0                                          # :92-93  ... and id 2 under a 'table' directory
        return labels

    def _get_default_path(self):
        from . import ROOT_DIR
        return os.path.join(ROOT_DIR, "data", "OCID")
