"""OSD loader — counterpart of /root/reference/lib/datasets/osd_object.py:21-113 for the evaluation path, PIL in place
of cv2, datasets/pcd.py in place of python-pcl.  Layout: <root>/{image_color,annotation,pcd}/<frame>.{png,png,pcd}."""
from __future__ import annotations

import glob
import os

from .common import SceneDataset


class OSDObject(SceneDataset):
    _marker = "OSD"

    def __init__(self, image_set, osd_object_path=None):
        self._image_set = image_set
        self._osd_object_path = self._get_default_path() if osd_object_path is None else osd_object_path
        self._setup("osd_object_" + image_set, self._osd_object_path,
                    sorted(glob.glob(os.path.join(self._osd_object_path, "image_color") + "/*.png")))   # :36-37
        self.image_files = self._files

    def _label_file(self, filename):
        return filename.replace("image_color", "annotation")                 # :77

    def _cloud_file(self, filename):
        return filename.replace("image_color", "pcd").replace("png", "pcd")  # :90-91

    def _get_default_path(self):
        from . import ROOT_DIR
        return os.path.join(ROOT_DIR, "data", "OSD")
