"""Evaluation datasets of the reference (lib/datasets/): OCID and OSD, python-pcl / OpenCV free."""
import os.path as osp

from .imdb import imdb
from .ocid_object import OCIDObject
from .osd_object import OSDObject
from .factory import get_dataset, list_datasets

# like lib/datasets/__init__.py:10-11: the checkout root, under which data/OCID and data/OSD are expected
ROOT_DIR = osp.join(osp.dirname(__file__), "..", "..")
