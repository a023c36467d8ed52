"""experiments/cfgs/*.yml (inference configurations, named like the reference's experiments) parse into the mode their
name says, and experiments/scripts/*.sh reference files that exist."""
import copy
import glob
import os
import re

import pytest

from unseenobjectclustering_amd.fcn import config as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YMLS = sorted(glob.glob(os.path.join(ROOT, "experiments", "cfgs", "*.yml")))
MODE = {"rgbd_add": "RGBD_ADD", "rgbd_early": "RGBD_EARLY", "rgbd_cat": "RGBD_CAT", "color": "COLOR", "depth": "DEPTH"}


@pytest.fixture
def restore_cfg():
    saved = copy.deepcopy(dict(C.cfg))
    yield
    C.cfg.clear()
    C.cfg.update(saved)


def test_there_is_one_pair_per_modality():
    assert len(YMLS) == 10


@pytest.mark.parametrize("path", YMLS, ids=[os.path.basename(p) for p in YMLS])
def test_yml_parses_into_its_mode(path, restore_cfg):
    C.cfg_from_file(path)
    tag = re.search(r"cosine_(rgbd_add|rgbd_early|rgbd_cat|color|depth)", os.path.basename(path)).group(1)
    assert C.network_mode() == MODE[tag]
    assert C.cfg.TRAIN.EMBEDDING_METRIC == "cosine" and C.cfg.TRAIN.NUM_UNITS == 64
    assert tuple(C.cfg.TEST.SCALES_BASE) == (1.0,) and C.cfg.TEST.VISUALIZE is False
    assert C.cfg.TRAIN.SYN_CROP_SIZE == 224 and abs(2 * C.cfg.TRAIN.EMBEDDING_ALPHA - 0.04) < 1e-12


def test_wrapper_scripts_point_at_existing_files():
    for sh in glob.glob(os.path.join(ROOT, "experiments", "scripts", "*.sh")):
        text = open(sh).read()
        assert os.access(sh, os.X_OK), sh
        for rel in re.findall(r"python ((?:tools|ros)/\S+\.py|bench\.py)", text):
            assert os.path.exists(os.path.join(ROOT, rel)), (sh, rel)
        for rel in re.findall(r"(experiments/cfgs/[A-Za-z0-9_]+\.yml)", text):
            assert os.path.exists(os.path.join(ROOT, rel)), (sh, rel)
