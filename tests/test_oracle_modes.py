"""Pins the oracle's other input modalities (cfg.INPUT 'COLOR' / 'DEPTH', RGBD 'early' fusion;
SURVEY.md §8 f-3) and its depth-less glue to golden vectors captured from the reference
(tests/golden/modes.npz, made by tests/golden/make_golden.py modes).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import backbone_oracle as BO
from oracle import glue_oracle as G
from oracle import mean_shift_oracle as MS
from tests.golden.cases import (GLUE_CASES, MODES, MODE_BACKBONE_CASES, MODE_GLUE_CASES, MODE_E2E_CASES, RNG_SEED,
                                WIDE_MEANSHIFT_CASES, KAPPA, EPSILON, glue_inputs, crop_cluster_labels,
                                e2e_stub_features)
from unseenobjectclustering_amd import synth
from unseenobjectclustering_amd.fcn import config as C


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "modes.npz"))


@pytest.mark.parametrize("mode", list(MODES))
def test_backbone_oracle_modes_match_reference(golden, mode):
    c = MODE_BACKBONE_CASES["tiny_64x64"]
    branches = ("fcn", "fcn_depth") if mode == "RGBD_CAT" else ("fcn",)
    sd = {k: torch.from_numpy(np.asarray(v))
          for k, v in synth.synthetic_state_dict(c["wseed"], branches=branches, in_channels=MODES[mode]["in_channels"]).items()}
    fr = synth.rgbd_frame(c["frames"][0], c["H"], c["W"], 4)
    out = BO.segnet_forward(sd, torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"]), mode)
    assert out.shape[1] == (128 if mode == "RGBD_CAT" else 64)
    flat = out.permute(0, 2, 3, 1).reshape(1, -1, out.shape[1]).numpy()[:, golden[f"{mode}/tiny_64x64/pos"]]
    assert np.abs(flat - golden[f"{mode}/tiny_64x64/embed"]).max() < 1e-6


@pytest.mark.parametrize("name", MODE_GLUE_CASES)
def test_glue_oracle_without_depth_matches_reference(golden, name):
    c = GLUE_CASES[name]
    img, lab, depth, gt = glue_inputs(c)
    rgb_c, mask_c, rois, depth_c = G.crop_rois(img, lab.clone(), None)
    assert depth_c is None
    key = "COLOR/glue_" + name
    assert np.array_equal(rois.numpy().astype(np.int32), golden[key + "/rois"])
    assert np.array_equal(np.packbits(mask_c.numpy().astype(np.uint8), axis=None), golden[key + "/mask_crops"])
    labels_c = crop_cluster_labels(c, gt, rois)
    refined, labels_c2 = G.match_label_crop(lab, labels_c, mask_c, rois, None)
    assert np.array_equal(refined.numpy().astype(np.uint8), golden[key + "/refined"])
    assert np.array_equal(labels_c2.numpy().astype(np.int8), golden[key + "/labels_crop_out"])


@pytest.mark.parametrize("name", list(MODE_E2E_CASES))
def test_test_sample_oracle_color_matches_reference(golden, name):
    c = MODE_E2E_CASES[name]
    fr = synth.rgbd_frame(c["seed"], 480, 640, c["objects"])
    net = lambda img, label, depth: e2e_stub_features(c["seed"], 480, 640, c["objects"] + 2)
    net_crop = lambda rgb, label, depth: torch.cat(
        [e2e_stub_features(1000 + 10 * c["seed"] + k, 224, 224, 2 + k % 3) for k in range(rgb.shape[0])])
    out_label, refined = G.test_sample(torch.from_numpy(fr["image_color"]), None, net, net_crop,
                                       np.random.RandomState(RNG_SEED))
    assert np.array_equal(out_label.numpy().astype(np.uint8), golden[f"COLOR/{name}/out_label"])
    assert np.array_equal(refined.numpy().astype(np.uint8), golden[f"COLOR/{name}/refined"])


@pytest.mark.parametrize("name", ["wide_60x80", "wide_224"])
def test_mean_shift_oracle_128d_matches_reference(golden, name):
    """128-d fields ('cat' fusion): the oracle's clustering against the reference's own mean_shift_smart_init."""
    c = WIDE_MEANSHIFT_CASES[name]
    X, _ = synth.embedding_field(c["seed"], c["H"], c["W"], 128, c["num_objects"], c["noise"])
    first = int(golden[f"WIDE/{name}/indices"][0])
    labels, idx = MS.mean_shift_smart_init(torch.from_numpy(X), KAPPA, c["m"], c["iters"], first_index=first, epsilon=EPSILON)
    assert np.array_equal(idx.numpy().astype(np.int32), golden[f"WIDE/{name}/indices"])
    assert np.array_equal(labels.numpy().astype(np.uint8), golden[f"WIDE/{name}/labels"])


def test_config_modes():
    """cfg.INPUT / FUSION_TYPE -> network mode; unsupported combinations raise (no silent fallback)."""
    saved = (C.cfg.INPUT, C.cfg.TRAIN.FUSION_TYPE)
    try:
        for mode, m in MODES.items():
            C.cfg.INPUT, C.cfg.TRAIN.FUSION_TYPE = m["INPUT"], m["FUSION"]
            assert C.network_mode() == mode
            assert C.uses_depth() == (mode != "COLOR")
        C.cfg.INPUT = "XYZ"
        with pytest.raises(NotImplementedError):
            C.require_supported()
    finally:
        C.cfg.INPUT, C.cfg.TRAIN.FUSION_TYPE = saved
    assert C.network_mode() == "RGBD_ADD"


def test_cfg_from_file(tmp_path):
    """Experiment ymls of the reference's experiments/cfgs/ layout (incl. !!python/tuple tags and many keys
    this path never reads) merge into cfg; type mismatches and unsupported settings raise."""
    saved = (C.cfg.INPUT, C.cfg.TRAIN.FUSION_TYPE, C.cfg.TRAIN.EMBEDDING_ALPHA, C.cfg.TEST.VISUALIZE)
    y = tmp_path / "exp.yml"
    y.write_text("EXP_DIR: tabletop_object\nINPUT: RGBD\nTRAIN:\n  MILESTONES: !!python/tuple [3]\n  FUSION_TYPE: early\n"
                 "  SYN_CROP_SIZE: 224\n  EMBEDDING_METRIC: cosine\n  EMBEDDING_ALPHA: 0.03\nTEST:\n  VISUALIZE: True\n")
    try:
        C.cfg_from_file(str(y))
        assert C.network_mode() == "RGBD_EARLY" and C.cfg.TRAIN.EMBEDDING_ALPHA == 0.03 and C.cfg.TEST.VISUALIZE is True
        y.write_text("TRAIN:\n  SYN_CROP_SIZE: big\n")
        with pytest.raises(ValueError):
            C.cfg_from_file(str(y))
        y.write_text("TRAIN:\n  EMBEDDING_METRIC: euclidean\n")
        with pytest.raises(NotImplementedError):
            C.cfg_from_file(str(y))
    finally:
        C.cfg.TRAIN.EMBEDDING_METRIC = "cosine"
        C.cfg.INPUT, C.cfg.TRAIN.FUSION_TYPE, C.cfg.TRAIN.EMBEDDING_ALPHA, C.cfg.TEST.VISUALIZE = saved
