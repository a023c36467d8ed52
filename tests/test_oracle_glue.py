"""Pins oracle/glue_oracle.py (two-stage glue + test_sample orchestration) to golden vectors
captured from the reference's own functions.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import glue_oracle as G
from tests.golden.cases import (GLUE_CASES, E2E_CASES, RNG_SEED, glue_inputs, crop_cluster_labels,
                                e2e_stub_features)
from unseenobjectclustering_amd import synth


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "glue.npz"))


@pytest.mark.parametrize("name", list(GLUE_CASES))
def test_glue_oracle_matches_reference(golden, name):
    c = GLUE_CASES[name]
    img, lab, depth, gt = glue_inputs(c)
    filt = G.filter_labels_depth(lab, depth, 0.8)
    assert np.array_equal(filt.numpy().astype(np.uint8), golden[name + "/filtered"])
    rgb_c, mask_c, rois, depth_c = G.crop_rois(img, filt.clone(), depth)
    K = rgb_c.shape[0]
    assert np.array_equal(rois.numpy().astype(np.int32).reshape(-1, 4), golden[name + "/rois"].reshape(-1, 4))
    assert np.array_equal(np.packbits(mask_c.numpy().astype(np.uint8), axis=None), golden[name + "/mask_crops"])
    pos = golden[name + "/crop_pos"]
    if K:
        assert np.abs(rgb_c.reshape(K, -1)[:, pos].numpy() - golden[name + "/rgb_crops_s"]).max() < 1e-6
        assert np.abs(depth_c.reshape(K, -1)[:, pos].numpy() - golden[name + "/depth_crops_s"]).max() < 1e-6
        labels_c = crop_cluster_labels(c, gt, rois)
        refined, labels_c2 = G.match_label_crop(filt, labels_c, mask_c, rois, depth_c)
        assert np.array_equal(refined.numpy().astype(np.uint8), golden[name + "/refined"])
        assert np.array_equal(labels_c2.numpy().astype(np.int8), golden[name + "/labels_crop_out"])


@pytest.mark.parametrize("name", ["e2e_b"])
def test_test_sample_oracle_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "e2e.npz"))
    c = E2E_CASES[name]
    fr = synth.rgbd_frame(c["seed"], 480, 640, c["objects"])
    net = lambda img, label, depth: e2e_stub_features(c["seed"], 480, 640, c["objects"] + 2)
    net_crop = lambda rgb, label, depth: torch.cat(
        [e2e_stub_features(1000 + 10 * c["seed"] + k, 224, 224, 2 + k % 3) for k in range(rgb.shape[0])])
    rng = np.random.RandomState(RNG_SEED)
    out_label, refined = G.test_sample(torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"]), net, net_crop, rng)
    assert np.array_equal(out_label.numpy().astype(np.uint8), g[name + "/out_label"])
    assert np.array_equal(refined.numpy().astype(np.uint8), g[name + "/refined"])


def test_cpython_short_list_sort_restatement_equals_sorted():
    """The restatement of CPython's list.sort for n < 64 (oracle/glue_oracle.py, mirrored by csrc/roi.hip for NaN sort keys) gives
    exactly sorted(..., reverse=True) — with ties and with NaN keys, where every comparison is false and only the algorithm itself
    defines the result (lib/fcn/test_dataset.py:135,148)."""
    import random
    import torch
    from oracle import glue_oracle as G
    rng = random.Random(5)
    for trial in range(3000):
        n = rng.randint(0, 63)
        pool = [round(rng.random(), 1) for _ in range(max(1, n // 3))]
        keys = [float("nan") if rng.random() < rng.choice([0.0, 0.2, 0.8]) else rng.choice(pool) for _ in range(n)]
        want = [i for i, _ in sorted([(i, k) for i, k in enumerate(keys)], key=lambda x: x[1], reverse=True)]
        assert G.cpython_sort_small(keys) == want, (keys, want)
    # 0-dim float32 tensors as keys, like the reference's avg_depth values
    keys = [torch.tensor(v, dtype=torch.float32) for v in (0.5, float("nan"), 0.7, 0.5, float("nan"), 0.1)]
    want = [i for i, _ in sorted([(i, k) for i, k in enumerate(keys)], key=lambda x: x[1], reverse=True)]
    assert G.cpython_sort_small(keys) == want
