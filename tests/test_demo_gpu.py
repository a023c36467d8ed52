"""BASELINE config 1 on the GPU: a real 640x480 RGB-D pair (reference data/demo, copied as input
data) through read_sample -> SEGNET -> two-stage test_sample, against the label maps the
reference's own test_sample produced with the same (calibrated synthetic) weights.

With the real network in the loop the embeddings agree to ~1e-6, not bitwise; on this frame that flips NO pixel:
measured 0 mismatching pixels for both the stage-1 and the refined map (round 2, gpurun_out/demo_parity.json), so
the bar is the north_star's: identical partitions (equal up to a permutation of the ids)."""
import json
import os

import numpy as np
import pytest
import torch

from unseenobjectclustering_amd import io as uio, networks, synth
from unseenobjectclustering_amd.fcn import test_dataset as TD
from unseenobjectclustering_amd.fcn.config import cfg

pytestmark = pytest.mark.gpu


def _agreement(a, b):
    """Fraction of pixels on which the two partitions agree after greedy id matching."""
    a = a.reshape(-1).astype(np.int64)
    b = b.reshape(-1).astype(np.int64)
    conf = np.zeros((a.max() + 1, b.max() + 1), dtype=np.int64)
    np.add.at(conf, (a, b), 1)
    return conf.max(axis=1).sum() / a.size, conf


def test_demo_frame_matches_reference(golden_dir, device):
    cfg.device = device
    g = np.load(os.path.join(golden_dir, "demo.npz"))
    d = os.path.join(golden_dir, "demo")
    cam = json.load(open(os.path.join(d, "camera_params.json")))
    sample = uio.read_sample(os.path.join(d, "000002-color.png"), os.path.join(d, "000002-depth.png"), cam)
    assert sample["image_color"].shape == (1, 3, 480, 640) and sample["depth"].shape == (1, 3, 480, 640)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    net_crop = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    feat = net(sample["image_color"].to(device), None, sample["depth"].to(device))
    emb = feat.permute(0, 2, 3, 1).reshape(-1, 64)[torch.from_numpy(g["pos"]).to(device)].cpu().numpy()
    assert np.abs(emb - g["embed"]).max() < 1e-3
    np.random.seed(3)
    out_label, refined = TD.test_sample(sample, net, net_crop)
    agree, conf = _agreement(out_label.numpy(), g["out_label"])
    assert refined is not None
    agree2, _ = _agreement(refined.numpy(), g["refined"])
    n = out_label.numel()
    rec = {"stage1_mismatched_pixels": int(round((1 - agree) * n)), "refined_mismatched_pixels": int(round((1 - agree2) * n)),
           "pixels": n}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(root, "gpurun_out", "demo_parity.json"), "w"))
    print("demo frame vs reference golden:", rec)
    assert rec["stage1_mismatched_pixels"] == 0 and rec["refined_mismatched_pixels"] == 0, rec
    assert len(np.unique(out_label.numpy())) == len(np.unique(g["out_label"]))
    assert len(np.unique(refined.numpy())) == len(np.unique(g["refined"]))


def test_device_input_prep_is_bit_identical(golden_dir, device):
    """uoc_prep_rgbd (uint8 BGR + uint16 mm depth -> network inputs on the device) against the host-side
    mirror of read_sample/compute_xyz (tools/test_images.py:96-133): same float32 ops, bit-exact."""
    d = os.path.join(golden_dir, "demo")
    cam = json.load(open(os.path.join(d, "camera_params.json")))
    want = uio.read_sample(os.path.join(d, "000002-color.png"), os.path.join(d, "000002-depth.png"), cam)
    raw = uio.read_sample_raw(os.path.join(d, "000002-color.png"), os.path.join(d, "000002-depth.png"), cam)
    got = uio.prepare_on_device(raw, device)
    assert torch.equal(got["image_color"].cpu(), want["image_color"])
    assert torch.equal(got["depth"].cpu(), want["depth"])
    # and a synthetic frame with the full uint16 range / odd size
    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    dep = rng.integers(0, 65536, size=(37, 53)).astype(np.uint16)
    cam2 = dict(fx=500.5, fy=499.25, x_offset=26.1, y_offset=18.7)
    want = uio.make_sample(im, dep, cam2)
    got = uio.prepare_on_device(uio.make_sample_raw(im, dep, cam2), device)
    assert torch.equal(got["image_color"].cpu(), want["image_color"])
    assert torch.equal(got["depth"].cpu(), want["depth"])


def test_raw_sample_through_test_sample(golden_dir, device):
    cfg.device = device
    g = np.load(os.path.join(golden_dir, "demo.npz"))
    d = os.path.join(golden_dir, "demo")
    cam = json.load(open(os.path.join(d, "camera_params.json")))
    raw = uio.read_sample_raw(os.path.join(d, "000002-color.png"), os.path.join(d, "000002-depth.png"), cam)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    np.random.seed(3)
    out_label, refined = TD.test_sample(raw, net, net)
    agree, _ = _agreement(out_label.numpy(), g["out_label"])
    assert agree == 1.0, agree
