"""CPU: (1) the host-side input preparation (unseenobjectclustering_amd/io.py) against the REFERENCE's own read_sample
(tools/test_images.py:96-135) captured in tests/golden/prep.npz — bit-exact; (2) the oracle's restatement of the
reference's test_segnet loop (lib/fcn/test_dataset.py:271-381) against tests/golden/segnet.npz."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import glue_oracle as G
from tests.golden.cases import (PREP_SYNTH, RNG_SEED, SEGNET_RUNS, SegnetLoader, prep_synthetic_arrays, segnet_samples,
                                segnet_stub_networks)
from unseenobjectclustering_amd import io as uio
from unseenobjectclustering_amd.fcn.config import cfg


@pytest.fixture(scope="module")
def prep(golden_dir):
    return np.load(os.path.join(golden_dir, "prep.npz"))


def test_read_sample_demo_pair_is_bit_identical_to_the_reference(prep, golden_dir):
    d = os.path.join(golden_dir, "demo")
    cam = json.load(open(os.path.join(d, "camera_params.json")))
    s = uio.read_sample(os.path.join(d, "000002-color.png"), os.path.join(d, "000002-depth.png"), cam)
    for key in ("image_color", "depth"):
        a = np.ascontiguousarray(s[key].numpy())
        assert a.dtype == np.float32 and a.shape == (1, 3, 480, 640)
        assert np.array_equal(a.reshape(-1)[prep["demo/pos"]], prep[f"demo/{key}/samples"])
        assert hashlib.sha256(a.tobytes()).digest() == prep[f"demo/{key}/sha256"].tobytes(), key


def test_make_sample_full_range_pair_is_bit_identical_to_the_reference(prep):
    im, dep = prep_synthetic_arrays()
    s = uio.make_sample(im, dep, PREP_SYNTH["camera"])
    assert np.array_equal(s["image_color"].numpy(), prep["synth/image_color"])
    assert np.array_equal(s["depth"].numpy(), prep["synth/depth"])
    # the raw sample the device-side preparation consumes carries the same bits
    raw = uio.make_sample_raw(im, dep, PREP_SYNTH["camera"])
    assert np.array_equal(raw["image_u8"].numpy(), im)
    assert np.array_equal(raw["depth_u16"].numpy().view(np.uint16), dep)


@pytest.mark.parametrize("tag", list(SEGNET_RUNS))
def test_oracle_test_segnet_matches_reference_golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "segnet.npz"))
    run = SEGNET_RUNS[tag]
    net, net_crop = segnet_stub_networks(run)
    got = G.test_segnet(SegnetLoader(run["name"], segnet_samples(run)), net, net_crop, np.random.RandomState(RNG_SEED))
    assert len(got) == len(run["frames"])
    for i, (pred, refined) in enumerate(got):
        assert np.array_equal(pred.astype(np.uint8), g[f"{tag}/{i}/labels"])
        assert np.array_equal(refined.astype(np.uint8), g[f"{tag}/{i}/labels_refined"])


def test_npy_driver_read_sample_matches_the_reference(golden_dir, tmp_path):
    """tools/test_npy.py read_sample (both .npy layouts) against the reference's own tools/test_npy.py:105-144
    (tests/golden/npy.npz): bit-identical float32 tensors."""
    import importlib.util
    from tests.golden.cases import npy_frames
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("uoc_tools_test_npy", os.path.join(root, "tools", "test_npy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.load(os.path.join(golden_dir, "npy.npz"))
    saved = cfg.INPUT
    cfg.INPUT = "RGBD"
    try:
        for name, d in npy_frames().items():
            f = str(tmp_path / (name + ".npy"))
            np.save(f, d, allow_pickle=True)
            s = mod.read_sample(f, PREP_SYNTH["camera"])
            for key in ("image_color", "depth"):
                assert s[key].dtype == torch.float32
                assert np.array_equal(s[key].numpy(), g[f"{name}/{key}"]), (name, key)
        # float64 intrinsics (what np.array(msg.K) or a json round trip gives) must not promote the result
        d = npy_frames()["debug"]
        d["debug_info"]["intrinsics"] = d["debug_info"]["intrinsics"].astype(np.float64)
        f = str(tmp_path / "debug64.npy")
        np.save(f, d, allow_pickle=True)
        s = mod.read_sample(f, None)
        assert s["depth"].dtype == torch.float32 and np.array_equal(s["depth"].numpy(), g["debug/depth"])
    finally:
        cfg.INPUT = saved
