"""CPU: the OCID / OSD loaders (unseenobjectclustering_amd/datasets, mirrors of /root/reference/lib/datasets/
ocid_object.py, osd_object.py, factory.py) on synthetic directory trees, the pcl-free PCD reader in all three storage
modes, and the native LZF decoder."""
import ctypes
import os

import numpy as np
import pytest
import torch

from tests import dataset_tree as DT
from unseenobjectclustering_amd import _native, datasets, synth
from unseenobjectclustering_amd.datasets import pcd
from unseenobjectclustering_amd.fcn.config import cfg, get_output_dir


def test_lzf_round_trip_with_back_references():
    rng = np.random.default_rng(3)
    for raw in (b"", b"a", b"abcabcabcabcabcabcabcabcabc" * 40, bytes(1000),
                rng.integers(0, 4, size=5000, dtype=np.uint8).tobytes(), rng.integers(0, 256, size=3000, dtype=np.uint8).tobytes()):
        comp = DT.lzf_compress(raw)
        dst = np.empty(max(len(raw), 1), dtype=np.uint8)
        n = _native.lib().uoc_lzf_decompress(ctypes.c_char_p(comp), len(comp), ctypes.c_void_p(dst.ctypes.data), len(raw))
        assert n == len(raw) and dst[:n].tobytes() == raw
    assert len(DT.lzf_compress(bytes(1000))) < 40, "the test compressor must produce back references"
    # malformed streams are refused, not read out of bounds
    bad = bytes([0xE0, 0x10, 0x05])          # back reference before any output
    dst = np.empty(64, dtype=np.uint8)
    assert _native.lib().uoc_lzf_decompress(ctypes.c_char_p(bad), 3, ctypes.c_void_p(dst.ctypes.data), 64) == -22
    assert _native.lib().uoc_lzf_decompress(ctypes.c_char_p(b"\x05ab"), 3, ctypes.c_void_p(dst.ctypes.data), 64) == -22


@pytest.mark.parametrize("mode", ["ascii", "binary", "binary_compressed"])
def test_pcd_reader_modes(tmp_path, mode):
    rng = np.random.default_rng(11)
    xyz = rng.standard_normal((257, 3)).astype(np.float32)
    xyz[::9] = np.nan
    path = os.path.join(str(tmp_path), "c.pcd")
    DT.write_pcd(path, xyz, mode, with_rgb=(mode != "ascii"))
    got = pcd.load_xyz(path)
    assert got.dtype == np.float32 and got.shape == (257, 3)
    assert np.array_equal(np.isnan(got), np.isnan(xyz))
    assert np.array_equal(got[~np.isnan(xyz)], xyz[~np.isnan(xyz)])


def _expected_label(lab, drop_table, drop_second):
    lab = lab.copy()
    if drop_table:
        lab[lab == 1] = 0
    if drop_second:
        lab[lab == 2] = 0
    out = lab.copy()
    for k, v in enumerate(np.unique(lab)):
        out[lab == v] = k
    return out


def test_ocid_loader(tmp_path):
    root = str(tmp_path)
    written = DT.make_ocid(root)
    cfg.INPUT, cfg.MODE = "RGBD", "TEST"
    ds = datasets.OCIDObject("test", os.path.join(root, "OCID"))
    assert ds.name == "ocid_object_test" and ds.num_classes == 2 and len(ds) == 3
    rels = sorted(w[0] for w in written)
    assert sorted(s["filename"] for s in ds) == rels                      # path after '.../OCID/'
    for s in ds:
        rel, seed, objects, mode = next(w for w in written if w[0] == s["filename"])
        bgr, lab, xyz = DT.frame_files(seed, objects)
        want_img = torch.from_numpy(bgr) / 255.0
        assert torch.equal(s["image_color_bgr"], want_img.permute(2, 0, 1))
        assert torch.equal(s["image_color"], (want_img - torch.tensor(cfg.PIXEL_MEANS / 255.0).float()).permute(2, 0, 1))
        assert s["image_color"].dtype == torch.float32 and s["image_color"].shape == (3, 480, 640)
        # table (id 1) is background everywhere; under a 'table' directory id 2 as well (ocid_object.py:91-93)
        want_lab = _expected_label(lab, True, "table" in rel)
        assert s["label"].shape == (1, 480, 640) and np.array_equal(s["label"][0].numpy(), want_lab)
        want_xyz = np.nan_to_num(xyz, nan=0.0).reshape(480, 640, 3).transpose(2, 0, 1)
        assert s["depth"].shape == (3, 480, 640) and np.array_equal(s["depth"].numpy(), want_xyz), mode
    cfg.INPUT = "COLOR"
    try:
        assert "depth" not in ds[0]
    finally:
        cfg.INPUT = "RGBD"


def test_osd_loader_and_dataloader_batches(tmp_path):
    root = str(tmp_path)
    written = DT.make_osd(root, "binary_compressed")
    ds = datasets.OSDObject("test", os.path.join(root, "OSD"))
    assert ds.name == "osd_object_test" and len(ds) == 2
    loader = torch.utils.data.DataLoader(ds, batch_size=cfg.TEST.IMS_PER_BATCH, shuffle=False, num_workers=0)
    batches = list(loader)
    assert [b["filename"] for b in batches] == [["image_color/learn0.png"], ["image_color/learn1.png"]]
    for b, (name, seed, objects, mode) in zip(batches, written):
        assert b["image_color"].shape == (1, 3, 480, 640) and b["depth"].shape == (1, 3, 480, 640)
        assert b["label"].shape == (1, 1, 480, 640)
        _, lab, _ = DT.frame_files(seed, objects)
        assert np.array_equal(b["label"][0, 0].numpy(), lab)           # ids 0, 11, 12, ... compacted back to 0, 1, 2, ...
    assert get_output_dir(ds, None).endswith(os.path.join("output", cfg.EXP_DIR, "osd_object_test"))


def test_factory():
    assert set(datasets.list_datasets()) == {"osd_object_test", "ocid_object_test"}
    with pytest.raises(KeyError):
        datasets.get_dataset("nope")
    with pytest.raises(NotImplementedError):
        datasets.get_dataset("tabletop_object_train")
