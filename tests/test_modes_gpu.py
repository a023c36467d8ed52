"""GPU parity of the other input modalities (cfg.INPUT 'COLOR' / 'DEPTH', RGBD 'early' fusion) and
of the depth-less two-stage glue against golden vectors captured from the reference
(tests/golden/modes.npz).  Embeddings within 1e-3 (north-star tolerance), integer outputs bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import backbone_oracle as BO
from oracle import mean_shift_oracle as MS
from tests.golden.cases import (GLUE_CASES, MODES, MODE_BACKBONE_CASES, MODE_GLUE_CASES, MODE_E2E_CASES, RNG_SEED,
                                WIDE_MEANSHIFT_CASES, KAPPA, EPSILON, glue_inputs, crop_cluster_labels,
                                e2e_stub_features)
from unseenobjectclustering_amd import networks, synth
from unseenobjectclustering_amd.fcn import test_dataset as TD
from unseenobjectclustering_amd.fcn.config import cfg
from unseenobjectclustering_amd.utils import mean_shift as UMS

pytestmark = pytest.mark.gpu
EMBED_TOL = 1e-3


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "modes.npz"))


@pytest.fixture()
def mode_cfg():
    saved = (cfg.INPUT, cfg.TRAIN.FUSION_TYPE)

    def use(mode):
        cfg.INPUT, cfg.TRAIN.FUSION_TYPE = MODES[mode]["INPUT"], MODES[mode]["FUSION"]
    yield use
    cfg.INPUT, cfg.TRAIN.FUSION_TYPE = saved


def _net(mode, wseed):
    m = MODES[mode]
    branches = ("fcn", "fcn_depth") if mode == "RGBD_CAT" else ("fcn",)      # SEG.py:69-71
    sd = {k: torch.from_numpy(np.asarray(v))
          for k, v in synth.synthetic_state_dict(wseed, branches=branches, in_channels=m["in_channels"]).items()}
    net = networks.__dict__[m["factory"]](2, 64, sd).eval()
    assert set(net.state_dict().keys()) == set(sd.keys())
    return net, sd


@pytest.mark.parametrize("name", list(MODE_BACKBONE_CASES))
@pytest.mark.parametrize("mode", list(MODES))
def test_network_modes_match_reference_golden(golden, device, mode_cfg, mode, name):
    mode_cfg(mode)
    c = MODE_BACKBONE_CASES[name]
    net, _ = _net(mode, c["wseed"])
    frames = [synth.rgbd_frame(s, c["H"], c["W"], 4) for s in c["frames"]]
    img = torch.from_numpy(np.concatenate([f["image_color"] for f in frames])).to(device)
    dep = torch.from_numpy(np.concatenate([f["depth"] for f in frames])).to(device)
    feat = net(img, None, None if mode == "COLOR" else dep)
    dim = 128 if mode == "RGBD_CAT" else 64
    assert feat.shape == (len(frames), dim, c["H"], c["W"])
    flat = feat.permute(0, 2, 3, 1).reshape(len(frames), -1, dim).cpu().numpy()
    key = f"{mode}/{name}"
    if key + "/pos" in golden:
        flat = flat[:, golden[key + "/pos"]]
    err = np.abs(flat - golden[key + "/embed"]).max()
    assert err < EMBED_TOL, err
    assert np.abs(np.linalg.norm(flat, axis=2) - 1).max() < 1e-5


@pytest.mark.parametrize("mode", list(MODES))
def test_network_modes_vs_oracle_crop_size(device, mode_cfg, mode):
    """224x224 batch of 2 (the stage-2 shape) against the CPU oracle."""
    mode_cfg(mode)
    net, sd = _net(mode, 9)
    frames = [synth.rgbd_frame(s, 224, 224, 3) for s in (21, 22)]
    img = torch.from_numpy(np.concatenate([f["image_color"] for f in frames]))
    dep = torch.from_numpy(np.concatenate([f["depth"] for f in frames]))
    want = BO.segnet_forward(sd, img, dep, mode)
    got = net(img.to(device), None, None if mode == "COLOR" else dep.to(device)).cpu()
    assert (got - want).abs().max().item() < EMBED_TOL


def test_mode_mismatch_raises(device, mode_cfg):
    mode_cfg("RGBD_EARLY")
    with pytest.raises(ValueError):
        networks.seg_resnet34_8s_embedding(2, 64, None)            # 3-channel factory under early fusion (SEG.py:103-105)
    mode_cfg("COLOR")
    with pytest.raises(ValueError):
        networks.seg_resnet34_8s_embedding_early(2, 64, None)
    mode_cfg("DEPTH")
    net, _ = _net("DEPTH", 1)
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 64, 64, device=device), None, None)  # DEPTH network without the XYZ tensor


@pytest.mark.parametrize("name", MODE_GLUE_CASES)
def test_glue_without_depth_matches_reference_golden(golden, device, mode_cfg, name):
    mode_cfg("COLOR")
    cfg.device = device
    c = GLUE_CASES[name]
    img, lab, depth, gt = glue_inputs(c)
    rgb_c, mask_c, rois, depth_c = TD.crop_rois(img.to(device), lab.clone(), None)
    assert depth_c is None
    K = rgb_c.shape[0]
    key = "COLOR/glue_" + name
    assert np.array_equal(rois.cpu().numpy().astype(np.int32), golden[key + "/rois"])
    assert np.array_equal(np.packbits(mask_c.cpu().numpy().astype(np.uint8), axis=None), golden[key + "/mask_crops"])
    assert np.abs(rgb_c.double().sum(dim=(1, 2, 3)).cpu().numpy() - golden[key + "/rgb_crops_sum"]).max() < 1e-2
    labels_c = crop_cluster_labels(c, gt, rois.cpu()).to(device)
    refined, labels_c2 = TD.match_label_crop(lab, labels_c, mask_c, rois, None)
    assert np.array_equal(refined.cpu().numpy().astype(np.uint8), golden[key + "/refined"])
    assert np.array_equal(labels_c2.cpu().numpy().astype(np.int8), golden[key + "/labels_crop_out"])
    assert K == golden[key + "/rois"].shape[0]


@pytest.mark.parametrize("name", list(MODE_E2E_CASES))
def test_test_sample_color_matches_reference_golden(golden, device, mode_cfg, name):
    """COLOR input: the sample has no depth, no depth-coverage filter runs (test_dataset.py:250), ROIs are
    painted in order of box area (:138-146)."""
    mode_cfg("COLOR")
    cfg.device = device
    c = MODE_E2E_CASES[name]
    fr = synth.rgbd_frame(c["seed"], 480, 640, c["objects"])
    sample = dict(image_color=torch.from_numpy(fr["image_color"]))
    seen = {}

    def net(img, label, depth):
        seen["depth"] = depth
        return e2e_stub_features(c["seed"], 480, 640, c["objects"] + 2).to(device)

    def net_crop(rgb, label, depth):
        seen["depth_crop"] = depth
        return torch.cat([e2e_stub_features(1000 + 10 * c["seed"] + k, 224, 224, 2 + k % 3)
                          for k in range(rgb.shape[0])]).to(device)
    np.random.seed(RNG_SEED)
    out_label, refined = TD.test_sample(sample, net, net_crop)
    assert seen["depth"] is None and seen["depth_crop"] is None
    assert np.array_equal(out_label.numpy().astype(np.uint8), golden[f"COLOR/{name}/out_label"])
    assert np.array_equal(refined.numpy().astype(np.uint8), golden[f"COLOR/{name}/refined"])


@pytest.mark.parametrize("name", list(WIDE_MEANSHIFT_CASES))
def test_cluster_128d_matches_reference_golden(golden, device, name):
    """128-d fields (the 'cat' embeddings) through uoc_ms_cluster_wide: seed indices and label maps bit-exact
    against the reference's mean_shift_smart_init, converged seeds within 1e-4."""
    c = WIDE_MEANSHIFT_CASES[name]
    X, _ = synth.embedding_field(c["seed"], c["H"], c["W"], 128, c["num_objects"], c["noise"])
    Xd = torch.from_numpy(X).to(device)
    first = int(golden[f"WIDE/{name}/indices"][0])
    labels, idx, Z, sl = UMS.cluster_batch(UMS.to_planes(Xd[None]), [first], KAPPA, c["m"], c["iters"], EPSILON,
                                           return_parts=True)
    assert np.array_equal(idx[0].cpu().numpy(), golden[f"WIDE/{name}/indices"])
    Zr = Z[0].permute(1, 0, 2).reshape(c["m"], 128).cpu().numpy()
    assert np.abs(Zr - golden[f"WIDE/{name}/Z"]).max() < 1e-4
    assert np.array_equal(sl[0].cpu().numpy(), golden[f"WIDE/{name}/seed_labels"])
    assert MS.labels_equal_up_to_permutation(labels[0].cpu().numpy(), golden[f"WIDE/{name}/labels"])
    assert np.array_equal(labels[0].cpu().numpy().astype(np.uint8), golden[f"WIDE/{name}/labels"])
    # the reference-named entry point on the row-major [n,128] tensor, global RNG draw included
    np.random.seed(RNG_SEED)
    lab2, idx2 = UMS.mean_shift_smart_init(Xd, KAPPA, c["m"], c["iters"])
    assert lab2.dtype == torch.int64 and np.array_equal(idx2.numpy().astype(np.int32), golden[f"WIDE/{name}/indices"])
    assert np.array_equal(lab2.cpu().numpy().astype(np.uint8), golden[f"WIDE/{name}/labels"])


def test_cat_two_stage_vs_oracle(device, mode_cfg):
    """'cat' fusion end to end against the CPU oracle's test_sample: (1) the real 128-d network on a small frame,
    (2) stub networks returning structured [B,128,h,w] fields so that stage 2 (crops, 128-d crop clustering,
    matching, paste) runs with several objects."""
    from oracle import glue_oracle as G
    mode_cfg("RGBD_CAT")
    cfg.device = device
    net, sd = _net("RGBD_CAT", 11)
    fr = synth.rgbd_frame(77, 128, 160, 3)
    img, dep = torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"])
    onet = lambda i, l, d: BO.segnet_forward(sd, i, d, "RGBD_CAT")
    want_label, want_refined = G.test_sample(img, dep, onet, onet, np.random.RandomState(RNG_SEED))
    np.random.seed(RNG_SEED)
    out_label, refined = TD.test_sample(dict(image_color=img, depth=dep), net, net)
    assert MS.labels_equal_up_to_permutation(out_label.numpy(), want_label.numpy())
    assert (refined is None) == (want_refined is None)
    if refined is not None:
        assert MS.labels_equal_up_to_permutation(refined.numpy(), want_refined.numpy())

    def field(seed, H, W, k):
        X, _ = synth.embedding_field(seed, H, W, 128, k, 0.05)
        return torch.from_numpy(X).view(H, W, 128).permute(2, 0, 1)[None].contiguous()
    fr = synth.rgbd_frame(78, 240, 320, 3)
    img, dep = torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"])
    s1 = lambda i, l, d: field(78, 240, 320, 5)
    s2 = lambda i, l, d: torch.cat([field(900 + k, 224, 224, 2 + k % 3) for k in range(i.shape[0])])
    want_label, want_refined = G.test_sample(img, dep, s1, s2, np.random.RandomState(RNG_SEED))
    np.random.seed(RNG_SEED)
    out_label, refined = TD.test_sample(dict(image_color=img, depth=dep), lambda i, l, d: s1(i, l, d).to(device),
                                        lambda i, l, d: s2(i, l, d).to(device))
    assert want_refined is not None and int(want_refined.max()) >= 2
    assert np.array_equal(out_label.numpy(), want_label.numpy())
    assert np.array_equal(refined.numpy(), want_refined.numpy())
