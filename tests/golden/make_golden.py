"""Golden-vector generator — runs ONLY in the build container, where /root/reference exists.

Imports the reference's own Python (through tests/golden/ref_harness.py) and records its
outputs on repo-owned synthetic inputs.  The .npz files written next to this script are the
fixtures tests/ compares the oracle (CPU) and the HIP path (GPU) against.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [meanshift|glue|backbone|e2e|all]
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
from cases import MEANSHIFT_CASES, KAPPA, EPSILON, RNG_SEED  # noqa: E402
from unseenobjectclustering_amd import synth  # noqa: E402


def make_meanshift(ref):
    ms = ref.mean_shift
    out = {}
    for name, c in MEANSHIFT_CASES.items():
        X, _ = synth.embedding_field(c["seed"], c["H"], c["W"], 64, c["num_objects"], c["noise"])
        Xt = torch.from_numpy(X)
        # full pipeline exactly as the reference runs it (global numpy RNG seeded like tools/test_images.py:154)
        np.random.seed(RNG_SEED)
        labels, idx = ms.mean_shift_smart_init(Xt, KAPPA, num_seeds=c["m"], max_iters=c["iters"], metric="cosine")
        # the same, stage by stage, to record intermediates
        np.random.seed(RNG_SEED)
        seeds, idx2 = ms.select_smart_seeds(Xt, c["m"], return_selected_indices=True, metric="cosine")
        assert torch.equal(idx, idx2)
        Z = ms.seed_hill_climbing_ball(Xt, seeds, KAPPA, max_iters=c["iters"], metric="cosine")
        seed_labels = ms.connected_components(Z, 2 * ref.cfg.TRAIN.EMBEDDING_ALPHA, metric="cosine")
        assert abs(2 * ref.cfg.TRAIN.EMBEDDING_ALPHA - EPSILON) < 1e-12
        assert int(labels.max()) < 255
        out[name + "/labels"] = labels.numpy().astype(np.uint8)
        out[name + "/indices"] = idx.numpy().astype(np.int32)
        out[name + "/Z"] = Z.numpy().astype(np.float32)
        out[name + "/seed_labels"] = seed_labels.numpy().astype(np.int32)
        print(name, "clusters:", np.unique(out[name + "/labels"]).tolist(), "first idx", int(idx[0]), flush=True)
    np.savez_compressed(os.path.join(HERE, "meanshift.npz"), **out)


def sample_positions(seed, n, count):
    return np.sort(np.random.default_rng(seed).choice(n, size=min(count, n), replace=False)).astype(np.int64)


BACKBONE_CASES = {
    # name -> (weight seed, frame seeds, H, W, num sampled pixels (0 = keep everything))
    "tiny_64x64":   dict(wseed=1, frames=[7], H=64, W=64, samples=0),
    "odd_72x104":   dict(wseed=2, frames=[8], H=72, W=104, samples=0),
    "crops_224":    dict(wseed=2, frames=[4, 5], H=224, W=224, samples=1024),
    "full_480x640": dict(wseed=1, frames=[1], H=480, W=640, samples=2048),
}


def make_backbone(ref):
    import contextlib
    import io
    out = {}
    for name, c in BACKBONE_CASES.items():
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synthetic_state_dict(c["wseed"]).items()}
        with contextlib.redirect_stdout(io.StringIO()):
            net = ref.networks.__dict__["seg_resnet34_8s_embedding"](2, 64, sd).eval()
        frames = [synth.rgbd_frame(s, c["H"], c["W"], 4) for s in c["frames"]]
        img = torch.from_numpy(np.concatenate([f["image_color"] for f in frames]))
        dep = torch.from_numpy(np.concatenate([f["depth"] for f in frames]))
        with torch.no_grad():
            feat = net(img, None, dep)                       # [B,64,H,W]
        B = feat.shape[0]
        flat = feat.permute(0, 2, 3, 1).reshape(B, -1, 64).numpy()
        if c["samples"]:
            pos = sample_positions(99, flat.shape[1], c["samples"])
            out[name + "/pos"] = pos
            out[name + "/embed"] = flat[:, pos].astype(np.float32)
        else:
            out[name + "/embed"] = flat.astype(np.float32)
        print(name, feat.shape, "norm check", float(feat.norm(dim=1).mean()), flush=True)
    np.savez_compressed(os.path.join(HERE, "backbone.npz"), **out)


def main():
    assert ref_harness.available(), "reference tree not present: golden vectors can only be made in the build container"
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ref = ref_harness.load_reference()
    torch.manual_seed(0)
    if what in ("meanshift", "all"):
        make_meanshift(ref)
    if what in ("backbone", "all"):
        make_backbone(ref)


if __name__ == "__main__":
    main()
