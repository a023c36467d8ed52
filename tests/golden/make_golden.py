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


def main():
    assert ref_harness.available(), "reference tree not present: golden vectors can only be made in the build container"
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ref = ref_harness.load_reference()
    torch.manual_seed(0)
    if what in ("meanshift", "all"):
        make_meanshift(ref)


if __name__ == "__main__":
    main()
