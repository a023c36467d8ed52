"""Pins the oracle (oracle/mean_shift_oracle.py) to golden vectors captured from the reference
itself (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import mean_shift_oracle as O
from tests.golden.cases import MEANSHIFT_CASES, KAPPA, EPSILON, SEED_CONTINUATION_CASES, continuation_inputs
from unseenobjectclustering_amd import synth

SMALL = [k for k, c in MEANSHIFT_CASES.items() if c["H"] * c["W"] <= 224 * 224]


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "meanshift.npz"))


@pytest.mark.parametrize("name", SMALL + ["full_480x640_a"])
def test_oracle_matches_reference_golden(golden, name):
    c = MEANSHIFT_CASES[name]
    X, _ = synth.embedding_field(c["seed"], c["H"], c["W"], 64, c["num_objects"], c["noise"])
    Xt = torch.from_numpy(X)
    first = int(golden[name + "/indices"][0])
    labels, idx, parts = O.mean_shift_smart_init(Xt, KAPPA, c["m"], c["iters"], first_index=first,
                                                 epsilon=EPSILON, return_parts=True)
    # integer work: bit-exact
    assert np.array_equal(idx.numpy().astype(np.int32), golden[name + "/indices"])
    assert np.array_equal(parts["seed_labels"].numpy().astype(np.int32), golden[name + "/seed_labels"])
    assert np.array_equal(labels.numpy().astype(np.uint8), golden[name + "/labels"])
    # floating point: same torch CPU ops => tight
    assert np.abs(parts["Z"].numpy() - golden[name + "/Z"]).max() < 1e-6


def test_seed_cc_quirks():
    """Mode-of-present-labels + overwrite (mean_shift.py:66-74): chain a-b-c where b joins a's
    component first, then c's ball contains b (labelled) and takes the mode."""
    def unit(theta):
        v = np.zeros(64, np.float32)
        v[0], v[1] = np.cos(theta), np.sin(theta)
        return v
    # cosine distance 0.5(1-cos(dtheta)) <= 0.04  <=>  dtheta <= ~0.4027 rad
    Z = torch.from_numpy(np.stack([unit(0.0), unit(0.35), unit(0.70), unit(2.0)]))
    lab = O.seed_connected_components(Z, 0.04).numpy()
    # seed0 ball = {0,1} -> label 0 ; seed2 ball = {1,2}: present {0,-1} -> mode 0, overwrites ; seed3 alone
    assert lab.tolist() == [0, 0, 0, 1]


def test_partition_compare():
    assert O.labels_equal_up_to_permutation([0, 0, 1, 2], [5, 5, 3, 1])
    assert not O.labels_equal_up_to_permutation([0, 0, 1, 2], [5, 4, 3, 1])
    assert not O.labels_equal_up_to_permutation([0, 1, 1, 2], [5, 5, 5, 1])


@pytest.mark.parametrize("name", list(SEED_CONTINUATION_CASES))
def test_oracle_seed_continuation_matches_reference(golden_dir, name):
    """select_smart_seeds(init_seeds=..., num_init_seeds=k), mean_shift.py:142-170, against the reference's own run."""
    g = np.load(os.path.join(golden_dir, "seedcont.npz"))
    c = SEED_CONTINUATION_CASES[name]
    X, init = continuation_inputs(c)
    Xt, it = torch.from_numpy(X), torch.from_numpy(init)
    if c["init"] == "rows":
        plain, pidx = O.select_seeds(Xt, c["m"], int(g[name + "/plain_indices"][0]))
        assert np.array_equal(pidx.numpy().astype(np.int32), g[name + "/plain_indices"])
        it[:c["k"]] = plain[:c["k"]]
    first = int(g[name + "/indices"][0]) if c["k"] == 0 else None
    seeds, idx = O.select_seeds(Xt, c["m"], first, init_seeds=it, num_init_seeds=c["k"])
    assert np.array_equal(idx.numpy().astype(np.int32), g[name + "/indices"])
    assert np.array_equal(seeds.numpy(), g[name + "/seeds"])
    assert (g[name + "/indices"][:c["k"]] == -1).all()
