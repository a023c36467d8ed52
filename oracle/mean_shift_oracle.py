"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the reference's
seeded von-Mises-Fisher mean-shift, cosine metric only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker / reported CPU baseline.  The product path
(unseenobjectclustering_amd.*) never imports anything under oracle/.

Pinned: tests/test_oracle_golden.py checks every function here against golden vectors
captured from the reference itself (tests/golden/make_golden.py imports /root/reference/lib
in the build container).  Same torch CPU ops as the reference (torch.mm / exp / argmax /
argmin / F.normalize) so that the arithmetic is the reference's arithmetic.

Reference: /root/reference/lib/utils/mean_shift.py (cited per function below).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def cosine_distance_to(X: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """0.5 * (1 - X v) for one unit vector v [d]  (mean_shift.py:162,184)."""
    return 0.5 * (1 - torch.mm(X, v.unsqueeze(1))[:, 0])


def select_seeds(X: torch.Tensor, num_seeds: int, first_index, init_seeds: torch.Tensor = None, num_init_seeds: int = 0):
    """Farthest-point seed selection in cosine distance (mean_shift.py:128-189).

    The reference draws ``first_index`` from the global NumPy RNG (:155); here it is an
    argument.  The reference keeps a [n, num_seeds] distance matrix and re-reduces
    ``min(distances[:, :i])`` each step (:174); a running minimum is the same value
    bit for bit (min is exact), so that is what is kept here.
    With ``init_seeds`` [num_seeds, d] the first ``num_init_seeds`` rows are taken as chosen (:142-170): their
    distances enter the minimum, their indices stay -1, no first index is used unless num_init_seeds == 0.
    Returns (seeds [m,d], indices [m] int64).
    """
    n, d = X.shape
    idx = torch.full((num_seeds,), -1, dtype=torch.long)
    if init_seeds is None or num_init_seeds == 0:
        seeds = torch.empty((num_seeds, d), dtype=X.dtype) if init_seeds is None else init_seeds.clone()
        idx[0] = int(first_index)
        seeds[0] = X[int(first_index)]
        dmin = cosine_distance_to(X, seeds[0])
        start = 1
    else:
        seeds = init_seeds.clone()
        dmin = cosine_distance_to(X, seeds[0])
        for i in range(1, num_init_seeds):
            dmin = torch.minimum(dmin, cosine_distance_to(X, seeds[i]))          # :165-170
        start = num_init_seeds
    for i in range(start, num_seeds):
        j = torch.argmax(dmin)                       # :175 (first maximal index)
        idx[i] = j
        seeds[i] = X[j]
        dmin = torch.minimum(dmin, cosine_distance_to(X, seeds[i]))   # :184 + :174 of next step
    return seeds, idx


def hill_climb(X: torch.Tensor, Z: torch.Tensor, kappa: float, max_iters: int = 10) -> torch.Tensor:
    """max_iters x { W = exp(kappa Z X^T); Z = l2normalize(W X) }  (mean_shift.py:79-109, :26)."""
    for _ in range(max_iters):
        W = torch.exp(kappa * torch.mm(Z, X.t()))
        Z = F.normalize(torch.mm(W, X), p=2, dim=1)
    return Z


def seed_connected_components(Z: torch.Tensor, epsilon: float) -> torch.Tensor:
    """Sequential epsilon-ball labelling of the converged seeds (mean_shift.py:41-76).

    Quirks kept: a component that already contains labelled seeds takes the MODE of those
    labels (ties -> smallest label, :30-38) and OVERWRITES every member (:74), so label
    ids can end up with gaps.
    """
    n = Z.shape[0]
    labels = np.full(n, -1, dtype=np.int64)
    next_label = 0
    for i in range(n):
        if labels[i] != -1:
            continue
        dist = 0.5 * (1 - torch.mm(Z, Z[i:i + 1].t()))[:, 0]
        member = (dist <= epsilon).numpy()
        present = labels[member]
        if np.unique(present).shape[0] > 1:
            known = present[present != -1]
            vals, counts = np.unique(known, return_counts=True)
            lab = int(vals[np.argmax(counts)])
        else:
            lab = next_label
            next_label += 1
        labels[member] = lab
    return torch.from_numpy(labels)


def assign_to_seeds(X: torch.Tensor, Z: torch.Tensor, seed_labels: torch.Tensor):
    """Nearest-seed labels + 'largest cluster becomes 0' swap (mean_shift.py:211-227).

    Quirk kept: only labels in range(len(unique(seed_labels))) are counted (:217-221).
    Returns (labels [n] int64, closest_seed [n] int64).
    """
    dist = 0.5 * (1 - torch.mm(X, Z.t()))
    closest = torch.argmin(dist, dim=1)
    labels = seed_labels[closest].clone()
    num = int(torch.unique(seed_labels).numel())
    count = torch.zeros(num, dtype=torch.long)
    for i in range(num):
        count[i] = (labels == i).sum()
    big = int(torch.argmax(count)) if num > 0 else 0
    if big != 0:
        zero = labels == 0
        other = labels == big
        labels[zero] = big
        labels[other] = 0
    return labels, closest


def mean_shift_smart_init(X: torch.Tensor, kappa: float, num_seeds: int = 100, max_iters: int = 10,
                          first_index: int = 0, epsilon: float = 0.04, return_parts: bool = False):
    """Full clustering of one embedding field (mean_shift.py:192-229 with :112-125).

    epsilon = 2 * cfg.TRAIN.EMBEDDING_ALPHA = 0.04 (:123, config.py:254).
    """
    seeds, idx = select_seeds(X, num_seeds, first_index)
    Z = hill_climb(X, seeds, kappa, max_iters)
    seed_labels = seed_connected_components(Z, epsilon)
    labels, closest = assign_to_seeds(X, Z, seed_labels)
    if return_parts:
        return labels, idx, dict(seeds=seeds, Z=Z, seed_labels=seed_labels, closest=closest)
    return labels, idx


def labels_equal_up_to_permutation(a, b) -> bool:
    """True iff the two integer label maps induce the same partition (bijection between ids)."""
    a = np.asarray(a).reshape(-1).astype(np.int64)
    b = np.asarray(b).reshape(-1).astype(np.int64)
    if a.shape != b.shape:
        return False
    pairs = np.unique(np.stack([a, b], 1), axis=0)
    return len(np.unique(pairs[:, 0])) == len(pairs) and len(np.unique(pairs[:, 1])) == len(pairs)
