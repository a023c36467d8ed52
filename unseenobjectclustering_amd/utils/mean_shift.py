"""Host-side mirror of the reference's lib/utils/mean_shift.py (cosine metric), backed by the
HIP kernels in csrc/meanshift.hip through the C ABI (include/uoc_hip.h).

Same names, argument meaning and return types as the reference:
    mean_shift_smart_init(X, kappa, num_seeds=100, max_iters=10, metric='cosine')
        -> (cluster_labels [n] int64, selected_indices [num_seeds] int64)      mean_shift.py:192
    select_smart_seeds / seed_hill_climbing_ball / connected_components / mean_shift_with_seeds

The first seed of every call is drawn with np.random.randint(0, n) exactly like the reference
(mean_shift.py:155), on the host, and handed to the kernels.  There is no CPU fallback: tensors
must live on a ROCm device and libuoc_hip.so must be built.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _native
from ..fcn.config import cfg

EMBED_DIM = 64          # one 64-channel "half"; 128-d ('cat' fusion) fields are two of them
_ws_cache = {}


def _workspace(device, nbytes: int) -> torch.Tensor:
    key = _native.stream_key(device)      # one scratch buffer per stream (frames in flight do not share it)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        _native.retire(ws)          # a captured hipGraph may have baked the old buffer's address
        ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def _check_points(X: torch.Tensor, what="X") -> torch.Tensor:
    if not X.is_cuda:
        raise _native.NativeError(f"{what} must be on a ROCm device (no CPU fallback); got {X.device}")
    if X.dtype != torch.float32:
        raise TypeError(f"{what} must be float32, got {X.dtype}")
    if X.shape[-1] != EMBED_DIM:
        raise NotImplementedError(f"kernels are specialised for d={EMBED_DIM}; got d={X.shape[-1]}")
    return X.contiguous()


def _require_cosine(metric):
    if metric != "cosine":
        raise NotImplementedError("only metric='cosine' is implemented on gfx950")


def _first_indices(first_index, B: int, n: int) -> torch.Tensor:
    """The B first-seed indices as an int32 host tensor, validated: the kernels read row X[first] directly.
    A device int32 tensor [B] passes through as it is (fcn/graph_replay.py: the indices live in a static buffer that is
    refilled before every replay; the caller validated them)."""
    if torch.is_tensor(first_index) and first_index.is_cuda:
        if first_index.dtype != torch.int32 or first_index.numel() != B or not first_index.is_contiguous():
            raise ValueError(f"first_index on the device must be a contiguous int32 tensor of {B} indices")
        return first_index
    first = np.asarray(first_index, dtype=np.int64).reshape(-1)
    if first.shape[0] != B:
        raise ValueError(f"first_index: expected {B} indices, got {first.shape[0]}")
    if first.size and (first.min() < 0 or first.max() >= n):
        raise ValueError(f"first_index out of range [0, {n}): {first.tolist()}")
    return torch.from_numpy(first.astype(np.int32)).pin_memory()     # pinned: the upload never blocks on the stream


def to_planes(X: torch.Tensor) -> torch.Tensor:
    """[B, n, 128] pixel-major rows -> the [B, 2, n, 64] plane layout of the 128-d kernels (a copy)."""
    B, n, d = X.shape
    return X.view(B, n, d // EMBED_DIM, EMBED_DIM).permute(0, 2, 1, 3).contiguous()


def cluster_batch(X: torch.Tensor, first_index, kappa: float = 20.0, num_seeds: int = 100, max_iters: int = 10,
                  epsilon: float = None, return_parts: bool = False):
    """Cluster B independent fields in one set of launches.

    X [B, n, 64] float32 unit rows (pixel-major) — or, for 128-d embeddings, the plane layout
    [B, 2, n, 64] (to_planes) — first_index: B ints.
    Returns labels [B, n] int32 and indices [B, num_seeds] int32 (device tensors); with
    return_parts also the converged seeds Z [B, m, 64] ([B, 2, m, 64]) and their labels [B, m].
    """
    X = _check_points(X)
    if X.dim() == 4:
        return _cluster_batch_wide(X, first_index, kappa, num_seeds, max_iters, epsilon, return_parts)
    assert X.dim() == 3
    B, n, _ = X.shape
    if epsilon is None:
        epsilon = 2 * cfg.TRAIN.EMBEDDING_ALPHA
    dev = X.device
    L = _native.lib()
    first = _first_indices(first_index, B, n).to(dev, non_blocking=True)
    labels = torch.empty((B, n), dtype=torch.int32, device=dev)
    indices = torch.empty((B, num_seeds), dtype=torch.int32, device=dev)
    Z = torch.empty((B, num_seeds, EMBED_DIM), dtype=torch.float32, device=dev)
    seed_labels = torch.empty((B, num_seeds), dtype=torch.int32, device=dev)
    nbytes = L.uoc_ms_workspace_bytes(B, n, num_seeds)
    ws = _workspace(dev, nbytes)
    with torch.cuda.device(dev):
        rc = L.uoc_ms_cluster(_native.ptr(X), B, n, num_seeds, float(kappa), int(max_iters), float(epsilon),
                              _native.ptr(first), _native.ptr(labels), _native.ptr(indices), _native.ptr(Z),
                              _native.ptr(seed_labels), _native.ptr(ws), ws.numel(), _native.stream_ptr(dev))
    _native.check(rc, "uoc_ms_cluster")
    if return_parts:
        return labels, indices, Z, seed_labels
    return labels, indices


def _cluster_batch_wide(X, first_index, kappa, num_seeds, max_iters, epsilon, return_parts):
    B, H2, n, _ = X.shape
    if H2 != 2:
        raise NotImplementedError("embedding dimension must be 64 or 128 (two 64-channel planes)")
    if epsilon is None:
        epsilon = 2 * cfg.TRAIN.EMBEDDING_ALPHA
    dev = X.device
    L = _native.lib()
    first = _first_indices(first_index, B, n).to(dev, non_blocking=True)
    labels = torch.empty((B, n), dtype=torch.int32, device=dev)
    indices = torch.empty((B, num_seeds), dtype=torch.int32, device=dev)
    Z = torch.empty((B, H2, num_seeds, EMBED_DIM), dtype=torch.float32, device=dev)
    seed_labels = torch.empty((B, num_seeds), dtype=torch.int32, device=dev)
    ws = _workspace(dev, L.uoc_ms_workspace_bytes_wide(B, n, num_seeds, H2))
    with torch.cuda.device(dev):
        rc = L.uoc_ms_cluster_wide(_native.ptr(X), H2, B, n, num_seeds, float(kappa), int(max_iters), float(epsilon),
                                   _native.ptr(first), _native.ptr(labels), _native.ptr(indices), _native.ptr(Z),
                                   _native.ptr(seed_labels), _native.ptr(ws), ws.numel(), _native.stream_ptr(dev))
    _native.check(rc, "uoc_ms_cluster_wide")
    if return_parts:
        return labels, indices, Z, seed_labels
    return labels, indices


def mean_shift_smart_init(X, kappa, num_seeds=100, max_iters=10, metric="cosine"):
    """mean_shift.py:192-229.  X [n, d] unit rows on the GPU, d = 64 or 128 -> (labels [n] int64, indices [m] int64)."""
    _require_cosine(metric)
    n = X.shape[0]
    first = np.random.randint(0, n)          # mean_shift.py:155 — same global-RNG draw as the reference
    Xb = X.unsqueeze(0)
    if X.shape[-1] == 2 * EMBED_DIM:
        if not X.is_cuda:
            raise _native.NativeError("X must be on a ROCm device (no CPU fallback)")
        Xb = to_planes(Xb.float())
    labels, indices = cluster_batch(Xb, [first], kappa, num_seeds, max_iters)
    idx = indices[0].long().cpu()            # synchronises; then ask whether the grid exchange completed
    with torch.cuda.device(labels.device):
        _native.check(_native.lib().uoc_ms_check(_native.stream_ptr(labels.device)), "uoc_ms_check")
    return labels[0].long(), idx


def select_smart_seeds(X, num_seeds, return_selected_indices=False, init_seeds=None, num_init_seeds=None,
                       metric="cosine"):
    """mean_shift.py:128-189.  With init_seeds [num_seeds, d] (first num_init_seeds rows already chosen, :142-170) the
    selection continues from them and — as in the reference, where `seeds = init_seeds` — is written into that tensor;
    the returned indices are -1 for the given rows.  The global RNG is drawn only when no seed is given yet (:154-155)."""
    _require_cosine(metric)
    X = _check_points(X)
    n = X.shape[0]
    dev = X.device
    L = _native.lib()
    if init_seeds is None:
        seeds = torch.empty((num_seeds, EMBED_DIM), dtype=torch.float32, device=dev)
        num_init = 0
    else:
        if num_init_seeds is None:
            raise TypeError("num_init_seeds is required with init_seeds")      # the reference fails on range(None) / None == 0
        num_init = int(num_init_seeds)
        if tuple(init_seeds.shape) != (num_seeds, EMBED_DIM) or init_seeds.device != dev:
            raise ValueError(f"init_seeds must be [{num_seeds}, {EMBED_DIM}] on {dev}; got {tuple(init_seeds.shape)} on "
                             f"{init_seeds.device}")
        if not 0 <= num_init <= num_seeds:
            raise ValueError(f"num_init_seeds={num_init} out of range [0, {num_seeds}]")
        if init_seeds.dtype != torch.float32 or not init_seeds.is_contiguous():
            raise TypeError("init_seeds must be a contiguous float32 tensor (it receives the selection in place)")
        seeds = init_seeds
    first = None
    if num_init == 0:
        first = torch.tensor([np.random.randint(0, n)], dtype=torch.int32).to(dev)
    indices = torch.empty((num_seeds,), dtype=torch.int32, device=dev)
    ws = _workspace(dev, L.uoc_ms_workspace_bytes(1, n, num_seeds))
    with torch.cuda.device(dev):
        rc = L.uoc_ms_select_seeds_from(_native.ptr(X), 1, n, num_seeds, num_init, _native.ptr(first), _native.ptr(seeds),
                                        _native.ptr(indices), _native.ptr(ws), ws.numel(), _native.stream_ptr(dev))
    _native.check(rc, "uoc_ms_select_seeds_from")
    if return_selected_indices:
        return seeds, indices.long().cpu()
    return (seeds,)


def seed_hill_climbing_ball(X, Z, kappa, max_iters=10, metric="cosine"):
    """mean_shift.py:79-109.  Returns the updated seeds (a new tensor, Z is not modified)."""
    _require_cosine(metric)
    X = _check_points(X)
    Zc = _check_points(Z, "Z").clone()
    n, m = X.shape[0], Zc.shape[0]
    L = _native.lib()
    ws = _workspace(X.device, L.uoc_ms_workspace_bytes(1, n, m))
    with torch.cuda.device(X.device):
        rc = L.uoc_ms_hill_climb(_native.ptr(X), 1, n, _native.ptr(Zc), m, float(kappa), int(max_iters),
                                 _native.ptr(ws), ws.numel(), _native.stream_ptr(X.device))
    _native.check(rc, "uoc_ms_hill_climb")
    return Zc


def connected_components(Z, epsilon, metric="cosine"):
    """mean_shift.py:41-76.  Z [m, d] -> [m] int64 labels (CPU tensor, like the reference)."""
    _require_cosine(metric)
    Zc = _check_points(Z, "Z")
    m = Zc.shape[0]
    L = _native.lib()
    out = torch.empty((m,), dtype=torch.int32, device=Zc.device)
    nu = torch.empty((1,), dtype=torch.int32, device=Zc.device)
    with torch.cuda.device(Zc.device):
        rc = L.uoc_ms_seed_components(_native.ptr(Zc), 1, m, float(epsilon), _native.ptr(out), _native.ptr(nu),
                                      _native.stream_ptr(Zc.device))
    _native.check(rc, "uoc_ms_seed_components")
    return out.long().cpu()


def mean_shift_with_seeds(X, Z, kappa, max_iters=10, metric="cosine"):
    """mean_shift.py:112-125: hill climbing then seed connected components (eps = 2*alpha)."""
    Znew = seed_hill_climbing_ball(X, Z, kappa, max_iters=max_iters, metric=metric)
    labels = connected_components(Znew, 2 * cfg.TRAIN.EMBEDDING_ALPHA, metric=metric)
    return labels, Znew
