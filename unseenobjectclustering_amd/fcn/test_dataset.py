"""Host-side mirror of the reference's lib/fcn/test_dataset.py inference surface
(clustering_features :44, crop_rois :62, match_label_crop :116, filter_labels_depth :183,
test_sample :232, test_segnet :271), backed by libuoc_hip.so.

Same names, argument meaning and return conventions as the reference (labels are float32
tensors, `out_label` lives on the CPU, ROIs are [K,4] float x0,y0,x1,y1 inclusive, the global
NumPy RNG is consumed once per clustered field in the reference's order).  Internally labels stay
int32 on the device and the per-object Python loops of the reference are single batched launches.
There is no CPU fallback: inputs are moved to the ROCm device and every stage runs as HIP kernels.
"""
from __future__ import annotations

import ctypes
import os
import time

import numpy as np
import torch

from .. import _native
from ..utils.mean_shift import cluster_batch, to_planes
from .config import cfg, require_supported, uses_depth

KAPPA = 20            # test_dataset.py:51
MAX_ITERS = 10        # :56
PAD_FRACTION = 0.25   # :66
DEPTH_FILTER = 0.8    # :252
MAX_LABELS = 128

_roi_ws = {}
LAST_FRAME_STATS = {"rois": 0}      # number of stage-1 ROIs of the most recent frame (measurement bookkeeping)


def _device():
    if cfg.device is not None and torch.device(cfg.device).type == "cuda":
        return torch.device(cfg.device)
    if not torch.cuda.is_available():
        raise _native.NativeError("no ROCm device visible: the HIP path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _ws(dev):
    key = _native.stream_key(dev)      # one scratch buffer per stream: two frames in flight must not share it
    if key not in _roi_ws:
        _roi_ws[key] = torch.empty(_native.lib().uoc_roi_workspace_bytes() + 256, dtype=torch.uint8, device=dev)
    return _roi_ws[key]


def _pixel_major(features: torch.Tensor) -> torch.Tensor:
    """[B,C,h,w] -> [B,h*w,C] contiguous; zero-copy for the channels-last view SEGNET.forward returns."""
    B, C, h, w = features.shape
    x = features.permute(0, 2, 3, 1)
    if not x.is_contiguous():
        x = x.contiguous()
    return x.reshape(B, h * w, C)


def _detach_keep_planes(features: torch.Tensor) -> torch.Tensor:
    """features.detach() like the reference (:247), keeping the kernel-layout buffer the 128-d 'cat' output of
    SEGNET.forward carries as `_uoc_planes` (detach() returns a new tensor object without Python attributes)."""
    planes = getattr(features, "_uoc_planes", None)
    features = features.detach()
    if planes is not None:
        features._uoc_planes = planes
    return features


def _cluster_device(features: torch.Tensor, num_seeds: int = 100, rng=None):
    """Device-resident clustering of every batch item: int32 labels [B, h*w], indices [B, m].
    `rng`: where the first-seed draws come from — the global NumPy RNG like the reference (mean_shift.py:155), or a
    per-frame np.random.RandomState when several frames are in flight and must not interleave their draws."""
    require_supported()
    if not features.is_cuda:
        raise _native.NativeError("features must be on a ROCm device (no CPU fallback)")
    planes = getattr(features, "_uoc_planes", None)     # 128-d output of SEGNET ('cat' fusion): already in kernel layout
    if planes is not None and planes.shape[0] == features.shape[0]:
        X = planes
    else:
        X = _pixel_major(features.float())
        if X.shape[-1] == 128:
            X = to_planes(X)
    B, n = X.shape[0], X.shape[-2]
    draw = (rng if rng is not None else np.random).randint
    firsts = [draw(0, n) for _ in range(B)]                # mean_shift.py:155, one draw per field, in order
    return cluster_batch(X, firsts, KAPPA, num_seeds, MAX_ITERS, 2 * cfg.TRAIN.EMBEDDING_ALPHA)


def clustering_features(features, num_seeds=100):
    """test_dataset.py:44-59 -> (out_label [B,h,w] float32 on the CPU, list of [m] int64 seed indices)."""
    labels, indices = _cluster_device(features, num_seeds)
    B, _, h, w = features.shape
    out_label = labels.view(B, h, w).float().cpu()
    _check_clustering(labels.device)
    return out_label, [indices[j].long().cpu() for j in range(B)]


def _labels_to_device(labels: torch.Tensor, dev) -> torch.Tensor:
    lab = labels.to(dev)
    if lab.dtype != torch.int32:
        lab = lab.round().to(torch.int32)
    lab = lab.contiguous()
    if lab.numel() and (int(lab.max()) >= MAX_LABELS or int(lab.min()) < 0):
        raise ValueError(f"label ids must be in [0, {MAX_LABELS})")
    return lab


def filter_labels_depth(labels, depth, threshold):
    """test_dataset.py:183-198.  Returns a new tensor like `labels`."""
    dev = depth.device if depth.is_cuda else _device()
    lab = _labels_to_device(labels, dev).clone()
    d = depth.to(dev).float().contiguous()
    B, H, W = lab.shape
    L = _native.lib()
    ws = _ws(dev)
    zptr = ctypes.c_void_p(d.data_ptr() + 2 * H * W * 4)
    with torch.cuda.device(dev):
        rc = L.uoc_filter_labels_depth(_native.ptr(lab), zptr, 3 * H * W, B, H, W, float(threshold), _native.ptr(ws),
                                       ws.numel(), _native.stream_ptr(dev))
    _native.check(rc, "uoc_filter_labels_depth")
    return lab.to(labels.dtype).to(labels.device)


def _read_table(table_dev: torch.Tensor) -> _native.RoiTable:
    host = table_dev.cpu().numpy().tobytes()      # one small D2H (2.5 KB); synchronises the stream
    return _native.RoiTable.from_buffer_copy(host)


def _check_clustering(dev):
    """Raise if a clustering launch since the last check reported a timed-out grid exchange (uoc_ms_check)."""
    with torch.cuda.device(dev):
        _native.check(_native.lib().uoc_ms_check(_native.stream_ptr(dev)), "uoc_ms_check")


def _build_rois(lab0: torch.Tensor, z_plane_ptr, H, W, dev, threshold=DEPTH_FILTER):
    L = _native.lib()
    table = torch.empty(_native.ROI_TABLE_BYTES, dtype=torch.uint8, device=dev)
    ws = _ws(dev)
    with torch.cuda.device(dev):
        rc = L.uoc_roi_build(_native.ptr(lab0), z_plane_ptr, H, W, float(threshold), PAD_FRACTION, _native.ptr(table),
                             _native.ptr(ws), ws.numel(), _native.stream_ptr(dev))
    _native.check(rc, "uoc_roi_build")
    return table


def _crop(rgb, depth, lab0, table, K, H, W, dev):
    S = cfg.TRAIN.SYN_CROP_SIZE
    L = _native.lib()
    rgb_crops = torch.empty((K, 3, S, S), dtype=torch.float32, device=dev)
    depth_crops = torch.empty((K, 3, S, S), dtype=torch.float32, device=dev) if depth is not None else None   # :73-76
    mask_crops = torch.empty((K, S, S), dtype=torch.float32, device=dev)
    if K > 0:
        with torch.cuda.device(dev):
            rc = L.uoc_roi_crop(_native.ptr(rgb), _native.ptr(depth), _native.ptr(lab0), H, W, _native.ptr(table), K, S,
                                _native.ptr(rgb_crops), _native.ptr(depth_crops), _native.ptr(mask_crops),
                                _native.stream_ptr(dev))
        _native.check(rc, "uoc_roi_crop")
    return rgb_crops, mask_crops, depth_crops


def crop_rois(rgb, initial_masks, depth):
    """test_dataset.py:62-112 -> (rgb_crops [K,3,S,S], mask_crops [K,S,S], rois [K,4], depth_crops)."""
    dev = rgb.device if rgb.is_cuda else _device()
    rgb = rgb.to(dev).float().contiguous()
    depth = depth.to(dev).float().contiguous() if depth is not None else None
    N, H, W = initial_masks.shape
    lab0 = _labels_to_device(initial_masks[0], dev)
    table = _build_rois(lab0, ctypes.c_void_p(0), H, W, dev)
    t = _read_table(table)
    K = int(t.K)
    rgb_crops, mask_crops, depth_crops = _crop(rgb, depth, lab0, table, K, H, W, dev)
    rois = torch.tensor([[t.box[k][i] for i in range(4)] for k in range(K)], dtype=torch.float32, device=dev).reshape(K, 4)
    return rgb_crops, mask_crops, rois, depth_crops


def _order_and_map(keep: np.ndarray, sort_key: torch.Tensor):
    """Host part of match_label_crop: ROI paint order by mean depth, far first — or by box area, large
    first, when there is no depth (:129-151, Python's stable sort with reverse=True on the 0-dim
    tensors, NaN behaviour included) and the global renumbering of kept clusters in that order (:156-163)."""
    K = keep.shape[0]
    keys = [(i, sort_key[i]) for i in range(K)]
    order = [i for i, _ in sorted(keys, key=lambda kv: kv[1], reverse=True)]
    mapping = np.zeros((K, MAX_LABELS), dtype=np.int32)
    count = 0
    for i in order:
        for c in np.nonzero(keep[i])[0]:
            count += 1
            mapping[i, c] = count
    return np.asarray(order, dtype=np.int32), mapping


class _HostMirror:
    """Pinned host buffers of one frame slot: the two small device->host reads of a frame (ROI table; keep table + mean
    depths) and the host->device paint plan go through them as asynchronous copies guarded by an event, so the host
    can queue another frame's work on another stream while a read is in flight."""

    def __init__(self, frames=1):
        n = MAX_LABELS * MAX_LABELS + MAX_LABELS
        self.frames = frames
        self.tables = torch.empty((frames, _native.ROI_TABLE_BYTES), dtype=torch.uint8).pin_memory()
        self.stats_all = torch.empty((frames, n), dtype=torch.int32).pin_memory()
        self.plans = torch.empty((frames, n), dtype=torch.int32).pin_memory()
        self.table, self.stats, self.plan = self.tables[0], self.stats_all[0], self.plans[0]
        # blocking events: a host thread that waits on one sleeps instead of spinning (the runner's core budget per rank)
        blocking = os.environ.get("UOC_PIPE_BLOCKING_EVENTS", "0") == "1"
        self.table_ready = torch.cuda.Event(blocking=blocking)
        self.stats_ready = torch.cuda.Event(blocking=blocking)


_mirrors = {}


def _mirror(dev, frames=1) -> _HostMirror:
    key = _native.stream_key(dev)
    if key not in _mirrors or _mirrors[key].frames < frames:
        _mirrors[key] = _HostMirror(frames)
    return _mirrors[key]


def _match_stats(labels_crop_i32, mask_crops, depth_crops, K, dev):
    """Overlap / mean-depth statistics of the K clustered crops (:118-148), one launch over all ROIs.
    Returns the device buffer [K*128 keep flags | K mean depths]."""
    S = cfg.TRAIN.SYN_CROP_SIZE
    L = _native.lib()
    ws = _ws(dev)
    # keep table and mean depths share one buffer (one D2H), paint order and id map another (one H2D)
    stats = torch.empty((K * MAX_LABELS + K,), dtype=torch.int32, device=dev)
    keep = stats[:K * MAX_LABELS].view(K, MAX_LABELS)
    meanz = stats[K * MAX_LABELS:].view(torch.float32) if depth_crops is not None else None
    with torch.cuda.device(dev):
        rc = L.uoc_roi_match_stats(_native.ptr(labels_crop_i32), _native.ptr(mask_crops), _native.ptr(depth_crops), K, S,
                                   _native.ptr(keep), _native.ptr(meanz), _native.ptr(ws), ws.numel(),
                                   _native.stream_ptr(dev))
    _native.check(rc, "uoc_roi_match_stats")
    return stats


def _paste(labels_crop_i32, table_dev, table_host, stats_h, has_depth, K, H, W, dev, plan_host=None):
    """Host ordering of the ROIs (:129-151) from the statistics read back, then one paste launch (:156-177)."""
    S = cfg.TRAIN.SYN_CROP_SIZE
    L = _native.lib()
    keep_h = stats_h[:K * MAX_LABELS].view(K, MAX_LABELS).numpy()
    if has_depth:
        sort_key = stats_h[K * MAX_LABELS:K * MAX_LABELS + K].view(torch.float32)
    else:       # :138-146 roi_size = (y_max - y_min + 1) * (x_max - x_min + 1), float32 like the reference's rois
        box = torch.tensor(np.ctypeslib.as_array(table_host.box)[:K].astype(np.float32))
        sort_key = (box[:, 3] - box[:, 1] + 1) * (box[:, 2] - box[:, 0] + 1)
    order, mapping = _order_and_map(keep_h, sort_key)
    flat = np.concatenate([order.reshape(-1), mapping.reshape(-1)]).astype(np.int32)
    if plan_host is not None:
        plan_host[:flat.size].copy_(torch.from_numpy(flat))
        plan = plan_host[:flat.size].to(dev, non_blocking=True)
    else:
        plan = torch.from_numpy(flat).to(dev)
    order_d, map_d = plan[:K], plan[K:]
    refined = torch.empty((H * W,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L.uoc_roi_paste(_native.ptr(labels_crop_i32), _native.ptr(table_dev), _native.ptr(map_d), _native.ptr(order_d),
                             K, S, H, W, _native.ptr(refined), _native.stream_ptr(dev))
    _native.check(rc, "uoc_roi_paste")
    return refined


_order_status = {}


def _status_word(dev) -> torch.Tensor:
    """Per-stream sticky device flag of uoc_roi_match (bit 0: NaN sort keys with >= 64 ROIs — the one ordering case the
    device kernel does not restate; the caller then orders on the host)."""
    key = _native.stream_key(dev)
    if key not in _order_status:
        _order_status[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return _order_status[key]


def _match_device(labels_crop_i32, mask_crops, depth_crops, table, K, H, W, dev, want_keep=False):
    """match_label_crop (:116-179) with no host round trip (round 6): statistics, ROI paint order (Python's
    sorted(reverse=True) semantics restated on the device), renumbering and paste in one call (uoc_roi_match).
    -> refined int32 [H*W] (device), keep table [K,128] (device) or None."""
    S = cfg.TRAIN.SYN_CROP_SIZE
    L = _native.lib()
    ws = _ws(dev)
    refined = torch.empty((H * W,), dtype=torch.int32, device=dev)
    keep = torch.empty((K, MAX_LABELS), dtype=torch.int32, device=dev) if want_keep else None
    status = _status_word(dev)
    with torch.cuda.device(dev):
        rc = L.uoc_roi_match(_native.ptr(labels_crop_i32), _native.ptr(mask_crops), _native.ptr(depth_crops), _native.ptr(table),
                             K, S, H, W, _native.ptr(refined), _native.ptr(keep), None, _native.ptr(status), _native.ptr(ws),
                             ws.numel(), _native.stream_ptr(dev))
    _native.check(rc, "uoc_roi_match")
    return refined, keep


def _order_needs_host(dev) -> bool:
    """Reads and clears the sticky ordering flag of this stream (synchronises)."""
    status = _status_word(dev)
    flagged = int(status.item()) != 0
    if flagged:
        status.zero_()
    return flagged


def _finish_block(dev):
    """End of a block of frames whose per-frame checks were deferred (the frame-parallel runner): the clustering status
    (uoc_ms_check) and the ordering flag of every stream of `dev`."""
    _check_clustering(dev)
    flagged = False
    for key, status in _order_status.items():
        if key[:2] == (dev.type, dev.index) and int(status.item()) != 0:
            status.zero_()
            flagged = True
    if flagged:
        raise HostOrderNeeded("a frame had NaN ROI sort keys with >= 64 ROIs: re-run with the ROI ordering on the host")


def _match_host_order(labels_crop_i32, mask_crops, depth_crops, table, K, H, W, dev):
    """The same with the ROI ordering on the host (Python's own sorted on the statistics read back): the fallback for NaN
    keys with >= 64 ROIs, and the path of rounds 1-5."""
    stats = _match_stats(labels_crop_i32, mask_crops, depth_crops, K, dev)
    stats_h = stats.cpu()
    table_host = _read_table(table) if depth_crops is None else None
    refined = _paste(labels_crop_i32, table, table_host, stats_h, depth_crops is not None, K, H, W, dev)
    return refined, stats[:K * MAX_LABELS].view(K, MAX_LABELS)


def _match(labels_crop_i32, mask_crops, depth_crops, table, K, H, W, dev):
    """labels_crop_i32 [K, S*S] int32 (device) -> refined int32 [H*W] (device), keep table (device)."""
    refined, keep = _match_device(labels_crop_i32, mask_crops, depth_crops, table, K, H, W, dev, want_keep=True)
    if _order_needs_host(dev):
        return _match_host_order(labels_crop_i32, mask_crops, depth_crops, table, K, H, W, dev)
    return refined, keep


def match_label_crop(initial_masks, labels_crop, out_label_crop, rois, depth_crop):
    """test_dataset.py:116-179 -> (refined_masks like initial_masks (float), labels_crop with rejected = -1)."""
    dev = labels_crop.device if labels_crop.is_cuda else _device()
    K = labels_crop.shape[0]
    N, H, W = initial_masks.shape
    refined_masks = torch.zeros_like(initial_masks).float()
    if K == 0:
        return refined_masks, labels_crop
    lab = _labels_to_device(labels_crop, dev).reshape(K, -1)
    t = _native.RoiTable()
    t.K = K
    r = rois.detach().cpu().numpy().astype(np.int64)
    for k in range(K):
        for i in range(4):
            t.box[k][i] = int(r[k, i])
    table = torch.frombuffer(bytearray(bytes(t)), dtype=torch.uint8).to(dev)
    refined, keep = _match(lab, out_label_crop.to(dev).float().contiguous(),
                           depth_crop.to(dev).float().contiguous() if depth_crop is not None else None,
                           table, K, H, W, dev)
    refined_masks[0] = refined.view(H, W).float().to(refined_masks.device)
    kept = torch.gather(keep, 1, lab.long()) != 0
    labels_out = torch.where(kept, lab, torch.full_like(lab, -1)).view(labels_crop.shape).to(labels_crop.dtype)
    return refined_masks, labels_out.to(labels_crop.device)


def test_sample(sample, network, network_crop):
    """test_dataset.py:232-267: (out_label [B,H,W] float32 CPU, out_label_refined [B,H,W] float32 CPU — only item 0 is
    refined, the rest stays zero, as in the reference — or None)."""
    return _run_frame(sample, network, network_crop, DEPTH_FILTER)


FORCE_HOST_ORDER = os.environ.get("UOC_HOST_ORDER", "0") == "1"      # True: ROI ordering on the host (rounds 1-5; the fallback for NaN keys with >= 64 ROIs; A/B)


class HostOrderNeeded(_native.NativeError):
    """A frame of the block had NaN sort keys with >= 64 ROIs: the device ordering does not restate that case, the
    caller re-runs with FORCE_HOST_ORDER (runner.run_sharded does)."""


class FrameJob:
    """One frame on its way through the two-stage path, cut at the ONE point where the host needs a small result from
    the device (the number of ROIs, which sizes the stage-2 launches):

        stage1  embed, cluster, depth filter + ROI table          -> async D2H of the table
        stage2  (needs K) crop, embed + cluster the K crops, match statistics, ROI order + renumbering + paste (device)

    Round 6: the second cut of rounds 1-5 (statistics to the host, Python's sorted, plan back to the device) is gone —
    uoc_roi_match orders on the device.  With host_order=True (FORCE_HOST_ORDER) the old cut is back as stage3.
    `_run_frame` runs the stages back to back; the pipelined runner keeps several FrameGroupJobs in flight.  Results do
    not depend on the interleaving: every job draws its first seeds from its own `rng`."""

    def __init__(self, sample, network, network_crop, depth_threshold, rng=None, host_order=None):
        self.sample, self.network, self.network_crop = sample, network, network_crop
        self.depth_threshold, self.rng = depth_threshold, rng
        self.host_order = FORCE_HOST_ORDER if host_order is None else host_order
        self.K = 0
        self.labels = self.refined = None

    def stage1(self):
        require_supported()
        dev = self.dev = _device()
        sample = self.sample
        if "image_u8" in sample:  # raw uint8 / uint16 sample: input preparation runs on the device (io.prepare_on_device)
            from ..io import prepare_on_device
            sample = dict(sample, **prepare_on_device(sample, dev))
        # non_blocking is a no-op for pageable host memory (the runtime stages it synchronously) and asynchronous for
        # pinned memory; asking the tensor (`is_pinned()`) costs a driver query per call, so just always pass it
        pin = lambda t: t.to(dev, non_blocking=True)
        self.image = image = pin(sample["image_color"]).float().contiguous()
        self.depth = depth = pin(sample["depth"]).float().contiguous() if uses_depth() else None     # :236-239
        if depth is None:
            self.depth_threshold = None                                                           # :250 `if depth is not None`
        thr = self.depth_threshold
        label = sample["label"].to(dev) if "label" in sample else None
        B, _, H, W = image.shape
        self.B, self.H, self.W = B, H, W

        features = _detach_keep_planes(self.network(image, label, depth))          # :247
        labels, _ = _cluster_device(features, num_seeds=100, rng=self.rng)         # [B, H*W] int32 on the device
        self.labels = labels

        # depth filter (:250-252) fused with the ROI table build for item 0; other items filter only
        zptr = ctypes.c_void_p(depth.data_ptr() + 2 * H * W * 4) if thr is not None else ctypes.c_void_p(0)
        self.table = _build_rois(labels[0], zptr, H, W, dev, thr if thr is not None else 0.0)
        if B > 1 and thr is not None:
            L = _native.lib()
            ws = _ws(dev)
            z1 = ctypes.c_void_p(depth.data_ptr() + (3 * H * W + 2 * H * W) * 4)
            with torch.cuda.device(dev):
                _native.check(L.uoc_filter_labels_depth(_native.ptr(labels[1:]), z1, 3 * H * W, B - 1, H, W, float(thr),
                                                        _native.ptr(ws), ws.numel(), _native.stream_ptr(dev)),
                              "uoc_filter_labels_depth")
        if self.network_crop is not None:
            self.host = _mirror(dev)
            self.host.table.copy_(self.table, non_blocking=True)
            self.host.table_ready.record(torch.cuda.current_stream(dev))

    def _set_refined(self, refined):
        dev, H, W, B = self.dev, self.H, self.W, self.B
        out = refined.view(1, H, W)
        if B > 1:    # match_label_crop returns zeros_like(initial_masks) with only item 0 painted (:153,:176-177)
            full = torch.zeros((B, H, W), dtype=refined.dtype, device=dev)
            full[0] = out[0]
            out = full
        self.refined = out

    def stage2(self):
        if self.network_crop is None:
            return
        dev, H, W = self.dev, self.H, self.W
        self.host.table_ready.synchronize()
        self.table_host = _native.RoiTable.from_buffer_copy(self.host.table.numpy().tobytes())
        K = self.K = int(self.table_host.K)
        if K == 0:
            return
        rgb_crop, mask_crop, depth_crop = _crop(self.image, self.depth, self.labels[0], self.table, K, H, W, dev)
        features_crop = _detach_keep_planes(self.network_crop(rgb_crop, mask_crop, depth_crop))     # :259
        self.labels_crop, _ = _cluster_device(features_crop, rng=self.rng)      # K fields, one launch set
        self.has_depth = depth_crop is not None
        self.mask_crop, self.depth_crop = mask_crop, depth_crop       # kept for the host-order fallback
        if not self.host_order:
            refined, _ = _match_device(self.labels_crop, mask_crop, depth_crop, self.table, K, H, W, dev)
            self._set_refined(refined)
            return
        stats = _match_stats(self.labels_crop, mask_crop, depth_crop, K, dev)
        self.host.stats[:stats.numel()].copy_(stats, non_blocking=True)
        self.host.stats_ready.record(torch.cuda.current_stream(dev))

    def stage3(self):
        if self.network_crop is None or self.K == 0 or not self.host_order:
            return
        dev, H, W, K = self.dev, self.H, self.W, self.K
        self.host.stats_ready.synchronize()
        refined = _paste(self.labels_crop, self.table, self.table_host, self.host.stats, self.has_depth, K, H, W, dev,
                         plan_host=self.host.plan)
        self._set_refined(refined)

    def redo_with_host_order(self):
        """The device ordering flagged this frame (NaN keys, >= 64 ROIs): order on the host from the same crop labels."""
        refined, _ = _match_host_order(self.labels_crop, self.mask_crop, self.depth_crop, self.table, self.K, self.H, self.W, self.dev)
        self._set_refined(refined)

    def result_device(self):
        """(labels [B,H,W] int32, refined [B,H,W] int32 or None), on the device."""
        return self.labels.view(self.B, self.H, self.W), self.refined


class FrameGroupJob:
    """N independent single-image frames through the two-stage path as ONE set of launches per stage (the frame-parallel
    runner's unit of work): both stage-1 embeddings in one network forward (batch N), both clusterings in one launch
    set, and the crops of all N frames in one stage-2 forward (batch K_1 + ... + K_N).  Twice the pixels / Winograd
    tiles per launch is what the convolution kernels need to fill 256 CUs with full-height tiles.  Every
    frame keeps its own ROI table, RNG and output; label maps are bit-identical to one-frame-at-a-time processing (the
    kernels' per-output summation order does not depend on the batch).  Same stages as FrameJob."""

    def __init__(self, samples, network, network_crop, depth_threshold, rngs, host_order=None):
        self.samples, self.network, self.network_crop = list(samples), network, network_crop
        self.depth_threshold, self.rngs = depth_threshold, list(rngs)
        self.host_order = FORCE_HOST_ORDER if host_order is None else host_order
        self.N = len(self.samples)
        self.K = [0] * self.N
        self.refined = [None] * self.N

    def stage1(self):
        require_supported()
        dev = self.dev = _device()
        N = self.N
        # non_blocking is a no-op for pageable host memory (the runtime stages it synchronously) and asynchronous for
        # pinned memory; asking the tensor (`is_pinned()`) costs a driver query per call, so just always pass it
        pin = lambda t: t.to(dev, non_blocking=True)
        if all("image_u8" in sm for sm in self.samples) and len({tuple(sm["depth_u16"].shape) for sm in self.samples}) == 1:
            # raw samples (uint8 BGR + uint16 depth): asynchronous uploads, prepared on the device straight into the launch
            # set's batched inputs
            from ..io import prepare_on_device
            Hs, Ws = self.samples[0]["depth_u16"].shape
            image = torch.empty((N, 3, Hs, Ws), dtype=torch.float32, device=dev)
            xyz = torch.empty((N, 3, Hs, Ws), dtype=torch.float32, device=dev)
            for f, sm in enumerate(self.samples):
                prepare_on_device(sm, dev, out=(image[f], xyz[f]))
            self.image = image
            self.depth = depth = xyz if uses_depth() else None
        else:
            images, depths = [], []
            for sm in self.samples:
                if "image_u8" in sm:
                    from ..io import prepare_on_device
                    sm = dict(sm, **prepare_on_device(sm, dev))
                images.append(pin(sm["image_color"]).float())
                if uses_depth():
                    depths.append(pin(sm["depth"]).float())
            self.image = image = (torch.cat(images) if N > 1 else images[0]).contiguous()
            self.depth = depth = ((torch.cat(depths) if N > 1 else depths[0]).contiguous()) if depths else None
        thr = self.depth_threshold if depth is not None else None
        assert image.shape[0] == N, "FrameGroupJob takes single-image samples"
        _, _, H, W = image.shape
        self.H, self.W = H, W
        features = _detach_keep_planes(self.network(image, None, depth))
        firsts = [self.rngs[f].randint(0, H * W) for f in range(N)]          # mean_shift.py:155, one draw per frame
        self.labels = labels = _cluster_fields(features, firsts)             # [N, H*W] int32
        self.tables = []
        for f in range(N):
            zptr = ctypes.c_void_p(depth.data_ptr() + ((3 * f + 2) * H * W) * 4) if thr is not None else ctypes.c_void_p(0)
            self.tables.append(_build_rois(labels[f], zptr, H, W, dev, thr if thr is not None else 0.0))
        if self.network_crop is not None:
            self.host = _mirror(dev, N)
            for f in range(N):
                self.host.tables[f].copy_(self.tables[f], non_blocking=True)
            self.host.table_ready.record(torch.cuda.current_stream(dev))

    def stage2(self):
        if self.network_crop is None:
            return
        dev, H, W, N = self.dev, self.H, self.W, self.N
        S = cfg.TRAIN.SYN_CROP_SIZE
        self.host.table_ready.synchronize()
        self.table_host = [_native.RoiTable.from_buffer_copy(self.host.tables[f].numpy().tobytes()) for f in range(N)]
        self.K = [int(t.K) for t in self.table_host]
        Kt = sum(self.K)
        if Kt == 0:
            return
        self.off = np.concatenate([[0], np.cumsum(self.K)]).astype(int)
        rgb = torch.empty((Kt, 3, S, S), dtype=torch.float32, device=dev)
        dep = torch.empty((Kt, 3, S, S), dtype=torch.float32, device=dev) if self.depth is not None else None
        mask = torch.empty((Kt, S, S), dtype=torch.float32, device=dev)
        L = _native.lib()
        for f in range(N):
            if self.K[f] == 0:
                continue
            a, b = self.off[f], self.off[f + 1]
            with torch.cuda.device(dev):
                rc = L.uoc_roi_crop(_native.ptr(self.image[f]), _native.ptr(self.depth[f]) if dep is not None else None,
                                    _native.ptr(self.labels[f]), H, W, _native.ptr(self.tables[f]), self.K[f], S,
                                    _native.ptr(rgb[a:b]), _native.ptr(dep[a:b]) if dep is not None else None,
                                    _native.ptr(mask[a:b]), _native.stream_ptr(dev))
            _native.check(rc, "uoc_roi_crop")
        features_crop = _detach_keep_planes(self.network_crop(rgb, mask, dep))
        firsts = [self.rngs[f].randint(0, S * S) for f in range(N) for _ in range(self.K[f])]   # K_f draws per frame, in order
        self.labels_crop = _cluster_fields(features_crop, firsts)              # [Kt, S*S]
        self.has_depth = dep is not None
        for f in range(N):
            if self.K[f] == 0:
                continue
            a, b = self.off[f], self.off[f + 1]
            if not self.host_order:
                refined, _ = _match_device(self.labels_crop[a:b], mask[a:b], dep[a:b] if dep is not None else None,
                                           self.tables[f], self.K[f], H, W, dev)
                self.refined[f] = refined.view(H, W)
                continue
            stats = _match_stats(self.labels_crop[a:b], mask[a:b], dep[a:b] if dep is not None else None, self.K[f], dev)
            self.host.stats_all[f, :stats.numel()].copy_(stats, non_blocking=True)
        # host ordering: the statistics are on their way to the host.  Device ordering: the same event marks "stage 2 has
        # run" — the runner frees the slot only then (see pending_event)
        self.host.stats_ready.record(torch.cuda.current_stream(dev))
        if not self.host_order:
            self.labels_crop = self.image = self.depth = None

    def stage3(self):
        if self.network_crop is None or sum(self.K) == 0 or not self.host_order:
            return
        dev, H, W = self.dev, self.H, self.W
        self.host.stats_ready.synchronize()
        for f in range(self.N):
            if self.K[f] == 0:
                continue
            a, b = self.off[f], self.off[f + 1]
            refined = _paste(self.labels_crop[a:b], self.tables[f], self.table_host[f], self.host.stats_all[f], self.has_depth,
                             self.K[f], H, W, dev, plan_host=self.host.plans[f])
            self.refined[f] = refined.view(H, W)
        self.labels_crop = self.image = self.depth = None

    def pending_event(self, state):
        """The event the next stage waits for (state 1: the ROI tables have landed on the host; state 2: stage 2 has run —
        with host ordering that is the arrival of the match statistics), or None if there is nothing to wait for.
        State 2 is a wait although the device ordering needs nothing from the device any more: a stream that is handed its
        next job's stage 1 while this job's stage 2 is still queued puts that job's sampling kernel into the per-device event
        chain (csrc/meanshift.hip) AHEAD of the other streams' stage-2 sampling kernels, which then wait behind this
        stream's whole stage 2 — measured 168 -> 161 frames/s (same box, round 6)."""
        if self.network_crop is None:
            return None
        if state == 1:
            return self.host.table_ready
        return self.host.stats_ready if sum(self.K) > 0 else None

    def final_maps(self):
        """Per frame: the refined map if stage 2 produced one, else the stage-1 map ([H,W] int32, device)."""
        return [self.refined[f] if self.refined[f] is not None else self.labels[f].view(self.H, self.W) for f in range(self.N)]


def _cluster_fields(features, firsts):
    """Clusters the B fields of `features` [B,C,h,w] with the given first-seed indices -> int32 labels [B, h*w]."""
    planes = getattr(features, "_uoc_planes", None)
    if planes is not None and planes.shape[0] == features.shape[0]:
        X = planes
    else:
        X = _pixel_major(features.float())
        if X.shape[-1] == 128:
            X = to_planes(X)
    labels, _ = cluster_batch(X, firsts, KAPPA, 100, MAX_ITERS, 2 * cfg.TRAIN.EMBEDDING_ALPHA)
    return labels


def _run_frame_graphed(sample, network, network_crop, depth_threshold, checked):
    """The frame as hipGraph replays (fcn/graph_replay.py) when the call qualifies — cfg.TEST.GRAPH_REPLAY, a single-image
    sample, SEGNET networks, and not the first call of this (networks, size, configuration) combination — else None and
    the caller runs the eager FrameJob.  Same label maps either way (tests/test_graph_replay_gpu.py)."""
    from . import graph_replay as GR
    if FORCE_HOST_ORDER or not getattr(cfg.TEST, "GRAPH_REPLAY", False) or "label" in sample:
        return None
    from ..networks.SEG import SEGNET
    if not isinstance(network, SEGNET) or not (network_crop is None or isinstance(network_crop, SEGNET)):
        return None
    require_supported()
    dev = _device()
    if "image_u8" in sample:
        from ..io import prepare_on_device
        sample = dict(sample, **prepare_on_device(sample, dev))
    image = sample["image_color"]
    if image.dim() != 4 or image.shape[0] != 1:
        return None
    _, _, H, W = image.shape
    has_depth = uses_depth()
    thr = depth_threshold if has_depth else None
    gf = GR.frame_for(network, network_crop, H, W, dev, thr)
    if gf is None:
        return None
    image = image.to(dev, non_blocking=True).float()
    depth = sample["depth"].to(dev, non_blocking=True).float() if has_depth else None
    labels, refined, K, redo = gf.run(image, depth, None)
    LAST_FRAME_STATS["rois"] = K
    if checked:
        _check_clustering(dev)              # synchronises
        if K > 0 and gf.order_flagged():
            refined = redo()
    return labels, refined


def _run_frame(sample, network, network_crop, depth_threshold, return_device=False, checked=None):
    """Per-frame body shared by test_sample (:247-261) and test_segnet (:288-321).
    depth_threshold None = no depth-coverage filter.  return_device=True keeps the int32 label
    maps on the device ([B,H,W], [1,H,W] or None).  checked (default: not return_device): synchronise and check the
    clustering status + the ordering flag before returning; the frame-parallel runner passes False and checks once per
    block (_finish_block)."""
    if checked is None:
        checked = not return_device
    graphed = _run_frame_graphed(sample, network, network_crop, depth_threshold, checked)
    if graphed is not None:
        labels, out_label_refined = graphed
    else:
        job = FrameJob(sample, network, network_crop, depth_threshold)
        job.stage1()
        job.stage2()
        job.stage3()
        LAST_FRAME_STATS["rois"] = job.K
        if checked:
            _check_clustering(job.dev)          # synchronises
            if job.K > 0 and not job.host_order and _order_needs_host(job.dev):
                job.redo_with_host_order()
        labels, out_label_refined = job.result_device()
    if return_device:
        return labels, out_label_refined
    out_label = labels.float().cpu()
    if out_label_refined is not None:
        out_label_refined = out_label_refined.float().cpu()
    return out_label, out_label_refined


def _print_average(metrics_all):
    """test_dataset.py:346-357 — sum per key over the frames, divide by the frame count, print sorted."""
    result = {}
    num = len(metrics_all)
    for metrics in metrics_all:
        for k in metrics.keys():
            result[k] = result.get(k, 0) + metrics[k]
    for k in sorted(result.keys()):
        result[k] /= num
        print("%s: %f" % (k, result[k]))
    return result


def test_segnet(test_loader, network, output_dir, network_crop):
    """test_dataset.py:271-381: dataset loop around the same per-frame body; per-frame segmentation metrics of
    the stage-1 and the refined label maps against sample['label'] (multilabel_metrics, :308-329), one .mat
    per sample (labels, labels_refined, filename; :337-340) and the averaged report (:346-381).  Samples
    without a 'label' entry are segmented and saved but not scored."""
    import scipy.io
    from ..utils.evaluation import multilabel_metrics
    network.eval()
    if network_crop is not None:
        network_crop.eval()
    epoch_size = len(test_loader)
    results = []
    metrics_all, metrics_all_refined = [], []
    name = str(getattr(getattr(test_loader, "dataset", None), "name", ""))
    threshold = 0.5 if "ocid" in name else (0.8 if "osd" in name else None)      # :299-305
    for i, sample in enumerate(test_loader):
        end = time.time()
        labels_dev, refined_dev = _run_frame(sample, network, network_crop, threshold, return_device=True, checked=True)
        prediction_dev = labels_dev.reshape(labels_dev.shape[-2:]) if labels_dev.shape[0] == 1 else labels_dev[0]
        refined_2d = refined_dev[0] if refined_dev is not None else prediction_dev
        prediction = labels_dev.float().cpu().squeeze().numpy()
        prediction_refined = refined_dev.float().cpu().squeeze().numpy() if refined_dev is not None else prediction.copy()
        result = {"labels": prediction, "labels_refined": prediction_refined, "filename": sample.get("filename", "")}
        if "label" in sample:
            gt = sample["label"].squeeze()
            metrics = multilabel_metrics(prediction_dev, gt)                     # :308-312
            metrics_all.append(metrics)
            print(metrics)
            metrics_refined = multilabel_metrics(refined_2d, gt)                 # :324-329
            metrics_all_refined.append(metrics_refined)
            print(metrics_refined)
            result["metrics"], result["metrics_refined"] = metrics, metrics_refined
        if output_dir is not None:
            filename = os.path.join(output_dir, "%06d.mat" % i)
            print(filename)
            scipy.io.savemat(filename, {k: result[k] for k in ("labels", "labels_refined", "filename")},
                             do_compression=True)
        results.append(result)
        print("[%d/%d], batch time %.2f" % (i, epoch_size, time.time() - end))
    if metrics_all:
        print("========================================================")
        print("%d images" % len(metrics_all))
        print("========================================================")
        result = _print_average(metrics_all)
        for k in ("Objects Precision", "Objects Recall", "Objects F-measure", "Boundary Precision", "Boundary Recall",
                  "Boundary F-measure", "obj_detected_075_percentage"):
            print("%.6f" % (result[k]))
        print("========================================================")
        print(result)
        print("====================Refined=============================")
        print(_print_average(metrics_all_refined))
        print("========================================================")
    return results
