"""Frame-parallel runner: frames are independent units, so they shard across ranks (one process
per GPU, torch.distributed; backend 'nccl' = RCCL over xGMI on the GPU node, 'gloo' in CPU tests)
with no communication during compute and ONE all_gather of the fixed-size per-rank label-map
block at the end.  The reference has no counterpart (inference pins one device,
tools/test_images.py:199).

Sharding-independent results: the NumPy RNG that picks the first mean-shift seed
(mean_shift.py:155) is re-seeded per frame from the GLOBAL frame index.
"""
from __future__ import annotations

import os
from collections import deque
from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist

from .fcn.config import cfg


def shard_range(num_frames: int, rank: int, world: int):
    """Contiguous block [lo, hi) of ceil(F/G) frames for `rank` (last blocks may be short/empty)."""
    per = (num_frames + world - 1) // world
    lo = min(rank * per, num_frames)
    return lo, min(lo + per, num_frames)


def config_fingerprint() -> int:
    """63 bits of uoc_config_fingerprint() (what could make two ranks compute different bits).  Tests of the mismatch path
    replace this function (monkeypatch); nothing in the environment overrides it."""
    from . import _native
    return _native.config_fingerprint() & 0x7FFFFFFFFFFFFFFF


def frame_rng_seed(frame_index: int) -> int:
    return int(cfg.RNG_SEED) + int(frame_index)


def run_sharded(num_frames: int, frame_fn: Callable[[int], torch.Tensor], height: int, width: int,
                device: torch.device, rank: int = 0, world: int = 1, gather: bool = True,
                force_collective: bool = False, inflight: Optional[int] = None,
                timing: Optional[dict] = None, block_range: Optional[tuple] = None) -> Optional[torch.Tensor]:
    """Runs frame_fn(global_index) -> [H,W] integer label map (on `device`) for this rank's block
    and all-gathers the uint8 blocks.  Returns [num_frames, H, W] uint8 on `device` (every rank),
    or only the local block when gather=False / world == 1.

    Error handling: a rank whose block failed (an exception in frame_fn, the clustering status check, or a
    label id that does not fit uint8) must not leave the others waiting in the collective, so every rank
    first all-reduces an error flag and ALL ranks raise when any rank failed.

    `inflight` (default: $UOC_FRAMES_IN_FLIGHT or 3): when frame_fn offers `make_job` (two_stage_frame_fn does), that
    many frames are kept in flight on this GPU, each on its own stream (_run_block_pipelined); 1 = one frame at a
    time on the current stream.  The label maps are the same either way.

    `block_range` = (lo, hi): run exactly the global frames [lo, hi) on this process (world must be 1, no gather) — what a
    rank of a larger job does for its block, without emulating a rank / world pair (tests that walk a long frame list in
    resident chunks).

    `timing`: if given, receives this rank's 'compute_s' (its frame block, device-synchronised), 'host_cpu_s' (CPU seconds
    the process spent on it: the event-driven host loop polls, so about one core per rank) and 'gather_s' (error-flag
    all-reduce + all_gather) — the per-rank breakdown bench.py prints for multi-GPU runs."""
    import time
    t_start, t_cpu = time.perf_counter(), time.process_time()
    per = (num_frames + world - 1) // world
    lo, hi = shard_range(num_frames, rank, world) if block_range is None else block_range
    if block_range is not None:
        assert world == 1 and not gather and 0 <= lo <= hi, "block_range: an explicit global frame block, single rank"
        per = hi - lo
    block = torch.zeros((per, height, width), dtype=torch.uint8, device=device)
    collective = gather and (world > 1 or force_collective)
    error = None
    if inflight is None:
        inflight = int(os.environ.get("UOC_FRAMES_IN_FLIGHT", "3"))
    def run_block():
        if (inflight > 1 or getattr(frame_fn, "frames_per_launch", 1) > 1) and device.type == "cuda" and hasattr(frame_fn, "make_job"):
            top = _run_block_pipelined(frame_fn, lo, hi, block, device, inflight)
        else:
            top = torch.zeros((), dtype=torch.int64, device=device)
            for i in range(lo, hi):
                np.random.seed(frame_rng_seed(i))
                m = frame_fn(i)
                top = torch.maximum(top, m.max().to(torch.int64))
                block[i - lo] = m.to(torch.uint8)
        if hasattr(frame_fn, "finish") and device.type == "cuda":
            frame_fn.finish(device)
        if hi > lo and int(top) > 255:       # one sync per block, after the last frame
            raise ValueError(f"label id {int(top)} does not fit the uint8 label-map block")

    try:
        from .fcn import test_dataset as TD
        counted = len(getattr(frame_fn, "roi_counts", ()))
        try:
            run_block()
        except TD.HostOrderNeeded:
            # a frame had NaN ROI sort keys with >= 64 ROIs (the one ordering case uoc_roi_match leaves to Python's own
            # sorted): the block runs again with the ordering on the host, same seeds -> same draws
            del getattr(frame_fn, "roi_counts", [])[counted:]
            was, TD.FORCE_HOST_ORDER = TD.FORCE_HOST_ORDER, True
            try:
                run_block()
            finally:
                TD.FORCE_HOST_ORDER = was
    except Exception as e:       # noqa: BLE001 - re-raised below, on every rank
        if not collective:
            raise
        error = e
    if timing is not None:
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        timing["compute_s"] = time.perf_counter() - t_start
        timing["host_cpu_s"] = time.process_time() - t_cpu     # CPU seconds of this process (all threads) over its block: the rank's core budget
        timing["frames"] = hi - lo
    if not collective:
        return block[:hi - lo]
    t_gather = time.perf_counter()
    # gloo has no device collectives for all_gather: a CUDA block is staged through the host (tests that run several ranks
    # on one GPU; on the GPU node the backend is nccl = RCCL and everything stays on the device)
    staged = device.type == "cuda" and dist.get_backend() == "gloo"
    coll_dev = torch.device("cpu") if staged else device
    # one MAX all-reduce carries the error flag AND the configuration fingerprint (library version / development build and
    # its rounding-affecting knobs) as (fp, -fp): max(fp) != -max(-fp) means two ranks would not compute the same bits for
    # the same frame — sharding independence broken silently — so every rank fails before the gather
    # (a rank whose library is missing or too old to HAVE a fingerprint must still join the all-reduce: it reports an error
    # and fingerprint 0 instead of raising in front of a collective the other ranks are already waiting in)
    try:
        fp = config_fingerprint()
    except Exception as e:       # noqa: BLE001
        fp = 0
        if error is None:
            error = e
    flag = torch.tensor([1 if error is not None else 0, fp, -fp], dtype=torch.int64, device=coll_dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    flag = flag.tolist()
    if flag[0] != 0:
        if error is not None:
            raise error
        raise RuntimeError("another rank failed in its frame block; aborting before the all_gather")
    if flag[1] != -flag[2]:
        raise RuntimeError("ranks run different libuoc_hip configurations (uoc_config_fingerprint differs: library version, "
                           f"development build or a rounding-affecting knob); this rank: {fp}")
    full = torch.empty((world * per, height, width), dtype=torch.uint8, device=coll_dev)
    dist.all_gather_into_tensor(full, block.to(coll_dev))
    full = full.to(device)
    if timing is not None:
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        timing["gather_s"] = time.perf_counter() - t_gather
    return full[:num_frames]


def launch_set_sizes(n: int, group: int, depth: int, balanced_tail: bool = True):
    """Launch sets of a block of n frames on `depth` streams: full sets of `group` frames; when the frames left over after the
    full rounds (depth sets each) would occupy only SOME of the streams with full sets — 20 frames on 3 x 4: 12 + (4, 4) — they
    are spread over all streams in smaller sets instead ((3, 3, 2)), so that the streams finish together.  Launch-set
    composition never changes a result (tests/test_pipeline_gpu.py).  Same-box A/B, ten alternating runs each (round 5): the
    driver's 20-frame form 161.9 -> 164.1 frames/s (ranges do not overlap), 32 frames 165.8 -> 166.3; blocks that are whole
    rounds, or whose remainder is at most one set, are untouched.  Evening out EVERY set of a short block ((4, 4, 3, 3, 3, 3) for
    20 frames) measured the same.  UOC_PIPE_TAIL=0 (speed-only knob) keeps full sets."""
    sizes, rem = [], n % (group * depth) if depth > 1 else 0
    if not (balanced_tail and depth > 1 and group < rem < group * depth):
        rem = 0
    left = n - rem
    while left > 0:
        sizes.append(min(group, left))
        left -= sizes[-1]
    if rem:
        base, extra = divmod(rem, depth)
        sizes += [base + (1 if k < extra else 0) for k in range(depth)]
    return [s for s in sizes if s > 0]


def _run_block_pipelined(frame_fn, lo: int, hi: int, block: torch.Tensor, device: torch.device, depth: int):
    """Frames lo..hi-1 of this rank on `depth` streams, each stream working on one job (fcn.test_dataset.FrameGroupJob:
    `frames_per_launch` frames batched into one set of launches per stage).

    A job has ONE point where the host needs a few bytes from the device (the ROI counts that size the stage-2 launches;
    until round 5 also the crop keep tables for the ROI order, which uoc_roi_match now computes on the device).
    With one frame at a time the matrix pipes idle through those reads and through the latency-bound kernels around
    them (farthest-point sampling 2 x 0.5 ms, seed components, glue).  Here every job runs on its own stream and the
    host, a single thread, is event-driven: it queues the next stage of whichever job's read has landed and never waits
    on one job while another could be fed, so the GPU always has other jobs' kernels to fill the gaps.  Per-stream
    workspaces; per-frame RandomState, so the label maps do not depend on the interleaving.  Returns the largest label
    id seen (device scalar).
    (Spreading the last, partial round of a block over the streams in smaller launch sets was measured on 20- and
    60-frame blocks: 121.6 vs 122.0 and 123.4 vs 123.9 frames/s — no gain, not kept.)"""
    from . import _native
    _native.lib().uoc_ms_set_stream_ordering(1 if depth > 1 else 0)   # persistent sampling grids: one at a time per device
    main = torch.cuda.current_stream(device)
    streams = _slot_streams(device, depth)
    tops = torch.zeros(len(streams), dtype=torch.int32, device=device)   # largest label id per stream (uoc_labels_to_u8)
    for st in streams:
        st.wait_stream(main)                      # inputs / weights (and `tops`) were produced on the caller's stream
    L = _native.lib()
    nxt = lo
    group = max(1, int(getattr(frame_fn, "frames_per_launch", 1)))
    poll = os.environ.get("UOC_PIPE_POLL", "1") != "0"
    slots = [None] * depth          # per stream: None or [idx, job, state]; state 1 = waits for the tables, 2 = for the statistics
    done_counts = {}

    plan = deque(launch_set_sizes(hi - lo, group, depth, os.environ.get("UOC_PIPE_TAIL", PIPE_TAIL_DEFAULT) != "0"))

    def issue_stage1(slot, i):
        n = min(plan.popleft() if plan else group, hi - i)
        if hasattr(frame_fn, "group_size"):
            m = max(1, min(n, frame_fn.group_size(i, n)))   # only frames of one size share a launch set
            if m < n:
                plan.appendleft(n - m)
            n = m
        idx = list(range(i, i + n))
        with torch.cuda.stream(streams[slot]):
            job = frame_fn.make_job(idx)
            job.stage1()
        slots[slot] = [idx, job, 1]
        return n

    def advance(slot, block_host):
        """Moves the slot's job one stage on if its pending device->host read has completed (or, with block_host, waits
        for it).  Returns True if something was issued."""
        idx, job, state = slots[slot]
        ev = job.pending_event(state)
        if ev is not None:
            if block_host:
                ev.synchronize()
            elif not ev.query():
                return False
        with torch.cuda.stream(streams[slot]):
            if state == 1:
                job.stage2()
                slots[slot][2] = 2
                if job.pending_event(2) is not None:      # the slot is freed when stage 2 has run (FrameGroupJob.pending_event)
                    return True
            job.stage3()
            for i, m in zip(idx, job.final_maps()):       # int32 [H, W] (contiguous) -> the uint8 row of the block
                with torch.cuda.device(device):
                    _native.check(L.uoc_labels_to_u8(_native.ptr(m), m.numel(), _native.ptr(block[i - lo]),
                                                     _native.ptr(tops[slot:slot + 1]), _native.stream_ptr(device)),
                                  "uoc_labels_to_u8")
        done_counts[idx[0]] = list(job.K)
        slots[slot] = None
        return True

    # Event-driven: the host never waits on one job while another stream has a stage that could be queued.  Each job
    # is advanced as soon as its small device->host read has landed (Event.query, non-blocking); only when nothing is
    # ready does the host wait — on the oldest job.  With UOC_PIPE_POLL=0 the host walks the jobs strictly in order.
    order = deque()
    while nxt < hi or any(s is not None for s in slots):
        progressed = False
        for slot in range(depth):
            if slots[slot] is None and nxt < hi:
                nxt += issue_stage1(slot, nxt)
                order.append(slot)
                progressed = True
        for slot in list(order):
            if slots[slot] is not None and poll and advance(slot, False):
                progressed = True
                if slots[slot] is None:
                    order.remove(slot)
        if not progressed and order:
            slot = order[0]
            advance(slot, True)
            if slots[slot] is None:
                order.popleft()
    for k in sorted(done_counts):
        frame_fn.roi_counts.extend(done_counts[k])
    for st in streams:
        main.wait_stream(st)
    return tops.max().to(torch.int64)


_streams = {}
PIPE_TAIL_DEFAULT = "1"     # balanced tail of a frame block: see launch_set_sizes


def _slot_streams(device, depth):
    key = (device.type, device.index)
    pool = _streams.setdefault(key, [])
    while len(pool) < depth:
        pool.append(torch.cuda.Stream(device))
    return pool[:depth]


def two_stage_frame_fn(samples, network, network_crop, first_index: int = 0, frames_per_launch: Optional[int] = None):
    """frame_fn over pre-uploaded samples: final label map = refined map if stage 2 produced one,
    else the stage-1 map (what test_segnet stores as labels_refined, test_dataset.py:324-327).
    Global frame i reads samples[(i - first_index) % len(samples)] (a rank passes the start of its block).
    frames_per_launch (default $UOC_FRAMES_PER_LAUNCH or 4): frames the pipelined runner batches into one launch set
    (fcn.test_dataset.FrameGroupJob)."""
    from .fcn.test_dataset import _run_frame, _finish_block, DEPTH_FILTER, LAST_FRAME_STATS, FrameGroupJob

    def fn(i: int) -> torch.Tensor:
        out, refined = _run_frame(samples[(i - first_index) % len(samples)], network, network_crop, DEPTH_FILTER, return_device=True)
        fn.roi_counts.append(LAST_FRAME_STATS["rois"])
        return (refined if refined is not None else out)[0]

    def make_job(indices):
        """The frames `indices` (global) as one FrameGroupJob, each with its own RNG seeded from its global index."""
        return FrameGroupJob([samples[(i - first_index) % len(samples)] for i in indices], network, network_crop, DEPTH_FILTER,
                             [np.random.RandomState(frame_rng_seed(i)) for i in indices])
    def group_size(i: int, want: int) -> int:
        """How many consecutive frames from global index i have frame i's image size (they are batched into one forward)."""
        shape = lambda k: tuple(samples[(k - first_index) % len(samples)].get("image_color", samples[(k - first_index) % len(samples)].get("image_u8")).shape)
        n = 1
        while n < want and shape(i + n) == shape(i):
            n += 1
        return n
    fn.make_job = make_job
    fn.group_size = group_size
    fn.frames_per_launch = frames_per_launch if frames_per_launch is not None else int(os.environ.get("UOC_FRAMES_PER_LAUNCH", "4"))
    fn.roi_counts = []          # stage-1 ROIs per processed frame (the bench derives the algorithmic work from it)
    # fn.finish() checks the sticky device flags once per block: the clustering status (uoc_ms_check) and the ordering flag
    # of uoc_roi_match (HostOrderNeeded -> run_sharded re-runs the block with the ROI ordering on the host)
    fn.finish = lambda dev: _finish_block(dev)
    return fn
