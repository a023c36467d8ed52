"""Frames in flight (runner._run_block_pipelined): the label-map block must not depend on how many frames share the
GPU — each frame draws its first seeds from its own RandomState and owns its stream's workspaces."""
import numpy as np
import pytest
import torch

from unseenobjectclustering_amd import _native, networks, runner, synth
from unseenobjectclustering_amd.fcn.config import cfg

pytestmark = pytest.mark.gpu


def test_block_is_independent_of_frames_in_flight(device):
    cfg.device = device
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    net_crop = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    samples = []
    for g in range(5):
        s = 10_000 + g
        fr = synth.palette_frame(s, 480, 640, 5 + s % 3)
        samples.append(dict(image_color=torch.from_numpy(fr["image_color"]).to(device),
                            depth=torch.from_numpy(fr["depth"]).to(device)))
    before = _native.lib().uoc_ms_fps_fallbacks()
    blocks, counts = {}, {}
    # (streams, frames per launch): sequential / two streams / three streams, one or two frames per launch set
    for depth, group in ((1, 1), (2, 1), (3, 1), (2, 2), (2, 3), (1, 2), (3, 4)):
        fn = runner.two_stage_frame_fn(samples, net, net_crop, frames_per_launch=group)
        blocks[depth, group] = runner.run_sharded(5, fn, 480, 640, device, 0, 1, False, inflight=depth).cpu()
        torch.cuda.synchronize()
        counts[depth, group] = list(fn.roi_counts)
    # (3, 4) and (2, 3) above ran with the balanced tail (5 frames: (2, 2, 1) and (3, 2)); the same with full sets ((4, 1), (3, 2))
    import os
    os.environ["UOC_PIPE_TAIL"] = "0"
    try:
        fn = runner.two_stage_frame_fn(samples, net, net_crop, frames_per_launch=4)
        blocks[3, 40] = runner.run_sharded(5, fn, 480, 640, device, 0, 1, False, inflight=3).cpu()
        counts[3, 40] = list(fn.roi_counts)
    finally:
        os.environ.pop("UOC_PIPE_TAIL", None)
    ref = blocks[1, 1]
    assert ref.shape == (5, 480, 640) and int(ref.max()) >= 5
    for key, b in blocks.items():
        assert torch.equal(ref, b), f"streams={key[0]} frames_per_launch={key[1]} changed a label map"
        assert counts[key] == counts[1, 1]
    # the on-chip sampling kernel served every field (no silent fallback to the streaming kernel's other summation order)
    assert _native.lib().uoc_ms_fps_fallbacks() == before
    # stage 2 really ran: ROI counts recorded per frame
    assert len(counts[1, 1]) == 5 and min(counts[1, 1]) >= 5


def test_launch_sets_with_frames_that_have_no_roi(device):
    """Frames whose objects are all rejected by the depth filter (no ROI, stage 2 skipped, test_dataset.py:251-254)
    mixed into launch sets with ordinary frames, and a launch set made only of such frames: same label maps and ROI
    counts as one frame at a time."""
    cfg.device = device
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    net_crop = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    samples = []
    for g, blind in enumerate((False, True, False, True, True, True, False)):
        fr = synth.palette_frame(10_000 + g, 480, 640, 5 + g % 3)
        depth = np.zeros_like(fr["depth"]) if blind else fr["depth"]      # z = 0 everywhere: every label fails the filter
        samples.append(dict(image_color=torch.from_numpy(fr["image_color"]).to(device), depth=torch.from_numpy(depth).to(device)))
    blocks, counts = {}, {}
    for depth, group in ((1, 1), (2, 3), (1, 4), (3, 2)):
        fn = runner.two_stage_frame_fn(samples, net, net_crop, frames_per_launch=group)
        blocks[depth, group] = runner.run_sharded(len(samples), fn, 480, 640, device, 0, 1, False, inflight=depth).cpu()
        torch.cuda.synchronize()
        counts[depth, group] = list(fn.roi_counts)
    ref = blocks[1, 1]
    assert counts[1, 1][1] == counts[1, 1][3] == counts[1, 1][4] == counts[1, 1][5] == 0 and min(counts[1, 1][0::2][:2]) >= 5
    for blind in (1, 3, 4, 5):
        assert int(ref[blind].max()) == 0          # everything filtered: the stage-1 map that is returned is background
    for key, b in blocks.items():
        assert torch.equal(ref, b), f"streams={key[0]} frames_per_launch={key[1]} changed a label map"
        assert counts[key] == counts[1, 1]


def test_rank_views_concatenate_to_the_single_rank_block(device):
    """Sharding independence with the REAL frame function (BASELINE configs[4]): the blocks that
    run_sharded computes as rank r of a world of 2 / 3 / 4 over the same 8 frames, concatenated, equal the block of a
    single rank — every frame seeds its RNG from its GLOBAL index and no kernel's summation order depends on which
    frames share a launch set (a rank's launch sets start at its block start, so their composition differs per view:
    world 3 gives blocks of 3 / 3 / 2 frames)."""
    cfg.device = device
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    net_crop = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    F = 8
    samples = []
    for g in range(F):
        s = 10_000 + g
        fr = synth.palette_frame(s, 480, 640, 5 + s % 3)
        samples.append(dict(image_color=torch.from_numpy(fr["image_color"]).to(device),
                            depth=torch.from_numpy(fr["depth"]).to(device)))
    fn = runner.two_stage_frame_fn(samples, net, net_crop, first_index=0, frames_per_launch=4)
    whole = runner.run_sharded(F, fn, 480, 640, device, 0, 1, False, inflight=3).cpu()
    assert whole.shape == (F, 480, 640)
    for world in (2, 3, 4):
        parts = []
        for rank in range(world):
            lo, hi = runner.shard_range(F, rank, world)
            # a rank holds only its own frames, like bench.py: samples[lo:hi] with first_index = lo
            fr_fn = runner.two_stage_frame_fn(samples[lo:hi], net, net_crop, first_index=lo, frames_per_launch=4)
            parts.append(runner.run_sharded(F, fr_fn, 480, 640, device, rank, world, False, inflight=3).cpu())
        got = torch.cat(parts)
        assert got.shape == whole.shape
        assert torch.equal(got, whole), f"world={world}: the concatenated rank blocks differ from the single-rank block"


def test_host_order_and_the_rerun_when_the_device_ordering_flags_a_block(device):
    """Round 6: (a) the ROI paint order on the host (rounds 1-5, FORCE_HOST_ORDER) and on the device (uoc_roi_match) give the same
    block; (b) when a block ends with the ordering flag up — NaN sort keys with >= 64 ROIs, the one case the kernel leaves to
    Python's own sort — run_sharded runs it again with the host ordering, same seeds, and does not double-count the ROIs.  The flag
    is raised artificially here (the kernel's own flagging is covered by test_device_roi_order_is_pythons_sorted)."""
    from unseenobjectclustering_amd.fcn import test_dataset as TD
    cfg.device = device
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    net_crop = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    samples = []
    for g in range(6):
        fr = synth.palette_frame(10_100 + g, 240, 320, 4 + g % 3)
        samples.append(dict(image_color=torch.from_numpy(fr["image_color"]).to(device), depth=torch.from_numpy(fr["depth"]).to(device)))
    fn = runner.two_stage_frame_fn(samples, net, net_crop, frames_per_launch=2)
    want = runner.run_sharded(6, fn, 240, 320, device, 0, 1, False, inflight=3).cpu()
    counts = list(fn.roi_counts)
    assert len(counts) == 6 and max(counts) >= 1
    was = TD.FORCE_HOST_ORDER
    try:
        TD.FORCE_HOST_ORDER = True
        fn2 = runner.two_stage_frame_fn(samples, net, net_crop, frames_per_launch=2)
        assert torch.equal(runner.run_sharded(6, fn2, 240, 320, device, 0, 1, False, inflight=3).cpu(), want)
        assert torch.equal(runner.run_sharded(6, fn2, 240, 320, device, 0, 1, False, inflight=1).cpu(), want)
    finally:
        TD.FORCE_HOST_ORDER = was
    # (b) one artificial flag: the first pass of the block "fails" with HostOrderNeeded, the second runs with the host ordering
    fn3 = runner.two_stage_frame_fn(samples, net, net_crop, frames_per_launch=2)
    seen = []
    real_finish = fn3.finish

    def finish(dev):
        seen.append(TD.FORCE_HOST_ORDER)
        real_finish(dev)
        if len(seen) == 1:
            raise TD.HostOrderNeeded("injected")
    fn3.finish = finish
    got = runner.run_sharded(6, fn3, 240, 320, device, 0, 1, False, inflight=3).cpu()
    assert seen == [False, True] and TD.FORCE_HOST_ORDER is was
    assert torch.equal(got, want) and list(fn3.roi_counts) == counts
