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
    for depth, group in ((1, 1), (2, 1), (3, 1), (2, 2), (2, 3), (1, 2)):
        fn = runner.two_stage_frame_fn(samples, net, net_crop, frames_per_launch=group)
        blocks[depth, group] = runner.run_sharded(5, fn, 480, 640, device, 0, 1, False, inflight=depth).cpu()
        torch.cuda.synchronize()
        counts[depth, group] = list(fn.roi_counts)
    ref = blocks[1, 1]
    assert ref.shape == (5, 480, 640) and int(ref.max()) >= 5
    for key, b in blocks.items():
        assert torch.equal(ref, b), f"streams={key[0]} frames_per_launch={key[1]} changed a label map"
        assert counts[key] == counts[1, 1]
    # the on-chip sampling kernel served every field (no silent fallback to the streaming kernel's other summation order)
    assert _native.lib().uoc_ms_fps_fallbacks() == before
    # stage 2 really ran: ROI counts recorded per frame
    assert len(counts[1, 1]) == 5 and min(counts[1, 1]) >= 5
