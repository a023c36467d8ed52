"""Round 6: one frame at a time as hipGraph replays (fcn/graph_replay.py) against the eager FrameJob path — the same
kernels in the same order, so both label maps must be torch.equal and the NumPy RNG must be left in the same state
(1 + K draws per frame, lib/utils/mean_shift.py:155 through lib/fcn/test_dataset.py:247-261).  Frames with K = 0 (no
ROI: stage 2 skipped, refined is None), K = 1 and several K; every K is seen twice so that both the first-use path (eager
run + capture) and the replay are compared."""
import numpy as np
import pytest
import torch

from unseenobjectclustering_amd import _native, networks, synth
from unseenobjectclustering_amd.fcn import graph_replay as GR, test_dataset as TD
from unseenobjectclustering_amd.fcn.config import cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nets(device):
    cfg.device = device
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    return (networks.seg_resnet34_8s_embedding(2, 64, sd).eval(), networks.seg_resnet34_8s_embedding(2, 64, sd).eval())


def _frames(device, H, W, specs):
    out = []
    for seed, objects in specs:
        fr = synth.palette_frame(seed, H, W, objects)
        out.append(dict(image_color=torch.from_numpy(fr["image_color"]).to(device), depth=torch.from_numpy(fr["depth"]).to(device)))
    return out


def _run(sample, nets, seed, graph):
    cfg.TEST.GRAPH_REPLAY = graph
    np.random.seed(seed)
    out, refined = TD._run_frame(sample, nets[0], nets[1], TD.DEPTH_FILTER, return_device=True, checked=True)
    after = np.random.randint(0, 1 << 30)
    return out.clone(), (refined.clone() if refined is not None else None), TD.LAST_FRAME_STATS["rois"], after


@pytest.mark.parametrize("hw", [(240, 320), (480, 640)])
def test_graph_replay_equals_eager(device, nets, hw):
    H, W = hw
    specs = [(10001, 5), (10002, 0), (10003, 1), (10004, 7), (10005, 3), (10006, 5)]
    frames = _frames(device, H, W, specs)
    old = cfg.TEST.GRAPH_REPLAY
    GR.reset()
    try:
        eager = [_run(f, nets, 100 + i, False) for i, f in enumerate(frames)]
        Ks = [e[2] for e in eager]
        assert 0 in Ks or min(Ks) <= 1, f"the frame set should contain a frame without ROIs or with one: K = {Ks}"
        for rnd in range(3):        # round 0: first call of the combination is eager by design, then captures; 1-2: replays
            for i, f in enumerate(frames):
                got = _run(f, nets, 100 + i, True)
                want = eager[i]
                assert got[2] == want[2], (rnd, i, got[2], want[2])
                assert got[3] == want[3], f"round {rnd} frame {i}: the RNG was not consumed like the eager path (1 + K draws)"
                assert torch.equal(got[0], want[0]), f"round {rnd} frame {i}: stage-1 map differs from the eager path"
                assert (got[1] is None) == (want[1] is None)
                if got[1] is not None:
                    assert torch.equal(got[1], want[1]), (
                        f"round {rnd} frame {i} (K = {got[2]}): refined map differs in {(got[1] != want[1]).sum().item()} pixels; "
                        f"sampling fallbacks {_native.lib().uoc_ms_fps_fallbacks()}, status {_native.lib().uoc_ms_check(_native.stream_ptr(device))}")
        gfs = [v for v in GR._frames.values() if v]
        assert len(gfs) == 1 and gfs[0].g1 is not None
        assert set(gfs[0].g2) == {k for k in Ks if k > 0}, (sorted(gfs[0].g2), Ks)
        assert gfs[0].calls == 3 * len(frames) - 1
    finally:
        cfg.TEST.GRAPH_REPLAY = old
        GR.reset()


def test_graph_replay_through_test_sample_and_new_weights(device, nets):
    """test_sample (the reference's call surface, CPU tensors in / CPU float maps out) takes the replay path from its second
    call on; loading new weights rebuilds the native copy and must not replay graphs that baked the old pointers."""
    fr = synth.palette_frame(10011, 240, 320, 4)
    sample = dict(image_color=torch.from_numpy(fr["image_color"]), depth=torch.from_numpy(fr["depth"]))
    old = cfg.TEST.GRAPH_REPLAY
    GR.reset()
    try:
        cfg.TEST.GRAPH_REPLAY = False
        np.random.seed(5)
        want = TD.test_sample(sample, nets[0], nets[1])
        cfg.TEST.GRAPH_REPLAY = True
        for _ in range(3):
            np.random.seed(5)
            got = TD.test_sample(sample, nets[0], nets[1])
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        assert sum(1 for v in GR._frames.values() if v) == 1
        sd2 = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
        sd2["fcn.resnet34_8s.fc.bias"] = sd2["fcn.resnet34_8s.fc.bias"] * 1.05
        net2 = networks.seg_resnet34_8s_embedding(2, 64, sd2).eval()
        cfg.TEST.GRAPH_REPLAY = False
        np.random.seed(5)
        want2 = TD.test_sample(sample, net2, nets[1])
        nets[0].load_state_dict(net2.state_dict())          # same module object, new weights -> new native copy
        cfg.TEST.GRAPH_REPLAY = True
        for _ in range(3):
            np.random.seed(5)
            got2 = TD.test_sample(sample, nets[0], nets[1])
            assert torch.equal(got2[0], want2[0]) and (got2[1] is None) == (want2[1] is None)
            assert got2[1] is None or torch.equal(got2[1], want2[1])
        assert not torch.equal(want2[0], want[0]), "the perturbed weights should change the stage-1 map (else this test proves nothing)"
    finally:
        cfg.TEST.GRAPH_REPLAY = old
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
        nets[0].load_state_dict(sd)
        GR.reset()
