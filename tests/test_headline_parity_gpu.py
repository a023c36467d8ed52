"""Parity ON THE HEADLINE WORKLOAD (BASELINE.json configs[3], the frames bench.py times): 640x480 palette frames
through both calibrated networks and the whole two-stage path on the GPU, against the oracle (oracle/glue_oracle.py +
backbone_oracle.py, the torch-CPU restatement pinned to the reference by tests/golden) on the same inputs, RNG seeds
and weights.  Reference path: lib/fcn/test_dataset.py:232-267, lib/utils/mean_shift.py:211-227.

north_star states two bars — integer label maps bit-exact up to a permutation of the ids, embeddings within 1e-3 — and
the first is only well defined GIVEN the embeddings: ten kappa = 20 hill-climbing iterations amplify an fp32
summation-order difference of ~1e-6 in the embeddings to ~1e-4 in sparsely supported seeds, enough to move a pixel whose two
nearest clusters are equally far.  The tests therefore split the claim exactly there:

  (a) embeddings: the HIP networks against the oracle's, on the stage-1 frames AND on the oracle's own crops  <= 1e-3
  (b) integer path: the ORACLE's embeddings (stage 1 and every crop) fed through the HIP clustering + ROI / match / paste
      kernels must reproduce the oracle's label maps bit-exactly — on every bench frame tested, no tolerance
  (c) end to end (HIP embeddings -> HIP integer path): a measured mismatch histogram over the frames of
      tests/golden/bench_oracle/ (default: the first 256 of its 1 024), asserted against the measured bounds

The oracle's label maps come from tests/golden/bench_oracle/*.npz (tests/golden/make_bench_oracle.py: oracle runs in the
build container, 1024 frames); the oracle's two network passes per frame run here on the host cores.  Reports go to
gpurun_out/ (copied to profiles/ by hand)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import backbone_oracle as BO, glue_oracle as GO, mean_shift_oracle as O
from unseenobjectclustering_amd import networks, runner, synth
from unseenobjectclustering_amd.fcn import test_dataset as TD
from unseenobjectclustering_amd.fcn.config import cfg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 480, 640
DECOMPOSED_FRAMES = int(os.environ.get("UOC_PARITY_FRAMES", "24"))     # frames of tests (a) + (b)
# measured bounds of the end-to-end comparison (profiles/r03_parity_histogram.json, 1 024 frames: 958 identical up to
# permutation, 39 x 1, 12 x 2, 7 x 3, 3 x 4, 3 x 5 pixels, one frame 17 and one 24 pixels of 307 200 — and on those worst
# frames the integer path is bit-exact given the oracle's embeddings, profiles/r03_parity_decomposed_outlier_frames.json)
E2E_MIN_EXACT_FRACTION = 0.90          # share of frames identical up to a permutation of the ids (measured 0.936)
E2E_P99_MISMATCHED_PIXELS = 5          # 99 % of the frames differ by at most this many pixels (measured: 99.8 %)
E2E_MAX_MISMATCHED_PIXELS = 32         # worst frame (measured 24 = 0.008 % of a frame)


def _fixture():
    """frame index -> (stage-1 map after the depth filter, final map), uint8 [480, 640]."""
    out = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "bench_oracle", "frames_*.npz"))):
        z = np.load(path)
        first = int(z["first"])
        for i in range(len(z["final"])):
            out[first + i] = (z["stage1"][i], z["final"][i])
    return out


def _bench_frame(g):
    s = 10_000 + g
    fr = synth.palette_frame(s, H, W, 5 + s % 3)
    return torch.from_numpy(fr["image_color"]), torch.from_numpy(fr["depth"])


def _mismatch(a, b):
    """Pixels on which two partitions disagree under the best one-to-one relabelling (Hungarian on the contingency table)."""
    from scipy.optimize import linear_sum_assignment
    a, b = np.asarray(a).reshape(-1).astype(np.int64), np.asarray(b).reshape(-1).astype(np.int64)
    kb = int(b.max()) + 1
    table = np.bincount(a * kb + b, minlength=(int(a.max()) + 1) * kb).reshape(-1, kb)
    r, c = linear_sum_assignment(-table)
    return int(a.size - table[r, c].sum())


@pytest.fixture(scope="module")
def nets(device):
    cfg.device = device
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    net = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    net_crop = networks.seg_resnet34_8s_embedding(2, 64, sd).eval()
    torch.set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))
    return sd, net, net_crop


def test_embeddings_and_integer_path_separately(device, nets):
    """(a) + (b) on bench frames 0 .. DECOMPOSED_FRAMES-1."""
    sd, net, net_crop = nets
    fix = _fixture()
    frames = [g for g in range(DECOMPOSED_FRAMES) if g in fix]
    if os.environ.get("UOC_PARITY_FRAME_LIST"):        # ad hoc: the decomposition on chosen frames, e.g. the histogram's outliers
        frames = [int(v) for v in os.environ["UOC_PARITY_FRAME_LIST"].split(",")]
    assert len(frames) >= min(DECOMPOSED_FRAMES, 8) or os.environ.get("UOC_PARITY_FRAME_LIST"), "tests/golden/bench_oracle/ is missing"
    report, worst_embed = [], 0.0
    for g in frames:
        img, dep = _bench_frame(g)
        want_out = torch.from_numpy(fix[g][0].astype(np.float32))[None]
        want_final = fix[g][1]
        # the oracle's embeddings: stage 1, and its own crops rebuilt from its stage-1 map (test_dataset.py:62-112)
        f1 = BO.segnet_forward(sd, img, dep)
        rgb_c, mask_c, rois, dep_c = GO.crop_rois(img, want_out.clone(), dep)
        K = rgb_c.shape[0]
        assert K >= 5, "the headline frames must exercise stage 2"
        f2 = BO.segnet_forward(sd, rgb_c, dep_c)
        # (a) HIP embeddings on the same inputs
        e1 = net(img.to(device), None, dep.to(device)).cpu()
        e2 = net_crop(rgb_c.to(device), None, dep_c.to(device)).cpu()
        err1, err2 = float((e1 - f1).abs().max()), float((e2 - f2).abs().max())
        worst_embed = max(worst_embed, err1, err2)
        # (b) the oracle's embeddings through the HIP integer path (clustering, depth filter, ROI table, crops' masks,
        # match statistics, paste).  The stubs ignore their inputs: stage 2 clusters the oracle's crop embeddings.
        stub1 = lambda image, label, depth: f1.to(device)
        stub2 = lambda image, label, depth: f2.to(device)
        np.random.seed(runner.frame_rng_seed(g))
        got_out, got_ref = TD.test_sample(dict(image_color=img, depth=dep), stub1, stub2)
        s1_same = bool(np.array_equal(got_out[0].numpy().astype(np.int64), want_out[0].numpy().astype(np.int64)))
        fin_same = bool(got_ref is not None and np.array_equal(got_ref[0].numpy().astype(np.int64), want_final.astype(np.int64)))
        np.random.seed(runner.frame_rng_seed(g))
        e2e_out, e2e_ref = TD.test_sample(dict(image_color=img, depth=dep), net, net_crop)
        report.append({
            "frame": g, "rois": K, "embed_err_stage1": err1, "embed_err_crops": err2,
            "end_to_end_mismatched_pixels": _mismatch((e2e_ref if e2e_ref is not None else e2e_out)[0].numpy(), want_final),
            "end_to_end_objects": int((e2e_ref if e2e_ref is not None else e2e_out).max()), "oracle_objects": int(want_final.max()),
            "given_oracle_embeddings": {
                "stage1_identical_ids": s1_same, "final_identical_ids": fin_same,
                "stage1_exact_up_to_permutation": bool(O.labels_equal_up_to_permutation(got_out.numpy(), want_out.numpy())),
                "final_exact_up_to_permutation": bool(got_ref is not None and O.labels_equal_up_to_permutation(got_ref[0].numpy(), want_final)),
                "final_mismatched_pixels": _mismatch(got_ref[0].numpy(), want_final) if got_ref is not None else H * W}})
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    summary = {"frames": len(frames), "embed_max_err": worst_embed,
               "exact_given_oracle_embeddings": all(r["given_oracle_embeddings"]["final_exact_up_to_permutation"] and
                                                    r["given_oracle_embeddings"]["stage1_exact_up_to_permutation"] for r in report),
               "per_frame": report}
    json.dump(summary, open(os.path.join(ROOT, "gpurun_out", "parity_decomposed.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "per_frame"}))
    assert worst_embed <= 1e-3, worst_embed                                  # north_star: embeddings within 1e-3 fp32
    for r in report:                                                         # north_star: integer labels bit-exact
        assert r["given_oracle_embeddings"]["stage1_exact_up_to_permutation"], r
        assert r["given_oracle_embeddings"]["final_exact_up_to_permutation"], r


def test_end_to_end_mismatch_histogram(device, nets):
    """(c): HIP embeddings -> HIP integer path over every frame of the fixture, through the frame-parallel runner in the
    launch shape bench.py times (streams x frames per launch), against the oracle's final maps."""
    sd, net, net_crop = nets
    fix = _fixture()
    n = 0
    while n in fix:
        n += 1
    # default: the first 256 frames (~1 min, most of it generating the synthetic frames on the host);
    # UOC_PARITY_E2E_FRAMES=1024 = all of BASELINE configs[4]'s frames -> profiles/r03_parity_histogram.json
    n = min(n, int(os.environ.get("UOC_PARITY_E2E_FRAMES", "256")))
    assert n >= 8, "tests/golden/bench_oracle/ is missing"
    hist, worst, per_frame = {}, 0, []
    CH = 64                                   # frames resident at a time (7.4 MB each)
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        samples = []
        for g in range(lo, hi):
            img, dep = _bench_frame(g)
            samples.append(dict(image_color=img.to(device), depth=dep.to(device)))
        fn = runner.two_stage_frame_fn(samples, net, net_crop, first_index=lo, frames_per_launch=4)
        # run_sharded as "rank lo/CH of ceil(n/CH)": exactly the global frames [lo, hi) (their RNG seeds are global), no
        # collective — the same call a rank of bench.py --gpus N makes for its block
        maps = runner.run_sharded(n, fn, H, W, device, lo // CH, (n + CH - 1) // CH, False, inflight=3).cpu().numpy()
        assert len(maps) == hi - lo
        for g in range(lo, hi):
            bad = _mismatch(maps[g - lo], fix[g][1])
            per_frame.append(bad)
            hist[bad] = hist.get(bad, 0) + 1
            worst = max(worst, bad)
    exact = hist.get(0, 0) / n
    out = {"frames": n, "histogram_mismatched_pixels": {str(k): hist[k] for k in sorted(hist)}, "worst_frame": worst,
           "exact_fraction": exact, "total_mismatched_pixels": int(sum(per_frame)), "pixels": n * H * W,
           "per_frame": per_frame}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_histogram.json"), "w"))
    print(json.dumps({k: v for k, v in out.items() if k != "per_frame"}))
    assert worst <= E2E_MAX_MISMATCHED_PIXELS, out["histogram_mismatched_pixels"]
    assert exact >= E2E_MIN_EXACT_FRACTION, out["histogram_mismatched_pixels"]
    assert sum(1 for v in per_frame if v <= E2E_P99_MISMATCHED_PIXELS) >= 0.99 * n, out["histogram_mismatched_pixels"]
