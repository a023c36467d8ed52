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
  (c) end to end (HIP embeddings -> HIP integer path) over all 1 024 frames of tests/golden/bench_oracle/ — round 5: the claim
      rests on the EXACT leg.  For EVERY frame whose map differs from the fixture (~68 of 1 024) leg (b) runs automatically:
      the oracle's stage-1 and crop embeddings through the HIP integer path must reproduce the oracle's maps with identical
      ids.  Next to it, as the explanation of WHY a pixel moved: hard count bounds (worst frame <= 32 px, P99 <= 5), the margin
      rule (nearest-seed margin of the oracle's own run, tests/golden/bench_margins/, at most TAU), and for pixels beyond TAU
      a FROZEN perturbation protocol (oracle/margins.PERTURB_RUNS = 8 seeded runs of the oracle at 1x the measured embedding
      error; the HIP networks are NOT an admissible witness; anything that needs more runs or a larger eps is reported as
      `escalated`, bounded and never silently passed)
  (d) TAU is not a free parameter: the margin of a pixel between seeds a and b moves by at most |dx| + (|dz_a| + |dz_b|) / 2
      when its embedding moves by dx and the seeds by dz (L2 norms, unit vectors); (d) measures |dx| (HIP vs oracle
      embeddings) and |dz| (HIP vs oracle converged seeds, i.e. the embedding error through ten kappa = 20 iterations)
      on live oracle runs and asserts their sum stays below TAU

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
EMBED_EPS = 2.5e-6                     # per-component embedding error of the perturbed oracle runs: just below the measured HIP-vs-oracle maximum ((a): 2.5e-6 .. 2.8e-6; bar 1e-3)
MAX_FLAGGED_FRAMES = 12                # frames (of 1 024) that may need the perturbation analysis (~60 s of oracle each); more = a regression
MAX_ESCALATED_FRAMES = 2               # frames the frozen protocol (8 runs at 1x) does not cover and that needed more runs / 2x / 4x eps: reported, bounded (measured: see profiles/r05_parity_margins.json)
E2E_MIN_EXACT_FRACTION = 0.90          # secondary alarm only: share of frames identical up to a permutation (measured 0.934)
# exact leg: a pixel may differ between the HIP and the oracle's integer path ON THE SAME EMBEDDINGS only as a near-tie of the
# oracle's own run — margin <= TAU, the same rule as end to end: the HIP kernels and torch's CPU mm add the same products in
# different orders (converged seeds differ by 1e-7 .. 1e-6, `dz_kernel` of test (d); a sparsely supported seed amplifies that
# through the ten kappa = 20 iterations like any other perturbation).  Measured (profiles/r05_parity_flagged_decomposed.json):
# 3 of 68 frames, ONE pixel each, margins 1.4e-6, 2.5e-6 and 5.4e-5.  Bounded: at most 2 pixels per frame, at most 8 frames.
# Regression alarm next to the north_star's 1e-3 bar (VERDICT r5 "what's weak" 1: the margin rule would let a backbone regression that
# doubles the embedding error pass): the measured HIP-vs-oracle maximum over the mismatching frames and their crops is 3.8e-6 (fp32
# path; 3.9e-6 in the split-precision experiment) — four times that fails the test.
EMBED_REGRESSION_ALARM = 1.5e-5
MAX_KERNEL_ROUNDING_FRAMES = 8
MAX_KERNEL_ROUNDING_PIXELS = 2
E2E_MAX_MISMATCHED_PIXELS = 32         # hard count bounds next to the margin rule: worst frame (measured 24) ...
E2E_P99_MISMATCHED_PIXELS = 5          # ... and the 99th percentile over the frames (measured 4)


def _fixture():
    """frame index -> (stage-1 map after the depth filter, final map), uint8 [480, 640]."""
    out = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "bench_oracle", "frames_*.npz"))):
        z = np.load(path)
        first, s1, fin = int(z["first"]), z["stage1"], z["final"]       # an NpzFile decompresses the whole array on every access
        for i in range(len(fin)):
            out[first + i] = (s1[i], fin[i])
    return out


def _fixture_frames(frames):
    """_fixture() restricted to a few frames (worker processes)."""
    out = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "bench_oracle", "frames_*.npz"))):
        lo, hi = (int(v) for v in os.path.basename(path)[len("frames_"):-len(".npz")].split("_"))
        if any(lo <= g < hi for g in frames):
            z = np.load(path)
            first, s1, fin = int(z["first"]), z["stage1"], z["final"]
            for g in frames:
                if first <= g < first + len(fin):
                    out[g] = (s1[g - first].copy(), fin[g - first].copy())
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


def _memo(network):
    """Caches a CPU network's output per input content (the perturbation runs of a frame reuse stage 1 and, while the stage-1
    map is unchanged, the crops)."""
    import hashlib
    cache = {}

    def net(image, label, depth):
        key = (tuple(image.shape), hashlib.blake2b(image.numpy().tobytes(), digest_size=16).digest(),
               hashlib.blake2b(depth.numpy().tobytes(), digest_size=16).digest())
        if key not in cache:
            if len(cache) > 6:
                cache.clear()
            cache[key] = network(image, label, depth)
        return cache[key]
    return net


def _perturb_worker(args):
    """One run of the frozen perturbation protocol in a worker process (the 8 runs of a frame are independent): the oracle's
    base embeddings come from the parent through /dev/shm (stage 1; the crops as long as the perturbed stage-1 map leaves them
    unchanged), anything else is recomputed with the oracle's network."""
    g, index, eps, shm = args
    torch.set_num_threads(8)
    from oracle import margins as M
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    z = {k: torch.from_numpy(np.load(os.path.join(shm, k + ".npy"))) for k in ("f1", "rgb_c", "f2", "base")}
    img, dep = _bench_frame(g)

    def net(image, label, depth):
        if image.shape == img.shape and torch.equal(image, img):
            return z["f1"]
        if image.shape == z["rgb_c"].shape and torch.equal(image, z["rgb_c"]):
            return z["f2"]
        return BO.segnet_forward(sd, image, depth)
    return M.perturbation_run(img, dep, net, runner.frame_rng_seed(g), z["base"].numpy(), eps, index)


_worker_sd = None


def _oracle_embed_worker(args):
    """The oracle's two network passes of bench frame g (stage 1; its own crops rebuilt from the fixture's stage-1 map) in a
    worker process: the exact leg walks ~68 frames and the CPU backbone is its cost.  Results go through /dev/shm."""
    global _worker_sd
    g, shm, threads = args
    torch.set_num_threads(threads)
    if _worker_sd is None:
        _worker_sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    fix = _fixture_frames([g])
    img, dep = _bench_frame(g)
    f1 = BO.segnet_forward(_worker_sd, img, dep)
    rgb_c, mask_c, rois, dep_c = GO.crop_rois(img, torch.from_numpy(fix[g][0].astype(np.float32))[None].clone(), dep)
    f2 = BO.segnet_forward(_worker_sd, rgb_c, dep_c)
    np.save(os.path.join(shm, f"f1_{g}.npy"), f1.numpy())
    np.save(os.path.join(shm, f"f2_{g}.npy"), f2.numpy())
    return g


def _decompose(g, fix, sd, net, net_crop, device, end_to_end=True, live_oracle_fallback=False, shm=None):
    """Legs (a) + (b) on bench frame g.  (b): the ORACLE's embeddings (stage 1, and its own crops rebuilt from its stage-1 map,
    test_dataset.py:62-112) through the HIP clustering, depth filter, ROI table, crops' masks, match statistics and paste —
    compared with the committed oracle maps id for id.  `live_oracle_fallback`: if the maps are not identical to the fixture
    (made on another host), the oracle's own integer path runs HERE on the very same embeddings and the comparison is
    repeated against that (which is what "given the oracle's embeddings" means to the last bit)."""
    img, dep = _bench_frame(g)
    want_out = torch.from_numpy(fix[g][0].astype(np.float32))[None]
    want_final = fix[g][1]
    rgb_c, mask_c, rois, dep_c = GO.crop_rois(img, want_out.clone(), dep)
    K = rgb_c.shape[0]
    assert K >= 5, "the headline frames must exercise stage 2"
    if shm is not None:        # computed by _oracle_embed_worker
        f1 = torch.from_numpy(np.load(os.path.join(shm, f"f1_{g}.npy")))
        f2 = torch.from_numpy(np.load(os.path.join(shm, f"f2_{g}.npy")))
        os.remove(os.path.join(shm, f"f1_{g}.npy"))
        os.remove(os.path.join(shm, f"f2_{g}.npy"))
    else:
        f1 = BO.segnet_forward(sd, img, dep)
        f2 = BO.segnet_forward(sd, rgb_c, dep_c)
    # (a) HIP embeddings on the same inputs
    e1 = net(img.to(device), None, dep.to(device)).cpu()
    e2 = net_crop(rgb_c.to(device), None, dep_c.to(device)).cpu()
    err1, err2 = float((e1 - f1).abs().max()), float((e2 - f2).abs().max())
    # (b) the stubs ignore their inputs: stage 2 clusters the oracle's crop embeddings
    stub1 = lambda image, label, depth: f1.to(device)
    stub2 = lambda image, label, depth: f2.to(device)
    np.random.seed(runner.frame_rng_seed(g))
    got_out, got_ref = TD.test_sample(dict(image_color=img, depth=dep), stub1, stub2)
    s1_same = bool(np.array_equal(got_out[0].numpy().astype(np.int64), want_out[0].numpy().astype(np.int64)))
    fin_same = bool(got_ref is not None and np.array_equal(got_ref[0].numpy().astype(np.int64), want_final.astype(np.int64)))
    row = {"frame": g, "rois": K, "embed_err_stage1": err1, "embed_err_crops": err2, "oracle_objects": int(want_final.max()),
           "given_oracle_embeddings": {
               "stage1_identical_ids": s1_same, "final_identical_ids": fin_same, "identical_to": "fixture" if s1_same and fin_same else None,
               "stage1_exact_up_to_permutation": bool(O.labels_equal_up_to_permutation(got_out.numpy(), want_out.numpy())),
               "final_exact_up_to_permutation": bool(got_ref is not None and O.labels_equal_up_to_permutation(got_ref[0].numpy(), want_final)),
               "final_mismatched_pixels": _mismatch(got_ref[0].numpy(), want_final) if got_ref is not None else H * W}}
    if live_oracle_fallback and not (s1_same and fin_same):
        from oracle import margins as M
        live_out, live_ref, info = M.test_sample_with_margins(img, dep, lambda *a: f1, lambda *a: f2,
                                                              np.random.RandomState(runner.frame_rng_seed(g)))
        l1 = bool(np.array_equal(got_out[0].numpy().astype(np.int64), live_out[0].numpy().astype(np.int64)))
        lF = bool(got_ref is not None and live_ref is not None and
                  np.array_equal(got_ref[0].numpy().astype(np.int64), live_ref[0].numpy().astype(np.int64)))
        upd = dict(identical_to="the oracle's integer path on this host, same embeddings" if l1 and lF else None,
                   live_stage1_identical_ids=l1, live_final_identical_ids=lF,
                   live_oracle_vs_fixture_pixels=_mismatch(live_ref[0].numpy(), want_final) if live_ref is not None else None)
        if not (l1 and lF) and got_ref is not None and live_ref is not None:
            # the HIP kernels and torch's CPU mm sum the same products in different orders: the converged seeds differ by
            # ~1e-7 .. 1e-6 (`dz_kernel` of test (d)), and a pixel whose margin in the oracle's OWN run is below that is decided by
            # the last bit of either sum.  Report every such pixel with that margin; the caller bounds them.
            px1 = M.label_changes(live_out[0].numpy(), got_out[0].numpy())
            pxF = M.label_changes(live_ref[0].numpy(), got_ref[0].numpy())
            upd["kernel_rounding_pixels"] = (
                [{"map": "stage1", "y": int(p // W), "x": int(p % W), "margin": float(info["margin1"][p])} for p in px1.tolist()] +
                [{"map": "final", "y": int(p // W), "x": int(p % W), "margin": float(info["marginF"].reshape(-1)[p])} for p in pxF.tolist()])
        row["given_oracle_embeddings"].update(upd)
    if end_to_end:
        np.random.seed(runner.frame_rng_seed(g))
        e2e_out, e2e_ref = TD.test_sample(dict(image_color=img, depth=dep), net, net_crop)
        row["end_to_end_mismatched_pixels"] = _mismatch((e2e_ref if e2e_ref is not None else e2e_out)[0].numpy(), want_final)
        row["end_to_end_objects"] = int((e2e_ref if e2e_ref is not None else e2e_out).max())
    return row


def _decompose_many(frames, fix, sd, net, net_crop, device, end_to_end, live_oracle_fallback):
    """_decompose over a list of frames with the oracle's network passes (the cost: two CPU backbone passes per frame)
    computed a few frames ahead in worker processes and handed over through files."""
    import shutil
    import tempfile
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    tmp = tempfile.mkdtemp(prefix="uoc_parity_")       # (a container's /dev/shm may hold 64 MB; the page cache does the same job)
    ncpu = len(os.sched_getaffinity(0))
    nwork = max(1, min(8, ncpu // 16))
    out = []
    try:
        with ProcessPoolExecutor(nwork, mp_context=mp.get_context("spawn")) as pool:     # spawn: this process holds a HIP context
            for lo in range(0, len(frames), 2 * nwork):       # bounded run-ahead: ~170 MB of embeddings per frame
                for g in pool.map(_oracle_embed_worker, [(g, tmp, max(4, min(16, ncpu // nwork))) for g in frames[lo:lo + 2 * nwork]]):
                    out.append(_decompose(g, fix, sd, net, net_crop, device, end_to_end=end_to_end,
                                          live_oracle_fallback=live_oracle_fallback, shm=tmp))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def test_embeddings_and_integer_path_separately(device, nets):
    """(a) + (b) on bench frames 0 .. DECOMPOSED_FRAMES-1."""
    sd, net, net_crop = nets
    fix = _fixture()
    frames = [g for g in range(DECOMPOSED_FRAMES) if g in fix]
    if os.environ.get("UOC_PARITY_FRAME_LIST"):        # ad hoc: the decomposition on chosen frames, e.g. the histogram's outliers
        frames = [int(v) for v in os.environ["UOC_PARITY_FRAME_LIST"].split(",")]
    assert len(frames) >= min(DECOMPOSED_FRAMES, 8) or os.environ.get("UOC_PARITY_FRAME_LIST"), "tests/golden/bench_oracle/ is missing"
    report = _decompose_many(frames, fix, sd, net, net_crop, device, end_to_end=True,
                             live_oracle_fallback=bool(os.environ.get("UOC_PARITY_FRAME_LIST")))
    worst_embed = max(max(r["embed_err_stage1"], r["embed_err_crops"]) for r in report)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    summary = {"frames": len(frames), "embed_max_err": worst_embed,
               "exact_given_oracle_embeddings": all(r["given_oracle_embeddings"]["final_exact_up_to_permutation"] and
                                                    r["given_oracle_embeddings"]["stage1_exact_up_to_permutation"] for r in report),
               "identical_ids_given_oracle_embeddings": all(r["given_oracle_embeddings"]["final_identical_ids"] and
                                                            r["given_oracle_embeddings"]["stage1_identical_ids"] for r in report),
               "per_frame": report}
    json.dump(summary, open(os.path.join(ROOT, "gpurun_out", "parity_decomposed.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "per_frame"}))
    assert worst_embed <= 1e-3, worst_embed                                  # north_star: embeddings within 1e-3 fp32
    for r in report:                                                         # north_star: integer labels bit-exact
        assert r["given_oracle_embeddings"]["stage1_exact_up_to_permutation"], r
        assert r["given_oracle_embeddings"]["final_exact_up_to_permutation"], r


def _margins():
    from oracle import margins as M
    return M.load_bench_margins(ROOT)


def _mismatching_pixels(a, b):
    """Flat indices of the pixels on which partition `a` disagrees with `b` under the best one-to-one relabelling."""
    from oracle import margins as M
    return M.label_changes(b, a)


def _lookup(idx, val, pix):
    from oracle import margins as M
    return M.lookup_margins(idx, val, pix)


def _frames_on_host(indices):
    return [synth.palette_frame(10_000 + g, H, W, 5 + (10_000 + g) % 3) for g in indices]


def _frame_pair(g):
    fr = synth.palette_frame(10_000 + g, H, W, 5 + (10_000 + g) % 3)
    return fr["image_color"], fr["depth"]


def test_end_to_end_margin_bounded(device, nets):
    """(c) north_star's "integer labels bit-exact up to label permutation" on every frame of BASELINE configs[4] (1 024 by
    default): HIP embeddings -> HIP integer path through the frame-parallel runner in the launch shape bench.py times,
    against the oracle's final maps.  The claim, in the order of its strength:
      1. EXACT: for every frame whose map differs from the oracle's, the oracle's embeddings through the HIP integer path
         reproduce the oracle's maps id for id (leg (b), `_decompose`; compared with the committed fixture, and — only if the
         fixture's host rounded an embedding differently — with the oracle's integer path run here on the same embeddings).
         Measured: 65 of 68 frames; on the other 3 ONE pixel differs, a near-tie of the oracle's own run (margins 1.4e-6,
         2.5e-6, 5.4e-5 <= TAU: the HIP kernels and torch's CPU mm add the same products in different orders, and the hill
         climbing amplifies that like any perturbation) — listed with their margins, at most MAX_KERNEL_ROUNDING_FRAMES
         frames of at most MAX_KERNEL_ROUNDING_PIXELS pixels;
      2. BOUNDED: worst frame <= 32 mismatching pixels, 99th percentile <= 5;
      3. EXPLAINED: every mismatching pixel is a near-tie of the oracle's own run (margin <= TAU) or, beyond TAU, a pixel the
         oracle itself flips under the FROZEN perturbation protocol (8 seeded runs at 1x the measured embedding error).
         Pixels that need more are `escalated`: reported, at most MAX_ESCALATED_FRAMES frames; unexplained pixels fail."""
    import time
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing as mp
    from oracle import margins as M
    sd, net, net_crop = nets
    fix, mar = _fixture(), _margins()
    n = 0
    while n in fix and n in mar:
        n += 1
    n = min(n, int(os.environ.get("UOC_PARITY_E2E_FRAMES", "1024")))
    assert n >= 8, "tests/golden/bench_oracle/ or bench_margins/ is missing"
    CH = 64                                   # frames resident at a time (7.4 MB each)
    hist, per_frame, pixels, beyond, worst_margin, flagged, bifurcated, mismatching = {}, [], [], [], 0.0, [], [], []
    workers = max(1, min(32, len(os.sched_getaffinity(0)) - 2))
    t_start = time.time()
    with ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn")) as pool:     # spawn: this process holds a HIP context
        chunks = [(lo, min(n, lo + CH)) for lo in range(0, n, CH)]
        pending = pool.map(_frame_pair, range(0, n), chunksize=4)       # host-side synthesis runs ahead of the GPU
        for lo, hi in chunks:
            samples = []
            for g in range(lo, hi):
                img, dep = next(pending)
                samples.append(dict(image_color=torch.from_numpy(img).to(device), depth=torch.from_numpy(dep).to(device)))
            fn = runner.two_stage_frame_fn(samples, net, net_crop, first_index=lo, frames_per_launch=4)
            maps = runner.run_sharded(n, fn, H, W, device, gather=False, inflight=3, block_range=(lo, hi)).cpu().numpy()
            assert len(maps) == hi - lo
            for g in range(lo, hi):
                bad = _mismatching_pixels(maps[g - lo], fix[g][1])
                per_frame.append(len(bad))
                hist[len(bad)] = hist.get(len(bad), 0) + 1
                if len(bad) == 0:
                    continue
                mismatching.append(g)
                m = _lookup(mar[g]["idxF"], mar[g]["valF"], bad)
                over = m > M.TAU
                if over.any():
                    flagged.append((g, bad, m))
                    continue
                worst_margin = max(worst_margin, float(m.max()))
                for p, v in zip(bad.tolist(), m.tolist()):
                    pixels.append({"frame": g, "y": p // W, "x": p % W, "margin": round(v, 9), "class": "near_tie"})
    t_e2e = time.time() - t_start
    # ---- 1. the exact leg on EVERY mismatching frame ----
    import shutil
    import tempfile
    t_start = time.time()
    decomposed = _decompose_many(mismatching, fix, sd, net, net_crop, device, end_to_end=False, live_oracle_fallback=True)
    t_dec = time.time() - t_start
    shm = tempfile.mkdtemp(prefix="uoc_parity_")
    def _rounding_only(r):
        px = r["given_oracle_embeddings"].get("kernel_rounding_pixels")
        return bool(px) and len(px) <= MAX_KERNEL_ROUNDING_PIXELS and all(p["margin"] <= M.TAU for p in px)
    rounding = [r["frame"] for r in decomposed if r["given_oracle_embeddings"]["identical_to"] is None and _rounding_only(r)]
    not_exact = [r["frame"] for r in decomposed if r["given_oracle_embeddings"]["identical_to"] is None and not _rounding_only(r)]
    dec = {"frames": [r["frame"] for r in decomposed], "count": len(decomposed),
           "differ_by_near_tie_pixels_of_the_oracle_run": rounding,
           "margins_of_those_pixels": sorted(p["margin"] for r in decomposed for p in r["given_oracle_embeddings"].get("kernel_rounding_pixels", [])),
           "largest_margin_of_such_a_pixel": max([p["margin"] for r in decomposed for p in r["given_oracle_embeddings"].get("kernel_rounding_pixels", [])] or [0.0]),
           "identical_ids_to_the_fixture": sum(r["given_oracle_embeddings"]["identical_to"] == "fixture" for r in decomposed),
           "identical_ids_to_the_oracle_on_this_host_only": sum((r["given_oracle_embeddings"]["identical_to"] or "").startswith("the oracle") for r in decomposed),
           "not_identical": not_exact, "embed_max_err": max([max(r["embed_err_stage1"], r["embed_err_crops"]) for r in decomposed] or [0.0]),
           "seconds": round(t_dec, 1), "per_frame": decomposed}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dec, open(os.path.join(ROOT, "gpurun_out", "parity_flagged_decomposed.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in dec.items() if k != "per_frame"}))
    # ---- 3. frames with a pixel beyond the margin: either a regression, or a seed between two modes (oracle/margins.py: the
    # margin bounds the assignment GIVEN the seeds; at such a seed the oracle's own result flips with the last bit of a sum —
    # bench frame 246: 1 vs 4 torch threads).  Ask the ORACLE, by the frozen protocol: its whole path again with its own
    # embeddings perturbed by the measured embedding error, M.PERTURB_RUNS seeded sign patterns at 1x.  A mismatching pixel is
    # explained if its label changes in one of those runs, or it is a near-tie of this host's oracle run, or this host's oracle
    # run itself differs from the committed one there.  What is left goes to the escalation (more runs, 2x / 4x eps) and is
    # REPORTED as escalated; what even that does not cover fails the test.
    assert len(flagged) <= MAX_FLAGGED_FRAMES, [f[0] for f in flagged]
    cpu_net = _memo(lambda image, label, depth: BO.segnet_forward(sd, image, depth))
    t_start = time.time()
    escalated_frames = []
    pert_pool = ProcessPoolExecutor(min(M.PERTURB_RUNS, max(1, len(os.sched_getaffinity(0)) // 8)), mp_context=mp.get_context("spawn"))
    for g, bad, m in flagged:
        img, dep = _bench_frame(g)
        seen = []
        rec_net = lambda image, label, depth: (seen.append((image, cpu_net(image, label, depth))) or seen[-1][1])

        def run_many(base, indices, g=g):        # the protocol's runs, side by side in worker processes
            np.save(os.path.join(shm, "f1.npy"), seen[0][1].numpy())
            np.save(os.path.join(shm, "rgb_c.npy"), seen[-1][0].numpy())
            np.save(os.path.join(shm, "f2.npy"), seen[-1][1].numpy())
            np.save(os.path.join(shm, "base.npy"), np.asarray(base))
            return list(pert_pool.map(_perturb_worker, [(g, i, EMBED_EPS, shm) for i in indices]))
        changed, base, info, per_run = M.unresolved_pixels(img, dep, rec_net, runner.frame_rng_seed(g), EMBED_EPS, run_many=run_many)
        here_vs_there = M.label_changes(fix[g][1], base)
        near = (m <= M.TAU) | (info["marginF"].reshape(-1)[bad] <= M.TAU)
        ok = np.isin(bad, changed) | np.isin(bad, here_vs_there) | near
        esc_ok, esc_used, esc_factor = np.zeros(len(bad), bool), 0, 1
        if not ok.all():
            extra, esc_used, esc_factor = M.escalated_pixels(img, dep, cpu_net, runner.frame_rng_seed(g), base, EMBED_EPS, bad[~ok])
            esc_ok = ~ok & np.isin(bad, extra)
            escalated_frames.append(g)
        for p, v, good, nr, esc in zip(bad.tolist(), m.tolist(), ok.tolist(), near.tolist(), esc_ok.tolist()):
            pixels.append({"frame": g, "y": p // W, "x": p % W, "margin": (round(v, 9) if np.isfinite(v) else None),
                           "class": "near_tie" if nr else "unresolved_by_the_oracle" if good else "escalated" if esc else "BEYOND_MARGIN"})
        need, acc, first_cover = bad[m > M.TAU], np.zeros(0, np.int64), None
        for i, c in enumerate(per_run):
            acc = np.union1d(acc, c)
            if np.isin(need, acc).all():
                first_cover = i + 1
                break
        bifurcated.append({"frame": g, "mismatching_pixels": int(len(bad)), "beyond_tau": int((m > M.TAU).sum()),
                           "pixels_the_oracle_flips_under_perturbation": int(len(changed)), "protocol_runs": len(per_run),
                           "runs_until_covered": first_cover, "escalation_runs": esc_used, "escalation_eps_factor": esc_factor,
                           "escalated_pixels": int(esc_ok.sum()),
                           "oracle_here_vs_fixture_pixels": int(len(here_vs_there)), "unexplained": int((~ok & ~esc_ok).sum())})
        if not (ok | esc_ok).all():
            beyond.append(bifurcated[-1])
    pert_pool.shutdown()
    shutil.rmtree(shm, ignore_errors=True)
    t_pert = time.time() - t_start
    exact = hist.get(0, 0) / n
    srt = sorted(per_frame)
    p99 = srt[min(len(srt) - 1, int(np.ceil(0.99 * len(srt))) - 1)]
    out = {"frames": n, "tau": M.TAU, "mismatching_frames": len(mismatching), "mismatching_pixels": len(pixels), "pixels_total": n * H * W,
           "worst_frame_pixels": max(per_frame), "p99_frame_pixels": p99,
           "exact_leg_on_every_mismatching_frame": {k: v for k, v in dec.items() if k not in ("per_frame", "frames")},
           "mismatches_unexplained": int(sum(b["unexplained"] for b in beyond)), "largest_margin_of_a_near_tie_mismatch": worst_margin,
           "embedding_perturbation": EMBED_EPS, "perturbation_protocol": f"{M.PERTURB_RUNS} seeded runs of the oracle at 1x eps, oracle embeddings only",
           "escalated_frames": escalated_frames, "frames_with_an_unresolved_seed": bifurcated,
           "exact_fraction": exact, "histogram_mismatched_pixels": {str(k): hist[k] for k in sorted(hist)},
           "seconds": {"end_to_end": round(t_e2e, 1), "exact_leg": round(t_dec, 1), "perturbation": round(t_pert, 1)},
           "beyond": beyond, "pixels": pixels, "per_frame": per_frame}
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_margins.json"), "w"))
    print(json.dumps({k: v for k, v in out.items() if k not in ("per_frame", "pixels")}))
    assert not not_exact, f"integer path not identical given the oracle's embeddings on frames {not_exact}"      # 1. the exact leg
    assert len(rounding) <= MAX_KERNEL_ROUNDING_FRAMES, rounding
    assert dec["embed_max_err"] <= 1e-3
    assert dec["embed_max_err"] <= EMBED_REGRESSION_ALARM, f"embedding error {dec['embed_max_err']:.2e}: inside the 1e-3 bar but 4x the measured 3.8e-6"
    assert max(per_frame) <= E2E_MAX_MISMATCHED_PIXELS and p99 <= E2E_P99_MISMATCHED_PIXELS, (max(per_frame), p99)   # 2. counts
    assert not beyond, beyond                                   # 3. nothing differs beyond the margin unexplained
    assert worst_margin <= M.TAU
    assert len(escalated_frames) <= MAX_ESCALATED_FRAMES, escalated_frames
    # secondary (a coarse regression alarm, not the claim): most frames are identical outright
    assert exact >= E2E_MIN_EXACT_FRACTION, out["histogram_mismatched_pixels"]


AMP_FRAMES = int(os.environ.get("UOC_PARITY_AMP_FRAMES", "3"))


def test_tau_is_the_measured_perturbation(device, nets):
    """(d) the perturbation that TAU stands for, measured: the oracle's full two-stage run (oracle/margins) on the host
    next to the HIP kernels on the first AMP_FRAMES bench frames.  Reports, for stage 1 and for the crops,
      dx            max L2 error of a pixel's embedding (HIP network vs oracle network, same inputs)
      dz_kernel     max L2 distance of a converged seed, HIP hill climbing vs oracle, SAME (oracle) embeddings
      dz            the same with the HIP embeddings: dx amplified by ten kappa = 20 iterations (99th percentile over the
                    seeds; dz_max and the number of seeds beyond TAU next to it — seeds between two modes)
    and asserts dx + dz <= TAU.  Also checks the oracle run on THIS host against the committed near-tie sets."""
    from oracle import margins as M
    from unseenobjectclustering_amd.utils import mean_shift as MS
    sd, net, net_crop = nets
    mar = _margins()
    cpu_net = lambda image, label, depth: BO.segnet_forward(sd, image, depth)
    rows = []
    for g in range(AMP_FRAMES):
        img, dep = _bench_frame(g)
        rng = np.random.RandomState(runner.frame_rng_seed(g))
        firsts = np.random.RandomState(runner.frame_rng_seed(g))
        out, refined, info = M.test_sample_with_margins(img, dep, cpu_net, cpu_net, rng)
        K = info["X2"].shape[0]
        first1 = [int(firsts.randint(0, H * W))]
        first2 = [int(firsts.randint(0, 224 * 224)) for _ in range(K)]
        rgb_c, mask_c, rois, dep_c = GO.crop_rois(img, out.clone(), dep)
        e1 = net(img.to(device), None, dep.to(device))                       # [1, 64, H, W] view of pixel-major rows
        e2 = net_crop(rgb_c.to(device), None, dep_c.to(device))
        X1h = e1[0].permute(1, 2, 0).reshape(-1, 64).contiguous()
        X2h = e2.permute(0, 2, 3, 1).reshape(K, -1, 64).contiguous()
        row = {"frame": g, "rois": K}
        for tag, Xh, Xo, Zo, first in (("stage1", X1h[None], info["X1"][None], info["Z1"][None], first1),
                                       ("crops", X2h, info["X2"], info["Z2"], first2)):
            dx = float((Xh.cpu() - Xo).norm(dim=-1).max())
            _, _, Zk, _ = MS.cluster_batch(Xo.to(device), first, 20.0, 100, 10, 0.04, return_parts=True)
            _, _, Zh, _ = MS.cluster_batch(Xh, first, 20.0, 100, 10, 0.04, return_parts=True)
            dzk, dzh = (Zk.cpu() - Zo).norm(dim=-1).reshape(-1), (Zh.cpu() - Zo).norm(dim=-1).reshape(-1)
            # a seed between two modes has no bounded amplification (oracle/margins.py, bench frame 246): the bound is stated
            # for the 99th percentile of the seeds, the rest is counted (and covered by (c)'s perturbation analysis)
            row[tag] = {"dx": dx, "dz_kernel": float(dzk.quantile(0.99)), "dz": float(dzh.quantile(0.99)), "dz_max": float(dzh.max()),
                        "seeds": int(dzh.numel()), "seeds_moved_more_than_tau": int((dzh > M.TAU).sum())}
            row[tag]["amplification"] = row[tag]["dz"] / max(dx, 1e-12)
        # the oracle on this host against the committed near-tie sets (made in the build container): same pixels below
        # TAU_STORE / 2 (the edge of the stored set may move by an ulp of the margin)
        idxF, valF = M.sparse_below(info["marginF"], M.TAU_STORE / 2)
        stored = _lookup(mar[g]["idxF"], mar[g]["valF"], idxF.astype(np.int64))
        row["near_tie_set_reproduced"] = bool(np.isfinite(stored).all() and np.abs(stored - valF).max() < 1e-4)
        rows.append(row)
    worst = max(r[t]["dx"] + r[t]["dz"] for r in rows for t in ("stage1", "crops"))
    seeds = sum(r[t]["seeds"] for r in rows for t in ("stage1", "crops"))
    moved = sum(r[t]["seeds_moved_more_than_tau"] for r in rows for t in ("stage1", "crops"))
    dz_max = max(r[t]["dz_max"] for r in rows for t in ("stage1", "crops"))
    out = {"tau": M.TAU, "worst_dx_plus_dz_q99": worst, "largest_seed_movement": dz_max, "seeds": seeds,
           "seeds_moved_more_than_tau": moved, "frames": rows}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_tau.json"), "w"), indent=1)
    print(json.dumps(out))
    # TAU bounds what the embedding error does to a margin: |dx| + the movement of the two seeds involved.  Well supported
    # seeds contract the error (q99 of |dz| < |dx|); sparsely supported ones amplify it up to a few 1e-4 (measured maximum
    # over 2 400 seeds: 3.9e-4); seeds beyond TAU must be rare — they are the ones (c) hands to the perturbation analysis
    assert worst <= M.TAU, out
    assert moved <= 0.005 * seeds, out
    assert all(r["near_tie_set_reproduced"] for r in rows), rows
