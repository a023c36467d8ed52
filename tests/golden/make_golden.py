"""Golden-vector generator — runs ONLY in the build container, where /root/reference exists.

Imports the reference's own Python (through tests/golden/ref_harness.py) and records its
outputs on repo-owned synthetic inputs.  The .npz files written next to this script are the
fixtures tests/ compares the oracle (CPU) and the HIP path (GPU) against.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [meanshift|seedcont|glue|backbone|e2e|demo|modes|evaluation|prep|npy|segnet|all]
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
from cases import (SEED_CONTINUATION_CASES, continuation_inputs, MEANSHIFT_CASES, KAPPA, EPSILON, RNG_SEED, BACKBONE_CASES, GLUE_CASES, E2E_CASES,  # noqa: E402
                   MODES, MODE_BACKBONE_CASES, MODE_GLUE_CASES, MODE_E2E_CASES, WIDE_MEANSHIFT_CASES,
                   EVAL_CASES, eval_pair, munkres_cases, PREP_SYNTH, prep_synthetic_arrays, SEGNET_RUNS, SegnetLoader, npy_frames,
                   segnet_samples, segnet_stub_networks, SEGNET_METRIC_KEYS,
                   sample_positions, glue_inputs, crop_cluster_labels, e2e_stub_features)
from unseenobjectclustering_amd import synth  # noqa: E402


def make_meanshift(ref):
    ms = ref.mean_shift
    out = {}
    for name, c in MEANSHIFT_CASES.items():
        X, _ = synth.embedding_field(c["seed"], c["H"], c["W"], 64, c["num_objects"], c["noise"])
        Xt = torch.from_numpy(X)
        # full pipeline exactly as the reference runs it (global numpy RNG seeded like tools/test_images.py:154)
        np.random.seed(RNG_SEED)
        labels, idx = ms.mean_shift_smart_init(Xt, KAPPA, num_seeds=c["m"], max_iters=c["iters"], metric="cosine")
        # the same, stage by stage, to record intermediates
        np.random.seed(RNG_SEED)
        seeds, idx2 = ms.select_smart_seeds(Xt, c["m"], return_selected_indices=True, metric="cosine")
        assert torch.equal(idx, idx2)
        Z = ms.seed_hill_climbing_ball(Xt, seeds, KAPPA, max_iters=c["iters"], metric="cosine")
        seed_labels = ms.connected_components(Z, 2 * ref.cfg.TRAIN.EMBEDDING_ALPHA, metric="cosine")
        assert abs(2 * ref.cfg.TRAIN.EMBEDDING_ALPHA - EPSILON) < 1e-12
        assert int(labels.max()) < 255
        out[name + "/labels"] = labels.numpy().astype(np.uint8)
        out[name + "/indices"] = idx.numpy().astype(np.int32)
        out[name + "/Z"] = Z.numpy().astype(np.float32)
        out[name + "/seed_labels"] = seed_labels.numpy().astype(np.int32)
        print(name, "clusters:", np.unique(out[name + "/labels"]).tolist(), "first idx", int(idx[0]), flush=True)
    np.savez_compressed(os.path.join(HERE, "meanshift.npz"), **out)


def make_seedcont(ref):
    """select_smart_seeds with init_seeds / num_init_seeds, the reference's own code (mean_shift.py:142-170)."""
    ms = ref.mean_shift
    out = {}
    for name, c in SEED_CONTINUATION_CASES.items():
        X, init = continuation_inputs(c)
        Xt, it = torch.from_numpy(X), torch.from_numpy(init.copy())
        if c["init"] == "rows":
            np.random.seed(RNG_SEED)
            plain, pidx = ms.select_smart_seeds(Xt, c["m"], return_selected_indices=True, metric="cosine")
            it[:c["k"]] = plain[:c["k"]]
            out[name + "/plain_indices"] = pidx.numpy().astype(np.int32)
        np.random.seed(RNG_SEED)
        seeds, idx = ms.select_smart_seeds(Xt, c["m"], return_selected_indices=True, init_seeds=it,
                                           num_init_seeds=c["k"], metric="cosine")
        assert seeds.data_ptr() == it.data_ptr()                       # the reference selects in place
        out[name + "/rng_draws"] = np.int32(0 if c["k"] else 1)
        out[name + "/next_rng"] = np.int64(np.random.randint(0, 1 << 30))   # where the global RNG stands afterwards
        out[name + "/indices"] = idx.numpy().astype(np.int32)
        out[name + "/seeds"] = seeds.numpy().astype(np.float32)
        if c["init"] == "rows":
            assert np.array_equal(out[name + "/indices"][c["k"]:], out[name + "/plain_indices"][c["k"]:])
        print(name, "indices", out[name + "/indices"][:10].tolist(), flush=True)
    np.savez_compressed(os.path.join(HERE, "seedcont.npz"), **out)


def make_backbone(ref):
    import contextlib
    import io
    out = {}
    for name, c in BACKBONE_CASES.items():
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.synthetic_state_dict(c["wseed"]).items()}
        with contextlib.redirect_stdout(io.StringIO()):
            net = ref.networks.__dict__["seg_resnet34_8s_embedding"](2, 64, sd).eval()
        frames = [synth.rgbd_frame(s, c["H"], c["W"], 4) for s in c["frames"]]
        img = torch.from_numpy(np.concatenate([f["image_color"] for f in frames]))
        dep = torch.from_numpy(np.concatenate([f["depth"] for f in frames]))
        with torch.no_grad():
            feat = net(img, None, dep)                       # [B,64,H,W]
        B = feat.shape[0]
        flat = feat.permute(0, 2, 3, 1).reshape(B, -1, 64).numpy()
        if c["samples"]:
            pos = sample_positions(99, flat.shape[1], c["samples"])
            out[name + "/pos"] = pos
            out[name + "/embed"] = flat[:, pos].astype(np.float32)
        else:
            out[name + "/embed"] = flat.astype(np.float32)
        print(name, feat.shape, "norm check", float(feat.norm(dim=1).mean()), flush=True)
    np.savez_compressed(os.path.join(HERE, "backbone.npz"), **out)


def make_glue(ref):
    td = ref.test_dataset
    out = {}
    for name, c in GLUE_CASES.items():
        img, lab, depth, gt = glue_inputs(c)
        filt = td.filter_labels_depth(lab, depth, 0.8)
        out[name + "/filtered"] = filt.numpy().astype(np.uint8)
        rgb_c, mask_c, rois, depth_c = td.crop_rois(img, filt.clone(), depth)
        K = rgb_c.shape[0]
        out[name + "/rois"] = rois.numpy().astype(np.int32)
        out[name + "/mask_crops"] = np.packbits(mask_c.numpy().astype(np.uint8), axis=None)
        pos = sample_positions(5, 3 * 224 * 224, 768)
        out[name + "/crop_pos"] = pos
        out[name + "/rgb_crops_s"] = rgb_c.reshape(K, -1)[:, pos].numpy() if K else np.zeros((0, 768), np.float32)
        out[name + "/depth_crops_s"] = depth_c.reshape(K, -1)[:, pos].numpy() if K else np.zeros((0, 768), np.float32)
        out[name + "/rgb_crops_sum"] = rgb_c.double().sum(dim=(1, 2, 3)).numpy()
        if K:
            labels_c = crop_cluster_labels(c, gt, rois)
            refined, labels_c2 = td.match_label_crop(filt, labels_c.clone(), mask_c, rois, depth_c)
            out[name + "/refined"] = refined.numpy().astype(np.uint8)
            out[name + "/labels_crop_out"] = labels_c2.numpy().astype(np.int8)
        print(name, "K =", K, "rois", rois.numpy().astype(int).tolist(), flush=True)
    np.savez_compressed(os.path.join(HERE, "glue.npz"), **out)


def make_e2e(ref):
    td = ref.test_dataset
    out = {}
    for name, c in E2E_CASES.items():
        fr = synth.rgbd_frame(c["seed"], 480, 640, c["objects"])
        sample = dict(image_color=torch.from_numpy(fr["image_color"]), depth=torch.from_numpy(fr["depth"]))
        net = lambda img, label, depth, c=c: e2e_stub_features(c["seed"], 480, 640, c["objects"] + 2)
        net_crop = lambda rgb, label, depth, c=c: torch.cat(
            [e2e_stub_features(1000 + 10 * c["seed"] + k, 224, 224, 2 + k % 3) for k in range(rgb.shape[0])])
        np.random.seed(RNG_SEED)
        out_label, refined = td.test_sample(sample, net, net_crop)
        out[name + "/out_label"] = out_label.numpy().astype(np.uint8)
        out[name + "/refined"] = refined.numpy().astype(np.uint8)
        print(name, "labels", np.unique(out[name + "/out_label"]).tolist(), "refined", np.unique(out[name + "/refined"]).tolist(), flush=True)
    np.savez_compressed(os.path.join(HERE, "e2e.npz"), **out)


def make_demo(ref):
    """BASELINE config 1: one 640x480 RGB-D pair of the reference's data/demo through the reference's
    own test_sample (real SEGNET, calibrated synthetic weights for both networks).  The PNG pair is
    copied next to the fixture as INPUT DATA; expected outputs are the reference's label maps."""
    import contextlib
    import io
    import json
    import shutil
    from unseenobjectclustering_amd import io as uio
    demo = os.path.join(ref_harness.REFERENCE_ROOT, "data", "demo")
    dst = os.path.join(HERE, "demo")
    os.makedirs(dst, exist_ok=True)
    for f in ("000002-color.png", "000002-depth.png", "camera_params.json"):
        shutil.copy(os.path.join(demo, f), os.path.join(dst, f))
        os.chmod(os.path.join(dst, f), 0o644)
    cam = json.load(open(os.path.join(dst, "camera_params.json")))
    sample = uio.read_sample(os.path.join(dst, "000002-color.png"), os.path.join(dst, "000002-depth.png"), cam)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.calibrated_state_dict().items()}
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref.networks.__dict__["seg_resnet34_8s_embedding"](2, 64, sd).eval()
        net_crop = ref.networks.__dict__["seg_resnet34_8s_embedding"](2, 64, sd).eval()
    np.random.seed(RNG_SEED)
    with torch.no_grad():
        out_label, refined = ref.test_dataset.test_sample(sample, net, net_crop)
        feat = net(sample["image_color"], None, sample["depth"])
    pos = sample_positions(7, 480 * 640, 1024)
    np.savez_compressed(os.path.join(HERE, "demo.npz"), out_label=out_label.numpy().astype(np.uint8),
                        refined=refined.numpy().astype(np.uint8), pos=pos,
                        embed=feat.permute(0, 2, 3, 1).reshape(-1, 64)[pos].numpy().astype(np.float32))
    print("demo: labels", np.unique(out_label.numpy()).tolist(), "refined", np.unique(refined.numpy()).tolist(), flush=True)


def make_modes(ref):
    """Other modalities: the reference's SEGNET built under cfg.INPUT / FUSION_TYPE of each mode, and its
    glue / test_sample with depth=None (COLOR input: no depth filter, ROI order by box area)."""
    import contextlib
    import io
    td = ref.test_dataset
    out = {}
    saved = (ref.cfg.INPUT, ref.cfg.TRAIN.FUSION_TYPE)
    try:
        for mode, m in MODES.items():
            ref.cfg.INPUT, ref.cfg.TRAIN.FUSION_TYPE = m["INPUT"], m["FUSION"]
            for name, c in MODE_BACKBONE_CASES.items():
                branches = ("fcn", "fcn_depth") if mode == "RGBD_CAT" else ("fcn",)
                sd = {k: torch.from_numpy(np.asarray(v))
                      for k, v in synth.synthetic_state_dict(c["wseed"], branches=branches, in_channels=m["in_channels"]).items()}
                with contextlib.redirect_stdout(io.StringIO()):
                    net = ref.networks.__dict__[m["factory"]](2, 64, sd).eval()
                assert hasattr(net, "fcn_depth") == (mode == "RGBD_CAT")
                got = net.state_dict()
                assert all(torch.equal(got[k], v) for k, v in sd.items()), "update_model dropped an entry"
                frames = [synth.rgbd_frame(s_, c["H"], c["W"], 4) for s_ in c["frames"]]
                img = torch.from_numpy(np.concatenate([f["image_color"] for f in frames]))
                dep = torch.from_numpy(np.concatenate([f["depth"] for f in frames]))
                with torch.no_grad():
                    feat = net(img, None, None if mode == "COLOR" else dep)
                B = feat.shape[0]
                flat = feat.permute(0, 2, 3, 1).reshape(B, -1, feat.shape[1]).numpy()
                key = f"{mode}/{name}"
                if c["samples"]:
                    pos = sample_positions(99, flat.shape[1], c["samples"])
                    out[key + "/pos"] = pos
                    out[key + "/embed"] = flat[:, pos].astype(np.float32)
                else:
                    out[key + "/embed"] = flat.astype(np.float32)
                print(key, feat.shape, "norm check", float(feat.norm(dim=1).mean()), flush=True)

        # 128-d clustering, the reference's mean_shift_smart_init stage by stage (as make_meanshift does for 64-d)
        ms = ref.mean_shift
        for name, c in WIDE_MEANSHIFT_CASES.items():
            X, _ = synth.embedding_field(c["seed"], c["H"], c["W"], 128, c["num_objects"], c["noise"])
            Xt = torch.from_numpy(X)
            np.random.seed(RNG_SEED)
            labels, idx = ms.mean_shift_smart_init(Xt, KAPPA, num_seeds=c["m"], max_iters=c["iters"], metric="cosine")
            np.random.seed(RNG_SEED)
            seeds, idx2 = ms.select_smart_seeds(Xt, c["m"], return_selected_indices=True, metric="cosine")
            assert torch.equal(idx, idx2)
            Z = ms.seed_hill_climbing_ball(Xt, seeds, KAPPA, max_iters=c["iters"], metric="cosine")
            seed_labels = ms.connected_components(Z, 2 * ref.cfg.TRAIN.EMBEDDING_ALPHA, metric="cosine")
            out[f"WIDE/{name}/labels"] = labels.numpy().astype(np.uint8)
            out[f"WIDE/{name}/indices"] = idx.numpy().astype(np.int32)
            out[f"WIDE/{name}/Z"] = Z.numpy().astype(np.float32)
            out[f"WIDE/{name}/seed_labels"] = seed_labels.numpy().astype(np.int32)
            print("WIDE", name, "clusters:", np.unique(labels.numpy()).tolist(), flush=True)

        ref.cfg.INPUT, ref.cfg.TRAIN.FUSION_TYPE = "COLOR", "add"
        for name in MODE_GLUE_CASES:
            c = GLUE_CASES[name]
            img, lab, depth, gt = glue_inputs(c)
            rgb_c, mask_c, rois, depth_c = td.crop_rois(img, lab.clone(), None)        # no depth filter either (:250)
            assert depth_c is None
            K = rgb_c.shape[0]
            key = "COLOR/glue_" + name
            out[key + "/rois"] = rois.numpy().astype(np.int32)
            out[key + "/mask_crops"] = np.packbits(mask_c.numpy().astype(np.uint8), axis=None)
            out[key + "/rgb_crops_sum"] = rgb_c.double().sum(dim=(1, 2, 3)).numpy()
            labels_c = crop_cluster_labels(c, gt, rois)
            refined, labels_c2 = td.match_label_crop(lab, labels_c.clone(), mask_c, rois, None)
            out[key + "/refined"] = refined.numpy().astype(np.uint8)
            out[key + "/labels_crop_out"] = labels_c2.numpy().astype(np.int8)
            print(key, "K =", K, flush=True)

        for name, c in MODE_E2E_CASES.items():
            fr = synth.rgbd_frame(c["seed"], 480, 640, c["objects"])
            sample = dict(image_color=torch.from_numpy(fr["image_color"]))           # COLOR samples carry no depth
            net = lambda img, label, depth, c=c: e2e_stub_features(c["seed"], 480, 640, c["objects"] + 2)
            net_crop = lambda rgb, label, depth, c=c: torch.cat(
                [e2e_stub_features(1000 + 10 * c["seed"] + k, 224, 224, 2 + k % 3) for k in range(rgb.shape[0])])
            np.random.seed(RNG_SEED)
            out_label, refined = td.test_sample(sample, net, net_crop)
            out[f"COLOR/{name}/out_label"] = out_label.numpy().astype(np.uint8)
            out[f"COLOR/{name}/refined"] = refined.numpy().astype(np.uint8)
            print("COLOR", name, "labels", np.unique(out_label.numpy()).tolist(), "refined", np.unique(refined.numpy()).tolist(), flush=True)
    finally:
        ref.cfg.INPUT, ref.cfg.TRAIN.FUSION_TYPE = saved
    np.savez_compressed(os.path.join(HERE, "modes.npz"), **out)


def make_evaluation(ref):
    """Reference metrics: munkres.py as is; evaluation.py with (a) np.bool aliased (removed from numpy 1.24) and
    (b) boundary_overlap — which needs cv2.dilate + skimage.disk, absent here — replaced by a stub returning
    (0, 0), so only the overlap metrics / counts of multilabel_metrics are recorded from it.  seg2bmap is recorded
    separately (it is plain numpy)."""
    import importlib
    if not hasattr(np, "bool"):
        np.bool = bool
    ev = importlib.import_module("utils.evaluation")
    mk = importlib.import_module("utils.munkres")
    out = {}
    for name, cost in munkres_cases().items():
        res = mk.Munkres().compute(cost.copy())
        out[f"munkres/{name}/cost"] = cost
        out[f"munkres/{name}/assign"] = np.asarray(res, dtype=np.int32).reshape(-1, 2)
        print("munkres", name, res, flush=True)
    real = ev.boundary_overlap
    ev.boundary_overlap = lambda p, g, bound_th=0.003: (0, 0)
    try:
        keys = ["Objects F-measure", "Objects Precision", "Objects Recall", "obj_detected", "obj_detected_075", "obj_gt",
                "obj_detected_075_percentage"]
        for name, c in EVAL_CASES.items():
            pred, gt = eval_pair(c)
            m = ev.multilabel_metrics(pred.copy(), gt.copy())
            out[f"metrics/{name}"] = np.array([float(m[k]) for k in keys], dtype=np.float64)
            labs = [l for l in np.unique(pred) if l != 0][:3]
            for l in labs:
                out[f"bmap/{name}/{int(l)}"] = np.packbits(ev.seg2bmap(pred == l).astype(np.uint8), axis=None)
            print("metrics", name, {k: round(float(m[k]), 4) for k in keys}, flush=True)
        out["metrics/keys"] = np.array(keys)
    finally:
        ev.boundary_overlap = real
    np.savez_compressed(os.path.join(HERE, "evaluation.npz"), **out)


def _install_cv2_reader():
    """cv2 is not installed here; the reference's read_sample only needs cv2.imread (tools/test_images.py:108,112).
    The stand-in decodes with PIL and returns what cv2 would: BGR uint8 [H,W,3] for a colour read, the raw uint16
    array for IMREAD_ANYDEPTH."""
    from PIL import Image
    cv2 = sys.modules["cv2"]
    cv2.IMREAD_ANYDEPTH = 2

    def imread(path, flags=1):
        im = Image.open(path)
        if flags == cv2.IMREAD_ANYDEPTH:
            return np.asarray(im).copy()
        return np.asarray(im.convert("RGB"))[:, :, ::-1].copy()
    cv2.imread = imread
    cv2.imwrite = lambda *a, **k: True


def make_prep(ref):
    """a15 / f-2: the reference's OWN read_sample + compute_xyz (tools/test_images.py:96-135), imported from
    /root/reference/tools with cv2.imread served by PIL, on (1) the demo pair and (2) a synthetic pair written to a
    temporary directory as 8-bit / 16-bit PNGs.  Recorded: SHA-256 of the raw float32 bytes of both tensors, sums and
    4096 sampled values for the demo pair; the complete tensors for the small synthetic pair."""
    import hashlib
    import importlib
    import json
    import tempfile
    from PIL import Image
    _install_cv2_reader()
    tools = os.path.join(ref_harness.REFERENCE_ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    ti = importlib.import_module("test_images")           # module level only defines functions; main() is guarded
    out = {}
    demo = os.path.join(HERE, "demo")
    cam = json.load(open(os.path.join(demo, "camera_params.json")))
    s = ti.read_sample(os.path.join(demo, "000002-color.png"), os.path.join(demo, "000002-depth.png"), cam)
    pos = sample_positions(11, 3 * 480 * 640, 4096)
    for key in ("image_color", "depth"):
        a = np.ascontiguousarray(s[key].numpy())
        assert a.dtype == np.float32 and a.shape == (1, 3, 480, 640), (a.dtype, a.shape)
        out[f"demo/{key}/sha256"] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)
        out[f"demo/{key}/sum"] = np.array(a.astype(np.float64).sum())
        out[f"demo/{key}/samples"] = a.reshape(-1)[pos]
    out["demo/pos"] = pos
    im, dep = prep_synthetic_arrays()
    with tempfile.TemporaryDirectory() as td:
        fc, fd = os.path.join(td, "c.png"), os.path.join(td, "d.png")
        Image.fromarray(im[:, :, ::-1].copy()).save(fc)           # file is RGB; cv2.imread hands back BGR = `im`
        Image.fromarray(dep).save(fd)                               # 16-bit greyscale PNG
        assert np.array_equal(sys.modules["cv2"].imread(fc), im)
        assert np.array_equal(sys.modules["cv2"].imread(fd, 2), dep)
        s2 = ti.read_sample(fc, fd, PREP_SYNTH["camera"])
    out["synth/image_color"] = s2["image_color"].numpy()
    out["synth/depth"] = s2["depth"].numpy()
    print("prep: demo image sum %.6f depth sum %.6f; synth shapes %s %s" % (
        float(out["demo/image_color/sum"]), float(out["demo/depth/sum"]), out["synth/image_color"].shape, out["synth/depth"].shape), flush=True)
    np.savez_compressed(os.path.join(HERE, "prep.npz"), **out)


def make_npy(ref):
    """The reference's OWN tools/test_npy.py read_sample (:105-144) on the two .npy layouts (tests/golden/cases.py
    npy_frames), written to a temporary directory."""
    import importlib
    import tempfile
    _install_cv2_reader()
    tools = os.path.join(ref_harness.REFERENCE_ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    tn = importlib.import_module("test_npy")
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, d in npy_frames().items():
            f = os.path.join(td, name + ".npy")
            np.save(f, d, allow_pickle=True)
            s = tn.read_sample(f, PREP_SYNTH["camera"])
            for key in ("image_color", "depth"):
                a = s[key].numpy()
                assert a.dtype == np.float32, (name, key, a.dtype)
                out[f"{name}/{key}"] = a
            print(name, {k: tuple(v.shape) for k, v in s.items()}, flush=True)
    np.savez_compressed(os.path.join(HERE, "npy.npz"), **out)


def make_segnet(ref):
    """a14: the reference's OWN test_segnet (lib/fcn/test_dataset.py:271-381) on three samples per dataset name, stub
    networks, cfg.TEST.VISUALIZE False (so it writes the .mat files), boundary_overlap stubbed to (0, 0) like
    make_evaluation (cv2.dilate / skimage are absent).  Recorded per run: the labels / labels_refined / filename of
    every .mat, the per-frame metrics before and after refinement, and the printed report."""
    import contextlib
    import importlib
    import io
    import tempfile
    import scipy.io
    if not hasattr(np, "bool"):
        np.bool = bool
    ev = importlib.import_module("utils.evaluation")
    td = ref.test_dataset
    real = ev.boundary_overlap
    ev.boundary_overlap = lambda p, g, bound_th=0.003: (0, 0)
    out = {}
    try:
        for tag, run in SEGNET_RUNS.items():
            samples = segnet_samples(run)
            net, net_crop = segnet_stub_networks(run)
            metrics_log = []
            real_mm = td.multilabel_metrics

            def spy(prediction, gt, _real=real_mm, _log=metrics_log):
                m = _real(prediction, gt)
                _log.append(dict(m))
                return m
            td.multilabel_metrics = spy
            buf = io.StringIO()
            with tempfile.TemporaryDirectory() as tmp:
                np.random.seed(RNG_SEED)
                try:
                    with contextlib.redirect_stdout(buf):
                        td.test_segnet(SegnetLoader(run["name"], samples), net, tmp, net_crop)
                finally:
                    td.multilabel_metrics = real_mm
                for i in range(len(samples)):
                    mat = scipy.io.loadmat(os.path.join(tmp, "%06d.mat" % i))
                    out[f"{tag}/{i}/labels"] = mat["labels"].astype(np.uint8)
                    out[f"{tag}/{i}/labels_refined"] = mat["labels_refined"].astype(np.uint8)
                    out[f"{tag}/{i}/filename"] = np.array(str(np.asarray(mat["filename"]).reshape(-1)[0]))
            assert len(metrics_log) == 2 * len(samples)
            for i in range(len(samples)):
                out[f"{tag}/{i}/metrics"] = np.array([float(metrics_log[2 * i][k]) for k in SEGNET_METRIC_KEYS])
                out[f"{tag}/{i}/metrics_refined"] = np.array([float(metrics_log[2 * i + 1][k]) for k in SEGNET_METRIC_KEYS])
            report = [ln for ln in buf.getvalue().splitlines() if "batch time" not in ln and not ln.endswith(".mat")]
            out[f"{tag}/report"] = np.array("\n".join(report))
            print(tag, "frames", len(samples), "segments", [int(out[f"{tag}/{i}/labels"].max()) for i in range(len(samples))],
                  "refined", [int(out[f"{tag}/{i}/labels_refined"].max()) for i in range(len(samples))], flush=True)
    finally:
        ev.boundary_overlap = real
    np.savez_compressed(os.path.join(HERE, "segnet.npz"), **out)


def main():
    assert ref_harness.available(), "reference tree not present: golden vectors can only be made in the build container"
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ref = ref_harness.load_reference()
    torch.manual_seed(0)
    if what in ("meanshift", "all"):
        make_meanshift(ref)
    if what in ("seedcont", "all"):
        make_seedcont(ref)
    if what in ("npy", "all"):
        make_npy(ref)
    if what in ("backbone", "all"):
        make_backbone(ref)
    if what in ("glue", "all"):
        make_glue(ref)
    if what in ("e2e", "all"):
        make_e2e(ref)
    if what in ("demo", "all"):
        make_demo(ref)
    if what in ("modes", "all"):
        make_modes(ref)
    if what in ("evaluation", "all"):
        make_evaluation(ref)
    if what in ("prep", "all"):
        make_prep(ref)
    if what in ("segnet", "all"):
        make_segnet(ref)


if __name__ == "__main__":
    main()
