"""Shared definition of the clustering golden cases (inputs are regenerated from seeds by
unseenobjectclustering_amd.synth; only reference OUTPUTS are stored in the .npz)."""

# name -> dict(seed, H, W, num_objects, noise, m, iters)
MEANSHIFT_CASES = {
    "tiny_60x80":    dict(seed=11, H=60,  W=80,  num_objects=4, noise=0.05, m=100, iters=10),
    "ragged_37x53":  dict(seed=12, H=37,  W=53,  num_objects=3, noise=0.05, m=100, iters=10),
    "fewseeds_m20":  dict(seed=13, H=96,  W=128, num_objects=5, noise=0.05, m=20,  iters=10),
    "oneiter":       dict(seed=14, H=96,  W=128, num_objects=5, noise=0.05, m=50,  iters=1),
    "crop_224_a":    dict(seed=2,  H=224, W=224, num_objects=3, noise=0.05, m=100, iters=10),
    "crop_224_b":    dict(seed=3,  H=224, W=224, num_objects=6, noise=0.08, m=100, iters=10),
    "full_480x640_a": dict(seed=1, H=480, W=640, num_objects=7, noise=0.05, m=100, iters=10),
    "full_480x640_b": dict(seed=5, H=480, W=640, num_objects=5, noise=0.05, m=100, iters=10),
    "full_480x640_c": dict(seed=9, H=480, W=640, num_objects=8, noise=0.10, m=100, iters=10),
}
KAPPA = 20.0
EPSILON = 0.04
RNG_SEED = 3

# select_smart_seeds(init_seeds=..., num_init_seeds=k) (mean_shift.py:142-170).  init: 'rows' = the first k seeds of a
# plain selection on the same field (a continued selection), 'free' = k unit vectors that are not rows of X,
# 'empty' = an init matrix with num_init_seeds = 0 (the first seed is then drawn from the RNG as usual)
SEED_CONTINUATION_CASES = {
    "cont_rows_60x80":   dict(seed=21, H=60,  W=80,  num_objects=4, noise=0.05, m=30,  k=6,  init="rows"),
    "cont_free_96x128":  dict(seed=22, H=96,  W=128, num_objects=5, noise=0.05, m=40,  k=5,  init="free"),
    "cont_free_224":     dict(seed=23, H=224, W=224, num_objects=6, noise=0.08, m=100, k=17, init="free"),
    "cont_empty_37x53":  dict(seed=24, H=37,  W=53,  num_objects=3, noise=0.05, m=20,  k=0,  init="empty"),
    "cont_all_given":    dict(seed=25, H=37,  W=53,  num_objects=3, noise=0.05, m=8,   k=8,  init="free"),
}


# ---------------------------------------------------------------------------------------------
# Shared input builders (used by make_golden.py in the build container AND by the tests).
# ---------------------------------------------------------------------------------------------
import numpy as np
import torch

from unseenobjectclustering_amd import synth


def continuation_inputs(c):
    """X [n,64] and the init_seeds [m,64] matrix (rows >= k are NaN-free filler the selection overwrites) of a
    SEED_CONTINUATION_CASES entry; for 'rows' the caller fills the first k rows from a plain selection."""
    X, _ = synth.embedding_field(c["seed"], c["H"], c["W"], 64, c["num_objects"], c["noise"])
    rng = np.random.default_rng(1000 + c["seed"])
    init = rng.standard_normal((c["m"], 64)).astype(np.float32)
    init /= np.linalg.norm(init, axis=1, keepdims=True)
    return X, init


def sample_positions(seed, n, count):
    return np.sort(np.random.default_rng(seed).choice(n, size=min(count, n), replace=False)).astype(np.int64)


BACKBONE_CASES = {
    # name -> (weight seed, frame seeds, H, W, num sampled pixels (0 = keep everything))
    "tiny_64x64":   dict(wseed=1, frames=[7], H=64, W=64, samples=0),
    "odd_72x104":   dict(wseed=2, frames=[8], H=72, W=104, samples=1024),
    "crops_224":    dict(wseed=2, frames=[4, 5], H=224, W=224, samples=1024),
    "full_480x640": dict(wseed=1, frames=[1], H=480, W=640, samples=2048),
}


GLUE_CASES = {
    # name -> (frame seed, H, W, objects, tweak)
    "normal_5":     dict(seed=31, H=480, W=640, objects=5, tweak=None),
    "border_8":     dict(seed=32, H=480, W=640, objects=8, tweak="border"),
    "zero_depth":   dict(seed=33, H=480, W=640, objects=4, tweak="zero_depth"),
    "no_objects":   dict(seed=34, H=240, W=320, objects=0, tweak="empty"),
    "small_ragged": dict(seed=35, H=123, W=157, objects=3, tweak=None),
}


def glue_inputs(c):
    """Label map (float, label 0 = background+table as after the largest-cluster swap), image, xyz."""
    fr = synth.rgbd_frame(c["seed"], c["H"], c["W"], c["objects"])
    lab = fr["label"].copy()
    lab = np.where(lab <= 1, 0, lab - 1).astype(np.float32)      # objects 1..K
    depth = fr["depth"].copy()
    if c["tweak"] == "border":
        lab[:40, :60] = lab.max() + 1                               # object touching the image corner
        lab[-1, :] = lab.max() + 1                                  # one-pixel-high object on the last row
    if c["tweak"] == "zero_depth" and lab.max() >= 2:
        depth[0, 2][lab == 2] = 0.0                                 # object 2 has no valid depth -> filtered
        half = (lab == 1) & (np.arange(lab.shape[1])[None, :] % 4 != 0)
        depth[0, 2][half] = 0.0                                     # object 1: 25 % coverage -> filtered
    if c["tweak"] == "empty":
        lab[:] = 0
    return torch.from_numpy(fr["image_color"]), torch.from_numpy(lab)[None], torch.from_numpy(depth), fr["label"]


def crop_cluster_labels(c, gt, rois):
    """Synthetic stage-2 cluster maps: nearest crop of the generating label map, object split in two."""
    out = []
    for k in range(rois.shape[0]):
        x0, y0, x1, y1 = [int(v) for v in rois[k]]
        g = torch.from_numpy(gt[y0:y1 + 1, x0:x1 + 1].astype(np.float32))
        g = torch.nn.functional.interpolate(g[None, None], size=(224, 224), mode="nearest")[0, 0]
        g[:, 112:][g[:, 112:] == g.max()] += 3                      # split the top label into two clusters
        out.append(g)
    return torch.stack(out) if out else torch.zeros((0, 224, 224))


E2E_CASES = {"e2e_a": dict(seed=41, objects=5), "e2e_b": dict(seed=42, objects=3)}


def e2e_stub_features(seed, H, W, objects):
    X, _ = synth.embedding_field(seed, H, W, 64, objects, 0.05)
    return torch.from_numpy(X).view(H, W, 64).permute(2, 0, 1)[None].contiguous()


# ---------------------------------------------------------------------------------------------
# Other input modalities: cfg.INPUT 'COLOR' / 'DEPTH', RGBD with FUSION_TYPE 'early'.
# ---------------------------------------------------------------------------------------------
MODES = {
    # mode -> (cfg.INPUT, cfg.TRAIN.FUSION_TYPE, factory name, stem input channels)
    "COLOR":      dict(INPUT="COLOR", FUSION="add",   factory="seg_resnet34_8s_embedding",       in_channels=3),
    "DEPTH":      dict(INPUT="DEPTH", FUSION="add",   factory="seg_resnet34_8s_embedding",       in_channels=3),
    "RGBD_EARLY": dict(INPUT="RGBD",  FUSION="early", factory="seg_resnet34_8s_embedding_early", in_channels=6),
    "RGBD_CAT":   dict(INPUT="RGBD",  FUSION="cat",   factory="seg_resnet34_8s_embedding",       in_channels=3),
}
# 128-d clustering (the 'cat' embeddings): reference mean_shift_smart_init on 128-channel synthetic fields
WIDE_MEANSHIFT_CASES = {
    "wide_60x80":   dict(seed=21, H=60,  W=80,  num_objects=4, noise=0.05, m=100, iters=10),
    "wide_224":     dict(seed=22, H=224, W=224, num_objects=5, noise=0.06, m=100, iters=10),
    "wide_480x640": dict(seed=23, H=480, W=640, num_objects=6, noise=0.05, m=100, iters=10),
}
MODE_BACKBONE_CASES = {
    "tiny_64x64":  dict(wseed=5, frames=[7], H=64, W=64, samples=512),
    "odd_72x104":  dict(wseed=6, frames=[8, 9], H=72, W=104, samples=512),
}
MODE_GLUE_CASES = ["normal_5", "border_8", "small_ragged"]       # GLUE_CASES re-run without depth (COLOR input)
MODE_E2E_CASES = {"color_a": dict(seed=43, objects=4)}


# ---------------------------------------------------------------------------------------------
# Evaluation: multilabel_metrics on synthetic (prediction, ground truth) pairs.
# ---------------------------------------------------------------------------------------------
EVAL_CASES = {
    # name -> (frame seed, H, W, objects, perturbation of the prediction)
    "shifted":     dict(seed=51, H=240, W=320, objects=4, mode="shift"),
    "merged":      dict(seed=52, H=240, W=320, objects=5, mode="merge"),
    "split_extra": dict(seed=53, H=480, W=640, objects=3, mode="split"),
    "permuted":    dict(seed=54, H=123, W=157, objects=6, mode="permute"),
    "no_pred":     dict(seed=55, H=120, W=160, objects=3, mode="empty_pred"),
    "no_gt":       dict(seed=56, H=120, W=160, objects=3, mode="empty_gt"),
    "nothing":     dict(seed=57, H=120, W=160, objects=0, mode="empty_both"),
}


def eval_pair(c):
    """(prediction, gt) int64 [H,W] maps: gt = generated scene (0 background, 1 table, 2.. objects), prediction = a
    perturbed copy."""
    gt = synth.rgbd_frame(c["seed"], c["H"], c["W"], c["objects"])["label"].astype(np.int64)
    pred = gt.copy()
    mode = c["mode"]
    if mode == "shift":
        pred = np.roll(np.roll(gt, 3, axis=0), -2, axis=1)
    elif mode == "merge":
        pred[pred == pred.max()] = pred.max() - 1
        pred = np.roll(pred, 1, axis=1)
    elif mode == "split":
        top = pred.max()
        ys = np.nonzero((pred == top).any(axis=1))[0]
        pred[(pred == top) & (np.arange(pred.shape[0])[:, None] > ys.mean())] = top + 1
        pred[5:25, 5:45] = top + 2                                   # a false-positive object
    elif mode == "permute":
        perm = np.arange(pred.max() + 1)
        perm[1:] = np.roll(perm[1:], 2)
        pred = perm[pred]
        pred = np.roll(pred, 2, axis=0)
    elif mode == "empty_pred":
        pred[:] = 0
    elif mode == "empty_gt":
        gt = np.zeros_like(gt)
    elif mode == "empty_both":
        gt = np.zeros_like(gt)
        pred = np.zeros_like(gt)
    return pred, gt


def munkres_cases():
    """Cost matrices for the Hungarian matcher: square / rectangular, integer ties, floats, the F.max() - F form."""
    rng = np.random.default_rng(20240611)
    out = {"1x1": np.array([[3.0]]), "ties_4x4": rng.integers(0, 3, size=(4, 4)).astype(np.float64),
           "ties_6x6": rng.integers(0, 4, size=(6, 6)).astype(np.float64), "tall_5x3": rng.random((5, 3)),
           "wide_3x6": rng.random((3, 6)), "float_8x8": rng.random((8, 8)), "zeros_3x3": np.zeros((3, 3))}
    F = rng.random((7, 5)) * (rng.random((7, 5)) > 0.5)
    out["fmeasure_7x5"] = F.max() - F
    F = rng.random((4, 9)) * (rng.random((4, 9)) > 0.6)
    out["fmeasure_4x9"] = F.max() - F
    return out


# ---------------------------------------------------------------------------------------------
# Input preparation: the reference's own read_sample (tools/test_images.py:105-135) on the
# demo pair and on a synthetic pair that spans the uint8 / uint16 ranges at an odd size.
# ---------------------------------------------------------------------------------------------
PREP_SYNTH = dict(seed=5, H=37, W=53, camera=dict(fx=500.5, fy=499.25, x_offset=26.1, y_offset=18.7))


def npy_frames():
    """The two .npy layouts tools/test_npy.py reads (reference :107-123): a {'rgb', 'depth' mm} frame that needs
    camera_params.json, and a {'debug_info': ...} frame that carries float32 intrinsics and a depth image in metres."""
    im_bgr, dep = prep_synthetic_arrays()
    rgb = np.ascontiguousarray(im_bgr[:, :, ::-1])
    plain = {"rgb": rgb, "depth": dep}
    K = np.array([[500.5, 0, 26.1], [0, 499.25, 18.7], [0, 0, 1]], dtype=np.float32)
    debug = {"debug_info": {"rgb": rgb[::-1].copy(), "depth_image": (dep[::-1].astype(np.float32) / 1000.0), "intrinsics": K}}
    return {"plain": plain, "debug": debug}


def prep_synthetic_arrays():
    """(BGR uint8 [H,W,3], depth uint16 [H,W] millimetres): full value ranges, a few zero-depth holes."""
    c = PREP_SYNTH
    rng = np.random.default_rng(c["seed"])
    im = rng.integers(0, 256, size=(c["H"], c["W"], 3), dtype=np.uint8)
    dep = rng.integers(0, 65536, size=(c["H"], c["W"])).astype(np.uint16)
    dep[::7, ::5] = 0
    dep[0, 0], dep[0, 1] = 65535, 1
    return im, dep


# ---------------------------------------------------------------------------------------------
# test_segnet: the reference's dataset loop (lib/fcn/test_dataset.py:271-381) over three samples
# with stub networks, under an 'ocid...' and an 'osd...' dataset name (depth-coverage thresholds 0.5 / 0.8, :299-305).
# ---------------------------------------------------------------------------------------------
SEGNET_RUNS = {
    "ocid": dict(name="ocid_object_test", frames=[dict(seed=61, objects=4), dict(seed=62, objects=5), dict(seed=63, objects=3)]),
    "osd":  dict(name="osd_object_test",  frames=[dict(seed=64, objects=5), dict(seed=65, objects=3), dict(seed=66, objects=4)]),
}
SEGNET_H, SEGNET_W = 240, 320


class SegnetLoader:
    """What test_segnet needs from a DataLoader: len(), iteration over sample dicts, `.dataset.name`."""

    def __init__(self, name, samples):
        self.samples = samples
        self.dataset = type("Dataset", (), {"name": name})()

    def __len__(self):
        return len(self.samples)

    def __iter__(self):
        return iter(self.samples)


def segnet_samples(run):
    """Sample dicts like the reference's dataset classes yield them (batch 1): image_color, depth, label (the
    generating scene as ground truth, [1,H,W] float like ocid_object.py), filename.  Frame f's depth has a block of the
    object with the largest id knocked out (z = 0 on ~45 % of it), so the two coverage thresholds act differently."""
    out = []
    for k, f in enumerate(run["frames"]):
        fr = synth.rgbd_frame(f["seed"], SEGNET_H, SEGNET_W, f["objects"])
        depth = fr["depth"].copy()
        lab = fr["label"]
        ys, xs = np.nonzero(lab == lab.max())
        cut = ys < np.quantile(ys, 0.45)
        depth[0, :, ys[cut], xs[cut]] = 0.0
        out.append(dict(image_color=torch.from_numpy(fr["image_color"]), depth=torch.from_numpy(depth),
                        label=torch.from_numpy(lab.astype(np.float32))[None], filename="scene_%02d/%06d" % (k, f["seed"])))
    return out


def segnet_stub_networks(run):
    """(network, network_crop) callables returning fixed embedding fields: stage 1 by frame (looked up by the image's
    checksum-free order of calls), stage 2 by crop index — like the e2e fixtures."""
    frames = run["frames"]
    state = {"i": 0}

    def net(img, label, depth):
        f = frames[state["i"] % len(frames)]
        state["i"] += 1
        return e2e_stub_features(f["seed"], SEGNET_H, SEGNET_W, f["objects"] + 2)

    def net_crop(rgb, label, depth):
        i = (state["i"] - 1) % len(frames)
        return torch.cat([e2e_stub_features(2000 + 10 * frames[i]["seed"] + k, 224, 224, 2 + k % 3) for k in range(rgb.shape[0])])

    for fn in (net, net_crop):
        fn.eval = lambda: None
    return net, net_crop


SEGNET_METRIC_KEYS = ["Objects F-measure", "Objects Precision", "Objects Recall", "obj_detected", "obj_detected_075",
                      "obj_gt", "obj_detected_075_percentage"]
