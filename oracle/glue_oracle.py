"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the reference's two-stage
glue: depth-coverage filter, ROI crop/resize, crop-label matching and paste-back, and the
test_sample orchestration.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinned against the reference's own functions by tests/golden/glue.npz.

Reference: /root/reference/lib/fcn/test_dataset.py (cited per function), lib/utils/mask.py:180-187.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import mean_shift_oracle as MS

CROP_SIZE = 224          # cfg.TRAIN.SYN_CROP_SIZE (config.py:129)
PAD_FRACTION = 0.25      # test_dataset.py:66


def filter_labels_depth(labels: torch.Tensor, depth: torch.Tensor, threshold: float) -> torch.Tensor:
    """test_dataset.py:183-198.  labels [B,H,W] float, depth [B,3,H,W]; a non-zero label whose
    fraction of pixels with z > 0 is below `threshold` becomes 0."""
    out = labels.clone()
    for i in range(labels.shape[0]):
        lab = labels[i]
        for mid in torch.unique(lab):
            if mid == 0:
                continue
            sel = lab == mid
            frac = torch.sum(depth[i, 2][sel] > 0).float() / torch.sum(sel.float())
            if frac < threshold:
                out[i][sel] = 0
    return out


def roi_boxes(label_map: torch.Tensor):
    """Padded, clamped boxes for every non-zero label of a [H,W] map, ascending label order
    (test_dataset.py:68-93, mask.py:180-187).  Returns (ids list, boxes int64 [K,4] x0,y0,x1,y1)."""
    H, W = label_map.shape
    ids = [int(v) for v in torch.unique(label_map).tolist() if v != 0]
    boxes = []
    for mid in ids:
        ys, xs = torch.nonzero(label_map == mid, as_tuple=True)
        x0, x1, y0, y1 = int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())
        # torch.round is round-half-to-even (:83-84)
        px = int(torch.round(torch.tensor(float(x1 - x0)) * PAD_FRACTION).item())
        py = int(torch.round(torch.tensor(float(y1 - y0)) * PAD_FRACTION).item())
        boxes.append([max(x0 - px, 0), max(y0 - py, 0), min(x1 + px, W - 1), min(y1 + py, H - 1)])
    return ids, torch.tensor(boxes, dtype=torch.int64).reshape(-1, 4)


def crop_rois(rgb: torch.Tensor, initial_masks: torch.Tensor, depth: torch.Tensor, crop_size: int = CROP_SIZE):
    """test_dataset.py:62-112.  Returns rgb_crops [K,3,S,S], mask_crops [K,S,S], rois [K,4] float,
    depth_crops [K,3,S,S]."""
    ids, boxes = roi_boxes(initial_masks[0])
    K = len(ids)
    S = crop_size
    rgb_crops = torch.zeros((K, 3, S, S))
    depth_crops = torch.zeros((K, 3, S, S)) if depth is not None else None        # :73-76
    mask_crops = torch.zeros((K, S, S))
    for k, mid in enumerate(ids):
        x0, y0, x1, y1 = [int(v) for v in boxes[k]]
        m = (initial_masks[0] == mid).float()[y0:y1 + 1, x0:x1 + 1]
        rgb_crops[k] = F.interpolate(rgb[0:1, :, y0:y1 + 1, x0:x1 + 1], size=(S, S), mode="bilinear", align_corners=True)[0]
        if depth is not None:
            depth_crops[k] = F.interpolate(depth[0:1, :, y0:y1 + 1, x0:x1 + 1], size=(S, S), mode="bilinear", align_corners=True)[0]
        mask_crops[k] = F.interpolate(m[None, None], size=(S, S), mode="nearest")[0, 0]
    return rgb_crops, mask_crops, boxes.float(), depth_crops


def match_label_crop(initial_masks: torch.Tensor, labels_crop: torch.Tensor, out_label_crop: torch.Tensor,
                     rois: torch.Tensor, depth_crop: torch.Tensor):
    """test_dataset.py:116-179.  labels_crop [K,S,S] float (modified: rejected clusters -> -1),
    returns (refined_masks like initial_masks, labels_crop)."""
    labels_crop = labels_crop.clone()
    K = labels_crop.shape[0]
    for i in range(K):                                         # :118-125 overlap test
        for mid in torch.unique(labels_crop[i]):
            sel = labels_crop[i] == mid
            frac = torch.sum(sel.float() * out_label_crop[i]) / torch.sum(sel.float())
            if frac < 0.5:
                labels_crop[i][sel] = -1
    keys = []
    for i in range(K):                                         # :129-138 mean depth of kept pixels
        if depth_crop is None:                                 # :139-146 no depth: box area, large first
            keys.append((i, (rois[i, 3] - rois[i, 1] + 1) * (rois[i, 2] - rois[i, 0] + 1)))
            continue
        kept = labels_crop[i] > -1
        z = depth_crop[i, 2][kept] if torch.sum(kept) > 0 else depth_crop[i, 2]
        keys.append((i, torch.mean(z[z > 0])))
    order = [i for i, _ in sorted(keys, key=lambda kv: kv[1], reverse=True)]   # :150-151 far first
    refined = torch.zeros_like(initial_masks).float()
    count = 0
    for i in order:                                            # :156-177
        relabeled = torch.zeros_like(labels_crop[i])
        for mid in torch.unique(labels_crop[i]):
            if mid == -1:
                continue
            count += 1
            relabeled[labels_crop[i] == mid] = count
        x0, y0, x1, y1 = [int(v) for v in rois[i].tolist()]
        h, w = y1 - y0 + 1, x1 - x0 + 1
        back = F.interpolate(relabeled[None, None].float(), size=(h, w), mode="nearest")[0, 0]
        window = refined[0, y0:y1 + 1, x0:x1 + 1]
        window[back != 0] = back[back != 0]
    return refined, labels_crop


def clustering_features(features: torch.Tensor, first_indices, num_seeds: int = 100, kappa: float = 20.0,
                        max_iters: int = 10, epsilon: float = 0.04):
    """test_dataset.py:44-59: per batch item cluster X = features[j].view(C,-1).T.
    `first_indices`: the np.random.randint draws, one per item."""
    B, C, H, W = features.shape
    out = torch.zeros((B, H, W))
    for j in range(B):
        X = torch.transpose(features[j].view(C, -1), 0, 1)     # :54-55: the STRIDED transpose, as the reference multiplies it
        lab, _ = MS.mean_shift_smart_init(X, kappa, num_seeds, max_iters, first_index=int(first_indices[j]), epsilon=epsilon)
        out[j] = lab.view(H, W).float()
    return out


def test_sample(image: torch.Tensor, depth: torch.Tensor, network, network_crop, rng: np.random.RandomState):
    """test_dataset.py:232-267 with `network(image, label, depth)` callables returning [B,64,H,W].
    RNG: one np.random.randint(0, n) per clustered field, in call order (mean_shift.py:155)."""
    features = network(image, None, depth)
    n = features.shape[2] * features.shape[3]
    out_label = clustering_features(features, [rng.randint(0, n) for _ in range(features.shape[0])])
    if depth is not None:                                      # :250
        out_label = filter_labels_depth(out_label, depth, 0.8)
    refined = None
    if network_crop is not None:
        rgb_c, mask_c, rois, depth_c = crop_rois(image, out_label.clone(), depth)
        if rgb_c.shape[0] > 0:
            f2 = network_crop(rgb_c, mask_c, depth_c)
            n2 = f2.shape[2] * f2.shape[3]
            labels_c = clustering_features(f2, [rng.randint(0, n2) for _ in range(f2.shape[0])])
            refined, _ = match_label_crop(out_label, labels_c, mask_c, rois, depth_c)
    return out_label, refined


def test_segnet(test_loader, network, network_crop, rng: np.random.RandomState):
    """test_dataset.py:271-381, the per-frame body only (the metrics are restated in evaluation_oracle): cluster,
    depth-coverage filter with the per-dataset threshold (:299-305: 0.5 for 'ocid', 0.8 for 'osd', none otherwise),
    refine.  Returns [(prediction [H,W] float32, prediction_refined [H,W] float32)], what the reference stores in
    the .mat files (:334-340; without a refined map the stage-1 prediction is stored twice, :324-327)."""
    name = test_loader.dataset.name
    out = []
    for sample in test_loader:
        image, depth = sample["image_color"], sample.get("depth")
        features = network(image, sample["label"], depth)
        n = features.shape[2] * features.shape[3]
        out_label = clustering_features(features, [rng.randint(0, n) for _ in range(features.shape[0])])
        if "ocid" in name and depth is not None:
            out_label = filter_labels_depth(out_label, depth, 0.5)
        if "osd" in name and depth is not None:
            out_label = filter_labels_depth(out_label, depth, 0.8)
        refined = None
        if network_crop is not None:
            rgb_c, mask_c, rois, depth_c = crop_rois(image, out_label.clone(), depth)
            if rgb_c.shape[0] > 0:
                f2 = network_crop(rgb_c, mask_c, depth_c)
                n2 = f2.shape[2] * f2.shape[3]
                labels_c = clustering_features(f2, [rng.randint(0, n2) for _ in range(f2.shape[0])])
                refined, _ = match_label_crop(out_label, labels_c, mask_c, rois, depth_c)
        prediction = out_label.squeeze().numpy()
        out.append((prediction, refined.squeeze().numpy() if refined is not None else prediction.copy()))
    return out


def cpython_sort_small(keys):
    """list.sort(reverse=True) of range(len(keys)) by keys[i] for n < 64, restated from CPython's Objects/listobject.c
    (reverse the list, count_run, binary insertion sort with `<` only, reverse again; minrun = n below 64) — the algorithm
    csrc/roi.hip::cpython_sort_small runs on the device when a ROI sort key is NaN (lib/fcn/test_dataset.py:135 can yield one
    and :148 sorts with Python's own sort, whose result for NaN keys is whatever this algorithm does).  Test infrastructure:
    tests/test_oracle_glue.py pins it to Python's sorted() here; the GPU test pins the kernel to Python's sorted() there."""
    n = len(keys)
    assert n < 64
    a = [(keys[i], i) for i in range(n)]
    lt = lambda x, y: bool(x[0] < y[0])

    def rev(lo, hi):
        a[lo:hi] = a[lo:hi][::-1]
    rev(0, n)
    if n >= 2:
        run, desc = 2, lt(a[1], a[0])
        lo = 2
        while lo < n:
            if desc != lt(a[lo], a[lo - 1]):
                break
            lo += 1
            run += 1
        if desc:
            rev(0, run)
        for start in range(run, n):
            l, r, pivot = 0, start, a[start]
            while l < r:
                p = l + ((r - l) >> 1)
                if lt(pivot, a[p]):
                    r = p
                else:
                    l = p + 1
            a[l + 1:start + 1] = a[l:start]
            a[l] = pivot
    rev(0, n)
    return [i for _, i in a]
