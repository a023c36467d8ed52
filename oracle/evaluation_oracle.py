"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the reference's segmentation metrics
(lib/utils/evaluation.py: seg2bmap :15-73, boundary_overlap :75-107, multilabel_metrics :109-257) in numpy.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pinning (tests/golden/evaluation.npz, made by tests/golden/make_golden.py evaluation):
  * seg2bmap, the overlap metrics, the detection counts and the Hungarian assignment are pinned against the
    reference's own code run in the build container (its munkres.py imports as is; evaluation.py needs the alias
    np.bool = bool on numpy >= 1.24);
  * PARITY UNPINNED for the dilation step of boundary_overlap only: it calls cv2.dilate with skimage's disk(), and
    neither OpenCV nor scikit-image exists in this image.  The restatement below uses their documented semantics
    (disk: x^2 + y^2 <= r^2; dilate: maximum over the footprint, out-of-image neighbours ignored).
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage

BACKGROUND_LABEL = 0
OBJECTS_LABEL = 1


def seg2bmap(seg):
    """evaluation.py:15-73 for width/height = the mask's own size."""
    seg = np.asarray(seg).astype(bool)
    e = np.zeros_like(seg)
    s = np.zeros_like(seg)
    se = np.zeros_like(seg)
    e[:, :-1] = seg[:, 1:]
    s[:-1, :] = seg[1:, :]
    se[:-1, :-1] = seg[1:, 1:]
    b = seg ^ e | seg ^ s | seg ^ se
    b[-1, :] = seg[-1, :] ^ e[-1, :]
    b[:, -1] = seg[:, -1] ^ s[:, -1]
    b[-1, -1] = 0
    return b


def disk(radius):
    r = int(radius)
    y, x = np.mgrid[-r:r + 1, -r:r + 1]
    return (x * x + y * y <= r * r).astype(np.uint8)


def boundary_overlap(predicted_mask, gt_mask, bound_th=0.003):
    """evaluation.py:75-107 -> (precision true positives, recall true positives)."""
    bound_pix = bound_th if bound_th >= 1 else np.ceil(bound_th * np.linalg.norm(predicted_mask.shape))
    fg_boundary = seg2bmap(predicted_mask)
    gt_boundary = seg2bmap(gt_mask)
    fp = disk(bound_pix).astype(bool)
    gt_dil = ndimage.binary_dilation(gt_boundary, structure=fp)
    fg_dil = ndimage.binary_dilation(fg_boundary, structure=fp)
    return np.sum(np.logical_and(fg_boundary, gt_dil)), np.sum(np.logical_and(gt_boundary, fg_dil))


def pair_tables(prediction, gt):
    """The integer tables the HIP kernels produce (uoc_eval_pair_stats), by brute force over label pairs."""
    prediction, gt = np.asarray(prediction).astype(np.int64), np.asarray(gt).astype(np.int64)
    cont = np.zeros((128, 128), np.int64)
    np.add.at(cont, (gt.reshape(-1), prediction.reshape(-1)), 1)
    prec_tp = np.zeros((128, 128), np.int64)
    rec_tp = np.zeros((128, 128), np.int64)
    bnd_pred = np.zeros(128, np.int64)
    bnd_gt = np.zeros(128, np.int64)
    lp = [l for l in np.unique(prediction) if l != 0]
    lg = [l for l in np.unique(gt) if l != 0]
    for j in lp:
        bnd_pred[j] = seg2bmap(prediction == j).sum()
    for i in lg:
        bnd_gt[i] = seg2bmap(gt == i).sum()
        for j in lp:
            prec_tp[i, j], rec_tp[i, j] = boundary_overlap(prediction == j, gt == i)
    return dict(cont=cont, prec_tp=prec_tp, rec_tp=rec_tp, bnd_pred=bnd_pred, bnd_gt=bnd_gt)
