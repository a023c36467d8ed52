"""Host-side mirror of the reference's lib/utils/evaluation.py `multilabel_metrics` (:109-257) and of the
Hungarian matcher it uses (lib/utils/munkres.py:244-623), backed by the evaluation kernels in csrc/eval.hip.

The per-pixel work — the gt x pred contingency table, the 1-pixel boundary maps of every mask (seg2bmap,
:15-73) and the dilated-boundary matches (boundary_overlap, :75-107) — runs on the GPU as integer tables
(uoc_eval_pair_stats); precision / recall / F-measure and the assignment are the reference's float64
arithmetic on those tables.  Same dictionary keys and edge cases as the reference.
"""
from __future__ import annotations

import ctypes
import sys

import numpy as np
import torch

from .. import _native

BACKGROUND_LABEL = 0
OBJECTS_LABEL = 1
MAX_LABELS = 128


class _Tables(ctypes.Structure):
    _fields_ = [("cont", ctypes.c_int32 * (128 * 128)), ("prec_tp", ctypes.c_int32 * (128 * 128)),
                ("rec_tp", ctypes.c_int32 * (128 * 128)), ("bnd_pred", ctypes.c_int32 * 128),
                ("bnd_gt", ctypes.c_int32 * 128), ("bad_label", ctypes.c_int32)]


_ws = {}


def _device():
    if not torch.cuda.is_available():
        raise _native.NativeError("no ROCm device visible: the evaluation kernels have no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def bound_pixels(shape, bound_th=0.003):
    """evaluation.py:88-89."""
    return bound_th if bound_th >= 1 else np.ceil(bound_th * np.linalg.norm(shape))


def pair_stats(prediction, gt, bound_th=0.003):
    """Integer tables for one (prediction, gt) pair of [H,W] label maps (numpy or torch, any device).
    Returns dict(cont, prec_tp, rec_tp [128,128] int64 indexed [gt, pred]; bnd_pred, bnd_gt [128])."""
    dev = prediction.device if torch.is_tensor(prediction) and prediction.is_cuda else _device()

    def as_dev(a):
        t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
        t = t.to(dev)
        if t.dtype != torch.int32:
            if t.is_floating_point():
                t = t.round()
            t = t.to(torch.int32)
        return t.contiguous()
    p, g = as_dev(prediction), as_dev(gt)
    if p.dim() != 2 or p.shape != g.shape:
        raise ValueError("prediction and gt must be [H,W] maps of the same shape")
    H, W = p.shape
    L = _native.lib()
    tables = torch.empty(ctypes.sizeof(_Tables), dtype=torch.uint8, device=dev)
    nbytes = L.uoc_eval_workspace_bytes(H, W)
    key = (dev.type, dev.index)
    if key not in _ws or _ws[key].numel() < nbytes:
        _ws[key] = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=dev)
    ws = _ws[key]
    radius = int(bound_pixels((H, W), bound_th))
    with torch.cuda.device(dev):
        rc = L.uoc_eval_pair_stats(_native.ptr(p), _native.ptr(g), H, W, radius, _native.ptr(tables), _native.ptr(ws),
                                   ws.numel(), _native.stream_ptr(dev))
    _native.check(rc, "uoc_eval_pair_stats")
    t = _Tables.from_buffer_copy(tables.cpu().numpy().tobytes())
    if t.bad_label:
        raise ValueError(f"label ids must be in [0, {MAX_LABELS})")
    as2 = lambda a: np.ctypeslib.as_array(a).astype(np.int64).reshape(128, 128)
    return dict(cont=as2(t.cont), prec_tp=as2(t.prec_tp), rec_tp=as2(t.rec_tp),
                bnd_pred=np.ctypeslib.as_array(t.bnd_pred).astype(np.int64),
                bnd_gt=np.ctypeslib.as_array(t.bnd_gt).astype(np.int64))


# ---------------------------------------------------------------------------------------------
# Hungarian assignment with the step structure AND the scan orders of lib/utils/munkres.py, because
# the assignment among equal-cost alternatives depends on them (the metrics sum over the assignment).
# ---------------------------------------------------------------------------------------------
class Munkres:
    """munkres.py:244-623.  compute(cost) -> list of (row, col) for the lowest-cost pairing; a rectangular
    matrix is zero-padded on its short side (:304-316) and only in-range pairs are returned (:364-370)."""

    def compute(self, cost_matrix):
        cost = np.asarray(cost_matrix)
        rows, cols = cost.shape
        n = max(rows, cols)
        C = np.zeros((n, n), dtype=cost.dtype)
        C[:rows, :cols] = cost
        self.C, self.n = C, n
        self.row_cov = np.zeros(n, dtype=bool)
        self.col_cov = np.zeros(n, dtype=bool)
        self.mark = np.zeros((n, n), dtype=np.int8)        # 1 = starred zero, 2 = primed zero
        C -= C.min(axis=1, keepdims=True)                                        # step 1 (:385-399)
        for i in range(n):                                                       # step 2 (:401-418)
            for j in range(n):
                if C[i, j] == 0 and not self.col_cov[j] and not self.row_cov[i]:
                    self.mark[i, j] = 1
                    self.col_cov[j] = True
                    self.row_cov[i] = True
        self.row_cov[:] = False
        self.col_cov[:] = False
        step = 3
        while step != 7:
            if step == 3:                                                        # (:420-439)
                self.col_cov |= (self.mark == 1).any(axis=0)
                step = 7 if int((self.mark == 1).sum()) >= n else 4
            elif step == 4:
                step = self._prime_zeros()
            elif step == 5:
                self._augment()
                step = 3
            else:                                                                # step 6 (:510-524)
                open_cells = (~self.row_cov).any() and (~self.col_cov).any()
                minval = C[np.ix_(~self.row_cov, ~self.col_cov)].min() if open_cells else sys.maxsize
                C[self.row_cov, :] += minval
                C[:, ~self.col_cov] -= minval
                step = 4
        return [(i, j) for i in range(rows) for j in range(cols) if self.mark[i, j] == 1]

    def _find_zero(self):
        """:536-560 — the FIRST row holding an uncovered zero, and within it the LAST such column (the
        reference's scan does not leave the row once it has found one)."""
        free = (self.C == 0) & ~self.row_cov[:, None] & ~self.col_cov[None, :]
        rows = np.nonzero(free.any(axis=1))[0]
        if rows.size == 0:
            return -1, -1
        r = int(rows[0])
        return r, int(np.nonzero(free[r])[0][-1])

    def _prime_zeros(self):
        """step 4 (:441-472)."""
        while True:
            r, c = self._find_zero()
            if r < 0:
                return 6
            self.mark[r, c] = 2
            stars = np.nonzero(self.mark[r] == 1)[0]
            if stars.size:
                self.row_cov[r] = True
                self.col_cov[int(stars[0])] = False
            else:
                self.z0 = (r, c)
                return 5

    def _augment(self):
        """step 5 (:474-508): alternating path of primed and starred zeros from z0; flip it."""
        path = [self.z0]
        while True:
            col = path[-1][1]
            stars = np.nonzero(self.mark[:, col] == 1)[0]
            if stars.size == 0:
                break
            r = int(stars[0])
            path.append((r, col))
            path.append((r, int(np.nonzero(self.mark[r] == 2)[0][0])))
        for r, c in path:
            self.mark[r, c] = 0 if self.mark[r, c] == 1 else 1
        self.row_cov[:] = False
        self.col_cov[:] = False
        self.mark[self.mark == 2] = 0


def metrics_from_tables(tabs, obj_detect_threshold=0.75):
    """evaluation.py:125-257 on the integer tables of pair_stats()."""
    cont = tabs["cont"]
    pred_count, gt_count = cont.sum(axis=0), cont.sum(axis=1)
    labels_gt = np.array([l for l in range(MAX_LABELS) if gt_count[l] > 0 and l != BACKGROUND_LABEL], dtype=np.int64)
    labels_pred = np.array([l for l in range(MAX_LABELS) if pred_count[l] > 0 and l != BACKGROUND_LABEL], dtype=np.int64)
    num_labels_gt, num_labels_pred = labels_gt.shape[0], labels_pred.shape[0]

    def fixed(f, p, r, det_pct):
        return {"Objects F-measure": f, "Objects Precision": p, "Objects Recall": r, "Boundary F-measure": f,
                "Boundary Precision": p, "Boundary Recall": r, "obj_detected": num_labels_pred, "obj_detected_075": 0.,
                "obj_gt": num_labels_gt, "obj_detected_075_percentage": det_pct}
    if num_labels_pred == 0 and num_labels_gt > 0:          # all false negatives (:145-156)
        return fixed(0., 1., 0., 0.)
    if num_labels_pred > 0 and num_labels_gt == 0:          # all false positives (:157-168)
        return fixed(0., 0., 1., 0.)
    if num_labels_pred == 0 and num_labels_gt == 0:         # correctly predicted nothing (:169-180)
        return fixed(1., 1., 1., 1.)

    sel = np.ix_(labels_gt, labels_pred)
    true_positives = cont[sel].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        prec = true_positives / pred_count[labels_pred][None, :]             # :193
        rec = true_positives / gt_count[labels_gt][:, None]                  # :196
        F = np.where(prec + rec > 0, (2 * prec * rec) / (prec + rec), 0.0)   # :199-200
    boundary_prec_tp = tabs["prec_tp"][sel].astype(np.float64)
    boundary_rec_tp = tabs["rec_tp"][sel].astype(np.float64)
    boundary_prec_denom = float(tabs["bnd_pred"][labels_pred].sum())         # :211-215
    boundary_rec_denom = float(tabs["bnd_gt"][labels_gt].sum())              # :216-219

    F[np.isnan(F)] = 0
    assignments = Munkres().compute(F.max() - F.copy())                      # :222-224
    num_obj_detected = sum(1 for a in assignments if F[a] > obj_detect_threshold)
    idx = tuple(np.array(assignments).T)

    with np.errstate(divide="ignore", invalid="ignore"):
        objects_pred = float(pred_count[1:].sum())          # prediction.clip(0,1) == 1  (:236)
        objects_gt = float(gt_count[1:].sum())
        precision = np.float64(np.sum(true_positives[idx])) / objects_pred
        recall = np.float64(np.sum(true_positives[idx])) / objects_gt
        F_measure = (2 * precision * recall) / (precision + recall)
        if np.isnan(F_measure):
            F_measure = 0
        boundary_precision = np.float64(np.sum(boundary_prec_tp[idx])) / np.float64(boundary_prec_denom)
        boundary_recall = np.float64(np.sum(boundary_rec_tp[idx])) / np.float64(boundary_rec_denom)
        boundary_F_measure = (2 * boundary_precision * boundary_recall) / (boundary_precision + boundary_recall)
        if np.isnan(boundary_F_measure):
            boundary_F_measure = 0
    return {"Objects F-measure": F_measure, "Objects Precision": precision, "Objects Recall": recall,
            "Boundary F-measure": boundary_F_measure, "Boundary Precision": boundary_precision,
            "Boundary Recall": boundary_recall, "obj_detected": num_labels_pred, "obj_detected_075": num_obj_detected,
            "obj_gt": num_labels_gt, "obj_detected_075_percentage": num_obj_detected / num_labels_gt}


def multilabel_metrics(prediction, gt, obj_detect_threshold=0.75):
    """evaluation.py:109-257: overlap and boundary precision / recall / F-measure of the object masks (labels
    >= 1) with Hungarian matching on the F-measure, plus the detection counts.  prediction, gt: [H,W]."""
    return metrics_from_tables(pair_stats(prediction, gt), obj_detect_threshold)
