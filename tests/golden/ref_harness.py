"""Import harness for the upstream reference (BUILD CONTAINER ONLY).

Used by tests/golden/make_golden.py to capture golden vectors from the reference's own
Python code under /root/reference/lib.  /root/reference does not exist on the GPU box and
nothing under tests/ -m gpu, bench.py or smoke() imports this file.

Shims: stub modules for easydict / torchvision / cv2 / transforms3d that
the reference imports but does not need on this path; `.cuda()` made a no-op; cfg switched
to cosine / RGBD / add / CPU.  Nothing is written into /root/reference.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "lib"))


class _AttrDict(dict):
    """Minimal stand-in for easydict.EasyDict (attribute access on a dict, recursive)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            v = _AttrDict(v)
        super().__setitem__(k, v)
        super().__setattr__(k, v)

    __setitem__ = __setattr__


def load_reference():
    """Returns a namespace with the reference modules needed for golden capture."""
    import torch

    sys.dont_write_bytecode = True
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict")
        m.EasyDict = _AttrDict
        sys.modules["easydict"] = m
    for name in ("torchvision", "cv2", "transforms3d", "transforms3d.quaternions"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    q = sys.modules["transforms3d.quaternions"]
    for fn in ("mat2quat", "quat2mat", "qmult"):
        setattr(q, fn, lambda *a, **k: None)
    sys.modules["transforms3d"].quaternions = q
    import matplotlib
    matplotlib.use("Agg")

    lib = os.path.join(REFERENCE_ROOT, "lib")
    if lib not in sys.path:
        sys.path.insert(0, lib)

    from fcn.config import cfg
    cfg.TRAIN.EMBEDDING_METRIC = "cosine"
    cfg.INPUT = "RGBD"
    cfg.TRAIN.FUSION_TYPE = "add"
    cfg.TRAIN.EMBEDDING_PRETRAIN = False
    cfg.TEST.VISUALIZE = False
    cfg.device = torch.device("cpu")
    torch.Tensor.cuda = lambda self, *a, **k: self

    import utils.mean_shift as mean_shift
    import fcn.test_dataset as test_dataset
    import networks
    import networks.SEG as SEG

    ns = types.SimpleNamespace(cfg=cfg, mean_shift=mean_shift, test_dataset=test_dataset,
                               networks=networks, SEG=SEG)
    return ns
