"""Process-global config object `cfg` — the subset of the reference's lib/fcn/config.py that the
inference hot path reads, with the values the RGB-D `add` / cosine experiment
config sets (experiments/cfgs/seg_resnet34_8s_embedding_cosine_rgbd_add_tabletop.yml).  cfg.INPUT
'COLOR' / 'DEPTH' and FUSION_TYPE 'early' / 'cat' (the other shipped experiment configs) are
implemented too.

The reference's default EMBEDDING_METRIC is 'euclidean' (config.py:261) and every shipped
experiment overrides it to 'cosine'; this build implements the cosine path only and raises if
asked for anything else.
"""
from __future__ import annotations

import os

import numpy as np


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


cfg = AttrDict()
cfg.INPUT = "RGBD"                     # config.py:30
cfg.MODE = "TEST"
cfg.RNG_SEED = 3                       # config.py:379
cfg.PIXEL_MEANS = np.array([[[102.9801, 115.9465, 122.7717]]])   # config.py:376 (BGR)
cfg.gpu_id = 0
cfg.device = None                      # set by the driver (tools/test_images.py:157-158)

cfg.EXP_DIR = "default"                # config.py:388
cfg.ROOT_DIR = __import__("os").path.abspath(__import__("os").path.join(__import__("os").path.dirname(__file__), "..", ".."))   # config.py:385
cfg.INTRINSICS = ()                    # config.py:38 (tools/test_net.py:98-101 overwrites the dataset's K with it)

cfg.TRAIN = AttrDict()
cfg.TRAIN.CLASSES = (0, 1, 2, 3)       # config.py:66
cfg.TRAIN.CHROMATIC = True             # config.py:159 (training-time augmentation flags the loaders read; TEST mode ignores them)
cfg.TRAIN.ADD_NOISE = False            # config.py:160
cfg.TRAIN.NUM_UNITS = 64               # config.py:165
cfg.TRAIN.FUSION_TYPE = "add"          # config.py:92
cfg.TRAIN.SYN_CROP_SIZE = 224          # config.py:129
cfg.TRAIN.EMBEDDING_PRETRAIN = False
cfg.TRAIN.EMBEDDING_METRIC = "cosine"  # yml:56 (code default 'euclidean' is not implemented here)
cfg.TRAIN.EMBEDDING_NORMALIZATION = True
cfg.TRAIN.EMBEDDING_ALPHA = 0.02       # config.py:254 ; epsilon = 2*alpha (mean_shift.py:123)

cfg.TEST = AttrDict()
cfg.TEST.VISUALIZE = False             # config.py:319
cfg.TEST.IMS_PER_BATCH = 1             # config.py:330
# (no reference counterpart) one-frame-at-a-time calls replay hipGraphs (fcn/graph_replay.py).  OFF by default: bit-identical
# to the eager path but measured no faster (7.78 vs 7.76 ms per frame, profiles/r06_latency.md — the frame is kernel-bound once
# the ROI ordering runs on the device), and captured graphs in a process cost its multi-stream throughput schedule 5 %.
cfg.TEST.GRAPH_REPLAY = os.environ.get("UOC_GRAPH_REPLAY", "0") == "1"
# EXPERIMENT (no reference counterpart; never the default, never the headline): newly built SEGNETs run their Winograd plane GEMMs in
# split precision (bf16 x 3, fp32 accumulation; csrc/wino4_split.hip).  UOC_SPLIT_GEMM=1 runs a whole test / bench session that way.
cfg.TEST.SPLIT_PRECISION_GEMM = os.environ.get("UOC_SPLIT_GEMM", "0") == "1"
cfg.TEST.ROS_CAMERA = "camera"         # config.py:327 ('D415' | 'Azure' | a kinect-style namespace)
cfg.TEST.SCALES_BASE = (0.25, 0.5, 1.0, 2.0, 3.0)   # config.py:353 (every shipped yml sets (1.0,))
cfg.TEST.CLASSES = (0, 1, 2, 3)        # config.py:346 (if emptied, tools/test_net.py:68-69 copies TRAIN.CLASSES)


def get_output_dir(imdb, net):
    """config.py:395-405: <ROOT_DIR>/output/<EXP_DIR>/<dataset name>[/<net>]."""
    import os.path as osp
    path = osp.abspath(osp.join(cfg.ROOT_DIR, "output", cfg.EXP_DIR, imdb.name))
    return path if net is None else osp.join(path, net)


def network_mode() -> str:
    """'RGBD_ADD' | 'COLOR' | 'DEPTH' | 'RGBD_EARLY' | 'RGBD_CAT' from cfg.INPUT / cfg.TRAIN.FUSION_TYPE
    (SEG.py:69-71,97-110)."""
    if cfg.INPUT == "COLOR":
        return "COLOR"
    if cfg.INPUT == "DEPTH":
        return "DEPTH"
    if cfg.INPUT == "RGBD":
        if cfg.TRAIN.FUSION_TYPE == "add":
            return "RGBD_ADD"
        if cfg.TRAIN.FUSION_TYPE == "early":
            return "RGBD_EARLY"
        return "RGBD_CAT"       # SEG.py:107-110: anything that is neither 'add' nor 'early' concatenates
    raise NotImplementedError("cfg.INPUT=%r is not one of 'RGBD', 'COLOR', 'DEPTH'" % (cfg.INPUT,))


def uses_depth() -> bool:
    """test_dataset.py:236 / tools/test_images.py:110: depth is read for DEPTH and RGBD inputs."""
    return cfg.INPUT in ("DEPTH", "RGBD")


def require_supported():
    """Fail loudly on configurations the HIP path does not implement (there is no fallback)."""
    if cfg.TRAIN.EMBEDDING_METRIC != "cosine":
        raise NotImplementedError("only cfg.TRAIN.EMBEDDING_METRIC='cosine' is implemented on gfx950")
    network_mode()
    if not cfg.TRAIN.EMBEDDING_NORMALIZATION:
        raise NotImplementedError("EMBEDDING_NORMALIZATION=False is not implemented")


def _merge(src: dict, dst: AttrDict, path=""):
    for k, v in src.items():
        if k not in dst:
            continue        # the reference's cfg has ~200 training / dataset keys this path never reads
        if isinstance(dst[k], AttrDict):
            if not isinstance(v, dict):
                raise ValueError("Type mismatch for config key: %s%s" % (path, k))
            _merge(v, dst[k], path + k + ".")
        else:
            if dst[k] is not None and not isinstance(dst[k], np.ndarray) and type(dst[k]) is not type(v):
                raise ValueError("Type mismatch (%s vs. %s) for config key: %s%s" % (type(dst[k]), type(v), path, k))
            dst[k] = v


def cfg_from_file(filename):
    """config.py:436-442: merge an experiment yml (experiments/cfgs/*.yml) into `cfg`.  Keys the inference
    path does not read are ignored; the `!!python/tuple` tags those files carry are accepted.  Unlike the
    reference's `yaml.load` without a Loader this never constructs arbitrary Python objects."""
    import yaml

    class Loader(yaml.SafeLoader):
        pass

    Loader.add_constructor("tag:yaml.org,2002:python/tuple", lambda l, n: tuple(l.construct_sequence(n)))
    with open(filename, "r") as f:
        data = yaml.load(f, Loader=Loader) or {}
    _merge(data, cfg)
    require_supported()
