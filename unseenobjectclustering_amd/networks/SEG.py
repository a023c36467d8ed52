"""SEGNET for network 'Resnet34_8s' — cfg.INPUT='RGBD' with FUSION_TYPE 'add' / 'cat' (two backbones) or
'early' (one 6-channel backbone), 'COLOR', 'DEPTH' — host-side mirror of
/root/reference/lib/networks/SEG.py:26-181.

The module only OWNS the parameters (same state-dict keys as the reference:
``fcn.resnet34_8s.*`` / ``fcn_depth.resnet34_8s.*``, SEG.py:69-71) and hands them to the native
network (csrc/net.hip), which folds BatchNorm and runs every layer as a hand-written HIP kernel.
forward(img, label, depth) -> [B, 64, H, W] unit-norm embeddings, like SEG.py:88-119 in eval mode;
the returned tensor is a channels-last VIEW of the pixel-major [B, H*W, 64] buffer the kernels
write, so clustering consumes it without a transpose.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import _native
from ..fcn.config import cfg, network_mode, require_supported
from ..synth import resnet34_8s_param_shapes

BRANCHES = ("fcn", "fcn_depth")
_MODE_ID = {"RGBD_ADD": 0, "COLOR": 1, "DEPTH": 2, "RGBD_EARLY": 3, "RGBD_CAT": 4}     # include/uoc_hip.h UOC_NET_*


class _Node(nn.Module):
    """Bare container used to reproduce the reference's parameter tree (names only)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container; the network runs in libuoc_hip.so")


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, as_buffer: bool):
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        if not hasattr(node, p):
            node.add_module(p, _Node())
        node = getattr(node, p)
    if as_buffer:
        node.register_buffer(parts[-1], tensor)
    else:
        node.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class SEGNET(nn.Module):
    def __init__(self, init_weights=True, batch_norm=False, in_channels=3, network_name="Resnet34_8s",
                 num_units=64, use_coordconv=False):
        super().__init__()
        require_supported()
        if network_name != "Resnet34_8s" or num_units != 64:
            raise NotImplementedError("only Resnet34_8s with num_units=64 is implemented on gfx950")
        self.mode = network_mode()
        if in_channels != (6 if self.mode == "RGBD_EARLY" else 3):
            raise ValueError("in_channels=%d does not fit cfg.INPUT=%r / FUSION_TYPE=%r (SEG.py:103-105: early fusion "
                             "concatenates image and XYZ into 6 channels)" % (in_channels, cfg.INPUT, cfg.TRAIN.FUSION_TYPE))
        self.network_name = network_name
        self.in_channels = in_channels
        self.num_units = num_units
        self.metric = cfg.TRAIN.EMBEDDING_METRIC
        self.normalize = cfg.TRAIN.EMBEDDING_NORMALIZATION
        self.input_type = cfg.INPUT
        self.fusion_type = cfg.TRAIN.FUSION_TYPE
        # SEG.py:69-71: fcn always; fcn_depth only for RGBD without early fusion
        self.branches = BRANCHES if self.mode in ("RGBD_ADD", "RGBD_CAT") else BRANCHES[:1]
        for br in self.branches:
            for name, shape in resnet34_8s_param_shapes(num_units, in_channels):
                if name.endswith("num_batches_tracked"):
                    t = torch.zeros((), dtype=torch.long)
                elif len(shape) == 4:
                    t = torch.empty(shape)
                    nn.init.xavier_normal_(t)              # SEG.py:77-81 re-initialises every conv
                elif name.endswith("running_var") or name.endswith(".weight"):
                    t = torch.ones(shape)                  # BN weight = 1, running_var = 1
                else:
                    t = torch.zeros(shape)                 # BN bias, running_mean, fc bias
                is_buf = name.endswith(("running_mean", "running_var", "num_batches_tracked"))
                _attach(self, f"{br}.resnet34_8s.{name}", t, is_buf)
        self._handle = None
        self._handle_device = None
        self._native_gen = 0          # bumped whenever the native weight copy is rebuilt (captured hipGraphs bake its pointers)
        # EXPERIMENT (never the default, never the headline): plane GEMMs in split precision (uoc_net_set_split_precision)
        self.split_precision = bool(getattr(cfg.TEST, "SPLIT_PRECISION_GEMM", False))
        self.train(False)

    # -- native network management -----------------------------------------------------------
    def _release(self):
        if getattr(self, "_handle", None) is not None:
            _native.lib().uoc_net_destroy(self._handle)
            self._handle = None
            self._handle_device = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self._release()           # weights changed: rebuild the native copy on next forward
        return out

    def set_split_precision(self, on: bool):
        """EXPERIMENT: run the Winograd layers' plane GEMMs in split precision (three bf16 terms per fp32 operand, six bf16 MFMA
        products, fp32 accumulation; csrc/wino4_split.hip).  Not bit-identical to the fp32 path; reported under its own key."""
        self.split_precision = bool(on)
        if self._handle is not None:
            with torch.cuda.device(self._handle_device):
                _native.check(_native.lib().uoc_net_set_split_precision(self._handle, 1 if on else 0), "uoc_net_set_split_precision")
            self._native_gen += 1         # captured graphs bake the launch sequence
        return self

    def refresh(self):
        """Drop the native weight copy (call after mutating parameters in place)."""
        self._release()

    def _ensure_native(self, device):
        if self._handle is not None and self._handle_device == device:
            return
        self._release()
        L = _native.lib()
        h = ctypes.c_void_p()
        _native.check(L.uoc_net_create_mode(ctypes.byref(h), _MODE_ID[self.mode]), "uoc_net_create_mode")
        for key, t in self.state_dict().items():
            if key.endswith("num_batches_tracked"):
                continue
            ht = t.detach().to("cpu", torch.float32).contiguous()
            _native.check(L.uoc_net_load_param(h, key.encode(), ctypes.c_void_p(ht.data_ptr()), ht.numel()),
                          f"uoc_net_load_param({key})")
        with torch.cuda.device(device):
            _native.check(L.uoc_net_finalize(h), "uoc_net_finalize")
        if getattr(self, "split_precision", False):
            with torch.cuda.device(device):
                _native.check(L.uoc_net_set_split_precision(h, 1), "uoc_net_set_split_precision")
        self._handle = h
        self._handle_device = device
        self._native_gen = getattr(self, "_native_gen", 0) + 1

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, img, label=None, depth=None):
        if self.training:
            raise NotImplementedError("training (EmbeddingLoss) is out of scope; call .eval()")
        need_img, need_depth = self.mode != "DEPTH", self.mode != "COLOR"
        if need_depth and depth is None:
            raise ValueError("this network reads the XYZ `depth` tensor (SEG.py:97-106)")
        lead = img if need_img else depth
        if not lead.is_cuda:
            raise _native.NativeError("SEGNET.forward needs ROCm tensors (no CPU fallback); call .cuda() on the inputs")
        dev = lead.device
        img = img.to(dev).contiguous().float() if need_img else None
        depth = depth.to(dev).contiguous().float() if need_depth else None
        B, C, H, W = (img if need_img else depth).shape
        assert C == 3 and (img is None or depth is None or depth.shape == img.shape), \
            "expects [B,3,H,W] image and XYZ tensors"
        self._ensure_native(dev)
        L = _native.lib()
        cat = self.mode == "RGBD_CAT"
        embed = torch.empty((B, 2, H * W, 64) if cat else (B, H * W, 64), dtype=torch.float32, device=dev)
        nbytes = L.uoc_net_workspace_bytes(self._handle, B, H, W)
        ws = _net_workspace(dev, nbytes)
        with torch.cuda.device(dev):
            rc = L.uoc_net_forward(self._handle, _native.ptr(img), _native.ptr(depth), B, H, W, _native.ptr(embed),
                                   _native.ptr(ws), ws.numel(), _native.stream_ptr(dev))
        _native.check(rc, "uoc_net_forward")
        if cat:
            # [B,128,H,W] like SEG.py:110 (a copy: the kernels keep 128-d fields as two 64-channel planes);
            # the plane tensor rides along so that clustering_features consumes it without converting back
            out = embed.view(B, 2, H, W, 64).permute(0, 1, 4, 2, 3).reshape(B, 128, H, W)
            out._uoc_planes = embed
            return out
        return embed.view(B, H, W, 64).permute(0, 3, 1, 2)

    def weight_parameters(self):
        return [p for n, p in self.named_parameters() if "weight" in n]

    def bias_parameters(self):
        return [p for n, p in self.named_parameters() if "bias" in n]


_net_ws = {}


def _net_workspace(device, nbytes):
    key = _native.stream_key(device)      # one activation arena per stream (frames in flight do not share it)
    ws = _net_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        _native.retire(ws)          # a captured hipGraph may have baked the old arena's address
        ws = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
        _net_ws[key] = ws
    return ws


def update_model(model, data):
    """SEG.py:130-159: strip 'module.' prefixes, keep entries whose key and shape match, load."""
    if data is None:
        return
    model_dict = model.state_dict()
    data_new = dict(data)
    for k, v in data.items():
        if "module." in k:
            data_new[k[7:]] = v
    picked = {}
    for k, v in data_new.items():
        if k in model_dict:
            v = v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))
            if tuple(v.shape) == tuple(model_dict[k].shape):
                picked[k] = v
    model_dict.update(picked)
    model.load_state_dict(model_dict)


def seg_resnet34_8s_embedding(num_classes=2, num_units=64, data=None):
    """SEG.py:173-176."""
    model = SEGNET(in_channels=3, network_name="Resnet34_8s", num_units=num_units)
    update_model(model, data)
    return model


def seg_resnet34_8s_embedding_early(num_classes=2, num_units=64, data=None):
    """SEG.py:178-181 (needs cfg.INPUT='RGBD', cfg.TRAIN.FUSION_TYPE='early')."""
    model = SEGNET(in_channels=6, network_name="Resnet34_8s", num_units=num_units)
    update_model(model, data)
    return model
