"""ctypes binding of libuoc_hip.so (include/uoc_hip.h).  No CPU fallback: if the library is
missing or a call fails, this raises — the product path never routes around the HIP kernels."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_size_t, c_void_p, POINTER

from .build import LIB_PATH

_lib = None


class NativeError(RuntimeError):
    pass


def _declare(lib):
    P = c_void_p
    lib.uoc_version.restype = c_int
    lib.uoc_is_dev_build.argtypes = []
    lib.uoc_is_dev_build.restype = c_int
    lib.uoc_config_fingerprint.argtypes = []
    lib.uoc_config_fingerprint.restype = ctypes.c_ulonglong
    lib.uoc_shutdown.restype = c_int
    lib.uoc_reload_env.argtypes = []
    lib.uoc_reload_env.restype = c_int
    lib.uoc_shutdown.argtypes = []
    lib.uoc_last_error.restype = c_char_p
    lib.uoc_ms_set_persistent_fps.argtypes = [c_int]
    lib.uoc_ms_set_persistent_fps.restype = c_int
    lib.uoc_ms_set_stream_ordering.argtypes = [c_int]
    lib.uoc_ms_set_stream_ordering.restype = c_int
    lib.uoc_ms_fps_fallbacks.argtypes = []
    lib.uoc_ms_fps_fallbacks.restype = c_int
    lib.uoc_ms_workspace_bytes.restype = c_size_t
    lib.uoc_ms_workspace_bytes.argtypes = [c_int, c_int, c_int]
    lib.uoc_ms_select_seeds.argtypes = [P, c_int, c_int, c_int, P, P, P, P, c_size_t, P]
    lib.uoc_ms_select_seeds_from.argtypes = [P, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, P]
    lib.uoc_ms_hill_climb.argtypes = [P, c_int, c_int, P, c_int, c_float, c_int, P, c_size_t, P]
    lib.uoc_ms_seed_components.argtypes = [P, c_int, c_int, c_float, P, P, P]
    lib.uoc_ms_assign.argtypes = [P, c_int, c_int, P, P, P, c_int, P, P, P, c_size_t, P]
    lib.uoc_ms_cluster.argtypes = [P, c_int, c_int, c_int, c_float, c_int, c_float, P, P, P, P, P, P, c_size_t, P]
    lib.uoc_ms_check.argtypes = [P]
    lib.uoc_ms_check.restype = c_int
    lib.uoc_ms_workspace_bytes_wide.restype = c_size_t
    lib.uoc_ms_workspace_bytes_wide.argtypes = [c_int, c_int, c_int, c_int]
    lib.uoc_ms_cluster_wide.argtypes = [P, c_int, c_int, c_int, c_int, c_float, c_int, c_float, P, P, P, P, P, P, c_size_t, P]
    lib.uoc_ms_cluster_wide.restype = c_int
    lib.uoc_net_embed_dim.argtypes = [P]
    lib.uoc_net_embed_dim.restype = c_int
    lib.uoc_net_create.argtypes = [POINTER(P)]
    lib.uoc_net_create_mode.argtypes = [POINTER(P), c_int]
    lib.uoc_net_destroy.argtypes = [P]
    lib.uoc_net_load_param.argtypes = [P, c_char_p, P, c_size_t]
    lib.uoc_net_finalize.argtypes = [P]
    lib.uoc_net_workspace_bytes.restype = c_size_t
    lib.uoc_net_workspace_bytes.argtypes = [P, c_int, c_int, c_int]
    lib.uoc_net_forward.argtypes = [P, P, P, c_int, c_int, c_int, P, P, c_size_t, P]
    lib.uoc_net_set_split_precision.argtypes = [P, c_int]
    lib.uoc_net_set_split_precision.restype = c_int
    lib.uoc_conv2d_nhwc.argtypes = [P, P, P, P, P] + [c_int] * 11 + [P]
    lib.uoc_conv2d_nhwc_algo.argtypes = [P, P, P, P, P] + [c_int] * 12 + [P]
    lib.uoc_conv2d_nhwc_algo.restype = c_int
    for name in ("uoc_net_create", "uoc_net_create_mode", "uoc_net_destroy", "uoc_net_load_param", "uoc_net_finalize", "uoc_net_forward",
                 "uoc_conv2d_nhwc"):
        getattr(lib, name).restype = c_int
    lib.uoc_eval_workspace_bytes.restype = c_size_t
    lib.uoc_eval_workspace_bytes.argtypes = [c_int, c_int]
    lib.uoc_eval_pair_stats.argtypes = [P, P, c_int, c_int, c_int, P, P, c_size_t, P]
    lib.uoc_eval_pair_stats.restype = c_int
    lib.uoc_roi_workspace_bytes.restype = c_size_t
    lib.uoc_roi_workspace_bytes.argtypes = []
    lib.uoc_prep_rgbd.argtypes = [P, P, c_int, c_int] + [c_float] * 7 + [P, P, P]
    lib.uoc_prep_rgbd.restype = c_int
    lib.uoc_filter_labels_depth.argtypes = [P, P, ctypes.c_long, c_int, c_int, c_int, c_float, P, c_size_t, P]
    lib.uoc_roi_build.argtypes = [P, P, c_int, c_int, c_float, c_float, P, P, c_size_t, P]
    lib.uoc_roi_crop.argtypes = [P, P, P, c_int, c_int, P, c_int, c_int, P, P, P, P]
    lib.uoc_roi_match_stats.argtypes = [P, P, P, c_int, c_int, P, P, P, c_size_t, P]
    lib.uoc_roi_paste.argtypes = [P, P, P, P, c_int, c_int, c_int, c_int, P, P]
    lib.uoc_labels_to_u8.argtypes = [P, ctypes.c_long, P, P, P]
    lib.uoc_roi_match.argtypes = [P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, c_size_t, P]
    for name in ("uoc_filter_labels_depth", "uoc_roi_build", "uoc_roi_crop", "uoc_roi_match_stats", "uoc_roi_paste", "uoc_labels_to_u8", "uoc_roi_match"):
        getattr(lib, name).restype = c_int
    lib.uoc_lzf_decompress.argtypes = [P, c_size_t, P, c_size_t]
    lib.uoc_lzf_decompress.restype = ctypes.c_long
    lib.uoc_prof_enable.argtypes = [c_int]
    lib.uoc_prof_reset.argtypes = []
    lib.uoc_prof_report.argtypes = [ctypes.c_char_p, c_size_t]
    for name in ("uoc_prof_enable", "uoc_prof_reset", "uoc_prof_report"):
        getattr(lib, name).restype = c_int
    for name in ("uoc_ms_select_seeds", "uoc_ms_select_seeds_from", "uoc_ms_hill_climb", "uoc_ms_seed_components", "uoc_ms_assign",
                 "uoc_ms_cluster"):
        getattr(lib, name).restype = c_int


# every symbol include/uoc_hip.h declares (tests check the .so exports them all)
EXPORTED_SYMBOLS = (
    "uoc_version", "uoc_is_dev_build", "uoc_config_fingerprint", "uoc_shutdown", "uoc_last_error", "uoc_reload_env", "uoc_ms_set_persistent_fps", "uoc_ms_set_stream_ordering", "uoc_ms_fps_fallbacks", "uoc_ms_check", "uoc_ms_workspace_bytes", "uoc_ms_select_seeds", "uoc_ms_select_seeds_from", "uoc_ms_hill_climb",
    "uoc_ms_seed_components", "uoc_ms_assign", "uoc_ms_cluster", "uoc_ms_workspace_bytes_wide", "uoc_ms_cluster_wide",
    "uoc_net_embed_dim",
    "uoc_net_create", "uoc_net_create_mode", "uoc_net_destroy", "uoc_net_load_param", "uoc_net_finalize", "uoc_net_workspace_bytes",
    "uoc_net_forward", "uoc_net_set_split_precision", "uoc_conv2d_nhwc", "uoc_conv2d_nhwc_algo",
    "uoc_roi_workspace_bytes", "uoc_prep_rgbd", "uoc_filter_labels_depth", "uoc_roi_build", "uoc_roi_crop", "uoc_roi_match_stats",
    "uoc_roi_paste", "uoc_roi_match", "uoc_labels_to_u8", "uoc_eval_workspace_bytes", "uoc_eval_pair_stats", "uoc_lzf_decompress", "uoc_prof_enable", "uoc_prof_reset", "uoc_prof_report",
)


class RoiTable(ctypes.Structure):
    """Mirror of uoc_roi_table (include/uoc_hip.h)."""
    _fields_ = [("K", c_int32), ("label", c_int32 * 128), ("box", (c_int32 * 4) * 128)]


ROI_TABLE_BYTES = ctypes.sizeof(RoiTable)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build it with `python -m unseenobjectclustering_amd.build` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be the one the process
        # loads first, otherwise this library would pull /opt/rocm's copy in and the two runtimes clash
        # ("no ROCm-capable device is detected").  Importing torch before dlopen guarantees sharing.
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        _declare(l)
        _lib = l
        import atexit
        atexit.register(l.uoc_shutdown)
    return _lib


CONV_DIRECT, CONV_WINOGRAD4, CONV_WINOGRAD4_BF16X3 = 0, 4, 5       # include/uoc_hip.h: UOC_CONV_*


def config_fingerprint() -> int:
    """uoc_config_fingerprint(): what could make two ranks compute different bits (library version / development build and
    its rounding-affecting knobs).  runner.run_sharded compares it across ranks."""
    return int(lib().uoc_config_fingerprint())


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().uoc_last_error().decode("utf-8", "replace")
        raise NativeError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device (or host) pointer of a torch tensor as c_void_p; None -> NULL."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def stream_ptr(device=None):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def stream_key(device):
    """(device type, index, current HIP stream handle): the key of every per-stream scratch cache, so that frames in
    flight on different streams never share a workspace."""
    import torch
    return (device.type, device.index, int(torch.cuda.current_stream(device).cuda_stream))


_retired = []
GRAPHS_ALIVE = 0          # captured hipGraphs in this process (fcn/graph_replay.py)


def retire(t):
    """A per-stream scratch buffer that is being replaced by a larger one.  Captured hipGraphs bake device addresses, so
    once any graph exists the old buffer is kept alive (a bounded leak: buffers grow geometrically / by ROI count)
    instead of going back to the allocator."""
    if t is not None and GRAPHS_ALIVE > 0:
        _retired.append(t)


def prof_enable(on: bool):
    lib().uoc_prof_reset()
    lib().uoc_prof_enable(1 if on else 0)


def prof_report():
    """Per-kernel-class totals recorded since prof_enable(True): list of dicts."""
    import json
    buf = ctypes.create_string_buffer(1 << 19)
    check(lib().uoc_prof_report(buf, len(buf)), "uoc_prof_report")
    return json.loads(buf.value.decode())
