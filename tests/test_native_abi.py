"""The C-ABI library builds, loads, and exports every symbol include/uoc_hip.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from unseenobjectclustering_amd import _native
from unseenobjectclustering_amd.build import LIB_PATH, build_native

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "uoc_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uoc_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_loads():
    path = build_native()
    assert os.path.exists(path)
    lib = _native.lib()
    assert lib.uoc_version() >= 100


def test_every_declared_symbol_is_exported():
    build_native()
    lib = ctypes.CDLL(LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/uoc_hip.h but not exported by libuoc_hip.so"
    assert set(_native.EXPORTED_SYMBOLS) == set(syms)


def test_argument_validation_without_gpu():
    """Validation errors are reported through the error channel before any HIP call."""
    lib = _native.lib()
    assert lib.uoc_ms_workspace_bytes(1, 307200, 100) > 307200 * 4
    rc = lib.uoc_ms_cluster(None, 1, 100, 100, 20.0, 10, 0.04, None, None, None, None, None, None, 0, None)
    assert rc == -22
    assert b"null" in lib.uoc_last_error().lower()
    rc = lib.uoc_ms_seed_components(None, 1, 1000, 0.04, None, None, None)
    assert rc == -22
    rc = lib.uoc_ms_select_seeds_from(None, 1, 100, 10, 3, None, None, None, None, 0, None)      # continuation entry
    assert rc == -22
    # 128-d entry points: halves in {1, 2}; the 2-plane workspace is larger
    assert lib.uoc_ms_workspace_bytes_wide(1, 50176, 100, 2) > lib.uoc_ms_workspace_bytes_wide(1, 50176, 100, 1) > 0
    assert lib.uoc_ms_workspace_bytes_wide(1, 50176, 100, 1) == lib.uoc_ms_workspace_bytes(1, 50176, 100)
    assert lib.uoc_ms_workspace_bytes_wide(1, 50176, 100, 3) == 0
    rc = lib.uoc_ms_cluster_wide(None, 3, 1, 100, 100, 20.0, 10, 0.04, None, None, None, None, None, None, 0, None)
    assert rc == -22 and b"halves" in lib.uoc_last_error().lower()
    # network handle: unknown mode rejected, embedding width per mode, parameter intake is host-only
    h = ctypes.c_void_p()
    assert lib.uoc_net_create_mode(ctypes.byref(h), 9) == -22
    for mode, dim in ((0, 64), (1, 64), (2, 64), (3, 64), (4, 128)):
        h = ctypes.c_void_p()
        assert lib.uoc_net_create_mode(ctypes.byref(h), mode) == 0
        assert lib.uoc_net_embed_dim(h) == dim
        assert lib.uoc_net_workspace_bytes(h, 1, 480, 640) > 100 << 20
        arr = (ctypes.c_float * 4)(1, 2, 3, 4)
        assert lib.uoc_net_load_param(h, b"fcn.resnet34_8s.fc.bias", arr, 4) == 0
        assert lib.uoc_net_forward(h, None, None, 1, 64, 64, None, None, 0, None) == -22      # not finalized
        assert lib.uoc_net_destroy(h) == 0
    # evaluation and glue entries validate their pointers first
    assert lib.uoc_eval_workspace_bytes(480, 640) >= 2 * 480 * 640 * 4
    assert lib.uoc_eval_pair_stats(None, None, 480, 640, 3, None, None, 0, None) == -22
    assert lib.uoc_roi_crop(None, None, None, 480, 640, None, 1, 224, None, None, None, None) == -22
    # round 6: the device-side match_label_crop and the split-precision switch validate first, too
    assert lib.uoc_roi_match(None, None, None, None, 1, 224, 480, 640, None, None, None, None, None, 0, None) == -22
    assert lib.uoc_net_set_split_precision(None, 1) == -22
    h = ctypes.c_void_p()
    assert lib.uoc_net_create(ctypes.byref(h)) == 0
    assert lib.uoc_net_set_split_precision(h, 1) == -22            # not finalized
    assert lib.uoc_net_destroy(h) == 0


def test_shipped_library_has_no_result_affecting_knobs():
    """The library reads only the speed-only variables INTEGRATION.md lists, and its configuration fingerprint does not move with
    the rounding-affecting knobs the development builds of rounds 3-5 had (there is no development build any more)."""
    lib = _native.lib()
    assert lib.uoc_is_dev_build() == 0
    blob = open(LIB_PATH, "rb").read()
    names = set(m.decode() for m in re.findall(rb"UOC_[A-Z0-9_]{3,}\x00", blob))
    names = {n.rstrip("\x00") for n in names}
    assert names <= {"UOC_CONV_AUTOTUNE", "UOC_CONV_TUNE_CACHE", "UOC_CONV_VERBOSE", "UOC_FPS_PERSISTENT", "UOC_SPLIT_MAX_MB", "UOC_HC_PARTS"}, names
    fp = _native.config_fingerprint()
    assert fp != 0
    old = {k: os.environ.get(k) for k in ("UOC_WINOGRAD_F", "UOC_HC_QUAD", "UOC_WINOGRAD_MIN_CIN")}
    try:
        os.environ.update(UOC_WINOGRAD_F="2", UOC_HC_QUAD="0", UOC_WINOGRAD_MIN_CIN="0")
        lib.uoc_reload_env()
        assert _native.config_fingerprint() == fp
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        lib.uoc_reload_env()
    # the explicit-algorithm conv entry validates before any HIP call
    assert lib.uoc_conv2d_nhwc_algo(None, None, None, None, None, 1, 1, 8, 8, 32, 64, 5, 1, 1, 1, 0, 0, None) == -22


def test_product_path_refuses_cpu_tensors():
    import torch
    from unseenobjectclustering_amd.utils.mean_shift import mean_shift_smart_init
    X = torch.nn.functional.normalize(torch.randn(128, 64), dim=1)
    with pytest.raises(_native.NativeError):
        mean_shift_smart_init(X, 20.0)
