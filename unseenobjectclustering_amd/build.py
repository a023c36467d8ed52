"""Builds libuoc_hip.so (the C-ABI library, include/uoc_hip.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the repo snapshot (git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("UOC_LIB_PATH") or os.path.join(PKG_DIR, "libuoc_hip.so")   # override: load another build (A/B measurements)
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Compile every csrc/*.hip into libuoc_hip.so.  Returns the library path."""
    if not force and not _stale():
        return LIB_PATH
    return _build(LIB_PATH, "build", [], force, verbose)


def _build(lib_path: str, objdir: str, flags, force: bool, verbose: bool) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libuoc_hip.so")
    objs, jobs = [], []
    os.makedirs(os.path.join(CSRC, objdir), exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h"))
    newest_header = max([os.path.getmtime(h) for h in headers], default=0.0)
    for src in sources():
        obj = os.path.join(CSRC, objdir, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            jobs.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + list(flags) + ["-c", src, "-o", obj])
        objs.append(obj)
    if jobs:      # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as pool:
            list(pool.map(run, jobs))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return lib_path


if __name__ == "__main__":
    import sys
    print(build_native(force=True, verbose=True))
