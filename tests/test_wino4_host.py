"""CPU check of the Winograd F(4x4,3x3) arithmetic the HIP kernels share (csrc/wino4_math.h is host + device code):
tests/csrc/wino4_host_test.cpp drives the weight / input / output transform bodies with plain loops and compares with
a direct dilated 3x3 convolution in double (partial tiles, every dilation the backbone uses, ragged sizes)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wino4_transforms_on_the_host(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "wino4_host_test")
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-o", exe,
                    os.path.join(ROOT, "tests", "csrc", "wino4_host_test.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok") == 6, r.stdout
