"""The LDS-DMA helpers (csrc/dma.h) write M0 from inline asm without saving, restoring or
declaring it (two scalar moves per DMA less).  That is safe only while the compiler never keeps a value of its own in
M0 across those statements.  This test disassembles the kernels that use them and checks, per kernel:
  * every instruction that mentions m0 is one of the helpers' own `s_mov_b32 m0, <lds address>`;
  * every LDS-DMA load is directly preceded by such a move (then `s_nop`);
  * no instruction that reads M0 implicitly for another purpose (movrel, sendmsg, GWS, ds_*_addtid, interp) occurs.
Needs hipcc (cross-compiles without a GPU)."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unseenobjectclustering_amd", "csrc")
FILES = ("conv.hip", "wino4.hip")
IMPLICIT_M0 = re.compile(r"^\s*(s_movrel|v_movrel|s_sendmsg|ds_gws|ds_\w*addtid|v_interp|ds_\w+_gs\b)")


def _asm(tmp, name):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(tmp, name + ".s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", os.path.join(CSRC, name),
                    "-o", out], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def test_no_compiler_use_of_m0_across_the_dma_helpers(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    with ThreadPoolExecutor(3) as pool:
        texts = list(pool.map(lambda n: _asm(str(tmp_path), n), FILES))
    checked = 0
    for name, text in zip(FILES, texts):
        for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*s_endpgm", text, re.S | re.M):
            kernel, body = m.group(1), m.group(2)
            ins = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
            dma = [i for i, l in enumerate(ins) if re.match(r"(global_load_lds_dwordx4|buffer_load_dwordx4 .*\blds\b)", l)]
            if not dma:
                continue
            checked += 1
            for i, l in enumerate(ins):
                if re.search(r"\bm0\b", l):
                    assert l.startswith("s_mov_b32 m0,"), f"{name}:{kernel}: unexpected use of m0: {l}"
                assert not IMPLICIT_M0.match(l), f"{name}:{kernel}: instruction with an implicit M0 operand: {l}"
            for i in dma:
                prev = [l for l in ins[max(0, i - 2):i]]
                assert any(l.startswith("s_mov_b32 m0,") for l in prev), f"{name}:{kernel}: DMA without its m0 move: {ins[i]} after {prev}"
    assert checked >= 8, f"only {checked} DMA kernels found — did the kernel names change?"
