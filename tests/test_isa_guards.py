"""Compile-time facts the hand-written kernels rely on, checked in the gfx950 ISA hipcc emits (no GPU needed):
  * the LDS-DMA macro writes M0 from inline asm without a clobber (hipcc rejects "m0" as a clobber): nothing else in the
    Winograd kernels may read or write M0;
  * the main loops of the Winograd kernels must not touch scratch (a spill inside the matrix stream is a memory round trip
    per stage); the epilogues may keep a handful of spills."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _isa(src, tmp_path, defines=()):
    out = os.path.join(str(tmp_path), src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "score_sde_pytorch_amd", "csrc"), *defines,
           os.path.join(ROOT, "score_sde_pytorch_amd", "csrc", src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def _kernels(isa, name):
    """(symbol, body) of every instantiation of a kernel"""
    found = []
    for m in re.finditer(r"^(_ZN\S*%s\S*):" % name, isa, re.M):
        body = isa[m.start():]
        found.append((m.group(1), body[:body.index(".Lfunc_end")]))
    assert found, name
    return found


def _main_loop(body):
    """the innermost backward-branch region that holds the kernel's MFMAs"""
    lines = body.split("\n")
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    total = sum("v_mfma" in l for l in lines)
    best = None
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), i) < i:
            lo = labels[m.group(1)]
            n = sum("v_mfma" in x for x in lines[lo:i])
            if n >= max(8, total // 4) and (best is None or i - lo < best[1] - best[0]):
                best = (lo, i)
    assert best is not None
    return lines[best[0]:best[1] + 1]


@pytest.mark.parametrize("src,kernel", [("conv_wino4.hip", "conv_wino4_kernel"), ("conv_wino.hip", "conv_wino_kernel")])
def test_winograd_kernels_own_m0_and_keep_scratch_out_of_the_loop(src, kernel, tmp_path):
    isa = _isa(src, tmp_path)
    for sym, body in _kernels(isa, kernel):
        m0_users = [l.strip() for l in body.split("\n") if re.search(r"\bm0\b", l.split(";")[0]) and "s_mov_b32 m0" not in l]
        assert not m0_users, (sym, m0_users[:4])
        loop = _main_loop(body)
        assert sum("v_mfma" in l for l in loop) >= 8
        scratch = [l.strip() for l in loop if re.match(r"\s+scratch_", l)]
        assert not scratch, (sym, scratch[:4])
    spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", isa)]
    assert max(spills) <= 16, spills
