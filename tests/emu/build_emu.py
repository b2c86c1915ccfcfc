"""TEST INFRASTRUCTURE ONLY: builds tests/emu/build/libssde_emu.so -- the product's kernel sources
(score_sde_pytorch_amd/csrc/*.hip, unmodified) compiled as plain host C++ against the emulator's
<hip/hip_runtime.h> (tests/emu/hip/), exporting the same C ABI as libssde_hip.so.  "Device" pointers
are host pointers (torch CPU tensors).  Used by `pytest -m "not gpu"` to execute the kernels' index
math, LDS protocols and MFMA fragment handling without a GPU; the product never loads it.
"""
import os
import platform
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "score_sde_pytorch_amd", "csrc")
OUT = os.path.join(HERE, "build")
LIB = os.path.join(OUT, "libssde_emu.so")
CLANG = os.environ.get("SSDE_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
         "-Wno-unknown-attributes", "-Wno-unused-value", "-I", HERE, "-I", os.path.join(ROOT, "include")]


def available():
    return platform.machine() == "x86_64" and os.path.exists(CLANG)


def _sources():
    from score_sde_pytorch_amd import _build
    return [os.path.join(CSRC, s) for s in _build.SOURCES] + [os.path.join(HERE, "emu_runtime.cpp")]


def build(verbose=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = _sources()
    deps = [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "rocrand", "rocrand_kernel.h"),
            os.path.join(CSRC, "ssde_common.h"), os.path.join(ROOT, "include", "ssde.h")]
    dep_m = max(os.path.getmtime(d) for d in deps)

    def one(src):
        obj = os.path.join(OUT, os.path.basename(src).rsplit(".", 1)[0] + ".emu.o")
        if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), dep_m):
            return obj, False
        cmd = [CLANG] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emulator build failed for %s:\n%s" % (src, r.stderr[-6000:]))
        return obj, True

    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(one, srcs))
    if any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = [CLANG, "-shared", "-fPIC", "-o", LIB] + [o for o, _ in res]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("emulator link failed:\n%s" % r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    print(build(verbose=True))
