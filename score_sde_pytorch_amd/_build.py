"""Builds libssde_hip.so (HIP kernels + C ABI, gfx950 only) in-tree with hipcc.

No torch extension machinery is involved: the library has a plain C ABI
(include/ssde.h) and is loaded with ctypes (see _lib.py).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libssde_hip.so")
SOURCES = ["runtime.hip", "conv_mfma.hip", "conv_wino.hip", "conv_wino4.hip", "wino4_xform.hip", "conv_wino4r.hip", "conv_small.hip", "conv1x1.hip", "wgrad.hip", "wgrad_wino.hip", "wgrad_wino4.hip", "groupnorm.hip", "resample.hip", "attention.hip",
           "elementwise.hip", "backward.hip", "plan.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fgpu-rdc" if False else "-fno-gpu-rdc",
         "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=False):
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "ssde_common.h"), os.path.join(HERE, "..", "include", "ssde.h")]
    hdr_mtime = max(os.path.getmtime(h) for h in headers)

    def compile_one(name):
        src = os.path.join(CSRC, name)
        obj = os.path.join(objdir, name.replace(".hip", ".o"))
        if force or _newer(src, obj) or hdr_mtime > os.path.getmtime(obj):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (name, r.stdout, r.stderr))
            return obj, True
        return obj, False

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    if force or any(ch for _, ch in results) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


def build_variant(name, defines):
    """A/B timing only: libssde_hip_<name>.so under csrc/build/variants, every source compiled with the given -D flags
    (e.g. ["-DSSDE_WINO_SCHED=0"]); loaded through SSDE_LIB_PATH.  Not part of build()."""
    vdir = os.path.join(CSRC, "build", "variants", name)
    os.makedirs(vdir, exist_ok=True)

    def compile_one(src_name):
        obj = os.path.join(vdir, src_name.replace(".hip", ".o"))
        r = subprocess.run([HIPCC] + FLAGS + list(defines) + ["-c", os.path.join(CSRC, src_name), "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src_name, r.stderr))
        return obj
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    lib = os.path.join(HERE, "..", "tools", "variants", "libssde_hip_%s.so" % name)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr)
    return os.path.abspath(lib)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
