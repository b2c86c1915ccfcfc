#!/usr/bin/env python
"""Energy per launch of the candidate kernels of a layer (GPU only; development tool).

The sampler runs AT the board's power cap (profiles/r4_power_during_sampler.txt: 1190-1200 W flat while bench.py runs, 261 W
idle), where end-to-end time follows the ENERGY of the kernels rather than their stand-alone durations (a kernel that gets
faster without getting cheaper only raises the power the others are throttled against: the bf16x6 GEMMs are 1.0 ms per
evaluation faster and the iteration gains 0.33 ms, profiles/r4_bf16x6_gemm_shapes_trace.txt).  This tool launches one kernel
back to back for a few seconds, samples the socket power beside it (sysfs hwmon, else rocm-smi) and prints
ms per launch, mean W, and J per launch = W x ms, with the idle power subtracted as well."""
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from score_sde_pytorch_amd import hipops as ops, _lib as L  # noqa: E402
from score_sde_pytorch_amd.engine import pack_conv_weight, pack_wino_weight, pack_wino4_weight, pack_matrix  # noqa: E402


def power_reader():
    # (the hwmon files of this image belong to other cards and read a constant: rocm-smi is the source that follows the load)
    def smi():
        r = subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True)
        d = json.loads(r.stdout[r.stdout.index("{"):])
        for k, v in d.get("card0", {}).items():
            if "ower" in k:
                return float(v)
        return float("nan")
    return smi, "rocm-smi --showpower"


READ, SRC = power_reader()


def measure(launch, dur=4.0):
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                samples.append((time.perf_counter(), READ()))
            except Exception:
                pass
            time.sleep(0.02)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < dur:
        for _ in range(200):
            launch()
        n += 200
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    stop.set()
    th.join()
    ws = [w for t, w in samples if t - t0 > 1.2 and w == w]
    ms = (t1 - t0) / n * 1e3
    return ms, (float(np.mean(ws)) if ws else float("nan"))


def conv_launcher(n, cin, cout, h, tile, gn=1):
    x = torch.randn(n, h, h, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / np.sqrt(9 * cin)
    a = L.ConvArgs()
    G = min(cin // 4, 32)
    mean, rstd = ops.groupnorm_stats(x, G, 1e-6)
    keep = [x, w, mean, rstd, torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")]
    ops._fill_src(a.main, x, None, L.PRO_GN_SILU, (mean, rstd, keep[4], keep[5], G))
    wp = {L.TILE_WINOGRAD: pack_wino_weight, L.TILE_WINOGRAD4: pack_wino4_weight}.get(tile, pack_conv_weight)(w)
    dst = torch.empty(n, h, h, cout, device="cuda")
    keep += [wp, dst]
    a.w_main, a.ksize, a.stride, a.pad, a.h_in, a.w_in = wp.data_ptr(), 3, 1, 1, h, h
    a.n, a.h_out, a.w_out, a.c_out, a.out_scale, a.dst, a.tile = n, h, h, cout, 1.0, dst.data_ptr(), tile
    lib, st = L.load(), ops._stream()
    return (lambda: L.check(lib.ssde_conv2d(C.byref(a), st))), keep


def gemm_launcher(n, h, k, cout):
    x = torch.randn(n, h, h, k, device="cuda")
    w = torch.randn(cout, k, device="cuda") / np.sqrt(k)
    a = L.ConvArgs()
    ops._fill_src(a.aux, x, None, L.PRO_NONE, None)
    wp = pack_matrix(w)
    dst = torch.empty(n, h, h, cout, device="cuda")
    a.w_aux, a.ksize, a.stride, a.pad = wp.data_ptr(), 0, 1, 0
    a.n, a.h_out, a.w_out, a.c_out, a.out_scale, a.dst, a.tile = n, h, h, cout, 1.0, dst.data_ptr(), L.TILE_AUTO
    lib, st = L.load(), ops._stream()
    return (lambda: L.check(lib.ssde_conv2d(C.byref(a), st))), [x, w, wp, dst]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    print("power source:", SRC)
    time.sleep(1.0)
    idle = float(np.mean([READ() for _ in range(10)]))
    print("idle %.0f W" % idle)
    for cin, cout, h in [(128, 128, 32), (256, 256, 16), (256, 256, 8)]:
        for tile, name in ((L.TILE_AUTO, "direct"), (L.TILE_WINOGRAD, "F(2x2,3x3)"), (L.TILE_WINOGRAD4, "F(4x4,3x3)")):
            f, keep = conv_launcher(n, cin, cout, h, tile)
            ms, w = measure(f)
            print("conv3x3 %3d->%3d @%2dx%-2d %-11s %.4f ms  %5.0f W  %.4f J/launch  (%.4f J above idle)" % (cin, cout, h, h, name, ms, w, w * ms * 1e-3, (w - idle) * ms * 1e-3), flush=True)
            del f, keep
    for h, k, cout in [(16, 256, 256), (16, 512, 256), (32, 256, 128)]:
        for label, mode, pipe in (("f32", "f32", "0"), ("bf16x6", "bf16x6", "0"), ("bf16x6+pipe", "bf16x6", "2")):
            os.environ["SSDE_MATRIX"], os.environ["SSDE_GEMM_PIPE"] = mode, pipe
            f, keep = gemm_launcher(n, h, k, cout)
            ms, w = measure(f)
            print("gemm %2dx%-2d K=%3d N=%3d %-12s %.4f ms  %5.0f W  %.4f J/launch  (%.4f J above idle)" % (h, h, k, cout, label, ms, w, w * ms * 1e-3, (w - idle) * ms * 1e-3), flush=True)
            del f, keep
