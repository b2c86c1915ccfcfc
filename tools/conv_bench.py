#!/usr/bin/env python
"""Micro-benchmark of the convolution kernels at the BASELINE shapes (GPU only): direct MFMA kernel vs the
Winograd kernel, and the weight-gradient kernel.  Prints algorithmic TFLOP/s (2*9*Cin*Cout*pixels)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from score_sde_pytorch_amd import hipops as ops, _lib as L  # noqa: E402
from score_sde_pytorch_amd.engine import pack_conv_weight, pack_wino_weight, pack_wino4_weight, pack_wino4r_weight  # noqa: E402


def time_conv(n, cin, cout, h, tile, gn=False, reps=5, resid=False, flags=0):
    dev = "cuda"
    x = torch.randn(n, h, h, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / np.sqrt(9 * cin)
    a = L.ConvArgs()
    gnt = None
    if int(gn) in (1, 2):
        G = min(cin // 4, 32)
        mean, rstd = ops.groupnorm_stats(x, G, 1e-6)
        gnt = (mean, rstd, torch.ones(cin, device=dev), torch.zeros(cin, device=dev), G)
    ops._fill_src(a.main, x, None, {0: L.PRO_NONE, 1: L.PRO_GN_SILU, 2: L.PRO_GN, 3: L.PRO_SILU}[int(gn)], gnt)
    wp = {L.TILE_WINOGRAD: pack_wino_weight, L.TILE_WINOGRAD4: pack_wino4_weight, L.TILE_WINOGRAD4R: pack_wino4r_weight}.get(tile, pack_conv_weight)(w)
    dst = torch.empty(n, h, h, cout, device=dev)
    if resid:
        rs = torch.randn(n, h, h, cout, device=dev)
        a.resid = rs.data_ptr()
    a.w_main, a.ksize, a.stride, a.pad, a.h_in, a.w_in = wp.data_ptr(), 3, 1, 1, h, h
    a.n, a.h_out, a.w_out, a.c_out, a.out_scale, a.dst, a.tile = n, h, h, cout, 1.0, dst.data_ptr(), tile
    a.flags = L.conv_route_flags() | flags
    if tile == L.TILE_WINOGRAD4R or os.environ.get("CONV_BENCH_EMIT_V"):      # the transformed-input buffer (36 / 16 of the tensor)
        vbuf = torch.empty(36 * n * (h // 4) * (h // 4) * cin, device=dev)
        a.wino_v = vbuf.data_ptr()
    lib = L.load()
    st = ops._stream()
    L.check(lib.ssde_conv2d(C.byref(a), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.ssde_conv2d(C.byref(a), st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return 2.0 * 9 * cin * cout * n * h * h / ms / 1e9, ms


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    shapes = [(128, 128, 32), (256, 128, 32), (256, 256, 16), (512, 256, 16), (256, 256, 8), (384, 128, 32)]
    if os.environ.get("CONV_BENCH_SHAPES"):          # "cin,cout,h;cin,cout,h;..."
        shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["CONV_BENCH_SHAPES"].split(";")]
    elif os.environ.get("CONV_BENCH_SMALL"):
        shapes = [(256, 256, 4), (512, 256, 4), (128, 256, 8), (256, 256, 8)]
    for cin, cout, h in shapes:
        for gn in ((0, 1, 2, 3) if os.environ.get("CONV_BENCH_PROLOGUES") else (0, 1)):   # 1 GN+SiLU, 2 GN, 3 SiLU
            rs = bool(os.environ.get("CONV_BENCH_RESID"))     # + residual input (the second convolution of a block)
            d, dms = time_conv(n, cin, cout, h, L.TILE_AUTO, gn, resid=rs)
            wv, wms = time_conv(n, cin, cout, h, L.TILE_WINOGRAD, gn, resid=rs) if h >= 8 else (0.0, float('inf'))
            w4, w4ms = time_conv(n, cin, cout, h, L.TILE_WINOGRAD4, gn, resid=rs) if h >= 8 and h % 4 == 0 else (0.0, float('inf'))
            print("B=%d %4d->%4d @%2dx%-2d gn=%d  direct %6.1f TF/s (%.3f ms)   winograd %6.1f TF/s (%.3f ms)   x%.2f   F(4x4,3x3) %6.1f TF/s (%.3f ms)"
                  % (n, cin, cout, h, h, gn, d, dms, wv, wms, dms / wms, w4, w4ms), flush=True)
