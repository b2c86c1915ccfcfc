#!/usr/bin/env python
"""Micro-benchmark of the 1x1 / NIN / Linear GEMM kernel (conv1x1.hip) at the BASELINE shapes (GPU only)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from score_sde_pytorch_amd import hipops as ops, _lib as L  # noqa: E402
from score_sde_pytorch_amd.engine import pack_matrix  # noqa: E402


def time_gemm(n, h, k, cout, tile=L.TILE_AUTO, resid=False, reps=10):
    dev = "cuda"
    x = torch.randn(n, h, h, k, device=dev)
    w = torch.randn(cout, k, device=dev) / np.sqrt(k)
    a = L.ConvArgs()
    ops._fill_src(a.aux, x, None, L.PRO_NONE, None)
    wp = pack_matrix(w)
    dst = torch.empty(n, h, h, cout, device=dev)
    r = torch.randn(n, h, h, cout, device=dev) if resid else None
    a.w_aux, a.ksize, a.stride, a.pad = wp.data_ptr(), 0, 1, 0
    a.n, a.h_out, a.w_out, a.c_out, a.out_scale, a.dst, a.tile = n, h, h, cout, 1.0, dst.data_ptr(), tile
    if resid:
        a.resid = r.data_ptr()
    a.flags = L.conv_route_flags()
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
    return 2.0 * k * cout * n * h * h / ms / 1e9, ms


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for h, k, cout in [(16, 256, 256), (16, 256, 768), (16, 512, 256), (32, 256, 128), (32, 256, 256), (8, 512, 256), (1, 512, 9984), (16, 1024, 256), (16, 2048, 256), (16, 2048, 128)]:
        g, gms = time_gemm(n, h, k, cout)
        o, oms = time_gemm(n, h, k, cout, tile=L.TILE_128x64)
        gr, grms = time_gemm(n, h, k, cout, resid=True)
        print("B=%d %2dx%-2d K=%4d N=%5d   gemm %6.1f TF/s (%.3f ms)   +resid %6.1f TF/s   general kernel(128x64) %6.1f TF/s (%.3f ms)"
              % (n, h, h, k, cout, g, gms, gr, o, oms), flush=True)
