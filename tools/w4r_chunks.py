#!/usr/bin/env python
"""Two-kernel F(4x4,3x3) layers cut into batch chunks that share ONE transformed-input window (GPU only).

Premise under test (tools/microbench/mall_window.hip measures it in isolation): the transform pass writes V = 2.25 x the
layer input and the matrix kernel reads it back in the next launch; if a chunk's V (75-150 MB) stays in the 256 MiB
Infinity Cache between the two launches, pass(chunk) -> matrix(chunk) -> pass(next chunk) on the same window takes V off
the HBM.  This script issues the chunks itself through ssde_conv2d (n = chunk, pointers offset by the chunk's first
image), so it measures the real kernels without any library change; the residual / GroupNorm prologue pointers follow.
usage: w4r_chunks.py [n]     (CONV_BENCH_SHAPES="cin,cout,h;..." to override the shapes)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from score_sde_pytorch_amd import hipops as ops, _lib as L  # noqa: E402
from score_sde_pytorch_amd.engine import pack_wino4r_weight  # noqa: E402


def run(n, cin, cout, h, chunks, reps=6, check=None):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(n, h, h, cin, device=dev, generator=g)
    w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) / np.sqrt(9 * cin)
    rs = torch.randn(n, h, h, cout, device=dev, generator=g)
    G = min(cin // 4, 32)
    mean, rstd = ops.groupnorm_stats(x, G, 1e-6)
    gamma, beta = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)
    wp = pack_wino4r_weight(w)
    dst = torch.empty(n, h, h, cout, device=dev)
    nc = n // chunks
    vbuf = torch.empty(36 * nc * (h // 4) * (h // 4) * cin, device=dev)
    args = []
    for c in range(chunks):
        a = L.ConvArgs()
        i0 = c * nc
        ops._fill_src(a.main, x[i0:i0 + nc], None, L.PRO_GN_SILU, (mean[i0:i0 + nc], rstd[i0:i0 + nc], gamma, beta, G))
        a.resid = rs[i0:i0 + nc].data_ptr()
        a.w_main, a.ksize, a.stride, a.pad, a.h_in, a.w_in = wp.data_ptr(), 3, 1, 1, h, h
        a.n, a.h_out, a.w_out, a.c_out, a.out_scale, a.tile = nc, h, h, cout, 1.0, L.TILE_WINOGRAD4R
        a.dst = dst[i0:i0 + nc].data_ptr()
        a.flags = L.conv_route_flags()
        a.wino_v = vbuf.data_ptr()
        args.append(a)
    lib, st = L.load(), ops._stream()

    def once():
        for a in args:
            L.check(lib.ssde_conv2d(C.byref(a), st))
    once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    keep = (x, w, rs, mean, rstd, gamma, beta, wp, vbuf)   # noqa: F841  (alive until the launches are done)
    return e0.elapsed_time(e1) / reps, dst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    shapes = [(128, 128, 32), (256, 128, 32), (256, 256, 32), (384, 128, 32), (256, 256, 16), (512, 256, 16)]
    if os.environ.get("CONV_BENCH_SHAPES"):
        shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["CONV_BENCH_SHAPES"].split(";")]
    for cin, cout, h in shapes:
        line, ref = [], None
        for chunks in (1, 2, 4, 8, 1, 2, 4, 8):
            ms, dst = run(n, cin, cout, h, chunks)
            if ref is None:
                ref = dst.clone()
            same = bool(torch.equal(ref, dst))
            vmb = 36 * (n // chunks) * (h // 4) ** 2 * cin * 4 / 1e6
            line.append("%d chunk%s (V window %.0f MB) %.4f ms%s" % (chunks, "s" if chunks > 1 else "", vmb, ms, "" if same else " MISMATCH"))
        print("%d->%d@%d n=%d | " % (cin, cout, h, n) + " | ".join(line), flush=True)
