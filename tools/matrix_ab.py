#!/usr/bin/env python
"""A/B of SSDE_MATRIX=f32 (exact-fp32 MFMA) against SSDE_MATRIX=bf16x6 (3-way bf16 split on the BF16 matrix pipe) for the
1x1 / NIN GEMM kernel at the BASELINE shapes: time per launch and the error of both forms against an fp64 product of the
same operands (GPU only; development tool, not part of the product path)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from score_sde_pytorch_amd import hipops as ops, _lib as L  # noqa: E402
from score_sde_pytorch_amd.engine import pack_matrix  # noqa: E402


def run(n, h, k, cout, mode, x, w, reps=20):
    os.environ["SSDE_MATRIX"] = mode
    a = L.ConvArgs()
    ops._fill_src(a.aux, x, None, L.PRO_NONE, None)
    wp = pack_matrix(w)
    dst = torch.empty(n, h, h, cout, device="cuda")
    a.w_aux, a.ksize, a.stride, a.pad = wp.data_ptr(), 0, 1, 0
    a.n, a.h_out, a.w_out, a.c_out, a.out_scale, a.dst, a.tile = n, h, h, cout, 1.0, dst.data_ptr(), L.TILE_AUTO
    lib, st = L.load(), ops._stream()
    for _ in range(3):
        L.check(lib.ssde_conv2d(C.byref(a), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.ssde_conv2d(C.byref(a), st))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, dst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    torch.manual_seed(0)
    print("GEMM [N*H*W, K] x [K, Cout], batch %d: ms per launch and direct-equivalent TF/s; relative L2 / max-abs error vs fp64" % n)
    for h, k, cout in [(16, 256, 256), (16, 256, 768), (16, 512, 256), (32, 256, 128), (32, 128, 128), (8, 512, 256), (16, 1024, 256), (16, 2048, 256)]:
        x = torch.randn(n, h, h, k, device="cuda") * (1 + torch.rand(1, 1, 1, k, device="cuda") * 3)
        w = torch.randn(cout, k, device="cuda") / np.sqrt(k)
        rows = min(n * h * h, 1 << 15)
        ref = x.reshape(-1, k)[:rows].double() @ w.double().t()
        line = "%2dx%-2d K=%4d N=%4d" % (h, h, k, cout)
        for label, mode, bm, pf, pipe in (("f32", "f32", 0, 0, "0"), ("f32+pipe", "f32", 0, 0, "1"), ("x6[128,1]", "bf16x6", 128, 1, "0"),
                                          ("x6[64,1]", "bf16x6", 64, 1, "0"), ("x6+pipe", "bf16x6", 128, 1, "1")):
            os.environ["SSDE_X6_BM"], os.environ["SSDE_X6_PF"], os.environ["SSDE_GEMM_PIPE"] = str(bm), str(pf), pipe
            ms, y = run(n, h, k, cout, mode, x, w)
            e = y.reshape(-1, cout)[:rows].double() - ref
            line += "   %s %.4f ms %5.1f TF/s err %.1e" % (label, ms, 2.0 * k * cout * n * h * h / ms / 1e9, (e.norm() / ref.norm()).item())
        print(line, flush=True)
