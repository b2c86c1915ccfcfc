#!/usr/bin/env python
"""A/B of the bf16x6 1x1 GEMM shapes at the BASELINE layer shapes (batch 256): the 128 x 128 tile, the persistent pipelined
kernel, the 128 x 256 tile (SSDE_X6_WIDE) -- with and without the GroupNorm prologue and the residual epilogue the layers of
the network have.  GPU only; development tool."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from score_sde_pytorch_amd import hipops as ops, _lib as L  # noqa: E402
from score_sde_pytorch_amd.engine import pack_matrix  # noqa: E402


def run(n, h, k, cout, x, w, gn, resid, reps=30):
    a = L.ConvArgs()
    keep = []
    if gn:
        G = 32
        mean, rstd = ops.groupnorm_stats(x, G)
        gam, bet = torch.rand(k, device="cuda") + 0.5, torch.randn(k, device="cuda") * 0.1
        keep += [mean, rstd, gam, bet]
        ops._fill_src(a.aux, x, None, L.PRO_GN, (mean, rstd, gam, bet, G))
    else:
        ops._fill_src(a.aux, x, None, L.PRO_NONE, None)
    wp = pack_matrix(w)
    dst = torch.empty(n, h, h, cout, device="cuda")
    a.w_aux, a.ksize, a.stride, a.pad = wp.data_ptr(), 0, 1, 0
    a.n, a.h_out, a.w_out, a.c_out, a.out_scale, a.dst, a.tile = n, h, h, cout, 1.0, dst.data_ptr(), L.TILE_AUTO
    if resid is not None:
        a.resid = resid.data_ptr()
    a.flags = L.conv_route_flags()
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
    os.environ["SSDE_MATRIX"] = "bf16x6"
    print("bf16x6 GEMM [N*H*W, K] x [K, Cout], batch %d: ms per launch (direct-equivalent TF/s); max rel err vs fp64 over the first rows" % n)
    base = dict(SSDE_X6_WIDE="0", SSDE_GEMM_PIPE="0", SSDE_X6_BM="128", SSDE_X6_PF="1")
    variants = (("128x128", dict(base)), ("pipe", dict(base, SSDE_GEMM_PIPE="2")), ("128x256", dict(base, SSDE_X6_WIDE="1")),
                ("128pf2", dict(base, SSDE_X6_PF="2")), ("64pf1", dict(base, SSDE_X6_BM="64")), ("64pf2", dict(base, SSDE_X6_BM="64", SSDE_X6_PF="2")),
                ("default", dict(SSDE_X6_WIDE="", SSDE_GEMM_PIPE="", SSDE_X6_BM="", SSDE_X6_PF="")))
    for h, k, cout, gn, res in [(16, 256, 768, True, False), (16, 256, 768, False, False), (16, 256, 256, False, True), (16, 512, 256, False, True),
                                (16, 384, 256, False, True), (32, 256, 256, False, True), (8, 512, 256, False, True), (8, 256, 768, True, False),
                                (32, 256, 128, False, True), (32, 384, 128, False, True), (16, 128, 256, False, True), (1, 512, 9984, False, False), (1, 512, 512, False, False),
                                (8, 256, 256, False, True), (4, 256, 256, False, True), (4, 256, 768, True, False)]:
        x = torch.randn(n, h, h, k, device="cuda") * (1 + torch.rand(1, 1, 1, k, device="cuda") * 3)
        w = torch.randn(cout, k, device="cuda") / np.sqrt(k)
        resid = torch.randn(n, h, h, cout, device="cuda") if res else None
        line = "%2dx%-2d K=%4d N=%4d %s %s" % (h, h, k, cout, "gn" if gn else "  ", "resid" if res else "     ")
        outs = []
        for rnd in range(2):
            for label, env in variants:
                os.environ.update(env)
                ms, y = run(n, h, k, cout, x, w, gn, resid)
                outs.append(y)
                line += "  %s %.4f (%3.0f)" % (label, ms, 2.0 * k * cout * n * h * h / ms / 1e9)
            line += "  |"
        d = max(float((o - outs[0]).abs().max()) for o in outs[1:])
        line += "  max |diff| between variants %.2e (scale %.2f)" % (d, float(outs[0].abs().max()))
        print(line, flush=True)
