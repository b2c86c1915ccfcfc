#!/usr/bin/env python
"""The input gradient of the batched temb projection, [batch, 9984] x [9984, 512] (8 workgroups with 624 stages each on the
GEMM kernel: 0.53 ms of the training step), under every route the library has.  GPU only; development tool."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from score_sde_pytorch_amd import hipops as ops, _lib as L  # noqa: E402
from score_sde_pytorch_amd.engine import pack_matrix  # noqa: E402


def run(n, k, cout, x, w, tile, reps=10):
    a = L.ConvArgs()
    ops._fill_src(a.aux, x, None, L.PRO_NONE, None)
    wp = pack_matrix(w)
    dst = torch.empty(n, 1, 1, cout, device="cuda")
    a.w_aux, a.ksize, a.stride, a.pad = wp.data_ptr(), 0, 1, 0
    a.n, a.h_out, a.w_out, a.c_out, a.out_scale, a.dst, a.tile = n, 1, 1, cout, 1.0, dst.data_ptr(), tile
    a.flags = L.conv_route_flags()
    lib, st = L.load(), ops._stream()
    for _ in range(2):
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
    torch.manual_seed(0)
    for n, k, cout in [(128, 9984, 512), (256, 9984, 512), (128, 4096, 512), (128, 512, 9984)]:
        x = torch.randn(n, 1, 1, k, device="cuda")
        w = torch.randn(cout, k, device="cuda") / np.sqrt(k)
        ref = x.reshape(n, k).double() @ w.double().t()
        line = "[%d, %d] x [%d, %d]:" % (n, k, k, cout)
        base = dict(SSDE_MATRIX="bf16x6", SSDE_X6_WIDE="", SSDE_GEMM_PIPE="", SSDE_X6_BM="", SSDE_X6_PF="")
        for label, env, tile in (("default", {}, L.TILE_AUTO), ("bm128", dict(SSDE_X6_WIDE="1") if cout % 256 == 0 else {}, L.TILE_AUTO),
                                 ("64pf2", dict(SSDE_X6_BM="64", SSDE_X6_PF="2"), L.TILE_AUTO), ("128pf2", dict(SSDE_X6_PF="2"), L.TILE_AUTO),
                                 ("f32", dict(SSDE_MATRIX="f32"), L.TILE_AUTO), ("general 64x64", dict(SSDE_MATRIX="f32"), L.TILE_64x64),
                                 ("general 128x64", dict(SSDE_MATRIX="f32"), L.TILE_128x64)):
            os.environ.update(base); os.environ.update(env)
            try:
                ms, y = run(n, k, cout, x, w, tile)
                err = float((y.reshape(n, cout).double() - ref).norm() / ref.norm())
                line += "  %s %.4f ms (err %.1e)" % (label, ms, err)
            except Exception as ex:  # noqa: BLE001
                line += "  %s failed (%s)" % (label, str(ex)[:60])
        print(line, flush=True)
