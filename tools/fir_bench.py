#!/usr/bin/env python
"""Micro-benchmark of ssde_upfirdn2d at the shapes the BASELINE networks use (GPU only): algorithmic bytes
(input + output(s), fp32) / time against the 8 TB/s HBM peak.  `dual` = act(GroupNorm(x)) and x filtered in one launch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from score_sde_pytorch_amd import hipops as ops, _lib as L  # noqa: E402
from score_sde_pytorch_amd.engine import fir_taps  # noqa: E402


def time_fir(n, c, h, up, down, pad, pro, dual, reps=10):
    x = torch.randn(n, h, h, c, device="cuda")
    G = min(c // 4, 32)
    gn = None
    if pro:
        mean, rstd = ops.groupnorm_stats(x, G, 1e-6)
        gn = (mean, rstd, torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), G)
    k = torch.tensor(fir_taps([1, 3, 3, 1], gain=float(up * up)))
    kw = dict(up=up, down=down, pad=pad, pro=L.PRO_GN_SILU if pro else L.PRO_NONE, gn=gn, dual=dual)
    y = ops.upfirdn2d_nhwc(x, k, **kw)
    y = y[0] if dual else y
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.upfirdn2d_nhwc(x, k, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = 4.0 * (x.numel() + y.numel() * (2 if dual else 1))
    return nbytes / ms / 1e9, ms, nbytes      # bytes / ms / 1e9 = TB/s


if __name__ == "__main__":
    shapes = [(256, 128, 32, 1, 2, (1, 1)), (256, 256, 16, 1, 2, (1, 1)), (256, 256, 8, 1, 2, (1, 1)),
              (256, 256, 4, 2, 1, (2, 1)), (256, 256, 8, 2, 1, (2, 1)), (256, 256, 16, 2, 1, (2, 1)),
              (16, 128, 256, 1, 2, (1, 1)), (16, 128, 128, 2, 1, (2, 1))]
    for n, c, h, up, down, pad in shapes:
        for pro, dual in ((0, 0), (1, 0), (1, 1)):
            tbs, ms, nb = time_fir(n, c, h, up, down, pad, pro, dual)
            print("N=%3d C=%3d %3dx%-3d up=%d down=%d pro=%d dual=%d  %7.1f MB  %.3f ms  %.2f TB/s = %.0f%% of 8 TB/s"
                  % (n, c, h, h, up, down, pro, dual, nb / 1e6, ms, tbs, tbs / 8.0 * 100.0), flush=True)
