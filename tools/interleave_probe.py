#!/usr/bin/env python
"""Does a kernel run slower right after a bf16x6 GEMM than right after the fp32 GEMM?  A conv_wino4 launch (128->128
@32x32, GN+SiLU, batch 256) alternates with a 1x1 GEMM (256->256 @16x16) in one stream; HIP events bracket every launch of
both; the sums are compared between SSDE_MATRIX=f32 and bf16x6 for the GEMM.  GPU only; development tool."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import energy_probe as ep  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

if __name__ == "__main__":
    conv, keep1 = ep.conv_launcher(256, 128, 128, 32, L.TILE_WINOGRAD4)
    reps = 300
    for rnd in range(2):
        for mode in ("f32", "bf16x6"):
            os.environ["SSDE_MATRIX"], os.environ["SSDE_GEMM_PIPE"] = mode, "0"
            gemm, keep2 = ep.gemm_launcher(256, 16, 256, 256)
            for _ in range(20):
                gemm(); conv()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps + 1)]
            ev[0].record()
            for i in range(reps):
                gemm(); ev[2 * i + 1].record()
                conv(); ev[2 * i + 2].record()
            torch.cuda.synchronize()
            tg = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(reps)) / reps
            tc = sum(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(reps)) / reps
            print("GEMM %-7s %.4f ms per launch | the conv_wino4 launch right after it %.4f ms | pair %.4f ms" % (mode, tg, tc, tg + tc), flush=True)
