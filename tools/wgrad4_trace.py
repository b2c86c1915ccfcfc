#!/usr/bin/env python
"""Where do a wgrad4_gemm_kernel workgroup's cycles go?  Variant library built with -DSSDE_WG4_TRACE (SSDE_LIB_PATH); s_memtime
stamps of wave 0 of the first workgroup: fill, then per K stage (16 tiles): loads issued + 32 MFMAs | LDS stores of the next
stage (waits for its loads) | barrier.  GPU only; a development tool."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import wgrad_bench as wb  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

lib = L.load()
buf = torch.zeros(128, dtype=torch.int64, device="cuda")
assert lib.ssde_debug_wg4_trace(C.c_void_p(buf.data_ptr())) == 0
for (h, c1, c2, co) in [(32, 128, 0, 128), (16, 256, 0, 256), (16, 256, 256, 256)]:
    buf.zero_()
    tf, ms = wb.time_wgrad(128, h, c1, c2, co, 2, reps=1)
    torch.cuda.synchronize()
    r = buf.cpu().numpy().astype(np.int64)
    d = lambda a, b: int(r[b] - r[a]) if r[a] and r[b] else -1   # noqa: E731
    print("== %dx%d cin=%d+%d cout=%d: whole weight gradient %.3f ms" % (h, h, c1, c2, co, ms))
    print("   fill %d | loop %d (%d stages) | slab store %d | total %d" % (d(0, 1), d(1, 2), int(r[100]), d(2, 3), d(0, 3)))
    for st in range(8):
        b = 4 + 4 * st
        print("   st%d: loads issued + 32 MFMAs %d | LDS stores of the next stage %d | barrier %d" % (st, d(b, b + 1), d(b + 1, b + 2), d(b + 2, b + 3)))
