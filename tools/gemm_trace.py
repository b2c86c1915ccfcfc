#!/usr/bin/env python
"""Where do a gemm1x1_kernel workgroup's cycles go?  Variant library built with -DSSDE_GEMM_TRACE (loaded through
SSDE_LIB_PATH); s_memtime deltas of thread 0 of the first workgroup.  GPU only; a development tool."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gemm_bench as gb  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

lib = L.load()
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
assert lib.ssde_debug_gemm_trace(C.c_void_p(buf.data_ptr())) == 0
for h, k, cout, resid in [(16, 256, 256, True), (16, 512, 256, True), (32, 256, 128, False), (16, 256, 768, False)]:
    buf.zero_()
    tf, ms = gb.time_gemm(256, h, k, cout, resid=resid, reps=1)
    torch.cuda.synchronize()
    r = buf.cpu().numpy().astype(np.int64)
    d = lambda a, b: int(r[b] - r[a]) if r[a] and r[b] else -1   # noqa: E731
    print("== M=%d K=%d N=%d resid=%d: %.1f TF/s %.3f ms | setup %d, fill %d, loop %d (%d stages), epilogue: half0 store %d sync %d, half1 store %d sync %d | total %d"
          % (256 * h * h, k, cout, resid, tf, ms, d(0, 1), d(1, 2), d(2, 40), k // 16, d(40, 41), d(41, 42), d(42, 43), d(43, 44), d(0, 44)))
    for st in range(min(8, k // 16)):
        s0 = 4 + st * 4
        prev = 2 if st == 0 else s0 - 2
        print("   st%d: loads+mfma %d | store_stage %d | barrier %d" % (st, d(prev, s0), d(s0, s0 + 1), d(s0 + 1, s0 + 2)))
