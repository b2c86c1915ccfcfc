#!/usr/bin/env python
"""Where do a conv_wino4x_kernel workgroup's cycles go?  Variant library built with -DSSDE_W4X_TRACE (SSDE_LIB_PATH); s_memtime
deltas of waves 0 and 7 of the first workgroup, stages 4..11.  GPU only; a development tool."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import conv_bench as cb  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

lib = L.load()
buf = torch.zeros(2 * 128, dtype=torch.int64, device="cuda")
assert lib.ssde_debug_w4x_trace(C.c_void_p(buf.data_ptr())) == 0
for (cin, cout, h, gn) in [(128, 128, 32, 1), (128, 128, 32, 0), (256, 256, 16, 1)]:
    buf.zero_()
    tf, ms = cb.time_conv(256, cin, cout, h, L.TILE_WINOGRAD4X, gn, reps=1)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(2, 128)
    print("== %d->%d @%dx%d gn=%d: %.1f TF/s %.3f ms (traced)" % (cin, cout, h, h, gn, tf, ms))
    for wv in range(2):
        r = t[wv].astype(np.int64)
        d = lambda a, b: int(r[b] - r[a]) if r[a] and r[b] else -1   # noqa: E731
        print(" wave %d: setup+fill %d | loop %d | epilogue half0 %d half1 %d | total %d" % (wv * 7, d(0, 2), d(2, 3), d(3, 4), d(4, 5), d(0, 5)))
        for k in range(8):
            b = 8 + k * 10
            prev = b - 2 if k else b + 1
            print("   st%d: head %d | slots 0-1 %d | pass 1 %d | slots 2-3 %d | pass-2 read + slots 4-5 %d | pass 2 %d | slots 6-8 %d | barrier %d"
                  % (k + 4, d(prev, b + 1) if k else -1, d(b + 1, b + 2), d(b + 2, b + 3), d(b + 3, b + 4), d(b + 4, b + 5), d(b + 5, b + 6), d(b + 6, b + 7), d(b + 7, b + 8)))
