#!/usr/bin/env python
"""Where do a conv_wino4g_kernel workgroup's cycles go?  Variant library built with -DSSDE_W4G_TRACE (SSDE_LIB_PATH); s_memtime
deltas of waves 0 and 7 of the first workgroup, stages 0..7.  GPU only; a development tool."""
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
assert lib.ssde_debug_w4g_trace(C.c_void_p(buf.data_ptr())) == 0
os.environ["SSDE_W4G_V_GIVEN"] = "1"
for (cin, cout, h) in [(128, 128, 32), (256, 256, 16), (512, 256, 16)]:
    buf.zero_()
    tf, ms = cb.time_conv(256, cin, cout, h, L.TILE_WINOGRAD4G, 1, reps=1)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(2, 128)
    print("== %d->%d @%dx%d: %.1f TF/s %.3f ms (matrix kernel alone, traced)" % (cin, cout, h, h, tf, ms))
    for wv in range(2):
        r = t[wv].astype(np.int64)
        d = lambda a, b: int(r[b] - r[a]) if r[a] and r[b] else -1   # noqa: E731
        print(" wave %d: setup %d | fill %d | loop %d (%d stages: %.0f per stage) | epilogue half0 %d half1 %d | total %d"
              % (wv * 7, d(0, 1), d(1, 2), d(2, 3), cin // 4, d(2, 3) / (cin // 4), d(3, 4), d(4, 5), d(0, 5)))
        for k in range(8):
            b = 8 + k * 4
            print("   st%d: head + slots 0-4 %d | counted wait + barrier %d | slots 5-8 %d" % (k, d(b, b + 1), d(b + 1, b + 2), d(b + 2, b + 3)))
