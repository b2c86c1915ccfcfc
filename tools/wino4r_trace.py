#!/usr/bin/env python
"""Where do a conv_wino4r_kernel workgroup's cycles go?  Variant library built with -DSSDE_W4R_TRACE (SSDE_LIB_PATH); s_memtime
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
assert lib.ssde_debug_w4r_trace(C.c_void_p(buf.data_ptr())) == 0
which = int(os.environ.get("W4R_TRACE_BID", "0"))      # the workgroup to stamp (0 = the first one, on a cold chip)
for (cin, cout, h) in [(128, 128, 32), (256, 256, 16), (512, 256, 16)]:
    buf.zero_()
    buf[255] = which
    tf, ms = cb.time_conv(256, cin, cout, h, L.TILE_WINOGRAD4R, 1, reps=1, flags=L.CONVF_V_GIVEN)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(2, 128)
    print("== %d->%d @%dx%d: %.1f TF/s %.3f ms (matrix kernel alone, traced; workgroup %d)" % (cin, cout, h, h, tf, ms, which))
    for wv in range(2):
        r = t[wv].astype(np.int64)
        d = lambda a, b: int(r[b] - r[a]) if r[a] and r[b] else -1   # noqa: E731
        print(" wave %s: setup %d | first loads issued %d | loop %d (%d stages: %.0f per stage) | epilogue half0 %d half1 %d | total %d"
              % (("0", "last")[wv], d(0, 1), d(1, 2), d(2, 3), cin // 4, d(2, 3) / (cin // 4), d(3, 4), d(4, 5), d(0, 5)))
        print("   epilogue rounds: " + " | ".join("products %d, barrier %d, transform %d, barrier + park %d, store (+ barrier) %d"
              % (d(3 if k == 0 else 4, 32 + 4 * k), d(32 + 4 * k, 33 + 4 * k), d(33 + 4 * k, 34 + 4 * k), d(34 + 4 * k, 35 + 4 * k), d(35 + 4 * k, 4 + k)) for k in range(2)))
        print("   stages 0-7: " + " | ".join("%d + %d" % (d(8 + 2 * k, 9 + 2 * k), d(9 + 2 * k, 10 + 2 * k)) for k in range(7)) + "   (slots 0-5 + slots 6-8)")
