#!/usr/bin/env python
"""conv_wino4r_kernel alone (V given) with and without a residual input at the big-layer shapes; run once per library
(SSDE_LIB_PATH) for A/B of epilogue changes.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_bench as cb  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

for cin, cout, h in [(128, 128, 32), (256, 128, 32), (256, 256, 16), (512, 256, 16), (256, 256, 32)]:
    row = []
    for resid in (False, True):
        best = min(cb.time_conv(256, cin, cout, h, L.TILE_WINOGRAD4R, 1, reps=20, resid=resid, flags=L.CONVF_V_GIVEN)[1] for _ in range(3))
        row.append(best)
    pair = min(cb.time_conv(256, cin, cout, h, L.TILE_WINOGRAD4R, 1, reps=20, resid=True)[1] for _ in range(3))
    print("%4d->%4d @%2dx%-2d  matrix kernel %.4f ms, + residual %.4f ms; pass + matrix kernel with residual %.4f ms" % (cin, cout, h, h, row[0], row[1], pair), flush=True)
