#!/usr/bin/env python
"""One F(4x4,3x3) layer, a few launches of each form -- the command rocprofv3 wraps for per-layer PMC passes (GPU only).
usage: w4r_layer.py cin cout h [n]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import conv_bench as cb  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

cin, cout, h = (int(v) for v in sys.argv[1:4])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 256
for label, tile, fl in (("fused", L.TILE_WINOGRAD4, 0), ("register-fed", L.TILE_WINOGRAD4R, L.CONVF_V_GIVEN),
                        ("transform+register-fed", L.TILE_WINOGRAD4R, 0)):
    tf, ms = cb.time_conv(n, cin, cout, h, tile, 1, reps=4, flags=fl)
    print("%s %d->%d@%d %.4f ms" % (label, cin, cout, h, ms), flush=True)
