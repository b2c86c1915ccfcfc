#!/usr/bin/env python
"""Which 3x3 kernel wins where (GPU only): direct / F(2x2,3x3) / F(4x4,3x3) at the small-batch shapes of the FFHQ-256 and
training configurations, to set the thresholds of engine.Lowering.wino_ok."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_bench as cb  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

if __name__ == "__main__":
    for n, cin, cout, h in [(16, 256, 256, 32), (16, 512, 256, 32), (16, 256, 256, 16), (16, 512, 256, 16), (16, 256, 256, 64),
                            (128, 256, 256, 8), (128, 512, 256, 8), (128, 256, 256, 16), (64, 256, 256, 16), (32, 128, 128, 32),
                            (8, 128, 128, 256), (8, 256, 256, 64), (8, 256, 256, 32)]:
        d, dms = cb.time_conv(n, cin, cout, h, L.TILE_AUTO, 1)
        w2, w2ms = cb.time_conv(n, cin, cout, h, L.TILE_WINOGRAD, 1)
        w4, w4ms = cb.time_conv(n, cin, cout, h, L.TILE_WINOGRAD4, 1)
        wg2 = -(-(n * h * h) // 256) * -(-cout // 64)
        wg4 = -(-(n * h * h) // 512) * -(-cout // 64)
        best = min((dms, "direct"), (w2ms, "F(2,3)"), (w4ms, "F(4,3)"))[1]
        print("B=%3d %4d->%4d @%3dx%-3d  direct %.3f ms | F(2,3) %.3f ms (%d wgs) | F(4,3) %.3f ms (%d wgs)  -> %s"
              % (n, cin, cout, h, h, dms, w2ms, wg2, w4ms, wg4, best), flush=True)
