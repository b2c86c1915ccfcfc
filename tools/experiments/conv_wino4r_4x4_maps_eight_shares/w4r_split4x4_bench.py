#!/usr/bin/env python
"""The 4x4-map layers of the BASELINE sampler (batch 256): the direct kernel (split over two workgroups per tile) against the
two-kernel F(4x4,3x3) whose register-fed matrix kernel splits its reduction over EIGHT workgroups per tile (one tile per image:
32 workgroup tiles -> 256 workgroups).  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_bench as cb  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for cin, cout, h in [(256, 256, 4), (512, 256, 4)]:
    row = []
    for label, tile, env in (("direct", L.TILE_AUTO, {}), ("4R unsplit", L.TILE_WINOGRAD4R, {"SSDE_CONV_KSPLIT": "0"}),
                             ("4R 8 shares", L.TILE_WINOGRAD4R, {}), ("4R 4 shares", L.TILE_WINOGRAD4R, {"SSDE_NUM_CUS": "128"}),
                             ("direct", L.TILE_AUTO, {}), ("4R 8 shares", L.TILE_WINOGRAD4R, {})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            _, ms = cb.time_conv(n, cin, cout, h, tile, 1, reps=10, resid=True)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        row.append("%s %.4f ms" % (label, ms))
    print("%d->%d@%d n=%d | " % (cin, cout, h, n) + " | ".join(row), flush=True)
