#!/usr/bin/env python
"""The 8x8-map layers of the BASELINE sampler (batch 256): F(2x2,3x3) against the two-kernel F(4x4,3x3) whose register-fed matrix
kernel splits its reduction over 2 / 4 workgroups per tile (conv_wino4r.hip, ssde_conv_wino4r_splits).  GPU only.
SSDE_CONV_KSPLIT=0 -> no split; SSDE_NUM_CUS=512 makes the rule pick four shares where it picks two on the real 256 CUs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_bench as cb  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for cin, cout, h in [(256, 256, 8), (512, 256, 8), (128, 256, 8)]:
    row = []
    for label, tile, env in (("F(2x2)", L.TILE_WINOGRAD, {}), ("4R unsplit", L.TILE_WINOGRAD4R, {"SSDE_CONV_KSPLIT": "0"}),
                             ("4R split", L.TILE_WINOGRAD4R, {}), ("4R split, rule for 512 CUs", L.TILE_WINOGRAD4R, {"SSDE_NUM_CUS": "512"}),
                             ("F(2x2)", L.TILE_WINOGRAD, {}), ("4R split", L.TILE_WINOGRAD4R, {})):
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
