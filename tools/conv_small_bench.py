#!/usr/bin/env python
"""The image-head convolutions (3x3 onto 4 channels, GroupNorm + SiLU prologue): conv_small.hip against the general direct kernel
(SSDE_CONV_SMALL=0 -> SSDE_CONVF_NO_SMALL_COUT).  GPU only.  usage: conv_small_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_bench as cb  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

for n, cin, h in [(256, 128, 32), (16, 128, 256), (16, 128, 128), (16, 256, 64), (16, 256, 32), (16, 512, 16)]:
    row = []
    for small in ("1", "0", "1", "0"):
        os.environ["SSDE_CONV_SMALL"] = small
        _, ms = cb.time_conv(n, cin, 4, h, L.TILE_AUTO, 1, reps=10)
        row.append("%s %.4f ms" % ("conv_small" if small == "1" else "direct", ms))
    del os.environ["SSDE_CONV_SMALL"]
    gb = n * h * h * (cin + 4) * 4 / 1e9
    print("%d->4@%d n=%d (%.3f GB in + out) | " % (cin, h, n, gb) + " | ".join(row), flush=True)
