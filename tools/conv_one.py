#!/usr/bin/env python
"""One 3x3 convolution shape, a few launches, for rocprofv3 counter passes: conv_one.py <tile> <cin> <cout> <h> <gn> [batch]"""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import conv_bench as cb  # noqa: E402

if __name__ == "__main__":
    tile, cin, cout, h, gn = (int(v) for v in sys.argv[1:6])
    n = int(sys.argv[6]) if len(sys.argv) > 6 else 256
    tf, ms = cb.time_conv(n, cin, cout, h, tile, gn, reps=5)
    print("tile %d %d->%d @%dx%d gn=%d: %.1f TF/s %.3f ms" % (tile, cin, cout, h, h, gn, tf, ms))
