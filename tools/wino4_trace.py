#!/usr/bin/env python
"""Where do a conv_wino4_kernel workgroup's cycles go?  Runs ONE convolution with the trace variant of the library
(-DSSDE_W4_TRACE, score_sde_pytorch_amd/_build.build_variant, loaded through SSDE_LIB_PATH) and prints the s_memtime
deltas of waves 0 and 7 of the first workgroup.  GPU only; a development tool, not part of the product path."""
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


def main():
    lib = L.load()
    buf = torch.zeros(2 * 128, dtype=torch.int64, device="cuda")
    assert lib.ssde_debug_w4_trace(C.c_void_p(buf.data_ptr())) == 0
    for (cin, cout, h, gn) in [(128, 128, 32, 1), (128, 128, 32, 0), (256, 256, 16, 1), (512, 256, 16, 1)]:
        buf.zero_()
        tf, ms = cb.time_conv(int(os.environ.get('W4_TRACE_BATCH', '256')), cin, cout, h, L.TILE_WINOGRAD4, gn, reps=1)
        torch.cuda.synchronize()
        t = buf.cpu().numpy().reshape(2, 128)
        print("== %d->%d @%dx%d gn=%d: %.1f TF/s %.3f ms (traced)" % (cin, cout, h, h, gn, tf, ms))
        for wv in range(2):
            r = t[wv].astype(np.int64)
            if r[0] == 0:
                continue

            def d(a, b):
                return int(r[b] - r[a]) if r[a] and r[b] else -1
            if r[110] and r[111] and r[111] > r[110]:
                print(" wave %d: shader clock during the workgroup = %.0f MHz (s_memtime %d ticks over %d ticks of the 100 MHz counter)"
                      % (wv * 7, 100.0 * (r[5] - r[0]) / (r[111] - r[110]), r[5] - r[0], r[111] - r[110]))
            print(" wave %d: setup %d | fill %d | loop %d | epilogue half0 %d half1 %d | total %d"
                  % (wv * 7, d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(0, 5)))
            print("   epilogue half 0: M write %d | sync %d | transform %d | sync %d | park %d | sync %d | store %d | sync %d"
                  % (d(3, 100), d(100, 101), d(101, 102), d(102, 103), d(103, 104), d(104, 105), d(105, 106), d(106, 4)))
            for st in range(8):
                b = 8 + st * 10
                prev = 2 if st == 0 else b - 2
                print("   st%d: head (waits, reads, prologue) %d | slots 0-1 %d | pass 1 %d | slots 2-3 %d | slots 4-5 + pass 2 %d | slots 6-8 %d | barrier %d"
                      % (st, d(prev, b + 1), d(b + 1, b + 2), d(b + 2, b + 3), d(b + 3, b + 4), d(b + 4, b + 5), d(b + 5, b + 6),
                         d(b + 6, b + 8)))

if __name__ == "__main__":
    main()
