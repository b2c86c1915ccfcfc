#!/usr/bin/env python
"""Where do a conv_wino_kernel workgroup's cycles go?  Runs ONE convolution with the trace variant of the library
(built with -DSSDE_WINO_TRACE by score_sde_pytorch_amd/_build.build_variant, loaded through SSDE_LIB_PATH) and prints
the s_memtime deltas of waves 0 (matrix phase 1) and 4 (matrix phase 2) of the first and of the last workgroup.
GPU only; a development tool, not part of the product path."""
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
    buf = torch.zeros(4 * 64, dtype=torch.int64, device="cuda")
    assert lib.ssde_debug_wino_trace(C.c_void_p(buf.data_ptr())) == 0
    for (cin, cout, h, gn) in [(128, 128, 32, 1), (256, 256, 16, 1), (256, 256, 16, 0), (512, 256, 16, 1)]:
        buf.zero_()
        tf, ms = cb.time_conv(256, cin, cout, h, L.TILE_WINOGRAD, gn, reps=1)
        torch.cuda.synchronize()
        t = buf.cpu().numpy().reshape(2, 2, 64)
        print("== %d->%d @%dx%d gn=%d: %.1f TF/s %.3f ms (traced)" % (cin, cout, h, h, gn, tf, ms))
        for blk in range(2):
            for wv in range(2):
                r = t[blk, wv].astype(np.int64)
                if r[0] == 0:
                    continue
                base = r[0]
                def d(a, b):
                    return int(r[b] - r[a]) if r[a] and r[b] else -1
                print(" block %s wave %d: setup %d | fill: to barrier1 %d, to loop %d | loop %d | epilogue: sync %d, transform+xch %d, sync %d, park %d, sync %d, store %d | total %d"
                      % ("first" if blk == 0 else "last", wv * 4, d(0, 1), d(1, 2), d(2, 3), d(3, 40), d(40, 41), d(41, 42), d(42, 43), d(43, 44),
                         d(44, 45), d(45, 46), d(0, 46)))
                rows = []
                for st in range(8):
                    s0 = 4 + st * 4
                    prev = 3 if st == 0 else s0 - 1
                    rows.append("st%d: ph1 work %d wait %d | ph2 work %d wait %d" % (st, d(prev, s0), d(s0, s0 + 1), d(s0 + 1, s0 + 2), d(s0 + 2, s0 + 3)))
                print("   " + "\n   ".join(rows))


if __name__ == "__main__":
    main()
