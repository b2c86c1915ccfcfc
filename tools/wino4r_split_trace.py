#!/usr/bin/env python
"""Where do the cycles of a SPLIT conv_wino4r_kernel launch go?  (8x8 maps at batch 256, two shares per tile.)  Variant library built
with -DSSDE_W4R_TRACE (SSDE_LIB_PATH); s_memtime deltas of wave 0 of the two workgroups of the first tile: block 0 and block
8 * n_tiles (tickets are dealt in start order, so block 0 is share 0 unless the stamps say otherwise).  GPU only; a development tool."""
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
for (cin, cout, h) in [(256, 256, 8), (512, 256, 8)]:
    rows = {}
    for which in (0, 8 * (cout // 64)):
        buf.zero_()
        buf[255] = which
        tf, ms = cb.time_conv(256, cin, cout, h, L.TILE_WINOGRAD4R, 1, reps=1, resid=True, flags=L.CONVF_V_GIVEN)
        torch.cuda.synchronize()
        rows[which] = (buf.cpu().numpy().reshape(2, 128)[0].astype(np.int64), ms)
    t0 = min(int(r[0]) for r, _ in rows.values() if r[0])
    print("== %d->%d @%dx%d batch 256, two shares: %.4f ms (matrix kernel alone, traced)" % (cin, cout, h, h, list(rows.values())[0][1]))
    for which, (r, _) in rows.items():
        d = lambda a, b: int(r[b] - r[a]) if r[a] and r[b] else -1   # noqa: E731
        last = bool(r[5])                                                 # only the last share reaches the end of the epilogue
        print(" block %3d (%s share), wave 0: set-up %d | first loads issued %d | loop %d (%d stages of this share: %.0f per stage)"
              % (which, "last" if last else "first", d(0, 1), d(1, 2), d(2, 3), cin // 8, d(2, 3) / (cin // 8)))
        for k in range(2):
            base = 3 if k == 0 else (4 if last else 44)
            line = "   round %d: products %d, barrier %d, transform %d, barrier + park %d" % (
                k, d(base, 32 + 4 * k), d(32 + 4 * k, 33 + 4 * k), d(33 + 4 * k, 34 + 4 * k), d(34 + 4 * k, 35 + 4 * k))
            line += ", wait for the previous share %d, sums (loads + adds%s) %d" % (d(35 + 4 * k, 40 + k), "" if last else " + stores", d(40 + k, 42 + k))
            line += (", epilogue store %d" % d(42 + k, 4 + k)) if last else (", stores acknowledged + signal %d" % d(42 + k, 44 + k))
            print(line)
        print("   whole workgroup %d cycles" % (d(0, 5) if last else d(0, 45)))
