#!/usr/bin/env python
"""The life of every workgroup of one bf16x6 GEMM launch (variant library built with -DSSDE_GEMM_TRACE, loaded through
SSDE_LIB_PATH): start, end of the K loop, end, and where it ran -- how many rounds a launch has, how far the workgroups
of a round run in lock step, how long the matrix pipe idles while a round stores.  GPU only; a development tool."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gemm_bench as gb  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

os.environ["SSDE_MATRIX"] = "bf16x6"
os.environ["SSDE_GEMM_PIPE"] = "0"
lib = L.load()
NWG = 1 << 14
buf = torch.zeros(NWG * 4, dtype=torch.int64, device="cuda")
assert lib.ssde_debug_gemm_wg_trace(C.c_void_p(buf.data_ptr())) == 0
for wide in ("0", "1"):
    os.environ["SSDE_X6_WIDE"] = wide
    for h, k, cout, resid in [(16, 256, 768, False), (16, 256, 256, True), (16, 512, 256, True), (32, 256, 128, True)]:
        gb.time_gemm(256, h, k, cout, resid=resid, reps=2)
        buf.zero_()
        tf, ms = gb.time_gemm(256, h, k, cout, resid=resid, reps=1)     # (time_gemm runs one untimed launch first: the trace holds the last)
        torch.cuda.synchronize()
        r = buf.cpu().numpy().astype(np.int64).reshape(NWG, 4)
        r = r[r[:, 0] != 0]
        t0 = r[:, 0].min()
        start, loop_end, end = r[:, 0] - t0, r[:, 1] - t0, r[:, 2] - t0
        # s_memtime counts at 100 MHz on gfx9 (constant clock); report in microseconds
        us = 1e-2
        print("== wide=%s M=%d K=%d N=%d resid=%d: %.3f ms per launch (%.0f TF/s), %d workgroups; span of the launch %.1f us"
              % (wide, 256 * h * h, k, cout, resid, ms, tf, len(r), end.max() * us))
        life, loop, epi = (end - start) * us, (loop_end - start) * us, (end - loop_end) * us
        print("   per workgroup: life %.1f us (min %.1f max %.1f), loop %.1f, epilogue %.1f (min %.1f max %.1f)"
              % (life.mean(), life.min(), life.max(), loop.mean(), epi.mean(), epi.min(), epi.max()))
        order = np.argsort(start)
        ss = start[order] * us
        # rounds: clusters of start times
        edges = [0] + [i for i in range(1, len(ss)) if ss[i] - ss[i - 1] > 2.0] + [len(ss)]
        print("   start-time clusters (gap > 2 us): " + ", ".join("%d wgs @%.1f-%.1f us" % (edges[i + 1] - edges[i], ss[edges[i]], ss[edges[i + 1] - 1]) for i in range(min(len(edges) - 1, 12))))
        hist, _ = np.histogram(end * us, bins=12, range=(0, end.max() * us))
        print("   ends per 1/12 of the span: " + " ".join(str(x) for x in hist))
        busy = np.zeros(int(end.max() * us) + 2)
        for a_, b_ in zip(start * us, loop_end * us):
            busy[int(a_):int(b_) + 1] += 1
        print("   workgroups inside their K loop per microsecond: " + " ".join("%d" % x for x in busy[:: max(1, len(busy) // 40)]))
