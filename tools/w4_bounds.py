#!/usr/bin/env python
"""conv_wino4_kernel at the BASELINE shapes: ms per launch for the library named by SSDE_LIB_PATH (product or a timing-experiment
variant: -DSSDE_W4_EXP_NOBARRIER=1 / -DSSDE_W4_EXP_NOSTORE=1, which compute wrong results on purpose).  GPU only."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import conv_bench as cb  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    out = []
    for cin, cout, h in [(128, 128, 32), (256, 128, 32), (256, 256, 16), (512, 256, 16), (384, 128, 32)]:
        tf, ms = cb.time_conv(n, cin, cout, h, L.TILE_WINOGRAD4, 1, reps=10)
        out.append("%d->%d@%d %.4f ms (%.0f TF/s)" % (cin, cout, h, ms, tf))
        if os.environ.get("W4_BOUNDS_TWO"):                        # + the two-kernel forms: both kernels, the matrix kernel alone
            forms = (("two kernels", L.TILE_WINOGRAD4R, 0),)
            for label, tile, fl in forms:
                tf, ms = cb.time_conv(n, cin, cout, h, tile, 1, reps=10, flags=fl)
                tf2, ms2 = cb.time_conv(n, cin, cout, h, tile, 1, reps=10, flags=fl | L.CONVF_V_GIVEN)
                out[-1] += " ; %s %.4f ms (%.0f TF/s) = transform pass %.4f + matrix kernel %.4f (%.0f TF/s)" % (label, ms, tf, ms - ms2, ms2, tf2)
            out[-1] += "\n"
    print(os.path.basename(os.environ.get("SSDE_LIB_PATH", "product")), " | ".join(out), flush=True)
