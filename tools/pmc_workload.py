"""A few launches of each matrix kernel at one BASELINE shape (workload of tools/pmc_kernels.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import conv_bench, gemm_bench, wgrad_bench  # noqa: E402
from score_sde_pytorch_amd import _lib as L  # noqa: E402

for gn in (0, 1):
    conv_bench.time_conv(256, 256, 256, 16, L.TILE_WINOGRAD, gn, reps=3)
conv_bench.time_conv(256, 256, 256, 4, L.TILE_AUTO, 1, reps=3)
gemm_bench.time_gemm(256, 16, 256, 256, reps=3)
wgrad_bench.time_wgrad(128, 16, 256, 0, 256, 0, reps=3)
