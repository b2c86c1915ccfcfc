#!/usr/bin/env python
"""Micro-benchmark of the 3x3 weight-gradient kernels (Winograd: wgrad_wino.hip, direct: wgrad.hip) at BASELINE shapes.
SSDE_WGRAD_WINOGRAD=0 selects the direct kernel for every launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from score_sde_pytorch_amd import hipops as ops  # noqa: E402


def time_wgrad(n, h, c1, c2, cout, pro=0, reps=5, splits=0, ks=3):
    dev = "cuda"
    xa = torch.randn(n, h, h, c1, device=dev)
    xb = torch.randn(n, h, h, c2, device=dev) if c2 else None
    gy = torch.randn(n, h, h, cout, device=dev)
    k = c1 + c2
    gn = None
    if pro in (1, 2):
        G = min(32, k // 4)
        mean, rstd = ops.groupnorm_stats(xa, G, x2=xb)
        gn = (mean, rstd, torch.ones(k, device=dev), torch.zeros(k, device=dev), G)
    dw = torch.zeros(cout, k, ks, ks, device=dev)
    ops.conv_wgrad(xa, gy, ks, dw, pad=ks // 2, x2=xb, pro=pro, gn=gn, splits=splits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_wgrad(xa, gy, ks, dw, pad=ks // 2, x2=xb, pro=pro, gn=gn, splits=splits)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return 2.0 * ks * ks * k * cout * n * h * h / ms / 1e9, ms


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    if os.environ.get("WG_SPLITS"):
        for sp in [int(v) for v in os.environ["WG_SPLITS"].split(",")]:
            t, ms = time_wgrad(n, 32, 384, 0, 128, 0, splits=sp)
            print("cin=384 cout=128 32x32 splits=%d  %6.1f TF/s (%.3f ms)" % (sp, t, ms), flush=True)
        sys.exit(0)
    if os.environ.get("WG_1X1"):       # the 1x1 / NIN gradients of the CIFAR network (SSDE_WGRAD_1X1_PIPELINED=0: chunked kernel)
        for h, c1, c2, cout in [(32, 256, 0, 128), (32, 256, 128, 128), (16, 256, 0, 256), (16, 256, 0, 768), (16, 256, 256, 256),
                                (16, 384, 0, 256), (8, 256, 256, 256), (4, 256, 256, 256)]:
            for pro in (0, 1):
                for sp in [int(v) for v in os.environ.get("WG_SPLITS_1X1", "0").split(",")]:
                    t, ms = time_wgrad(n, h, c1, c2, cout, pro, ks=1, splits=sp)
                    print("1x1 B=%d %2dx%-2d cin=%3d+%3d cout=%3d pro=%d splits=%d   %6.1f TF/s (%.3f ms)" % (n, h, h, c1, c2, cout, pro, sp, t, ms), flush=True)
        sys.exit(0)
    for h, c1, c2, cout in [(32, 128, 0, 128), (32, 256, 0, 128), (32, 256, 128, 128), (32, 384, 0, 128), (16, 256, 0, 256),
                            (16, 256, 128, 256), (16, 256, 256, 256), (8, 256, 0, 256), (8, 256, 256, 256)]:
        for pro in (0, 2):
            t, ms = time_wgrad(n, h, c1, c2, cout, pro)
            print("B=%d %2dx%-2d cin=%3d+%3d cout=%3d pro=%d   %6.1f TF/s (%.3f ms)" % (n, h, h, c1, c2, cout, pro, t, ms), flush=True)
