#!/usr/bin/env python
"""Attention forward at the sampler's shape (GPU only): the fp32-MFMA kernel against the BF16-pipe kernel (attn_x6_kernel), HIP events."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from score_sde_pytorch_amd import hipops as ops  # noqa: E402


def time_mode(qkv, c, x6, reps=50):
    os.environ["SSDE_MATRIX"] = "bf16x6"
    os.environ["SSDE_ATTN_X6"] = "1" if x6 else "0"
    for _ in range(5):
        ops.attention(qkv, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.attention(qkv, c)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if __name__ == "__main__":
    for n, c in [(256, 256), (128, 256), (64, 256), (256, 128), (256, 64)]:
        qkv = torch.randn(n, 256, 3 * c, device="cuda")
        t32, t6 = time_mode(qkv, c, False), time_mode(qkv, c, True)
        fl = 4.0 * n * 256 * 256 * c
        print("N=%3d L=256 C=%3d  fp32 kernel %.4f ms (%.1f TF/s)   x6 kernel %.4f ms (%.1f TF/s)" % (n, c, t32, fl / t32 / 1e9, t6, fl / t6 / 1e9), flush=True)
