#!/usr/bin/env python
"""Go / no-go error budget for fp32 contractions evaluated on the BF16 matrix pipe through a 3-way split (CPU only).

An fp32 value is the exact sum of three bf16 pieces (a = a0 + a1 + a2, 8 significand bits each, round-to-nearest
residuals); a product a*b = sum_ij a_i*b_j with every partial product exact in an fp32 accumulator (8 x 8 significand
bits).  `v_mfma_f32_32x32x16_bf16` runs at 16x the fp32-MFMA rate, so
    x9 : all nine partial products         -> 9/16 of the fp32-MFMA time, product error 0 (only accumulation rounding)
    x6 : a0b0, a0b1, a1b0, a1b1, a0b2, a2b0 -> 6/16 of the time, drops a1b2 + a2b1 + a2b2 ~ 2^-23 |a||b| per product
    x3 : a0b0, a0b1, a1b0                  -> 3/16, drops ~ 2^-15 |a||b| (shown for scale: this is NOT fp32)
This script runs the CPU oracle of the CIFAR-10 NCSN++ (bench.py's synthetic weights) with EVERY contraction the matrix
kernels execute -- 3x3 convolutions (Winograd F(4x4,3x3) products where the production heuristic uses them, here: all
stride-1 3x3 layers on maps >= 4x4, the pessimistic case; direct form elsewhere), 1x1 / NIN / Linear, attention QK^T and
PV -- evaluated through the split, accumulating the partial GEMMs in fp32 from the smallest term to the largest, and
reports the relative L2 error of the score against an fp64 run of the same network at sigma = 0.01 / 1 / 50, next to the
plain-fp32 evaluation of the same structure.  Decision rule: go only if x6 (or x9) is no worse than
the F(4x4,3x3) fp32 path (1.3e-5 / 4.5e-6 / 2.5e-6).  Test infrastructure: imports oracle/.
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _util  # noqa: E402
from oracle import unet_oracle  # noqa: E402
from score_sde_pytorch_amd.models import utils as mutils  # noqa: E402
import wino43_error_budget as w43  # noqa: E402

TERMS = {
    9: [(2, 2), (1, 2), (2, 1), (0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)],
    6: [(0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)],
    3: [(0, 1), (1, 0), (0, 0)],
}


def split3(t):
    """t (fp32) = p0 + p1 + p2 exactly (up to the last bit of p2), every piece representable in bf16"""
    p0 = t.to(torch.bfloat16).to(torch.float32)
    r = t - p0
    p1 = r.to(torch.bfloat16).to(torch.float32)
    p2 = (r - p1).to(torch.bfloat16).to(torch.float32)
    return (p0, p1, p2)


def contract(fn, a, b, nterms):
    """fn(a, b) bilinear; nterms = 0 -> plain fp32"""
    if not nterms or a.dtype != torch.float32:
        return fn(a, b)
    pa, pb = split3(a), split3(b)
    out = None
    for i, j in TERMS[nterms]:
        t = fn(pa[i], pb[j])
        out = t if out is None else out + t
    return out


def wino_conv(x, w, b, nterms):
    BT, G, AT = (torch.tensor(t, dtype=torch.float64) for t in (w43.BT4, w43.G4, w43.AT4))
    m, a = 4, 6
    n, c, h, wd = x.shape
    th, tw = -(-h // m), -(-wd // m)
    xp = F.pad(x, (1, 1 + tw * m - wd, 1, 1 + th * m - h))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)
    U = (G @ w.to(torch.float64) @ G.T).to(x.dtype)
    BTd, ATd = BT.to(x.dtype), AT.to(x.dtype)
    V = BTd @ tiles @ BTd.T
    M = contract(lambda p, q: torch.einsum("ncijab,ocab->noijab", p, q), V, U, nterms)
    Y = ATd @ M @ ATd.T
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], th * m, tw * m)[:, :, :h, :wd]
    return y + b[None, :, None, None] if b is not None else y


def run(wino, nterms, dtype, config, sd, x, sig):
    o_conv, o_einsum, o_linear = F.conv2d, torch.einsum, F.linear

    def conv2d(inp, w, b=None, stride=1, padding=0, *a, **k):
        if w.shape[1] == 1 and w.shape[0] == 1:                      # the FIR filters run on the vector ALUs in fp32
            return o_conv(inp, w, b, stride, padding, *a, **k)
        if wino and w.shape[-1] == 3 and stride == 1 and padding == 1 and inp.shape[-1] >= 4:
            return wino_conv(inp, w, b, nterms)
        y = contract(lambda p, q: o_conv(p, q, None, stride, padding, *a, **k), inp, w, nterms)
        return y + b[None, :, None, None] if b is not None else y

    def einsum(eq, p, q):
        return contract(lambda u, v: o_einsum(eq, u, v), p, q, nterms)

    def linear(inp, w, b=None):
        y = contract(lambda u, v: o_linear(u, v), inp, w, nterms)
        return y + b if b is not None else y

    F.conv2d, torch.einsum, F.linear = conv2d, einsum, linear
    try:
        sdd = {k: v.to(dtype) if v.is_floating_point() else v for k, v in sd.items()}
        return unet_oracle.ncsnpp_forward(config, sdd, x.to(dtype), sig.to(dtype))
    finally:
        F.conv2d, torch.einsum, F.linear = o_conv, o_einsum, o_linear


if __name__ == "__main__":
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    config = _util.cfgs.get_config("ve/cifar10_ncsnpp_continuous")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(config)
    sd = dict(_util.load_seeded(model, seed=1)); sd["sigmas"] = model.sigmas.clone()
    g = torch.Generator().manual_seed(3)
    sig = torch.tensor([0.01, 1.0, 50.0])
    x = torch.rand(3, 3, 32, 32, generator=g) + sig[:, None, None, None] * torch.randn(3, 3, 32, 32, generator=g)
    ref = run(False, 0, torch.float64, config, sd, x, sig)
    print("relative L2 error of the score vs fp64, CIFAR NCSN++, sigma = 0.01 / 1 / 50 (per image)")
    for label, wino, nt in (("direct   fp32", False, 0), ("direct   bf16x9", False, 9), ("direct   bf16x6", False, 6), ("direct   bf16x3", False, 3),
                            ("F(4x4,3x3) fp32", True, 0), ("F(4x4,3x3) bf16x9", True, 9), ("F(4x4,3x3) bf16x6", True, 6), ("F(4x4,3x3) bf16x3", True, 3)):
        out = run(wino, nt, torch.float32, config, sd, x, sig).double()
        per = [((out[i] - ref[i]).norm() / ref[i].norm()).item() for i in range(3)]
        print("%-20s %s" % (label, "  ".join("%.3g" % p for p in per)), flush=True)
