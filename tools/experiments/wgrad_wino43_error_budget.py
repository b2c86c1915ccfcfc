#!/usr/bin/env python
"""fp32 rounding of a Winograd F(4x4,3x3) WEIGHT GRADIENT against the direct form and F(2x2,3x3) (CPU, torch):

    dL/dg[co, ci] = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G          (6x6 positions, 4x4 output tiles)

evaluated in fp32 with the transforms written out and the tile sum accumulated in fp32 in chunks (as the kernel's
split-K slabs), against an fp64 evaluation of the direct weight gradient.  Shapes of the CIFAR NCSN++ training step
(batch 128): 128 channels @ 32x32 and 256 channels @ 16x16 -- reduced in channels to keep the CPU time down, the
reduction length (tiles per (co, ci)) is what matters and is kept.  Prints relative L2 / max errors."""
import sys
import torch

torch.manual_seed(0)
BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                   [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                  [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def tiles(x, t, halo):
    """[N, C, H, W] (already padded by 1) -> [N, C, ty, tx, t + halo, t + halo]"""
    return x.unfold(2, t + halo, t).unfold(3, t + halo, t)


def wgrad_wino(x, dy, bt, g, at, t, dtype, chunk):
    n, ci, h, w = x.shape
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1)).to(dtype)
    d = tiles(xp, t, 2)                                               # [N, Ci, ty, tx, t+2, t+2]
    v = torch.einsum("ak,nctukl,bl->nctuab", bt.to(dtype), d, bt.to(dtype))
    z = torch.einsum("ka,nctukl,lb->nctuab", at.to(dtype), tiles(dy.to(dtype), t, 0), at.to(dtype))
    v = v.permute(0, 2, 3, 1, 4, 5).reshape(-1, ci, (t + 2) ** 2)      # [tiles, Ci, pos]
    z = z.permute(0, 2, 3, 1, 4, 5).reshape(-1, dy.shape[1], (t + 2) ** 2)
    m = torch.zeros(dy.shape[1], ci, (t + 2) ** 2, dtype=dtype)
    parts = []
    for s in range(0, v.shape[0], chunk):                             # split-K slabs, each accumulated in `dtype`
        parts.append(torch.einsum("top,tip->oip", z[s:s + chunk], v[s:s + chunk]))
    for p_ in parts:
        m = m + p_
    m = m.reshape(dy.shape[1], ci, t + 2, t + 2)
    return torch.einsum("ak,ocab,bl->ockl", g.to(dtype), m, g.to(dtype))


def direct(x, dy, dtype):
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1)).to(dtype)
    dyd = dy.to(dtype)
    out = torch.zeros(dy.shape[1], x.shape[1], 3, 3, dtype=dtype)
    for k in range(3):
        for l in range(3):
            out[:, :, k, l] = torch.einsum("nohw,nchw->oc", dyd, xp[:, :, k:k + x.shape[2], l:l + x.shape[3]])
    return out


def main():
    for (n, c, h) in [(128, 8, 32), (128, 8, 16), (128, 8, 8)]:
        x = torch.nn.functional.silu(torch.randn(n, c, h, h) * 1.3 + 0.2)      # an activated, normalised tensor
        dy = torch.randn(n, c, h, h) * 1e-3
        ref = direct(x, dy, torch.float64)
        scale = ref.abs().max()
        res = {"direct fp32": direct(x, dy, torch.float32).double(),
               "F(2x2,3x3) fp32": wgrad_wino(x, dy, BT2, G2, AT2, 2, torch.float32, 4096).double(),
               "F(4x4,3x3) fp32": wgrad_wino(x, dy, BT, G, AT, 4, torch.float32, 1024).double(),
               "F(4x4,3x3) fp64 (algebra check)": wgrad_wino(x, dy, BT, G, AT, 4, torch.float64, 1024)}
        print("batch %d, %dx%d maps (%d tiles of 4x4 per (co, ci))" % (n, h, h, n * h * h // 16))
        for k, v in res.items():
            e = v - ref
            print("   %-34s rel L2 %.3e   max-abs / max %.3e" % (k, float(e.norm() / ref.norm()), float(e.abs().max() / scale)))


if __name__ == "__main__":
    main()
