#!/usr/bin/env python
"""Error budget of Winograd F(4x4,3x3) in fp32 against the 1e-4 forward tolerance of the parity tests (CPU only).

Runs the CPU oracle of the CIFAR-10 NCSN++ (configs/ve/cifar10_ncsnpp_continuous, random-init weights as in bench.py)
three ways -- every stride-1 3x3 convolution as (a) the direct form, (b) F(2x2,3x3), (c) F(4x4,3x3), all in fp32 with
the transforms written out as the kernel would do them (fp32 adds / multiplies, fp32 accumulation over input channels) --
and reports the relative L2 error of each against an fp64 run of the same network.  Test infrastructure: imports oracle/.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402
from oracle import unet_oracle  # noqa: E402
from score_sde_pytorch_amd.models import utils as mutils  # noqa: E402

# F(2x2,3x3) and F(4x4,3x3) matrices (Lavin & Gray 2015; interpolation points 0, +-1, +-2 for the latter)
BT2 = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
G2 = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
AT2 = [[1, 1, 1, 0], [0, 1, -1, -1]]
BT4 = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
G4 = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]
AT4 = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]


def wino_conv(x, w, b, m, dtype):
    """3x3, stride 1, pad 1 through F(m x m, 3x3); every step in `dtype`"""
    BT, G, AT = (BT2, G2, AT2) if m == 2 else (BT4, G4, AT4)
    BT, G, AT = (torch.tensor(t, dtype=torch.float64) for t in (BT, G, AT))
    a = m + 2
    n, c, h, wd = x.shape
    th, tw = -(-h // m), -(-wd // m)
    xp = F.pad(x, (1, 1 + tw * m - wd, 1, 1 + th * m - h))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                        # [n, c, th, tw, a, a]
    U = (G.to(torch.float64) @ w.to(torch.float64) @ G.T).to(dtype)    # weights transformed offline in fp64, stored in dtype
    BTd, ATd = BT.to(dtype), AT.to(dtype)
    V = BTd @ tiles.to(dtype) @ BTd.T                                  # [n, c, th, tw, a, a]
    M = torch.einsum("ncijab,ocab->noijab", V, U)                      # accumulation over c in dtype
    Y = ATd @ M @ ATd.T                                                # [n, o, th, tw, m, m]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], th * m, tw * m)[:, :, :h, :wd]
    return y + b.to(dtype)[None, :, None, None] if b is not None else y


def run(mode, dtype, config, sd, x, sig):
    orig = F.conv2d

    def conv2d(inp, w, b=None, stride=1, padding=0, *a, **k):
        if mode and w.shape[-1] == 3 and stride == 1 and padding == 1 and inp.shape[-1] >= 4:
            return wino_conv(inp, w, b, mode, dtype)
        return orig(inp, w, b, stride, padding, *a, **k)
    F.conv2d = conv2d
    try:
        sdd = {k: v.to(dtype) if v.is_floating_point() else v for k, v in sd.items()}
        return unet_oracle.ncsnpp_forward(config, sdd, x.to(dtype), sig.to(dtype))
    finally:
        F.conv2d = orig


if __name__ == "__main__":
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    config = _util.cfgs.get_config("ve/cifar10_ncsnpp_continuous")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(config)
    sd = dict(_util.load_seeded(model, seed=1)); sd["sigmas"] = model.sigmas.clone()     # bench.py's synthetic weights
    g = torch.Generator().manual_seed(3)
    for name, sig in (("sigma 0.01 / 1 / 50", torch.tensor([0.01, 1.0, 50.0])),):
        x = torch.rand(3, 3, 32, 32, generator=g) + sig[:, None, None, None] * torch.randn(3, 3, 32, 32, generator=g)
        ref = run(0, torch.float64, config, sd, x, sig)
        for label, mode in (("direct fp32", 0), ("F(2x2,3x3) fp32", 2), ("F(4x4,3x3) fp32", 4)):
            out = run(mode, torch.float32, config, sd, x, sig).double()
            rel = ((out - ref).norm() / ref.norm()).item()
            per = [((out[i] - ref[i]).norm() / ref[i].norm()).item() for i in range(3)]
            print("%-18s %s: rel L2 error vs fp64 %.3g  (per image %s)" % (label, name, rel, " ".join("%.3g" % p for p in per)), flush=True)
