#!/usr/bin/env python
"""Per-launch HIP-event times of ONE U-Net evaluation of the CIFAR-10 NCSN++ sampler at batch 256 (GPU only): every op of the
lowered program with its kind, shape and ms, the list sorted by time, and the sums per (kind, shape).  What to look at next."""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402
from score_sde_pytorch_amd import engine as E, _lib as L  # noqa: E402
from score_sde_pytorch_amd.models import utils as mutils  # noqa: E402

KIND = {v: k for k, v in vars(L).items() if k.startswith("OP_") and isinstance(v, int)}


def describe(op):
    k = int(op.kind)
    if k == L.OP_CONV:
        c = op.u.conv
        cin3, cin1 = c.main.c0 + c.main.c1, c.aux.c0 + c.aux.c1
        return "conv%s %4d%s->%4d @%2dx%-2d tile %d s%d%s%s%s" % ("3x3" if c.ksize else "1x1", cin3 if c.ksize else cin1, ("+%d" % cin1) if c.ksize and cin1 else "",
                                                             c.c_out, c.h_out, c.w_out, c.tile, c.stride, " gn" if (c.main.gn_groups or c.aux.gn_groups) else "",
                                                             " resid" if c.resid else "", " part" if c.gn_part else "")
    if k == L.OP_UPFIRDN:
        a = op.u.fir
        return "fir %d ch %dx%d -> %dx%d up %d down %d%s" % (a.c, a.h_in, a.w_in, a.h_out, a.w_out, a.up, a.down, " pair" if a.dst2 else "")
    if k == L.OP_ATTN:
        return "attn l=%d c=%d" % (op.u.attn.l, op.u.attn.c)
    if k == L.OP_GN_FINALIZE:
        a = op.u.gn_fin
        return "gn_finalize c %d+%d slices %d,%d" % (a.c0, a.c1, a.slices0, a.slices1)
    if k == L.OP_GN_STATS:
        a = op.u.gn
        return "gn_stats c %d+%d hw %d" % (a.c0, a.c1, a.hw)
    return KIND.get(k, str(k))


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda")
    name = sys.argv[2] if len(sys.argv) > 2 else "ve/cifar10_ncsnpp_continuous"      # e.g. 16 ve/ffhq_256_ncsnpp_continuous
    cfg = _util.cfgs.get_config(name)
    R = cfg.data.image_size
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.to(dev).eval()
    eng = E.UNetEngine(model, B, R, R, dev)
    eng.weights.refresh()
    g = torch.Generator().manual_seed(3)
    eng.load_inputs((torch.randn(B, 3, R, R, generator=g) * 5).to(dev), torch.full((B,), 3.0, device=dev))
    prog = eng.program
    prog.run_timed()
    reps = 5
    ms = np.zeros(prog.n)
    for _ in range(reps):
        ms += np.array(prog.run_timed())
    ms /= reps
    print("# %s, batch %d: %d launches, %.3f ms per evaluation (HIP events around every launch, program launches)" % (name, B, prog.n, ms.sum()))
    groups = collections.OrderedDict()
    for i in range(prog.n):
        d = describe(prog.ops[i])
        print("%4d  %8.4f ms  %s" % (i, ms[i], d))
        e = groups.setdefault(d, [0, 0.0])
        e[0] += 1; e[1] += ms[i]
    print("\n# sums per (kind, shape), by time")
    for d, (cnt, t) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        print("%8.4f ms  %3d x %8.4f  %s" % (t, cnt, t / cnt, d))
