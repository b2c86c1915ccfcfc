#!/usr/bin/env python
"""Per-launch HIP-event times of ONE forward + backward of the CIFAR-10 NCSN++ training program at batch 128 (GPU only): every op
with its kind, shape and ms, and the sums per (kind, shape) by time.  The training-step twin of tools/op_times.py."""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _util  # noqa: E402
import op_times as OT  # noqa: E402
from score_sde_pytorch_amd import backward as B, _lib as L  # noqa: E402
from score_sde_pytorch_amd.models import utils as mutils  # noqa: E402


def describe(op):
    k = int(op.kind)
    if k == L.OP_WGRAD:
        a = op.u.wgrad
        return "wgrad%s %4d->%4d @%2dx%-2d s%d flags %d%s" % ("3x3" if a.ksize == 3 else "1x1", a.src.c0 + a.src.c1, a.c_out, a.h_out, a.w_out, a.stride,
                                                             a.flags, " v_pre" if a.v_pre else "")
    if k == L.OP_COLSUM:
        a = op.u.colsum
        return "colsum c %d hw %d%s" % (a.c, a.hw, " per-sample" if a.per_sample else "")
    if k == L.OP_GN_BWD_REDUCE:
        a = op.u.gn_bwd
        return "gn_bwd c %d+%d hw %d slices %d flags %d" % (a.src.c0, a.src.c1, a.hw, a.slices, a.flags)
    if k == L.OP_PROLOGUE_BWD:
        a = op.u.pro_bwd
        return "prologue_bwd c %d+%d hw %d mode %d" % (a.src.c0, a.src.c1, a.hw, a.src.pro_mode)
    if k == L.OP_ATTN_BWD:
        a = op.u.attn_bwd
        return "attn_bwd l=%d c=%d" % (a.l, a.c)
    return OT.describe(op)


if __name__ == "__main__":
    Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda")
    cfg = _util.cfgs.get_config("ve/cifar10_ncsnpp_continuous")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.to(dev).train()
    eng = B.TrainEngine(model, Bn, 32, 32, dev, dropout=True)
    eng.weights.refresh()
    g = torch.Generator().manual_seed(3)
    eng.load_inputs((torch.randn(Bn, 3, 32, 32, generator=g) * 5).to(dev), torch.full((Bn,), 3.0, device=dev))
    eng.gout.tensor.view(-1)[: Bn * 3 * 32 * 32].copy_(torch.randn(Bn * 3 * 32 * 32, generator=g).to(dev))
    prog = eng.program
    prog.run_timed()
    reps = 5
    ms = np.zeros(prog.n)
    for _ in range(reps):
        ms += np.array(prog.run_timed())
    ms /= reps
    print("# %d launches (%d forward, %d backward), %.3f ms: forward %.3f, backward %.3f (HIP events around every launch, program launches)"
          % (prog.n, eng.n_fwd, eng.n_bwd, ms.sum(), ms[:eng.n_fwd].sum(), ms[eng.n_fwd:].sum()))
    groups = collections.OrderedDict()
    for i in range(prog.n):
        d = ("F " if i < eng.n_fwd else "B ") + describe(prog.ops[i])
        print("%4d  %8.4f ms  %s" % (i, ms[i], d))
        e = groups.setdefault(d, [0, 0.0])
        e[0] += 1; e[1] += ms[i]
    print("\n# sums per (kind, shape), by time")
    for d, (cnt, t) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        print("%8.4f ms  %3d x %8.4f  %s" % (t, cnt, t / cnt, d))
