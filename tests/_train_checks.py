"""Training-path parity checks shared by the CPU-emulated suite (tests/test_emulated_training.py) and the
GPU suite (tests/test_train_gpu.py).  Every check compares libssde kernels, called through the C ABI,
with torch CPU autograd over plain fp32 torch ops / the CPU oracle (oracle/unet_oracle.py is functional
torch code, so `backward()` on its output IS the reference gradient: the reference itself trains by
autograd over the same ops, losses.py:196).

Tolerances (fp32; max-abs error / max-abs reference value per tensor):
  TOL_OP   2e-5   single kernels (contractions over up to ~1e5 terms)
  TOL_GRAD 2e-4   whole-network parameter / input gradients (tensors whose reference gradient is
                  numerically zero -- e.g. the key bias of attention, NIN_1.b -- are compared absolutely)
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

import _util
from _util import rel_err

TOL_OP = 2e-5
TOL_GRAD = 2e-4


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def hash_keep(elem, key, thresh, scale):
    """numpy restatement of ssde_keep (csrc/ssde_common.h): murmur3 finaliser on elem*0x9E3779B1 + key."""
    x = (elem.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64(key)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return np.where(x >= np.uint64(thresh), np.float32(scale), np.float32(0.0)).astype(np.float32)


def check_backward_ops(dev):
    from score_sde_pytorch_amd import hipops as ops, _lib as L
    g = torch.Generator().manual_seed(5)
    d = lambda t: t.to(dev)  # noqa: E731
    # ---- 3x3 weight gradient with GroupNorm+SiLU prologue and a concatenated source
    for (n, c0, c1, cout, h, splits) in [(3, 32, 32, 64, 8, 0), (2, 16, 0, 40, 16, 3), (5, 8, 0, 8, 4, 0)]:
        x1 = torch.randn(n, c0, h, h, generator=g)
        x2 = torch.randn(n, c1, h, h, generator=g) if c1 else None
        xc = torch.cat([x1, x2], 1) if c1 else x1
        C = c0 + c1
        G = min(C // 4, 32)
        gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
        w = (torch.randn(cout, C, 3, 3, generator=g) / np.sqrt(9 * C)).requires_grad_()
        gout = torch.randn(n, cout, h, h, generator=g)
        F.conv2d(F.silu(F.group_norm(xc, G, gamma, beta, 1e-6)), w, padding=1).backward(gout)
        mean, rstd = ops.groupnorm_stats(d(nhwc(x1)), G, 1e-6, x2=d(nhwc(x2)) if c1 else None)
        dw = torch.zeros(cout, C, 3, 3, device=dev)
        ops.conv_wgrad(d(nhwc(x1)), d(nhwc(gout)), 3, dw, x2=d(nhwc(x2)) if c1 else None, pro=L.PRO_GN_SILU,
                       gn=(mean, rstd, d(gamma), d(beta), G), splits=splits, scale=0.5)
        assert rel_err(dw, 0.5 * w.grad) < TOL_OP, ("wgrad3", n, c0, c1, cout, h)
    # ---- 1x1 weight gradient on a column slice, plain and transposed (NIN)
    x = torch.randn(2, 48, 8, 8, generator=g)
    w = torch.randn(72, 48, generator=g).requires_grad_()
    gout = torch.randn(2, 8, 8, 80, generator=g)
    torch.einsum('nchw,oc->nhwo', x, w).backward(gout[..., 4:76].contiguous())
    dw = torch.zeros(72, 48, device=dev)
    ops.conv_wgrad(d(nhwc(x)), d(gout), 1, dw, pad=0, g_off=4, c_out=72)
    assert rel_err(dw, w.grad) < TOL_OP
    dwt = torch.zeros(48, 72, device=dev)
    ops.conv_wgrad(d(nhwc(x)), d(gout), 1, dwt, pad=0, g_off=4, c_out=72, transpose_out=True)
    assert rel_err(dwt, w.grad.t()) < TOL_OP
    # ---- strided VALID conv (conv_downsample_2d) with a padded input channel
    x = torch.randn(2, 4, 17, 17, generator=g)
    x[:, 3] = 0
    w = torch.randn(24, 3, 3, 3, generator=g).requires_grad_()
    xr = x[:, :3].clone().requires_grad_()
    gout = torch.randn(2, 24, 8, 8, generator=g)
    F.conv2d(xr, w, stride=2).backward(gout)
    dw = torch.zeros(24, 3, 3, 3, device=dev)
    ops.conv_wgrad(d(nhwc(x)), d(nhwc(gout)), 3, dw, stride=2, pad=0, cin_store=3)
    assert rel_err(dw, w.grad) < TOL_OP
    # its input gradient: zero insertion + stride-1 conv with the rotated, transposed weights, cropped
    gz = ops.upfirdn2d_nhwc(d(nhwc(gout)), torch.ones(1, 1), up=2, pad=(0, 0))
    wd = torch.nn.functional.pad(w.detach(), (0, 0, 0, 0, 0, 1)).permute(1, 0, 2, 3).flip(2, 3).contiguous()
    dx = ops.conv2d(gz, wd, None, pad=2, out_hw=(17, 17))
    assert rel_err(nchw(dx.cpu())[:, :3], xr.grad) < TOL_OP
    # ---- stride-1 input gradient = forward kernel with rotated weights, accumulating in place
    x = torch.randn(2, 32, 8, 8, generator=g).requires_grad_()
    w = torch.randn(64, 32, 3, 3, generator=g) / 17
    go = torch.randn(2, 64, 8, 8, generator=g)
    F.conv2d(x, w, padding=1).backward(go)
    base = torch.randn(2, 8, 8, 32, generator=g)
    buf = d(base.clone())
    a = L.ConvArgs()
    # (through the wrapper: resid given, resid_post exercised by the network-level checks)
    dx = ops.conv2d(d(nhwc(go)), w.permute(1, 0, 2, 3).flip(2, 3).contiguous(), None, scale=0.7)
    assert rel_err(nchw(dx.cpu()), 0.7 * x.grad) < TOL_OP
    # ---- column sums
    gg = torch.randn(3, 8, 8, 40, generator=g)
    per, tot, tot2 = torch.zeros(3, 100, device=dev), torch.zeros(40, device=dev), torch.zeros(40, device=dev)
    ops.colsum(d(gg), scale=0.7, per_sample=per, ps_off=20, total=tot, total2=tot2)
    assert rel_err(per[:, 20:60], 0.7 * gg.sum((1, 2))) < TOL_OP and rel_err(tot, 0.7 * gg.sum((0, 1, 2))) < TOL_OP
    assert torch.equal(tot, tot2)
    tot = torch.zeros(3, device=dev)
    gg4 = torch.randn(2, 4, 4, 4, generator=g)
    ops.colsum(d(gg4), c=3, total=tot)
    assert rel_err(tot, gg4.sum((0, 1, 2))[:3]) < TOL_OP
    # several pixel slices per sample, a ragged channel count, both destinations (the one-launch form of round 4 lives on as
    # tools/experiments/colsum_fused.patch)
    gb = torch.randn(5, 16, 16, 72, generator=g)
    for _once in (0,):
        per, tot = torch.zeros(5, 80, device=dev), torch.zeros(68, device=dev)
        ops.colsum(d(gb), c=68, g_off=4, scale=1.3, per_sample=per, ps_off=8, total=tot)
        assert rel_err(per[:, 8:76], 1.3 * gb[..., 4:].sum((1, 2))) < TOL_OP and rel_err(tot, 1.3 * gb[..., 4:].sum((0, 1, 2))) < TOL_OP
        tot1 = torch.zeros(72, device=dev)
        ops.colsum(d(gb), scale=0.5, total=tot1)                 # total only: the slices cut the rows of the whole batch
        assert rel_err(tot1, 0.5 * gb.sum((0, 1, 2))) < TOL_OP
        for _ in range(2):                                       # fixed summation order: the same launch again, bit for bit
            tot2 = torch.zeros(72, device=dev)
            ops.colsum(d(gb), scale=0.5, total=tot2)
            assert torch.equal(tot2, tot1)
        # the same column sums DEFERRED (ssde_colsum with SSDE_COLSUMF_DEFER leaves the pixel-slice partials behind) and finished by
        # ONE ssde_colsum_finish launch for all of them -- jobs of different widths, sample counts and destinations: bit for bit
        # what the per-call kernels wrote
        perd, totd, tot1d = torch.zeros(5, 80, device=dev), torch.zeros(68, device=dev), torch.zeros(72, device=dev)
        per3, tot3, tot3b = torch.zeros(3, 100, device=dev), torch.zeros(40, device=dev), torch.zeros(40, device=dev)
        tot4 = torch.zeros(3, device=dev)
        jobs = [ops.colsum(d(gb), c=68, g_off=4, scale=1.3, per_sample=perd, ps_off=8, total=totd, defer=True),
                ops.colsum(d(gb), scale=0.5, total=tot1d, defer=True),
                ops.colsum(d(gg), scale=0.7, per_sample=per3, ps_off=20, total=tot3, total2=tot3b, defer=True),
                ops.colsum(d(gg4), c=3, total=tot4, defer=True),
                ops.colsum(d(gg), scale=0.7, per_sample=per3, ps_off=60, defer=True)]
        ops.colsum_finish(jobs)
        assert torch.equal(perd, per) and torch.equal(totd, tot) and torch.equal(tot1d, tot1)
        # 37 samples: a sample lane of the finishing kernel owns five of them (four per trip of its loop, then one)
        gm = torch.randn(37, 8, 8, 12, generator=g)
        perm, totm = torch.zeros(37, 12, device=dev), torch.zeros(12, device=dev)
        ops.colsum(d(gm), scale=1.1, per_sample=perm, total=totm)
        permd, totmd = torch.zeros(37, 12, device=dev), torch.zeros(12, device=dev)
        ops.colsum_finish([ops.colsum(d(gm), scale=1.1, per_sample=permd, total=totmd, defer=True)])
        assert torch.equal(permd, perm) and torch.equal(totmd, totm)
        assert rel_err(totm, 1.1 * gm.sum((0, 1, 2))) < TOL_OP
        per3r, tot3r = torch.zeros(3, 100, device=dev), torch.zeros(40, device=dev)
        ops.colsum(d(gg), scale=0.7, per_sample=per3r, ps_off=20, total=tot3r)
        ops.colsum(d(gg), scale=0.7, per_sample=per3r, ps_off=60)
        tot4r = torch.zeros(3, device=dev)
        ops.colsum(d(gg4), c=3, total=tot4r)
        assert torch.equal(per3, per3r) and torch.equal(tot3, tot3r) and torch.equal(tot3b, tot3r) and torch.equal(tot4, tot4r)
    # ---- GroupNorm + SiLU backward over a concatenated source (group straddles nothing; 12 groups of 4)
    n, c0, c1, h = 3, 32, 16, 8
    x1 = torch.randn(n, c0, h, h, generator=g).requires_grad_()
    x2 = torch.randn(n, c1, h, h, generator=g).requires_grad_()
    C, G = c0 + c1, 12
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).requires_grad_()
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_()
    dp = torch.randn(n, C, h, h, generator=g)
    F.silu(F.group_norm(torch.cat([x1, x2], 1), G, gamma, beta, 1e-6)).backward(dp)
    a1, a2 = d(nhwc(x1.detach())), d(nhwc(x2.detach()))
    mean, rstd = ops.groupnorm_stats(a1, G, 1e-6, x2=a2)
    gn_t = (mean, rstd, d(gamma.detach()), d(beta.detach()), G)
    for slices in (1, 4):
        dx, dx2, dga, dbe = ops.gn_backward(a1, d(nhwc(dp)), gn_t, L.PRO_GN_SILU, x2=a2, slices=slices, one_call=False)
        assert rel_err(nchw(dx.cpu()), x1.grad) < TOL_OP and rel_err(nchw(dx2.cpu()), x2.grad) < TOL_OP
        assert rel_err(dga, gamma.grad) < TOL_OP and rel_err(dbe, beta.grad) < TOL_OP
    # the same through ONE call (ABI 7): the one-pass kernel with every group in one workgroup / ragged runs of 5, 5, 2 groups /
    # one group per workgroup, and the library's own three-kernel sequence; then accumulation, a scale, one source without gradient
    for env in ({"SSDE_NUM_CUS": "1"}, {"SSDE_NUM_CUS": "4"}, {}, {"SSDE_GN_BWD_FUSED": "0"}):
        os.environ.update(env)
        try:
            dx, dx2, dga, dbe = ops.gn_backward(a1, d(nhwc(dp)), gn_t, L.PRO_GN_SILU, x2=a2, slices=4)
            assert rel_err(nchw(dx.cpu()), x1.grad) < TOL_OP and rel_err(nchw(dx2.cpu()), x2.grad) < TOL_OP, env
            assert rel_err(dga, gamma.grad) < TOL_OP and rel_err(dbe, beta.grad) < TOL_OP, env
            # dgamma / dbeta deferred to a finishing launch (SSDE_GNBWDF_DEFER_PARAMS + ssde_gn_bwd_finish): the same bits
            dxd, dx2d, dgad, dbed = ops.gn_backward(a1, d(nhwc(dp)), gn_t, L.PRO_GN_SILU, x2=a2, slices=4, defer_params=True)
            assert torch.equal(dxd, dx) and torch.equal(dx2d, dx2) and torch.equal(dgad, dga) and torch.equal(dbed, dbe), env
            base = torch.randn(n, h, h, c1, generator=g)
            acc2 = d(base.clone())
            dx, dx2, dga, dbe = ops.gn_backward(a1, d(nhwc(dp)), gn_t, L.PRO_GN_SILU, x2=a2, scale=0.25, acc=(False, True), want=(False, True),
                                                dx2=acc2)
            assert dx is None and rel_err(nchw(dx2.cpu()), nchw(base) + 0.25 * x2.grad) < TOL_OP, env
            assert rel_err(dga, gamma.grad) < TOL_OP and rel_err(dbe, beta.grad) < TOL_OP, env
        finally:
            for k in env:
                os.environ.pop(k)
    # a 32x32 map (8 pixels of a channel quad per thread), plain GroupNorm, one source; and a map too large for the one-pass
    # kernel (96x96 at 4 channels per group: 9 pixels per thread) that the same call runs as three kernels
    for (n_, c_, g_, h_, mode) in [(2, 32, 8, 32, L.PRO_GN), (1, 8, 2, 96, L.PRO_GN_SILU)]:
        xb = torch.randn(n_, c_, h_, h_, generator=g).requires_grad_()
        gb_ = (1 + 0.1 * torch.randn(c_, generator=g)).requires_grad_()
        bb_ = (0.1 * torch.randn(c_, generator=g)).requires_grad_()
        dpb = torch.randn(n_, c_, h_, h_, generator=g)
        yb = F.group_norm(xb, g_, gb_, bb_, 1e-6)
        (F.silu(yb) if mode == L.PRO_GN_SILU else yb).backward(dpb)
        ab = d(nhwc(xb.detach()))
        mb, rb = ops.groupnorm_stats(ab, g_, 1e-6)
        dx, _, dga, dbe = ops.gn_backward(ab, d(nhwc(dpb)), (mb, rb, d(gb_.detach()), d(bb_.detach()), g_), mode, slices=2)
        assert rel_err(nchw(dx.cpu()), xb.grad) < TOL_OP, (h_, rel_err(nchw(dx.cpu()), xb.grad))
        assert rel_err(dga, gb_.grad) < 2 * TOL_OP and rel_err(dbe, bb_.grad) < 2 * TOL_OP, h_
    # ---- attention backward
    for (n, Lt, Cc) in [(2, 64, 32), (1, 16, 64), (2, 256, 32)]:
        qkv = torch.randn(n, Lt, 3 * Cc, generator=g).requires_grad_()
        q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]
        o = torch.softmax(q @ k.transpose(1, 2) * Cc ** -0.5, -1) @ v
        do = torch.randn(n, Lt, Cc, generator=g)
        o.backward(do)
        dqkv = ops.attention_bwd(d(qkv.detach()), d(o.detach().contiguous()), d(do), Cc)
        assert rel_err(dqkv, qkv.grad) < TOL_OP, ("attention_bwd", n, Lt, Cc)
    # ---- DSM loss head
    sc = torch.randn(4, 3, 8, 8, generator=g).requires_grad_()
    z = torch.randn(4, 3, 8, 8, generator=g)
    s = torch.rand(4, generator=g) + 0.5
    g2 = torch.rand(4, generator=g) + 0.5
    for rm in (False, True):
        for lw in (False, True):
            sb = s[:, None, None, None]
            losses = torch.square(sc + z / sb) if lw else torch.square(sc * sb + z)
            losses = losses.reshape(4, -1)
            losses = losses.mean(-1) if rm else 0.5 * losses.sum(-1)
            if lw:
                losses = losses * g2
            loss = losses.mean()
            sc.grad = None
            loss.backward()
            l, ls, ds = ops.dsm_loss(d(sc.detach()), d(z), d(s), g2=d(g2) if lw else None, reduce_mean=rm, likelihood_weighting=lw)
            assert rel_err(l, loss.detach().reshape(1)) < 2e-6 and rel_err(ds, sc.grad) < 2e-6 and rel_err(ls, losses.detach()) < 2e-6
    xt = ops.perturb(d(z), d(z * 0.5), d(s), a_coef=d(g2))
    assert rel_err(xt, g2[:, None, None, None] * z + s[:, None, None, None] * z * 0.5) < 2e-6
    # ---- clip + Adam + EMA against torch.optim.Adam / clip_grad_norm_ / the reference EMA rule
    P = torch.randn(1003, generator=g)
    Gd = torch.randn(1003, generator=g) * 3
    p_ref = P.clone().requires_grad_()
    opt = torch.optim.Adam([p_ref], lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    m, v, ema, pp, ema_ref = (torch.zeros(1003, device=dev), torch.zeros(1003, device=dev), d(P.clone()), d(P.clone()), P.clone())
    for step in range(1, 4):
        p_ref.grad = Gd.clone()
        torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
        opt.step()
        decay = min(0.999, (1 + step) / (10 + step))
        ema_ref.sub_((1 - decay) * (ema_ref - p_ref.detach()))
        hyper = d(torch.tensor([2e-4, 0.9, 0.999, 1e-8, 0.0, 1.0, 1 - 0.9 ** step, (1 - 0.999 ** step) ** 0.5, 1 - decay, 0, 0, 0]))
        ops.adam_clip_ema(pp, d(Gd), m, v, ema, hyper, ops.sumsq_flat(d(Gd)))
        assert rel_err(pp, p_ref.detach()) < 1e-6 and rel_err(ema, ema_ref) < 1e-6


def check_wgrad_1x1(dev):
    """1x1 / NIN weight gradients on wgrad1x1_gemm_kernel (software-pipelined, the default) and on the chunked
    wgrad_kernel<1,2,2> (SSDE_WGRAD_1X1_PIPELINED=0): several (co, ci) tiles, pixel splits with ragged last stages, concatenated
    sources, GroupNorm (+SiLU) prologue, the regenerated dropout mask, column slices of the gradient and the transposed
    (NIN) destination -- against torch autograd over the same arithmetic."""
    from score_sde_pytorch_amd import hipops as ops, _lib as L
    g = torch.Generator().manual_seed(11)
    d = lambda t: t.to(dev)                                                                           # noqa: E731
    cases = [  # n, c0, c1, cout, h, pro, splits, dropout
        (3, 96, 64, 136, 12, L.PRO_GN_SILU, 0, False),
        (2, 160, 0, 72, 10, L.PRO_GN, 3, False),
        (5, 64, 32, 260, 6, L.PRO_NONE, 2, False),
        (3, 128, 0, 128, 8, L.PRO_GN_SILU, 4, True),
        (2, 32, 0, 40, 5, L.PRO_SILU, 0, False),
    ]
    for (n, c0, c1, cout, h, pro, splits, drop) in cases:
        C = c0 + c1
        x = (torch.randn(n, C, h, h, generator=g) * 1.3 + 0.2).requires_grad_()
        w = (torch.randn(cout, C, generator=g) / np.sqrt(C)).requires_grad_()
        gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
        G = 32 if C % 32 == 0 and (C // 32) % 4 == 0 else C // 4
        gfull = torch.randn(n, h, h, cout + 8, generator=g)
        u = x
        if pro in (L.PRO_GN, L.PRO_GN_SILU):
            u = F.group_norm(u, G, gamma, beta, 1e-6)
        if pro in (L.PRO_SILU, L.PRO_GN_SILU):
            u = F.silu(u)
        xa = d(nhwc(x.detach()))
        seed_t, salt, pdrop = torch.tensor([0x1234567], dtype=torch.int32, device=dev), 77, 0.25
        if drop:
            thresh = min(int(round(pdrop * 2.0 ** 32)), 2 ** 32 - 1)
            keep = hash_keep(np.arange(n * h * h * C, dtype=np.uint64), (0x1234567 ^ salt) & 0xFFFFFFFF, thresh, 1.0 / (1.0 - pdrop))
            u = u * torch.from_numpy(keep.reshape(n, h, h, C)).permute(0, 3, 1, 2)
        torch.einsum('nchw,oc->nhwo', u, w).backward(gfull[..., 4:4 + cout].contiguous())
        gn = None
        if pro in (L.PRO_GN, L.PRO_GN_SILU):
            mean, rstd = ops.groupnorm_stats(xa, G, 1e-6)
            gn = (mean, rstd, d(gamma), d(beta), G)
        x0 = xa[..., :c0].contiguous()
        x1 = xa[..., c0:].contiguous() if c1 else None
        for mode in ("1", "0"):
            os.environ["SSDE_WGRAD_1X1_PIPELINED"] = mode
            try:
                for tr in (False, True):
                    dw = torch.full((C, cout) if tr else (cout, C), 0.5, device=dev)
                    ops.conv_wgrad(x0, d(gfull), 1, dw, pad=0, x2=x1, pro=pro, gn=gn, g_off=4, c_out=cout, transpose_out=tr, scale=0.5,
                                   splits=splits, dropout=(pdrop, seed_t, salt) if drop else None)
                    ref = 0.5 + 0.5 * (w.grad.t() if tr else w.grad)
                    assert rel_err(dw, ref) < TOL_OP, (n, c0, c1, cout, h, pro, splits, drop, mode, tr)
            finally:
                del os.environ["SSDE_WGRAD_1X1_PIPELINED"]


def check_upfirdn_tiles(dev):
    """The LDS-staged FIR kernel (channels % 32 == 0, 4x4 taps) in all three modes and their gradient shapes, with
    ragged sizes (partial tiles, odd maps), an asymmetric kernel (flip), the GroupNorm+SiLU prologue applied once in LDS,
    the dual output (act(GN(x)) and x filtered in one launch) and accumulation, against the oracle's restatement of
    upfirdn2d_native (op/upfirdn2d.py:159-200).  Tolerance 2e-6 (16-term sums) / 1e-5 with the prologue."""
    from score_sde_pytorch_amd import hipops as ops, _lib as L
    from oracle import unet_oracle as uo
    g = torch.Generator().manual_seed(31)
    kasym = (torch.arange(16, dtype=torch.float32).reshape(4, 4) + 1) / 30.0
    ksym = torch.tensor(uo.setup_fir_kernel([1, 3, 3, 1]))
    d = lambda t: t.to(dev)   # noqa: E731
    for (n, c, h, w) in [(2, 32, 5, 7), (1, 64, 20, 12), (3, 32, 16, 16), (2, 32, 1, 1)]:
        x = torch.randn(n, c, h, w, generator=g)
        for up, down, pad in [(2, 1, (2, 1)), (1, 2, (1, 1)), (1, 1, (2, 2)), (1, 1, (1, 2)), (2, 1, (1, 2))]:
            if (h * up + pad[0] + pad[1] - 4) // down + 1 < 1 or (w * up + pad[0] + pad[1] - 4) // down + 1 < 1:
                continue
            for k in (kasym, ksym * up * up):
                y = ops.upfirdn2d_nhwc(d(nhwc(x)), k, up=up, down=down, pad=pad)
                ref = uo.upfirdn2d(x, k, up=up, down=down, pad=pad)
                assert tuple(nchw(y.cpu()).shape) == tuple(ref.shape), (up, down, pad)
                assert rel_err(nchw(y.cpu()), ref) < 2e-6, (n, c, h, w, up, down, pad)
    # prologue once in LDS + dual output + accumulate
    x = torch.randn(2, 64, 12, 20, generator=g) + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(64, generator=g), 0.1 * torch.randn(64, generator=g)
    xd = d(nhwc(x))
    mean, rstd = ops.groupnorm_stats(xd, 16)
    act = F.silu(F.group_norm(x, 16, gamma, beta, 1e-6))
    for up, down, pad, gain in [(2, 1, (2, 1), 4.0), (1, 2, (1, 1), 1.0)]:
        k = ksym * gain
        y, y2 = ops.upfirdn2d_nhwc(xd, k, up=up, down=down, pad=pad, pro=L.PRO_GN_SILU, gn=(mean, rstd, d(gamma), d(beta), 16),
                                   dual=True)
        assert rel_err(nchw(y.cpu()), uo.upfirdn2d(act, k, up=up, down=down, pad=pad)) < 1e-5
        assert rel_err(nchw(y2.cpu()), uo.upfirdn2d(x, k, up=up, down=down, pad=pad)) < 2e-6
        base = torch.randn(y.shape, generator=g)
        acc = d(base.clone())
        ops.upfirdn2d_nhwc(xd, k, up=up, down=down, pad=pad, accumulate_into=acc)
        assert rel_err(acc.cpu(), base + nhwc(uo.upfirdn2d(x, k, up=up, down=down, pad=pad))) < 2e-6


def check_fused_gn_statistics(dev, monkeypatch):
    """GroupNorm statistics from the producers' epilogues (ssde_conv_args.gn_part + ssde_gn_finalize) against the
    standalone reduction kernel (SSDE_GN_FUSE=0) on the same network: the lowered program must contain finalize ops,
    every (mean, rstd) pair must match the standalone kernel's to 1e-5, and both outputs must match the oracle."""
    import ctypes as C
    from score_sde_pytorch_amd import engine as E, _lib as L
    from score_sde_pytorch_amd.models import utils as mutils
    from oracle import unet_oracle
    for ch_mult, attn, n_min in (((1, 2, 2), (16,), 20), ((1, 2, 2, 2), (16, 4), 28)):
        _fused_gn_case(dev, monkeypatch, ch_mult, attn, n_min)


def _fused_gn_case(dev, monkeypatch, ch_mult, attn, n_min):
    """3 images (a batch that does not fill the multi-image tiles of the 8x8 / 4x4 maps), maps from 32x32 down to 4x4:
    tiles that are part of one image, tiles holding several whole images, and the tail guard"""
    import ctypes as C
    from score_sde_pytorch_amd import engine as E, _lib as L
    from score_sde_pytorch_amd.models import utils as mutils
    from oracle import unet_oracle
    cfg = _util.small_config("ncsnpp", image_size=32, ch_mult=ch_mult, num_res_blocks=2, attn=attn)
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    sd = dict(_util.load_seeded(model, seed=1)); sd["sigmas"] = model.sigmas.clone()
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(21)
    x = torch.rand(3, 3, 32, 32, generator=g) + 2 * torch.randn(3, 3, 32, 32, generator=g)
    sig = torch.tensor([0.05, 1.3, 20.0])
    lib = L.load()
    cuda = str(dev).startswith("cuda")
    outs, seqs, counts = {}, {}, {}
    # "consumer": the transform pass of a two-kernel F(4x4,3x3) convolution merges the partials of its source itself
    # (ssde_conv_args.gn_in_part0, the default); "launch": one ssde_gn_finalize launch per normalised tensor
    # (SSDE_GN_MERGE_IN_CONSUMER=0); "0": the standalone reduction kernel over the activations (SSDE_GN_FUSE=0)
    for mode in ("consumer", "launch", "0"):
        monkeypatch.setenv("SSDE_GN_FUSE", "0" if mode == "0" else "1")
        monkeypatch.setenv("SSDE_GN_MERGE_IN_CONSUMER", "1" if mode == "consumer" else "0")
        eng = E.UNetEngine(model, 3, 32, 32, torch.device(dev))
        prog = eng.program
        kinds = [int(prog.ops[i].kind) for i in range(prog.n)]
        n_fin, n_std = kinds.count(L.OP_GN_FINALIZE), kinds.count(L.OP_GN_STATS)
        n_in = sum(1 for kind, f, _, _ in eng.b.specs if kind == L.OP_CONV and f.get("gn_in_part0") is not None)
        counts[mode] = (n_fin, n_in)
        if mode == "launch":
            assert n_fin >= n_min and n_in == 0, (mode, n_fin, n_std, n_in)
        elif mode == "0":
            assert n_fin == 0 and n_in == 0, (mode, n_fin, n_std, n_in)
        eng.weights.refresh()
        eng.load_inputs(x.to(dev).contiguous(), sig.to(dev))
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream if cuda else 0)
        seq = []
        for si, (kind, f, _, _) in enumerate(eng.b.specs):          # spec by spec: statistics buffers are recycled later
            lo, hi = prog.spec_start[si], prog.spec_start[si + 1]
            if hi > lo:
                ptr = C.cast(C.byref(prog.ops, lo * C.sizeof(L.Op)), C.POINTER(L.Op))
                L.check(lib.ssde_program_run(ptr, hi - lo, st))
            if kind in (L.OP_GN_FINALIZE, L.OP_GN_STATS):
                n = f["n"] * f["groups"]
                seq.append((f["mean"].tensor[:n].cpu().clone(), f["rstd"].tensor[:n].cpu().clone()))
            elif kind == L.OP_CONV and f.get("gn_in_part0") is not None:     # (the pass leaves them in main.gn_mean / gn_rstd too)
                n = f["n"] * f["main"]["gn_groups"]
                seq.append((f["main"]["gn_mean"].tensor[:n].cpu().clone(), f["main"]["gn_rstd"].tensor[:n].cpu().clone()))
        outs[mode] = eng.output_view().cpu().clone()
        seqs[mode] = seq
    ref = unet_oracle.ncsnpp_forward(cfg, sd, x, sig)
    assert all(rel_err(outs[m], ref) < 1e-4 for m in outs)
    # every statistic is either merged by its consumer or by a finalize launch; where the two-kernel F(4x4,3x3) form runs
    # (SSDE_WINOGRAD=6 / the product's default on the GPU) the passes take theirs
    assert sum(counts["consumer"]) == counts["launch"][0], counts
    if os.environ.get("SSDE_WINOGRAD", "") == "4" and os.environ.get("SSDE_WINO4_TWO", "2") != "0":
        assert counts["consumer"][1] >= n_min // 2, counts
    assert len(seqs["launch"]) == len(seqs["0"]) >= n_min
    for (m1, r1), (m0, r0) in zip(seqs["launch"], seqs["0"]):
        assert float((m1 - m0).abs().max()) <= 1e-5 * max(1.0, float(m0.abs().max()))
        assert rel_err(r1, r0) < 1e-5
    # a pass merges with the finalize kernel's own function over the same partials in the same order: the same bits, and so
    # is the network's output (the specs come in the same order: a statistic is finished right in front of its first reader)
    assert len(seqs["consumer"]) == len(seqs["launch"])
    for (m1, r1), (m0, r0) in zip(seqs["consumer"], seqs["launch"]):
        assert torch.equal(m1, m0) and torch.equal(r1, r0)
    assert torch.equal(outs["consumer"], outs["launch"])


def check_dropout_mask(dev):
    """The dropout mask is a pure function of (seed word, salt, element index): forward conv, weight gradient
    and GroupNorm backward regenerate the SAME mask (numpy restatement of the hash as the witness)."""
    from score_sde_pytorch_amd import hipops as ops, _lib as L
    g = torch.Generator().manual_seed(9)
    n, C, h, cout, p = 2, 32, 8, 64, 0.3
    x = torch.randn(n, C, h, h, generator=g).requires_grad_()
    G = 8
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).requires_grad_()
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_()
    w = (torch.randn(cout, C, 3, 3, generator=g) / np.sqrt(9 * C)).requires_grad_()
    seed_word, salt = 12345, 0x9E3779B1 * 3
    thresh = min(int(round(p * 2.0 ** 32)), 2 ** 32 - 1)
    elem = np.arange(n * h * h * C, dtype=np.uint64)
    keep = hash_keep(elem, (seed_word ^ (salt & 0xFFFFFFFF)) & 0xFFFFFFFF, thresh, 1.0 / (1.0 - p))
    mask = torch.from_numpy(keep.reshape(n, h, h, C)).permute(0, 3, 1, 2)
    assert 0.6 < float((mask > 0).float().mean()) < 0.8
    y = F.conv2d(F.silu(F.group_norm(x, G, gamma, beta, 1e-6)) * mask, w, padding=1)
    gout = torch.randn(n, cout, h, h, generator=g)
    y.backward(gout)
    d = lambda t: t.to(dev)  # noqa: E731
    seed_t = torch.tensor([seed_word], dtype=torch.int32, device=dev)
    xa = d(nhwc(x.detach()))
    mean, rstd = ops.groupnorm_stats(xa, G, 1e-6)
    gn = (mean, rstd, d(gamma.detach()), d(beta.detach()), G)
    # forward through the C ABI with the dropout fields armed
    a = L.ConvArgs()
    ops._fill_src(a.main, xa, None, L.PRO_GN_SILU, gn)
    ops.set_dropout(a.main, p, seed_t, salt)
    wp = ops.pack_conv_weight(w.detach().to(dev))
    dst = torch.empty(n, h, h, cout, device=dev)
    a.w_main, a.ksize, a.stride, a.pad, a.h_in, a.w_in = wp.data_ptr(), 3, 1, 1, h, h
    a.n, a.h_out, a.w_out, a.c_out, a.out_scale, a.dst = n, h, h, cout, 1.0, dst.data_ptr()
    import ctypes as C_
    L.check(L.load().ssde_conv2d(C_.byref(a), ops._stream()), "ssde_conv2d")
    assert rel_err(nchw(dst.cpu()), y.detach()) < TOL_OP
    # the same mask in the F(4x4,3x3) kernels: the fused one applies it in its halo prologue, the two-kernel form in its
    # transform pass (wino4_xform.hip) -- whose output is also what the weight gradient below may be handed (v_pre)
    from score_sde_pytorch_amd.engine import pack_wino4_weight, pack_wino4r_weight
    wp4, wp4r = pack_wino4_weight(w.detach().to(dev)), pack_wino4r_weight(w.detach().to(dev))
    vbuf = torch.full((36 * n * (h // 4) ** 2 * C,), float("nan"), device=dev)
    for tile in (L.TILE_WINOGRAD4, L.TILE_WINOGRAD4R):
        dst4 = torch.full((n, h, h, cout), float("nan"), device=dev)
        a.w_main, a.tile, a.dst, a.wino_v = (wp4r if tile == L.TILE_WINOGRAD4R else wp4).data_ptr(), tile, dst4.data_ptr(), vbuf.data_ptr()
        L.check(L.load().ssde_conv2d(C_.byref(a), ops._stream()), "ssde_conv2d")
        assert rel_err(nchw(dst4.cpu()), y.detach()) < 2e-5, tile
    dw = torch.zeros(cout, C, 3, 3, device=dev)
    ops.conv_wgrad(xa, d(nhwc(gout)), 3, dw, pro=L.PRO_GN_SILU, gn=gn, dropout=(p, seed_t, salt))
    assert rel_err(dw, w.grad) < TOL_OP
    dP = ops.conv2d(d(nhwc(gout)), w.detach().permute(1, 0, 2, 3).flip(2, 3).contiguous(), None)
    dx, _, dga, dbe = ops.gn_backward(xa, dP, gn, L.PRO_GN_SILU, dropout=(p, seed_t, salt))
    assert rel_err(nchw(dx.cpu()), x.grad) < 5 * TOL_OP
    assert rel_err(dga, gamma.grad) < 5 * TOL_OP and rel_err(dbe, beta.grad) < 5 * TOL_OP


def small_cfg(kind):
    return {"ncsnpp": lambda: _util.small_config("ncsnpp"), "ddpmpp": lambda: _util.small_config("ddpmpp"),
            "ffhq": lambda: _util.small_config("ffhq", image_size=32, ch_mult=(1, 1, 2), attn=(16,))}[kind]()


def oracle_grads(cfg, sd, x, cond, gout):
    from oracle import unet_oracle
    sd_req = {k: (v.clone().requires_grad_() if (v.dtype == torch.float32 and k != "sigmas") else v) for k, v in sd.items()}
    xr = x.clone().requires_grad_()
    y = unet_oracle.ncsnpp_forward(cfg, sd_req, xr, cond)
    y.backward(gout)
    return y.detach(), xr.grad, {k: v.grad for k, v in sd_req.items() if torch.is_tensor(v) and v.requires_grad}


def compare_param_grads(model, flat, ref, tol=TOL_GRAD):
    worst = 0.0
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        gr, gv = ref[name], flat.grad_view(p).cpu()
        scale = float(gr.abs().max())
        if scale < 1e-4:       # analytically-zero gradients (softmax shift invariance): absolute comparison
            assert float((gv - gr).abs().max()) < 1e-4, name
            continue
        e = rel_err(gv, gr)
        worst = max(worst, e)
        assert e < tol, (name, e)
    return worst


def check_unet_grads(kind, dev, cfg=None, batch=2):
    """All parameter gradients and the input gradient of a whole network vs autograd through the CPU oracle."""
    from score_sde_pytorch_amd.models import utils as mutils
    from score_sde_pytorch_amd import backward as B
    cfg = cfg or small_cfg(kind)
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    sd = {k: v.clone() for k, v in _util.load_seeded(model, seed=1).items()}
    sd["sigmas"] = model.sigmas.clone()
    model = model.to(dev)
    R = cfg.data.image_size
    g = torch.Generator().manual_seed(11)
    x = torch.randn(batch, 3, R, R, generator=g) * 2
    cond = torch.exp(torch.rand(batch, generator=g) * 4 - 2) if cfg.model.embedding_type == "fourier" \
        else torch.rand(batch, generator=g) * 900 + 50
    gout = torch.randn(batch, 3, R, R, generator=g)
    y_ref, gx_ref, ref = oracle_grads(cfg, sd, x, cond, gout)
    eng = B.TrainEngine(model, batch, R, R, torch.device(dev), input_grad=True, dropout=False)
    y = eng.forward_train(x.to(dev), cond.to(dev)).clone()
    assert rel_err(y, y_ref) < 1e-4
    eng.backward(gout.to(dev))
    assert rel_err(eng.gx_view(), gx_ref) < TOL_GRAD
    return compare_param_grads(model, eng.flat, ref)


def check_deferred_finish_is_bit_identical(dev, monkeypatch, kind="ncsnpp"):
    """SSDE_DEFER_FINISH=1 (default: bias / temb / dgamma / dbeta gradients left as partials and finished 16 jobs per launch,
    flushed where a reader needs them) against =0 (every call finishes itself): two TrainEngines over the same weights, the
    same forward and backward -- every gradient of the flat buffer and the input gradient bit for bit."""
    from score_sde_pytorch_amd.models import utils as mutils
    from score_sde_pytorch_amd import backward as B, _lib as L
    cfg = small_cfg(kind)
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.to(dev)
    R, batch = cfg.data.image_size, 3
    g = torch.Generator().manual_seed(12)
    x = torch.randn(batch, 3, R, R, generator=g) * 2
    cond = torch.exp(torch.rand(batch, generator=g) * 4 - 2) if cfg.model.embedding_type == "fourier" \
        else torch.rand(batch, generator=g) * 900 + 50
    gout = torch.randn(batch, 3, R, R, generator=g)
    grads, gxs, kinds = {}, {}, {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SSDE_DEFER_FINISH", mode)
        eng = B.TrainEngine(model, batch, R, R, torch.device(dev), input_grad=True, dropout=False)
        prog = eng.program
        kinds[mode] = [int(prog.ops[i].kind) for i in range(prog.n)]
        eng.forward_train(x.to(dev), cond.to(dev))
        eng.backward(gout.to(dev))
        grads[mode] = eng.flat.grad.detach().cpu().clone()
        gxs[mode] = eng.gx_view().cpu().clone()
    assert kinds["1"].count(L.OP_COLSUM_FINISH) > 0 and kinds["0"].count(L.OP_COLSUM_FINISH) == 0, "the two forms must differ"
    assert torch.isfinite(grads["1"]).all() and float(grads["1"].abs().max()) > 0
    assert torch.equal(grads["1"], grads["0"]) and torch.equal(gxs["1"], gxs["0"])


def check_wino_v_from_forward(dev, monkeypatch, dropout=False):
    """The F(4x4,3x3) weight gradient fed by the forward launch's by-product (ssde_conv_args.wino_v -> ssde_wgrad_args.v_pre,
    conv_wino4.hip kEmitV): with every legal 3x3 layer on the F(4x4,3x3) kernels, forward and weight gradient, the training
    program must carry such pairs, every gradient must match autograd through the oracle, and the gradients must equal -- to
    rounding of the two transform implementations -- those of the same program with the by-product switched off."""
    from score_sde_pytorch_amd.models import utils as mutils
    from score_sde_pytorch_amd import backward as B, _lib as L
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    monkeypatch.setenv("SSDE_WGRAD_WINOGRAD", "44")
    cfg = small_cfg("ncsnpp")
    if dropout:
        cfg.model.dropout = 0.1
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    sd = {k: v.clone() for k, v in _util.load_seeded(model, seed=1).items()}
    sd["sigmas"] = model.sigmas.clone()
    model = model.to(dev)
    R, batch = cfg.data.image_size, 3
    g = torch.Generator().manual_seed(12)
    x = torch.randn(batch, 3, R, R, generator=g) * 2
    cond = torch.exp(torch.rand(batch, generator=g) * 4 - 2)
    gout = torch.randn(batch, 3, R, R, generator=g)
    y_ref, gx_ref, ref = oracle_grads(cfg, sd, x, cond, gout)
    grads = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SSDE_WINO_V_FROM_FORWARD", mode)
        eng = B.TrainEngine(model, batch, R, R, torch.device(dev), input_grad=True, dropout=dropout)
        prog = eng.program
        vs = {prog.ops[i].u.conv.wino_v for i in range(prog.n) if prog.ops[i].kind == L.OP_CONV and prog.ops[i].u.conv.wino_v}
        pre = [prog.ops[i].u.wgrad.v_pre for i in range(prog.n) if prog.ops[i].kind == L.OP_WGRAD and prog.ops[i].u.wgrad.v_pre]
        # (training programs lower every F(4x4,3x3) layer -- forward and input-gradient -- to the two-kernel form, whose transform
        #  pass always owns a V buffer; the weight gradients take the forward ones)
        assert (len(pre) >= 4 and set(pre) <= vs and len(set(pre)) == len(pre)) if mode == "1" else not pre, (mode, len(vs), len(pre))
        y = eng.forward_train(x.to(dev), cond.to(dev), seed=77).clone()
        eng.backward(gout.to(dev))
        if not dropout:                                   # (the oracle has no counterpart of the hashed dropout mask)
            assert rel_err(y, y_ref) < 1e-4
            assert rel_err(eng.gx_view(), gx_ref) < TOL_GRAD
            compare_param_grads(model, eng.flat, ref)
        grads[mode] = (eng.flat.grad.detach().cpu().clone(), y.cpu().clone())
    # with dropout on: the transform pass regenerates the same mask for the forward convolution and for the weight gradient
    assert torch.equal(grads["1"][1], grads["0"][1])
    assert rel_err(grads["1"][0], grads["0"][0]) < 1e-5


def check_autograd_bridge(dev):
    """model(x, labels) under torch autograd: loss.backward() fills p.grad and x.grad via the HIP backward program."""
    from score_sde_pytorch_amd.models import utils as mutils
    cfg = small_cfg("ncsnpp")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    sd = {k: v.clone() for k, v in _util.load_seeded(model, seed=1).items()}
    sd["sigmas"] = model.sigmas.clone()
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 16, 16, generator=g)
    cond = torch.tensor([0.3, 9.0])
    gout = torch.randn(2, 3, 16, 16, generator=g)
    _, gx_ref, ref = oracle_grads(cfg, sd, x, cond, gout)
    xd = x.to(dev).requires_grad_()
    y = model(xd, cond.to(dev))
    (y * gout.to(dev)).sum().backward()
    assert rel_err(xd.grad, gx_ref) < TOL_GRAD
    for name, p in model.named_parameters():
        if p.requires_grad and float(ref[name].abs().max()) > 1e-4:
            assert rel_err(p.grad, ref[name]) < TOL_GRAD, name


def check_fused_step(dev, steps=3, sde_kind="vesde"):
    """FusedTrainStep's two halves (loss_and_grads, optimizer_step: perturb -> forward -> loss -> backward -> clip+Adam+EMA)
    for a few steps with injected t and z against oracle/train_oracle.py -- the restatement of losses.py:151-210 +
    models/ema.py that oracle/gen_golden_train.py pins to the reference's own run (rel err 0).  Same cases as the fixture test
    (check_step_fn_against_reference_run), other draws and the engine-level entry points."""
    from score_sde_pytorch_amd.models import utils as mutils, ema as ema_mod
    from score_sde_pytorch_amd import losses, sde_lib
    from oracle import train_oracle
    cfg = small_cfg("ncsnpp" if sde_kind in ("vesde", "smld") else "ddpmpp")
    if sde_kind == "smld":      # the discrete NCSN++ configs (configs/ve/cifar10_ncsnpp.py): positional labels, sigma gather
        cfg.model.embedding_type, cfg.model.num_scales, cfg.training.continuous = "positional", 24, False
    if sde_kind == "ddpm":      # configs/vp/cifar10_ddpmpp.py
        cfg.model.num_scales, cfg.training.continuous = 24, False
    cfg.optim.warmup = 2
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    sd = {k: v.clone() for k, v in _util.load_seeded(model, seed=1).items()}
    sd["sigmas"] = model.sigmas.clone()
    model = model.to(dev)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    if sde_kind in ("vesde", "smld"):
        sde = sde_lib.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales)
        okind, okw = "vesde", dict(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    else:
        sde = (sde_lib.VPSDE if sde_kind == "ddpm" else sde_lib.subVPSDE)(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales)
        okind = "vpsde" if sde_kind == "ddpm" else "subvpsde"
        okw = dict(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=cfg.model.num_scales)
    discrete = sde_kind in ("smld", "ddpm")
    orc = train_oracle.TrainState(cfg, sd, okind, okw, continuous=not discrete)
    assert orc.names == names
    R, Bn = cfg.data.image_size, 3
    g = torch.Generator().manual_seed(3)
    opt = losses.get_optimizer(cfg, model.parameters())
    ema = ema_mod.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    optimize_fn = losses.optimization_manager(cfg)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=optimize_fn, reduce_mean=False, continuous=not discrete,
                                 likelihood_weighting=False)
    state = dict(optimizer=opt, model=model, ema=ema, step=0)
    fs = step_fn.fused_for(state, torch.zeros(Bn, 3, R, R, device=dev))
    for step in range(steps):
        batch = torch.rand(Bn, 3, R, R, generator=g)
        u = torch.rand(Bn, generator=g)
        z = torch.randn(Bn, 3, R, R, generator=g)
        labels = torch.randint(0, sde.N, (Bn,), generator=g)
        ref_loss = orc.train_step(batch, u, labels, z)
        t = labels if discrete else u * (1 - 1e-5) + 1e-5                          # losses.py:84 / :116,136
        loss = fs.loss_and_grads(batch.to(dev), t=t.to(dev), z=z.to(dev)).clone()
        fs.optimizer_step(opt, ema, state['step'], optimize_fn.ssde_hyper)
        state['step'] += 1
        assert abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)) < 1e-5
        for n_, p in model.named_parameters():
            if not p.requires_grad:
                continue
            if "NIN_1.b" in n_:      # analytically-zero gradient: Adam turns rounding noise into +-lr steps
                assert float((p.detach().cpu() - orc.params[n_].detach()).abs().max()) < 4 * cfg.optim.lr * (step + 1)
                continue
            assert rel_err(p.detach(), orc.params[n_].detach()) < 2e-5, (step, n_)
            if step > 0:    # Adam normalises away the gradient scale: zero-gradient tensors follow rounding noise
                assert rel_err(p.detach().cpu() - sd[n_], orc.params[n_].detach() - sd[n_]) < 2e-2, (step, n_)
        for s_, r_, n_ in zip(ema.shadow_params, orc.shadow, names):
            assert "NIN_1.b" in n_ or rel_err(s_, r_) < 2e-5
    # the optimizer object still exposes torch.optim.Adam state for checkpoints (utils.save_checkpoint)
    osd = opt.state_dict()
    assert len(osd["state"]) == len(names) and all(float(v["step"]) == steps for v in osd["state"].values())
    return float(loss)


def _reference_train_state(dev, case, seed=1):
    """state = {optimizer, model, ema, step} + the train / eval step functions of a tests/_util.TRAIN_CASES entry, built with
    the product's public surface exactly as run_lib.train builds them (run_lib.py:63-112)."""
    from score_sde_pytorch_amd.models import utils as mutils, ema as ema_mod
    from score_sde_pytorch_amd import losses, sde_lib
    _, _, _, continuous, reduce_mean, lw = case
    cfg = _util.train_case_config(case)
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    init = {k: v.clone() for k, v in _util.load_seeded(model, seed=seed).items()}
    model = model.to(dev)
    sde = _util.train_case_sde(sde_lib, case, cfg)
    opt = losses.get_optimizer(cfg, model.parameters())
    ema = ema_mod.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    optimize_fn = losses.optimization_manager(cfg)
    kw = dict(optimize_fn=optimize_fn, reduce_mean=reduce_mean, continuous=continuous, likelihood_weighting=lw)
    train_step = losses.get_step_fn(sde, train=True, **kw)
    eval_step = losses.get_step_fn(sde, train=False, **kw)
    return cfg, init, dict(optimizer=opt, model=model, ema=ema, step=0), train_step, eval_step


def _compare_with_reference_step(gold, name, step, state, init, last):
    """parameters and EMA shadows after `step` against the REFERENCE's run (tests/golden/train_small.npz)"""
    model, ema = state['model'], state['ema']
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    cur = dict(model.named_parameters())
    norms, ema_norms = gold[name + "/norms"][step], gold[name + "/ema_norms"][step]
    assert len(names) == norms.shape[0] == len(ema.shadow_params)
    for i, n in enumerate(names):
        if "NIN_1.b" in n:        # analytically-zero gradient (softmax shift invariance): Adam turns rounding noise into +-lr steps
            continue
        p, s, p0 = cur[n].detach().cpu(), ema.shadow_params[i].detach().cpu(), init[n]
        for got, ref in ((p, norms[i]), (s, ema_norms[i])):
            assert abs(float(got.double().norm()) - ref[0]) <= 2e-5 * ref[0], (name, step, n)
            d = float((got - p0).double().norm())
            if ref[1] == 0.0:     # step 0 of the warm-up: lr = 0, nothing may move (losses.py:44-46)
                assert d == 0.0, (name, step, n, d)
            else:
                assert abs(d - ref[1]) <= 2e-2 * ref[1], (name, step, n, d, ref[1])
    if last:
        probes = [k.split("/", 2)[2] for k in gold.files if k.startswith(name + "/p/")]
        assert len(probes) >= 20
        for n in probes:
            if "NIN_1.b" in n:
                continue
            p, s, p0 = cur[n].detach().cpu(), ema.shadow_params[names.index(n)].detach().cpu(), init[n]
            rp, rs = torch.from_numpy(gold["%s/p/%s" % (name, n)]), torch.from_numpy(gold["%s/e/%s" % (name, n)])
            assert rel_err(p, rp) < 2e-5 and rel_err(s, rs) < 2e-5, (name, n)
            assert rel_err(p - p0, rp - p0) < 2e-2 and rel_err(s - p0, rs - p0) < 2e-2, (name, n, rel_err(p - p0, rp - p0))


def check_step_fn_against_reference_run(dev, name):
    """losses.get_step_fn(...)(state, batch) -- the public entry, train and eval branches -- against what the REFERENCE's
    get_step_fn + optimization_manager + get_optimizer + ExponentialMovingAverage produced on the same weights, batches and
    injected draws (oracle/gen_golden_train.py ran /root/reference's losses.py:151-210 and models/ema.py:32-51)."""
    gold = np.load(os.path.join(_util.GOLDEN, "train_small.npz"))
    case = _util.TRAIN_CASES[name]
    cfg, init, state, train_step, eval_step = _reference_train_state(dev, case)
    inputs = _util.train_case_inputs(name, cfg.model.num_scales, size=cfg.data.image_size)
    ref_loss = gold[name + "/loss"]
    for step in range(_util.TRAIN_STEPS):
        batch, u, labels, z = inputs[step]
        with _util.inject_rng(u, labels, z):
            loss = train_step(state, batch.to(dev))
        assert state['step'] == step + 1 and state['ema'].num_updates == step + 1
        assert abs(float(loss) - ref_loss[step]) <= 1e-5 * abs(ref_loss[step]), (name, step, float(loss), ref_loss[step])
        _compare_with_reference_step(gold, name, step, state, init, last=step == _util.TRAIN_STEPS - 1)
    assert int(gold[name + "/num_updates"]) == state['ema'].num_updates
    # eval branch (losses.py:200-206): the loss of the EMA weights; raw parameters, EMA and counters untouched
    batch, u, labels, z = inputs[_util.TRAIN_STEPS]
    before = [p.detach().clone() for p in state['model'].parameters()]
    shadows = [s_.clone() for s_ in state['ema'].shadow_params]
    with _util.inject_rng(u, labels, z):
        eval_loss = eval_step(state, batch.to(dev))
    ref_eval = float(gold[name + "/eval_loss"])
    assert abs(float(eval_loss) - ref_eval) <= 1e-5 * abs(ref_eval), (name, float(eval_loss), ref_eval)
    assert state['step'] == _util.TRAIN_STEPS and state['ema'].num_updates == _util.TRAIN_STEPS
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, state['model'].parameters()))
    assert all(torch.equal(a, b) for a, b in zip(shadows, state['ema'].shadow_params))
    osd = state['optimizer'].state_dict()
    assert all(float(v["step"]) == _util.TRAIN_STEPS for v in osd["state"].values())
    return float(loss)


def check_reference_checkpoint_resume(dev, tmp_path, warm):
    """utils.restore_checkpoint on a checkpoint WRITTEN BY THE REFERENCE (tests/golden/ref_checkpoint_small.pth: DataParallel's
    `module.` keys, torch.optim.Adam.state_dict(), the reference EMA's {decay, num_updates, shadow_params}; utils.py:22-29 after two
    of the reference's steps), then the third step against the reference's third step.  warm: the state has already run a fused
    step (flat buffers, re-homed moments and shadows exist) before the restore."""
    from score_sde_pytorch_amd import utils as ssde_utils
    gold = np.load(os.path.join(_util.GOLDEN, "train_small.npz"))
    case = _util.TRAIN_CKPT_CASE
    cfg, _, state, train_step, _ = _reference_train_state(dev, case, seed=9)       # other weights than the checkpoint's
    inputs = _util.train_case_inputs("ckpt", cfg.model.num_scales, size=cfg.data.image_size)
    if warm:
        batch, u, labels, z = inputs[3]
        with _util.inject_rng(u, labels, z):
            train_step(state, batch.to(dev))
    path = os.path.join(_util.GOLDEN, "ref_checkpoint_small.pth")
    assert ssde_utils.restore_checkpoint(path, state, dev) is state
    assert state['step'] == 2 and state['ema'].num_updates == 2 and state['ema'].decay == cfg.model.ema_rate
    raw = ssde_utils.load_checkpoint_file(path, "cpu")
    assert all(k.startswith("module.") for k in raw['model'])
    for k, v in state['model'].state_dict().items():
        assert torch.equal(v.cpu(), raw['model']["module." + k]), k
    # the initial weights of the reference run are tests/_util's seed 1: the fixture's norms are relative to them
    from score_sde_pytorch_amd.models import utils as mutils
    init = _util.load_seeded(mutils.get_model("ncsnpp")(cfg), seed=1)
    batch, u, labels, z = inputs[2]
    with _util.inject_rng(u, labels, z):
        loss = train_step(state, batch.to(dev))
    ref = gold["ckpt/loss"][2]
    assert abs(float(loss) - ref) <= 1e-5 * abs(ref), (float(loss), ref)
    assert state['step'] == 3
    _compare_with_reference_step(gold, "ckpt", 2, state, init, last=True)
    # and back: what save_checkpoint writes for the reference to read has the reference's own structure
    out = os.path.join(str(tmp_path), "checkpoint_3.pth")
    ssde_utils.save_checkpoint(out, state, data_parallel_prefix=True)
    mine = ssde_utils.load_checkpoint_file(out, "cpu")
    assert set(mine) == set(raw) and list(mine['model']) == list(raw['model'])
    assert set(mine['ema']) == set(raw['ema']) and len(mine['ema']['shadow_params']) == len(raw['ema']['shadow_params'])
    assert mine['optimizer']['param_groups'][0]['params'] == raw['optimizer']['param_groups'][0]['params']
    assert set(mine['optimizer']['state']) == set(raw['optimizer']['state'])
    assert all(set(v) == {"step", "exp_avg", "exp_avg_sq"} for v in mine['optimizer']['state'].values())
    assert mine['step'] == 3 and mine['ema']['num_updates'] == 3


def check_device_repack(dev, kind="ncsnpp", need_kind=None):
    """ssde_pack_weights (four launches for the whole model, driven by a device-resident descriptor table) against the
    torch packers in engine.py, for every packed operand incl. Winograd and input-gradient variants."""
    from score_sde_pytorch_amd.models import utils as mutils
    from score_sde_pytorch_amd import backward as B
    cfg = small_cfg(kind)
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.to(dev)
    R = cfg.data.image_size
    eng = B.TrainEngine(model, 2, R, R, torch.device(dev), input_grad=True, dropout=False)
    ws = eng.weights
    eng.flat.data.add_(torch.randn(eng.flat.numel, generator=torch.Generator().manual_seed(5)).to(dev) * 0.1)
    ws.refresh(force=True, on_device=False)
    ref = [e[0].clone() for e in ws.entries]
    for e in ws.entries:
        if e[4] is not None:
            e[0].fill_(7.0) if e[4][0]["kind"] in (1, 2) else e[0].zero_()   # matrix / vector kernels keep the zero padding
    assert ws.device_refresh()
    n = 0
    for e, r in zip(ws.entries, ref):
        if e[4] is not None:
            assert float((e[0] - r).abs().max()) < 1e-6, (e[4][0]["kind"], e[4][0].get("flags"), tuple(e[0].shape))
            n += 1
    assert n >= len(ws.entries) - 2
    if need_kind is not None:                     # the pack kind under test is really in this store, forward and input-gradient
        flags = {d.get("flags", 0) for e in ws.entries if e[4] is not None for d in e[4] if d["kind"] == need_kind}
        assert flags == {0, 1}, flags


def check_op_package(dev):
    """score_sde_pytorch_amd.op: upfirdn2d / fused_leaky_relu (NCHW, autograd) vs the oracle's upfirdn2d_native
    restatement and F.leaky_relu, forward and gradients."""
    from oracle import unet_oracle
    import score_sde_pytorch_amd.op as op
    g = torch.Generator().manual_seed(21)
    k = torch.from_numpy(unet_oracle.setup_fir_kernel([1, 3, 3, 1]))
    for up, down, pad, c in [(2, 1, (2, 1), 6), (1, 2, (1, 1), 8), (1, 1, (2, 2), 3)]:
        x = torch.randn(2, c, 8, 8, generator=g)
        go = None
        xr = x.clone().requires_grad_()
        ref = unet_oracle.upfirdn2d(xr, k * (up ** 2), up=up, down=down, pad=pad)
        go = torch.randn(ref.shape, generator=g)
        ref.backward(go)
        xd = x.to(dev).requires_grad_()
        y = op.upfirdn2d(xd, (k * (up ** 2)).to(dev), up=up, down=down, pad=pad)
        y.backward(go.to(dev))
        assert rel_err(y.detach(), ref.detach()) < 2e-6 and rel_err(xd.grad, xr.grad) < 2e-6, (up, down, pad)
    x = torch.randn(3, 5, 4, 4, generator=g)
    b = torch.randn(5, generator=g)
    xr, br = x.clone().requires_grad_(), b.clone().requires_grad_()
    ref = F.leaky_relu(xr + br[None, :, None, None], 0.2) * 2 ** 0.5
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd, bd = x.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    y = op.fused_leaky_relu(xd, bd)
    y.backward(go.to(dev))
    assert rel_err(y.detach(), ref.detach()) < 2e-6 and rel_err(xd.grad, xr.grad) < 2e-6 and rel_err(bd.grad, br.grad) < 2e-6
    # double backward (op/fused_act.py:20-51: a penalty on the gradient differentiates through the gradient op)
    w1, w2 = torch.randn(x.shape, generator=g), torch.randn(5, generator=g)
    outs = []
    for which, d in (("torch", "cpu"), ("op", dev)):
        xx, bb, gg = x.to(d).requires_grad_(), b.to(d).requires_grad_(), go.to(d).requires_grad_()
        yy = (F.leaky_relu(xx + bb[None, :, None, None], 0.2) * 2 ** 0.5) if which == "torch" else op.fused_leaky_relu(xx, bb)
        gx, gb = torch.autograd.grad(yy, (xx, bb), gg, create_graph=True)
        pen = (gx * w1.to(d)).sum() + (gb * w2.to(d)).sum()
        (ggo,) = torch.autograd.grad(pen, gg)
        outs.append((gx.detach().cpu(), gb.detach().cpu(), ggo.cpu()))
    for a_, r_ in zip(outs[1], outs[0]):
        assert rel_err(a_, r_) < 2e-6


def check_input_gradient_only(dev):
    """TrainEngine(param_grads=False): d out / d x without any weight-gradient kernel, vs oracle autograd."""
    from score_sde_pytorch_amd.models import utils as mutils
    from score_sde_pytorch_amd import backward as B, _lib as L
    cfg = small_cfg("ddpmpp")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    sd = {k: v.clone() for k, v in _util.load_seeded(model, seed=1).items()}
    sd["sigmas"] = model.sigmas.clone()
    model = model.to(dev)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 3, 16, 16, generator=g)
    cond = torch.tensor([120.0, 800.0])
    gout = torch.randn(2, 3, 16, 16, generator=g)
    _, gx_ref, _ = oracle_grads(cfg, sd, x, cond, gout)
    eng = B.TrainEngine(model, 2, 16, 16, torch.device(dev), input_grad=True, dropout=False, param_grads=False)
    kinds = [int(eng.program.ops[i].kind) for i in range(eng.program.n)]
    assert L.OP_WGRAD not in kinds and L.OP_MEMSET not in kinds
    eng.forward_train(x.to(dev), cond.to(dev))
    eng.backward(gout.to(dev))
    assert rel_err(eng.gx_view(), gx_ref) < TOL_GRAD


def _ode_case_model(dev):
    from score_sde_pytorch_amd.models import utils as mutils
    from score_sde_pytorch_amd import sde_lib
    cfg = small_cfg("ddpmpp")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.to(dev).eval()
    return cfg, model, sde_lib.subVPSDE(**_util.ODE_CASE["sde_kwargs"])


def check_likelihood(dev):
    """likelihood.get_likelihood_fn against the REFERENCE's likelihood.get_likelihood_fn output stored in
    tests/golden/ode_small.npz (oracle/gen_golden_ode.py ran /root/reference/likelihood.py:69-111 on the same
    network, data and Hutchinson probe; sub-VP, RK45 rtol = atol = 1e-5, eps = 1e-5 -- BASELINE config #5's settings).

    Tolerances, and why: the integration is adaptive over ~330 accepted steps through a random-weight network, so it
    amplifies the per-evaluation fp32 difference between the HIP U-Net and the CPU one (<= 1e-4, typically 1e-5).  The
    fixture records the amplification measured on the reference itself: scaling the input by (1 + 1e-6) moves bpd by
    1.3e-5 and the latent by 1.3e-3 relative (`lik_sens_*`), i.e. x13 and x1300.  Hence: one right-hand-side
    evaluation 1e-4; bpd 1e-3 relative; latent 5e-2 relative; NFE within 3 % (an accept/reject decision can flip)."""
    from score_sde_pytorch_amd.models import utils as mutils
    from score_sde_pytorch_amd import likelihood
    gold = np.load(os.path.join(_util.GOLDEN, "ode_small.npz"))
    case = _util.ODE_CASE
    cfg, model, sde = _ode_case_model(dev)
    _, data, epsilon = _util.ode_case_inputs()
    inv = _util.ode_inverse_scaler
    shape = data.shape
    # the probe is drawn with torch.randint_like on the data's device (likelihood.py:76): inject the fixture's
    real = torch.randint_like
    torch.randint_like = lambda t, low=0, high=2, **kw: ((epsilon + 1.) / 2.).to(t.device)
    try:
        lik = likelihood.get_likelihood_fn(sde, inv, rtol=case["rtol"], atol=case["atol"], eps=case["lik_eps"])
        bpd, z, nfe = lik(model, data.to(dev))
    finally:
        torch.randint_like = real
    assert dev == "cpu" or lik.last_path == "fused"       # ode.FusedLikelihoodRhs: one device program per evaluation
    # (a) one evaluation of the augmented right-hand side at a fixed point (reference closures likelihood.py:26-37,59-67)
    drift_fn = lambda xx, tt: sde.reverse(mutils.get_score_fn(sde, model, train=False, continuous=True),     # noqa: E731
                                          probability_flow=True).sde(xx, tt)[0]
    with torch.no_grad(), likelihood._frozen(model):
        xd = data.to(dev).clone()
        vt = torch.ones(shape[0], device=dev) * case["t_probe"]
        div = likelihood.get_div_fn(drift_fn)(xd, vt, epsilon.to(dev))
        dr = drift_fn(xd, vt)
    assert rel_err(dr, torch.from_numpy(gold["rhs_drift"])) < 1e-4
    # eps^T J eps is a sum of 768 terms of mixed sign: compare on the scale of the number of terms
    scale = float(np.prod(shape[1:]))
    assert float((div.cpu() - torch.from_numpy(gold["rhs_div"])).abs().max()) / scale < 1e-4, (div, gold["rhs_div"])
    # (b) the whole integration
    ref_nfe = int(gold["lik_nfe"])
    assert abs(nfe - ref_nfe) <= 0.03 * ref_nfe, (nfe, ref_nfe)
    assert torch.isfinite(bpd).all() and torch.isfinite(z).all()
    assert rel_err(bpd, torch.from_numpy(gold["lik_bpd"])) < 1e-3, (bpd, gold["lik_bpd"])
    assert rel_err(z, torch.from_numpy(gold["lik_z"])) < 5e-2


def check_fused_likelihood_rhs(dev):
    """ode.FusedLikelihoodRhs: one evaluation of d/dt [x, delta log p] as a single device program (forward, drift kernel,
    input-gradient program seeded with the Hutchinson probe, per-sample divergence kernel) against the REFERENCE's
    right-hand side stored in tests/golden/ode_small.npz (`rhs_drift`, `rhs_div`: likelihood.py:29-35,59-67 evaluated by
    oracle/gen_golden_ode.py at t_probe), then a second evaluation at another time to show that the per-evaluation
    scalars really come from the device record (same program, new record)."""
    from score_sde_pytorch_amd import ode
    from oracle import ode_oracle, sampler_oracle
    gold = np.load(os.path.join(_util.GOLDEN, "ode_small.npz"))
    case = _util.ODE_CASE
    cfg, model, sde = _ode_case_model(dev)
    _, data, epsilon = _util.ode_case_inputs()
    rhs = ode.FusedLikelihoodRhs(model, sde, data.shape, epsilon, torch.device(dev))
    n, Bn = data.numel(), data.shape[0]
    scale = float(np.prod(data.shape[1:]))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    osde = sampler_oracle.make_sde("subvpsde", **case["sde_kwargs"])
    for t in (case["t_probe"], 0.37):
        rhs.x32.copy_(data.reshape(-1).to(dev))
        out = torch.full((n + Bn,), float("nan"), dtype=torch.float64, device=dev)
        rhs(t, None, out=out)
        got_d = out[:n].reshape(data.shape).to("cpu", torch.float32)
        got_v = out[n:].to("cpu", torch.float32)
        if t == case["t_probe"]:
            ref_d, ref_v = torch.from_numpy(gold["rhs_drift"]), torch.from_numpy(gold["rhs_div"])
        else:
            ref_d, ref_v = ode_oracle.rhs_augmented(cfg, sd, osde, data, torch.ones(Bn) * t, epsilon)
        assert rel_err(got_d, ref_d) < 1e-4, (t, rel_err(got_d, ref_d))
        # eps^T J eps is a sum of 768 terms of mixed sign: compare on the scale of the number of terms
        assert float((got_v - ref_v).abs().max()) / scale < 1e-4, (t, got_v, ref_v)
    assert rhs.last_path == ("graph" if dev != "cpu" else "eager")


def check_ode_sampler(dev, denoise):
    """sampling.get_ode_sampler against the REFERENCE's get_ode_sampler output in tests/golden/ode_small.npz
    (/root/reference/sampling.py:449-483 run by oracle/gen_golden_ode.py; sub-VP, RK45 rtol = atol = 1e-5, eps = 1e-3).
    The reverse-time integration is well conditioned -- the fixture's `ode_sens`: input x (1 + 1e-6) moves the samples
    by 1.7e-6 -- so the samples must agree to 1e-3 relative (ten times the 1e-4 forward tolerance) and the NFE to within
    two steps (12 evaluations: an accept/reject decision near error_norm = 1 can flip)."""
    from score_sde_pytorch_amd import sampling
    gold = np.load(os.path.join(_util.GOLDEN, "ode_small.npz"))
    case = _util.ODE_CASE
    cfg, model, sde = _ode_case_model(dev)
    z, _, _ = _util.ode_case_inputs()
    smp = sampling.get_ode_sampler(sde, tuple(z.shape), _util.ode_inverse_scaler, denoise=denoise, rtol=case["rtol"],
                                   atol=case["atol"], eps=case["sample_eps"], device=dev)
    x, nfe = smp(model, z=z.to(dev))
    assert smp.last_path == "fused"       # stage arithmetic, U-Net program and drift all as HIP launches (ode.FusedDrift)
    tag = "ode_denoise" if denoise else "ode"
    assert abs(nfe - int(gold[tag + "_nfe"])) <= 12, (nfe, int(gold[tag + "_nfe"]))
    assert rel_err(x, torch.from_numpy(gold[tag + "_samples"])) < 1e-3


def check_fused_drift_discrete_ve(dev):
    """ode.FusedDrift on a discrete-label VE model (positional embedding + scale_by_sigma, e.g. the reference's
    ve/cifar10_ncsnpp): the probability-flow ODE always asks for continuous labels (sampling.py:431), so the network sees
    labels = sigma(t) and divides its output by sigmas[labels.long()] (ncsnpp.py:245,377-379).  The fused right-hand
    side has to fill that separate per-sample sigma buffer of the U-Net program itself (round-2 advisor finding: it stayed
    zero and the score was inf / NaN).  Compared with the CPU oracle's drift (oracle/ode_oracle.py:19-22)."""
    from score_sde_pytorch_amd.models import utils as mutils
    from score_sde_pytorch_amd import sde_lib, ode
    from oracle import ode_oracle, sampler_oracle
    cfg = small_cfg("ncsnpp")
    cfg.model.embedding_type = "positional"
    cfg.model.fourier_scale = 16
    cfg.training.continuous = False
    assert cfg.model.scale_by_sigma
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    sd = dict(_util.load_seeded(model, seed=1)); sd["sigmas"] = model.sigmas.clone()
    model = model.to(dev).eval()
    kw = dict(sigma_min=cfg.model.sigma_min, sigma_max=cfg.model.sigma_max, N=cfg.model.num_scales)
    sde = sde_lib.VESDE(**kw)
    R, Bn = cfg.data.image_size, 3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(Bn, 3, R, R, generator=g) * 4.0
    rhs = ode.FusedDrift(model, sde, x.shape, torch.device(dev))
    assert rhs.unet.sig is not rhs.unet.cond
    for t in (0.9, 0.45):
        rhs.x32.copy_(x.reshape(-1).to(dev))
        out = torch.full((x.numel(),), float("nan"), dtype=torch.float64, device=dev)
        rhs(t, None, out=out)
        with torch.no_grad():
            ref = ode_oracle.drift(cfg, sd, sampler_oracle.make_sde("vesde", **kw), x, torch.ones(Bn) * t)
        got = out.reshape(x.shape).to("cpu", torch.float32)
        assert torch.isfinite(got).all()
        assert rel_err(got, ref) < 1e-4, (t, rel_err(got, ref))


def check_forced_exchange_one_rank(dev, backend, monkeypatch):
    """The data-parallel path on ONE GPU: a world-size-1 `nccl` (= RCCL) process group with the exchange forced on
    (SSDE_FORCE_GRAD_EXCHANGE=1) drives TrainEngine.run_backward_bucketed -> dist.all_reduce(flat.grad[lo:hi],
    async_op=True) per bucket -> work.wait() before the fused optimizer, i.e. the stream ordering between the
    ctypes-launched backward segments (torch's current stream) and RCCL's own stream (losses.py, the reference's
    nn.DataParallel at models/utils.py:93 is what it replaces).  A one-rank SUM is the identity and the 1/world factor is 1,
    so three steps must be BIT-identical to the same three steps without the exchange; small buckets so that many
    collectives are in flight while the backward program is still being enqueued."""
    import socket
    import torch.distributed as dist
    from score_sde_pytorch_amd.models import utils as mutils, ema as ema_mod
    from score_sde_pytorch_amd import losses, sde_lib
    cfg = _util.small_config("ncsnpp")
    cfg.model.dropout = 0.1
    sde = sde_lib.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales)
    g = torch.Generator().manual_seed(3)
    batches = [(torch.rand(4, 3, 16, 16, generator=g), torch.rand(4, generator=g) * 0.9 + 0.05, torch.randn(4, 3, 16, 16, generator=g))
               for _ in range(3)]

    def run(whole_step=False):
        """whole_step: FusedTrainStep.train_step (on the GPU: one hipGraph without the exchange, one hipGraph per gradient bucket
        with it -- _step_graph_segments); else its two halves as program launches"""
        torch.manual_seed(0)
        model = mutils.get_model("ncsnpp")(cfg)
        _util.load_seeded(model, seed=1)
        model = model.to(dev)
        opt = losses.get_optimizer(cfg, model.parameters())
        ema = ema_mod.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
        optimize_fn = losses.optimization_manager(cfg)
        step_fn = losses.get_step_fn(sde, train=True, optimize_fn=optimize_fn, continuous=True)
        state = dict(optimizer=opt, model=model, ema=ema, step=0)
        fs = step_fn.fused_for(state, torch.zeros(4, 3, 16, 16, device=dev))
        ls, buckets = [], 0
        for i, (b, t, z) in enumerate(batches):
            if whole_step:
                ls.append(float(fs.train_step(b.to(dev), opt, ema, state["step"], optimize_fn.ssde_hyper, t=t.to(dev), z=z.to(dev),
                                              seed=11 + i)))
                buckets = max(buckets, getattr(fs, "collectives_last_step", 0))
            else:
                ls.append(float(fs.loss_and_grads(b.to(dev), t=t.to(dev), z=z.to(dev), seed=11 + i)))
                buckets = max(buckets, len(fs._pending))
                fs.optimizer_step(opt, ema, state["step"], optimize_fn.ssde_hyper)
            state["step"] += 1
        if dev != "cpu":
            torch.cuda.synchronize()
        segmented = getattr(fs, "_graph_seg", None) is not None
        return ls, fs.flat.data.clone(), fs.flat.grad.clone(), buckets, segmented

    ref_loss, ref_p, ref_g, nb0, _ = run()
    assert nb0 == 0
    ref_loss_w, ref_p_w, ref_g_w, _, seg0 = run(whole_step=True)        # the single-replica step (one graph on the GPU)
    assert not seg0 and ref_loss_w == ref_loss and torch.equal(ref_p_w, ref_p)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    monkeypatch.setenv("SSDE_FORCE_GRAD_EXCHANGE", "1")
    monkeypatch.setenv("SSDE_GRAD_BUCKET_MB", "0.05")
    kw = dict(device_id=torch.device("cuda", torch.cuda.current_device())) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, **kw)
    try:
        loss, p, gr, nb, _ = run()
        loss_w, p_w, gr_w, nb_w, seg = run(whole_step=True)
    finally:
        dist.destroy_process_group()
    assert nb >= 4, nb                          # several async all-reduces were in flight behind the backward program
    assert loss == ref_loss
    assert torch.equal(gr, ref_g) and torch.equal(p, ref_p)
    # the whole step with the exchange: on the GPU one captured graph per gradient bucket + the closing optimizer graph
    assert nb_w >= 4 and seg == (dev != "cpu"), (nb_w, seg)
    assert loss_w == ref_loss and torch.equal(gr_w, ref_g) and torch.equal(p_w, ref_p)


def _train_plan_case(dev, dropout):
    from score_sde_pytorch_amd.models import utils as mutils, ema as ema_mod
    from score_sde_pytorch_amd import losses, sde_lib
    cfg = small_cfg("ncsnpp")
    cfg.model.dropout = dropout
    cfg.optim.warmup = 0
    sde = sde_lib.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales)
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    sd = dict(_util.load_seeded(model, seed=1)); sd["sigmas"] = model.sigmas.clone()
    model = model.to(dev)
    opt = losses.get_optimizer(cfg, model.parameters())
    ema = ema_mod.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    optimize_fn = losses.optimization_manager(cfg)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=optimize_fn, reduce_mean=False, continuous=True)
    state = dict(optimizer=opt, model=model, ema=ema, step=0)
    R, Bn = cfg.data.image_size, 3
    fs = step_fn.fused_for(state, torch.zeros(Bn, 3, R, R, device=dev))
    return cfg, sd, sde, model, opt, ema, optimize_fn, state, fs, R, Bn


def check_train_plan(dev, tmp_path=None, c_host=None):
    """SURVEY 8(b)'s training entries of the plan-level C ABI (csrc/plan.hip): a losses.FusedTrainStep exported with
    plan_export.export_train_plan and driven through ssde_train_step reproduces the Python-driven fused step (same kernels,
    same order: loss and every parameter bit for bit over two steps, Adam moments and EMA carried inside the plan);
    ssde_train_forward + ssde_unet_backward return the forward output and ALL parameter gradients of the network, compared
    with torch autograd through the CPU oracle (= the reference's loss.backward(), losses.py:196).
    c_host: optional callable(blob_path, files, loss_ref) that runs tests/c_host/plan_host.c in `train` mode."""
    from score_sde_pytorch_amd import plan_export
    # ---- (1) whole optimisation steps
    cfg, sd, sde, model, opt, ema, optimize_fn, state, fs, R, Bn = _train_plan_case(dev, dropout=0.1)
    blob = plan_export.export_train_plan(fs, opt, ema)
    g = torch.Generator().manual_seed(21)
    steps = [(torch.rand(Bn, 3, R, R, generator=g), torch.rand(Bn, generator=g) * 0.9 + 0.05, torch.randn(Bn, 3, R, R, generator=g))
             for _ in range(2)]
    ref_loss, hypers, coefs = [], [], []
    for i, (b, t, z) in enumerate(steps):
        ref_loss.append(float(fs.loss_and_grads(b.to(dev), t=t.to(dev), z=z.to(dev), seed=40 + i)))
        coefs.append((fs.a.clone(), fs.s.clone(), fs.eng.cond.tensor[:Bn].clone()))
        fs.optimizer_step(opt, ema, state["step"], optimize_fn.ssde_hyper)
        state["step"] += 1
        hypers.append(fs.hyper.detach().cpu()[:9].tolist())
    ref_params = fs.flat.data.clone()
    plan = plan_export.LoadedPlan(blob)
    assert plan.header.kind == plan_export.PLAN_TRAIN and plan.header.n_flat == fs.flat.numel and plan.header.seg[3] > plan.header.seg[2]
    for i, (b, t, z) in enumerate(steps):
        a_, s_, lab = coefs[i]
        loss = plan.train_step(b.to(dev), z.to(dev), a_, s_, lab, hypers[i], 40 + i)
        assert abs(float(loss) - ref_loss[i]) <= (0.0 if dev != "cpu" else 1e-6 * abs(ref_loss[i])), (i, float(loss), ref_loss[i])
    got = plan.read_io(plan_export.IO_PARAMS, ref_params)
    if dev != "cpu":
        assert torch.equal(got, ref_params), float((got - ref_params).abs().max())
    else:
        # on the emulator's host tensors the Python path re-packs the Winograd weights with the torch packers after a step
        # (WeightStore.refresh: device re-pack only on the GPU) while the plan always uses ssde_pack_weights: G g G^T in
        # another summation order, a last-bit difference in the packed weights of the second step
        assert rel_err(got, ref_params) < 1e-5
    plan.close()
    if c_host is not None:
        files = {}
        b, t, z = steps[0]
        a_, s_, lab = coefs[0]
        for name, v in (("batch", b), ("z", z), ("a", a_), ("s", s_), ("labels", lab), ("hyper", torch.tensor(hypers[0]))):
            files[name] = os.path.join(str(tmp_path), name + ".f32")
            np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32).tofile(files[name])
        blob_path = os.path.join(str(tmp_path), "train.blob")
        open(blob_path, "wb").write(blob)
        c_host(blob_path, files, 40, ref_loss[0])
    # ---- (2) forward + all gradients through the plan, against autograd over the oracle (no dropout)
    cfg, sd, sde, model, opt, ema, optimize_fn, state, fs, R, Bn = _train_plan_case(dev, dropout=0.0)
    plan = plan_export.LoadedPlan(plan_export.export_train_plan(fs))          # no optimizer segment
    assert plan.header.seg[3] == 0
    g = torch.Generator().manual_seed(22)
    x = torch.randn(Bn, 3, R, R, generator=g) * 2
    cond = torch.exp(torch.rand(Bn, generator=g) * 4 - 2)
    gout = torch.randn(Bn, 3, R, R, generator=g)
    y_ref, _, gref = oracle_grads(cfg, sd, x, cond, gout)
    y = plan.train_forward(x.to(dev), cond.to(dev))
    assert rel_err(y, y_ref) < 1e-4
    _, dparams = plan.unet_backward(gout.to(dev))
    worst = 0.0
    checked = 0
    for name, p in model.named_parameters():
        if not p.requires_grad:                      # (the Fourier frequencies are not trained: ncsnpp.py:60-64 / layerspp.py:35)
            continue
        o, n = fs.flat.index[id(p)]
        checked += 1
        gv, gr = dparams[o:o + n].view(p.shape).cpu(), gref[name]
        scale = float(gr.abs().max())
        if scale < 1e-4:
            assert float((gv - gr).abs().max()) < 1e-4, name
            continue
        worst = max(worst, rel_err(gv, gr))
        assert rel_err(gv, gr) < TOL_GRAD, (name, rel_err(gv, gr))
    assert checked >= 100
    plan.close()
    return worst


def check_generic_loss_closures(dev):
    """losses.get_sde_loss_fn / get_smld_loss_fn / get_ddpm_loss_fn on a model the fused step cannot lower (a plain
    torch module): perturbation and loss head run as libssde_hip kernels (ssde_perturb, ssde_dsm_loss) under a custom
    autograd function.  Loss value and parameter gradients against the reference's formulas (losses.py:77-101, 104-125,
    128-148) written out in torch on the same RNG draws."""
    from score_sde_pytorch_amd import losses, sde_lib

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = torch.nn.Conv2d(3, 8, 3, padding=1)
            self.c2 = torch.nn.Conv2d(8, 3, 3, padding=1)

        def forward(self, x, labels):
            h = F.silu(self.c1(x)) * (1.0 + 0.01 * torch.log1p(labels.float().abs()))[:, None, None, None]
            return self.c2(h)

    torch.manual_seed(3)
    model = Tiny().to(dev)
    batch = torch.randn(4, 3, 8, 8, device=dev)
    bc = lambda v: v[:, None, None, None]                                                            # noqa: E731

    def ref_sde(sde, reduce_mean, lw, eps=1e-5):
        t = torch.rand(batch.shape[0], device=dev) * (sde.T - eps) + eps
        z = torch.randn_like(batch)
        mean, std = sde.marginal_prob(batch, t)
        xt = mean + bc(std) * z
        if isinstance(sde, sde_lib.VESDE):
            score = model(xt, sde.marginal_prob(torch.zeros_like(xt), t)[1])
        else:
            score = -model(xt, t * 999) / bc(std)
        red = (lambda v: v.flatten(1).mean(1)) if reduce_mean else (lambda v: 0.5 * v.flatten(1).sum(1))
        if lw:
            g2 = sde.sde(torch.zeros_like(batch), t)[1] ** 2
            return (red((score + z / bc(std)) ** 2) * g2).mean()
        return red((score * bc(std) + z) ** 2).mean()

    def ref_smld(sde, reduce_mean):
        labels = torch.randint(0, sde.N, (batch.shape[0],), device=dev)
        sig = torch.flip(sde.discrete_sigmas, dims=(0,)).to(dev)[labels]
        z = torch.randn_like(batch)
        score = model(batch + bc(sig) * z, labels)
        red = (lambda v: v.flatten(1).mean(1)) if reduce_mean else (lambda v: 0.5 * v.flatten(1).sum(1))
        return (red((score + z / bc(sig)) ** 2) * sig ** 2).mean()

    def ref_ddpm(sde, reduce_mean):
        labels = torch.randint(0, sde.N, (batch.shape[0],), device=dev)
        a, s = sde.sqrt_alphas_cumprod.to(dev)[labels], sde.sqrt_1m_alphas_cumprod.to(dev)[labels]
        z = torch.randn_like(batch)
        out = model(bc(a) * batch + bc(s) * z, labels)
        red = (lambda v: v.flatten(1).mean(1)) if reduce_mean else (lambda v: 0.5 * v.flatten(1).sum(1))
        return red((out - z) ** 2).mean()

    ve, vp, sub = sde_lib.VESDE(N=50), sde_lib.VPSDE(N=50), sde_lib.subVPSDE(N=50)
    cases = [
        (losses.get_sde_loss_fn(ve, True, reduce_mean=False, likelihood_weighting=False), lambda: ref_sde(ve, False, False)),
        (losses.get_sde_loss_fn(vp, True, reduce_mean=True, likelihood_weighting=False), lambda: ref_sde(vp, True, False)),
        (losses.get_sde_loss_fn(sub, True, reduce_mean=False, likelihood_weighting=True), lambda: ref_sde(sub, False, True)),
        (losses.get_sde_loss_fn(vp, True, reduce_mean=True, likelihood_weighting=True), lambda: ref_sde(vp, True, True)),
        (losses.get_smld_loss_fn(ve, True, reduce_mean=False), lambda: ref_smld(ve, False)),
        (losses.get_smld_loss_fn(ve, True, reduce_mean=True), lambda: ref_smld(ve, True)),
        (losses.get_ddpm_loss_fn(vp, True, reduce_mean=True), lambda: ref_ddpm(vp, True)),
        (losses.get_ddpm_loss_fn(vp, True, reduce_mean=False), lambda: ref_ddpm(vp, False)),
    ]
    for i, (fn, ref) in enumerate(cases):
        got = []
        for f in (lambda: fn(model, batch), ref):
            torch.manual_seed(100 + i)
            model.zero_grad(set_to_none=True)
            loss = f()
            loss.backward()
            got.append((loss.detach().clone(), [p.grad.detach().clone() for p in model.parameters()]))
        (l1, g1), (l0, g0) = got
        assert l1.shape == () and abs(float(l1) - float(l0)) <= 2e-5 * abs(float(l0)), (i, float(l1), float(l0))
        for a, b in zip(g1, g0):
            assert rel_err(a, b) < 2e-5, (i, rel_err(a, b))


def check_checkpoint_and_ema_swap(dev, tmp_path):
    """utils.save_checkpoint / restore_checkpoint (reference utils.py:7-28) around the fused step, the EMA
    store / copy_to / restore swap (models/ema.py:53-89) on the flat buffers, and weight re-packing of an inference
    engine after the fused optimizer wrote the parameters behind torch's back."""
    from score_sde_pytorch_amd.models import utils as mutils, ema as ema_mod
    from score_sde_pytorch_amd import losses, sde_lib, utils as ssde_utils
    cfg = small_cfg("ncsnpp")
    cfg.optim.warmup = 0
    sde = sde_lib.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales)
    R, Bn = cfg.data.image_size, 3
    g = torch.Generator().manual_seed(11)
    batches = [(torch.rand(Bn, 3, R, R, generator=g), torch.rand(Bn, generator=g) * 0.9 + 0.05, torch.randn(Bn, 3, R, R, generator=g))
               for _ in range(3)]
    xq = torch.rand(2, 3, R, R, generator=g).to(dev)
    sq = torch.tensor([0.5, 3.0]).to(dev)

    def infer(m, x, sig):
        """inference engine of the model (NCSNpp.forward's no-grad path; called directly so that the emulated run,
        whose tensors are host tensors, passes the product's is_cuda guard)"""
        return m._engine_for(x).forward(x, sig).clone()

    def make(seed):
        torch.manual_seed(seed)
        model = mutils.get_model("ncsnpp")(cfg)
        _util.load_seeded(model, seed=seed + 1)
        model = model.to(dev)
        opt = losses.get_optimizer(cfg, model.parameters())
        ema = ema_mod.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
        optimize_fn = losses.optimization_manager(cfg)
        step_fn = losses.get_step_fn(sde, train=True, optimize_fn=optimize_fn, reduce_mean=False, continuous=True)
        state = dict(optimizer=opt, model=model, ema=ema, step=0)
        fs = step_fn.fused_for(state, torch.zeros(Bn, 3, R, R, device=dev))

        def step(i):
            b, t, z = (v.to(dev) for v in batches[i])
            loss = float(fs.loss_and_grads(b, t=t, z=z, seed=5))
            fs.optimizer_step(opt, ema, state['step'], optimize_fn.ssde_hyper)
            state['step'] += 1
            return loss
        return model, state, step

    model, state, step = make(0)
    with torch.no_grad():
        y0 = infer(model, xq, sq)
    step(0)
    with torch.no_grad():
        y1 = infer(model, xq, sq)          # the inference engine was lowered before the fused step touched the weights
    fresh = mutils.get_model("ncsnpp")(cfg).to(dev)
    fresh.load_state_dict(model.state_dict())
    with torch.no_grad():
        assert rel_err(y1, y0) > 1e-6
        assert rel_err(y1, infer(fresh, xq, sq)) < 2e-5    # (device vs torch packing of the Winograd weights: rounding)
    step(1)
    path = os.path.join(str(tmp_path), "ckpt", "checkpoint_1.pth")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    ssde_utils.save_checkpoint(path, state, data_parallel_prefix=True)      # the form the reference writes
    loss_a = step(2)
    params_a = [p.detach().clone() for p in model.parameters()]
    ema_a = [s.clone() for s in state['ema'].shadow_params]

    model2, state2, step2 = make(7)
    step2(0)                                # flat buffers and moments exist before the restore
    missing = ssde_utils.restore_checkpoint(os.path.join(str(tmp_path), "nope", "x.pth"), state2, dev)
    assert missing is state2 and state2['step'] == 1
    ssde_utils.restore_checkpoint(path, state2, dev)
    assert state2['step'] == 2
    loss_b = step2(2)
    assert loss_a == loss_b
    for a, b in zip(params_a, model2.parameters()):
        assert torch.equal(a, b.detach())
    for a, b in zip(ema_a, state2['ema'].shadow_params):
        assert torch.equal(a, b)

    # EMA swap (what the eval branch of step_fn and run_lib's sampling snapshots do)
    ema = state['ema']
    before = [p.detach().clone() for p in model.parameters()]
    with torch.no_grad():
        y_raw = infer(model, xq, sq)
    ema.store(model.parameters())
    ema.copy_to(model.parameters())
    for p, s in zip([p for p in model.parameters() if p.requires_grad], ema.shadow_params):
        assert torch.equal(p.detach(), s)
    fresh.load_state_dict(model.state_dict())
    with torch.no_grad():
        y_ema = infer(model, xq, sq)
        assert rel_err(y_ema, infer(fresh, xq, sq)) < 2e-5 and rel_err(y_ema, y_raw) > 1e-7
    ema.restore(model.parameters())
    for a, b in zip(before, model.parameters()):
        assert torch.equal(a, b.detach())
    with torch.no_grad():
        assert torch.equal(infer(model, xq, sq), y_raw)


def check_conv_split_reduction(dev, monkeypatch, n, repeats=3):
    """3x3 convolutions on small maps split the input-channel reduction over two workgroups that meet in dst
    (conv_mfma.hip, g_conv_sync).  With the full fused form (concat source, GroupNorm+SiLU prologue, bias, per-image
    addend, residual, scale, GroupNorm partials of the result): the split launch must match the torch reference, give
    the same bits on every repeat (the hand-over order is free, x + y == y + x), leave its flags clear for the next
    launch, and its GroupNorm partials must finalize to the statistics of its own output."""
    import ctypes as C
    import numpy as np
    import torch.nn.functional as F
    from score_sde_pytorch_amd import hipops as ops, _lib as L
    from score_sde_pytorch_amd.engine import pack_conv_weight
    lib = L.load()
    g = torch.Generator().manual_seed(12)
    for (c0, c1, cout, h) in ((256, 0, 256, 4), (128, 128, 192, 4), (128, 0, 64, 2)):
        cin = c0 + c1
        x0 = (torch.randn(n, h, h, c0, generator=g) * 1.5 + 0.3).to(dev)
        x1 = torch.randn(n, h, h, c1, generator=g).to(dev) if c1 else None
        xcat = torch.cat([x0, x1], -1) if c1 else x0
        w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
        b, gamma, beta = torch.randn(cout, generator=g), torch.randn(cin, generator=g), torch.randn(cin, generator=g)
        resid, ca = torch.randn(n, h, h, cout, generator=g).to(dev), torch.randn(n, cout, generator=g).to(dev)
        G = 32
        mean, rstd = ops.groupnorm_stats(x0, G, 1e-6, x2=x1)
        gn = (mean, rstd, gamma.to(dev), beta.to(dev), G)
        wp, bd = pack_conv_weight(w.to(dev)), b.to(dev)
        outs = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("SSDE_CONV_KSPLIT", mode)
            for rep in range(repeats if mode == "1" else 1):
                a = L.ConvArgs()
                ops._fill_src(a.main, x0, x1, L.PRO_GN_SILU, gn)
                dst = torch.full((n, h, h, cout), float("nan"), device=dev)
                a.w_main, a.ksize, a.stride, a.pad, a.h_in, a.w_in = wp.data_ptr(), 3, 1, 1, h, h
                a.n, a.h_out, a.w_out, a.c_out, a.out_scale, a.dst, a.tile = n, h, h, cout, 0.7, dst.data_ptr(), L.TILE_AUTO
                a.bias, a.chan_add, a.chan_add_ld, a.resid = bd.data_ptr(), ca.data_ptr(), cout, resid.data_ptr()
                a.flags = L.conv_route_flags()                 # (SSDE_CONV_KSPLIT=0 -> SSDE_CONVF_NO_KSPLIT)
                sl = lib.ssde_conv_gn_slices(C.byref(a))
                part = torch.full((n, max(sl, 1), cout // 4, 3), float("nan"), device=dev)
                if sl > 0:
                    a.gn_part = part.data_ptr()
                L.check(lib.ssde_conv2d(C.byref(a), ops._stream()))
                outs.setdefault(mode, []).append((dst, part, sl))
        ref = F.conv2d(F.silu(F.group_norm(xcat.cpu().permute(0, 3, 1, 2), G, gamma, beta, 1e-6)), w, b, padding=1)
        ref = ((ref + ca.cpu()[:, :, None, None]).permute(0, 2, 3, 1) + resid.cpu()) * 0.7
        y = outs["1"][0][0]
        assert _util.rel_err(y.cpu(), ref) < 2e-5, (c0, c1, cout, h)
        assert _util.rel_err(outs["0"][0][0].cpu(), ref) < 2e-5
        for dst, _, _ in outs["1"][1:]:
            assert torch.equal(dst, y), "split reduction is not reproducible"
        dst, part, sl = outs["1"][-1]
        if sl > 0:
            Go = 32 if (cout // 32) % 4 == 0 else 16
            f = L.GnFinalizeArgs()
            m2, r2 = torch.zeros(n, Go, device=dev), torch.zeros(n, Go, device=dev)
            f.part0, f.c0, f.slices0, f.n, f.groups, f.eps = part.data_ptr(), cout, sl, n, Go, 1e-6
            f.mean, f.rstd = m2.data_ptr(), r2.data_ptr()
            L.check(lib.ssde_gn_finalize(C.byref(f), ops._stream()))
            mr, rr = ops.groupnorm_stats(dst, Go, 1e-6)
            assert (m2 - mr).abs().max().item() < 1e-5 and ((r2 - rr).abs() / rr).max().item() < 1e-4


def check_gn_finalize_entry_counts(dev):
    """ssde_gn_finalize over synthetic partials ([N][slices][C/4][3] = mean, M2, count per image, slice, channel quad) against an
    fp64 merge: a group with few entries (a team of 16 lanes) and with many (SSDE_GN_TEAM_MAX_ENTRIES: a workgroup per group --
    the 128x128 / 256x256 levels of FFHQ-256), a channel concatenation whose boundary cuts a group, empty entries."""
    import ctypes as C
    from score_sde_pytorch_amd import _lib as L, hipops as ops
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    for n, c0, c1, groups, s0, s1 in [(2, 64, 32, 8, 200, 100), (3, 32, 0, 8, 1024, 0), (2, 64, 32, 8, 8, 4), (1, 128, 0, 32, 4096, 0)]:
        def parts(c, sl):
            if c == 0:
                return None
            cnt = torch.randint(0, 3, (n, sl, c // 4, 1), generator=g).float() * 32.0          # some entries are empty
            mean = torch.randn(n, sl, c // 4, 1, generator=g) * 2 + 0.5
            M2 = torch.rand(n, sl, c // 4, 1, generator=g) * cnt * 3
            return torch.cat([mean * (cnt > 0), M2, cnt], -1).contiguous()
        p0, p1 = parts(c0, s0), parts(c1, s1)
        cpq = (c0 + c1) // groups // 4
        mean_ref, rstd_ref = torch.zeros(n, groups, dtype=torch.float64), torch.zeros(n, groups, dtype=torch.float64)
        for i in range(n):
            for gi in range(groups):
                ent = []
                for q in range(gi * cpq, (gi + 1) * cpq):
                    src, qq = (p0, q) if q < c0 // 4 else (p1, q - c0 // 4)
                    ent.append(src[i, :, qq, :].double())
                e = torch.cat(ent, 0)
                cnt = e[:, 2].sum()
                m = (e[:, 0] * e[:, 2]).sum() / cnt
                M2 = (e[:, 1] + e[:, 2] * (e[:, 0] - m) ** 2).sum()
                mean_ref[i, gi], rstd_ref[i, gi] = m, 1.0 / torch.sqrt(M2 / cnt + 1e-6)
        f = L.GnFinalizeArgs()
        d0, d1 = p0.to(dev), (p1.to(dev) if p1 is not None else None)
        mo, ro = torch.zeros(n, groups, device=dev), torch.zeros(n, groups, device=dev)
        f.part0, f.part1 = d0.data_ptr(), (d1.data_ptr() if d1 is not None else None)
        f.c0, f.c1, f.slices0, f.slices1, f.n, f.groups, f.eps = c0, c1, s0, s1, n, groups, 1e-6
        f.mean, f.rstd = mo.data_ptr(), ro.data_ptr()
        L.check(lib.ssde_gn_finalize(C.byref(f), ops._stream()))
        mo2, ro2 = torch.zeros(n, groups, device=dev), torch.zeros(n, groups, device=dev)
        f.mean, f.rstd = mo2.data_ptr(), ro2.data_ptr()
        L.check(lib.ssde_gn_finalize(C.byref(f), ops._stream()))
        assert torch.equal(mo, mo2) and torch.equal(ro, ro2), "the merge must be reproducible"
        assert float((mo.cpu().double() - mean_ref).abs().max()) < 1e-5, (n, c0, c1, s0, s1)
        assert float(((ro.cpu().double() - rstd_ref).abs() / rstd_ref).max()) < 1e-5, (n, c0, c1, s0, s1)


def check_conv_small_cout(dev, big=False):
    """conv_small.hip: 3x3 / stride 1 convolutions onto at most four channels (the image heads, ncsnpp.py:329-337,368-375) --
    plain, then fully fused (concat source, GroupNorm + SiLU prologue, bias, per-image addend, residual before / after the
    scale), on maps that are not multiples of the 8x8 tile, with one, two and three 128-channel chunks and a ragged last
    chunk; against torch at the direct kernel's tolerance, and against the general direct kernel (SSDE_CONVF_NO_SMALL_COUT)."""
    import ctypes as C
    import numpy as np
    import torch.nn.functional as F
    from score_sde_pytorch_amd import hipops as ops, _lib as L
    from score_sde_pytorch_amd.engine import pack_conv_weight
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    cases = [(2, 128, 0, 4, 32, 32), (3, 16, 0, 3, 8, 8), (1, 8, 0, 1, 12, 20), (2, 96, 64, 4, 16, 16), (1, 256, 128, 4, 8, 8), (5, 40, 0, 2, 4, 4)]
    if big:
        cases += [(256, 128, 0, 4, 32, 32), (16, 128, 0, 4, 128, 128), (7, 256, 0, 3, 64, 64)]
    for (n, c0, c1, cout, h, w_) in cases:
        cin = c0 + c1
        x0 = torch.randn(n, h, w_, c0, generator=g) * 1.5 + 0.3
        x1 = torch.randn(n, h, w_, c1, generator=g) if c1 else None
        xcat = torch.cat([x0, x1], -1) if c1 else x0
        wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
        b, gamma, beta = torch.randn(cout, generator=g), torch.randn(cin, generator=g), torch.randn(cin, generator=g)
        resid, ca = torch.randn(n, h, w_, cout, generator=g), torch.randn(n, cout, generator=g)
        G = 32 if cin % 32 == 0 and (cin // 32) % 4 == 0 else cin // 4
        x0d, x1d = x0.to(dev), (x1.to(dev) if c1 else None)
        mean, rstd = ops.groupnorm_stats(x0d, G, 1e-6, x2=x1d)
        gn = (mean, rstd, gamma.to(dev), beta.to(dev), G)
        wp, bd, cad, rd = pack_conv_weight(wt.to(dev)), b.to(dev), ca.to(dev), resid.to(dev)
        xn = F.silu(F.group_norm(xcat.permute(0, 3, 1, 2), G, gamma, beta, 1e-6))
        for fused, post in ((False, 0), (True, 0), (True, 1)):
            outs = []
            for flags in (0, L.CONVF_NO_SMALL_COUT):
                a = L.ConvArgs()
                ops._fill_src(a.main, x0d, x1d, L.PRO_GN_SILU if fused else L.PRO_NONE, gn if fused else None)
                dst = torch.full((n, h, w_, cout), float("nan"), device=dev)
                a.w_main, a.ksize, a.stride, a.pad, a.h_in, a.w_in = wp.data_ptr(), 3, 1, 1, h, w_
                a.n, a.h_out, a.w_out, a.c_out, a.dst, a.tile = n, h, w_, cout, dst.data_ptr(), L.TILE_AUTO
                a.out_scale, a.flags = (0.7 if fused else 1.0), L.conv_route_flags() | flags
                a.bias = bd.data_ptr()
                if fused:
                    a.chan_add, a.chan_add_ld, a.resid, a.resid_post = cad.data_ptr(), cout, rd.data_ptr(), post
                assert flags or lib.ssde_conv_gn_slices(C.byref(a)) == 0
                assert lib.ssde_conv_lds_bytes(C.byref(a)) > 0, lib.ssde_last_error()
                L.check(lib.ssde_conv2d(C.byref(a), ops._stream()))
                outs.append(dst.cpu())
            if fused:
                core = (F.conv2d(xn, wt, b, padding=1) + ca[:, :, None, None]).permute(0, 2, 3, 1)
                ref = (core + resid) * 0.7 if not post else core * 0.7 + resid
            else:
                ref = F.conv2d(xcat.permute(0, 3, 1, 2), wt, b, padding=1).permute(0, 2, 3, 1)
            assert _util.rel_err(outs[0], ref) < 2e-5, (n, c0, c1, cout, h, w_, fused, post, _util.rel_err(outs[0], ref))
            assert _util.rel_err(outs[1], ref) < 2e-5 and _util.rel_err(outs[0], outs[1]) < 2e-5


def check_conv_winograd4(dev, big=False, regs=False):
    """conv_wino4.hip (F(4x4,3x3)): plain convolutions over the tilings it knows (part of one image, several whole
    images, ragged batch tails, cout tiles that are not full), then the fully fused form (concat source, GroupNorm + SiLU
    prologue, bias, per-image addend, residual, scale, GroupNorm partials of the result) against torch.  Tolerance 2e-5:
    F(4x4,3x3) rounds ~5x coarser than the direct kernel (tools/experiments/wino43_error_budget.py)."""
    import ctypes as C
    import numpy as np
    import torch.nn.functional as F
    from score_sde_pytorch_amd import hipops as ops, _lib as L
    from score_sde_pytorch_amd.engine import pack_wino4_weight, pack_wino4r_weight
    # regs=True: the same cases as a transform pass + the register-fed matrix kernel (conv_wino4r.hip, SSDE_TILE_WINOGRAD4R, weights
    # packed per lane), tolerances unchanged
    TILE = L.TILE_WINOGRAD4R if regs else L.TILE_WINOGRAD4
    if regs:
        pack_wino4_weight = pack_wino4r_weight
    lib = L.load()
    g = torch.Generator().manual_seed(2)
    # (>= 128 / 256 input channels and few workgroups: the reduction is split over two / four workgroups per tile, checked
    # both ways)
    plain = [(1, 8, 64, 16), (2, 32, 64, 8), (3, 16, 96, 8), (1, 12, 128, 32), (3, 128, 64, 8), (9, 132, 72, 8), (2, 260, 96, 8)]
    if regs:
        # the register-fed kernel's own split (ssde_conv_wino4r_splits: shares of an even number of stages that fit one round of
        # workgroups): two shares of 32 / 40 stages (256 / 320 channels, a ragged batch), four shares of 32 stages (512 channels)
        plain += [(5, 256, 128, 8), (17, 320, 80, 8), (3, 512, 64, 8), (2, 256, 64, 16)]
    if big:
        plain += [(16, 128, 128, 32), (9, 256, 256, 16), (20, 256, 256, 8), (2, 128, 128, 64)]
        if regs:
            plain += [(256, 256, 256, 8), (256, 512, 256, 8), (128, 256, 256, 8)]      # the BASELINE sampler's / training step's 8x8 shapes: 2 / 2 / 4 shares
    for (n, cin, cout, h) in plain:
        if regs and cin % 8:
            continue                                # (the register-fed kernel runs its stages in pairs: input channels % 8 == 0)
        x = torch.randn(n, cin, h, h, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
        b = torch.randn(cout, generator=g)
        ref = F.conv2d(x, w, b, padding=1)
        for ksplit in (("1", "0") if cin >= 128 else ("1",)):
            os.environ["SSDE_CONV_KSPLIT"] = ksplit
            try:
                y = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().to(dev), w.to(dev), b.to(dev), tile=TILE)
            finally:
                del os.environ["SSDE_CONV_KSPLIT"]
            assert _util.rel_err(y.cpu().permute(0, 3, 1, 2), ref) < 2e-5, (n, cin, cout, h, ksplit)
    fused = [(5, 32, 16, 64, 16), (9, 64, 0, 96, 8), (3, 24, 32, 128, 32), (10, 96, 32, 64, 8), (4, 192, 64, 64, 8)]
    if big:
        fused += [(12, 128, 128, 128, 32), (10, 256, 256, 256, 16)]
    for (n, c0, c1, cout, h) in fused:
        cin = c0 + c1
        x0 = torch.randn(n, h, h, c0, generator=g) * 1.5 + 0.3
        x1 = torch.randn(n, h, h, c1, generator=g) if c1 else None
        xcat = torch.cat([x0, x1], -1) if c1 else x0
        w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
        b, gamma, beta = torch.randn(cout, generator=g), torch.randn(cin, generator=g), torch.randn(cin, generator=g)
        resid, ca = torch.randn(n, h, h, cout, generator=g), torch.randn(n, cout, generator=g)
        G = 32 if cin % 32 == 0 and (cin // 32) % 4 == 0 else cin // 4
        x0d, x1d = x0.to(dev), (x1.to(dev) if c1 else None)
        mean, rstd = ops.groupnorm_stats(x0d, G, 1e-6, x2=x1d)
        a = L.ConvArgs()
        gn = (mean, rstd, gamma.to(dev), beta.to(dev), G)          # _fill_src borrows the pointers: keep the tensors
        ops._fill_src(a.main, x0d, x1d, L.PRO_GN_SILU, gn)
        wp, bd, cad, rd = pack_wino4_weight(w.to(dev)), b.to(dev), ca.to(dev), resid.to(dev)
        dst = torch.full((n, h, h, cout), float("nan"), device=dev)
        a.w_main, a.ksize, a.stride, a.pad, a.h_in, a.w_in = wp.data_ptr(), 3, 1, 1, h, h
        a.n, a.h_out, a.w_out, a.c_out, a.out_scale, a.dst, a.tile = n, h, h, cout, 0.7, dst.data_ptr(), TILE
        a.bias, a.chan_add, a.chan_add_ld, a.resid = bd.data_ptr(), cad.data_ptr(), cout, rd.data_ptr()
        vbuf = torch.full((36 * n * (h // 4) * (h // 4) * cin,), float("nan"), device=dev) if regs else None
        a.wino_v = vbuf.data_ptr() if regs else None
        sl = lib.ssde_conv_gn_slices(C.byref(a))
        assert sl > 0
        part = torch.full((n, sl, cout // 4, 3), float("nan"), device=dev)
        a.gn_part = part.data_ptr()
        L.check(lib.ssde_conv2d(C.byref(a), ops._stream()))
        xn = F.silu(F.group_norm(xcat.permute(0, 3, 1, 2), G, gamma, beta, 1e-6))
        ref = ((F.conv2d(xn, w, b, padding=1) + ca[:, :, None, None]).permute(0, 2, 3, 1) + resid) * 0.7
        assert _util.rel_err(dst.cpu(), ref) < 2e-5, (n, c0, c1, cout, h)
        Go = cout // 4
        f = L.GnFinalizeArgs()
        m2, r2 = torch.zeros(n, Go, device=dev), torch.zeros(n, Go, device=dev)
        f.part0, f.c0, f.slices0, f.n, f.groups, f.eps = part.data_ptr(), cout, sl, n, Go, 1e-6
        f.mean, f.rstd = m2.data_ptr(), r2.data_ptr()
        L.check(lib.ssde_gn_finalize(C.byref(f), ops._stream()))
        mr, rr = ops.groupnorm_stats(dst, Go, 1e-6)
        assert (m2 - mr).abs().max().item() < 1e-5 and ((r2 - rr).abs() / rr).max().item() < 1e-4


def check_conv_winograd4_two_kernels(dev, big=False):
    """F(4x4,3x3) as an input-transform pass (wino4_xform.hip) + the register-fed matrix kernel (conv_wino4r.hip,
    SSDE_TILE_WINOGRAD4R) against torch and against conv_wino4.hip: with the transformed input taken from conv_wino4's own
    by-product (SSDE_CONVF_V_GIVEN: the caller filled wino_v) the matrix kernel must reproduce conv_wino4's output BIT FOR BIT
    (same products, same order); with its own transform
    pass the result is within rounding of it and within the F(4x4,3x3) tolerance of torch.  Tilings: part of one image, whole
    images, ragged batch tails, cout tiles that are not full; the fused epilogue and the GroupNorm partials."""
    import ctypes as C
    import numpy as np
    import torch.nn.functional as F
    from score_sde_pytorch_amd import hipops as ops, _lib as L
    from score_sde_pytorch_amd.engine import pack_wino4_weight, pack_wino4r_weight
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    cases = [(5, 32, 16, 64, 16), (9, 64, 0, 96, 8), (3, 24, 32, 128, 32), (1, 8, 0, 64, 16), (2, 128, 0, 256, 16), (1, 16, 0, 40, 64)]
    if big:
        cases += [(12, 128, 128, 128, 32), (10, 256, 256, 256, 16), (33, 256, 0, 256, 16)]
    for (n, c0, c1, cout, h) in cases:
        cin = c0 + c1
        x0 = torch.randn(n, h, h, c0, generator=g) * 1.5 + 0.3
        x1 = torch.randn(n, h, h, c1, generator=g) if c1 else None
        xcat = torch.cat([x0, x1], -1) if c1 else x0
        w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
        b, gamma, beta = torch.randn(cout, generator=g), torch.randn(cin, generator=g), torch.randn(cin, generator=g)
        resid, ca = torch.randn(n, h, h, cout, generator=g), torch.randn(n, cout, generator=g)
        G = 32 if cin % 32 == 0 and (cin // 32) % 4 == 0 else cin // 4
        x0d, x1d = x0.to(dev), (x1.to(dev) if c1 else None)
        mean, rstd = ops.groupnorm_stats(x0d, G, 1e-6, x2=x1d)
        a = L.ConvArgs()
        gn = (mean, rstd, gamma.to(dev), beta.to(dev), G)
        ops._fill_src(a.main, x0d, x1d, L.PRO_GN_SILU, gn)
        wp, bd, cad, rd = pack_wino4_weight(w.to(dev)), b.to(dev), ca.to(dev), resid.to(dev)
        wpr = pack_wino4r_weight(w.to(dev))
        assert wpr.numel() == wp.numel()
        a.ksize, a.stride, a.pad, a.h_in, a.w_in = 3, 1, 1, h, h
        a.n, a.h_out, a.w_out, a.c_out, a.out_scale = n, h, h, cout, 0.7
        a.bias, a.chan_add, a.chan_add_ld, a.resid = bd.data_ptr(), cad.data_ptr(), cout, rd.data_ptr()
        xn = F.silu(F.group_norm(xcat.permute(0, 3, 1, 2), G, gamma, beta, 1e-6))
        ref = ((F.conv2d(xn, w, b, padding=1) + ca[:, :, None, None]).permute(0, 2, 3, 1) + resid) * 0.7
        out, parts = {}, {}
        v = torch.full((36 * n * (h // 4) * (h // 4) * cin,), float("nan"), device=dev)
        a.wino_v = v.data_ptr()
        # (SSDE_CONVF_NO_KSPLIT: a split reduction of conv_wino4.hip sums in another order)
        for name, tile, flags in (("one kernel", L.TILE_WINOGRAD4, 0),
                                  ("register-fed matrix kernel on the by-product", L.TILE_WINOGRAD4R, L.CONVF_V_GIVEN),
                                  ("two kernels", L.TILE_WINOGRAD4R, 0)):
            a.tile, a.flags = tile, flags | L.CONVF_NO_KSPLIT
            a.w_main = (wpr if tile == L.TILE_WINOGRAD4R else wp).data_ptr()
            dst = torch.full((n, h, h, cout), float("nan"), device=dev)
            a.dst, a.gn_part = dst.data_ptr(), None
            sl = lib.ssde_conv_gn_slices(C.byref(a))
            assert sl > 0, (name, lib.ssde_last_error())
            part = torch.full((n, sl, cout // 4, 3), float("nan"), device=dev)
            a.gn_part = part.data_ptr()
            if name.startswith("two kernels"):
                v.fill_(float("nan"))
            L.check(lib.ssde_conv2d(C.byref(a), ops._stream()))
            assert _util.rel_err(dst.cpu(), ref) < 2e-5, (name, n, c0, c1, cout, h, _util.rel_err(dst.cpu(), ref))
            out[name], parts[name] = dst.cpu(), part.cpu()
        name = "register-fed matrix kernel on the by-product"
        assert torch.equal(out[name], out["one kernel"]), (name, n, c0, c1, cout, h)
        assert torch.equal(parts[name], parts["one kernel"]), name
        assert _util.rel_err(out["two kernels"], out["one kernel"]) < 5e-6


def check_wgrad_wino4_streamk(dev, big=False):
    """F(4x4,3x3) weight gradient with the stream-K work split of its GEMM (the default; SSDE_WGRAD4_STREAMK=0 = the plain split: equal runs of K stages cut
    across the 36 positions, a run that crosses a position boundary leaves two partial slabs) against torch autograd and
    against the plain split: GroupNorm + SiLU prologue, a concatenated source, (co, ci) tile grids of 1 and 4, ragged last
    stages, runs of different lengths (SSDE_NUM_CUS)."""
    import torch.nn.functional as F
    from score_sde_pytorch_amd import hipops as ops, _lib as L
    g = torch.Generator().manual_seed(21)
    cases = [(5, 32, 64, 0, 64, "256"), (3, 32, 96, 32, 160, "256"), (9, 16, 128, 0, 128, "64"), (2, 32, 32, 0, 32, "40")]
    if big:
        cases += [(32, 16, 256, 0, 256, "256"), (16, 32, 128, 128, 128, "256")]
    for (n, h, c1, c2, cout, cus) in cases:
        k = c1 + c2
        x = torch.randn(n, k, h, h, generator=g).requires_grad_(False)
        G = min(32, k // 4)
        gamma, beta = 1 + 0.1 * torch.randn(k, generator=g), 0.1 * torch.randn(k, generator=g)
        w = (torch.randn(cout, k, 3, 3, generator=g) / np.sqrt(9 * k)).requires_grad_()
        gy = torch.randn(n, cout, h, h, generator=g)
        F.conv2d(F.silu(F.group_norm(x, G, gamma, beta, 1e-6)), w, padding=1).backward(gy)
        xa = x[:, :c1].permute(0, 2, 3, 1).contiguous().to(dev)
        xb = x[:, c1:].permute(0, 2, 3, 1).contiguous().to(dev) if c2 else None
        mean, rstd = ops.groupnorm_stats(xa, G, 1e-6, x2=xb)
        gn = (mean, rstd, gamma.to(dev), beta.to(dev), G)
        gyd = gy.permute(0, 2, 3, 1).contiguous().to(dev)
        out = {}
        for sk in ("0", "1"):
            env = {"SSDE_WGRAD_WINOGRAD": "44", "SSDE_WGRAD4_STREAMK": sk, "SSDE_NUM_CUS": cus}
            os.environ.update(env)
            try:
                dw = torch.zeros(cout, k, 3, 3, device=dev)
                ops.conv_wgrad(xa, gyd, 3, dw, pad=1, x2=xb, pro=L.PRO_GN_SILU, gn=gn)
                dw2 = torch.zeros(cout, k, 3, 3, device=dev)
                ops.conv_wgrad(xa, gyd, 3, dw2, pad=1, x2=xb, pro=L.PRO_GN_SILU, gn=gn)
            finally:
                for key in env:
                    os.environ.pop(key)
            assert torch.equal(dw, dw2), (sk, n, h, k, cout)                     # fixed order of the partial sums
            assert rel_err(dw, w.grad) < 2e-4, (sk, n, h, k, cout, rel_err(dw, w.grad))
            out[sk] = dw.cpu()
        assert rel_err(out["1"], out["0"]) < 2e-5, (n, h, k, cout)
