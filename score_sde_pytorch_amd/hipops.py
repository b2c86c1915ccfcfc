"""Single-op Python entry points over the C ABI (one HIP launch each, current torch stream).

Tensors are NHWC float32 CUDA tensors unless stated.  These wrappers exist for the op-level
parity tests and for `score_sde_pytorch_amd.op` (the reference's `op` package surface); the
U-Net itself runs through `engine.UNetEngine` programs, not through here.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .engine import pack_conv_weight, pack_matrix, pack_wino_weight, pack_wino4_weight, pack_wino4r_weight


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("score_sde_pytorch_amd HIP op called with a %s tensor (no CPU fallback)" % t.device.type)


def _p(t):
    return None if t is None else t.data_ptr()


def _fill_src(s, t, t2=None, pro=L.PRO_NONE, gn=None):
    s.p0, s.c0 = _p(t), t.shape[-1]
    s.p1, s.c1 = (_p(t2), t2.shape[-1]) if t2 is not None else (None, 0)
    s.pro_mode = pro
    if gn is not None:
        mean, rstd, gamma, beta, groups = gn
        s.gn_groups, s.gn_mean, s.gn_rstd, s.gn_gamma, s.gn_beta = groups, _p(mean), _p(rstd), _p(gamma), _p(beta)


def groupnorm_stats(x, groups, eps=1e-6, x2=None, slices=1):
    """(mean, rstd) [N, G] of the channel-concat of NHWC tensors x (and x2)."""
    _need_cuda(x, x2)
    n = x.shape[0]
    hw = int(np.prod(x.shape[1:-1]))
    mean = torch.empty(n, groups, device=x.device)
    rstd = torch.empty(n, groups, device=x.device)
    a = L.GnStatsArgs()
    a.p0, a.c0 = _p(x), x.shape[-1]
    a.p1, a.c1 = (_p(x2), x2.shape[-1]) if x2 is not None else (None, 0)
    a.n, a.hw, a.groups, a.eps = n, hw, groups, eps
    a.mean, a.rstd = _p(mean), _p(rstd)
    scratch = torch.empty(n * slices * groups * 2, device=x.device) if slices > 1 else None
    a.scratch, a.slices = _p(scratch), slices
    L.check(L.load().ssde_groupnorm_stats(C.byref(a), _stream()), "ssde_groupnorm_stats")
    return mean, rstd


def conv2d(x=None, weight=None, bias=None, stride=1, pad=1, x2=None, pro=L.PRO_NONE, gn=None,
           aux=None, aux2=None, aux_weight=None, aux_pro=L.PRO_NONE, aux_gn=None,
           chan_add=None, resid=None, scale=1.0, tile=L.TILE_AUTO, out_hw=None, flags=None):
    """k x k (k=3) conv of NHWC `x` (weight OIHW) plus optional 1x1 conv of `aux` (weight [Cout, Cin]).
    flags: SSDE_CONVF_* routing switches (None: the A/B variables of the environment, _lib.conv_route_flags)."""
    _need_cuda(x, aux, resid)
    a = L.ConvArgs()
    keep = []
    if x is not None:
        n, h_in, w_in = x.shape[0], x.shape[1], x.shape[2]
        c_out = weight.shape[0]
        h_out = (h_in + 2 * pad - 3) // stride + 1
        w_out = (w_in + 2 * pad - 3) // stride + 1
        if out_hw is not None:      # top-left crop of the full result (transposed strided convolutions)
            h_out, w_out = out_hw
        _fill_src(a.main, x, x2, pro, gn)
        packer = {L.TILE_WINOGRAD: pack_wino_weight, L.TILE_WINOGRAD4: pack_wino4_weight,
                  L.TILE_WINOGRAD4R: pack_wino4r_weight}.get(tile, pack_conv_weight)
        wp = packer(weight.to(x.device)); keep.append(wp)
        if tile == L.TILE_WINOGRAD4R:      # the two-kernel form needs the transformed-input buffer
            wv = torch.empty(36 * n * (h_in // 4) * (w_in // 4) * (x.shape[-1] + (x2.shape[-1] if x2 is not None else 0)), device=x.device)
            keep.append(wv)
            a.wino_v = _p(wv)
        a.w_main, a.ksize, a.stride, a.pad, a.h_in, a.w_in = _p(wp), 3, stride, pad, h_in, w_in
    else:
        n, h_out, w_out = aux.shape[0], aux.shape[1], aux.shape[2]
        c_out = aux_weight.shape[0]
        a.ksize, a.stride, a.pad = 0, 1, 0
    if aux is not None:
        _fill_src(a.aux, aux, aux2, aux_pro, aux_gn)
        wa = pack_matrix(aux_weight.to(aux.device)); keep.append(wa)
        a.w_aux = _p(wa)
    dev = (x if x is not None else aux).device
    dst = torch.empty(n, h_out, w_out, c_out, device=dev)
    a.n, a.h_out, a.w_out, a.c_out, a.tile = n, h_out, w_out, c_out, tile
    a.bias = _p(bias)
    if chan_add is not None:
        a.chan_add, a.chan_add_ld = _p(chan_add), chan_add.shape[-1]
    a.resid, a.out_scale, a.dst = _p(resid), scale, _p(dst)
    a.flags = L.conv_route_flags() if flags is None else flags
    L.check(L.load().ssde_conv2d(C.byref(a), _stream()), "ssde_conv2d")
    return dst


def upfirdn2d_nhwc(x, kernel, up=1, down=1, pad=(0, 0), pro=L.PRO_NONE, gn=None, dual=False, accumulate_into=None):
    """dual=True also returns the same filter applied to the source without its prologue (one launch, dst2);
    accumulate_into: an existing output tensor the result is added to."""
    _need_cuda(x)
    n, h, w, c = x.shape
    kh, kw = kernel.shape
    h_out = (h * up + pad[0] + pad[1] - kh) // down + 1
    w_out = (w * up + pad[0] + pad[1] - kw) // down + 1
    dst = torch.empty(n, h_out, w_out, c, device=x.device) if accumulate_into is None else accumulate_into
    dst2 = torch.empty(n, h_out, w_out, c, device=x.device) if dual else None
    a = L.UpfirdnArgs()
    _fill_src(a.src, x, None, pro, gn)
    a.accumulate = int(accumulate_into is not None)
    a.dst2 = _p(dst2) if dual else None
    a.n, a.h_in, a.w_in, a.c, a.h_out, a.w_out = n, h, w, c, h_out, w_out
    a.up, a.down, a.pad0, a.pad1, a.kh, a.kw = up, down, pad[0], pad[1], kh, kw
    for i, v in enumerate(np.asarray(kernel.detach().cpu(), dtype=np.float32).reshape(-1).tolist()):
        a.k[i] = v
    a.dst = _p(dst)
    L.check(L.load().ssde_upfirdn2d(C.byref(a), _stream()), "ssde_upfirdn2d")
    return (dst, dst2) if dual else dst


def attention(qkv, channels):
    """qkv [N, L, 3C] -> softmax(q k^T / sqrt(C)) v  [N, L, C]."""
    _need_cuda(qkv)
    n, l = qkv.shape[0], qkv.shape[1]
    dst = torch.empty(n, l, channels, device=qkv.device)
    a = L.AttnArgs()
    a.qkv, a.dst, a.n, a.l, a.c, a.scale = _p(qkv), _p(dst), n, l, channels, float(int(channels) ** (-0.5))
    a.flags = L.attn_route_flags()
    L.check(L.load().ssde_attention(C.byref(a), _stream()), "ssde_attention")
    return dst


def embed(cond, table, dim, kind):
    _need_cuda(cond, table)
    dst = torch.empty(cond.shape[0], dim, device=cond.device)
    a = L.EmbedArgs()
    a.cond, a.w, a.dst, a.n, a.dim, a.kind = _p(cond), _p(table), _p(dst), cond.shape[0], dim, kind
    L.check(L.load().ssde_embed(C.byref(a), _stream()), "ssde_embed")
    return dst


def to_nhwc(x, c_pad=None, a=1.0, b=0.0):
    _need_cuda(x)
    n, c, h, w = x.shape
    c_pad = c_pad or c
    dst = torch.empty(n, h, w, c_pad, device=x.device)
    args = L.ToNhwcArgs()
    args.src, args.dst, args.n, args.c, args.h, args.w, args.c_pad, args.a, args.b = _p(x), _p(dst), n, c, h, w, c_pad, a, b
    L.check(L.load().ssde_to_nhwc(C.byref(args), _stream()), "ssde_to_nhwc")
    return dst


def to_nchw(x, c=None, mode=0, v=None):
    _need_cuda(x)
    n, h, w, c_src = x.shape
    c = c or c_src
    dst = torch.empty(n, c, h, w, device=x.device)
    args = L.ToNchwArgs()
    args.src, args.dst, args.n, args.c, args.h, args.w, args.c_src, args.mode, args.v = _p(x), _p(dst), n, c, h, w, c_src, mode, _p(v)
    L.check(L.load().ssde_to_nchw(C.byref(args), _stream()), "ssde_to_nchw")
    return dst


def fused_bias_act(x, bias=None, channels=1, inner=1, act=1, alpha=0.2, scale=1.0, grad=0, ref=None):
    _need_cuda(x)
    x = x.contiguous()
    dst = torch.empty_like(x)
    a = L.BiasActArgs()
    a.src, a.bias, a.dst, a.numel, a.channels, a.inner, a.act, a.alpha, a.scale = \
        _p(x), _p(bias), _p(dst), x.numel(), channels, inner, act, alpha, scale
    a.grad, a.ref = grad, _p(ref)
    L.check(L.load().ssde_fused_bias_act(C.byref(a), _stream()), "ssde_fused_bias_act")
    return dst


def randn(numel, seed, device, step_ptr=None, stream_id=0):
    dst = torch.empty(numel, device=device)
    a = L.RandnArgs()
    a.dst, a.numel, a.seed, a.step_ptr, a.stream_id = _p(dst), numel, seed, _p(step_ptr), stream_id
    L.check(L.load().ssde_randn(C.byref(a), _stream()), "ssde_randn")
    return dst


# ------------------------------------------------------------------------------- training-path ops
def set_dropout(s, p, seed_t, salt):
    """Arm the train-mode dropout of a source (ssde.h: drop_thresh / drop_scale / drop_seed / drop_salt)."""
    s.drop_thresh = min(int(round(p * 2.0 ** 32)), 2 ** 32 - 1) if p > 0 else 0
    s.drop_scale = 1.0 / (1.0 - p) if p > 0 else 1.0
    s.drop_seed = _p(seed_t) if p > 0 else None
    s.drop_salt = salt & 0xFFFFFFFF


def conv_wgrad(x, g, ksize, dw, stride=1, pad=1, x2=None, pro=L.PRO_NONE, gn=None, g_off=0, c_out=None, cin_store=None,
               transpose_out=False, scale=1.0, splits=0, dropout=None, flags=None):
    """dw += scale * sum_pixels g[:, g_off:g_off+c_out]^T pro(x)[shifted]; x, g NHWC; dw in the reference layout.
    flags: SSDE_WGRADF_* (None: the A/B variables of the environment, _lib.wgrad_route_flags)."""
    _need_cuda(x, g, dw)
    a = L.WgradArgs()
    a.flags = L.wgrad_route_flags() if flags is None else flags
    _fill_src(a.src, x, x2, pro, gn)
    if dropout is not None:
        set_dropout(a.src, *dropout)
    a.g, a.g_ld, a.g_off = _p(g), g.shape[-1], g_off
    a.n, a.h_in, a.w_in, a.h_out, a.w_out = x.shape[0], x.shape[1], x.shape[2], g.shape[1], g.shape[2]
    a.c_out = c_out if c_out is not None else g.shape[-1] - g_off
    ctot = x.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
    a.ksize, a.stride, a.pad = ksize, stride, pad
    a.cin_store = cin_store if cin_store is not None else ctot
    a.transpose_out, a.splits, a.scale, a.dw = int(transpose_out), splits, scale, _p(dw)
    a.splits = 0
    need = wgrad_scratch_floats(a)       # slabs of the library's preferred split
    a.splits = splits
    if splits > 1:                       # explicit split (tests): a slab is at most one padded copy of dw per tile grid
        bt = 64 if ksize == 3 else 128
        slab = ksize * ksize * (-(-a.c_out // bt) * bt) * (-(-a.cin_store // bt) * bt)
        need = max(need, slab * splits)
    scratch = torch.empty(max(int(need), 1), device=x.device)
    a.scratch, a.scratch_floats = _p(scratch), scratch.numel()
    L.check(L.load().ssde_conv_wgrad(C.byref(a), _stream()), "ssde_conv_wgrad")
    return dw


def wgrad_scratch_floats(a):
    """Floats of scratch the library's preferred pixel split of this weight-gradient launch needs (host-side query)."""
    r = int(L.load().ssde_wgrad_scratch_floats(C.byref(a)))
    if r < 0:
        L.check(r, "ssde_wgrad_scratch_floats")
    return r


def colsum(g, c=None, g_off=0, scale=1.0, per_sample=None, ps_off=0, total=None, total2=None, defer=False):
    """defer=True: only the pass over g runs; returns the job (a dict with the partials) for colsum_finish()."""
    _need_cuda(g)
    n = g.shape[0]
    hw = int(np.prod(g.shape[1:-1])) if g.dim() > 2 else 1
    a = L.ColsumArgs()
    a.g, a.g_ld, a.g_off, a.n, a.hw = _p(g), g.shape[-1], g_off, n, hw
    a.c = c if c is not None else g.shape[-1] - g_off
    a.scale = scale
    if per_sample is not None:
        a.per_sample, a.ps_ld, a.ps_off = _p(per_sample), per_sample.shape[-1], ps_off
    scratch = torch.empty(n * (colsum_slices(hw) + 1) * a.c, device=g.device)
    a.scratch = _p(scratch)
    if defer:
        a.per_sample, a.flags = None, L.COLSUMF_DEFER
        L.check(L.load().ssde_colsum(C.byref(a), _stream()), "ssde_colsum")
        return dict(part=scratch, per_sample=per_sample, total=total, total2=total2, n=n, slices=colsum_slices(hw), c=a.c,
                    ps_ld=per_sample.shape[-1] if per_sample is not None else 0, ps_off=ps_off)
    a.total, a.total2 = _p(total), _p(total2)
    L.check(L.load().ssde_colsum(C.byref(a), _stream()), "ssde_colsum")


def colsum_finish(jobs):
    """One launch that finishes up to L.FINISH_JOBS deferred column sums (ssde_colsum_finish)."""
    a = L.ColsumFinishArgs()
    a.count = len(jobs)
    for i, j in enumerate(jobs):
        q = a.job[i]
        q.part, q.per_sample, q.total, q.total2 = _p(j["part"]), _p(j["per_sample"]), _p(j["total"]), _p(j["total2"])
        q.n, q.slices, q.c, q.ps_ld, q.ps_off = j["n"], j["slices"], j["c"], j["ps_ld"], j["ps_off"]
    L.check(L.load().ssde_colsum_finish(C.byref(a), _stream()), "ssde_colsum_finish")


def gn_bwd_finish(jobs):
    """One launch that writes dgamma / dbeta of up to L.FINISH_JOBS deferred GroupNorm backward calls (ssde_gn_bwd_finish)."""
    a = L.GnBwdFinishArgs()
    a.count = len(jobs)
    for i, j in enumerate(jobs):
        q = a.job[i]
        q.scratch, q.dgamma, q.dbeta, q.rows, q.c = _p(j["scratch"]), _p(j["dgamma"]), _p(j["dbeta"]), j["rows"], j["c"]
    L.check(L.load().ssde_gn_bwd_finish(C.byref(a), _stream()), "ssde_gn_bwd_finish")


def colsum_slices(hw):
    return max(1, min(32, hw // 64))


def gn_backward(x, dp, gn, pro, x2=None, dropout=None, slices=1, scale=1.0, acc=(False, False), want=(True, True), one_call=True,
                dx=None, dx2=None, defer_params=False):
    """GroupNorm(+SiLU)(+dropout) backward: returns (dx, dx2, dgamma, dbeta) for dp = d loss / d pro(x).
    one_call: ssde_gn_bwd_reduce with its gradient destinations set (ABI 7: one pass over dp and x where the shape allows);
    otherwise the reduction and ssde_prologue_bwd as two calls."""
    _need_cuda(x, dp)
    n = x.shape[0]
    hw = int(np.prod(x.shape[1:-1]))
    ctot = x.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
    groups = gn[4]
    r = L.GnBwdReduceArgs()
    r.flags = L.gn_bwd_route_flags()
    _fill_src(r.src, x, x2, pro, gn)
    if dropout is not None:
        set_dropout(r.src, *dropout)
    sums = torch.empty(n, groups, 2, device=x.device)
    dgamma, dbeta = torch.empty(ctot, device=x.device), torch.empty(ctot, device=x.device)
    scratch = torch.empty(n * slices * ctot * 2, device=x.device)
    r.dp, r.n, r.hw, r.sums, r.dgamma, r.dbeta, r.scratch, r.slices = _p(dp), n, hw, _p(sums), _p(dgamma), _p(dbeta), _p(scratch), slices
    if dx is None:
        dx = torch.zeros_like(x) if want[0] else None
    if dx2 is None:
        dx2 = torch.zeros_like(x2) if (x2 is not None and want[1]) else None
    if one_call and (dx is not None or dx2 is not None):
        r.g0, r.g1, r.acc0, r.acc1, r.scale = _p(dx), _p(dx2), int(acc[0]), int(acc[1]), scale
        if defer_params:         # dgamma / dbeta by a finishing launch (here: of this one job) instead of the call's own
            r.flags |= L.GNBWDF_DEFER_PARAMS
            r.dgamma = r.dbeta = None
            L.check(L.load().ssde_gn_bwd_reduce(C.byref(r), _stream()), "ssde_gn_bwd_reduce")
            gn_bwd_finish([dict(scratch=scratch, dgamma=dgamma, dbeta=dbeta, rows=int(L.load().ssde_gn_bwd_scratch_rows(C.byref(r))), c=ctot)])
            return dx, dx2, dgamma, dbeta
        L.check(L.load().ssde_gn_bwd_reduce(C.byref(r), _stream()), "ssde_gn_bwd_reduce")
        return dx, dx2, dgamma, dbeta
    L.check(L.load().ssde_gn_bwd_reduce(C.byref(r), _stream()), "ssde_gn_bwd_reduce")
    prologue_bwd(x, dp, pro, dx, dx2, x2=x2, gn=gn, sums=sums, dropout=dropout, scale=scale, acc=acc)
    return dx, dx2, dgamma, dbeta


def prologue_bwd(x, dp, pro, g0, g1=None, x2=None, gn=None, sums=None, dropout=None, scale=1.0, acc=(False, False),
                 dp_off=0, c0=None, c1=0):
    a = L.PrologueBwdArgs()
    if x is not None:
        _fill_src(a.src, x, x2, pro, gn)
    else:
        a.src.c0, a.src.c1, a.src.pro_mode = c0, c1, pro
    if dropout is not None:
        set_dropout(a.src, *dropout)
    ref = x if x is not None else (g0 if g0 is not None else g1)
    a.dp, a.dp_ld, a.dp_off = _p(dp), dp.shape[-1], dp_off
    a.n = ref.shape[0]
    a.hw = int(np.prod(ref.shape[1:-1])) if ref.dim() > 2 else 1
    a.sums, a.scale, a.acc0, a.acc1, a.g0, a.g1 = _p(sums), scale, int(acc[0]), int(acc[1]), _p(g0), _p(g1)
    L.check(L.load().ssde_prologue_bwd(C.byref(a), _stream()), "ssde_prologue_bwd")


def attention_bwd(qkv, o, d_o, channels):
    _need_cuda(qkv, o, d_o)
    n, l = qkv.shape[0], qkv.shape[1]
    dqkv = torch.empty_like(qkv)
    stats = torch.empty(n, l, 4, device=qkv.device)
    a = L.AttnBwdArgs()
    a.qkv, a.o, a.d_o, a.dqkv, a.stats = _p(qkv), _p(o), _p(d_o), _p(dqkv), _p(stats)
    a.n, a.l, a.c, a.scale = n, l, channels, float(int(channels) ** (-0.5))
    L.check(L.load().ssde_attention_bwd(C.byref(a), _stream()), "ssde_attention_bwd")
    return dqkv


def sample_update(x, y=None, b=None, z=None, c=None, a=None):
    """x_mean = a[n] x + b[n] y ; x_out = x_mean + c[n] z with per-sample coefficient vectors (ssde_sample_update): the
    arithmetic of one predictor / corrector update of the generic sampler path.  Returns (x_out, x_mean)."""
    _need_cuda(x)
    x = x.contiguous()
    n, per = x.shape[0], x[0].numel()
    vec = lambda v: None if v is None else torch.as_tensor(v, dtype=torch.float32, device=x.device).expand(n).contiguous()   # noqa: E731
    ten = lambda t: None if t is None else t.to(torch.float32).contiguous()                                                   # noqa: E731
    y, z, a, b, c = ten(y), ten(z), vec(a), vec(b), vec(c)
    x_mean, x_out = torch.empty_like(x), torch.empty_like(x)
    args = L.SampleUpdateArgs()
    args.x, args.y, args.z, args.a, args.b, args.c = _p(x), _p(y), _p(z), _p(a), _p(b), _p(c)
    args.x_mean, args.x_out, args.n, args.per = _p(x_mean), _p(x_out), n, per
    L.check(L.load().ssde_sample_update(C.byref(args), _stream()), "ssde_sample_update")
    return x_out, x_mean


def dsm_loss(score, z, s, g2=None, reduce_mean=False, likelihood_weighting=False, want_grad=True, grad_scale=1.0):
    _need_cuda(score, z, s)
    n = score.shape[0]
    per = score.numel() // n
    dscore = torch.empty_like(score) if want_grad else None
    losses, loss = torch.empty(n, device=score.device), torch.empty(1, device=score.device)
    a = L.DsmLossArgs()
    a.score, a.z, a.s, a.g2, a.dscore, a.losses, a.loss = _p(score), _p(z), _p(s), _p(g2), _p(dscore), _p(losses), _p(loss)
    a.n, a.per, a.reduce_mean, a.likelihood_weighting, a.grad_scale = n, per, int(reduce_mean), int(likelihood_weighting), grad_scale
    L.check(L.load().ssde_dsm_loss(C.byref(a), _stream()), "ssde_dsm_loss")
    return loss, losses, dscore


def perturb(x, z, s, a_coef=None):
    _need_cuda(x, z, s)
    dst = torch.empty_like(x)
    a = L.PerturbArgs()
    a.x, a.z, a.a, a.s, a.dst, a.n, a.per = _p(x), _p(z), _p(a_coef), _p(s), _p(dst), x.shape[0], x.numel() // x.shape[0]
    L.check(L.load().ssde_perturb(C.byref(a), _stream()), "ssde_perturb")
    return dst


def sumsq_flat(x):
    partial, out = torch.empty(1024, device=x.device), torch.empty(1, device=x.device)
    a = L.SumsqFlatArgs()
    a.x, a.numel, a.partial, a.out = _p(x), x.numel(), _p(partial), _p(out)
    L.check(L.load().ssde_sumsq_flat(C.byref(a), _stream()), "ssde_sumsq_flat")
    return out


def adam_clip_ema(p, g, m, v, ema, hyper, gnorm_sq=None):
    a = L.AdamArgs()
    a.p, a.g, a.m, a.v, a.ema, a.numel, a.hyper, a.gnorm_sq = _p(p), _p(g), _p(m), _p(v), _p(ema), p.numel(), _p(hyper), _p(gnorm_sq)
    L.check(L.load().ssde_adam_clip_ema(C.byref(a), _stream()), "ssde_adam_clip_ema")
