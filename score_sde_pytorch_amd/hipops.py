"""Single-op Python entry points over the C ABI (one HIP launch each, current torch stream).

Tensors are NHWC float32 CUDA tensors unless stated.  These wrappers exist for the op-level
parity tests and for `score_sde_pytorch_amd.op` (the reference's `op` package surface); the
U-Net itself runs through `engine.UNetEngine` programs, not through here.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .engine import pack_conv_weight, pack_matrix


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
           chan_add=None, resid=None, scale=1.0, tile=L.TILE_AUTO, out_hw=None):
    """k x k (k=3) conv of NHWC `x` (weight OIHW) plus optional 1x1 conv of `aux` (weight [Cout, Cin])."""
    _need_cuda(x, aux, resid)
    a = L.ConvArgs()
    keep = []
    if x is not None:
        n, h_in, w_in = x.shape[0], x.shape[1], x.shape[2]
        c_out = weight.shape[0]
        h_out = (h_in + 2 * pad - 3) // stride + 1
        w_out = (w_in + 2 * pad - 3) // stride + 1
        _fill_src(a.main, x, x2, pro, gn)
        wp = pack_conv_weight(weight.to(x.device)); keep.append(wp)
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
    L.check(L.load().ssde_conv2d(C.byref(a), _stream()), "ssde_conv2d")
    return dst


def upfirdn2d_nhwc(x, kernel, up=1, down=1, pad=(0, 0), pro=L.PRO_NONE, gn=None):
    _need_cuda(x)
    n, h, w, c = x.shape
    kh, kw = kernel.shape
    h_out = (h * up + pad[0] + pad[1] - kh) // down + 1
    w_out = (w * up + pad[0] + pad[1] - kw) // down + 1
    dst = torch.empty(n, h_out, w_out, c, device=x.device)
    a = L.UpfirdnArgs()
    _fill_src(a.src, x, None, pro, gn)
    a.n, a.h_in, a.w_in, a.c, a.h_out, a.w_out = n, h, w, c, h_out, w_out
    a.up, a.down, a.pad0, a.pad1, a.kh, a.kw = up, down, pad[0], pad[1], kh, kw
    for i, v in enumerate(np.asarray(kernel.detach().cpu(), dtype=np.float32).reshape(-1).tolist()):
        a.k[i] = v
    a.dst = _p(dst)
    L.check(L.load().ssde_upfirdn2d(C.byref(a), _stream()), "ssde_upfirdn2d")
    return dst


def attention(qkv, channels):
    """qkv [N, L, 3C] -> softmax(q k^T / sqrt(C)) v  [N, L, C]."""
    _need_cuda(qkv)
    n, l = qkv.shape[0], qkv.shape[1]
    dst = torch.empty(n, l, channels, device=qkv.device)
    a = L.AttnArgs()
    a.qkv, a.dst, a.n, a.l, a.c, a.scale = _p(qkv), _p(dst), n, l, channels, float(int(channels) ** (-0.5))
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


def fused_bias_act(x, bias=None, channels=1, inner=1, act=1, alpha=0.2, scale=1.0):
    _need_cuda(x)
    x = x.contiguous()
    dst = torch.empty_like(x)
    a = L.BiasActArgs()
    a.src, a.bias, a.dst, a.numel, a.channels, a.inner, a.act, a.alpha, a.scale = \
        _p(x), _p(bias), _p(dst), x.numel(), channels, inner, act, alpha, scale
    L.check(L.load().ssde_fused_bias_act(C.byref(a), _stream()), "ssde_fused_bias_act")
    return dst


def randn(numel, seed, device, step_ptr=None, stream_id=0):
    dst = torch.empty(numel, device=device)
    a = L.RandnArgs()
    a.dst, a.numel, a.seed, a.step_ptr, a.stream_id = _p(dst), numel, seed, _p(step_ptr), stream_id
    L.check(L.load().ssde_randn(C.byref(a), _stream()), "ssde_randn")
    return dst
