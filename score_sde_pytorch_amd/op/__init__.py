"""Drop-in for the reference's `op` package (op/__init__.py:1-2): `upfirdn2d` and `fused_leaky_relu` /
`FusedLeakyReLU`, same signatures, NCHW tensors, differentiable -- backed by libssde_hip.so instead of the
JIT-compiled CUDA extensions (op/upfirdn2d.py:10-16, op/fused_act.py:11-17).  No model in the reference calls
`fused_leaky_relu`; it is provided for API parity.  CPU tensors are refused (the reference falls
back to `upfirdn2d_native` there; this package is the MI355X path only).
"""
import numpy as np
import torch
from torch import nn

from .. import hipops


def _flip_taps(kernel):
    return torch.flip(kernel, [0, 1])


class UpFirDn2d(torch.autograd.Function):
    """op/upfirdn2d.py:88-142: forward = ssde_upfirdn2d; backward = the same kernel with up/down swapped, the
    flipped taps and the gradient pads of :111-116 (and again the forward op for the double backward)."""

    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        n, c, h, w = input.shape
        kh, kw = kernel.shape
        p0, p1 = pad
        x = hipops.to_nhwc(input.contiguous().float(), c_pad=(c + 3) // 4 * 4)
        y = hipops.upfirdn2d_nhwc(x, kernel, up=up, down=down, pad=pad)
        out = hipops.to_nchw(y, c=c)
        g_pad = (kw - p0 - 1, w * up - out.shape[3] * down + p0 - up + 1)
        if min(g_pad) < 0:
            raise NotImplementedError("upfirdn2d backward with negative gradient pads")
        ctx.cfg = (up, down, pad, g_pad)
        ctx.save_for_backward(kernel)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (kernel,) = ctx.saved_tensors
        up, down, pad, g_pad = ctx.cfg
        grad_input = UpFirDn2d.apply(grad_output, _flip_taps(kernel), down, up, g_pad)
        return grad_input, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """op/upfirdn2d.py:145-156."""
    if input.device.type == "cpu":
        raise RuntimeError("score_sde_pytorch_amd.op.upfirdn2d runs on the MI355X only (no CPU fallback)")
    if kernel.dim() != 2 or max(kernel.shape) > 4:
        raise ValueError("upfirdn2d: kernel must be 2-D, at most 4x4")
    return UpFirDn2d.apply(input, kernel.to(input.device, torch.float32), up, down, (pad[0], pad[1]))


class FusedLeakyReLUFunction(torch.autograd.Function):
    """op/fused_act.py:54-74 (forward) and :20-51 (backward), act=3 (leaky relu), bias over dim 1."""

    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        inner = int(np.prod(input.shape[2:])) if input.dim() > 2 else 1
        out = hipops.fused_bias_act(input.float(), bias.float(), channels=input.shape[1], inner=inner, act=3,
                                    alpha=negative_slope, scale=scale)
        ctx.save_for_backward(out)
        ctx.cfg = (negative_slope, scale, input.shape[1], inner)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        grad_input, grad_bias = FusedLeakyReLUBackward.apply(grad_output, out, ctx.cfg)
        return grad_input, grad_bias, None, None


class FusedLeakyReLUBackward(torch.autograd.Function):
    """The gradient of fused_leaky_relu as a differentiable op of its own (op/fused_act.py:20-51: gradient penalties
    differentiate through it).  Both directions are the same device kernel in gradient mode: the slope is chosen by the sign
    of the saved forward output, so the map grad_output -> grad_input is linear and its adjoint is itself (plus the bias
    broadcast, whose adjoint is the per-channel sum)."""

    @staticmethod
    def forward(ctx, grad_output, out, cfg):
        slope, scale, channels, inner = cfg
        ctx.save_for_backward(out)
        ctx.cfg = cfg
        grad_input = hipops.fused_bias_act(grad_output.contiguous().float(), None, act=3, alpha=slope, scale=scale, grad=1, ref=out)
        n = grad_input.shape[0]
        # grad_bias = grad_input summed over every dim but the channel one (op/fused_act.py:33-38): column sums in NHWC
        g = hipops.to_nhwc(grad_input.reshape(n, channels, inner, 1), c_pad=(channels + 3) // 4 * 4)
        total = torch.zeros(channels, device=g.device)
        hipops.colsum(g, c=channels, total=total)
        return grad_input, total

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        (out,) = ctx.saved_tensors
        slope, scale, channels, inner = ctx.cfg
        if gradgrad_input is None:
            gradgrad_input = torch.zeros_like(out)
        bias = gradgrad_bias.contiguous().float() if gradgrad_bias is not None else None
        gradgrad_out = hipops.fused_bias_act(gradgrad_input.contiguous().float(), bias, channels=channels, inner=inner, act=3,
                                             alpha=slope, scale=scale, grad=1, ref=out)
        return gradgrad_out, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    """op/fused_act.py:86-97."""
    if input.device.type == "cpu":
        raise RuntimeError("score_sde_pytorch_amd.op.fused_leaky_relu runs on the MI355X only (no CPU fallback)")
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    """op/fused_act.py:77-83."""

    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
