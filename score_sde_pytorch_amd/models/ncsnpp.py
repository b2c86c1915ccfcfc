"""NCSN++ / DDPM++ score network, MI355X-native.

Drop-in for the reference's `models.ncsnpp.NCSNpp` (models/ncsnpp.py:34-381): same
constructor (`NCSNpp(config)`), same `forward(x[B,C,H,W], time_cond[B])`, same
`all_modules` ModuleList ordering and parameter names, so reference checkpoints
(`state_dict` keys `all_modules.<i>.<Leaf>...`, buffer `sigmas`) load unchanged.

The modules below are *parameter containers*: they own tensors with the reference's
names, shapes and initial distributions, and describe themselves to the engine
(`score_sde_pytorch_amd.engine`), which lowers the whole forward to one program of
hand-written HIP kernels (libssde_hip.so).  There is no PyTorch-eager forward in this
package: running on a CPU tensor, or without the HIP library, raises.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import utils


# --------------------------------------------------------------------------- init
def _fan_avg_uniform_(tensor, scale=1.0, in_axis=1, out_axis=0):
    """DDPM's default initialiser: variance_scaling(scale, 'fan_avg', 'uniform')
    (models/layers.py:54-91); scale 0 means 1e-10."""
    scale = 1e-10 if scale == 0 else scale
    shape = tensor.shape
    receptive = float(np.prod(shape)) / shape[in_axis] / shape[out_axis]
    fan_in, fan_out = shape[in_axis] * receptive, shape[out_axis] * receptive
    bound = math.sqrt(3.0 * scale / ((fan_in + fan_out) / 2.0))
    with torch.no_grad():
        tensor.copy_((torch.rand(*shape, dtype=tensor.dtype) * 2.0 - 1.0) * bound)
    return tensor


def _conv(in_ch, out_ch, k, init_scale=1.0, stride=1, padding=0):
    """ddpm_conv3x3 / ddpm_conv1x1 (models/layers.py:100-124)."""
    m = nn.Conv2d(in_ch, out_ch, kernel_size=k, stride=stride, padding=padding, bias=True)
    _fan_avg_uniform_(m.weight.data, init_scale)
    nn.init.zeros_(m.bias)
    return m


def _dense(in_dim, out_dim):
    m = nn.Linear(in_dim, out_dim)
    _fan_avg_uniform_(m.weight.data, 1.0)
    nn.init.zeros_(m.bias)
    return m


def _group_norm(ch):
    return nn.GroupNorm(num_groups=min(ch // 4, 32), num_channels=ch, eps=1e-6)


# --------------------------------------------------------------------------- containers
class GaussianFourierProjection(nn.Module):
    """Frozen random frequencies W ~ N(0, scale^2) (models/layerspp.py:32-41)."""
    kind = "fourier"

    def __init__(self, embedding_size=256, scale=1.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)


class NIN(nn.Module):
    """Per-pixel linear map y = x W + b with W [in, out] (models/layers.py:546-555)."""

    def __init__(self, in_dim, num_units, init_scale=0.1):
        super().__init__()
        self.W = nn.Parameter(_fan_avg_uniform_(torch.empty(in_dim, num_units), init_scale))
        self.b = nn.Parameter(torch.zeros(num_units))


class AttnBlockpp(nn.Module):
    """GroupNorm -> q,k,v NIN -> softmax(q k / sqrt(C)) v -> NIN -> skip (models/layerspp.py:62-91)."""
    kind = "attn"

    def __init__(self, channels, skip_rescale=False, init_scale=0.0):
        super().__init__()
        self.GroupNorm_0 = _group_norm(channels)
        self.NIN_0 = NIN(channels, channels)
        self.NIN_1 = NIN(channels, channels)
        self.NIN_2 = NIN(channels, channels)
        self.NIN_3 = NIN(channels, channels, init_scale=init_scale)
        self.channels = channels
        self.skip_rescale = skip_rescale


class ResnetBlockBigGANpp(nn.Module):
    """BigGAN residual block with optional FIR / box 2x resampling (models/layerspp.py:212-274)."""
    kind = "res"

    def __init__(self, in_ch, out_ch=None, temb_dim=None, up=False, down=False, dropout=0.1, fir=False,
                 fir_kernel=(1, 3, 3, 1), skip_rescale=True, init_scale=0.0):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        self.GroupNorm_0 = _group_norm(in_ch)
        self.Conv_0 = _conv(in_ch, out_ch, 3, padding=1)
        if temb_dim is not None:
            self.Dense_0 = _dense(temb_dim, out_ch)
        self.GroupNorm_1 = _group_norm(out_ch)
        self.Dropout_0 = nn.Dropout(dropout)
        self.Conv_1 = _conv(out_ch, out_ch, 3, init_scale=init_scale, padding=1)
        if in_ch != out_ch or up or down:
            self.Conv_2 = _conv(in_ch, out_ch, 1)
        self.in_ch, self.out_ch = in_ch, out_ch
        self.up, self.down, self.fir, self.fir_kernel = up, down, fir, tuple(fir_kernel)
        self.skip_rescale = skip_rescale
        self.dropout = dropout


class FirConv2d(nn.Module):
    """3x3 conv fused with FIR down-sampling (up_or_down_sampling.Conv2d(down=True),
    models/up_or_down_sampling.py:23-56,144-178).  Only `weight`/`bias` live here."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.weight = nn.Parameter(_fan_avg_uniform_(torch.empty(out_ch, in_ch, 3, 3), 1.0))
        self.bias = nn.Parameter(torch.zeros(out_ch))


class Downsample(nn.Module):
    """models/layerspp.py:129-163."""
    kind = "down"

    def __init__(self, in_ch=None, out_ch=None, with_conv=False, fir=False, fir_kernel=(1, 3, 3, 1)):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        if with_conv:
            if fir:
                self.Conv2d_0 = FirConv2d(in_ch, out_ch)
            else:
                self.Conv_0 = _conv(in_ch, out_ch, 3, stride=2, padding=0)
        self.in_ch, self.out_ch, self.with_conv, self.fir, self.fir_kernel = in_ch, out_ch, with_conv, fir, tuple(fir_kernel)


class Upsample(nn.Module):
    """models/layerspp.py:94-126 (the with_conv+fir variant reaches the broken upsample_conv_2d)."""
    kind = "up"

    def __init__(self, in_ch=None, out_ch=None, with_conv=False, fir=False, fir_kernel=(1, 3, 3, 1)):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        if with_conv:
            if fir:
                raise NotImplementedError("Upsample(with_conv=True, fir=True) is dead code in the reference "
                                          "(upsample_conv_2d raises, models/up_or_down_sampling.py:126)")
            self.Conv_0 = _conv(in_ch, out_ch, 3, padding=1)
        self.in_ch, self.out_ch, self.with_conv, self.fir, self.fir_kernel = in_ch, out_ch, with_conv, fir, tuple(fir_kernel)


class Combine(nn.Module):
    """1x1 conv on the image pyramid, then sum / cat with the feature map (models/layerspp.py:44-59)."""
    kind = "combine"

    def __init__(self, dim1, dim2, method="cat"):
        super().__init__()
        self.Conv_0 = _conv(dim1, dim2, 1)
        self.method = method


# --------------------------------------------------------------------------- the network
@utils.register_model(name="ncsnpp")
class NCSNpp(nn.Module):
    """NCSN++ (fir=True, progressive input/output) and DDPM++ (fir=False) score network."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        m = config.model
        if m.nonlinearity.lower() != "swish":
            raise NotImplementedError("the HIP path fuses SiLU; nonlinearity=%r is not built" % m.nonlinearity)
        if m.resblock_type.lower() != "biggan":
            raise NotImplementedError("resblock_type=%r: every shipped ncsnpp config uses 'biggan'" % m.resblock_type)
        self.register_buffer("sigmas", torch.tensor(utils.get_sigmas(config)))
        self.nf = nf = m.nf
        ch_mult = tuple(m.ch_mult)
        self.num_res_blocks = nrb = m.num_res_blocks
        self.attn_resolutions = tuple(m.attn_resolutions)
        self.num_resolutions = nres = len(ch_mult)
        self.all_resolutions = [config.data.image_size // (2 ** i) for i in range(nres)]
        self.conditional = m.conditional
        self.skip_rescale = m.skip_rescale
        self.resblock_type = "biggan"
        self.progressive = m.progressive.lower()
        self.progressive_input = m.progressive_input.lower()
        self.embedding_type = m.embedding_type.lower()
        self.combine_method = m.progressive_combine.lower()
        self.fir, self.fir_kernel = m.fir, tuple(m.fir_kernel)
        if self.progressive not in ("none", "output_skip", "residual"):
            raise ValueError("progressive=%r" % self.progressive)
        if self.progressive_input not in ("none", "input_skip", "residual"):
            raise ValueError("progressive_input=%r" % self.progressive_input)
        if self.embedding_type not in ("fourier", "positional"):
            raise ValueError("embedding type %s unknown." % self.embedding_type)
        if self.progressive == "residual":
            raise NotImplementedError("progressive='residual' needs Upsample(with_conv, fir) which the reference "
                                      "cannot run either (SURVEY F7)")

        mods = []
        add = mods.append
        if self.embedding_type == "fourier":
            if not config.training.continuous:
                raise AssertionError("Fourier features are only used for continuous training.")
            add(GaussianFourierProjection(embedding_size=nf, scale=m.fourier_scale))
            embed_dim = 2 * nf
        else:
            embed_dim = nf
        if self.conditional:
            add(_dense(embed_dim, nf * 4))
            add(_dense(nf * 4, nf * 4))

        def res_block(in_ch, out_ch=None, up=False, down=False):
            return ResnetBlockBigGANpp(in_ch, out_ch, temb_dim=nf * 4 if self.conditional else None, up=up, down=down,
                                       dropout=m.dropout, fir=self.fir, fir_kernel=self.fir_kernel,
                                       skip_rescale=self.skip_rescale, init_scale=m.init_scale)

        def attn_block(ch):
            return AttnBlockpp(ch, skip_rescale=self.skip_rescale, init_scale=m.init_scale)

        channels = config.data.num_channels
        self.channels = channels
        if self.progressive == "output_skip":
            self.pyramid_upsample = Upsample(fir=self.fir, fir_kernel=self.fir_kernel, with_conv=False)
        if self.progressive_input == "input_skip":
            self.pyramid_downsample = Downsample(fir=self.fir, fir_kernel=self.fir_kernel, with_conv=False)

        # ---- encoder
        pyr_ch = channels
        add(_conv(channels, nf, 3, padding=1))
        skip_chs = [nf]
        cur = nf
        for lvl in range(nres):
            for _ in range(nrb):
                out_ch = nf * ch_mult[lvl]
                add(res_block(cur, out_ch))
                cur = out_ch
                if self.all_resolutions[lvl] in self.attn_resolutions:
                    add(attn_block(cur))
                skip_chs.append(cur)
            if lvl != nres - 1:
                add(res_block(cur, down=True))
                if self.progressive_input == "input_skip":
                    add(Combine(dim1=pyr_ch, dim2=cur, method=self.combine_method))
                    if self.combine_method == "cat":
                        cur *= 2
                elif self.progressive_input == "residual":
                    add(Downsample(in_ch=pyr_ch, out_ch=cur, with_conv=True, fir=self.fir, fir_kernel=self.fir_kernel))
                    pyr_ch = cur
                skip_chs.append(cur)

        # ---- bottleneck
        cur = skip_chs[-1]
        add(res_block(cur))
        add(attn_block(cur))
        add(res_block(cur))

        # ---- decoder
        for lvl in reversed(range(nres)):
            for _ in range(nrb + 1):
                out_ch = nf * ch_mult[lvl]
                add(res_block(cur + skip_chs.pop(), out_ch))
                cur = out_ch
            if self.all_resolutions[lvl] in self.attn_resolutions:
                add(attn_block(cur))
            if self.progressive == "output_skip":
                add(_group_norm(cur))
                add(_conv(cur, channels, 3, init_scale=m.init_scale, padding=1))
            if lvl != 0:
                add(res_block(cur, up=True))
        assert not skip_chs
        if self.progressive != "output_skip":
            add(_group_norm(cur))
            add(_conv(cur, channels, 3, init_scale=m.init_scale, padding=1))

        self.all_modules = nn.ModuleList(mods)
        self._engines = {}

    def flat_param_groups(self):
        """Parameters the HIP programs use as one concatenated operand (backward.FlatParams stores them back to back)."""
        dense = [m.Dense_0 for m in self.all_modules if getattr(m, "kind", "") == "res" and hasattr(m, "Dense_0")]
        return [[d.weight for d in dense], [d.bias for d in dense]] if dense else []

    # ------------------------------------------------------------------ forward
    def _engine_for(self, x):
        from .. import engine as _engine
        key = (x.device.index, x.shape[0], x.shape[2], x.shape[3])
        eng = self._engines.get(key)
        if eng is None:
            eng = _engine.UNetEngine(self, batch=x.shape[0], height=x.shape[2], width=x.shape[3], device=x.device)
            self._engines[key] = eng
        return eng

    def forward(self, x, time_cond):
        if not x.is_cuda:
            raise RuntimeError("score_sde_pytorch_amd.NCSNpp runs on MI355X through libssde_hip.so only; "
                               "got a %s tensor (there is no CPU fallback)" % x.device.type)
        if x.dtype != torch.float32 or x.dim() != 4:
            raise TypeError("NCSNpp.forward expects a float32 [B, C, H, W] tensor")
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            from .. import autograd as _autograd
            return _autograd.unet_apply(self, x, time_cond)
        return self._engine_for(x).forward(x, time_cond)
