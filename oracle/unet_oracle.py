"""CPU oracle for the NCSN++ / DDPM++ U-Net forward (TEST INFRASTRUCTURE ONLY).

A functional fp32 restatement of the reference's forward pass, driven by a plain
`state_dict` (the reference's key names) and a config object.  Nothing in the
product package imports this file; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg do, as the checker.

Pinning: the reference ships no tests or golden vectors, so this
restatement is pinned against the reference implementation itself, imported from
/root/reference in the build container by oracle/gen_golden.py; the resulting
vectors live in tests/golden/ and tests/test_oracle_golden.py re-checks them.

Every function cites the reference lines it follows (paths relative to
yang-song/score_sde_pytorch).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _cfg(obj, name, default=None):
    try:
        return getattr(obj, name)
    except (AttributeError, KeyError):
        return default


# --------------------------------------------------------------------------- FIR
def setup_fir_kernel(k):
    """models/up_or_down_sampling.py:181-188 (_setup_kernel): outer product, normalised."""
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k = k / np.sum(k)
    return k


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """op/upfirdn2d.py:159-200 (upfirdn2d_native) with square up/down/pad, NCHW in/out."""
    n, c, in_h, in_w = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    v = x.reshape(n * c, 1, in_h, in_w)
    # zero insertion (lines 168-170)
    if up > 1:
        z = v.new_zeros(n * c, 1, in_h * up, in_w * up)
        z[:, :, ::up, ::up] = v
        v = z
    # padding / cropping (lines 172-180)
    v = F.pad(v, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    v = v[:, :, max(-p0, 0): v.shape[2] - max(-p1, 0), max(-p0, 0): v.shape[3] - max(-p1, 0)]
    # correlation with the flipped kernel (lines 186-187)
    w = torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw).to(v.dtype)
    v = F.conv2d(v, w)
    # decimation (line 195)
    v = v[:, :, ::down, ::down]
    return v.reshape(n, c, v.shape[2], v.shape[3])


def upsample_2d(x, k, factor=2):
    """models/up_or_down_sampling.py:195-224."""
    kk = setup_fir_kernel(k) * (factor ** 2)
    p = kk.shape[0] - factor
    return upfirdn2d(x, torch.tensor(kk), up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))


def downsample_2d(x, k, factor=2):
    """models/up_or_down_sampling.py:227-257."""
    kk = setup_fir_kernel(k)
    p = kk.shape[0] - factor
    return upfirdn2d(x, torch.tensor(kk), down=factor, pad=((p + 1) // 2, p // 2))


def conv_downsample_2d(x, w, k, factor=2):
    """models/up_or_down_sampling.py:144-178: FIR with pad (p+1)//2, p//2 then stride-2 VALID conv."""
    kk = setup_fir_kernel(k)
    conv = w.shape[2]
    p = (kk.shape[0] - factor) + (conv - 1)
    x = upfirdn2d(x, torch.tensor(kk), pad=((p + 1) // 2, p // 2))
    return F.conv2d(x, w, stride=factor, padding=0)


def naive_upsample_2d(x, factor=2):
    """models/up_or_down_sampling.py:59-63."""
    return x.repeat_interleave(factor, dim=2).repeat_interleave(factor, dim=3)


def naive_downsample_2d(x, factor=2):
    """models/up_or_down_sampling.py:66-69."""
    n, c, h, w = x.shape
    return x.reshape(n, c, h // factor, factor, w // factor, factor).mean(dim=(3, 5))


# --------------------------------------------------------------------------- layers
def group_norm(x, sd, prefix):
    """nn.GroupNorm(num_groups=min(C // 4, 32), eps=1e-6): models/layerspp.py:67,219,231."""
    c = x.shape[1]
    return F.group_norm(x, min(c // 4, 32), sd[prefix + ".weight"], sd[prefix + ".bias"], eps=1e-6)


def nin(x, sd, prefix):
    """models/layers.py:546-555: y[b,h,w,:] = x[b,h,w,:] @ W + b."""
    y = torch.einsum("bchw,cd->bdhw", x, sd[prefix + ".W"])
    return y + sd[prefix + ".b"][None, :, None, None]


def conv(x, sd, prefix, stride=1, padding=0):
    return F.conv2d(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"), stride=stride, padding=padding)


def timestep_embedding(t, dim, max_positions=10000):
    """models/layers.py:515-529."""
    half = dim // 2
    e = math.log(max_positions) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -e)
    arg = t.float()[:, None] * freqs[None, :]
    out = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if dim % 2 == 1:
        out = F.pad(out, (0, 1))
    return out


def attn_block(x, sd, p, skip_rescale):
    """models/layerspp.py:75-91 (AttnBlockpp.forward)."""
    b, c, h, w = x.shape
    hn = group_norm(x, sd, p + ".GroupNorm_0")
    q, k, v = nin(hn, sd, p + ".NIN_0"), nin(hn, sd, p + ".NIN_1"), nin(hn, sd, p + ".NIN_2")
    s = torch.einsum("bchw,bcij->bhwij", q, k) * (int(c) ** (-0.5))
    s = F.softmax(s.reshape(b, h, w, h * w), dim=-1).reshape(b, h, w, h, w)
    o = torch.einsum("bhwij,bcij->bchw", s, v)
    o = nin(o, sd, p + ".NIN_3")
    return (x + o) / np.sqrt(2.) if skip_rescale else x + o


def resblock_biggan(x, temb, sd, p, up, down, fir, fir_kernel, skip_rescale, dropout_mask=None):
    """models/layerspp.py:242-274 (ResnetBlockBigGANpp.forward), eval mode (dropout = identity)."""
    h = F.silu(group_norm(x, sd, p + ".GroupNorm_0"))
    if up:
        if fir:
            h, x = upsample_2d(h, fir_kernel), upsample_2d(x, fir_kernel)
        else:
            h, x = naive_upsample_2d(h), naive_upsample_2d(x)
    elif down:
        if fir:
            h, x = downsample_2d(h, fir_kernel), downsample_2d(x, fir_kernel)
        else:
            h, x = naive_downsample_2d(h), naive_downsample_2d(x)
    h = conv(h, sd, p + ".Conv_0", padding=1)
    if temb is not None:
        h = h + F.linear(F.silu(temb), sd[p + ".Dense_0.weight"], sd[p + ".Dense_0.bias"])[:, :, None, None]
    h = F.silu(group_norm(h, sd, p + ".GroupNorm_1"))
    if dropout_mask is not None:
        h = h * dropout_mask
    h = conv(h, sd, p + ".Conv_1", padding=1)
    if (p + ".Conv_2.weight") in sd:
        x = conv(x, sd, p + ".Conv_2")
    return (x + h) / np.sqrt(2.) if skip_rescale else x + h


def fir_conv_down(x, sd, p, fir_kernel):
    """layerspp.Downsample(with_conv=True, fir=True).forward -> up_or_down_sampling.Conv2d(down=True)
    (models/layerspp.py:149-163, models/up_or_down_sampling.py:45-56)."""
    y = conv_downsample_2d(x, sd[p + ".Conv2d_0.weight"], fir_kernel)
    return y + sd[p + ".Conv2d_0.bias"].reshape(1, -1, 1, 1)


# --------------------------------------------------------------------------- U-Net
def ncsnpp_forward(config, sd, x, time_cond):
    """models/ncsnpp.py:232-381 (NCSNpp.forward) for resblock_type='biggan'.

    `sd` maps the reference's state-dict keys (without any 'module.' prefix) to CPU fp32 tensors.
    """
    m = config.model
    nf = m.nf
    ch_mult = tuple(m.ch_mult)
    num_res_blocks = m.num_res_blocks
    attn_resolutions = tuple(m.attn_resolutions)
    num_resolutions = len(ch_mult)
    fir = m.fir
    fir_kernel = list(m.fir_kernel)
    skip_rescale = m.skip_rescale
    progressive = m.progressive.lower()
    progressive_input = m.progressive_input.lower()
    embedding_type = m.embedding_type.lower()
    combine_method = m.progressive_combine.lower()
    assert m.resblock_type.lower() == "biggan", "oracle covers the BigGAN residual block only (all ncsnpp configs)"
    assert m.nonlinearity.lower() == "swish"
    act = F.silu
    P = lambda i: "all_modules.%d" % i  # noqa: E731
    i = 0

    # ncsnpp.py:236-249
    if embedding_type == "fourier":
        used_sigmas = time_cond
        x_proj = torch.log(used_sigmas)[:, None] * sd[P(i) + ".W"][None, :] * 2 * np.pi   # layerspp.py:40
        temb = torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)
        i += 1
    elif embedding_type == "positional":
        timesteps = time_cond
        used_sigmas = sd["sigmas"][time_cond.long()]
        temb = timestep_embedding(timesteps, nf)
    else:
        raise ValueError(embedding_type)
    # ncsnpp.py:251-257
    if m.conditional:
        temb = F.linear(temb, sd[P(i) + ".weight"], sd[P(i) + ".bias"]); i += 1
        temb = F.linear(act(temb), sd[P(i) + ".weight"], sd[P(i) + ".bias"]); i += 1
    else:
        temb = None
    # ncsnpp.py:259-261
    if not config.data.centered:
        x = 2 * x - 1.

    def res(idx, h, up=False, down=False):
        return resblock_biggan(h, temb, sd, P(idx), up, down, fir, fir_kernel, skip_rescale)

    def is_attn(h):
        return h.shape[-1] in attn_resolutions

    # ---- down path, ncsnpp.py:263-303
    input_pyramid = x if progressive_input != "none" else None
    hs = [conv(x, sd, P(i), padding=1)]; i += 1
    for i_level in range(num_resolutions):
        for _ in range(num_res_blocks):
            h = res(i, hs[-1]); i += 1
            if is_attn(h):
                h = attn_block(h, sd, P(i), skip_rescale); i += 1
            hs.append(h)
        if i_level != num_resolutions - 1:
            h = res(i, hs[-1], down=True); i += 1
            if progressive_input == "input_skip":
                # self.pyramid_downsample = Downsample(fir, with_conv=False): ncsnpp.py:110,290
                input_pyramid = downsample_2d(input_pyramid, fir_kernel) if fir else F.avg_pool2d(input_pyramid, 2, 2)
                hc = conv(input_pyramid, sd, P(i) + ".Conv_0"); i += 1       # Combine: layerspp.py:52-59
                h = torch.cat([hc, h], dim=1) if combine_method == "cat" else hc + h
            elif progressive_input == "residual":
                input_pyramid = fir_conv_down(input_pyramid, sd, P(i), fir_kernel) if fir else \
                    conv(F.pad(input_pyramid, (0, 1, 0, 1)), sd, P(i) + ".Conv_0", stride=2)
                i += 1
                input_pyramid = (input_pyramid + h) / np.sqrt(2.) if skip_rescale else input_pyramid + h
                h = input_pyramid
            hs.append(h)

    # ---- bottleneck, ncsnpp.py:305-311
    h = hs[-1]
    h = res(i, h); i += 1
    h = attn_block(h, sd, P(i), skip_rescale); i += 1
    h = res(i, h); i += 1

    # ---- up path, ncsnpp.py:313-364
    pyramid = None
    for i_level in reversed(range(num_resolutions)):
        for _ in range(num_res_blocks + 1):
            h = res(i, torch.cat([h, hs.pop()], dim=1)); i += 1
        if is_attn(h):
            h = attn_block(h, sd, P(i), skip_rescale); i += 1
        if progressive != "none":
            if progressive == "output_skip":
                if i_level != num_resolutions - 1:
                    # self.pyramid_upsample = Upsample(fir, with_conv=False): ncsnpp.py:101,341
                    pyramid = upsample_2d(pyramid, fir_kernel) if fir else \
                        F.interpolate(pyramid, scale_factor=2, mode="nearest")
                ph = act(group_norm(h, sd, P(i))); i += 1
                ph = conv(ph, sd, P(i), padding=1); i += 1
                pyramid = ph if i_level == num_resolutions - 1 else pyramid + ph
            else:
                raise NotImplementedError("progressive='residual' reaches the broken upsample_conv_2d (SURVEY F7)")
        if i_level != 0:
            h = res(i, h, up=True); i += 1
    assert not hs

    # ---- head, ncsnpp.py:368-381
    if progressive == "output_skip":
        h = pyramid
    else:
        h = act(group_norm(h, sd, P(i))); i += 1
        h = conv(h, sd, P(i), padding=1); i += 1
    n_modules = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("all_modules."))
    assert i == n_modules, (i, n_modules)
    if m.scale_by_sigma:
        h = h / used_sigmas.reshape(-1, 1, 1, 1)
    return h
