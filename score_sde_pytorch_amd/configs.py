"""Experiment configurations for the BASELINE workloads.

The reference keeps one `ml_collections.ConfigDict` python file per experiment
(configs/*.py).  Those files run unchanged against this package (any object with the
same attribute tree works -- see `from_reference`).  Because `ml_collections` is not
a dependency here, the BASELINE configs are also available as presets built from one
shared default table plus per-experiment overrides:

    cfg = get_config("ve/cifar10_ncsnpp_continuous")

Values follow configs/default_cifar10_configs.py, configs/default_lsun_configs.py and
the named experiment files of the reference.
"""
import copy

import torch


class ConfigDict(dict):
    """Attribute-style nested dict (the subset of ml_collections.ConfigDict the hot path reads)."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _tree(d):
    return ConfigDict({k: _tree(v) if isinstance(v, dict) else v for k, v in d.items()})


_COMMON = {
    "training": dict(batch_size=128, n_iters=1300001, snapshot_freq=50000, log_freq=50, eval_freq=100,
                     snapshot_freq_for_preemption=10000, snapshot_sampling=True, likelihood_weighting=False,
                     continuous=True, reduce_mean=False, sde="vesde"),
    "sampling": dict(n_steps_each=1, noise_removal=True, probability_flow=False, snr=0.16, method="pc",
                     predictor="reverse_diffusion", corrector="langevin"),
    "eval": dict(begin_ckpt=9, end_ckpt=26, batch_size=1024, enable_sampling=False, num_samples=50000,
                 enable_loss=True, enable_bpd=False, bpd_dataset="test"),
    "data": dict(dataset="CIFAR10", image_size=32, random_flip=True, centered=False,
                 uniform_dequantization=False, num_channels=3),
    "model": dict(name="ncsnpp", sigma_min=0.01, sigma_max=50, num_scales=1000, beta_min=0.1, beta_max=20.0,
                  dropout=0.1, embedding_type="fourier", scale_by_sigma=True, ema_rate=0.999,
                  normalization="GroupNorm", nonlinearity="swish", nf=128, ch_mult=(1, 2, 2, 2), num_res_blocks=4,
                  attn_resolutions=(16,), resamp_with_conv=True, conditional=True, fir=True,
                  fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type="biggan", progressive="none",
                  progressive_input="residual", progressive_combine="sum", attention_type="ddpm",
                  init_scale=0.0, fourier_scale=16, conv_size=3),
    "optim": dict(weight_decay=0, optimizer="Adam", lr=2e-4, beta1=0.9, eps=1e-8, warmup=5000, grad_clip=1.0),
    "seed": 42,
}

# 256-px defaults differ from the CIFAR-10 ones in these entries (configs/default_lsun_configs.py)
_LSUN = {
    "training": dict(batch_size=64, n_iters=2400001, snapshot_freq_for_preemption=5000),
    "sampling": dict(snr=0.075),
    "eval": dict(begin_ckpt=50, end_ckpt=96, batch_size=512, enable_sampling=True),
    "data": dict(dataset="LSUN", image_size=256),
    "model": dict(sigma_max=378, num_scales=2000, dropout=0.0),
}

_PRESETS = {
    # configs/ve/cifar10_ncsnpp_continuous.py
    "ve/cifar10_ncsnpp_continuous": [],
    # configs/ve/cifar10_ncsnpp_deep_continuous.py
    "ve/cifar10_ncsnpp_deep_continuous": [{"training": dict(n_iters=950001), "model": dict(num_res_blocks=8)}],
    # configs/subvp/cifar10_ddpmpp_continuous.py
    "subvp/cifar10_ddpmpp_continuous": [{
        "training": dict(sde="subvpsde", reduce_mean=True),
        "sampling": dict(predictor="euler_maruyama", corrector="none"),
        "data": dict(centered=True),
        "model": dict(scale_by_sigma=False, ema_rate=0.9999, fir=False, progressive_input="none",
                      embedding_type="positional"),
    }],
    # configs/vp/cifar10_ddpmpp_continuous.py
    "vp/cifar10_ddpmpp_continuous": [{
        "training": dict(sde="vpsde", reduce_mean=True),
        "sampling": dict(predictor="euler_maruyama", corrector="none"),
        "data": dict(centered=True),
        "model": dict(scale_by_sigma=False, ema_rate=0.9999, fir=False, progressive_input="none",
                      embedding_type="positional"),
    }],
    # configs/ve/ffhq_256_ncsnpp_continuous.py
    "ve/ffhq_256_ncsnpp_continuous": [_LSUN, {
        "data": dict(dataset="FFHQ", image_size=256),
        "model": dict(sigma_max=348, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, progressive="output_skip",
                      progressive_input="input_skip"),
    }],
}


def get_config(name, **model_overrides):
    """Return a fresh ConfigDict for a named BASELINE experiment; keyword args override `config.model`."""
    if name not in _PRESETS:
        raise KeyError("unknown config %r; known: %s" % (name, sorted(_PRESETS)))
    table = copy.deepcopy(_COMMON)
    for layer in _PRESETS[name]:
        for section, values in layer.items():
            table[section].update(copy.deepcopy(values))
    table["model"].update(model_overrides)
    cfg = _tree(table)
    cfg.device = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")
    return cfg


def from_reference(ref_config):
    """Deep-convert a reference `ml_collections.ConfigDict` (or any nested mapping) into ConfigDict."""
    def conv(v):
        if hasattr(v, "items"):
            return ConfigDict({k: conv(x) for k, x in v.items()})
        return v
    return conv(ref_config)
