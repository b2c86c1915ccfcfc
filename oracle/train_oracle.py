"""CPU oracle of the training step (TEST INFRASTRUCTURE ONLY).

Restates, over the functional U-Net of oracle/unet_oracle.py and with injected draws:
  * get_sde_loss_fn.loss_fn        losses.py:73-99     (through sampler_oracle.dsm_loss)
  * get_smld_loss_fn.loss_fn       losses.py:111-124
  * get_ddpm_loss_fn.loss_fn       losses.py:135-147
  * get_step_fn.step_fn            losses.py:177-208   (train branch :190-199, eval branch :200-206)
  * optimization_manager           losses.py:41-50     (lr warm-up, global-norm clip, optimizer.step)
  * get_optimizer                  losses.py:26-35     (torch.optim.Adam: installed third-party code, executed, not restated)
  * ExponentialMovingAverage       models/ema.py:32-51 (update), :53-89 (copy_to / store / restore)
Nothing in the product package imports this file.

Pinning: oracle/gen_golden_train.py runs the REFERENCE's own get_step_fn / optimization_manager / get_optimizer /
ExponentialMovingAverage (imported from /root/reference in the build container) on the cases of
tests/_util.TRAIN_CASES, asserts that this restatement reproduces every loss, parameter and shadow parameter of those
runs, and stores the reference's results in tests/golden/train_small.npz; tests/test_oracle_golden.py re-checks
the restatement against the stored vectors.
"""
import numpy as np
import torch

from . import sampler_oracle, unet_oracle


def _b(v):
    return v[:, None, None, None]


def loss_value(cfg, sd, sde_kind, sde_kwargs, continuous, reduce_mean, likelihood_weighting, batch, u, labels, z, eps=1e-5):
    """The scalar loss of one batch.  `u` is the uniform draw behind t (losses.py:84), `labels` the integer noise levels
    of the discrete losses (losses.py:116,136)."""
    if continuous:
        sde = sampler_oracle.make_sde(sde_kind, **sde_kwargs)
        t = u * (sde.T - eps) + eps                                                                   # losses.py:84
        return sampler_oracle.dsm_loss(cfg, sd, sde_kind, sde_kwargs, batch, t, z, reduce_mean=reduce_mean,
                                       likelihood_weighting=likelihood_weighting)
    assert not likelihood_weighting                                                                   # losses.py:170
    red = (lambda v: torch.mean(v, dim=-1)) if reduce_mean else (lambda v: 0.5 * torch.sum(v, dim=-1))
    sde = sampler_oracle.make_sde(sde_kind, **sde_kwargs)
    n = batch.shape[0]
    if sde_kind == "vesde":                                                                           # losses.py:111-124
        sigmas = torch.flip(sde.discrete_sigmas, dims=(0,))[labels]
        noise = z * _b(sigmas)
        score = unet_oracle.ncsnpp_forward(cfg, sd, noise + batch, labels)
        target = -noise / _b(sigmas ** 2)
        return torch.mean(red(torch.square(score - target).reshape(n, -1)) * sigmas ** 2)
    if sde_kind == "vpsde":                                                                           # losses.py:135-147
        alphas_cumprod = torch.cumprod(sde.alphas, dim=0)                                             # sde_lib.py:127-129
        a, s = torch.sqrt(alphas_cumprod)[labels], torch.sqrt(1. - alphas_cumprod)[labels]
        out = unet_oracle.ncsnpp_forward(cfg, sd, _b(a) * batch + _b(s) * z, labels)
        return torch.mean(red(torch.square(out - z).reshape(n, -1)))
    raise ValueError("Discrete training for %s is not recommended." % sde_kind)                      # losses.py:175


class TrainState:
    """state = {model, optimizer, ema, step} of run_lib.train (run_lib.py:66-70) for the functional U-Net."""

    def __init__(self, cfg, sd, sde_kind, sde_kwargs, continuous=True, reduce_mean=False, likelihood_weighting=False,
                 trainable=None):
        self.cfg, self.sde_kind, self.sde_kwargs = cfg, sde_kind, dict(sde_kwargs)
        self.continuous, self.reduce_mean, self.likelihood_weighting = continuous, reduce_mean, likelihood_weighting
        # the Fourier frequencies are a frozen nn.Parameter (layerspp.py:38), `sigmas` is a buffer
        self.names = trainable if trainable is not None else \
            [k for k, v in sd.items() if v.dtype == torch.float32 and k != "sigmas" and not k.endswith("all_modules.0.W")]
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.params = {k: self.sd[k].clone().requires_grad_() for k in self.names}
        o = cfg.optim
        self.optimizer = torch.optim.Adam([self.params[k] for k in self.names], lr=o.lr, betas=(o.beta1, 0.999), eps=o.eps,
                                          weight_decay=o.weight_decay)                                # losses.py:29-30
        self.decay = cfg.model.ema_rate
        self.num_updates = 0                                                                          # ema.py:27
        self.shadow = [self.params[k].detach().clone() for k in self.names]                           # ema.py:28-29
        self.step = 0

    def _full(self, params):
        full = dict(self.sd)
        full.update(params)
        return full

    def _loss(self, params, batch, u, labels, z):
        return loss_value(self.cfg, self._full(params), self.sde_kind, self.sde_kwargs, self.continuous, self.reduce_mean,
                          self.likelihood_weighting, batch, u, labels, z)

    def train_step(self, batch, u, labels, z):
        o = self.cfg.optim
        self.optimizer.zero_grad()                                                                    # losses.py:194
        loss = self._loss(self.params, batch, u, labels, z)
        loss.backward()                                                                               # :196
        if o.warmup > 0:                                                                              # :44-46
            for g in self.optimizer.param_groups:
                g['lr'] = o.lr * np.minimum(self.step / o.warmup, 1.0)
        if o.grad_clip >= 0:                                                                          # :47-48
            torch.nn.utils.clip_grad_norm_([self.params[k] for k in self.names], max_norm=o.grad_clip)
        self.optimizer.step()                                                                         # :49
        self.step += 1                                                                                # :198
        self.num_updates += 1                                                                         # ema.py:44-47
        decay = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        with torch.no_grad():
            for s, k in zip(self.shadow, self.names):                                                 # ema.py:49-51
                s.sub_((1.0 - decay) * (s - self.params[k]))
        return loss.detach()

    def eval_step(self, batch, u, labels, z):
        """losses.py:200-206: the loss of the EMA parameters; the raw parameters are untouched afterwards."""
        with torch.no_grad():
            return self._loss({k: s for k, s in zip(self.names, self.shadow)}, batch, u, labels, z)
