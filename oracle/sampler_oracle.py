"""CPU oracle for the predictor-corrector sampler and the DSM loss (TEST INFRASTRUCTURE ONLY).

Restates, with injected noise (the reference draws torch.randn_like on the device, which a
Philox/rocRAND stream cannot reproduce -- SURVEY F9):
  * get_pc_sampler.pc_sampler             sampling.py:390-409
  * LangevinCorrector.update_fn           sampling.py:262-282
  * ReverseDiffusionPredictor.update_fn   sampling.py:195-200  (+ RSDE.discretize sde_lib.py:102-107)
  * EulerMaruyamaPredictor.update_fn      sampling.py:181-187  (+ RSDE.sde sde_lib.py:93-100)
  * VESDE / VPSDE / subVPSDE scalar maps  sde_lib.py:112-254
  * get_score_fn                          models/utils.py:129-178
  * get_sde_loss_fn.loss_fn               losses.py:73-99
Pinned against the reference by oracle/gen_golden.py (see unet_oracle.py header).
"""
import numpy as np
import torch

from . import unet_oracle


def _b(v):
    return v[:, None, None, None]


class _VE:
    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000):
        self.sigma_min, self.sigma_max, self.N, self.T = sigma_min, sigma_max, N, 1
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), N))   # sde_lib.py:219

    def sigma(self, t):                                                                              # sde_lib.py:233-236
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def sde(self, x, t):                                                                             # sde_lib.py:226-231
        g = self.sigma(t) * torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min))))
        return torch.zeros_like(x), g

    def marginal_prob(self, x, t):
        return x, self.sigma(t)

    def discretize(self, x, t):                                                                      # sde_lib.py:246-254
        idx = (t * (self.N - 1) / self.T).long()
        sigma = self.discrete_sigmas[idx]
        adj = torch.where(idx == 0, torch.zeros_like(t), self.discrete_sigmas[idx - 1])
        return torch.zeros_like(x), torch.sqrt(sigma ** 2 - adj ** 2)


class _VP:
    def __init__(self, beta_min=0.1, beta_max=20, N=1000, sub=False):
        self.beta_0, self.beta_1, self.N, self.T, self.sub = beta_min, beta_max, N, 1, sub
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)                          # sde_lib.py:125
        self.alphas = 1. - self.discrete_betas
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1. - torch.cumprod(self.alphas, dim=0))            # sde_lib.py:127-129

    def sde(self, x, t):                                                                             # sde_lib.py:135-139, 184-189
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
        drift = -0.5 * _b(beta_t) * x
        if self.sub:
            disc = 1. - torch.exp(-2 * self.beta_0 * t - (self.beta_1 - self.beta_0) * t ** 2)
            return drift, torch.sqrt(beta_t * disc)
        return drift, torch.sqrt(beta_t)

    def marginal_prob(self, x, t):                                                                   # sde_lib.py:141-145, 191-195
        lmc = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        if self.sub:
            return _b(torch.exp(lmc)) * x, 1 - torch.exp(2. * lmc)
        return torch.exp(_b(lmc)) * x, torch.sqrt(1. - torch.exp(2. * lmc))


    def discretize(self, x, t):
        if self.sub:                                                                                 # sde_lib.py:52-69 (Euler-Maruyama default)
            dt = 1 / self.N
            drift, diffusion = self.sde(x, t)
            return drift * dt, diffusion * torch.sqrt(torch.tensor(dt))
        idx = (t * (self.N - 1) / self.T).long()                                                     # sde_lib.py:155-163 (DDPM)
        return _b(torch.sqrt(self.alphas[idx])) * x - x, torch.sqrt(self.discrete_betas[idx])


def make_sde(kind, **kw):
    kind = kind.lower()
    if kind == "vesde":
        return _VE(**kw)
    if kind == "vpsde":
        return _VP(sub=False, **kw)
    if kind == "subvpsde":
        return _VP(sub=True, **kw)
    raise ValueError(kind)


def score_fn(cfg, sd, sde, x, t, continuous=True):
    """models/utils.py:129-178: continuous models, and the discrete (DDPM / SMLD) label conventions."""
    if isinstance(sde, _VP):
        if continuous or sde.sub:                                                                    # :147-153
            labels = t * 999
            out = unet_oracle.ncsnpp_forward(cfg, sd, x, labels)
            std = sde.marginal_prob(torch.zeros_like(x), t)[1]
        else:                                                                                        # :154-158
            labels = t * (sde.N - 1)
            out = unet_oracle.ncsnpp_forward(cfg, sd, x, labels)
            std = sde.sqrt_1m_alphas_cumprod[labels.long()]
        return -out / _b(std)
    if continuous:                                                                                   # :165-166
        labels = sde.marginal_prob(torch.zeros_like(x), t)[1]
    else:                                                                                            # :168-171
        labels = torch.round((sde.T - t) * (sde.N - 1)).long()
    return unet_oracle.ncsnpp_forward(cfg, sd, x, labels)


def pc_sample(cfg, sd, sde_kind, sde_kwargs, x_T, noises, snr, n_steps=1, eps=1e-3, denoise=True,
              predictor="reverse_diffusion", corrector="langevin", max_steps=None, continuous=True,
              probability_flow=False, project=None):
    """sampling.py:390-409 with noise injected: noises[i, 0] feeds the corrector, noises[i, 1] the predictor.
    `project(sde, x, t, z) -> (x, x_mean)`, when given, runs after the corrector (noise noises[i, 2]) and after the
    predictor (noises[i, 3]): the data-consistency step of controllable_generation.py:44-52 / 136-144."""
    sde = make_sde(sde_kind, **sde_kwargs)
    B = x_T.shape[0]
    x = x_T.clone()
    x_mean = x
    timesteps = torch.linspace(sde.T, eps, sde.N)
    x_steps, score_norms = [], []
    steps = sde.N if max_steps is None else max_steps
    with torch.no_grad():
        for i in range(steps):
            t = torch.ones(B) * timesteps[i]
            if corrector == "langevin":                                                   # sampling.py:262-282
                alpha = torch.ones_like(t) if isinstance(sde, _VE) else sde.alphas[(t * (sde.N - 1) / sde.T).long()]
                for _ in range(n_steps):
                    grad = score_fn(cfg, sd, sde, x, t, continuous)
                    z = noises[i, 0]
                    score_norms.append(float(torch.norm(grad.reshape(B, -1), dim=-1).mean()))
                    gnorm = torch.norm(grad.reshape(B, -1), dim=-1).mean()
                    znorm = torch.norm(z.reshape(B, -1), dim=-1).mean()
                    step = (snr * znorm / gnorm) ** 2 * 2 * alpha
                    x_mean = x + _b(step) * grad
                    x = x_mean + _b(torch.sqrt(step * 2)) * z
            elif corrector == "ald":                                                      # sampling.py:300-319
                alpha = torch.ones_like(t) if isinstance(sde, _VE) else sde.alphas[(t * (sde.N - 1) / sde.T).long()]
                std = sde.marginal_prob(x, t)[1]
                for _ in range(n_steps):
                    grad = score_fn(cfg, sd, sde, x, t, continuous)
                    score_norms.append(float(torch.norm(grad.reshape(B, -1), dim=-1).mean()))
                    step = (snr * std) ** 2 * 2 * alpha
                    x_mean = x + _b(step) * grad
                    x = x_mean + noises[i, 0] * _b(torch.sqrt(step * 2))
            elif corrector != "none":
                raise ValueError(corrector)
            if project is not None:
                x, x_mean = project(sde, x, t, noises[i, 2])
            if predictor == "reverse_diffusion":                                          # sampling.py:195-200
                f, G = sde.discretize(x, t)
                s = score_fn(cfg, sd, sde, x, t, continuous)
                score_norms.append(float(torch.norm(s.reshape(B, -1), dim=-1).mean()))
                rev_f = f - _b(G) ** 2 * s * (0.5 if probability_flow else 1.)            # sde_lib.py:105
                rev_G = torch.zeros_like(G) if probability_flow else G                    # sde_lib.py:106
                x_mean = x - rev_f
                x = x_mean + _b(rev_G) * noises[i, 1]
            elif predictor == "euler_maruyama":                                           # sampling.py:181-187
                dt = -1. / sde.N
                drift, diffusion = sde.sde(x, t)
                s = score_fn(cfg, sd, sde, x, t, continuous)
                score_norms.append(float(torch.norm(s.reshape(B, -1), dim=-1).mean()))
                drift = drift - _b(diffusion) ** 2 * s                                    # sde_lib.py:96
                x_mean = x + drift * dt
                x = x_mean + _b(diffusion) * np.sqrt(-dt) * noises[i, 1]
            elif predictor == "ancestral_sampling":                                       # sampling.py:213-239
                idx = (t * (sde.N - 1) / sde.T).long()
                s = score_fn(cfg, sd, sde, x, t, continuous)
                score_norms.append(float(torch.norm(s.reshape(B, -1), dim=-1).mean()))
                if isinstance(sde, _VE):
                    sigma = sde.discrete_sigmas[idx]
                    adj = torch.where(idx == 0, torch.zeros_like(t), sde.discrete_sigmas[idx - 1])
                    x_mean = x + s * _b(sigma ** 2 - adj ** 2)
                    std = torch.sqrt((adj ** 2 * (sigma ** 2 - adj ** 2)) / (sigma ** 2))
                    x = x_mean + _b(std) * noises[i, 1]
                else:
                    beta = sde.discrete_betas[idx]
                    x_mean = (x + _b(beta) * s) / _b(torch.sqrt(1. - beta))
                    x = x_mean + _b(torch.sqrt(beta)) * noises[i, 1]
            elif predictor != "none":
                raise ValueError(predictor)
            if project is not None:
                x, x_mean = project(sde, x, t, noises[i, 3])
            x_steps.append(x.clone())
    return dict(samples=(x_mean if denoise else x), x_steps=x_steps, score_norms=score_norms)


_COLOR_M = torch.tensor([[5.7735014e-01, -8.1649649e-01, 4.7008697e-08],                              # controllable_generation.py:103-105
                         [5.7735026e-01, 4.0824834e-01, 7.0710671e-01],
                         [5.7735026e-01, 4.0824822e-01, -7.0710683e-01]])


def inpaint(cfg, sd, sde_kind, sde_kwargs, data, mask, prior, noises, **kw):
    """controllable_generation.py:42-81 (get_pc_inpainter) with injected prior / noises."""
    def project(sde, x, t, z):                                                                        # :44-52
        mean, std = sde.marginal_prob(data, t)
        known = mean + z * _b(std)
        x = x * (1. - mask) + known * mask
        return x, x * (1. - mask) + mean * mask
    x0 = data * mask + prior * (1. - mask)                                                            # :74
    return pc_sample(cfg, sd, sde_kind, sde_kwargs, x0, noises, project=project, **kw)


def colorize(cfg, sd, sde_kind, sde_kwargs, gray, prior, noises, **kw):
    """controllable_generation.py:103-178 (get_pc_colorizer) with injected prior / noises."""
    M, invM = _COLOR_M, torch.inverse(_COLOR_M)
    dec = lambda v: torch.einsum('bihw,ij->bjhw', v, M)
    cpl = lambda v: torch.einsum('bihw,ij->bjhw', v, invM)
    mask = torch.cat([torch.ones_like(gray[:, :1]), torch.zeros_like(gray[:, 1:])], dim=1)            # :148-151

    def project(sde, x, t, z):                                                                        # :136-144
        mean, std = sde.marginal_prob(dec(gray), t)
        known = mean + z * _b(std)
        x = cpl(dec(x) * (1. - mask) + known * mask)
        return x, cpl(dec(x) * (1. - mask) + mean * mask)
    x0 = cpl(dec(gray) * mask + dec(prior * (1. - mask)))                                             # :170-172
    return pc_sample(cfg, sd, sde_kind, sde_kwargs, x0, noises, project=project, **kw)


def dsm_loss(cfg, sd, sde_kind, sde_kwargs, batch, t, z, reduce_mean=False, likelihood_weighting=False):
    """losses.py:73-99 with injected (t, z); eval-mode network (dropout off)."""
    sde = make_sde(sde_kind, **sde_kwargs)
    mean, std = sde.marginal_prob(batch, t)
    perturbed = mean + _b(std) * z
    s = score_fn(cfg, sd, sde, perturbed, t)
    red = (lambda v: torch.mean(v, dim=-1)) if reduce_mean else (lambda v: 0.5 * torch.sum(v, dim=-1))
    if not likelihood_weighting:
        losses = red(torch.square(s * _b(std) + z).reshape(batch.shape[0], -1))
    else:
        g2 = sde.sde(torch.zeros_like(batch), t)[1] ** 2
        losses = red(torch.square(s + z / _b(std)).reshape(batch.shape[0], -1)) * g2
    return torch.mean(losses)
