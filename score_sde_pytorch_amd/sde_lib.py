"""Forward SDEs, their marginals / discretisations, and the reverse-time SDE factory.

API-compatible with the reference's sde_lib.py (SDE base class :7-109, VPSDE :112-164,
subVPSDE :167-204, VESDE :207-254): constructors, `T`, `sde`, `marginal_prob`,
`prior_sampling` (CPU tensor), `prior_logp`, `discretize`, `reverse`, and the attributes
samplers read (`N`, `discrete_sigmas`, `discrete_betas`, `alphas`, `alphas_cumprod`,
`sqrt_alphas_cumprod`, `sqrt_1m_alphas_cumprod`).  User-defined subclasses plug in the
same way.  These are [B]-sized scalar computations; the per-pixel arithmetic they feed is
executed by the fused HIP update kernels (engine.PCSampler) when the stock
predictor/corrector pair is used.
"""
import abc
import math

import numpy as np
import torch


def _bcast(v):
    """[B] -> [B,1,1,1]"""
    return v[:, None, None, None]


class SDE(abc.ABC):
    """dx = f(x,t) dt + g(t) dw on t in [0, T], discretised in N steps."""

    def __init__(self, N):
        super().__init__()
        self.N = N

    @property
    @abc.abstractmethod
    def T(self):
        """End time of the SDE."""

    @abc.abstractmethod
    def sde(self, x, t):
        """Return (drift [B,C,H,W], diffusion [B])."""

    @abc.abstractmethod
    def marginal_prob(self, x, t):
        """Return (mean, std[B]) of p_t(x(t) | x(0)=x)."""

    @abc.abstractmethod
    def prior_sampling(self, shape):
        """One CPU sample from p_T."""

    @abc.abstractmethod
    def prior_logp(self, z):
        """log p_T(z), [B]."""

    def discretize(self, x, t):
        """Euler-Maruyama step x_{i+1} = x_i + f_i + G_i z_i with dt = 1/N (sde_lib.py:52-69)."""
        dt = 1 / self.N
        drift, diffusion = self.sde(x, t)
        return drift * dt, diffusion * torch.sqrt(torch.tensor(dt, device=t.device))

    def reverse(self, score_fn, probability_flow=False):
        """Reverse-time SDE (or probability-flow ODE) driven by `score_fn` (sde_lib.py:71-109)."""
        n_steps, t_end = self.N, self.T
        fwd_sde, fwd_discretize = self.sde, self.discretize
        half = 0.5 if probability_flow else 1.0

        class RSDE(self.__class__):
            def __init__(self):
                self.N = n_steps
                self.probability_flow = probability_flow

            @property
            def T(self):
                return t_end

            def sde(self, x, t):
                drift, diffusion = fwd_sde(x, t)
                drift = drift - _bcast(diffusion) ** 2 * score_fn(x, t) * half
                return drift, (0. if probability_flow else diffusion)

            def discretize(self, x, t):
                f, G = fwd_discretize(x, t)
                rev_f = f - _bcast(G) ** 2 * score_fn(x, t) * half
                return rev_f, (torch.zeros_like(G) if probability_flow else G)

        return RSDE()


def _std_normal_logp(z):
    d = int(np.prod(z.shape[1:]))
    return -d / 2. * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.


class VPSDE(SDE):
    """Variance-preserving SDE, beta(t) linear in t (sde_lib.py:112-164)."""

    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0, self.beta_1 = beta_min, beta_max
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1. - self.discrete_betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1. - self.alphas_cumprod)

    @property
    def T(self):
        return 1

    def _beta(self, t):
        return self.beta_0 + t * (self.beta_1 - self.beta_0)

    def _log_mean_coeff(self, t):
        return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0

    def sde(self, x, t):
        beta_t = self._beta(t)
        return -0.5 * _bcast(beta_t) * x, torch.sqrt(beta_t)

    def marginal_prob(self, x, t):
        lmc = self._log_mean_coeff(t)
        return torch.exp(_bcast(lmc)) * x, torch.sqrt(1. - torch.exp(2. * lmc))

    def prior_sampling(self, shape):
        return torch.randn(*shape)

    def prior_logp(self, z):
        return _std_normal_logp(z)

    def discretize(self, x, t):
        """DDPM ancestral discretisation (sde_lib.py:155-164)."""
        idx = (t * (self.N - 1) / self.T).long()
        beta = self.discrete_betas.to(x.device)[idx]
        alpha = self.alphas.to(x.device)[idx]
        return _bcast(torch.sqrt(alpha)) * x - x, torch.sqrt(beta)


class subVPSDE(SDE):
    """Sub-VP SDE (sde_lib.py:167-204)."""

    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0, self.beta_1 = beta_min, beta_max

    @property
    def T(self):
        return 1

    def sde(self, x, t):
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
        discount = 1. - torch.exp(-2 * self.beta_0 * t - (self.beta_1 - self.beta_0) * t ** 2)
        return -0.5 * _bcast(beta_t) * x, torch.sqrt(beta_t * discount)

    def marginal_prob(self, x, t):
        lmc = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return _bcast(torch.exp(lmc)) * x, 1 - torch.exp(2. * lmc)

    def prior_sampling(self, shape):
        return torch.randn(*shape)

    def prior_logp(self, z):
        return _std_normal_logp(z)


class VESDE(SDE):
    """Variance-exploding SDE, sigma(t) = sigma_min (sigma_max/sigma_min)^t (sde_lib.py:207-254)."""

    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000):
        super().__init__(N)
        self.sigma_min, self.sigma_max = sigma_min, sigma_max
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), N))

    @property
    def T(self):
        return 1

    def _sigma(self, t):
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def sde(self, x, t):
        g = self._sigma(t) * torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min)),
                                                     device=t.device))
        return torch.zeros_like(x), g

    def marginal_prob(self, x, t):
        return x, self._sigma(t)

    def prior_sampling(self, shape):
        return torch.randn(*shape) * self.sigma_max

    def prior_logp(self, z):
        d = int(np.prod(z.shape[1:]))
        return -d / 2. * np.log(2 * np.pi * self.sigma_max ** 2) \
            - torch.sum(z ** 2, dim=(1, 2, 3)) / (2 * self.sigma_max ** 2)

    def discretize(self, x, t):
        """SMLD discretisation G_i = sqrt(sigma_i^2 - sigma_{i-1}^2), sigma_{-1} = 0 (sde_lib.py:246-254)."""
        idx = (t * (self.N - 1) / self.T).long()
        table = self.discrete_sigmas.to(t.device)
        sigma = table[idx]
        prev = torch.where(idx == 0, torch.zeros_like(t), table[idx - 1])
        return torch.zeros_like(x), torch.sqrt(sigma ** 2 - prev ** 2)


__all__ = ["SDE", "VPSDE", "subVPSDE", "VESDE", "math"]
