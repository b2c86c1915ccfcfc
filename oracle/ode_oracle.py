"""CPU oracle for the probability-flow ODE sampler and the likelihood (TEST INFRASTRUCTURE ONLY).

Restates, on top of unet_oracle / sampler_oracle (functional torch fp32, scipy's solve_ivp as the reference uses it):
  * get_ode_sampler.ode_sampler            sampling.py:449-483   (drift_fn :444-448, denoise_update_fn :435-442)
  * get_div_fn / get_likelihood_fn         likelihood.py:26-37, 69-111
  * RSDE.sde (probability flow)            sde_lib.py:93-100
  * to/from_flattened_numpy                models/utils.py:181-188  (fp64 numpy state <-> fp32 tensors)
  * prior_logp                             sde_lib.py:150-154, 201-204, 241-244
Pinned against the reference by oracle/gen_golden_ode.py, which runs the reference's own two functions and asserts that
this restatement reproduces their outputs (tests/golden/ode_small.npz).
"""
import numpy as np
import torch
from scipy import integrate

from . import sampler_oracle as S


def drift(cfg, sd, sde, x, t):
    """RSDE(probability_flow=True).sde(x, t)[0]  (sde_lib.py:93-97)."""
    f, g = sde.sde(x, t)
    return f - g[:, None, None, None] ** 2 * S.score_fn(cfg, sd, sde, x, t, continuous=True) * 0.5


def prior_logp(sde, z):
    n = float(np.prod(z.shape[1:]))
    if isinstance(sde, S._VE):                                                     # sde_lib.py:241-244
        return -n / 2. * np.log(2 * np.pi * sde.sigma_max ** 2) - torch.sum(z ** 2, dim=(1, 2, 3)) / (2 * sde.sigma_max ** 2)
    return -n / 2. * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.     # sde_lib.py:150-154, 201-204


def ode_sample(cfg, sd, sde_kind, sde_kwargs, z, rtol=1e-5, atol=1e-5, method="RK45", eps=1e-3, denoise=False):
    """sampling.py:449-483 with the latent injected; returns (samples, nfe)."""
    sde = S.make_sde(sde_kind, **sde_kwargs)
    shape = tuple(z.shape)
    with torch.no_grad():
        def ode_func(t, y):                                                        # sampling.py:466-470
            x = torch.from_numpy(y.reshape(shape)).type(torch.float32)
            vec_t = torch.ones(shape[0]) * t
            return drift(cfg, sd, sde, x, vec_t).numpy().reshape((-1,))
        sol = integrate.solve_ivp(ode_func, (sde.T, eps), z.numpy().reshape((-1,)), rtol=rtol, atol=atol, method=method)
        x = torch.tensor(sol.y[:, -1]).reshape(shape).type(torch.float32)
        if denoise:                                                                # sampling.py:435-442 -> :195-200
            vec_eps = torch.ones(shape[0]) * eps
            f, G = sde.discretize(x, vec_eps)
            x = x - (f - G[:, None, None, None] ** 2 * S.score_fn(cfg, sd, sde, x, vec_eps, continuous=True))
    return x, int(sol.nfev)


def rhs_augmented(cfg, sd, sde, x, t, epsilon):
    """One evaluation of the likelihood ODE's right-hand side: (drift, Hutchinson divergence eps^T J eps), likelihood.py:29-35."""
    with torch.enable_grad():
        xg = x.detach().clone().requires_grad_(True)
        d = drift(cfg, sd, sde, xg, t)
        grad_fn_eps = torch.autograd.grad(torch.sum(d * epsilon), xg)[0]
    return d.detach(), torch.sum(grad_fn_eps * epsilon, dim=(1, 2, 3))


def likelihood(cfg, sd, sde_kind, sde_kwargs, data, epsilon, inverse_scaler, rtol=1e-5, atol=1e-5, method="RK45", eps=1e-5):
    """likelihood.py:69-111 with the Hutchinson probe injected; returns (bpd, z, nfe)."""
    sde = S.make_sde(sde_kind, **sde_kwargs)
    shape = tuple(data.shape)
    B = shape[0]

    def ode_func(t, y):                                                            # likelihood.py:90-95
        x = torch.from_numpy(y[:-B].reshape(shape)).type(torch.float32)
        vec_t = torch.ones(B) * t
        d, div = rhs_augmented(cfg, sd, sde, x, vec_t, epsilon)
        return np.concatenate([d.numpy().reshape((-1,)), div.numpy().reshape((-1,))], axis=0)
    init = np.concatenate([data.numpy().reshape((-1,)), np.zeros((B,))], axis=0)
    sol = integrate.solve_ivp(ode_func, (eps, sde.T), init, rtol=rtol, atol=atol, method=method)
    zp = sol.y[:, -1]
    z = torch.from_numpy(zp[:-B].reshape(shape)).type(torch.float32)
    delta_logp = torch.from_numpy(zp[-B:].reshape((B,))).type(torch.float32)
    bpd = -(prior_logp(sde, z) + delta_logp) / np.log(2)                           # likelihood.py:103-110
    bpd = bpd / np.prod(shape[1:]) + (7. - inverse_scaler(-1.))
    return bpd, z, int(sol.nfev)
