"""Bits/dim by the probability-flow ODE with a Hutchinson-Skilling divergence estimate (drop-in for the reference's
likelihood.py:26-113: `get_div_fn`, `get_likelihood_fn`, same arguments and return values).

The drift and its vector-Jacobian product run on the HIP forward / backward programs through the autograd bridge
(autograd.py).  Only d drift / d x is needed, so the parameters are frozen for the duration of a call: the backward
program is then lowered without its weight-gradient kernels (backward.TrainEngine(param_grads=False)).  With
method='RK45' on GPU data the integrator is ode.solve_rk45 (scipy's RK45 algorithm, fp64 state resident on the device);
other methods, or SSDE_HOST_ODE=1, run scipy.integrate.solve_ivp on the host exactly as the reference does.
"""
import os
import contextlib

import numpy as np
import torch
from scipy import integrate

from .models import utils as mutils


def get_div_fn(fn):
    """Divergence of `fn` by the Hutchinson-Skilling trace estimator (likelihood.py:26-37)."""

    def div_fn(x, t, eps):
        with torch.enable_grad():
            x.requires_grad_(True)
            fn_eps = torch.sum(fn(x, t) * eps)
            grad_fn_eps = torch.autograd.grad(fn_eps, x)[0]
        x.requires_grad_(False)
        return torch.sum(grad_fn_eps * eps, dim=tuple(range(1, len(x.shape))))

    return div_fn


@contextlib.contextmanager
def _frozen(model):
    flags = [p.requires_grad for p in model.parameters()]
    for p in model.parameters():
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p, f in zip(model.parameters(), flags):
            p.requires_grad_(f)


def get_likelihood_fn(sde, inverse_scaler, hutchinson_type='Rademacher', rtol=1e-5, atol=1e-5, method='RK45', eps=1e-5):
    """Returns likelihood_fn(model, data) -> (bpd [B], z, nfe)  (likelihood.py:40-113)."""

    def drift_fn(model, x, t):
        score_fn = mutils.get_score_fn(sde, model, train=False, continuous=True)
        rsde = sde.reverse(score_fn, probability_flow=True)     # the probability-flow ODE is a special reverse SDE
        return rsde.sde(x, t)[0]

    def div_fn(model, x, t, noise):
        return get_div_fn(lambda xx, tt: drift_fn(model, xx, tt))(x, t, noise)

    def likelihood_fn(model, data):
        with torch.no_grad(), _frozen(model):
            shape = data.shape
            if hutchinson_type == 'Gaussian':
                epsilon = torch.randn_like(data)
            elif hutchinson_type == 'Rademacher':
                epsilon = torch.randint_like(data, low=0, high=2).float() * 2 - 1.
            else:
                raise NotImplementedError(f"Hutchinson type {hutchinson_type} unknown.")

            def ode_func(t, x):
                sample = mutils.from_flattened_numpy(x[:-shape[0]], shape).to(data.device).type(torch.float32)
                vec_t = torch.ones(sample.shape[0], device=sample.device) * t
                drift = mutils.to_flattened_numpy(drift_fn(model, sample, vec_t))
                logp_grad = mutils.to_flattened_numpy(div_fn(model, sample, vec_t, epsilon))
                return np.concatenate([drift, logp_grad], axis=0)

            if method == 'RK45' and data.is_cuda and os.environ.get("SSDE_HOST_ODE", "0") != "1":
                from . import ode

                def dev_func(t, y):
                    sample = y[:-shape[0]].reshape(shape).to(torch.float32)
                    vec_t = torch.ones(shape[0], device=sample.device) * t
                    drift = drift_fn(model, sample, vec_t).reshape(-1)
                    logp_grad = div_fn(model, sample, vec_t, epsilon).reshape(-1)
                    return torch.cat([drift, logp_grad]).to(torch.float64)
                init = torch.cat([data.reshape(-1).to(torch.float64), torch.zeros(shape[0], dtype=torch.float64, device=data.device)])
                yT, nfe = ode.solve_rk45(dev_func, (eps, sde.T), init, rtol=rtol, atol=atol)
                z = yT[:-shape[0]].reshape(shape).to(torch.float32)
                delta_logp = yT[-shape[0]:].to(torch.float32)
            else:
                init = np.concatenate([mutils.to_flattened_numpy(data), np.zeros((shape[0],))], axis=0)
                solution = integrate.solve_ivp(ode_func, (eps, sde.T), init, rtol=rtol, atol=atol, method=method)
                nfe = solution.nfev
                zp = solution.y[:, -1]
                z = mutils.from_flattened_numpy(zp[:-shape[0]], shape).to(data.device).type(torch.float32)
                delta_logp = mutils.from_flattened_numpy(zp[-shape[0]:], (shape[0],)).to(data.device).type(torch.float32)
            prior_logp = sde.prior_logp(z)
            bpd = -(prior_logp + delta_logp) / np.log(2)
            N = np.prod(shape[1:])
            bpd = bpd / N
            offset = 7. - inverse_scaler(-1.)       # the reference's conversion of log-likelihoods to bits/dim
            bpd = bpd + offset
            return bpd, z, nfe

    return likelihood_fn
