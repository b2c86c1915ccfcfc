"""Bits/dim by the probability-flow ODE with a Hutchinson-Skilling divergence estimate (drop-in for the reference's
likelihood.py:26-113: `get_div_fn`, `get_likelihood_fn`, same arguments and return values).

For an NCSNpp model and a stock SDE the whole right-hand side -- drift and eps^T (d drift / d x) eps -- is ONE device
program per evaluation (ode.FusedLikelihoodRhs: forward program, drift kernel, input-gradient program with the probe as
the cotangent, per-sample divergence kernel; captured into a hipGraph).  Other models / SDEs go through `get_div_fn`
below: the drift and its vector-Jacobian product run on the HIP forward / backward programs through the autograd bridge
(autograd.py).  Only d drift / d x is needed, so the parameters are frozen for the duration of a call: the backward
program is then lowered without its weight-gradient kernels (backward.TrainEngine(param_grads=False)).
The augmented state [x (B*D values) | accumulated log-density change (B values)] is one fp64 tensor; with method='RK45'
on GPU data it is integrated by ode.solve_rk45 (scipy's RK45 algorithm, state resident on the device), otherwise -- or
with SSDE_HOST_ODE=1 -- by scipy.integrate.solve_ivp on the host, as the reference does (ode.solve_host).
"""
import contextlib
import math

import torch

from . import ode
from .models import utils as mutils


def get_div_fn(fn):
    """div fn(x, t) estimated as eps^T (d fn / d x) eps (Hutchinson-Skilling, likelihood.py:26-37): one vector-Jacobian
    product with the probe as the cotangent."""

    def div_fn(x, t, eps):
        xg = x.detach().requires_grad_(True)
        with torch.enable_grad():
            (vjp,) = torch.autograd.grad(fn(xg, t), xg, grad_outputs=eps)
        return (vjp * eps).flatten(1).sum(dim=1)

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


def _probe(data, kind):
    if kind == 'Rademacher':
        return torch.randint_like(data, low=0, high=2).float() * 2 - 1.
    if kind == 'Gaussian':
        return torch.randn_like(data)
    raise NotImplementedError(f"Hutchinson type {kind} unknown.")


def get_likelihood_fn(sde, inverse_scaler, hutchinson_type='Rademacher', rtol=1e-5, atol=1e-5, method='RK45', eps=1e-5):
    """Returns likelihood_fn(model, data) -> (bpd [B], z, nfe)  (likelihood.py:40-113)."""

    def likelihood_fn(model, data):
        batch, dims = data.shape[0], data[0].numel()
        with torch.no_grad(), _frozen(model):
            probe = _probe(data, hutchinson_type)
            y0 = torch.cat([data.reshape(-1), data.new_zeros(batch)]).to(torch.float64)
            if method == 'RK45' and ode.FusedLikelihoodRhs.applies(model, sde, data):
                # the whole right-hand side as one device program per evaluation (ode.FusedLikelihoodRhs): no torch
                # arithmetic, no autograd graph, no dtype round trips between U-Net evaluations
                # shared by every likelihood_fn of this SDE object; a small LRU (ode.rhs_cache_get)
                key = ("likelihood", id(sde), tuple(data.shape), data.device.index)
                rhs, fresh = ode.rhs_cache_get(model, key, lambda: ode.FusedLikelihoodRhs(model, sde, data.shape, probe, data.device))
                if not fresh:
                    rhs.set_probe(probe)
                likelihood_fn.last_path = "fused"
            else:
                score_fn = mutils.get_score_fn(sde, model, train=False, continuous=True)
                flow = sde.reverse(score_fn, probability_flow=True)          # the probability-flow ODE as a reverse SDE

                def drift(x, t):
                    return flow.sde(x, t)[0]
                divergence = get_div_fn(drift)

                def rhs(t, y):                                                # d/dt [x, delta log p] = [drift, div drift]
                    x = y[: batch * dims].reshape(data.shape).to(torch.float32)
                    vec_t = torch.full((batch,), float(t), device=x.device)
                    return torch.cat([drift(x, vec_t).reshape(-1), divergence(x, vec_t, probe)]).to(torch.float64)
                likelihood_fn.last_path = "generic"

            y1, nfe = ode.integrate_ode(rhs, (eps, sde.T), y0, rtol, atol, method)
            z = y1[: batch * dims].reshape(data.shape).to(torch.float32)
            delta_logp = y1[batch * dims:].to(torch.float32)
            nats = -(sde.prior_logp(z) + delta_logp)
            # bits per dimension, shifted for the [0, 255] pixel scale of the data (the reference's offset, likelihood.py:107-111)
            bpd = nats / (math.log(2.) * dims) + (7. - inverse_scaler(-1.))
            return bpd, z, nfe

    likelihood_fn.last_path = None
    return likelihood_fn
