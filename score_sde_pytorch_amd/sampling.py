"""Predictor-corrector and probability-flow samplers.

Plugin surface identical to the reference's sampling.py: `register_predictor`,
`register_corrector`, `get_predictor`, `get_corrector`, `get_sampling_fn`, `Predictor`,
`Corrector`, the stock predictors/correctors, `shared_predictor_update_fn`,
`shared_corrector_update_fn`, `get_pc_sampler`, `get_ode_sampler` (sampling.py:30-485).
User-registered predictors/correctors keep working: they go through the generic python
loop, with every score evaluation running as one HIP program.

Fast path (what BASELINE measures): when predictor, corrector and SDE are the stock classes,
`pc_sampler` lowers one whole PC iteration -- label fill, U-Net forward, rocRAND noise,
per-sample norms, Langevin update, U-Net forward, noise, predictor update, step counter --
into ONE static program captured as a hipGraph and replayed N times.  Per-step scalars
(sigma_i, G_i, ...) live in device tables indexed by a device-resident step counter, so the
graph is replayed with unchanged arguments and the host issues one call per PC step instead
of ~1,900 kernel launches.  Sampling needs no collectives: one independent
sampler per GPU.
"""
import abc
import functools

import numpy as np
import torch

from . import sde_lib
from .models import utils as mutils
from .models.utils import get_score_fn

_CORRECTORS = {}
_PREDICTORS = {}


def _registrar(table, cls, name):
    def _do(c):
        key = c.__name__ if name is None else name
        if key in table:
            raise ValueError(f'Already registered model with name: {key}')
        table[key] = c
        return c
    return _do if cls is None else _do(cls)


def register_predictor(cls=None, *, name=None):
    """Class decorator registering a Predictor under `name` (sampling.py:34-50)."""
    return _registrar(_PREDICTORS, cls, name)


def register_corrector(cls=None, *, name=None):
    """Class decorator registering a Corrector under `name` (sampling.py:53-69)."""
    return _registrar(_CORRECTORS, cls, name)


def get_predictor(name):
    return _PREDICTORS[name]


def get_corrector(name):
    return _CORRECTORS[name]


def get_sampling_fn(config, sde, shape, inverse_scaler, eps):
    """Build the sampler named by `config.sampling.method` ('pc' or 'ode'), sampling.py:80-123."""
    method = config.sampling.method.lower()
    if method == 'ode':
        return get_ode_sampler(sde=sde, shape=shape, inverse_scaler=inverse_scaler,
                               denoise=config.sampling.noise_removal, eps=eps, device=config.device)
    if method == 'pc':
        return get_pc_sampler(sde=sde, shape=shape,
                              predictor=get_predictor(config.sampling.predictor.lower()),
                              corrector=get_corrector(config.sampling.corrector.lower()),
                              inverse_scaler=inverse_scaler, snr=config.sampling.snr,
                              n_steps=config.sampling.n_steps_each,
                              probability_flow=config.sampling.probability_flow,
                              continuous=config.training.continuous, denoise=config.sampling.noise_removal,
                              eps=eps, device=config.device)
    raise ValueError(f"Sampler name {config.sampling.method} unknown.")


def _b(v):
    return v[:, None, None, None]


class Predictor(abc.ABC):
    """update_fn(x, t) -> (x, x_mean); owns the reverse SDE built from `score_fn` (sampling.py:126-148)."""

    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__()
        self.sde = sde
        self.rsde = sde.reverse(score_fn, probability_flow)
        self.score_fn = score_fn

    @abc.abstractmethod
    def update_fn(self, x, t):
        pass


class Corrector(abc.ABC):
    """update_fn(x, t) -> (x, x_mean) (sampling.py:151-173)."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__()
        self.sde, self.score_fn, self.snr, self.n_steps = sde, score_fn, snr, n_steps

    @abc.abstractmethod
    def update_fn(self, x, t):
        pass


def _apply(x, y=None, b=None, z=None, c=None, a=None):
    """x_mean = a[n] x + b[n] y ; x = x_mean + c[n] z as ONE launch of ssde_sample_update (per-sample coefficients, the
    reference's fp32 rounding per product and sum, except the VP ancestral rule -- see there).  This is the arithmetic of every stock update below when it runs in
    the GENERIC loop -- a user-registered subclass, a model other than NCSNpp, per-sample times; the stock classes on an
    NCSNpp model never get here (pc_engine.FusedPCSampler runs whole iterations as one program)."""
    from . import hipops
    return hipops.sample_update(x, y=y, b=b, z=z, c=c, a=a)


@register_predictor(name='euler_maruyama')
class EulerMaruyamaPredictor(Predictor):
    """x <- x + drift dt + g sqrt(-dt) z with dt = -1/N (sampling.py:176-187)."""

    def update_fn(self, x, t):
        dt = -1. / self.rsde.N
        drift, diffusion = self.rsde.sde(x, t)
        return _apply(x, y=drift, b=dt, z=torch.randn_like(x), c=diffusion * np.sqrt(-dt))


@register_predictor(name='reverse_diffusion')
class ReverseDiffusionPredictor(Predictor):
    """x <- x - rev_f + G z from the SDE's own discretisation (sampling.py:190-200)."""

    def update_fn(self, x, t):
        f, G = self.rsde.discretize(x, t)
        return _apply(x, y=f, b=-1.0, z=torch.randn_like(x), c=G)


@register_predictor(name='ancestral_sampling')
class AncestralSamplingPredictor(Predictor):
    """DDPM / SMLD ancestral step (sampling.py:203-239); VE and VP only."""

    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__(sde, score_fn, probability_flow)
        if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
        assert not probability_flow, "Probability flow not supported by ancestral sampling"

    def vesde_update_fn(self, x, t):
        sde = self.sde
        idx = (t * (sde.N - 1) / sde.T).long()
        table = sde.discrete_sigmas.to(t.device)
        sigma = table[idx]
        prev = torch.where(idx == 0, torch.zeros_like(t), table[idx - 1])
        gap = sigma ** 2 - prev ** 2                                        # variance removed by this step
        std = torch.sqrt((prev ** 2 * gap) / (sigma ** 2))
        return _apply(x, y=self.score_fn(x, t), b=gap, z=torch.randn_like(x), c=std)

    def vpsde_update_fn(self, x, t):
        sde = self.sde
        idx = (t * (sde.N - 1) / sde.T).long()
        beta = sde.discrete_betas.to(t.device)[idx]
        # (x + beta score) / sqrt(1 - beta) as a x + b score with the division folded into the coefficients.  NOT bit-faithful
        # to sampling.py:236 (a reciprocal and two products instead of one sum and one division: differences of an ulp or
        # two per step) -- the one stock rule whose rounding differs from the reference's operation order; it is covered by
        # the 1e-4 / 1e-3 variant goldens, not by a bit-equality claim
        inv = 1. / torch.sqrt(1. - beta)
        return _apply(x, y=self.score_fn(x, t), a=inv, b=beta * inv, z=torch.randn_like(x), c=torch.sqrt(beta))

    def update_fn(self, x, t):
        if isinstance(self.sde, sde_lib.VESDE):
            return self.vesde_update_fn(x, t)
        return self.vpsde_update_fn(x, t)


@register_predictor(name='none')
class NonePredictor(Predictor):
    def __init__(self, sde, score_fn, probability_flow=False):
        pass

    def update_fn(self, x, t):
        return x, x


def _check_corrector_sde(sde):
    if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE, sde_lib.subVPSDE)):
        raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")


def _langevin_alpha(sde, t):
    if isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE)):
        return sde.alphas.to(t.device)[(t * (sde.N - 1) / sde.T).long()]
    return torch.ones_like(t)


def _batch_mean_norm(v):
    return torch.norm(v.reshape(v.shape[0], -1), dim=-1).mean()


@register_corrector(name='langevin')
class LangevinCorrector(Corrector):
    """Langevin MCMC with the step size set from BATCH-MEAN norms (sampling.py:253-282)."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        _check_corrector_sde(sde)

    def update_fn(self, x, t):
        alpha = _langevin_alpha(self.sde, t)
        x_mean = x
        for _ in range(self.n_steps):
            grad = self.score_fn(x, t)
            noise = torch.randn_like(x)
            step = (self.snr * _batch_mean_norm(noise) / _batch_mean_norm(grad)) ** 2 * 2 * alpha      # [B]
            x, x_mean = _apply(x, y=grad, b=step, z=noise, c=torch.sqrt(step * 2))
        return x, x_mean


@register_corrector(name='ald')
class AnnealedLangevinDynamics(Corrector):
    """NCSN's annealed Langevin dynamics, step from the marginal std (sampling.py:285-319)."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        _check_corrector_sde(sde)

    def update_fn(self, x, t):
        step = (self.snr * self.sde.marginal_prob(x, t)[1]) ** 2 * 2 * _langevin_alpha(self.sde, t)  # [B]
        x_mean = x
        for _ in range(self.n_steps):
            x, x_mean = _apply(x, y=self.score_fn(x, t), b=step, z=torch.randn_like(x), c=torch.sqrt(step * 2))
        return x, x_mean


@register_corrector(name='none')
class NoneCorrector(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        pass

    def update_fn(self, x, t):
        return x, x


def shared_predictor_update_fn(x, t, sde, model, predictor, probability_flow, continuous):
    """One predictor update with a freshly built score_fn (sampling.py:333-341)."""
    score_fn = mutils.get_score_fn(sde, model, train=False, continuous=continuous)
    cls = NonePredictor if predictor is None else predictor
    return cls(sde, score_fn, probability_flow).update_fn(x, t)


def shared_corrector_update_fn(x, t, sde, model, corrector, continuous, snr, n_steps):
    """One corrector update with a freshly built score_fn (sampling.py:344-352)."""
    score_fn = mutils.get_score_fn(sde, model, train=False, continuous=continuous)
    cls = NoneCorrector if corrector is None else corrector
    return cls(sde, score_fn, snr, n_steps).update_fn(x, t)


def get_pc_sampler(sde, shape, predictor, corrector, inverse_scaler, snr, n_steps=1, probability_flow=False,
                   continuous=False, denoise=True, eps=1e-3, device='cuda'):
    """Predictor-corrector sampler factory (sampling.py:355-411).

    Returns `pc_sampler(model) -> (samples, nfe)`.  Two extra keyword arguments exist for parity
    testing (the reference's RNG stream cannot be reproduced): `x_init` (the prior sample)
    and `noises` ([N, 2, *shape]: corrector / predictor noise per step).  `pc_sampler.last_path` tells
    which path ran ('fused-graph', 'fused-eager' or 'generic')."""
    predictor_update_fn = functools.partial(shared_predictor_update_fn, sde=sde, predictor=predictor,
                                            probability_flow=probability_flow, continuous=continuous)
    corrector_update_fn = functools.partial(shared_corrector_update_fn, sde=sde, corrector=corrector,
                                            continuous=continuous, snr=snr, n_steps=n_steps)
    fused_cache = {}

    def pc_sampler(model, x_init=None, noises=None, seed=None, use_graph=True, max_steps=None):
        from . import pc_engine
        with torch.no_grad():
            x = (sde.prior_sampling(shape) if x_init is None else x_init).to(device)
            plan = pc_engine.plan_fused(sde, predictor, corrector, model, continuous, x, probability_flow)
            if plan is not None:
                key = id(model)
                eng = fused_cache.get(key)
                if eng is None:
                    eng = pc_engine.FusedPCSampler(model, sde, plan, shape, snr=snr, n_steps=n_steps,
                                                   probability_flow=probability_flow, eps=eps, device=x.device)
                    fused_cache[key] = eng
                x_fin, x_mean = eng.run(x, noises=noises, seed=seed, use_graph=use_graph, max_steps=max_steps)
                pc_sampler.last_path = eng.last_path
                pc_sampler.engine = eng
                return inverse_scaler(x_mean if denoise else x_fin), sde.N * (n_steps + 1)
            # generic path: arbitrary registered predictors / correctors / SDE subclasses
            pc_sampler.last_path = 'generic'
            timesteps = torch.linspace(sde.T, eps, sde.N, device=device)
            x_mean = x
            steps = sde.N if max_steps is None else max_steps
            real_randn_like = torch.randn_like
            for i in range(steps):
                vec_t = torch.ones(shape[0], device=timesteps.device) * timesteps[i]
                if noises is not None:
                    torch.randn_like = lambda t, _z=noises[i, 0]: _z.to(t.device)
                try:
                    x, x_mean = corrector_update_fn(x, vec_t, model=model)
                    if noises is not None:
                        torch.randn_like = lambda t, _z=noises[i, 1]: _z.to(t.device)
                    x, x_mean = predictor_update_fn(x, vec_t, model=model)
                finally:
                    torch.randn_like = real_randn_like
            return inverse_scaler(x_mean if denoise else x), sde.N * (n_steps + 1)

    pc_sampler.last_path = None
    return pc_sampler


def get_ode_sampler(sde, shape, inverse_scaler, denoise=False, rtol=1e-5, atol=1e-5, method='RK45', eps=1e-3,
                    device='cuda'):
    """Probability-flow ODE sampler with adaptive RK45 (sampling.py:414-485).

    The drift evaluation (one U-Net forward per function evaluation) runs as a HIP program.  With method='RK45' on a
    GPU tensor the integrator is ode.solve_rk45 -- scipy's RK45 algorithm with the fp64 state resident on the device
    (no per-evaluation host round trip, models/utils.py:181-188); other methods, or SSDE_HOST_ODE=1, use
    scipy.integrate.solve_ivp on the host exactly as the reference does."""

    def denoise_update_fn(model, x):
        score_fn = get_score_fn(sde, model, train=False, continuous=True)
        vec_eps = torch.ones(x.shape[0], device=x.device) * eps
        _, x = ReverseDiffusionPredictor(sde, score_fn, probability_flow=False).update_fn(x, vec_eps)
        return x

    def drift_fn(model, x, t):
        score_fn = get_score_fn(sde, model, train=False, continuous=True)
        return sde.reverse(score_fn, probability_flow=True).sde(x, t)[0]

    def ode_sampler(model, z=None):
        from . import ode
        with torch.no_grad():
            x = sde.prior_sampling(shape).to(device) if z is None else z

            if ode.FusedDrift.applies(model, sde, x):
                # stock SDE + NCSNpp: stage arithmetic, U-Net program and drift are all HIP launches, no torch arithmetic
                # cached on the model (not in this closure): every sampler built for the same SDE object, shape and
                # device shares the lowered program, its packed weights and its captured graph (the rhs keeps `sde` alive,
                # so the id cannot be recycled)
                key = ("drift", id(sde), tuple(shape), x.device.index)
                rhs, _ = ode.rhs_cache_get(model, key, lambda: ode.FusedDrift(model, sde, shape, x.device))
                ode_sampler.last_path = "fused"
            else:
                def rhs(t, y):
                    xt = y.reshape(shape).to(torch.float32)
                    return drift_fn(model, xt, torch.full((shape[0],), float(t), device=xt.device)).reshape(-1).to(torch.float64)
                ode_sampler.last_path = "generic"
            y, nfev = ode.integrate_ode(rhs, (sde.T, eps), x.reshape(-1).to(torch.float64), rtol, atol, method)
            x = y.reshape(shape).to(torch.float32)
            if denoise:
                x = denoise_update_fn(model, x)
            return inverse_scaler(x), nfev

    ode_sampler.last_path = None
    return ode_sampler
