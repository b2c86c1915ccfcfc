"""Controllable generation with PC samplers: inpainting and colorization
(reference controllable_generation.py:8-180; same factories, arguments and return conventions).

Both are the PC loop of `sampling.get_pc_sampler` with a data-consistency projection after the corrector
update and after the predictor update.  For the stock predictor / corrector / SDE classes on an NCSNpp
model the whole iteration -- both U-Net evaluations, the updates and the two projections
(`ssde_project_update`, include/ssde.h) -- is one step program replayed as a hipGraph; user-registered
predictors / correctors run the generic loop, calling their `update_fn` with our `score_fn`.

Extra keyword arguments of the returned functions exist for parity testing (the reference's RNG stream
cannot be reproduced): `prior` (the prior sample) and `noises` ([N, 4, *shape]: corrector,
predictor, projection-after-corrector, projection-after-predictor noise of every iteration).
"""
import functools

import torch

from .sampling import shared_corrector_update_fn, shared_predictor_update_fn

# orthonormal colour decoupling: channel 0 of `decouple(img)` is the grey-scale image (controllable_generation.py:103-107)
_M = [[5.7735014e-01, -8.1649649e-01, 4.7008697e-08],
      [5.7735026e-01, 4.0824834e-01, 7.0710671e-01],
      [5.7735026e-01, 4.0824822e-01, -7.0710683e-01]]


def _controlled_pc(sde, predictor, corrector, inverse_scaler, snr, n_steps, probability_flow, continuous, denoise, eps,
                   to_space, from_space, matrices):
    """Shared driver: `to_space` / `from_space` map the state into / out of the space where the mask applies."""
    predictor_update_fn = functools.partial(shared_predictor_update_fn, sde=sde, predictor=predictor,
                                            probability_flow=probability_flow, continuous=continuous)
    corrector_update_fn = functools.partial(shared_corrector_update_fn, sde=sde, corrector=corrector,
                                            continuous=continuous, snr=snr, n_steps=n_steps)
    fused_cache = {}

    def project(x, known_mean, std, mask, z):
        # controllable_generation.py:47-51 / :139-143 (the new x enters x_mean, as in the reference)
        known = known_mean + z * std[:, None, None, None]
        x = from_space(to_space(x) * (1. - mask) + known * mask)
        x_mean = from_space(to_space(x) * (1. - mask) + known_mean * mask)
        return x, x_mean

    def run(model, x, data_in_space, mask, noises, seed, use_graph, max_steps):
        from . import pc_engine
        plan = pc_engine.plan_fused(sde, predictor, corrector, model, continuous, x, probability_flow)
        if plan is not None:
            key = (id(model), tuple(x.shape))
            eng = fused_cache.get(key)
            if eng is None:
                eng = fused_cache[key] = pc_engine.FusedPCSampler(
                    model, sde, plan, tuple(x.shape), snr=snr, n_steps=n_steps, probability_flow=probability_flow,
                    eps=eps, device=x.device, projection=matrices)
            eng.set_projection_inputs(data_in_space, mask)
            x_fin, x_mean = eng.run(x, noises=noises, seed=seed, use_graph=use_graph, max_steps=max_steps)
            run.last_path = eng.last_path
            return x_fin, x_mean
        run.last_path = 'generic'
        timesteps = torch.linspace(sde.T, eps, sde.N)
        x_mean = x
        real_randn_like = torch.randn_like
        for i in range(sde.N if max_steps is None else max_steps):
            vec_t = torch.ones(x.shape[0], device=x.device) * timesteps[i]
            known_mean, std = sde.marginal_prob(data_in_space, vec_t)
            for k, update_fn in enumerate((corrector_update_fn, predictor_update_fn)):
                if noises is not None:
                    torch.randn_like = lambda t, _z=noises[i, k]: _z.to(t.device)
                try:
                    x, x_mean = update_fn(x, vec_t, model=model)
                finally:
                    torch.randn_like = real_randn_like
                z = torch.randn_like(x) if noises is None else noises[i, 2 + k].to(x.device)
                x, x_mean = project(x, known_mean, std, mask, z)
        return x, x_mean

    run.last_path = None
    return run


def get_pc_inpainter(sde, predictor, corrector, inverse_scaler, snr, n_steps=1, probability_flow=False, continuous=False,
                     denoise=True, eps=1e-5):
    """Image inpainting with a PC sampler (controllable_generation.py:8-83).

    Returns `pc_inpainter(model, data, mask)`: `mask` is 1 on known pixels, 0 where pixels are to be generated."""
    ident = lambda v: v
    run = _controlled_pc(sde, predictor, corrector, inverse_scaler, snr, n_steps, probability_flow, continuous, denoise,
                         eps, ident, ident, dict(M=None, invM=None))

    def pc_inpainter(model, data, mask, prior=None, noises=None, seed=None, use_graph=True, max_steps=None):
        with torch.no_grad():
            prior = sde.prior_sampling(data.shape) if prior is None else prior
            x = data * mask + prior.to(data.device) * (1. - mask)                       # :74
            x, x_mean = run(model, x, data, mask.expand_as(data), noises, seed, use_graph, max_steps)
            pc_inpainter.last_path = run.last_path
            return inverse_scaler(x_mean if denoise else x)

    pc_inpainter.last_path = None
    return pc_inpainter


def get_pc_colorizer(sde, predictor, corrector, inverse_scaler, snr, n_steps=1, probability_flow=False, continuous=False,
                     denoise=True, eps=1e-5):
    """Image colorization with a PC sampler (controllable_generation.py:86-180).

    Returns `pc_colorizer(model, gray_scale_img)`; the R, G, B channels of `gray_scale_img` hold the same values."""
    M = torch.tensor(_M)
    invM = torch.inverse(M)

    def decouple(inputs):
        return torch.einsum('bihw,ij->bjhw', inputs, M.to(inputs.device))

    def couple(inputs):
        return torch.einsum('bihw,ij->bjhw', inputs, invM.to(inputs.device))

    def get_mask(image):
        return torch.cat([torch.ones_like(image[:, :1, ...]), torch.zeros_like(image[:, 1:, ...])], dim=1)

    run = _controlled_pc(sde, predictor, corrector, inverse_scaler, snr, n_steps, probability_flow, continuous, denoise,
                         eps, decouple, couple, dict(M=M.flatten().tolist(), invM=invM.flatten().tolist()))

    def pc_colorizer(model, gray_scale_img, prior=None, noises=None, seed=None, use_graph=True, max_steps=None):
        with torch.no_grad():
            shape = gray_scale_img.shape
            mask = get_mask(gray_scale_img)
            prior = sde.prior_sampling(shape) if prior is None else prior
            x = couple(decouple(gray_scale_img) * mask + decouple(prior.to(gray_scale_img.device) * (1. - mask)))   # :170-172
            x, x_mean = run(model, x, decouple(gray_scale_img), mask, noises, seed, use_graph, max_steps)
            pc_colorizer.last_path = run.last_path
            return inverse_scaler(x_mean if denoise else x)

    pc_colorizer.last_path = None
    return pc_colorizer
