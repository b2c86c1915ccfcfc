"""Generate tests/golden/ode_small.npz by running the REFERENCE's probability-flow ODE sampler and likelihood
(build container only; BASELINE config #5's path at a CPU-sized problem).

    python oracle/gen_golden_ode.py     # needs /root/reference; writes tests/golden/ode_small.npz  (~5 min of CPU)

What runs is the reference's own `sampling.get_ode_sampler` (sampling.py:414-485, scipy call :473) and
`likelihood.get_likelihood_fn` (likelihood.py:40-113, scipy call :99) on the down-sized sub-VP DDPM++ network of
tests/_util.small_config("ddpmpp") with seeded weights, an injected latent `z` (ode_sampler's own argument) and an
injected Hutchinson probe (torch.randint_like patched: the reference draws it on the data's device, likelihood.py:76).
The script also
  * asserts that oracle/ode_oracle.py reproduces every stored output, and
  * measures how far each output moves when the input is scaled by (1 + 1e-6) -- the conditioning of the adaptive
    integration through a random-weight network -- and stores it, because the GPU tests' tolerances are stated against
    it (an fp32 U-Net evaluation differs from the CPU one by ~1e-5 relative, ten such perturbations).
TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G                                  # noqa: E402  (sets sys.path for tests/ and the repo root)

import _util                                            # noqa: E402


def main():
    G.import_reference()
    from oracle import ode_oracle
    import models.utils as ref_mutils            # noqa  (reference)
    import models.ncsnpp                         # noqa
    import sde_lib as ref_sde_lib                # noqa
    import sampling as ref_sampling              # noqa
    import likelihood as ref_likelihood          # noqa
    import ml_collections

    def ref_cfg_like(cfg):
        def conv(v):
            if hasattr(v, "items"):
                d = ml_collections.ConfigDict()
                for k, x in v.items():
                    d[k] = conv(x)
                return d
            return v
        return conv(cfg)

    torch.set_num_threads(min(16, os.cpu_count()))
    case = _util.ODE_CASE
    cfg = _util.small_config("ddpmpp")
    cfg.device = torch.device("cpu")
    torch.manual_seed(0)
    ref_model = ref_mutils.get_model("ncsnpp")(ref_cfg_like(cfg)).eval()
    sd = _util.fix_top_level_groupnorm(_util.seeded_state_dict(ref_model, seed=1), ref_model)
    ref_model.load_state_dict(sd, strict=False)
    full_sd = dict(sd); full_sd["sigmas"] = ref_model.sigmas
    kw = case["sde_kwargs"]
    sde = ref_sde_lib.subVPSDE(**kw)
    z, data, epsilon = _util.ode_case_inputs()
    shape = tuple(z.shape)
    inv = _util.ode_inverse_scaler
    out = {}

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max())

    # ---- probability-flow ODE sampler (sampling.py:449-483), with and without the denoising step
    for tag, denoise in (("ode", False), ("ode_denoise", True)):
        smp = ref_sampling.get_ode_sampler(sde, shape, inv, denoise=denoise, rtol=case["rtol"], atol=case["atol"],
                                           method="RK45", eps=case["sample_eps"], device="cpu")
        x, nfe = smp(ref_model, z=z.clone())
        xo, nfe_o = ode_oracle.ode_sample(cfg, full_sd, "subvpsde", kw, z.clone(), rtol=case["rtol"], atol=case["atol"],
                                          eps=case["sample_eps"], denoise=denoise)
        e = rel(inv(xo), x)
        print("%-12s nfe %d (oracle %d)  |x| max %.4g  oracle-vs-reference rel err %.3g" % (tag, nfe, nfe_o, float(x.abs().max()), e))
        assert nfe == nfe_o and e < 1e-5, (tag, nfe, nfe_o, e)
        out[tag + "_samples"] = x.numpy()
        out[tag + "_nfe"] = np.int64(nfe)
        if not denoise:
            x2, nfe2 = smp(ref_model, z=z.clone() * (1 + 1e-6))
            out["ode_sens"] = np.float64(rel(x2, x))
            print("             conditioning: input x (1 + 1e-6) moves the samples by %.3g relative, nfe %d" % (out["ode_sens"], nfe2))

    # ---- likelihood (likelihood.py:69-111) with the Hutchinson probe injected
    real = torch.randint_like
    torch.randint_like = lambda t, low=0, high=2, **k: ((epsilon + 1.) / 2.).to(t.device)
    try:
        lf = ref_likelihood.get_likelihood_fn(sde, inv, hutchinson_type="Rademacher", rtol=case["rtol"], atol=case["atol"],
                                              method="RK45", eps=case["lik_eps"])
        bpd, zz, nfe = lf(ref_model, data.clone())
        bpd2, zz2, nfe2 = lf(ref_model, data.clone() * (1 + 1e-6))
        # one evaluation of the augmented right-hand side with the reference's own closures (likelihood.py:26-37, 59-67)
        t_probe = torch.ones(shape[0]) * case["t_probe"]
        score_fn = ref_mutils.get_score_fn(sde, ref_model, train=False, continuous=True)
        drift_fn = lambda xx, tt: sde.reverse(score_fn, probability_flow=True).sde(xx, tt)[0]   # noqa: E731
        with torch.no_grad():
            d0 = drift_fn(data.clone(), t_probe)
            div0 = ref_likelihood.get_div_fn(drift_fn)(data.clone(), t_probe, epsilon)
    finally:
        torch.randint_like = real
    bo, zo, nfe_o = ode_oracle.likelihood(cfg, full_sd, "subvpsde", kw, data.clone(), epsilon, inv, rtol=case["rtol"],
                                          atol=case["atol"], eps=case["lik_eps"])
    do, divo = ode_oracle.rhs_augmented(cfg, full_sd, ode_oracle.S.make_sde("subvpsde", **kw), data.clone(), t_probe, epsilon)
    print("likelihood   nfe %d (oracle %d)  bpd %s  oracle-vs-reference: bpd %.3g  z %.3g  drift %.3g  div %.3g" % (
        nfe, nfe_o, bpd.tolist(), rel(bo, bpd), rel(zo, zz), rel(do, d0), rel(divo, div0)))
    assert nfe == nfe_o and rel(bo, bpd) < 1e-5 and rel(zo, zz) < 1e-4 and rel(do, d0) < 1e-5 and rel(divo, div0) < 1e-4
    out.update(lik_bpd=bpd.numpy(), lik_z=zz.numpy(), lik_nfe=np.int64(nfe), rhs_drift=d0.numpy(), rhs_div=div0.numpy(),
               lik_sens_bpd=np.float64(rel(bpd2, bpd)), lik_sens_z=np.float64(rel(zz2, zz)), lik_sens_nfe=np.int64(nfe2))
    print("             conditioning: input x (1 + 1e-6) moves bpd by %.3g, the latent by %.3g relative, nfe %d" % (
        out["lik_sens_bpd"], out["lik_sens_z"], nfe2))
    path = os.path.join(G.ROOT, "tests", "golden", "ode_small.npz")
    np.savez_compressed(path, **out)
    print("written", path)


if __name__ == "__main__":
    main()
