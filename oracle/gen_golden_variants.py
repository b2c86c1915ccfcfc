"""Generate tests/golden/pc_small_variants.npz by running the REFERENCE sampler (build container only).

    python oracle/gen_golden_variants.py     # needs /root/reference; writes tests/golden/pc_small_variants.npz

One entry per stock predictor / corrector / SDE combination that the fused sampler lowers
(score_sde_pytorch_amd/pc_engine.py) beyond the BASELINE reverse_diffusion + langevin pair of
gen_golden.py: the reference's `get_pc_sampler` (sampling.py:355-411) runs on CPU with the down-sized
configs of tests/_util.small_config, seeded weights and injected noise (torch.randn_like patched);
the script also asserts that oracle/sampler_oracle.pc_sample reproduces every entry.
TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G                                  # noqa: E402  (sets sys.path for tests/ and the repo root)

import _util                                            # noqa: E402
VARIANTS = _util.PC_VARIANTS
BATCH, SIZE = _util.PC_VARIANT_BATCH, _util.PC_VARIANT_SIZE


def variant_inputs(name, n_steps_sde, sigma_max):
    """x_T and noises[i, 0|1] of a variant (recipe shared with the tests through tests/_util.pc_variant_inputs)."""
    import _util
    return _util.pc_variant_inputs(name, BATCH, n_steps_sde, SIZE, sigma_max)


def main():
    G.import_reference()
    import _util
    from oracle import sampler_oracle
    import models.utils as ref_mutils            # noqa  (reference)
    import models.ncsnpp                         # noqa
    import sde_lib as ref_sde_lib                # noqa
    import sampling as ref_sampling              # noqa
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
    out = {}
    for name, (kind, sde_kind, kw, pred, corr, n_steps, continuous, pflow, denoise, eps) in VARIANTS.items():
        cfg = _util.small_config(kind)
        cfg.device = torch.device("cpu")
        torch.manual_seed(0)
        ref_model = ref_mutils.get_model("ncsnpp")(ref_cfg_like(cfg)).eval()
        sd = _util.fix_top_level_groupnorm(_util.seeded_state_dict(ref_model, seed=1), ref_model)
        ref_model.load_state_dict(sd, strict=False)
        full_sd = dict(sd); full_sd["sigmas"] = ref_model.sigmas
        sde = {"vesde": ref_sde_lib.VESDE, "vpsde": ref_sde_lib.VPSDE, "subvpsde": ref_sde_lib.subVPSDE}[sde_kind](**kw)
        N = kw["N"]
        x_T, noises = variant_inputs(name, N, kw.get("sigma_max", 1.0))
        seq = []
        for i in range(N):
            if corr != "none":
                seq += [noises[i, 0]] * n_steps
            if pred != "none":
                seq.append(noises[i, 1])
        it = iter(seq)
        real_randn_like = torch.randn_like
        torch.randn_like = lambda t, **k: next(it).to(t.device)
        sde.prior_sampling = lambda shape: x_T.clone()
        try:
            sampler = ref_sampling.get_pc_sampler(sde, (BATCH, 3, SIZE, SIZE), ref_sampling.get_predictor(pred),
                                                  ref_sampling.get_corrector(corr), lambda v: v, snr=0.16, n_steps=n_steps,
                                                  probability_flow=pflow, continuous=continuous, denoise=denoise, eps=eps,
                                                  device="cpu")
            samples_ref, nfe = sampler(ref_model)
        finally:
            torch.randn_like = real_randn_like
        assert next(it, None) is None, "noise sequence not consumed as expected"
        assert torch.isfinite(samples_ref).all()
        orc = sampler_oracle.pc_sample(cfg, full_sd, sde_kind, kw, x_T, noises, snr=0.16, n_steps=n_steps, eps=eps,
                                       denoise=denoise, predictor=pred, corrector=corr, continuous=continuous,
                                       probability_flow=pflow)
        err = float((orc["samples"] - samples_ref).abs().max() / samples_ref.abs().max())
        print("%-28s |x| max %.4g  nfe %d  oracle-vs-reference rel err %.3g" % (name, float(samples_ref.abs().max()), nfe, err))
        assert err < 1e-4, (name, err)
        out[name] = samples_ref.numpy()
    path = os.path.join(G.ROOT, "tests", "golden", "pc_small_variants.npz")
    np.savez_compressed(path, **out)
    print("written", path)


if __name__ == "__main__":
    main()
