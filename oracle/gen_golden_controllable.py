"""Generate tests/golden/controllable_small.npz by running the REFERENCE's controllable_generation.py
(get_pc_inpainter :8-83, get_pc_colorizer :86-180) on CPU (build container only; needs /root/reference).

Down-sized configs, seeded weights, injected prior sample and noise (torch.randn_like patched);
the script also asserts that oracle/sampler_oracle.{inpaint,colorize} reproduce every entry.
TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G                                  # noqa: E402
import _util                                            # noqa: E402


def main():
    G.import_reference()
    from oracle import sampler_oracle
    import models.utils as ref_mutils            # noqa  (reference)
    import models.ncsnpp                         # noqa
    import sde_lib as ref_sde_lib                # noqa
    import sampling as ref_sampling              # noqa
    import controllable_generation as ref_cg     # noqa
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
    B, R = _util.PC_VARIANT_BATCH, _util.PC_VARIANT_SIZE
    out = {}
    for name, (task, variant, pred, corr) in _util.CONTROLLABLE_CASES.items():
        kind, sde_kind, kw, _, _, _, continuous, _, _, eps = _util.PC_VARIANTS[variant]
        cfg = _util.small_config(kind)
        cfg.device = torch.device("cpu")
        torch.manual_seed(0)
        ref_model = ref_mutils.get_model("ncsnpp")(ref_cfg_like(cfg)).eval()
        sd = _util.fix_top_level_groupnorm(_util.seeded_state_dict(ref_model, seed=1), ref_model)
        ref_model.load_state_dict(sd, strict=False)
        full_sd = dict(sd); full_sd["sigmas"] = ref_model.sigmas
        sde = {"vesde": ref_sde_lib.VESDE, "vpsde": ref_sde_lib.VPSDE, "subvpsde": ref_sde_lib.subVPSDE}[sde_kind](**kw)
        N = kw["N"]
        data, mask, prior, noises = _util.controllable_inputs(name, B, N, R, kw.get("sigma_max", 1.0))
        seq = []
        for i in range(N):                         # randn_like call order of one iteration (controllable_generation.py:44-52,78-80)
            if corr != "none":
                seq.append(noises[i, 0])
            seq.append(noises[i, 2])
            if pred != "none":
                seq.append(noises[i, 1])
            seq.append(noises[i, 3])
        it = iter(seq)
        real_randn_like = torch.randn_like
        torch.randn_like = lambda t, **k: next(it).to(t.device)
        sde.prior_sampling = lambda shape: prior.clone()
        args = (sde, ref_sampling.get_predictor(pred), ref_sampling.get_corrector(corr), lambda v: v)
        kws = dict(snr=0.16, n_steps=1, probability_flow=False, continuous=continuous, denoise=True, eps=eps)
        try:
            if task == "inpaint":
                ref = ref_cg.get_pc_inpainter(*args, **kws)(ref_model, data, mask)
            else:
                ref = ref_cg.get_pc_colorizer(*args, **kws)(ref_model, data)
        finally:
            torch.randn_like = real_randn_like
        assert next(it, None) is None, "noise sequence not consumed as expected"
        assert torch.isfinite(ref).all()
        okw = dict(snr=0.16, n_steps=1, eps=eps, denoise=True, predictor=pred, corrector=corr, continuous=continuous)
        if task == "inpaint":
            orc = sampler_oracle.inpaint(cfg, full_sd, sde_kind, kw, data, mask, prior, noises, **okw)
        else:
            orc = sampler_oracle.colorize(cfg, full_sd, sde_kind, kw, data, prior, noises, **okw)
        err = float((orc["samples"] - ref).abs().max() / ref.abs().max())
        print("%-28s |x| max %.4g  oracle-vs-reference rel err %.3g" % (name, float(ref.abs().max()), err))
        assert err < 1e-4, (name, err)
        out[name] = ref.numpy()
    path = os.path.join(G.ROOT, "tests", "golden", "controllable_small.npz")
    np.savez_compressed(path, **out)
    print("written", path)


if __name__ == "__main__":
    main()
