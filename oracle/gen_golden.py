"""Generate tests/golden/*.npz by running the REFERENCE implementation (build container only).

    python oracle/gen_golden.py            # needs /root/reference; writes tests/golden/

The reference has no tests or golden vectors of its own, so parity is pinned to the
reference *implementation*: it is imported from a scratch copy (its op/ package JIT-builds into its
own directory), fed deterministic weights (tests/_util.seeded_state_dict) and seeded
inputs, and its outputs are stored here.  The same script asserts that oracle/ reproduces the
reference, and that score_sde_pytorch_amd.configs presets equal the reference's config files.
The GPU box has no /root/reference: tests there use only the committed .npz files.
"""
import os
import shutil
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_SRC = "/root/reference"
SCRATCH = "/tmp/ssde_refcopy"
SHIM = "/tmp/ssde_refshim"


def import_reference():
    if not os.path.isdir(REF_SRC):
        raise SystemExit("reference not present at %s (this script only runs in the build container)" % REF_SRC)
    if not os.path.isdir(SCRATCH):
        shutil.copytree(REF_SRC, SCRATCH)
        os.system("chmod -R u+w %s" % SCRATCH)
    os.makedirs(os.path.join(SHIM, "ml_collections"), exist_ok=True)
    with open(os.path.join(SHIM, "ml_collections", "__init__.py"), "w") as f:
        f.write("class ConfigDict(dict):\n"
                "    def __getattr__(self, k):\n"
                "        try: return self[k]\n"
                "        except KeyError: raise AttributeError(k)\n"
                "    def __setattr__(self, k, v): self[k] = v\n")
    os.environ.setdefault("TORCH_EXTENSIONS_DIR", "/tmp/torch_ext")
    sys.dont_write_bytecode = True
    sys.path[:0] = [SHIM, SCRATCH]


def main():
    import_reference()
    import importlib
    import _util
    from oracle import unet_oracle, sampler_oracle
    from score_sde_pytorch_amd import configs as my_cfgs
    import models.utils as ref_mutils            # noqa  (reference)
    import models.ncsnpp                         # noqa  registers 'ncsnpp' in the reference registry
    import sde_lib as ref_sde_lib                # noqa
    import sampling as ref_sampling              # noqa

    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    torch.set_num_threads(os.cpu_count())

    # ---- 1. config presets equal the reference's files
    for name in ["ve/cifar10_ncsnpp_continuous", "ve/cifar10_ncsnpp_deep_continuous", "subvp/cifar10_ddpmpp_continuous",
                 "vp/cifar10_ddpmpp_continuous", "ve/ffhq_256_ncsnpp_continuous"]:
        ref = importlib.import_module("configs." + name.replace("/", ".")).get_config()
        mine = my_cfgs.get_config(name)
        for sec in ["training", "sampling", "eval", "data", "model", "optim"]:
            for k, v in ref[sec].items():
                if k == "tfrecords_path":
                    continue
                mv = mine[sec][k]
                same = (tuple(v) == tuple(mv)) if isinstance(v, (list, tuple)) else (v == mv)
                assert same, (name, sec, k, v, mv)
        print("config preset ok:", name)

    def ref_cfg_like(cfg):
        """Reference ConfigDict carrying the values of one of our configs."""
        import ml_collections
        def conv(v):
            if hasattr(v, "items"):
                d = ml_collections.ConfigDict()
                for k, x in v.items():
                    d[k] = conv(x)
                return d
            return v
        return conv(cfg)

    cases = {
        "unet_small_ncsnpp": (_util.small_config("ncsnpp"), 3, "sigma"),
        "unet_small_ddpmpp": (_util.small_config("ddpmpp"), 3, "label"),
        "unet_small_ffhq": (_util.small_config("ffhq", image_size=32, ch_mult=(1, 1, 2), attn=(16,)), 2, "sigma"),
        "unet_small_ncsnpp_3lvl": (_util.small_config("ncsnpp", image_size=32, ch_mult=(1, 2, 2), num_res_blocks=2, attn=(16,)), 2, "sigma"),
        "unet_cifar_ncsnpp": (my_cfgs.get_config("ve/cifar10_ncsnpp_continuous"), 2, "sigma"),
        "unet_cifar_ddpmpp": (my_cfgs.get_config("subvp/cifar10_ddpmpp_continuous"), 2, "label"),
    }
    for name, (cfg, batch, cond_kind) in cases.items():
        cfg.device = torch.device("cpu")
        rcfg = ref_cfg_like(cfg)
        torch.manual_seed(0)
        ref_model = ref_mutils.get_model("ncsnpp")(rcfg).eval()
        sd = _util.fix_top_level_groupnorm(_util.seeded_state_dict(ref_model, seed=1), ref_model)
        missing = ref_model.load_state_dict(sd, strict=False)
        assert set(missing.missing_keys) <= {"sigmas"} and not missing.unexpected_keys, missing
        g = torch.Generator().manual_seed(123)
        R = cfg.data.image_size
        x = torch.rand(batch, 3, R, R, generator=g) if not cfg.data.centered else torch.rand(batch, 3, R, R, generator=g) * 2 - 1
        if cond_kind == "sigma":
            cond = torch.exp(torch.rand(batch, generator=g) * (np.log(50.0) - np.log(0.01)) + np.log(0.01)).float()
            x = x + cond[:, None, None, None] * torch.randn(batch, 3, R, R, generator=g)
        else:
            cond = (torch.rand(batch, generator=g) * 999).float()
        with torch.no_grad():
            y_ref = ref_model(x, cond)
            full_sd = dict(sd); full_sd["sigmas"] = ref_model.sigmas
            y_orc = unet_oracle.ncsnpp_forward(cfg, full_sd, x, cond)
        err = float((y_ref - y_orc).abs().max() / y_ref.abs().max())
        print("%-28s out absmax %.4g  oracle-vs-reference rel err %.3g" % (name, float(y_ref.abs().max()), err))
        assert err < 2e-5, err
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), x=x.numpy(), cond=cond.numpy(), y=y_ref.numpy())

    # ---- 3. PC sampler trajectory (BASELINE config #1: B=8, VESDE N=10, reverse_diffusion + langevin)
    cfg = my_cfgs.get_config("ve/cifar10_ncsnpp_continuous")
    cfg.device = torch.device("cpu")
    rcfg = ref_cfg_like(cfg)
    ref_model = ref_mutils.get_model("ncsnpp")(rcfg).eval()
    sd = _util.fix_top_level_groupnorm(_util.seeded_state_dict(ref_model, seed=1), ref_model)
    ref_model.load_state_dict(sd, strict=False)
    full_sd = dict(sd); full_sd["sigmas"] = ref_model.sigmas
    B, N = 8, 10
    sde = ref_sde_lib.VESDE(sigma_min=0.01, sigma_max=50, N=N)
    x_T, noises = _util.pc_case_inputs(B, N)

    # reference run with injected noise: patch torch.randn_like / prior_sampling
    it = iter(noises.reshape(2 * N, B, 3, 32, 32))
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: next(it).to(t.device)
    sde.prior_sampling = lambda shape: x_T.clone()
    traj_x, traj_norm = [], []
    real_pred = ref_sampling.shared_predictor_update_fn

    def spy_pred(x, t, **kw):
        xn, xm = real_pred(x, t, **kw)
        traj_x.append(xn.clone())
        return xn, xm
    ref_sampling.shared_predictor_update_fn = spy_pred
    # ||score|| per function evaluation (batch mean of the per-sample norm) of the REFERENCE run: for the continuous VE
    # score_fn the score is the network output itself (models/utils.py:166-176), so a forward hook sees it
    ref_norms = []
    hook = ref_model.register_forward_hook(
        lambda mod, inp, outp: ref_norms.append(float(torch.norm(outp.reshape(outp.shape[0], -1), dim=-1).mean())))
    try:
        sampler = ref_sampling.get_pc_sampler(sde, (B, 3, 32, 32), ref_sampling.ReverseDiffusionPredictor,
                                              ref_sampling.LangevinCorrector, lambda v: v, snr=0.16, n_steps=1,
                                              probability_flow=False, continuous=True, denoise=True, eps=1e-5, device="cpu")
        # get_pc_sampler bound the original function through functools.partial at creation: rebuild after patching
        samples_ref, nfe = sampler(ref_model)
    finally:
        torch.randn_like = real_randn_like
        ref_sampling.shared_predictor_update_fn = real_pred
        hook.remove()
    assert nfe == 2 * N and len(ref_norms) == 2 * N
    out = sampler_oracle.pc_sample(cfg, full_sd, sde_kind="vesde", sde_kwargs=dict(sigma_min=0.01, sigma_max=50, N=N),
                                   x_T=x_T, noises=noises, snr=0.16, n_steps=1, eps=1e-5, denoise=True)
    err = float((out["samples"] - samples_ref).abs().max() / samples_ref.abs().max())
    print("pc_sampler oracle-vs-reference rel err %.3g (|x| max %.3g)" % (err, float(samples_ref.abs().max())))
    assert err < 1e-4, err
    # per-step parity of the oracle against the reference trajectory (spy on the predictor update)
    assert len(traj_x) == N
    for i in range(N):
        e = float((out["x_steps"][i] - traj_x[i]).abs().max() / traj_x[i].abs().max())
        assert e < 1e-4, (i, e)
    ne = max(abs(a - b) / b for a, b in zip(out["score_norms"], ref_norms))
    print("score-norm trajectory oracle-vs-reference rel err %.3g" % ne)
    assert ne < 1e-5, ne
    # x_T and the noises are regenerated from the seed by the tests (recipe: tests/_util.pc_case_inputs)
    np.savez_compressed(os.path.join(out_dir, "pc_cifar_ncsnpp_n10.npz"), samples=samples_ref.numpy(),
                        score_norms=np.asarray(ref_norms, dtype=np.float64),         # from the reference run (the oracle's agree, below)
                        x_step0=traj_x[0].numpy(), x_step4=traj_x[4].numpy(), x_step9=traj_x[9].numpy())
    print("golden vectors written to", out_dir)


if __name__ == "__main__":
    main()
