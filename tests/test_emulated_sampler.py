"""The fused PC step program (pc_engine.FusedPCSampler: U-Net program + FILL / SUMSQ / LANGEVIN / PREDICTOR ops of the
C ABI) run under the CPU emulator of the HIP sources, against the samples of the REFERENCE sampler
(tests/golden/pc_small_variants.npz).  The same cases run on the MI355X in tests/test_sampler_gpu.py."""
import os
import types

import numpy as np
import pytest
import torch

import _util
from _util import rel_err
from emu import emulated, available

pytestmark = pytest.mark.skipif(not available(), reason="g++ not available for the kernel emulator")


@pytest.mark.parametrize("name", ["ve_ancestral_ald", "vp_ancestral_ald_discrete", "vp_em_langevin", "subvp_rd_none_pflow"])
def test_fused_step_program_variants(name):
    from score_sde_pytorch_amd import sde_lib, sampling, pc_engine
    from score_sde_pytorch_amd.models import utils as mutils
    gold = np.load(os.path.join(_util.GOLDEN, "pc_small_variants.npz"))
    kind, sde_kind, kw, pred, corr, n_steps, continuous, pflow, denoise, eps = _util.PC_VARIANTS[name]
    cfg = _util.small_config(kind)
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg).eval()
    _util.load_seeded(model, seed=1)
    sde = {"vesde": sde_lib.VESDE, "vpsde": sde_lib.VPSDE, "subvpsde": sde_lib.subVPSDE}[sde_kind](**kw)
    B, R = _util.PC_VARIANT_BATCH, _util.PC_VARIANT_SIZE
    x_T, noises = _util.pc_variant_inputs(name, B, kw["N"], R, kw.get("sigma_max", 1.0))
    with emulated():
        plan = pc_engine.plan_fused(sde, sampling.get_predictor(pred), sampling.get_corrector(corr), model, continuous,
                                    types.SimpleNamespace(is_cuda=True), pflow)   # only the device kind of x is inspected
        assert plan is not None and plan["predictor"] == pred and plan["corrector"] == corr
        eng = pc_engine.FusedPCSampler(model, sde, plan, (B, 3, R, R), snr=0.16, n_steps=n_steps, probability_flow=pflow,
                                       eps=eps, device=torch.device("cpu"))
        x, x_mean = eng.run(x_T, noises=noises)
    out = x_mean if denoise else x
    assert rel_err(out, torch.from_numpy(gold[name])) < 2e-4


def test_conditioning_chain_evaluated_once_per_iteration(monkeypatch):
    """SSDE_PC_SHARE_COND=1: the evaluations of one PC iteration run at one noise level, so the second one skips the network's
    conditioning chain (embedding, temb MLP, Dense_0 projections: persistent buffers) -- fewer launches, the same bits."""
    from score_sde_pytorch_amd import sde_lib, sampling, pc_engine
    from score_sde_pytorch_amd.models import utils as mutils
    name = "vp_em_langevin"
    kind, sde_kind, kw, pred, corr, n_steps, continuous, pflow, denoise, eps = _util.PC_VARIANTS[name]
    cfg = _util.small_config(kind)
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg).eval()
    _util.load_seeded(model, seed=1)
    sde = sde_lib.VPSDE(**kw)
    B, R = _util.PC_VARIANT_BATCH, _util.PC_VARIANT_SIZE
    x_T, noises = _util.pc_variant_inputs(name, B, kw["N"], R, 1.0)
    outs, counts = {}, {}
    with emulated():
        for share in ("0", "1"):
            monkeypatch.setenv("SSDE_PC_SHARE_COND", share)
            plan = pc_engine.plan_fused(sde, sampling.get_predictor(pred), sampling.get_corrector(corr), model, continuous,
                                        types.SimpleNamespace(is_cuda=True), pflow)
            eng = pc_engine.FusedPCSampler(model, sde, plan, (B, 3, R, R), snr=0.16, n_steps=n_steps, probability_flow=pflow,
                                           eps=eps, device=torch.device("cpu"))
            outs[share] = eng.run(x_T, noises=noises, max_steps=3)
            counts[share] = (eng.step_program(with_rng=False).n, eng.unet.cond_only_ops, eng.nfe_per_step())
    n_cond, nfe = counts["1"][1], counts["1"][2]
    assert n_cond >= 3 and nfe >= 2 and counts["0"][0] - counts["1"][0] == n_cond * (nfe - 1), counts
    assert torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1])


@pytest.mark.parametrize("name", ["inpaint_vp_ancestral_none", "colorize_ve_rd_langevin"])
def test_fused_controllable_generation(name, monkeypatch):
    """get_pc_inpainter / get_pc_colorizer: the step program with ssde_project_update after the corrector and the
    predictor blocks, against the REFERENCE's controllable_generation.py output (controllable_small.npz)"""
    from score_sde_pytorch_amd import sde_lib, sampling, pc_engine, controllable_generation as cg
    from score_sde_pytorch_amd.models import utils as mutils
    gold = np.load(os.path.join(_util.GOLDEN, "controllable_small.npz"))
    task, variant, pred, corr = _util.CONTROLLABLE_CASES[name]
    kind, sde_kind, kw, _, _, _, continuous, _, _, eps = _util.PC_VARIANTS[variant]
    cfg = _util.small_config(kind)
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg).eval()
    _util.load_seeded(model, seed=1)
    sde = {"vesde": sde_lib.VESDE, "vpsde": sde_lib.VPSDE, "subvpsde": sde_lib.subVPSDE}[sde_kind](**kw)
    data, mask, prior, noises = _util.controllable_inputs(name, _util.PC_VARIANT_BATCH, kw["N"], _util.PC_VARIANT_SIZE,
                                                          kw.get("sigma_max", 1.0))
    real_plan = pc_engine.plan_fused
    monkeypatch.setattr(pc_engine, "plan_fused",         # host tensors are only acceptable under the emulator
                        lambda sde_, p, c, m, cont, x, pf=False: real_plan(sde_, p, c, m, cont, types.SimpleNamespace(is_cuda=True), pf))
    args = (sde, sampling.get_predictor(pred), sampling.get_corrector(corr), lambda v: v)
    kws = dict(snr=0.16, n_steps=1, probability_flow=False, continuous=continuous, denoise=True, eps=eps)
    with emulated():
        if task == "inpaint":
            fn = cg.get_pc_inpainter(*args, **kws)
            out = fn(model, data, mask, prior=prior, noises=noises)
        else:
            fn = cg.get_pc_colorizer(*args, **kws)
            out = fn(model, data, prior=prior, noises=noises)
    assert fn.last_path == "fused-eager"
    assert rel_err(out, torch.from_numpy(gold[name])) < 2e-4
