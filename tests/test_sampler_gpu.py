"""PC-sampler parity on the GPU (BASELINE config #1: configs/ve/cifar10_ncsnpp_continuous, B=8,
VESDE N=10, reverse_diffusion + langevin, snr 0.16, eps 1e-5, denoise) with INJECTED noise
(SURVEY F9), against the trajectory of the reference implementation stored in
tests/golden/pc_cifar_ncsnpp_n10.npz.

Stated tolerances (north_star: pixel MSE and score-norm trajectory):
  pixel MSE per step  <= 1e-12 * max|x|^2, max-abs error <= 5e-6 max|x| (_util.TRAJ_*)     score-norm trajectory <= 1e-4 relative
"""
import os

import numpy as np
import pytest
import torch

import _util
from _util import rel_err

pytestmark = pytest.mark.gpu


def _setup(n_steps_sde=10, batch=8):
    from score_sde_pytorch_amd import sde_lib, sampling
    from score_sde_pytorch_amd.models import utils as mutils
    cfg = _util.cfgs.get_config("ve/cifar10_ncsnpp_continuous")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.cuda().eval()
    sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=50, N=n_steps_sde)
    sampler = sampling.get_pc_sampler(sde, (batch, 3, 32, 32), sampling.ReverseDiffusionPredictor,
                                      sampling.LangevinCorrector, lambda v: v, snr=0.16, n_steps=1,
                                      probability_flow=False, continuous=True, denoise=True, eps=1e-5, device="cuda")
    return cfg, model, sde, sampler


# kernel modes of the 3x3 convolutions (engine.Lowering.wino_ok): "2" = Winograd F(2x2,3x3) wherever legal (conftest's
# default), "4" = F(4x4,3x3) wherever legal -- the kernel that carries 60 % of the benchmarked PC iteration and rounds ~5x
# coarser -- and "0" = the direct (bitwise fmaf-chain) kernel
WINO_MODES = ["2", "4", "0"]


@pytest.mark.parametrize("wino", WINO_MODES)
def test_fused_pc_sampler_matches_reference_trajectory(wino, monkeypatch):
    monkeypatch.setenv("SSDE_WINOGRAD", wino)
    gold = np.load(os.path.join(_util.GOLDEN, "pc_cifar_ncsnpp_n10.npz"))
    cfg, model, sde, sampler = _setup()
    x_T, noises = _util.pc_case_inputs(8, 10)
    # full run
    samples, nfe = sampler(model, x_init=x_T, noises=noises)
    assert nfe == 20 and sampler.last_path == "fused-eager"
    ref = torch.from_numpy(gold["samples"])
    _util.assert_trajectory_close(samples, ref, "samples")
    # intermediate states: re-run truncated
    for k, name in [(1, "x_step0"), (5, "x_step4"), (10, "x_step9")]:
        sampler(model, x_init=x_T, noises=noises, max_steps=k)
        x_k = sampler.engine.x.view(8, 3, 32, 32).cpu()
        r = torch.from_numpy(gold[name])
        _util.assert_trajectory_close(x_k, r, name)


@pytest.mark.parametrize("wino", WINO_MODES)
def test_score_norm_trajectory(wino, monkeypatch):
    """||score|| per function evaluation, batch mean, vs the reference run (north_star's second yardstick)."""
    monkeypatch.setenv("SSDE_WINOGRAD", wino)
    gold = np.load(os.path.join(_util.GOLDEN, "pc_cifar_ncsnpp_n10.npz"))
    cfg, model, sde, sampler = _setup()
    x_T, noises = _util.pc_case_inputs(8, 10)
    sampler(model, x_init=x_T, noises=noises, max_steps=0)          # builds the engine, loads x_T
    eng = sampler.engine
    eng.reset(x_T.cuda())
    prog = eng.step_program(with_rng=False)
    norms = []
    # run op by op so the score can be read after each U-Net evaluation
    from score_sde_pytorch_amd import _lib as L
    import ctypes as C
    lib = L.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nz = noises.cuda()
    for i in range(10):
        eng.z_c.copy_(nz[i, 0].reshape(-1)); eng.z_p.copy_(nz[i, 1].reshape(-1))
        for j in range(prog.n):
            op = prog.ops[j]
            L.check(lib.ssde_program_run(C.byref(op), 1, st))
            if op.kind == L.OP_TO_NCHW:
                s = eng.unet.output_view()
                norms.append(float(torch.norm(s.reshape(8, -1), dim=-1).mean()))
    ref = gold["score_norms"]
    assert len(norms) == len(ref) == 20
    assert np.max(np.abs(np.array(norms) - ref) / ref) < 1e-4


def test_generic_path_equals_fused_path():
    """a user-registered (non-stock) predictor goes through the generic loop; wrapping the stock update must
    reproduce the fused result -- same kernels for the score, torch ops for the update."""
    from score_sde_pytorch_amd import sampling
    cfg, model, sde, sampler = _setup()
    x_T, noises = _util.pc_case_inputs(8, 10)

    class MyPredictor(sampling.ReverseDiffusionPredictor):
        pass

    generic = sampling.get_pc_sampler(sde, (8, 3, 32, 32), MyPredictor, sampling.LangevinCorrector, lambda v: v,
                                      snr=0.16, n_steps=1, continuous=True, denoise=True, eps=1e-5, device="cuda")
    a, _ = sampler(model, x_init=x_T, noises=noises, max_steps=3)
    b, _ = generic(model, x_init=x_T, noises=noises, max_steps=3)
    assert generic.last_path == "generic"
    assert rel_err(a, b) < 2e-5


def test_graph_replay_equals_eager_and_is_reproducible():
    """hipGraph replay with in-kernel rocRAND noise: same seed -> same samples; graph == eager launch order."""
    cfg, model, sde, sampler = _setup(n_steps_sde=6)
    x_T, _ = _util.pc_case_inputs(8, 6)
    a, _ = sampler(model, x_init=x_T, seed=11, use_graph=True)
    assert sampler.last_path == "fused-graph"
    b, _ = sampler(model, x_init=x_T, seed=11, use_graph=False)
    c, _ = sampler(model, x_init=x_T, seed=11, use_graph=True)
    d, _ = sampler(model, x_init=x_T, seed=12, use_graph=True)
    assert torch.isfinite(a).all()
    assert torch.equal(a, b) and torch.equal(a, c)
    assert not torch.equal(a, d)
    # one program / one captured graph serves every seed (the seed is a device word, not a baked constant)
    assert len(sampler.engine._programs) == 1


def test_default_noise_is_fresh_per_call_like_randn_like():
    """Without an explicit seed every call draws new noise, as the reference's torch.randn_like does
    (sampling.py:197,275); torch.manual_seed makes the sequence of calls reproducible."""
    cfg, model, sde, sampler = _setup(n_steps_sde=4)
    x_T, _ = _util.pc_case_inputs(8, 4)
    torch.manual_seed(5)
    a, _ = sampler(model, x_init=x_T)
    b, _ = sampler(model, x_init=x_T)
    torch.manual_seed(5)
    a2, _ = sampler(model, x_init=x_T)
    assert not torch.equal(a, b)
    assert torch.equal(a, a2)


def test_vp_euler_maruyama_fused_matches_oracle():
    """sub-VP / DDPM++ with euler_maruyama + none (the VP/sub-VP config default): fused tables vs CPU oracle."""
    from oracle import sampler_oracle
    from score_sde_pytorch_amd import sde_lib, sampling
    from score_sde_pytorch_amd.models import utils as mutils
    cfg = _util.small_config("ddpmpp")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    sd = dict(_util.load_seeded(model, seed=1)); sd["sigmas"] = model.sigmas.clone()
    model = model.cuda().eval()
    N, B = 8, 4
    sde = sde_lib.subVPSDE(beta_min=0.1, beta_max=20, N=N)
    g = torch.Generator().manual_seed(5)
    x_T = torch.randn(B, 3, 16, 16, generator=g)
    noises = torch.randn(N, 2, B, 3, 16, 16, generator=g)
    sampler = sampling.get_pc_sampler(sde, (B, 3, 16, 16), sampling.EulerMaruyamaPredictor, sampling.NoneCorrector,
                                      lambda v: v, snr=0.16, n_steps=1, continuous=True, denoise=True, eps=1e-3, device="cuda")
    out, nfe = sampler(model, x_init=x_T, noises=noises)
    assert sampler.last_path == "fused-eager"
    ref = sampler_oracle.pc_sample(cfg, sd, "subvpsde", dict(beta_min=0.1, beta_max=20, N=N), x_T, noises, snr=0.16,
                                   eps=1e-3, denoise=True, predictor="euler_maruyama", corrector="none")
    assert rel_err(out, ref["samples"]) < 2e-4


def _variant(name):
    from score_sde_pytorch_amd import sde_lib, sampling
    from score_sde_pytorch_amd.models import utils as mutils
    kind, sde_kind, kw, pred, corr, n_steps, continuous, pflow, denoise, eps = _util.PC_VARIANTS[name]
    cfg = _util.small_config(kind)
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.cuda().eval()
    sde = {"vesde": sde_lib.VESDE, "vpsde": sde_lib.VPSDE, "subvpsde": sde_lib.subVPSDE}[sde_kind](**kw)
    B, R = _util.PC_VARIANT_BATCH, _util.PC_VARIANT_SIZE
    x_T, noises = _util.pc_variant_inputs(name, B, kw["N"], R, kw.get("sigma_max", 1.0))
    mk = lambda p, c: sampling.get_pc_sampler(sde, (B, 3, R, R), p, c, lambda v: v, snr=0.16, n_steps=n_steps,
                                              probability_flow=pflow, continuous=continuous, denoise=denoise, eps=eps,
                                              device="cuda")
    return model, mk, sampling.get_predictor(pred), sampling.get_corrector(corr), x_T, noises


@pytest.mark.parametrize("name", list(_util.PC_VARIANTS))
def test_fused_variants_match_reference_samples(name):
    """ancestral / Euler-Maruyama / reverse-diffusion / none x ald / langevin / none on VE, VP (continuous and discrete
    labels) and sub-VP: the fused step program against the REFERENCE sampler's output (pc_small_variants.npz)"""
    gold = np.load(os.path.join(_util.GOLDEN, "pc_small_variants.npz"))
    model, mk, pred, corr, x_T, noises = _variant(name)
    sampler = mk(pred, corr)
    out, nfe = sampler(model, x_init=x_T, noises=noises)
    assert sampler.last_path == "fused-eager"
    assert rel_err(out, torch.from_numpy(gold[name])) < 2e-4


@pytest.mark.parametrize("name", ["ve_ancestral_ald", "vp_ancestral_ald_discrete"])
def test_generic_path_variants_match_reference_samples(name):
    """the same combinations through user subclasses (generic loop calling update_fn with our score_fn)"""
    gold = np.load(os.path.join(_util.GOLDEN, "pc_small_variants.npz"))
    model, mk, pred, corr, x_T, noises = _variant(name)
    sampler = mk(type("MyPredictor", (pred,), {}), type("MyCorrector", (corr,), {}))
    out, nfe = sampler(model, x_init=x_T, noises=noises)
    assert sampler.last_path == "generic"
    assert rel_err(out, torch.from_numpy(gold[name])) < 2e-4


def test_unusable_combinations_keep_reference_errors():
    from score_sde_pytorch_amd import sde_lib, sampling
    model, mk, pred, corr, x_T, noises = _variant("subvp_em_none")
    sub = sde_lib.subVPSDE(0.1, 20, N=4)
    bad = sampling.get_pc_sampler(sub, (4, 3, 16, 16), sampling.AncestralSamplingPredictor, sampling.NoneCorrector,
                                  lambda v: v, snr=0.16, continuous=True, device="cuda")
    with pytest.raises(NotImplementedError):
        bad(model, x_init=x_T)
    bad = sampling.get_pc_sampler(sub, (4, 3, 16, 16), sampling.NonePredictor, sampling.LangevinCorrector,
                                  lambda v: v, snr=0.16, continuous=True, device="cuda")
    with pytest.raises(AttributeError):
        bad(model, x_init=x_T)


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("name", list(_util.CONTROLLABLE_CASES))
def test_controllable_generation_matches_reference(name, generic):
    """inpainting / colorization: fused step program with ssde_project_update (and the generic loop for user
    subclasses) against the REFERENCE's controllable_generation.py output (controllable_small.npz)"""
    from score_sde_pytorch_amd import sde_lib, sampling, controllable_generation as cg
    from score_sde_pytorch_amd.models import utils as mutils
    gold = np.load(os.path.join(_util.GOLDEN, "controllable_small.npz"))
    task, variant, pred, corr = _util.CONTROLLABLE_CASES[name]
    kind, sde_kind, kw, _, _, _, continuous, _, _, eps = _util.PC_VARIANTS[variant]
    cfg = _util.small_config(kind)
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.cuda().eval()
    sde = {"vesde": sde_lib.VESDE, "vpsde": sde_lib.VPSDE, "subvpsde": sde_lib.subVPSDE}[sde_kind](**kw)
    data, mask, prior, noises = _util.controllable_inputs(name, _util.PC_VARIANT_BATCH, kw["N"], _util.PC_VARIANT_SIZE,
                                                          kw.get("sigma_max", 1.0))
    P, Cr = sampling.get_predictor(pred), sampling.get_corrector(corr)
    if generic:
        P, Cr = type("MyPredictor", (P,), {}), type("MyCorrector", (Cr,), {})
    kws = dict(snr=0.16, n_steps=1, probability_flow=False, continuous=continuous, denoise=True, eps=eps)
    if task == "inpaint":
        fn = cg.get_pc_inpainter(sde, P, Cr, lambda v: v, **kws)
        out = fn(model, data.cuda(), mask.cuda(), prior=prior, noises=noises)
    else:
        fn = cg.get_pc_colorizer(sde, P, Cr, lambda v: v, **kws)
        out = fn(model, data.cuda(), prior=prior, noises=noises)
    assert fn.last_path == ("generic" if generic else "fused-eager")
    assert rel_err(out, torch.from_numpy(gold[name])) < 2e-4


def test_inpainting_graph_replay_keeps_known_pixels():
    """hipGraph replay with in-kernel noise: at the last step (t = eps) the known half equals the data up to std(eps)"""
    from score_sde_pytorch_amd import sde_lib, sampling, controllable_generation as cg
    model, mk, pred, corr, x_T, noises = _variant("ve_em_langevin")
    sde = sde_lib.VESDE(0.01, 50, N=12)
    data, mask, prior, _ = _util.controllable_inputs("inpaint_ve_rd_langevin", 4, 1, 16, 50.0)
    fn = cg.get_pc_inpainter(sde, sampling.ReverseDiffusionPredictor, sampling.LangevinCorrector, lambda v: v, snr=0.16,
                             continuous=True, denoise=True, eps=1e-5)
    a = fn(model, data.cuda(), mask.cuda(), prior=prior, seed=3)
    b = fn(model, data.cuda(), mask.cuda(), prior=prior, seed=3)
    assert fn.last_path == "fused-graph" and torch.equal(a, b) and torch.isfinite(a).all()
    known = mask.cuda() > 0
    assert float((a - data.cuda())[known].abs().max()) < 1e-4       # x_mean on known pixels = marginal mean = data (VE)


def test_ode_sampler_on_device_matches_host_scipy(monkeypatch):
    """get_ode_sampler: the on-device RK45 driver and the reference's host scipy loop give the same samples and NFE"""
    import _util
    from score_sde_pytorch_amd import sde_lib, sampling
    from score_sde_pytorch_amd.models import utils as mutils
    cfg = _util.small_config("ddpmpp")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.cuda().eval()
    sde = sde_lib.subVPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales)
    shape = (2, 3, 16, 16)
    z = torch.randn(*shape, generator=torch.Generator().manual_seed(1)).cuda()
    smp = sampling.get_ode_sampler(sde, shape, lambda v: v, rtol=1e-4, atol=1e-4, eps=1e-3, device="cuda")
    x_dev, nfe_dev = smp(model, z=z.clone())
    monkeypatch.setenv("SSDE_HOST_ODE", "1")
    x_host, nfe_host = smp(model, z=z.clone())
    assert abs(nfe_dev - nfe_host) <= 12 and torch.isfinite(x_dev).all()
    assert _util.rel_err(x_dev, x_host) < 2e-3


def test_long_trajectory_error_growth_f4x4(monkeypatch):
    """50 predictor-corrector iterations (100 U-Net evaluations, Langevin batch-mean step size feeding back every
    iteration) with injected noise and every legal 3x3 layer on the F(4x4,3x3) kernel, against the CPU oracle's trajectory
    (oracle/sampler_oracle.pc_sample = /root/reference/sampling.py:390-409).  The 10-step golden above pins the start; this
    one shows that the coarser Winograd rounding does not accumulate: the per-step pixel MSE must stay under the
    bounds of _util.TRAJ_* (1e-12 max|x|^2, 5e-6 max|x|) at EVERY step, and the error may not grow by more than 4x from the first decade of
    steps to the last.  The per-step table goes to gpurun_out/ (copied to profiles/ by the builder)."""
    from oracle import sampler_oracle
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    N, B = 50, 4
    cfg, model, sde, sampler = _setup(n_steps_sde=N, batch=B)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x_T, noises = _util.pc_case_inputs(B, N)
    sampler(model, x_init=x_T, noises=noises, max_steps=0)          # builds the engine
    eng = sampler.engine
    from score_sde_pytorch_amd import _lib as L
    tiles = [eng.unet.program.ops[i].u.conv.tile for i in range(eng.unet.program.n)
             if eng.unet.program.ops[i].kind == L.OP_CONV and eng.unet.program.ops[i].u.conv.ksize == 3]
    assert sum(t in L.TILES_WINOGRAD4 for t in tiles) >= 40, tiles
    eng.unet.weights.refresh()
    eng.reset(x_T.cuda())
    prog = eng.step_program(with_rng=False)
    nz = noises.cuda()
    xs = []
    for i in range(N):
        eng.z_c.copy_(nz[i, 0].reshape(-1)); eng.z_p.copy_(nz[i, 1].reshape(-1))
        prog.run()
        xs.append(eng.x.view(B, 3, 32, 32).cpu().clone())
    ref = sampler_oracle.pc_sample(cfg, sd, "vesde", dict(sigma_min=0.01, sigma_max=50, N=N), x_T, noises, snr=0.16,
                                   n_steps=1, eps=1e-5, denoise=False)
    rows, rel = [], []
    for i in range(N):
        r = ref["x_steps"][i]
        mx = float(r.abs().max())
        mse = float(((xs[i].double() - r.double()) ** 2).mean())
        rel.append(float((xs[i] - r).abs().max()) / mx)
        rows.append("step %2d  max|x| %9.4f  pixel-MSE/max^2 %.3e  max-abs-err/max %.3e" % (i, mx, mse / mx ** 2, rel[-1]))
        assert mse <= _util.TRAJ_MSE * mx ** 2 and rel[-1] <= _util.TRAJ_MAXABS, rows[-1]
    out = os.path.join(os.path.dirname(_util.GOLDEN), "..", "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "f4x4_trajectory_error_growth.txt"), "w") as f:
            f.write("\n".join(rows) + "\n")
    assert np.mean(rel[-10:]) < 4 * max(np.mean(rel[:10]), 1e-6), (np.mean(rel[:10]), np.mean(rel[-10:]))
