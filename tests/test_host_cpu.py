"""CPU: host-side logic -- C-ABI library loads and exports every declared symbol, ctypes mirror matches the
header, plugin registries behave like the reference's, engines lower every BASELINE config (dry), step tables
follow the reference's formulas, and the HIP entry points fail loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import _util


def test_library_exports_every_header_symbol():
    from score_sde_pytorch_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(_util.ROOT, "include", "ssde.h")).read()
    declared = set(re.findall(r"\b(ssde_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert set(_lib.EXPORTS) == declared
    assert lib.ssde_abi_version() == _lib.ABI_VERSION
    assert lib.ssde_sizeof_op() == C.sizeof(_lib.Op)


def test_conv_plan_validation_errors_are_reported():
    from score_sde_pytorch_amd import _lib as L
    lib = L.load()
    a = L.ConvArgs()
    a.dst = 0x1000
    assert lib.ssde_conv_lds_bytes(C.byref(a)) < 0
    assert b"neither" in lib.ssde_last_error()
    a.main.p0, a.main.c0, a.w_main, a.ksize, a.stride, a.pad = 0x1000, 6, 0x1000, 3, 1, 1
    a.n, a.h_in, a.w_in, a.h_out, a.w_out, a.c_out = 1, 8, 8, 8, 8, 8
    assert lib.ssde_conv_lds_bytes(C.byref(a)) < 0 and b"multiples of 4" in lib.ssde_last_error()
    a.main.c0, a.h_out = 8, 9
    assert lib.ssde_conv_lds_bytes(C.byref(a)) < 0 and b"larger than" in lib.ssde_last_error()
    a.h_out, a.stride = 8, 3
    assert lib.ssde_conv_lds_bytes(C.byref(a)) < 0 and b"stride" in lib.ssde_last_error()


def test_registries_match_reference_behaviour():
    from score_sde_pytorch_amd import sampling
    from score_sde_pytorch_amd.models import utils as mutils
    assert sampling.get_predictor("reverse_diffusion") is sampling.ReverseDiffusionPredictor
    assert sampling.get_corrector("langevin") is sampling.LangevinCorrector
    assert set(["euler_maruyama", "reverse_diffusion", "ancestral_sampling", "none"]) <= set(sampling._PREDICTORS)
    assert set(["langevin", "ald", "none"]) <= set(sampling._CORRECTORS)
    with pytest.raises(ValueError, match="Already registered"):
        @sampling.register_predictor(name="none")
        class _P(sampling.Predictor):       # noqa
            def update_fn(self, x, t):
                return x, x
    with pytest.raises(ValueError, match="Already registered"):
        @mutils.register_model(name="ncsnpp")
        class _M:                             # noqa
            pass

    @sampling.register_corrector(name="unit_test_corrector")
    class _C(sampling.Corrector):
        def update_fn(self, x, t):
            return x, x
    assert sampling.get_corrector("unit_test_corrector") is _C
    del sampling._CORRECTORS["unit_test_corrector"]


def test_score_fn_rejects_unknown_sde_and_cpu_model_input():
    from score_sde_pytorch_amd import sde_lib
    from score_sde_pytorch_amd.models import utils as mutils

    class Odd(sde_lib.SDE):
        T = 1
        def sde(self, x, t): return x, t
        def marginal_prob(self, x, t): return x, t
        def prior_sampling(self, shape): return torch.zeros(shape)
        def prior_logp(self, z): return z
    with pytest.raises(NotImplementedError, match="not yet supported"):
        mutils.get_score_fn(Odd(10), None)
    model = mutils.get_model("ncsnpp")(_util.small_config("ncsnpp"))
    score_fn = mutils.get_score_fn(sde_lib.VESDE(N=10), model, continuous=True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        score_fn(torch.zeros(1, 3, 16, 16), torch.ones(1))


def test_sde_lib_matches_closed_forms():
    from score_sde_pytorch_amd import sde_lib
    t = torch.tensor([0.0, 0.3, 1.0])
    x = torch.ones(3, 1, 2, 2)
    ve = sde_lib.VESDE(0.01, 50, N=1000)
    assert torch.allclose(ve.marginal_prob(x, t)[1], torch.tensor([0.01, 0.01 * 5000 ** 0.3, 50.0]), rtol=1e-5)
    f, G = ve.discretize(x, torch.tensor([0.0, 0.5, 1.0]))
    s = ve.discrete_sigmas
    assert float(f.abs().max()) == 0 and torch.allclose(G[0], s[0]) and torch.allclose(G[2], torch.sqrt(s[999] ** 2 - s[998] ** 2))
    vp = sde_lib.VPSDE(0.1, 20, N=1000)
    mean, std = vp.marginal_prob(x, t)
    assert torch.allclose(mean[:, 0, 0, 0] ** 2 + std ** 2, torch.ones(3), atol=1e-6)
    sub = sde_lib.subVPSDE(0.1, 20, N=1000)
    _, std_sub = sub.marginal_prob(x, t)
    assert torch.allclose(std_sub, std ** 2, atol=1e-6)                       # sub-VP std = (VP std)^2
    score = lambda xx, tt: -xx                                               # noqa: E731
    r = vp.reverse(score, probability_flow=True)
    d, g = r.sde(x, t)
    assert g == 0.
    assert ve.prior_sampling((2, 3, 4, 4)).device.type == "cpu"
    assert vp.prior_logp(torch.zeros(2, 3, 4, 4)).shape == (2,)


def test_step_tables_follow_reference_formulas():
    from score_sde_pytorch_amd import sde_lib, pc_engine
    N, eps = 50, 1e-5
    sde = sde_lib.VESDE(0.01, 50, N=N)
    plan = dict(predictor="reverse_diffusion", corrector="langevin", vp_like=False, continuous=True)
    tabs = pc_engine.step_tables(sde, plan, eps, probability_flow=False)
    ts = torch.linspace(1, eps, N)
    assert torch.equal(tabs["label"], sde.marginal_prob(torch.zeros(N, 1, 1, 1), ts)[1])
    for i in [0, 7, N - 1]:
        t = torch.ones(3) * ts[i]
        f, G = sde.discretize(torch.zeros(3, 1, 1, 1), t)
        assert float(tabs["coef"][i, 0]) == 1.0
        assert float(tabs["coef"][i, 1]) == float(G[0] ** 2) and float(tabs["coef"][i, 2]) == float(G[0])
    # VP reverse diffusion: x_mean = (2 - sqrt(alpha)) x + beta score ; sub-VP Euler-Maruyama drift
    vp = sde_lib.VPSDE(0.1, 20, N=N)
    plan = dict(predictor="reverse_diffusion", corrector="langevin", vp_like=True, continuous=True)
    tabs = pc_engine.step_tables(vp, plan, 1e-3, probability_flow=False)
    idx = (torch.linspace(1, 1e-3, N) * (N - 1)).long()
    assert torch.allclose(tabs["coef"][:, 0], 2 - torch.sqrt(vp.alphas[idx]))
    assert torch.allclose(tabs["coef"][:, 1], vp.discrete_betas[idx]) and torch.equal(tabs["alpha"], vp.alphas[idx])
    plan = dict(predictor="euler_maruyama", corrector="none", vp_like=True, continuous=True)
    tabs = pc_engine.step_tables(sde_lib.subVPSDE(0.1, 20, N=N), plan, 1e-3, probability_flow=True)
    assert float(tabs["coef"][:, 2].abs().max()) == 0.0                      # ODE: no noise
    # ancestral sampling (sampling.py:213-239) and annealed Langevin dynamics (:300-319)
    plan = dict(predictor="ancestral_sampling", corrector="ald", vp_like=False, continuous=True)
    tabs = pc_engine.step_tables(sde, plan, eps, probability_flow=False)
    idx = (ts * (N - 1)).long()
    sig = sde.discrete_sigmas[idx]
    adj = torch.where(idx == 0, torch.zeros(N), sde.discrete_sigmas[idx - 1])
    assert torch.equal(tabs["coef"][:, 1], sig ** 2 - adj ** 2) and float(tabs["coef"][-1, 2]) == 0.0
    assert torch.equal(tabs["ald_std2"], sde.marginal_prob(torch.zeros(N, 1, 1, 1), ts)[1] ** 2) and "alpha" not in tabs
    vp3 = sde_lib.VPSDE(0.1, 3, N=N)
    plan = dict(predictor="ancestral_sampling", corrector="ald", vp_like=True, continuous=False)
    tabs = pc_engine.step_tables(vp3, plan, 1e-3, probability_flow=False)
    beta = vp3.discrete_betas[(torch.linspace(1, 1e-3, N) * (N - 1)).long()]
    assert torch.allclose(tabs["coef"][:, 0] ** 2 * (1 - beta), torch.ones(N)) and torch.equal(tabs["coef"][:, 2], torch.sqrt(beta))
    assert torch.equal(tabs["label"], torch.linspace(1, 1e-3, N) * (N - 1))  # discrete DDPM labels (models/utils.py:154-156)


@pytest.mark.parametrize("name,batch,gflop_per_img", [("ve/cifar10_ncsnpp_continuous", 256, 21.76),
                                                       ("subvp/cifar10_ddpmpp_continuous", 8, 21.70),
                                                       ("ve/ffhq_256_ncsnpp_continuous", 2, 532.1)])
def test_dry_lowering_of_baseline_configs(name, batch, gflop_per_img):
    """every BASELINE network lowers to a program whose conv plans validate, whose FLOP census matches the survey
    (21.84 / 21.77 / 534.6 GFLOP per image including the non-GEMM ops), and whose liveness-planned
    activation arena stays far below the 288 GB of HBM."""
    from score_sde_pytorch_amd import engine
    from score_sde_pytorch_amd.models import utils as mutils
    cfg = _util.cfgs.get_config(name)
    model = mutils.get_model("ncsnpp")(cfg)
    R = cfg.data.image_size
    eng = engine.UNetEngine(model, batch, R, R, torch.device("cpu"))
    lds = eng.validate_plans()
    assert max(lds) <= 160 * 1024
    assert abs(eng.flops_per_forward() / batch / 1e9 - gflop_per_img) / gflop_per_img < 5e-3
    assert eng.b.arena_bytes < 8e9
    with pytest.raises(RuntimeError, match="MI355X only"):
        eng.program.run()


def test_two_kernel_winograd_is_the_form_of_every_f4x4_layer(monkeypatch):
    """engine.Lowering._wino4_two_kernels: at the BASELINE sampler shape every F(4x4,3x3) layer is lowered as transform pass +
    register-fed matrix kernel (SSDE_TILE_WINOGRAD4R) and owns a transformed-input buffer of 36/16 of its input; SSDE_WINO4_TWO=3
    is round 4's rule (from four cout tiles up: the 128-cout layers on the fused kernel), SSDE_WINO4_TWO=0 switches the form off."""
    from score_sde_pytorch_amd import engine, _lib as L
    from score_sde_pytorch_amd.models import utils as mutils
    monkeypatch.setenv("SSDE_WINOGRAD", "1")               # the production heuristic (the suite's default forces F(2x2,3x3))
    cfg = _util.cfgs.get_config("ve/cifar10_ncsnpp_continuous")
    model = mutils.get_model("ncsnpp")(cfg)

    def convs(eng):
        out = []
        for kind, f, _, _ in eng.b.specs:
            if kind == L.OP_CONV and f["ksize"] == 3:
                out.append((f["tile"], f["c_out"], f["h_out"], f["main"]["c0"] + f["main"]["c1"], f["wino_v"]))
        return out
    cs = convs(engine.UNetEngine(model, 256, 32, 32, torch.device("cpu")))
    two = [c for c in cs if c[0] == L.TILE_WINOGRAD4R]
    assert len(two) >= 40 and not [c for c in cs if c[0] == L.TILE_WINOGRAD4]
    assert all(c[2] >= 8 and c[3] % 8 == 0 for c in two)
    # the 8x8 maps (128 workgroup tiles at batch 256): two shares of the split matrix kernel fill the chip -- every layer with
    # 256 input channels or more; the 128-channel one stays on F(2x2,3x3); SSDE_W4R_SPLIT=0 is round 4's choice
    at8 = [c for c in cs if c[2] == 8 and c[0] in (L.TILE_WINOGRAD4R, L.TILE_WINOGRAD)]
    assert len(at8) >= 20 and all((c[0] == L.TILE_WINOGRAD4R) == (c[3] >= 256) for c in at8)
    monkeypatch.setenv("SSDE_W4R_SPLIT", "0")
    assert not [c for c in convs(engine.UNetEngine(model, 256, 32, 32, torch.device("cpu"))) if c[0] == L.TILE_WINOGRAD4R and c[2] < 16]
    monkeypatch.delenv("SSDE_W4R_SPLIT")
    # batch 128 (the training step): four shares
    cs128 = convs(engine.UNetEngine(model, 128, 32, 32, torch.device("cpu")))
    assert len([c for c in cs128 if c[2] == 8 and c[0] == L.TILE_WINOGRAD4R]) >= 20
    monkeypatch.setenv("SSDE_W4R_SPLIT4", "0")
    assert not [c for c in convs(engine.UNetEngine(model, 128, 32, 32, torch.device("cpu"))) if c[2] == 8 and c[0] == L.TILE_WINOGRAD4R]
    monkeypatch.delenv("SSDE_W4R_SPLIT4")
    for tile, c_out, h, c_in, v in two:
        assert v is not None and v.numel == 36 * 256 * (h // 4) ** 2 * c_in
    assert all(c[4] is None for c in cs if c[0] != L.TILE_WINOGRAD4R)          # inference: nobody else wants the transformed input
    monkeypatch.setenv("SSDE_WINO4_TWO", "3")
    cs = convs(engine.UNetEngine(model, 256, 32, 32, torch.device("cpu")))
    two, one = [c for c in cs if c[0] == L.TILE_WINOGRAD4R], [c for c in cs if c[0] == L.TILE_WINOGRAD4]
    assert len(two) >= 15 and len(one) >= 15 and all(c[1] >= 256 for c in two) and all(c[1] < 256 for c in one)
    monkeypatch.setenv("SSDE_WINO4_TWO", "0")
    assert not [c for c in convs(engine.UNetEngine(model, 256, 32, 32, torch.device("cpu"))) if c[0] == L.TILE_WINOGRAD4R]


def test_weight_packing_layout():
    from score_sde_pytorch_amd.engine import pack_conv_weight, pack_matrix
    w = torch.arange(5 * 3 * 3 * 3, dtype=torch.float32).reshape(5, 3, 3, 3)
    p = pack_conv_weight(w)
    assert p.shape == (1, 9, 64, 8)
    for (co, ci, ky, kx) in [(0, 0, 0, 0), (4, 2, 2, 1), (3, 1, 0, 2)]:
        assert float(p[ci // 8, ky * 3 + kx, co, ci % 8]) == float(w[co, ci, ky, kx])
    assert float(p[0, :, 5:, :].abs().max()) == 0 and float(p[0, :, :, 3:].abs().max()) == 0
    m = torch.randn(70, 20)
    pm = pack_matrix(m)
    assert pm.shape == (3, 1, 128, 8) and float(pm[2, 0, 69, 3]) == float(m[69, 19])


def test_state_dict_keys_and_counts_match_reference_census():
    """parameter tree compatibility: 572 tensors, 62,758,915 parameters for CIFAR NCSN++"""
    from score_sde_pytorch_amd.models import utils as mutils
    cfg = _util.cfgs.get_config("ve/cifar10_ncsnpp_continuous")
    model = mutils.get_model("ncsnpp")(cfg)
    sd = model.state_dict()
    assert len(sd) == 572 and sum(p.numel() for p in model.parameters()) == 62758915
    for k in ["sigmas", "all_modules.0.W", "all_modules.1.weight", "all_modules.3.bias", "all_modules.4.GroupNorm_0.weight",
              "all_modules.4.Conv_0.weight", "all_modules.4.Dense_0.bias", "all_modules.4.Conv_1.weight"]:
        assert k in sd, k
    assert any(k.endswith("NIN_3.W") for k in sd) and any(k.endswith("Conv2d_0.weight") for k in sd)
    ckpt = {"module." + k: v for k, v in sd.items()}             # DataParallel-style reference checkpoint
    model.load_state_dict(mutils.strip_data_parallel_prefix(ckpt), strict=True)
    assert not model.all_modules[0].W.requires_grad


def test_ema_update_and_optimizer_state_follow_the_reference_run():
    """Host classes against data the REFERENCE produced (tests/golden/ref_checkpoint_small.pth after two of its steps,
    train_small.npz `ckpt/*` after its third; oracle/gen_golden_train.py): models.ema.ExponentialMovingAverage.load_state_dict +
    update() reproduce the reference's third shadow parameters (models/ema.py:32-51), and losses.get_optimizer's Adam loads
    the reference's optimizer.state_dict() (the optimizer arithmetic itself is checked by the step_fn fixture tests)."""
    import os
    from score_sde_pytorch_amd.models.ema import ExponentialMovingAverage
    from score_sde_pytorch_amd.models import utils as mutils
    from score_sde_pytorch_amd import utils as ssde_utils, losses
    gold = np.load(os.path.join(_util.GOLDEN, "train_small.npz"))
    raw = ssde_utils.load_checkpoint_file(os.path.join(_util.GOLDEN, "ref_checkpoint_small.pth"), "cpu")
    cfg = _util.train_case_config(_util.TRAIN_CKPT_CASE)
    model = mutils.get_model("ncsnpp")(cfg)
    state = dict(model=model, optimizer=losses.get_optimizer(cfg, model.parameters()),
                 ema=ExponentialMovingAverage(model.parameters(), decay=0.5), step=0)
    ssde_utils.restore_checkpoint(os.path.join(_util.GOLDEN, "ref_checkpoint_small.pth"), state, "cpu")
    ema = state['ema']
    assert state['step'] == 2 and ema.num_updates == 2 and ema.decay == cfg.model.ema_rate
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert len(ema.shadow_params) == len(names) == len(raw['ema']['shadow_params'])
    osd = state['optimizer'].state_dict()
    assert set(osd['state']) == set(raw['optimizer']['state']) and all(float(v['step']) == 2 for v in osd['state'].values())
    probes = [k.split("/", 2)[2] for k in gold.files if k.startswith("ckpt/p/")]
    cur = dict(model.named_parameters())
    with torch.no_grad():
        for n in probes:                    # the reference's third optimizer step, taken from the fixture
            cur[n].copy_(torch.from_numpy(gold["ckpt/p/" + n]))
    ema.update(model.parameters())
    assert ema.num_updates == 3
    for n in probes:
        e_ref = torch.from_numpy(gold["ckpt/e/" + n])
        assert torch.equal(ema.shadow_params[names.index(n)], e_ref) or \
            _util.rel_err(ema.shadow_params[names.index(n)], e_ref) < 1e-7, n


def test_ema_matches_reference_formula():
    from score_sde_pytorch_amd.models.ema import ExponentialMovingAverage
    p = [torch.nn.Parameter(torch.ones(4)), torch.nn.Parameter(torch.zeros(2), requires_grad=False)]
    ema = ExponentialMovingAverage(p, decay=0.999)
    with torch.no_grad():
        p[0].mul_(3.0)
    ema.update(p)
    d = min(0.999, 2 / 11)                                       # (1 + n) / (10 + n) warm-up, models/ema.py:45-47
    assert torch.allclose(ema.shadow_params[0], torch.ones(4) - (1 - d) * (torch.ones(4) - 3.0))
    assert len(ema.shadow_params) == 1
    ema.store(p); ema.copy_to(p)
    assert torch.allclose(p[0], ema.shadow_params[0])
    ema.restore(p)
    assert torch.allclose(p[0].detach(), torch.full((4,), 3.0))
    with pytest.raises(ValueError):
        ExponentialMovingAverage(p, decay=1.5)


def test_on_device_rk45_reproduces_scipy():
    """ode.solve_rk45 is scipy's RK45 (same initial step, error norm, step control): identical nfev, result to rounding"""
    import math
    from scipy import integrate
    from score_sde_pytorch_amd import ode
    A = np.array([[-0.5, 2.0, 0.0], [-2.0, -0.5, 0.3], [0.0, -0.3, -1.0]])
    At = torch.from_numpy(A)
    b = np.array([1.0, 0.0, 0.5])
    y0 = np.array([1.0, -0.5, 2.0])
    for span in [(0.0, 5.0), (1.0, 1e-3)]:
        sol = integrate.solve_ivp(lambda t, y: A @ y + math.sin(3 * t) * b, span, y0, rtol=1e-5, atol=1e-5, method="RK45")
        y, nfev = ode.solve_rk45(lambda t, y: At @ y + math.sin(3 * t) * torch.from_numpy(b), span, torch.from_numpy(y0), 1e-5, 1e-5)
        assert nfev == sol.nfev
        assert float(np.abs(sol.y[:, -1] - y.numpy()).max()) < 1e-12


def test_ema_swap_and_restore_invalidate_packed_weights():
    """ExponentialMovingAverage.copy_to / restore (non-flat path) must be visible to WeightStore.refresh: an engine or
    captured sampler graph lowered BEFORE the swap keeps packed kernel-layout copies keyed on Tensor._version /
    the flat buffer's generation (`param.data.copy_` bumps neither)."""
    from score_sde_pytorch_amd import engine as E
    from score_sde_pytorch_amd.models import utils as mutils, ema as ema_mod
    cfg = _util.small_config("ncsnpp")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    eng = E.UNetEngine(model, 2, 16, 16, torch.device("cpu"))          # dry lowering: packs weights, runs nothing
    ema = ema_mod.ExponentialMovingAverage(model.parameters(), decay=0.999)
    with torch.no_grad():
        for s in ema.shadow_params:
            s.mul_(1.25)
    w = model.all_modules[3].weight                                    # first conv
    entry = next(e for e in eng.weights.entries if any(src is w for src in e[1]))
    before = entry[0].clone()
    ema.store(model.parameters())
    ema.copy_to(model.parameters())
    eng.weights.refresh(on_device=False)
    swapped = entry[0].clone()
    assert not torch.equal(before, swapped)
    assert torch.allclose(swapped, before * 1.25)
    ema.restore(model.parameters())
    eng.weights.refresh(on_device=False)
    assert torch.equal(entry[0], before)


def test_grad_run_extents_are_recorded():
    """FlatParams.write_extent: a pointer handed out by grad_run covers the whole multi-parameter run (grad_buckets
    uses it so that a bucket is not all-reduced before an op writing across its lower boundary has run)."""
    from score_sde_pytorch_amd import backward as B
    from score_sde_pytorch_amd.models import utils as mutils
    cfg = _util.small_config("ncsnpp")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    flat = B.FlatParams(model, torch.device("cpu"))
    ws, bs = model.flat_param_groups()
    run = flat.grad_run(ws)
    o = flat.index[id(ws[0])][0]
    assert flat.write_extent(o) == run.numel() == sum(w.numel() for w in ws)
    o1, n1 = flat.index[id(ws[1])]
    assert flat.write_extent(o1) == n1                                  # a plain parameter pointer covers that parameter
    assert flat.write_extent(o1 + 5) == n1 - 5


def test_fused_rhs_cache_is_bounded_per_kind(monkeypatch):
    """ode.rhs_cache_get: the ODE sampler's drift programs and the likelihood programs of a model are kept in one dict but bounded
    per kind, newest last; a hit re-inserts the entry as the most recent one; SSDE_ODE_RHS_CACHE sets the bound."""
    from score_sde_pytorch_amd import ode

    class M:
        pass
    m, made = M(), []

    def get(kind, i, **kw):
        return ode.rhs_cache_get(m, (kind, i), lambda: made.append((kind, i)) or (kind, i), **kw)
    for i in range(4):
        assert get("drift", i) == (("drift", i), True)
    for i in range(4):
        assert get("likelihood", i) == (("likelihood", i), True)          # ... without evicting a drift entry
    assert all(get("drift", i) == (("drift", i), False) for i in range(4)) and len(made) == 8
    assert get("drift", 9)[1] and get("drift", 0)[1]                      # a fifth drift entry evicts the oldest one (0) only
    assert not get("drift", 2)[1] and not get("likelihood", 0)[1]
    monkeypatch.setenv("SSDE_ODE_RHS_CACHE", "1")
    assert get("likelihood", 7)[1] and [k for k in m._ode_rhs if k[0] == "likelihood"] == [("likelihood", 7)]
    assert len([k for k in m._ode_rhs if k[0] == "drift"]) == 4           # the other kind is untouched by that call
    assert get("drift", 5, limit=2)[1] and len([k for k in m._ode_rhs if k[0] == "drift"]) == 2
