"""INTEGRATION.md Level 1, executed (build container only: needs /root/reference and the CPU emulator).

The reference's OWN glue files -- `controllable_generation.py`, `likelihood.py` and `configs/ve/cifar10_ncsnpp_continuous.py`,
`configs/subvp/cifar10_ddpmpp_continuous.py` with their `configs/default_cifar10_configs.py` -- are copied UNMODIFIED
(byte-identical, asserted) into a scratch directory and imported on top of the `sys.modules` aliasing of INTEGRATION.md
(`sde_lib`, `sampling`, `losses`, `op`, `models.utils`, `models.ncsnpp`, `models.ema` -> score_sde_pytorch_amd), i.e. exactly what a
maintainer who takes Level 1 runs.  They drive this repository's plugin surface on the down-sized networks of
tests/_util.small_config (the reference's config objects, with the `--config.model.*` overrides a user would pass on the command
line), the kernels execute on the emulator, and the results are compared with what the reference's own stack produced:
tests/golden/controllable_small.npz and ode_small.npz (oracle/gen_golden_controllable.py, gen_golden_ode.py).
Nothing here runs on the GPU box (no /root/reference there); the GPU suite covers the same entry points through the package.
"""
import hashlib
import importlib
import os
import shutil
import sys

import numpy as np
import pytest
import torch

import emu
import _util
from _util import rel_err

REF = "/root/reference"
GLUE = ["controllable_generation.py", "likelihood.py", "configs/__init__.py", "configs/default_cifar10_configs.py",
        "configs/ve/__init__.py", "configs/ve/cifar10_ncsnpp_continuous.py", "configs/subvp/__init__.py",
        "configs/subvp/cifar10_ddpmpp_continuous.py", "configs/vp/__init__.py", "configs/vp/cifar10_ddpmpp.py"]

pytestmark = pytest.mark.skipif(not (os.path.isdir(REF) and emu.available()),
                                reason="needs the reference checkout (build container) and the CPU emulator")


def _sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


@pytest.fixture(scope="module")
def level1(tmp_path_factory):
    """scratch copy of the glue files + the aliasing of INTEGRATION.md; yields the imported reference modules"""
    root = str(tmp_path_factory.mktemp("level1"))
    for rel in GLUE:
        src = os.path.join(REF, rel)
        dst = os.path.join(root, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        if os.path.exists(src):
            shutil.copyfile(src, dst)
        else:                                   # the reference's configs/ directories are namespace packages (no __init__.py)
            assert rel.endswith("__init__.py"), rel
    os.makedirs(os.path.join(root, "ml_collections"))
    with open(os.path.join(root, "ml_collections", "__init__.py"), "w") as f:      # the 6-line ConfigDict shim
        f.write("class ConfigDict(dict):\n"
                "    def __getattr__(self, k):\n"
                "        try: return self[k]\n"
                "        except KeyError: raise AttributeError(k)\n"
                "    def __setattr__(self, k, v): self[k] = v\n")
    saved_modules = dict(sys.modules)
    saved_path = list(sys.path)
    # ---- INTEGRATION.md Level 1, verbatim (minus the two glue modules under test, which stay the reference's)
    import score_sde_pytorch_amd as amd                                                        # noqa: F401
    from score_sde_pytorch_amd import sde_lib, sampling, losses, op
    from score_sde_pytorch_amd.models import utils as mutils, ncsnpp, ema
    import score_sde_pytorch_amd.models as amd_models
    for name in ("configs", "controllable_generation", "likelihood", "ml_collections"):
        for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
            del sys.modules[k]
    sys.modules.update({
        "sde_lib": sde_lib, "sampling": sampling, "losses": losses, "op": op,
        "models": amd_models, "models.utils": mutils, "models.ncsnpp": ncsnpp, "models.ema": ema,
    })
    sys.path.insert(0, root)
    try:
        cg = importlib.import_module("controllable_generation")
        lik = importlib.import_module("likelihood")
        cfg_ve = importlib.import_module("configs.ve.cifar10_ncsnpp_continuous")
        cfg_subvp = importlib.import_module("configs.subvp.cifar10_ddpmpp_continuous")
        cfg_vp = importlib.import_module("configs.vp.cifar10_ddpmpp")
        for mod, rel in ((cg, GLUE[0]), (lik, GLUE[1]), (cfg_ve, GLUE[5]), (cfg_subvp, GLUE[7]), (cfg_vp, GLUE[9])):
            assert os.path.realpath(mod.__file__).startswith(os.path.realpath(root)), mod.__file__
            assert _sha(mod.__file__) == _sha(os.path.join(REF, rel)), "%s is not the reference's file" % rel
        assert cg.shared_predictor_update_fn is sampling.shared_predictor_update_fn        # the glue sits on OUR sampler
        assert lik.mutils is mutils
        yield dict(cg=cg, lik=lik, cfg_ve=cfg_ve, cfg_subvp=cfg_subvp, cfg_vp=cfg_vp)
    finally:
        sys.path[:] = saved_path
        for k in list(sys.modules):
            if k not in saved_modules:
                del sys.modules[k]
        sys.modules.update(saved_modules)


@pytest.fixture(autouse=True)
def _emulated(monkeypatch):
    """kernels on the emulator; the product refuses CPU tensors, under the emulator "device" memory IS host memory"""
    from score_sde_pytorch_amd.models import ncsnpp
    from score_sde_pytorch_amd import autograd as A

    def fwd(self, x, time_cond):
        if torch.is_grad_enabled() and x.requires_grad:
            return A.unet_apply(self, x, time_cond)
        return self._engine_for(x).forward(x, time_cond).clone()
    monkeypatch.setattr(ncsnpp.NCSNpp, "forward", fwd)
    with emu.emulated():
        yield


def _shrink(config, kind):
    """the `--config.model.*` / `--config.data.*` overrides that turn the reference's config into tests/_util.small_config(kind)"""
    small = _util.small_config(kind)
    for sec in ("model", "data"):
        for k, v in small[sec].items():
            if k in config[sec] and config[sec][k] != v:
                config[sec][k] = v
    config.device = torch.device("cpu")
    return config


def _model(config, seed=1):
    from score_sde_pytorch_amd.models import utils as mutils
    torch.manual_seed(0)
    model = mutils.create_model(config)                 # run_lib.py:83 (the reference's own call, on its own config object)
    _util.load_seeded(model, seed=seed)
    return model.eval()


@pytest.mark.parametrize("name", list(_util.CONTROLLABLE_CASES))
def test_reference_controllable_generation_runs_on_the_aliased_modules(name, level1):
    """the reference's get_pc_inpainter / get_pc_colorizer (controllable_generation.py:8-180, unmodified) on top of this
    repository's sampling / sde_lib / models, against the output of the reference's own stack"""
    import sampling
    import sde_lib
    cg = level1["cg"]
    gold = np.load(os.path.join(_util.GOLDEN, "controllable_small.npz"))
    task, variant, pred, corr = _util.CONTROLLABLE_CASES[name]
    kind, sde_kind, kw, _, _, _, continuous, _, _, eps = _util.PC_VARIANTS[variant]
    cfg_mod = {"vesde": level1["cfg_ve"], "subvpsde": level1["cfg_subvp"], "vpsde": level1["cfg_vp"]}[sde_kind]
    config = _shrink(cfg_mod.get_config(), kind)
    model = _model(config)
    sde = {"vesde": sde_lib.VESDE, "vpsde": sde_lib.VPSDE, "subvpsde": sde_lib.subVPSDE}[sde_kind](**kw)
    B, R, N = _util.PC_VARIANT_BATCH, _util.PC_VARIANT_SIZE, kw["N"]
    data, mask, prior, noises = _util.controllable_inputs(name, B, N, R, kw.get("sigma_max", 1.0))
    seq = []
    for i in range(N):                          # randn_like call order of one iteration (controllable_generation.py:44-52,78-80)
        if corr != "none":
            seq.append(noises[i, 0])
        seq.append(noises[i, 2])
        if pred != "none":
            seq.append(noises[i, 1])
        seq.append(noises[i, 3])
    it = iter(seq)
    real = torch.randn_like
    torch.randn_like = lambda t, **k: next(it).to(t.device)
    sde.prior_sampling = lambda shape: prior.clone()
    args = (sde, sampling.get_predictor(pred), sampling.get_corrector(corr), lambda v: v)
    kws = dict(snr=0.16, n_steps=1, probability_flow=False, continuous=continuous, denoise=True, eps=eps)
    try:
        if task == "inpaint":
            out = cg.get_pc_inpainter(*args, **kws)(model, data, mask)
        else:
            out = cg.get_pc_colorizer(*args, **kws)(model, data)
    finally:
        torch.randn_like = real
    assert next(it, None) is None, "noise sequence not consumed as the reference consumes it"
    assert rel_err(out, torch.from_numpy(gold[name])) < 1e-4


def test_reference_likelihood_closures_run_on_the_aliased_modules(level1):
    """the reference's likelihood.py (unmodified): get_div_fn's Hutchinson estimator differentiates THROUGH this repository's
    U-Net (autograd bridge -> HIP backward program) -- one evaluation against the reference stack's values (ode_small.npz) -- and
    get_likelihood_fn's scipy integration runs end to end (loose tolerance: the 2000-evaluation solve of the fixture is a
    GPU-suite test through the package's own likelihood.py)."""
    import sde_lib
    from models import utils as mutils
    lik = level1["lik"]
    gold = np.load(os.path.join(_util.GOLDEN, "ode_small.npz"))
    case = _util.ODE_CASE
    config = _shrink(level1["cfg_subvp"].get_config(), "ddpmpp")
    model = _model(config)
    sde = sde_lib.subVPSDE(**case["sde_kwargs"])
    z, data, epsilon = _util.ode_case_inputs()
    t_probe = torch.ones(data.shape[0]) * case["t_probe"]
    score_fn = mutils.get_score_fn(sde, model, train=False, continuous=True)
    drift_fn = lambda xx, tt: sde.reverse(score_fn, probability_flow=True).sde(xx, tt)[0]   # noqa: E731  (likelihood.py:59-63)
    with torch.no_grad():
        d0 = drift_fn(data.clone(), t_probe)
    div0 = lik.get_div_fn(drift_fn)(data.clone(), t_probe, epsilon)                          # likelihood.py:26-37
    assert rel_err(d0, torch.from_numpy(gold["rhs_drift"])) < 1e-4
    assert rel_err(div0, torch.from_numpy(gold["rhs_div"])) < 2e-3
    # the whole likelihood_fn of the reference (scipy RK45 on the host, fp64 numpy <-> fp32 tensors per evaluation) at a tolerance a CPU run affords
    real = torch.randint_like
    torch.randint_like = lambda t, low=0, high=2, **k: ((epsilon + 1.) / 2.).to(t.device)
    try:
        lf = lik.get_likelihood_fn(sde, _util.ode_inverse_scaler, hutchinson_type="Rademacher", rtol=1e-1, atol=1e-1,
                                   method="RK45", eps=case["lik_eps"])
        for p in model.parameters():            # (a user's own economy: without it the bridge also returns the 154 parameter
            p.requires_grad_(False)             #  gradients the reference's get_div_fn never asks for -- 3x the emulator time)
        bpd, zz, nfe = lf(model, data.clone())
    finally:
        torch.randint_like = real
    # (an adaptive solve at rtol = atol = 1e-1 sits within ~5e-3 of the converged value; the step sequence -- hence the last digits -- follows
    #  the error estimates, so the yardstick is the reference stack's converged bits/dim, not a second loose solve)
    print("reference likelihood_fn on the aliased modules: nfe %d, bpd %s (reference stack, rtol 1e-5: %s)" % (nfe, bpd.tolist(), gold["lik_bpd"].tolist()))
    assert 20 <= nfe <= int(gold["lik_nfe"]) and torch.isfinite(zz).all()
    assert rel_err(bpd, torch.from_numpy(gold["lik_bpd"])) < 2e-2


def test_reference_config_file_builds_the_full_model(level1):
    """configs/ve/cifar10_ncsnpp_continuous.py, unmodified and unshrunk: create_model + the VE SDE + get_sampling_fn accept it and
    lower the full 62.8 M-parameter network (a dry lowering: nothing is executed)"""
    import sampling
    import sde_lib
    from models import utils as mutils
    config = level1["cfg_ve"].get_config()
    config.device = torch.device("cpu")
    model = mutils.create_model(config)
    assert sum(p.numel() for p in model.parameters()) == 62758915 and len(model.state_dict()) == 572
    sde = sde_lib.VESDE(sigma_min=config.model.sigma_min, sigma_max=config.model.sigma_max, N=config.model.num_scales)
    shape = (config.eval.batch_size if False else 4, config.data.num_channels, config.data.image_size, config.data.image_size)
    fn = sampling.get_sampling_fn(config, sde, shape, lambda v: v, 1e-5)        # run_lib.py:100 / :255
    assert callable(fn)
