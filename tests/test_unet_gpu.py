"""End-to-end U-Net parity on the GPU: score_sde_pytorch_amd.NCSNpp (HIP program through the
C ABI) against (a) golden outputs of the REFERENCE implementation (tests/golden/*.npz, made by
oracle/gen_golden.py) and (b) the CPU oracle on fresh seeded inputs.

Tolerance (fp32 path): max |y - y_ref| <= 1e-4 * max |y_ref| for a full forward."""
import os

import numpy as np
import pytest
import torch

import _util
from _util import rel_err

pytestmark = pytest.mark.gpu
TOL_FWD = 1e-4

CASES = {
    "unet_small_ncsnpp": lambda: _util.small_config("ncsnpp"),
    "unet_small_ddpmpp": lambda: _util.small_config("ddpmpp"),
    "unet_small_ffhq": lambda: _util.small_config("ffhq", image_size=32, ch_mult=(1, 1, 2), attn=(16,)),
    "unet_small_ncsnpp_3lvl": lambda: _util.small_config("ncsnpp", image_size=32, ch_mult=(1, 2, 2), num_res_blocks=2, attn=(16,)),
    "unet_cifar_ncsnpp": lambda: _util.cfgs.get_config("ve/cifar10_ncsnpp_continuous"),
    "unet_cifar_ddpmpp": lambda: _util.cfgs.get_config("subvp/cifar10_ddpmpp_continuous"),
}


def _model(cfg):
    from score_sde_pytorch_amd.models import utils as mutils
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    sd = _util.load_seeded(model, seed=1)
    return model.cuda().eval(), sd


@pytest.mark.parametrize("name", ["unet_small_ncsnpp", "unet_small_ffhq", "unet_cifar_ncsnpp"])
@pytest.mark.parametrize("mode", ["0", "1", "4"])
def test_forward_direct_kernel_and_size_heuristic(name, mode, monkeypatch):
    """SSDE_WINOGRAD=0: every 3x3 on the direct (bitwise fmaf-chain) kernel; =1: the production size heuristic; =4: the
    F(4x4,3x3) kernel wherever it is legal (the suite's default is 2: F(2x2,3x3) wherever legal)"""
    monkeypatch.setenv("SSDE_WINOGRAD", mode)
    test_forward_matches_reference_golden(name)


@pytest.mark.parametrize("name", ["unet_small_ncsnpp", "unet_small_ffhq", "unet_cifar_ncsnpp"])
def test_forward_on_the_bf16_matrix_pipe(name, monkeypatch):
    """SSDE_MATRIX=bf16x6: the 1x1 / NIN / Linear GEMMs as exact-fp32 products of a 3-way bf16 split on the BF16 matrix pipe --
    against the reference goldens at the unchanged tolerances"""
    monkeypatch.setenv("SSDE_MATRIX", "bf16x6")
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    test_forward_matches_reference_golden(name)


@pytest.mark.parametrize("name", ["unet_small_ncsnpp", "unet_small_ffhq", "unet_cifar_ncsnpp"])
def test_forward_winograd_f4x4_in_two_kernels(name, monkeypatch):
    """SSDE_WINO4_TWO=2: every F(4x4,3x3) layer as a transform pass + the register-fed matrix kernel (conv_wino4r.hip); the production rule
    (from four cout tiles up) is what test_forward_matches_reference_golden runs on the CIFAR network"""
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    monkeypatch.setenv("SSDE_WINO4_TWO", "2")
    test_forward_matches_reference_golden(name)


@pytest.mark.parametrize("name", [n for n in CASES if "cifar" in n])
def test_forward_with_the_wide_gemm_tile_forced(name, monkeypatch):
    """SSDE_X6_WIDE=1: every 1x1 / NIN layer with Cout % 256 == 0 on the 128 x 256 tile of the split kernel (the production rule
    takes it from 512 workgroups up, i.e. at bench sizes: test_bench_sizes_gpu.py) -- its GroupNorm partials feed the next layer"""
    monkeypatch.setenv("SSDE_MATRIX", "bf16x6")
    monkeypatch.setenv("SSDE_X6_WIDE", "1")
    test_forward_matches_reference_golden(name)


@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_reference_golden(name):
    path = os.path.join(_util.GOLDEN, name + ".npz")
    assert os.path.exists(path), "golden fixture missing: run oracle/gen_golden.py in the build container"
    gold = np.load(path)
    model, _ = _model(CASES[name]())
    x, cond, y_ref = (torch.from_numpy(gold[k]) for k in ("x", "cond", "y"))
    with torch.no_grad():
        y = model(x.cuda(), cond.cuda())
    assert y.shape == y_ref.shape
    assert torch.isfinite(y).all()
    assert rel_err(y, y_ref) < TOL_FWD, rel_err(y, y_ref)
    # per sample (a sample at sigma 0.01 next to one at sigma 50 does not hide behind the batch maximum)
    assert _util.per_sample_err(y, y_ref) < 2 * TOL_FWD, _util.per_sample_err(y, y_ref)


@pytest.mark.parametrize("batch", [1, 5, 8])
def test_forward_matches_oracle_other_batches(batch):
    """ragged batch sizes exercise the tile tails (M not a multiple of any tile)"""
    from oracle import unet_oracle
    cfg = _util.small_config("ncsnpp", image_size=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn=(16,))
    model, sd = _model(cfg)
    sd = dict(sd); sd["sigmas"] = model.sigmas.cpu()
    g = torch.Generator().manual_seed(batch)
    x = torch.rand(batch, 3, 32, 32, generator=g) + 3 * torch.randn(batch, 3, 32, 32, generator=g)
    sig = torch.exp(torch.rand(batch, generator=g) * 6 - 3)
    with torch.no_grad():
        y = model(x.cuda(), sig.cuda())
        ref = unet_oracle.ncsnpp_forward(cfg, sd, x, sig)
    assert rel_err(y, ref) < TOL_FWD


def test_weight_update_is_picked_up():
    """engines cache packed weights; an in-place parameter change must re-pack (optimizer.step semantics)"""
    cfg = _util.small_config("ncsnpp")
    model, _ = _model(cfg)
    x = torch.rand(2, 3, 16, 16).cuda(); sig = torch.tensor([1.0, 7.0]).cuda()
    with torch.no_grad():
        y0 = model(x, sig)
        model.all_modules[3].weight.mul_(1.5)
        y1 = model(x, sig)
    assert rel_err(y1, y0) > 1e-3


def test_cpu_tensor_fails_loudly():
    cfg = _util.small_config("ncsnpp")
    from score_sde_pytorch_amd.models import utils as mutils
    model = mutils.get_model("ncsnpp")(cfg)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.rand(1, 3, 16, 16), torch.ones(1))


def test_forward_ffhq256_full_architecture():
    """BASELINE configs[3]: the 65.6 M-parameter 256x256 NCSN++ (output-skip pyramid, input-skip Combine), batch 1, vs the CPU oracle"""
    from oracle import unet_oracle
    cfg = _util.cfgs.get_config("ve/ffhq_256_ncsnpp_continuous")
    model, sd = _model(cfg)
    sd = dict(sd); sd["sigmas"] = model.sigmas.cpu()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 3, 256, 256, generator=g) * 3
    sig = torch.tensor([2.5])
    with torch.no_grad():
        y = model(x.cuda(), sig.cuda())
        ref = unet_oracle.ncsnpp_forward(cfg, sd, x, sig)
    assert rel_err(y, ref) < TOL_FWD
