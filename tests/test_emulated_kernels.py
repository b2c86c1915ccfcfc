"""CPU execution of the product's HIP kernel sources under the test-only emulator (tests/emu/):
the same .hip files, compiled as host C++ against an emulated <hip/hip_runtime.h> (fibers for
threads, wave64 collectives, exact-f32 MFMA lane maps), driven through the same C ABI.

This is NOT the product path and proves nothing about speed; it pins the kernels' index math,
LDS protocols and fragment handling to the reference golden vectors in the build container,
where no GPU exists.  The `-m gpu` suites remain the parity tests proper (real hardware).
Tolerances as in the GPU suites (fp32): 2e-5 per contraction, 1e-4 per full forward.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _util
from _util import rel_err
import emu

pytestmark = pytest.mark.skipif(not emu.available(), reason="emulator needs x86-64 + ROCm's clang++")


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.fixture()
def ops():
    from score_sde_pytorch_amd import hipops
    with emu.emulated():
        yield hipops


@pytest.mark.parametrize("n,cin,cout,h,tile", [(2, 32, 64, 8, 0), (1, 64, 96, 16, 3), (2, 4, 128, 8, 0),
                                                (3, 16, 4, 8, 0), (1, 32, 64, 16, 1), (1, 32, 64, 16, 2)])
def test_conv3x3(ops, n, cin, cout, h, tile):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, cin, h, h, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    b = torch.randn(cout, generator=g)
    y = ops.conv2d(nhwc(x), w, b, tile=tile)
    assert rel_err(nchw(y), F.conv2d(x, w, b, padding=1)) < 2e-5


def test_groupnorm_stats_and_attention(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 64, 8, 8, generator=g) * 2 + 1
    mean, rstd = ops.groupnorm_stats(nhwc(x), 16, 1e-6)
    xr = x.reshape(3, 16, -1).double()
    assert rel_err(mean, xr.mean(-1)) < 2e-6
    assert rel_err(rstd, 1 / torch.sqrt(xr.var(-1, unbiased=False) + 1e-6)) < 1e-5
    qkv = torch.randn(2, 64, 96, generator=g)
    o = ops.attention(qkv, 32)
    q, k, v = qkv[..., :32], qkv[..., 32:64], qkv[..., 64:]
    assert rel_err(o, torch.softmax(q @ k.transpose(1, 2) * 32 ** -0.5, -1) @ v) < 2e-5


@pytest.mark.parametrize("n,c", [(2, 64), (1, 256), (1, 192)])
def test_attention_on_the_bf16_matrix_pipe(ops, n, c, monkeypatch):
    """attn_x6_kernel (SSDE_ATTNF_BF16X6: L = 256 tokens; Q K^T and P V as exact-fp32 products of a 3-way bf16 split, fp32
    softmax) against fp64 -- at the fp32 kernel's tolerance -- and a permutation check that catches a transposed operand."""
    monkeypatch.setenv("SSDE_MATRIX", "bf16x6")
    g = torch.Generator().manual_seed(30 + c)
    l = 256
    qkv = torch.randn(n, l, 3 * c, generator=g)
    qkv[:, :, :c] *= 2.0                      # sharpen the softmax
    qkv[0, 3, :c] *= 6.0                      # one very peaked query row
    y = ops.attention(qkv, c)
    q, k, v = qkv[..., :c].double(), qkv[..., c:2 * c].double(), qkv[..., 2 * c:].double()
    ref = torch.softmax(q @ k.transpose(1, 2) * (c ** -0.5), dim=-1) @ v
    # (the emulator adds the 16 products of a bf16 MFMA to the accumulator one by one -- 6 to 8 fp32 roundings per fp32 product;
    #  the hardware's bound against the fp32 kernel is asserted on the GPU: test_ops_gpu.py)
    assert rel_err(y, ref) < 6e-6
    monkeypatch.setenv("SSDE_ATTN_X6", "0")
    y32 = ops.attention(qkv, c)
    assert rel_err(y32, ref) < 3e-6 and rel_err(y, y32) < 8e-6 and not torch.equal(y, y32)      # (two different kernels ran)
    monkeypatch.delenv("SSDE_ATTN_X6")
    if c == 64:
        q = torch.zeros(1, l, c); k = torch.zeros(1, l, c)
        perm = torch.randperm(l, generator=g)
        for i in range(l):                    # query i picks key perm[i] through two channels: rows, keys and channels all asymmetric
            q[0, i, i % c] = 400.0; q[0, i, (i // c) * 9 % c] += 300.0
            k[0, perm[i], i % c] += 1.0; k[0, perm[i], (i // c) * 9 % c] += 1.0
        vv = torch.arange(l * c, dtype=torch.float32).reshape(1, l, c) / 100.0
        ref2 = torch.softmax(q.double() @ k.double().transpose(1, 2) * (c ** -0.5), dim=-1) @ vv.double()
        assert rel_err(ops.attention(torch.cat([q, k, vv], -1), c), ref2) < 1e-5


@pytest.mark.parametrize("up,down,pad", [(2, 1, (2, 1)), (1, 2, (1, 1)), (1, 1, (2, 2))])
def test_upfirdn2d(ops, up, down, pad):
    from oracle import unet_oracle
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 8, 8, 8, generator=g)
    k = torch.from_numpy(unet_oracle.setup_fir_kernel([1, 3, 3, 1])) * (up ** 2)
    y = ops.upfirdn2d_nhwc(nhwc(x), k, up=up, down=down, pad=pad)
    assert rel_err(nchw(y), unet_oracle.upfirdn2d(x, k, up=up, down=down, pad=pad)) < 2e-6


CASES = {
    "unet_small_ncsnpp": lambda: _util.small_config("ncsnpp"),
    "unet_small_ddpmpp": lambda: _util.small_config("ddpmpp"),
    "unet_small_ffhq": lambda: _util.small_config("ffhq", image_size=32, ch_mult=(1, 1, 2), attn=(16,)),
}


def test_unet_forward_direct_kernel(monkeypatch):
    monkeypatch.setenv("SSDE_WINOGRAD", "0")
    test_unet_forward_matches_reference_golden("unet_small_ncsnpp")


@pytest.mark.parametrize("name", ["unet_small_ncsnpp", "unet_small_ffhq"])
def test_unet_forward_winograd_f4x4_kernel(name, monkeypatch):
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    test_unet_forward_matches_reference_golden(name)


def test_unet_forward_on_the_bf16_matrix_pipe(monkeypatch):
    """the 1x1 / NIN / Linear GEMMs through the 3-way bf16 split (SSDE_MATRIX=bf16x6 -> SSDE_CONVF_BF16X6), emulated"""
    monkeypatch.setenv("SSDE_MATRIX", "bf16x6")
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    test_unet_forward_matches_reference_golden("unet_small_ffhq")


def test_unet_forward_winograd_f4x4_in_two_kernels(monkeypatch):
    """every legal 3x3 layer as a transform pass + a matrix kernel (SSDE_WINO4_TWO=2 forces what the lowering otherwise takes by
    its cout-tile rule), emulated"""
    from score_sde_pytorch_amd import engine as E, _lib as L
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    monkeypatch.setenv("SSDE_WINO4_TWO", "2")
    seen = []
    real = E.Lowering.conv

    def spy(self, *a, **k):
        real(self, *a, **k)
        seen.append(self.b.specs[-1][1]["tile"])
    monkeypatch.setattr(E.Lowering, "conv", spy)
    test_unet_forward_matches_reference_golden("unet_small_ffhq")
    assert seen.count(L.TILE_WINOGRAD4R) >= 8 and L.TILE_WINOGRAD4 not in seen


@pytest.mark.parametrize("name", list(CASES))
def test_unet_forward_matches_reference_golden(name):
    from score_sde_pytorch_amd.models import utils as mutils
    from score_sde_pytorch_amd import engine as E
    gold = np.load(os.path.join(_util.GOLDEN, name + ".npz"))
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(CASES[name]())
    _util.load_seeded(model, seed=1)
    x, cond, y_ref = (torch.from_numpy(gold[k]) for k in ("x", "cond", "y"))
    with emu.emulated():
        eng = E.UNetEngine(model, x.shape[0], x.shape[2], x.shape[3], torch.device("cpu"))
        y = eng.forward(x, cond)
    assert rel_err(y, y_ref) < 1e-4


@pytest.mark.parametrize("n,cin,cout,h", [(1, 8, 64, 16), (2, 32, 64, 8), (3, 16, 96, 8), (5, 24, 40, 8), (9, 8, 64, 8), (1, 64, 128, 32)])
def test_conv3x3_winograd(ops, n, cin, cout, h):
    """F(2x2,3x3) kernel (conv_wino.hip) against the plain fp32 convolution: same tolerance as the direct kernel"""
    pass
    from score_sde_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, cin, h, h, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    b = torch.randn(cout, generator=g)
    y = ops.conv2d(nhwc(x), w, b, tile=L.TILE_WINOGRAD)
    assert rel_err(nchw(y.cpu()), F.conv2d(x, w, b, padding=1)) < 2e-5


def test_conv3x3_winograd_fused_prologue_epilogue(ops):
    """GroupNorm+SiLU prologue over a concatenated source, bias, temb addend, residual and scale on the Winograd kernel"""
    pass
    from score_sde_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(7)
    n, c0, c1, cout, h = 2, 32, 32, 64, 8
    x1, x2 = torch.randn(n, c0, h, h, generator=g), torch.randn(n, c1, h, h, generator=g)
    C, G = c0 + c1, 16
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    w, b = torch.randn(cout, C, 3, 3, generator=g) / np.sqrt(9 * C), torch.randn(cout, generator=g)
    ca, res = torch.randn(n, cout, generator=g), torch.randn(n, cout, h, h, generator=g)
    mean, rstd = ops.groupnorm_stats(nhwc(x1), G, 1e-6, x2=nhwc(x2))
    y = ops.conv2d(nhwc(x1), w, b, x2=nhwc(x2), pro=L.PRO_GN_SILU, gn=(mean, rstd, gamma, beta, G),
                   chan_add=ca, resid=nhwc(res), scale=0.7, tile=L.TILE_WINOGRAD)
    ref = 0.7 * (F.conv2d(F.silu(F.group_norm(torch.cat([x1, x2], 1), G, gamma, beta, 1e-6)), w, b, padding=1) + ca[:, :, None, None] + res)
    assert rel_err(nchw(y.cpu()), ref) < 2e-5


# (n, h, cin1, cin2, cout, prologue, extras): shapes that take the GEMM kernel of conv1x1.hip (M >= 128, Cout >= 96),
# with ragged M / K / Cout tails, a virtual concat, every prologue and the whole epilogue
GEMM1X1_CASES = [(2, 8, 64, 0, 128, 0, False), (3, 8, 40, 0, 96, 0, True), (1, 16, 32, 24, 200, 2, True),
                 (5, 6, 8, 0, 130, 3, False), (2, 16, 64, 32, 256, 1, True)]


@pytest.mark.parametrize("matrix", ["f32", "bf16x6"])     # SSDE_MATRIX: exact-fp32 MFMA / 3-way bf16 split on the BF16 pipe
@pytest.mark.parametrize("pipe", ["0", "2"])               # SSDE_GEMM_PIPE: plain kernels / the persistent pipelined one, forced
@pytest.mark.parametrize("n,h,c1,c2,cout,pro,extras", GEMM1X1_CASES)
def test_conv1x1_gemm_kernel(ops, n, h, c1, c2, cout, pro, extras, matrix, pipe, monkeypatch):
    monkeypatch.setenv("SSDE_MATRIX", matrix)
    monkeypatch.setenv("SSDE_GEMM_PIPE", pipe)
    if pipe == "2":
        monkeypatch.setenv("SSDE_NUM_CUS", "3")      # a pretend 3-CU device: 8 persistent workgroups, several tiles each
    g = torch.Generator().manual_seed(n * 1000 + cout)
    xa = torch.randn(n, c1, h, h, generator=g)
    xb = torch.randn(n, c2, h, h, generator=g) if c2 else None
    cat = xa if xb is None else torch.cat([xa, xb], 1)
    k = c1 + c2
    w = torch.randn(cout, k, generator=g) / np.sqrt(k)
    gn, act = None, cat
    if pro in (1, 2):
        G = k // 4 if k // 4 <= 32 else 32
        while k % G or (k // G) % 4:
            G -= 1
        gam, bet = 1 + 0.1 * torch.randn(k, generator=g), 0.1 * torch.randn(k, generator=g)
        mean, rstd = ops.groupnorm_stats((nhwc(xa)), G, x2=None if xb is None else (nhwc(xb)))
        gn = (mean, rstd, (gam), (bet), G)
        act = F.group_norm(cat, G, gam, bet, 1e-6)
    if pro in (2, 3):
        act = F.silu(act)
    kw = {}
    ref = F.conv2d(act, w[:, :, None, None])
    if extras:
        bias, tproj, res = torch.randn(cout, generator=g), torch.randn(n, cout, generator=g), torch.randn(n, cout, h, h, generator=g)
        kw = dict(bias=(bias), chan_add=(tproj), resid=(nhwc(res)), scale=float(1 / np.sqrt(2)))
        ref = (ref + bias[None, :, None, None] + tproj[:, :, None, None] + res) / np.sqrt(2)
    y = ops.conv2d(aux=(nhwc(xa)), aux2=None if xb is None else (nhwc(xb)), aux_weight=w, aux_pro=pro, aux_gn=gn, **kw)
    assert rel_err(nchw(y.cpu()), ref) < 2e-5


# rows per workgroup / load-ahead depth of the split kernel (SSDE_X6_BM, SSDE_X6_PF): the two-register-set loop issues every load
# unconditionally with clamped stage indices -- odd and even stage counts, a ragged last stage, a single stage
@pytest.mark.parametrize("bm,pf", [("128", "2"), ("64", "1"), ("64", "2")])
@pytest.mark.parametrize("n,h,c1,c2,cout,pro,extras", [(3, 8, 40, 0, 96, 0, True), (1, 16, 32, 24, 200, 2, True), (2, 16, 64, 32, 256, 1, True),
                                                       (5, 6, 8, 0, 130, 3, False), (2, 8, 16, 0, 128, 0, False)])
def test_conv1x1_gemm_kernel_rows_and_load_ahead(ops, n, h, c1, c2, cout, pro, extras, bm, pf, monkeypatch):
    monkeypatch.setenv("SSDE_X6_BM", bm)
    monkeypatch.setenv("SSDE_X6_PF", pf)
    monkeypatch.setenv("SSDE_X6_WIDE", "0")
    test_conv1x1_gemm_kernel(ops, n, h, c1, c2, cout, pro, extras, "bf16x6", "0", monkeypatch)


# the 128 x 256 tile of the split kernel (SSDE_X6_WIDE=1 forces it; it needs Cout % 256 == 0): ragged M and K, a virtual concat,
# GroupNorm prologue, the whole epilogue, GroupNorm partials of the output checked through the next layer's statistics
@pytest.mark.parametrize("n,h,c1,c2,cout,pro,extras", [(3, 8, 40, 0, 256, 0, True), (1, 16, 32, 24, 512, 2, True),
                                                       (2, 16, 64, 32, 256, 1, True), (5, 4, 64, 0, 256, 3, False)])
def test_conv1x1_gemm_kernel_wide_tile(ops, n, h, c1, c2, cout, pro, extras, monkeypatch):
    monkeypatch.setenv("SSDE_X6_WIDE", "1")
    test_conv1x1_gemm_kernel(ops, n, h, c1, c2, cout, pro, extras, "bf16x6", "0", monkeypatch)


# (n, h, w, cin1, cin2, cout, prologue, dropout): shapes that take the Winograd weight-gradient kernel (wgrad_wino.hip:
# 3x3 / stride 1 / pad 1, h % 4 == 0, w % 8 == 0, channels % 64 == 0), with a virtual concat, every prologue, dropout
WGRAD_WINO_CASES = [(1, 8, 8, 64, 0, 64, 0, 0.0), (2, 8, 16, 64, 0, 128, 3, 0.0), (3, 4, 8, 64, 64, 64, 2, 0.0),
                    (1, 16, 16, 128, 0, 64, 1, 0.0), (2, 8, 8, 64, 0, 64, 2, 0.25),
                    # F(4x4,3x3) only (wgrad_wino4.hip: maps that are multiples of 4, channels multiples of 32): channel
                    # blocks that do not fill the 128 x 128 GEMM tile, a ragged last K stage, a virtual concat, dropout
                    (1, 8, 8, 64, 0, 32, 0, 0.0), (5, 16, 8, 64, 64, 96, 2, 0.0), (3, 8, 24, 128, 0, 32, 2, 0.3),
                    (2, 12, 8, 160, 0, 160, 3, 0.0)]


@pytest.mark.parametrize("wmode", ["44", "2"])
@pytest.mark.parametrize("n,h,w,c1,c2,cout,pro,drop", WGRAD_WINO_CASES)
def test_wgrad_winograd_kernel(ops, n, h, w, c1, c2, cout, pro, drop, wmode, monkeypatch):
    # SSDE_WGRAD_WINOGRAD: "44" = F(4x4,3x3) (wgrad_wino4.hip) wherever it is legal, F(2x2,3x3) elsewhere; "2" = F(2x2,3x3) only
    # (channel counts the latter does not take fall to the direct kernel: still checked against autograd)
    monkeypatch.setenv("SSDE_WGRAD_WINOGRAD", wmode)
    g = torch.Generator().manual_seed(n * 100 + cout + c1)
    xa = torch.randn(n, c1, h, w, generator=g)
    xb = torch.randn(n, c2, h, w, generator=g) if c2 else None
    cat = xa if xb is None else torch.cat([xa, xb], 1)
    k = c1 + c2
    gy = torch.randn(n, cout, h, w, generator=g)
    gn, act = None, cat
    if pro in (1, 2):
        G = min(32, k // 4)
        gam, bet = 1 + 0.1 * torch.randn(k, generator=g), 0.1 * torch.randn(k, generator=g)
        mean, rstd = ops.groupnorm_stats((nhwc(xa)), G, x2=None if xb is None else (nhwc(xb)))
        gn = (mean, rstd, (gam), (bet), G)
        act = F.group_norm(cat, G, gam, bet, 1e-6)
    if pro in (2, 3):
        act = F.silu(act)
    dropout = None
    if drop > 0:
        seed = (torch.tensor([1234], dtype=torch.int32))
        dropout = (drop, seed, 77)
    wt = torch.zeros(cout, k, 3, 3, requires_grad=True)
    dw = (torch.zeros(cout, k, 3, 3))
    kw = dict(x2=None if xb is None else (nhwc(xb)), pro=pro, gn=gn, scale=0.5, dropout=dropout)
    ops.conv_wgrad((nhwc(xa)), (nhwc(gy)), 3, dw, stride=1, pad=1, **kw)
    if drop > 0:
        # oracle mask: numpy restatement of ssde_keep over the element indices of the virtual concat tensor (NHWC)
        import _train_checks as T
        thresh = min(int(round(drop * 2.0 ** 32)), 2 ** 32 - 1)
        keep = T.hash_keep(np.arange(n * h * w * k, dtype=np.uint64), (1234 ^ 77) & 0xFFFFFFFF, thresh, 1.0 / (1.0 - drop))
        act = act * torch.from_numpy(keep.reshape(n, h, w, k)).permute(0, 3, 1, 2)
    F.conv2d(act.clone().requires_grad_(False), wt, padding=1).backward(gy)
    assert rel_err(dw.cpu(), 0.5 * wt.grad) < 2e-5


def test_rk45_stage_kernels_reproduce_scipy(ops):
    """ssde_rk_combine / ssde_rk_error_norm (the integrator's fp64 stage arithmetic and RMS error norm) driving the same
    step-size controller as scipy's RK45: identical function-evaluation count and solution on an analytic system"""
    from scipy import integrate
    from score_sde_pytorch_amd import ode
    g = torch.Generator().manual_seed(3)
    n = 301                                            # not a multiple of the block size
    A = (torch.randn(n, n, generator=g, dtype=torch.float64) / n ** 0.5 - 0.5 * torch.eye(n, dtype=torch.float64))
    y0 = torch.randn(n, generator=g, dtype=torch.float64)
    calls = []

    def fun(t, y):
        calls.append(t)
        return A @ y + torch.sin(torch.tensor(3.0 * t, dtype=torch.float64))
    stages = ode._HipStages(n, y0)
    y, nfev = ode.solve_rk45(fun, (0.0, 2.0), y0, rtol=1e-6, atol=1e-8, stages=stages)
    sol = integrate.solve_ivp(lambda t, v: (A.numpy() @ v + np.sin(3.0 * t)), (0.0, 2.0), y0.numpy(), rtol=1e-6, atol=1e-8, method="RK45")
    assert nfev == sol.nfev
    assert float((y - torch.from_numpy(sol.y[:, -1])).abs().max()) < 1e-11
    # the fp32 copy that accompanies every stage argument (the fused drift's input)
    x32 = torch.zeros(n, dtype=torch.float32)
    stages2 = ode._HipStages(n, y0, x32=x32)
    tmp = torch.empty_like(y0)
    stages2.K[0].copy_(y0 * 2)
    stages2.combine(y0, [0.25], tmp)
    assert torch.equal(tmp, y0 + (y0 * 2) * 0.25) and torch.equal(x32, tmp.to(torch.float32))


def test_pf_drift_kernel(ops):
    """ssde_pf_drift: (double)(a x - (g2 score) 0.5) with the reference's fp32 operation order (sde_lib.py:93-97)"""
    import ctypes as C
    from score_sde_pytorch_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    x, s = torch.randn(1000, generator=g), torch.randn(1000, generator=g) * 30
    a, g2 = torch.tensor(-0.5 * 7.3, dtype=torch.float32), torch.tensor(3.7, dtype=torch.float32) ** 2
    out = torch.empty(1000, dtype=torch.float64)
    args = L.PfDriftArgs()
    args.x, args.score, args.dst, args.numel, args.a, args.g2 = x.data_ptr(), s.data_ptr(), out.data_ptr(), 1000, float(a), float(g2)
    L.check(L.load().ssde_pf_drift(C.byref(args), ops._stream()))
    ref = (a * x - g2 * s * 0.5).to(torch.float64)
    assert torch.equal(out, ref)
