"""The plan-level C ABI driven by a host without Python: tests/c_host/plan_host.c is compiled with gcc,
loads a plan blob exported by score_sde_pytorch_amd/plan_export.py, runs one U-Net evaluation with ssde_unet_forward and
compares it with the golden output of the REFERENCE implementation (tests/golden/*.npz), tolerance 1e-4 (full forward).

  * CPU (`-m "not gpu"`): the blob is exported from a dry lowering and the C program is linked against the test-only
    emulator library -- the loader, relocation and runner code of csrc/plan.hip execute for real, kernels on the emulator;
  * GPU (`-m gpu`): the same C source linked against libssde_hip.so + libamdhip64, on the MI355X.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import _util

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c_host", "plan_host.c")
INC = os.path.join(_util.ROOT, "include")


def _case(name, tmp_path):
    from score_sde_pytorch_amd.models import utils as mutils
    cfg = {"unet_small_ncsnpp": lambda: _util.small_config("ncsnpp"),
           "unet_small_ffhq": lambda: _util.small_config("ffhq", image_size=32, ch_mult=(1, 1, 2), attn=(16,))}[name]()
    gold = np.load(os.path.join(_util.GOLDEN, name + ".npz"))
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    for k in ("x", "cond", "y"):
        np.ascontiguousarray(gold[k], dtype=np.float32).tofile(str(tmp_path / (k + ".f32")))
    return cfg, model, gold


def _run(exe, blob_path, tmp_path, env=None, extra=()):
    cmd = [exe, blob_path, str(tmp_path / "x.f32"), str(tmp_path / "cond.f32"), str(tmp_path / "y.f32"), "1e-4"] + list(extra)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "plan_host:" in r.stdout
    return r.stdout


@pytest.mark.parametrize("name", ["unet_small_ncsnpp", "unet_small_ffhq"])
def test_c_host_runs_exported_plan_on_emulator(name, tmp_path):
    import emu
    if not emu.available():
        pytest.skip("emulator needs x86-64 + ROCm's clang++")
    from score_sde_pytorch_amd import engine as E, plan_export
    cfg, model, gold = _case(name, tmp_path)
    B, _, H, W = gold["x"].shape
    with emu.emulated():
        eng = E.UNetEngine(model, B, H, W, torch.device("cpu"))
        blob = plan_export.export_unet_plan(eng)
        # the Python binding of the same entry points (what the C program calls), on the emulator
        plan = plan_export.LoadedPlan(blob)
        y = plan.unet_forward(torch.from_numpy(gold["x"]).contiguous(), torch.from_numpy(gold["cond"]).contiguous())
        assert _util.rel_err(y, torch.from_numpy(gold["y"])) < 1e-4
        plan.refresh_weights()
        y2 = plan.unet_forward(torch.from_numpy(gold["x"]).contiguous(), torch.from_numpy(gold["cond"]).contiguous())
        # (the device re-pack computes G g G^T in a different summation order than the torch packing of the first fill)
        assert _util.rel_err(y2, y) < 1e-5 and _util.rel_err(y2, torch.from_numpy(gold["y"])) < 1e-4
        plan.close()
    blob_path = str(tmp_path / "plan.blob")
    open(blob_path, "wb").write(blob)
    emu_lib = emu.build_emu.build()
    exe = str(tmp_path / "plan_host_emu")
    r = subprocess.run(["gcc", "-O1", "-std=c11", "-DHOST_IS_DEVICE", "-I", INC, SRC, "-o", exe, emu_lib, "-lm",
                        "-Wl,-rpath," + os.path.dirname(emu_lib)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = _run(exe, blob_path, tmp_path, extra=["refresh"])
    assert "first parameter: all_modules." in out


def test_blob_is_position_independent_and_rejects_corruption(tmp_path):
    """no absolute address survives in the op array; a truncated or foreign blob is refused with a message"""
    import ctypes as C
    import emu
    if not emu.available():
        pytest.skip("emulator needs x86-64 + ROCm's clang++")
    from score_sde_pytorch_amd import engine as E, plan_export, _lib as L
    cfg, model, gold = _case("unet_small_ncsnpp", tmp_path)
    with emu.emulated():
        eng = E.UNetEngine(model, 2, 16, 16, torch.device("cpu"))
        blob = plan_export.export_unet_plan(eng)
        hdr = plan_export.PlanHeader.from_buffer_copy(blob[:C.sizeof(plan_export.PlanHeader)])
        assert hdr.n_ops == eng.program.n and hdr.n_params == len(list(model.named_parameters())) and hdr.n_relocs > hdr.n_ops
        off = C.sizeof(plan_export.PlanHeader) + hdr.n_regions * C.sizeof(plan_export.PlanRegion)
        for i in range(hdr.n_ops):
            op = L.Op.from_buffer_copy(blob[off + i * C.sizeof(L.Op): off + (i + 1) * C.sizeof(L.Op)])
            for o in plan_export._op_pointer_offsets(int(op.kind)):
                assert bytes(op)[o:o + 8] == b"\x00" * 8
        lib = plan_export.bind(L.load())
        h = C.c_void_p()
        bad = (C.c_char * 64).from_buffer_copy(blob[:64])
        assert lib.ssde_plan_load(C.cast(bad, C.c_void_p), 64, C.byref(h)) != 0 and b"blob" in lib.ssde_last_error()
        foreign = bytearray(blob); foreign[0:8] = b"NOTAPLAN"
        fb = (C.c_char * len(foreign)).from_buffer_copy(bytes(foreign))
        assert lib.ssde_plan_load(C.cast(fb, C.c_void_p), len(foreign), C.byref(h)) != 0 and b"magic" in lib.ssde_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["unet_small_ncsnpp", "unet_small_ffhq"])
def test_c_host_runs_exported_plan_on_gpu(name, tmp_path):
    from score_sde_pytorch_amd import engine as E, plan_export, _lib as L
    cfg, model, gold = _case(name, tmp_path)
    B, _, H, W = gold["x"].shape
    model = model.cuda().eval()
    eng = E.UNetEngine(model, B, H, W, torch.device("cuda"))
    blob = plan_export.export_unet_plan(eng)
    blob_path = str(tmp_path / "plan.blob")
    open(blob_path, "wb").write(blob)
    exe = str(tmp_path / "plan_host")
    libdir = os.path.dirname(L.LIB_PATH)
    r = subprocess.run(["gcc", "-O1", "-std=c11", "-I", INC, "-I", "/opt/rocm/include", SRC, "-o", exe, L.LIB_PATH,
                        "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    _run(exe, blob_path, tmp_path, extra=["refresh"])


@pytest.mark.gpu
def test_sampler_plan_through_c_abi_matches_python_sampler():
    """ssde_pc_reset / ssde_pc_run / ssde_pc_state on an exported sampler plan: same seed word, same iterations ->
    the state equals the Python-driven FusedPCSampler's bit for bit (graph replay and op-by-op launch)"""
    import ctypes as C
    from score_sde_pytorch_amd import sde_lib, sampling, plan_export, _lib as L
    from score_sde_pytorch_amd.models import utils as mutils
    cfg = _util.small_config("ncsnpp")
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.cuda().eval()
    N, B = 6, 4
    sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=50, N=N)
    sampler = sampling.get_pc_sampler(sde, (B, 3, 16, 16), sampling.ReverseDiffusionPredictor, sampling.LangevinCorrector,
                                      lambda v: v, snr=0.16, n_steps=1, continuous=True, denoise=False, eps=1e-5, device="cuda")
    x_T = torch.randn(B, 3, 16, 16, generator=torch.Generator().manual_seed(3)) * 50
    ref, _ = sampler(model, x_init=x_T, seed=77, use_graph=False)
    blob = plan_export.export_pc_plan(sampler.engine)
    plan = plan_export.LoadedPlan(blob)
    assert plan.header.kind == plan_export.PLAN_PC and plan.header.sde_steps == N and plan.header.nfe_per_iteration == 2
    lib = plan.lib
    xd = x_T.cuda().contiguous()
    out = torch.empty_like(xd)
    for use_graph in (0, 1):
        st = torch.cuda.Stream()
        L.check(lib.ssde_pc_reset(plan.handle, C.c_void_p(xd.data_ptr()), 77, C.c_void_p(st.cuda_stream)))
        L.check(lib.ssde_pc_run(plan.handle, N, use_graph, C.c_void_p(st.cuda_stream)))
        L.check(lib.ssde_pc_state(plan.handle, C.c_void_p(out.data_ptr()), None, C.c_void_p(st.cuda_stream)))
        st.synchronize()
        assert torch.equal(out.cpu(), ref.cpu()), use_graph
    plan.close()


def _c_host_train(exe):
    def run(blob_path, files, seed, loss_ref):
        cmd = [exe, "train", blob_path] + [files[k] for k in ("batch", "z", "a", "s", "labels", "hyper")] + [str(seed), repr(loss_ref), "1e-6"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
        assert "plan_host train:" in r.stdout
    return run


def test_training_plan_through_c_abi_on_emulator(tmp_path):
    """ssde_train_step / ssde_train_forward / ssde_unet_backward (include/ssde.h) on exported training plans: Python binding
    and the plain-C host, kernels on the emulator"""
    import emu
    import _train_checks as T
    if not emu.available():
        pytest.skip("emulator needs x86-64 + ROCm's clang++")
    emu_lib = emu.build_emu.build()
    exe = str(tmp_path / "plan_host_emu")
    r = subprocess.run(["gcc", "-O1", "-std=c11", "-DHOST_IS_DEVICE", "-I", INC, SRC, "-o", exe, emu_lib, "-lm",
                        "-Wl,-rpath," + os.path.dirname(emu_lib)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    with emu.emulated():
        worst = T.check_train_plan("cpu", tmp_path, c_host=_c_host_train(exe))
    assert worst < T.TOL_GRAD


@pytest.mark.gpu
def test_training_plan_through_c_abi_on_gpu(tmp_path):
    import _train_checks as T
    from score_sde_pytorch_amd import _lib as L
    exe = str(tmp_path / "plan_host")
    libdir = os.path.dirname(L.LIB_PATH)
    r = subprocess.run(["gcc", "-O1", "-std=c11", "-I", INC, "-I", "/opt/rocm/include", SRC, "-o", exe, L.LIB_PATH,
                        "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    worst = T.check_train_plan("cuda", tmp_path, c_host=_c_host_train(exe))
    assert worst < T.TOL_GRAD
