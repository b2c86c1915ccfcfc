"""Training path under the test-only CPU emulator (tests/emu/): the backward / optimizer kernels and the
backward lowering, compared with torch autograd over the CPU oracle.  Same checks as tests/test_train_gpu.py
(which runs them on the MI355X); see tests/_train_checks.py for what is compared and the tolerances."""
import pytest
import torch

import emu
import _train_checks as T

pytestmark = pytest.mark.skipif(not emu.available(), reason="emulator needs x86-64 + ROCm's clang++")


@pytest.fixture(autouse=True)
def _emulated():
    with emu.emulated():
        yield


def test_backward_kernels():
    T.check_backward_ops("cpu")


def test_dropout_mask_is_regenerated_identically():
    T.check_dropout_mask("cpu")


def test_upfirdn_lds_tiles():
    T.check_upfirdn_tiles("cpu")


@pytest.mark.parametrize("pipe", ["0", "2"])      # the plain GEMM kernels / the persistent pipelined one (its own partials code)
def test_groupnorm_statistics_from_producer_epilogues(pipe, monkeypatch):
    monkeypatch.setenv("SSDE_GEMM_PIPE", pipe)
    if pipe == "2":
        monkeypatch.setenv("SSDE_NUM_CUS", "3")      # persistent workgroups loop over several tiles
    T.check_fused_gn_statistics("cpu", monkeypatch)


def test_groupnorm_statistics_merged_by_the_transform_pass(monkeypatch):
    # F(4x4,3x3) in two kernels wherever it is legal: the pass merges the partials of its source itself (gn_in_part0)
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    T.check_fused_gn_statistics("cpu", monkeypatch)


def test_conv3x3_winograd_f4x4():
    T.check_conv_winograd4("cpu", big=False)


def test_fused_likelihood_right_hand_side():
    T.check_fused_likelihood_rhs("cpu")


def test_fused_ode_drift_on_a_discrete_label_ve_model():
    T.check_fused_drift_discrete_ve("cpu")


def test_bucketed_gradient_exchange_over_a_one_rank_group(monkeypatch):
    """the same check tests/test_train_gpu.py runs over RCCL, here over gloo on the emulator's host tensors"""
    T.check_forced_exchange_one_rank("cpu", "gloo", monkeypatch)


def test_conv_reduction_split_over_two_workgroups(monkeypatch):
    T.check_conv_split_reduction("cpu", monkeypatch, n=5)


@pytest.mark.parametrize("kind", ["ncsnpp", "ddpmpp", "ffhq"])
def test_whole_network_gradients(kind):
    T.check_unet_grads(kind, "cpu")


def test_whole_network_gradients_winograd_f4x4(monkeypatch):
    """forward and input-gradient convolutions on the F(4x4,3x3) kernel wherever it is legal"""
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    T.check_unet_grads("ncsnpp", "cpu")


def test_whole_network_gradients_winograd_f4x4_in_two_kernels(monkeypatch):
    """forward and input-gradient convolutions as transform pass + matrix kernel (conv_wino4r.hip); the weight gradients read
    the transform pass's output (ssde_wgrad_args.v_pre)"""
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    monkeypatch.setenv("SSDE_WINO4_TWO", "2")
    monkeypatch.setenv("SSDE_WGRAD_WINOGRAD", "44")
    T.check_unet_grads("ncsnpp", "cpu")


def test_autograd_bridge(monkeypatch):
    from score_sde_pytorch_amd.models import ncsnpp
    # the product refuses CPU tensors; under the emulator "device" memory IS host memory
    orig = ncsnpp.NCSNpp.forward

    def fwd(self, x, time_cond):
        from score_sde_pytorch_amd import autograd as A
        return A.unet_apply(self, x, time_cond)
    monkeypatch.setattr(ncsnpp.NCSNpp, "forward", fwd)
    T.check_autograd_bridge("cpu")
    monkeypatch.setattr(ncsnpp.NCSNpp, "forward", orig)


@pytest.mark.parametrize("sde_kind", ["vesde", "subvpsde", "smld", "ddpm"])
def test_fused_training_step(sde_kind):
    T.check_fused_step("cpu", steps=2, sde_kind=sde_kind)


@pytest.fixture
def host_is_device(monkeypatch):
    """step_fn itself under the emulator: the product asks losses._on_device whether a tensor is HIP memory"""
    from score_sde_pytorch_amd import losses
    monkeypatch.setattr(losses, "_on_device", lambda t: True)


@pytest.mark.parametrize("name", ["ve_cont", "subvp_cont", "smld", "ddpm", "ve_cont_lw", "subvp_cont_lw_rm"])
def test_step_fn_matches_the_reference_run(name, host_is_device):
    """three optimisation steps + the eval step of losses.get_step_fn against the REFERENCE's own run of them"""
    T.check_step_fn_against_reference_run("cpu", name)


@pytest.mark.parametrize("warm", [False, True])
def test_checkpoint_written_by_the_reference_resumes(tmp_path, warm, host_is_device):
    T.check_reference_checkpoint_resume("cpu", tmp_path, warm)


@pytest.mark.parametrize("kind", ["ncsnpp", "ffhq"])
def test_device_weight_repack(kind):
    T.check_device_repack("cpu", kind)


def test_device_weight_repack_of_the_per_lane_f4x4_image(monkeypatch):
    """every legal 3x3 layer on the two-kernel F(4x4,3x3) form: SSDE_PACK_WINO4R (conv_wino4r.hip's per-lane image), forward and
    input-gradient variants, against engine.pack_wino4r_weight"""
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    monkeypatch.setenv("SSDE_WINO4_TWO", "2")
    T.check_device_repack("cpu", "ffhq", need_kind=6)


def test_op_package(monkeypatch):
    import score_sde_pytorch_amd.op as op
    # the package refuses CPU tensors; under the emulator host memory is the device
    import torch as _t
    real = _t.Tensor.device.__get__
    T.check_op_package.__globals__  # noqa
    class _Dev:
        type = "cuda"
    monkeypatch.setattr(op, "upfirdn2d", lambda input, kernel, up=1, down=1, pad=(0, 0):
                        op.UpFirDn2d.apply(input, kernel.to(_t.float32), up, down, (pad[0], pad[1])))
    monkeypatch.setattr(op, "fused_leaky_relu", lambda input, bias, negative_slope=0.2, scale=2 ** 0.5:
                        op.FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale))
    T.check_op_package("cpu")


def test_input_gradient_only_program():
    T.check_input_gradient_only("cpu")


def test_checkpoint_roundtrip_and_ema_swap(tmp_path):
    T.check_checkpoint_and_ema_swap("cpu", tmp_path)


def test_generic_loss_closures_run_the_hip_loss_head():
    T.check_generic_loss_closures("cpu")


def test_1x1_weight_gradients_pipelined_and_chunked():
    T.check_wgrad_1x1("cpu")


@pytest.mark.parametrize("dropout", [False, True])
def test_weight_gradient_fed_by_the_forward_launch(monkeypatch, dropout):
    T.check_wino_v_from_forward("cpu", monkeypatch, dropout=dropout)


def test_conv3x3_winograd_f4x4_register_fed_matrix_kernel():
    T.check_conv_winograd4("cpu", big=False, regs=True)


def test_conv3x3_onto_the_image_channels():
    T.check_conv_small_cout("cpu")


def test_conv3x3_winograd_f4x4_in_two_kernels():
    T.check_conv_winograd4_two_kernels("cpu")


def test_conv3x3_winograd_f4x4_matrix_kernel_persistent_workgroups(monkeypatch):
    """the same cases on a pretend 8-CU device: launches of more than 8 tiles run as ONE workgroup per CU walking several tiles
    (conv_wino4r_kernel<1, true>: the next tile's first loads are issued from the epilogue of the current one, empty padding tiles
    are skipped) -- bit-identical to the one-kernel form when fed its by-product, like the one-tile-per-workgroup launch"""
    monkeypatch.setenv("SSDE_NUM_CUS", "8")
    T.check_conv_winograd4_two_kernels("cpu")


def test_winograd_f4x4_weight_gradient_with_the_stream_k_split():
    T.check_wgrad_wino4_streamk("cpu")


@pytest.mark.parametrize("kind", ["ncsnpp", "ddpmpp"])
def test_deferred_finishing_launches_are_bit_identical(kind, monkeypatch):
    T.check_deferred_finish_is_bit_identical("cpu", monkeypatch, kind)


def test_groupnorm_finalize_with_few_and_with_many_entries_per_group():
    T.check_gn_finalize_entry_counts("cpu")
