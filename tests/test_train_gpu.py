"""Training path on the MI355X through the C ABI: backward / optimizer kernels, whole-network gradients,
the torch.autograd bridge and the fused DSM training step, against torch CPU autograd over the oracle.
Checks and tolerances: tests/_train_checks.py."""
import pytest
import torch

import _util
import _train_checks as T

pytestmark = pytest.mark.gpu


def test_backward_kernels():
    T.check_backward_ops("cuda")


def test_dropout_mask_is_regenerated_identically():
    T.check_dropout_mask("cuda")


def test_upfirdn_lds_tiles():
    T.check_upfirdn_tiles("cuda")


@pytest.mark.parametrize("pipe", ["0", "2"])      # the plain GEMM kernels / the persistent pipelined one (its own partials code)
def test_groupnorm_statistics_from_producer_epilogues(pipe, monkeypatch):
    monkeypatch.setenv("SSDE_GEMM_PIPE", pipe)
    if pipe == "2":
        monkeypatch.setenv("SSDE_NUM_CUS", "3")      # persistent workgroups loop over several tiles
    T.check_fused_gn_statistics("cuda", monkeypatch)


def test_groupnorm_statistics_merged_by_the_transform_pass(monkeypatch):
    # F(4x4,3x3) in two kernels wherever it is legal: the pass merges the partials of its source itself (gn_in_part0)
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    T.check_fused_gn_statistics("cuda", monkeypatch)


def test_conv3x3_winograd_f4x4():
    T.check_conv_winograd4("cuda", big=True)


def test_conv_reduction_split_over_two_workgroups(monkeypatch):
    T.check_conv_split_reduction("cuda", monkeypatch, n=256)


@pytest.mark.parametrize("kind", ["ncsnpp", "ddpmpp", "ffhq"])
def test_whole_network_gradients_small(kind):
    T.check_unet_grads(kind, "cuda", batch=3)


def test_whole_network_gradients_cifar_ncsnpp():
    """the full BASELINE architecture (62.8 M parameters), batch 2"""
    T.check_unet_grads("ncsnpp", "cuda", cfg=_util.cfgs.get_config("ve/cifar10_ncsnpp_continuous", dropout=0.0), batch=2)


def test_whole_network_gradients_winograd_f4x4(monkeypatch):
    """forward and input-gradient convolutions on the F(4x4,3x3) kernel wherever it is legal"""
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    T.check_unet_grads("ncsnpp", "cuda", batch=3)


def test_autograd_bridge():
    T.check_autograd_bridge("cuda")


@pytest.mark.parametrize("sde_kind", ["vesde", "subvpsde", "smld", "ddpm"])
def test_fused_training_step(sde_kind):
    T.check_fused_step("cuda", steps=3, sde_kind=sde_kind)


@pytest.mark.parametrize("graph", ["1", "0"])       # the whole step as one hipGraph replay / as program runs
@pytest.mark.parametrize("name", list(_util.TRAIN_CASES))
def test_step_fn_matches_the_reference_run(name, graph, monkeypatch):
    """three optimisation steps + the eval step of losses.get_step_fn against the REFERENCE's own run of them
    (tests/golden/train_small.npz: /root/reference's losses.py:151-210 + models/ema.py executed by oracle/gen_golden_train.py)"""
    monkeypatch.setenv("SSDE_TRAIN_GRAPH", graph)
    T.check_step_fn_against_reference_run("cuda", name)


@pytest.mark.parametrize("warm", [False, True])
def test_checkpoint_written_by_the_reference_resumes(tmp_path, warm):
    T.check_reference_checkpoint_resume("cuda", tmp_path, warm)


def test_step_fn_selects_fused_path_and_trains():
    """get_step_fn end to end with torch's own RNG (no injection): loss is finite and parameters move"""
    from score_sde_pytorch_amd.models import utils as mutils, ema as ema_mod
    from score_sde_pytorch_amd import losses, sde_lib
    cfg = _util.small_config("ncsnpp")
    cfg.model.dropout = 0.1
    torch.manual_seed(0)
    model = mutils.get_model("ncsnpp")(cfg)
    _util.load_seeded(model, seed=1)
    model = model.cuda()
    sde = sde_lib.VESDE(cfg.model.sigma_min, cfg.model.sigma_max, cfg.model.num_scales)
    opt = losses.get_optimizer(cfg, model.parameters())
    ema = ema_mod.ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg))
    eval_fn = losses.get_step_fn(sde, train=False, optimize_fn=losses.optimization_manager(cfg))
    state = dict(optimizer=opt, model=model, ema=ema, step=0)
    p0 = model.all_modules[3].weight.detach().clone()
    batch = torch.rand(4, 3, 16, 16, device="cuda")
    ls = [float(step_fn(state, batch)) for _ in range(4)]
    assert all(l == l and l > 0 for l in ls) and state["step"] == 4
    assert float((model.all_modules[3].weight.detach() - p0).abs().max()) > 0
    le = float(eval_fn(state, batch))
    assert le == le and le > 0


@pytest.mark.parametrize("kind", ["ncsnpp", "ffhq"])
def test_device_weight_repack(kind):
    T.check_device_repack("cuda", kind)


def test_device_weight_repack_of_the_per_lane_f4x4_image(monkeypatch):
    """every legal 3x3 layer on the two-kernel F(4x4,3x3) form: SSDE_PACK_WINO4R (conv_wino4r.hip's per-lane image), forward and
    input-gradient variants, against engine.pack_wino4r_weight"""
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    monkeypatch.setenv("SSDE_WINO4_TWO", "2")
    T.check_device_repack("cuda", "ffhq", need_kind=6)


def test_op_package():
    T.check_op_package("cuda")


def test_op_package_refuses_cpu():
    import score_sde_pytorch_amd.op as op
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        op.upfirdn2d(torch.zeros(1, 4, 4, 4), torch.ones(2, 2))


def test_input_gradient_only_program():
    T.check_input_gradient_only("cuda")


def test_likelihood_matches_reference_fixture():
    T.check_likelihood("cuda")


@pytest.mark.parametrize("denoise", [False, True])
def test_ode_sampler_matches_reference_fixture(denoise):
    T.check_ode_sampler("cuda", denoise)


def test_fused_likelihood_right_hand_side():
    T.check_fused_likelihood_rhs("cuda")


def test_fused_ode_drift_on_a_discrete_label_ve_model():
    T.check_fused_drift_discrete_ve("cuda")


def test_checkpoint_roundtrip_and_ema_swap(tmp_path):
    T.check_checkpoint_and_ema_swap("cuda", tmp_path)


def test_bucketed_gradient_exchange_over_a_one_rank_rccl_group(monkeypatch):
    T.check_forced_exchange_one_rank("cuda", "nccl", monkeypatch)


def test_generic_loss_closures_run_the_hip_loss_head():
    T.check_generic_loss_closures("cuda")


def test_1x1_weight_gradients_pipelined_and_chunked():
    T.check_wgrad_1x1("cuda")


@pytest.mark.parametrize("dropout", [False, True])
def test_weight_gradient_fed_by_the_forward_launch(monkeypatch, dropout):
    T.check_wino_v_from_forward("cuda", monkeypatch, dropout=dropout)


def test_conv3x3_winograd_f4x4_register_fed_matrix_kernel():
    T.check_conv_winograd4("cuda", big=True, regs=True)


def test_conv3x3_onto_the_image_channels():
    T.check_conv_small_cout("cuda", big=True)


def test_conv3x3_winograd_f4x4_in_two_kernels():
    T.check_conv_winograd4_two_kernels("cuda", big=True)


def test_conv3x3_winograd_f4x4_matrix_kernel_persistent_workgroups(monkeypatch):
    """the same cases on a pretend 8-CU device: launches of more than 8 tiles run as ONE workgroup per CU walking several tiles
    (conv_wino4r_kernel<1, true>: the next tile's first loads are issued from the epilogue of the current one, empty padding tiles
    are skipped) -- bit-identical to the one-kernel form when fed its by-product, like the one-tile-per-workgroup launch"""
    monkeypatch.setenv("SSDE_NUM_CUS", "8")
    T.check_conv_winograd4_two_kernels("cuda", big=True)


def test_whole_network_gradients_winograd_f4x4_in_two_kernels(monkeypatch):
    """forward and input-gradient convolutions as transform pass + matrix kernel; weight gradients fed by the transform pass"""
    monkeypatch.setenv("SSDE_WINOGRAD", "4")
    monkeypatch.setenv("SSDE_WINO4_TWO", "2")
    monkeypatch.setenv("SSDE_WGRAD_WINOGRAD", "44")
    T.check_unet_grads("ncsnpp", "cuda", batch=3)


def test_winograd_f4x4_weight_gradient_with_the_stream_k_split():
    T.check_wgrad_wino4_streamk("cuda", big=True)


@pytest.mark.parametrize("kind", ["ncsnpp", "ddpmpp"])
def test_deferred_finishing_launches_are_bit_identical(kind, monkeypatch):
    T.check_deferred_finish_is_bit_identical("cuda", monkeypatch, kind)


def test_groupnorm_finalize_with_few_and_with_many_entries_per_group():
    T.check_gn_finalize_entry_counts("cuda")
