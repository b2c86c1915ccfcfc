"""ctypes binding of libssde_hip.so (C ABI declared in include/ssde.h).

The structures below mirror include/ssde.h field for field; `load()` verifies the
mirror against the library (`ssde_sizeof_op`, `ssde_abi_version`) and fails loudly
if the shared object is missing -- there is no CPU or eager-PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSDE_LIB_PATH: developer switch for A/B timing of kernel variants built by _build.build_variant (tools/ab_bench.sh)
LIB_PATH = os.environ.get("SSDE_LIB_PATH") or os.path.join(_HERE, "libssde_hip.so")
ABI_VERSION = 10

PRO_NONE, PRO_GN, PRO_GN_SILU, PRO_SILU = 0, 1, 2, 3
TILE_AUTO, TILE_256x64, TILE_128x64, TILE_64x64, TILE_256x32, TILE_WINOGRAD, TILE_WINOGRAD4 = 0, 1, 2, 3, 4, 5, 6
TILE_WINOGRAD4R = 9       # (7, 8: the bf16-split and the LDS-fed F(4x4,3x3) kernels of round 4, tools/experiments/)
TILES_WINOGRAD4 = (TILE_WINOGRAD4, TILE_WINOGRAD4R)
# routing switches of a launch (include/ssde.h: SSDE_CONVF_*, SSDE_WGRADF_*, SSDE_GNBWDF_*)
CONVF_V_GIVEN, CONVF_BF16X6, CONVF_NO_KSPLIT, CONVF_BKC8, CONVF_GEMM_PIPE, CONVF_NO_GEMM_PIPE, CONVF_X6_BM64, CONVF_X6_PF2, \
    CONVF_NO_SMALL_COUT, CONVF_X6_WIDE, CONVF_X6_NO_WIDE = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024
WGRADF_DIRECT, WGRADF_F2, WGRADF_F4_FORCE, WGRADF_NO_STREAMK, WGRADF_NO_XCD_ORDER, WGRADF_1X1_CHUNKED, WGRADF_XVEC1 = 1, 2, 4, 8, 16, 32, 64
GNBWDF_THREE_KERNELS, GNBWDF_DEFER_PARAMS = 1, 2
(OP_CONV, OP_GN_STATS, OP_UPFIRDN, OP_ATTN, OP_EMBED, OP_TO_NHWC, OP_TO_NCHW, OP_BIAS_ACT, OP_SUMSQ,
 OP_RANDN, OP_LANGEVIN, OP_PREDICTOR, OP_FILL, OP_STEP_INC, OP_WGRAD, OP_COLSUM, OP_GN_BWD_REDUCE, OP_PROLOGUE_BWD,
 OP_ATTN_BWD, OP_PERTURB, OP_DSM_LOSS, OP_SUMSQ_FLAT, OP_ADAM, OP_MEMSET, OP_AXPY, OP_PACK, OP_PROJECT,
 OP_GN_FINALIZE, OP_PF_DRIFT, OP_HUTCH_DIV, OP_COLSUM_FINISH, OP_GN_BWD_FINISH) = range(1, 33)
FINISH_JOBS = 16
COLSUMF_DEFER = 1
PACK_CONV3, PACK_WINO3, PACK_MATRIX, PACK_VECTOR, PACK_WINO4, PACK_WINO4R = 1, 2, 3, 4, 5, 6

_fp = C.c_void_p  # device pointers are passed as integers


class Src(C.Structure):
    _fields_ = [("p0", _fp), ("p1", _fp), ("c0", C.c_int32), ("c1", C.c_int32),
                ("pro_mode", C.c_int32), ("gn_groups", C.c_int32),
                ("gn_mean", _fp), ("gn_rstd", _fp), ("gn_gamma", _fp), ("gn_beta", _fp),
                ("drop_thresh", C.c_uint32), ("drop_scale", C.c_float), ("drop_seed", _fp),
                ("drop_salt", C.c_uint32), ("_pad", C.c_int32)]


class ConvArgs(C.Structure):
    _fields_ = [("main", Src), ("aux", Src), ("w_main", _fp), ("w_aux", _fp),
                ("n", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32),
                ("h_out", C.c_int32), ("w_out", C.c_int32), ("c_out", C.c_int32),
                ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("tile", C.c_int32),
                ("bias", _fp), ("chan_add", _fp), ("chan_add_ld", C.c_int32), ("resid_post", C.c_int32),
                ("resid", _fp), ("out_scale", C.c_float), ("flags", C.c_uint32), ("dst", _fp), ("gn_part", _fp), ("wino_v", _fp),
                ("gn_in_part0", _fp), ("gn_in_part1", _fp), ("gn_in_slices0", C.c_int32), ("gn_in_slices1", C.c_int32),
                ("gn_in_eps", C.c_float), ("_pad_gn_in", C.c_int32)]


class GnStatsArgs(C.Structure):
    _fields_ = [("p0", _fp), ("p1", _fp), ("c0", C.c_int32), ("c1", C.c_int32),
                ("n", C.c_int32), ("hw", C.c_int32), ("groups", C.c_int32), ("eps", C.c_float),
                ("mean", _fp), ("rstd", _fp), ("scratch", _fp), ("slices", C.c_int32), ("_pad0", C.c_int32)]


class RkCombineArgs(C.Structure):
    _fields_ = [("y", _fp), ("k", _fp), ("n", C.c_int64), ("terms", C.c_int32), ("_pad0", C.c_int32), ("coef", C.c_double * 7),
                ("dst", _fp), ("dst32", _fp), ("n32", C.c_int64)]


class RkErrorArgs(C.Structure):
    _fields_ = [("y", _fp), ("y_new", _fp), ("k", _fp), ("n", C.c_int64), ("coef", C.c_double * 7), ("atol", C.c_double),
                ("rtol", C.c_double), ("partial", _fp), ("partial_len", C.c_int32), ("_pad0", C.c_int32), ("out", _fp)]


class PfDriftArgs(C.Structure):
    _fields_ = [("x", _fp), ("score", _fp), ("dst", _fp), ("numel", C.c_int64), ("a", C.c_float), ("g2", C.c_float), ("dyn", _fp)]


class OdeDyn(C.Structure):
    """per-evaluation scalars of an ODE right-hand side (device record, include/ssde.h: ssde_ode_dyn)"""
    _fields_ = [("label", C.c_float), ("std", C.c_float), ("a", C.c_float), ("g2", C.c_float), ("dst", _fp)]


class HutchDivArgs(C.Structure):
    _fields_ = [("gx", _fp), ("eps", _fp), ("dst", _fp), ("dst_off", C.c_int64), ("n", C.c_int32), ("per", C.c_int32),
                ("a", C.c_float), ("g2", C.c_float), ("dyn", _fp)]


class GnFinalizeArgs(C.Structure):
    _fields_ = [("part0", _fp), ("part1", _fp), ("c0", C.c_int32), ("c1", C.c_int32), ("slices0", C.c_int32), ("slices1", C.c_int32),
                ("n", C.c_int32), ("groups", C.c_int32), ("eps", C.c_float), ("_pad0", C.c_int32), ("mean", _fp), ("rstd", _fp)]


class UpfirdnArgs(C.Structure):
    _fields_ = [("src", Src), ("n", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32), ("c", C.c_int32),
                ("h_out", C.c_int32), ("w_out", C.c_int32),
                ("up", C.c_int32), ("down", C.c_int32), ("pad0", C.c_int32), ("pad1", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("k", C.c_float * 16), ("dst", _fp),
                ("accumulate", C.c_int32), ("_pad0", C.c_int32), ("dst2", _fp)]


class AttnArgs(C.Structure):
    _fields_ = [("qkv", _fp), ("dst", _fp), ("n", C.c_int32), ("l", C.c_int32), ("c", C.c_int32), ("scale", C.c_float),
                ("flags", C.c_uint32), ("_pad0", C.c_int32)]


class EmbedArgs(C.Structure):
    _fields_ = [("cond", _fp), ("w", _fp), ("dst", _fp), ("n", C.c_int32), ("dim", C.c_int32),
                ("kind", C.c_int32), ("_pad0", C.c_int32)]


class ToNhwcArgs(C.Structure):
    _fields_ = [("src", _fp), ("dst", _fp), ("n", C.c_int32), ("c", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("c_pad", C.c_int32), ("a", C.c_float), ("b", C.c_float), ("mode", C.c_int32), ("v", _fp)]


class ToNchwArgs(C.Structure):
    _fields_ = [("src", _fp), ("dst", _fp), ("n", C.c_int32), ("c", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("c_src", C.c_int32), ("mode", C.c_int32), ("v", _fp), ("alpha", C.c_float), ("accumulate", C.c_int32)]


class BiasActArgs(C.Structure):
    _fields_ = [("src", _fp), ("bias", _fp), ("dst", _fp), ("numel", C.c_int64), ("channels", C.c_int32),
                ("inner", C.c_int32), ("act", C.c_int32), ("alpha", C.c_float), ("scale", C.c_float), ("grad", C.c_int32),
                ("ref", _fp)]


class SumsqArgs(C.Structure):
    _fields_ = [("a", _fp), ("b", _fp), ("out_a", _fp), ("out_b", _fp), ("n", C.c_int32), ("per", C.c_int32)]


class RandnArgs(C.Structure):
    _fields_ = [("dst", _fp), ("numel", C.c_int64), ("seed", C.c_uint64), ("step_ptr", _fp),
                ("stream_id", C.c_int32), ("_pad0", C.c_int32), ("seed_ptr", _fp)]


class LangevinArgs(C.Structure):
    _fields_ = [("x", _fp), ("x_mean", _fp), ("grad", _fp), ("noise", _fp), ("grad_sumsq", _fp), ("noise_sumsq", _fp),
                ("alpha_tab", _fp), ("step_ptr", _fp), ("n", C.c_int32), ("per", C.c_int32), ("snr", C.c_float),
                ("_pad0", C.c_int32)]


class PredictorArgs(C.Structure):
    _fields_ = [("x", _fp), ("x_mean", _fp), ("score", _fp), ("noise", _fp), ("coef", _fp), ("step_ptr", _fp),
                ("numel", C.c_int64)]


class SampleUpdateArgs(C.Structure):
    _fields_ = [("x", _fp), ("y", _fp), ("z", _fp), ("a", _fp), ("b", _fp), ("c", _fp), ("x_mean", _fp), ("x_out", _fp),
                ("n", C.c_int32), ("per", C.c_int32)]


class FillArgs(C.Structure):
    _fields_ = [("dst", _fp), ("tab", _fp), ("step_ptr", _fp), ("n", C.c_int32), ("_pad0", C.c_int32)]


class StepIncArgs(C.Structure):
    _fields_ = [("step_ptr", _fp), ("delta", C.c_int32), ("_pad0", C.c_int32)]


class ProjectArgs(C.Structure):
    _fields_ = [("x", _fp), ("x_mean", _fp), ("data", _fp), ("mask", _fp), ("noise", _fp), ("coef", _fp), ("step_ptr", _fp),
                ("n", C.c_int32), ("c", C.c_int32), ("hw", C.c_int32), ("use_matrix", C.c_int32),
                ("M", C.c_float * 9), ("invM", C.c_float * 9)]


class WgradArgs(C.Structure):
    _fields_ = [("src", Src), ("g", _fp), ("g_ld", C.c_int32), ("g_off", C.c_int32),
                ("n", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32), ("h_out", C.c_int32), ("w_out", C.c_int32),
                ("c_out", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("cin_store", C.c_int32), ("transpose_out", C.c_int32), ("splits", C.c_int32), ("scale", C.c_float),
                ("flags", C.c_uint32), ("dw", _fp), ("scratch", _fp), ("scratch_floats", C.c_int64), ("v_pre", _fp)]


class ColsumArgs(C.Structure):
    _fields_ = [("g", _fp), ("g_ld", C.c_int32), ("g_off", C.c_int32), ("n", C.c_int32), ("hw", C.c_int32),
                ("c", C.c_int32), ("scale", C.c_float), ("per_sample", _fp), ("ps_ld", C.c_int32), ("ps_off", C.c_int32),
                ("total", _fp), ("total2", _fp), ("scratch", _fp), ("flags", C.c_uint32), ("_pad0", C.c_int32)]


class ColsumJob(C.Structure):
    _fields_ = [("part", _fp), ("per_sample", _fp), ("total", _fp), ("total2", _fp), ("n", C.c_int32), ("slices", C.c_int32),
                ("c", C.c_int32), ("ps_ld", C.c_int32), ("ps_off", C.c_int32), ("_pad0", C.c_int32)]


class ColsumFinishArgs(C.Structure):
    _fields_ = [("count", C.c_int32), ("_pad0", C.c_int32), ("job", ColsumJob * FINISH_JOBS)]


class GnBwdJob(C.Structure):
    _fields_ = [("scratch", _fp), ("dgamma", _fp), ("dbeta", _fp), ("rows", C.c_int32), ("c", C.c_int32)]


class GnBwdFinishArgs(C.Structure):
    _fields_ = [("count", C.c_int32), ("_pad0", C.c_int32), ("job", GnBwdJob * FINISH_JOBS)]


class GnBwdReduceArgs(C.Structure):
    _fields_ = [("src", Src), ("dp", _fp), ("n", C.c_int32), ("hw", C.c_int32), ("sums", _fp),
                ("dgamma", _fp), ("dbeta", _fp), ("scratch", _fp), ("slices", C.c_int32), ("flags", C.c_uint32),
                ("g0", _fp), ("g1", _fp), ("acc0", C.c_int32), ("acc1", C.c_int32), ("scale", C.c_float), ("_pad1", C.c_int32)]


class PrologueBwdArgs(C.Structure):
    _fields_ = [("src", Src), ("dp", _fp), ("dp_ld", C.c_int32), ("dp_off", C.c_int32), ("n", C.c_int32), ("hw", C.c_int32),
                ("sums", _fp), ("scale", C.c_float), ("acc0", C.c_int32), ("acc1", C.c_int32), ("g0", _fp), ("g1", _fp)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("qkv", _fp), ("o", _fp), ("d_o", _fp), ("dqkv", _fp), ("stats", _fp),
                ("n", C.c_int32), ("l", C.c_int32), ("c", C.c_int32), ("scale", C.c_float)]


class PerturbArgs(C.Structure):
    _fields_ = [("x", _fp), ("z", _fp), ("a", _fp), ("s", _fp), ("dst", _fp), ("n", C.c_int32), ("per", C.c_int32)]


class DsmLossArgs(C.Structure):
    _fields_ = [("score", _fp), ("z", _fp), ("s", _fp), ("g2", _fp), ("dscore", _fp), ("losses", _fp), ("loss", _fp),
                ("n", C.c_int32), ("per", C.c_int32), ("reduce_mean", C.c_int32), ("likelihood_weighting", C.c_int32),
                ("grad_scale", C.c_float), ("_pad0", C.c_int32)]


class SumsqFlatArgs(C.Structure):
    _fields_ = [("x", _fp), ("numel", C.c_int64), ("partial", _fp), ("out", _fp)]


class AdamArgs(C.Structure):
    _fields_ = [("p", _fp), ("g", _fp), ("m", _fp), ("v", _fp), ("ema", _fp), ("numel", C.c_int64),
                ("hyper", _fp), ("gnorm_sq", _fp)]


class MemsetArgs(C.Structure):
    _fields_ = [("dst", _fp), ("bytes", C.c_int64), ("value", C.c_int32), ("_pad0", C.c_int32)]


class AxpyArgs(C.Structure):
    _fields_ = [("x", _fp), ("gate", _fp), ("dst", _fp), ("numel", C.c_int64), ("alpha", C.c_float), ("acc", C.c_int32)]


class PackDesc(C.Structure):
    _fields_ = [("src", _fp), ("src2", _fp), ("dst", _fp), ("kind", C.c_int32), ("cout", C.c_int32), ("cin", C.c_int32),
                ("cout_l", C.c_int32), ("cin_l", C.c_int32), ("flags", C.c_int32), ("r_off", C.c_int32), ("c_off", C.c_int32),
                ("n", C.c_int64)]


class PackArgs(C.Structure):
    _fields_ = [("table", _fp), ("count", C.c_int32), ("kind", C.c_int32), ("max_n", C.c_int64)]


class _OpUnion(C.Union):
    _fields_ = [("conv", ConvArgs), ("gn", GnStatsArgs), ("fir", UpfirdnArgs), ("attn", AttnArgs),
                ("embed", EmbedArgs), ("to_nhwc", ToNhwcArgs), ("to_nchw", ToNchwArgs), ("bias_act", BiasActArgs),
                ("sumsq", SumsqArgs), ("randn", RandnArgs), ("langevin", LangevinArgs), ("predictor", PredictorArgs),
                ("fill", FillArgs), ("step_inc", StepIncArgs),
                ("wgrad", WgradArgs), ("colsum", ColsumArgs), ("gn_bwd", GnBwdReduceArgs), ("pro_bwd", PrologueBwdArgs),
                ("attn_bwd", AttnBwdArgs), ("perturb", PerturbArgs), ("dsm_loss", DsmLossArgs),
                ("sumsq_flat", SumsqFlatArgs), ("adam", AdamArgs), ("memset", MemsetArgs), ("axpy", AxpyArgs),
                ("pack", PackArgs), ("project", ProjectArgs), ("gn_fin", GnFinalizeArgs),
                ("pf_drift", PfDriftArgs), ("hutch_div", HutchDivArgs), ("colsum_fin", ColsumFinishArgs), ("gn_bwd_fin", GnBwdFinishArgs)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("flops_class", C.c_int32), ("u", _OpUnion)]


_UNION_FIELD = {OP_CONV: "conv", OP_GN_STATS: "gn", OP_UPFIRDN: "fir", OP_ATTN: "attn", OP_EMBED: "embed",
                OP_TO_NHWC: "to_nhwc", OP_TO_NCHW: "to_nchw", OP_BIAS_ACT: "bias_act", OP_SUMSQ: "sumsq",
                OP_RANDN: "randn", OP_LANGEVIN: "langevin", OP_PREDICTOR: "predictor", OP_FILL: "fill",
                OP_STEP_INC: "step_inc", OP_WGRAD: "wgrad", OP_COLSUM: "colsum", OP_GN_BWD_REDUCE: "gn_bwd",
                OP_PROLOGUE_BWD: "pro_bwd", OP_ATTN_BWD: "attn_bwd", OP_PERTURB: "perturb", OP_DSM_LOSS: "dsm_loss",
                OP_SUMSQ_FLAT: "sumsq_flat", OP_ADAM: "adam", OP_MEMSET: "memset", OP_AXPY: "axpy", OP_PACK: "pack", OP_PROJECT: "project",
                OP_GN_FINALIZE: "gn_fin", OP_PF_DRIFT: "pf_drift", OP_HUTCH_DIV: "hutch_div", OP_COLSUM_FINISH: "colsum_fin",
                OP_GN_BWD_FINISH: "gn_bwd_fin"}

EXPORTS = ["ssde_conv2d", "ssde_groupnorm_stats", "ssde_upfirdn2d", "ssde_attention", "ssde_embed", "ssde_to_nhwc",
           "ssde_to_nchw", "ssde_fused_bias_act", "ssde_sumsq", "ssde_randn", "ssde_langevin_update",
           "ssde_predictor_update", "ssde_fill_from_table", "ssde_step_inc", "ssde_program_run",
           "ssde_program_run_timed", "ssde_graph_capture", "ssde_graph_launch", "ssde_graph_destroy",
           "ssde_abi_version", "ssde_sizeof_op", "ssde_last_error", "ssde_conv_lds_bytes",
           "ssde_conv_wgrad", "ssde_colsum", "ssde_gn_bwd_reduce", "ssde_prologue_bwd", "ssde_attention_bwd",
           "ssde_perturb", "ssde_dsm_loss", "ssde_sumsq_flat", "ssde_adam_clip_ema", "ssde_memset", "ssde_axpy",
           "ssde_wgrad_scratch_floats", "ssde_wgrad_wants_winograd4", "ssde_pack_weights", "ssde_project_update", "ssde_gn_finalize", "ssde_conv_gn_slices", "ssde_rk_combine", "ssde_rk_error_norm", "ssde_pf_drift", "ssde_hutch_div", "ssde_sample_update", "ssde_mfma_probe", "ssde_colsum_finish", "ssde_gn_bwd_finish", "ssde_gn_bwd_scratch_rows",
           # plan-level entry points (csrc/plan.hip; argument types: plan_export.bind)
           "ssde_plan_load", "ssde_plan_load_file", "ssde_plan_destroy", "ssde_plan_info", "ssde_plan_param",
           "ssde_plan_refresh_weights", "ssde_unet_forward", "ssde_pc_reset", "ssde_pc_run", "ssde_pc_state",
           "ssde_train_step", "ssde_train_forward", "ssde_unet_backward", "ssde_plan_copy_io"]

_lib = None


# ---- environment -> routing flags -----------------------------------------------------------------------------------------
# The library itself reads no routing switch from the environment (ABI 8): a launch's route is part of its arguments.  For
# A/B runs and the tests of alternative kernels the HOST side maps the round-2..4 variables to flags at the moment it builds
# arguments (engine.ProgramBuilder.finalize, the hipops wrappers), so a lowered program -- and a plan blob exported from it --
# carries its routes with it and two programs in one process may differ.
def conv_route_flags(env=None):
    e = os.environ if env is None else env
    f = 0
    if e.get("SSDE_MATRIX", "").startswith("b"):
        f |= CONVF_BF16X6
    if e.get("SSDE_CONV_KSPLIT", "1") == "0":
        f |= CONVF_NO_KSPLIT
    if e.get("SSDE_CONV_BKC64", "") == "8":
        f |= CONVF_BKC8
    if e.get("SSDE_GEMM_PIPE", "") == "0":
        f |= CONVF_NO_GEMM_PIPE
    if e.get("SSDE_GEMM_PIPE", "") == "2":
        f |= CONVF_GEMM_PIPE
    if e.get("SSDE_X6_BM", "") == "64":
        f |= CONVF_X6_BM64
    if e.get("SSDE_X6_PF", "") == "2":
        f |= CONVF_X6_PF2
    if e.get("SSDE_CONV_SMALL", "1") == "0":
        f |= CONVF_NO_SMALL_COUT
    if e.get("SSDE_X6_WIDE", "") == "1":
        f |= CONVF_X6_WIDE
    if e.get("SSDE_X6_WIDE", "") == "0":
        f |= CONVF_X6_NO_WIDE
    return f


ATTNF_BF16X6 = 1


def attn_route_flags(env=None):
    """SSDE_MATRIX=bf16x6 also moves the attention forward onto the BF16 matrix pipe (SSDE_ATTN_X6=0 keeps the fp32 kernel: A/B runs)"""
    e = os.environ if env is None else env
    return ATTNF_BF16X6 if e.get("SSDE_MATRIX", "").startswith("b") and e.get("SSDE_ATTN_X6", "1") != "0" else 0


def wgrad_route_flags(env=None):
    e = os.environ if env is None else env
    f = {"0": WGRADF_DIRECT, "2": WGRADF_F2, "44": WGRADF_F4_FORCE}.get(e.get("SSDE_WGRAD_WINOGRAD", ""), 0)
    if e.get("SSDE_WGRAD4_STREAMK", "1") == "0":
        f |= WGRADF_NO_STREAMK
    if e.get("SSDE_WGRAD4_XCD", "1") == "0":
        f |= WGRADF_NO_XCD_ORDER
    if e.get("SSDE_WGRAD_1X1_PIPELINED", "1") == "0":
        f |= WGRADF_1X1_CHUNKED
    if e.get("SSDE_WGRAD4_XVEC", "2") == "1":
        f |= WGRADF_XVEC1
    return f


def gn_bwd_route_flags(env=None):
    e = os.environ if env is None else env
    return GNBWDF_THREE_KERNELS if e.get("SSDE_GN_BWD_FUSED", "1") == "0" else 0


class SsdeError(RuntimeError):
    pass


def bind(lib):
    """Declare argument types on a loaded library and verify it against this mirror."""
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise SsdeError("libssde_hip.so does not export %s" % name)
    lib.ssde_last_error.restype = C.c_char_p
    lib.ssde_program_run.argtypes = [C.POINTER(Op), C.c_int32, C.c_void_p]
    lib.ssde_program_run_timed.argtypes = [C.POINTER(Op), C.c_int32, C.c_void_p, C.POINTER(C.c_float)]
    lib.ssde_graph_capture.argtypes = [C.POINTER(Op), C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.ssde_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
    lib.ssde_graph_destroy.argtypes = [C.c_void_p]
    for name, typ in [("ssde_conv2d", ConvArgs), ("ssde_groupnorm_stats", GnStatsArgs), ("ssde_upfirdn2d", UpfirdnArgs),
                      ("ssde_attention", AttnArgs), ("ssde_embed", EmbedArgs), ("ssde_to_nhwc", ToNhwcArgs),
                      ("ssde_to_nchw", ToNchwArgs), ("ssde_fused_bias_act", BiasActArgs), ("ssde_sumsq", SumsqArgs),
                      ("ssde_randn", RandnArgs), ("ssde_langevin_update", LangevinArgs),
                      ("ssde_predictor_update", PredictorArgs), ("ssde_fill_from_table", FillArgs),
                      ("ssde_step_inc", StepIncArgs), ("ssde_conv_wgrad", WgradArgs), ("ssde_colsum", ColsumArgs),
                      ("ssde_gn_bwd_reduce", GnBwdReduceArgs), ("ssde_prologue_bwd", PrologueBwdArgs),
                      ("ssde_attention_bwd", AttnBwdArgs), ("ssde_perturb", PerturbArgs), ("ssde_dsm_loss", DsmLossArgs),
                      ("ssde_sumsq_flat", SumsqFlatArgs), ("ssde_adam_clip_ema", AdamArgs), ("ssde_memset", MemsetArgs),
                      ("ssde_axpy", AxpyArgs), ("ssde_pack_weights", PackArgs), ("ssde_project_update", ProjectArgs),
                      ("ssde_gn_finalize", GnFinalizeArgs), ("ssde_rk_combine", RkCombineArgs), ("ssde_hutch_div", HutchDivArgs), ("ssde_sample_update", SampleUpdateArgs),
                      ("ssde_rk_error_norm", RkErrorArgs), ("ssde_pf_drift", PfDriftArgs),
                      ("ssde_colsum_finish", ColsumFinishArgs), ("ssde_gn_bwd_finish", GnBwdFinishArgs)]:
        getattr(lib, name).argtypes = [C.POINTER(typ), C.c_void_p]
    lib.ssde_conv_lds_bytes.argtypes = [C.POINTER(ConvArgs)]
    lib.ssde_conv_gn_slices.argtypes = [C.POINTER(ConvArgs)]
    lib.ssde_wgrad_scratch_floats.argtypes = [C.POINTER(WgradArgs)]
    lib.ssde_wgrad_scratch_floats.restype = C.c_int64
    lib.ssde_wgrad_wants_winograd4.argtypes = [C.POINTER(WgradArgs)]
    lib.ssde_gn_bwd_scratch_rows.argtypes = [C.POINTER(GnBwdReduceArgs)]
    if lib.ssde_abi_version() != ABI_VERSION:
        raise SsdeError("ABI mismatch: library %d, binding %d" % (lib.ssde_abi_version(), ABI_VERSION))
    if lib.ssde_sizeof_op() != C.sizeof(Op):
        raise SsdeError("struct layout mismatch: sizeof(ssde_op) library %d, ctypes %d" % (lib.ssde_sizeof_op(), C.sizeof(Op)))
    return lib


def load():
    """Load (once) and return the ctypes handle; raises if the library is absent or mismatched."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SsdeError(
            "libssde_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(score_sde_pytorch_amd has no CPU/eager fallback)" % LIB_PATH)
    lib = bind(C.CDLL(LIB_PATH))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ssde_last_error()
        raise SsdeError("%s failed (%d): %s" % (what or "libssde_hip call", rc, msg.decode() if msg else "?"))


def pointer_offsets(struct_cls, base=0):
    """Byte offsets of every pointer field of a ctypes structure: nested structures and arrays of structures included."""
    out = []
    for name, typ in struct_cls._fields_:
        off = base + getattr(struct_cls, name).offset
        if typ is C.c_void_p:
            out.append(off)
        elif isinstance(typ, type) and issubclass(typ, C.Structure):
            out.extend(pointer_offsets(typ, off))
        elif isinstance(typ, type) and issubclass(typ, C.Array) and issubclass(typ._type_, C.Structure):
            for i in range(typ._length_):
                out.extend(pointer_offsets(typ._type_, off + i * C.sizeof(typ._type_)))
    return out


def make_op(kind, args, flops_class=0):
    op = Op()
    op.kind = kind
    op.flops_class = flops_class
    setattr(op.u, _UNION_FIELD[kind], args)
    return op


def op_array(ops):
    arr = (Op * len(ops))()
    for i, o in enumerate(ops):
        arr[i] = o
    return arr
