// Error reporting, program runner, hipGraph capture for libssde_hip.so.
#include "ssde_common.h"
#include <string.h>
#include <vector>

static thread_local char g_err[512] = "";

void ssde_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ssde_last_error(void) { return g_err; }

int ssde_num_cus() {
  // SSDE_NUM_CUS overrides (read per call: the tests run persistent kernels on a pretend 3-CU device so that workgroups
  // loop over several tiles; engine.Lowering reads the same variable)
  if (const char* e = getenv("SSDE_NUM_CUS")) { const int v = atoi(e); if (v > 0) return v; }
  // cached per device id: a process may drive devices of different size (partition modes, mixed nodes), and the planners
  // (conv_wino4 splits, stream-K groups, gn_bwd_fused_shape) must agree with the device the launch goes to
  constexpr int kMaxDev = 64;
  static std::atomic<int> cus[kMaxDev];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
  int c = cus[dev].load(std::memory_order_relaxed);
  if (c <= 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev].store(c = v, std::memory_order_relaxed);
  }
  return c;
}
extern "C" int ssde_abi_version(void) { return SSDE_ABI_VERSION; }
extern "C" int ssde_sizeof_op(void) { return (int)sizeof(ssde_op); }

// ---- ssde_mfma_probe: the matrix rate the device sustains (see include/ssde.h) ----
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_probe_kernel(int iters, float* sink) {
  probe_f32x16 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float a = 1.0f + (float)(threadIdx.x & 7) * 0.125f, b = 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);   // 8 independent chains
  }
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) t += acc[j][0];
  if (t == -1.f) sink[threadIdx.x & 63] = t;          // never true: keeps the chains alive
}

extern "C" int ssde_mfma_probe(int32_t workgroups, int32_t iters, float* sink, void* stream) {
  SSDE_REQUIRE(workgroups > 0 && iters > 0 && sink, "mfma_probe: bad args");
  // a workgroup is 4 waves = one wave per SIMD of the CU it lands on
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(workgroups), dim3(256), 0, static_cast<hipStream_t>(stream), iters, sink);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

static int run_one(const ssde_op& op, void* stream) {
  switch (op.kind) {
    case SSDE_OP_CONV: return ssde_conv2d(&op.u.conv, stream);
    case SSDE_OP_GN_STATS: return ssde_groupnorm_stats(&op.u.gn, stream);
    case SSDE_OP_UPFIRDN: return ssde_upfirdn2d(&op.u.fir, stream);
    case SSDE_OP_ATTN: return ssde_attention(&op.u.attn, stream);
    case SSDE_OP_EMBED: return ssde_embed(&op.u.embed, stream);
    case SSDE_OP_TO_NHWC: return ssde_to_nhwc(&op.u.to_nhwc, stream);
    case SSDE_OP_TO_NCHW: return ssde_to_nchw(&op.u.to_nchw, stream);
    case SSDE_OP_BIAS_ACT: return ssde_fused_bias_act(&op.u.bias_act, stream);
    case SSDE_OP_SUMSQ: return ssde_sumsq(&op.u.sumsq, stream);
    case SSDE_OP_RANDN: return ssde_randn(&op.u.randn, stream);
    case SSDE_OP_LANGEVIN: return ssde_langevin_update(&op.u.langevin, stream);
    case SSDE_OP_PROJECT: return ssde_project_update(&op.u.project, stream);
    case SSDE_OP_PREDICTOR: return ssde_predictor_update(&op.u.predictor, stream);
    case SSDE_OP_FILL: return ssde_fill_from_table(&op.u.fill, stream);
    case SSDE_OP_STEP_INC: return ssde_step_inc(&op.u.step_inc, stream);
    case SSDE_OP_WGRAD: return ssde_conv_wgrad(&op.u.wgrad, stream);
    case SSDE_OP_COLSUM: return ssde_colsum(&op.u.colsum, stream);
    case SSDE_OP_GN_BWD_REDUCE: return ssde_gn_bwd_reduce(&op.u.gn_bwd, stream);
    case SSDE_OP_PROLOGUE_BWD: return ssde_prologue_bwd(&op.u.pro_bwd, stream);
    case SSDE_OP_ATTN_BWD: return ssde_attention_bwd(&op.u.attn_bwd, stream);
    case SSDE_OP_PERTURB: return ssde_perturb(&op.u.perturb, stream);
    case SSDE_OP_DSM_LOSS: return ssde_dsm_loss(&op.u.dsm_loss, stream);
    case SSDE_OP_SUMSQ_FLAT: return ssde_sumsq_flat(&op.u.sumsq_flat, stream);
    case SSDE_OP_ADAM: return ssde_adam_clip_ema(&op.u.adam, stream);
    case SSDE_OP_MEMSET: return ssde_memset(&op.u.memset, stream);
    case SSDE_OP_AXPY: return ssde_axpy(&op.u.axpy, stream);
    case SSDE_OP_PACK: return ssde_pack_weights(&op.u.pack, stream);
    case SSDE_OP_GN_FINALIZE: return ssde_gn_finalize(&op.u.gn_fin, stream);
    case SSDE_OP_PF_DRIFT: return ssde_pf_drift(&op.u.pf_drift, stream);
    case SSDE_OP_HUTCH_DIV: return ssde_hutch_div(&op.u.hutch_div, stream);
    case SSDE_OP_COLSUM_FINISH: return ssde_colsum_finish(&op.u.colsum_fin, stream);
    case SSDE_OP_GN_BWD_FINISH: return ssde_gn_bwd_finish(&op.u.gn_bwd_fin, stream);
  }
  ssde_set_error("program: unknown op kind %d", op.kind);
  return SSDE_EINVAL;
}

extern "C" int ssde_program_run(const ssde_op* ops, int32_t n_ops, void* stream) {
  SSDE_REQUIRE(ops && n_ops >= 0, "program: null ops");
  for (int i = 0; i < n_ops; ++i) {
    if (int rc = run_one(ops[i], stream)) {
      char tmp[400];
      strncpy(tmp, g_err, sizeof(tmp) - 1); tmp[sizeof(tmp) - 1] = 0;
      ssde_set_error("op %d (kind %d): %s", i, ops[i].kind, tmp);
      return rc;
    }
  }
  return SSDE_OK;
}

extern "C" int ssde_program_run_timed(const ssde_op* ops, int32_t n_ops, void* stream, float* ms) {
  SSDE_REQUIRE(ops && n_ops >= 0 && ms, "program: null args");
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::vector<hipEvent_t> ev(n_ops + 1);
  for (auto& e : ev) SSDE_HIP_CHECK(hipEventCreate(&e));
  int rc = SSDE_OK;
  SSDE_HIP_CHECK(hipEventRecord(ev[0], st));
  for (int i = 0; i < n_ops && rc == SSDE_OK; ++i) {
    rc = run_one(ops[i], stream);
    if (rc == SSDE_OK && hipEventRecord(ev[i + 1], st) != hipSuccess) { ssde_set_error("hipEventRecord failed"); rc = SSDE_EHIP; }
  }
  if (rc == SSDE_OK) {
    if (hipStreamSynchronize(st) != hipSuccess) { ssde_set_error("hipStreamSynchronize failed"); rc = SSDE_EHIP; }
    for (int i = 0; i < n_ops && rc == SSDE_OK; ++i)
      if (hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]) != hipSuccess) { ssde_set_error("hipEventElapsedTime failed"); rc = SSDE_EHIP; }
  }
  for (auto& e : ev) hipEventDestroy(e);
  return rc;
}

struct SsdeGraph { hipGraph_t graph; hipGraphExec_t exec; };

extern "C" int ssde_graph_capture(const ssde_op* ops, int32_t n_ops, void* stream, void** graph_out) {
  SSDE_REQUIRE(ops && graph_out, "graph: null args");
  hipStream_t st = static_cast<hipStream_t>(stream);
  SSDE_REQUIRE(st != nullptr, "graph: capture needs a non-default stream");
  SSDE_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  int rc = ssde_program_run(ops, n_ops, stream);
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(st, &graph);
  if (rc != SSDE_OK) { if (graph) hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) { ssde_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return SSDE_EHIP; }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) { hipGraphDestroy(graph); ssde_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e)); return SSDE_EHIP; }
  SsdeGraph* g = new SsdeGraph{graph, exec};
  *graph_out = g;
  return SSDE_OK;
}

extern "C" int ssde_graph_launch(void* graph, void* stream) {
  SSDE_REQUIRE(graph, "graph: null handle");
  SsdeGraph* g = static_cast<SsdeGraph*>(graph);
  SSDE_HIP_CHECK(hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream)));
  return SSDE_OK;
}

extern "C" int ssde_graph_destroy(void* graph) {
  if (!graph) return SSDE_OK;
  SsdeGraph* g = static_cast<SsdeGraph*>(graph);
  hipGraphExecDestroy(g->exec);
  hipGraphDestroy(g->graph);
  delete g;
  return SSDE_OK;
}
