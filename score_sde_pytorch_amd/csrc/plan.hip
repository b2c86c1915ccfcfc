// Plan-level C entry points: a whole lowered program (U-Net evaluation, or a predictor-corrector iteration) loaded from a
// PLAN BLOB and driven by a host that has no Python -- SURVEY 8(b): ssde_plan_*, ssde_unet_forward, ssde_pc_*.
//
// The reference's host is Python (NCSNpp.forward models/ncsnpp.py:232-381, pc_sampler sampling.py:390-409); its lowering
// to kernels stays in ONE place, score_sde_pytorch_amd/engine.py + pc_engine.py.  plan_export.py serialises what that
// lowering produced -- the flat ssde_op array, the liveness-planned activation arena, the packed kernel-layout weights,
// the reference-layout parameters with their state_dict names, the device-side re-pack descriptor tables, the step
// tables of the sampler -- as a position-independent blob: every pointer field is a (region, byte offset) relocation.
// This file allocates the regions on the device, uploads their initial contents, patches the pointers and runs the
// program; a C / C++ / Go / Rust host needs only libssde_hip.so, the blob and (optionally) a checkpoint to copy into
// the parameter regions followed by ssde_plan_refresh_weights.
#include "ssde_common.h"
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

namespace {

struct Region { void* dev = nullptr; int64_t bytes = 0; int kind = 0; char name[32]; };

}  // namespace

struct ssde_plan {
  ssde_plan_header hdr;
  std::vector<Region> regions;
  std::vector<ssde_op> ops, refresh_ops;
  std::vector<ssde_plan_param_entry> params;
  void* graph = nullptr;
  hipStream_t graph_stream = nullptr;
};

namespace {

int fail_free(ssde_plan* p, int rc) {
  if (p) {
    for (auto& r : p->regions)
      if (r.dev) hipFree(r.dev);
    delete p;
  }
  return rc;
}

void* region_ptr(const ssde_plan* p, int id) {
  return (id >= 0 && id < (int)p->regions.size()) ? p->regions[id].dev : nullptr;
}

}  // namespace

extern "C" int ssde_plan_load(const void* blob, size_t bytes, ssde_plan** out) {
  SSDE_REQUIRE(blob && out && bytes >= sizeof(ssde_plan_header), "plan: blob too small");
  const char* base = static_cast<const char*>(blob);
  ssde_plan_header h;
  memcpy(&h, base, sizeof(h));
  SSDE_REQUIRE(memcmp(h.magic, "SSDEPLN1", 8) == 0, "plan: bad magic");
  SSDE_REQUIRE(h.abi_version == SSDE_ABI_VERSION && h.sizeof_op == (int)sizeof(ssde_op),
               "plan: blob built for ABI %d / op size %d, library has %d / %d", h.abi_version, h.sizeof_op, SSDE_ABI_VERSION,
               (int)sizeof(ssde_op));
  const size_t need = sizeof(h) + (size_t)h.n_regions * sizeof(ssde_plan_region) + (size_t)(h.n_ops + h.n_refresh_ops) * sizeof(ssde_op) +
                      (size_t)h.n_relocs * sizeof(ssde_plan_reloc) + (size_t)h.n_params * sizeof(ssde_plan_param_entry) + (size_t)h.data_bytes;
  SSDE_REQUIRE(bytes >= need, "plan: blob truncated (%zu of %zu bytes)", bytes, need);
  const ssde_plan_region* regs = reinterpret_cast<const ssde_plan_region*>(base + sizeof(h));
  const char* ops_raw = reinterpret_cast<const char*>(regs + h.n_regions);
  const ssde_plan_reloc* rel = reinterpret_cast<const ssde_plan_reloc*>(ops_raw + (size_t)(h.n_ops + h.n_refresh_ops) * sizeof(ssde_op));
  const ssde_plan_param_entry* par = reinterpret_cast<const ssde_plan_param_entry*>(rel + h.n_relocs);
  const char* data = reinterpret_cast<const char*>(par + h.n_params);

  ssde_plan* p = new ssde_plan;
  p->hdr = h;
  p->regions.resize(h.n_regions);
  // staging copies of the constant regions (pointers inside them are patched before the upload)
  std::vector<std::vector<char>> staged(h.n_regions);
  for (int i = 0; i < h.n_regions; ++i) {
    Region& r = p->regions[i];
    r.bytes = regs[i].bytes; r.kind = regs[i].kind;
    memcpy(r.name, regs[i].name, sizeof(r.name));
    if (r.bytes <= 0) continue;
    if (hipMalloc(&r.dev, (size_t)r.bytes) != hipSuccess) {
      ssde_set_error("plan: hipMalloc of %lld bytes for region %d (%s) failed", (long long)r.bytes, i, r.name);
      return fail_free(p, SSDE_EHIP);
    }
    if (regs[i].kind == SSDE_REGION_CONST) {
      if (regs[i].data_offset < 0 || regs[i].data_offset + r.bytes > h.data_bytes) { ssde_set_error("plan: region %d data out of range", i); return fail_free(p, SSDE_EINVAL); }
      staged[i].assign(data + regs[i].data_offset, data + regs[i].data_offset + r.bytes);
    }
  }
  p->ops.resize(h.n_ops);
  p->refresh_ops.resize(h.n_refresh_ops);
  if (h.n_ops) memcpy(p->ops.data(), ops_raw, (size_t)h.n_ops * sizeof(ssde_op));
  if (h.n_refresh_ops) memcpy(p->refresh_ops.data(), ops_raw + (size_t)h.n_ops * sizeof(ssde_op), (size_t)h.n_refresh_ops * sizeof(ssde_op));
  for (int i = 0; i < h.n_relocs; ++i) {
    const ssde_plan_reloc& q = rel[i];
    if (q.region < 0 || q.region >= h.n_regions || q.offset < 0 || q.offset > p->regions[q.region].bytes) {
      ssde_set_error("plan: relocation %d points outside region %d", i, q.region);
      return fail_free(p, SSDE_EINVAL);
    }
    char* target = static_cast<char*>(p->regions[q.region].dev) + q.offset;
    char* where = nullptr;
    if (q.target_kind == SSDE_RELOC_OP && q.target >= 0 && q.target < h.n_ops) where = reinterpret_cast<char*>(&p->ops[q.target]);
    else if (q.target_kind == SSDE_RELOC_REFRESH_OP && q.target >= 0 && q.target < h.n_refresh_ops) where = reinterpret_cast<char*>(&p->refresh_ops[q.target]);
    else if (q.target_kind == SSDE_RELOC_REGION && q.target >= 0 && q.target < h.n_regions && !staged[q.target].empty()) where = staged[q.target].data();
    const int64_t limit = q.target_kind == SSDE_RELOC_REGION ? (where ? (int64_t)staged[q.target].size() : 0) : (int64_t)sizeof(ssde_op);
    if (!where || q.byte_offset < 0 || q.byte_offset + (int64_t)sizeof(void*) > limit) {
      ssde_set_error("plan: relocation %d has a bad target", i);
      return fail_free(p, SSDE_EINVAL);
    }
    memcpy(where + q.byte_offset, &target, sizeof(void*));
  }
  for (int i = 0; i < h.n_regions; ++i) {
    Region& r = p->regions[i];
    if (r.bytes <= 0) continue;
    hipError_t e = staged[i].empty() ? hipMemset(r.dev, 0, (size_t)r.bytes)
                                     : hipMemcpy(r.dev, staged[i].data(), (size_t)r.bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { ssde_set_error("plan: initialising region %d failed", i); return fail_free(p, SSDE_EHIP); }
  }
  p->params.assign(par, par + h.n_params);
  *out = p;
  return SSDE_OK;
}

extern "C" int ssde_plan_load_file(const char* path, ssde_plan** out) {
  SSDE_REQUIRE(path && out, "plan: null args");
  FILE* f = fopen(path, "rb");
  SSDE_REQUIRE(f, "plan: cannot open %s", path);
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<char> buf(n > 0 ? (size_t)n : 0);
  const size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
  fclose(f);
  SSDE_REQUIRE(n > 0 && got == (size_t)n, "plan: short read of %s", path);
  return ssde_plan_load(buf.data(), buf.size(), out);
}

extern "C" int ssde_plan_destroy(ssde_plan* p) {
  if (!p) return SSDE_OK;
  if (p->graph) ssde_graph_destroy(p->graph);
  fail_free(p, 0);
  return SSDE_OK;
}

extern "C" int ssde_plan_info(const ssde_plan* p, ssde_plan_header* out) {
  SSDE_REQUIRE(p && out, "plan: null args");
  *out = p->hdr;
  return SSDE_OK;
}

// Device address of a reference-layout parameter ("all_modules.3.weight", ...) for the host to copy checkpoint data
// into; index < 0 looks the name up, else `name_out` receives the index-th name.
extern "C" int ssde_plan_param(const ssde_plan* p, const char* name, int index, float** dev, int64_t* numel, const char** name_out) {
  SSDE_REQUIRE(p, "plan: null plan");
  int found = -1;
  if (index >= 0) found = index < (int)p->params.size() ? index : -1;
  else if (name)
    for (size_t i = 0; i < p->params.size(); ++i)
      if (strncmp(p->params[i].name, name, sizeof(p->params[i].name)) == 0) { found = (int)i; break; }
  SSDE_REQUIRE(found >= 0, "plan: no parameter %s", name ? name : "(index out of range)");
  const ssde_plan_param_entry& q = p->params[found];
  if (dev) *dev = reinterpret_cast<float*>(static_cast<char*>(region_ptr(p, q.region)) + q.offset);
  if (numel) *numel = q.numel;
  if (name_out) *name_out = q.name;
  return SSDE_OK;
}

extern "C" int ssde_plan_refresh_weights(ssde_plan* p, void* stream) {
  SSDE_REQUIRE(p, "plan: null plan");
  return ssde_program_run(p->refresh_ops.data(), (int)p->refresh_ops.size(), stream);
}

namespace {
int copy_in(const ssde_plan* p, int slot, const void* src, size_t bytes, hipStream_t st) {
  void* dst = region_ptr(p, p->hdr.io[slot]);
  SSDE_REQUIRE(dst && src, "plan: I/O slot %d missing", slot);
  SSDE_REQUIRE((int64_t)bytes <= p->regions[p->hdr.io[slot]].bytes, "plan: I/O slot %d too small", slot);
  SSDE_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
  return SSDE_OK;
}
}  // namespace

// out = model(x, cond): x, out [B, C, H, W] fp32, cond [B] (the noise level sigma or the time label, exactly what
// NCSNpp.forward receives), all DEVICE pointers.  `sigma` (discrete-label models with scale_by_sigma) and `std`
// (VP score head) may be NULL when the plan has no such input.
extern "C" int ssde_unet_forward(ssde_plan* p, const float* x, const float* cond, const float* sigma, const float* std_,
                                 float* out, void* stream) {
  SSDE_REQUIRE(p && p->hdr.kind == SSDE_PLAN_UNET, "unet_forward: not a U-Net plan");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ssde_plan_header& h = p->hdr;
  const size_t img = (size_t)h.batch * h.channels * h.height * h.width * sizeof(float);
  if (int rc = copy_in(p, SSDE_IO_X, x, img, st)) return rc;
  if (int rc = copy_in(p, SSDE_IO_COND, cond, (size_t)h.batch * sizeof(float), st)) return rc;
  if (h.io[SSDE_IO_SIGMA] >= 0 && h.io[SSDE_IO_SIGMA] != h.io[SSDE_IO_COND])
    if (int rc = copy_in(p, SSDE_IO_SIGMA, sigma, (size_t)h.batch * sizeof(float), st)) return rc;
  if (h.io[SSDE_IO_STD] >= 0)
    if (int rc = copy_in(p, SSDE_IO_STD, std_, (size_t)h.batch * sizeof(float), st)) return rc;
  if (int rc = ssde_program_run(p->ops.data(), (int)p->ops.size(), stream)) return rc;
  SSDE_REQUIRE(out, "unet_forward: null output");
  SSDE_HIP_CHECK(hipMemcpyAsync(out, region_ptr(p, h.io[SSDE_IO_OUT]), img, hipMemcpyDeviceToDevice, st));
  return SSDE_OK;
}

// ---- predictor-corrector sampler plans (pc_engine.FusedPCSampler: one program = one PC iteration) ----
extern "C" int ssde_pc_reset(ssde_plan* p, const float* x_T, uint64_t seed, void* stream) {
  SSDE_REQUIRE(p && p->hdr.kind == SSDE_PLAN_PC, "pc_reset: not a sampler plan");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ssde_plan_header& h = p->hdr;
  const size_t img = (size_t)h.batch * h.channels * h.height * h.width * sizeof(float);
  if (int rc = copy_in(p, SSDE_IO_X, x_T, img, st)) return rc;
  if (int rc = copy_in(p, SSDE_IO_XMEAN, x_T, img, st)) return rc;
  SSDE_HIP_CHECK(hipMemsetAsync(region_ptr(p, h.io[SSDE_IO_STEP]), 0, sizeof(int32_t), st));
  SSDE_HIP_CHECK(hipStreamSynchronize(st));
  SSDE_HIP_CHECK(hipMemcpy(region_ptr(p, h.io[SSDE_IO_SEED]), &seed, sizeof(seed), hipMemcpyHostToDevice));
  return SSDE_OK;
}

// n PC iterations on `stream`; use_graph != 0: the iteration is captured into a hipGraph on first use (stream must then
// be a non-default stream) and replayed -- the device step counter and seed word make every replay a new iteration.
extern "C" int ssde_pc_run(ssde_plan* p, int32_t n_iterations, int32_t use_graph, void* stream) {
  SSDE_REQUIRE(p && p->hdr.kind == SSDE_PLAN_PC && n_iterations >= 0, "pc_run: bad args");
  if (use_graph) {
    if (!p->graph || p->graph_stream != static_cast<hipStream_t>(stream)) {
      if (p->graph) { ssde_graph_destroy(p->graph); p->graph = nullptr; }
      if (int rc = ssde_graph_capture(p->ops.data(), (int)p->ops.size(), stream, &p->graph)) return rc;
      p->graph_stream = static_cast<hipStream_t>(stream);
    }
    for (int i = 0; i < n_iterations; ++i)
      if (int rc = ssde_graph_launch(p->graph, stream)) return rc;
    return SSDE_OK;
  }
  for (int i = 0; i < n_iterations; ++i)
    if (int rc = ssde_program_run(p->ops.data(), (int)p->ops.size(), stream)) return rc;
  return SSDE_OK;
}

extern "C" int ssde_pc_state(ssde_plan* p, float* x, float* x_mean, void* stream) {
  SSDE_REQUIRE(p && p->hdr.kind == SSDE_PLAN_PC, "pc_state: not a sampler plan");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ssde_plan_header& h = p->hdr;
  const size_t img = (size_t)h.batch * h.channels * h.height * h.width * sizeof(float);
  if (x) SSDE_HIP_CHECK(hipMemcpyAsync(x, region_ptr(p, h.io[SSDE_IO_X]), img, hipMemcpyDeviceToDevice, st));
  if (x_mean) SSDE_HIP_CHECK(hipMemcpyAsync(x_mean, region_ptr(p, h.io[SSDE_IO_XMEAN]), img, hipMemcpyDeviceToDevice, st));
  return SSDE_OK;
}

// ---- training plans (losses.FusedTrainStep: perturb | forward | loss head | backward | clip + Adam + EMA) ----
namespace {
int run_segment(ssde_plan* p, int lo, int hi, void* stream) {
  SSDE_REQUIRE(0 <= lo && lo <= hi && hi <= (int)p->ops.size(), "plan: bad segment [%d, %d)", lo, hi);
  return ssde_program_run(p->ops.data() + lo, hi - lo, stream);
}
// host scalars of a call (dropout seed word, hyper-parameters) are copied from the caller's memory: the stream is
// synchronised before returning to the caller's frame
int set_seed(ssde_plan* p, const int32_t* seed_word, hipStream_t st) {
  if (p->hdr.io[SSDE_IO_DROP_SEED] < 0) return SSDE_OK;              // a plan lowered without dropout
  SSDE_HIP_CHECK(hipMemcpyAsync(region_ptr(p, p->hdr.io[SSDE_IO_DROP_SEED]), seed_word, sizeof(int32_t), hipMemcpyHostToDevice, st));
  return SSDE_OK;
}
}  // namespace

extern "C" int ssde_train_step(ssde_plan* p, const float* batch, const float* z, const float* a, const float* s, const float* labels,
                               const float* g2, const float* hyper, uint32_t dropout_seed, float* loss_out, void* stream) {
  SSDE_REQUIRE(p && p->hdr.kind == SSDE_PLAN_TRAIN && p->hdr.seg[3] > 0, "train_step: not a training plan with an optimizer segment");
  SSDE_REQUIRE(hyper, "train_step: null hyper-parameters");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ssde_plan_header& h = p->hdr;
  const size_t img = (size_t)h.batch * h.channels * h.height * h.width * sizeof(float), vec = (size_t)h.batch * sizeof(float);
  if (int rc = copy_in(p, SSDE_IO_BATCH, batch, img, st)) return rc;
  if (int rc = copy_in(p, SSDE_IO_Z, z, img, st)) return rc;
  if (int rc = copy_in(p, SSDE_IO_A, a, vec, st)) return rc;
  if (int rc = copy_in(p, SSDE_IO_S, s, vec, st)) return rc;
  if (int rc = copy_in(p, SSDE_IO_COND, labels, vec, st)) return rc;
  if (h.io[SSDE_IO_STD] >= 0)                                         // VP score head: score = -h / std (models/utils.py:159)
    if (int rc = copy_in(p, SSDE_IO_STD, s, vec, st)) return rc;
  if (h.io[SSDE_IO_G2] >= 0 && g2)
    if (int rc = copy_in(p, SSDE_IO_G2, g2, vec, st)) return rc;
  SSDE_REQUIRE(region_ptr(p, h.io[SSDE_IO_HYPER]), "train_step: plan has no hyper-parameter record");
  SSDE_HIP_CHECK(hipMemcpyAsync(region_ptr(p, h.io[SSDE_IO_HYPER]), hyper, 9 * sizeof(float), hipMemcpyHostToDevice, st));
  const int32_t seed_word = (int32_t)(dropout_seed & 0x7FFFFFFFu);
  if (int rc = set_seed(p, &seed_word, st)) return rc;
  SSDE_HIP_CHECK(hipStreamSynchronize(st));                           // `hyper` and `seed_word` are the caller's / this frame's
  if (int rc = run_segment(p, 0, (int)p->ops.size(), stream)) return rc;
  if (int rc = ssde_plan_refresh_weights(p, stream)) return rc;       // packed copies follow the in-place parameter update
  if (loss_out) SSDE_HIP_CHECK(hipMemcpyAsync(loss_out, region_ptr(p, h.io[SSDE_IO_LOSS]), sizeof(float), hipMemcpyDeviceToDevice, st));
  return SSDE_OK;
}

extern "C" int ssde_train_forward(ssde_plan* p, const float* x, const float* cond, const float* sigma, const float* std_,
                                  uint32_t dropout_seed, float* out, void* stream) {
  SSDE_REQUIRE(p && p->hdr.kind == SSDE_PLAN_TRAIN, "train_forward: not a training plan");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ssde_plan_header& h = p->hdr;
  const size_t img = (size_t)h.batch * h.channels * h.height * h.width * sizeof(float), vec = (size_t)h.batch * sizeof(float);
  if (int rc = copy_in(p, SSDE_IO_X, x, img, st)) return rc;
  if (int rc = copy_in(p, SSDE_IO_COND, cond, vec, st)) return rc;
  if (h.io[SSDE_IO_SIGMA] >= 0 && h.io[SSDE_IO_SIGMA] != h.io[SSDE_IO_COND])
    if (int rc = copy_in(p, SSDE_IO_SIGMA, sigma, vec, st)) return rc;
  if (h.io[SSDE_IO_STD] >= 0)
    if (int rc = copy_in(p, SSDE_IO_STD, std_, vec, st)) return rc;
  const int32_t seed_word = (int32_t)(dropout_seed & 0x7FFFFFFFu);
  if (int rc = set_seed(p, &seed_word, st)) return rc;
  SSDE_HIP_CHECK(hipStreamSynchronize(st));
  if (int rc = run_segment(p, h.seg[0], h.seg[1], stream)) return rc;
  if (out) SSDE_HIP_CHECK(hipMemcpyAsync(out, region_ptr(p, h.io[SSDE_IO_OUT]), img, hipMemcpyDeviceToDevice, st));
  return SSDE_OK;
}

extern "C" int ssde_unet_backward(ssde_plan* p, const float* dout, float* dx, float* dparams, void* stream) {
  SSDE_REQUIRE(p && p->hdr.kind == SSDE_PLAN_TRAIN && dout, "unet_backward: not a training plan / null cotangent");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const ssde_plan_header& h = p->hdr;
  const size_t img = (size_t)h.batch * h.channels * h.height * h.width * sizeof(float);
  if (int rc = copy_in(p, SSDE_IO_GOUT, dout, img, st)) return rc;
  if (int rc = run_segment(p, h.seg[2], h.seg[3] > 0 ? h.seg[3] : (int)p->ops.size(), stream)) return rc;
  if (dx) {
    SSDE_REQUIRE(region_ptr(p, h.io[SSDE_IO_GX]), "unet_backward: the plan was exported without the input gradient");
    SSDE_HIP_CHECK(hipMemcpyAsync(dx, region_ptr(p, h.io[SSDE_IO_GX]), img, hipMemcpyDeviceToDevice, st));
  }
  if (dparams) {
    SSDE_REQUIRE(region_ptr(p, h.io[SSDE_IO_GRAD]) && h.n_flat > 0, "unet_backward: the plan carries no parameter gradients");
    SSDE_HIP_CHECK(hipMemcpyAsync(dparams, region_ptr(p, h.io[SSDE_IO_GRAD]), (size_t)h.n_flat * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  return SSDE_OK;
}

extern "C" int ssde_plan_copy_io(ssde_plan* p, int32_t slot, void* buf, int64_t bytes, int32_t to_plan, void* stream) {
  SSDE_REQUIRE(p && buf && slot >= 0 && slot < SSDE_IO_SLOTS && bytes >= 0, "plan_copy_io: bad args");
  void* reg = region_ptr(p, p->hdr.io[slot]);
  SSDE_REQUIRE(reg, "plan_copy_io: the plan has no I/O slot %d", slot);
  SSDE_REQUIRE(bytes <= p->regions[p->hdr.io[slot]].bytes, "plan_copy_io: %lld bytes exceed slot %d (%lld)", (long long)bytes, slot,
               (long long)p->regions[p->hdr.io[slot]].bytes);
  SSDE_HIP_CHECK(hipMemcpyAsync(to_plan ? reg : buf, to_plan ? buf : reg, (size_t)bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
  return SSDE_OK;
}
