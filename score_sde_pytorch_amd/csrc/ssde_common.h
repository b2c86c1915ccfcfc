// Shared host/device helpers for libssde_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>
#include "../../include/ssde.h"

#define SSDE_OK 0
#define SSDE_EINVAL (-22)
#define SSDE_EHIP (-5)

void ssde_set_error(const char* fmt, ...);
// conv_wino.hip: launch (lds_out == NULL) or plan-only (returns the LDS bytes through lds_out)
int ssde_conv_wino_launch(const ssde_conv_args* a, void* stream, int* lds_out);
int ssde_conv_wino4_launch(const ssde_conv_args* a, void* stream, int* lds_out);   // conv_wino4.hip, same forms
bool ssde_wino4_xform_merges_gn(const ssde_conv_args* a);                               // wino4_xform.hip: the pass merges the source's GroupNorm partials itself (gn_in_part0)
int ssde_wino4_xform_vq_launch(const ssde_conv_args* a, void* stream);              // wino4_xform.hip: V = B^T pro(x) B into a->wino_v
int ssde_conv_wino4r_launch(const ssde_conv_args* a, void* stream, int* lds_out);  // conv_wino4r.hip (two-kernel form, operands from registers)
bool ssde_conv1x1_wants(const ssde_conv_args* a);                                    // conv1x1.hip
int ssde_conv1x1_launch(const ssde_conv_args* a, void* stream, int* lds_out);
unsigned* ssde_conv_sync_slots(int need);                                            // conv_mfma.hip
int ssde_conv_wino4_splits(int wgs, int ctot, int c_out, unsigned flags);                         // conv_wino4.hip
int ssde_conv_wino4r_splits(int wgs, int ctot, int c_out, unsigned flags);                        // conv_wino4r.hip
bool ssde_conv_small_wants(const ssde_conv_args* a);                                             // conv_small.hip (image heads: at most 4 couts)
int ssde_conv_small_launch(const ssde_conv_args* a, void* stream, int* lds_out);
bool ssde_wgrad_wino_wants(const ssde_wgrad_args* a);                                // wgrad_wino.hip
int64_t ssde_wgrad_wino_scratch_floats(const ssde_wgrad_args* a);
int ssde_wgrad_wino_launch(const ssde_wgrad_args* a, void* stream);
bool ssde_wgrad_wino4_wants(const ssde_wgrad_args* a);                               // wgrad_wino4.hip (asked first)
int64_t ssde_wgrad_wino4_scratch_floats(const ssde_wgrad_args* a);
int ssde_wgrad_wino4_launch(const ssde_wgrad_args* a, void* stream);

#define SSDE_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ssde_set_error(__VA_ARGS__);         \
      return SSDE_EINVAL;                  \
    }                                      \
  } while (0)

#define SSDE_HIP_CHECK(expr)                                                        \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      ssde_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return SSDE_EHIP;                                                             \
    }                                                                               \
  } while (0)

#define SSDE_LAUNCH_CHECK()  SSDE_HIP_CHECK(hipGetLastError())

int ssde_num_cus();                                                                   // runtime.hip: CUs of the current device (cached)
static inline int ssde_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline int ssde_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline bool ssde_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Every LDS byte is carved from the dynamic region: a static __shared__ precedes the dynamic region
// unpadded and can knock its base off 16-byte alignment (cdna_hip_programming.md Guideline 17).  The
// float4 element type keeps the base 16-byte aligned for ds_read_b128.
#define SSDE_LDS(var) HIP_DYNAMIC_SHARED(float4, var##_f4) float* var = reinterpret_cast<float*>(var##_f4)

// Async global -> LDS copy of 16 bytes per lane with no VGPR round trip (global_load_lds_dwordx4): lane l's
// bytes land at lds_wave_base + imm + 16*l (M0 holds the wave-uniform LDS base; the 13-bit signed immediate is added
// to BOTH the global and the LDS address, so a run of copies with equal strides on both sides needs one address setup).
// Issued through inline assembly on purpose: the compiler builtin (__builtin_amdgcn_global_load_lds) is modelled as
// an LDS write, and hipcc then puts s_waitcnt vmcnt(0) in front of the next ds_read of ANY LDS address -- the issuing
// wave sat out the whole copy latency before its first MFMA.  The asm form is opaque: ordering is the caller's job
// (the data is visible to other waves after the issuer's SSDE_WAIT_VMCNT + a workgroup barrier).
// M0 is written without a clobber (hipcc rejects "m0" as reserved): on gfx9+ nothing else in these kernels uses M0 (LDS
// instructions need no M0 initialisation; no s_movrel / s_sendmsg / LDS-direct), checked in the ISA of conv_wino.hip.
#ifndef SSDE_GLDS16_OFF
#define SSDE_GLDS16_OFF(gptr, lds_wave_base, imm)                                                                        \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%2"                               \
               :                                                                                                          \
               : "s"(__builtin_amdgcn_readfirstlane(                                                                      \
                     (int)(uintptr_t)(__attribute__((address_space(3))) void*)(lds_wave_base))),                          \
                 "v"(gptr), "n"(imm)                                                                                      \
               :)
#endif

// The same piece with M0 left as the previous SSDE_GLDS16_OFF of this wave set it (same lds_wave_base, another immediate
// offset): tests/test_isa_guards.py checks that nothing else in these kernels touches M0.
#ifndef SSDE_GLDS16_OFF_SAME_BASE
#define SSDE_GLDS16_OFF_SAME_BASE(gptr, lds_wave_base, imm) \
  asm volatile("global_load_lds_dwordx4 %0, off offset:%1" : : "v"(gptr), "n"(imm) :)
#endif

// The scalar-base forms: the lane's global address is sbase (SGPR pair, wave-uniform) + voff (32-bit VGPR byte offset) +
// imm -- a stream that advances by a wave-uniform stride needs no vector address arithmetic at all.  _LO32 issues the
// piece on lanes 0-31 only (a run of 512 bytes): exec is narrowed and restored inside the statement.
#ifndef SSDE_GLDS16_S
#define SSDE_GLDS16_S(voff, sbase, lds_wave_base, imm)                                                                   \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"                               \
               :                                                                                                          \
               : "s"(__builtin_amdgcn_readfirstlane(                                                                      \
                     (int)(uintptr_t)(__attribute__((address_space(3))) void*)(lds_wave_base))),                          \
                 "v"(voff), "s"(sbase), "n"(imm)                                                                          \
               :)
#define SSDE_GLDS16_S_SAME_BASE(voff, sbase, lds_wave_base, imm) \
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(sbase), "n"(imm) :)
#define SSDE_GLDS16_S_SAME_BASE_LO32(voff, sbase, lds_wave_base, imm)                                                    \
  do {                                                                                                                    \
    unsigned long long ssde_exec_save_;                                                                                   \
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_hi, 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3\n\t"          \
                 "s_mov_b64 exec, %0"                                                                                     \
                 : "=&s"(ssde_exec_save_)                                                                                 \
                 : "v"(voff), "s"(sbase), "n"(imm)                                                                        \
                 :);                                                                                                      \
  } while (0)
#endif

// A 16-byte global load whose completion hipcc does not track (asm): the kernel counts it with SSDE_WAIT_VMCNT_FOR itself,
// next to its LDS-DMA pieces.  (A load hipcc knows about gets `s_waitcnt vmcnt(n)` with n = the loads hipcc knows to be
// younger -- it cannot see the asm LDS-DMA pieces issued after it, so every use waited for ALL of them.)
// SSDE_WAIT_VMCNT_FOR(n, a, b) is the wait that "defines" the destinations a, b for the compiler: nothing may read them
// between the load and this statement (tests/test_isa_guards.py checks the emitted ISA for a stray copy).
typedef float ssde_f32x4 __attribute__((ext_vector_type(4)));
#ifndef SSDE_GLOAD16
#define SSDE_GLOAD16(dst, voff, sbase) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) :)
#define SSDE_WAIT_VMCNT_FOR(n, a, b) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(n) : "memory")
#define SSDE_WAIT_VMCNT_FENCE(n) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(n) : "memory")
#endif
// The same with a 13-bit signed immediate offset, and the 8-byte form (conv_wino4r.hip: operands straight into MFMA registers)
#ifndef SSDE_GLOAD16_I
#define SSDE_GLOAD16_I(dst, voff, sbase, imm) \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(imm) :)
#define SSDE_GLOAD8_I(dst, voff, sbase, imm) \
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(imm) :)
#define SSDE_WAIT_VMCNT_FOR3(n, a, b, c) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(n) : "memory")
#endif
// The same loads for places OUTSIDE a kernel's hand-scheduled loop, where hipcc may have parked the scalar base in a VGPR lane
// (an SGPR spill) and restores it with v_readlane_b32 right in front of the asm statement: a VALU write of an SGPR needs five
// wait states before a VMEM instruction reads that SGPR as its address (gfx9 data hazard), and the hazard recogniser does not
// look inside inline asm -- the load went out with a stale half of its 64-bit base (a memory access fault, round 6).  The
// s_nop pays those wait states unconditionally.
#ifndef SSDE_GLOAD16_I_SAFE
#define SSDE_GLOAD16_I_SAFE(dst, voff, sbase, imm) \
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(imm) :)
#define SSDE_GLOAD8_I_SAFE(dst, voff, sbase, imm) \
  asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(imm) :)
#endif

// 16-byte accesses at AGENT scope (sc1, what a relaxed agent-scope atomic dword access compiles to): the partial sums a split
// reduction hands from workgroup to workgroup cross XCDs, whose L2s are not coherent with each other; as dword atomics they
// were four times the transactions (conv_wino4r.hip).  The load is asynchronous to the compiler: SSDE_WAIT_VMCNT_FOR4 before use.
#ifndef SSDE_GLOAD16_AGENT
#define SSDE_GLOAD16_AGENT(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(ptr) : "memory")
#define SSDE_GSTORE16_AGENT(ptr, val) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(ptr), "v"(val) : "memory")
#define SSDE_WAIT_VMCNT_FOR4(n, a, b, c, d) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(n) : "memory")
#endif

// An LDS address the compiler must treat as one opaque 32-bit register (so that constant distances from it become the
// immediate offsets of ds_read / ds_write instead of one address register and one add per access).
typedef __attribute__((address_space(3))) const float ssde_lds_cfloat;
typedef float ssde_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const ssde_f32x2 ssde_lds_cfloat2;
typedef __attribute__((address_space(3))) float ssde_lds_float;
typedef __attribute__((address_space(3))) ssde_f32x2 ssde_lds_float2;
#ifndef SSDE_OPAQUE_VGPR
#define SSDE_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, which hipcc lowers
// to s_waitcnt vmcnt(0) whenever an LDS-DMA (or any global load) is in flight: a full memory latency exposed at every
// barrier.  This form waits for the wave's own LDS operations (lgkmcnt(0)), pins the compiler's ordering of memory
// accesses around it, and leaves VMEM in flight; a wave that must publish LDS-DMA'd data calls SSDE_WAIT_VMCNT(n)
// before the barrier its readers pass (cdna_hip_programming.md 5: "issuer's counted vmcnt + a barrier").
// s_waitcnt simm16 (gfx9): vmcnt = [15:14|3:0], expcnt = [6:4], lgkmcnt = [11:8]; all-ones = no wait.
#define SSDE_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt((((n) & 0xF) | (((n) >> 4) << 14)) | 0x0F70)
#define SSDE_LDS_BARRIER()                       \
  do {                                           \
    asm volatile("" ::: "memory");               \
    __builtin_amdgcn_s_waitcnt(0xC07F);          \
    __builtin_amdgcn_s_barrier();                \
    asm volatile("" ::: "memory");               \
  } while (0)

// LDS hand-over between the lanes of ONE wave (no workgroup barrier): the hardware executes a wave's LDS operations in order,
// so all that is needed is that the compiler keeps them in order and the data has landed; the test emulator runs lanes as
// fibers and needs a real rendez-vous.
#ifdef SSDE_EMULATED
#define SSDE_WAVE_SYNC() ((void)emu::wave_exchange(0u))
#else
#define SSDE_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

#ifdef __HIPCC__
// x * sigmoid(x) = x * rcp(1 + exp2(-x log2 e)): v_mul + v_exp_f32 + v_add + v_rcp_f32 + v_mul, each ~1 ulp.
// (__frcp_rn is the correctly rounded reciprocal: hipcc expands it to the 12-instruction IEEE division sequence, and
//  on gfx950 VALU cycles are not hidden behind fp32 MFMAs of another wave -- tools/microbench/mfma_valu_overlap.hip --
//  so in the staging prologues that sequence cost ~10% of a Winograd stage.)
__device__ __forceinline__ float ssde_silu(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
// d silu(u) / du = sig * (1 + u * (1 - sig))
__device__ __forceinline__ float ssde_silu_grad(float u) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
  return sg * (1.0f + u * (1.0f - sg));
}

// ---- prologue (GroupNorm apply, SiLU, dropout) shared by every kernel that stages a source ----
struct SsdePro {
  bool gn, silu, drop;
  uint32_t thresh, key;
  float dscale;
};
__device__ __forceinline__ SsdePro ssde_pro_decode(const ssde_src& s) {
  SsdePro p;
  p.gn = (s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU);
  p.silu = (s.pro_mode == SSDE_PRO_GN_SILU || s.pro_mode == SSDE_PRO_SILU);
  p.drop = s.drop_thresh != 0u;
  p.thresh = s.drop_thresh;
  p.dscale = s.drop_scale;
  p.key = p.drop ? (*s.drop_seed ^ s.drop_salt) : 0u;
  return p;
}
__device__ __forceinline__ uint32_t ssde_hash32(uint32_t x) {   // murmur3 finaliser
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
// dropout multiplier (0 or 1/(1-p)) of element `elem` of the virtual concat tensor
__device__ __forceinline__ float ssde_keep(uint32_t elem, const SsdePro& p) {
  return ssde_hash32(elem * 0x9E3779B1u + p.key) >= p.thresh ? p.dscale : 0.f;
}
// pro(x) on 4 consecutive channels; elem0 = linear index of v.x in the virtual concat tensor
__device__ __forceinline__ float4 ssde_pro_apply(float4 v, float mu, float rs, const float4& gam, const float4& bet,
                                                 uint32_t elem0, const SsdePro& p) {
  if (p.gn) {
    v.x = (v.x - mu) * rs * gam.x + bet.x;
    v.y = (v.y - mu) * rs * gam.y + bet.y;
    v.z = (v.z - mu) * rs * gam.z + bet.z;
    v.w = (v.w - mu) * rs * gam.w + bet.w;
  }
  if (p.silu) { v.x = ssde_silu(v.x); v.y = ssde_silu(v.y); v.z = ssde_silu(v.z); v.w = ssde_silu(v.w); }
  // unlikely: the four hashes stay behind a wave-uniform branch, out of line of the inference staging loops (a version
  // with two separate template instantiations behind the branch cost conv_wino.hip 2-5 %, profiles/r2_ab_prologue_split.txt)
  if (__builtin_expect(p.drop, 0)) {
    v.x *= ssde_keep(elem0, p); v.y *= ssde_keep(elem0 + 1u, p);
    v.z *= ssde_keep(elem0 + 2u, p); v.w *= ssde_keep(elem0 + 3u, p);
  }
  return v;
}

// ---- coalesced epilogue ---------------------------------------------------------------------------------------
// The MFMA accumulator layout gives a lane one output channel of scattered pixels (4-byte stores in 64..128-byte
// runs: store-ISSUE bound -- for K <= 256 the epilogue took as long as the main loop).  Instead the workgroup parks
// its output tile in LDS as [rows = local pixels][ld] and every thread moves float4s along channels: 16-byte
// stores / residual loads, a whole NHWC pixel row per 16 lanes.
struct SsdeEpi {
  const float* bias; const float* chan_add; int chan_add_ld;
  const float* resid; int resid_post; float scale; float* dst; int Cout;
  float* gn_part = nullptr;   // optional GroupNorm partial statistics of the stored tensor, see ssde_store_tile
};

// Chan's pairwise merge of (count, mean, M2 = sum of squared deviations); exact for n_b == 0
// (no contraction inside: the same merge is inlined into several kernels -- the finalize launch, the transform pass that merges for
//  itself, the producers' epilogues -- and "bit-identical whoever merges" holds only if hipcc does not fuse a multiply-add in one
//  of them and not in the other, which -ffp-contract=fast decides per call site; a clean build of round 6's sources showed it:
//  test_groupnorm_statistics_merged_by_the_transform_pass on the GPU)
__device__ __forceinline__ void ssde_stat_merge(float& n, float& m, float& M2, float nb, float mb, float M2b) {
#pragma clang fp contract(off)
  const float nt = n + nb;
  const float d = mb - m;
  const float f = nt > 0.f ? nb * __builtin_amdgcn_rcpf(nt) : 0.f;      // counts are small integers: ~1 ulp is plenty
  m += d * f;
  M2 += M2b + d * d * n * f;
  n = nt;
}
// The merge of the producers' partials into the statistics of one (image n, group g) by a TEAM OF 16 LANES (lanes 16 k .. 16 k + 15
// of a wave; l16 = lane & 15): lane l merges the entries l, l + 16, ... of the group's (slice, channel quad) list in order -- 16
// to 64 triples per group in the CIFAR networks --, then the 16 lane results are merged by xor shuffles, lower lane first: a fixed
// order, so the result is deterministic, and every lane of the team ends with the same (count, mean, M2).  Run by
// ssde_gn_finalize's kernel AND by the consumers that merge for themselves (wino4_xform.hip: ssde_conv_args.gn_in_part0): the
// same function, the same bits.  The groups of a channel concatenation straddle the boundary: [q0, qm) in part0, the rest in part1.
// (two phases, so that a consumer can put its own loads between them: the entries' loads are issued first and the chain of merges
//  runs while the consumer's loads are in flight -- loads return in order, so a merge that FOLLOWED 36 pixel loads per thread
//  waited for all of them first: wino4_xform.hip)
struct SsdeGnTeam {
  const float* part0; const float* part1;
  int Q0, Q1, s0, s1, q0, qm, cpq, e0, etot, n;
  float v[4][3];                                  // entries l16, l16 + 16, l16 + 32, l16 + 48 of the list (those that exist)
  __device__ __forceinline__ const float* entry(int k) const {
    if (k < e0) {
      const int nq = qm - q0, s = k / nq, q = q0 + (k - s * nq);
      return part0 + (((size_t)n * s0 + s) * Q0 + q) * 3;
    }
    const int kk = k - e0, nq = q0 + cpq - qm, s = kk / nq, q = qm - Q0 + (kk - s * nq);
    return part1 + (((size_t)n * s1 + s) * Q1 + q) * 3;
  }
};
// the entry list of one (image n, group g): everything but the loads
__device__ __forceinline__ void ssde_gn_team_init(SsdeGnTeam& t, const float* part0, const float* part1, int c0, int c1, int s0, int s1,
                                                  int groups, int n, int g) {
  t.part0 = part0; t.part1 = part1; t.s0 = s0; t.s1 = s1; t.n = n;
  t.cpq = (c0 + c1) / groups / 4;                 // channel quads per group
  t.q0 = g * t.cpq; t.Q0 = c0 >> 2; t.Q1 = c1 >> 2;
  t.qm = min(max(t.Q0, t.q0), t.q0 + t.cpq);
  t.e0 = (t.qm - t.q0) * s0;
  t.etot = t.e0 + (t.q0 + t.cpq - t.qm) * s1;
}
// More entries per group than this (the 128x128 and 256x256 levels of the FFHQ-256 network: 1024-4096 slices per image) and a
// team of 16 lanes is the wrong shape -- 64-256 dependent loads per lane --: ssde_gn_finalize gives such a group a whole
// workgroup (groupnorm.hip), and no consumer merges it for itself (wino4_xform.hip).  Host and device use the same bound.
#define SSDE_GN_TEAM_MAX_ENTRIES 256
__host__ __device__ __forceinline__ bool ssde_gn_group_is_big(int c0, int c1, int s0, int s1, int groups) {
  return (long long)((c0 + c1) / groups / 4) * ((long long)s0 + s1) > SSDE_GN_TEAM_MAX_ENTRIES;
}
__device__ __forceinline__ void ssde_gn_merge16_load(SsdeGnTeam& t, const float* part0, const float* part1, int c0, int c1, int s0, int s1,
                                                     int groups, int n, int g, int l16) {
  ssde_gn_team_init(t, part0, part1, c0, c1, s0, s1, groups, n, g);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = l16 + 16 * i;
    const float* e = t.entry(k < t.etot ? k : 0);
    t.v[i][0] = e[0]; t.v[i][1] = e[1]; t.v[i][2] = e[2];
  }
}
__device__ __forceinline__ void ssde_gn_merge16_finish(const SsdeGnTeam& t, int l16, float& cnt, float& m, float& M2) {
  cnt = 0.f; m = 0.f; M2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (l16 + 16 * i < t.etot) ssde_stat_merge(cnt, m, M2, t.v[i][2], t.v[i][0], t.v[i][1]);
  for (int k = l16 + 64; k < t.etot; k += 16) {   // (more than 64 entries per group: none of the shipped configurations)
    const float* e = t.entry(k);
    ssde_stat_merge(cnt, m, M2, e[2], e[0], e[1]);
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    const float nb = __shfl_xor(cnt, o, 64), mb = __shfl_xor(m, o, 64), Mb = __shfl_xor(M2, o, 64);
    if (l16 & o) { float tn = nb, tm = mb, tM = Mb; ssde_stat_merge(tn, tm, tM, cnt, m, M2); cnt = tn; m = tm; M2 = tM; }
    else ssde_stat_merge(cnt, m, M2, nb, mb, Mb);
  }
}
// (mean, M2, count) of a group -> rstd, by whoever merged (same reason: one function, no contraction)
__device__ __forceinline__ float ssde_gn_rstd(float cnt, float M2, float eps) {
#pragma clang fp contract(off)
  const float var = cnt > 0.f ? M2 / cnt : 0.f;
  return 1.0f / sqrtf(var + eps);
}
__device__ __forceinline__ void ssde_gn_merge16(const float* part0, const float* part1, int c0, int c1, int s0, int s1, int groups,
                                                int n, int g, int l16, float& cnt, float& m, float& M2) {
  SsdeGnTeam t;
  ssde_gn_merge16_load(t, part0, part1, c0, c1, s0, s1, groups, n, g, l16);
  ssde_gn_merge16_finish(t, l16, cnt, m, M2);
}

// pixfn(row, pix, img) -> false when the row is outside the tensor.  ncols must be a multiple of 4.
//
// GroupNorm statistics of the tensor being written (e.gn_part != nullptr, gn_entry >= 0): the next layer normalises this
// tensor per (image, group of channels) (layerspp.py:67,219,231); instead of re-reading it from HBM with
// gn_stats_kernel, the workgroup reduces the values it stores -- all rows of ONE image (the launcher guarantees it) --
// to one (mean, M2, count) triple per wave and channel quad: gn_part[((gn_entry * nwaves + wave) * Cout/4 + quad) * 3 ...],
// gn_entry = image * tiles_per_image + tile.  ssde_gn_finalize merges slices and quads into groups in a fixed order
// (deterministic, no atomics).  A thread owns one channel quad (nthreads % (ncols/4) == 0): sums relative to its first
// value (no cancellation), lanes of equal quad merged by wave shuffles.
// SWZ = 1: the tile was parked with column bit 4 flipped on rows with bit 4 set (conv_wino.hip: makes its 64 scalar
// LDS writes per lane bank-conflict free); undone here on the float4 row reads.
// Tiles that hold several whole images (maps of 8x8 and smaller): rpi_log2 = log2(rows per image); all threads are on
// the same image in one loop trip (rows per image >= NT / (NCOLS/4), checked by the launchers), the partial of an image
// is flushed when its last row has been stored, entries gn_entry, gn_entry + 1, ... (< gn_entry_max: the batch tail).
template <int ROWS, int NCOLS, int NT, int BATCH = 4, int SWZ = 0, class PixFn>
__device__ __forceinline__ void ssde_store_tile(float* tile, int ld, int n0, const SsdeEpi& e, PixFn pixfn, int gn_entry = -1,
                                                int rpi_log2 = 30, int gn_entry_max = 0x7fffffff, int tid_in = -1) {
  // tid_in: the caller's (opaque) copy of the thread id -- a persistent kernel passes one it re-derives per tile so that the row /
  // pixel arithmetic below is not hoisted out of its tile loop and kept in registers across the matrix loop (conv_wino4r.hip)
  const int tidx = tid_in >= 0 ? tid_in : (int)threadIdx.x;
  constexpr int C4N = NCOLS / 4, TOTAL = ROWS * C4N, ITERS = TOTAL / NT, RSTEP = NT / C4N;
  static_assert(TOTAL % NT == 0 && NT % C4N == 0 && 64 % C4N == 0, "a thread owns one channel quad of ITERS rows");
  const int c = (tidx % C4N) * 4, row0 = tidx / C4N;
  const int j = n0 + c;
  const bool stats = e.gn_part != nullptr && gn_entry >= 0;
  float st_p = 0.f, st_s1 = 0.f, st_s2 = 0.f, st_n = 0.f;
  const int rpi_mask = (rpi_log2 < 30 ? (1 << rpi_log2) : ROWS) - 1;
  // partial statistics of the image whose rows end with loop trip `it`: lanes of equal channel quad merged by shuffles,
  // one entry per WAVE (no workgroup barrier, no LDS round trip in the exposed tail of the kernel)
  auto flush = [&](int it) {
    float n = st_n, m = 0.f, M2 = 0.f;
    if (n > 0.f) { const float rn = __builtin_amdgcn_rcpf(n); m = st_p + st_s1 * rn; M2 = st_s2 - st_s1 * st_s1 * rn; M2 = M2 < 0.f ? 0.f : M2; }
    for (int o = C4N; o < 64; o <<= 1) {        // lanes l, l + C4N, l + 2 C4N, ... hold the same channel quad
      const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(m, o, 64), Mb = __shfl_xor(M2, o, 64);
      if (tidx & o) { float tn = nb, tm = mb, tM = Mb; ssde_stat_merge(tn, tm, tM, n, m, M2); n = tn; m = tm; M2 = tM; }
      else ssde_stat_merge(n, m, M2, nb, mb, Mb);   // both partners merge lower-lane-first: identical results
    }
    const int lane = tidx & 63, wave = tidx >> 6;
    const int entry = gn_entry + ((it * RSTEP) >> (rpi_log2 < 30 ? rpi_log2 : 30));
    if (lane < C4N && n0 + 4 * lane < e.Cout && entry < gn_entry_max) {
      float* o = e.gn_part + (((size_t)entry * (NT / 64) + wave) * (e.Cout >> 2) + (n0 >> 2) + lane) * 3;
      o[0] = m; o[1] = M2; o[2] = n;
    }
    st_p = st_s1 = st_s2 = st_n = 0.f;
  };
  if ((e.Cout & 3) == 0) {
    // The rows of a thread are processed in batches of up to 4: all residual / per-sample loads of a batch are issued
    // before the first value is needed.  (As one load per loop trip the 4..16 trips each waited out a full memory
    // latency: this loop was ~5000 of the ~10000 exposed epilogue cycles of a conv_wino workgroup, tools/wino_trace.py.)
    const bool col_ok = j < e.Cout;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e.bias && col_ok) bias4 = *reinterpret_cast<const float4*>(e.bias + j);
    constexpr int B = ITERS >= BATCH ? BATCH : ITERS;      // BATCH: 12 registers per row in flight
    static_assert(ITERS % B == 0, "row batches");
#pragma unroll
    for (int it0 = 0; it0 < ITERS; it0 += B) {
      float4 t[B], r[B], a[B];
      size_t pix[B];
      bool ok[B];
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int row = row0 + (it0 + b) * RSTEP;
        int img = 0;
        pix[b] = 0;
        ok[b] = col_ok && pixfn(row, pix[b], img);
        t[b] = *reinterpret_cast<const float4*>(tile + row * ld + (SWZ ? (c ^ (((row >> 4) & 1) << 4)) : c));
        r[b] = make_float4(0.f, 0.f, 0.f, 0.f);
        a[b] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok[b] && e.resid) r[b] = *reinterpret_cast<const float4*>(e.resid + pix[b] * e.Cout + j);
        if (ok[b] && e.chan_add) a[b] = *reinterpret_cast<const float4*>(e.chan_add + (size_t)img * e.chan_add_ld + j);
      }
      // Three straight passes over the batch -- values, stores, statistics -- instead of one loop trip per row that stored,
      // updated the statistics and possibly flushed them: with the flush's branches between two stores hipcc put an
      // `s_waitcnt vmcnt(0)` behind every global_store (gfx9 counts stores in vmcnt, and the count state it merges at those
      // joins is "unknown"), so the 4-8 stores of a thread each waited out a write acknowledgement: 7.0-7.7 k cycles of a
      // round's store phase in conv_wino4r_kernel (profiles/r5_wino4r_epilogue_trace.txt), the same pattern in every kernel
      // that ends in this function.  SSDE_STORE_ROW_BY_ROW=1 (an A/B variant only) keeps the old loop.
#if !defined(SSDE_STORE_ROW_BY_ROW) || !SSDE_STORE_ROW_BY_ROW
#pragma unroll
      for (int b = 0; b < B; ++b) {
        float v[4] = {t[b].x + bias4.x + a[b].x, t[b].y + bias4.y + a[b].y, t[b].z + bias4.z + a[b].z, t[b].w + bias4.w + a[b].w};
        const float4 rr = r[b];
        if (!e.resid_post) { v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= e.scale;
        if (e.resid_post) { v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
        t[b] = make_float4(v[0], v[1], v[2], v[3]);
      }
#pragma unroll
      for (int b = 0; b < B; ++b)
        if (ok[b]) *reinterpret_cast<float4*>(e.dst + pix[b] * e.Cout + j) = t[b];
      if (stats) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
          if (ok[b]) {
            if (st_n == 0.f) st_p = t[b].x;
            const float d0 = t[b].x - st_p, d1 = t[b].y - st_p, d2 = t[b].z - st_p, d3 = t[b].w - st_p;
            st_s1 += (d0 + d1) + (d2 + d3);
            st_s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            st_n += 4.f;
          }
          // uniform over the workgroup: the rows of an image are complete after this trip
          if ((((it0 + b + 1) * RSTEP) & rpi_mask) == 0) flush(it0 + b);
        }
      }
#else
#pragma unroll
      for (int b = 0; b < B; ++b) {
        if (ok[b]) {
          float v[4] = {t[b].x + bias4.x + a[b].x, t[b].y + bias4.y + a[b].y, t[b].z + bias4.z + a[b].z, t[b].w + bias4.w + a[b].w};
          const float4 rr = r[b];
          if (!e.resid_post) { v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] *= e.scale;
          if (e.resid_post) { v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
          *reinterpret_cast<float4*>(e.dst + pix[b] * e.Cout + j) = make_float4(v[0], v[1], v[2], v[3]);
          if (stats) {
            if (st_n == 0.f) st_p = v[0];
            const float d0 = v[0] - st_p, d1 = v[1] - st_p, d2 = v[2] - st_p, d3 = v[3] - st_p;
            st_s1 += (d0 + d1) + (d2 + d3);
            st_s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            st_n += 4.f;
          }
        }
        // uniform over the workgroup: the rows of an image are complete after this trip
        if (stats && (((it0 + b + 1) * RSTEP) & rpi_mask) == 0) flush(it0 + b);
      }
#endif
    }
  } else {
    // channel counts that are not a multiple of 4 (the 3-channel image ends): element-wise
    for (int it = 0; it < ITERS; ++it) {
      const int row = row0 + it * RSTEP;
      size_t pix; int img;
      if (j >= e.Cout || !pixfn(row, pix, img)) continue;
      const float4 t = *reinterpret_cast<const float4*>(tile + row * ld + (SWZ ? (c ^ (((row >> 4) & 1) << 4)) : c));
      const float v[4] = {t.x, t.y, t.z, t.w};
      for (int k = 0; k < 4 && j + k < e.Cout; ++k) {
        float x = v[k];
        if (e.bias) x += e.bias[j + k];
        if (e.chan_add) x += e.chan_add[(size_t)img * e.chan_add_ld + j + k];
        const float r = e.resid ? e.resid[pix * e.Cout + j + k] : 0.f;
        if (e.resid && !e.resid_post) x += r;
        x *= e.scale;
        if (e.resid && e.resid_post) x += r;
        e.dst[pix * e.Cout + j + k] = x;
      }
    }
  }
}


// ---- exact fp32 products on the BF16 matrix pipe (SSDE_MATRIX=bf16x6) --------------------------------------------------
// An fp32 value is the exact sum of three bf16 pieces taken by truncation (8 significand bits each: p0 = the top 16 bits of
// a, p1 = the top 16 bits of a - p0, p2 = a - p0 - p1, which has at most 8 significant bits left).  a * b = sum_ij a_i b_j
// with every partial product exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16 (8 x 8 significand bits); the six
// terms a0b0, a0b1, a1b0, a1b1, a0b2, a2b0 drop a1b2 + a2b1 + a2b2 <= 2^-23 |a b| per product -- the size of the fp32
// rounding of the product itself -- and cost 6/16 of the fp32 MFMA time (the BF16 pipe runs 16x the fp32 rate and, unlike
// fp32 MFMAs, co-issues with VALU).  tools/experiments/bf16_split_error_budget.py: the whole CIFAR NCSN++ through this
// split is as close to fp64 as the fp32 evaluation (profiles/r4_bf16_split_error_budget.txt).
typedef __bf16 ssde_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ssde_u32x4 __attribute__((ext_vector_type(4)));
// (hi's upper half, lo's upper half) as one dword of two bf16: v_perm_b32
__device__ __forceinline__ uint32_t ssde_pack_hi16(uint32_t lo, uint32_t hi) {
#ifdef SSDE_EMULATED
  return (hi & 0xffff0000u) | (lo >> 16);
#else
  return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
#endif
}
// 4 consecutive channels -> 3 pieces of 4 bf16 (8 bytes each), channel order kept
__device__ __forceinline__ void ssde_split3(const float4& v, uint2& p0, uint2& p1, uint2& p2) {
  const float f[4] = {v.x, v.y, v.z, v.w};
  uint32_t a[4], b[4], c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(uint32_t, f[i]);
    const float r = f[i] - __builtin_bit_cast(float, a[i] & 0xffff0000u);
    b[i] = __builtin_bit_cast(uint32_t, r);
    c[i] = __builtin_bit_cast(uint32_t, r - __builtin_bit_cast(float, b[i] & 0xffff0000u));
  }
  p0 = make_uint2(ssde_pack_hi16(a[0], a[1]), ssde_pack_hi16(a[2], a[3]));
  p1 = make_uint2(ssde_pack_hi16(b[0], b[1]), ssde_pack_hi16(b[2], b[3]));
  p2 = make_uint2(ssde_pack_hi16(c[0], c[1]), ssde_pack_hi16(c[2], c[3]));
}
// the six partial products of one 32x32 block over 16 channels, smallest terms first
#define SSDE_MFMA_BF16X6(acc, A, B)                                                                                       \
  do {                                                                                                                     \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssde_bf16x8, (A)[0]), __builtin_bit_cast(ssde_bf16x8, (B)[2]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssde_bf16x8, (A)[2]), __builtin_bit_cast(ssde_bf16x8, (B)[0]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssde_bf16x8, (A)[1]), __builtin_bit_cast(ssde_bf16x8, (B)[1]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssde_bf16x8, (A)[0]), __builtin_bit_cast(ssde_bf16x8, (B)[1]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssde_bf16x8, (A)[1]), __builtin_bit_cast(ssde_bf16x8, (B)[0]), acc, 0, 0, 0); \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssde_bf16x8, (A)[0]), __builtin_bit_cast(ssde_bf16x8, (B)[0]), acc, 0, 0, 0); \
  } while (0)

// SSDE_CONVF_BF16X6 in a launch's flags routes the contractions that have a split kernel through the BF16 matrix pipe;
// without it = the exact-fp32 MFMA kernels (the host side maps SSDE_MATRIX=bf16x6 to the flag when it builds the arguments)

__device__ __forceinline__ float ssde_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float ssde_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif
