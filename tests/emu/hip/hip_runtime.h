// TEST INFRASTRUCTURE ONLY -- a host-side functional emulator of the small slice of the HIP
// device/runtime API that score_sde_pytorch_amd/csrc/*.hip uses, so the kernels' index math,
// LDS protocols and MFMA fragment handling can be executed (slowly) on the build container's
// CPU by `pytest -m "not gpu"`.  It is found INSTEAD of ROCm's <hip/hip_runtime.h> only when a
// kernel source is compiled with `-I tests/emu` by tests/emu/build_emu.py; nothing in the
// product package references it and libssde_hip.so is never built against it.
//
// Execution model: one workgroup at a time; every thread of the workgroup is a fiber (own
// stack, hand-written x86-64 context switch).  __syncthreads() and the wave64 collectives
// (__shfl_xor, MFMA) park a fiber until its workgroup / wave has arrived; a collective that
// can never complete (divergent barrier, early-exited lane) aborts with a diagnostic.
// MFMA semantics follow /opt/skills/guides/cdna_hip_programming.md section 3: operand and
// C/D lane maps of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, k-ordered fmaf chain.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>

#ifndef __HIPCC__
#define __HIPCC__ 1
#endif
#define SSDE_EMULATED 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
// static LDS is banned in the product sources (all LDS is carved from the dynamic region,
// guide Guideline 17); make any re-introduction a compile error under emulation.
#define __shared__ static_assert(false, "use SSDE_LDS / HIP_DYNAMIC_SHARED: static __shared__ is not emulated");
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(emu::dyn_lds());

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));

namespace emu {
struct Fiber {
  dim3 tid;
  int linear, wave, lane;
  int state;        // 0 runnable, 1 at workgroup barrier, 2 at wave collective, 3 done
  void* sp;
  char* stack;
};
extern Fiber* cur;
extern dim3 g_blockIdx, g_blockDim, g_gridDim;
void* dyn_lds();
void syncthreads();
// wave64 exchange: every live lane of the wave deposits up to 4 words, all lanes may then read any lane's words
struct Xchg { uint32_t w[64][4]; };
const Xchg& wave_exchange(uint32_t a, uint32_t b = 0, uint32_t c = 0, uint32_t d = 0);
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body);
inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)
#define warpSize 64

static inline void __syncthreads() { emu::syncthreads(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---- wave64 cross-lane ----
static inline float __shfl_xor(float v, int mask, int width = 64) {
  (void)width;
  const emu::Xchg& x = emu::wave_exchange(emu::f2u(v));
  return emu::u2f(x.w[(emu::cur->lane ^ mask) & 63][0]);
}
static inline int __shfl_xor(int v, int mask, int width = 64) {
  (void)width;
  const emu::Xchg& x = emu::wave_exchange((uint32_t)v);
  return (int)x.w[(emu::cur->lane ^ mask) & 63][0];
}
static inline double __shfl_xor(double v, int mask, int width = 64) {      // two 32-bit exchanges, as the hardware does
  uint64_t u; memcpy(&u, &v, 8);
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)u, mask, width);
  const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(u >> 32), mask, width);
  u = ((uint64_t)hi << 32) | lo;
  double r; memcpy(&r, &u, 8);
  return r;
}
static inline float __shfl(float v, int src, int width = 64) {
  (void)width;
  const emu::Xchg& x = emu::wave_exchange(emu::f2u(v));
  return emu::u2f(x.w[src & 63][0]);
}
static inline float __shfl_down(float v, unsigned delta, int width = 64) {
  (void)width;
  const emu::Xchg& x = emu::wave_exchange(emu::f2u(v));
  const int s = emu::cur->lane + (int)delta;
  return emu::u2f(x.w[s < 64 ? s : emu::cur->lane][0]);
}

// ---- MFMA (exact f32; k-ordered fmaf chain, one rounding per product) ----
static inline emu_f32x16 emu_mfma_32x32x2(float a, float b, emu_f32x16 c) {
  const emu::Xchg& x = emu::wave_exchange(emu::f2u(a), emu::f2u(b));
  const int l = emu::cur->lane, j = l & 31, hi = l >> 5;
  emu_f32x16 d;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float v = c[r];
    for (int k = 0; k < 2; ++k) v = fmaf(emu::u2f(x.w[k * 32 + row][0]), emu::u2f(x.w[k * 32 + j][1]), v);
    d[r] = v;
  }
  return d;
}
static inline emu_f32x4 emu_mfma_16x16x4(float a, float b, emu_f32x4 c) {
  const emu::Xchg& x = emu::wave_exchange(emu::f2u(a), emu::f2u(b));
  const int l = emu::cur->lane, j = l & 15, q = l >> 4;
  emu_f32x4 d;
  for (int r = 0; r < 4; ++r) {
    const int row = q * 4 + r;
    float v = c[r];
    for (int k = 0; k < 4; ++k) v = fmaf(emu::u2f(x.w[k * 16 + row][0]), emu::u2f(x.w[k * 16 + j][1]), v);
    d[r] = v;
  }
  return d;
}
// v_mfma_f32_32x32x16_bf16: lane l holds A[i = l & 31][k = 8 (l >> 5) .. + 7] and B[k][j = l & 31] as 8 bf16 (element e in the
// low / high half of dword e / 2); C/D as the 32x32x2 form.  Products of two bf16 are exact in fp32; they are summed into the
// accumulator one by one in k order (the hardware's internal summation order is not specified: the GPU tests carry the
// tolerance, this model carries the index math).
template <class V>
static inline emu_f32x16 emu_mfma_32x32x16_bf16(V a, V b, emu_f32x16 c) {
  static_assert(sizeof(V) == 16, "8 bf16 per lane");
  uint32_t aw[4], bw[4];
  memcpy(aw, &a, 16); memcpy(bw, &b, 16);
  const emu::Xchg xa = emu::wave_exchange(aw[0], aw[1], aw[2], aw[3]);      // (copied: the buffer is reused two exchanges on)
  const emu::Xchg& xb = emu::wave_exchange(bw[0], bw[1], bw[2], bw[3]);
  const int l = emu::cur->lane, j = l & 31, hi = l >> 5;
  emu_f32x16 d;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float v = c[r];
    for (int kh = 0; kh < 2; ++kh)
      for (int e = 0; e < 8; ++e) {
        const uint32_t ua = xa.w[kh * 32 + row][e >> 1], ub = xb.w[kh * 32 + j][e >> 1];
        const float fa = emu::u2f((e & 1) ? (ua & 0xffff0000u) : (ua << 16)), fb = emu::u2f((e & 1) ? (ub & 0xffff0000u) : (ub << 16));
        v += fa * fb;
      }
    d[r] = v;
  }
  return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu_mfma_32x32x16_bf16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_32x32x2((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma_16x16x4((a), (b), (c))
// LDS-DMA: lane l's `size` bytes land at the wave-uniform LDS base + size * l (executed synchronously here)
static inline void emu_global_load_lds(const void* g, __attribute__((address_space(3))) void* l, unsigned size, int off) {
  memcpy((char*)(void*)l + off + (size_t)size * emu::cur->lane, (const char*)g + off, size);   // imm offset: both sides
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu_global_load_lds((const void*)(g), (l), (size), (off))
// the product issues the copy through inline assembly (ssde_common.h); the emulator substitutes its model of the instruction
#define SSDE_GLDS16_OFF(gptr, lds_wave_base, imm) emu_global_load_lds((const void*)(gptr), (__attribute__((address_space(3))) void*)(lds_wave_base), 16, (imm))
#define SSDE_OPAQUE_VGPR(x) ((void)0)
// scalar-base forms and the asm global load of ssde_common.h (the copies are synchronous here, the waits are no-ops)
#define SSDE_GLDS16_S(voff, sbase, lds_wave_base, imm) \
  emu_global_load_lds((const char*)(sbase) + (voff), (__attribute__((address_space(3))) void*)(lds_wave_base), 16, (imm))
#define SSDE_GLDS16_S_SAME_BASE(voff, sbase, lds_wave_base, imm) SSDE_GLDS16_S(voff, sbase, lds_wave_base, imm)
#define SSDE_GLDS16_S_SAME_BASE_LO32(voff, sbase, lds_wave_base, imm) \
  do { if (emu::cur->lane < 32) SSDE_GLDS16_S(voff, sbase, lds_wave_base, imm); } while (0)
#define SSDE_GLDS16_S_LO48(voff, sbase, lds_wave_base, imm) \
  do { if (emu::cur->lane < 48) SSDE_GLDS16_S(voff, sbase, lds_wave_base, imm); } while (0)
#define SSDE_GLDS16_S_SAME_BASE_LO48(voff, sbase, lds_wave_base, imm) SSDE_GLDS16_S_LO48(voff, sbase, lds_wave_base, imm)
#define SSDE_GLDS16_S_LO16(voff, sbase, lds_wave_base, imm) \
  do { if (emu::cur->lane < 16) SSDE_GLDS16_S(voff, sbase, lds_wave_base, imm); } while (0)
#define SSDE_GLOAD16(dst, voff, sbase) memcpy(&(dst), (const char*)(sbase) + (voff), 16)
#define SSDE_WAIT_VMCNT_FOR(n, a, b) ((void)0)
#define SSDE_GLOAD16_I(dst, voff, sbase, imm) memcpy(&(dst), (const char*)(sbase) + (voff) + (imm), 16)
#define SSDE_GLOAD8_I(dst, voff, sbase, imm) memcpy(&(dst), (const char*)(sbase) + (voff) + (imm), 8)
#define SSDE_GLOAD16_I_SAFE(dst, voff, sbase, imm) SSDE_GLOAD16_I(dst, voff, sbase, imm)
#define SSDE_GLOAD8_I_SAFE(dst, voff, sbase, imm) SSDE_GLOAD8_I(dst, voff, sbase, imm)
#define SSDE_WAIT_VMCNT_FOR3(n, a, b, c) ((void)0)
#define SSDE_GLOAD16_AGENT(dst, ptr) memcpy(&(dst), (const void*)(ptr), 16)
#define SSDE_GSTORE16_AGENT(ptr, val) memcpy((void*)(ptr), &(val), 16)
#define SSDE_WAIT_VMCNT_FOR4(n, a, b, c, d) ((void)0)
#define SSDE_WAIT_VMCNT_FENCE(n) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)      /* only used on wave-uniform values */
#define SSDE_GLDS16_OFF_SAME_BASE(gptr, lds_wave_base, imm) SSDE_GLDS16_OFF(gptr, lds_wave_base, imm)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
// timestamps: a counter that advances on every read (a loop that waits for time to pass terminates)
static inline unsigned long long __builtin_amdgcn_s_memtime() { static unsigned long long t = 0; return t += 1000; }
static inline unsigned long long __builtin_amdgcn_s_memrealtime() { static unsigned long long t = 0; return t += 50; }
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_sched_barrier(a) ((void)0)
#define __builtin_amdgcn_s_barrier() emu::syncthreads()

// ---- device math ----
// glibc declares __expf/__logf itself: map the HIP fast-math names with macros
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float __frcp_rn(float x) { return 1.0f / x; }
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
static inline float __fsqrt_rn(float x) { return sqrtf(x); }
static inline float __frsqrt_rn(float x) { return 1.0f / sqrtf(x); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
using std::max;
using std::min;
static inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }

#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
// atomics: workgroups run one after another and fibers are cooperative, so plain RMW is atomic
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline float unsafeAtomicAdd(float* p, float v) { return atomicAdd(p, v); }

// ---- host runtime subset ----
typedef void* hipStream_t;
typedef int hipError_t;
typedef struct emu_event* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum { hipSuccess = 0, hipErrorNotSupported = 801 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum { hipStreamCaptureModeThreadLocal = 1 };
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulator: unsupported"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
enum { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 256; return hipSuccess; }   /* (SSDE_NUM_CUS=3 in the tests of persistent kernels) */
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorNotSupported; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, int) { return hipErrorNotSupported; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
#define HIP_SYMBOL(X) (X)
template <class T> static inline hipError_t hipGetSymbolAddress(void** p, const T& sym) { *p = const_cast<void*>(static_cast<const void*>(&sym)); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
  emu::launch((grid), (block), (size_t)(lds), [&]() { kernel(__VA_ARGS__); })
