// Shared host/device helpers for libssde_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ssde.h"

#define SSDE_OK 0
#define SSDE_EINVAL (-22)
#define SSDE_EHIP (-5)

void ssde_set_error(const char* fmt, ...);

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

static inline int ssde_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline int ssde_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline bool ssde_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Every LDS byte is carved from the dynamic region: a static __shared__ precedes the dynamic region
// unpadded and can knock its base off 16-byte alignment (cdna_hip_programming.md Guideline 17).  The
// float4 element type keeps the base 16-byte aligned for ds_read_b128.
#define SSDE_LDS(var) HIP_DYNAMIC_SHARED(float4, var##_f4) float* var = reinterpret_cast<float*>(var##_f4)

#ifdef __HIPCC__
// x * sigmoid(x); v_exp_f32 + v_rcp_f32 (each ~1 ulp)
__device__ __forceinline__ float ssde_silu(float x) {
  return x * __frcp_rn(1.0f + __expf(-x));
}
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
