// Single-head spatial self-attention core of AttnBlockpp (models/layerspp.py:82-86):
//   w = softmax_j( sum_c q[i,c] k[j,c] * C^-1/2 ),  h[i,c] = sum_j w[i,j] v[j,c]
// with L = H*W <= 256 tokens (attention only ever runs at 16x16 and at the 4x4
// bottleneck) and d = C channels.  fp32 operands on the exact-fp32
// MFMA (v_mfma_f32_32x32x2_f32).  One workgroup = 64 query rows of one image:
//   1. S = Q K^T  (4 waves x 64 keys each), Q/K channel chunks staged in LDS
//   2. row softmax on the 64 x L score tile held in LDS (never written to HBM,
//      the reference materialises [B, L, L])
//   3. O = P V   (4 waves x 64 channels each), V chunks staged in LDS as stored
//      (token-major); the k-strided B fragment is read with 4 ds_read_b32.
// qkv layout [N, L, 3C]: the fused NIN_0..2 projection output (q | k | v).
#include "ssde_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kQB = 64;          // query rows per workgroup
constexpr int kLMax = 256;
constexpr int kLDC = 36;         // channel-chunk row stride (32 + 4)
constexpr int kLDP = 260;        // score / V row stride (256 + 4)
constexpr int kAttnLdsFloats = kQB * kLDP + 32 * kLDP;

__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ qkv, float* __restrict__ dst,
                                                   int N, int L, int C, float scale) {
  SSDE_LDS(smem);
  float* Qs = smem;                       // [64][36]
  float* Ks = smem + kQB * kLDC;          // [256][36]
  float* Ps = smem;                       // [64][260]   (after step 1)
  float* Vs = smem + kQB * kLDP;          // [32][260]
  const int n = blockIdx.y, q0 = blockIdx.x * kQB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int C3 = 3 * C;
  const float* base = qkv + (size_t)n * L * C3;
  const int Lk = (L + 63) & ~63;          // keys rounded to a wave block
  const int kb = wave * 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- 1. S = Q K^T ----
  for (int c0 = 0; c0 < C; c0 += 32) {
    __syncthreads();
    for (int q = tid; q < kQB * 8; q += 256) {
      const int row = q >> 3, f = q & 7;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q0 + row < L) v = *reinterpret_cast<const float4*>(base + (size_t)(q0 + row) * C3 + c0 + f * 4);
      *reinterpret_cast<float4*>(Qs + row * kLDC + f * 4) = v;
    }
    for (int q = tid; q < Lk * 8; q += 256) {
      const int row = q >> 3, f = q & 7;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < L) v = *reinterpret_cast<const float4*>(base + (size_t)row * C3 + C + c0 + f * 4);
      *reinterpret_cast<float4*>(Ks + row * kLDC + f * 4) = v;
    }
    __syncthreads();
    if (kb < Lk) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float4 af[2], bf[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const float4*>(Qs + (a * 32 + li) * kLDC + kk * 8 + lh * 4);
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const float4*>(Ks + (kb + b * 32 + li) * kLDC + kk * 8 + lh * 4);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
          }
      }
    }
  }
  __syncthreads();
  if (kb < Lk) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          Ps[row * kLDP + kb + b * 32 + li] = acc[a][b][r] * scale;
        }
  }
  __syncthreads();

  // ---- 2. softmax over keys, 4 lanes per query row ----
  {
    const int row = tid >> 2, sub = tid & 3;
    float* prow = Ps + row * kLDP;
    float m = -INFINITY;
    for (int j = sub; j < L; j += 4) m = fmaxf(m, prow[j]);
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    float sum = 0.f;
    for (int j = sub; j < L; j += 4) { const float e = __expf(prow[j] - m); prow[j] = e; sum += e; }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    const float inv = 1.0f / sum;
    for (int j = sub; j < L; j += 4) prow[j] *= inv;
    const int Lp = (L + 31) & ~31;
    for (int j = L + sub; j < Lp; j += 4) prow[j] = 0.f;
  }

  // ---- 3. O = P V ----
  const int Lp = (L + 31) & ~31;
  for (int cp = 0; cp < C; cp += 256) {
    const int Cw = min(256, C - cp);
    const int cb = wave * 64;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int k0 = 0; k0 < Lp; k0 += 32) {
      __syncthreads();
      const int f4n = Cw >> 2;
      for (int q = tid; q < 32 * f4n; q += 256) {
        const int row = q / f4n, f = q - row * f4n;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 + row < L) v = *reinterpret_cast<const float4*>(base + (size_t)(k0 + row) * C3 + 2 * C + cp + f * 4);
        *reinterpret_cast<float4*>(Vs + row * kLDP + f * 4) = v;
      }
      __syncthreads();
      if (cb < Cw) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          float4 af[2];
          float bf[2][4];
#pragma unroll
          for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const float4*>(Ps + (a * 32 + li) * kLDP + k0 + kk * 8 + lh * 4);
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[b][j] = Vs[(kk * 8 + lh * 4 + j) * kLDP + cb + b * 32 + li];
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b][0], acc[a][b], 0, 0, 0);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b][1], acc[a][b], 0, 0, 0);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b][2], acc[a][b], 0, 0, 0);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b][3], acc[a][b], 0, 0, 0);
            }
        }
      }
    }
    if (cb < Cw) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int col = cp + cb + b * 32 + li;
          if (col >= C) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = q0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < L) dst[((size_t)n * L + row) * C + col] = acc[a][b][r];
          }
        }
    }
  }
}

}  // namespace

extern "C" int ssde_attention(const ssde_attn_args* a, void* stream) {
  SSDE_REQUIRE(a && a->qkv && a->dst, "attention: null args");
  SSDE_REQUIRE(a->n > 0 && a->l > 0 && a->l <= kLMax, "attention: token count %d outside 1..%d", a->l, kLMax);
  SSDE_REQUIRE(a->c > 0 && a->c % 32 == 0, "attention: channels must be a multiple of 32 (got %d)", a->c);
  const int lds = kAttnLdsFloats * 4;
  auto kfn = attn_kernel;
  static bool attr_set = false;   // set once, outside any stream capture
  if (!attr_set) {
    SSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, dim3(ssde_cdiv(a->l, kQB), a->n), dim3(256), lds, static_cast<hipStream_t>(stream),
                     a->qkv, a->dst, a->n, a->l, a->c, a->scale);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
