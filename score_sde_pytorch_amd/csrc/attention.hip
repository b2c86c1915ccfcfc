// Single-head spatial self-attention core of AttnBlockpp (models/layerspp.py:82-86):
//   w = softmax_j( sum_c q[i,c] k[j,c] * C^-1/2 ),  h[i,c] = sum_j w[i,j] v[j,c]
// with L = H*W <= 256 tokens (attention only ever runs at 16x16 and at the 4x4
// bottleneck) and d = C channels, and its backward (what autograd derives
// for those three lines).  fp32 operands on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Forward, one workgroup = 64 query rows of one image:
//   1. S = Q K^T  (4 waves x 64 keys each), Q/K channel chunks staged in LDS
//   2. row softmax on the 64 x L score tile held in LDS (never written to HBM,
//      the reference materialises [B, L, L])
//   3. O = P V   (4 waves x 64 channels each), V chunks staged in LDS as stored
//      (token-major); the k-strided B fragment is read with 4 ds_read_b32.
// Backward re-uses the same two GEMM phases (P is recomputed, never stored):
//   kernel A (per 64 query rows):  P, D_i = dO_i . O_i;  dP = dO V^T;  dS = P o (dP - D);  dQ = scale dS K
//   kernel B (per 64 key rows):    P^T from the saved row max / row sum;  dV = P^T dO;
//                                  dP^T = V dO^T;  dS^T = P^T o (dP^T - D);  dK = scale dS^T Q
// qkv layout [N, L, 3C]: the fused NIN_0..2 projection output (q | k | v).
#include "ssde_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kQB = 64;          // rows per workgroup
constexpr int kLMax = 256;
constexpr int kLDC = 36;         // channel-chunk row stride (32 + 4)
constexpr int kLDP = 260;        // score / V row stride (256 + 4)
constexpr int kTileFloats = kQB * kLDP;                 // the 64 x L tile
constexpr int kStageFloats = (kQB + kLMax) * kLDC;      // Q/K chunk staging (phase 1); >= 32 * kLDP (phase 3)
constexpr int kFwdVch = 8;                              // V tokens per chunk of the forward kernel
constexpr int kAttnLdsFloats = kQB * kLDP + kFwdVch * kLDP;  // forward: Q/K staging aliases the tile in phase 1; 75 KB: two workgroups per CU
constexpr int kBwdLdsFloats = kTileFloats + kStageFloats + 3 * kLMax;

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}

// acc (wave: rows 0..63 x columns kb..kb+63, kb = wave*64) = A[64 x C] * B[Lk x C]^T.
// A: 64 rows from pointer A (row stride lda), rows >= a_valid read as zero; B likewise with b_valid rows.
__device__ __forceinline__ void gemm_nt(const float* __restrict__ A, int a_valid, size_t lda,
                                        const float* __restrict__ B, int b_valid, size_t ldb, int C, int Lk,
                                        float* Qs, float* Ks, f32x16 (&acc)[2][2]) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int kb = wave * 64;
  zero_acc(acc);
  // The 32-channel chunk c0+32 is fetched into registers while the MFMAs of chunk c0 run (as fixed-count, branch-free
  // loads: rows past the valid range read a clamped address and are zeroed); it is parked in LDS behind the barrier.
  const int f = tid & 7, r0 = tid >> 3;                   // item (row r0 + 32 it, channel quad f)
  float4 ra[2], rb[8];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = r0 + it * 32;
      const float4 v = *reinterpret_cast<const float4*>(A + (size_t)max(min(row, a_valid - 1), 0) * lda + c0 + f * 4);
      ra[it] = row < a_valid ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = r0 + it * 32;
      const float4 v = *reinterpret_cast<const float4*>(B + (size_t)max(min(row, b_valid - 1), 0) * ldb + c0 + f * 4);
      rb[it] = row < b_valid ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  load_chunk(0);
  for (int c0 = 0; c0 < C; c0 += 32) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) *reinterpret_cast<float4*>(Qs + (r0 + it * 32) * kLDC + f * 4) = ra[it];
#pragma unroll
    for (int it = 0; it < 8; ++it)
      if (r0 + it * 32 < Lk) *reinterpret_cast<float4*>(Ks + (r0 + it * 32) * kLDC + f * 4) = rb[it];
    __syncthreads();
    if (c0 + 32 < C) load_chunk(c0 + 32);
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
}

// out[64 x C] = out_scale * Ps[64 x Lp] * Bm[L x C]   (Bm token-major rows, row stride ldb, rows >= b_valid zero)
// VCH = tokens of V staged per chunk (a multiple of 8): 32 in the backward kernels; 8 in the forward kernel, whose LDS
// footprint (64 x 260 score tile + the chunk) then lets two workgroups share a CU.
template <int VCH = 32>
__device__ __forceinline__ void gemm_pv(const float* Ps, int Lp, const float* __restrict__ Bm, int b_valid, size_t ldb, int C,
                                        float* Vs, float* __restrict__ out, size_t ldo, int out_valid, float out_scale) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
  for (int cp = 0; cp < C; cp += 256) {
    const int Cw = min(256, C - cp);
    const int cb = wave * 64;
    zero_acc(acc);
    // VCH-token chunk of V: up to VCH / 4 float4 per thread, prefetched into registers during the MFMAs of the previous chunk
    constexpr int NIT = VCH / 4;
    const int f4n = Cw >> 2;
    int vrow[NIT], vf[NIT];
    {
      int row = tid / f4n, ff = tid - row * f4n;
      const int dr = 256 / f4n, df = 256 - dr * f4n;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        vrow[it] = row; vf[it] = ff;
        row += dr; ff += df;
        if (ff >= f4n) { ff -= f4n; ++row; }
      }
    }
    float4 rv[NIT];
    auto load_chunk = [&](int k0) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int tok = k0 + vrow[it];
        const float4 v = *reinterpret_cast<const float4*>(Bm + (size_t)max(min(tok, b_valid - 1), 0) * ldb + cp + vf[it] * 4);
        rv[it] = (vrow[it] < VCH && tok < b_valid) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    load_chunk(0);
    for (int k0 = 0; k0 < Lp; k0 += VCH) {
      __syncthreads();
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (vrow[it] < VCH) *reinterpret_cast<float4*>(Vs + vrow[it] * kLDP + vf[it] * 4) = rv[it];
      __syncthreads();
      if (k0 + VCH < Lp) load_chunk(k0 + VCH);
      if (cb < Cw) {
#pragma unroll
        for (int kk = 0; kk < VCH / 8; ++kk) {
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
            const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < out_valid) out[(size_t)row * ldo + col] = acc[a][b][r] * out_scale;
          }
        }
    }
  }
}

// accumulators (wave's 64 x 64 block of the 64 x Lk tile) -> LDS tile, times `mul`
__device__ __forceinline__ void acc_to_tile(const f32x16 (&acc)[2][2], float* Ps, int Lk, float mul) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5, kb = wave * 64;
  if (kb < Lk) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          Ps[row * kLDP + kb + b * 32 + li] = acc[a][b][r] * mul;
        }
  }
}

// softmax over the first L columns of each of the 64 rows (4 lanes per row); columns L..Lp-1 zeroed.
// Optionally returns the row max and the row sum of exp (lane sub == 0 of each row holds them).
__device__ __forceinline__ void softmax_rows(float* Ps, int L, float* m_out, float* l_out) {
  const int tid = threadIdx.x;
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
  *m_out = m; *l_out = sum;
}

__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ qkv, float* __restrict__ dst,
                                                   int N, int L, int C, float scale) {
  SSDE_LDS(smem);
  float* Qs = smem;                       // [64][36]
  float* Ks = smem + kQB * kLDC;          // [256][36]
  float* Ps = smem;                       // [64][260]   (after step 1)
  float* Vs = smem + kQB * kLDP;          // [kFwdVch][260]
  const int n = blockIdx.y, q0 = blockIdx.x * kQB;
  const int C3 = 3 * C;
  const float* base = qkv + (size_t)n * L * C3;
  const int Lk = (L + 63) & ~63;          // keys rounded to a wave block
  const int Lp = (L + 31) & ~31;

  f32x16 acc[2][2];
  gemm_nt(base + (size_t)q0 * C3, L - q0, C3, base + C, L, C3, C, Lk, Qs, Ks, acc);
  __syncthreads();
  acc_to_tile(acc, Ps, Lk, scale);
  __syncthreads();
  float m, l;
  softmax_rows(Ps, L, &m, &l);
  gemm_pv<kFwdVch>(Ps, Lp, base + 2 * C, L, C3, C, Vs, dst + ((size_t)n * L + q0) * C, C, L - q0, 1.0f);
}

// ---- backward A: dQ (and the per-row softmax statistics + D for kernel B) ----
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                         const float* __restrict__ d_o, float* __restrict__ dqkv,
                                                         float* __restrict__ stats, int N, int L, int C, float scale) {
  SSDE_LDS(smem);
  float* Ps = smem;                       // [64][260]
  float* Qs = smem + kTileFloats;         // staging
  float* Ks = Qs + kQB * kLDC;
  float* Vs = Qs;
  float* Ds = smem + kTileFloats + kStageFloats;   // [64]
  const int n = blockIdx.y, q0 = blockIdx.x * kQB, tid = threadIdx.x;
  const int C3 = 3 * C;
  const float* base = qkv + (size_t)n * L * C3;
  const int Lk = (L + 63) & ~63, Lp = (L + 31) & ~31;
  const int valid = L - q0;

  f32x16 acc[2][2];
  gemm_nt(base + (size_t)q0 * C3, valid, C3, base + C, L, C3, C, Lk, Qs, Ks, acc);
  __syncthreads();
  acc_to_tile(acc, Ps, Lk, scale);
  __syncthreads();
  float m, l;
  softmax_rows(Ps, L, &m, &l);
  {   // D_i = sum_c dO[i,c] * O[i,c]; 4 lanes per row
    const int row = tid >> 2, sub = tid & 3;
    float d = 0.f;
    if (row < valid) {
      const float* po = o + ((size_t)n * L + q0 + row) * C;
      const float* pd = d_o + ((size_t)n * L + q0 + row) * C;
      for (int c = sub * 4; c < C; c += 16) {
        const float4 a = *reinterpret_cast<const float4*>(po + c);
        const float4 b = *reinterpret_cast<const float4*>(pd + c);
        d += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
      }
    }
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    if (sub == 0) {
      Ds[row] = d;
      if (row < valid) {
        float* st = stats + ((size_t)n * L + q0 + row) * 4;
        st[0] = m; st[1] = l; st[2] = d; st[3] = 0.f;
      }
    }
  }
  // dP = dO V^T
  gemm_nt(d_o + ((size_t)n * L + q0) * C, valid, C, base + 2 * C, L, C3, C, Lk, Qs, Ks, acc);
  __syncthreads();
  {   // dS = P o (dP - D), in place over P
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5, kb = wave * 64;
    if (kb < Lk) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int col = kb + b * 32 + li;
            if (col < Lp) Ps[row * kLDP + col] *= (acc[a][b][r] - Ds[row]);
          }
    }
  }
  // dQ = scale * dS K
  gemm_pv(Ps, Lp, base + C, L, C3, C, Vs, dqkv + ((size_t)n * L + q0) * C3, C3, valid, scale);
}

// ---- backward B: dK, dV for 64 key rows ----
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                          float* __restrict__ dqkv, const float* __restrict__ stats,
                                                          int N, int L, int C, float scale) {
  SSDE_LDS(smem);
  float* Ps = smem;                       // [64 keys][260 queries]
  float* Qs = smem + kTileFloats;
  float* Ks = Qs + kQB * kLDC;
  float* Vs = Qs;
  float* St = smem + kTileFloats + kStageFloats;   // [3][256]: m, l, D per query
  const int n = blockIdx.y, k0 = blockIdx.x * kQB, tid = threadIdx.x;
  const int C3 = 3 * C;
  const float* base = qkv + (size_t)n * L * C3;
  const int Lk = (L + 63) & ~63, Lp = (L + 31) & ~31;
  const int valid = L - k0;
  for (int j = tid; j < kLMax; j += 256) {
    float m = 0.f, l = 1.f, d = 0.f;
    if (j < L) { const float* st = stats + ((size_t)n * L + j) * 4; m = st[0]; l = st[1]; d = st[2]; }
    St[j] = m; St[kLMax + j] = l; St[2 * kLMax + j] = d;
  }
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5, kb = wave * 64;

  f32x16 acc[2][2];
  // S^T = K_tile Q^T
  gemm_nt(base + (size_t)k0 * C3 + C, valid, C3, base, L, C3, C, Lk, Qs, Ks, acc);
  __syncthreads();
  if (kb < Lk) {   // P^T[key, query] = exp(scale * s - m_q) / l_q ; rows (keys) >= valid and columns (queries) >= L are zero
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int col = kb + b * 32 + li;
          float pv = 0.f;
          if (row < valid && col < L) pv = __expf(acc[a][b][r] * scale - St[col]) / St[kLMax + col];
          Ps[row * kLDP + col] = pv;
        }
  }
  __syncthreads();
  // dV = P^T dO
  gemm_pv(Ps, Lp, d_o + (size_t)n * L * C, L, C, C, Vs, dqkv + ((size_t)n * L + k0) * C3 + 2 * C, C3, valid, 1.0f);
  // dP^T = V_tile dO^T
  gemm_nt(base + (size_t)k0 * C3 + 2 * C, valid, C3, d_o + (size_t)n * L * C, L, C, C, Lk, Qs, Ks, acc);
  __syncthreads();
  if (kb < Lk) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int col = kb + b * 32 + li;
          if (col < Lp) Ps[row * kLDP + col] *= (acc[a][b][r] - St[2 * kLMax + col]);
        }
  }
  // dK = scale * dS^T Q
  gemm_pv(Ps, Lp, base, L, C3, C, Vs, dqkv + ((size_t)n * L + k0) * C3 + C, C3, valid, scale);
}

template <typename K>
int set_lds_once(K kfn, int bytes, std::atomic<bool>* done) {   // idempotent: a lost race only sets the attribute twice
  if (!*done) {
    SSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    *done = true;
  }
  return SSDE_OK;
}

}  // namespace

extern "C" int ssde_attention(const ssde_attn_args* a, void* stream) {
  SSDE_REQUIRE(a && a->qkv && a->dst, "attention: null args");
  SSDE_REQUIRE(a->n > 0 && a->l > 0 && a->l <= kLMax, "attention: token count %d outside 1..%d", a->l, kLMax);
  SSDE_REQUIRE(a->c > 0 && a->c % 32 == 0, "attention: channels must be a multiple of 32 (got %d)", a->c);
  const int lds = kAttnLdsFloats * 4;
  static std::atomic<bool> attr_set{false};   // set once, outside any stream capture
  if (int rc = set_lds_once(attn_kernel, lds, &attr_set)) return rc;
  hipLaunchKernelGGL(attn_kernel, dim3(ssde_cdiv(a->l, kQB), a->n), dim3(256), lds, static_cast<hipStream_t>(stream),
                     a->qkv, a->dst, a->n, a->l, a->c, a->scale);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

extern "C" int ssde_attention_bwd(const ssde_attn_bwd_args* a, void* stream) {
  SSDE_REQUIRE(a && a->qkv && a->o && a->d_o && a->dqkv && a->stats, "attention_bwd: null args");
  SSDE_REQUIRE(a->n > 0 && a->l > 0 && a->l <= kLMax, "attention_bwd: token count %d outside 1..%d", a->l, kLMax);
  SSDE_REQUIRE(a->c > 0 && a->c % 32 == 0, "attention_bwd: channels must be a multiple of 32 (got %d)", a->c);
  const int lds = kBwdLdsFloats * 4;
  static std::atomic<bool> set_q{false}, set_kv{false};
  if (int rc = set_lds_once(attn_bwd_q_kernel, lds, &set_q)) return rc;
  if (int rc = set_lds_once(attn_bwd_kv_kernel, lds, &set_kv)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(ssde_cdiv(a->l, kQB), a->n);
  hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(256), lds, st, a->qkv, a->o, a->d_o, a->dqkv, a->stats, a->n, a->l, a->c, a->scale);
  SSDE_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_bwd_kv_kernel, grid, dim3(256), lds, st, a->qkv, a->d_o, a->dqkv, a->stats, a->n, a->l, a->c, a->scale);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
