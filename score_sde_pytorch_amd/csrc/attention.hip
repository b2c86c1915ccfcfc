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
#include <type_traits>

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

// ---- forward on the BF16 matrix pipe (SSDE_ATTNF_BF16X6): both contractions as exact-fp32 products of a 3-way bf16 split ------
// The fp32-MFMA kernel above spends 2 x 32 k matrix cycles per wave on the two contractions of a 64-row block and sits at 0.48 of
// that pipe's peak (0.217 ms per launch at 16x16, batch 256: five launches per U-Net evaluation).  v_mfma_f32_32x32x16_bf16 does
// the same contraction in 6/16 of the time (ssde_common.h: SSDE_MFMA_BF16X6, the 1x1 GEMMs' split -- every partial product exact
// in the fp32 accumulator, what is dropped is below the rounding of the product itself), so the kernel is re-cut around what then
// bounds it -- staging and LDS traffic:
//   * one workgroup of EIGHT waves per CU owns 64 query rows of one image and (almost) all of the CU's LDS;
//   * Q is split ONCE into three bf16 planes for all channels (98 KB at C = 256); K arrives in 16-channel stages, V in
//     16-token stages, split by the thread that loaded them, two stages in LDS and two more in flight in registers
//     (loads return in order; the barriers wait for LDS only);
//   * phase 1, S = Q K^T: wave w owns keys 32 w .. 32 w + 31 (two 32 x 32 blocks); phase 3, O = P V: channels 32 w .. 32 w + 31;
//   * phase 2: scaled scores -> fp32 tile (over the dead Q planes) -> every thread takes 32 consecutive keys of one row into
//     registers -> max / exp / sum over the 8 threads of a row by shuffles (the arithmetic of softmax_rows) -> P written as three
//     bf16 planes in the A-fragment layout;
//   * V is staged TRANSPOSED ([channel][16 tokens] bf16, the B fragment wants 8 consecutive tokens of a channel per lane): a
//     thread holds two tokens x four channels, packs token pairs into dwords and writes channel (j + (quad >> 1)) & 3 in its j-th
//     store, which spreads the 64 lanes of a store over the 64 banks;
//   * the four workgroups of an image run on ONE XCD (they read the same K and V: L2 hits).
// L == 256 and C % 64 == 0, C <= 256 only (the 16x16 attention of every shipped config); anything else stays on attn_kernel.
#ifndef SSDE_ATTN_X6_SCORE_TERMS
#define SSDE_ATTN_X6_SCORE_TERMS 8          // (6: an A/B variant)
#endif
namespace x6 {
constexpr int kT = 512, kRows = 64, kL = 256;
constexpr int kKStage = 3 * kL * 32;                      // bytes of one K stage: [3 pieces][256 keys][16 channels] bf16
constexpr int kPPlane = kL * kRows * 2;                   // bytes of one P piece: [16 token steps][64 rows][16 tokens] bf16
constexpr int kMain = 3 * kPPlane;                        // Q planes (<= this for C <= 256) | fp32 score tile | P planes
constexpr int kLds = kMain + 2 * kKStage;                 // (a V stage is 3 * C * 32 <= kKStage bytes)
static_assert(kRows * kLDP * 4 <= kMain, "score tile");

// x -> its three bf16 pieces as the upper halves of three dwords (ssde_split3's arithmetic)
__device__ __forceinline__ void split_hi(float x, uint32_t& a, uint32_t& b, uint32_t& c) {
  a = __builtin_bit_cast(uint32_t, x);
  const float r = x - __builtin_bit_cast(float, a & 0xffff0000u);
  b = __builtin_bit_cast(uint32_t, r);
  c = __builtin_bit_cast(uint32_t, r - __builtin_bit_cast(float, b & 0xffff0000u));
}
}  // namespace x6

// kNst = C / 16, the number of K stages, is a template parameter: the producers' code is then straight-line from the first request to
// the last wait (no loop-carried copies of a register whose load is in flight; tools/isa_inflight_check.py walks the emitted ISA)
template <int kNst>
__global__ __launch_bounds__(x6::kT) void attn_x6_kernel(const float* __restrict__ qkv, float* __restrict__ dst, int N, float scale) {
  using namespace x6;
  constexpr int C = 16 * kNst;
  SSDE_LDS(smem);
  char* lds = reinterpret_cast<char*>(smem);
  char* stage0 = lds + kMain;
  char* stage1 = stage0 + kKStage;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  // workgroup -> (image, 64-row block): the four blocks of an image are consecutive workgroups of ONE XCD
  const int xcd = blockIdx.x & 7, k_in = blockIdx.x >> 3;
  const int n = (k_in >> 2) * 8 + xcd, q0 = (k_in & 3) * kRows;
  if (n >= N) return;
  const int C3 = 3 * C;
  const float* base = qkv + (size_t)n * kL * C3;
  const int qplane = C * kRows * 2;                        // bytes of one Q piece: [C / 16][64 rows][16 channels] bf16
  constexpr int nst = kNst;                                 // K stages (a multiple of 4: C % 64 == 0)

  // ---- roles.  Waves 0-3 are CONSUMERS: each owns a 64 x 64 block of the products (64 query rows x 64 keys in phase 1, x 64
  // channels in phase 3: four 32 x 32 accumulators) and does nothing but read fragments and issue MFMAs.  Waves 4-7 are PRODUCERS:
  // they stream K and V (requests, counted waits, the 3-way split, the parking in LDS).  A SIMD holds one wave of each kind, so the
  // split runs beside the MFMAs by itself, and the fragments of a stage are read by four waves instead of eight (48 KB per
  // stage instead of 72: in the first form -- eight waves with 64 x 32 blocks, each also staging -- the LDS reads behind every barrier,
  // then the MFMAs, then the split took turns: 0.163 ms per launch, profiles/r6_attention_x6_versions.txt).
  const bool consumer = wave < 4;                          // (uniform)
  f32x16 acc[2][2];
  auto zero = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  };
  // The operands of a stage's MFMAs: A = rows (a * 32 + li) of the planes at `ap` (piece pitch a_pitch), B = rows (64 wave + b * 32
  // + li) of the stage at `bp` (piece pitch b_pitch); a lane reads 16 bytes = 8 consecutive k of its row
  struct Frags { ssde_u32x4 A[2][3], B[2][3]; };
  auto read_frags = [&](Frags& f, const char* ap, int a_pitch, const char* bp, int b_pitch) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int a = 0; a < 2; ++a) f.A[a][q] = *reinterpret_cast<const ssde_u32x4*>(ap + q * a_pitch + (a * 32 + li) * 32 + lh * 16);
#pragma unroll
      for (int b = 0; b < 2; ++b) f.B[b][q] = *reinterpret_cast<const ssde_u32x4*>(bp + q * b_pitch + (wave * 64 + b * 32 + li) * 32 + lh * 16);
    }
  };
  // smallest terms first.  Six terms (SSDE_MFMA_BF16X6) drop a1 b2 + a2 b1 + a2 b2 <= 2^-23 |a b| per product; the scores take
  // EIGHT (all but a2 b2 <= 2^-32 |a b|): with pieces taken by truncation the dropped terms are a one-sided error, and the scores
  // of a peaked query row (|q| ~ 10) carry it into the exponent.  P V, with P in [0, 1], keeps six.
  auto mfmas = [&](auto Terms, const Frags& f) __attribute__((always_inline)) {
    constexpr int kTerms = decltype(Terms)::value;
    constexpr int TI[8] = {1, 2, 0, 2, 1, 0, 1, 0}, TJ[8] = {2, 1, 2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = 8 - kTerms; t < 8; ++t)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssde_bf16x8, f.A[a][TI[t]]),
                                                              __builtin_bit_cast(ssde_bf16x8, f.B[b][TJ[t]]), acc[a][b], 0, 0, 0);
  };
  using Six = std::integral_constant<int, 6>;
  using ScoreTerms = std::integral_constant<int, SSDE_ATTN_X6_SCORE_TERMS>;

  // ---- the streamed operands (producers): K in nst stages of 16 channels, then V in 16 stages of 16 tokens, ONE stream through a
  // register ring of four elements.  Element u + 4 is requested at the top of stage u -- the first V stages during the last K
  // stages, so they land under the softmax -- and has three stages to land.  The loads are asm statements hipcc does not track,
  // four per element and thread, counted by the kernel itself: "element u + 1 has landed" is vmcnt(12) (elements u + 2 .. u + 4 are
  // younger), 8 / 4 / 0 over the last V stages.  Every requested element is consumed (no load lands in a register hipcc considers
  // free) and the whole stream lives inside the producers' arm of the role branch (no copy of a register in flight at a join).
  // History: loads hipcc tracked ran one stage ahead in effect -- it waits vmcnt(0) in front of every store once a guard sits
  // near them: 50 us per workgroup.
  const int ptid = tid & 255;                               // thread of the producer half
  const float* kbase = base + C;
  const float* vbase = base + 2 * C;
  // K: a producer thread stages (key row (ptid >> 2) + 64 i, channel quad ptid & 3), i = 0 .. 3, of the stage's 256 x 4
  const int krow = ptid >> 2, kf = ptid & 3;
  const uint32_t koff = (uint32_t)((krow * C3 + kf * 4) * 4), kstep = (uint32_t)(64 * C3 * 4);
  // V: tokens 2 tp, 2 tp + 1 of the channel quads fq0 and fq0 + 32
  const int tp = ptid & 7, fq0 = ptid >> 3;
  const bool v_on0 = fq0 * 4 < C, v_on1 = (fq0 + 32) * 4 < C;
  const uint32_t voff = (uint32_t)((2 * tp * C3 + (v_on0 ? fq0 * 4 : 0)) * 4), vtok = (uint32_t)(C3 * 4), vquad = v_on1 ? 512u : 0u;
  const int vpitch = C * 32;                                // bytes of one piece of a V stage: [C channels][16 tokens] bf16
  auto request = [&](int u, ssde_f32x4 (&Rs)[4]) __attribute__((always_inline)) {      // stream element u (uniform)
    const bool is_k = u < nst;
    const float* sb = is_k ? kbase + u * 16 : vbase + (size_t)(u - nst) * 16 * C3;       // (scalar)
    const uint32_t o0 = is_k ? koff : voff;
    const uint32_t o1 = o0 + (is_k ? kstep : vtok);                    // K: row + 64       V: token 2 tp + 1
    const uint32_t o2 = is_k ? o0 + 2 * kstep : o0 + vquad;            // K: row + 128      V: quad fq0 + 32, token 2 tp
    const uint32_t o3 = is_k ? o0 + 3 * kstep : o2 + vtok;             // K: row + 192      V: quad fq0 + 32, token 2 tp + 1
    SSDE_GLOAD16_I_SAFE(Rs[0], o0, sb, 0);
    SSDE_GLOAD16_I_SAFE(Rs[1], o1, sb, 0);
    SSDE_GLOAD16_I_SAFE(Rs[2], o2, sb, 0);
    SSDE_GLOAD16_I_SAFE(Rs[3], o3, sb, 0);
  };
#ifdef SSDE_EMULATED
#define SSDE_X6_WAIT(n, Rs) ((void)0)
#else
#define SSDE_X6_WAIT(n, Rs) asm volatile("s_waitcnt vmcnt(%4)" : "+v"((Rs)[0]), "+v"((Rs)[1]), "+v"((Rs)[2]), "+v"((Rs)[3]) : "n"(n) : "memory")
#endif
  // a K element split and parked: [3 pieces][256 keys][16 channels] bf16
  auto park_k = [&](const ssde_f32x4 (&Rs)[4], char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint2 p0, p1, p2;
      ssde_split3(make_float4(Rs[i].x, Rs[i].y, Rs[i].z, Rs[i].w), p0, p1, p2);
      char* d = buf + (krow + 64 * i) * 32 + kf * 8;
      *reinterpret_cast<uint2*>(d) = p0;
      *reinterpret_cast<uint2*>(d + kL * 32) = p1;
      *reinterpret_cast<uint2*>(d + 2 * kL * 32) = p2;
    }
  };
  // a V element, TRANSPOSED: [3 pieces][C channels][16 tokens] bf16, a dword = (token 2 tp) | (token 2 tp + 1) << 16.  The j-th
  // store of a thread writes channel (j + rot) & 3 of its quad, rot = (fq >> 1) & 3: the 64 lanes of one store hit 64 different
  // banks.  The rotation of the four words is two rounds of selects (v_cndmask), no branches
  const int rot = (fq0 >> 1) & 3;                           // (the same for fq0 + 32)
  const bool rot1 = (rot & 1) != 0, rot2 = (rot & 2) != 0;
  auto park_v = [&](const ssde_f32x4 (&Rs)[4], char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {                           // quad fq0 + 32 h
      const float f0[4] = {Rs[2 * h].x, Rs[2 * h].y, Rs[2 * h].z, Rs[2 * h].w};
      const float f1[4] = {Rs[2 * h + 1].x, Rs[2 * h + 1].y, Rs[2 * h + 1].z, Rs[2 * h + 1].w};
      uint32_t w[3][4], y[3][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t a0, b0, c0, a1, b1, c1;
        split_hi(f0[i], a0, b0, c0);
        split_hi(f1[i], a1, b1, c1);
        w[0][i] = ssde_pack_hi16(a0, a1); w[1][i] = ssde_pack_hi16(b0, b1); w[2][i] = ssde_pack_hi16(c0, c1);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        uint32_t x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = rot1 ? w[q][(i + 1) & 3] : w[q][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) y[q][i] = rot2 ? x[(i + 2) & 3] : x[i];
      }
      if (h == 0 ? v_on0 : v_on1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          char* d = buf + ((fq0 + 32 * h) * 4 + ((j + rot) & 3)) * 32 + tp * 4;
#pragma unroll
          for (int q = 0; q < 3; ++q) *reinterpret_cast<uint32_t*>(d + q * vpitch) = y[q][j];
        }
      }
    }
  };

  // ---- phase 1: S = Q K^T ----
  ssde_f32x4 R[4][4];                                       // (producers)
  zero();
  if (!consumer) { request(0, R[0]); request(1, R[1]); request(2, R[2]); request(3, R[3]); }       // before Q: they land under its split
  {
    // Q: every channel of the 64 rows, split once by all eight waves (up to 8 float4 of a thread requested together)
    const int cq4 = C >> 2, total = kRows * cq4;            // channel quads per row
    for (int item0 = tid; item0 < total; item0 += 8 * kT) {
      float4 qv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int item = min(item0 + u * kT, total - 1);
        const int row = item / cq4, cq = item - row * cq4;
        qv[u] = *reinterpret_cast<const float4*>(base + (size_t)(q0 + row) * C3 + cq * 4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int item = item0 + u * kT;
        if (item < total) {
          const int row = item / cq4, cq = item - row * cq4;
          uint2 p0, p1, p2;
          ssde_split3(qv[u], p0, p1, p2);
          char* d = lds + (cq >> 2) * (kRows * 32) + row * 32 + (cq & 3) * 8;
          *reinterpret_cast<uint2*>(d) = p0;
          *reinterpret_cast<uint2*>(d + qplane) = p1;
          *reinterpret_cast<uint2*>(d + 2 * qplane) = p2;
        }
      }
    }
  }
  if (consumer) {
    SSDE_LDS_BARRIER();                                    // (the Q planes and K stage 0 are complete)
    for (int u = 0; u < nst; ++u) {
      Frags f;
      read_frags(f, lds + u * (kRows * 32), qplane, (u & 1) ? stage1 : stage0, kL * 32);
      mfmas(ScoreTerms{}, f);
      SSDE_LDS_BARRIER();
    }
  } else {
    // (the Q loads above are younger than the four requests and have been consumed: loads return in order, so the ring has landed)
    SSDE_X6_WAIT(12, R[0]);
    park_k(R[0], stage0);
    SSDE_LDS_BARRIER();
#pragma unroll
    for (int u = 0; u < nst; ++u) {                        // (straight-line: every index is a compile-time constant)
      const int j = u & 3;
      request(u + 4, R[j]);                                 // R[j] held element u: parked in LDS during stage u - 1
      SSDE_X6_WAIT(12, R[(j + 1) & 3]);                     // element u + 1 (at u = nst - 1: V stage 0, parked in phase 3)
      if (u + 1 < nst) park_k(R[(j + 1) & 3], (j & 1) ? stage0 : stage1);
      SSDE_LDS_BARRIER();
    }
    // V stages 1 .. 3 are still in flight: waited for HERE, before the softmax -- its 32 scores per thread could make hipcc move
    // the ring, and a register must not be copied while its load is in flight (they were requested 1-3 stages ago)
    SSDE_X6_WAIT(8, R[1]); SSDE_X6_WAIT(4, R[2]); SSDE_X6_WAIT(0, R[3]);
  }

  // ---- phase 2: softmax of the 64 x 256 scores; P as three bf16 planes (all eight waves) ----
  {
    float* tile = smem;                                    // [64][kLDP] over the Q planes (every wave is behind the last barrier)
    if (consumer) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            tile[row * kLDP + wave * 64 + b * 32 + li] = acc[a][b][r] * scale;
          }
    }
    SSDE_LDS_BARRIER();
    const int row = tid >> 3, sub = tid & 7;               // keys 32 sub .. 32 sub + 31 of the row
    float x[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(tile + row * kLDP + sub * 32 + i * 4);
      x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
    SSDE_LDS_BARRIER();                                    // the tile is in registers: the P planes may overwrite it
    float m = x[0];
#pragma unroll
    for (int i = 1; i < 32; ++i) m = fmaxf(m, x[i]);
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    m = fmaxf(m, __shfl_xor(m, 4, 64));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { x[i] = __expf(x[i] - m); sum += x[i]; }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int u = 0; u < 4; ++u) {                          // 16-byte units: token step 2 sub + (u >> 1), half u & 1
      uint2 a0, a1, a2, b0, b1, b2;
      ssde_split3(make_float4(x[8 * u] * inv, x[8 * u + 1] * inv, x[8 * u + 2] * inv, x[8 * u + 3] * inv), a0, a1, a2);
      ssde_split3(make_float4(x[8 * u + 4] * inv, x[8 * u + 5] * inv, x[8 * u + 6] * inv, x[8 * u + 7] * inv), b0, b1, b2);
      char* d = lds + (2 * sub + (u >> 1)) * (kRows * 32) + row * 32 + (u & 1) * 16;
      *reinterpret_cast<uint4*>(d) = make_uint4(a0.x, a0.y, b0.x, b0.y);
      *reinterpret_cast<uint4*>(d + kPPlane) = make_uint4(a1.x, a1.y, b1.x, b1.y);
      *reinterpret_cast<uint4*>(d + 2 * kPPlane) = make_uint4(a2.x, a2.y, b2.x, b2.y);
    }
  }

  // ---- phase 3: O = P V over 16 stages of 16 tokens ----
  if (consumer) {
    const bool w_on = wave * 64 < C;                        // (C < 256: the upper consumers only keep the barriers)
    zero();
    SSDE_LDS_BARRIER();                                    // (the P planes and V stage 0 are complete)
    for (int v = 0; v < 16; ++v) {
      if (w_on) {
        Frags f;
        read_frags(f, lds + v * (kRows * 32), kPPlane, (v & 1) ? stage1 : stage0, vpitch);
        mfmas(Six{}, f);
      }
      SSDE_LDS_BARRIER();
    }
    if (w_on) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float* out = dst + ((size_t)n * kL + q0) * C + wave * 64 + b * 32 + li;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            out[(size_t)row * C] = acc[a][b][r];
          }
      }
    }
  } else {
    park_v(R[0], stage0);                                  // (V stages 0 .. 3 have landed: end of phase 1)
    SSDE_LDS_BARRIER();
#pragma unroll
    for (int v = 0; v < 16; ++v) {                         // (straight-line: every index below is a compile-time constant)
      const int j = v & 3;
      if (v + 4 < 16) request(nst + v + 4, R[j]);
      if (v + 1 < 16) {
        // element v + 1: stages 1 .. 3 landed before the softmax; from 4 on, the stages requested behind it are younger
        // (v = 3: stages 5 .. 7 and nothing older in flight; the last ones: 8 / 4 / 0)
        if (v + 1 >= 4) {
          if (v + 4 < 16) SSDE_X6_WAIT(12, R[(j + 1) & 3]);
          else if (v + 3 < 16) SSDE_X6_WAIT(8, R[(j + 1) & 3]);
          else if (v + 2 < 16) SSDE_X6_WAIT(4, R[(j + 1) & 3]);
          else SSDE_X6_WAIT(0, R[(j + 1) & 3]);
        }
        park_v(R[(j + 1) & 3], (j & 1) ? stage0 : stage1);
      }
      SSDE_LDS_BARRIER();
    }
  }
#undef SSDE_X6_WAIT
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
  if ((a->flags & SSDE_ATTNF_BF16X6) && a->l == x6::kL && a->c <= 256 && a->c % 64 == 0) {
    static std::atomic<bool> x6_set[4];
    const dim3 grid(ssde_cdiv(a->n, 8) * 8 * 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define SSDE_ATTN_X6_GO(NST, SLOT)                                                                       \
    do {                                                                                                 \
      if (int rc = set_lds_once(attn_x6_kernel<NST>, x6::kLds, &x6_set[SLOT])) return rc;                \
      hipLaunchKernelGGL(attn_x6_kernel<NST>, grid, dim3(x6::kT), x6::kLds, st, a->qkv, a->dst, a->n, a->scale); \
    } while (0)
    switch (a->c) {
      case 64: SSDE_ATTN_X6_GO(4, 0); break;
      case 128: SSDE_ATTN_X6_GO(8, 1); break;
      case 192: SSDE_ATTN_X6_GO(12, 2); break;
      default: SSDE_ATTN_X6_GO(16, 3); break;
    }
#undef SSDE_ATTN_X6_GO
    SSDE_LAUNCH_CHECK();
    return SSDE_OK;
  }
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
