// 1x1 convolutions / NIN / Linear layers as a plain fp32 GEMM on the matrix pipe (ddpm_conv1x1 models/layers.py:100-105,
// NIN layers.py:546-555, nn.Linear at ncsnpp.py:86-91 and layerspp.py:227,263, and the input-gradients of all of them):
//
//   out[m, j] = scale * ( sum_k pro(src)[m, k] * W[k, j] + bias[j] + chan_add[img(m), j] (+ resid[m, j]) ) (+ resid[m, j])
//
// m runs over the N*H*W pixels (NHWC rows are contiguous, so the pixel operand is a row-major [M, K] matrix and no
// spatial tiling is needed), the source may be a virtual concat (p0 | p1) with the usual GroupNorm / SiLU / dropout
// prologue.  The general kernel (conv_mfma.hip) runs these layers as its "aux" phase with 64-wide output tiles and a
// single LDS stage: the pixel operand is re-read Cout/64 times and every 32-channel chunk costs two barriers -- 60-85
// TF/s at the BASELINE shapes.  Here:
//   workgroup = 4 waves = 128 pixels x 128 output channels, each wave a 64 x 64 block (2 x 2 v_mfma_f32_32x32x2_f32 tiles,
//   64 accumulator registers); K advances 32 channels per stage through DOUBLE-BUFFERED LDS, one barrier per stage:
//   the global loads of stage s+1 are issued before the 64 MFMAs of stage s (4096 matrix cycles = 1.7 us, an L2 / HBM
//   latency; with 16-channel stages the kernel was load-latency bound) and parked in LDS after them.
//   Both operands sit K-major in LDS with a row pitch of 34 floats: a lane's ds_read_b64 is the channel pair
//   (4t + 2h, 4t + 2h + 1) of its row (h = lane >> 5 = the MFMA k index), one read feeds two MFMAs, and the 32 rows of a
//   half-wave cover all 64 banks (34 i mod 64 hits every even bank once).
//   Workgroups that share a pixel tile are consecutive on ONE XCD (blocks are dealt round-robin to the 8 XCDs), so the
//   second output-channel tile finds the pixel rows in that XCD's L2.
// The epilogue is the shared coalesced one (ssde_store_tile): accumulators -> LDS tile -> float4 rows.
#include "ssde_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// -DSSDE_GEMM_TRACE (tools/gemm_trace.py, a variant library only): s_memtime stamps of wave 0 of the first workgroup
#ifdef SSDE_GEMM_TRACE
__device__ unsigned long long* g_gemm_trace;
extern "C" int ssde_debug_gemm_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -5;
}
#define SSDE_GT(slot)                                                                  \
  do {                                                                                 \
    if (gt_on) g_gemm_trace[(slot)] = __builtin_amdgcn_s_memtime();                    \
  } while (0)
// every workgroup's start / end of loop / end (+ its hardware id): the rounds of a launch and how far they run in lock step
__device__ unsigned long long* g_gemm_wg_trace;
extern "C" int ssde_debug_gemm_wg_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_wg_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -5;
}
#define SSDE_GW(slot)                                                                                          \
  do {                                                                                                         \
    if (threadIdx.x == 0 && g_gemm_wg_trace != nullptr) {                                                      \
      g_gemm_wg_trace[(size_t)blockIdx.x * 4 + (slot)] = __builtin_amdgcn_s_memtime();                         \
      if ((slot) == 0) g_gemm_wg_trace[(size_t)blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32); \
    }                                                                                                          \
  } while (0)
#else
#define SSDE_GT(slot) do { } while (0)
#define SSDE_GW(slot) do { } while (0)
#endif

// epilogue rows in flight per thread (12 registers each) and workgroups per CU the kernel is compiled for
#ifndef SSDE_GEMM_BATCH
#define SSDE_GEMM_BATCH 1
#endif
#ifndef SSDE_GEMM_OCC
#define SSDE_GEMM_OCC 4
#endif

namespace {

constexpr int kThreads = 256;
constexpr int BM = 128, BN = 128, BK = 16, LDK = 18;
constexpr int kStage = (BM + BN) * LDK;           // floats per LDS stage (A rows then B rows)

struct GemmParams {
  ssde_src src;
  const float* wpk;        // [ceil(K/8)][CoutPad][8]
  int M, HW, K, Cout, CoutPad, m_tiles, n_tiles;
  const float* bias; const float* chan_add; int chan_add_ld;
  const float* resid; int resid_post;
  float scale;
  float* dst;
  float* gn_part;          // GroupNorm partials of dst: per 64-row half tile (HW % 64 == 0) or per whole image (HW < 64), see ssde_store_tile
  int lHW, gn_entries;
};

template <bool kGn>
__global__ __launch_bounds__(kThreads, SSDE_GEMM_OCC) void gemm1x1_kernel(const GemmParams p) {
  SSDE_LDS(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
  const int nt = lin % p.n_tiles, mt = (lin / p.n_tiles) * 8 + xcd;
#ifdef SSDE_GEMM_TRACE
  const bool gt_on = tid == 0 && blockIdx.x == 0 && g_gemm_trace != nullptr;
#endif
  SSDE_GT(0);
  if (mt >= p.m_tiles) return;
  const int m0 = mt * BM, n0 = nt * BN;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const ssde_src& s = p.src;
  SsdePro pro = ssde_pro_decode(s);
  pro.gn = kGn;
  const int Ctot = s.c0 + s.c1;
  const int cpg = kGn ? Ctot / s.gn_groups : 1;
  const int nst = (p.K + BK - 1) / BK;
  const int ncin8 = (p.K + 7) >> 3;

  // ---- staging plan: thread = (rows r0 = tid >> 3 + 32 i, channel quad f = tid & 7) of both operands ----
  constexpr int NI = 2, RS = 64;
  const int f = tid & 3, r0 = tid >> 2;
  int arow[NI], aimg[NI];
  bool aok[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = m0 + r0 + i * RS;
    aok[i] = m < p.M;
    arow[i] = aok[i] ? m : 0;
    aimg[i] = arow[i] / p.HW;
  }
  bool bok[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) bok[i] = (n0 + r0 + i * RS) < p.CoutPad;
  const float* bptr = p.wpk + ((size_t)n0 + r0) * 8 + (f & 1) * 4;      // + 8-channel chunk * CoutPad * 8

  float4 av[NI], bv[NI];
  float mu[NI], rs[NI];
  float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
  bool k_ok = false;
  int c_cur = 0;

  // branch-free global loads of stage st (rows / channels outside the problem read a clamped address and are zeroed below)
  auto load_stage = [&](int st) {
    const int c_base = st * BK;
    const bool second = c_base >= s.c0;
    const float* base = second ? s.p1 : s.p0;
    const int C = second ? s.c1 : s.c0;
    const int cthr = (second ? c_base - s.c0 : c_base) + f * 4;
    c_cur = c_base + f * 4;
    k_ok = c_cur < p.K;
    const float* ap = base + (k_ok ? cthr : 0);
#pragma unroll
    for (int i = 0; i < NI; ++i) av[i] = *reinterpret_cast<const float4*>(ap + (size_t)arow[i] * C);
    const int cin8 = min(st * 2 + (f >> 1), ncin8 - 1);
    const float* bp = bptr + (size_t)cin8 * p.CoutPad * 8;
#pragma unroll
    for (int i = 0; i < NI; ++i) bv[i] = *reinterpret_cast<const float4*>(bp + (bok[i] ? i * RS * 8 : 0));
    if (kGn) {
      const int cg = k_ok ? c_cur : 0;
      gam = *reinterpret_cast<const float4*>(s.gn_gamma + cg);
      bet = *reinterpret_cast<const float4*>(s.gn_beta + cg);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int gi = aimg[i] * s.gn_groups + cg / cpg;
        mu[i] = s.gn_mean[gi];
        rs[i] = s.gn_rstd[gi];
      }
    }
  };
  auto store_stage = [&](int st, float* buf) {
    const int cin8 = st * 2 + (f >> 1);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (aok[i] && k_ok)
        v = ssde_pro_apply(av[i], mu[i], rs[i], gam, bet, (uint32_t)arow[i] * (uint32_t)Ctot + (uint32_t)c_cur, pro);
      float* d = buf + (r0 + i * RS) * LDK + f * 4;
      *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
      float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bok[i] && cin8 < ncin8) w = bv[i];
      float* e = buf + (BM + r0 + i * RS) * LDK + f * 4;
      *reinterpret_cast<float2*>(e) = make_float2(w.x, w.y);
      *reinterpret_cast<float2*>(e + 2) = make_float2(w.z, w.w);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  int aoff[2], boff[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) aoff[a] = (wm0 + a * 32 + li) * LDK + 2 * lh;
#pragma unroll
  for (int b = 0; b < 2; ++b) boff[b] = (BM + wn0 + b * 32 + li) * LDK + 2 * lh;

  SSDE_GT(1);
  load_stage(0);
  store_stage(0, smem);
  __syncthreads();
  SSDE_GT(2);
  for (int st = 0; st < nst; ++st) {
    const float* cur = smem + (st & 1) * kStage;
    const bool has_next = st + 1 < nst;
    if (has_next) load_stage(st + 1);
    float2 af[2][2], bf[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) af[0][a] = *reinterpret_cast<const float2*>(cur + aoff[a]);
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[0][b] = *reinterpret_cast<const float2*>(cur + boff[b]);
#pragma unroll
    for (int t = 0; t < BK / 4; ++t) {
      const int c = t & 1;
      if (t + 1 < BK / 4) {
#pragma unroll
        for (int a = 0; a < 2; ++a) af[c ^ 1][a] = *reinterpret_cast<const float2*>(cur + aoff[a] + (t + 1) * 4);
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[c ^ 1][b] = *reinterpret_cast<const float2*>(cur + boff[b] + (t + 1) * 4);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][a].x, bf[c][b].x, acc[a][b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][a].y, bf[c][b].y, acc[a][b], 0, 0, 0);
      // the fragment reads of step t+1 go out before the 8 MFMAs of step t (512 matrix cycles cover their latency)
      if (t + 1 < BK / 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
    if (st < 8) SSDE_GT(4 + st * 4);
    if (has_next) store_stage(st + 1, smem + ((st + 1) & 1) * kStage);
    if (st < 8) SSDE_GT(5 + st * 4);
    __syncthreads();
    if (st < 8) SSDE_GT(6 + st * 4);
  }
  SSDE_GT(40);

  // ---- epilogue: accumulators -> LDS tile [64][BN + 4] -> coalesced float4 rows, the two 64-row halves in turn
  // (a half tile is 33 KB: with the 37 KB of operand stages four workgroups fit a CU and cover each other's pipeline
  // fill and epilogue; the loops are short -- K = 128 .. 512 -- so those fixed costs matter)
  constexpr int LDT = BN + 4;
  SsdeEpi e{p.bias, p.chan_add, p.chan_add_ld, p.resid, p.resid_post, p.scale, p.dst, p.Cout, p.gn_part};
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if ((wave >> 1) == half) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            smem[m * LDT + wn0 + b * 32 + li] = acc[a][b][r];
          }
    }
    __syncthreads();
    ssde_store_tile<64, BN, kThreads, SSDE_GEMM_BATCH>(smem, LDT, n0, e, [&](int row, size_t& pix, int& img) {
      const int m = m0 + half * 64 + row;
      if (m >= p.M) return false;
      pix = (size_t)m;
      img = m / p.HW;
      return true;
    }, p.gn_part ? (p.HW >= 64 ? (m0 >> 6) + half : (m0 + half * 64) / p.HW) : -1, p.HW >= 64 ? 30 : p.lHW, p.gn_entries);
    SSDE_GT(41 + half * 2);
    __syncthreads();
    SSDE_GT(42 + half * 2);
  }
}

// ---- the same GEMM through the BF16 matrix pipe: exact-fp32 products by a 3-way bf16 split (SSDE_MATRIX=bf16x6) ----------
// Both operands are split while they are staged (ssde_split3: 5.5 VALU per element, which the BF16 MFMAs -- unlike the
// fp32 ones -- co-issue with); LDS holds six planes per stage, [piece][rows][16 channels] bf16 for the pixel rows and
// the same for the weight rows: a lane's 16 bytes of a fragment are channels 8 k .. 8 k + 7 (k = lane >> 5) of its row, the 64
// lanes of a ds_read_b128 cover 1 KB contiguously.  Per 16-channel stage a wave issues 6 MFMAs of 32 cycles per 32 x 32 block
// (768 matrix cycles for a 64 x 64 wave tile against 2048 of the fp32 kernel for the same channels).
// With the matrix time cut by 2.7x the kernel is bound by what surrounds it -- the latency of a stage's global loads, the
// store burst of the epilogue, the tail of the last partial round of workgroups -- so the shape is a parameter:
//   kBM  = 128 (waves 2 x 2 of 64 x 64; 48 KB LDS, 3 workgroups per CU) or 64 (waves 2 x 2 of 32 x 64; 36 KB, 4 per CU: twice
//          the workgroups, each a quarter of a CU's registers -- fuller rounds, half the reuse of a staged weight tile)
//   kPF  = stages the loads of the pixel rows run ahead of their use (1: issued before the MFMAs of the previous stage; 2: one
//          stage earlier still -- a second register set and a loop unrolled by two, which costs a workgroup per CU of
//          occupancy: hipcc needs ~190 / ~125 registers for it)
// SSDE_X6_BM / SSDE_X6_PF pick the instantiation per call (A/B: tools/matrix_ab.py, profiles/r4_bf16x6_gemm_*.txt).
//   kBN  = 128, or 256 (round 6, "wide": waves 2 x 2 of 64 x 128, 128 accumulator registers, 72 KB LDS, 2 workgroups per CU).
//          With BN = 128 a 16-channel stage stages 4 float4 per thread (GroupNorm + split: ~140 VALU) for 24 MFMAs of a wave, and
//          a layer with Cout = 768 normalises and splits its pixel rows six times; the wide tile stages 6 float4 for 48 MFMAs
//          and reads 18 instead of 24 fragments per 48 MFMAs from LDS.
#ifndef SSDE_X6_PF2_OCC_LOSS
#define SSDE_X6_PF2_OCC_LOSS 1
#endif
constexpr int XBK = 16;
template <int kBM, int kBN = 128> struct X6 {
  static constexpr int kPlaneA = kBM * XBK * 2, kPlaneB = kBN * XBK * 2;     // bytes of one piece plane
  static constexpr int kStageBytes = 3 * kPlaneA + 3 * kPlaneB;
  static constexpr int NA = kBM / 64;                                        // 32-row blocks of a wave / staged rows per thread
  static constexpr int NBW = kBN / 64;                                       // 32-column blocks of a wave / staged weight rows per thread
};

template <bool kGn, int kBM, int kPF, int kBN = 128>
__global__ __launch_bounds__(kThreads, kBN == 256 ? 2 : (kBM == 64 ? 4 : 3) - (kPF == 2 ? SSDE_X6_PF2_OCC_LOSS : 0)) void gemm1x1_bf16x6_kernel(const GemmParams p) {
  using X = X6<kBM, kBN>;
  constexpr int NA = X::NA, NB = X::NBW, NBW = X::NBW, RS = 64;
  SSDE_LDS(smem);
  char* lds = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
  const int nt = lin % p.n_tiles, mt = (lin / p.n_tiles) * 8 + xcd;
#ifdef SSDE_GEMM_TRACE
  const bool gt_on = tid == 0 && blockIdx.x == 0 && g_gemm_trace != nullptr;
#endif
  SSDE_GT(0);
  if (mt >= p.m_tiles) return;
  SSDE_GW(0);
  const int m0 = mt * kBM, n0 = nt * kBN;
  const int wm0 = (wave >> 1) * (kBM / 2), wn0 = (wave & 1) * (kBN / 2);
  const ssde_src& s = p.src;
  SsdePro pro = ssde_pro_decode(s);
  pro.gn = kGn;
  const int Ctot = s.c0 + s.c1;
  const int cpg = kGn ? Ctot / s.gn_groups : 1;
  const int nst = (p.K + XBK - 1) / XBK;
  const int ncin8 = (p.K + 7) >> 3;

  // staging plan: thread = (rows r0 + 64 i, channel quad f) of both operands
  const int f = tid & 3, r0 = tid >> 2;
  int arow[NA], aimg[NA];
  bool aok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int m = m0 + r0 + i * RS;
    aok[i] = m < p.M;
    arow[i] = aok[i] ? m : 0;
    aimg[i] = arow[i] / p.HW;
  }
  bool bok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) bok[i] = (n0 + r0 + i * RS) < p.CoutPad;
  const float* bptr = p.wpk + ((size_t)n0 + r0) * 8 + (f & 1) * 4;

  // registers of a stage in flight: the pixel rows (the operand that comes from HBM: these are the loads that run kPF stages
  // ahead) apart from the rest (weights and GroupNorm parameters: L2 / L1 hits, always one stage ahead)
  struct ARegs { float4 av[NA]; };
  struct StageRegs {
    float4 bv[NB];
    float mu[NA], rs[NA];
    float4 gam, bet;
  };
  auto load_a = [&](int st, ARegs& A) __attribute__((always_inline)) {
    const int c_base = st * XBK;
    const bool second = c_base >= s.c0;
    const float* base = second ? s.p1 : s.p0;
    const int C = second ? s.c1 : s.c0;
    const int cthr = (second ? c_base - s.c0 : c_base) + f * 4;
    const float* ap = base + (c_base + f * 4 < p.K ? cthr : 0);
#pragma unroll
    for (int i = 0; i < NA; ++i) A.av[i] = *reinterpret_cast<const float4*>(ap + (size_t)arow[i] * C);
  };
  auto load_rest = [&](int st, StageRegs& R) __attribute__((always_inline)) {
    const int c_cur = st * XBK + f * 4;
    const int cin8 = min(st * 2 + (f >> 1), ncin8 - 1);
    const float* bp = bptr + (size_t)cin8 * p.CoutPad * 8;
#pragma unroll
    for (int i = 0; i < NB; ++i) R.bv[i] = *reinterpret_cast<const float4*>(bp + (bok[i] ? i * RS * 8 : 0));
    R.gam = make_float4(1.f, 1.f, 1.f, 1.f); R.bet = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NA; ++i) { R.mu[i] = 0.f; R.rs[i] = 1.f; }
    if (kGn) {
      const int cg = c_cur < p.K ? c_cur : 0;
      R.gam = *reinterpret_cast<const float4*>(s.gn_gamma + cg);
      R.bet = *reinterpret_cast<const float4*>(s.gn_beta + cg);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int gi = aimg[i] * s.gn_groups + cg / cpg;
        R.mu[i] = s.gn_mean[gi];
        R.rs[i] = s.gn_rstd[gi];
      }
    }
  };
  auto store_stage = [&](int st, const ARegs& A, const StageRegs& R, char* buf) __attribute__((always_inline)) {
    const int cin8 = st * 2 + (f >> 1);
    const int c_cur = st * XBK + f * 4;
    const bool k_ok = c_cur < p.K;
    uint2 q0, q1, q2;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (aok[i] && k_ok)
        v = ssde_pro_apply(A.av[i], R.mu[i], R.rs[i], R.gam, R.bet, (uint32_t)arow[i] * (uint32_t)Ctot + (uint32_t)c_cur, pro);
      char* d = buf + (r0 + i * RS) * (XBK * 2) + f * 8;
      ssde_split3(v, q0, q1, q2);
      *reinterpret_cast<uint2*>(d) = q0;
      *reinterpret_cast<uint2*>(d + X::kPlaneA) = q1;
      *reinterpret_cast<uint2*>(d + 2 * X::kPlaneA) = q2;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bok[i] && cin8 < ncin8) w = R.bv[i];
      char* d = buf + 3 * X::kPlaneA + (r0 + i * RS) * (XBK * 2) + f * 8;
      ssde_split3(w, q0, q1, q2);
      *reinterpret_cast<uint2*>(d) = q0;
      *reinterpret_cast<uint2*>(d + X::kPlaneB) = q1;
      *reinterpret_cast<uint2*>(d + 2 * X::kPlaneB) = q2;
    }
  };

  f32x16 acc[NA][NBW];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int aoff = (wm0 + li) * (XBK * 2) + lh * 16, boff = 3 * X::kPlaneA + (wn0 + li) * (XBK * 2) + lh * 16;
  // the 6 x NA x NBW MFMAs of a stage, term by term over the blocks of a column pair: consecutive MFMAs never share an
  // accumulator; the weight fragments of one column pair at a time (12 registers per block: the wide tile would hold 48)
  auto mfma_stage = [&](const char* cur) __attribute__((always_inline)) {
    ssde_u32x4 A[NA][3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int a = 0; a < NA; ++a) A[a][q] = *reinterpret_cast<const ssde_u32x4*>(cur + aoff + q * X::kPlaneA + a * 32 * (XBK * 2));
    constexpr int TI[6] = {0, 2, 1, 0, 1, 0}, TJ[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int b0 = 0; b0 < NBW; b0 += 2) {
      ssde_u32x4 B[2][3];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) B[b][q] = *reinterpret_cast<const ssde_u32x4*>(cur + boff + q * X::kPlaneB + (b0 + b) * 32 * (XBK * 2));
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b0 + b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssde_bf16x8, A[a][TI[t]]),
                                                                     __builtin_bit_cast(ssde_bf16x8, B[b][TJ[t]]), acc[a][b0 + b], 0, 0, 0);
    }
  };

  SSDE_GT(1);
  char* buf0 = lds;
  char* buf1 = lds + X::kStageBytes;
  StageRegs R;
  if constexpr (kPF == 1) {
    ARegs A0;
    load_a(0, A0); load_rest(0, R);
    store_stage(0, A0, R, buf0);
    __syncthreads();
    SSDE_GT(2);
    for (int st = 0; st < nst; ++st) {
      const bool has_next = st + 1 < nst;
      if (has_next) { load_a(st + 1, A0); load_rest(st + 1, R); }
      mfma_stage((st & 1) ? buf1 : buf0);
      if (st < 8) SSDE_GT(4 + st * 4);
      if (has_next) store_stage(st + 1, A0, R, (st & 1) ? buf0 : buf1);
      if (st < 8) SSDE_GT(5 + st * 4);
      __syncthreads();
      if (st < 8) SSDE_GT(6 + st * 4);
    }
  } else {
    // two register sets for the pixel rows: the loads of stage st + 2 are issued before the MFMAs of stage st and are not
    // needed before the end of stage st + 1 (two stage times of cover for their HBM latency); the loop is unrolled by two so
    // that both sets are compile-time names
    ARegs A0, A1;
    // every load of the loop is UNCONDITIONAL (a stage index past the end is clamped to the last stage: a redundant L2 hit whose
    // registers nobody parks): a load behind a guard is one hipcc's wait-count pass cannot rely on, and it then waits vmcnt(0/1)
    // for the oldest set -- i.e. for the set it had just requested as well
    const int last = nst - 1;
    load_a(0, A0); load_rest(0, R);
    load_a(min(1, last), A1);
    store_stage(0, A0, R, buf0);
    __syncthreads();
    SSDE_GT(2);
    for (int st = 0; st < nst; st += 2) {
      load_rest(min(st + 1, last), R);
      __builtin_amdgcn_sched_barrier(0);          // (hipcc's scheduler would put the pixel rows' loads first again)
      load_a(min(st + 2, last), A0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_stage(buf0);
      if (st < 8) SSDE_GT(4 + st * 4);
      if (st + 1 < nst) store_stage(st + 1, A1, R, buf1);
      if (st < 8) SSDE_GT(5 + st * 4);
      __syncthreads();
      if (st < 8) SSDE_GT(6 + st * 4);
      if (st + 1 >= nst) break;
      load_rest(min(st + 2, last), R);
      __builtin_amdgcn_sched_barrier(0);
      load_a(min(st + 3, last), A1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_stage(buf1);
      if (st + 1 < 8) SSDE_GT(4 + (st + 1) * 4);
      if (st + 2 < nst) store_stage(st + 2, A0, R, buf0);
      if (st + 1 < 8) SSDE_GT(5 + (st + 1) * 4);
      __syncthreads();
      if (st + 1 < 8) SSDE_GT(6 + (st + 1) * 4);
    }
  }
  SSDE_GT(40);
  SSDE_GW(1);

  // epilogue: 64-row halves of the tile (kBM = 64: one) through the shared coalesced store
  constexpr int LDT = kBN + 4;
  SsdeEpi e{p.bias, p.chan_add, p.chan_add_ld, p.resid, p.resid_post, p.scale, p.dst, p.Cout, p.gn_part};
#pragma unroll
  for (int half = 0; half < kBM / 64; ++half) {
    if (kBM == 64 || (wave >> 1) == half) {
#pragma unroll
      for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = (kBM == 64 ? wm0 : 0) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            smem[m * LDT + wn0 + b * 32 + li] = acc[a][b][r];
          }
    }
    __syncthreads();
    ssde_store_tile<64, kBN, kThreads, SSDE_GEMM_BATCH>(smem, LDT, n0, e, [&](int row, size_t& pix, int& img) {
      const int m = m0 + half * 64 + row;
      if (m >= p.M) return false;
      pix = (size_t)m;
      img = m / p.HW;
      return true;
    }, p.gn_part ? (p.HW >= 64 ? (m0 >> 6) + half : (m0 + half * 64) / p.HW) : -1, p.HW >= 64 ? 30 : p.lHW, p.gn_entries);
    SSDE_GT(41 + half * 2);
    __syncthreads();
    SSDE_GT(42 + half * 2);
  }
  SSDE_GW(2);
}

// ---- persistent, software-pipelined form (SSDE_GEMM_PIPE) -----------------------------------------------------------
// The kernels above run compute-then-write rounds: every workgroup of a launch reaches its epilogue at the same time, the
// epilogue runs at HBM speed (output + residual, profiles/r2_gemm_trace.txt: 36 k of a workgroup's 201 k cycles) and the
// matrix pipe idles meanwhile -- 256 -> 256 @16x16 at batch 256 is 55 us of MFMA and 21 us of epilogue traffic and takes 99 us.
// Here a workgroup owns SEVERAL 128 x 128 tiles (grid = 2 workgroups per CU) and keeps TWO accumulator sets in registers:
// while the K loop of tile i + 1 fills one, the finished tile i drains from the other, one 8-row x 64-column unit per wave at
// a time, spread evenly over the stages of that loop: 8 accumulator registers -> the wave's own 2 KB LDS slab (no workgroup
// barrier: SSDE_WAVE_SYNC) -> two float4 per lane along the channels -> bias / temb addend / residual / scale -> 16-byte
// stores.  The residual rows of the next unit are fetched while the current one is stored.  GroupNorm partials of the
// stored tensor: a lane always holds the same channel quad, so it accumulates pivoted sums over the tile and the four lanes of
// a quad merge at the tile's end (the layout ssde_gn_finalize reads; the three wave slots this kernel does not use hold
// zero counts).  Both matrix modes: exact-fp32 MFMA, or the 3-way bf16 split (kX6).
constexpr int kSlabLd = 68;                          // floats per row of a wave's 8 x 64 slab (bank spread of the b32 writes)
constexpr int kSlabFloats = 8 * kSlabLd;

template <bool kGn, bool kX6>
__global__ __launch_bounds__(kThreads, 2) void gemm1x1_pipe_kernel(const GemmParams p) {
  SSDE_LDS(smem);
  constexpr int kStageB = kX6 ? X6<128>::kStageBytes : kStage * 4;
  constexpr int KB = 16;                             // channels per stage, both modes
  char* lds = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  float* slab = reinterpret_cast<float*>(lds + 2 * kStageB) + wave * kSlabFloats;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const ssde_src& s = p.src;
  SsdePro pro = ssde_pro_decode(s);
  pro.gn = kGn;
  const int Ctot = s.c0 + s.c1;
  const int cpg = kGn ? Ctot / s.gn_groups : 1;
  const int nst = (p.K + KB - 1) / KB;
  const int ncin8 = (p.K + 7) >> 3;
  const int f = tid & 3, r0 = tid >> 2;
  constexpr int NI = 2, RS = 64;
  // drain plan of a lane: float4 (row8, c4) and (row8 + 4, c4) of every unit
  const int d_row = lane >> 4, d_c4 = lane & 15;

  struct Pending {                                   // the finished tile whose accumulators are being stored
    int m0, n0, done;                                // done: units stored so far (8 per tile)
    bool valid;
    float4 bias4;
    float4 rnext[2];                                 // residual rows of unit `done`, fetched while unit done - 1 was stored
    float st_p, st_s1, st_s2, st_n;
  };
  Pending pd;
  pd.valid = false; pd.done = 8; pd.m0 = pd.n0 = 0;
  pd.bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  pd.rnext[0] = pd.rnext[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  pd.st_p = pd.st_s1 = pd.st_s2 = pd.st_n = 0.f;

  // (the row of a unit goes through an opaque register: left alone, hipcc hoists the 16 rows x 3 pointers of all eight units
  //  out of the K loop and spills)
  auto unit_row = [&](int u, int k) {
    int r = wm0 + (u >> 2) * 32 + (u & 3) * 8 + d_row;
    SSDE_OPAQUE_VGPR(r);
    return r + 4 * k;
  };
  auto fetch_resid = [&](int u) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int m = pd.m0 + unit_row(u, k), j = pd.n0 + wn0 + d_c4 * 4;
      pd.rnext[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.resid && u < 8 && m < p.M && j < p.Cout) pd.rnext[k] = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.Cout + j);
    }
  };
  // one unit (a, q) of the pending tile: compile-time register indices
  auto drain_unit = [&](auto A_, auto Q_, f32x16 (&acc)[2][2]) __attribute__((always_inline)) {
    constexpr int a = decltype(A_)::value, q = decltype(Q_)::value, u = a * 4 + q;
    const float4 rr[2] = {pd.rnext[0], pd.rnext[1]};
    fetch_resid(u + 1);
    SSDE_WAVE_SYNC();                                // the reads of the previous unit are done (emulator; in-order on the GPU)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(r + 4 * lh) * kSlabLd + b * 32 + li] = acc[a][b][4 * q + r];
    SSDE_WAVE_SYNC();
    const int j = pd.n0 + wn0 + d_c4 * 4;
    // the 8 rows of a unit lie in one image (the launcher takes this kernel with a per-image addend only for H*W % 8 == 0)
    const int img = p.chan_add ? (pd.m0 + unit_row(u, 0) - d_row) / p.HW : 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float4 t = *reinterpret_cast<const float4*>(slab + (d_row + 4 * k) * kSlabLd + d_c4 * 4);
      const int m = pd.m0 + unit_row(u, k);
      if (m < p.M && j < p.Cout) {
        float4 ca = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.chan_add) ca = *reinterpret_cast<const float4*>(p.chan_add + (size_t)img * p.chan_add_ld + j);
        float v[4] = {t.x + pd.bias4.x + ca.x, t.y + pd.bias4.y + ca.y, t.z + pd.bias4.z + ca.z, t.w + pd.bias4.w + ca.w};
        if (!p.resid_post) { v[0] += rr[k].x; v[1] += rr[k].y; v[2] += rr[k].z; v[3] += rr[k].w; }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= p.scale;
        if (p.resid_post) { v[0] += rr[k].x; v[1] += rr[k].y; v[2] += rr[k].z; v[3] += rr[k].w; }
        *reinterpret_cast<float4*>(p.dst + (size_t)m * p.Cout + j) = make_float4(v[0], v[1], v[2], v[3]);
        if (p.gn_part) {
          if (pd.st_n == 0.f) pd.st_p = v[0];
          const float d0 = v[0] - pd.st_p, d1 = v[1] - pd.st_p, d2 = v[2] - pd.st_p, d3 = v[3] - pd.st_p;
          pd.st_s1 += (d0 + d1) + (d2 + d3);
          pd.st_s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
          pd.st_n += 4.f;
        }
      }
    }
  };
  auto drain_next = [&](f32x16 (&acc)[2][2]) __attribute__((always_inline)) {
    using std::integral_constant;
    switch (pd.done) {
      case 0: drain_unit(integral_constant<int, 0>{}, integral_constant<int, 0>{}, acc); break;
      case 1: drain_unit(integral_constant<int, 0>{}, integral_constant<int, 1>{}, acc); break;
      case 2: drain_unit(integral_constant<int, 0>{}, integral_constant<int, 2>{}, acc); break;
      case 3: drain_unit(integral_constant<int, 0>{}, integral_constant<int, 3>{}, acc); break;
      case 4: drain_unit(integral_constant<int, 1>{}, integral_constant<int, 0>{}, acc); break;
      case 5: drain_unit(integral_constant<int, 1>{}, integral_constant<int, 1>{}, acc); break;
      case 6: drain_unit(integral_constant<int, 1>{}, integral_constant<int, 2>{}, acc); break;
      default: drain_unit(integral_constant<int, 1>{}, integral_constant<int, 3>{}, acc); break;
    }
    ++pd.done;
  };
  // GroupNorm partials of the finished tile: the wave's 64 rows x 64 columns are ONE 64-row slice of one image (the launcher
  // takes this kernel only for H*W % 64 == 0); lanes l, l ^ 16, l ^ 32 hold the same channel quad
  auto finish_tile = [&](f32x16 (&acc)[2][2]) __attribute__((always_inline)) {
    while (pd.done < 8) drain_next(acc);
    if (p.gn_part) {
      float n = pd.st_n, m = 0.f, M2 = 0.f;
      if (n > 0.f) { const float rn = __builtin_amdgcn_rcpf(n); m = pd.st_p + pd.st_s1 * rn; M2 = pd.st_s2 - pd.st_s1 * pd.st_s1 * rn; M2 = M2 < 0.f ? 0.f : M2; }
      for (int o = 16; o < 64; o <<= 1) {
        const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(m, o, 64), Mb = __shfl_xor(M2, o, 64);
        if (lane & o) { float tn = nb, tm = mb, tM = Mb; ssde_stat_merge(tn, tm, tM, n, m, M2); n = tn; m = tm; M2 = tM; }
        else ssde_stat_merge(n, m, M2, nb, mb, Mb);
      }
      const int entry = (pd.m0 + wm0) >> 6, j = pd.n0 + wn0 + 4 * lane;
      if (lane < 16 && j < p.Cout && entry < p.gn_entries) {
#pragma unroll
        for (int slot = 0; slot < 4; ++slot) {
          float* o = p.gn_part + (((size_t)entry * 4 + slot) * (p.Cout >> 2) + (j >> 2)) * 3;
          o[0] = slot == 0 ? m : 0.f; o[1] = slot == 0 ? M2 : 0.f; o[2] = slot == 0 ? n : 0.f;
        }
      }
    }
    pd.valid = false;
  };

  // ---- one tile: K loop into `acc`, the pending tile drains from `old` ----
  auto run_tile = [&](int vb, f32x16 (&acc)[2][2], f32x16 (&old)[2][2]) __attribute__((always_inline)) {
    const int xcd = vb & 7, lin = vb >> 3;
    const int nt = lin % p.n_tiles, mt = (lin / p.n_tiles) * 8 + xcd;
    const int m0 = mt * BM, n0 = nt * BN;
    int arow[NI], aimg[NI];
    bool aok[NI], bok[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int m = m0 + r0 + i * RS;
      aok[i] = m < p.M;
      arow[i] = aok[i] ? m : 0;
      aimg[i] = arow[i] / p.HW;
      bok[i] = (n0 + r0 + i * RS) < p.CoutPad;
    }
    const float* bptr = p.wpk + ((size_t)n0 + r0) * 8 + (f & 1) * 4;
    float4 av[NI], bv[NI];
    float mu[NI] = {0.f, 0.f}, rs[NI] = {1.f, 1.f};
    float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_stage = [&](int st) __attribute__((always_inline)) {
      const int c_base = st * KB;
      const bool second = c_base >= s.c0;
      const float* base = second ? s.p1 : s.p0;
      const int C = second ? s.c1 : s.c0;
      const int cthr = (second ? c_base - s.c0 : c_base) + f * 4;
      const int c_cur = c_base + f * 4;
      const float* ap = base + (c_cur < p.K ? cthr : 0);
#pragma unroll
      for (int i = 0; i < NI; ++i) av[i] = *reinterpret_cast<const float4*>(ap + (size_t)arow[i] * C);
      const int cin8 = min(st * 2 + (f >> 1), ncin8 - 1);
      const float* bp = bptr + (size_t)cin8 * p.CoutPad * 8;
#pragma unroll
      for (int i = 0; i < NI; ++i) bv[i] = *reinterpret_cast<const float4*>(bp + (bok[i] ? i * RS * 8 : 0));
      if (kGn) {
        const int cg = c_cur < p.K ? c_cur : 0;
        gam = *reinterpret_cast<const float4*>(s.gn_gamma + cg);
        bet = *reinterpret_cast<const float4*>(s.gn_beta + cg);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int gi = aimg[i] * s.gn_groups + cg / cpg;
          mu[i] = s.gn_mean[gi];
          rs[i] = s.gn_rstd[gi];
        }
      }
    };
    auto store_stage = [&](int st, char* buf) __attribute__((always_inline)) {
      const int cin8 = st * 2 + (f >> 1), c_cur = st * KB + f * 4;
      const bool k_ok = c_cur < p.K;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (aok[i] && k_ok)
          v = ssde_pro_apply(av[i], mu[i], rs[i], gam, bet, (uint32_t)arow[i] * (uint32_t)Ctot + (uint32_t)c_cur, pro);
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bok[i] && cin8 < ncin8) w = bv[i];
        if constexpr (kX6) {
          using X = X6<128>;
          uint2 q0, q1, q2;
          char* d = buf + (r0 + i * RS) * (KB * 2) + f * 8;
          ssde_split3(v, q0, q1, q2);
          *reinterpret_cast<uint2*>(d) = q0;
          *reinterpret_cast<uint2*>(d + X::kPlaneA) = q1;
          *reinterpret_cast<uint2*>(d + 2 * X::kPlaneA) = q2;
          ssde_split3(w, q0, q1, q2);
          *reinterpret_cast<uint2*>(d + 3 * X::kPlaneA) = q0;
          *reinterpret_cast<uint2*>(d + 3 * X::kPlaneA + X::kPlaneB) = q1;
          *reinterpret_cast<uint2*>(d + 3 * X::kPlaneA + 2 * X::kPlaneB) = q2;
        } else {
          float* fb = reinterpret_cast<float*>(buf);
          float* d = fb + (r0 + i * RS) * LDK + f * 4;
          *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
          *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
          float* e = fb + (BM + r0 + i * RS) * LDK + f * 4;
          *reinterpret_cast<float2*>(e) = make_float2(w.x, w.y);
          *reinterpret_cast<float2*>(e + 2) = make_float2(w.z, w.w);
        }
      }
    };
    auto mfma_stage = [&](const char* cur) __attribute__((always_inline)) {
      if constexpr (kX6) {
        using X = X6<128>;
        const int aoff = (wm0 + li) * (KB * 2) + lh * 16, boff = 3 * X::kPlaneA + (wn0 + li) * (KB * 2) + lh * 16;
        ssde_u32x4 A[2][3], B[2][3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            A[a][q] = *reinterpret_cast<const ssde_u32x4*>(cur + aoff + q * X::kPlaneA + a * 32 * (KB * 2));
            B[a][q] = *reinterpret_cast<const ssde_u32x4*>(cur + boff + q * X::kPlaneB + a * 32 * (KB * 2));
          }
        constexpr int TI[6] = {0, 2, 1, 0, 1, 0}, TJ[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssde_bf16x8, A[a][TI[t]]),
                                                                  __builtin_bit_cast(ssde_bf16x8, B[b][TJ[t]]), acc[a][b], 0, 0, 0);
      } else {
        const float* cf = reinterpret_cast<const float*>(cur);
        int aoff[2], boff[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { aoff[a] = (wm0 + a * 32 + li) * LDK + 2 * lh; boff[a] = (BM + wn0 + a * 32 + li) * LDK + 2 * lh; }
        float2 af[2][2], bf[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { af[0][a] = *reinterpret_cast<const float2*>(cf + aoff[a]); bf[0][a] = *reinterpret_cast<const float2*>(cf + boff[a]); }
#pragma unroll
        for (int t = 0; t < KB / 4; ++t) {
          const int c = t & 1;
          if (t + 1 < KB / 4) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              af[c ^ 1][a] = *reinterpret_cast<const float2*>(cf + aoff[a] + (t + 1) * 4);
              bf[c ^ 1][a] = *reinterpret_cast<const float2*>(cf + boff[a] + (t + 1) * 4);
            }
          }
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][a].x, bf[c][b].x, acc[a][b], 0, 0, 0);
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][a].y, bf[c][b].y, acc[a][b], 0, 0, 0);
        }
      }
    };
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    char* buf0 = lds;
    char* buf1 = lds + kStageB;
    load_stage(0);
    store_stage(0, buf0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
      const bool has_next = st + 1 < nst;
      if (has_next) load_stage(st + 1);
      mfma_stage((st & 1) ? buf1 : buf0);
      // the pending tile's units, spread evenly over this loop (behind the MFMAs in program order: they run beside them)
      if (pd.valid) {
        const int target = ((st + 1) * 8 + nst - 1) / nst;
        while (pd.done < target && pd.done < 8) drain_next(old);
      }
      if (has_next) store_stage(st + 1, (st & 1) ? buf0 : buf1);
      __syncthreads();
    }
    if (pd.valid) finish_tile(old);
    // this tile becomes the pending one
    pd.valid = true; pd.done = 0; pd.m0 = m0; pd.n0 = n0;
    pd.st_p = pd.st_s1 = pd.st_s2 = pd.st_n = 0.f;
    pd.bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int j = n0 + wn0 + d_c4 * 4;
    if (p.bias && j < p.Cout) pd.bias4 = *reinterpret_cast<const float4*>(p.bias + j);
    fetch_resid(0);
  };

  // `acc` accumulates, `old` drains; a finished tile moves from one to the other with 64 register copies (nothing next to a
  // tile's ~60 k cycles, and the roles stay compile-time: alternating them behind a run-time flag made hipcc spill hundreds
  // of registers)
  f32x16 acc[2][2], old[2][2];
  const int total_vb = ((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
  for (int vb = blockIdx.x; vb < total_vb; vb += gridDim.x) {
    if (((vb >> 3) / p.n_tiles) * 8 + (vb & 7) >= p.m_tiles) continue;      // grid padding (uniform over the workgroup)
    run_tile(vb, acc, old);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) old[a][b] = acc[a][b];
  }
  if (pd.valid) finish_tile(old);
}

}  // namespace

// Used by ssde_conv2d for 1x1-only launches it judges large enough (conv_mfma.hip); shapes it does not take stay on the
// general kernel.
bool ssde_conv1x1_wants(const ssde_conv_args* a) {
  if (a->ksize != 0 || a->aux.p0 == nullptr || a->tile != SSDE_TILE_AUTO) return false;
  const long long M = (long long)a->n * a->h_out * a->w_out;
  return a->c_out >= 96 && M >= 128 && M < (1ll << 31);
}

int ssde_conv1x1_launch(const ssde_conv_args* a, void* stream, int* lds_out) {
  const ssde_src& s = a->aux;
  SSDE_REQUIRE(s.c0 % 4 == 0 && s.c1 % 4 == 0, "conv1x1: channels must be multiples of 4 (got %d,%d)", s.c0, s.c1);
  SSDE_REQUIRE(s.c1 == 0 || (s.p1 != nullptr && s.c0 % 32 == 0), "conv1x1: concat boundary must be a multiple of 32");
  const bool gn = s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU;
  if (gn) {
    SSDE_REQUIRE(s.gn_groups > 0 && (s.c0 + s.c1) % s.gn_groups == 0 && ((s.c0 + s.c1) / s.gn_groups) % 4 == 0,
                 "conv1x1: GroupNorm needs channels-per-group %% 4 == 0");
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "conv1x1: GroupNorm pointers missing");
  }
  SSDE_REQUIRE(a->w_aux && a->dst, "conv1x1: null weights / destination");
  GemmParams p;
  p.src = s; p.wpk = a->w_aux;
  p.M = a->n * a->h_out * a->w_out; p.HW = a->h_out * a->w_out; p.K = s.c0 + s.c1;
  p.Cout = a->c_out; p.CoutPad = ssde_cdiv(a->c_out, 64) * 64;
  p.m_tiles = ssde_cdiv(p.M, BM); p.n_tiles = ssde_cdiv(a->c_out, BN);
  p.bias = a->bias; p.chan_add = a->chan_add; p.chan_add_ld = a->chan_add_ld;
  p.resid = a->resid; p.resid_post = a->resid_post; p.scale = a->out_scale; p.dst = a->dst;
  p.gn_part = a->gn_part;
  const bool gn_ok = a->c_out % 4 == 0 && (p.HW % 64 == 0 || (p.HW >= 8 && p.HW < 64 && 64 % p.HW == 0));
  SSDE_REQUIRE(!a->gn_part || gn_ok, "conv1x1: GroupNorm partials need H*W %% 64 == 0 or a power of two in 8..32");
  p.lHW = ssde_ilog2(p.HW);
  p.gn_entries = p.HW >= 64 ? a->n * (p.HW / 64) : a->n;
  const bool x6 = (a->flags & SSDE_CONVF_BF16X6) != 0;
  // the persistent, software-pipelined form: when workgroups get more than one tile each (otherwise there is nothing to overlap
  // and the plain kernels' 3-4 workgroups per CU cover each other better than its 2)
  {
    const int total = ssde_cdiv(p.m_tiles, 8) * 8 * p.n_tiles;
    const int wgs = (2 * ssde_num_cus() + 7) / 8 * 8;
    const bool pipe_ok = !(a->flags & SSDE_CONVF_NO_GEMM_PIPE) && a->c_out % 4 == 0 && (!a->gn_part || p.HW % 64 == 0) && (!a->chan_add || p.HW % 8 == 0);
    // Measured (profiles/r4_gemm_pipe_ab.txt, batch 256): with exact-fp32 MFMAs the pipelined form LOSES 0-8 % -- the drain's
    // VALU and the fp32 MFMAs share a datapath, and two workgroups per CU hide less than four -- so it is not taken there; with
    // the bf16 split it wins 9-10 % from 512 input or output channels up and loses up to 10 % below (the loop is too short for
    // the drain).  SSDE_CONVF_NO_GEMM_PIPE = never, SSDE_CONVF_GEMM_PIPE = always (tests), neither = this rule
    // (round 6: from 512 output channels up the 128 x 256 tile of the plain kernel is ahead of it -- 256 -> 768 @16x16 with a
    //  GroupNorm prologue 0.21-0.23 -> 0.17-0.18 ms, profiles/r6_gemm_wide_tile_ab.txt: the pixel rows are normalised and split
    //  three times instead of six -- and level with it at 512 -> 256)
    const bool wide_first = a->c_out >= 512 && a->c_out % 256 == 0 && !(a->flags & (SSDE_CONVF_X6_NO_WIDE | SSDE_CONVF_X6_BM64 | SSDE_CONVF_X6_PF2)) &&
                            (long long)ssde_cdiv(p.M, 128) * (a->c_out / 256) >= 2ll * ssde_num_cus();
    // (the REAL tiles decide, not the grid padded to the 8 XCDs, and 512 output channels alone no longer do: where the wide tile
    //  does not fill the device the plain kernel is ahead -- the temb projections [256, 512] x [512, 9984], 2 x 78 tiles, took this
    //  kernel through the padded count and 0.088 instead of 0.046 ms; 256 -> 768 @8x8 0.062 against 0.052)
    const bool pays = x6 && p.K >= 512 && (long long)p.m_tiles * p.n_tiles > wgs && !wide_first;
    // (round 5: the fp32 instantiation of the pipelined kernel is no longer built -- it lost everywhere)
    if (pipe_ok && x6 && (pays || (a->flags & SSDE_CONVF_GEMM_PIPE))) {
      const int lds = 2 * (x6 ? X6<128>::kStageBytes : kStage * 4) + 4 * kSlabFloats * 4;
      if (lds_out) { *lds_out = lds; return SSDE_OK; }
      const dim3 grid(total < wgs ? total : wgs);
      auto go = [&](auto kfn, std::atomic<bool>& attr_set) {
        if (!attr_set) {
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
          attr_set = true;
        }
        hipLaunchKernelGGL(kfn, grid, dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
        return true;
      };
      static std::atomic<bool> pset[2];
      const bool ok = gn ? go(gemm1x1_pipe_kernel<true, true>, pset[0]) : go(gemm1x1_pipe_kernel<false, true>, pset[1]);
      SSDE_REQUIRE(ok, "conv1x1: hipFuncSetAttribute failed");
      SSDE_LAUNCH_CHECK();
      return SSDE_OK;
    }
  }
  // shape of the split kernel (see there): rows per workgroup and load-ahead depth; A/B-selectable per call
  int xbm = 128, xpf = 1, xbn = BN;
  if (x6) {
    if (a->flags & SSDE_CONVF_X6_BM64) xbm = 64;
    if (a->flags & SSDE_CONVF_X6_PF2) xpf = 2;
    // launches whose 128-row tiles would not give every CU a workgroup (the 8x8 and 4x4 maps with 256 output channels, the temb
    // projections [batch, 512] x [512, 9984]): 64 rows -- twice the workgroups, four per CU.  512 -> 256 @8x8 at batch 256:
    // 0.044-0.047 -> 0.037-0.038 ms (profiles/r6_gemm_wide_tile_ab.txt)
    if ((long long)ssde_cdiv(p.M, 128) * ssde_cdiv(a->c_out, BN) <= ssde_num_cus() && !(a->flags & SSDE_CONVF_X6_WIDE)) xbm = 64;
    // the wide tile (128 x 256): where the output channels fill it and its workgroups (two per CU) fill the device at least once
    const bool wide_fits = a->c_out % 256 == 0 && xbm == 128 && xpf == 1;
    const bool wide_pays = (long long)ssde_cdiv(p.M, 128) * (a->c_out / 256) >= 2ll * ssde_num_cus();
    if (!(a->flags & SSDE_CONVF_X6_NO_WIDE) && wide_fits && (wide_pays || (a->flags & SSDE_CONVF_X6_WIDE))) xbn = 256;
    p.m_tiles = ssde_cdiv(p.M, xbm);
    p.n_tiles = ssde_cdiv(a->c_out, xbn);
  }
  const int lds_ops = x6 ? 2 * (xbn == 256 ? X6<128, 256>::kStageBytes : xbm == 64 ? X6<64>::kStageBytes : X6<128>::kStageBytes) : 2 * kStage * 4;
  const int lds_epi = 64 * (xbn + 4) * 4;
  const int lds = lds_ops > lds_epi ? lds_ops : lds_epi;
  if (lds_out) { *lds_out = lds; return SSDE_OK; }
  const dim3 grid(ssde_cdiv(p.m_tiles, 8) * 8 * p.n_tiles);
  auto go = [&](auto kfn, std::atomic<bool>& attr_set) {
    if (!attr_set) {                            // once per instantiation, before any stream capture
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return false;
      attr_set = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
    return true;
  };
  static std::atomic<bool> set[12];
  bool ok;
  if (!x6) ok = gn ? go(gemm1x1_kernel<true>, set[0]) : go(gemm1x1_kernel<false>, set[1]);
  else if (xbn == 256) ok = gn ? go(gemm1x1_bf16x6_kernel<true, 128, 1, 256>, set[10]) : go(gemm1x1_bf16x6_kernel<false, 128, 1, 256>, set[11]);
  else if (xbm == 128 && xpf == 1) ok = gn ? go(gemm1x1_bf16x6_kernel<true, 128, 1>, set[2]) : go(gemm1x1_bf16x6_kernel<false, 128, 1>, set[3]);
  else if (xbm == 128) ok = gn ? go(gemm1x1_bf16x6_kernel<true, 128, 2>, set[4]) : go(gemm1x1_bf16x6_kernel<false, 128, 2>, set[5]);
  else if (xpf == 1) ok = gn ? go(gemm1x1_bf16x6_kernel<true, 64, 1>, set[6]) : go(gemm1x1_bf16x6_kernel<false, 64, 1>, set[7]);
  else ok = gn ? go(gemm1x1_bf16x6_kernel<true, 64, 2>, set[8]) : go(gemm1x1_bf16x6_kernel<false, 64, 2>, set[9]);
  SSDE_REQUIRE(ok, "conv1x1: hipFuncSetAttribute failed");
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
