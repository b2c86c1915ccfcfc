// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions by Winograd F(4x4, 3x3) on the exact-fp32 matrix pipe.
//
// Forward (conv_wino4.hip):  Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A   per 4x4 output tile, 6x6 transform positions.
// Differentiating,
//
//   dL/dg[co, ci] = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G
//
// so per transform position p (36 of them) the gradient is a GEMM  dU_p[co, ci] = sum_t Z_p[t, co] * V_p[t, ci]  whose
// reduction runs over the 4x4 tiles of the whole batch (t = 0 .. N H W / 16): 36 multiply-adds per (tile, co, ci) for 16
// output pixels, against 144 of the direct form (wgrad.hip) and 64 of F(2x2,3x3) (wgrad_wino.hip) -- the matrix pipe does
// 1.78x less work than the kernel this replaces on the layers that dominate the training step (the weight gradients of
// ddpm_conv3x3 in loss.backward(), /root/reference losses.py:196 through models/layers.py:118-124).
// Rounding: tools/experiments/wgrad_wino43_error_budget.py -- relative L2 error 2.6e-6 .. 3.0e-6 against fp64 at the
// reduction lengths of the training step (direct fp32: 2.8e-7 .. 3.7e-7, F(2x2,3x3): 4.5e-7 .. 5.7e-7), against the 2e-4
// the gradient parity tests allow.
//
// Unlike the forward convolution, BOTH operands of this GEMM have to be transformed, and the reduction dimension (tiles)
// is long while the result (36 x Cout x Cin) is small.  A fused kernel in the style of conv_wino4.hip -- transforms of a
// 2 x 2-tile chunk riding between the MFMAs of a 32 co x 64 ci block -- was built first and measured 0-11 % over the
// F(2x2,3x3) kernel (profiles/r3_wgrad_wino4_fused_v1_ab.txt, git history of this file): three times conv_wino4's transform
// work per MFMA, every stage bound by LDS traffic and VALU that the matrix pipe cannot overlap, and the input transform
// (with its SiLU) redone for every 32-co block.  This file therefore splits the work the classic way, which suits a
// long-K, small-output GEMM:
//   wino4_xform_v_kernel    V[p][t][ci] = B^T pro(x) B      one thread = one (tile, channel), lanes over channels: 36
//                                                            coalesced loads (GroupNorm / SiLU / dropout applied once per
//                                                            element and tile, not once per cout block), the 6x6 transform in
//                                                            registers, 36 coalesced stores.  HBM-bound: reads the tensor,
//                                                            writes 2.25x of it
//   wino4_xform_z_kernel    Z[p][t][co] = A dY A^T          the same with 16 loads
//   wgrad4_gemm_kernel      dU_p = Z_p^T V_p                 batched over the 36 positions, 128 x 128 blocks of (co, ci), 4
//                                                            waves x (64 x 64) on v_mfma_f32_32x32x2_f32, K = tiles in
//                                                            stages of 16 through double-buffered LDS (both operands are
//                                                            K-major rows of channels: no transposition anywhere), split-K
//                                                            over workgroups into slabs
//   wgrad4_sum_splits_kernel, wgrad4_reduce_kernel    dw += scale G^T (sum_splits dU) G  fixed order (deterministic)
// The transformed operands live in the launch's scratch (2.25 x (input + output gradient) of the layer).
#include "ssde_common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// -DSSDE_WG4_TRACE (tools/wgrad4_trace.py, a variant library only): s_memtime stamps of wave 0 of the first GEMM workgroup
#ifdef SSDE_WG4_TRACE
__device__ unsigned long long* g_wg4_trace;
extern "C" int ssde_debug_wg4_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wg4_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -5;
}
#define SSDE_WT(slot)                                                                                     \
  do {                                                                                                    \
    if (tr_on) g_wg4_trace[slot] = __builtin_amdgcn_s_memtime();                                          \
  } while (0)
#else
#define SSDE_WT(slot) do { } while (0)
#endif

namespace {

constexpr int kPos = 36;
constexpr int BM = 128, BN = 128, BK = 16, LDP = BM + 4;      // LDS row pitch: rows 4 banks apart
constexpr int kGemmThreads = 256;
constexpr int kStage = 2 * BK * LDP;                           // floats of one LDS stage (Z rows then V rows)

struct W4Params {
  ssde_src src;
  const float* g;
  int g_ld, g_off;
  int N, H, W, Cout, Ctot;
  int tx, ty, T;           // tiles per row / column of an image, tiles in total
  int co_tiles, ci_tiles, splits, k_per_split;
  int xcd_order;           // 1: the blocks of one (position, split) share an XCD (SSDE_WGRAD4_XCD=0: plain order, A/B only)
  float scale;
  float* dw;
  float* V; float* Z; float* slabs;
  int sk_on, sk_per, sk_S, sk_groups;   // stream-K (default): workgroup group g owns stages [g sk_per, (g+1) sk_per) of the flattened
                           // (position, K stage) space, sk_S stages per position (see wgrad4_gemm_kernel); splits = the most segments a position has
  int v_quads;             // 1: V is the forward launch's by-product, [pos][Ctot / 4][t][4] (conv_wino4.hip kEmitV: a workgroup's 32
                           // tiles are one 512-byte run per position and stage); 0: wino4_xform_v_kernel's [pos][t][Ctot]
};

// ---- transforms: thread = (tile t, VEC channels), consecutive threads = consecutive channels: 36 (16) coalesced loads and 36
// stores in flight per thread.  VEC = 2 (8-byte accesses, 112-126 registers, four waves per SIMD) is 3-7 % faster per layer
// than VEC = 1; four channels per thread (16-byte accesses, 256 registers, two waves per SIMD) measured 5-12 % SLOWER end to
// end: profiles/r3_wgrad_wino4_ab.txt. ----
// B^T rows [4,0,-5,0,1,0] [0,-4,-4,1,1,0] [0,4,-4,-1,1,0] [0,-2,-1,2,1,0] [0,2,-1,-2,1,0] [0,4,0,-5,0,1]
__device__ __forceinline__ void bt6(const float (&d)[6], float (&o)[6]) {
  const float t1 = d[4] - 4.f * d[2], t2 = d[3] - 4.f * d[1], t3 = d[4] - d[2], t4 = d[3] - d[1];
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = t1 + t2;
  o[2] = t1 - t2;
  o[3] = t3 + 2.f * t4;
  o[4] = t3 - 2.f * t4;
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// A rows [1,0,0,0] [1,1,1,1] [1,-1,1,-1] [1,2,4,8] [1,-2,4,-8] [0,0,0,1]
__device__ __forceinline__ void at6(const float (&d)[4], float (&o)[6]) {
  const float s = d[0] + d[2], t = d[1] + d[3], u = d[0] + 4.f * d[2], v = 2.f * d[1] + 8.f * d[3];
  o[0] = d[0];
  o[1] = s + t;
  o[2] = s - t;
  o[3] = u + v;
  o[4] = u - v;
  o[5] = d[3];
}

// VEC = channels per thread (1, or 2 with 8-byte accesses: SSDE_WGRAD4_XVEC, A/B in profiles/r3_wgrad_wino4_ab.txt)
template <int VEC> struct XfVec;
template <> struct XfVec<1> { typedef float type; };
template <> struct XfVec<2> { typedef float2 type; };
template <int VEC> __device__ __forceinline__ float xf_get(const typename XfVec<VEC>::type& v, int e);
template <> __device__ __forceinline__ float xf_get<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float xf_get<2>(const float2& v, int e) { return e ? v.y : v.x; }
template <int VEC> __device__ __forceinline__ typename XfVec<VEC>::type xf_make(const float (&o)[VEC]);
template <> __device__ __forceinline__ float xf_make<1>(const float (&o)[1]) { return o[0]; }
template <> __device__ __forceinline__ float2 xf_make<2>(const float (&o)[2]) { return make_float2(o[0], o[1]); }

template <bool kGn, int VEC>
__global__ __launch_bounds__(256) void wino4_xform_v_kernel(const W4Params p) {
  typedef typename XfVec<VEC>::type vec_t;
  const ssde_src& s = p.src;
  SsdePro pro = ssde_pro_decode(s);
  pro.gn = kGn;
  const int cpg = kGn ? p.Ctot / s.gn_groups : 1;
  const int CV = p.Ctot / VEC;
  const long long total = (long long)p.T * CV;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % CV) * VEC, t = (int)(idx / CV);
    const int img = t / (p.tx * p.ty), r = t - img * (p.tx * p.ty);
    const int ty = r / p.tx, tx = r - ty * p.tx;
    const bool second = c >= s.c0;                    // (c0 % 4 == 0: a pair never straddles the sources)
    const float* base = second ? s.p1 + (c - s.c0) : s.p0 + c;
    const int C = second ? s.c1 : s.c0;
    float mu = 0.f, rs = 1.f, ga[VEC], be[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { ga[e] = 1.f; be[e] = 0.f; }
    if (kGn) {
      mu = s.gn_mean[img * s.gn_groups + c / cpg];    // (cpg % 4 == 0: a pair lies in one group)
      rs = s.gn_rstd[img * s.gn_groups + c / cpg];
#pragma unroll
      for (int e = 0; e < VEC; ++e) { ga[e] = s.gn_gamma[c + e]; be[e] = s.gn_beta[c + e]; }
    }
    float v[VEC][6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const int iy = ty * 4 - 1 + a, ix = tx * 4 - 1 + b;
        const bool inb = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int pix = (img * p.H + (inb ? iy : 0)) * p.W + (inb ? ix : 0);
        const vec_t xv = *reinterpret_cast<const vec_t*>(base + (size_t)pix * C);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          float x = xf_get<VEC>(xv, e);
          if (kGn) x = (x - mu) * rs * ga[e] + be[e];
          if (pro.silu) x = ssde_silu(x);
          if (__builtin_expect(pro.drop, 0)) x *= ssde_keep((uint32_t)pix * (uint32_t)p.Ctot + (uint32_t)(c + e), pro);
          v[e][a][b] = inb ? x : 0.f;               // the convolution pads the ACTIVATED tensor with zeros
        }
      }
    // columns (over the rows a of every column b), then rows
#pragma unroll
    for (int e = 0; e < VEC; ++e)
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        float d[6], o[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) d[a] = v[e][a][b];
        bt6(d, o);
#pragma unroll
        for (int a = 0; a < 6; ++a) v[e][a][b] = o[a];
      }
    float* dst = p.V + (size_t)t * p.Ctot + c;
    const size_t plane = (size_t)p.T * p.Ctot;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      float o[VEC][6];
#pragma unroll
      for (int e = 0; e < VEC; ++e) bt6(v[e][a], o[e]);
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        float w[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) w[e] = o[e][b];
        *reinterpret_cast<vec_t*>(dst + (size_t)(a * 6 + b) * plane) = xf_make<VEC>(w);
      }
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void wino4_xform_z_kernel(const W4Params p) {
  typedef typename XfVec<VEC>::type vec_t;
  const int CV = p.Cout / VEC;
  const long long total = (long long)p.T * CV;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % CV) * VEC, t = (int)(idx / CV);
    const int img = t / (p.tx * p.ty), r = t - img * (p.tx * p.ty);
    const int ty = r / p.tx, tx = r - ty * p.tx;
    const float* base = p.g + p.g_off + c + (size_t)((img * p.H + ty * 4) * p.W + tx * 4) * p.g_ld;
    float tcol[VEC][6][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      vec_t dv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) dv[a] = *reinterpret_cast<const vec_t*>(base + (size_t)(a * p.W + b) * p.g_ld);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float d[4], o[6];
#pragma unroll
        for (int a = 0; a < 4; ++a) d[a] = xf_get<VEC>(dv[a], e);
        at6(d, o);
#pragma unroll
        for (int a = 0; a < 6; ++a) tcol[e][a][b] = o[a];
      }
    }
    float* dst = p.Z + (size_t)t * p.Cout + c;
    const size_t plane = (size_t)p.T * p.Cout;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      float o[VEC][6];
#pragma unroll
      for (int e = 0; e < VEC; ++e) at6(tcol[e][a], o[e]);
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        float w[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) w[e] = o[e][b];
        *reinterpret_cast<vec_t*>(dst + (size_t)(a * 6 + b) * plane) = xf_make<VEC>(w);
      }
    }
  }
}

// ---- batched split-K GEMM: slab[split][pos][co][ci] = sum_{t in the split} Z[pos][t][co] * V[pos][t][ci] ----
// Workgroup = 4 waves = a 128 co x 128 ci block, wave (a, b) = the 64 x 64 quadrant (2 x 2 MFMA blocks, 64 accumulators);
// a stage = 16 tiles: rows [t][128 channels] of both operands, K-major exactly as they lie in memory, double buffered, one
// barrier per stage.  MFMA operands: lane (li, lh) reads row (k + lh), column (block + li) -- 32 consecutive floats per
// half wave, conflict-free; the rows of a stage sit LDP = 132 floats apart.
template <bool kVQuads>
__global__ __launch_bounds__(kGemmThreads, 4) void wgrad4_gemm_kernel(const W4Params p) {
  SSDE_LDS(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware block order: the (co, ci) blocks of one (position, split) read the same rows of both operands, so they run
  // on ONE XCD (blocks are dealt round-robin over the 8 XCDs, each with its own L2) -- as consecutive blocks they landed on
  // different XCDs and every operand row was fetched from HBM once per block that uses it (PMC: 469 MB per launch
  // against 263 MB of operands, profiles/r3_train_pmc.json)
#ifdef SSDE_WG4_TRACE
  const bool tr_on = blockIdx.x == 0 && tid == 0 && g_wg4_trace != nullptr;
#endif
  SSDE_WT(0);
  const int ntiles = p.co_tiles * p.ci_tiles;
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
  const int tile = p.xcd_order ? lin % ntiles : (int)blockIdx.x % ntiles;
  const int grp = p.xcd_order ? (lin / ntiles) * 8 + xcd : (int)blockIdx.x / ntiles;       // (position, split) pair / stream-K group
  if (grp >= (p.sk_on ? p.sk_groups : kPos * p.splits)) return;
  const int ci_t = tile % p.ci_tiles, co_t = tile / p.ci_tiles;
  const int co0 = co_t * BM, ci0 = ci_t * BN;
  // Stream-K (default; SSDE_WGRAD4_STREAMK=0 switches it off): the 36 positions x sk_S stages are ONE line of work that the groups cut into equal runs of
  // sk_per stages, so every CU gets the same number of stages whatever 36 x blocks x splits comes to (the plain split leaves
  // 576 workgroups on 256 CUs for the 256 -> 256 @16x16 and 128 -> 128 @32x32 layers: 0.75 of the chip).  A group is the
  // ntiles workgroups that share operand rows (same XCD, same K range at the same time).  A run that crosses a position
  // boundary is two segments: each accumulates from zero and leaves its own partial slab, slot = group - first group of
  // that position; the reduction adds a position's slots in order (deterministic).
  int sk_cur = p.sk_on ? grp * p.sk_per : 0;
  const int sk_end = p.sk_on ? min(kPos * p.sk_S, sk_cur + p.sk_per) : 1;
  for (; sk_cur < sk_end;) {
  int pos, k0, k1, slot;
  if (p.sk_on) {
    pos = sk_cur / p.sk_S;
    const int s0 = sk_cur - pos * p.sk_S;
    const int s1 = min(p.sk_S, s0 + (sk_end - sk_cur));
    k0 = s0 * BK; k1 = min(p.T, s1 * BK);
    slot = grp - (pos * p.sk_S) / p.sk_per;
    sk_cur += s1 - s0;
  } else {
    const int split = grp % p.splits;
    pos = grp / p.splits;
    k0 = split * p.k_per_split;
    k1 = min(p.T, k0 + p.k_per_split);
    slot = split;
    sk_cur = sk_end;
  }
  const int nst = (k1 - k0 + BK - 1) / BK;
  const float* Zp = p.Z + (size_t)pos * p.T * p.Cout;
  const float* Vp = p.V + (size_t)pos * p.T * p.Ctot;

  // staging: thread = (row r0 = tid >> 5 (+8), channel quad q = tid & 31) of both operands: 2 float4 per operand and stage
  const int q4 = (tid & 31) * 4, r0 = tid >> 5;
  const bool zc_ok = co0 + q4 < p.Cout, vc_ok = ci0 + q4 < p.Ctot;
  // V in the by-product layout [pos][Ctot / 4][t][4]: thread = (tile row t = tid & 15, channel quads (tid >> 4) + 16 i) -- 16 lanes
  // read 256 contiguous bytes, and the float4 of a 16-lane group land in LDS rows 132 floats apart (all 64 banks once)
  const int vt = tid & 15, vq = tid >> 4;
  const float* Vq = p.V + (size_t)pos * (p.Ctot >> 2) * p.T * 4;
  float4 zv[2], vv[2];
  auto load_stage = [&](int st) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = k0 + st * BK + r0 + i * 8;
      const bool ok = t < k1;
      const size_t row = (size_t)(ok ? t : k0);
      zv[i] = *reinterpret_cast<const float4*>(Zp + row * p.Cout + (zc_ok ? co0 + q4 : 0));
      if (!ok || !zc_ok) zv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kVQuads) {
        const int tq = k0 + st * BK + vt, cq = (ci0 >> 2) + vq + 16 * i;
        const bool okq = tq < k1 && cq * 4 < p.Ctot;
        vv[i] = *reinterpret_cast<const float4*>(Vq + ((size_t)(okq ? cq : 0) * p.T + (okq ? tq : k0)) * 4);
        if (!okq) vv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        vv[i] = *reinterpret_cast<const float4*>(Vp + row * p.Ctot + (vc_ok ? ci0 + q4 : 0));
        if (!ok || !vc_ok) vv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_stage = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<float4*>(buf + (r0 + i * 8) * LDP + q4) = zv[i];
      if (kVQuads) *reinterpret_cast<float4*>(buf + (BK + vt) * LDP + (vq + 16 * i) * 4) = vv[i];
      else *reinterpret_cast<float4*>(buf + (BK + r0 + i * 8) * LDP + q4) = vv[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int aoff = lh * LDP + wm0 + li, boff = (BK + lh) * LDP + wn0 + li;

  load_stage(0);
  store_stage(smem);
  __syncthreads();
  SSDE_WT(1);
#ifdef SSDE_WG4_TRACE
  if (tr_on) g_wg4_trace[100] = (unsigned long long)nst;
#endif
  for (int st = 0; st < nst; ++st) {
    const float* cur = smem + (st & 1) * kStage;
    const bool has_next = st + 1 < nst;
    if (st < 8) SSDE_WT(4 + 4 * st);
    if (has_next) load_stage(st + 1);
    float af[2][2], bf[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) af[0][a] = cur[aoff + a * 32];
#pragma unroll
    for (int c = 0; c < 2; ++c) bf[0][c] = cur[boff + c * 32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int w = kk & 1;
      if (kk + 1 < BK / 2) {
#pragma unroll
        for (int a = 0; a < 2; ++a) af[w ^ 1][a] = cur[aoff + (kk + 1) * 2 * LDP + a * 32];
#pragma unroll
        for (int c = 0; c < 2; ++c) bf[w ^ 1][c] = cur[boff + (kk + 1) * 2 * LDP + c * 32];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[w][a], bf[w][c], acc[a][c], 0, 0, 0);
      // the fragment reads of step kk+1 go out before the 4 MFMAs of step kk
      if (kk + 1 < BK / 2) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    if (st < 8) SSDE_WT(4 + 4 * st + 1);
    if (has_next) store_stage(smem + ((st + 1) & 1) * kStage);
    if (st < 8) SSDE_WT(4 + 4 * st + 2);
    __syncthreads();
    if (st < 8) SSDE_WT(4 + 4 * st + 3);
  }
  SSDE_WT(2);

  // slab[split][pos][co][ci]: a lane's 32 consecutive ci are a 128-byte run.
  // Addresses = one SCALAR base per 32 x 32 block and accumulator row (the wave index read into an SGPR) + ONE per-lane byte
  // offset.  Written with per-element 64-bit pointers hipcc hoisted sixteen row offsets out of the stream-K segment loop, spilled
  // them (54 VGPRs), and every reload -- scratch loads count in vmcnt like the stores before them -- drew an s_waitcnt vmcnt(0)
  // between two stores: 23-39 of a segment's 64 stores each waited out the write acknowledgement of its predecessor.
  // Both channel counts are multiples of 32 (ssde_wgrad_wino4_wants): a block is inside or outside as a whole.
#if defined(SSDE_WG4_ELEMENT_POINTERS) && SSDE_WG4_ELEMENT_POINTERS      // (rounds 3-5, A/B variant only)
  float* slab = p.slabs + ((size_t)slot * kPos + pos) * (size_t)p.Cout * p.Ctot;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ci = ci0 + wn0 + c * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < p.Cout && ci < p.Ctot) slab[(size_t)co * p.Ctot + ci] = acc[a][c][r];
      }
    }
#else
  char* slab = reinterpret_cast<char*>(p.slabs + ((size_t)slot * kPos + pos) * (size_t)p.Cout * p.Ctot);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm0u = (wave_u >> 1) * 64, wn0u = (wave_u & 1) * 64;
  const uint32_t lane_off = ((uint32_t)(4 * lh) * (uint32_t)p.Ctot + (uint32_t)li) * 4u;
  const size_t row_bytes = (size_t)p.Ctot * 4;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (co0 + wm0u + a * 32 < p.Cout && ci0 + wn0u + c * 32 < p.Ctot) {
        char* blk = slab + ((size_t)(co0 + wm0u + a * 32) * p.Ctot + (size_t)(ci0 + wn0u + c * 32)) * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          *reinterpret_cast<float*>(blk + (size_t)((r & 3) + 8 * (r >> 2)) * row_bytes + lane_off) = acc[a][c][r];
      }
    }
#endif
  }                                              // next segment of a stream-K run
  SSDE_WT(3);
}

// Reduction in two launches.  (a) slab[0] += slab[1] + .. + slab[splits-1], element-wise in float4s over all 36 x Cout x Cin
// values: thousands of workgroups at HBM speed (one thread per (co, ci) walking its 36 x splits values was 16-64 workgroups
// for the whole chip).  Splits are added in order (deterministic).
__global__ __launch_bounds__(256) void wgrad4_sum_splits_kernel(const W4Params p) {
  const size_t quads = (size_t)kPos * p.Cout * p.Ctot / 4;
  const size_t stride = (size_t)kPos * p.Cout * p.Ctot;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = *reinterpret_cast<const float4*>(p.slabs + i * 4);
    int ns = p.splits;
    if (p.sk_on) {                                 // stream-K: the slots this position's segments filled
      const int pos = (int)(i * 4 / ((size_t)p.Cout * p.Ctot));
      ns = ((pos + 1) * p.sk_S - 1) / p.sk_per - (pos * p.sk_S) / p.sk_per + 1;
    }
    for (int sp = 1; sp < ns; ++sp) {
      const float4 v = *reinterpret_cast<const float4*>(p.slabs + (size_t)sp * stride + i * 4);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(p.slabs + i * 4) = a;
  }
}
// (b) dw[co, ci, :, :] += scale * G^T dU[co, ci] G with the 6 x 3 G of conv_wino4.hip; thread = one (co, ci), consecutive
// threads -> consecutive ci
__global__ __launch_bounds__(256) void wgrad4_reduce_kernel(const W4Params p) {
  const int total = p.Cout * p.Ctot;
  const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                         {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    float u[kPos];
#pragma unroll
    for (int q = 0; q < kPos; ++q) u[q] = p.slabs[(size_t)q * total + idx];
    float t[3][6];                                 // G^T u
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        float v = 0.f;
#pragma unroll
        for (int a = 0; a < 6; ++a)
          if (G[a][k] != 0.f) v += G[a][k] * u[a * 6 + b];
        t[k][b] = v;
      }
    float* dst = p.dw + (size_t)idx * 9;           // idx = co * Ctot + ci
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int l = 0; l < 3; ++l) {
        float v = 0.f;
#pragma unroll
        for (int b = 0; b < 6; ++b)
          if (G[b][l] != 0.f) v += t[k][b] * G[b][l];
        dst[k * 3 + l] += p.scale * v;
      }
  }
}

// ssde_wgrad_args.flags: SSDE_WGRADF_DIRECT = direct kernel everywhere (0), SSDE_WGRADF_F2 = F(2x2,3x3) wherever legal (2),
// SSDE_WGRADF_F4_FORCE = F(4x4,3x3) wherever legal (44, tests), none = F(4x4,3x3) where it is legal and pays, F(2x2,3x3) for
// the rest (4)
int mode(const ssde_wgrad_args* a) {
  return (a->flags & SSDE_WGRADF_DIRECT) ? 0 : (a->flags & SSDE_WGRADF_F2) ? 2 : (a->flags & SSDE_WGRADF_F4_FORCE) ? 44 : 4;
}

void plan(const ssde_wgrad_args* a, W4Params* p) {
  const ssde_src& s = a->src;
  p->src = s; p->g = a->g; p->g_ld = a->g_ld; p->g_off = a->g_off;
  p->N = a->n; p->H = a->h_out; p->W = a->w_out; p->Cout = a->c_out; p->Ctot = s.c0 + s.c1;
  p->tx = a->w_out / 4; p->ty = a->h_out / 4;
  p->T = a->n * p->tx * p->ty;
  p->co_tiles = ssde_cdiv(a->c_out, BM); p->ci_tiles = ssde_cdiv(p->Ctot, BN);
  // four workgroups share a CU: aim at 768 of them, at least 32 stages (512 tiles) each; every
  // split costs a slab of 36 x Cout x Cin floats that the reduction reads back.  (The K loop itself runs near the MFMA bound --
  // 6.4 k cycles per 16-tile stage with three workgroups on the CU, tools/wgrad4_trace.py -- and what a launch loses is
  // quantisation: 576 workgroups are 3 on a quarter of the CUs and 2 on the rest.  A rule that minimises
  // ceil(workgroups / CUs) x (stages + 3) + 1.2 per split was built and measured: 256 -> 128 @32x32 0.455 -> 0.42 ms and 8x8
  // 512 -> 256 0.113 -> 0.104, but 384 -> 256 @16x16 0.248 -> 0.284 and 128 -> 128 @32x32 0.265 -> 0.278; the weight-gradient
  // class of the training step 16.59 -> 16.54 ms -- not adopted, profiles/r4_wgrad4_kernel_times.txt.)
  const int target = 768;
  const int blocks = kPos * p->co_tiles * p->ci_tiles;
  int splits = (target + blocks / 2) / blocks;
  const int max_splits = p->T / 512;
  if (splits > max_splits) splits = max_splits;
  if (a->splits > 0) splits = a->splits;
  if (splits < 1) splits = 1;
  p->k_per_split = ssde_cdiv(ssde_cdiv(p->T, splits), BK) * BK;
  p->splits = ssde_cdiv(p->T, p->k_per_split);
  p->sk_on = 0; p->sk_per = p->sk_S = p->sk_groups = 0;
  {
    if (!(a->flags & SSDE_WGRADF_NO_STREAMK) && a->splits <= 0) {      // (default on; the flag: the plain split, for A/B runs and tests)
      // three workgroups per CU are what the kernel's registers and LDS keep resident beside each other at full rate
      const int ntiles = p->co_tiles * p->ci_tiles;
      const int S = ssde_cdiv(p->T, BK), total = kPos * S;
      int groups = 3 * ssde_num_cus() / ntiles;
      if (groups < 1) groups = 1;
      int per = ssde_cdiv(total, groups);
      if (per < 8) per = 8;                        // (at least 8 stages per run)
      if (per < S || groups >= kPos) {             // otherwise whole positions per group: the plain form with one split
        p->sk_on = 1; p->sk_per = per; p->sk_S = S; p->sk_groups = ssde_cdiv(total, per);
        p->splits = ssde_cdiv(S, per) + 1;         // the most slots a position can have
      }
    }
  }
  p->scale = a->scale; p->dw = a->dw;
  p->xcd_order = !(a->flags & SSDE_WGRADF_NO_XCD_ORDER);
}

int64_t scratch_floats(const W4Params& p) {
  return (int64_t)kPos * p.T * (p.Ctot + p.Cout) + (int64_t)p.splits * kPos * p.Cout * p.Ctot;
}

}  // namespace

// ssde_conv_wgrad / ssde_wgrad_scratch_floats (wgrad.hip) route eligible launches here, before wgrad_wino.hip
bool ssde_wgrad_wino4_wants(const ssde_wgrad_args* a) {
  const int m = mode(a);
  if ((m != 4 && m != 44) || a->ksize != 3 || a->stride != 1 || a->pad != 1 || a->transpose_out) return false;
  if (a->h_in != a->h_out || a->w_in != a->w_out || a->h_out % 4 != 0 || a->w_out % 4 != 0 || a->h_out < 8 || a->w_out < 8) return false;
  const ssde_src& s = a->src;
  const int Ctot = s.c0 + s.c1;
  if (a->c_out % 32 != 0 || Ctot % 32 != 0 || a->cin_store != Ctot) return false;
  if (a->g_ld % 4 != 0 || a->g_off % 4 != 0 || s.c0 % 4 != 0 || s.c1 % 4 != 0) return false;
  // what wgrad.hip's make_plan would have checked had it been asked first: the transforms read one (mean, rstd) per channel
  // quad, and the gradient block must lie inside a row of the parameter's gradient
  if (a->g_off + a->c_out > a->g_ld) return false;
  if ((s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU) &&
      (s.gn_groups <= 0 || Ctot % s.gn_groups != 0 || (Ctot / s.gn_groups) % 4 != 0)) return false;
  const long long T = (long long)a->n * (a->h_out / 4) * (a->w_out / 4);
  // where it pays (profiles/r3_wgrad_wino4_ab.txt, batch 128): from 16x16 maps up when a channel count exceeds 128, 128 -> 128
  // from 8192 tiles (32x32 maps at batch 128: +3-4 % since the transforms move two channels per thread), and on 8x8 maps from
  // 512 input channels; the rest stays on the fused F(2x2,3x3) kernel
  if (m == 4 && !((T >= 1024 && (Ctot > 128 || a->c_out > 128)) || T >= 8192 || (T >= 512 && Ctot >= 512))) return false;
  return (long long)a->n * a->h_out * a->w_out < (1ll << 30) && T * (Ctot > a->c_out ? Ctot : a->c_out) * kPos < (1ll << 40);
}

int64_t ssde_wgrad_wino4_scratch_floats(const ssde_wgrad_args* a) {
  W4Params p;
  ssde_wgrad_args b = *a;
  b.splits = 0;
  plan(&b, &p);
  return scratch_floats(p);
}

int ssde_wgrad_wino4_launch(const ssde_wgrad_args* a, void* stream) {
  const ssde_src& s = a->src;
  SSDE_REQUIRE(a->g && a->dw && s.p0 && (s.c1 == 0 || s.p1), "wgrad(winograd 4x4): null tensors");
  const bool gn = s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU;
  if (gn) {
    SSDE_REQUIRE(s.gn_groups > 0 && (s.c0 + s.c1) % s.gn_groups == 0, "wgrad(winograd 4x4): GroupNorm groups");
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "wgrad(winograd 4x4): GroupNorm pointers missing");
  }
  SSDE_REQUIRE(s.drop_thresh == 0 || s.drop_seed, "wgrad(winograd 4x4): dropout seed pointer missing");
  W4Params p;
  plan(a, &p);
  if (a->scratch_floats < scratch_floats(p)) {          // fewer splits when scratch is short (the operands must fit)
    const int64_t left = a->scratch_floats - (int64_t)kPos * p.T * (p.Ctot + p.Cout);
    const int fit = (int)(left / ((int64_t)kPos * p.Cout * p.Ctot));
    SSDE_REQUIRE(fit >= 1 && a->scratch, "wgrad(winograd 4x4): scratch of %lld floats needed at least",
                 (long long)kPos * p.T * (p.Ctot + p.Cout) + (long long)kPos * p.Cout * p.Ctot);
    p.k_per_split = ssde_cdiv(ssde_cdiv(p.T, fit), BK) * BK;
    p.splits = ssde_cdiv(p.T, p.k_per_split);
    p.sk_on = 0;
  }
  SSDE_REQUIRE(a->scratch, "wgrad(winograd 4x4): scratch missing");
  // v_pre: the forward launch of this layer left V behind (conv_wino4.hip, kEmitV): no input-transform pass here
  const bool have_v = a->v_pre != nullptr;
  p.V = have_v ? const_cast<float*>(a->v_pre) : a->scratch;
  p.v_quads = have_v ? 1 : 0;
  if (have_v) {
    p.Z = a->scratch;
    p.slabs = p.Z + (size_t)kPos * p.T * p.Cout;
  } else {
  p.Z = p.V + (size_t)kPos * p.T * p.Ctot;
  p.slabs = p.Z + (size_t)kPos * p.T * p.Cout;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto grid_for = [](long long total) { long long b = (total + 255) / 256; return (unsigned)(b > 256 * 32 ? 256 * 32 : (b < 1 ? 1 : b)); };
  const int xvec = (a->flags & SSDE_WGRADF_XVEC1) ? 1 : 2;      // channels per transform thread
  const bool v2 = xvec == 2 && p.Ctot % 2 == 0 && p.Cout % 2 == 0 && s.c0 % 2 == 0 && p.g_ld % 2 == 0 && p.g_off % 2 == 0;
  const dim3 gv(grid_for((long long)p.T * p.Ctot / (v2 ? 2 : 1))), gz(grid_for((long long)p.T * p.Cout / (v2 ? 2 : 1)));
  if (v2) {
    if (!have_v) {
      if (gn) hipLaunchKernelGGL((wino4_xform_v_kernel<true, 2>), gv, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((wino4_xform_v_kernel<false, 2>), gv, dim3(256), 0, st, p);
      SSDE_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(wino4_xform_z_kernel<2>, gz, dim3(256), 0, st, p);
  } else {
    if (!have_v) {
      if (gn) hipLaunchKernelGGL((wino4_xform_v_kernel<true, 1>), gv, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((wino4_xform_v_kernel<false, 1>), gv, dim3(256), 0, st, p);
      SSDE_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(wino4_xform_z_kernel<1>, gz, dim3(256), 0, st, p);
  }
  SSDE_LAUNCH_CHECK();
  const int lds = 2 * kStage * 4;
  const dim3 gg(ssde_cdiv(p.sk_on ? p.sk_groups : kPos * p.splits, 8) * 8 * p.co_tiles * p.ci_tiles);
  if (have_v) hipLaunchKernelGGL(wgrad4_gemm_kernel<true>, gg, dim3(kGemmThreads), lds, st, p);
  else hipLaunchKernelGGL(wgrad4_gemm_kernel<false>, gg, dim3(kGemmThreads), lds, st, p);
  SSDE_LAUNCH_CHECK();
  if (p.splits > 1) {
    hipLaunchKernelGGL(wgrad4_sum_splits_kernel, dim3(grid_for((long long)kPos * p.Cout * p.Ctot / 4)), dim3(256), 0, st, p);
    SSDE_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(wgrad4_reduce_kernel, dim3(grid_for((long long)p.Cout * p.Ctot)), dim3(256), 0, st, p);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
