// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions by Winograd F(4x4, 3x3) on the exact-fp32 matrix pipe.
//
// Forward (conv_wino4.hip):  Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A   per 4x4 output tile, 6x6 transform positions.
// Differentiating,
//
//   dL/dg[co, ci] = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G
//
// so per transform position p (36 of them) the gradient is a GEMM  dU_p[co, ci] = sum_tiles Z_p[tile, co] * V_p[tile, ci]
// whose reduction runs over the 4x4 tiles of the whole batch: 36 multiply-adds per (tile, co, ci) for 16 output pixels,
// against 144 of the direct form (wgrad.hip) and 64 of F(2x2,3x3) (wgrad_wino.hip) -- the matrix pipe does 1.78x less
// work than the kernel this replaces on the layers that dominate the training step (replaces loss.backward()'s weight
// gradients of ddpm_conv3x3, /root/reference losses.py:196 through models/layers.py:118-124).
// Rounding: tools/experiments/wgrad_wino43_error_budget.py -- relative L2 error 2.6e-6 .. 3.0e-6 against fp64 at the
// reduction lengths of the training step (direct fp32: 2.8e-7 .. 3.7e-7, F(2x2,3x3): 4.5e-7 .. 5.7e-7), against the 2e-4
// the gradient parity tests allow.
//
// One workgroup (8 waves, two per SIMD) owns a 32 co x 64 ci block of dU for all 36 positions -- the matrix layout of
// conv_wino4.hip with the roles changed: M = 32 output channels (Z), N = 64 input channels (V), K = tiles; wave (q, h)
// = positions q + 4 j (j = 0..8) x the ci half h on v_mfma_f32_32x32x2_f32, 144 accumulator registers -- and walks a
// contiguous range of CHUNKS: one chunk = an 8 x 8 output patch of one image = 2 x 2 tiles = one stage (K = 4: two MFMAs
// per position).  Per stage:
//   global -> registers: the 10 x 10 halo of pro(src) (GroupNorm / SiLU / dropout recomputed, as everywhere) for the 64
//                        input channels, the 8 x 8 patch of dY for the 32 output channels           (3 stages ahead)
//   registers -> raw LDS [pixel][channel] (single buffered: written behind a second, LDS-only barrier once every wave has
//                        read the previous chunk's columns)                                          (2 stages ahead)
//   raw -> V = B^T d B [pos][ci][4 tiles] and Z = A dY A^T [pos][co][4 tiles], each in two 1-D passes through the
//                        destination buffer; a lane transforms the horizontally adjacent tiles (2 ty, 2 ty + 1) as one
//                        float2, which is exactly the k pair a fragment read needs; a wave owns whole items, so pass 2
//                        reads what the same wave wrote in pass 1 (no barrier between the passes)    (1 stage ahead)
// all of it riding between the wave's 18 MFMAs, every wave running the same branch-free program (flags are template
// parameters, the last stages are peeled), as conv_wino4.hip.  The chunk range is split over <= 256 workgroups with equal
// counts per XCD; every workgroup writes its partial dU block to a slab and wgrad_wino4_reduce_kernel sums the slabs in a
// fixed order (deterministic) and applies G^T . G.
#include "ssde_common.h"
#include <cstdlib>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kThreads = 512, kNP = 9, kPS = 4, kPos = 36;
constexpr int kCo = 32, kCi = 64;                      // block of dU per workgroup
constexpr int kZP = kCo * 4 + 2;                       // floats per position of Z [co][4 tiles]; = 2 mod 4: the b64 writes of
constexpr int kVP = kCi * 4 + 2;                       // two transform columns x 8 channels interleave over the banks
constexpr int kZFloats = kPos * kZP, kVFloats = kPos * kVP;
constexpr int kXP = kCi + 8, kGP = kCo + 8;            // channel pitch of the raw buffers (columns 8 banks apart)
constexpr int kRawX = 100 * kXP, kRawG = 64 * kGP;     // 10 x 10 halo pixels, 8 x 8 patch pixels
constexpr int kSlab = kPos * kCo * kCi;                // floats of one partial dU block
constexpr int kLds = (2 * kVFloats + 2 * kZFloats + kRawX + kRawG) * 4;
static_assert(kLds <= 160 * 1024, "LDS budget");

struct Ww4Params {
  ssde_src src;
  const float* g;
  int g_ld, g_off;
  int N, H, W, Cout, Ctot;
  int cx, cy;              // chunks per image row / column (W / 8, H / 8)
  int chunks, chunks_per_split, splits;
  int co_tiles, ci_tiles;
  float scale;
  float* dw;
  float* scratch;
};

// 1-D input transform, two tiles at once (B^T rows as in conv_wino4.hip)
__device__ __forceinline__ void bt6(const ssde_f32x2 (&d)[6], ssde_f32x2 (&o)[6]) {
  const ssde_f32x2 t1 = d[4] - 4.f * d[2], t2 = d[3] - 4.f * d[1], t3 = d[4] - d[2], t4 = d[3] - d[1];
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = t1 + t2;
  o[2] = t1 - t2;
  o[3] = t3 + 2.f * t4;
  o[4] = t3 - 2.f * t4;
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// 1-D transform of the output gradient with A (6 x 4: rows [1,0,0,0] [1,1,1,1] [1,-1,1,-1] [1,2,4,8] [1,-2,4,-8] [0,0,0,1])
__device__ __forceinline__ void at6(const ssde_f32x2 (&d)[4], ssde_f32x2 (&o)[6]) {
  const ssde_f32x2 s = d[0] + d[2], t = d[1] + d[3], u = d[0] + 4.f * d[2], v = 2.f * d[1] + 8.f * d[3];
  o[0] = d[0];
  o[1] = s + t;
  o[2] = s - t;
  o[3] = u + v;
  o[4] = u - v;
  o[5] = d[3];
}

template <bool kGn>
__global__ __launch_bounds__(kThreads, 2) void wgrad_wino4_kernel(const Ww4Params p) {
  SSDE_LDS(smem);
  float* Vb = smem;                              // [2][kVFloats]   B operand (input channels)
  float* Zb = Vb + 2 * kVFloats;                 // [2][kZFloats]   A operand (output channels)
  float* rawx = Zb + 2 * kZFloats;               // [100 halo pixels][kXP]
  float* rawg = rawx + kRawX;                    // [64 patch pixels][kGP]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware order (wgrad_wino.hip): a workgroup needs a whole CU, so every XCD gets the same number of workgroups; the
  // (co, ci) blocks of one chunk range sit on one XCD (at most two) and share its L2
  const int ntiles = p.co_tiles * p.ci_tiles;
  const int total = p.splits * ntiles, per_xcd = (total + 7) >> 3;
  const int jb = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (jb >= total || (blockIdx.x >> 3) >= per_xcd) return;
  const int tile = jb % ntiles, split = jb / ntiles;
  const int co0 = (tile / p.ci_tiles) * kCo, ci0 = (tile % p.ci_tiles) * kCi;
  const int ch_begin = split * p.chunks_per_split;
  const int nst = min(p.chunks, ch_begin + p.chunks_per_split) - ch_begin;

  const ssde_src& s = p.src;
  SsdePro pro = ssde_pro_decode(s);
  pro.gn = kGn;
  const int cpg = kGn ? p.Ctot / s.gn_groups : 1;
  // the 64 input channels of this block lie in one tensor of the virtual concat
  const bool second = ci0 >= s.c0;
  const float* xbase = second ? s.p1 : s.p0;
  const int xC = second ? s.c1 : s.c0;
  const int xc0 = second ? ci0 - s.c0 : ci0;

  // ---- staging plan: x item = (halo pixel, channel quad), 1600 of them (the fourth exists for tid < 64); dY item = (patch
  // pixel, channel quad), one per thread.  The quad of a thread is the same for all its items ----
  const int quad = tid & 15, gquad = tid & 7, gpx = tid >> 3;
  int hy[4], hx[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int px = min((tid + it * kThreads) >> 4, 99);
    hy[it] = px / 10; hx[it] = px - hy[it] * 10;
  }
  const bool x_item3 = tid < 64;
  const float* xq = xbase + xc0 + quad * 4;
  const float* gq = p.g + p.g_off + co0 + gquad * 4;
  float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kGn) {
    gam = *reinterpret_cast<const float4*>(s.gn_gamma + ci0 + quad * 4);
    bet = *reinterpret_cast<const float4*>(s.gn_beta + ci0 + quad * 4);
  }
  const int ggrp = (ci0 + quad * 4) / cpg;
  const int per_img = p.cx * p.cy;

  float4 xv[4], gv;
  float mu = 0.f, rs = 1.f;
  // chunk cursors (image, chunk row, chunk column), wave-uniform: one walked by the loads, one by the raw stores; both
  // visit the chunks of the range in order, so a step is two compares instead of two integer divisions
  int l_img, l_cy, l_cx, s_img, s_cy, s_cx;
  {
    l_img = ch_begin / per_img;
    const int r = ch_begin - l_img * per_img;
    l_cy = r / p.cx; l_cx = r - l_cy * p.cx;
    s_img = l_img; s_cy = l_cy; s_cx = l_cx;
  }
  auto advance = [&](int& img, int& cy, int& cx) __attribute__((always_inline)) {
    cx += 1;
    const bool wx = cx == p.cx;
    cx = wx ? 0 : cx;
    cy += wx ? 1 : 0;
    const bool wy = cy == p.cy;
    cy = wy ? 0 : cy;
    img += wy ? 1 : 0;
  };
  // global loads of the next chunk (branch-free: pixels outside the image read a clamped address and are zeroed in
  // store_raw, which re-derives the coordinates)
  auto load_chunk = [&]() __attribute__((always_inline)) {
    const int img = l_img, cy = l_cy, cx = l_cx;
    if (kGn) {
      mu = s.gn_mean[img * s.gn_groups + ggrp];
      rs = s.gn_rstd[img * s.gn_groups + ggrp];
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int iy = cy * 8 - 1 + hy[it], ix = cx * 8 - 1 + hx[it];
      const bool inb = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int pix = (img * p.H + (inb ? iy : 0)) * p.W + (inb ? ix : 0);
      xv[it] = *reinterpret_cast<const float4*>(xq + (size_t)pix * xC);
    }
    const int pix = (img * p.H + cy * 8 + (gpx >> 3)) * p.W + cx * 8 + (gpx & 7);
    gv = *reinterpret_cast<const float4*>(gq + (size_t)pix * p.g_ld);
    advance(l_img, l_cy, l_cx);
  };
  // prologue + raw LDS stores of the chunk in the registers
  auto store_raw = [&]() __attribute__((always_inline)) {
    const int img = s_img, cy = s_cy, cx = s_cx;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (it == 3 && !x_item3) continue;
      const int iy = cy * 8 - 1 + hy[it], ix = cx * 8 - 1 + hx[it];
      const bool inb = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (inb) {
        const int pix = (img * p.H + iy) * p.W + ix;
        v = ssde_pro_apply(xv[it], mu, rs, gam, bet, (uint32_t)pix * (uint32_t)p.Ctot + (uint32_t)(ci0 + quad * 4), pro);
      }
      *reinterpret_cast<float4*>(rawx + ((tid + it * kThreads) >> 4) * kXP + quad * 4) = v;
    }
    *reinterpret_cast<float4*>(rawg + gpx * kGP + gquad * 4) = gv;
    advance(s_img, s_cy, s_cx);
  };

  // ---- transform plan.  V: wave w owns the input channels w*8 .. w*8+7, round r = tile row ty; lane = line * 8 + channel
  // (lines 6, 7 repeat lines 0, 1: same addresses, same values -- straight-line code for every lane).  Z: wave w owns the
  // tile row ty = w >> 2 of the output channels (w & 3) * 8 .. + 7; pass 1 lane = column (4) * 8 + channel, pass 2 lane =
  // row (6) * 8 + channel, the remaining lanes repeat the first ones ----
  const int t_ch = lane & 7;
  const int t_l6 = (lane >> 3) >= 6 ? (lane >> 3) - 6 : (lane >> 3);         // line of a 6-line pass
  const int t_l4 = (lane >> 3) & 3;                                            // column of the 4-column pass
  const int v_ci = wave * 8 + t_ch;
  const int z_co = (wave & 3) * 8 + t_ch, z_ty = wave >> 2;

  const int wq = wave >> 1, wh = wave & 1;
  f32x16 acc[kNP];
#pragma unroll
  for (int j = 0; j < kNP; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int a_off = li * 4 + 2 * lh, b_off = (wh * 32 + li) * 4 + 2 * lh;

  // pass 1 of V, round r: reads of the lane's column of both tiles (x and x + 4), arithmetic, writes
  auto v1_read = [&](int r, ssde_f32x2 (&d)[6]) __attribute__((always_inline)) {
    const float* rp = rawx + ((4 * r) * 10 + t_l6) * kXP + v_ci;
#pragma unroll
    for (int a = 0; a < 6; ++a) { d[a].x = rp[a * 10 * kXP]; d[a].y = rp[a * 10 * kXP + 4 * kXP]; }
  };
  auto v1_write = [&](float* Vn, int r, const ssde_f32x2 (&d)[6]) __attribute__((always_inline)) {
    ssde_f32x2 o[6];
    bt6(d, o);
    float* vp = Vn + t_l6 * kVP + v_ci * 4 + 2 * r;
#pragma unroll
    for (int a = 0; a < 6; ++a) *reinterpret_cast<ssde_f32x2*>(vp + a * 6 * kVP) = o[a];
  };
  auto v2_read = [&](const float* Vn, int r, ssde_f32x2 (&d)[6]) __attribute__((always_inline)) {
    const float* vp = Vn + (t_l6 * 6) * kVP + v_ci * 4 + 2 * r;
#pragma unroll
    for (int b = 0; b < 6; ++b) d[b] = *reinterpret_cast<const ssde_f32x2*>(vp + b * kVP);
  };
  auto v2_write = [&](float* Vn, int r, const ssde_f32x2 (&d)[6]) __attribute__((always_inline)) {
    ssde_f32x2 o[6];
    bt6(d, o);
    float* vp = Vn + (t_l6 * 6) * kVP + v_ci * 4 + 2 * r;
#pragma unroll
    for (int b = 0; b < 6; ++b) *reinterpret_cast<ssde_f32x2*>(vp + b * kVP) = o[b];
  };
  auto z1_read = [&](ssde_f32x2 (&d)[4]) __attribute__((always_inline)) {
    const float* gp = rawg + ((4 * z_ty) * 8 + t_l4) * kGP + z_co;
#pragma unroll
    for (int r = 0; r < 4; ++r) { d[r].x = gp[r * 8 * kGP]; d[r].y = gp[r * 8 * kGP + 4 * kGP]; }
  };
  auto z1_write = [&](float* Zn, const ssde_f32x2 (&d)[4]) __attribute__((always_inline)) {
    ssde_f32x2 o[6];
    at6(d, o);
    float* zp = Zn + t_l4 * kZP + z_co * 4 + 2 * z_ty;
#pragma unroll
    for (int a = 0; a < 6; ++a) *reinterpret_cast<ssde_f32x2*>(zp + a * 6 * kZP) = o[a];
  };
  auto z2_read = [&](const float* Zn, ssde_f32x2 (&d)[4]) __attribute__((always_inline)) {
    const float* zp = Zn + (t_l6 * 6) * kZP + z_co * 4 + 2 * z_ty;
#pragma unroll
    for (int c = 0; c < 4; ++c) d[c] = *reinterpret_cast<const ssde_f32x2*>(zp + c * kZP);
  };
  auto z2_write = [&](float* Zn, const ssde_f32x2 (&d)[4]) __attribute__((always_inline)) {
    ssde_f32x2 o[6];
    at6(d, o);
    float* zp = Zn + (t_l6 * 6) * kZP + z_co * 4 + 2 * z_ty;
#pragma unroll
    for (int b = 0; b < 6; ++b) *reinterpret_cast<ssde_f32x2*>(zp + b * kZP) = o[b];
  };

  // ---- pipeline fill.  On entry to stage st: V / Z[st & 1] = chunk st transformed, raw = chunk st + 1, registers = chunk
  // st + 2 (in flight) ----
  load_chunk();
  store_raw();
  if (nst > 1) load_chunk();
  SSDE_LDS_BARRIER();
  {
    ssde_f32x2 d[6], z[4];
    v1_read(0, d); v1_write(Vb, 0, d);
    v1_read(1, d); v1_write(Vb, 1, d);
    z1_read(z); z1_write(Zb, z);
    SSDE_LDS_BARRIER();                          // raw may be overwritten; (and the test emulator's fibers meet here)
    // pass 2 is in place and lanes 48-63 repeat lines 0, 1: all reads of a wave before its writes (a wave's lanes run in
    // lockstep; the barrier is for the test emulator, whose lanes are fibers -- in the stage body MFMAs sit in between)
    ssde_f32x2 d1[6];
    v2_read(Vb, 0, d); v2_read(Vb, 1, d1); z2_read(Zb, z);
    SSDE_LDS_BARRIER();
    v2_write(Vb, 0, d); v2_write(Vb, 1, d1); z2_write(Zb, z);
  }
  if (nst > 1) store_raw();
  if (nst > 2) load_chunk();
  SSDE_LDS_BARRIER();

  // ---- one stage.  HAS1: chunk st + 1 exists (it is transformed here), HAS2: the chunk in the registers (st + 2) is
  // activated and stored, HAS3: chunk st + 3 is fetched.
  //   head     fragment reads of positions 0..2 | pass-1 reads of V round 0
  //   slots    2 MFMAs of position j and the fragment reads of position j + 3, and between them:
  //            after 1: V round 0 pass 1 -> writes, reads of round 1 | after 2: round 1 -> writes, pass-1 reads of Z |
  //            after 3: Z pass 1 -> writes, LDS-ONLY BARRIER (every wave has read its columns of raw), prologue of the
  //            registers -> raw, pass-2 reads of V round 0 | after 4: round 0 pass 2 -> writes, reads of round 1 |
  //            after 5: round 1 -> writes, pass-2 reads of Z | after 6: Z pass 2 -> writes | after 7: global loads of chunk
  //            st + 3
  //   tail     LDS-only barrier
  auto stage = [&](auto H1, auto H2, auto H3, const int st) __attribute__((always_inline)) {
    constexpr bool has1 = decltype(H1)::value, has2 = decltype(H2)::value, has3 = decltype(H3)::value;
    const int cur = st & 1, nxt = cur ^ 1;
    const float* Vc = Vb + cur * kVFloats;
    const float* Zc = Zb + cur * kZFloats;
    float* Vn = Vb + nxt * kVFloats;
    float* Zn = Zb + nxt * kZFloats;
    ssde_f32x2 td[6], tz[4];
    ssde_f32x2 af[4], bf[4];
    ssde_lds_cfloat* za = (ssde_lds_cfloat*)(Zc + wq * kZP + a_off);
    ssde_lds_cfloat* va = (ssde_lds_cfloat*)(Vc + wq * kVP + b_off);
    SSDE_OPAQUE_VGPR(za);
    SSDE_OPAQUE_VGPR(va);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      af[j] = *(ssde_lds_cfloat2*)(za + kPS * j * kZP);
      bf[j] = *(ssde_lds_cfloat2*)(va + kPS * j * kVP);
    }
    if (has1) v1_read(0, td);
    __builtin_amdgcn_sched_barrier(0);
#define SSDE_WW4_POS(J)                                                                                          \
    do {                                                                                                         \
      if ((J) + 3 < kNP) {                                                                                         \
        af[((J) + 3) & 3] = *(ssde_lds_cfloat2*)(za + kPS * ((J) + 3) * kZP);                                      \
        bf[((J) + 3) & 3] = *(ssde_lds_cfloat2*)(va + kPS * ((J) + 3) * kVP);                                      \
      }                                                                                                          \
      acc[J] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[(J) & 3].x, bf[(J) & 3].x, acc[J], 0, 0, 0);               \
      acc[J] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[(J) & 3].y, bf[(J) & 3].y, acc[J], 0, 0, 0);               \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
    } while (0)
    SSDE_WW4_POS(0);
    SSDE_WW4_POS(1);
    if (has1) { v1_write(Vn, 0, td); v1_read(1, td); }
    __builtin_amdgcn_sched_barrier(0);
    SSDE_WW4_POS(2);
    if (has1) { v1_write(Vn, 1, td); z1_read(tz); }
    __builtin_amdgcn_sched_barrier(0);
    SSDE_WW4_POS(3);
    if (has1) z1_write(Zn, tz);
    SSDE_LDS_BARRIER();                          // every wave has read its columns of raw (chunk st + 1)
    if (has2) store_raw();
    if (has1) v2_read(Vn, 0, td);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_WW4_POS(4);
    if (has1) { v2_write(Vn, 0, td); v2_read(Vn, 1, td); }
    __builtin_amdgcn_sched_barrier(0);
    SSDE_WW4_POS(5);
    if (has1) { v2_write(Vn, 1, td); z2_read(Zn, tz); }
    __builtin_amdgcn_sched_barrier(0);
    SSDE_WW4_POS(6);
    if (has1) z2_write(Zn, tz);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_WW4_POS(7);
    if (has3) load_chunk();
    __builtin_amdgcn_sched_barrier(0);
    SSDE_WW4_POS(8);
#undef SSDE_WW4_POS
    SSDE_LDS_BARRIER();
  };
  {
    using T = std::true_type; using F = std::false_type;
    int st = 0;
    if (nst >= 3) {
      for (; st + 3 < nst; ++st) stage(T{}, T{}, T{}, st);
      stage(T{}, T{}, F{}, st); ++st;
      stage(T{}, F{}, F{}, st); ++st;
    } else if (nst == 2) {
      stage(T{}, F{}, F{}, st); ++st;
    }
    stage(F{}, F{}, F{}, st);
  }

  // ---- partial block dU[pos][co][ci] -> slab [split][block][36][32][64]: a lane's row of 32 ci is a 128-byte run ----
  float* slab = p.scratch + ((size_t)split * ntiles + tile) * (size_t)kSlab;
#pragma unroll
  for (int j = 0; j < kNP; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = (r & 3) + 8 * (r >> 2) + 4 * lh;
      slab[((size_t)(wq + kPS * j) * kCo + co) * kCi + wh * 32 + li] = acc[j][r];
    }
}

// dw[co, ci, :, :] += scale * G^T (sum_splits dU[co, ci]) G with the 6 x 3 G of conv_wino4.hip; thread = one (block, co, ci),
// consecutive threads -> consecutive ci (coalesced slab reads); splits summed in order (deterministic)
__global__ __launch_bounds__(256) void wgrad_wino4_reduce_kernel(const Ww4Params p) {
  const int ntiles = p.co_tiles * p.ci_tiles;
  const int total = ntiles * kCo * kCi;
  const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                         {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int tile = idx / (kCo * kCi), co_l = (idx / kCi) % kCo, ci_l = idx % kCi;
    float u[kPos];
#pragma unroll
    for (int q = 0; q < kPos; ++q) u[q] = 0.f;
    for (int sp = 0; sp < p.splits; ++sp) {
      const float* slab = p.scratch + ((size_t)sp * ntiles + tile) * (size_t)kSlab + co_l * kCi + ci_l;
#pragma unroll
      for (int q = 0; q < kPos; ++q) u[q] += slab[q * (kCo * kCi)];
    }
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
    const int co = (tile / p.ci_tiles) * kCo + co_l, ci = (tile % p.ci_tiles) * kCi + ci_l;
    float* dst = p.dw + ((size_t)co * p.Ctot + ci) * 9;
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

// SSDE_WGRAD_WINOGRAD: 0 = direct kernel everywhere, 2 = F(2x2,3x3) wherever legal, anything else = F(4x4,3x3) where it is
// legal (maps that are multiples of 8), F(2x2,3x3) for the rest
int mode() {                                       // (read per call: a host-side getenv, tests switch kernels in one process)
  const char* e = getenv("SSDE_WGRAD_WINOGRAD");
  const int v = e ? atoi(e) : 4;
  return (v == 0 || v == 2) ? v : 4;
}

int plan(const ssde_wgrad_args* a, Ww4Params* p) {
  const ssde_src& s = a->src;
  p->src = s; p->g = a->g; p->g_ld = a->g_ld; p->g_off = a->g_off;
  p->N = a->n; p->H = a->h_out; p->W = a->w_out; p->Cout = a->c_out; p->Ctot = s.c0 + s.c1;
  p->cx = a->w_out / 8; p->cy = a->h_out / 8;
  p->chunks = a->n * p->cx * p->cy;
  p->co_tiles = a->c_out / kCo; p->ci_tiles = p->Ctot / kCi;
  const int ntiles = p->co_tiles * p->ci_tiles;
  // one workgroup fills a CU: ONE round of at most 256 workgroups, at least 4 stages each
  int splits = 256 / ntiles;
  const int max_splits = p->chunks >= 4 ? p->chunks / 4 : 1;
  if (splits > max_splits) splits = max_splits;
  if (a->splits > 0) splits = a->splits;
  if (splits > p->chunks) splits = p->chunks;
  if (splits < 1) splits = 1;
  p->chunks_per_split = ssde_cdiv(p->chunks, splits);
  p->splits = ssde_cdiv(p->chunks, p->chunks_per_split);
  p->scale = a->scale; p->dw = a->dw; p->scratch = a->scratch;
  return SSDE_OK;
}

}  // namespace

// ssde_conv_wgrad / ssde_wgrad_scratch_floats (wgrad.hip) route eligible launches here, before wgrad_wino.hip
bool ssde_wgrad_wino4_wants(const ssde_wgrad_args* a) {
  if (mode() != 4 || a->ksize != 3 || a->stride != 1 || a->pad != 1 || a->transpose_out) return false;
  if (a->h_in != a->h_out || a->w_in != a->w_out || a->h_out % 8 != 0 || a->w_out % 8 != 0) return false;
  const ssde_src& s = a->src;
  const int Ctot = s.c0 + s.c1;
  if (a->c_out % kCo != 0 || Ctot % kCi != 0 || a->cin_store != Ctot || (s.c1 > 0 && s.c0 % kCi != 0)) return false;
  if (a->g_ld % 4 != 0 || a->g_off % 4 != 0) return false;
  if ((a->c_out / kCo) * (Ctot / kCi) > 256) return false;            // (more blocks than CUs: the F(2x2,3x3) kernel's 64 x 64 blocks)
  return (long long)a->n * a->h_out * a->w_out < (1ll << 30);
}

int64_t ssde_wgrad_wino4_scratch_floats(const ssde_wgrad_args* a) {
  Ww4Params p;
  ssde_wgrad_args b = *a;
  b.splits = 0;
  plan(&b, &p);
  return (int64_t)p.splits * p.co_tiles * p.ci_tiles * kSlab;
}

int ssde_wgrad_wino4_launch(const ssde_wgrad_args* a, void* stream) {
  const ssde_src& s = a->src;
  SSDE_REQUIRE(a->g && a->dw && s.p0 && (s.c1 == 0 || s.p1), "wgrad(winograd 4x4): null tensors");
  const bool gn = s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU;
  if (gn) {
    SSDE_REQUIRE(s.gn_groups > 0 && (s.c0 + s.c1) % s.gn_groups == 0 && ((s.c0 + s.c1) / s.gn_groups) % 4 == 0,
                 "wgrad(winograd 4x4): GroupNorm channels-per-group %% 4");
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "wgrad(winograd 4x4): GroupNorm pointers missing");
  }
  SSDE_REQUIRE(s.drop_thresh == 0 || s.drop_seed, "wgrad(winograd 4x4): dropout seed pointer missing");
  Ww4Params p;
  plan(a, &p);
  const int64_t need = (int64_t)p.splits * p.co_tiles * p.ci_tiles * kSlab;
  if (a->scratch_floats < need) {          // fewer, longer workgroups when scratch is short
    const int fit = (int)(a->scratch_floats / ((int64_t)p.co_tiles * p.ci_tiles * kSlab));
    SSDE_REQUIRE(fit >= 1 && a->scratch, "wgrad(winograd 4x4): scratch of %lld floats needed at least", (long long)p.co_tiles * p.ci_tiles * kSlab);
    p.chunks_per_split = ssde_cdiv(p.chunks, fit);
    p.splits = ssde_cdiv(p.chunks, p.chunks_per_split);
  }
  SSDE_REQUIRE(a->scratch, "wgrad(winograd 4x4): scratch missing");
  static std::atomic<bool> attr_set{false};   // once, before any stream capture
  if (!attr_set) {
    SSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_wino4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_wino4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ntiles = p.co_tiles * p.ci_tiles;
  const dim3 grid(ssde_cdiv(p.splits * ntiles, 8) * 8);
  if (gn) hipLaunchKernelGGL(wgrad_wino4_kernel<true>, grid, dim3(kThreads), kLds, st, p);
  else hipLaunchKernelGGL(wgrad_wino4_kernel<false>, grid, dim3(kThreads), kLds, st, p);
  SSDE_LAUNCH_CHECK();
  int blocks = ntiles * 8;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wgrad_wino4_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
