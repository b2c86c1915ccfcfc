// The input-transform pass of the two-kernel F(4x4,3x3) convolution (SSDE_TILE_WINOGRAD4R):
//   wino4_xform_vq_kernel (this file)        V[pos][Cin/4][t][4] = B^T pro(x) B      HBM-bound: reads x once, writes 2.25x of it
//   conv_wino4r_kernel (conv_wino4r.hip)     Y = A^T [ sum_ci U .* V ] A             the matrix kernel: no prologue, no input transform
// conv_wino4.hip does both in one kernel, and on gfx950 that costs matrix time twice over: the GroupNorm / SiLU prologue and the
// two transform passes are VALU work that SERIALISES with fp32 MFMAs on a SIMD (~950 of a stage's ~4600 cycles, beside 2304
// matrix cycles), and every 64-cout workgroup of a pixel tile repeats them.  The price of the split is the extra pass: x read
// once more and V (2.25 x) written and read; it is paid back where a V tile feeds four or more cout tiles (the 256-cout layers)
// and in training programs, where V is wanted anyway by the F(4x4,3x3) weight gradient (ssde_wgrad_args.v_pre).
#include "ssde_common.h"

namespace {

// ---- the input-transform pass: V[pos][Q][t][4] = B^T pro(x) B, Q = Ctot / 4 channel quads, t = (img * tiles_h + ty) * tiles_w + tx.
// One thread = one 6x6 input tile of one channel quad: 36 float4 loads (the prologue -- GroupNorm, SiLU, dropout -- applied once
// per element and tile; out-of-image pixels are zeros of the ACTIVATED tensor), both 1-D passes in registers, 36 float4 stores.
// A wave = 16 consecutive tiles x 4 consecutive quads, lane = quad * 16 + tile: a store instruction writes four 256-byte runs,
// a load instruction reads 64 contiguous bytes (4 quads) of 16 pixels.  HBM-bound: x read ~2.25x through the tile overlap (L2),
// V written once.
// GroupNorm statistics merged by the pass itself (ssde_conv_args.gn_in_part0, ABI 10): a workgroup covers 16 tiles -- part of
// one image, or up to four whole ones on the 8x8 maps -- and 16 channel quads, i.e. at most 4 x 16 (image, group) pairs.  A
// merging workgroup does it FIRST: its sixteen teams of 16 lanes merge the producers' partials of those pairs (ssde_gn_merge16,
// the finalize kernel's function: the same bits) into an LDS table -- one trip of ~0.5 us on the 16x16 and 32x32 maps --, then
// every thread requests its 36 pixels and reads its (mean, rstd) from the table behind one LDS barrier, the pixels in flight.
// (Round 6's first form requested the partials before the pixels and merged while the pixels were in flight: a team's twelve
// partials and its index arithmetic were live across the 144 registers of pixels, the kernel went from 2 to 41 spilled
// registers, and the pass paid back what the launches had cost -- profiles/r6_gn_merge_in_transform_pass_ab.txt.  A fifth wave
// that only merges would halve the occupancy: five waves of 256 registers do not fit a CU twice.)  The workgroup whose tiles hold
// an image's first tile also writes the
// statistics to src.gn_mean / gn_rstd for the backward pass of a training program.  The merge costs a few KB of L2 reads per
// workgroup (147 KB in, 147 KB out otherwise); what it removes is a launch of 6-9 us in front of every pass (59 of the 95 per
// U-Net evaluation).
struct XformVqParams {
  ssde_src src; float* v;
  int N, H, W, Ctot, T, tiles_h, tiles_w;
  const float* part0; const float* part1;      // part0 != nullptr: merge here
  int s0, s1; float eps;
};
constexpr int kXfPairs = 64;                   // (image, group) pairs of a workgroup, at most

__device__ __forceinline__ void bt6q(const float4 (&d)[6], float4 (&o)[6]) {
#define SSDE_BT6_LANE(c)                                                                                             \
  {                                                                                                                  \
    const float t1 = d[4].c - 4.f * d[2].c, t2 = d[3].c - 4.f * d[1].c, t3 = d[4].c - d[2].c, t4 = d[3].c - d[1].c;   \
    o[0].c = 4.f * d[0].c - 5.f * d[2].c + d[4].c;                                                                    \
    o[1].c = t1 + t2; o[2].c = t1 - t2; o[3].c = t3 + 2.f * t4; o[4].c = t3 - 2.f * t4;                               \
    o[5].c = 4.f * d[1].c - 5.f * d[3].c + d[5].c;                                                                    \
  }
  SSDE_BT6_LANE(x) SSDE_BT6_LANE(y) SSDE_BT6_LANE(z) SSDE_BT6_LANE(w)
#undef SSDE_BT6_LANE
}

constexpr int kXfThreads = 256;

template <bool kGn, bool kMerge>
__global__ __launch_bounds__(kXfThreads, 2) void wino4_xform_vq_kernel(const XformVqParams p) {
  static_assert(kGn || !kMerge, "only a GroupNorm prologue has statistics to merge");
  const ssde_src& s = p.src;
  SsdePro pro = ssde_pro_decode(s);
  pro.gn = kGn;
  const int Q = p.Ctot >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per_img = p.tiles_h * p.tiles_w;
  const int cpg = kGn ? p.Ctot / s.gn_groups : 4;     // (cpg % 4 == 0: a quad lies in one group)
  // the pairs of this workgroup: images of tiles 16 bx .. 16 bx + 15, groups of quads 16 by .. 16 by + 15
  const int t0 = blockIdx.x * 16, t1 = min(t0 + 15, p.T - 1);
  const int img_lo = t0 / per_img;
  const int qa = blockIdx.y * 16, qb = min(qa + 15, Q - 1);
  const int g_lo = (qa * 4) / cpg, ng = (qb * 4) / cpg - g_lo + 1;
  if constexpr (kMerge) {
    SSDE_LDS(tab);
    const int team = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    const int npairs = (t1 / per_img - img_lo + 1) * ng;            // <= kXfPairs
    for (int pr0 = 0; pr0 < npairs; pr0 += 16) {                    // (uniform trip count; a team beyond the end repeats the last pair)
      const int pr = min(pr0 + team, npairs - 1);
      const int il = pr / ng;
      const int n = img_lo + il, g = g_lo + (pr - il * ng);
      float cnt, m, M2;
      ssde_gn_merge16(p.part0, p.part1, s.c0, s.c1, p.s0, p.s1, s.gn_groups, n, g, l16, cnt, m, M2);
      if (l16 == 0 && pr0 + team < npairs) {
        const float r = ssde_gn_rstd(cnt, M2, p.eps);
        tab[2 * pr] = m; tab[2 * pr + 1] = r;
        if (n * per_img >= t0) {                      // the image's first tile is one of this workgroup's
          const_cast<float*>(s.gn_mean)[n * s.gn_groups + g] = m;
          const_cast<float*>(s.gn_rstd)[n * s.gn_groups + g] = r;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);                // (the pixel loads stay behind the merge: nothing of it is live across them)
  }
  const int t_raw = blockIdx.x * 16 + (lane & 15);
  const int q_raw = (blockIdx.y * 4 + wave) * 4 + (lane >> 4);
  const bool valid = t_raw < p.T && q_raw < Q;
  if (!kMerge && !valid) return;                  // (the waves of a merging workgroup all reach its barrier)
  const int t = valid ? t_raw : 0, q = valid ? q_raw : 0;
  const int c = q * 4;
  const int img = t / per_img, r = t - img * per_img;
  const int ty = r / p.tiles_w, tx = r - ty * p.tiles_w;
  const bool second = c >= s.c0;                      // (c0 % 4 == 0: a quad never straddles the sources)
  const float* base = second ? s.p1 + (c - s.c0) : s.p0 + c;
  const int C = second ? s.c1 : s.c0;
  float mu = 0.f, rs = 1.f;
  float4 ga = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kGn) {
    if (!kMerge) {
      mu = s.gn_mean[img * s.gn_groups + c / cpg];
      rs = s.gn_rstd[img * s.gn_groups + c / cpg];
    }
    ga = *reinterpret_cast<const float4*>(s.gn_gamma + c);
    be = *reinterpret_cast<const float4*>(s.gn_beta + c);
  }
  float4 v[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int iy = ty * 4 - 1 + a, ix = tx * 4 - 1 + b;
      const bool inb = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int pix = (img * p.H + (inb ? iy : 0)) * p.W + (inb ? ix : 0);
      v[a][b] = *reinterpret_cast<const float4*>(base + (size_t)pix * C);
    }
  if constexpr (kMerge) {
    SSDE_LDS(tab);
    SSDE_LDS_BARRIER();                               // (LDS only: the pixel loads stay in flight)
    if (!valid) return;
    const int pr = (img - img_lo) * ng + (c / cpg - g_lo);
    mu = tab[2 * pr]; rs = tab[2 * pr + 1];
  }
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int iy = ty * 4 - 1 + a, ix = tx * 4 - 1 + b;
      const bool inb = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int pix = (img * p.H + (inb ? iy : 0)) * p.W + (inb ? ix : 0);
      const float4 x = ssde_pro_apply(v[a][b], mu, rs, ga, be, (uint32_t)pix * (uint32_t)p.Ctot + (uint32_t)c, pro);
      v[a][b] = inb ? x : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  // columns (over the rows a of every column b), then rows
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    float4 d[6], o[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) d[a] = v[a][b];
    bt6q(d, o);
#pragma unroll
    for (int a = 0; a < 6; ++a) v[a][b] = o[a];
  }
  float* dst = p.v + ((size_t)q * p.T + t) * 4;
  const size_t plane = (size_t)Q * p.T * 4;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float4 o[6];
    bt6q(v[a], o);
#pragma unroll
    for (int b = 0; b < 6; ++b) *reinterpret_cast<float4*>(dst + (size_t)(a * 6 + b) * plane) = o[b];
  }
}

}  // namespace

// does the pass of this launch merge the GroupNorm partials of its source itself (ssde_conv_args.gn_in_part0)?  Otherwise
// ssde_conv2d issues the finalize launch in front of it.  16 tiles of a workgroup must not span more than four images.
bool ssde_wino4_xform_merges_gn(const ssde_conv_args* a) {
  const ssde_src& s = a->main;
  const bool gn = s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU;
  if (!gn || !a->gn_in_part0 || a->tile != SSDE_TILE_WINOGRAD4R || (a->flags & SSDE_CONVF_V_GIVEN)) return false;
  if (a->gn_in_slices0 <= 0 || (s.c1 > 0 && (!a->gn_in_part1 || a->gn_in_slices1 <= 0))) return false;
  const int per_img = (a->h_in / 4) * (a->w_in / 4);
  if (s.gn_groups > 0 && ssde_gn_group_is_big(s.c0, s.c1, a->gn_in_slices0, s.c1 > 0 ? a->gn_in_slices1 : 0, s.gn_groups)) return false;
  return per_img >= 4 && s.gn_groups > 0 && s.gn_mean && s.gn_rstd;
}

int ssde_wino4_xform_vq_launch(const ssde_conv_args* a, void* stream) {
  SSDE_REQUIRE(a && a->main.p0 && a->wino_v, "conv(winograd 4x4, two kernels): the transformed-input buffer (ssde_conv_args.wino_v) is missing");
  const ssde_src& s = a->main;
  SSDE_REQUIRE(s.c0 > 0 && s.c0 % 4 == 0 && s.c1 % 4 == 0 && (s.c1 == 0 || s.p1), "conv(winograd 4x4, two kernels): channels must be multiples of 4");
  SSDE_REQUIRE(a->h_in % 4 == 0 && a->w_in % 4 == 0 && a->n > 0, "conv(winograd 4x4, two kernels): bad shape");
  const bool gn = s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU;
  if (gn) {
    SSDE_REQUIRE(s.gn_groups > 0 && (s.c0 + s.c1) % s.gn_groups == 0 && ((s.c0 + s.c1) / s.gn_groups) % 4 == 0,
                 "conv(winograd 4x4, two kernels): GroupNorm needs channels-per-group %% 4 == 0");
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "conv(winograd 4x4, two kernels): GroupNorm pointers missing");
  }
  SSDE_REQUIRE(s.drop_thresh == 0 || s.drop_seed, "conv(winograd 4x4, two kernels): dropout seed pointer missing");
  XformVqParams p{s, a->wino_v, a->n, a->h_in, a->w_in, s.c0 + s.c1, a->n * (a->h_in / 4) * (a->w_in / 4), a->h_in / 4, a->w_in / 4,
                  nullptr, nullptr, 0, 0, 0.f};
  SSDE_REQUIRE((unsigned long long)a->n * a->h_in * a->w_in * (unsigned)p.Ctot < (1ull << 32), "conv(winograd 4x4, two kernels): tensor too large");
  if (ssde_wino4_xform_merges_gn(a)) {
    p.part0 = a->gn_in_part0; p.part1 = s.c1 > 0 ? a->gn_in_part1 : nullptr;
    p.s0 = a->gn_in_slices0; p.s1 = s.c1 > 0 ? a->gn_in_slices1 : 0; p.eps = a->gn_in_eps;
  }
  const dim3 grid(ssde_cdiv(p.T, 16), ssde_cdiv(p.Ctot >> 2, 16));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (gn && p.part0) hipLaunchKernelGGL((wino4_xform_vq_kernel<true, true>), grid, dim3(kXfThreads), kXfPairs * 2 * sizeof(float), st, p);
  else if (gn) hipLaunchKernelGGL((wino4_xform_vq_kernel<true, false>), grid, dim3(kXfThreads), 0, st, p);
  else hipLaunchKernelGGL((wino4_xform_vq_kernel<false, false>), grid, dim3(kXfThreads), 0, st, p);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
