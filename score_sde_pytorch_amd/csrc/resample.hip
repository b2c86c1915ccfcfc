// upfirdn2d for NHWC fp32: zero-insert upsample, zero pad, 2-D FIR (true
// convolution, i.e. correlation with the flipped kernel), decimate.
// Reference semantics: upfirdn2d_native (op/upfirdn2d.py:159-200), which the CUDA
// kernel op/upfirdn2d_kernel.cu:107-207 reproduces; callers upsample_2d /
// downsample_2d / conv_downsample_2d (models/up_or_down_sampling.py:144-257) and
// the 2x2 box filters of naive_upsample_2d / naive_downsample_2d (:59-69).
// The optional GroupNorm(+SiLU) prologue implements "h = act(GroupNorm_0(x))"
// of an up/down ResnetBlockBigGANpp (layerspp.py:243-258) without writing h.
//
// HBM-bound: algorithmic traffic = (in + out) * 4 B.  Two kernels:
//  * upfirdn_tile_kernel<UP, DOWN> -- the three shapes the networks use with the 4x4 FIR (up 2 / pad (2,1),
//    down 2 / pad (1,1), pad (2,2)) and their gradients (the same shapes with up and down swapped): one workgroup owns
//    an output tile x 32 channels, stages the input tile it needs ONCE into LDS (zero padding resolved there, the
//    GroupNorm / SiLU prologue applied once per input element instead of once per tap: an input pixel feeds 4 outputs
//    when upsampling and 4..16 taps when downsampling), then every tap is an LDS read with compile-time parity logic.
//    A residual block resamples BOTH act(GroupNorm(x)) and x (layerspp.py:250-258): with `dst2` the raw tile is
//    filtered first, then transformed in place in LDS and filtered again -- x is read from HBM once for both outputs.
//  * upfirdn_kernel -- the general form (any up / down / pads, kernels up to 4x4, channel counts that are not a
//    multiple of 32 such as the 4-channel image pyramids): one lane = 4 channels of one output pixel.
#include "ssde_common.h"

namespace {

struct FirParams {
  ssde_src src;
  int n, h_in, w_in, c, h_out, w_out, up, down, pad0, pad1, kh, kw;
  float kf[16];   // flipped kernel
  float* dst;
  int accumulate;
};

__global__ __launch_bounds__(256) void upfirdn_kernel(const FirParams p) {
  const int c4n = p.c >> 2;
  const size_t total = (size_t)p.n * p.h_out * p.w_out * c4n;
  const SsdePro pro = ssde_pro_decode(p.src);
  const bool use_gn = pro.gn;
  const int cpg = use_gn ? p.c / p.src.gn_groups : 1;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    size_t t = idx / c4n;
    const int ox = (int)(t % p.w_out); t /= p.w_out;
    const int oy = (int)(t % p.h_out);
    const int n = (int)(t / p.h_out);
    const int ch = c4 * 4;
    float mu = 0.f, rs = 1.f;
    float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
    if (use_gn) {
      const int g = ch / cpg;
      mu = p.src.gn_mean[n * p.src.gn_groups + g];
      rs = p.src.gn_rstd[n * p.src.gn_groups + g];
      gam = *reinterpret_cast<const float4*>(p.src.gn_gamma + ch);
      bet = *reinterpret_cast<const float4*>(p.src.gn_beta + ch);
    }
    const float* xin = p.src.p0 + (size_t)n * p.h_in * p.w_in * p.c + ch;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < p.kh; ++ky) {
      const int uy = oy * p.down + ky - p.pad0;
      if (uy < 0 || uy % p.up != 0) continue;
      const int iy = uy / p.up;
      if (iy >= p.h_in) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        const int ux = ox * p.down + kx - p.pad0;
        if (ux < 0 || ux % p.up != 0) continue;
        const int ix = ux / p.up;
        if (ix >= p.w_in) continue;
        float4 v = *reinterpret_cast<const float4*>(xin + ((size_t)iy * p.w_in + ix) * p.c);
        v = ssde_pro_apply(v, mu, rs, gam, bet,
                           (uint32_t)(((size_t)n * p.h_in + iy) * p.w_in + ix) * (uint32_t)p.c + (uint32_t)ch, pro);
        const float w = p.kf[ky * p.kw + kx];
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
      }
    }
    float4* o = reinterpret_cast<float4*>(p.dst + (((size_t)n * p.h_out + oy) * p.w_out + ox) * p.c + ch);
    if (p.accumulate) { const float4 old = *o; acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w; }
    *o = acc;
  }
}


#ifndef SSDE_FIR_STAGE_U
#define SSDE_FIR_STAGE_U 4       // (1 = one load in flight per thread, the form of rounds 2-5: A/B variant only)
#endif

// ---- LDS-staged tiles for the 4x4 FIR -----------------------------------------------------------------------------
constexpr int kFirCh = 32;        // channels per workgroup: 8 lanes x float4 = one 128-byte line per pixel
struct FirTileParams {
  ssde_src src;
  int n, h_in, w_in, c, h_out, w_out, pad0;
  int lth, ltw;                   // log2 of the output tile (th x tw outputs)
  int ih, iw;                     // staged input tile
  int tiles_x, tiles_y, cchunks;
  float kf[16];                   // flipped kernel
  float* dst; float* dst2;
  int accumulate;
};

template <int UP, int DOWN>
__global__ __launch_bounds__(256) void upfirdn_tile_kernel(const FirTileParams p) {
  SSDE_LDS(tile);                 // [ih * iw][32 channels]; then 16 weights re-ordered by tap parity
  const int tid = threadIdx.x, c4 = tid & 7, slot = tid >> 3;
  int b = blockIdx.x;
  const int cc = b % p.cchunks; b /= p.cchunks;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int n = b / p.tiles_y;
  const int TH = 1 << p.lth, TW = 1 << p.ltw;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int ch = cc * kFirCh + c4 * 4;
  // first staged input row / column: floor((o0 * DOWN - pad0) / UP)
  const int uy0 = oy0 * DOWN - p.pad0, ux0 = ox0 * DOWN - p.pad0;
  const int iy0 = UP == 2 ? (uy0 >> 1) : uy0, ix0 = UP == 2 ? (ux0 >> 1) : ux0;
  const int npix = p.ih * p.iw;
  float* wtab = tile + npix * kFirCh;      // [ky & 1][kx & 1][ky >> 1][kx >> 1] for UP == 2
  if (UP == 2 && tid < 16) {
    const int ky = ((tid >> 3) & 1) + 2 * ((tid >> 1) & 1), kx = ((tid >> 2) & 1) + 2 * (tid & 1);
    wtab[tid] = p.kf[ky * 4 + kx];
  }
  const float* xin = p.src.p0 + (size_t)n * p.h_in * p.w_in * p.c + ch;
  // ---- stage the raw input tile (zeros outside the image) ----
  // kStageU loads of a thread are issued before the first of them is parked: with one load in flight per thread (a loop the
  // compiler cannot unroll: the tile size is a launch argument) the 4-12 round trips of a tile were paid one after the other
  // and the pass sat at 0.3-0.5 of the HBM peak on every shape but the largest (profiles/r6_fir_staging_unroll.txt)
  constexpr int kStageU = SSDE_FIR_STAGE_U;
  for (int base = slot; base < npix; base += 32 * kStageU) {
    float4 v[kStageU];
#pragma unroll
    for (int u = 0; u < kStageU; ++u) {
      const int pix = base + 32 * u;
      const int ly = pix / p.iw, lx = pix - ly * p.iw;
      const int iy = iy0 + ly, ix = ix0 + lx;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pix < npix && iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in)
        v[u] = *reinterpret_cast<const float4*>(xin + ((size_t)iy * p.w_in + ix) * p.c);
    }
#pragma unroll
    for (int u = 0; u < kStageU; ++u) {
      const int pix = base + 32 * u;
      if (pix < npix) *reinterpret_cast<float4*>(tile + pix * kFirCh + c4 * 4) = v[u];
    }
  }
  __syncthreads();

  auto filter = [&](float* out) {
    for (int op = slot; op < TH * TW; op += 32) {
      const int oy = oy0 + (op >> p.ltw), ox = ox0 + (op & (TW - 1));
      if (oy >= p.h_out || ox >= p.w_out) continue;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (UP == 2) {
        // u = o * DOWN + k - pad0 must be even: k = k0, k0 + 2 with k0 = (pad0 - o * DOWN) & 1; input index u >> 1
        const int ky0 = (p.pad0 - oy * DOWN) & 1, kx0 = (p.pad0 - ox * DOWN) & 1;
        const int ly = ((oy * DOWN + ky0 - p.pad0) >> 1) - iy0, lx = ((ox * DOWN + kx0 - p.pad0) >> 1) - ix0;
        const float4 w = *reinterpret_cast<const float4*>(wtab + (ky0 * 2 + kx0) * 4);
        const float* t0 = tile + (ly * p.iw + lx) * kFirCh + c4 * 4;
        const float4 v00 = *reinterpret_cast<const float4*>(t0);
        const float4 v01 = *reinterpret_cast<const float4*>(t0 + kFirCh);
        const float4 v10 = *reinterpret_cast<const float4*>(t0 + p.iw * kFirCh);
        const float4 v11 = *reinterpret_cast<const float4*>(t0 + p.iw * kFirCh + kFirCh);
        acc.x = w.x * v00.x + w.y * v01.x + w.z * v10.x + w.w * v11.x;
        acc.y = w.x * v00.y + w.y * v01.y + w.z * v10.y + w.w * v11.y;
        acc.z = w.x * v00.z + w.y * v01.z + w.z * v10.z + w.w * v11.z;
        acc.w = w.x * v00.w + w.y * v01.w + w.z * v10.w + w.w * v11.w;
      } else {
        const int ly = oy * DOWN - p.pad0 - iy0, lx = ox * DOWN - p.pad0 - ix0;
        const float* t0 = tile + (ly * p.iw + lx) * kFirCh + c4 * 4;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
          for (int kx = 0; kx < 4; ++kx) {
            const float4 v = *reinterpret_cast<const float4*>(t0 + (ky * p.iw + kx) * kFirCh);
            const float w = p.kf[ky * 4 + kx];
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
          }
      }
      float4* o = reinterpret_cast<float4*>(out + (((size_t)n * p.h_out + oy) * p.w_out + ox) * p.c + ch);
      if (p.accumulate) { const float4 old = *o; acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w; }
      *o = acc;
    }
  };

  const SsdePro pro = ssde_pro_decode(p.src);
  const bool has_pro = pro.gn || pro.silu || pro.drop;
  if (p.dst2) {                    // the un-activated source, filtered with the same taps
    filter(p.dst2);
    __syncthreads();
  }
  if (has_pro) {
    // in place: every thread transforms exactly the elements it staged; padding stays zero (upfirdn2d pads AFTER the
    // activation: the reference filters act(GroupNorm(x)) as its own tensor)
    float mu = 0.f, rs = 1.f;
    float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pro.gn) {
      const int g = ch / (p.c / p.src.gn_groups);
      mu = p.src.gn_mean[n * p.src.gn_groups + g];
      rs = p.src.gn_rstd[n * p.src.gn_groups + g];
      gam = *reinterpret_cast<const float4*>(p.src.gn_gamma + ch);
      bet = *reinterpret_cast<const float4*>(p.src.gn_beta + ch);
    }
    for (int pix = slot; pix < npix; pix += 32) {
      const int ly = pix / p.iw, lx = pix - ly * p.iw;
      const int iy = iy0 + ly, ix = ix0 + lx;
      if (iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in) {
        float4* e = reinterpret_cast<float4*>(tile + pix * kFirCh + c4 * 4);
        *e = ssde_pro_apply(*e, mu, rs, gam, bet,
                            (uint32_t)(((size_t)n * p.h_in + iy) * p.w_in + ix) * (uint32_t)p.c + (uint32_t)ch, pro);
      }
    }
    __syncthreads();
  }
  filter(p.dst);
}

int pow2_ceil(int v) { int q = 1; while (q < v) q *= 2; return q; }

}  // namespace

extern "C" int ssde_upfirdn2d(const ssde_upfirdn_args* a, void* stream) {
  SSDE_REQUIRE(a && a->src.p0 && a->dst, "upfirdn2d: null args");
  SSDE_REQUIRE(a->src.p1 == nullptr && a->src.c1 == 0, "upfirdn2d: concatenated source not supported");
  SSDE_REQUIRE(a->c > 0 && a->c % 4 == 0 && a->src.c0 == a->c, "upfirdn2d: channels must be a multiple of 4 and match src.c0");
  SSDE_REQUIRE(a->kh >= 1 && a->kh <= 4 && a->kw >= 1 && a->kw <= 4, "upfirdn2d: kernel larger than 4x4");
  SSDE_REQUIRE(a->up >= 1 && a->down >= 1, "upfirdn2d: bad up/down");
  // output size exactly as upfirdn2d_native computes it (op/upfirdn2d.py:196-197)
  const int eh = (a->h_in * a->up + a->pad0 + a->pad1 - a->kh) / a->down + 1;
  const int ew = (a->w_in * a->up + a->pad0 + a->pad1 - a->kw) / a->down + 1;
  SSDE_REQUIRE(eh == a->h_out && ew == a->w_out, "upfirdn2d: output must be %dx%d (got %dx%d)", eh, ew, a->h_out, a->w_out);
  SSDE_REQUIRE(a->pad0 >= 0 && a->pad1 >= 0, "upfirdn2d: negative pads unsupported");
  if (a->src.pro_mode == SSDE_PRO_GN || a->src.pro_mode == SSDE_PRO_GN_SILU) {
    SSDE_REQUIRE(a->src.gn_groups > 0 && a->c % a->src.gn_groups == 0 && (a->c / a->src.gn_groups) % 4 == 0,
                 "upfirdn2d: GroupNorm channels-per-group must be a multiple of 4");
    SSDE_REQUIRE(a->src.gn_mean && a->src.gn_rstd && a->src.gn_gamma && a->src.gn_beta, "upfirdn2d: GroupNorm pointers missing");
  }
  float kf[16] = {0.f};
  for (int y = 0; y < a->kh; ++y)
    for (int x = 0; x < a->kw; ++x) kf[y * a->kw + x] = a->k[(a->kh - 1 - y) * a->kw + (a->kw - 1 - x)];
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool tiled = a->kh == 4 && a->kw == 4 && a->c % kFirCh == 0 && a->up <= 2 && a->down <= 2 &&
                     !(a->up == 2 && a->down == 2);
  if (tiled) {
    FirTileParams p;
    p.src = a->src;
    p.n = a->n; p.h_in = a->h_in; p.w_in = a->w_in; p.c = a->c; p.h_out = a->h_out; p.w_out = a->w_out; p.pad0 = a->pad0;
    const int want = a->up == 2 ? 16 : 8;             // 16x16 outputs from an 11x11 input tile, or 8x8 from 19x19 / 12x12
    const int th = pow2_ceil(a->h_out < want ? a->h_out : want), tw = pow2_ceil(a->w_out < want ? a->w_out : want);
    p.lth = ssde_ilog2(th); p.ltw = ssde_ilog2(tw);
    p.ih = ((th - 1) * a->down + 3) / a->up + 2;
    p.iw = ((tw - 1) * a->down + 3) / a->up + 2;
    p.tiles_x = ssde_cdiv(a->w_out, tw); p.tiles_y = ssde_cdiv(a->h_out, th); p.cchunks = a->c / kFirCh;
    for (int i = 0; i < 16; ++i) p.kf[i] = kf[i];
    p.dst = a->dst; p.dst2 = a->dst2; p.accumulate = a->accumulate;
    const int lds = (p.ih * p.iw * kFirCh + 16) * 4;
    const dim3 grid((unsigned)((size_t)a->n * p.tiles_y * p.tiles_x * p.cchunks));
    if (a->up == 2) hipLaunchKernelGGL((upfirdn_tile_kernel<2, 1>), grid, dim3(256), lds, st, p);
    else if (a->down == 2) hipLaunchKernelGGL((upfirdn_tile_kernel<1, 2>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((upfirdn_tile_kernel<1, 1>), grid, dim3(256), lds, st, p);
    SSDE_LAUNCH_CHECK();
    return SSDE_OK;
  }
  FirParams p;
  p.src = a->src;
  p.n = a->n; p.h_in = a->h_in; p.w_in = a->w_in; p.c = a->c; p.h_out = a->h_out; p.w_out = a->w_out;
  p.up = a->up; p.down = a->down; p.pad0 = a->pad0; p.pad1 = a->pad1; p.kh = a->kh; p.kw = a->kw;
  for (int i = 0; i < 16; ++i) p.kf[i] = kf[i];
  p.accumulate = a->accumulate;
  const size_t total = (size_t)a->n * a->h_out * a->w_out * (a->c / 4);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (a->dst2) {                  // general form: the raw source is a second pass
    FirParams q = p;
    q.src.pro_mode = SSDE_PRO_NONE; q.src.drop_thresh = 0; q.dst = a->dst2;
    hipLaunchKernelGGL(upfirdn_kernel, dim3((unsigned)blocks), dim3(256), 0, st, q);
    SSDE_LAUNCH_CHECK();
  }
  p.dst = a->dst;
  hipLaunchKernelGGL(upfirdn_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
