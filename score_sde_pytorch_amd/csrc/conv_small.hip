// 3x3 / stride 1 / pad 1 convolutions with at most FOUR output channels: the image heads of the network --
// `conv3x3(act(GroupNorm(h)))` onto the 3 image channels at models/ncsnpp.py:368-375 (every configuration) and the
// output-skip pyramid convolutions of the progressive architectures (ncsnpp.py:329-337: FFHQ / CelebA-HQ / LSUN NCSN++, one per
// resolution level up to 256x256 and 1024x1024).
//
// The matrix kernels tile output channels by 32 (conv_mfma.hip, `v_mfma_f32_32x32x2_f32`), so these layers ran with 7/8 of
// their MFMAs on zero padding: 128 -> 4 @ 32x32 at batch 256 took 0.21 ms = 11.5 TFLOP/s of useful work, the time of a
// 128 -> 128 Winograd layer for 1/32 of its FLOPs.  With 4 couts there is no operand reuse for a matrix instruction to exploit
// on the weight side; the natural machine is the vector ALU (v_pk_fma_f32: two couts per instruction and lane, 128 FMA / clk /
// CU) and the bounds are the one read of the input (HBM) and the LDS reads that feed the FMAs:
//   workgroup = 256 threads = one 8x8 output tile of one image, all input channels in chunks of 128
//   LDS        halo tile [10 x 10 px][2 halves][16 slices][4 ch] -- the GroupNorm / SiLU / dropout prologue applied once per
//              element while staging, zeros outside the image -- and the chunk's weights [tap][slice][8 ci][4 co]
//              (slice pitch 36 floats: the 16 slices of a ds_read_b128 hit 16 distinct bank quads)
//   thread     = (8-channel slice s = lane & 15, strip of 4 pixels in a row): 16 accumulators (4 px x 4 co); per tap row it
//              reads 6 px x 8 ch once (12 ds_read_b128) and uses them for the three taps of the row, 32 weights per tap
//              (8 ds_read_b128, the same address for the lanes of a slice: broadcast) -> 1152 FMAs on 108 LDS reads
//   reduction  the 16 slices of a pixel are summed through LDS in slice order (fixed: deterministic), one thread per
//              (pixel, cout), then bias / per-image addend / residual / scale as everywhere else (ssde_store_tile's order)
// fp32 FMA chains throughout: closer to the direct sum than the Winograd kernels (tests: the direct kernel's tolerance).
#include "ssde_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 8, kHalo = kTile + 2, kHaloPx = kHalo * kHalo;   // 8x8 outputs, 10x10 inputs
constexpr int kChunk = 128, kSlices = kChunk / 8;                      // channels per LDS fill, 8-channel slices
constexpr int kWPitch = 36;                                            // floats per (tap, slice): 32 weights + 4 of padding
constexpr int kHaloFloats = kHaloPx * kChunk;                          // 12800
constexpr int kWFloats = 9 * kSlices * kWPitch;                        // 5184
constexpr int kGnFloats = 64 + 2 * kChunk;                             // (mean, rstd) per channel quad, gamma, beta of a chunk
constexpr int kLdsBytes = (kHaloFloats + kWFloats + kGnFloats) * 4;    // 73216: two workgroups per CU
static_assert(kSlices * kTile * kTile * 4 <= kHaloFloats, "the slice partials reuse the halo area");

struct SmallParams {
  ssde_src src;
  const float* wpk;        // SSDE_PACK_CONV3: [ceil(C/8)][9][CoutPad][8]
  int N, H, W, Cout, CoutPad;
  int tiles_x, tiles_y;
  const float* bias; const float* chan_add; int chan_add_ld;
  const float* resid; int resid_post;
  float scale;
  float* dst;
};

__global__ __launch_bounds__(kThreads, 2) void conv_small_cout_kernel(const SmallParams p) {
  SSDE_LDS(smem);
  float* halo = smem;                       // [px][half][slice][4]
  float* wl = smem + kHaloFloats;           // [tap][slice][ci 8][co 4] (+ 4)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int img = b / p.tiles_y;
  const int oy0 = ty * kTile, ox0 = tx * kTile;

  const ssde_src& s = p.src;
  const int Ctot = s.c0 + s.c1;
  const SsdePro pro = ssde_pro_decode(s);
  const int cpg = pro.gn ? Ctot / s.gn_groups : 1;

  // compute role: slice sl of the chunk, pixels (row, col0 .. col0 + 3) of the tile
  const int sl = lane & 15, strip = wave * 4 + (lane >> 4);
  const int row = strip >> 1, col0 = (strip & 1) * 4;
  ssde_f32x2 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) { acc[j][0] = ssde_f32x2{0.f, 0.f}; acc[j][1] = ssde_f32x2{0.f, 0.f}; }

  constexpr int kItems = (kHaloPx * (kChunk / 4) + kThreads - 1) / kThreads;      // 13 halo float4s per thread and chunk
  float* gnl = wl + kWFloats;               // this chunk's GroupNorm tables: [32 quads](mean, rstd) | gamma[128] | beta[128]
  for (int c_base = 0; c_base < Ctot; c_base += kChunk) {
    if (c_base) __syncthreads();            // the previous chunk has been consumed
    // ---- GroupNorm parameters of the chunk -> LDS (one division per channel quad here instead of one per staged item)
    if (pro.gn) {
      const int c = c_base + (tid & 31) * 4;
      if (c < Ctot) {
        if (tid < 32) {
          const int gi = img * s.gn_groups + c / cpg;
          *reinterpret_cast<float2*>(gnl + 2 * tid) = make_float2(s.gn_mean[gi], s.gn_rstd[gi]);
        } else if (tid < 64) {
          *reinterpret_cast<float4*>(gnl + 64 + (tid & 31) * 4) = *reinterpret_cast<const float4*>(s.gn_gamma + c);
        } else if (tid < 96) {
          *reinterpret_cast<float4*>(gnl + 64 + kChunk + (tid & 31) * 4) = *reinterpret_cast<const float4*>(s.gn_beta + c);
        }
      }
    }
    // ---- the halo tile of this chunk: item = (halo pixel, channel quad), the 32 quads of a pixel are 512 contiguous bytes.
    // ALL loads of a thread are issued before the first one is used (branch-free: items outside the image / the channel range
    // read a valid address and are zeroed below); the first version issued one load per loop trip and waited for each
    float4 hv[kItems];
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const int it = tid + k * kThreads;
      const int hp = it >> 5, q = it & 31;
      const int hy = hp / kHalo, hx = hp - hy * kHalo;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      const int c = c_base + q * 4;
      const bool ok = it < kHaloPx * (kChunk / 4) && c < Ctot && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const size_t pix = ok ? ((size_t)img * p.H + iy) * p.W + ix : 0;
      const bool second = ok && c >= s.c0;                  // (c0 % 4 == 0: a quad never straddles the sources)
      hv[k] = *reinterpret_cast<const float4*>(second ? s.p1 + pix * s.c1 + (c - s.c0) : s.p0 + pix * s.c0 + (ok ? c : 0));
    }
    // ---- the chunk's weights: global [ci8 chunk][tap][CoutPad][8 ci] -> LDS [tap][slice][ci][co]
    for (int it = tid; it < 9 * kSlices * 8; it += kThreads) {
      const int ci = it & 7, ts = it >> 3;
      const int sx = ts % kSlices, tap = ts / kSlices;
      const int c8 = (c_base >> 3) + sx;
      float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c8 * 8 < Ctot) {
        const float* g = p.wpk + (((size_t)c8 * 9 + tap) * p.CoutPad) * 8 + ci;
        w4 = make_float4(g[0], g[8], g[16], g[24]);          // couts 0..3 (CoutPad >= 64: rows past Cout are zero)
      }
      *reinterpret_cast<float4*>(wl + (tap * kSlices + sx) * kWPitch + ci * 4) = w4;
    }
    if (pro.gn) __syncthreads();            // the GroupNorm tables are in place
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const int it = tid + k * kThreads;
      if (it >= kHaloPx * (kChunk / 4)) continue;
      const int hp = it >> 5, q = it & 31;
      const int hy = hp / kHalo, hx = hp - hy * kHalo;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      const int c = c_base + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < Ctot && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
        const size_t pix = ((size_t)img * p.H + iy) * p.W + ix;
        float mu = 0.f, rs = 1.f;
        float4 ga = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pro.gn) {
          const float2 mr = *reinterpret_cast<const float2*>(gnl + 2 * q);
          mu = mr.x; rs = mr.y;
          ga = *reinterpret_cast<const float4*>(gnl + 64 + q * 4);
          be = *reinterpret_cast<const float4*>(gnl + 64 + kChunk + q * 4);
        }
        v = ssde_pro_apply(hv[k], mu, rs, ga, be, (uint32_t)pix * (uint32_t)Ctot + (uint32_t)c, pro);
      }
      // channel c = slice * 8 + half * 4 + k  ->  [px][half][slice][k]
      *reinterpret_cast<float4*>(halo + hp * kChunk + ((q & 1) * kSlices + (q >> 1)) * 4) = v;
    }
    __syncthreads();
    // ---- 9 taps x 8 channels x 4 pixels x 4 couts
    if (c_base + sl * 8 < Ctot) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        float4 x[6][2];
        const float* hrow = halo + ((row + dy) * kHalo + col0) * kChunk + sl * 4;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          x[k][0] = *reinterpret_cast<const float4*>(hrow + k * kChunk);
          x[k][1] = *reinterpret_cast<const float4*>(hrow + k * kChunk + kSlices * 4);
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float* wt = wl + ((dy * 3 + dx) * kSlices + sl) * kWPitch;
          ssde_f32x2 w[8][2];
#pragma unroll
          for (int ci = 0; ci < 8; ++ci) {
            const float4 t = *reinterpret_cast<const float4*>(wt + ci * 4);
            w[ci][0] = ssde_f32x2{t.x, t.y};
            w[ci][1] = ssde_f32x2{t.z, t.w};
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xv[8] = {x[j + dx][0].x, x[j + dx][0].y, x[j + dx][0].z, x[j + dx][0].w,
                                 x[j + dx][1].x, x[j + dx][1].y, x[j + dx][1].z, x[j + dx][1].w};
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
              const ssde_f32x2 xx = ssde_f32x2{xv[ci], xv[ci]};
              acc[j][0] = __builtin_elementwise_fma(xx, w[ci][0], acc[j][0]);
              acc[j][1] = __builtin_elementwise_fma(xx, w[ci][1], acc[j][1]);
            }
          }
        }
      }
    }
  }
  // ---- sum the slices (fixed order), then the epilogue of one (pixel, cout) per thread
  __syncthreads();
  float* red = smem;                        // [slice][64 px][4]
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *reinterpret_cast<float4*>(red + (sl * 64 + row * kTile + col0 + j) * 4) = make_float4(acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y);
  __syncthreads();
  const int px = tid >> 2, co = tid & 3;
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < kSlices; ++k) v += red[(k * 64 + px) * 4 + co];
  const int oy = oy0 + (px >> 3), ox = ox0 + (px & 7);
  if (co < p.Cout && oy < p.H && ox < p.W) {
    const size_t pix = ((size_t)img * p.H + oy) * p.W + ox;
    if (p.bias) v += p.bias[co];
    if (p.chan_add) v += p.chan_add[(size_t)img * p.chan_add_ld + co];
    const float r = p.resid ? p.resid[pix * p.Cout + co] : 0.f;
    if (!p.resid_post) v += r;
    v *= p.scale;
    if (p.resid_post) v += r;
    p.dst[pix * p.Cout + co] = v;
  }
}

}  // namespace

// Does ssde_conv2d hand this launch to the kernel above?  (plain 3x3 / stride 1 / pad 1, no fused 1x1 source, at most four
// output channels, input channels a multiple of 8, nobody asked for GroupNorm partials of the result or for another tile)
bool ssde_conv_small_wants(const ssde_conv_args* a) {
  if (!a || (a->flags & SSDE_CONVF_NO_SMALL_COUT) || a->tile != SSDE_TILE_AUTO) return false;
  const int ctot = a->main.c0 + a->main.c1;
  return a->ksize == 3 && a->stride == 1 && a->pad == 1 && a->aux.p0 == nullptr && a->main.p0 && a->c_out >= 1 && a->c_out <= 4 &&
         a->h_in == a->h_out && a->w_in == a->w_out && ctot >= 8 && ctot % 8 == 0 && a->main.c0 % 4 == 0 && a->gn_part == nullptr;
}

int ssde_conv_small_launch(const ssde_conv_args* a, void* stream, int* lds_out) {
  SSDE_REQUIRE(ssde_conv_small_wants(a) && a->dst && a->w_main, "conv(small cout): not a launch for this kernel");
  if (lds_out) { *lds_out = kLdsBytes; return SSDE_OK; }
  const ssde_src& s = a->main;
  SSDE_REQUIRE(s.c1 == 0 || s.p1, "conv(small cout): second source missing");
  const bool gn = s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU;
  if (gn) {
    SSDE_REQUIRE(s.gn_groups > 0 && (s.c0 + s.c1) % s.gn_groups == 0 && ((s.c0 + s.c1) / s.gn_groups) % 4 == 0,
                 "conv(small cout): GroupNorm needs channels-per-group %% 4 == 0");
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "conv(small cout): GroupNorm pointers missing");
  }
  SSDE_REQUIRE(s.drop_thresh == 0 || s.drop_seed, "conv(small cout): dropout seed pointer missing");
  SSDE_REQUIRE((unsigned long long)a->n * a->h_in * a->w_in * (unsigned)(s.c0 + s.c1) < (1ull << 32), "conv(small cout): tensor too large");
  SmallParams p;
  p.src = s; p.wpk = a->w_main;
  p.N = a->n; p.H = a->h_out; p.W = a->w_out; p.Cout = a->c_out; p.CoutPad = ssde_cdiv(a->c_out, 64) * 64;
  p.tiles_x = ssde_cdiv(a->w_out, kTile); p.tiles_y = ssde_cdiv(a->h_out, kTile);
  p.bias = a->bias; p.chan_add = a->chan_add; p.chan_add_ld = a->chan_add_ld;
  p.resid = a->resid; p.resid_post = a->resid_post; p.scale = a->out_scale; p.dst = a->dst;
  const long long wgs = (long long)a->n * p.tiles_x * p.tiles_y;
  SSDE_REQUIRE(wgs > 0 && wgs < (1ll << 31), "conv(small cout): bad grid");
  static std::atomic<bool> attr_set;
  if (!attr_set) {                              // once, before any stream capture
    SSDE_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_small_cout_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024) == hipSuccess, "conv(small cout): hipFuncSetAttribute failed");
    attr_set = true;
  }
  hipLaunchKernelGGL(conv_small_cout_kernel, dim3((unsigned)wgs), dim3(kThreads), kLdsBytes, static_cast<hipStream_t>(stream), p);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
