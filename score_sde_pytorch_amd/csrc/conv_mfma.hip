// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 64 FLOP/clk/SIMD = the fp32 peak).
//
//   out[m, j] = scale * ( sum_{tap,ci} pro(main)[pix(m)+tap, ci] * Wm[tap, ci, j]
//                       + sum_{cx}     pro(aux )[pix(m),     cx] * Wa[cx, j]
//                       + bias[j] + chan_add[n(m), j] + resid[m, j] )
//
// Reference semantics: nn.Conv2d 3x3 / 1x1 (models/layers.py:100-124), NIN
// (models/layers.py:546-555), nn.Linear (layerspp.py:227,263), the strided conv
// of conv_downsample_2d (up_or_down_sampling.py:178), fused with the GroupNorm
// apply + SiLU that precede them (layerspp.py:243,264,77) and the residual tail
// (layerspp.py:268-274).
//
// Data layout: activations NHWC fp32; a workgroup (4 waves) owns BM output pixels
// (a TH x TW patch of IMGS images) x BN output channels.  Per input-channel chunk
// the (TH+2)x(TW+2) halo is staged ONCE into LDS (rows padded by 4 floats: the
// ds_read_b128 fragment reads are bank-conflict free) and re-used by all 9 taps;
// weights are host-packed [cin/8][tap][cout][8] so a tap's BN x 8 panel is one
// contiguous run.  An MFMA k-slot pair (h = lane>>5) is mapped to 4 consecutive
// channels so one ds_read_b128 feeds 4 MFMAs per operand.
#include "ssde_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct ConvGeom {
  int N, Hin, Win, Hout, Wout, Cout, CoutPad;
  int stride, pad;
  int lTW, lTH;
  int tiles_x, tiles_per_img;
  int m_tiles, n_tiles;
};

struct ConvKParams {
  ssde_src main, aux;
  const float* w_main;
  const float* w_aux;
  ConvGeom g;
  const float* bias;
  const float* chan_add;
  int chan_add_ld;
  const float* resid;
  int resid_post;
  float scale;
  float* dst;
  float* gn_part;          // GroupNorm partials of dst (one image per tile), see ssde_store_tile
  int ksplit;              // 1, or 2: two workgroups per tile, each reducing half of the input-channel chunks
  unsigned* sync;          // ksplit == 2: this launch's (ticket, ready) pairs, one per tile (zero before and after)
};

constexpr int kThreads = 256;
// input channels per LDS stage of the 3x3 phase: 8; 16 for the 64 x 64 tile when its halo fits the staging plan (small
// maps: short loops and two barriers per stage -- 4x4 at batch 256: 62 -> 72 TF/s)
#ifndef SSDE_CONV_BKC64
#define SSDE_CONV_BKC64 16
#endif

// Small feature maps (4x4 at batch 256: 256 tiles of 64 pixels x 64 channels) give a launch of one workgroup per CU,
// and a lone workgroup has nobody to cover its staging latency.  Such launches split the reduction over two
// workgroups on the same XCD; whichever finishes first leaves its raw sums in the tile's own part of dst, the other adds
// them to its own and runs the epilogue.  x + y == y + x bit for bit, so the result does not depend on who came first.
constexpr int kSyncSlots = 1 << 17;            // pairs; a launch takes m_tiles * n_tiles of them from a rotating cursor
__device__ unsigned g_conv_sync[kSyncSlots * 2];

// One reduction phase: KS x KS taps over source `s`, BKC input channels per stage.
template <int KS, int BKC, int BM, int BN, int TM, int TN, int MAXI>
__device__ __forceinline__ void conv_phase(const ssde_src& s, const float* __restrict__ wpk,
                                           const ConvGeom& g, int stride, int pad, int Hs, int Ws,
                                           int img0, int ty, int tx, int n0, int wm0, int wn0,
                                           f32x16 (&acc)[TM][TN], float* smem, int ks, int ksplit) {
  constexpr int T = KS * KS;
  constexpr int F4 = BKC / 4;
  constexpr int LDA = BKC + 4;
  constexpr int NB8 = BKC / 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int TW = 1 << g.lTW, TH = 1 << g.lTH;
  const int IMGS = BM >> (g.lTW + g.lTH);
  const int HWd = (TW - 1) * stride + KS;
  const int HH = (TH - 1) * stride + KS;
  const int halo_px = IMGS * HH * HWd;
  const int items = halo_px * F4;
  float* As = smem;
  float* Bs = smem + ((halo_px * LDA + 3) & ~3);

  // ---- per-thread staging plan (identical for every channel chunk) ----
  const int f4 = tid % F4;
  int goff[MAXI];   // input pixel index, -1 = zero padding, -2 = no item
  int gimg[MAXI];
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int q = tid + it * kThreads;
    goff[it] = -2;
    gimg[it] = 0;
    if (q < items) {
      const int hp = q / F4;
      const int il = hp / (HH * HWd);
      const int rem = hp - il * (HH * HWd);
      const int hy = rem / HWd;
      const int hx = rem - hy * HWd;
      const int iy = ty * TH * stride - pad + hy;
      const int ix = tx * TW * stride - pad + hx;
      const int img = img0 + il;
      const bool inb = (img < g.N) && (iy >= 0) && (iy < Hs) && (ix >= 0) && (ix < Ws);
      goff[it] = inb ? (img * Hs + iy) * Ws + ix : -1;
      gimg[it] = img;
    }
  }
  // ---- per-lane fragment bases ----
  int hb[TM], bb[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int m = wm0 + a * 32 + li;
    const int c = m & (TW - 1);
    const int r = (m >> g.lTW) & (TH - 1);
    const int il = m >> (g.lTW + g.lTH);
    hb[a] = ((il * HH + r * stride) * HWd + c * stride) * LDA + lh * 4;
  }
#pragma unroll
  for (int b = 0; b < TN; ++b) bb[b] = (wn0 + b * 32 + li) * LDA + lh * 4;

  const int Ctot = s.c0 + s.c1;
  const int ncin8 = (Ctot + 7) >> 3;
  const int nchunks = (Ctot + BKC - 1) / BKC;
  const int ch0 = nchunks * ks / ksplit, ch1 = nchunks * (ks + 1) / ksplit;     // this workgroup's share of the reduction
  const SsdePro pro = ssde_pro_decode(s);
  const int cpg = pro.gn ? Ctot / s.gn_groups : 1;
  constexpr int B_ITEMS = T * BN * F4;
  constexpr int B_ITERS = (B_ITEMS + kThreads - 1) / kThreads;

  // ---- staging registers of the stage in flight ----
  float4 av[MAXI];
  float4 bv[B_ITERS];
  float mu[MAXI], rs[MAXI];
  float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
  bool chan_ok = false;
  int c_stage = 0;

  // global loads of stage `ch` into registers (activation halo, weights, GroupNorm parameters)
  auto load_stage = [&](int ch) {
    const int c_base = ch * BKC;
    c_stage = c_base;
    const float* base;
    int C, cc;
    if (c_base < s.c0) { base = s.p0; C = s.c0; cc = c_base; }
    else               { base = s.p1; C = s.c1; cc = c_base - s.c0; }
    const int cthr = cc + f4 * 4;
    chan_ok = cthr < C;
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
      av[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (goff[it] >= 0 && chan_ok)
        av[it] = *reinterpret_cast<const float4*>(base + (size_t)goff[it] * C + cthr);
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int q = tid + it * kThreads;
      bv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (B_ITEMS % kThreads == 0 || q < B_ITEMS) {
        const int f = q & 1;
        const int j = (q >> 1) % BN;
        const int r = (q >> 1) / BN;       // = cb * T + tap
        const int tap = r % T, cb = r / T;
        const int cin8 = ch * NB8 + cb;
        if (cin8 < ncin8)
          bv[it] = *reinterpret_cast<const float4*>(
              wpk + ((size_t)(cin8 * T + tap) * g.CoutPad + n0 + j) * 8 + f * 4);
      }
    }
#pragma unroll
    for (int it = 0; it < MAXI; ++it) { mu[it] = 0.f; rs[it] = 1.f; }
    if (pro.gn && (c_base + f4 * 4) < Ctot) {
      gam = *reinterpret_cast<const float4*>(s.gn_gamma + c_base + f4 * 4);
      bet = *reinterpret_cast<const float4*>(s.gn_beta + c_base + f4 * 4);
      const int gidx = (c_base + f4 * 4) / cpg;
#pragma unroll
      for (int it = 0; it < MAXI; ++it)
        if (goff[it] >= 0) {
          mu[it] = s.gn_mean[gimg[it] * s.gn_groups + gidx];
          rs[it] = s.gn_rstd[gimg[it] * s.gn_groups + gidx];
        }
    }
  };
  // prologue transform (GroupNorm apply, SiLU, dropout) + LDS stores of the staged registers
  auto store_stage = [&]() {
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
      if (goff[it] == -2) continue;
      float4 v = av[it];
      if (goff[it] >= 0 && chan_ok)
        v = ssde_pro_apply(v, mu[it], rs[it], gam, bet, (uint32_t)goff[it] * (uint32_t)Ctot + (uint32_t)(c_stage + f4 * 4), pro);
      const int q = tid + it * kThreads;
      *reinterpret_cast<float4*>(As + (q / F4) * LDA + f4 * 4) = v;
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int q = tid + it * kThreads;
      if (B_ITEMS % kThreads == 0 || q < B_ITEMS) {
        const int f = q & 1;
        const int j = (q >> 1) % BN;
        const int r = (q >> 1) / BN;
        const int tap = r % T, cb = r / T;
        *reinterpret_cast<float4*>(Bs + (tap * BN + j) * LDA + cb * 8 + f * 4) = bv[it];
      }
    }
  };

  // software pipeline: the global loads of stage ch+1 are in flight while stage ch runs on the matrix pipe
  if (ch0 >= ch1) return;
  load_stage(ch0);
  __syncthreads();       // the previous phase's fragment reads are done
  store_stage();
  __syncthreads();
  for (int ch = ch0; ch < ch1; ++ch) {
    const bool has_next = ch + 1 < ch1;
    if (has_next) load_stage(ch + 1);
    // ---- MFMA: every tap re-uses the staged halo ----
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
      const int dy = tap / KS, dx = tap % KS;
      const int aoff = (dy * HWd + dx) * LDA;
#pragma unroll
      for (int kk = 0; kk < NB8; ++kk) {
        float4 af[TM], bf[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const float4*>(As + hb[a] + aoff + kk * 8);
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const float4*>(Bs + bb[b] + tap * BN * LDA + kk * 8);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
          }
      }
    }
    if (has_next) {
      __syncthreads();   // every wave is done reading this stage
      store_stage();
      __syncthreads();
    }
  }
}

template <int WM, int WN, int TM, int TN, bool HAS3, bool HAS1, int BKC3>
__global__ __launch_bounds__(kThreads, 2) void conv_mfma_kernel(const ConvKParams p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  SSDE_LDS(smem);
  const ConvGeom& g = p.g;

  // XCD-aware tile order: the n-tiles of one m-tile are consecutive on ONE XCD
  // (blocks are dispatched round-robin over the 8 XCDs), so the activation halo
  // is fetched from HBM once and re-read from that XCD's L2.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, l = bid >> 3;
  const int nt = l % g.n_tiles;
  const int ks = (l / g.n_tiles) % p.ksplit;                  // the two halves of a tile: same XCD, 8 * n_tiles blocks apart
  const int mt = (l / (g.n_tiles * p.ksplit)) * 8 + xcd;
  if (mt >= g.m_tiles) return;

  const int IMGS = BM >> (g.lTW + g.lTH);
  const int img0 = (mt / g.tiles_per_img) * IMGS;
  const int trem = mt % g.tiles_per_img;
  const int ty = trem / g.tiles_x, tx = trem % g.tiles_x;
  const int n0 = nt * BN;

  const int wave = threadIdx.x >> 6;
  const int wm0 = (wave / WN) * (TM * 32);
  const int wn0 = (wave % WN) * (TN * 32);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if constexpr (HAS3)
    conv_phase<3, BKC3, BM, BN, TM, TN, 5>(p.main, p.w_main, g, g.stride, g.pad, g.Hin, g.Win,
                                         img0, ty, tx, n0, wm0, wn0, acc, smem, ks, p.ksplit);
  if constexpr (HAS1)
    conv_phase<1, 32, BM, BN, TM, TN, (BM * 8) / kThreads>(p.aux, p.w_aux, g, 1, 0, g.Hout, g.Wout,
                                                          img0, ty, tx, n0, wm0, wn0, acc, smem, ks, p.ksplit);

  // ---- epilogue: accumulators -> LDS tile [BM pixels][BN + 4] -> coalesced float4 stores (ssde_store_tile) ----
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int TW = 1 << g.lTW, TH = 1 << g.lTH;
  constexpr int LDT = BN + 4;
  __syncthreads();                       // every wave is done with the operand stages
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        smem[m * LDT + wn0 + b * 32 + li] = acc[a][b][r];
      }
  __syncthreads();
  auto pixfn = [&](int m, size_t& pix, int& img) {
    const int c = m & (TW - 1);
    const int rr = (m >> g.lTW) & (TH - 1);
    img = img0 + (m >> (g.lTW + g.lTH));
    const int oy = ty * TH + rr, ox = tx * TW + c;
    if (img >= g.N || oy >= g.Hout || ox >= g.Wout) return false;
    pix = ((size_t)img * g.Hout + oy) * g.Wout + ox;
    return true;
  };
  if constexpr (BM == 64 && BN == 64) if (p.ksplit == 2) {
    // ---- hand-over between the two halves of the reduction (see g_conv_sync) ----
    unsigned* sy = p.sync + 2 * (mt * g.n_tiles + nt);
    int* ticket = reinterpret_cast<int*>(smem + BM * LDT);
    if (threadIdx.x == 0) *ticket = (int)atomicAdd(sy, 1u);
    __syncthreads();
    const bool first = *ticket == 0;
    // The sums travel as agent-scope (sc1) dword accesses, which are coherent at the device level by themselves: a
    // __threadfence() here is a write-back of the whole L2 per workgroup (measured: the split launch 40 % SLOWER than the
    // unsplit one), and all that is needed is that the stores are acknowledged before the flag goes up.
    if (!first) {
      if (threadIdx.x == 0)
        while (__hip_atomic_load(sy + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(4);
      __syncthreads();
    }
    constexpr int QN = BN / 4;            // float4 columns of a tile row
    constexpr int ITERS = BM * QN / kThreads;
    float* tp[ITERS];
    float* dp[ITERS];
    float o[ITERS][4];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int q = threadIdx.x + it * kThreads;
      const int m = q / QN, j = (q % QN) * 4;
      size_t pix; int img;
      const bool ok = pixfn(m, pix, img) && n0 + j < g.Cout;     // c_out % 4 == 0 on this path
      tp[it] = smem + m * LDT + j;
      dp[it] = ok ? p.dst + pix * g.Cout + n0 + j : nullptr;
    }
    if (first) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it)
        if (dp[it]) {
#pragma unroll
          for (int k = 0; k < 4; ++k) __hip_atomic_store(dp[it] + k, tp[it][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
      // all loads in flight before the first add
#pragma unroll
      for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          o[it][k] = dp[it] ? __hip_atomic_load(dp[it] + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
      for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k) tp[it][k] += o[it][k];
    }
    if (first) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0);      // every store of this thread is acknowledged ...
      __syncthreads();                    // ... and every thread's
      if (threadIdx.x == 0) atomicAdd(sy + 1, 1u);
      return;
    }
    __syncthreads();
    if (threadIdx.x == 0) { sy[0] = 0u; sy[1] = 0u; }            // ready for the next launch that is dealt these slots
  }
  SsdeEpi e{p.bias, p.chan_add, p.chan_add_ld, p.resid, p.resid_post, p.scale, p.dst, g.Cout, p.gn_part};
  const int gn_entry = p.gn_part ? img0 * g.tiles_per_img + trem : -1;        // IMGS > 1: tiles_per_img == 1
  const int rpi_log2 = IMGS > 1 ? g.lTW + g.lTH : 30;                          // rows per image = TW * TH
  ssde_store_tile<BM, BN, kThreads>(smem, LDT, n0, e, pixfn, gn_entry, rpi_log2, g.N * g.tiles_per_img);
}

struct TileCfg { int bm, bn; };
const TileCfg kTiles[5] = {{0, 0}, {256, 64}, {128, 64}, {64, 64}, {256, 32}};

struct ConvPlan {
  ConvKParams kp;
  int tile;
  int lds_bytes;
  int grid;
  bool has3, has1;
  int bkc;                 // input channels per stage of the 3x3 phase
  int gn_slices;           // slices per image of the GroupNorm partials (0: a tile spans several images)
};

int src_check(const ssde_src& s, const char* what) {
  SSDE_REQUIRE(s.p0 != nullptr && s.c0 > 0, "conv: %s source missing", what);
  SSDE_REQUIRE(s.c0 % 4 == 0 && s.c1 % 4 == 0, "conv: %s channels must be multiples of 4 (got %d,%d)", what, s.c0, s.c1);
  SSDE_REQUIRE(s.c1 == 0 || s.p1 != nullptr, "conv: %s second tensor missing", what);
  SSDE_REQUIRE(s.c1 == 0 || s.c0 % 32 == 0, "conv: %s concat boundary must be a multiple of 32", what);
  if (s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU) {
    SSDE_REQUIRE(s.gn_groups > 0 && (s.c0 + s.c1) % s.gn_groups == 0 && ((s.c0 + s.c1) / s.gn_groups) % 4 == 0,
                 "conv: %s GroupNorm needs channels-per-group %% 4 == 0", what);
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "conv: %s GroupNorm pointers missing", what);
  }
  return SSDE_OK;
}

int pow2_floor(int v) { int p = 1; while (p * 2 <= v) p *= 2; return p; }

// tile geometry: TW x TH output pixels (powers of two; the output itself may be any size, edge tiles are
// bounds-checked) of IMGS images
int halo_px(int bm, int w_out, int h_out, int stride, int ks, int* lTW, int* lTH) {
  int tw = pow2_floor(w_out < 16 ? w_out : 16);
  int th = bm / tw; if (th > h_out) th = h_out;
  th = pow2_floor(th);
  *lTW = ssde_ilog2(tw); *lTH = ssde_ilog2(th);
  const int imgs = bm / (tw * th);
  return imgs * ((th - 1) * stride + ks) * ((tw - 1) * stride + ks);
}

int make_plan(const ssde_conv_args* a, ConvPlan* pl) {
  SSDE_REQUIRE(a && a->dst, "conv: null args");
  pl->has3 = a->ksize != 0;
  pl->has1 = a->aux.p0 != nullptr;
  SSDE_REQUIRE(pl->has3 || pl->has1, "conv: neither a k x k nor a 1 x 1 source");
  SSDE_REQUIRE(a->ksize == 0 || a->ksize == 3, "conv: ksize must be 0 or 3 (1x1 goes through aux)");
  SSDE_REQUIRE(a->n > 0 && a->h_out > 0 && a->w_out > 0 && a->c_out > 0, "conv: bad output shape");
  if (pl->has3) {
    if (int rc = src_check(a->main, "main")) return rc;
    SSDE_REQUIRE(a->w_main, "conv: w_main missing");
    SSDE_REQUIRE(a->stride == 1 || a->stride == 2, "conv: stride must be 1 or 2");
    // the output may be a top-left crop of the full convolution result (used by strided input-gradients)
    SSDE_REQUIRE((a->h_in + 2 * a->pad - 3) / a->stride + 1 >= a->h_out && (a->w_in + 2 * a->pad - 3) / a->stride + 1 >= a->w_out,
                 "conv: output %dx%d larger than input %dx%d stride %d pad %d allows", a->h_out, a->w_out, a->h_in, a->w_in, a->stride, a->pad);
  }
  if (pl->has1) {
    if (int rc = src_check(a->aux, "aux")) return rc;
    SSDE_REQUIRE(a->w_aux, "conv: w_aux missing");
  }
  const int M = a->n * a->h_out * a->w_out;
  int tile = a->tile;
  if (tile == SSDE_TILE_AUTO) {
    if (a->c_out <= 32) tile = SSDE_TILE_256x32;
    else {
      const int nt = ssde_cdiv(a->c_out, 64);
      if (ssde_cdiv(M, 256) * nt >= 512) tile = SSDE_TILE_256x64;
      else if (ssde_cdiv(M, 128) * nt >= 384) tile = SSDE_TILE_128x64;
      else tile = SSDE_TILE_64x64;
    }
    if (a->w_out * a->h_out * a->n < kTiles[tile].bm && tile != SSDE_TILE_256x32) tile = SSDE_TILE_64x64;
  }
  SSDE_REQUIRE(tile >= 1 && tile <= 4, "conv: bad tile id %d", tile);
  (void)SSDE_TILE_WINOGRAD4;
  (void)SSDE_TILE_WINOGRAD;
  int lTW = 0, lTH = 0;
  pl->bkc = 8;
  const bool deep_ok = !(a->flags & SSDE_CONVF_BKC8);     // (the flag: 8-channel stages everywhere -- A/B runs, tests)
  if (pl->has3) {
    // the halo of the chosen tile must fit the per-thread staging plan (5 x 256 float4 items)
    while (true) {
      const int px = halo_px(kTiles[tile].bm, a->w_out, a->h_out, a->stride, 3, &lTW, &lTH);
      pl->bkc = (deep_ok && tile == SSDE_TILE_64x64 && px * (SSDE_CONV_BKC64 / 4) <= 5 * kThreads) ? SSDE_CONV_BKC64 : 8;
      if (px * 2 <= 5 * kThreads) break;
      SSDE_REQUIRE(tile != SSDE_TILE_64x64, "conv: halo too large for any tile");
      tile = (tile == SSDE_TILE_256x32) ? SSDE_TILE_64x64 : tile + 1;   // 256x64 -> 128x64 -> 64x64
    }
  } else {
    halo_px(kTiles[tile].bm, a->w_out, a->h_out, 1, 1, &lTW, &lTH);
  }
  const int bm = kTiles[tile].bm, bn = kTiles[tile].bn;
  const int tw = 1 << lTW, th = 1 << lTH, imgs = bm / (tw * th);

  ConvKParams& kp = pl->kp;
  kp.main = a->main; kp.aux = a->aux; kp.w_main = a->w_main; kp.w_aux = a->w_aux;
  ConvGeom& g = kp.g;
  g.N = a->n; g.Hin = a->h_in; g.Win = a->w_in; g.Hout = a->h_out; g.Wout = a->w_out;
  g.Cout = a->c_out; g.CoutPad = ssde_cdiv(a->c_out, 64) * 64;
  g.stride = pl->has3 ? a->stride : 1; g.pad = pl->has3 ? a->pad : 0;
  g.lTW = lTW; g.lTH = lTH;
  g.tiles_x = ssde_cdiv(a->w_out, tw);
  g.tiles_per_img = g.tiles_x * ssde_cdiv(a->h_out, th);
  g.m_tiles = ssde_cdiv(a->n, imgs) * g.tiles_per_img;
  g.n_tiles = ssde_cdiv(a->c_out, bn);
  kp.bias = a->bias; kp.chan_add = a->chan_add; kp.chan_add_ld = a->chan_add_ld;
  kp.resid = a->resid; kp.resid_post = a->resid_post; kp.scale = a->out_scale; kp.dst = a->dst;
  kp.gn_part = a->gn_part;
  // part of one image per tile, or whole images of at least one epilogue trip (kThreads / (bn / 4) rows) each
  const bool gn_ok = a->c_out % 4 == 0 && (imgs == 1 || (g.tiles_per_img == 1 && tw * th >= kThreads / (bn / 4)));
  pl->gn_slices = gn_ok ? g.tiles_per_img * (kThreads / 64) : 0;
  SSDE_REQUIRE(!a->gn_part || pl->gn_slices > 0, "conv: GroupNorm partials need one image per tile and c_out %% 4 == 0");

  int lds = 0;
  if (pl->has3) {
    const int px = imgs * ((th - 1) * g.stride + 3) * ((tw - 1) * g.stride + 3);
    const int lda = pl->bkc + 4;
    lds = (((px * lda + 3) & ~3) + 9 * bn * lda) * 4;
  }
  if (pl->has1) {
    const int l1 = (bm * 36 + bn * 36) * 4;
    if (l1 > lds) lds = l1;
  }
  const int epi_bytes = bm * (bn + 4) * 4 + 16;     // the epilogue parks the output tile in LDS (+ the hand-over ticket)
  pl->lds_bytes = lds > epi_bytes ? lds : epi_bytes;
  pl->tile = tile;
  // split the reduction when the launch would leave every CU with at most one workgroup (see g_conv_sync)
  const bool split_ok = !(a->flags & SSDE_CONVF_NO_KSPLIT);
  const int wgs = ssde_cdiv(g.m_tiles, 8) * 8 * g.n_tiles;
  const int ctot = pl->has3 ? a->main.c0 + a->main.c1 : 0;
  kp.ksplit = (split_ok && pl->has3 && tile == SSDE_TILE_64x64 && wgs <= 320 && ctot >= 128 && a->c_out % 4 == 0 &&
               a->resid != a->dst && g.m_tiles * g.n_tiles <= kSyncSlots / 4) ? 2 : 1;
  kp.sync = nullptr;
  pl->grid = wgs * kp.ksplit;
  return SSDE_OK;
}

template <int WM, int WN, int TM, int TN, int BKC3 = 8>
int launch_cfg(ConvPlan& pl, hipStream_t st) {
  dim3 grid(pl.grid), block(kThreads);
  if (pl.kp.ksplit == 2) {
    pl.kp.sync = ssde_conv_sync_slots(pl.kp.g.m_tiles * pl.kp.g.n_tiles);
    SSDE_REQUIRE(pl.kp.sync, "conv: no hand-over slots for a split reduction");
  }
#define SSDE_CONV_LAUNCH(H3, H1)                                                                     \
  do {                                                                                               \
    auto kfn = conv_mfma_kernel<WM, WN, TM, TN, H3, H1, BKC3>;                                       \
    static std::atomic<bool> attr_set{false}; /* once per instantiation, before any stream capture */            \
    if (!attr_set) {                                                                                 \
      SSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));   \
      attr_set = true;                                                                               \
    }                                                                                                \
    hipLaunchKernelGGL(kfn, grid, block, pl.lds_bytes, st, pl.kp);                                   \
  } while (0)
  if (pl.has3 && pl.has1) SSDE_CONV_LAUNCH(true, true);
  else if (pl.has3) SSDE_CONV_LAUNCH(true, false);
  else SSDE_CONV_LAUNCH(false, true);
#undef SSDE_CONV_LAUNCH
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

}  // namespace

// A fresh run of (ticket, ready) pairs for a launch that splits its reduction (this file and conv_wino4.hip): launches in
// flight on different streams (or captured into different graphs) never share a pair unless more than kSyncSlots pairs are
// dealt in between.  nullptr: more pairs asked for than a launch may take.
unsigned* ssde_conv_sync_slots(int need) {
  static std::atomic<unsigned> cursor{0};
  if (need <= 0 || need > kSyncSlots / 4) return nullptr;
  // CAS loop: a run that would straddle the end of the table starts at 0 AND moves the cursor behind itself, so the next
  // caller cannot be dealt [0, need) again (a bare fetch_add left the cursor inside the run just handed out)
  unsigned cur = cursor.load(std::memory_order_relaxed), at, next;
  do {
    at = cur % (unsigned)kSyncSlots;
    if (at + (unsigned)need > (unsigned)kSyncSlots) at = 0;
    next = at + (unsigned)need;
  } while (!cursor.compare_exchange_weak(cur, next, std::memory_order_relaxed));
  unsigned* base = nullptr;
  if (hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(g_conv_sync)) != hipSuccess) return nullptr;
  return base + 2 * (size_t)at;
}

// the Winograd kernels have their own launchers: launch (lds_out == NULL), LDS query, or GroupNorm-slice query (stream == 1)
typedef int (*ssde_wino_launcher)(const ssde_conv_args*, void*, int*);
static ssde_wino_launcher wino_launcher(int tile) {
  switch (tile) {
    case SSDE_TILE_WINOGRAD: return ssde_conv_wino_launch;
    case SSDE_TILE_WINOGRAD4: return ssde_conv_wino4_launch;
    case SSDE_TILE_WINOGRAD4R: return ssde_conv_wino4r_launch;
    default: return nullptr;
  }
}

extern "C" int ssde_conv2d(const ssde_conv_args* a, void* stream) {
  if (a && a->gn_in_part0 && !ssde_wino4_xform_merges_gn(a)) {
    // the statistics of main are still partials and this route has no kernel that merges them for itself: the finalize launch
    // of ABI 3-9, issued here in front of the kernel
    const ssde_src& s = a->main;
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_groups > 0 && a->ksize != 0, "conv: gn_in_part0 needs a k x k main source with a GroupNorm prologue");
    ssde_gn_finalize_args f;
    f.part0 = a->gn_in_part0; f.part1 = s.c1 > 0 ? a->gn_in_part1 : nullptr;
    f.c0 = s.c0; f.c1 = s.c1; f.slices0 = a->gn_in_slices0; f.slices1 = s.c1 > 0 ? a->gn_in_slices1 : 0;
    f.n = a->n; f.groups = s.gn_groups; f.eps = a->gn_in_eps;
    f.mean = const_cast<float*>(s.gn_mean); f.rstd = const_cast<float*>(s.gn_rstd);
    if (int rc = ssde_gn_finalize(&f, stream)) return rc;
  }
  if (a) {
    if (ssde_wino_launcher fn = wino_launcher(a->tile)) {
      // the two-kernel forms: the input-transform pass into wino_v first, unless the caller says wino_v already holds it
      if (a->tile == SSDE_TILE_WINOGRAD4R && !(a->flags & SSDE_CONVF_V_GIVEN)) {
        // (the matrix kernel's launcher validates the arguments -- aux source, kernel size / stride / pad, map and channel limits,
        //  V < 4 GB --: ask it in its plan-only form first, so that an invalid call enqueues nothing)
        int lds = 0;
        if (int rc = fn(a, nullptr, &lds)) return rc;
        if (int rc = ssde_wino4_xform_vq_launch(a, stream)) return rc;
      }
      return fn(a, stream, nullptr);
    }
  }
  if (a && a->dst && ssde_conv1x1_wants(a)) return ssde_conv1x1_launch(a, stream, nullptr);   // 1x1-only: GEMM kernel (conv1x1.hip)
  if (a && a->dst && ssde_conv_small_wants(a)) return ssde_conv_small_launch(a, stream, nullptr);   // image heads (conv_small.hip)
  ConvPlan pl;
  if (int rc = make_plan(a, &pl)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (pl.tile) {
    case SSDE_TILE_256x64: return launch_cfg<4, 1, 2, 2>(pl, st);
    case SSDE_TILE_128x64: return launch_cfg<4, 1, 1, 2>(pl, st);
    case SSDE_TILE_64x64:  return pl.bkc == 8 ? launch_cfg<2, 2, 1, 1>(pl, st) : launch_cfg<2, 2, 1, 1, SSDE_CONV_BKC64>(pl, st);
    case SSDE_TILE_256x32: return launch_cfg<4, 1, 2, 1>(pl, st);
  }
  ssde_set_error("conv: unreachable tile %d", pl.tile);
  return SSDE_EINVAL;
}

extern "C" int ssde_conv_gn_slices(const ssde_conv_args* a) {
  if (!a || a->c_out % 4 != 0) return 0;
  ssde_conv_args q = *a;
  q.gn_part = nullptr;
  if (ssde_wino_launcher fn = wino_launcher(q.tile)) {
    int s = 0;
    if (fn(&q, reinterpret_cast<void*>(1), &s)) return 0;     // plan-only query form
    return s;
  }
  if (q.dst && ssde_conv1x1_wants(&q)) {
    // the GEMM kernel stores 64-row half tiles of linear pixel rows, 4 waves each: part of one image, or whole images
    // of at least one epilogue trip (8 rows) each
    const int hw = q.h_out * q.w_out;
    if (hw % 64 == 0) return (hw / 64) * 4;
    return (hw >= 8 && hw < 64 && 64 % hw == 0) ? 4 : 0;
  }
  if (q.dst && ssde_conv_small_wants(&q)) return 0;             // (the image heads are not normalised by anybody)
  ConvPlan pl;
  if (make_plan(&q, &pl)) return 0;
  return pl.gn_slices;
}

extern "C" int ssde_conv_lds_bytes(const ssde_conv_args* a) {
  if (ssde_wino_launcher fn = a ? wino_launcher(a->tile) : nullptr) {
    int lds = 0;
    if (int rc = fn(a, nullptr, &lds)) return rc;
    return lds;
  }
  if (a && a->dst && ssde_conv1x1_wants(a)) {
    int lds = 0;
    if (int rc = ssde_conv1x1_launch(a, nullptr, &lds)) return rc;
    return lds;
  }
  if (a && a->dst && ssde_conv_small_wants(a)) {
    int lds = 0;
    if (int rc = ssde_conv_small_launch(a, nullptr, &lds)) return rc;
    return lds;
  }
  ConvPlan pl;
  if (int rc = make_plan(a, &pl)) return rc;
  return pl.lds_bytes;
}
