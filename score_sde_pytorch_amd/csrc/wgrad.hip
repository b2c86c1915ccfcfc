// Weight gradient of the fused convolutions (autograd of nn.Conv2d / NIN / nn.Linear as used at
// models/layers.py:100-124,546-555 and layerspp.py:227,263) on the exact-fp32 matrix pipe:
//
//   dw[co, ci, tap] += scale * sum_{pixels m} g[m, co] * pro(src)[m (+) tap, ci]
//
// This is a GEMM whose reduction dimension is the pixel index (N*H*W: 1e5 at batch 128), so it is
// split over workgroups along pixels.  Every workgroup writes its partial tile to a scratch slab with
// coalesced stores (lane = input channel) and a second kernel sums the slabs in a fixed order into dw
// (deterministic; a first version met in dw by fp32 atomics and spent 75% of its time in them).
// Workgroups that share pixels (same split, different tile) are placed on the same XCD.
// pro(src) -- GroupNorm apply, SiLU and the train-mode dropout mask -- is RECOMPUTED while the halo
// tile is staged, exactly as the forward kernel does (conv_mfma.hip), so the normalised / activated
// tensor is never stored for backward.
//
// Workgroup (4 waves) = BCO x BCI x T accumulators: T = 9 taps -> 64 x 64 (each wave one 32x32 block
// per tap, 144 accumulator registers); T = 1 -> 128 x 128 (each wave 2x2 blocks).  Per pixel chunk
// (PX output pixels of IMGS images) the output-gradient rows and the input halo are staged once into
// LDS pixel-major; a wave's MFMA k-slot pair is two adjacent pixels, so both fragment reads are
// conflict-free ds_read_b32 of 32 consecutive floats and the halo is re-used by all 9 taps.
// dw is written in the REFERENCE layout (OIHW, or [in][out] for NIN with transpose_out).
#include "ssde_common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kThreads = 256;

struct WgParams {
  ssde_src src;
  const float* g;
  int g_ld, g_off;
  int N, Hin, Win, Hout, Wout, Cout, Ctot, cin_store;
  int stride, pad;
  int lTW, lTH, lPX;
  int tiles_x, tiles_per_img, chunks, chunks_per_split;
  int co_tiles, ci_tiles, splits;
  int transpose_out;
  float scale;
  float* dw;
  float* scratch;
  int rows, rows_per_split;    // wgrad1x1_gemm_kernel: pixels in total / per split (a multiple of its 16-row stage)
};

template <int KS, int CO_B, int CI_B>
__global__ __launch_bounds__(kThreads, 2) void wgrad_kernel(const WgParams p) {
  constexpr int T = KS * KS;
  constexpr int BCO = 2 * CO_B * 32, BCI = 2 * CI_B * 32;
  constexpr int GF4 = BCO / 4, PF4 = BCI / 4;
  constexpr int GU = 4, PU = 4;             // loads in flight per thread while staging
  SSDE_LDS(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int wco = wave >> 1, wci = wave & 1;

  // XCD-aware order: the tiles of one pixel split are consecutive on ONE XCD (blocks are dispatched
  // round-robin over the 8 XCDs), so a split's gradient rows / halo are fetched into one L2 only
  const int ntiles = p.co_tiles * p.ci_tiles;
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
  const int tile = lin % ntiles, split = (lin / ntiles) * 8 + xcd;
  if (split >= p.splits) return;
  const int co0 = (tile / p.ci_tiles) * BCO, ci0 = (tile % p.ci_tiles) * BCI;
  const int ch_begin = split * p.chunks_per_split;
  const int ch_end = min(p.chunks, ch_begin + p.chunks_per_split);

  const int TW = 1 << p.lTW, TH = 1 << p.lTH, PX = 1 << p.lPX;
  const int IMGS = PX >> (p.lTW + p.lTH);
  const int HWd = (TW - 1) * p.stride + KS, HH = (TH - 1) * p.stride + KS;
  const int halo_px = IMGS * HH * HWd;
  float* gs = smem;                 // [PX][BCO]
  float* ps = smem + PX * BCO;      // [halo_px][BCI]

  const SsdePro pro = ssde_pro_decode(p.src);
  const int cpg = pro.gn ? p.Ctot / p.src.gn_groups : 1;

  f32x16 acc[T][CO_B][CI_B];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int a = 0; a < CO_B; ++a)
#pragma unroll
      for (int b = 0; b < CI_B; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][a][b][r] = 0.f;

  for (int chunk = ch_begin; chunk < ch_end; ++chunk) {
    const int img0 = (chunk / p.tiles_per_img) * IMGS;
    const int trem = chunk % p.tiles_per_img;
    const int ty = trem / p.tiles_x, tx = trem % p.tiles_x;
    __syncthreads();   // previous chunk's fragment reads are done

    // ---- stage the output-gradient rows: gs[m][c] = g[pix(m), g_off + co0 + c] ----
    for (int q0 = 0; q0 < PX * GF4; q0 += kThreads * GU) {
      float4 v[GU];
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int q = q0 + u * kThreads + tid;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < PX * GF4) {
          const int m = q / GF4, c = (q - m * GF4) * 4;
          const int il = m >> (p.lTW + p.lTH);
          const int oy = ty * TH + ((m >> p.lTW) & (TH - 1)), ox = tx * TW + (m & (TW - 1));
          const int img = img0 + il;
          const int col = p.g_off + co0 + c;
          if (img < p.N && oy < p.Hout && ox < p.Wout && col + 4 <= p.g_ld)
            v[u] = *reinterpret_cast<const float4*>(p.g + (((size_t)img * p.Hout + oy) * p.Wout + ox) * p.g_ld + col);
        }
      }
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int q = q0 + u * kThreads + tid;
        if (q < PX * GF4) *reinterpret_cast<float4*>(gs + (size_t)q * 4) = v[u];
      }
    }
    // ---- stage the input halo with its prologue: ps[hp][c] = pro(src)[pixel(hp), ci0 + c] ----
    for (int q0 = 0; q0 < halo_px * PF4; q0 += kThreads * PU) {
      float4 v[PU];
      int pixi[PU], chv[PU], imgv[PU];
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        const int q = q0 + u * kThreads + tid;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        pixi[u] = -1; chv[u] = 0; imgv[u] = 0;
        if (q < halo_px * PF4) {
          const int hp = q / PF4, c = (q - hp * PF4) * 4;
          const int il = hp / (HH * HWd);
          const int rem = hp - il * (HH * HWd);
          const int hy = rem / HWd, hx = rem - hy * HWd;
          const int iy = ty * TH * p.stride - p.pad + hy, ix = tx * TW * p.stride - p.pad + hx;
          const int img = img0 + il;
          const int ch = ci0 + c;
          if (img < p.N && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win && ch < p.Ctot) {
            const int pix = (img * p.Hin + iy) * p.Win + ix;
            pixi[u] = pix; chv[u] = ch; imgv[u] = img;
            v[u] = (ch < p.src.c0) ? *reinterpret_cast<const float4*>(p.src.p0 + (size_t)pix * p.src.c0 + ch)
                                   : *reinterpret_cast<const float4*>(p.src.p1 + (size_t)pix * p.src.c1 + (ch - p.src.c0));
          }
        }
      }
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        const int q = q0 + u * kThreads + tid;
        if (q < halo_px * PF4) {
          float4 x = v[u];
          if (pixi[u] >= 0) {
            float mu = 0.f, rs = 1.f;
            float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pro.gn) {
              const int gi = imgv[u] * p.src.gn_groups + chv[u] / cpg;
              mu = p.src.gn_mean[gi]; rs = p.src.gn_rstd[gi];
              gam = *reinterpret_cast<const float4*>(p.src.gn_gamma + chv[u]);
              bet = *reinterpret_cast<const float4*>(p.src.gn_beta + chv[u]);
            }
            x = ssde_pro_apply(x, mu, rs, gam, bet, (uint32_t)pixi[u] * (uint32_t)p.Ctot + (uint32_t)chv[u], pro);
          }
          *reinterpret_cast<float4*>(ps + (size_t)q * 4) = x;
        }
      }
    }
    __syncthreads();

    // ---- MFMA: k-slot pair = pixels (2ks, 2ks+1) ----
#pragma unroll 2
    for (int ks = 0; ks < PX / 2; ++ks) {
      const int m = 2 * ks + lh;
      const int il = m >> (p.lTW + p.lTH);
      const int r = (m >> p.lTW) & (TH - 1), c = m & (TW - 1);
      const float* ga = gs + m * BCO + wco * (CO_B * 32) + li;
      const float* pb = ps + ((il * HH + r * p.stride) * HWd + c * p.stride) * BCI + wci * (CI_B * 32) + li;
      float af[CO_B];
#pragma unroll
      for (int a = 0; a < CO_B; ++a) af[a] = ga[a * 32];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int dy = t / KS, dx = t % KS;
#pragma unroll
        for (int b = 0; b < CI_B; ++b) {
          const float bf = pb[(dy * HWd + dx) * BCI + b * 32];
#pragma unroll
          for (int a = 0; a < CO_B; ++a)
            acc[t][a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf, acc[t][a][b], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: lane owns input channel ci (column), 16 output channels (rows) per block ----
  if (p.splits == 1) {
#pragma unroll
    for (int b = 0; b < CI_B; ++b) {
      const int ci = ci0 + wci * (CI_B * 32) + b * 32 + li;
      if (ci >= p.cin_store) continue;
#pragma unroll
      for (int a = 0; a < CO_B; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + wco * (CO_B * 32) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (co >= p.Cout) continue;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const size_t idx = p.transpose_out ? (size_t)ci * p.Cout + co : ((size_t)co * p.cin_store + ci) * T + t;
            p.dw[idx] += acc[t][a][b][r] * p.scale;
          }
        }
    }
    return;
  }
  // partial slab [split][tile][tap][BCO][BCI]: consecutive lanes -> consecutive ci (128-byte runs)
  float* slab = p.scratch + ((size_t)split * ntiles + tile) * (size_t)(T * BCO * BCI);
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int a = 0; a < CO_B; ++a)
#pragma unroll
      for (int b = 0; b < CI_B; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int col = wco * (CO_B * 32) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          slab[((size_t)t * BCO + col) * BCI + wci * (CI_B * 32) + b * 32 + li] = acc[t][a][b][r];
        }
}

// ---- 1x1 / NIN / Linear weight gradients as a software-pipelined GEMM ------------------------------------------------
// dw[co, ci] = sum_m g[m, co] * pro(src)[m, ci] over the pixels m of the whole batch: both operands are K-major rows of
// channels exactly as they lie in memory.  wgrad_kernel<1, 2, 2> above stages a 64-pixel chunk, waits, multiplies, waits
// (44-54 TF/s: the load latency of every chunk is exposed); here a stage is 16 pixels, double buffered in LDS, and the
// next stage's rows are fetched into registers -- prologue applied there -- while the MFMAs of the current one run (the
// schedule of wgrad_wino4.hip's wgrad4_gemm_kernel).  Workgroup = 4 waves = 128 co x 128 ci, wave = a 64 x 64 quadrant;
// splits over pixel ranges, slabs in wgrad_reduce_kernel<1, 128, 128>'s layout (the same reduction follows).
constexpr int kG1BK = 16, kG1LDP = 128 + 4;                    // LDS row pitch: rows 4 banks apart
constexpr int kG1Stage = 2 * kG1BK * kG1LDP;                   // floats of one LDS stage (g rows, then source rows)

__global__ __launch_bounds__(kThreads, 4) void wgrad1x1_gemm_kernel(const WgParams p) {
  SSDE_LDS(smem);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int ntiles = p.co_tiles * p.ci_tiles;
  const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;       // the tiles of one pixel split share an XCD (one L2)
  const int tile = lin % ntiles, split = (lin / ntiles) * 8 + xcd;
  if (split >= p.splits) return;
  const int co0 = (tile / p.ci_tiles) * 128, ci0 = (tile % p.ci_tiles) * 128;
  const int k0 = split * p.rows_per_split;
  const int k1 = min(p.rows, k0 + p.rows_per_split);
  const int nst = (k1 - k0 + kG1BK - 1) / kG1BK;
  const int hw = p.Hout * p.Wout;

  const ssde_src& s = p.src;
  const SsdePro pro = ssde_pro_decode(s);
  const int cpg = pro.gn ? p.Ctot / s.gn_groups : 1;
  // staging: thread = rows r0, r0 + 8 of a stage, channel quad q4 of both operands
  const int q4 = (tid & 31) * 4, r0 = tid >> 5;
  const bool zc_ok = co0 + q4 < p.Cout && p.g_off + co0 + q4 + 4 <= p.g_ld;
  const int ch = ci0 + q4;
  const bool vc_ok = ch < p.Ctot;
  const bool second = ch >= s.c0;
  const float* vbase = second ? s.p1 + (ch - s.c0) : s.p0 + ch;
  const int vld = second ? s.c1 : s.c0;
  float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pro.gn && vc_ok) {
    gam = *reinterpret_cast<const float4*>(s.gn_gamma + ch);
    bet = *reinterpret_cast<const float4*>(s.gn_beta + ch);
  }
  const int grp = pro.gn && vc_ok ? ch / cpg : 0;
  float4 zv[2], vv[2];
  auto load_stage = [&](int st) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = k0 + st * kG1BK + r0 + i * 8;
      const bool ok = m < k1;
      const size_t row = (size_t)(ok ? m : k0);
      zv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      vv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok && zc_ok) zv[i] = *reinterpret_cast<const float4*>(p.g + row * p.g_ld + p.g_off + co0 + q4);
      if (ok && vc_ok) {
        float4 x = *reinterpret_cast<const float4*>(vbase + row * vld);
        float mu = 0.f, rs = 1.f;
        if (pro.gn) {
          const int gi = (int)(row / (size_t)hw) * s.gn_groups + grp;
          mu = s.gn_mean[gi]; rs = s.gn_rstd[gi];
        }
        vv[i] = ssde_pro_apply(x, mu, rs, gam, bet, (uint32_t)row * (uint32_t)p.Ctot + (uint32_t)ch, pro);
      }
    }
  };
  auto store_stage = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<float4*>(buf + (r0 + i * 8) * kG1LDP + q4) = zv[i];
      *reinterpret_cast<float4*>(buf + (kG1BK + r0 + i * 8) * kG1LDP + q4) = vv[i];
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
  const int aoff = lh * kG1LDP + wm0 + li, boff = (kG1BK + lh) * kG1LDP + wn0 + li;

  load_stage(0);
  store_stage(smem);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const float* cur = smem + (st & 1) * kG1Stage;
    const bool has_next = st + 1 < nst;
    if (has_next) load_stage(st + 1);
    float af[2][2], bf[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) af[0][a] = cur[aoff + a * 32];
#pragma unroll
    for (int c = 0; c < 2; ++c) bf[0][c] = cur[boff + c * 32];
#pragma unroll
    for (int kk = 0; kk < kG1BK / 2; ++kk) {
      const int w = kk & 1;
      if (kk + 1 < kG1BK / 2) {
#pragma unroll
        for (int a = 0; a < 2; ++a) af[w ^ 1][a] = cur[aoff + (kk + 1) * 2 * kG1LDP + a * 32];
#pragma unroll
        for (int c = 0; c < 2; ++c) bf[w ^ 1][c] = cur[boff + (kk + 1) * 2 * kG1LDP + c * 32];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[w][a], bf[w][c], acc[a][c], 0, 0, 0);
    }
    if (has_next) store_stage(smem + ((st + 1) & 1) * kG1Stage);
    __syncthreads();
  }

  // lane owns input channel ci (column), 16 output channels (rows) per block -- as wgrad_kernel's epilogue
  if (p.splits == 1) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ci = ci0 + wn0 + c * 32 + li;
      if (ci >= p.cin_store) continue;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (co >= p.Cout) continue;
          const size_t idx = p.transpose_out ? (size_t)ci * p.Cout + co : (size_t)co * p.cin_store + ci;
          p.dw[idx] += acc[a][c][r] * p.scale;
        }
    }
    return;
  }
  float* slab = p.scratch + ((size_t)split * ntiles + tile) * (size_t)(128 * 128);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = wm0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        slab[(size_t)col * 128 + wn0 + c * 32 + li] = acc[a][c][r];
      }
}

// dw[co, ci, tap] += scale * sum_splits slab[split][tile][tap][co_l][ci_l]; thread = one (tile, tap, co_l, ci_l)
template <int T, int BCO, int BCI>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgParams p) {
  const int ntiles = p.co_tiles * p.ci_tiles;
  const size_t per_tile = (size_t)T * BCO * BCI;
  const size_t total = per_tile * ntiles;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int tile = (int)(idx / per_tile);
    const size_t rem = idx - (size_t)tile * per_tile;
    const int ci_l = (int)(rem % BCI);
    const int co_l = (int)((rem / BCI) % BCO);
    const int t = (int)(rem / ((size_t)BCI * BCO));
    const int co = (tile / p.ci_tiles) * BCO + co_l, ci = (tile % p.ci_tiles) * BCI + ci_l;
    if (co >= p.Cout || ci >= p.cin_store) continue;
    // eight slabs in flight per thread (two were 2.3 TB/s on the 100-200 slabs of the pipelined 1x1 kernel: a chain of
    // exposed load latencies); fixed order -> deterministic
    float sa[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) sa[u] = 0.f;
    int sp = 0;
    for (; sp + 8 <= p.splits; sp += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p.scratch[(size_t)(sp + u) * total + idx];
#pragma unroll
      for (int u = 0; u < 8; ++u) sa[u] += v[u];
    }
    float tail = 0.f;
    for (; sp < p.splits; ++sp) tail += p.scratch[(size_t)sp * total + idx];
    const float sum = (((sa[0] + sa[1]) + (sa[2] + sa[3])) + ((sa[4] + sa[5]) + (sa[6] + sa[7]))) + tail;
    const size_t o = p.transpose_out ? (size_t)ci * p.Cout + co : ((size_t)co * p.cin_store + ci) * T + t;
    p.dw[o] += sum * p.scale;
  }
}

template <int KS, int CO_B, int CI_B>
int launch(const WgParams& p, int lds_bytes, hipStream_t st) {
  constexpr int BCO = 2 * CO_B * 32, BCI = 2 * CI_B * 32;
  auto kfn = wgrad_kernel<KS, CO_B, CI_B>;
  static std::atomic<bool> attr_set{false};   // once per instantiation, before any stream capture
  if (!attr_set) {
    SSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int ntiles = p.co_tiles * p.ci_tiles;
  hipLaunchKernelGGL(kfn, dim3(ssde_cdiv(p.splits, 8) * 8 * ntiles), dim3(kThreads), lds_bytes, st, p);
  SSDE_LAUNCH_CHECK();
  if (p.splits > 1) {
    const size_t total = (size_t)KS * KS * BCO * BCI * ntiles;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((wgrad_reduce_kernel<KS * KS, BCO, BCI>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    SSDE_LAUNCH_CHECK();
  }
  return SSDE_OK;
}

int pow2_floor(int v) { int q = 1; while (q * 2 <= v) q *= 2; return q; }

struct WgPlan { WgParams p; int lds; int64_t slab_floats; int pref_splits; };

// 1x1 / NIN / Linear gradients go to wgrad1x1_gemm_kernel (SSDE_WGRADF_1X1_CHUNKED: the chunked kernel)
bool pipelined_1x1(const ssde_wgrad_args* a) {
  return a->ksize == 1 && !(a->flags & SSDE_WGRADF_1X1_CHUNKED);
}

int make_plan(const ssde_wgrad_args* a, WgPlan* pl) {
  SSDE_REQUIRE(a && a->ksize != 0, "wgrad: null args");
  SSDE_REQUIRE(a->ksize == 3 || a->ksize == 1, "wgrad: ksize must be 1 or 3");
  SSDE_REQUIRE(a->stride == 1 || a->stride == 2, "wgrad: stride must be 1 or 2");
  const ssde_src& s = a->src;
  const int Ctot = s.c0 + s.c1;
  SSDE_REQUIRE(s.c0 > 0 && s.c0 % 4 == 0 && s.c1 % 4 == 0, "wgrad: source channels must be multiples of 4");
  SSDE_REQUIRE(a->cin_store > 0 && a->cin_store <= Ctot, "wgrad: cin_store %d outside 1..%d", a->cin_store, Ctot);
  SSDE_REQUIRE(a->g_ld % 4 == 0 && a->g_off % 4 == 0 && a->g_off + a->c_out <= a->g_ld + 3, "wgrad: bad g columns");
  SSDE_REQUIRE(a->n > 0 && a->h_out > 0 && a->w_out > 0 && a->c_out > 0, "wgrad: bad shape");
  SSDE_REQUIRE(!a->transpose_out || a->ksize == 1, "wgrad: transpose_out needs ksize 1");
  if (a->ksize == 1) SSDE_REQUIRE(a->h_in == a->h_out && a->w_in == a->w_out && a->stride == 1 && a->pad == 0, "wgrad: 1x1 geometry");
  else SSDE_REQUIRE((a->h_in + 2 * a->pad - 3) / a->stride + 1 >= a->h_out && (a->w_in + 2 * a->pad - 3) / a->stride + 1 >= a->w_out,
                    "wgrad: output larger than the convolution produces");
  if (s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU)
    SSDE_REQUIRE(s.gn_groups > 0 && Ctot % s.gn_groups == 0 && (Ctot / s.gn_groups) % 4 == 0, "wgrad: GroupNorm channels-per-group %% 4");
  WgParams& p = pl->p;
  p.src = s; p.g = a->g; p.g_ld = a->g_ld; p.g_off = a->g_off;
  p.N = a->n; p.Hin = a->h_in; p.Win = a->w_in; p.Hout = a->h_out; p.Wout = a->w_out;
  p.Cout = a->c_out; p.Ctot = Ctot; p.cin_store = a->cin_store;
  p.stride = a->stride; p.pad = a->pad; p.transpose_out = a->transpose_out; p.scale = a->scale; p.dw = a->dw;
  p.scratch = a->scratch;
  const int hw = a->h_out * a->w_out;
  int px;
  if (a->ksize == 1) px = 64;
  else if (a->stride == 2) px = 32;
  else px = hw >= 128 ? 128 : 64;
  const int tw = pow2_floor(a->w_out < 16 ? a->w_out : 16);
  int th = px / tw; if (th > a->h_out) th = a->h_out;
  th = pow2_floor(th);
  const int imgs = px / (tw * th);
  p.lTW = ssde_ilog2(tw); p.lTH = ssde_ilog2(th); p.lPX = ssde_ilog2(px);
  p.tiles_x = ssde_cdiv(a->w_out, tw);
  p.tiles_per_img = p.tiles_x * ssde_cdiv(a->h_out, th);
  p.chunks = ssde_cdiv(a->n, imgs) * p.tiles_per_img;
  const int bco = a->ksize == 3 ? 64 : 128, bci = bco;
  p.co_tiles = ssde_cdiv(a->c_out, bco);
  p.ci_tiles = ssde_cdiv(a->cin_store, bci);
  const int ntiles = p.co_tiles * p.ci_tiles;
  pl->slab_floats = (int64_t)a->ksize * a->ksize * bco * bci * ntiles;
  // ~2 workgroups per CU; a split costs one slab write + read, so keep >= 2 chunks per workgroup
  int splits = ssde_cdiv(512, ntiles);
  const int max_splits = p.chunks >= 2 ? p.chunks / 2 : 1;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  pl->pref_splits = splits;
  if (a->splits > 0) splits = a->splits;
  if (splits > p.chunks) splits = p.chunks;
  if (splits > 1 && a->scratch_floats < pl->slab_floats * splits) {
    splits = (int)(a->scratch_floats / pl->slab_floats);     // fewer, longer workgroups when scratch is short
    if (splits < 1) splits = 1;
  }
  p.chunks_per_split = ssde_cdiv(p.chunks, splits);
  p.splits = ssde_cdiv(p.chunks, p.chunks_per_split);
  p.rows = a->n * hw;
  p.rows_per_split = 0;
  if (pipelined_1x1(a)) {
    // the pipelined GEMM: ~768 workgroups (3 of the 4 a CU holds: the sweep in profiles/r3_wgrad1x1_ab.txt peaks at
    // 768-1024 for every shape of the CIFAR network), at least 8 stages of 16 pixels each
    const int stages = ssde_cdiv(p.rows, kG1BK);
    int sp = ssde_cdiv(768, ntiles);
    if (sp > stages / 8) sp = stages / 8;
    if (sp < 1) sp = 1;
    pl->pref_splits = sp;
    if (a->splits > 0) sp = a->splits;
    if (sp > stages) sp = stages;
    if (sp > 1 && a->scratch_floats < pl->slab_floats * sp) { sp = (int)(a->scratch_floats / pl->slab_floats); if (sp < 1) sp = 1; }
    p.rows_per_split = ssde_cdiv(stages, sp) * kG1BK;
    p.splits = ssde_cdiv(p.rows, p.rows_per_split);
  }
  const int ks = a->ksize;
  const int halo = imgs * ((th - 1) * a->stride + ks) * ((tw - 1) * a->stride + ks);
  pl->lds = (px * bco + halo * bci) * 4;
  SSDE_REQUIRE(pl->lds <= 160 * 1024, "wgrad: %d bytes of LDS needed", pl->lds);
  return SSDE_OK;
}

}  // namespace

extern "C" int ssde_wgrad_wants_winograd4(const ssde_wgrad_args* a) { return a && ssde_wgrad_wino4_wants(a) ? 1 : 0; }

extern "C" int64_t ssde_wgrad_scratch_floats(const ssde_wgrad_args* a) {
  if (a && ssde_wgrad_wino4_wants(a)) return ssde_wgrad_wino4_scratch_floats(a);    // Winograd F(4x4,3x3): wgrad_wino4.hip
  if (a && ssde_wgrad_wino_wants(a)) return ssde_wgrad_wino_scratch_floats(a);      // Winograd F(2x2,3x3): wgrad_wino.hip
  WgPlan pl;
  if (int rc = make_plan(a, &pl)) return rc;
  return pl.pref_splits > 1 ? pl.slab_floats * pl.pref_splits : 0;
}

extern "C" int ssde_conv_wgrad(const ssde_wgrad_args* a, void* stream) {
  if (a && ssde_wgrad_wino4_wants(a)) return ssde_wgrad_wino4_launch(a, stream);
  if (a && ssde_wgrad_wino_wants(a)) return ssde_wgrad_wino_launch(a, stream);
  WgPlan pl;
  if (int rc = make_plan(a, &pl)) return rc;
  const ssde_src& s = a->src;
  SSDE_REQUIRE(a->g && a->dw && s.p0 && (s.c1 == 0 || s.p1), "wgrad: null tensors");
  if (s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU)
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "wgrad: GroupNorm pointers missing");
  SSDE_REQUIRE(s.drop_thresh == 0 || s.drop_seed, "wgrad: dropout seed pointer missing");
  SSDE_REQUIRE(pl.p.splits == 1 || a->scratch, "wgrad: scratch missing");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a->ksize == 3) return launch<3, 1, 1>(pl.p, pl.lds, st);
  if (pl.p.rows_per_split > 0) {
    const int ntiles = pl.p.co_tiles * pl.p.ci_tiles;
    hipLaunchKernelGGL(wgrad1x1_gemm_kernel, dim3(ssde_cdiv(pl.p.splits, 8) * 8 * ntiles), dim3(kThreads), 2 * kG1Stage * 4, st, pl.p);
    SSDE_LAUNCH_CHECK();
    if (pl.p.splits > 1) {
      size_t blocks = ((size_t)128 * 128 * ntiles + 255) / 256;
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL((wgrad_reduce_kernel<1, 128, 128>), dim3((unsigned)blocks), dim3(256), 0, st, pl.p);
      SSDE_LAUNCH_CHECK();
    }
    return SSDE_OK;
  }
  return launch<1, 2, 2>(pl.p, pl.lds, st);
}
