// HBM-bound kernels of the training step: bias / temb-addend gradients (column sums), GroupNorm
// backward (reduction + apply, fused with SiLU' and the regenerated dropout mask), the DSM loss head
// (losses.py:84-99) and the fused clip + Adam + EMA update over flat buffers (losses.py:41-51,
// models/ema.py:46-51).  All tensors NHWC fp32 unless stated; every kernel reads each operand once
// with 16-byte lane accesses.
#include "ssde_common.h"

namespace {

inline unsigned grid_for(size_t total, int block = 256, unsigned cap = 256 * 16) {
  size_t b = (total + block - 1) / block;
  if (b < 1) b = 1;
  return (unsigned)(b > cap ? cap : b);
}

// ---- column sums ------------------------------------------------------------------------------
// grid (ceil(CL/64), slices, N): block = 64 float4 channel lanes x 4 pixel lanes (one wave per pixel lane)
// over the pixels of one slice of one sample; out row = n * slices + slice
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ g, int g_ld, int g_off, int hw, int c,
                                                     float scale, float* __restrict__ out, int out_ld, int out_off) {
  SSDE_LDS(smem);                                      // [4][64] float4
  float4* red = reinterpret_cast<float4*>(smem);
  const int n = blockIdx.z, slices = gridDim.y, slice = blockIdx.y;
  const int cl = blockIdx.x * 64 + (threadIdx.x & 63);
  const int pl = threadIdx.x >> 6;
  const int col = g_off + cl * 4;
  const int per_slice = (hw + slices - 1) / slices;
  const int p0 = slice * per_slice, p1 = min(hw, p0 + per_slice);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cl * 4 < c && col + 4 <= g_ld) {
    const float* base = g + (size_t)n * hw * g_ld + col;
    int px = p0 + pl;
    for (; px + 12 < p1; px += 16) {
      const float4 a = *reinterpret_cast<const float4*>(base + (size_t)px * g_ld);
      const float4 b = *reinterpret_cast<const float4*>(base + (size_t)(px + 4) * g_ld);
      const float4 d = *reinterpret_cast<const float4*>(base + (size_t)(px + 8) * g_ld);
      const float4 e = *reinterpret_cast<const float4*>(base + (size_t)(px + 12) * g_ld);
      s.x += (a.x + b.x) + (d.x + e.x); s.y += (a.y + b.y) + (d.y + e.y);
      s.z += (a.z + b.z) + (d.z + e.z); s.w += (a.w + b.w) + (d.w + e.w);
    }
    for (; px < p1; px += 4) {
      const float4 a = *reinterpret_cast<const float4*>(base + (size_t)px * g_ld);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (pl == 0 && cl * 4 < c) {
    const float4 a = red[threadIdx.x], b = red[threadIdx.x + 64], d = red[threadIdx.x + 128], e = red[threadIdx.x + 192];
    const float v[4] = {(a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z), (a.w + b.w) + (d.w + e.w)};
    for (int k = 0; k < 4; ++k)
      if (cl * 4 + k < c) out[((size_t)n * slices + slice) * out_ld + out_off + cl * 4 + k] = v[k] * scale;
  }
}

// per[n, off + j] = sum_slices part[n * slices + s, j]   (fixed order)
__global__ void colsum_slices_kernel(const float* __restrict__ part, int slices, int c, float* __restrict__ per, int ld, int off) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (j >= c) return;
  float s = 0.f;
  for (int i = 0; i < slices; ++i) s += part[((size_t)n * slices + i) * c + j];
  per[(size_t)n * ld + off + j] = s;
}

// total[j] = sum_n per[n, off + j].  Block = 32 channels x 8 sample lanes (a serial loop over the samples is a chain
// of dependent L2 latencies: 28 us for 128 samples); fixed summation order -> deterministic.
__global__ __launch_bounds__(256) void colsum_total_kernel(const float* __restrict__ per, int ld, int off, int n, int c,
                                                           float* __restrict__ t0, float* __restrict__ t1) {
  SSDE_LDS(smem);                                        // [8][32]
  const int jl = threadIdx.x & 31, nl = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + jl;
  float s = 0.f;
  if (j < c)
    for (int i = nl; i < n; i += 8) s += per[(size_t)i * ld + off + j];
  smem[nl * 32 + jl] = s;
  __syncthreads();
  if (nl == 0 && j < c) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += smem[k * 32 + jl];
    t0[j] = t;
    if (t1) t1[j] = t;
  }
}

// ---- finishing launches for up to SSDE_FINISH_JOBS deferred calls at once (job = blockIdx.y) -----------------------------
// colsum: what colsum_slices_kernel + colsum_total_kernel do for one call, in the same order -- per[n] = the slices of sample n
// in order, total = the samples i = lane, lane + 8, ... per lane, the 8 lanes in order -- so the results are bit-identical.
__global__ __launch_bounds__(256) void colsum_finish_kernel(const ssde_colsum_finish_args a) {
  SSDE_LDS(smem);                                        // [8][32]
  const ssde_colsum_job& j = a.job[blockIdx.y];
  if ((int)blockIdx.x * 32 >= j.c) return;               // (uniform per block: the grid is as wide as the widest job)
  const int jl = threadIdx.x & 31, nl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + jl;
  // (four samples of a lane at a time: their loads are independent and go out together -- as one sample per trip the 16 trips x
  //  `slices` loads of a lane at batch 128 were one chain of L2 latencies, 52 us per launch, 9 launches per training step --;
  //  every sample still sums its slices in order and the lane its samples in order: the same bits)
  float s = 0.f;
#if defined(SSDE_COLSUM_FINISH_SERIAL) && SSDE_COLSUM_FINISH_SERIAL       // (an A/B variant only: the loop of rounds 4-5)
  if (col < j.c)
    for (int i = nl; i < j.n; i += 8) {
      const float* p = j.part + (size_t)i * j.slices * j.c + col;
      float t = 0.f;
      for (int sl = 0; sl < j.slices; ++sl) t += p[(size_t)sl * j.c];
      if (j.per_sample) j.per_sample[(size_t)i * j.ps_ld + j.ps_off + col] = t;
      s += t;
    }
#else
  if (col < j.c)
    for (int i0 = nl; i0 < j.n; i0 += 32) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      const float* p[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) p[k] = j.part + (size_t)min(i0 + 8 * k, j.n - 1) * j.slices * j.c + col;
      for (int sl = 0; sl < j.slices; ++sl) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = p[k][(size_t)sl * j.c];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] += v[k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + 8 * k;
        if (i < j.n) {
          if (j.per_sample) j.per_sample[(size_t)i * j.ps_ld + j.ps_off + col] = t[k];
          s += t[k];
        }
      }
    }
#endif
  if (!j.total) return;
  smem[nl * 32 + jl] = s;
  __syncthreads();
  if (nl == 0 && col < j.c) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += smem[k * 32 + jl];
    j.total[col] = t;
    if (j.total2) j.total2[col] = t;
  }
}

// GroupNorm backward: the dgamma / dbeta role of gn_bwd_finalize_kernel (same order)
__global__ __launch_bounds__(256) void gn_bwd_finish_kernel(const ssde_gn_bwd_finish_args a) {
  SSDE_LDS(smem);                                        // [2][8][32]
  const ssde_gn_bwd_job& j = a.job[blockIdx.y];
  if ((int)blockIdx.x * 32 >= j.c) return;
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float dg = 0.f, db = 0.f;
  if (c < j.c)
    for (int r = rl; r < j.rows; r += 8) {
      const float2 o = *reinterpret_cast<const float2*>(j.scratch + ((size_t)r * j.c + c) * 2);
      dg += o.x; db += o.y;
    }
  smem[rl * 32 + cl] = dg;
  smem[256 + rl * 32 + cl] = db;
  __syncthreads();
  if (rl == 0 && c < j.c) {
    float tg = 0.f, tb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { tg += smem[k * 32 + cl]; tb += smem[256 + k * 32 + cl]; }
    j.dgamma[c] = tg;
    j.dbeta[c] = tb;
  }
}

// ---- GroupNorm backward: reduction -----------------------------------------------------------------
constexpr int kGbThreads = 1024;

struct GbParams {
  ssde_src src; const float* dp; int n, hw, slices;
  float* sums; float* dgamma; float* dbeta; float* scratch;
};

// grid (slices, N): thread (pl, cl) walks pixels pl, pl+PL, ... of float4 channel lane cl and accumulates,
// per channel, S1 = sum du*xhat and S0 = sum du, du = dp * keep * silu'(u).  scratch[n][slice][C][2].
__global__ __launch_bounds__(kGbThreads) void gn_bwd_reduce_kernel(const GbParams p) {
  SSDE_LDS(smem);                                     // [1024][8]
  const int n = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x;
  const ssde_src& s = p.src;
  const int C = s.c0 + s.c1, CL = C >> 2, PL = kGbThreads / CL;
  const int cpg = C / s.gn_groups;
  const SsdePro pro = ssde_pro_decode(s);
  const int per_slice = (p.hw + p.slices - 1) / p.slices;
  const int px0 = slice * per_slice, px1 = min(p.hw, px0 + per_slice);
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a0[4] = {0.f, 0.f, 0.f, 0.f};
  if (tid < CL * PL) {
    const int cl = tid % CL, pl = tid / CL, ch = cl * 4;
    const float* base; int Cs, cc;
    if (ch < s.c0) { base = s.p0; Cs = s.c0; cc = ch; } else { base = s.p1; Cs = s.c1; cc = ch - s.c0; }
    const int g = ch / cpg;
    const float mu = s.gn_mean[n * s.gn_groups + g], rs = s.gn_rstd[n * s.gn_groups + g];
    const float4 gam = *reinterpret_cast<const float4*>(s.gn_gamma + ch);
    const float4 bet = *reinterpret_cast<const float4*>(s.gn_beta + ch);
    const float gm[4] = {gam.x, gam.y, gam.z, gam.w}, bt[4] = {bet.x, bet.y, bet.z, bet.w};
    for (int px = px0 + pl; px < px1; px += PL) {
      const size_t pix = (size_t)n * p.hw + px;
      const float4 xv = *reinterpret_cast<const float4*>(base + pix * Cs + cc);
      const float4 dv = *reinterpret_cast<const float4*>(p.dp + pix * C + ch);
      const float x[4] = {xv.x, xv.y, xv.z, xv.w}, d[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (x[k] - mu) * rs;
        float du = d[k];
        if (pro.silu) du *= ssde_silu_grad(xh * gm[k] + bt[k]);
        if (pro.drop) du *= ssde_keep((uint32_t)pix * (uint32_t)C + (uint32_t)(ch + k), pro);
        a1[k] += du * xh;
        a0[k] += du;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { smem[tid * 8 + k] = a1[k]; smem[tid * 8 + 4 + k] = a0[k]; }
  __syncthreads();
  if (tid < CL) {
    float t1[4] = {0.f, 0.f, 0.f, 0.f}, t0[4] = {0.f, 0.f, 0.f, 0.f};
    for (int pl = 0; pl < PL; ++pl)
#pragma unroll
      for (int k = 0; k < 4; ++k) { t1[k] += smem[(pl * CL + tid) * 8 + k]; t0[k] += smem[(pl * CL + tid) * 8 + 4 + k]; }
    float* o = p.scratch + (((size_t)n * p.slices + slice) * C + tid * 4) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[2 * k] = t1[k]; o[2 * k + 1] = t0[k]; }
  }
}

// role A (blockIdx.y == 0): sums[n][g] = (mean_g dxh, mean_g dxh*xhat), dxh = du*gamma; one thread per (n, g)
// role B (blockIdx.y == 1): dgamma[c], dbeta[c] = sum over samples and slices; block = 32 channels x 8 (sample,slice)
//                            lanes + an LDS tree (a serial loop over 128 samples x slices was 45 us of latency)
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const GbParams p, const int role0) {
  SSDE_LDS(smem);                                        // [2][8][32]
  const ssde_src& s = p.src;
  const int C = s.c0 + s.c1, G = s.gn_groups, cpg = C / G;
  if ((int)blockIdx.y + role0 == 0) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.n * G) return;
    const int n = idx / G, g = idx % G;
    float A = 0.f, B = 0.f;
    for (int sl = 0; sl < p.slices; ++sl) {
      const float* o = p.scratch + (((size_t)n * p.slices + sl) * C + g * cpg) * 2;
      for (int c = 0; c < cpg; ++c) { const float gm = s.gn_gamma[g * cpg + c]; B += gm * o[2 * c]; A += gm * o[2 * c + 1]; }
    }
    const float inv = 1.0f / ((float)cpg * (float)p.hw);
    p.sums[idx * 2] = A * inv;
    p.sums[idx * 2 + 1] = B * inv;
  } else {
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float dg = 0.f, db = 0.f;
    if (c < C) {
      const int rows = p.n * p.slices;
      for (int r = rl; r < rows; r += 8) {
        const float2 o = *reinterpret_cast<const float2*>(p.scratch + ((size_t)r * C + c) * 2);
        dg += o.x; db += o.y;
      }
    }
    smem[rl * 32 + cl] = dg;
    smem[256 + rl * 32 + cl] = db;
    __syncthreads();
    if (rl == 0 && c < C) {
      float tg = 0.f, tb = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) { tg += smem[k * 32 + cl]; tb += smem[256 + k * 32 + cl]; }
      p.dgamma[c] = tg;
      p.dbeta[c] = tb;
    }
  }
}

// ---- GroupNorm backward in ONE pass over dp and x (ABI 7: ssde_gn_bwd_reduce_args.g0 / g1) -------------------------------
// The reduction above and prologue_bwd_kernel below each read dp and x once: five tensor passes and three launches per
// GroupNorm.  A (sample, group) is small -- 32x32 pixels x 4 channels is 16 KB per tensor -- so a 1024-thread workgroup
// that owns one sample and a run of whole groups keeps its share of BOTH tensors in registers (<= 8 pixels x one channel
// quad of dp and x per thread = 64 registers, 256 KB per workgroup), reduces the two per-channel sums through LDS in a
// fixed order, and applies  dx = rstd (du gamma - A - xhat B)  to the values it still holds: dp and x are read once, three
// tensor passes (four where the gradient accumulates) and one launch; the per-channel sums go to scratch[n][C][2] for the
// dgamma / dbeta role of gn_bwd_finalize_kernel.  Maps whose (sample, group) does not fit (hw > 8192 at 4 channels per
// group: the 128x128 and 256x256 levels of FFHQ) stay on the three-kernel path.
constexpr int kGfMaxThreads = 1024, kGfR = 8;
constexpr int kGfMaxClc = 256;                      // channel quads per workgroup (>= 4 pixel lanes)
constexpr int kGfNxt = kGfMaxThreads + 8 * kGfMaxClc;  // floats of the second reduction area: ceil(PL / 8) rows of 8 clc values

struct GfParams {
  GbParams b; int acc0, acc1; float* g0; float* g1; float scale;
  int gpc, clc, pl;      // groups per workgroup, channel quads per workgroup (= gpc * cpg / 4), pixel lanes (= 1024 / clc)
};

template <int kGfThreads>
__global__ __launch_bounds__(kGfThreads, 4) void gn_bwd_fused_kernel(const GfParams p) {
  SSDE_LDS(smem);                                     // [1024][8] partial sums | [<= 1024 + 64][8 clc / clc] | group sums
  const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const ssde_src& s = p.b.src;
  const int C = s.c0 + s.c1, G = s.gn_groups, cpg = C / G, hw = p.b.hw;
  const int CLc = p.clc, PL = p.pl, T = CLc * 8;
  const SsdePro pro = ssde_pro_decode(s);
  const int cl = tid % CLc, pl = tid / CLc;
  const int ch0 = chunk * p.gpc * cpg, ch = ch0 + cl * 4;
  const bool live = pl < PL && ch < C;
  const bool first = ch < s.c0;
  const int Cs = first ? s.c0 : s.c1, cc = first ? ch : ch - s.c0;
  const float* xsrc = (first ? s.p0 : s.p1) + cc;
  float* gdst = first ? p.g0 : p.g1;
  const bool acc = first ? p.acc0 : p.acc1;
  float gm[4] = {0.f, 0.f, 0.f, 0.f}, bt[4] = {0.f, 0.f, 0.f, 0.f};
  float mu = 0.f, rs = 0.f;
  if (live) {
    const float4 gam = *reinterpret_cast<const float4*>(s.gn_gamma + ch);
    const float4 bet = *reinterpret_cast<const float4*>(s.gn_beta + ch);
    gm[0] = gam.x; gm[1] = gam.y; gm[2] = gam.z; gm[3] = gam.w;
    bt[0] = bet.x; bt[1] = bet.y; bt[2] = bet.z; bt[3] = bet.w;
    mu = s.gn_mean[n * G + ch / cpg]; rs = s.gn_rstd[n * G + ch / cpg];
  }
  // every load of the thread in flight before the first use
  float4 dv[kGfR], xv[kGfR];
#pragma unroll
  for (int r = 0; r < kGfR; ++r) {
    const int px = pl + r * PL;
    dv[r] = make_float4(0.f, 0.f, 0.f, 0.f); xv[r] = dv[r];
    if (live && px < hw) {
      const size_t pix = (size_t)n * hw + px;
      dv[r] = *reinterpret_cast<const float4*>(p.b.dp + pix * C + ch);
      xv[r] = *reinterpret_cast<const float4*>(xsrc + pix * Cs);
    }
  }
  // du = dp * keep * silu'(u) and xhat replace dp and x in the registers
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < kGfR; ++r) {
    const int px = pl + r * PL;
    if (!(live && px < hw)) continue;
    const uint32_t pix = (uint32_t)n * (uint32_t)hw + (uint32_t)px;
    float d[4] = {dv[r].x, dv[r].y, dv[r].z, dv[r].w}, x[4] = {xv[r].x, xv[r].y, xv[r].z, xv[r].w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (x[k] - mu) * rs;
      float du = d[k];
      if (pro.silu) du *= ssde_silu_grad(xh * gm[k] + bt[k]);
      if (pro.drop) du *= ssde_keep(pix * (uint32_t)C + (uint32_t)(ch + k), pro);
      a1[k] += du * xh;
      a0[k] += du;
      d[k] = du; x[k] = xh;
    }
    dv[r] = make_float4(d[0], d[1], d[2], d[3]); xv[r] = make_float4(x[0], x[1], x[2], x[3]);
  }
  // rows = pixel lanes, T = 8 clc values per row (S1[4], S0[4] of every channel quad); eight rows per thread and round,
  // the grouping depends on (PL, T) only
  float* cur = smem;
  float* nxt = smem + kGfThreads * 8;
  if (pl < PL) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { cur[pl * T + cl * 8 + k] = a1[k]; cur[pl * T + cl * 8 + 4 + k] = a0[k]; }
  }
  __syncthreads();
  int rows = PL;
  while (rows > 1) {
    const int J = (rows + 7) >> 3;
    for (int t = tid; t < J * T; t += kGfThreads) {
      const int j = t / T, v = t - j * T;
      float a = 0.f;
      for (int r = j; r < rows; r += J) a += cur[r * T + v];
      nxt[t] = a;
    }
    __syncthreads();
    float* sw = cur; cur = nxt; nxt = sw;
    rows = J;
  }
  // cur[q * 8 + k] = S1 = sum du xhat, cur[q * 8 + 4 + k] = S0 = sum du of channel ch0 + 4 q + k
  float* gsum = smem + kGfThreads * 8 + kGfNxt;                    // behind both ping-pong areas
  if (tid < CLc * 4 && ch0 + tid < C) {
    const int q = tid >> 2, k = tid & 3;
    *reinterpret_cast<float2*>(p.b.scratch + ((size_t)n * C + ch0 + tid) * 2) = make_float2(cur[q * 8 + k], cur[q * 8 + 4 + k]);
  }
  if (tid < p.gpc && chunk * p.gpc + tid < G) {
    float A = 0.f, B = 0.f;
    for (int c = 0; c < cpg; ++c) {
      const int lc = tid * cpg + c;
      const float g = s.gn_gamma[ch0 + lc];
      B += g * cur[(lc >> 2) * 8 + (lc & 3)];
      A += g * cur[(lc >> 2) * 8 + 4 + (lc & 3)];
    }
    const float inv = 1.0f / ((float)cpg * (float)hw);
    gsum[tid * 2] = A * inv;
    gsum[tid * 2 + 1] = B * inv;
    if (p.b.sums) { p.b.sums[(n * G + chunk * p.gpc + tid) * 2] = A * inv; p.b.sums[(n * G + chunk * p.gpc + tid) * 2 + 1] = B * inv; }
  }
  __syncthreads();
  if (!live || !gdst) return;
  const float A = gsum[(cl * 4 / cpg) * 2], B = gsum[(cl * 4 / cpg) * 2 + 1];
  constexpr int kB = 4;                                // rows whose old gradient is in flight together
#pragma unroll
  for (int r0 = 0; r0 < kGfR; r0 += kB) {
    float4 old[kB];
#pragma unroll
    for (int b = 0; b < kB; ++b) {
      const int px = pl + (r0 + b) * PL;
      old[b] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (acc && px < hw) old[b] = *reinterpret_cast<const float4*>(gdst + ((size_t)n * hw + px) * Cs + cc);
    }
    // values of the whole batch first, then its stores back to back: with the bounds test, the arithmetic and the store of a row
    // in one loop trip hipcc put an s_waitcnt vmcnt(0) between the stores (gfx9 counts stores in vmcnt; the count state it
    // merges at the joins of the skipped rows is "unknown"), and every store waited out its predecessor's acknowledgement
    float4 w[kB];
#pragma unroll
    for (int b = 0; b < kB; ++b) {
      const int r = r0 + b;
      const float du[4] = {dv[r].x, dv[r].y, dv[r].z, dv[r].w}, xh[4] = {xv[r].x, xv[r].y, xv[r].z, xv[r].w};
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = rs * (du[k] * gm[k] - A - xh[k] * B) * p.scale;
      w[b] = make_float4(o[0], o[1], o[2], o[3]);
      if (acc) { w[b].x += old[b].x; w[b].y += old[b].y; w[b].z += old[b].z; w[b].w += old[b].w; }
    }
#pragma unroll
    for (int b = 0; b < kB; ++b) {
      const int px = pl + (r0 + b) * PL;
      if (px < hw) *reinterpret_cast<float4*>(gdst + ((size_t)n * hw + px) * Cs + cc) = w[b];
    }
  }
}

// ---- prologue backward: apply ---------------------------------------------------------------------------
struct PbParams {
  ssde_src src; const float* dp; int dp_ld, dp_off; int n, hw; const float* sums; float scale;
  int acc0, acc1; float* g0; float* g1;
};

// A thread keeps ONE channel quad for the whole kernel (the launcher makes the grid's thread count a multiple of C/4) and
// walks pixels px_step apart: gamma / beta / group index / source pointers are loop invariants, the loop body has no
// 64-bit division (the flat-index form spent more instructions on idx / CL, idx % CL and pix / hw than on the arithmetic),
// and two pixels are in flight per trip.
__global__ __launch_bounds__(256) void prologue_bwd_kernel(const PbParams p, const unsigned px_step) {
  const ssde_src& s = p.src;
  const int C = s.c0 + s.c1, CL = C >> 2;
  const SsdePro pro = ssde_pro_decode(s);
  const int cpg = pro.gn ? C / s.gn_groups : 1;
  const unsigned gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const int cl = (int)(gtid % (unsigned)CL);
  const unsigned pix0 = gtid / (unsigned)CL, npix = (unsigned)p.n * (unsigned)p.hw;
  const int ch = cl * 4;
  const bool first = ch < s.c0;
  float* gdst = first ? p.g0 : p.g1;
  if (!gdst) return;
  const int Cs = first ? s.c0 : s.c1, cc = first ? ch : ch - s.c0;
  const float* xsrc = (first ? s.p0 : s.p1) + cc;
  const float* dsrc = p.dp + p.dp_off + ch;
  const bool acc = first ? p.acc0 : p.acc1;
  const bool need_x = pro.gn || pro.silu;
  const int g = ch / cpg;
  float gm[4] = {1.f, 1.f, 1.f, 1.f}, bt[4] = {0.f, 0.f, 0.f, 0.f};
  if (pro.gn) {
    const float4 gam = *reinterpret_cast<const float4*>(s.gn_gamma + ch);
    const float4 bet = *reinterpret_cast<const float4*>(s.gn_beta + ch);
    gm[0] = gam.x; gm[1] = gam.y; gm[2] = gam.z; gm[3] = gam.w;
    bt[0] = bet.x; bt[1] = bet.y; bt[2] = bet.z; bt[3] = bet.w;
  }
  constexpr int U = 2;
  for (unsigned pb = pix0; pb < npix; pb += U * px_step) {
    float4 dv[U], xv[U], old[U];
    float mu[U], rs[U], A[U], B[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned pix = pb + u * px_step;
      ok[u] = pix < npix;
      const size_t px = ok[u] ? pix : pb;
      dv[u] = *reinterpret_cast<const float4*>(dsrc + px * p.dp_ld);
      xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (need_x) xv[u] = *reinterpret_cast<const float4*>(xsrc + px * Cs);
      old[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (acc) old[u] = *reinterpret_cast<const float4*>(gdst + px * Cs + cc);
      mu[u] = 0.f; rs[u] = 1.f; A[u] = 0.f; B[u] = 0.f;
      if (pro.gn) {
        const int gi = (int)((unsigned)px / (unsigned)p.hw) * s.gn_groups + g;
        mu[u] = s.gn_mean[gi]; rs[u] = s.gn_rstd[gi];
        A[u] = p.sums[gi * 2]; B[u] = p.sums[gi * 2 + 1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      const unsigned pix = pb + u * px_step;
      float d[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
      const float x[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
      if (pro.gn) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = (x[k] - mu[u]) * rs[u];
          float du = d[k];
          if (pro.silu) du *= ssde_silu_grad(xh * gm[k] + bt[k]);
          if (pro.drop) du *= ssde_keep(pix * (uint32_t)C + (uint32_t)(ch + k), pro);
          d[k] = rs[u] * (du * gm[k] - A[u] - xh * B[u]);
        }
      } else if (pro.silu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] *= ssde_silu_grad(x[k]);
      }
      float4 r = make_float4(d[0] * p.scale, d[1] * p.scale, d[2] * p.scale, d[3] * p.scale);
      if (acc) { r.x += old[u].x; r.y += old[u].y; r.z += old[u].z; r.w += old[u].w; }
      *reinterpret_cast<float4*>(gdst + (size_t)pix * Cs + cc) = r;
    }
  }
}

// ---- DSM loss head ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void perturb_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                                      const float* __restrict__ a, const float* __restrict__ s,
                                                      float* __restrict__ dst, int per, size_t numel) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / per);
    // mean + std[:, None, None, None] * z  (losses.py:86-87); mean = a[n] * x
    const float mean = a ? __fmul_rn(a[n], x[i]) : x[i];
    dst[i] = __fadd_rn(mean, __fmul_rn(s[n], z[i]));
  }
}

// one block per sample: losses[n] and dscore[n, :]
__global__ __launch_bounds__(256) void dsm_loss_kernel(const float* __restrict__ score, const float* __restrict__ z,
                                                       const float* __restrict__ s, const float* __restrict__ g2,
                                                       float* __restrict__ dscore, float* __restrict__ losses,
                                                       int n_total, int per, int reduce_mean, int lw, float grad_scale) {
  SSDE_LDS(smem);
  const int n = blockIdx.x, tid = threadIdx.x;
  const float sd = s[n];
  const float w = lw ? g2[n] : 1.0f;
  // d(mean_n losses)/d score = w * red' * 2 r * dr/dscore / N ; red = 0.5*sum (factor 1) or mean (factor 2/per)
  const float red = reduce_mean ? 2.0f / (float)per : 1.0f;
  const float gfac = w * red * (lw ? 1.0f : sd) / (float)n_total * grad_scale;
  float acc = 0.f;
  for (int i = tid; i < per; i += 256) {
    const size_t k = (size_t)n * per + i;
    const float r = lw ? __fadd_rn(score[k], __fdiv_rn(z[k], sd)) : __fadd_rn(__fmul_rn(score[k], sd), z[k]);
    acc += r * r;
    if (dscore) dscore[k] = r * gfac;
  }
  acc = ssde_wave_sum(acc);
  if ((tid & 63) == 0) smem[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    const float t = (smem[0] + smem[1]) + (smem[2] + smem[3]);
    losses[n] = (reduce_mean ? t / (float)per : 0.5f * t) * w;
  }
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  SSDE_LDS(smem);
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int i = tid; i < n; i += 256) acc += v[i];
  acc = ssde_wave_sum(acc);
  if ((tid & 63) == 0) smem[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) out[0] = ((smem[0] + smem[1]) + (smem[2] + smem[3])) / (float)n;
}

// ---- optimizer ------------------------------------------------------------------------------------------------
constexpr int kSumsqBlocks = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, size_t numel, float* __restrict__ partial) {
  SSDE_LDS(smem);
  const int tid = threadIdx.x;
  float acc = 0.f;
  const size_t n4 = numel >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0 && tid == 0) for (size_t i = n4 * 4; i < numel; ++i) acc += x[i] * x[i];
  acc = ssde_wave_sum(acc);
  if ((tid & 63) == 0) smem[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) partial[blockIdx.x] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  SSDE_LDS(smem);
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int i = tid; i < n; i += 256) acc += partial[i];
  acc = ssde_wave_sum(acc);
  if ((tid & 63) == 0) smem[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) out[0] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
}

// hyper: [0] lr [1] beta1 [2] beta2 [3] eps [4] weight_decay [5] grad_clip (<0: off) [6] 1-beta1^t [7] sqrt(1-beta2^t)
//        [8] EMA one_minus_decay
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float* __restrict__ ema, size_t numel,
                                                   const float* __restrict__ hyper, const float* __restrict__ gnorm_sq) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], clip = hyper[5];
  const float bc1 = hyper[6], bc2s = hyper[7], omd = hyper[8];
  float coef = 1.0f;
  if (gnorm_sq && clip >= 0.f) {
    // clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    const float c = clip / (sqrtf(gnorm_sq[0]) + 1e-6f);
    coef = c < 1.0f ? c : 1.0f;
  }
  const float step_size = lr / bc1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (size_t)gridDim.x * blockDim.x) {
    float pi = p[i];
    float gi = g[i] * coef;
    if (wd != 0.f) gi += wd * pi;
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);            // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(vi) / bc2s + eps;
    pi -= step_size * (mi / denom);                               // param.addcdiv_(exp_avg, denom, value=-step_size)
    m[i] = mi; v[i] = vi; p[i] = pi;
    if (ema) { const float e = ema[i]; ema[i] = e - omd * (e - pi); }   // s_param.sub_(one_minus_decay * (s_param - param))
  }
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                   float* __restrict__ dst, size_t numel, float alpha, int acc) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (size_t)gridDim.x * blockDim.x) {
    float v = alpha * x[i];
    if (gate) v *= ssde_silu_grad(gate[i]);
    dst[i] = acc ? dst[i] + v : v;
  }
}

int src_ok(const ssde_src& s, const char* who, bool need_x) {
  const int C = s.c0 + s.c1;
  SSDE_REQUIRE(C > 0 && s.c0 % 4 == 0 && s.c1 % 4 == 0, "%s: channels must be multiples of 4", who);
  SSDE_REQUIRE(!need_x || (s.p0 && (s.c1 == 0 || s.p1)), "%s: source tensors missing", who);
  if (s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU) {
    SSDE_REQUIRE(s.gn_groups > 0 && C % s.gn_groups == 0 && (C / s.gn_groups) % 4 == 0, "%s: GroupNorm channels-per-group %% 4", who);
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "%s: GroupNorm pointers missing", who);
  }
  SSDE_REQUIRE(s.drop_thresh == 0 || s.drop_seed, "%s: dropout seed pointer missing", who);
  return SSDE_OK;
}

}  // namespace

extern "C" int ssde_colsum(const ssde_colsum_args* a, void* stream) {
  SSDE_REQUIRE(a && a->g && a->n > 0 && a->hw > 0 && a->c > 0, "colsum: bad args");
  SSDE_REQUIRE(a->g_ld % 4 == 0 && a->g_off % 4 == 0, "colsum: g columns must be 16-byte aligned");
  const bool defer = (a->flags & SSDE_COLSUMF_DEFER) != 0;
  SSDE_REQUIRE(defer || a->per_sample || a->total, "colsum: nothing to compute");
  hipStream_t st = static_cast<hipStream_t>(stream);
  // pixel slices: >= 64 pixels per block, at most 32 slices; scratch holds [N*slices][c] partials (+ [N][c] when
  // no per_sample destination exists)
  int slices = a->hw / 64;
  if (slices > 32) slices = 32;
  if (slices < 1) slices = 1;
  SSDE_REQUIRE((slices == 1 && a->per_sample && !defer) || a->scratch, "colsum: scratch of N*(slices+1)*c floats needed (slices=%d)", slices);
  if (defer) {                                 // only the pass over g: [N][slices][c] partials for a later ssde_colsum_finish
    hipLaunchKernelGGL(colsum_kernel, dim3(ssde_cdiv(ssde_cdiv(a->c, 4), 64), slices, a->n), dim3(256), 4 * 64 * 16, st,
                       a->g, a->g_ld, a->g_off, a->hw, a->c, a->scale, a->scratch, a->c, 0);
    SSDE_LAUNCH_CHECK();
    return SSDE_OK;
  }
  float* per = a->per_sample ? a->per_sample : a->scratch + (size_t)a->n * slices * a->c;
  const int ld = a->per_sample ? a->ps_ld : a->c, off = a->per_sample ? a->ps_off : 0;
  const int cl = ssde_cdiv(a->c, 4);
  if (slices == 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3(ssde_cdiv(cl, 64), 1, a->n), dim3(256), 4 * 64 * 16, st,
                       a->g, a->g_ld, a->g_off, a->hw, a->c, a->scale, per, ld, off);
    SSDE_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(colsum_kernel, dim3(ssde_cdiv(cl, 64), slices, a->n), dim3(256), 4 * 64 * 16, st,
                       a->g, a->g_ld, a->g_off, a->hw, a->c, a->scale, a->scratch, a->c, 0);
    SSDE_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_slices_kernel, dim3(ssde_cdiv(a->c, 256), a->n), dim3(256), 0, st, a->scratch, slices, a->c, per, ld, off);
    SSDE_LAUNCH_CHECK();
  }
  if (a->total) {
    hipLaunchKernelGGL(colsum_total_kernel, dim3(ssde_cdiv(a->c, 32)), dim3(256), 8 * 32 * 4, st, per, ld, off, a->n, a->c, a->total, a->total2);
    SSDE_LAUNCH_CHECK();
  }
  return SSDE_OK;
}

extern "C" int ssde_colsum_finish(const ssde_colsum_finish_args* a, void* stream) {
  SSDE_REQUIRE(a && a->count > 0 && a->count <= SSDE_FINISH_JOBS, "colsum_finish: bad job count");
  int cmax = 0;
  for (int i = 0; i < a->count; ++i) {
    const ssde_colsum_job& j = a->job[i];
    SSDE_REQUIRE(j.part && (j.per_sample || j.total) && j.n > 0 && j.slices > 0 && j.c > 0, "colsum_finish: job %d is incomplete", i);
    SSDE_REQUIRE(!j.total2 || j.total, "colsum_finish: job %d has total2 without total", i);
    if (j.c > cmax) cmax = j.c;
  }
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(ssde_cdiv(cmax, 32), a->count), dim3(256), 8 * 32 * 4, static_cast<hipStream_t>(stream), *a);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

extern "C" int ssde_gn_bwd_finish(const ssde_gn_bwd_finish_args* a, void* stream) {
  SSDE_REQUIRE(a && a->count > 0 && a->count <= SSDE_FINISH_JOBS, "gn_bwd_finish: bad job count");
  int cmax = 0;
  for (int i = 0; i < a->count; ++i) {
    const ssde_gn_bwd_job& j = a->job[i];
    SSDE_REQUIRE(j.scratch && j.dgamma && j.dbeta && j.rows > 0 && j.c > 0, "gn_bwd_finish: job %d is incomplete", i);
    if (j.c > cmax) cmax = j.c;
  }
  hipLaunchKernelGGL(gn_bwd_finish_kernel, dim3(ssde_cdiv(cmax, 32), a->count), dim3(256), 2 * 8 * 32 * 4, static_cast<hipStream_t>(stream), *a);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

// Shape of the one-pass kernel for (n, hw, C, G), or false: the largest run of whole groups whose pixels fit kGfR per
// thread, narrowed while the launch has fewer than two workgroups per CU (the per-thread row count only falls).
static bool gn_bwd_fused_shape(int kGfThreads, int n, int hw, int C, int G, int* gpc_out, int* clc_out, int* pl_out) {
  const int cpg = C / G;
  if (cpg % 4 != 0) return false;
  int best = 0;
  for (int gpc = G; gpc >= 1; --gpc) {
    const int clc = gpc * cpg / 4;
    if (clc > kGfMaxClc) continue;
    const int pl = kGfThreads / clc;
    if (ssde_cdiv(hw, pl) > kGfR) continue;
    if (!best) best = gpc;
    if ((long)n * ssde_cdiv(G, gpc) >= 2L * ssde_num_cus()) { best = gpc; break; }
    best = gpc;
  }
  if (!best) return false;
  *gpc_out = best; *clc_out = best * cpg / 4; *pl_out = kGfThreads / *clc_out;
  return true;
}

// does this call take the one-pass kernel?  (shape and flags only)
static bool gn_bwd_one_pass(const ssde_gn_bwd_reduce_args* a, int* gpc, int* clc, int* pl) {
  const int C = a->src.c0 + a->src.c1;
  return (a->g0 || a->g1) && !(a->flags & SSDE_GNBWDF_THREE_KERNELS) && (size_t)a->n * a->hw < (1ull << 31) && a->src.gn_groups > 0 &&
         gn_bwd_fused_shape(1024, a->n, a->hw, C, a->src.gn_groups, gpc, clc, pl);
}

extern "C" int ssde_gn_bwd_scratch_rows(const ssde_gn_bwd_reduce_args* a) {
  if (!a) return 0;
  int gpc, clc, pl;
  return gn_bwd_one_pass(a, &gpc, &clc, &pl) ? a->n : a->n * (a->slices > 0 ? a->slices : 1);
}

extern "C" int ssde_gn_bwd_reduce(const ssde_gn_bwd_reduce_args* a, void* stream) {
  const bool defer = a && (a->flags & SSDE_GNBWDF_DEFER_PARAMS);
  SSDE_REQUIRE(a && a->dp && (defer || (a->dgamma && a->dbeta)) && a->scratch, "gn_bwd_reduce: null args");
  const bool apply = a->g0 || a->g1;
  SSDE_REQUIRE(apply || a->sums, "gn_bwd_reduce: neither sums nor gradient destinations");
  if (int rc = src_ok(a->src, "gn_bwd_reduce", true)) return rc;
  SSDE_REQUIRE(a->src.pro_mode == SSDE_PRO_GN || a->src.pro_mode == SSDE_PRO_GN_SILU, "gn_bwd_reduce: source has no GroupNorm prologue");
  const int C = a->src.c0 + a->src.c1;
  SSDE_REQUIRE(C <= 4 * kGbThreads && a->n > 0 && a->hw > 0, "gn_bwd_reduce: bad shape");
  SSDE_REQUIRE(!(a->g1 && a->src.c1 == 0), "gn_bwd_reduce: g1 without a second source");
  const int slices = a->slices > 0 ? a->slices : 1;
  GbParams p{a->src, a->dp, a->n, a->hw, slices, a->sums, a->dgamma, a->dbeta, a->scratch};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int bx_a = ssde_cdiv(a->n * a->src.gn_groups, 256), bx_b = ssde_cdiv(C, 32);
  if (apply) {
    // ABI 7: the gradient of the sources in the same call -- one pass where a (sample, run of groups) fits the registers
    // of a workgroup (SSDE_GNBWDF_THREE_KERNELS: always the three kernels, for A/B timing and the tests of both forms)
    // (512-thread workgroups, two per CU, measured slower: the step 0.0587 -> 0.0591 s, profiles/r4_gn_bwd_one_pass_ab.txt --
    // a workgroup's rows shrink to 64-byte runs of 4 channel quads)
    constexpr int nt = 1024;
    int gpc = 0, clc = 0, pl = 0;
    if (gn_bwd_one_pass(a, &gpc, &clc, &pl)) {
      GfParams f{p, a->acc0, a->acc1, a->g0, a->g1, a->scale, gpc, clc, pl};
      f.b.slices = 1;                                   // scratch[n][C][2]
      const dim3 grid(ssde_cdiv(a->src.gn_groups, gpc), a->n);
      const size_t lds = (size_t)(nt * 8 + kGfNxt + 2 * kGfMaxClc) * sizeof(float);
      hipLaunchKernelGGL(gn_bwd_fused_kernel<nt>, grid, dim3(nt), lds, st, f);
      SSDE_LAUNCH_CHECK();
      if (defer) return SSDE_OK;                        // dgamma / dbeta: a later ssde_gn_bwd_finish sums scratch[n][C][2]
      // dgamma / dbeta: role B of the finalize kernel only (grid.y == 1 with the role offset)
      hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(bx_b, 1), dim3(256), 2 * 8 * 32 * 4, st, f.b, 1);
      SSDE_LAUNCH_CHECK();
      return SSDE_OK;
    }
    SSDE_REQUIRE(a->sums, "gn_bwd_reduce: the three-kernel path needs the sums buffer");
  }
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(slices, a->n), dim3(kGbThreads), kGbThreads * 8 * 4, st, p);
  SSDE_LAUNCH_CHECK();
  if (defer) hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(bx_a, 1), dim3(256), 2 * 8 * 32 * 4, st, p, 0);      // the group sums only
  else hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(bx_a > bx_b ? bx_a : bx_b, 2), dim3(256), 2 * 8 * 32 * 4, st, p, 0);
  SSDE_LAUNCH_CHECK();
  if (apply) {
    ssde_prologue_bwd_args q{};
    q.src = a->src; q.dp = a->dp; q.dp_ld = C; q.dp_off = 0; q.n = a->n; q.hw = a->hw; q.sums = a->sums; q.scale = a->scale;
    q.acc0 = a->acc0; q.acc1 = a->acc1; q.g0 = a->g0; q.g1 = a->g1;
    return ssde_prologue_bwd(&q, stream);
  }
  return SSDE_OK;
}

extern "C" int ssde_prologue_bwd(const ssde_prologue_bwd_args* a, void* stream) {
  SSDE_REQUIRE(a && a->dp && (a->g0 || a->g1), "prologue_bwd: null args");
  const bool need_x = a->src.pro_mode != SSDE_PRO_NONE;
  if (int rc = src_ok(a->src, "prologue_bwd", need_x)) return rc;
  const bool gn = a->src.pro_mode == SSDE_PRO_GN || a->src.pro_mode == SSDE_PRO_GN_SILU;
  SSDE_REQUIRE(!gn || a->sums, "prologue_bwd: GroupNorm sums missing");
  SSDE_REQUIRE(a->dp_ld % 4 == 0 && a->dp_off % 4 == 0, "prologue_bwd: dp columns must be 16-byte aligned");
  SSDE_REQUIRE(a->n > 0 && a->hw > 0, "prologue_bwd: bad shape");
  PbParams p{a->src, a->dp, a->dp_ld, a->dp_off, a->n, a->hw, a->sums, a->scale, a->acc0, a->acc1, a->g0, a->g1};
  const int CL = (a->src.c0 + a->src.c1) / 4;
  const size_t total = (size_t)a->n * a->hw * CL;
  SSDE_REQUIRE(CL > 0 && (size_t)a->n * a->hw < (1ull << 31), "prologue_bwd: bad shape");
  // thread count = a multiple of CL: blocks in steps of CL / gcd(CL, 256)
  int gcd = CL, b256 = 256;
  while (b256) { const int t = gcd % b256; gcd = b256; b256 = t; }
  const unsigned unit = (unsigned)(CL / gcd);
  unsigned blocks = grid_for(total);
  blocks = (blocks + unit - 1) / unit * unit;
  const unsigned px_step = blocks * 256u / (unsigned)CL;
  hipLaunchKernelGGL(prologue_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, px_step);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

extern "C" int ssde_perturb(const ssde_perturb_args* a, void* stream) {
  SSDE_REQUIRE(a && a->x && a->z && a->s && a->dst && a->n > 0 && a->per > 0, "perturb: bad args");
  const size_t numel = (size_t)a->n * a->per;
  hipLaunchKernelGGL(perturb_kernel, dim3(grid_for(numel)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     a->x, a->z, a->a, a->s, a->dst, a->per, numel);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

extern "C" int ssde_dsm_loss(const ssde_dsm_loss_args* a, void* stream) {
  SSDE_REQUIRE(a && a->score && a->z && a->s && a->losses && a->loss && a->n > 0 && a->per > 0, "dsm_loss: bad args");
  SSDE_REQUIRE(!a->likelihood_weighting || a->g2, "dsm_loss: g2 needed for likelihood weighting");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(dsm_loss_kernel, dim3(a->n), dim3(256), 64, st, a->score, a->z, a->s, a->g2, a->dscore, a->losses,
                     a->n, a->per, a->reduce_mean, a->likelihood_weighting, a->grad_scale == 0.f ? 1.0f : a->grad_scale);
  SSDE_LAUNCH_CHECK();
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 64, st, a->losses, a->n, a->loss);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

extern "C" int ssde_sumsq_flat(const ssde_sumsq_flat_args* a, void* stream) {
  SSDE_REQUIRE(a && a->x && a->partial && a->out && a->numel > 0, "sumsq_flat: bad args (partial needs %d floats)", kSumsqBlocks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int blocks = (int)grid_for((size_t)a->numel / 4, 256, kSumsqBlocks);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 64, st, a->x, (size_t)a->numel, a->partial);
  SSDE_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 64, st, a->partial, blocks, a->out);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

extern "C" int ssde_adam_clip_ema(const ssde_adam_args* a, void* stream) {
  SSDE_REQUIRE(a && a->p && a->g && a->m && a->v && a->hyper && a->numel > 0, "adam: bad args");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for((size_t)a->numel, 256, 256 * 32)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     a->p, a->g, a->m, a->v, a->ema, (size_t)a->numel, a->hyper, a->gnorm_sq);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

extern "C" int ssde_memset(const ssde_memset_args* a, void* stream) {
  SSDE_REQUIRE(a && a->dst && a->bytes >= 0, "memset: bad args");
  if (a->bytes == 0) return SSDE_OK;
  SSDE_HIP_CHECK(hipMemsetAsync(a->dst, a->value, (size_t)a->bytes, static_cast<hipStream_t>(stream)));
  return SSDE_OK;
}

extern "C" int ssde_axpy(const ssde_axpy_args* a, void* stream) {
  SSDE_REQUIRE(a && a->x && a->dst && a->numel > 0, "axpy: bad args");
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for((size_t)a->numel)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     a->x, a->gate, a->dst, (size_t)a->numel, a->alpha, a->acc);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

// ---- weight re-packing -----------------------------------------------------------------------------------
namespace {

// logical 3x3 weight of a packed operand: W[co][ci][ky][kx] (zero outside the parameter)
__device__ __forceinline__ float pack_w3(const ssde_pack_desc& d, int co, int ci, int ky, int kx) {
  if (d.flags & 1) {      // input-gradient weights: transposed + rotated by 180 degrees
    if (ci >= d.cout || co >= d.cin) return 0.f;
    return d.src[((size_t)(ci * d.cin + co) * 3 + (2 - ky)) * 3 + (2 - kx)];
  }
  if (co >= d.cout || ci >= d.cin) return 0.f;
  return d.src[((size_t)(co * d.cin + ci) * 3 + ky) * 3 + kx];
}

// grid (x, entries): every workgroup row re-packs one table entry, destination-driven (coalesced stores)
__global__ __launch_bounds__(256) void pack_conv3_kernel(const ssde_pack_desc* __restrict__ table) {
  const ssde_pack_desc d = table[blockIdx.y];
  const int cpad = (d.cout_l + 63) / 64 * 64;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < (size_t)d.n; idx += (size_t)gridDim.x * 256) {
    const int e = (int)(idx & 7);
    const int co = (int)((idx >> 3) % cpad);
    const size_t r = (idx >> 3) / cpad;
    const int tap = (int)(r % 9), c8 = (int)(r / 9);
    d.dst[idx] = pack_w3(d, co, c8 * 8 + e, tap / 3, tap % 3);
  }
}

// Winograd: U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], stored as conv_wino.hip's LDS image.
// One thread = one (cout, channel pair): it reads the two 3x3 filters once (the first version was one thread per OUTPUT
// element: every filter was re-read 16 times with a 9 KB lane stride and the kernel ran at 0.9 ms per training step)
// and writes the pair's 16 positions as float2 -- consecutive lanes = consecutive couts = 512-byte runs.
__global__ __launch_bounds__(256) void pack_wino3_kernel(const ssde_pack_desc* __restrict__ table) {
  const ssde_pack_desc d = table[blockIdx.y];
  const int ntl = (d.cout_l + 63) / 64;
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  const size_t items = (size_t)d.n >> 5;                  // 16 positions x 2 channels per item
  for (size_t it = (size_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (size_t)gridDim.x * 256) {
    const int cs = (int)(it & 63);
    const int q = (int)((it >> 6) & 3);
    const size_t r = it >> 8;
    const int nt = (int)(r % ntl), c8 = (int)(r / ntl);
    const int co = nt * 64 + (cs ^ ((q & 1) << 4));
    float u[2][16];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ci = c8 * 8 + 2 * q + e;
      float w[3][3];
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int l = 0; l < 3; ++l) w[k][l] = pack_w3(d, co, ci, k, l);
#pragma unroll
      for (int pa = 0; pa < 4; ++pa)
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
          float acc = 0.f;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            float t = 0.f;
#pragma unroll
            for (int l = 0; l < 3; ++l) t += w[k][l] * G[pb][l];
            acc += G[pa][k] * t;
          }
          u[e][pa * 4 + pb] = acc;
        }
    }
#pragma unroll
    for (int pos = 0; pos < 16; ++pos)
      *reinterpret_cast<float2*>(d.dst + ((((r * 16 + pos) * 4 + q) * 64 + cs) << 1)) = make_float2(u[0][pos], u[1][pos]);
  }
}

// Winograd F(4x4,3x3): U = G g G^T with the 6x3 G of conv_wino4.hip, stored as its LDS image
// [ceil(cin/4)][ceil(cout/64)][8 waves (q, h)][9 positions q + 4 j][32 couts of half h][4].  One thread = one (cout, cin):
// it reads the 3x3 filter once and writes its 36 positions (consecutive lanes = the 4 channels of consecutive couts: 512 B
// runs per position).  The products are formed in
// fp64 and rounded once, like the host packing (engine.pack_wino4_weight).
template <bool kPerLane>
__global__ __launch_bounds__(256) void pack_wino4_kernel(const ssde_pack_desc* __restrict__ table) {
  const ssde_pack_desc d = table[blockIdx.y];
  const int ntl = (d.cout_l + 63) / 64;
  const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                          {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
  const size_t items = (size_t)d.n / 36;
  for (size_t it = (size_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (size_t)gridDim.x * 256) {
    const int e = (int)(it & 3), cs = (int)((it >> 2) & 63);
    const size_t r = it >> 8;
    const int nt = (int)(r % ntl), c4 = (int)(r / ntl);
    const int co = nt * 64 + cs, ci = c4 * 4 + e;
    double w[3][3], t[6][3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int l = 0; l < 3; ++l) w[k][l] = (double)pack_w3(d, co, ci, k, l);
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int l = 0; l < 3; ++l) t[a][l] = G[a][0] * w[0][l] + G[a][1] * w[1][l] + G[a][2] * w[2][l];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b)
      {
        const int pos = a * 6 + b, wv = (pos & 3) * 2 + (cs >> 5), j = pos >> 2;      // (conv_wino4.hip: wave (q, h) owns positions q + 4 j of cout half h)
        const float u = (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
        if (!kPerLane) {
          d.dst[(((r * 8 + wv) * 9 + j) * 32 + (cs & 31)) * 4 + e] = u;
        } else {
          // SSDE_PACK_WINO4R (conv_wino4r.hip): wave w owns positions 4 w .. 4 w + 3 with both cout halves -- lane (lh = e >> 1,
          // li = cout & 31) holds channels 2 lh, 2 lh + 1 of couts li and 32 + li as four floats per position -- and cout half
          // w & 1 of position 32 + (w >> 1) as a last piece of two floats per lane
          const int lane = (e >> 1) * 32 + (cs & 31), half = cs >> 5;
          const int w = pos < 32 ? (pos >> 2) : 2 * (pos - 32) + half;
          const size_t base = (r * 8 + w) * (size_t)(9 * 32 * 4);
          d.dst[base + (pos < 32 ? ((pos & 3) * 64 + lane) * 4 + half * 2 + (e & 1) : 1024 + lane * 2 + (e & 1))] = u;
        }
      }
  }
}

// matrix parts (source-driven): dst[(col/8)*ld + row][col%8] = src[i][j]; flags&1 swaps the roles of i and j
__global__ __launch_bounds__(256) void pack_matrix_kernel(const ssde_pack_desc* __restrict__ table) {
  const ssde_pack_desc d = table[blockIdx.y];
  const int ld = (d.cout_l + 63) / 64 * 64;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < (size_t)d.n; idx += (size_t)gridDim.x * 256) {
    const int i = (int)(idx / d.cin), j = (int)(idx % d.cin);
    const int row = ((d.flags & 1) ? j : i) + d.r_off, col = ((d.flags & 1) ? i : j) + d.c_off;
    d.dst[((size_t)(col >> 3) * ld + row) * 8 + (col & 7)] = d.src[idx];
  }
}

__global__ __launch_bounds__(256) void pack_vector_kernel(const ssde_pack_desc* __restrict__ table) {
  const ssde_pack_desc d = table[blockIdx.y];
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < (size_t)d.n; idx += (size_t)gridDim.x * 256)
    d.dst[d.r_off + idx] = d.src[idx] + (d.src2 ? d.src2[idx] : 0.f);
}

}  // namespace

extern "C" int ssde_pack_weights(const ssde_pack_args* a, void* stream) {
  SSDE_REQUIRE(a && a->table && a->count > 0 && a->max_n > 0, "pack_weights: bad args");
  size_t bx = ((size_t)a->max_n + 255) / 256;
  if (bx > 256) bx = 256;
  const dim3 grid((unsigned)bx, (unsigned)a->count);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (a->kind) {
    case SSDE_PACK_CONV3: hipLaunchKernelGGL(pack_conv3_kernel, grid, dim3(256), 0, st, a->table); break;
    case SSDE_PACK_WINO3: hipLaunchKernelGGL(pack_wino3_kernel, grid, dim3(256), 0, st, a->table); break;
    case SSDE_PACK_WINO4: hipLaunchKernelGGL(pack_wino4_kernel<false>, grid, dim3(256), 0, st, a->table); break;
    case SSDE_PACK_WINO4R: hipLaunchKernelGGL(pack_wino4_kernel<true>, grid, dim3(256), 0, st, a->table); break;
    case SSDE_PACK_MATRIX: hipLaunchKernelGGL(pack_matrix_kernel, grid, dim3(256), 0, st, a->table); break;
    case SSDE_PACK_VECTOR: hipLaunchKernelGGL(pack_vector_kernel, grid, dim3(256), 0, st, a->table); break;
    default: ssde_set_error("pack_weights: unknown kind %d", a->kind); return SSDE_EINVAL;
  }
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
