// GroupNorm statistics (mean, rstd per (sample, group)) for NHWC fp32 tensors.
// Reference: nn.GroupNorm(num_groups=min(C//4, 32), eps=1e-6) as used at
// models/layerspp.py:67,219,231 and models/ncsnpp.py:194-227.  Only the reduction
// lives here; normalise + affine (+ SiLU) is fused into the consumer's LDS staging
// (conv_mfma.hip, resample.hip), so the normalised tensor never touches HBM.
//
// HBM-bound: every element is read exactly once with 16-byte lane loads; a wave
// covers 256 consecutive channels of a pixel run.  Sums are accumulated relative
// to a per-group pivot (the group's first element) so E[(x-K)^2] - E[x-K]^2 does
// not cancel catastrophically; lanes are combined with wave64 shuffles.
#include "ssde_common.h"

namespace {

constexpr int kGnThreads = 1024;

struct GnParams {
  const float* p0; const float* p1; int c0, c1;
  int n, hw, groups, slices; float eps;
  float* mean; float* rstd; float* scratch;
};

// grid = (slices, n).  Thread (pl, cl): channel float4 `cl`, pixels pl, pl+PL, ...
__global__ __launch_bounds__(kGnThreads) void gn_stats_kernel(const GnParams p) {
  SSDE_LDS(smem);
  float* s_sum = smem;
  float* s_sq = smem + kGnThreads;
  const int n = blockIdx.y, slice = blockIdx.x;
  const int C = p.c0 + p.c1;
  const int CL = C >> 2;                 // float4 lanes per pixel
  const int PL = kGnThreads / CL;        // pixel lanes
  const int tid = threadIdx.x;
  const int cpg = C / p.groups;
  const int px_per_slice = (p.hw + p.slices - 1) / p.slices;
  const int px0 = slice * px_per_slice;
  const int px1 = min(p.hw, px0 + px_per_slice);

  float sum = 0.f, sq = 0.f;
  if (tid < CL * PL) {
    const int cl = tid % CL, pl = tid / CL;
    const int ch = cl * 4;
    const float* base; int Cs, cc;
    if (ch < p.c0) { base = p.p0; Cs = p.c0; cc = ch; } else { base = p.p1; Cs = p.c1; cc = ch - p.c0; }
    // pivot: first element (pixel 0) of this lane's group
    const int g = ch / cpg;
    const int gch = g * cpg;
    const float pivot = (gch < p.c0) ? p.p0[(size_t)n * p.hw * p.c0 + gch]
                                     : p.p1[(size_t)n * p.hw * p.c1 + (gch - p.c0)];
    const float* src = base + (size_t)n * p.hw * Cs + cc;
    int px = px0 + pl;
    // 4 independent loads in flight per lane
    for (; px + 3 * PL < px1; px += 4 * PL) {
      const float4 a = *reinterpret_cast<const float4*>(src + (size_t)px * Cs);
      const float4 b = *reinterpret_cast<const float4*>(src + (size_t)(px + PL) * Cs);
      const float4 c = *reinterpret_cast<const float4*>(src + (size_t)(px + 2 * PL) * Cs);
      const float4 d = *reinterpret_cast<const float4*>(src + (size_t)(px + 3 * PL) * Cs);
      const float4 v[4] = {a, b, c, d};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float x0 = v[k].x - pivot, x1 = v[k].y - pivot, x2 = v[k].z - pivot, x3 = v[k].w - pivot;
        sum += (x0 + x1) + (x2 + x3);
        sq += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
      }
    }
    for (; px < px1; px += PL) {
      const float4 a = *reinterpret_cast<const float4*>(src + (size_t)px * Cs);
      const float x0 = a.x - pivot, x1 = a.y - pivot, x2 = a.z - pivot, x3 = a.w - pivot;
      sum += (x0 + x1) + (x2 + x3);
      sq += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
    }
  }
  s_sum[tid] = sum;
  s_sq[tid] = sq;
  __syncthreads();
  // one thread per group gathers its (cpg/4) x PL lanes in a fixed order (deterministic)
  if (tid < p.groups) {
    const int g = tid;
    const int l0 = g * (cpg >> 2), l1 = l0 + (cpg >> 2);
    float S = 0.f, Q = 0.f;
    for (int pl = 0; pl < PL; ++pl)
      for (int cl = l0; cl < l1; ++cl) { S += s_sum[pl * CL + cl]; Q += s_sq[pl * CL + cl]; }
    if (p.slices == 1) {
      const int gch = g * cpg;
      const float pivot = (gch < p.c0) ? p.p0[(size_t)n * p.hw * p.c0 + gch]
                                       : p.p1[(size_t)n * p.hw * p.c1 + (gch - p.c0)];
      const float cnt = (float)cpg * (float)p.hw;
      const float m = S / cnt;
      float var = Q / cnt - m * m;
      var = var < 0.f ? 0.f : var;
      p.mean[n * p.groups + g] = pivot + m;
      p.rstd[n * p.groups + g] = 1.0f / sqrtf(var + p.eps);
    } else {
      float* o = p.scratch + (((size_t)n * p.slices + slice) * p.groups + g) * 2;
      o[0] = S; o[1] = Q;
    }
  }
}

// slices > 1: combine the per-slice partial sums (same pivot in every slice).
__global__ void gn_finalize_kernel(const GnParams p) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.n * p.groups) return;
  const int n = idx / p.groups, g = idx % p.groups;
  const int C = p.c0 + p.c1, cpg = C / p.groups;
  float S = 0.f, Q = 0.f;
  for (int s = 0; s < p.slices; ++s) {
    const float* o = p.scratch + (((size_t)n * p.slices + s) * p.groups + g) * 2;
    S += o[0]; Q += o[1];
  }
  const int gch = g * cpg;
  const float pivot = (gch < p.c0) ? p.p0[(size_t)n * p.hw * p.c0 + gch]
                                   : p.p1[(size_t)n * p.hw * p.c1 + (gch - p.c0)];
  const float cnt = (float)cpg * (float)p.hw;
  const float m = S / cnt;
  float var = Q / cnt - m * m;
  var = var < 0.f ? 0.f : var;
  p.mean[idx] = pivot + m;
  p.rstd[idx] = 1.0f / sqrtf(var + p.eps);
}

// Statistics from the producers' partials (ssde_store_tile): a team of 16 lanes per (image, group), ssde_gn_merge16
// (ssde_common.h) -- 16 (image, group) pairs per workgroup.  (Until round 6: one wave per pair, most of its lanes idle behind 16
// to 64 entries and two more shuffle rounds in the chain.)
struct GnFinParams {
  const float* part0; const float* part1;
  int c0, c1, s0, s1, n, groups; float eps;
  float* mean; float* rstd;
};
__global__ __launch_bounds__(256) void gn_part_finalize_kernel(const GnFinParams p) {
  const int idx = blockIdx.x * 16 + (threadIdx.x >> 4), l16 = threadIdx.x & 15;
  const int tot = p.n * p.groups;
  const int pair = idx < tot ? idx : tot - 1;          // (whole waves run the shuffles: a pair beyond the end repeats the last one)
  const int n = pair / p.groups, g = pair - n * p.groups;
  float cnt, m, M2;
  ssde_gn_merge16(p.part0, p.part1, p.c0, p.c1, p.s0, p.s1, p.groups, n, g, l16, cnt, m, M2);
  if (l16 == 0 && idx < tot) {
    p.mean[idx] = m;
    p.rstd[idx] = ssde_gn_rstd(cnt, M2, p.eps);
  }
}

// The same merge for groups with MANY entries (SSDE_GN_TEAM_MAX_ENTRIES: the 128x128 / 256x256 levels of FFHQ-256, 1024-4096
// slices per image): one workgroup per (image, group).  Thread t merges the entries t, t + 256, ... in order -- four loads in
// flight at a time --, then the 256 thread results are merged pairwise through LDS, lower index first: a fixed order, so the
// result is deterministic (it is NOT the order of a 16-lane team: such a group is always merged here, by nobody else).
// With teams of 16 these launches were 64-256 dependent loads per lane: 109 launches of an FFHQ-256 evaluation took 3.2 ms
// (round 5's wave per group: 1.4 ms).
__global__ __launch_bounds__(256) void gn_part_finalize_big_kernel(const GnFinParams p) {
  SSDE_LDS(red);                                   // [256][3]
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int n = pair / p.groups, g = pair - n * p.groups;
  SsdeGnTeam t;
  ssde_gn_team_init(t, p.part0, p.part1, p.c0, p.c1, p.s0, p.s1, p.groups, n, g);
  float cnt = 0.f, m = 0.f, M2 = 0.f;
  for (int k0 = tid; k0 < t.etot; k0 += 4 * 256) {
    float v[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + 256 * i;
      const float* e = t.entry(k < t.etot ? k : 0);
      v[i][0] = e[0]; v[i][1] = e[1]; v[i][2] = e[2];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (k0 + 256 * i < t.etot) ssde_stat_merge(cnt, m, M2, v[i][2], v[i][0], v[i][1]);
  }
  red[tid * 3] = cnt; red[tid * 3 + 1] = m; red[tid * 3 + 2] = M2;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    if ((tid & (2 * o - 1)) == 0) {
      ssde_stat_merge(cnt, m, M2, red[(tid + o) * 3], red[(tid + o) * 3 + 1], red[(tid + o) * 3 + 2]);
      red[tid * 3] = cnt; red[tid * 3 + 1] = m; red[tid * 3 + 2] = M2;
    }
    __syncthreads();
  }
  if (tid == 0) {
    p.mean[pair] = m;
    p.rstd[pair] = ssde_gn_rstd(cnt, M2, p.eps);
  }
}

}  // namespace

extern "C" int ssde_gn_finalize(const ssde_gn_finalize_args* a, void* stream) {
  SSDE_REQUIRE(a && a->part0 && a->mean && a->rstd, "gn_finalize: null args");
  const int C = a->c0 + a->c1;
  SSDE_REQUIRE(a->c0 > 0 && a->c0 % 4 == 0 && a->c1 % 4 == 0 && (a->c1 == 0 || a->part1), "gn_finalize: bad channel counts");
  SSDE_REQUIRE(a->groups > 0 && C % a->groups == 0 && (C / a->groups) % 4 == 0, "gn_finalize: channels-per-group must be a multiple of 4");
  SSDE_REQUIRE(a->n > 0 && a->slices0 > 0 && (a->c1 == 0 || a->slices1 > 0), "gn_finalize: bad shape");
  GnFinParams p{a->part0, a->part1, a->c0, a->c1, a->slices0, a->slices1, a->n, a->groups, a->eps, a->mean, a->rstd};
  if (ssde_gn_group_is_big(a->c0, a->c1, a->slices0, a->slices1, a->groups))
    hipLaunchKernelGGL(gn_part_finalize_big_kernel, dim3(a->n * a->groups), dim3(256), 256 * 3 * sizeof(float), static_cast<hipStream_t>(stream), p);
  else
    hipLaunchKernelGGL(gn_part_finalize_kernel, dim3(ssde_cdiv(a->n * a->groups, 16)), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

extern "C" int ssde_groupnorm_stats(const ssde_gn_stats_args* a, void* stream) {
  SSDE_REQUIRE(a && a->p0 && a->mean && a->rstd, "gn_stats: null args");
  const int C = a->c0 + a->c1;
  SSDE_REQUIRE(a->c0 % 4 == 0 && a->c1 % 4 == 0 && C > 0 && C <= 4 * kGnThreads, "gn_stats: bad channel count %d", C);
  SSDE_REQUIRE(a->c1 == 0 || a->p1, "gn_stats: second tensor missing");
  SSDE_REQUIRE(a->groups > 0 && a->groups <= kGnThreads && C % a->groups == 0 && (C / a->groups) % 4 == 0,
               "gn_stats: channels-per-group must be a multiple of 4 (C=%d G=%d)", C, a->groups);
  SSDE_REQUIRE(a->n > 0 && a->hw > 0, "gn_stats: bad shape");
  int slices = a->slices > 0 ? a->slices : 1;
  SSDE_REQUIRE(slices == 1 || a->scratch, "gn_stats: scratch needed for slices > 1");
  GnParams p{a->p0, a->p1, a->c0, a->c1, a->n, a->hw, a->groups, slices, a->eps, a->mean, a->rstd, a->scratch};
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(slices, a->n), dim3(kGnThreads), 2 * kGnThreads * sizeof(float), st, p);
  SSDE_LAUNCH_CHECK();
  if (slices > 1) {
    const int tot = a->n * a->groups;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(ssde_cdiv(tot, 256)), dim3(256), 0, st, p);
    SSDE_LAUNCH_CHECK();
  }
  return SSDE_OK;
}
