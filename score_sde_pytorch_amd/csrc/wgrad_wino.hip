// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions by Winograd F(2x2, 3x3) on the exact-fp32 matrix pipe.
//
// Forward (conv_wino.hip):  Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A   per 2x2 output tile.  Differentiating,
//
//   dL/dg[co, ci] = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G
//
// so per transform position p (16 of them) the gradient is a GEMM  dU_p[co, ci] = sum_tiles Z_p[tile, co] * V_p[tile, ci]
// whose reduction runs over the 2x2 tiles of the whole batch -- 16 multiply-adds per (tile, co, ci) instead of the 36 of
// the direct form (wgrad.hip), i.e. 2.25x less matrix-pipe work for the layers that dominate the training step.
// V is the forward kernel's input transform of pro(src) (GroupNorm / SiLU / dropout recomputed while staging, as
// everywhere), Z = A dY A^T costs 12 adds per (tile, co); G^T . G is applied once, by the reduction kernel.
//
// One workgroup (8 waves, 128 accumulator registers per lane) owns a 64 co x 64 ci block for all 16 positions and walks
// a contiguous range of CHUNKS (one chunk = 8 tiles = a 4 x 8 output patch of one image = one MFMA stage, k = tile
// pairs); the chunk range is split over workgroups and the partial dU blocks are summed in a fixed order
// (deterministic).  The stage pipeline is the forward kernel's ping-pong: while waves 0-3 run their 64 MFMAs, waves 4-7
// apply the prologue and park the raw halo and the raw output-gradient patch in LDS; then waves 4-7 run theirs (with the
// global loads of the stage after next issued between the MFMAs) while waves 0-3 transform raw -> V, Z.  LDS layouts
// [pos][tile pair][64 channels, bit 4 ^= pair parity][2 tiles] make every fragment read a conflict-free ds_read_b64
// that feeds two MFMAs, exactly as in conv_wino.hip (with "tile" <-> channel and "channel pair" <-> tile pair).
#include "ssde_common.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kThreads = 512;
constexpr int kStageFloats = 16 * 4 * 64 * 2;     // one V or Z stage: [pos][tile pair][64][2]
constexpr int kHaloW = 10, kHaloH = 6;            // halo of a 4 x 8 output patch
constexpr int kHaloPx = kHaloW * kHaloH;
constexpr int kRawX = kHaloPx * 64, kRawG = 32 * 64;
constexpr int kSlab = 16 * 64 * 64;               // floats of one partial dU block

struct WwParams {
  ssde_src src;
  const float* g;
  int g_ld, g_off;
  int N, H, W, Cout, Ctot;
  int cx, cy;              // chunks per image row / column (W / 8, H / 4)
  int chunks, chunks_per_split, splits;
  int co_tiles, ci_tiles;
  float scale;
  float* dw;
  float* scratch;
};

template <bool kGn>
__global__ __launch_bounds__(kThreads, 2) void wgrad_wino_kernel(const WwParams p) {
  SSDE_LDS(smem);
  float* Vb = smem;                          // [2][kStageFloats]   B operand (input channels)
  float* Zb = smem + 2 * kStageFloats;       // [2][kStageFloats]   A operand (output channels)
  float* rawx = smem + 4 * kStageFloats;     // [60 halo pixels][64 ci]
  float* rawg = rawx + kRawX;                // [32 pixels][64 co]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lq = lane >> 4;
  const int ph = wave >> 2;

  // XCD-aware order: blocks are dealt round-robin to the 8 XCDs, and a workgroup needs a whole CU (155 KB of LDS), so
  // every XCD must get the SAME number (<= 32) of workgroups: XCD x takes the x-th run of ceil(total / 8) consecutive
  // (split, tile) pairs -- the (co, ci) blocks of one chunk range sit on one XCD (at most two) and share its L2.
  // (Dealing whole splits to XCDs left 5 XCDs with 36 workgroups for 32 CUs at 12 blocks per split: two rounds.)
  const int ntiles = p.co_tiles * p.ci_tiles;
  const int total = p.splits * ntiles, per_xcd = (total + 7) >> 3;
  const int j = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (j >= total) return;
  const int tile = j % ntiles, split = j / ntiles;
  const int co0 = (tile / p.ci_tiles) * 64, ci0 = (tile % p.ci_tiles) * 64;
  const int ch_begin = split * p.chunks_per_split;
  const int nst = min(p.chunks, ch_begin + p.chunks_per_split) - ch_begin;
  const int last = nst - 1;

  const ssde_src& s = p.src;
  SsdePro pro = ssde_pro_decode(s);
  pro.gn = kGn;
  const int cpg = kGn ? p.Ctot / s.gn_groups : 1;
  // the 64 input channels of this block lie in one tensor of the virtual concat
  const bool second = ci0 >= s.c0;
  const float* xbase = second ? s.p1 : s.p0;
  const int xC = second ? s.c1 : s.c0;
  const int xc0 = second ? ci0 - s.c0 : ci0;

  // ---- staging plan of waves 4-7: item = (pixel, channel quad); the quad is the same for all items of a thread ----
  const int sid = tid & 255;
  const int quad = sid & 15;
  int hy[4], hx[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int px = min((sid + it * 256) >> 4, kHaloPx - 1);
    hy[it] = px / kHaloW; hx[it] = px - hy[it] * kHaloW;
  }
  const bool x_item3 = (sid + 3 * 256) < kHaloPx * 16;      // 960 items: the fourth exists for sid < 192
  const float* xq = xbase + xc0 + quad * 4;
  const float* gq = p.g + p.g_off + co0 + quad * 4;
  float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kGn) {
    gam = *reinterpret_cast<const float4*>(s.gn_gamma + ci0 + quad * 4);
    bet = *reinterpret_cast<const float4*>(s.gn_beta + ci0 + quad * 4);
  }
  const int ggrp = (ci0 + quad * 4) / cpg;

  // chunk walked by the loader (stages are loaded in order; past the last stage the same chunk is loaded again)
  int l_idx = -1, l_img = 0, l_cy = 0, l_cx = 0;
  {
    const int per_img = p.cx * p.cy;
    l_img = ch_begin / per_img;
    const int r = ch_begin - l_img * per_img;
    l_cy = r / p.cx; l_cx = r - l_cy * p.cx;
  }
  float4 xv[4], gv[2];
  float mu = 0.f, rs = 1.f;
  int xpix[4];              // linear input pixel of the item, -1 = zero padding
  // branch-free pieces of one stage's global loads (issued between the MFMAs of the matrix phase)
  auto load_piece = [&](int st, int k) {
    if (k == 0) {
      // advance to the next chunk (image-major, then rows, then columns) when a NEW stage is loaded: selects, no branch
      const int adv = (st > l_idx && l_idx >= 0) ? 1 : 0;
      l_cx += adv;
      const int wx = (l_cx == p.cx) ? 1 : 0;
      l_cx = wx ? 0 : l_cx;
      l_cy += wx;
      const int wy = (l_cy == p.cy) ? 1 : 0;
      l_cy = wy ? 0 : l_cy;
      l_img += wy;
      l_idx = st;
      if (kGn) {
        mu = s.gn_mean[l_img * s.gn_groups + ggrp];
        rs = s.gn_rstd[l_img * s.gn_groups + ggrp];
      }
    }
    if (k >= 1 && k <= 4) {
      const int it = k - 1;
      const int iy = l_cy * 4 - 1 + hy[it], ix = l_cx * 8 - 1 + hx[it];
      const bool inb = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int pix = (l_img * p.H + (inb ? iy : 0)) * p.W + (inb ? ix : 0);
      xpix[it] = inb ? pix : -1;
      xv[it] = *reinterpret_cast<const float4*>(xq + (size_t)pix * xC);
    } else if (k >= 5 && k <= 6) {
      const int it = k - 5;
      const int px = (sid + it * 256) >> 4;                    // 0..31: row px >> 3, column px & 7 of the patch
      const int pix = (l_img * p.H + l_cy * 4 + (px >> 3)) * p.W + l_cx * 8 + (px & 7);
      gv[it] = *reinterpret_cast<const float4*>(gq + (size_t)pix * p.g_ld);
    }
  };
  auto load_stage = [&](int st) {
#pragma unroll
    for (int k = 0; k < 8; ++k) load_piece(st, k);
  };
  // prologue + raw LDS stores of the loaded stage
  auto store_stage = [&]() {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (it == 3 && !x_item3) continue;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (xpix[it] >= 0)
        v = ssde_pro_apply(xv[it], mu, rs, gam, bet, (uint32_t)xpix[it] * (uint32_t)p.Ctot + (uint32_t)(ci0 + quad * 4), pro);
      *reinterpret_cast<float4*>(rawx + (sid + it * 256) * 4) = v;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) *reinterpret_cast<float4*>(rawg + (sid + it * 256) * 4) = gv[it];
  };

  // ---- transforms of waves 0-3: thread = (tile pair tp, channel c); tiles 2 tp, 2 tp + 1 are horizontal neighbours ----
  const int tp = sid >> 6, tc = sid & 63;
  const int trow = tp >> 1, tcol = (tp & 1) * 2;               // tile grid 2 x 4; tile 2 tp at (trow, tcol)
  const int t_col = (tp * 64 + (tc ^ ((tp & 1) << 4))) * 2;
  auto transform = [&](float* Vn, float* Zn) {
    // V = B^T d B of both tiles at once: component .x = tile 2 tp, .y = tile 2 tp + 1 (its patch starts 2 columns right)
    {
      const float* rp = rawx + ((2 * trow) * kHaloW + 2 * tcol) * 64 + tc;
      float col[4][6];
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 6; ++x) col[y][x] = rp[(y * kHaloW + x) * 64];
      float2 r[4][4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        float2 d[4];
#pragma unroll
        for (int y = 0; y < 4; ++y) d[y] = make_float2(col[y][x], col[y][x + 2]);
        r[0][x] = make_float2(d[0].x - d[2].x, d[0].y - d[2].y);
        r[1][x] = make_float2(d[1].x + d[2].x, d[1].y + d[2].y);
        r[2][x] = make_float2(d[2].x - d[1].x, d[2].y - d[1].y);
        r[3][x] = make_float2(d[1].x - d[3].x, d[1].y - d[3].y);
      }
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        *reinterpret_cast<float2*>(Vn + (y * 4 + 0) * 512 + t_col) = make_float2(r[y][0].x - r[y][2].x, r[y][0].y - r[y][2].y);
        *reinterpret_cast<float2*>(Vn + (y * 4 + 1) * 512 + t_col) = make_float2(r[y][1].x + r[y][2].x, r[y][1].y + r[y][2].y);
        *reinterpret_cast<float2*>(Vn + (y * 4 + 2) * 512 + t_col) = make_float2(r[y][2].x - r[y][1].x, r[y][2].y - r[y][1].y);
        *reinterpret_cast<float2*>(Vn + (y * 4 + 3) * 512 + t_col) = make_float2(r[y][1].x - r[y][3].x, r[y][1].y - r[y][3].y);
      }
    }
    // Z = A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]
    {
      const float* gp = rawg + ((2 * trow) * 8 + 2 * tcol) * 64 + tc;
      float2 y[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) y[i][j] = make_float2(gp[(i * 8 + j) * 64], gp[(i * 8 + j + 2) * 64]);
      float2 t[4][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        t[0][j] = y[0][j];
        t[1][j] = make_float2(y[0][j].x + y[1][j].x, y[0][j].y + y[1][j].y);
        t[2][j] = make_float2(y[0][j].x - y[1][j].x, y[0][j].y - y[1][j].y);
        t[3][j] = make_float2(-y[1][j].x, -y[1][j].y);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        *reinterpret_cast<float2*>(Zn + (a * 4 + 0) * 512 + t_col) = t[a][0];
        *reinterpret_cast<float2*>(Zn + (a * 4 + 1) * 512 + t_col) = make_float2(t[a][0].x + t[a][1].x, t[a][0].y + t[a][1].y);
        *reinterpret_cast<float2*>(Zn + (a * 4 + 2) * 512 + t_col) = make_float2(t[a][0].x - t[a][1].x, t[a][0].y - t[a][1].y);
        *reinterpret_cast<float2*>(Zn + (a * 4 + 3) * 512 + t_col) = make_float2(-t[a][1].x, -t[a][1].y);
      }
    }
  };

  f32x4 acc[8][2][2];
#pragma unroll
  for (int ps = 0; ps < 8; ++ps)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ps][a][b][r] = 0.f;

  // wave block: 32 co x 32 ci for 8 positions (waves 0-3: positions 0-7, waves 4-7: 8-15)
  const int cob = ((wave >> 1) & 1) * 32, cib = (wave & 1) * 32;
  const int swz = (lq & 1) << 4;
  int aoff[2], boff[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) aoff[a] = ph * 8 * 512 + (lq * 64 + ((cob + a * 16 + li) ^ swz)) * 2;
#pragma unroll
  for (int b = 0; b < 2; ++b) boff[b] = ph * 8 * 512 + (lq * 64 + ((cib + b * 16 + li) ^ swz)) * 2;

  // `piece(ps)`: a slice of this wave's load issue work, placed between the two MFMA quartets of position ps
  auto mfma_stage = [&](const float* Zc, const float* Vc, auto&& piece) {
    float2 af[2][2], bf[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) af[0][a] = *reinterpret_cast<const float2*>(Zc + aoff[a]);
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[0][b] = *reinterpret_cast<const float2*>(Vc + boff[b]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int cur = ps & 1;
      if (ps + 1 < 8) {
#pragma unroll
        for (int a = 0; a < 2; ++a) af[cur ^ 1][a] = *reinterpret_cast<const float2*>(Zc + (ps + 1) * 512 + aoff[a]);
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[cur ^ 1][b] = *reinterpret_cast<const float2*>(Vc + (ps + 1) * 512 + boff[b]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[ps][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][a].x, bf[cur][b].x, acc[ps][a][b], 0, 0, 0);
      piece(ps);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[ps][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][a].y, bf[cur][b].y, acc[ps][a][b], 0, 0, 0);
      if (ps + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x026, 24, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- pipeline prologue: stage 0 transformed, stage 1 in flight ----
  if (ph == 1) { load_stage(0); store_stage(); }
  SSDE_LDS_BARRIER();
  if (ph == 0) transform(Vb, Zb);
  else load_stage(min(1, last));
  SSDE_LDS_BARRIER();

  for (int st = 0; st < nst; ++st) {
    const float* Vc = Vb + (st & 1) * kStageFloats;
    const float* Zc = Zb + (st & 1) * kStageFloats;
    float* Vn = Vb + ((st + 1) & 1) * kStageFloats;
    float* Zn = Zb + ((st + 1) & 1) * kStageFloats;
    // phase 1: waves 0-3 on the matrix pipe; waves 4-7 apply the prologue to stage st+1 and park it in the raw buffers
    if (ph == 0) mfma_stage(Zc, Vc, [&](int) {});
    else if (st + 1 < nst) store_stage();
    SSDE_LDS_BARRIER();
    // phase 2: waves 4-7 on the matrix pipe, issuing the loads of stage st+2 between their MFMAs; waves 0-3 transform st+1
    if (ph == 1) {
      const int sn = min(st + 2, last);
      mfma_stage(Zc, Vc, [&](int k) { load_piece(sn, k); });
    } else if (st + 1 < nst) {
      transform(Vn, Zn);
    }
    SSDE_LDS_BARRIER();
  }

  // ---- partial block dU[pos][co][ci] -> slab [split][tile][16][64][64]: a lane's 16 consecutive ci are a 64-byte run ----
  float* slab = p.scratch + ((size_t)split * ntiles + tile) * (size_t)kSlab;
#pragma unroll
  for (int ps = 0; ps < 8; ++ps)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          slab[((size_t)(ph * 8 + ps) * 64 + cob + a * 16 + 4 * lq + r) * 64 + cib + b * 16 + li] = acc[ps][a][b][r];
}

// dw[co, ci, :, :] += scale * G^T (sum_splits dU[co, ci]) G,  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];
// thread = one (tile, co, ci); consecutive threads -> consecutive ci (coalesced slab reads)
__global__ __launch_bounds__(256) void wgrad_wino_reduce_kernel(const WwParams p) {
  const int ntiles = p.co_tiles * p.ci_tiles;
  const int total = ntiles * 4096;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int tile = idx >> 12, co_l = (idx >> 6) & 63, ci_l = idx & 63;
    float u[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) u[q] = 0.f;
    for (int sp = 0; sp < p.splits; ++sp) {
      const float* slab = p.scratch + ((size_t)sp * ntiles + tile) * (size_t)kSlab + co_l * 64 + ci_l;
#pragma unroll
      for (int q = 0; q < 16; ++q) u[q] += slab[q * 4096];
    }
    float t[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      t[0][b] = u[0 * 4 + b] + 0.5f * (u[1 * 4 + b] + u[2 * 4 + b]);
      t[1][b] = 0.5f * (u[1 * 4 + b] - u[2 * 4 + b]);
      t[2][b] = u[3 * 4 + b] + 0.5f * (u[1 * 4 + b] + u[2 * 4 + b]);
    }
    const int co = (tile / p.ci_tiles) * 64 + co_l, ci = (tile % p.ci_tiles) * 64 + ci_l;
    float* dst = p.dw + ((size_t)co * p.Ctot + ci) * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      dst[i * 3 + 0] += p.scale * (t[i][0] + 0.5f * (t[i][1] + t[i][2]));
      dst[i * 3 + 1] += p.scale * (0.5f * (t[i][1] - t[i][2]));
      dst[i * 3 + 2] += p.scale * (t[i][3] + 0.5f * (t[i][1] + t[i][2]));
    }
  }
}

bool winograd_enabled(const ssde_wgrad_args* a) { return !(a->flags & SSDE_WGRADF_DIRECT); }

}  // namespace

// ssde_conv_wgrad / ssde_wgrad_scratch_floats (wgrad.hip) route eligible launches here.
bool ssde_wgrad_wino_wants(const ssde_wgrad_args* a) {
  if (!winograd_enabled(a) || a->ksize != 3 || a->stride != 1 || a->pad != 1 || a->transpose_out) return false;
  if (a->h_in != a->h_out || a->w_in != a->w_out || a->h_out % 4 != 0 || a->w_out % 8 != 0) return false;
  const ssde_src& s = a->src;
  const int Ctot = s.c0 + s.c1;
  if (a->c_out % 64 != 0 || Ctot % 64 != 0 || a->cin_store != Ctot || (s.c1 > 0 && s.c0 % 64 != 0)) return false;
  if (a->g_ld % 4 != 0 || a->g_off % 4 != 0) return false;
  return (long long)a->n * a->h_out * a->w_out < (1ll << 30);
}

static int ww_plan(const ssde_wgrad_args* a, WwParams* p) {
  const ssde_src& s = a->src;
  p->src = s; p->g = a->g; p->g_ld = a->g_ld; p->g_off = a->g_off;
  p->N = a->n; p->H = a->h_out; p->W = a->w_out; p->Cout = a->c_out; p->Ctot = s.c0 + s.c1;
  p->cx = a->w_out / 8; p->cy = a->h_out / 4;
  p->chunks = a->n * p->cx * p->cy;
  p->co_tiles = a->c_out / 64; p->ci_tiles = p->Ctot / 64;
  const int ntiles = p->co_tiles * p->ci_tiles;
  // one workgroup fills a CU (155 KB of LDS): aim at ONE round of at most 256 workgroups (257 would take as long as
  // 512), with at least 4 stages each
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

int64_t ssde_wgrad_wino_scratch_floats(const ssde_wgrad_args* a) {
  WwParams p;
  ssde_wgrad_args b = *a;
  b.splits = 0;
  ww_plan(&b, &p);
  return (int64_t)p.splits * p.co_tiles * p.ci_tiles * kSlab;
}

int ssde_wgrad_wino_launch(const ssde_wgrad_args* a, void* stream) {
  const ssde_src& s = a->src;
  SSDE_REQUIRE(a->g && a->dw && s.p0 && (s.c1 == 0 || s.p1), "wgrad(winograd): null tensors");
  const bool gn = s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU;
  if (gn) {
    SSDE_REQUIRE(s.gn_groups > 0 && (s.c0 + s.c1) % s.gn_groups == 0 && ((s.c0 + s.c1) / s.gn_groups) % 4 == 0,
                 "wgrad(winograd): GroupNorm channels-per-group %% 4");
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "wgrad(winograd): GroupNorm pointers missing");
  }
  SSDE_REQUIRE(s.drop_thresh == 0 || s.drop_seed, "wgrad(winograd): dropout seed pointer missing");
  WwParams p;
  ww_plan(a, &p);
  const int64_t need = (int64_t)p.splits * p.co_tiles * p.ci_tiles * kSlab;
  if (a->scratch_floats < need) {          // fewer, longer workgroups when scratch is short
    const int fit = (int)(a->scratch_floats / ((int64_t)p.co_tiles * p.ci_tiles * kSlab));
    SSDE_REQUIRE(fit >= 1 && a->scratch, "wgrad(winograd): scratch of %lld floats needed at least", (long long)p.co_tiles * p.ci_tiles * kSlab);
    p.chunks_per_split = ssde_cdiv(p.chunks, fit);
    p.splits = ssde_cdiv(p.chunks, p.chunks_per_split);
  }
  SSDE_REQUIRE(a->scratch, "wgrad(winograd): scratch missing");
  constexpr int lds = (4 * kStageFloats + kRawX + kRawG) * 4;
  static std::atomic<bool> attr_set{false};   // once, before any stream capture
  if (!attr_set) {
    SSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_wino_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_wino_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int ntiles = p.co_tiles * p.ci_tiles;
  const dim3 grid(ssde_cdiv(p.splits * ntiles, 8) * 8);
  if (gn) hipLaunchKernelGGL(wgrad_wino_kernel<true>, grid, dim3(kThreads), lds, st, p);
  else hipLaunchKernelGGL(wgrad_wino_kernel<false>, grid, dim3(kThreads), lds, st, p);
  SSDE_LAUNCH_CHECK();
  int blocks = ntiles * 16;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
