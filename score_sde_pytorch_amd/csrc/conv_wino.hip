// 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2, 3x3) on the exact-fp32 matrix pipe.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A          (Lavin & Gray; the fp32 algorithm cuDNN / MIOpen
//                                                          also pick for these layers in the reference)
// Per 2x2 output tile and (ci, co) pair the 36 multiply-adds of the direct form become 16: the 3x3
// layers are 90% of the network's FLOPs, so the matrix pipe -- the bound of the direct
// kernel (conv_mfma.hip) -- does 2.25x less work.  Everything stays fp32; the transforms only add
// and halve, so the result differs from the direct sum by fp32 rounding (tests: same tolerance).
//
// One workgroup (8 waves = two per SIMD, 128 accumulator registers per lane) owns 64 tiles (a 16x16
// output patch, or 8x8 patches of 4 images; 4x4 images go to the direct kernel) x 64 output channels x all 16 transform
// positions: waves 0-3 hold positions 0-7 (transform rows 0,1), waves 4-7 positions 8-15, each wave a
// 32-tile x 32-cout block.  The two waves of a SIMD (w and w+4) PING-PONG: while one runs its 64 MFMAs
// of the stage, the other does staging work for the next stage, then they swap (two barriers per
// stage); a version in which every wave staged and computed in the same phase kept the matrix pipe
// only 22% busy, because VALU/LDS/VMEM work of both waves coincided and nothing covered it.
//   phase 1: waves 0-3 start the LDS-DMA of the weights (k+1), then MFMA(stage k)
//            waves 4-7 prologue + LDS stores of the raw halo (k+1)
//   phase 2: waves 4-7 issue the halo loads of stage k+2 into registers, then MFMA(stage k)
//            waves 0-3 input transform raw -> V(k+1), then wait for their DMA
// Every asynchronous load is issued a full phase (>= 2048 matrix cycles) before it is needed and the barriers
// order LDS only (SSDE_LDS_BARRIER): with __syncthreads() the compiler drained vmcnt(0) -- an HBM latency --
// at both barriers of every stage and the matrix pipe was 47% busy.
// Per 8-input-channel stage:
//   raw halo (with the fused GroupNorm / SiLU / dropout prologue)  -> LDS, channel-pair major
//   input transform B^T d B, one (tile, channel pair) per thread   -> LDS V[pos][pair][tile][2]
//   host-transformed weights G g G^T, pre-arranged as the LDS image -> LDS U[pos][pair][cout][2]
//   16 positions x (32 tiles x 32 couts per wave) on v_mfma_f32_16x16x4_f32: a lane's ds_read_b64 is the
//   channel pair (2q, 2q+1), q = lane >> 4 = the MFMA k index, so one read feeds two MFMAs
// V/U stages are double buffered; tile / cout index bit 4 is XOR-swizzled with the pair's parity so
// the b64 fragment reads of a 32-lane group cover all 64 banks.  The accumulator layout of the 16x16
// MFMA keeps a wave's 8 positions of a (tile, cout) in ONE lane, so each wave applies its half of the
// (linear) output transform A^T M A in registers; waves 4-7 hand their 2x2 partial to waves 0-3 through
// LDS once per workgroup; the epilogue (bias, temb addend, residual, scale) is the direct kernel's.
#include "ssde_common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Schedule of the asynchronous issue work (A/B-timed on the MI355X with tools/ab_bench.sh, round 2):
//   0  weight-DMA pieces / halo loads ride between the MFMAs of the matrix phases (round 1)        203-229 TF/s
//   1  matrix phases are pure MFMA + fragment reads, waves 4-7 issue every VMEM instruction in their staging phase:
//      SLOWER (176-208 TF/s) -- the burst of 32 KB of LDS-DMA plus the halo loads, issued at once beside the other
//      waves' fragment reads, costs more than the issue slots it frees (an MFMA wave has idle issue slots anyway)
//   2  schedule 0, and waves 4-7 read their first fragments before the barrier that opens their matrix phase
#ifndef SSDE_WINO_SCHED
#define SSDE_WINO_SCHED 0
#endif
// Issue priority (s_setprio) of the two waves of a SIMD: 1 = the wave in its matrix phase runs at priority 1 (round 1);
// 0 = no priorities; 2 = the STAGING wave's short VALU / LDS bursts outrank the matrix wave's MFMA stream.
// s_memtime traces (tools/wino_trace.py) showed why 2 wins: at 1 the staging wave is starved until the matrix wave has
// issued its last MFMA, so the prologue (GroupNorm + SiLU, ~900 VALU cycles) ran AFTER the 2750-cycle matrix phase
// instead of inside it (phase 1: 3880 -> 3500 cycles); fp32 MFMA and VALU do not co-execute, so the VALU cycles are
// paid either way, but their LDS / VMEM latencies now hide under the other wave's MFMAs.
#ifndef SSDE_WINO_PRIO
#define SSDE_WINO_PRIO 2
#endif
// GroupNorm parameters of the prologue: 0 = global loads issued between the MFMAs of the staging waves' matrix phase
// (round 1: 10 extra VMEM instructions + their address arithmetic per stage in the matrix stream);
// 1 = mean / rstd of the tile's images and gamma / beta parked in LDS once per workgroup, read in store_stage.
// Together with priority 2: +3..5 % on every BASELINE shape (gpurun_out/conv_ab_r2e.txt).
#ifndef SSDE_WINO_GNLDS
#define SSDE_WINO_GNLDS 1
#endif
// s_sleep argument (units of 64 cycles) between the 8 LDS groups of the input transform; 0 = one burst (round 1).
// A/B on the MI355X (gpurun_out/conv_ab_r2g.txt): 1 = +3 % on every shape, 2 = neutral, 3 = slower (the transform itself
// becomes the critical path of the phase).
#ifndef SSDE_WINO_TSLEEP
#define SSDE_WINO_TSLEEP 1
#endif

// -DSSDE_WINO_TRACE (tools/wino_trace.py, a variant library only): s_memtime stamps of waves 0 and 4 of the first and of
// the last workgroup, to see where a workgroup's cycles go (fill, the two phases of a stage, barriers, epilogue).
#ifdef SSDE_WINO_TRACE
__device__ unsigned long long* g_wino_trace;
extern "C" int ssde_debug_wino_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wino_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -5;
}
#define SSDE_TR(slot)                                                                                     \
  do {                                                                                                    \
    if (tr_on) g_wino_trace[tr_base + (slot)] = __builtin_amdgcn_s_memtime();                             \
  } while (0)
#else
#define SSDE_TR(slot) do { } while (0)
#endif

namespace {

constexpr int kThreads = 512;
constexpr int kStageFloats = 16 * 4 * 64 * 2;     // one V or U stage: [pos][pair][64][2]
constexpr int kStagers = 256;                     // waves 4-7 stage (load / prologue / LDS store)
constexpr int kMaxRaw = 4;                        // raw float4 items per staging thread per stage (halo <= 512 pixels)
constexpr int kUItems = kStageFloats / 4 / kStagers;   // weight float4 items per staging thread per stage

struct WinoParams {
  ssde_src src;
  const float* wpk;        // [ceil(C/8)][n_tiles][16][4][64][2] (LDS image per stage)
  int N, H, W, Cout;
  int lTWt, lTHt;          // log2 tiles per patch row / column
  int tiles_x, tiles_per_img, m_tiles, n_tiles;
  const float* bias; const float* chan_add; int chan_add_ld;
  const float* resid; int resid_post;
  float scale;
  float* dst;
  float* gn_part;          // GroupNorm partials of dst (a tile holds part of one image, or several whole images), see ssde_store_tile
};

template <bool kGn>      // GroupNorm prologue: compile-time, so that the staging loads below are straight-line code
__global__ __launch_bounds__(kThreads, 2) void conv_wino_kernel(const WinoParams p) {
  SSDE_LDS(smem);
  float* Vb = smem;                          // [2][kStageFloats]
  float* Ub = smem + 2 * kStageFloats;       // [2][kStageFloats]
  float* raw = smem + 4 * kStageFloats;      // [4 pairs][halo_px][2]; then (SSDE_WINO_GNLDS) the GroupNorm tables
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lq = lane >> 4;

  // XCD-aware order (as conv_mfma.hip): the cout tiles of one pixel tile run on one XCD
  const int bid = blockIdx.x;
  const int xcd = bid & 7, l = bid >> 3;
  const int nt = l % p.n_tiles;
  const int mt = (l / p.n_tiles) * 8 + xcd;
#ifdef SSDE_WINO_TRACE
  const bool tr_on = lane == 0 && (wave & 3) == 0 && (bid == 0 || bid == (int)gridDim.x - 8) && g_wino_trace != nullptr;
  const int tr_base = ((bid == 0 ? 0 : 1) * 2 + (wave >> 2)) * 64;
#endif
  SSDE_TR(0);
  if (mt >= p.m_tiles) return;

  const int TWt = 1 << p.lTWt, THt = 1 << p.lTHt;       // tiles per patch row / column
  const int IMGS = 64 >> (p.lTWt + p.lTHt);
  const int HWd = 2 * TWt + 2, HH = 2 * THt + 2;
  const int halo_px = IMGS * HH * HWd;
  const int img0 = (mt / p.tiles_per_img) * IMGS;
  const int trem = mt % p.tiles_per_img;
  const int ty = trem / p.tiles_x, tx = trem % p.tiles_x;
  const int n0 = nt * 64;

  const ssde_src& s = p.src;
  const int Ctot = s.c0 + s.c1;
  const int nst = (Ctot + 7) >> 3;
  SsdePro pro = ssde_pro_decode(s);
  pro.gn = kGn;
  const int cpg = kGn ? Ctot / s.gn_groups : 1;
  const float inv_cpg = 1.0f / (float)cpg;

  // ---- per-thread raw staging plan: item = (halo pixel, channel half) ----
  const int sid = tid & (kStagers - 1);          // index among the 256 staging threads (waves 4-7) / transform threads (0-3)
  int goff[kMaxRaw], gimg[kMaxRaw];
#pragma unroll
  for (int it = 0; it < kMaxRaw; ++it) {
    const int q = sid + it * kStagers;
    goff[it] = -2; gimg[it] = 0;
    if (q < halo_px * 2) {
      const int hp = q >> 1;
      const int il = hp / (HH * HWd);
      const int rem = hp - il * (HH * HWd);
      const int hy = rem / HWd, hx = rem - hy * HWd;
      const int iy = ty * 2 * THt - 1 + hy, ix = tx * 2 * TWt - 1 + hx;
      const int img = img0 + il;
      const bool inb = img < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      goff[it] = inb ? (img * p.H + iy) * p.W + ix : -1;
      gimg[it] = img;
    }
  }
  // transform item (waves 0-3): this thread's tile and channel pair
  const int t_tile = sid & 63, t_pair = sid >> 6;
  int t_base;
  {
    const int il = t_tile >> (p.lTWt + p.lTHt);
    const int tr = (t_tile >> p.lTWt) & (THt - 1), tc = t_tile & (TWt - 1);
    t_base = (il * HH + 2 * tr) * HWd + 2 * tc;
  }
  const int t_vcol = ((t_tile ^ ((t_pair & 1) << 4)) + t_pair * 64) * 2;

#if SSDE_WINO_GNLDS
  // GroupNorm tables in LDS: (mean, rstd) of every (tile image, group), gamma and beta of every channel
  float* gn_tab = raw + 8 * halo_px;         // [IMGS][groups][2]
  float* gb_tab = gn_tab + 2 * IMGS * (kGn ? s.gn_groups : 0);   // [2][Ctot]
  if (kGn) {
    for (int q = tid; q < IMGS * s.gn_groups; q += kThreads) {
      const int il = q / s.gn_groups, img = img0 + il < p.N ? img0 + il : 0;
      const int gi = img * s.gn_groups + (q - il * s.gn_groups);
      *reinterpret_cast<float2*>(gn_tab + 2 * q) = make_float2(s.gn_mean[gi], s.gn_rstd[gi]);
    }
    for (int q = tid; q < Ctot; q += kThreads) { gb_tab[q] = s.gn_gamma[q]; gb_tab[Ctot + q] = s.gn_beta[q]; }
  }                                          // published by the first pipeline barrier
  int gil[kMaxRaw];
#pragma unroll
  for (int it = 0; it < kMaxRaw; ++it) gil[it] = (goff[it] >= 0 ? gimg[it] - img0 : 0) * (kGn ? s.gn_groups : 0);
#endif
  float4 rv[kMaxRaw];
  float mu[kMaxRaw], rs[kMaxRaw];
  float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
  bool chan_ok = false;
  int c_cur = 0;

  // The loads of a stage are BRANCH-FREE (items outside the image / halo / channel range read a clamped, valid address
  // and are zeroed in store_stage) and cut into 8 PIECES that mfma_stage issues between the MFMAs of its 8 positions:
  // as exec-masked blocks in front of the matrix phase they cost ~870 cycles of every stage.
  const float* ld_bp = nullptr; int ld_C = 0, ld_cg = 0;
  auto load_piece = [&](int st, int k) {
    const int half = sid & 1;
    if (k == 0) {
      const int c_base = st * 8;
      c_cur = c_base;
      const bool second = c_base >= s.c0;
      const float* base = second ? s.p1 : s.p0;
      ld_C = second ? s.c1 : s.c0;
      const int cthr = (second ? c_base - s.c0 : c_base) + half * 4;
      chan_ok = cthr < ld_C;
      ld_bp = base + (chan_ok ? cthr : 0);
      ld_cg = (c_base + half * 4) < Ctot ? c_base + half * 4 : 0;
    }
    // GroupNorm parameters first (two pieces), the halo float4s after them: the LAST load is issued at position 5 at the
    // latest, >= 500 matrix cycles before the barrier behind which store_stage consumes all of them
#if SSDE_WINO_GNLDS
    constexpr int kFirstRaw = 0;
    if (false) {
#else
    constexpr int kFirstRaw = kGn ? 2 : 0;
    if (kGn && k < 2) {
#endif
      if (k == 0) {
        gam = *reinterpret_cast<const float4*>(s.gn_gamma + ld_cg);
        bet = *reinterpret_cast<const float4*>(s.gn_beta + ld_cg);
      }
#pragma unroll
      for (int it = 2 * k; it < 2 * k + 2; ++it) {
        const int gi = (goff[it] >= 0 ? gimg[it] : 0) * s.gn_groups + ld_cg / cpg;
        mu[it] = s.gn_mean[gi];
        rs[it] = s.gn_rstd[gi];
      }
    } else if (k >= kFirstRaw && k < kFirstRaw + 4) {
      const int it = k - kFirstRaw;
      rv[it] = *reinterpret_cast<const float4*>(ld_bp + (size_t)(goff[it] >= 0 ? goff[it] : 0) * ld_C);
    }
  };
  auto load_stage = [&](int st) {
#pragma unroll
    for (int k = 0; k < 8; ++k) load_piece(st, k);
  };
  // weights of stage st: the host packed them as the LDS image, so they go global -> LDS by DMA
  // (a straight 32 KB copy: each of the 4 issuing waves moves a contiguous 8 KB with 8 instructions that differ only in
  // the immediate offset -4096 .. +3072, applied by the hardware to the global and the LDS address alike)
  static_assert(kUItems == 8, "the immediate-offset pieces below are written for 8 x 1 KiB per wave");
  auto dma_piece = [&](int st, float* Un, int k) {
    const int wv = (sid >> 6);
    const float* gsrc = p.wpk + ((size_t)st * p.n_tiles + nt) * kStageFloats + (size_t)wv * 2048 + 1024 + lane * 4;
    float* ldst = Un + wv * 2048 + 1024;
    switch (k) {
      case 0: SSDE_GLDS16_OFF(gsrc, ldst, -4096); break;
      case 1: SSDE_GLDS16_OFF(gsrc, ldst, -3072); break;
      case 2: SSDE_GLDS16_OFF(gsrc, ldst, -2048); break;
      case 3: SSDE_GLDS16_OFF(gsrc, ldst, -1024); break;
      case 4: SSDE_GLDS16_OFF(gsrc, ldst, 0); break;
      case 5: SSDE_GLDS16_OFF(gsrc, ldst, 1024); break;
      case 6: SSDE_GLDS16_OFF(gsrc, ldst, 2048); break;
      default: SSDE_GLDS16_OFF(gsrc, ldst, 3072); break;
    }
  };
  auto dma_weights = [&](int st, float* Un) {
#pragma unroll
    for (int k = 0; k < 8; ++k) dma_piece(st, Un, k);
  };
  // prologue + raw LDS store (channel-pair major)
  auto store_stage = [&]() {
    const int half = sid & 1;
#if SSDE_WINO_GNLDS
    if (kGn) {
      const int cg = (c_cur + half * 4) < Ctot ? c_cur + half * 4 : 0;
      gam = *reinterpret_cast<const float4*>(gb_tab + cg);
      bet = *reinterpret_cast<const float4*>(gb_tab + Ctot + cg);
      const int g = (int)(((float)cg + 0.5f) * inv_cpg);      // cg / cpg, exact for these small integers
#pragma unroll
      for (int it = 0; it < kMaxRaw; ++it) {
        const float2 mr = *reinterpret_cast<const float2*>(gn_tab + 2 * (gil[it] + g));
        mu[it] = mr.x; rs[it] = mr.y;
      }
    }
#endif
#pragma unroll
    for (int it = 0; it < kMaxRaw; ++it) {
      if (goff[it] == -2) continue;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (goff[it] >= 0 && chan_ok)
        v = ssde_pro_apply(rv[it], mu[it], rs[it], gam, bet, (uint32_t)goff[it] * (uint32_t)Ctot + (uint32_t)(c_cur + half * 4), pro);
      const int hp = (sid + it * kStagers) >> 1;
      *reinterpret_cast<float2*>(raw + ((2 * half) * halo_px + hp) * 2) = make_float2(v.x, v.y);
      *reinterpret_cast<float2*>(raw + ((2 * half + 1) * halo_px + hp) * 2) = make_float2(v.z, v.w);
    }
  };
  // V = B^T d B for this thread's (tile, channel pair), both channels at once.
  // The 16 LDS reads and 16 LDS writes are issued in 8 groups of 4 with s_sleep in between (SSDE_WINO_TSLEEP): as one
  // burst at the start of the phase, the 4 x 32 LDS operations of the transforming waves queue in front of the matrix
  // waves' fragment reads, which are prefetched only one position (256 matrix cycles) ahead -- the matrix phase of
  // waves 4-7 took 3260 cycles instead of 2750 (tools/wino_trace.py).  Groups of 4 keep the LDS queue shorter than
  // that prefetch distance, and the transforming waves have ~2000 idle cycles in this phase anyway.
  auto transform = [&](float* Vn) {
    const float* rp = raw + (t_pair * halo_px + t_base) * 2;
    float2 r[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const float2 d0 = *reinterpret_cast<const float2*>(rp + (0 * HWd + x) * 2);
      const float2 d1 = *reinterpret_cast<const float2*>(rp + (1 * HWd + x) * 2);
      const float2 d2 = *reinterpret_cast<const float2*>(rp + (2 * HWd + x) * 2);
      const float2 d3 = *reinterpret_cast<const float2*>(rp + (3 * HWd + x) * 2);
      r[0][x] = make_float2(d0.x - d2.x, d0.y - d2.y);
      r[1][x] = make_float2(d1.x + d2.x, d1.y + d2.y);
      r[2][x] = make_float2(d2.x - d1.x, d2.y - d1.y);
      r[3][x] = make_float2(d1.x - d3.x, d1.y - d3.y);
#if SSDE_WINO_TSLEEP > 0
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_sleep(SSDE_WINO_TSLEEP);
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const float2 v0 = make_float2(r[y][0].x - r[y][2].x, r[y][0].y - r[y][2].y);
      const float2 v1 = make_float2(r[y][1].x + r[y][2].x, r[y][1].y + r[y][2].y);
      const float2 v2 = make_float2(r[y][2].x - r[y][1].x, r[y][2].y - r[y][1].y);
      const float2 v3 = make_float2(r[y][1].x - r[y][3].x, r[y][1].y - r[y][3].y);
      *reinterpret_cast<float2*>(Vn + (y * 4 + 0) * 512 + t_vcol) = v0;
      *reinterpret_cast<float2*>(Vn + (y * 4 + 1) * 512 + t_vcol) = v1;
      *reinterpret_cast<float2*>(Vn + (y * 4 + 2) * 512 + t_vcol) = v2;
      *reinterpret_cast<float2*>(Vn + (y * 4 + 3) * 512 + t_vcol) = v3;
#if SSDE_WINO_TSLEEP > 0
      if (y < 3) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_sleep(SSDE_WINO_TSLEEP);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
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

  // fragment column offsets inside a [pos] slab of 512 floats: ((pair q) * 64 + (index ^ swizzle)) * 2
  const int ph = wave >> 2;                                   // position half: transform rows 2*ph, 2*ph+1
  const int tb0 = ((wave >> 1) & 1) * 32, cb0 = (wave & 1) * 32;
  const int swz = (lq & 1) << 4;
  int aoff[2], boff[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) aoff[a] = ph * 8 * 512 + (lq * 64 + ((tb0 + a * 16 + li) ^ swz)) * 2;
#pragma unroll
  for (int b = 0; b < 2; ++b) boff[b] = ph * 8 * 512 + (lq * 64 + ((cb0 + b * 16 + li) ^ swz)) * 2;

  // fragments of position ps+1 are read while the 8 MFMAs of position ps issue (two register sets)
  // `piece(ps)`: a slice of this wave's asynchronous issue work for a later stage (weight DMA / halo loads), placed
  // between the two MFMA quartets of position ps so that its VALU / VMEM issue slots hide under the matrix pipe.
  // sched_barrier(0) closes every position: nothing moves across, the pieces cannot be hoisted in front of the MFMAs.
  float2 af[2][2], bf[2][2];
  // first fragments of a stage; a wave whose V / U stage is already published issues this BEFORE the barrier that opens
  // its matrix phase (the barrier's lgkmcnt(0) completes it), so the phase starts with an MFMA instead of an LDS round trip
  auto preload = [&](const float* Vc, const float* Uc) {
#pragma unroll
    for (int a = 0; a < 2; ++a) af[0][a] = *reinterpret_cast<const float2*>(Vc + aoff[a]);
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[0][b] = *reinterpret_cast<const float2*>(Uc + boff[b]);
  };
  auto mfma_stage = [&](const float* Vc, const float* Uc, auto&& piece) {
    __builtin_amdgcn_sched_barrier(0);
#if SSDE_WINO_PRIO == 1
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int cur = ps & 1;
      if (ps + 1 < 8) {
#pragma unroll
        for (int a = 0; a < 2; ++a) af[cur ^ 1][a] = *reinterpret_cast<const float2*>(Vc + (ps + 1) * 512 + aoff[a]);
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[cur ^ 1][b] = *reinterpret_cast<const float2*>(Uc + (ps + 1) * 512 + boff[b]);
      }
      // first the even channel of the pair for all four (tile, cout) blocks, then the odd one: a block's second MFMA
      // depends on its first, four independent MFMAs and the piece sit in between
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
      // issue order inside the position: the 4 fragment reads of position ps+1 first (their LDS latency is covered by
      // 256 matrix cycles; left alone, hipcc sinks them below the MFMAs and waits), 4 MFMAs, the piece, 4 MFMAs
      if (ps + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x026, 24, 0);    // VALU | SALU | VMEM read of the piece
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#if SSDE_WINO_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#endif
  };
#if SSDE_WINO_PRIO == 2
#define SSDE_STAGING_HI() __builtin_amdgcn_s_setprio(2)
#define SSDE_STAGING_LO() __builtin_amdgcn_s_setprio(0)
#else
#define SSDE_STAGING_HI() do { } while (0)
#define SSDE_STAGING_LO() do { } while (0)
#endif

  // ---- pipeline prologue: stage 0 staged, stage 1 in flight ----
  const int last = nst - 1;
  SSDE_TR(1);
#if SSDE_WINO_SCHED != 1
  // the asynchronous issue work rides between the MFMAs of the matrix phases
#if SSDE_WINO_GNLDS
  if (ph == 1) load_stage(0);
  else dma_weights(0, Ub);
  if (kGn) __syncthreads();                  // publishes the GroupNorm tables filled above
  if (ph == 1) store_stage();
#else
  if (ph == 1) { load_stage(0); store_stage(); }
  else dma_weights(0, Ub);
#endif
  SSDE_LDS_BARRIER();
  SSDE_TR(2);
  if (ph == 0) { transform(Vb); SSDE_WAIT_VMCNT(0); }
  else load_stage(min(1, last));
  SSDE_LDS_BARRIER();
  SSDE_TR(3);

  for (int st = 0; st < nst; ++st) {
    const float* Vc = Vb + (st & 1) * kStageFloats;
    const float* Uc = Ub + (st & 1) * kStageFloats;
    float* Vn = Vb + ((st + 1) & 1) * kStageFloats;
    float* Un = Ub + ((st + 1) & 1) * kStageFloats;
    if (ph == 0) {
      const int sn = min(st + 1, last);
      preload(Vc, Uc);
      mfma_stage(Vc, Uc, [&](int k) { dma_piece(sn, Un, k); });
    } else {
      SSDE_STAGING_HI();
      if (st + 1 < nst) store_stage();
      SSDE_STAGING_LO();
#if SSDE_WINO_SCHED == 2
      preload(Vc, Uc);
#endif
    }
    if (st < 8) SSDE_TR(4 + st * 4);
    SSDE_LDS_BARRIER();
    if (st < 8) SSDE_TR(5 + st * 4);
    if (ph == 1) {
      const int sn = min(st + 2, last);
#if SSDE_WINO_SCHED != 2
      preload(Vc, Uc);
#endif
      mfma_stage(Vc, Uc, [&](int k) { load_piece(sn, k); });
    } else {
      SSDE_STAGING_HI();
      if (st + 1 < nst) transform(Vn);
      SSDE_STAGING_LO();
      SSDE_WAIT_VMCNT(0);
    }
    if (st < 8) SSDE_TR(6 + st * 4);
    SSDE_LDS_BARRIER();
    if (st < 8) SSDE_TR(7 + st * 4);
  }
#else
  // The matrix phases are PURE MFMA + fragment-read streams.  Every VMEM instruction (the 8 LDS-DMA pieces of the next
  // weight stage, the halo float4 loads and GroupNorm parameters of the stage after next) is issued by waves 4-7 in
  // phase 1, i.e. in the phase in which they do NOT own the matrix pipe: issued between MFMAs, an LDS-DMA piece held
  // the issuing wave's next MFMA back by 60-185 cycles (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"), ~1000
  // cycles per 4096-cycle stage.  In-order VMEM return + the staging waves' vmcnt(0) at the end of their matrix phase
  // (two phases after the issue) publishes the weights; the halo registers are consumed one stage later.
  //   phase 1: waves 0-3 MFMA(k)                       waves 4-7 prologue + raw store (k+1); halo loads (k+2); weight DMA (k+1)
  //   phase 2: waves 4-7 MFMA(k), then vmcnt(0)        waves 0-3 input transform raw -> V(k+1)
  // Waves 4-7 read their first fragments of stage k before the barrier that opens phase 2 (V(k), U(k) are published).
  if (ph == 1) {
    dma_weights(0, Ub);
    load_stage(0);
  }
#if SSDE_WINO_GNLDS
  if (kGn) __syncthreads();                  // publishes the GroupNorm tables filled above
#endif
  if (ph == 1) {
    store_stage();
    load_stage(min(1, last));
    SSDE_WAIT_VMCNT(0);
  }
  SSDE_LDS_BARRIER();
  SSDE_TR(2);
  if (ph == 0) transform(Vb);
  SSDE_LDS_BARRIER();
  SSDE_TR(3);

  for (int st = 0; st < nst; ++st) {
    const float* Vc = Vb + (st & 1) * kStageFloats;
    const float* Uc = Ub + (st & 1) * kStageFloats;
    float* Vn = Vb + ((st + 1) & 1) * kStageFloats;
    float* Un = Ub + ((st + 1) & 1) * kStageFloats;
    if (ph == 0) {
      preload(Vc, Uc);
      mfma_stage(Vc, Uc, [](int) {});
    } else {
      SSDE_STAGING_HI();
      if (st + 1 < nst) {
        store_stage();
        if (st + 2 < nst) load_stage(st + 2);
        dma_weights(st + 1, Un);
      }
      SSDE_STAGING_LO();
      preload(Vc, Uc);
    }
    if (st < 8) SSDE_TR(4 + st * 4);
    SSDE_LDS_BARRIER();
    if (st < 8) SSDE_TR(5 + st * 4);
    if (ph == 1) {
      mfma_stage(Vc, Uc, [](int) {});
      SSDE_WAIT_VMCNT(0);
    } else if (st + 1 < nst) {
      SSDE_STAGING_HI();
      transform(Vn);
      SSDE_STAGING_LO();
    }
    if (st < 8) SSDE_TR(6 + st * 4);
    SSDE_LDS_BARRIER();
    if (st < 8) SSDE_TR(7 + st * 4);
  }
#endif
  SSDE_TR(40);
  __syncthreads();
  SSDE_TR(41);

  // ---- output transform Y = A^T M A: this wave's two transform rows (register local) ----
  // A^T = [[1,1,1,0],[0,1,-1,-1]]; rows py = 2*ph, 2*ph+1 contribute  At[dy][py] * sum_px At[dx][px] M[py][px]
  float yp[2][2][4][4];     // [a][b][r][dy*2+dx]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t[2][2];
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
          t[yy][0] = acc[yy * 4 + 0][a][b][r] + acc[yy * 4 + 1][a][b][r] + acc[yy * 4 + 2][a][b][r];
          t[yy][1] = acc[yy * 4 + 1][a][b][r] - acc[yy * 4 + 2][a][b][r] - acc[yy * 4 + 3][a][b][r];
        }
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          // ph 0: rows 0,1 -> Y0 += t0 + t1, Y1 += t1 ; ph 1: rows 2,3 -> Y0 += t2, Y1 += -t2 - t3
          yp[a][b][r][0 * 2 + dx] = ph == 0 ? t[0][dx] + t[1][dx] : t[0][dx];
          yp[a][b][r][1 * 2 + dx] = ph == 0 ? t[1][dx] : -t[0][dx] - t[1][dx];
        }
      }
  // waves 4-7 hand their partial 2x2 to waves 0-3 (same tile / cout blocks) through LDS: [wave & 3][a][b][r][lane][4]
  float4* xch = reinterpret_cast<float4*>(smem);
  if (ph == 1) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          xch[((((wave & 3) * 2 + a) * 2 + b) * 4 + r) * 64 + lane] =
              make_float4(yp[a][b][r][0], yp[a][b][r][1], yp[a][b][r][2], yp[a][b][r][3]);
  }
  SSDE_TR(42);
  __syncthreads();
  SSDE_TR(43);
  // waves 0-3 complete Y and park it in LDS as [256 local pixels = tile * 4 + dy * 2 + dx][64 couts + 4] (above the
  // 64 KB exchange area); then all 8 waves store it with coalesced float4s (ssde_store_tile)
  constexpr int LDT = 68;
  float* outs = smem + 16384;
  if (ph == 0) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tile = tb0 + a * 16 + 4 * lq + r;
          const float4 o = xch[(((wave * 2 + a) * 2 + b) * 4 + r) * 64 + lane];
          // column bit 4 flipped for odd lq (= rows with bit 4 set): the two 16-lane groups of a 32-lane LDS write cycle
          // hit different banks (row pitch 68: 16 rows are a multiple of 32 banks apart)
          float* dstp = outs + (tile * 4) * LDT + cb0 + ((b ^ (lq & 1)) * 16) + li;
          dstp[0 * LDT] = yp[a][b][r][0] + o.x;
          dstp[1 * LDT] = yp[a][b][r][1] + o.y;
          dstp[2 * LDT] = yp[a][b][r][2] + o.z;
          dstp[3 * LDT] = yp[a][b][r][3] + o.w;
        }
  }
  SSDE_TR(44);
  __syncthreads();
  SSDE_TR(45);
  SsdeEpi e{p.bias, p.chan_add, p.chan_add_ld, p.resid, p.resid_post, p.scale, p.dst, p.Cout, p.gn_part};
  const int gn_entry = p.gn_part ? img0 * p.tiles_per_img + trem : -1;        // IMGS > 1: tiles_per_img == 1, trem == 0
  const int rpi_log2 = IMGS > 1 ? 8 - (6 - p.lTWt - p.lTHt) : 30;             // rows per image: 256 / IMGS
  ssde_store_tile<256, 64, kThreads, 4, 1>(outs, LDT, n0, e, [&](int row, size_t& pix, int& img) {
    const int tile = row >> 2, dy = (row >> 1) & 1, dx = row & 1;
    const int il = tile >> (p.lTWt + p.lTHt);
    const int tr = (tile >> p.lTWt) & (THt - 1), tc = tile & (TWt - 1);
    img = img0 + il;
    const int oy = (ty * THt + tr) * 2 + dy, ox = (tx * TWt + tc) * 2 + dx;
    if (img >= p.N || oy >= p.H || ox >= p.W) return false;
    pix = ((size_t)img * p.H + oy) * p.W + ox;
    return true;
  }, gn_entry, rpi_log2, p.N * p.tiles_per_img);
  SSDE_TR(46);
}

int pow2_floor(int v) { int q = 1; while (q * 2 <= v) q *= 2; return q; }

}  // namespace

// Launched by ssde_conv2d when a->tile == SSDE_TILE_WINOGRAD (the caller packs w_main for this kernel).
int ssde_conv_wino_launch(const ssde_conv_args* a, void* stream, int* lds_out) {
  SSDE_REQUIRE(a && a->dst && a->main.p0 && a->w_main, "conv(winograd): null args");
  SSDE_REQUIRE(a->ksize == 3 && a->stride == 1 && a->pad == 1, "conv(winograd): needs 3x3, stride 1, pad 1");
  SSDE_REQUIRE(a->aux.p0 == nullptr, "conv(winograd): fused 1x1 source not supported (issue it as a second conv)");
  SSDE_REQUIRE(a->h_in == a->h_out && a->w_in == a->w_out && a->h_out % 2 == 0 && a->w_out % 2 == 0 && a->h_out >= 8 && a->w_out >= 8,
               "conv(winograd): even same-size output of at least 8x8 needed (got %dx%d)", a->h_out, a->w_out);
  const ssde_src& s = a->main;
  SSDE_REQUIRE(s.c0 > 0 && s.c0 % 4 == 0 && s.c1 % 4 == 0 && (s.c1 == 0 || (s.p1 && s.c0 % 8 == 0)),
               "conv(winograd): channels must be multiples of 4 (concat boundary of 8)");
  if (s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU) {
    SSDE_REQUIRE(s.gn_groups > 0 && (s.c0 + s.c1) % s.gn_groups == 0 && ((s.c0 + s.c1) / s.gn_groups) % 4 == 0,
                 "conv(winograd): GroupNorm needs channels-per-group %% 4 == 0");
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "conv(winograd): GroupNorm pointers missing");
  }
  SSDE_REQUIRE(s.drop_thresh == 0 || s.drop_seed, "conv(winograd): dropout seed pointer missing");
  WinoParams p;
  p.src = s; p.wpk = a->w_main;
  p.N = a->n; p.H = a->h_out; p.W = a->w_out; p.Cout = a->c_out;
  const int twt = pow2_floor((a->w_out / 2) < 8 ? (a->w_out / 2) : 8);
  int tht = 64 / twt; if (tht > a->h_out / 2) tht = a->h_out / 2;
  tht = pow2_floor(tht);
  const int imgs = 64 / (twt * tht);
  p.lTWt = ssde_ilog2(twt); p.lTHt = ssde_ilog2(tht);
  p.tiles_x = ssde_cdiv(a->w_out, 2 * twt);
  p.tiles_per_img = p.tiles_x * ssde_cdiv(a->h_out, 2 * tht);
  p.m_tiles = ssde_cdiv(a->n, imgs) * p.tiles_per_img;
  p.n_tiles = ssde_cdiv(a->c_out, 64);
  p.bias = a->bias; p.chan_add = a->chan_add; p.chan_add_ld = a->chan_add_ld;
  p.resid = a->resid; p.resid_post = a->resid_post; p.scale = a->out_scale; p.dst = a->dst;
  p.gn_part = a->gn_part;
  // GroupNorm partials: a tile is part of one image (tiles_per_img slices) or holds `imgs` whole images (1 slice each,
  // tiles_per_img == 1); 256 / imgs rows per image >= the 32 rows of one epilogue trip
  const bool gn_ok = a->c_out % 4 == 0 && (imgs == 1 || (p.tiles_per_img == 1 && imgs <= 8));
  SSDE_REQUIRE(!a->gn_part || gn_ok, "conv(winograd): GroupNorm partials not available for this tiling");
  if (lds_out && stream == reinterpret_cast<void*>(1)) { *lds_out = gn_ok ? p.tiles_per_img * (kThreads / 64) : 0; return SSDE_OK; }
  const int halo_px = imgs * (2 * tht + 2) * (2 * twt + 2);
  SSDE_REQUIRE(halo_px * 2 <= kMaxRaw * kStagers, "conv(winograd): halo of %d pixels exceeds the staging plan", halo_px);
  int lds = (4 * kStageFloats + 4 * halo_px * 2) * 4;
#if SSDE_WINO_GNLDS
  if (s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU) lds += (2 * imgs * s.gn_groups + 2 * (s.c0 + s.c1)) * 4;
  SSDE_REQUIRE(lds <= 160 * 1024, "conv(winograd): %d bytes of LDS", lds);
#endif
  if (lds_out) { *lds_out = lds; return SSDE_OK; }
  static std::atomic<bool> attr_set{false};   // once, before any stream capture
  if (!attr_set) {
    SSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SSDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const bool gn = a->main.pro_mode == SSDE_PRO_GN || a->main.pro_mode == SSDE_PRO_GN_SILU;
  const dim3 grid(ssde_cdiv(p.m_tiles, 8) * 8 * p.n_tiles);
  if (gn) hipLaunchKernelGGL(conv_wino_kernel<true>, grid, dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
  else hipLaunchKernelGGL(conv_wino_kernel<false>, grid, dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
