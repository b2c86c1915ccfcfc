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
};

template <bool kGn>      // GroupNorm prologue: compile-time, so that the staging loads below are straight-line code
__global__ __launch_bounds__(kThreads, 2) void conv_wino_kernel(const WinoParams p) {
  SSDE_LDS(smem);
  float* Vb = smem;                          // [2][kStageFloats]
  float* Ub = smem + 2 * kStageFloats;       // [2][kStageFloats]
  float* raw = smem + 4 * kStageFloats;      // [4 pairs][halo_px][2]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, lq = lane >> 4;

  // XCD-aware order (as conv_mfma.hip): the cout tiles of one pixel tile run on one XCD
  const int bid = blockIdx.x;
  const int xcd = bid & 7, l = bid >> 3;
  const int nt = l % p.n_tiles;
  const int mt = (l / p.n_tiles) * 8 + xcd;
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
    constexpr int kFirstRaw = kGn ? 2 : 0;
    if (kGn && k < 2) {
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
  // V = B^T d B for this thread's (tile, channel pair), both channels at once
  auto transform = [&](float* Vn) {
    const float* rp = raw + (t_pair * halo_px + t_base) * 2;
    float2 d[4][4];
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int x = 0; x < 4; ++x) d[y][x] = *reinterpret_cast<const float2*>(rp + (y * HWd + x) * 2);
    float2 r[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      r[0][x] = make_float2(d[0][x].x - d[2][x].x, d[0][x].y - d[2][x].y);
      r[1][x] = make_float2(d[1][x].x + d[2][x].x, d[1][x].y + d[2][x].y);
      r[2][x] = make_float2(d[2][x].x - d[1][x].x, d[2][x].y - d[1][x].y);
      r[3][x] = make_float2(d[1][x].x - d[3][x].x, d[1][x].y - d[3][x].y);
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
  auto mfma_stage = [&](const float* Vc, const float* Uc, auto&& piece) {
    float2 af[2][2], bf[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) af[0][a] = *reinterpret_cast<const float2*>(Vc + aoff[a]);
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[0][b] = *reinterpret_cast<const float2*>(Uc + boff[b]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
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
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- pipeline prologue: stage 0 staged, stage 1 in flight ----
  const int last = nst - 1;
  if (ph == 1) { load_stage(0); store_stage(); }
  else dma_weights(0, Ub);
  SSDE_LDS_BARRIER();
  if (ph == 0) { transform(Vb); SSDE_WAIT_VMCNT(0); }
  else load_stage(min(1, last));
  SSDE_LDS_BARRIER();

  for (int st = 0; st < nst; ++st) {
    const float* Vc = Vb + (st & 1) * kStageFloats;
    const float* Uc = Ub + (st & 1) * kStageFloats;
    float* Vn = Vb + ((st + 1) & 1) * kStageFloats;
    float* Un = Ub + ((st + 1) & 1) * kStageFloats;
    // phase 1: waves 0-3 start the weight DMA of st+1 and run the matrix pipe; waves 4-7 apply the prologue to the
    // halo of st+1 (loaded during their previous matrix phase) and store it
    // (past the last stage both issue a redundant copy of it instead of branching: the issue code stays in the
    // basic block of the MFMAs; the stray DMA lands in the idle U buffer and is waited for before the epilogue)
    if (ph == 0) {
      const int sn = min(st + 1, last);
      mfma_stage(Vc, Uc, [&](int k) { dma_piece(sn, Un, k); });
    } else if (st + 1 < nst) {
      store_stage();
    }
    SSDE_LDS_BARRIER();
    // phase 2: waves 4-7 put the halo loads of st+2 in flight and run the matrix pipe; waves 0-3 transform st+1 and
    // make sure their DMA has landed before the barrier that publishes U(st+1)
    if (ph == 1) {
      const int sn = min(st + 2, last);
      mfma_stage(Vc, Uc, [&](int k) { load_piece(sn, k); });
    } else {
      if (st + 1 < nst) transform(Vn);
      SSDE_WAIT_VMCNT(0);
    }
    SSDE_LDS_BARRIER();
  }
  __syncthreads();

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
  __syncthreads();
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
          float* dstp = outs + (tile * 4) * LDT + cb0 + b * 16 + li;
          dstp[0 * LDT] = yp[a][b][r][0] + o.x;
          dstp[1 * LDT] = yp[a][b][r][1] + o.y;
          dstp[2 * LDT] = yp[a][b][r][2] + o.z;
          dstp[3 * LDT] = yp[a][b][r][3] + o.w;
        }
  }
  __syncthreads();
  SsdeEpi e{p.bias, p.chan_add, p.chan_add_ld, p.resid, p.resid_post, p.scale, p.dst, p.Cout};
  ssde_store_tile(outs, 256, LDT, 64, n0, e, kThreads, [&](int row, size_t& pix, int& img) {
    const int tile = row >> 2, dy = (row >> 1) & 1, dx = row & 1;
    const int il = tile >> (p.lTWt + p.lTHt);
    const int tr = (tile >> p.lTWt) & (THt - 1), tc = tile & (TWt - 1);
    img = img0 + il;
    const int oy = (ty * THt + tr) * 2 + dy, ox = (tx * TWt + tc) * 2 + dx;
    if (img >= p.N || oy >= p.H || ox >= p.W) return false;
    pix = ((size_t)img * p.H + oy) * p.W + ox;
    return true;
  });
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
  const int halo_px = imgs * (2 * tht + 2) * (2 * twt + 2);
  SSDE_REQUIRE(halo_px * 2 <= kMaxRaw * kStagers, "conv(winograd): halo of %d pixels exceeds the staging plan", halo_px);
  const int lds = (4 * kStageFloats + 4 * halo_px * 2) * 4;
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
