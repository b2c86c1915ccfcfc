// 3x3 / stride 1 / pad 1 convolution by Winograd F(4x4, 3x3), two-kernel form, matrix kernel fed from REGISTERS.
//
//   wino4_xform_vq_kernel (wino4_xform.hip)   V[pos][Cin/4][t][4] = B^T pro(x) B      HBM-bound pass
//   conv_wino4r_kernel (this file)            Y = A^T [ sum_ci U .* V ] A             no LDS, no barrier in the main loop
//
// Round 4's matrix kernel (conv_wino4g.hip, now tools/experiments/conv_wino4g_lds_fed/) moved both operands through LDS by LDS-DMA and sits at ~3050 cycles per 4-channel
// stage for 2304 matrix cycles.  tools/microbench/fill_path.hip (profiles/r5_fill_path_microbench.txt) separates the causes:
// beside a running fp32 MFMA chain the CU takes 64 KB per stage from L2 at NO cost by either route (LDS-DMA 2376 cycles per
// stage, global_load_dwordx4 into VGPRs 2304 = the matrix bound), so neither the TA, nor the TCP -> LDS write, nor the L2 is the
// 18 B/clk/CU "fill bound" of round 4.  What the product pays for is its lock step: one barrier per stage makes every wave wait
// for the slowest LDS-DMA piece of the stage (V streams from HBM), with the weights fetched only half a stage ahead.  And the
// 36 KB of weights per stage are wave-PRIVATE: the LDS round trip shares them with nobody.
//
// This kernel therefore keeps NOTHING in LDS during the main loop.  A workgroup is 8 waves = 32 tiles x 64 couts x 36 transform
// positions, as in conv_wino4.hip, but the 72 (position, cout half) blocks of 32 x 32 are dealt so that a wave needs as few
// DIFFERENT operand bytes as possible: wave w owns positions 4 w .. 4 w + 3 with BOTH cout halves (two MFMA blocks share one A
// fragment) and one cout half (w & 1) of position 32 + (w >> 1): nine accumulator blocks, 18 MFMAs per 4-channel stage, fed by
//   A: V[pos][stage][tile = lane & 31][2 (lane >> 5) .. + 1]        one global_load_dwordx2 per position: FIVE per stage
//   B: U[pos][cout = (lane & 31), 32 + (lane & 31)][2 (lane >> 5) .. + 1]   packed per LANE (SSDE_PACK_WINO4R): one dwordx4 per
//                                                                   full position + one dwordx2 for the half position
// straight into the registers its v_mfma_f32_32x32x2_f32 read.  (The first version dealt wave (q, h) the positions q + 4 j of cout
// half h: every V run was loaded by two waves, 72 KB per stage and CU -- and the CU's path from L2 saturates at ~28 B/clk when
// every CU pulls at once, tools/microbench/fill_path.hip and profiles/r5_wino4r_simd_pair_balance_ab.txt: ~2500 cycles per stage
// whatever the waves' priorities.  This dealing moves 56 KB.)  A register is re-loaded for stage st + 2 right after the MFMAs
// that consumed it in stage st (two register sets), so every load has two whole stages to land, ~18 loads are in flight per wave
// at any time, and the only synchronisation is the wave's own counted s_waitcnt vmcnt (five per stage).  Waves drift freely: a
// late V piece stalls one wave while its SIMD sibling keeps the matrix pipe busy.  LDS is used by the epilogue only
// (conv_wino4.hip's: products -> LDS, A^T M A, parked tile, shared coalesced store with GroupNorm partials).
#include "ssde_common.h"
#include <atomic>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// -DSSDE_W4R_TRACE (a variant library only): s_memtime stamps of waves 0 and 7 of the first workgroup
#ifdef SSDE_W4R_TRACE
__device__ unsigned long long* g_w4r_trace;
extern "C" int ssde_debug_w4r_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_w4r_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -5;
}
#define SSDE_RT(slot)                                                                                     \
  do {                                                                                                    \
    if (tr_on) g_w4r_trace[tr_base + (slot)] = __builtin_amdgcn_s_memtime();                              \
  } while (0)
#else
#define SSDE_RT(slot) do { } while (0)
#endif


#ifndef SSDE_W4R_PERSIST
#define SSDE_W4R_PERSIST 0          // (1: persistent workgroups for launches of more than one round -- until the GPU parity run is green, a variant only)
#endif

namespace {

constexpr int kNP = 9;                              // accumulator blocks per wave: (4 positions) x (2 cout halves) + 1
constexpr int kNV = 5;                              // positions a wave loads V for
constexpr int kPos = 36, kTiles = 32;
constexpr int kURegion = kNP * 32 * 4;              // floats of a stage's weight image only wave (q, h) reads (4.5 KB)
constexpr int kUFloats = 8 * kURegion;              // one (stage, 64-cout tile) of the image
// 8 waves, 32 tiles x 64 couts, one workgroup per CU (conv_wino4.hip's shape).  A 4-wave form with two workgroups per CU, the
// second one started half a tile late so that one's epilogue would run under the other's MFMAs, was built and measured level:
// the epilogue is VALU / issue work and fp32 MFMAs share the VALU datapath (tools/experiments/conv_wino4r_two_workgroups_per_cu/)
constexpr int kWaves = 8, kThreads = 64 * kWaves, kBN = 64;
constexpr int kLdm = kBN + 2, kLdt = kBN + 4;       // pitches of the product exchange and of the parked output tile
constexpr int kLds = kPos * 16 * kLdm * 4;          // the epilogue's product exchange (the parked tile fits inside)
static_assert(256 * kLdt <= kPos * 16 * kLdm, "parked tile");

struct Wino4rParams {
  const float* v;          // [36][Ctot / 4][T][4]
  const float* wpk;        // SSDE_PACK_WINO4R: [Ctot / 4][n_tiles][8 waves][4 positions x [64 lanes][4] | [64 lanes][2]]
  int N, H, W, Cout, Ctot;
  int lTWt, lTHt;
  int tiles_x, tiles_per_img, m_tiles, n_tiles;
  const float* bias; const float* chan_add; int chan_add_ld;
  const float* resid; int resid_post;
  float scale;
  float* dst;
  float* gn_part;
  int T, tiles_h, tiles_w;
  int ksplit;              // 1, 2 or 4 workgroups per tile, each reducing its share of the channel stages (the kernel's kKs)
  unsigned* sync;          // ksplit > 1: this launch's (shares started, shares handed over) pairs, one per tile (conv_mfma.hip's
                           // g_conv_sync: zero before and after)
  int ticket_off;          // float index of the hand-over ticket in LDS (behind the epilogue's exchange area)
  int total_work;          // kPersist: tiles (pixel tile, cout tile) of the launch, dealt bid, bid + grid, ...
};

// kKs = 2 or 4: that many workgroups per tile, each reducing its share of the channel stages (conv_wino4.hip's split, the same
// protocol: shares are dealt in the order the workgroups START -- an atomic counter per tile -- so share k only ever waits for
// shares < k, which are resident or done; the sums are formed in the fixed order ((s0 + s1) + s2) + s3).  The maps this is for
// are the 8x8 ones at batch 256: 1024 tiles = 32 workgroup tiles x 4 cout tiles = 128 workgroups on 256 CUs; two shares each
// cover the chip with one workgroup per CU, and the matrix loop of a share is half as long.  (Eight shares on the 4x4 maps -- one
// tile per image -- were built and measured slower than the direct kernel: the hand-over chain outlasts the 8 stages a share
// computes, profiles/r5_wino4r_4x4_maps_eight_shares_rejected.txt, tools/experiments/conv_wino4r_4x4_maps_eight_shares/.)
//
// kPersist (unsplit launches of more than one round of workgroups): the grid is one workgroup per CU and a workgroup walks the
// tiles bid, bid + grid, bid + 2 grid, ... itself (the same tile -> XCD map: the grid is a multiple of 8).  What that buys is the
// START of a tile: a freshly dispatched workgroup spends 6-15 k cycles before its first MFMA (dispatch of eight waves one after the
// other, kernel arguments, address arithmetic, 20 loads whose first touches of V miss everywhere --
// profiles/r5_wino4r_later_workgroups_trace.txt: "setup" + "first loads issued").  Here the first two stages' loads of the NEXT tile
// are issued from the epilogue of the current one, after its second output round is parked and before that round is stored: they
// land under the store phase, and the next tile's first MFMA follows the last store directly.  The loads sit in the very registers
// the loop reads (the register sets are free during the epilogue); tests/test_isa_guards.py checks that nothing touches them
// between the issue and the loop's first counted wait.  Stores count in vmcnt too: the waits of the first stage may see up to nine
// younger stores outstanding and wait for a few more of the (older) loads than necessary -- never for fewer.
template <int kKs, bool kPersist = false>
__global__ __launch_bounds__(kThreads, 2) void conv_wino4r_kernel(const Wino4rParams p) {
  constexpr bool kSplit = kKs > 1;
  static_assert(!(kSplit && kPersist), "the split shares are dealt by tickets: one tile per workgroup");
  SSDE_LDS(smem);                               // the epilogue's only (+ the split's ticket)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  // wave w: positions 4 w + i (i = 0..3), both cout halves -> blocks 2 i, 2 i + 1; position 32 + (w >> 1), cout half w & 1 -> block 8
  auto pos_of = [&](int i) { return i < 4 ? 4 * wave + i : 32 + (wave >> 1); };
  const int TWt = 1 << p.lTWt, THt = 1 << p.lTHt;
  const int IMGS = kTiles >> (p.lTWt + p.lTHt);
  // XCD-aware order (conv_wino4.hip): the cout tiles of one pixel tile run on one XCD at the same time -- they read the same V
  auto mt_of = [&](int w) { return ((w >> 3) / (p.n_tiles * kKs)) * 8 + (w & 7); };   // (the kKs workgroups of a tile: same XCD, 8 * n_tiles blocks apart)
  auto nt_of = [&](int w) { return (w >> 3) % p.n_tiles; };
  // ---- operand addresses of a tile.  A: the lane's byte offset of position slot j inside one stage's [36][Q][T][4] view of V
  // (32-bit: the launcher checks V < 4 GB); the stage rides in the scalar base.  Tiles outside the batch / the image read tile 0's
  // run (a valid address; their outputs are never stored).  B: the wave's 4.5 KB of a stage, lane-major; four 1 KB pieces at
  // immediates -2048 .. 1024 around the base and one 512-byte piece behind them
  const size_t v_stage = (size_t)p.T * 16;
  const size_t u_stage = (size_t)p.n_tiles * kUFloats * 4;
  const uint32_t u_off = (uint32_t)lane * 16u, u_off8 = 2048u + (uint32_t)lane * 8u;
  auto operands = [&](int mt_, int nt_, int st_off_, uint32_t (&vo)[kNV], const char*& vs_, const char*& us_) {
    const int img0_ = (mt_ / p.tiles_per_img) * IMGS, trem_ = mt_ % p.tiles_per_img;
    const int ty_ = trem_ / p.tiles_x, tx_ = trem_ % p.tiles_x;
    const int il = li >> (p.lTWt + p.lTHt);
    const int tr = (li >> p.lTWt) & (THt - 1), tc = li & (TWt - 1);
    const int img = img0_ + il, yy = ty_ * THt + tr, xx = tx_ * TWt + tc;
    const uint32_t t = (img < p.N && yy < p.tiles_h && xx < p.tiles_w) ? (uint32_t)((img * p.tiles_h + yy) * p.tiles_w + xx) : 0u;
    const uint32_t QT = (uint32_t)(p.Ctot >> 2) * (uint32_t)p.T;
#pragma unroll
    for (int j = 0; j < kNV; ++j) vo[j] = ((uint32_t)pos_of(j) * QT + t) * 16u + 8u * (uint32_t)lh;
    vs_ = reinterpret_cast<const char*>(p.v) + (size_t)st_off_ * v_stage;      // (local) stage st: + st * T * 16 bytes
    us_ = reinterpret_cast<const char*>(p.wpk + ((size_t)nt_ * kWaves + wave) * kURegion) + 2048 + (size_t)st_off_ * u_stage;
  };
  // TWO register sets per operand: stage st lives in set st & 1 and is loaded while stage st - 2 is consumed, so a wave has two
  // whole stages of loads in flight.  (With one set -- the first version -- a stage could not be shorter than one memory
  // latency: ~3000 cycles for a wave alone on its SIMD, profiles/r5_wino4r_v2_two_workgroups_per_cu.txt, against 1152 matrix
  // cycles; two waves per SIMD at 2304 cycles each sat right at that bound.)
  ssde_f32x2 va[2][kNV], u2[2];
  ssde_f32x4 u4[2][4];
  uint32_t v_off[kNV];
  const char* vs = nullptr;
  const char* us = nullptr;
  // the loads of one stage into register set S: V0 U0 | V1 U1 | V2 U2 | V3 U3 | V4 U4 (the order the loop's counted waits assume)
#define SSDE_W4R_ROUND_AT(S, VO, VN, UN)                                                           \
  do {                                                                                             \
    SSDE_GLOAD8_I_SAFE(va[S][0], VO[0], VN, 0); SSDE_GLOAD16_I_SAFE(u4[S][0], u_off, UN, -2048);             \
    SSDE_GLOAD8_I_SAFE(va[S][1], VO[1], VN, 0); SSDE_GLOAD16_I_SAFE(u4[S][1], u_off, UN, -1024);             \
    SSDE_GLOAD8_I_SAFE(va[S][2], VO[2], VN, 0); SSDE_GLOAD16_I_SAFE(u4[S][2], u_off, UN, 0);                 \
    SSDE_GLOAD8_I_SAFE(va[S][3], VO[3], VN, 0); SSDE_GLOAD16_I_SAFE(u4[S][3], u_off, UN, 1024);              \
    SSDE_GLOAD8_I_SAFE(va[S][4], VO[4], VN, 0); SSDE_GLOAD8_I_SAFE(u2[S], u_off8, UN, 0);                    \
  } while (0)
  const int total_work = kPersist ? p.total_work : 0;
  bool primed = false;                           // kPersist: the first two stages of this tile are already in flight

  for (int work = blockIdx.x;;) {                // (one trip unless kPersist)
  const int bid = work;
  const int nt = nt_of(work);
  const int mt = mt_of(work);
#ifdef SSDE_W4R_TRACE
  const bool tr_on = lane == 0 && (wave == 0 || wave == kWaves - 1) && g_w4r_trace != nullptr && bid == (int)g_w4r_trace[255];   // (the host names the workgroup)
  const int tr_base = (wave == 0 ? 0 : 1) * 128;
#endif
  SSDE_RT(0);
  if (mt >= p.m_tiles) {                         // (the grid is padded to whole groups of 8 pixel tiles)
    if constexpr (kPersist) { work += gridDim.x; if (work < total_work) continue; }
    return;
  }

  const int img0 = (mt / p.tiles_per_img) * IMGS;
  const int trem = mt % p.tiles_per_img;
  const int ty = trem / p.tiles_x, tx = trem % p.tiles_x;
  const int n0 = nt * kBN;
  // this workgroup's share of the reduction: stages st_off .. st_off + nst (an even count: the launcher splits only channel
  // counts that are multiples of 8 kKs)
  int nst = p.Ctot >> 2, ks = 0, st_off = 0;
  unsigned* sy = kSplit ? p.sync + 2 * ((size_t)mt * p.n_tiles + nt) : nullptr;
  if constexpr (kSplit) {
    int* ticket = reinterpret_cast<int*>(smem + p.ticket_off);
    if (tid == 0) *ticket = (int)atomicAdd(sy, 1u);
    __syncthreads();
    ks = __builtin_amdgcn_readfirstlane(*ticket);
    const int nst_all = nst;
    st_off = nst_all * ks / kKs;
    nst = nst_all * (ks + 1) / kKs - st_off;
  }
  if (!primed) operands(mt, nt, st_off, v_off, vs, us);

  f32x16 acc[kNP];
#pragma unroll
  for (int j = 0; j < kNP; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // The VMEM queue of a wave is a ring of 10 loads per stage, always in this order (loads return in order):
  //   V0 U0 | V1 U1 | V2 U2 | V3 U3 | V4 U4              (position i of the wave: its V run, then its weights)
  // each re-issued for stage st + 2 right after the MFMAs that consumed it in stage st.  Position i needs Ui of this stage (index
  // 2 i + 1; Vi is older): the loads issued since are the rest of its own round (8 - 2 i), the whole round of stage st + 1 (10)
  // and what this stage has re-issued so far (2 i) = 18, so s_waitcnt vmcnt(18) before every position.  kMode 1 = the stage
  // before the last (nothing is issued any more, the last stage's round is in flight): vmcnt(18 - 2 i); kMode 2 = the last
  // stage: vmcnt(8 - 2 i).
#define SSDE_W4R_LOADV(S, J) SSDE_GLOAD8_I(va[S][J], v_off[J], vn, 0)
#define SSDE_W4R_MFMA(S, J, BLK, B0, B1)                                                           \
  do {                                                                                             \
    acc[BLK] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[S][J].x, (B0), acc[BLK], 0, 0, 0);          \
    acc[BLK] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[S][J].y, (B1), acc[BLK], 0, 0, 0);          \
  } while (0)
#define SSDE_W4R_POS(S, I, IMM)                                                                    \
  do {                                                                                             \
    SSDE_WAIT_VMCNT_FOR(kMode == 0 ? 18 : kMode == 1 ? 18 - 2 * (I) : 8 - 2 * (I), va[S][I], u4[S][I]); \
    SSDE_W4R_MFMA(S, I, 2 * (I), u4[S][I].x, u4[S][I].y);                                          \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SSDE_W4R_MFMA(S, I, 2 * (I) + 1, u4[S][I].z, u4[S][I].w);                                      \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    if (kMode == 0) { SSDE_W4R_LOADV(S, I); SSDE_GLOAD16_I(u4[S][I], u_off, un, IMM); }            \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)

  SSDE_RT(1);
  if (!primed) {
    SSDE_W4R_ROUND_AT(0, v_off, vs, us);
    SSDE_W4R_ROUND_AT(1, v_off, vs + v_stage, us + u_stage);
  }
  SSDE_RT(2);
  // stage st in register set S (= st & 1, a compile-time constant: the loop below is unrolled by two)
  auto stage = [&](auto SetC, auto ModeC, const int st) __attribute__((always_inline)) {
    constexpr int S = decltype(SetC)::value, kMode = decltype(ModeC)::value;
    const char* vn = vs + (size_t)(st + 2) * v_stage;
    const char* un = us + (size_t)(st + 2) * u_stage;
    if (st < 8) SSDE_RT(8 + st * 2);
    SSDE_W4R_POS(S, 0, -2048);
    SSDE_W4R_POS(S, 1, -1024);
    SSDE_W4R_POS(S, 2, 0);
    if (st < 8) SSDE_RT(8 + st * 2 + 1);
    SSDE_W4R_POS(S, 3, 1024);
    SSDE_WAIT_VMCNT_FOR(kMode == 0 ? 18 : kMode == 1 ? 10 : 0, va[S][4], u2[S]);
    SSDE_W4R_MFMA(S, 4, 8, u2[S].x, u2[S].y);
    __builtin_amdgcn_sched_barrier(0);
    if (kMode == 0) { SSDE_W4R_LOADV(S, 4); SSDE_GLOAD8_I(u2[S], u_off8, un, 0); }
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
    using Steady = std::integral_constant<int, 0>; using Penult = std::integral_constant<int, 1>; using Last = std::integral_constant<int, 2>;
    // (the launcher admits even stage counts only -- input channels a multiple of 8 -- so the tail is the same straight-line
    //  code for every launch: a branch per tail shape made hipcc spill hundreds of registers around the joins)
    int st = 0;
    // (The two waves of a SIMD share its matrix pipe and the older one wins every arbitration; a balance -- each wave posts its
    //  stage number in LDS, reads its partner's and takes the lower issue priority while it is ahead -- evened them out and
    //  changed nothing: the pair is bound by the bytes it pulls, not by who runs first.  profiles/r5_wino4r_simd_pair_balance_ab.txt)
    for (; st + 3 < nst; st += 2) { stage(S0{}, Steady{}, st); stage(S1{}, Steady{}, st + 1); }
    stage(S0{}, Penult{}, st);
    stage(S1{}, Last{}, st + 1);
  }
#undef SSDE_W4R_POS
#undef SSDE_W4R_MFMA
#undef SSDE_W4R_LOADV
  SSDE_RT(3);

  // ---- epilogue (conv_wino4.hip): 16 tiles (accumulator rows r < 8, then r >= 8 of every wave) at a time: products -> LDS
  // M[pos][16 tiles][64 couts], A^T M A per (tile, cout pair), parked 4x4 outputs, shared coalesced store ----
  SsdeEpi e{p.bias, p.chan_add, p.chan_add_ld, p.resid, p.resid_post, p.scale, p.dst, p.Cout, p.gn_part};
  const int gn_base = !p.gn_part ? -1 : (IMGS == 1 ? (img0 * p.tiles_per_img + trem) * 2 : img0);
  const int rpi_log2 = IMGS > 2 ? 8 - (4 - p.lTWt - p.lTHt) : 30;
  // (kPersist: the per-lane indices of the epilogue are re-derived from an opaque copy of the thread id in every tile -- as loop
  //  invariants of the tile loop hipcc computed the ~70 LDS addresses of the exchange once, kept them across the matrix loop and
  //  spilled 105 registers around it)
  int tid_e = tid;
  if constexpr (kPersist) asm volatile("" : "+v"(tid_e));
  const int li_e = tid_e & 31, lh_e = (tid_e >> 5) & 1;
  const int e_tl = tid_e / (kBN / 2), e_cp = tid_e % (kBN / 2);
  float* park = smem;
  int next_work = 0;
#pragma unroll
  for (int rnd = 0; rnd < 2; ++rnd) {
    // (a later tile of a persistent workgroup: the exchange below overwrites the previous tile's parked rows -- every thread must be
    //  done reading them; eight waves that just left the same matrix loop arrive here within a few hundred cycles of each other)
    if (kPersist && rnd == 0 && primed) SSDE_LDS_BARRIER();
    auto pixfn = [&](int row, size_t& pix, int& img) {
      const int tile = rnd * 16 + (row >> 4), dy = (row >> 2) & 3, dx = row & 3;
      const int il = tile >> (p.lTWt + p.lTHt);
      const int tr = (tile >> p.lTWt) & (THt - 1), tc = tile & (TWt - 1);
      img = img0 + il;
      const int oy = (ty * THt + tr) * 4 + dy, ox = (tx * TWt + tc) * 4 + dx;
      if (img >= p.N || oy >= p.H || ox >= p.W) return false;
      pix = ((size_t)img * p.H + oy) * p.W + ox;
      return true;
    };
#pragma unroll
    for (int j = 0; j < kNP; ++j)
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int r = rnd * 8 + r8;
        const int tl = (r & 3) + 4 * lh_e + 8 * ((r >> 2) & 1);
        const int pos = pos_of(j >> 1), half = j < 8 ? (j & 1) : (wave & 1);      // block j of this wave
        smem[(pos * 16 + tl) * kLdm + half * 32 + li_e] = acc[j][r];
      }
    SSDE_RT(32 + rnd * 4);   // products written (before the barrier)
    SSDE_LDS_BARRIER();      // (LDS-only barriers throughout: a __syncthreads() would wait out the previous round's global stores)
    SSDE_RT(33 + rnd * 4);
    float2 y[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) y[a][b] = make_float2(0.f, 0.f);
    const float* mp = smem + e_tl * kLdm + 2 * e_cp;
#pragma unroll
    for (int px = 0; px < 6; ++px) {
      float2 m[6];
#pragma unroll
      for (int py = 0; py < 6; ++py) m[py] = *reinterpret_cast<const float2*>(mp + (py * 6 + px) * (16 * kLdm));
      float2 t[4];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float m0 = c ? m[0].y : m[0].x, m1 = c ? m[1].y : m[1].x, m2 = c ? m[2].y : m[2].x;
        const float m3 = c ? m[3].y : m[3].x, m4 = c ? m[4].y : m[4].x, m5 = c ? m[5].y : m[5].x;
        const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
        const float t0 = m0 + s1 + s2, t1 = d1 + 2.f * d2, t2 = s1 + 4.f * s2, t3 = d1 + 8.f * d2 + m5;
        if (c) { t[0].y = t0; t[1].y = t1; t[2].y = t2; t[3].y = t3; }
        else   { t[0].x = t0; t[1].x = t1; t[2].x = t2; t[3].x = t3; }
      }
      constexpr float kA[6][4] = {{1.f, 0.f, 0.f, 0.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, -1.f, 1.f, -1.f},
                                  {1.f, 2.f, 4.f, 8.f}, {1.f, -2.f, 4.f, -8.f}, {0.f, 0.f, 0.f, 1.f}};
#pragma unroll
      for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx)
          if (kA[px][dx] != 0.f) { y[dy][dx].x += kA[px][dx] * t[dy].x; y[dy][dx].y += kA[px][dx] * t[dy].y; }
    }
    SSDE_RT(34 + rnd * 4);                     // output transform done
    SSDE_LDS_BARRIER();                        // every thread has read its products: the parked tile may overwrite them
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
      for (int dx = 0; dx < 4; ++dx)
        *reinterpret_cast<float2*>(park + (e_tl * 16 + dy * 4 + dx) * kLdt + 2 * e_cp) = y[dy][dx];
    SSDE_LDS_BARRIER();
    SSDE_RT(35 + rnd * 4);                     // tile parked: the store phase starts
    if constexpr (kPersist) {
      if (rnd == 1) {
        // the next tile's first two stages, requested before this tile's last stores (the operand registers and v_off / vs / us
        // are dead since the matrix loop ended)
        next_work = work + (int)gridDim.x;
        while (next_work < total_work && mt_of(next_work) >= p.m_tiles) next_work += (int)gridDim.x;
        if (next_work < total_work) {
          operands(mt_of(next_work), nt_of(next_work), 0, v_off, vs, us);
          SSDE_W4R_ROUND_AT(0, v_off, vs, us);
          SSDE_W4R_ROUND_AT(1, v_off, vs + v_stage, us + u_stage);
        }
      }
    }
    const int gn_entry = gn_base < 0 ? -1 : (IMGS == 1 ? gn_base + rnd : gn_base + rnd * (IMGS >> 1));
    const int gn_max = IMGS == 1 ? p.N * p.tiles_per_img * 2 : p.N;
    if constexpr (kSplit) {
      // split reduction (conv_wino4.hip's hand-over): share 0 leaves its raw 4x4 outputs in the tile's own part of dst, shares
      // 1 .. kKs - 2 add theirs to them in turn, the last share adds its own and runs the epilogue, round by round.  The sums
      // travel as agent-scope 16-byte accesses, coherent at the device level by themselves (no __threadfence(): that is a
      // write-back of the whole L2 per workgroup, conv_mfma.hip)
      const bool first = ks == 0, last = ks == kKs - 1;
      // (sy[1]: the shares that have handed over round 0 in its low half, round 1 in its high half -- share k starts adding to
      //  round 0 while share k - 1 is still busy with its round 1)
      if (!first) {
        if (tid == 0)
          while (((__hip_atomic_load(sy + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (16 * rnd)) & 0xffffu) != (unsigned)ks)
            __builtin_amdgcn_s_sleep(4);
        __syncthreads();
      }
      SSDE_RT(40 + rnd);                     // (split) the previous share's round is there
      constexpr int kIters = 256 * 16 / kThreads, kBatch = 4;      // (4 float4 of a thread at a time: round 0 still holds half of the accumulators)
      static_assert(kBatch == 4, "SSDE_WAIT_VMCNT_FOR4");
#pragma unroll
      for (int ib = 0; ib < kIters; ib += kBatch) {
        float* tp[kBatch];
        float* dp[kBatch];
#pragma unroll
        for (int it = 0; it < kBatch; ++it) {
          const int q = tid + (ib + it) * kThreads;
          const int row = q >> 4, j = (q & 15) * 4;
          size_t pix; int img;
          const bool ok = pixfn(row, pix, img) && n0 + j < p.Cout;       // c_out % 4 == 0 on this path
          tp[it] = park + row * kLdt + j;
          dp[it] = ok ? p.dst + pix * p.Cout + n0 + j : nullptr;
        }
        // (16 bytes per access at agent scope: as dword atomics -- conv_wino4.hip's form -- the hand-over was 64 loads and 64
        //  stores per thread and share)
        ssde_f32x4 o[kBatch];
        if (!first) {
#pragma unroll
          for (int it = 0; it < kBatch; ++it) SSDE_GLOAD16_AGENT(o[it], dp[it] ? dp[it] : p.dst);     // the batch's loads in flight before the first add
          SSDE_WAIT_VMCNT_FOR4(0, o[0], o[1], o[2], o[3]);
#pragma unroll
          for (int it = 0; it < kBatch; ++it) o[it] += *reinterpret_cast<const ssde_f32x4*>(tp[it]);   // (sum so far) + own share
        } else {
#pragma unroll
          for (int it = 0; it < kBatch; ++it) o[it] = *reinterpret_cast<const ssde_f32x4*>(tp[it]);
        }
        if (last) {
#pragma unroll
          for (int it = 0; it < kBatch; ++it) *reinterpret_cast<ssde_f32x4*>(tp[it]) = o[it];
        } else {
#pragma unroll
          for (int it = 0; it < kBatch; ++it)
            if (dp[it]) SSDE_GSTORE16_AGENT(dp[it], o[it]);
        }
      }
      SSDE_RT(42 + rnd);                     // (split) sums formed: handed on, or parked for the epilogue
      if (!last) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0);      // every store of this thread is acknowledged ...
        __syncthreads();                    // ... and every thread's (and park may be refilled by round 1)
        if (tid == 0) atomicAdd(sy + 1, rnd == 0 ? 1u : 0x10000u);
        SSDE_RT(44 + rnd);                   // (split) stores acknowledged, round signalled
        if (rnd == 0) continue;
        return;
      }
      __syncthreads();
      if (rnd == 1 && tid == 0) { sy[0] = 0u; sy[1] = 0u; }              // ready for the next launch that is dealt these slots
    }
    // (Fetching the residual rows ahead of the exchange / transform / park phases was built twice and measured neutral:
    //  profiles/r6_w4r_epilogue_prefetch_v1_in_scratch_rejected.txt, r6_w4r_epilogue_prefetch_v2_neutral.txt)
    if (rnd == 0) ssde_store_tile<256, kBN, kThreads, 4, 0>(park, kLdt, n0, e, pixfn, gn_entry, rpi_log2, gn_max, kPersist ? tid_e : -1);
    else ssde_store_tile<256, kBN, kThreads, 8, 0>(park, kLdt, n0, e, pixfn, gn_entry, rpi_log2, gn_max, kPersist ? tid_e : -1);
    if (rnd == 0) { SSDE_LDS_BARRIER(); SSDE_RT(4); }
  }
  SSDE_RT(5);
  if constexpr (kPersist) {
    if (next_work < total_work) { work = next_work; primed = true; continue; }
  }
  return;
  }                                              // the next tile of a persistent workgroup
#undef SSDE_W4R_ROUND_AT
}

int pow2_floor(int v) { int q = 1; while (q * 2 <= v) q *= 2; return q; }

}  // namespace

// over how many workgroups per tile (1, 2 or 4) a launch of `wgs` tiles over `ctot` input channels splits its reduction: the
// shares must fit ONE round of workgroups (one per CU: the kernel owns a CU's registers), every share an even number of at least
// 8 channel stages.  SSDE_CONVF_NO_KSPLIT in the launch's flags switches it off.
int ssde_conv_wino4r_splits(int wgs, int ctot, int c_out, unsigned flags) {
  if ((flags & SSDE_CONVF_NO_KSPLIT) || c_out % 4 != 0) return 1;
  const int cus = ssde_num_cus();
  // (measured on 8x8 maps, profiles/r5_wino4r_split_8x8.txt, r5_wino4r_split_8x8_batch128.txt: at batch 256 two shares of 32 / 64
  //  stages take 0.077 / 0.112 ms against 0.098 / 0.163 unsplit, two shares of 16 stages -- 128 input channels -- are level; at
  //  batch 128 four shares 0.063 / 0.082 against 0.092 / 0.153.  The last share spends 13-15 k cycles on the hand-over,
  //  profiles/r5_wino4r_split_trace.txt: what ~6 stages cost)
  if (wgs * 4 <= cus && ctot % 32 == 0 && ctot >= 256) return 4;
  if (wgs * 2 <= cus && ctot % 16 == 0 && ctot >= 256) return 2;
  return 1;
}

// stream == (void*)1 with lds_out: plan-only query of the GroupNorm slices per image (conv_mfma.hip, ssde_conv_gn_slices)
int ssde_conv_wino4r_launch(const ssde_conv_args* a, void* stream, int* lds_out) {
  SSDE_REQUIRE(a && a->dst && a->main.p0 && a->w_main, "conv(winograd 4x4, register-fed): null args");
  SSDE_REQUIRE(a->ksize == 3 && a->stride == 1 && a->pad == 1, "conv(winograd 4x4, register-fed): needs 3x3, stride 1, pad 1");
  SSDE_REQUIRE(a->aux.p0 == nullptr, "conv(winograd 4x4, register-fed): fused 1x1 source not supported (issue it as a second conv)");
  SSDE_REQUIRE(a->h_in == a->h_out && a->w_in == a->w_out && a->h_out % 4 == 0 && a->w_out % 4 == 0 && a->h_out >= 8 && a->w_out >= 8,
               "conv(winograd 4x4, register-fed): same-size output, multiples of 4, at least 8x8 (got %dx%d)", a->h_out, a->w_out);
  const ssde_src& s = a->main;
  SSDE_REQUIRE(s.c0 > 0 && s.c0 % 4 == 0 && s.c1 % 4 == 0 && (s.c1 == 0 || s.p1), "conv(winograd 4x4, register-fed): channels must be multiples of 4");
  SSDE_REQUIRE((s.c0 + s.c1) % 8 == 0, "conv(winograd 4x4, register-fed): input channels must be a multiple of 8 (the stage loop runs in pairs)");
  Wino4rParams p;
  p.v = a->wino_v; p.wpk = a->w_main;
  p.N = a->n; p.H = a->h_out; p.W = a->w_out; p.Cout = a->c_out; p.Ctot = s.c0 + s.c1;
  const int twt = pow2_floor((a->w_out / 4) < 8 ? (a->w_out / 4) : 8);
  int tht = kTiles / twt; if (tht > a->h_out / 4) tht = a->h_out / 4;
  tht = pow2_floor(tht);
  const int imgs = kTiles / (twt * tht);
  p.lTWt = ssde_ilog2(twt); p.lTHt = ssde_ilog2(tht);
  p.tiles_x = ssde_cdiv(a->w_out, 4 * twt);
  p.tiles_per_img = p.tiles_x * ssde_cdiv(a->h_out, 4 * tht);
  p.m_tiles = ssde_cdiv(a->n, imgs) * p.tiles_per_img;
  p.n_tiles = ssde_cdiv(a->c_out, kBN);
  p.bias = a->bias; p.chan_add = a->chan_add; p.chan_add_ld = a->chan_add_ld;
  p.resid = a->resid; p.resid_post = a->resid_post; p.scale = a->out_scale; p.dst = a->dst;
  p.gn_part = a->gn_part;
  p.tiles_h = a->h_out / 4; p.tiles_w = a->w_out / 4; p.T = a->n * p.tiles_h * p.tiles_w;
  const bool gn_ok = a->c_out % 4 == 0 && (imgs == 1 || (p.tiles_per_img == 1 && imgs <= 8));
  SSDE_REQUIRE(!a->gn_part || gn_ok, "conv(winograd 4x4, register-fed): GroupNorm partials not available for this tiling");
  if (lds_out && stream == reinterpret_cast<void*>(1)) {
    *lds_out = gn_ok ? (imgs == 1 ? 2 * p.tiles_per_img : 1) * (kThreads / 64) : 0;
    return SSDE_OK;
  }
  p.ticket_off = kLds / 4;
  const int lds = kLds + 16;
  SSDE_REQUIRE(lds <= 160 * 1024, "conv(winograd 4x4, register-fed): %d bytes of LDS", lds);
  if (lds_out) { *lds_out = lds; return SSDE_OK; }
  SSDE_REQUIRE(a->wino_v, "conv(winograd 4x4, register-fed): the transformed-input buffer (ssde_conv_args.wino_v) is missing");
  SSDE_REQUIRE(36ull * (unsigned long long)p.T * (unsigned)p.Ctot * 4ull < (1ull << 32),
               "conv(winograd 4x4, register-fed): a transformed input of 4 GB or more is not addressable by this kernel");
  // Split the reduction when the launch would leave half of the CUs or more without a workgroup (8x8 maps at batch 256: 128
  // tiles of 8 images x 64 couts): ssde_conv_wino4r_splits, the rule engine.Lowering.wino_ok mirrors.  (dst is the hand-over
  // buffer: not when the residual aliases it)
  const int wgs = ssde_cdiv(p.m_tiles, 8) * 8 * p.n_tiles;
  p.ksplit = a->resid != a->dst ? ssde_conv_wino4r_splits(wgs, p.Ctot, a->c_out, a->flags) : 1;
  p.sync = p.ksplit > 1 ? ssde_conv_sync_slots(p.m_tiles * p.n_tiles) : nullptr;
  if (!p.sync) p.ksplit = 1;
  // more than one round of unsplit workgroups: one persistent workgroup per CU walks the tiles (the grid stays a multiple of 8:
  // the tile -> XCD map).  SSDE_W4R_PERSIST=0 at compile time (A/B variant) keeps one workgroup per tile.
  const int cus = ssde_num_cus() / 8 * 8;
  const bool persist = SSDE_W4R_PERSIST && p.ksplit == 1 && cus >= 8 && wgs > cus;
  p.total_work = wgs;
  const dim3 grid(persist ? cus : wgs * p.ksplit);
  auto go = [&](auto kfn, std::atomic<bool>& attr_set) {
    if (!attr_set) {                            // once per instantiation, before any stream capture
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return false;
      attr_set = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
    return true;
  };
  static std::atomic<bool> set[4];
  const bool ok = p.ksplit == 4 ? go(conv_wino4r_kernel<4>, set[2]) : p.ksplit == 2 ? go(conv_wino4r_kernel<2>, set[1])
                : persist ? go(conv_wino4r_kernel<1, true>, set[3]) : go(conv_wino4r_kernel<1>, set[0]);
  SSDE_REQUIRE(ok, "conv(winograd 4x4, register-fed): hipFuncSetAttribute failed");
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
