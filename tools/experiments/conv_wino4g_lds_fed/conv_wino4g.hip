// 3x3 / stride 1 / pad 1 convolution by Winograd F(4x4, 3x3), the TWO-kernel form: the transformed input arrives from HBM.
//
//   wino4_xform_vq_kernel (below)   V[pos][Cin/4][t][4] = B^T pro(x) B      HBM-bound: reads x once, writes 2.25x of it
//   conv_wino4g_kernel              Y = A^T [ sum_ci U .* V ] A             the matrix kernel: no prologue, no input transform
//
// conv_wino4.hip does both in one kernel, and on gfx950 that costs matrix time twice over: the GroupNorm / SiLU prologue and the
// two transform passes are VALU work that SERIALISES with fp32 MFMAs on a SIMD (~950 of a stage's ~4600 cycles, beside 2304
// matrix cycles), and every 64-cout workgroup of a pixel tile repeats them.  Measured bound (profiles/r4_wino4_two_kernel.txt,
// conv_wino4_kernel built with -DSSDE_W4_EXP_NOXFORM=1: the stage body without prologue and transform): 0.353 -> 0.250 ms at
// 128 -> 128 @32x32, 0.292 -> 0.204 ms at 256 -> 256 @16x16, 0.525 -> 0.359 ms at 512 -> 256 @16x16, batch 256.  The price is the
// extra pass: x read once more and V (2.25 x) written and read; it is paid back where a V tile feeds four or more cout tiles
// (the 256-cout layers), roughly level at two (the 128-cout layers at 32x32) -- the lowering chooses (engine.Lowering.wino_ok).
// In a training program V is wanted anyway, by the F(4x4,3x3) weight gradient (ssde_wgrad_args.v_pre).
//
// Main kernel: workgroup = 8 waves = 32 tiles x 64 couts x 36 positions as in conv_wino4.hip (same accumulator layout, same
// epilogue), K = 4 input channels per stage.  Both operands come by LDS-DMA:
//   U  the packed weight image of conv_wino4.hip (SSDE_PACK_WINO4), wave-private 4.5 KB per stage, double buffered;
//   V  36 runs of 512 bytes per stage (a workgroup's 32 tiles are consecutive t), a 3-deep ring: the pieces of stage s + 2 are
//      issued in stage s, three per wave (two full-wave pieces = 2 positions each, one 16-lane piece of the last four positions).
// A wave counts its own pieces (vmcnt, hand-counted below); the ONE LDS-only barrier per stage publishes V(s + 1), whose pieces
// have had a whole stage to land.  The stage body is 18 MFMAs, 18 ds_read_b64 and 8 vector-memory instructions per wave.
#include "ssde_common.h"
#include <atomic>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// -DSSDE_W4G_TRACE (a variant library only): s_memtime stamps of waves 0 and 7 of the first workgroup
#ifdef SSDE_W4G_TRACE
__device__ unsigned long long* g_w4g_trace;
extern "C" int ssde_debug_w4g_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_w4g_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -5;
}
#define SSDE_GT(slot)                                                                                     \
  do {                                                                                                    \
    if (tr_on) g_w4g_trace[tr_base + (slot)] = __builtin_amdgcn_s_memtime();                              \
  } while (0)
#else
#define SSDE_GT(slot) do { } while (0)
#endif

// one LDS-DMA piece on lanes 0-15 (a run of 256 bytes), M0 set: exec is narrowed and restored inside the statement
#ifndef SSDE_GLDS16_S_LO16
#define SSDE_GLDS16_S_LO16(voff, sbase, lds_wave_base, imm)                                                              \
  do {                                                                                                                    \
    unsigned long long ssde_exec_save_;                                                                                   \
    asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xffff\n\t"                                 \
                 "global_load_lds_dwordx4 %2, %3 offset:%4\n\ts_mov_b64 exec, %0"                                        \
                 : "=&s"(ssde_exec_save_)                                                                                 \
                 : "s"(__builtin_amdgcn_readfirstlane(                                                                    \
                       (int)(uintptr_t)(__attribute__((address_space(3))) void*)(lds_wave_base))),                        \
                   "v"(voff), "s"(sbase), "n"(imm)                                                                        \
                 :);                                                                                                      \
  } while (0)
#endif

namespace {

constexpr int kWaves = 8, kThreads = kWaves * 64;
constexpr int kNP = 9, kPS = 4;                     // positions per wave; wave (q, h) owns positions q + kPS * j
constexpr int kEpiThreads = 512;
constexpr int kPos = 36, kTiles = 32, kKc = 4;
constexpr int kVP = kTiles * kKc;                   // 128 floats per position: unpadded, the LDS-DMA destination of a 512-byte run
constexpr int kVFloats = kPos * kVP;                // one V stage, 18 KB
constexpr int kVRing = 3;
constexpr int kUFloats = kPos * 64 * kKc;           // one U stage, 36 KB
constexpr int kURegion = kNP * 32 * kKc;            // the floats of a stage only wave (q, h) reads
constexpr int kLdm = 66, kLdt = 68;                 // pitches of the product exchange and of the parked output tile (conv_wino4.hip)
constexpr int kPF = 3;                              // fragment reads run this many positions ahead of their MFMAs
// timing experiments (wrong results on purpose, variant libraries only): the stage body without its LDS-DMA pieces / without its
// fragment reads -- where the ~700 cycles per stage beside the 2304 matrix cycles go (profiles/r4_wino4_two_kernels.txt)
#ifndef SSDE_W4G_EXP_NODMA
#define SSDE_W4G_EXP_NODMA 0
#endif
#ifndef SSDE_W4G_EXP_NOREAD
#define SSDE_W4G_EXP_NOREAD 0
#endif
// 1: the stage barrier in the middle of the stage (see the stage body); 0: at its end.  Measured level (0.246-0.259 vs 0.2515 ms
// at 128 -> 128 @32x32, 0.191 vs 0.1905 at 256 -> 256 @16x16, profiles/r4_wino4_two_kernels.txt): the simpler form is the default
#ifndef SSDE_W4G_MID_BARRIER
#define SSDE_W4G_MID_BARRIER 0
#endif

struct Wino4gParams {
  const float* v;          // [36][Ctot / 4][T][4]
  const float* wpk;        // conv_wino4.hip's image: [Ctot / 4][n_tiles][8 waves][9][32][4]
  int N, H, W, Cout, Ctot;
  int lTWt, lTHt;
  int tiles_x, tiles_per_img, m_tiles, n_tiles;
  const float* bias; const float* chan_add; int chan_add_ld;
  const float* resid; int resid_post;
  float scale;
  float* dst;
  float* gn_part;
  int T, tiles_h, tiles_w;
};

__global__ __launch_bounds__(kThreads, kWaves / 4) void conv_wino4g_kernel(const Wino4gParams p) {
  SSDE_LDS(smem);
  float* Vb = smem;                            // [3][kVFloats]
  float* Ub = smem + kVRing * kVFloats;        // [2][8 waves][9 positions][32 couts][4]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware order (conv_wino4.hip): the cout tiles of one pixel tile run on one XCD -- they read the same V runs
  const int bid = blockIdx.x;
  const int xcd = bid & 7, l = bid >> 3;
  const int nt = l % p.n_tiles;
  const int mt = (l / p.n_tiles) * 8 + xcd;
#ifdef SSDE_W4G_TRACE
  const bool tr_on = lane == 0 && (wave == 0 || wave == 7) && bid == 0 && g_w4g_trace != nullptr;
  const int tr_base = (wave == 0 ? 0 : 1) * 128;
#endif
  SSDE_GT(0);
  if (mt >= p.m_tiles) return;

  const int TWt = 1 << p.lTWt, THt = 1 << p.lTHt;
  const int IMGS = kTiles >> (p.lTWt + p.lTHt);
  const int img0 = (mt / p.tiles_per_img) * IMGS;
  const int trem = mt % p.tiles_per_img;
  const int ty = trem / p.tiles_x, tx = trem % p.tiles_x;
  const int n0 = nt * 64;
  const int nst = p.Ctot >> 2;

  // ---- V pieces of this wave: piece i = 0, 1: positions 2 (wave + 8 i) + (lane >> 5), tile lane & 31; piece 2 (lanes 0-15):
  // position 32 + (wave >> 1), tile 16 (wave & 1) + lane.  The lane's byte offset inside a stage's [36][T][4] slab of V is a
  // 32-bit VGPR (the launcher checks V < 4 GB), the stage rides in the scalar base.  Tiles outside the batch / the image read
  // tile 0's run (a valid address; their outputs are never stored).
  auto tile_t = [&](int tile) {
    const int il = tile >> (p.lTWt + p.lTHt);
    const int tr = (tile >> p.lTWt) & (THt - 1), tc = tile & (TWt - 1);
    const int img = img0 + il, yy = ty * THt + tr, xx = tx * TWt + tc;
    return (img < p.N && yy < p.tiles_h && xx < p.tiles_w) ? (img * p.tiles_h + yy) * p.tiles_w + xx : 0;
  };
  const uint32_t Q = (uint32_t)(p.Ctot >> 2);
  uint32_t v_voff[3];
  v_voff[0] = ((uint32_t)(2 * wave + lh) * Q * (uint32_t)p.T + (uint32_t)tile_t(li)) * 16u;
  v_voff[1] = ((uint32_t)(2 * (wave + 8) + lh) * Q * (uint32_t)p.T + (uint32_t)tile_t(li)) * 16u;
  v_voff[2] = ((uint32_t)(32 + (wave >> 1)) * Q * (uint32_t)p.T + (uint32_t)tile_t(16 * (wave & 1) + (lane & 15))) * 16u;
  auto v_base = [&](int st) { return p.v + (size_t)st * p.T * 4; };
  auto v_issue = [&](int st, float* Vr, int i) __attribute__((always_inline)) {
    const float* vb = v_base(st);
    if (i == 0) SSDE_GLDS16_S(v_voff[0], vb, Vr + (2 * wave) * kVP, 0);
    else if (i == 1) SSDE_GLDS16_S(v_voff[1], vb, Vr + (2 * (wave + 8)) * kVP, 0);
    else SSDE_GLDS16_S_LO16(v_voff[2], vb, Vr + 32 * kVP + wave * 64, 0);
  };
  // ---- U pieces (conv_wino4.hip): wave (q, h) moves and is the only reader of its 4.5 KB of a stage, pieces of 1 KB at
  // immediates -2048 .. +2048 around one scalar base (the last one on lanes 0-31)
  const uint32_t w_voff = (uint32_t)((wave * kURegion + 2 * 256 + lane * 4) * 4);
  auto w_base = [&](int st) { return p.wpk + ((size_t)st * p.n_tiles + nt) * kUFloats; };
  auto w_ldst = [&](float* Un) { return Un + wave * kURegion + 2 * 256; };

  const int wq = wave >> 1;
  f32x16 acc[kNP];
#pragma unroll
  for (int j = 0; j < kNP; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int a_off = li * 4 + 2 * lh, b_off = wave * kURegion + li * 4 + 2 * lh;

  // ---- pipeline fill: V(0), U(0), V(1); V(0) landed (8 or 5 younger pieces may be in flight) and published ----
  SSDE_GT(1);
  v_issue(0, Vb, 0); v_issue(0, Vb, 1); v_issue(0, Vb, 2);
  {
    const float* wb = w_base(0);
    float* ld = w_ldst(Ub);
    SSDE_GLDS16_S(w_voff, wb, ld, -2048);
    SSDE_GLDS16_S_SAME_BASE(w_voff, wb, ld, -1024);
    SSDE_GLDS16_S_SAME_BASE(w_voff, wb, ld, 0);
    SSDE_GLDS16_S_SAME_BASE(w_voff, wb, ld, 1024);
    SSDE_GLDS16_S_SAME_BASE_LO32(w_voff, wb, ld, 2048);
  }
  if (nst > 1) {
    v_issue(1, Vb + kVFloats, 0); v_issue(1, Vb + kVFloats, 1); v_issue(1, Vb + kVFloats, 2);
    SSDE_WAIT_VMCNT_FENCE(8);
  } else {
    SSDE_WAIT_VMCNT_FENCE(5);
  }
  SSDE_LDS_BARRIER();
  SSDE_GT(2);

  // ---- one stage.  H1: a stage st + 1 exists (its weight pieces P' are issued here, its V is awaited at the end), H2: a stage
  // st + 2 exists (its V pieces V'' are issued here).  VMEM queue of a wave, in issue order (loads and LDS-DMA return in order):
  //   on entry             [P0 P1 P2 P3 P4 | V'0 V'1 V'2]        (V' = V(st + 1), only if H1)
  //   head   P0, P1 landed (positions 0..3)                       vmcnt(3 + 3 H1)
  //   slot 0 issues P'0; slot 1 reads position 4 = P2             vmcnt(2 + 3 H1 + H1)
  //   slots 1, 2 issue P'1, P'2; slot 3 reads position 6 = P3     vmcnt(1 + 3 H1 + 3 H1)
  //   slots 3, 4 issue P'3, P'4; slot 5 reads position 8 = P4     vmcnt(3 H1 + 5 H1)
  //   after slot 4 (H1)    V' landed, then the barrier             vmcnt(5)   (the five P' may fly; covers P4 as well)
  //   slots 5, 6, 7 issue V''0, V''1, V''2
  int vcur = 0;                                  // ring slot of V(st)
  auto stage = [&](auto H1, auto H2, const int st) __attribute__((always_inline)) {
    constexpr bool has1 = decltype(H1)::value && !SSDE_W4G_EXP_NODMA, has2 = decltype(H2)::value && !SSDE_W4G_EXP_NODMA;
    constexpr int n1 = has1 ? 1 : 0, n2 = has2 ? 1 : 0;
    const int cur = st & 1, nxt = cur ^ 1;
    const float* Vc = Vb + vcur * kVFloats;
    const float* Uc = Ub + cur * kUFloats;
    const int v2 = vcur == 0 ? 2 : vcur - 1;       // (vcur + 2) % 3
    float* Vn2 = Vb + v2 * kVFloats;
    ssde_f32x2 af[kPF + 1], bf[kPF + 1];
    const float* wb = w_base(has1 ? st + 1 : st);
    float* wl = w_ldst(Ub + nxt * kUFloats);
    ssde_lds_cfloat* va = (ssde_lds_cfloat*)(Vc + wq * kVP + a_off);
    ssde_lds_cfloat* ua = (ssde_lds_cfloat*)(Uc + b_off);
    SSDE_OPAQUE_VGPR(va);
    SSDE_OPAQUE_VGPR(ua);
    if (st < 8) SSDE_GT(8 + st * 4);
    SSDE_WAIT_VMCNT_FENCE(3 + 3 * n1);
#pragma unroll
    for (int j = 0; j < kPF; ++j) {
      af[j] = *(ssde_lds_cfloat2*)(va + kPS * j * kVP);
      bf[j] = *(ssde_lds_cfloat2*)(ua + j * 128);
    }
    __builtin_amdgcn_sched_barrier(0);
#define SSDE_W4G_POS(J)                                                                                          \
    do {                                                                                                         \
      if ((J) + kPF < kNP && !SSDE_W4G_EXP_NOREAD) {                                                             \
        af[((J) + kPF) % (kPF + 1)] = *(ssde_lds_cfloat2*)(va + kPS * ((J) + kPF) * kVP);                        \
        bf[((J) + kPF) % (kPF + 1)] = *(ssde_lds_cfloat2*)(ua + ((J) + kPF) * 128);                              \
      }                                                                                                          \
      acc[J] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[(J) % (kPF + 1)].x, bf[(J) % (kPF + 1)].x, acc[J], 0, 0, 0); \
      acc[J] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[(J) % (kPF + 1)].y, bf[(J) % (kPF + 1)].y, acc[J], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
    } while (0)
    SSDE_W4G_POS(0);
    if (has1) SSDE_GLDS16_S(w_voff, wb, wl, -2048);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_WAIT_VMCNT_FENCE(2 + 4 * n1);
    SSDE_W4G_POS(1);
    if (has1) SSDE_GLDS16_S_SAME_BASE(w_voff, wb, wl, -1024);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4G_POS(2);
    if (has1) SSDE_GLDS16_S_SAME_BASE(w_voff, wb, wl, 0);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_WAIT_VMCNT_FENCE(1 + 6 * n1);
    SSDE_W4G_POS(3);
    if (has1) SSDE_GLDS16_S_SAME_BASE(w_voff, wb, wl, 1024);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4G_POS(4);
    if (has1) SSDE_GLDS16_S_SAME_BASE_LO32(w_voff, wb, wl, 2048);
    __builtin_amdgcn_sched_barrier(0);
    if (st < 8) SSDE_GT(8 + st * 4 + 1);
#if SSDE_W4G_MID_BARRIER
    // the stage's ONE barrier sits HERE, not at the stage boundary: it publishes V(st + 1) (own pieces landed: only the five P'
    // may be in flight, which also covers P4) half a stage before its first reader, and it orders this stage's V'' pieces (slots
    // 5-7, ring slot of V(st - 1)) behind every wave's last read of V(st - 1).  The waves then run from one stage into the next
    // without meeting: the two waves of a SIMD drift up to half a stage apart, and one wave's head (counted wait + the first
    // fragment reads, ~300 cycles without an MFMA) falls under the other's MFMAs instead of beside its head.
    if (has1) { SSDE_WAIT_VMCNT_FENCE(5); SSDE_LDS_BARRIER(); }
    else SSDE_WAIT_VMCNT_FENCE(0);
#else
    SSDE_WAIT_VMCNT_FENCE(8 * n1);
#endif
    if (st < 8) SSDE_GT(8 + st * 4 + 2);
    SSDE_W4G_POS(5);
    if (has2) v_issue(st + 2, Vn2, 0);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4G_POS(6);
    if (has2) v_issue(st + 2, Vn2, 1);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4G_POS(7);
    if (has2) v_issue(st + 2, Vn2, 2);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4G_POS(8);
#undef SSDE_W4G_POS
#if !SSDE_W4G_MID_BARRIER
    if (has1) SSDE_WAIT_VMCNT_FENCE(5 + 3 * n2);
    SSDE_LDS_BARRIER();
#endif
    if (st < 8) SSDE_GT(8 + st * 4 + 3);
    vcur = vcur == 2 ? 0 : vcur + 1;
  };
  {
    using T = std::true_type; using F = std::false_type;
    int st = 0;
    for (; st + 2 < nst; ++st) stage(T{}, T{}, st);
    if (st + 1 < nst) { stage(T{}, F{}, st); ++st; }
    stage(F{}, F{}, st);
  }
#if SSDE_W4G_MID_BARRIER
  SSDE_LDS_BARRIER();                            // every wave has read its last fragments: the epilogue reuses the LDS
#endif
  SSDE_GT(3);

  // ---- epilogue (conv_wino4.hip): 16 tiles (accumulator rows r < 8, then r >= 8 of every wave) at a time: products -> LDS
  // M[pos][16 tiles][64 couts], A^T M A per (tile, cout pair), parked 4x4 outputs, shared coalesced store ----
  SsdeEpi e{p.bias, p.chan_add, p.chan_add_ld, p.resid, p.resid_post, p.scale, p.dst, p.Cout, p.gn_part};
  const int gn_base = !p.gn_part ? -1 : (IMGS == 1 ? (img0 * p.tiles_per_img + trem) * 2 : img0);
  const int rpi_log2 = IMGS > 2 ? 8 - (4 - p.lTWt - p.lTHt) : 30;
  const int wh = wave & 1;
  const int e_tl = tid >> 5, e_cp = tid & 31;
  float* park = smem;
#pragma unroll
  for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
    for (int j = 0; j < kNP; ++j)
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int r = rnd * 8 + r8;
        const int tl = (r & 3) + 4 * lh + 8 * ((r >> 2) & 1);
        smem[((wq + kPS * j) * 16 + tl) * kLdm + wh * 32 + li] = acc[j][r];
      }
    __syncthreads();
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
    __syncthreads();                           // every thread has read its products: the parked tile may overwrite them
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
      for (int dx = 0; dx < 4; ++dx)
        *reinterpret_cast<float2*>(park + (e_tl * 16 + dy * 4 + dx) * kLdt + 2 * e_cp) = y[dy][dx];
    __syncthreads();
    const int gn_entry = gn_base < 0 ? -1 : (IMGS == 1 ? gn_base + rnd : gn_base + rnd * (IMGS >> 1));
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
    const int gn_max = IMGS == 1 ? p.N * p.tiles_per_img * 2 : p.N;
    if (rnd == 0) ssde_store_tile<256, 64, kEpiThreads, 4, 0>(park, kLdt, n0, e, pixfn, gn_entry, rpi_log2, gn_max);
    else ssde_store_tile<256, 64, kEpiThreads, 8, 0>(park, kLdt, n0, e, pixfn, gn_entry, rpi_log2, gn_max);
    if (rnd == 0) { __syncthreads(); SSDE_GT(4); }
  }
  SSDE_GT(5);
}

int pow2_floor(int v) { int q = 1; while (q * 2 <= v) q *= 2; return q; }

// ---- the input-transform pass: V[pos][Q][t][4] = B^T pro(x) B, Q = Ctot / 4 channel quads, t = (img * tiles_h + ty) * tiles_w + tx.
// One thread = one 6x6 input tile of one channel quad: 36 float4 loads (the prologue -- GroupNorm, SiLU, dropout -- applied once
// per element and tile; out-of-image pixels are zeros of the ACTIVATED tensor), both 1-D passes in registers, 36 float4 stores.
// A wave = 16 consecutive tiles x 4 consecutive quads, lane = quad * 16 + tile: a store instruction writes four 256-byte runs,
// a load instruction reads 64 contiguous bytes (4 quads) of 16 pixels.  HBM-bound: x read ~2.25x through the tile overlap (L2),
// V written once.
struct XformVqParams {
  ssde_src src; float* v;
  int N, H, W, Ctot, T, tiles_h, tiles_w;
};

__device__ __forceinline__ void bt6q(const float4 (&d)[6], float4 (&o)[6]) {
#define SSDE_BT6_LANE(c)                                                                                             \
  {                                                                                                                  \
    const float t1 = d[4].c - 4.f * d[2].c, t2 = d[3].c - 4.f * d[1].c, t3 = d[4].c - d[2].c, t4 = d[3].c - d[1].c;   \
    o[0].c = 4.f * d[0].c - 5.f * d[2].c + d[4].c;                                                                    \
    o[1].c = t1 + t2; o[2].c = t1 - t2; o[3].c = t3 + 2.f * t4; o[4].c = t3 - 2.f * t4;                               \
    o[5].c = 4.f * d[1].c - 5.f * d[3].c + d[5].c;                                                                    \
  }
  SSDE_BT6_LANE(x) SSDE_BT6_LANE(y) SSDE_BT6_LANE(z) SSDE_BT6_LANE(w)
#undef SSDE_BT6_LANE
}

template <bool kGn>
__global__ __launch_bounds__(256, 2) void wino4_xform_vq_kernel(const XformVqParams p) {
  const ssde_src& s = p.src;
  SsdePro pro = ssde_pro_decode(s);
  pro.gn = kGn;
  const int Q = p.Ctot >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 16 + (lane & 15);
  const int q = (blockIdx.y * 4 + wave) * 4 + (lane >> 4);
  if (t >= p.T || q >= Q) return;
  const int c = q * 4;
  const int per_img = p.tiles_h * p.tiles_w;
  const int img = t / per_img, r = t - img * per_img;
  const int ty = r / p.tiles_w, tx = r - ty * p.tiles_w;
  const bool second = c >= s.c0;                      // (c0 % 4 == 0: a quad never straddles the sources)
  const float* base = second ? s.p1 + (c - s.c0) : s.p0 + c;
  const int C = second ? s.c1 : s.c0;
  float mu = 0.f, rs = 1.f;
  float4 ga = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kGn) {
    const int cpg = p.Ctot / s.gn_groups;             // (cpg % 4 == 0: a quad lies in one group)
    mu = s.gn_mean[img * s.gn_groups + c / cpg];
    rs = s.gn_rstd[img * s.gn_groups + c / cpg];
    ga = *reinterpret_cast<const float4*>(s.gn_gamma + c);
    be = *reinterpret_cast<const float4*>(s.gn_beta + c);
  }
  float4 v[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int iy = ty * 4 - 1 + a, ix = tx * 4 - 1 + b;
      const bool inb = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int pix = (img * p.H + (inb ? iy : 0)) * p.W + (inb ? ix : 0);
      v[a][b] = *reinterpret_cast<const float4*>(base + (size_t)pix * C);
    }
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int iy = ty * 4 - 1 + a, ix = tx * 4 - 1 + b;
      const bool inb = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int pix = (img * p.H + (inb ? iy : 0)) * p.W + (inb ? ix : 0);
      const float4 x = ssde_pro_apply(v[a][b], mu, rs, ga, be, (uint32_t)pix * (uint32_t)p.Ctot + (uint32_t)c, pro);
      v[a][b] = inb ? x : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  // columns (over the rows a of every column b), then rows
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    float4 d[6], o[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) d[a] = v[a][b];
    bt6q(d, o);
#pragma unroll
    for (int a = 0; a < 6; ++a) v[a][b] = o[a];
  }
  float* dst = p.v + ((size_t)q * p.T + t) * 4;
  const size_t plane = (size_t)Q * p.T * 4;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float4 o[6];
    bt6q(v[a], o);
#pragma unroll
    for (int b = 0; b < 6; ++b) *reinterpret_cast<float4*>(dst + (size_t)(a * 6 + b) * plane) = o[b];
  }
}

}  // namespace

int ssde_wino4_xform_vq_launch(const ssde_conv_args* a, void* stream) {
  SSDE_REQUIRE(a && a->main.p0 && a->wino_v, "conv(winograd 4x4, two kernels): the transformed-input buffer (ssde_conv_args.wino_v) is missing");
  const ssde_src& s = a->main;
  SSDE_REQUIRE(s.c0 > 0 && s.c0 % 4 == 0 && s.c1 % 4 == 0 && (s.c1 == 0 || s.p1), "conv(winograd 4x4, two kernels): channels must be multiples of 4");
  SSDE_REQUIRE(a->h_in % 4 == 0 && a->w_in % 4 == 0 && a->n > 0, "conv(winograd 4x4, two kernels): bad shape");
  const bool gn = s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU;
  if (gn) {
    SSDE_REQUIRE(s.gn_groups > 0 && (s.c0 + s.c1) % s.gn_groups == 0 && ((s.c0 + s.c1) / s.gn_groups) % 4 == 0,
                 "conv(winograd 4x4, two kernels): GroupNorm needs channels-per-group %% 4 == 0");
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "conv(winograd 4x4, two kernels): GroupNorm pointers missing");
  }
  SSDE_REQUIRE(s.drop_thresh == 0 || s.drop_seed, "conv(winograd 4x4, two kernels): dropout seed pointer missing");
  XformVqParams p{s, a->wino_v, a->n, a->h_in, a->w_in, s.c0 + s.c1, a->n * (a->h_in / 4) * (a->w_in / 4), a->h_in / 4, a->w_in / 4};
  SSDE_REQUIRE((unsigned long long)a->n * a->h_in * a->w_in * (unsigned)p.Ctot < (1ull << 32), "conv(winograd 4x4, two kernels): tensor too large");
  const dim3 grid(ssde_cdiv(p.T, 16), ssde_cdiv(p.Ctot >> 2, 16));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (gn) hipLaunchKernelGGL(wino4_xform_vq_kernel<true>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(wino4_xform_vq_kernel<false>, grid, dim3(256), 0, st, p);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}

// stream == (void*)1 with lds_out: plan-only query of the GroupNorm slices per image (conv_mfma.hip, ssde_conv_gn_slices)
int ssde_conv_wino4g_launch(const ssde_conv_args* a, void* stream, int* lds_out) {
  SSDE_REQUIRE(a && a->dst && a->main.p0 && a->w_main, "conv(winograd 4x4, two kernels): null args");
  SSDE_REQUIRE(a->ksize == 3 && a->stride == 1 && a->pad == 1, "conv(winograd 4x4, two kernels): needs 3x3, stride 1, pad 1");
  SSDE_REQUIRE(a->aux.p0 == nullptr, "conv(winograd 4x4, two kernels): fused 1x1 source not supported (issue it as a second conv)");
  SSDE_REQUIRE(a->h_in == a->h_out && a->w_in == a->w_out && a->h_out % 4 == 0 && a->w_out % 4 == 0 && a->h_out >= 8 && a->w_out >= 8,
               "conv(winograd 4x4, two kernels): same-size output, multiples of 4, at least 8x8 (got %dx%d)", a->h_out, a->w_out);
  const ssde_src& s = a->main;
  SSDE_REQUIRE(s.c0 > 0 && s.c0 % 4 == 0 && s.c1 % 4 == 0 && (s.c1 == 0 || s.p1), "conv(winograd 4x4, two kernels): channels must be multiples of 4");
  Wino4gParams p;
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
  p.n_tiles = ssde_cdiv(a->c_out, 64);
  p.bias = a->bias; p.chan_add = a->chan_add; p.chan_add_ld = a->chan_add_ld;
  p.resid = a->resid; p.resid_post = a->resid_post; p.scale = a->out_scale; p.dst = a->dst;
  p.gn_part = a->gn_part;
  p.tiles_h = a->h_out / 4; p.tiles_w = a->w_out / 4; p.T = a->n * p.tiles_h * p.tiles_w;
  const bool gn_ok = a->c_out % 4 == 0 && (imgs == 1 || (p.tiles_per_img == 1 && imgs <= 8));
  SSDE_REQUIRE(!a->gn_part || gn_ok, "conv(winograd 4x4, two kernels): GroupNorm partials not available for this tiling");
  if (lds_out && stream == reinterpret_cast<void*>(1)) {
    *lds_out = gn_ok ? (imgs == 1 ? 2 * p.tiles_per_img : 1) * (kEpiThreads / 64) : 0;
    return SSDE_OK;
  }
  int lds = (kVRing * kVFloats + 2 * kUFloats) * 4;
  const int lds_epi = kPos * 16 * kLdm * 4;
  if (lds < lds_epi) lds = lds_epi;
  SSDE_REQUIRE(lds <= 160 * 1024, "conv(winograd 4x4, two kernels): %d bytes of LDS", lds);
  if (lds_out) { *lds_out = lds; return SSDE_OK; }
  SSDE_REQUIRE(a->wino_v, "conv(winograd 4x4, two kernels): the transformed-input buffer (ssde_conv_args.wino_v) is missing");
  SSDE_REQUIRE(36ull * (unsigned long long)p.T * (unsigned)p.Ctot * 4ull < (1ull << 32),
               "conv(winograd 4x4, two kernels): a transformed input of 4 GB or more is not addressable by this kernel");
  const int wgs = ssde_cdiv(p.m_tiles, 8) * 8 * p.n_tiles;
  static std::atomic<bool> attr_set;
  if (!attr_set) {                              // once, before any stream capture
    SSDE_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4g_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024) == hipSuccess, "conv(winograd 4x4, two kernels): hipFuncSetAttribute failed");
    attr_set = true;
  }
  hipLaunchKernelGGL(conv_wino4g_kernel, dim3(wgs), dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
