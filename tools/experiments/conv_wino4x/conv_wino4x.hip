// 3x3 / stride 1 / pad 1 convolution by Winograd F(4x4, 3x3) on the BF16 matrix pipe: exact-fp32 products by a 3-way bf16 split
// of BOTH operands (SSDE_TILE_WINOGRAD4X; the engine lowers to it under SSDE_MATRIX=bf16x6, inference programs).
//
// Same algorithm, tiling, prologue and epilogue as conv_wino4.hip (read that file first): one workgroup = 32 tiles x 64 couts
// x 36 positions, 8 waves x 9 positions x (32 x 32) accumulators, K advances 4 input channels per stage.  What changes:
//   * a (position, tile) cell of V and a (position, cout) cell of U are 24 bytes -- [piece 0 | piece 1 | piece 2] x 4
//     channels bf16, the pieces taken by truncation (ssde_split3: they sum to the fp32 value exactly);
//   * K = 16 slots of v_mfma_f32_32x32x16_bf16 = 4 channels x 4, two MFMAs per position and stage (64 matrix cycles against
//     128 of the fp32 form), eight of the nine partial products:
//        lanes k = 0:  (a1 a0)·(b0 b0)      lanes k = 1:  (a0 a1)·(b1 b1)          -> a0b0 + a1b0 + a0b1 + a1b1
//        lanes k = 0:  (a0 a2)·(b2 b0)      lanes k = 1:  (a1 a2)·(b2 b1)          -> a0b2 + a2b0 + a1b2 + a2b1
//     (k = lane >> 5 supplies slots 8k .. 8k + 7.)  A lane loads the 6-register tuple (a_{1-k}, a_k, a_2): the first MFMA takes
//     registers 0-3, the second 2-5; likewise (b_2, b_k, b_k).  Dropped: a2·b2 <= 2^-32 |ab|.
//   * the weights of a stage are ONE wave-private image (55 KB; two do not fit): position j of the next stage is fetched by
//     LDS-DMA (768 bytes, lanes 0-47) into the cell range position j of this stage has just left, right after that
//     position's MFMAs were issued; every fragment read waits, with a counted vmcnt, for exactly its own piece (tables below);
//   * the input transform's second pass splits its fp32 results and writes the bf16 cells.  Both passes run in place in
//     V[next]: a position's 960-byte row is ten 96-byte chunks, chunk w = the 4 tiles (8 items) of wave w -- 64 bytes of fp32
//     intermediate, then 96 bytes of bf16 cells, both inside the wave's own chunk (a wave's LDS operations execute in order: all
//     its pass-2 reads precede its writes); chunks 8 and 9 take the padding lanes' writes.
// Matrix phase in isolation (tools/microbench/wino4_bf16x6_phase.hip): 1434 cycles per stage against 2347 of the fp32 form.
// LDS: V 2 x 33.75 KB + U 54 KB + raw 2 x 9.6 KB + GroupNorm tables = ~150 KB.  Not taken: split reductions, training.
#include "ssde_common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned ssde_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned ssde_u32x6 __attribute__((ext_vector_type(6)));

// 768-byte LDS-DMA piece on lanes 0-47 (M0 / scalar base as the previous SSDE_GLDS16_S of this wave left them)
#ifndef SSDE_GLDS16_S_SAME_BASE_LO48
#define SSDE_GLDS16_S_SAME_BASE_LO48(voff, sbase, lds_wave_base, imm)                                                    \
  do {                                                                                                                    \
    unsigned long long ssde_exec_save_;                                                                                   \
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_hi, 0xffff\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3\n\t"     \
                 "s_mov_b64 exec, %0"                                                                                     \
                 : "=&s"(ssde_exec_save_)                                                                                 \
                 : "v"(voff), "s"(sbase), "n"(imm)                                                                        \
                 :);                                                                                                      \
  } while (0)
#define SSDE_GLDS16_S_LO48(voff, sbase, lds_wave_base, imm)                                                              \
  do {                                                                                                                    \
    unsigned long long ssde_exec_save_;                                                                                   \
    asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 %0, exec\n\ts_mov_b32 exec_hi, 0xffff\n\t"                              \
                 "global_load_lds_dwordx4 %2, %3 offset:%4\n\ts_mov_b64 exec, %0"                                        \
                 : "=&s"(ssde_exec_save_)                                                                                 \
                 : "s"(__builtin_amdgcn_readfirstlane(                                                                    \
                       (int)(uintptr_t)(__attribute__((address_space(3))) void*)(lds_wave_base))),                        \
                   "v"(voff), "s"(sbase), "n"(imm)                                                                        \
                 :);                                                                                                      \
  } while (0)
#endif

// -DSSDE_W4X_TRACE (tools/wino4x_trace.py, a variant library only): s_memtime stamps of waves 0 and 7 of the first workgroup
#ifdef SSDE_W4X_TRACE
__device__ unsigned long long* g_w4x_trace;
extern "C" int ssde_debug_w4x_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_w4x_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -5;
}
#define SSDE_XT(slot)                                                                                     \
  do {                                                                                                    \
    if (tr_on) g_w4x_trace[tr_base + (slot)] = __builtin_amdgcn_s_memtime();                              \
  } while (0)
#else
#define SSDE_XT(slot) do { } while (0)
#endif

namespace {

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kNP = 9;                              // positions per wave
constexpr int kPS = 4;                              // wave (q, h) owns positions q + 4 j
constexpr int kPos = 36, kTiles = 32;
constexpr int kCell = 24;                           // bytes of a cell: 3 pieces x 4 channels bf16
constexpr int kChunk = 4 * kCell;                   // 96: the cells of one wave's 4 tiles
constexpr int kVPitch = 10 * kChunk + 8;            // 968 bytes per position: 8 wave chunks + 2 for the padding lanes, + 8 so that the six
                                                    // positions of a transform line are not 128-byte multiples apart (6 x 960 = 45 x 128:
                                                    // every lane group of a pass-2 write hit the same banks)
constexpr int kVBytes = kPos * kVPitch;             // one V stage
constexpr int kUPos = 32 * kCell;                   // 768: one position of a wave's weights
constexpr int kUWave = kNP * kUPos;                 // 6912
constexpr int kUBytes = kWaves * kUWave;            // 55296: one stage of weights (64 couts x 36 positions)
constexpr int kMaxRaw = 2;
constexpr int kLdm = 66, kLdt = 68;

struct Wino4xParams {
  ssde_src src;
  const char* wpk;         // [ceil(C/4)][n_tiles][8 waves][9][32 couts][24 bytes]
  int N, H, W, Cout;
  int lTWt, lTHt;
  int tiles_x, tiles_per_img, m_tiles, n_tiles;
  const float* bias; const float* chan_add; int chan_add_ld;
  const float* resid; int resid_post;
  float scale;
  float* dst;
  float* gn_part;
};

// B^T rows [4,0,-5,0,1,0] [0,-4,-4,1,1,0] [0,4,-4,-1,1,0] [0,-2,-1,2,1,0] [0,2,-1,-2,1,0] [0,4,0,-5,0,1], two channels at once
__device__ __forceinline__ void bt6x(const ssde_f32x2 (&d)[6], ssde_f32x2 (&o)[6]) {
  const ssde_f32x2 t1 = d[4] - 4.f * d[2], t2 = d[3] - 4.f * d[1], t3 = d[4] - d[2], t4 = d[3] - d[1];
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = t1 + t2;
  o[2] = t1 - t2;
  o[3] = t3 + 2.f * t4;
  o[4] = t3 - 2.f * t4;
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

// two fp32 values (a channel pair) -> three dwords, dword i = (piece i of x | piece i of y << 16)
__device__ __forceinline__ void split_pair(float x, float y, uint32_t (&q)[3]) {
  uint32_t ax = __builtin_bit_cast(uint32_t, x), ay = __builtin_bit_cast(uint32_t, y);
  q[0] = ssde_pack_hi16(ax, ay);
  float rx = x - __builtin_bit_cast(float, ax & 0xffff0000u), ry = y - __builtin_bit_cast(float, ay & 0xffff0000u);
  ax = __builtin_bit_cast(uint32_t, rx); ay = __builtin_bit_cast(uint32_t, ry);
  q[1] = ssde_pack_hi16(ax, ay);
  rx -= __builtin_bit_cast(float, ax & 0xffff0000u); ry -= __builtin_bit_cast(float, ay & 0xffff0000u);
  q[2] = ssde_pack_hi16(__builtin_bit_cast(uint32_t, rx), __builtin_bit_cast(uint32_t, ry));
}

template <bool kGn>
__global__ __launch_bounds__(kThreads, kWaves / 4) void conv_wino4x_kernel(const Wino4xParams p) {
  SSDE_LDS(smem);
  char* lds = reinterpret_cast<char*>(smem);
  char* Vb = lds;                              // [2][kVBytes]
  char* Ub = lds + 2 * kVBytes;                // [8 waves][9 positions][32 couts][24]: a wave reads only its own region
  float* rawb = reinterpret_cast<float*>(Ub + kUBytes);     // [2][2 pairs][halo_px][2], then the GroupNorm tables
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;

  const int bid = blockIdx.x;
  const int xcd = bid & 7, l = bid >> 3;
  const int nt = l % p.n_tiles;
  const int mt = (l / p.n_tiles) * 8 + xcd;
#ifdef SSDE_W4X_TRACE
  const bool tr_on = lane == 0 && (wave == 0 || wave == 7) && bid == 0 && g_w4x_trace != nullptr;
  const int tr_base = (wave == 0 ? 0 : 1) * 128;
#endif
  SSDE_XT(0);
  if (mt >= p.m_tiles) return;

  const int TWt = 1 << p.lTWt, THt = 1 << p.lTHt;
  const int IMGS = kTiles >> (p.lTWt + p.lTHt);
  const int HWd = 4 * TWt + 2, HH = 4 * THt + 2;
  const int halo_px = IMGS * HH * HWd;
  const int raw_plane = 2 * halo_px;           // floats between the two channel-pair planes of a raw buffer
  const int raw_stride = 2 * raw_plane;        // floats per raw buffer
  const int img0 = (mt / p.tiles_per_img) * IMGS;
  const int trem = mt % p.tiles_per_img;
  const int ty = trem / p.tiles_x, tx = trem % p.tiles_x;
  const int n0 = nt * 64;

  const ssde_src& s = p.src;
  const int Ctot = s.c0 + s.c1;
  const int nst = (Ctot + 3) >> 2;
  SsdePro pro = ssde_pro_decode(s);
  pro.gn = kGn;
  const int cpg = kGn ? Ctot / s.gn_groups : 1;
  const float inv_cpg = 1.0f / (float)cpg;

  // ---- raw staging plan (conv_wino4.hip): item = halo pixel, its 4 channels of the stage one float4 ----
  int goff[kMaxRaw], gil[kMaxRaw];
  uint32_t voff0[kMaxRaw], voff1[kMaxRaw];
#pragma unroll
  for (int it = 0; it < kMaxRaw; ++it) {
    const int q = tid + it * kThreads;
    goff[it] = -2; gil[it] = 0;
    if (q < halo_px) {
      const int il = q / (HH * HWd);
      const int rem = q - il * (HH * HWd);
      const int hy = rem / HWd, hx = rem - hy * HWd;
      const int iy = ty * 4 * THt - 1 + hy, ix = tx * 4 * TWt - 1 + hx;
      const int img = img0 + il;
      const bool inb = img < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      goff[it] = inb ? (img * p.H + iy) * p.W + ix : -1;
      gil[it] = inb ? il * (kGn ? s.gn_groups : 0) : 0;
    }
    const uint32_t px = (uint32_t)(goff[it] >= 0 ? goff[it] : 0);
    voff0[it] = px * (uint32_t)s.c0 * 4u;
    voff1[it] = px * (uint32_t)s.c1 * 4u;
  }
  // ---- transform plan: lane = line * 8 + item, a wave owns the 8 items (4 tiles x 2 channel pairs) of its chunk with all their
  // 6 lines; lanes 48-63 repeat lines 0 and 1 into chunks 8 and 9 (never read by a fragment) ----
  const int t_line6 = lane >> 3;
  const bool t_real = t_line6 < 6;
  const int t_line = t_real ? t_line6 : t_line6 - 6;
  const int t_tile = wave * 4 + ((lane & 7) >> 1), t_pair = lane & 1;
  int t_base;
  {
    const int il = t_tile >> (p.lTWt + p.lTHt);
    const int tr = (t_tile >> p.lTWt) & (THt - 1), tc = t_tile & (TWt - 1);
    t_base = (il * HH + 4 * tr) * HWd + 4 * tc;
  }
  const int t_chunk = (t_real ? wave : 8 + (t_line6 - 6)) * kChunk, t_within = (lane & 7) >> 1;
  const int t_ioff = t_chunk + t_within * 16 + t_pair * 8;          // fp32 intermediate (a float2) inside a position's row
  const int t_foff = t_chunk + t_within * kCell + t_pair * 4;       // the item's bf16 pair inside its cell (+ 8 per piece)
  const int t_rawoff = t_pair * raw_plane + (t_base + t_line) * 2;

  float* gn_tab = rawb + 2 * raw_stride;       // [IMGS][groups][2]
  float* gb_tab = gn_tab + 2 * IMGS * (kGn ? s.gn_groups : 0);   // [2][Ctot]
  ssde_f32x4 rv[kMaxRaw];
  auto load_piece = [&](int st, int k) __attribute__((always_inline)) {
    const int c_base = st * 4;
    const bool second = c_base >= s.c0;
    const float* sb = second ? s.p1 + (c_base - s.c0) : s.p0 + c_base;
    const uint32_t vo = second ? voff1[k] : voff0[k];
    SSDE_GLOAD16(rv[k], vo, sb);
  };
  struct GnRegs { float4 gam, bet; float2 mr[kMaxRaw]; };
  auto gn_fetch = [&](int st) __attribute__((always_inline)) {
    GnRegs r;
    r.gam = make_float4(1.f, 1.f, 1.f, 1.f); r.bet = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < kMaxRaw; ++it) r.mr[it] = make_float2(0.f, 1.f);
    if (kGn) {
      const int c_cur = st * 4;
      r.gam = *reinterpret_cast<const float4*>(gb_tab + c_cur);
      r.bet = *reinterpret_cast<const float4*>(gb_tab + Ctot + c_cur);
      const int g = (int)(((float)c_cur + 0.5f) * inv_cpg);
#pragma unroll
      for (int it = 0; it < kMaxRaw; ++it) r.mr[it] = *reinterpret_cast<const float2*>(gn_tab + 2 * (gil[it] + g));
    }
    return r;
  };
  auto store_raw_with = [&](float* rw, int st, const GnRegs& r) __attribute__((always_inline)) {
    const int c_cur = st * 4;
#pragma unroll
    for (int it = 0; it < kMaxRaw; ++it) {
      if (goff[it] == -2) continue;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (goff[it] >= 0)
        v = ssde_pro_apply(make_float4(rv[it].x, rv[it].y, rv[it].z, rv[it].w), r.mr[it].x, r.mr[it].y, r.gam, r.bet,
                           (uint32_t)goff[it] * (uint32_t)Ctot + (uint32_t)c_cur, pro);
      const int q = tid + it * kThreads;
      *reinterpret_cast<float2*>(rw + q * 2) = make_float2(v.x, v.y);
      *reinterpret_cast<float2*>(rw + raw_plane + q * 2) = make_float2(v.z, v.w);
    }
  };
  auto store_raw = [&](float* rw, int st) __attribute__((always_inline)) { store_raw_with(rw, st, gn_fetch(st)); };
  // pass 1 (lane = column x): raw -> fp32 intermediate at positions a * 6 + x; pass 2 (lane = row y): positions y * 6 + b,
  // read as fp32, written as bf16 cells
  auto pass1_read = [&](const float* rw, ssde_f32x2 (&d)[6]) __attribute__((always_inline)) {
    const float* rp = rw + t_rawoff;
#pragma unroll
    for (int a = 0; a < 6; ++a) { const float2 q = *reinterpret_cast<const float2*>(rp + a * HWd * 2); d[a].x = q.x; d[a].y = q.y; }
  };
  // (one opaque LDS base per pass: the positions become immediate offsets of the ds instructions)
  typedef __attribute__((address_space(3))) char lds_char;
  auto pass1_write = [&](char* Vn, const ssde_f32x2 (&o)[6]) __attribute__((always_inline)) {
    lds_char* w = (lds_char*)(Vn + t_line * kVPitch + t_ioff);
    SSDE_OPAQUE_VGPR(w);
#pragma unroll
    for (int a = 0; a < 6; ++a) *(ssde_lds_float2*)(w + a * 6 * kVPitch) = o[a];
  };
  auto pass2_read = [&](const char* Vn, ssde_f32x2 (&d)[6]) __attribute__((always_inline)) {
    lds_char* r = (lds_char*)(const_cast<char*>(Vn) + t_line * 6 * kVPitch + t_ioff);
    SSDE_OPAQUE_VGPR(r);
#pragma unroll
    for (int b = 0; b < 6; ++b) d[b] = *(ssde_lds_float2*)(r + b * kVPitch);
  };
  auto pass2_write = [&](char* Vn, const ssde_f32x2 (&o)[6]) __attribute__((always_inline)) {
    lds_char* w = (lds_char*)(Vn + t_line * 6 * kVPitch + t_foff);
    SSDE_OPAQUE_VGPR(w);
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      uint32_t q[3];
      split_pair(o[b].x, o[b].y, q);
      *(lds_u32*)(w + b * kVPitch) = q[0];
      *(lds_u32*)(w + b * kVPitch + 8) = q[1];
      *(lds_u32*)(w + b * kVPitch + 16) = q[2];
    }
  };
  // Weights: position j of stage st = 768 bytes at w_base(st) + wave * 6912 + j * 768, moved by lanes 0-47 into the same
  // offset of the wave's LDS region.  One scalar base and one M0 per stage (aimed at position 4), immediates -3072 .. +3072.
  const uint32_t w_voff = (uint32_t)(wave * kUWave + 4 * kUPos + lane * 16);          // (lanes 48-63 sit the copy out)
  auto w_base = [&](int st) { return p.wpk + ((size_t)st * p.n_tiles + nt) * kUBytes; };
  char* w_ldst = Ub + wave * kUWave + 4 * kUPos;

  const int wq = wave >> 1;
  f32x16 acc[kNP];
#pragma unroll
  for (int j = 0; j < kNP; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- pipeline fill: leaves V[0] transformed, ALL weight pieces of stage 0 landed, raw[1] = the activated halo of stage 1,
  // rv = the halo of stage 2 (landed) ----
  {
    const char* wb = w_base(0);
    SSDE_GLDS16_S_LO48(w_voff, wb, w_ldst, -3072);
    SSDE_GLDS16_S_SAME_BASE_LO48(w_voff, wb, w_ldst, -2304);
    SSDE_GLDS16_S_SAME_BASE_LO48(w_voff, wb, w_ldst, -1536);
    SSDE_GLDS16_S_SAME_BASE_LO48(w_voff, wb, w_ldst, -768);
    SSDE_GLDS16_S_SAME_BASE_LO48(w_voff, wb, w_ldst, 0);
    SSDE_GLDS16_S_SAME_BASE_LO48(w_voff, wb, w_ldst, 768);
    SSDE_GLDS16_S_SAME_BASE_LO48(w_voff, wb, w_ldst, 1536);
    SSDE_GLDS16_S_SAME_BASE_LO48(w_voff, wb, w_ldst, 2304);
    SSDE_GLDS16_S_SAME_BASE_LO48(w_voff, wb, w_ldst, 3072);
  }
#pragma unroll
  for (int k = 0; k < kMaxRaw; ++k) load_piece(0, k);
  if (kGn) {
    for (int q = tid; q < IMGS * s.gn_groups; q += kThreads) {
      const int il = q / s.gn_groups, img = img0 + il < p.N ? img0 + il : 0;
      const int gi = img * s.gn_groups + (q - il * s.gn_groups);
      *reinterpret_cast<float2*>(gn_tab + 2 * q) = make_float2(s.gn_mean[gi], s.gn_rstd[gi]);
    }
    for (int q = tid; q < Ctot; q += kThreads) { gb_tab[q] = s.gn_gamma[q]; gb_tab[Ctot + q] = s.gn_beta[q]; }
  }
  __syncthreads();                             // publishes the GroupNorm tables
  SSDE_WAIT_VMCNT_FOR(0, rv[0], rv[1]);
  store_raw(rawb, 0);
  if (nst > 1) {
#pragma unroll
    for (int k = 0; k < kMaxRaw; ++k) load_piece(1, k);
  }
  SSDE_LDS_BARRIER();
  {
    ssde_f32x2 d[6], o[6];
    pass1_read(rawb, d);
    bt6x(d, o);
    pass1_write(Vb, o);
    SSDE_LDS_BARRIER();                        // (lanes run in lockstep: only the test emulator's fibers need the rendez-vous)
    pass2_read(Vb, d);
    bt6x(d, o);
    SSDE_LDS_BARRIER();                        // (emulator: every lane has read before any lane overwrites)
    pass2_write(Vb, o);
  }
  SSDE_WAIT_VMCNT_FOR(0, rv[0], rv[1]);
  if (nst > 1) store_raw(rawb + raw_stride, 1);
  if (nst > 2) {
#pragma unroll
    for (int k = 0; k < kMaxRaw; ++k) load_piece(2, k);
  }
  SSDE_WAIT_VMCNT_FOR(0, rv[0], rv[1]);        // (also: every weight piece of stage 0 has landed)
  SSDE_LDS_BARRIER();
  SSDE_XT(2);

  // ---- one stage.  FULL (the main loop): stages st + 1, st + 2, st + 3 exist -- the weights of st + 1 are fetched position by
  // position, its input is transformed, the halo of st + 2 is activated at the head, the halo of st + 3 is fetched.  The last
  // three stages take the same body with the flags off and vmcnt(0) at their head (nothing of theirs is counted).
  //
  // VMEM issue order of a FULL stage (D'j = weight piece j of the NEXT stage, H = halo load):
  //      slot 0: D'0 | 1: D'1 H0 | 2: D'2 | 3: D'3 H1 | 4: D'4 | 5: D'5 | 6: D'6 | 7: D'7 | 8: D'8          (11 per stage)
  // The fragment read of position j (issued at the start of slot j - 2; positions 0, 1 at the head) needs piece j of THIS
  // stage = D'j of the previous one: younger operations = those behind D'j in the previous stage + those of this stage so far:
  //      head (positions 0, 1 and the halo registers)   vmcnt(5)      [behind H1 of the previous stage: D'4 .. D'8]
  //      slot 1 (position 3)  6 + 1 = 7      slot 2 (4)  4 + 3 = 7      slot 3 (5)  3 + 4 = 7
  //      slot 4 (position 6)  2 + 6 = 8      slot 5 (7)  1 + 7 = 8      slot 6 (8)  0 + 8 = 8        (position 2: 7 + 0, covered)
  auto stage = [&](auto FULLT, auto H1, auto H2, const int st) __attribute__((always_inline)) {
    constexpr bool full = decltype(FULLT)::value, has1 = decltype(H1)::value, has2 = decltype(H2)::value;
    const int cur = st & 1, nxt = cur ^ 1;
    const char* Vc = Vb + cur * kVBytes;
    char* Vn = Vb + nxt * kVBytes;
    ssde_f32x2 td[6], to[6];
    const char* wb = w_base(has1 ? st + 1 : st);
    // fragment bases: (a_{1-k}, a_k, a_2) of tile li at position wq + 4 j; (b_2, b_k, b_k) of cout li at the wave's position j
    const char* va_p = Vc + wq * kVPitch + li * kCell + 8 * (1 - lh);
    const char* va_q = Vc + wq * kVPitch + li * kCell + 8 * lh;
    const char* va_2 = Vc + wq * kVPitch + li * kCell + 16;
    const char* ub_k = Ub + wave * kUWave + li * kCell + 8 * lh;
    const char* ub_2 = Ub + wave * kUWave + li * kCell + 16;
    // a position's operands are two 6-register tuples: the first MFMA takes registers 0-3, the second 2-5 (no copies)
    ssde_u32x6 fa[3], fb[3];                       // [slot of the read-ahead ring]
    typedef __attribute__((address_space(3))) const ssde_u32x2 lds_u32x2;
    typedef __attribute__((address_space(3))) const char lds_cchar;
    lds_cchar* pa_p = (lds_cchar*)va_p; lds_cchar* pa_q = (lds_cchar*)va_q; lds_cchar* pa_2 = (lds_cchar*)va_2;
    lds_cchar* pb_k = (lds_cchar*)ub_k; lds_cchar* pb_2 = (lds_cchar*)ub_2;
    SSDE_OPAQUE_VGPR(pa_p); SSDE_OPAQUE_VGPR(pa_q); SSDE_OPAQUE_VGPR(pa_2); SSDE_OPAQUE_VGPR(pb_k); SSDE_OPAQUE_VGPR(pb_2);
    auto frag = [&](int j) __attribute__((always_inline)) {
      const int r = j % 3;
      const ssde_u32x2 x0 = *(lds_u32x2*)(pa_p + kPS * j * kVPitch), x1 = *(lds_u32x2*)(pa_q + kPS * j * kVPitch);
      const ssde_u32x2 x2 = *(lds_u32x2*)(pa_2 + kPS * j * kVPitch);
      const ssde_u32x2 y0 = *(lds_u32x2*)(pb_2 + j * kUPos), y1 = *(lds_u32x2*)(pb_k + j * kUPos), y2 = *(lds_u32x2*)(pb_k + j * kUPos);
      fa[r] = (ssde_u32x6){x0[0], x0[1], x1[0], x1[1], x2[0], x2[1]};
      fb[r] = (ssde_u32x6){y0[0], y0[1], y1[0], y1[1], y2[0], y2[1]};
    };
    auto mfma = [&](int j) __attribute__((always_inline)) {
      const int r = j % 3;
      const ssde_u32x4 A1 = __builtin_shufflevector(fa[r], fa[r], 0, 1, 2, 3), A2 = __builtin_shufflevector(fa[r], fa[r], 2, 3, 4, 5);
      const ssde_u32x4 B2 = __builtin_shufflevector(fb[r], fb[r], 0, 1, 2, 3), B1 = __builtin_shufflevector(fb[r], fb[r], 2, 3, 4, 5);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssde_bf16x8, A1), __builtin_bit_cast(ssde_bf16x8, B1), acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssde_bf16x8, A2), __builtin_bit_cast(ssde_bf16x8, B2), acc[j], 0, 0, 0);
    };
    // ---- head ----
    GnRegs gnr;
    if (has2) gnr = gn_fetch(st + 2);
    if (full) SSDE_WAIT_VMCNT_FOR(5, rv[0], rv[1]); else SSDE_WAIT_VMCNT_FOR(0, rv[0], rv[1]);
    frag(0);
    frag(1);
    if (has1) pass1_read(rawb + nxt * raw_stride, td);
    if (has2) store_raw_with(rawb + cur * raw_stride, st + 2, gnr);
    if (st >= 4 && st < 12) SSDE_XT(8 + (st - 4) * 10 + 1);
    __builtin_amdgcn_sched_barrier(0);
#define SSDE_W4X_SLOT(J, WAITN)                                                                                           \
    do {                                                                                                                  \
      if ((J) + 2 < kNP) {                                                                                                \
        if (full && (WAITN) >= 0) SSDE_WAIT_VMCNT_FENCE((WAITN) < 0 ? 0 : (WAITN));                                       \
        frag((J) + 2);                                                                                                    \
      }                                                                                                                   \
      mfma(J);                                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                                  \
      if (has1) {                                                                                                         \
        if ((J) == 0) SSDE_GLDS16_S_LO48(w_voff, wb, w_ldst, -3072);                                                      \
        else SSDE_GLDS16_S_SAME_BASE_LO48(w_voff, wb, w_ldst, ((J) - 4) * 768);                                           \
      }                                                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                                  \
    } while (0)
    SSDE_W4X_SLOT(0, -1);
    SSDE_W4X_SLOT(1, 7);
    if (full) load_piece(st + 3, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (st >= 4 && st < 12) SSDE_XT(8 + (st - 4) * 10 + 2);
    if (has1) {
      bt6x(td, to);
      pass1_write(Vn, to);
    }
    if (st >= 4 && st < 12) SSDE_XT(8 + (st - 4) * 10 + 3);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4X_SLOT(2, 7);
    SSDE_W4X_SLOT(3, 7);
    if (full) load_piece(st + 3, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (st >= 4 && st < 12) SSDE_XT(8 + (st - 4) * 10 + 4);
    if (has1) {
      SSDE_WAVE_SYNC();                          // (emulator: the wave's pass-1 writes before its pass-2 reads)
      pass2_read(Vn, td);
    }
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4X_SLOT(4, 8);
    SSDE_W4X_SLOT(5, 8);
    if (st >= 4 && st < 12) SSDE_XT(8 + (st - 4) * 10 + 5);
    if (has1) {
      bt6x(td, to);
      SSDE_WAVE_SYNC();                          // (emulator: every lane of the wave has read its row before any cell is written)
      pass2_write(Vn, to);
    }
    if (st >= 4 && st < 12) SSDE_XT(8 + (st - 4) * 10 + 6);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4X_SLOT(6, 8);
    SSDE_W4X_SLOT(7, -1);
    SSDE_W4X_SLOT(8, -1);
#undef SSDE_W4X_SLOT
    if (st >= 4 && st < 12) SSDE_XT(8 + (st - 4) * 10 + 7);
    SSDE_LDS_BARRIER();
    if (st >= 4 && st < 12) SSDE_XT(8 + (st - 4) * 10 + 8);
  };
  {
    using T = std::true_type; using F = std::false_type;
    int st = 0;
    for (; st + 3 < nst; ++st) stage(T{}, T{}, T{}, st);
    if (st + 2 < nst) { stage(F{}, T{}, T{}, st); ++st; }
    if (st + 1 < nst) { stage(F{}, T{}, F{}, st); ++st; }
    stage(F{}, F{}, F{}, st);
  }

  SSDE_XT(3);
  // ---- epilogue (conv_wino4.hip): products -> LDS M[pos][16 tiles][64 couts], A^T M A per (tile, cout pair), parked tile,
  // shared coalesced store; 16 tiles (accumulator rows r < 8, then r >= 8) at a time ----
  __syncthreads();
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
    __syncthreads();
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
    if (rnd == 0) ssde_store_tile<256, 64, kThreads, 4, 0>(park, kLdt, n0, e, pixfn, gn_entry, rpi_log2, gn_max);
    else ssde_store_tile<256, 64, kThreads, 8, 0>(park, kLdt, n0, e, pixfn, gn_entry, rpi_log2, gn_max);
    if (rnd == 0) { __syncthreads(); SSDE_XT(4); }
  }
  SSDE_XT(5);
}

int pow2_floor(int v) { int q = 1; while (q * 2 <= v) q *= 2; return q; }

}  // namespace

// stream == (void*)1 with lds_out: plan-only query of the GroupNorm slices per image (as ssde_conv_wino4_launch)
int ssde_conv_wino4x_launch(const ssde_conv_args* a, void* stream, int* lds_out) {
  SSDE_REQUIRE(a && a->dst && a->main.p0 && a->w_main, "conv(winograd 4x4, bf16 split): null args");
  SSDE_REQUIRE(a->ksize == 3 && a->stride == 1 && a->pad == 1, "conv(winograd 4x4, bf16 split): needs 3x3, stride 1, pad 1");
  SSDE_REQUIRE(a->aux.p0 == nullptr, "conv(winograd 4x4, bf16 split): fused 1x1 source not supported (issue it as a second conv)");
  SSDE_REQUIRE(a->wino_v == nullptr, "conv(winograd 4x4, bf16 split): no transformed-input by-product (inference programs only)");
  SSDE_REQUIRE(a->h_in == a->h_out && a->w_in == a->w_out && a->h_out % 4 == 0 && a->w_out % 4 == 0 && a->h_out >= 8 && a->w_out >= 8,
               "conv(winograd 4x4, bf16 split): same-size output, multiples of 4, at least 8x8 (got %dx%d)", a->h_out, a->w_out);
  const ssde_src& s = a->main;
  SSDE_REQUIRE(s.c0 > 0 && s.c0 % 4 == 0 && s.c1 % 4 == 0 && (s.c1 == 0 || s.p1), "conv(winograd 4x4, bf16 split): channels must be multiples of 4");
  const bool gn = s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU;
  if (gn) {
    SSDE_REQUIRE(s.gn_groups > 0 && (s.c0 + s.c1) % s.gn_groups == 0 && ((s.c0 + s.c1) / s.gn_groups) % 4 == 0,
                 "conv(winograd 4x4, bf16 split): GroupNorm needs channels-per-group %% 4 == 0");
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "conv(winograd 4x4, bf16 split): GroupNorm pointers missing");
  }
  SSDE_REQUIRE(s.drop_thresh == 0 || s.drop_seed, "conv(winograd 4x4, bf16 split): dropout seed pointer missing");
  SSDE_REQUIRE((unsigned long long)a->n * a->h_in * a->w_in * (unsigned)(s.c0 > s.c1 ? s.c0 : s.c1) * 4ull < (1ull << 32),
               "conv(winograd 4x4, bf16 split): a source of 4 GB or more is not addressable by this kernel");
  Wino4xParams p;
  p.src = s; p.wpk = reinterpret_cast<const char*>(a->w_main);
  p.N = a->n; p.H = a->h_out; p.W = a->w_out; p.Cout = a->c_out;
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
  const bool gn_ok = a->c_out % 4 == 0 && (imgs == 1 || (p.tiles_per_img == 1 && imgs <= 8));
  SSDE_REQUIRE(!a->gn_part || gn_ok, "conv(winograd 4x4, bf16 split): GroupNorm partials not available for this tiling");
  if (lds_out && stream == reinterpret_cast<void*>(1)) {
    *lds_out = gn_ok ? (imgs == 1 ? 2 * p.tiles_per_img : 1) * (kThreads / 64) : 0;
    return SSDE_OK;
  }
  const int halo_px = imgs * (4 * tht + 2) * (4 * twt + 2);
  SSDE_REQUIRE(halo_px <= kMaxRaw * kThreads, "conv(winograd 4x4, bf16 split): halo of %d pixels exceeds the staging plan", halo_px);
  int lds = 2 * kVBytes + kUBytes + 2 * 2 * 2 * halo_px * 4;
  if (gn) lds += (2 * imgs * s.gn_groups + 2 * (s.c0 + s.c1)) * 4;
  const int lds_epi = kPos * 16 * kLdm * 4;
  if (lds < lds_epi) lds = lds_epi;
  SSDE_REQUIRE(lds <= 160 * 1024, "conv(winograd 4x4, bf16 split): %d bytes of LDS", lds);
  if (lds_out) { *lds_out = lds; return SSDE_OK; }
  const dim3 grid(ssde_cdiv(p.m_tiles, 8) * 8 * p.n_tiles);
  static std::atomic<bool> set[2];
  auto go = [&](auto kfn, std::atomic<bool>& attr_set) {
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return false;
      attr_set = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
    return true;
  };
  const bool ok = gn ? go(conv_wino4x_kernel<true>, set[1]) : go(conv_wino4x_kernel<false>, set[0]);
  SSDE_REQUIRE(ok, "conv(winograd 4x4, bf16 split): hipFuncSetAttribute failed");
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
