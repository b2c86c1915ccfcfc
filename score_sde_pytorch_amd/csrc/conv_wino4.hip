// 3x3 / stride 1 / pad 1 convolution by Winograd F(4x4, 3x3) on the exact-fp32 matrix pipe.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A     with 6x6 transformed tiles, 4x4 outputs per tile (Lavin & Gray,
//                                                    interpolation points 0, +-1, +-2)
// Per 4x4 output tile and (ci, co) pair the 144 multiply-adds of the direct form become 36 (F(2x2,3x3), conv_wino.hip:
// 64): the matrix pipe does 4x less work than the direct kernel and 1.78x less than F(2x2,3x3).  The price is
// rounding: the transforms multiply by up to 8 and the products are ~4-5x less accurate than F(2x2,3x3) in fp32.
// tools/experiments/wino43_error_budget.py runs the whole CIFAR-10 NCSN++ with every 3x3 layer in this form against
// an fp64 run: 2.5e-6 .. 1.3e-5 relative L2 error of the score (direct fp32: 4e-7 .. 1.2e-5), against the 1e-4 the
// parity tests allow.
//
// One workgroup (8 waves, two per SIMD) owns 32 tiles (a 32x16 output patch, or whole small images) x 64 output
// channels x all 36 transform positions; K advances 4 input channels per stage:
//   raw halo (+ fused GroupNorm / SiLU / dropout prologue)      -> LDS raw[pair][pixel][2]
//   input transform B^T d B in two 1-D passes, in place         -> LDS V[pos][tile][4]      (pitch 144 floats)
//   host-transformed weights G g G^T, packed as the LDS image   -> LDS U[pos][cout][4]      (36 KB, by LDS-DMA)
//   wave (q, h) = 9 positions {q, q+4, ..} x (32 tiles x 32 couts) on v_mfma_f32_32x32x2_f32: a lane's ds_read_b64 is
//   the channel pair (2k, 2k+1), k = lane >> 5 = the MFMA k index, one read of each operand feeds two MFMAs; all 64
//   lanes of a fragment read are 512 contiguous bytes (no bank conflicts)
// V and raw are double buffered, ONE LDS-only barrier per stage: a wave transforms 8 of the 64 (tile, channel pair) items
// with all their 6 lines, so the second 1-D pass reads what the same wave wrote in the first.  Every wave runs the same
// program; its halo loads, its weight DMA pieces and the LDS round trips of the two transform passes ride between its 18
// MFMAs (an MFMA only occupies the matrix pipe; the wave keeps issuing).
// The 36 positions of an output live in 4 waves, so the workgroup exchanges the products through LDS -- 16 tiles (half of
// every wave's accumulators) at a time, M[pos][16 tiles][64 couts] = 150 KB -- every thread applies A^T M A to one
// (tile, cout pair), and the result goes through the shared coalesced epilogue (ssde_store_tile: bias, temb addend,
// residual, scale, GroupNorm partials).
//
// Round 3 (profiles/r3_wino4_*.txt; every figure a same-box A/B against the round-2 kernel built as a variant library):
//   * U is PRIVATE per wave: the packed image of a stage is ordered [wave][position j][32 couts][4], wave (q, h) moves its
//     own 4.5 KB by LDS-DMA and is their only reader.  The stage barrier therefore carries no vmcnt(0) any more (it cost
//     400-570 cycles per stage: the last piece landed ~1500 cycles after its issue); a wave counts its own pieces with
//     vmcnt right before the fragment read that needs them.  The halo loads are issued from inline asm as well, so that
//     hipcc's waitcnt insertion (which cannot see the asm DMA pieces behind them) does not turn every use into vmcnt(0).
//   * The halo prologue (GroupNorm, SiLU, dropout -> raw) moved from the END of a stage to the HEAD of the next one (one
//     stage more of load latency covered, nothing between the last MFMA and the barrier), with the GroupNorm table reads
//     ahead of the head's burst of fragment / transform reads.
//   * The stage body is branch-free: flags are template parameters, the last three stages are peeled, lanes 48-63 of the
//     transforms work on padding columns.  hipcc's waitcnt counts are exact only in straight-line code (with the exec-masked
//     transform blocks every MFMA waited for the transform's LDS writes as well).
//   * One vector-memory instruction per position slot (as a burst in slots 0-1 they queued behind each other: 1500 cycles
//     for two positions); addressing by scalar base + 32-bit lane offset (no vector address arithmetic per stage).
//   Together +7 .. +10 % per layer (241-315 TF/s direct-equivalent with GroupNorm + SiLU, 240-333 without).
//   Measured and NOT adopted: the transform as scalar v_fma_f32 instead of packed v_pk_* (neutral, -DSSDE_W4_SCALAR_BT=1);
//   the two waves of a SIMD running their prologue at opposite ends of the stage (neutral: a VALU instruction of one wave
//   waits out the 64-cycle MFMA the other has in the pipe, its head took 2100 cycles instead of 1500); the MFMAs of the
//   last three positions issued after the barrier inside the next head, fragments carried in registers (2-3 % slower: the
//   head grew by exactly the matrix cycles it gained -- LDS burst latency, VALU and fp32 MFMA time ADD on a SIMD).
//   Where a stage's ~4400 cycles go (s_memtime, tools/wino4_trace.py): 2304 are MFMAs of the SIMD's two waves, ~950 VALU
//   (prologue 2 x 130-260, transforms 2 x 120, bookkeeping), the rest LDS / VMEM issue latency that nothing overlaps
//   because all eight waves are in the same phase (one barrier per stage, double buffers: no wave may run ahead).
// Matrix-pipe occupancy, two ways.  Cycle trace of an unprofiled launch (tools/wino4_trace.py, profiles/r3_wino4_trace_e_*):
// matrix cycles / workgroup cycles = 0.40 for 128 -> 128 channels (73.7 k of 183 k: loop 146.5 k, fill 9.6 k, epilogue 27 k),
// 0.46 for 256 -> 256, 0.49 for 512 -> 256; inside the main loop 0.50-0.52.  PMC (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE
// over all launches of the bench command): 0.40 in round 2 and in round 3 -- under rocprofv3 --pmc the kernel runs 26 %
// slower than unprofiled (408 vs 323 us per launch in one process), so the ratio understates the unprofiled kernel and moved
// less than the +7-10 % the same-box A/B shows.  The shader clock inside the kernel (s_memtime / s_memrealtime): 2.09-2.17 GHz.
#include "ssde_common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// -DSSDE_W4_TRACE (tools/wino4_trace.py, a variant library only): s_memtime stamps of waves 0 and 7 of the first workgroup
#ifdef SSDE_W4_TRACE
__device__ unsigned long long* g_w4_trace;
extern "C" int ssde_debug_w4_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_w4_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -5;
}
#define SSDE_TR(slot)                                                                                     \
  do {                                                                                                    \
    if (tr_on) g_w4_trace[tr_base + (slot)] = __builtin_amdgcn_s_memtime();                               \
  } while (0)
// the same with the constant 100 MHz counter: (s_memtime delta) / (s_memrealtime delta) x 100 MHz = the shader clock the
// kernel actually ran at
#define SSDE_TRR(slot)                                                                                    \
  do {                                                                                                    \
    if (tr_on) g_w4_trace[tr_base + (slot)] = __builtin_amdgcn_s_memrealtime();                           \
  } while (0)
#else
#define SSDE_TR(slot) do { } while (0)
#define SSDE_TRR(slot) do { } while (0)
#endif


// fragment reads run this many positions (2 MFMAs = 128 matrix cycles each) ahead of their MFMAs
#ifndef SSDE_W4_PF
#define SSDE_W4_PF 3
#endif
// 1: the short VALU / LDS bursts of the transforms and of the halo staging run at s_setprio 2 (as conv_wino.hip's staging)
#ifndef SSDE_W4_PRIO
#define SSDE_W4_PRIO 1
#endif
// 1: the two waves of a SIMD run their halo prologue at opposite ends of a stage (see the stage body); measured neutral
// (profiles/r3_wino4_ab.txt: a VALU instruction of one wave waits out the running 64-cycle MFMA of the other, so the head of
// the wave that overlaps took 2100 cycles instead of 1500), kept as a switch
#ifndef SSDE_W4_STAGGER
#define SSDE_W4_STAGGER 0
#endif
// 1: a stage head reads the GroupNorm tables before its fragment / transform reads
#ifndef SSDE_W4_GNFIRST
#define SSDE_W4_GNFIRST 1
#endif
// rows of the output tile a thread has in flight in the second round of the shared epilogue (residual loads issued before
// the first use; the first round still holds half of the accumulators and keeps 4)
#ifndef SSDE_W4_EPI_BATCH
#define SSDE_W4_EPI_BATCH 8
#endif
// timing experiments (profiles/r4_wino4_upper_bounds.txt): variants that compute WRONG results on purpose, never the product
#ifndef SSDE_W4_EXP_NOBARRIER
#define SSDE_W4_EXP_NOBARRIER 0
#endif
#ifndef SSDE_W4_EXP_NOSTORE
#define SSDE_W4_EXP_NOSTORE 0
#endif
// 1: the stage body without the GroupNorm / SiLU prologue and without both passes of the input transform (V is garbage): what the
// main loop would cost if V arrived already transformed, from a separate HBM-bound pass (the classical two-kernel Winograd)
#ifndef SSDE_W4_EXP_NOXFORM
#define SSDE_W4_EXP_NOXFORM 0
#endif
#if SSDE_W4_PRIO
#define SSDE_W4_HI() __builtin_amdgcn_s_setprio(2)
#define SSDE_W4_LO() __builtin_amdgcn_s_setprio(0)
#else
#define SSDE_W4_HI() do { } while (0)
#define SSDE_W4_LO() do { } while (0)
#endif


namespace {

// 8 waves per workgroup, two per SIMD: 9 positions x one 32x32 block = 144 accumulator registers each.  (12 waves, three per
// SIMD at 96 accumulators, measured 11-25 % slower -- profiles/r2_wino4_ab_12_waves.txt: the stalls of this kernel are not
// latencies a third wave would cover, and the output transform spilled at 168 registers; the variant is no longer built.)
constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kNP = 72 / kWaves;                   // positions per wave
constexpr int kPS = kWaves / 2;                    // wave (q, h) owns positions q + kPS * j
constexpr int kEpiThreads = 512;                   // threads of the output transform and of the shared epilogue
constexpr int kPos = 36, kTiles = 32, kKc = 4;
// LDS paddings, A/B-timed (profiles/r2_wino4_ab_lds_padding.txt): a V pitch of 128 + 24 (second transform pass free of bank
// conflicts instead of the first) and raw channel-pair planes 32 banks apart are both neutral: 35 % of the LDS cycles are
// bank conflicts (PMC), but the LDS is active only a third of the time
#ifndef SSDE_W4_VPAD
#define SSDE_W4_VPAD 16
#endif
#ifndef SSDE_W4_RAWPAD
#define SSDE_W4_RAWPAD 0
#endif
constexpr int kVP = kTiles * kKc + SSDE_W4_VPAD;               // floats per position of V: 128 + 16, so that the 6 lines x 8 items of a
                                                    // wave's transform writes (b64) spread over all banks
constexpr int kVFloats = kPos * kVP;                // one V stage
constexpr int kUFloats = kPos * 64 * kKc;           // 9216: one U stage
constexpr int kURegion = kNP * 32 * kKc;            // 1152: the floats of a stage only wave (q, h) reads: [9 positions][32 couts][4]
static_assert(SSDE_W4_VPAD >= 16, "lanes 48-63 of the transform write the padding columns of V");
constexpr int kMaxRaw = 2;                          // float4 halo items per thread per stage (halo <= 1024 pixels)
constexpr int kLdm = 66;                            // row pitch of the product exchange [pos][16 tiles][64 couts]
constexpr int kLdt = 68;                            // row pitch of the parked output tile [256 pixels][64 couts]

struct Wino4Params {
  ssde_src src;
  const float* wpk;        // [ceil(C/4)][n_tiles][8 waves][9][32][4] (LDS image per stage, a wave's 4.5 KB contiguous)
  int N, H, W, Cout;
  int lTWt, lTHt;          // log2 tiles per patch row / column
  int tiles_x, tiles_per_img, m_tiles, n_tiles;
  const float* bias; const float* chan_add; int chan_add_ld;
  const float* resid; int resid_post;
  float scale;
  float* dst;
  float* gn_part;
  int ksplit;              // 1, 2 or 4 workgroups per tile, each reducing its share of the channel stages (the kernel's kKs)
  unsigned* sync;          // ksplit > 1: this launch's (shares started, shares handed over) pairs, one per tile, from
                           // conv_mfma.hip's g_conv_sync (zero before and after)
  int ticket_off;          // float index of the hand-over ticket in LDS (behind everything else)
  float* wino_v;           // kEmitV: V[pos][t][Ctot] = B^T pro(x) B for the F(4x4,3x3) weight gradient (ssde_conv_args.wino_v)
  int T, tiles_h, tiles_w; // 4x4 tiles of the whole batch / per image column / per image row
};

// 1-D input transform (one column / row of B^T d, B^T rows: [4,0,-5,0,1,0] [0,-4,-4,1,1,0] [0,4,-4,-1,1,0]
// [0,-2,-1,2,1,0] [0,2,-1,-2,1,0] [0,4,0,-5,0,1]), two channels at once
__device__ __forceinline__ void bt6(const float2 (&d)[6], float2 (&o)[6]) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float d0 = e ? d[0].y : d[0].x, d1 = e ? d[1].y : d[1].x, d2 = e ? d[2].y : d[2].x;
    const float d3 = e ? d[3].y : d[3].x, d4 = e ? d[4].y : d[4].x, d5 = e ? d[5].y : d[5].x;
    const float t1 = d4 - 4.f * d2, t2 = d3 - 4.f * d1, t3 = d4 - d2, t4 = d3 - d1;
    const float o0 = 4.f * d0 - 5.f * d2 + d4, o1 = t1 + t2, o2 = t1 - t2, o3 = t3 + 2.f * t4, o4 = t3 - 2.f * t4;
    const float o5 = 4.f * d1 - 5.f * d3 + d5;
    if (e) { o[0].y = o0; o[1].y = o1; o[2].y = o2; o[3].y = o3; o[4].y = o4; o[5].y = o5; }
    else   { o[0].x = o0; o[1].x = o1; o[2].x = o2; o[3].x = o3; o[4].x = o4; o[5].x = o5; }
  }
}

// -DSSDE_W4_SCALAR_BT=1 (A/B only, built with -fno-slp-vectorize): the same arithmetic as scalar v_fma_f32 / v_add_f32
// instead of packed v_pk_* (MI355X_MICROARCH.md prices packed f32 VALU beside bf16 MFMAs as an anti-lever; beside fp32
// MFMAs, which share the VALU datapath, it measured neutral: profiles/r3_wino4_ab.txt)
#ifndef SSDE_W4_SCALAR_BT
#define SSDE_W4_SCALAR_BT 0
#endif
__device__ __forceinline__ void bt6(const ssde_f32x2 (&d)[6], ssde_f32x2 (&o)[6]) {
#if SSDE_W4_SCALAR_BT
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    float d0 = d[0][e], d1 = d[1][e], d2 = d[2][e], d3 = d[3][e], d4 = d[4][e], d5 = d[5][e];
    asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5));    // keeps the halves apart (no SLP re-packing)
    const float t1 = d4 - 4.f * d2, t2 = d3 - 4.f * d1, t3 = d4 - d2, t4 = d3 - d1;
    o[0][e] = 4.f * d0 - 5.f * d2 + d4;
    o[1][e] = t1 + t2;
    o[2][e] = t1 - t2;
    o[3][e] = t3 + 2.f * t4;
    o[4][e] = t3 - 2.f * t4;
    o[5][e] = 4.f * d1 - 5.f * d3 + d5;
  }
  return;
#endif
  const ssde_f32x2 t1 = d[4] - 4.f * d[2], t2 = d[3] - 4.f * d[1], t3 = d[4] - d[2], t4 = d[3] - d[1];
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = t1 + t2;
  o[2] = t1 - t2;
  o[3] = t3 + 2.f * t4;
  o[4] = t3 - 2.f * t4;
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

// kKs = 2 or 4: that many workgroups per tile, each reducing its share of the channel stages (own instantiations: the
// hand-over costs the epilogue registers, and the unsplit kernel is the one the big layers run).  The shares are dealt in
// the order the workgroups START (an atomic counter per tile, as a decoupled look-back scan numbers its blocks): share k
// only ever waits for shares < k, which are resident or done -- no assumption about the dispatch order -- and the sums are
// formed in the fixed order ((s0 + s1) + s2) + s3, so the result does not depend on who arrived when.
// kEmitV (training forward, ssde_conv_args.wino_v): the workgroups of the FIRST cout tile also leave the transformed input of
// every stage in HBM, in the layout the F(4x4,3x3) weight-gradient GEMM reads (wgrad_wino4.hip: V[pos][t][ci]) -- the
// weight gradient of the layer then skips its own input-transform pass (wino4_xform_v_kernel: one read of x, the prologue and
// the 6x6 transform per element, 2.25x the tensor written: 2.4-2.7 ms of a training step).  Layout [pos][Ctot / 4][t][4].  Per stage 1152 float4 (36 positions
// x 32 tiles x 4 channels) leave LDS V[cur], 2.25 per thread, in the last three position slots, which carry no other
// vector-memory instruction.  Stores count in vmcnt like the loads: the counted waits of the stage stay correct (a wait for
// "at most n outstanding" can only become stricter when younger stores are still in flight), and the stores sit as far
// from the next counted wait -- the head of the next stage -- as the stage allows.
template <bool kGn, int kKs, bool kEmitV = false>
__global__ __launch_bounds__(kThreads, kWaves / 4) void conv_wino4_kernel(const Wino4Params p) {
  constexpr bool kSplit = kKs > 1;
  SSDE_LDS(smem);
  float* Vb = smem;                            // [2][kVFloats]
  float* Ub = smem + 2 * kVFloats;             // [2][8 waves][9 positions][32 couts][4]: a wave reads only its own region
  float* rawb = Ub + 2 * kUFloats;             // [2][2 pairs][halo_px][2], then the GroupNorm tables
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;

  // XCD-aware order (as conv_mfma.hip): the cout tiles of one pixel tile run on one XCD
  const int bid = blockIdx.x;
  const int xcd = bid & 7, l = bid >> 3;
  const int nt = l % p.n_tiles;
                                                             // (the kKs workgroups of a tile: same XCD, 8 * n_tiles blocks apart)
  const int mt = (l / (p.n_tiles * kKs)) * 8 + xcd;
#ifdef SSDE_W4_TRACE
  const bool tr_on = lane == 0 && (wave == 0 || wave == 7) && bid == 0 && g_w4_trace != nullptr;
  const int tr_base = (wave == 0 ? 0 : 1) * 128;
#endif
  SSDE_TR(0);
  SSDE_TRR(110);
  if (mt >= p.m_tiles) return;

  const int TWt = 1 << p.lTWt, THt = 1 << p.lTHt;
  const int IMGS = kTiles >> (p.lTWt + p.lTHt);
  const int HWd = 4 * TWt + 2, HH = 4 * THt + 2;
  const int halo_px = IMGS * HH * HWd;
  // floats between the two channel-pair planes of a raw buffer: with SSDE_W4_RAWPAD the planes are 32 banks apart (the 48
  // lanes of a transform read touch 4 tiles x 6 columns of BOTH planes; columns shared by neighbouring tiles are the same
  // address, but planes 8 banks apart collided)
  const int raw_plane = SSDE_W4_RAWPAD ? ((2 * halo_px + 63) & ~63) + 32 : 2 * halo_px;
  const int raw_stride = 2 * raw_plane;        // floats per raw buffer
  const int img0 = (mt / p.tiles_per_img) * IMGS;
  const int trem = mt % p.tiles_per_img;
  const int ty = trem / p.tiles_x, tx = trem % p.tiles_x;
  const int n0 = nt * 64;

  const ssde_src& s = p.src;
  const int Ctot = s.c0 + s.c1;
  const int nst_all = (Ctot + 3) >> 2;
  // this workgroup's share of the reduction: stages st_off .. st_off + nst (`st` below counts from 0: LDS buffer parity and
  // the peeled last stages go by the local count, channels and weights by st + st_off)
  unsigned* sy = kSplit ? p.sync + 2 * ((size_t)mt * p.n_tiles + nt) : nullptr;
  int* ticket = reinterpret_cast<int*>(smem + p.ticket_off);
  if (kSplit && tid == 0) *ticket = (int)atomicAdd(sy, 1u);       // (read after the index set-up below)
  int ks = 0, st_off = 0, nst = nst_all;
  SsdePro pro = ssde_pro_decode(s);
  pro.gn = kGn;
  const int cpg = kGn ? Ctot / s.gn_groups : 1;
  const float inv_cpg = 1.0f / (float)cpg;

  // ---- raw staging plan: item = halo pixel (its 4 channels of the stage are one float4).  The byte offset of the pixel in
  // either source is a 32-bit VGPR, the channel offset of a stage rides in the scalar base (global_load ... v, s[:]): no
  // vector address arithmetic per stage (the launcher checks that a source is smaller than 4 GB) ----
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
    const uint32_t px = (uint32_t)(goff[it] >= 0 ? goff[it] : 0);      // outside the image: a clamped, valid address
    voff0[it] = px * (uint32_t)s.c0 * 4u;
    voff1[it] = px * (uint32_t)s.c1 * 4u;
  }
  // ---- transform plan: a wave owns 8 of the 64 items (tile, channel pair) with all their 6 lines: lane = line * 8 + item,
  // so that pass 2 reads what the SAME wave wrote in pass 1 -- LDS executes a wave's operations in order, and no workgroup
  // barrier is needed between the passes.  Lanes 48-63 repeat lines 0 and 1 of the same items into the padding columns of V
  // (128 .. 143, never read by a fragment): every lane runs the same straight-line code, no exec-masked branch in the
  // stage body (hipcc's waitcnt insertion is exact only in straight-line code) ----
  const int t_line6 = lane >> 3;                          // 0..7
  const bool t_real = t_line6 < 6;
  const int t_line = t_real ? t_line6 : t_line6 - 6;
  const int t_item = wave * 8 + (lane & 7);
  const int t_tile = t_item >> 1, t_pair = t_item & 1;
  int t_base;
  {
    const int il = t_tile >> (p.lTWt + p.lTHt);
    const int tr = (t_tile >> p.lTWt) & (THt - 1), tc = t_tile & (TWt - 1);
    t_base = (il * HH + 4 * tr) * HWd + 4 * tc;
  }
  const int t_vcol = t_real ? t_tile * 4 + t_pair * 2 : kTiles * kKc + (lane & 7) * 2;
  const int t_rawoff = t_pair * raw_plane + (t_base + t_line) * 2;

  // V by-product plan: thread = (tile tid & 31, positions (tid >> 5) + 16 i).  One 32-bit byte offset per thread (the launcher
  // checks that V is smaller than 4 GB), the position block i and the stage's channels ride in the scalar base; 0xFFFFFFFF =
  // this thread stores nothing (another cout tile, a tile outside the batch)
  uint32_t ev_off = 0xFFFFFFFFu;
  const uint32_t ev_lds = (uint32_t)(((tid >> 5) * kVP + (tid & 31) * 4) * 4);
  if constexpr (kEmitV) {
    const int etile = tid & 31;
    const int eil = etile >> (p.lTWt + p.lTHt);
    const int etr = (etile >> p.lTWt) & (THt - 1), etc = etile & (TWt - 1);
    const int eimg = img0 + eil, eyy = ty * THt + etr, exx = tx * TWt + etc;
    // layout [pos][Ctot / 4][t][4]: the 32 tiles of a workgroup are consecutive t for every map the networks have (a patch is
    // whole tile rows of one image, or whole images) -> one 512-byte run per position and stage.  (As [pos][t][Ctot], the
    // layout wino4_xform_v_kernel writes, every lane's 16 bytes were a separate memory transaction: the forward + input-gradient
    // class went from 25.0 to 31.6 ms, profiles/r4_wino_v_from_forward_ab.txt)
    if (p.wino_v != nullptr && nt == 0 && eimg < p.N && eyy < p.tiles_h && exx < p.tiles_w)
      ev_off = (uint32_t)((((tid >> 5) * (Ctot >> 2)) * p.T + (eimg * p.tiles_h + eyy) * p.tiles_w + exx) * 16);
  }
  auto emit_v = [&](const float* Vc, int st, int it) __attribute__((always_inline)) {
    if constexpr (kEmitV) {
      if (ev_off != 0xFFFFFFFFu && (it < 2 || tid < 128)) {          // positions 32..35: the first four 32-thread rows
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(Vc) + it * (16 * kVP * 4) + ev_lds);
        char* base = reinterpret_cast<char*>(p.wino_v) + ((size_t)it * 16 * (Ctot >> 2) + (size_t)(st + st_off)) * p.T * 16;
        *reinterpret_cast<float4*>(base + ev_off) = v;
      }
    }
  };
  // GroupNorm tables in LDS: (mean, rstd) of every (tile image, group), gamma and beta of every channel
  float* gn_tab = rawb + 2 * raw_stride;       // [IMGS][groups][2]
  float* gb_tab = gn_tab + 2 * IMGS * (kGn ? s.gn_groups : 0);   // [2][Ctot]
  // halo loads of stage st: opaque to hipcc (SSDE_GLOAD16), their vmcnt accounting is the stage body's
  ssde_f32x4 rv[kMaxRaw];
  auto load_piece = [&](int st, int k) __attribute__((always_inline)) {
    const int c_base = (st + st_off) * 4;
    const bool second = c_base >= s.c0;
    const float* sb = second ? s.p1 + (c_base - s.c0) : s.p0 + c_base;
    const uint32_t vo = second ? voff1[k] : voff0[k];
    SSDE_GLOAD16(rv[k], vo, sb);
  };
  // prologue + raw LDS store (channel-pair major) of the halo in rv = stage st, in two steps so that a stage can put the few
  // LDS reads of the GroupNorm tables AHEAD of its burst of fragment / transform reads (LDS returns in order: the prologue
  // arithmetic then starts when 4 reads have come back instead of 16)
  struct GnRegs { float4 gam, bet; float2 mr[kMaxRaw]; };
  auto gn_fetch = [&](int st) __attribute__((always_inline)) {
    GnRegs r;
    r.gam = make_float4(1.f, 1.f, 1.f, 1.f); r.bet = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < kMaxRaw; ++it) r.mr[it] = make_float2(0.f, 1.f);
    if (kGn) {
      const int c_cur = (st + st_off) * 4;
      r.gam = *reinterpret_cast<const float4*>(gb_tab + c_cur);
      r.bet = *reinterpret_cast<const float4*>(gb_tab + Ctot + c_cur);
      const int g = (int)(((float)c_cur + 0.5f) * inv_cpg);          // c_cur / cpg, exact for these small integers
#pragma unroll
      for (int it = 0; it < kMaxRaw; ++it) r.mr[it] = *reinterpret_cast<const float2*>(gn_tab + 2 * (gil[it] + g));   // (item 1 of most lanes: entry 0)
    }
    return r;
  };
  auto store_raw_with = [&](float* rw, int st, const GnRegs& r) __attribute__((always_inline)) {
    const int c_cur = (st + st_off) * 4;
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
  // B^T d B in two 1-D passes over the 6x6 tile, the second in place: pass 1 lane = (column x), pass 2 lane = (row y)
  auto pass1 = [&](const float* rw, float* Vn) {
    const float* rp = rw + t_rawoff;
    float2 d[6], o[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) d[a] = *reinterpret_cast<const float2*>(rp + a * HWd * 2);
    bt6(d, o);
#pragma unroll
    for (int a = 0; a < 6; ++a) *reinterpret_cast<float2*>(Vn + (a * 6 + t_line) * kVP + t_vcol) = o[a];
  };
  auto pass2 = [&](float* Vn) {
    float* vp = Vn + (t_line * 6) * kVP + t_vcol;
    float2 d[6], o[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) d[b] = *reinterpret_cast<const float2*>(vp + b * kVP);
    bt6(d, o);
#pragma unroll
    for (int b = 0; b < 6; ++b) *reinterpret_cast<float2*>(vp + b * kVP) = o[b];
  };
  // Weights of a stage: the host packed them as the LDS image (36 KB per stage) in the order [wave][position j][32 couts][4]:
  // wave (q, h) moves -- and is the only reader of -- the 9 x 512 B of its positions q + 4 j and its cout half h, as 4.5
  // LDS-DMA pieces of 1 KB (the last one on lanes 0-31).  No other wave waits for these bytes: the wave counts its own
  // pieces with vmcnt right before the fragment read that needs them, and the stage barrier carries no vmcnt(0).
  // One scalar base per stage, the lane's bytes as a constant 32-bit offset, the pieces as immediates -2048 .. +2048.
  const uint32_t w_voff = (uint32_t)((wave * kURegion + 2 * 256 + lane * 4) * 4);
  auto w_base = [&](int st) { return p.wpk + ((size_t)(st + st_off) * p.n_tiles + nt) * kUFloats; };
  auto w_ldst = [&](float* Un) { return Un + wave * kURegion + 2 * 256; };

  const int wq = wave >> 1;                     // this wave's positions wq + 4 j (its cout half is wave & 1)
  f32x16 acc[kNP];
#pragma unroll
  for (int j = 0; j < kNP; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int a_off = li * 4 + 2 * lh, b_off = wave * kURegion + li * 4 + 2 * lh;

  if constexpr (kSplit) {
    __syncthreads();
    ks = __builtin_amdgcn_readfirstlane(*ticket);
    st_off = nst_all * ks / kKs;
    nst = nst_all * (ks + 1) / kKs - st_off;
  }
  // ---- pipeline fill.  Leaves what every stage expects on entry: V[cur] transformed, the wave's weight pieces of the stage
  // issued, raw[nxt] = the activated halo of stage st + 1, rv = the halo of stage st + 2 ----
  SSDE_TR(1);
  {
    const float* wb = w_base(0);
    float* ld = w_ldst(Ub);
    SSDE_GLDS16_S(w_voff, wb, ld, -2048);
    SSDE_GLDS16_S_SAME_BASE(w_voff, wb, ld, -1024);
    SSDE_GLDS16_S_SAME_BASE(w_voff, wb, ld, 0);
    SSDE_GLDS16_S_SAME_BASE(w_voff, wb, ld, 1024);
    SSDE_GLDS16_S_SAME_BASE_LO32(w_voff, wb, ld, 2048);
  }
#pragma unroll
  for (int k = 0; k < kMaxRaw; ++k) load_piece(0, k);
  // (the table loads below are issued behind the weight pieces and the halo of stage 0: one memory latency for all of them)
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
  pass1(rawb, Vb);
  SSDE_LDS_BARRIER();                          // (a wave's lanes run in lockstep: only the test emulator's fibers need this one)
  pass2(Vb);
  SSDE_WAIT_VMCNT_FOR(0, rv[0], rv[1]);        // unconditional: every path from a halo load to its use passes a counted wait
  if (nst > 1) store_raw(rawb + raw_stride, 1);
  // group Y (waves 4-7, the second wave of every SIMD, SSDE_W4_STAGGER): activates and stores its halo at the END of a
  // stage instead of at the head, see below
  const bool grp_y = SSDE_W4_STAGGER && wave >= 4;
  if (nst > 2 && !grp_y) {
#pragma unroll
    for (int k = 0; k < kMaxRaw; ++k) load_piece(2, k);
  }
  SSDE_WAIT_VMCNT_FOR(0, rv[0], rv[1]);
  SSDE_LDS_BARRIER();
  SSDE_TR(2);

  // ---- one stage.  HAS1: a stage st + 1 exists (its weights are fetched and its input transformed here), HAS2: the halo of
  // stage st + 2 is activated and stored, HASL: a halo is fetched (group X: that of stage st + 3, stored at the head of the
  // next stage; group Y: that of stage st + 2, stored at the end of this one).  Compile-time flags: the main loop runs the
  // all-true body, the last stages are peeled, and the body has no branch.
  //
  //   head     own weight pieces 0, 1 landed | fragment reads of positions 0..2 | pass-1 reads | X: prologue of rv -> raw[cur]
  //   slot j   fragment reads of position j + 3 | 2 MFMAs of position j | ONE vector-memory instruction (slots 0..6:
  //            P0' H0' P1' H1' P2' P3' P4' -- as a burst in slots 0-1 the halo loads and the first pieces queued behind each
  //            other in the address path: 1500 cycles for two positions, tools/wino4_trace.py / profiles/r3_wino4_trace_a.txt)
  //            after slot 1: pass-1 arithmetic and writes; after 3: pass-2 reads; after 5: pass-2 arithmetic and writes
  //   tail     Y: prologue of rv -> raw[cur] | LDS-only barrier
  // X and Y are the two waves of a SIMD: while X runs the prologue (VALU: GroupNorm, SiLU -- two quarter-rate
  // transcendentals per element) Y already issues MFMAs, and Y's prologue runs beside X's last positions.  With every wave
  // doing the same thing at the same time each phase was bound by ITS resource while the matrix pipe idled.
  //
  // VMEM queue of a wave (in issue order; P = weight piece of this stage, ' = of the next, H = halo load):
  //   on entry           [P0 H0 P1 H1 P2 P3 P4]  (Y: the H are done)   vmcnt(3): pieces 0, 1 (positions 0..3) and X's rv
  //   slot 1 (pos 4, 5)  [P2 P3 P4 P0']                                 piece 2 <=> vmcnt(2 + HAS1)
  //   slot 3 (pos 6, 7)  [P3 P4 P0' H0' P1']                            piece 3 <=> vmcnt(1 + 2 HAS1 + HASL)
  //   slot 5 (pos 8)     [P4 P0' H0' P1' H1' P2']                       piece 4 <=> vmcnt(3 HAS1 + 2 HASL)
  //   Y's tail           [P0' H0' P1' H1' P2' P3' P4']                  rv      <=> vmcnt(3 HAS1)
  // (fragment reads run SSDE_W4_PF = 3 positions ahead, so the read of position j + 3 is what a slot's count protects) ----
  static_assert(SSDE_W4_PF == 3 && kWaves == 8, "the vmcnt counts of the stage body assume reads 3 positions ahead");
  auto stage = [&](auto H1, auto H2, auto HL, auto GY, const int st) __attribute__((always_inline)) {
    constexpr bool has1 = decltype(H1)::value, has2 = decltype(H2)::value, hasl = decltype(HL)::value, gy = decltype(GY)::value;
    constexpr int n1 = has1 ? 1 : 0, nl = hasl ? 1 : 0;
    const int st_l = gy ? st + 2 : st + 3;              // the stage whose halo this stage fetches
    const int cur = st & 1, nxt = cur ^ 1;
    const float* Vc = Vb + cur * kVFloats;
    const float* Uc = Ub + cur * kUFloats;
    float* Vn = Vb + nxt * kVFloats;
    ssde_f32x2 td[6], to[6];
    ssde_lds_float* tw = (ssde_lds_float*)(Vn + t_line * kVP + t_vcol);            // pass-1 column of this lane
    ssde_lds_float* vp = (ssde_lds_float*)(Vn + (t_line * 6) * kVP + t_vcol);      // pass-2 row of this lane
    SSDE_OPAQUE_VGPR(tw);
    SSDE_OPAQUE_VGPR(vp);
    ssde_f32x2 af[SSDE_W4_PF + 1], bf[SSDE_W4_PF + 1];
    const float* wb = w_base(has1 ? st + 1 : st);
    float* wl = w_ldst(Ub + nxt * kUFloats);
    // one fragment base per operand, opaque to the compiler: the 9 positions are immediate offsets of the ds_read (left
    // alone, hipcc kept a VGPR and a 3-operand add per position and operand: 18 VALU per stage and 16 registers)
    ssde_lds_cfloat* va = (ssde_lds_cfloat*)(Vc + wq * kVP + a_off);
    ssde_lds_cfloat* ua = (ssde_lds_cfloat*)(Uc + b_off);
    SSDE_OPAQUE_VGPR(va);
    SSDE_OPAQUE_VGPR(ua);
    // ---- head ----
    GnRegs gnr;
    if (has2 && !gy && SSDE_W4_GNFIRST && !SSDE_W4_EXP_NOXFORM) gnr = gn_fetch(st + 2);        // (tables: no dependence on the weight pieces)
    if (has2 && !gy) SSDE_WAIT_VMCNT_FOR(3, rv[0], rv[1]); else SSDE_WAIT_VMCNT_FENCE(3);
#pragma unroll
    for (int j = 0; j < SSDE_W4_PF; ++j) {
      af[j] = *(ssde_lds_cfloat2*)(va + kPS * j * kVP);
      bf[j] = *(ssde_lds_cfloat2*)(ua + j * 128);
    }
    if (has1 && !SSDE_W4_EXP_NOXFORM) {
      const float* rp = rawb + nxt * raw_stride + t_rawoff;
#pragma unroll
      for (int a = 0; a < 6; ++a) { const float2 q = *reinterpret_cast<const float2*>(rp + a * HWd * 2); td[a].x = q.x; td[a].y = q.y; }
    }
    if (has2 && !gy && !SSDE_W4_EXP_NOXFORM) {
      SSDE_W4_HI();
      if (SSDE_W4_GNFIRST) store_raw_with(rawb + cur * raw_stride, st + 2, gnr);
      else store_raw(rawb + cur * raw_stride, st + 2);
      SSDE_W4_LO();
    }
    if (st < 8) SSDE_TR(8 + st * 10 + 1);
    __builtin_amdgcn_sched_barrier(0);
#define SSDE_W4_POS(J)                                                                                          \
    do {                                                                                                         \
      if ((J) + SSDE_W4_PF < kNP) {                                                                                \
        af[((J) + SSDE_W4_PF) % (SSDE_W4_PF + 1)] =                                                              \
            *(ssde_lds_cfloat2*)(va + kPS * ((J) + SSDE_W4_PF) * kVP);                                           \
        bf[((J) + SSDE_W4_PF) % (SSDE_W4_PF + 1)] =                                                              \
            *(ssde_lds_cfloat2*)(ua + ((J) + SSDE_W4_PF) * 128);                                                 \
      }                                                                                                          \
      acc[J] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[(J) % (SSDE_W4_PF + 1)].x, bf[(J) % (SSDE_W4_PF + 1)].x, acc[J], 0, 0, 0); \
      acc[J] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[(J) % (SSDE_W4_PF + 1)].y, bf[(J) % (SSDE_W4_PF + 1)].y, acc[J], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
    } while (0)
    // ---- slots ----
    SSDE_W4_POS(0);
    if (has1) SSDE_GLDS16_S(w_voff, wb, wl, -2048);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_WAIT_VMCNT_FENCE(2 + n1);
    SSDE_W4_POS(1);
    if (hasl) load_piece(st_l, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (st < 8) SSDE_TR(8 + st * 10 + 2);
    if (has1 && !SSDE_W4_EXP_NOXFORM) {
      SSDE_W4_HI();
      bt6(td, to);
#pragma unroll
      for (int a = 0; a < 6; ++a) *(ssde_lds_float2*)(tw + a * 6 * kVP) = to[a];
      SSDE_W4_LO();
    }
    if (st < 8) SSDE_TR(8 + st * 10 + 3);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4_POS(2);
    if (has1) SSDE_GLDS16_S_SAME_BASE(w_voff, wb, wl, -1024);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_WAIT_VMCNT_FENCE(1 + 2 * n1 + nl);
    SSDE_W4_POS(3);
    if (hasl) load_piece(st_l, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (st < 8) SSDE_TR(8 + st * 10 + 4);
    if (has1 && !SSDE_W4_EXP_NOXFORM) {
#pragma unroll
      for (int b = 0; b < 6; ++b) td[b] = *(ssde_lds_float2*)(vp + b * kVP);
    }
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4_POS(4);
    if (has1) SSDE_GLDS16_S_SAME_BASE(w_voff, wb, wl, 0);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_WAIT_VMCNT_FENCE(3 * n1 + 2 * nl);
    SSDE_W4_POS(5);
    if (has1) SSDE_GLDS16_S_SAME_BASE(w_voff, wb, wl, 1024);
    __builtin_amdgcn_sched_barrier(0);
    if (has1 && !SSDE_W4_EXP_NOXFORM) {
      SSDE_W4_HI();
      bt6(td, to);
#pragma unroll
      for (int b = 0; b < 6; ++b) *(ssde_lds_float2*)(vp + b * kVP) = to[b];
      SSDE_W4_LO();
    }
    if (st < 8) SSDE_TR(8 + st * 10 + 5);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4_POS(6);
    if (has1) SSDE_GLDS16_S_SAME_BASE_LO32(w_voff, wb, wl, 2048);
    emit_v(Vc, st, 0);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4_POS(7);
    emit_v(Vc, st, 1);
    __builtin_amdgcn_sched_barrier(0);
    SSDE_W4_POS(8);
    emit_v(Vc, st, 2);
#undef SSDE_W4_POS
    if (st < 8) SSDE_TR(8 + st * 10 + 6);
    if (has2 && gy && !SSDE_W4_EXP_NOXFORM) {
      SSDE_WAIT_VMCNT_FOR(3 * n1, rv[0], rv[1]);
      SSDE_W4_HI();
      store_raw(rawb + cur * raw_stride, st + 2);
      SSDE_W4_LO();
    }
#if SSDE_W4_EXP_NOBARRIER
    // TIMING EXPERIMENT ONLY (wrong results): no stage barrier -- the waves run free, an upper bound on what any scheme that
    // lets waves of a SIMD drift apart (third V buffer + LDS ready/free counters) could gain in the loop
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
#else
    SSDE_LDS_BARRIER();
#endif
    if (st < 8) SSDE_TR(8 + st * 10 + 8);
  };
  {
    using T = std::true_type; using F = std::false_type;
    // (written so that the stage that consumes the halo in flight is the only successor of the stage that fetched it: the
    //  registers of an asm load are not protected by hipcc on a path it merely cannot rule out, tests/test_isa_guards.py)
    int st = 0;
    if (!grp_y) {
      if (nst >= 3) {
        for (; st + 3 < nst; ++st) stage(T{}, T{}, T{}, F{}, st);
        stage(T{}, T{}, F{}, F{}, st); ++st;
        stage(T{}, F{}, F{}, F{}, st); ++st;
      } else if (nst == 2) {
        stage(T{}, F{}, F{}, F{}, st); ++st;
      }
      stage(F{}, F{}, F{}, F{}, st);
    } else {
      for (; st + 2 < nst; ++st) stage(T{}, T{}, T{}, T{}, st);
      if (nst >= 2) { stage(T{}, F{}, F{}, T{}, st); ++st; }
      stage(F{}, F{}, F{}, T{}, st);
    }
  }
  SSDE_TR(3);

  // ---- epilogue, 16 tiles (= accumulator rows r < 8, then r >= 8 of every wave) at a time ----
  // The products of a round go to LDS as M[pos][tile 16][64 couts] (pitch kLdm), every thread applies A^T M A to one
  // (tile, cout pair), parks the 4x4 outputs as [256 pixels][64 couts] and the shared epilogue stores them.  All waves
  // take part in both rounds and each round retires half of a wave's accumulators, so nothing spills.  (Rounds over
  // the two 32-cout halves instead kept 144 accumulators live through the transform of the other half: 81 scratch
  // stores per lane and 105 k cycles of epilogue per workgroup, tools/wino4_trace.py.)
  SsdeEpi e{p.bias, p.chan_add, p.chan_add_ld, p.resid, p.resid_post, p.scale, p.dst, p.Cout, p.gn_part};
  // GroupNorm partials: one entry per (image, workgroup tile, round) when the tile is part of one image; per image when a
  // round holds IMGS / 2 whole images (IMGS >= 2: 16 / (tiles per image) images per round)
  const int gn_base = !p.gn_part ? -1 : (IMGS == 1 ? (img0 * p.tiles_per_img + trem) * 2 : img0);
  const int rpi_log2 = IMGS > 2 ? 8 - (4 - p.lTWt - p.lTHt) : 30;             // rows per image in a round: 256 / (IMGS / 2)
  const bool e_on = tid < kEpiThreads;                                       // waves 8-11 (if any) only hand over products
  const int wh = wave & 1;
  const int e_tl = e_on ? tid >> 5 : 0, e_cp = tid & 31;
  float* park = smem;                                                         // [256][kLdt], aliases the products
  // split reduction: share 0 leaves its raw 4x4 outputs in the tile's own part of dst, shares 1 .. kKs - 2 add theirs to
  // them in turn, the last share adds its own and runs the epilogue.  sy[1] counts the shares that have handed over.
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
    if (rnd == 0) SSDE_TR(100);
    __syncthreads();
    if (rnd == 0) SSDE_TR(101);
    // Y = A^T M A for (tile, couts 2 cp, 2 cp + 1); A^T rows [1,1,1,1,1,0] [0,1,-1,2,-2,0] [0,1,1,4,4,0] [0,1,-1,8,-8,1]
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
      // column px of A: (1, 0, 0, 0), (1, 1, 1, 1), (1, -1, 1, -1), (1, 2, 4, 8), (1, -2, 4, -8), (0, 0, 0, 1)
      constexpr float kA[6][4] = {{1.f, 0.f, 0.f, 0.f}, {1.f, 1.f, 1.f, 1.f}, {1.f, -1.f, 1.f, -1.f},
                                  {1.f, 2.f, 4.f, 8.f}, {1.f, -2.f, 4.f, -8.f}, {0.f, 0.f, 0.f, 1.f}};
#pragma unroll
      for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx)
          if (kA[px][dx] != 0.f) { y[dy][dx].x += kA[px][dx] * t[dy].x; y[dy][dx].y += kA[px][dx] * t[dy].y; }
    }
    if (rnd == 0) SSDE_TR(102);
    __syncthreads();                           // every thread has read its products: the parked tile may overwrite them
    if (rnd == 0) SSDE_TR(103);
    if (e_on) {
#pragma unroll
      for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx)
          *reinterpret_cast<float2*>(park + (e_tl * 16 + dy * 4 + dx) * kLdt + 2 * e_cp) = y[dy][dx];
    }
    if (rnd == 0) SSDE_TR(104);
    __syncthreads();
    if (rnd == 0) SSDE_TR(105);
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
    if constexpr (kSplit) {
      const bool first = ks == 0, last = ks == kKs - 1;
      // The sums travel as agent-scope dword accesses, coherent at the device level by themselves (no __threadfence():
      // that is a write-back of the whole L2 per workgroup, conv_mfma.hip)
      if (!first && rnd == 0) {
        if (tid == 0)
          while (__hip_atomic_load(sy + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)ks) __builtin_amdgcn_s_sleep(4);
        __syncthreads();
      }
      // (4 float4 of a thread at a time: round 0 still holds half of the accumulators)
      constexpr int kIters = 256 * 16 / kThreads, kBatch = 4;
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
        float o[kBatch][4];
        if (!first) {
          // the batch's loads in flight before the first add
#pragma unroll
          for (int it = 0; it < kBatch; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              o[it][k] = dp[it] ? __hip_atomic_load(dp[it] + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
          for (int it = 0; it < kBatch; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) o[it][k] += tp[it][k];            // (sum so far) + own share
        } else {
#pragma unroll
          for (int it = 0; it < kBatch; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) o[it][k] = tp[it][k];
        }
        if (last) {
#pragma unroll
          for (int it = 0; it < kBatch; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) tp[it][k] = o[it][k];
        } else {
#pragma unroll
          for (int it = 0; it < kBatch; ++it)
            if (dp[it]) {
#pragma unroll
              for (int k = 0; k < 4; ++k) __hip_atomic_store(dp[it] + k, o[it][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
      }
      if (!last) {
        if (rnd == 0) { __syncthreads(); continue; }                     // (park is refilled by round 1)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0);      // every store of this thread is acknowledged ...
        __syncthreads();                    // ... and every thread's
        if (tid == 0) atomicAdd(sy + 1, 1u);
        return;
      }
      __syncthreads();
      if (rnd == 1 && tid == 0) { sy[0] = 0u; sy[1] = 0u; }              // ready for the next launch that is dealt these slots
    }
#if SSDE_W4_EXP_NOSTORE
    // TIMING EXPERIMENT ONLY (no output): the epilogue's coalesced store skipped -- an upper bound on what overlapping the
    // store burst with the next tile's fill could gain
    if (p.scale == 12345.f)
#endif
    // round 1 has no accumulators left: all 8 rows of a thread (residual loads) in flight instead of 4
    if (rnd == 0) ssde_store_tile<256, 64, kEpiThreads, 4, 0>(park, kLdt, n0, e, pixfn, gn_entry, rpi_log2, gn_max);
    else ssde_store_tile<256, 64, kEpiThreads, SSDE_W4_EPI_BATCH, 0>(park, kLdt, n0, e, pixfn, gn_entry, rpi_log2, gn_max);
    if (rnd == 0) SSDE_TR(106);
    if (rnd == 0) { __syncthreads(); SSDE_TR(4); }
  }
  SSDE_TR(5);
  SSDE_TRR(111);
}

int pow2_floor(int v) { int q = 1; while (q * 2 <= v) q *= 2; return q; }

}  // namespace

// over how many workgroups per tile (1, 2 or 4) a launch of `wgs` tiles over `ctot` input channels splits its reduction: up
// to 256 workgroups, at least 16 channel stages each (engine.py picks this kernel for 8x8 maps only where the split fills
// the chip -- same rule there).  SSDE_CONVF_NO_KSPLIT in the launch's flags switches it off.
int ssde_conv_wino4_splits(int wgs, int ctot, int c_out, unsigned flags) {
  if ((flags & SSDE_CONVF_NO_KSPLIT) || c_out % 4 != 0) return 1;
  const int cus = ssde_num_cus();                   // (256 on the MI355X: a quarter / half of the chip covered)
  if (wgs <= cus / 4 && ctot >= 256) return 4;
  if (wgs <= cus / 2 && ctot >= 128) return 2;
  return 1;
}

// stream == (void*)1 with lds_out: plan-only query of the GroupNorm slices per image (conv_mfma.hip, ssde_conv_gn_slices)
int ssde_conv_wino4_launch(const ssde_conv_args* a, void* stream, int* lds_out) {
  SSDE_REQUIRE(a && a->dst && a->main.p0 && a->w_main, "conv(winograd 4x4): null args");
  SSDE_REQUIRE(a->ksize == 3 && a->stride == 1 && a->pad == 1, "conv(winograd 4x4): needs 3x3, stride 1, pad 1");
  SSDE_REQUIRE(a->aux.p0 == nullptr, "conv(winograd 4x4): fused 1x1 source not supported (issue it as a second conv)");
  SSDE_REQUIRE(a->h_in == a->h_out && a->w_in == a->w_out && a->h_out % 4 == 0 && a->w_out % 4 == 0 && a->h_out >= 8 && a->w_out >= 8,
               "conv(winograd 4x4): same-size output, multiples of 4, at least 8x8 (got %dx%d)", a->h_out, a->w_out);
  const ssde_src& s = a->main;
  SSDE_REQUIRE(s.c0 > 0 && s.c0 % 4 == 0 && s.c1 % 4 == 0 && (s.c1 == 0 || s.p1), "conv(winograd 4x4): channels must be multiples of 4");
  const bool gn = s.pro_mode == SSDE_PRO_GN || s.pro_mode == SSDE_PRO_GN_SILU;
  if (gn) {
    SSDE_REQUIRE(s.gn_groups > 0 && (s.c0 + s.c1) % s.gn_groups == 0 && ((s.c0 + s.c1) / s.gn_groups) % 4 == 0,
                 "conv(winograd 4x4): GroupNorm needs channels-per-group %% 4 == 0");
    SSDE_REQUIRE(s.gn_mean && s.gn_rstd && s.gn_gamma && s.gn_beta, "conv(winograd 4x4): GroupNorm pointers missing");
  }
  SSDE_REQUIRE(s.drop_thresh == 0 || s.drop_seed, "conv(winograd 4x4): dropout seed pointer missing");
  // the halo loads address a source as scalar base + 32-bit byte offset
  SSDE_REQUIRE((unsigned long long)a->n * a->h_in * a->w_in * (unsigned)(s.c0 > s.c1 ? s.c0 : s.c1) * 4ull < (1ull << 32),
               "conv(winograd 4x4): a source of 4 GB or more is not addressable by this kernel");
  Wino4Params p;
  p.src = s; p.wpk = a->w_main;
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
  p.wino_v = a->wino_v; p.tiles_h = a->h_out / 4; p.tiles_w = a->w_out / 4; p.T = a->n * p.tiles_h * p.tiles_w;
  SSDE_REQUIRE(!a->wino_v || 36ull * (unsigned long long)p.T * (unsigned)(s.c0 + s.c1) * 4ull < (1ull << 32),
               "conv(winograd 4x4): a transformed-input by-product of 4 GB or more is not addressable by this kernel");
  // GroupNorm partials: a workgroup tile is part of one image (two epilogue rounds: 2 x tiles_per_img slices of 8 wave
  // entries) or holds `imgs` >= 2 whole images (imgs / 2 per round; 512 / imgs rows per image >= the 32 rows of one
  // epilogue trip)
  const bool gn_ok = a->c_out % 4 == 0 && (imgs == 1 || (p.tiles_per_img == 1 && imgs <= 8));
  SSDE_REQUIRE(!a->gn_part || gn_ok, "conv(winograd 4x4): GroupNorm partials not available for this tiling");
  if (lds_out && stream == reinterpret_cast<void*>(1)) {
    *lds_out = gn_ok ? (imgs == 1 ? 2 * p.tiles_per_img : 1) * (kEpiThreads / 64) : 0;
    return SSDE_OK;
  }
  const int halo_px = imgs * (4 * tht + 2) * (4 * twt + 2);
  SSDE_REQUIRE(halo_px <= kMaxRaw * kThreads, "conv(winograd 4x4): halo of %d pixels exceeds the staging plan", halo_px);
  const int raw_plane = SSDE_W4_RAWPAD ? ((2 * halo_px + 63) & ~63) + 32 : 2 * halo_px;
  int lds = (2 * kVFloats + 2 * kUFloats + 2 * 2 * raw_plane) * 4;
  if (gn) lds += (2 * imgs * s.gn_groups + 2 * (s.c0 + s.c1)) * 4;
  const int lds_epi = kPos * 16 * kLdm * 4;
  if (lds < lds_epi) lds = lds_epi;
  p.ticket_off = lds / 4;
  lds += 16;
  SSDE_REQUIRE(lds <= 160 * 1024, "conv(winograd 4x4): %d bytes of LDS", lds);
  if (lds_out) { *lds_out = lds; return SSDE_OK; }
  // Split the reduction when the launch would leave half of the CUs or more without a workgroup (8x8 maps at batch 256:
  // 128 tiles of 8 images x 64 couts)
  const int wgs = ssde_cdiv(p.m_tiles, 8) * 8 * p.n_tiles;
  p.ksplit = a->resid != a->dst ? ssde_conv_wino4_splits(wgs, s.c0 + s.c1, a->c_out, a->flags) : 1;
  p.sync = p.ksplit > 1 ? ssde_conv_sync_slots(p.m_tiles * p.n_tiles) : nullptr;
  if (!p.sync) p.ksplit = 1;
  const dim3 grid(wgs * p.ksplit);
  auto go = [&](auto kfn, std::atomic<bool>& attr_set) {
    if (!attr_set) {                            // once per instantiation, before any stream capture
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return false;
      attr_set = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(kThreads), lds, static_cast<hipStream_t>(stream), p);
    return true;
  };
  static std::atomic<bool> set[2][3], setv[2][3];
  const int si = p.ksplit == 4 ? 2 : p.ksplit == 2 ? 1 : 0;
  const bool ok = a->wino_v
      ? (gn ? (si == 2 ? go(conv_wino4_kernel<true, 4, true>, setv[1][2]) : si == 1 ? go(conv_wino4_kernel<true, 2, true>, setv[1][1])
                                                                                    : go(conv_wino4_kernel<true, 1, true>, setv[1][0]))
            : (si == 2 ? go(conv_wino4_kernel<false, 4, true>, setv[0][2]) : si == 1 ? go(conv_wino4_kernel<false, 2, true>, setv[0][1])
                                                                                     : go(conv_wino4_kernel<false, 1, true>, setv[0][0])))
      : (gn ? (si == 2 ? go(conv_wino4_kernel<true, 4>, set[1][2]) : si == 1 ? go(conv_wino4_kernel<true, 2>, set[1][1])
                                                                            : go(conv_wino4_kernel<true, 1>, set[1][0]))
            : (si == 2 ? go(conv_wino4_kernel<false, 4>, set[0][2]) : si == 1 ? go(conv_wino4_kernel<false, 2>, set[0][1])
                                                                             : go(conv_wino4_kernel<false, 1>, set[0][0])));
  SSDE_REQUIRE(ok, "conv(winograd 4x4): hipFuncSetAttribute failed");
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
