// What feeds a CU's operands beside a running fp32 MFMA chain, and at what price?  
//
// conv_wino4g_kernel's stage is ~3050 cycles for 2304 matrix cycles; without its 8 LDS-DMA pieces per wave and stage it runs at
// the matrix bound (profiles/r4_wino4_two_kernels.txt).  36 of the 54 KB a stage moves are a wave-PRIVATE weight image that goes
// L2 -> LDS -> the same wave's registers.  This benchmark runs the stage skeleton of that kernel -- 8 waves (two per SIMD), 9
// position slots of kMf v_mfma_f32_32x32x2_f32 each on 9 independent accumulator blocks -- and varies only how operands arrive:
//   none   nothing arrives (the matrix bound)
//   dma    kD pieces of 1 KB per wave and stage by global_load_lds_dwordx4 (wave-private LDS region), the B fragments of a slot
//          read back with ds_read_b64 (kMf / 2 per slot) as the product does
//   vgpr   kG pieces of 1 KB per wave and stage by global_load_dwordx4 straight into the registers the slot's MFMAs use as their
//          B operand; refilled right after the slot's MFMAs for the NEXT stage, awaited by a counted s_waitcnt vmcnt
//   mixed  kD LDS-DMA pieces (the shared V) + kG register pieces (the private U), A fragments read from LDS
// Sources: "L2" = every workgroup streams the same 16-stage window (the weight image of a layer: L2-hot), "stream" = every
// workgroup its own addresses, never revisited (the transformed input V: HBM / infinity cache).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/fill_path.hip -o tools/microbench/fill_path && tools/microbench/fill_path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kNP = 9;

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(n) : "memory")
#define WAIT_VMCNT_FOR(n, a) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(n) : "memory")
#define GLOAD16(dst, voff, sbase, imm) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(imm) :)
#define GLDS16(voff, sbase, ldsaddr, imm)                                                                \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3"                 \
               : : "s"(__builtin_amdgcn_readfirstlane((int)(ldsaddr))), "v"(voff), "s"(sbase), "n"(imm) :)

// kD: LDS-DMA pieces per wave and stage (slots 0 .. kD-1), kG: register pieces (slots 0 .. kG-1), kMf: MFMAs per slot (2 = a
// stage of 4 channels, 4 = 8 channels), kReadA: A fragments from a shared LDS image, kReadB: B fragments from the wave's LDS
// region (the product's form; with kG > 0 the B operand is the loaded register instead), kIss: waves that issue the DMA pieces
// (8 = every wave its own; 4 / 2 / 1 = that many waves issue 8 / kIss times as many).
template <int kD, int kG, int kMf, bool kReadA, bool kReadB, int kIss>
__global__ __launch_bounds__(512, 2) void fill(const float* src, float* out, unsigned long long* cyc, int stages, int stream) {
  extern __shared__ float4 smem4[];
  char* lds = reinterpret_cast<char*>(smem4);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 147456 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = 1.0f + (i & 7);
  __syncthreads();
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[kNP];
  for (int p = 0; p < kNP; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  // LDS: [0, 18 KB) shared A image [36][32][4]; wave-private regions of 16 KB from 18 KB on (two 8 KB halves = stage parity)
  const char* va = lds + ((wave >> 1) * 512 + li * 16 + 8 * lh);
  const int priv = 18432 + wave * 16384;
  const char* ua = lds + priv + li * 16 + 8 * lh;
  constexpr int kPieces = kD * (8 / kIss);          // DMA pieces an issuing wave moves per stage
  const bool issuer = (wave % (8 / kIss)) == 0;
  const uint32_t voff = (uint32_t)wave * 16384u + (uint32_t)lane * 16u;   // the wave's slot rides in the lane offset: the base stays scalar
  // bytes per (stage, wave): 16 KB slots keep every configuration's addresses apart.  stream bit 0: the DMA pieces stream
  // (workgroup-private addresses, never revisited), bit 1: the register pieces do; otherwise the 16-stage L2-hot window
  const size_t per_stage = (size_t)8 * 16384;
  const char* base_l2 = reinterpret_cast<const char*>(src);
  const char* base_st = base_l2 + (size_t)blockIdx.x * stages * per_stage;
  const char* base_d = (stream & 1) ? base_st : base_l2;
  const char* base_g = (stream & 2) ? base_st : base_l2;
  f32x4 u[kG > 0 ? kG : 1];
  if (kG > 0) {
#pragma unroll
    for (int j = 0; j < kG; ++j) GLOAD16(u[j], voff, base_g + 4096 + (j >> 3) * 8192, ((j & 7) - 4) * 1024);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int st = 0; st < stages; ++st) {
    const char* sd = base_d + (size_t)((stream & 1) ? st : (st & 15)) * per_stage + 4096;
    const char* sg = base_g + (size_t)((stream & 2) ? st : (st & 15)) * per_stage + 4096;
    const int half = (st & 1) * 8192;
    // fragment reads run one slot ahead of their MFMAs (the product: three)
    f32x2 af[2][2], bf[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) af[0][m] = af[1][m] = bf[0][m] = bf[1][m] = f32x2{1.f, 2.f};
    auto frag = [&](int j, int s) __attribute__((always_inline)) {
#pragma unroll
      for (int m = 0; m < kMf / 2; ++m) {
        if (kReadA) af[s][m] = *reinterpret_cast<const f32x2*>(va + 4 * j * 512 + m * 8);
        if (kReadB && kG == 0) bf[s][m] = *reinterpret_cast<const f32x2*>(ua + (st & 1 ? 0 : 8192) + j * 512 + m * 8);
      }
    };
    if (kIss == 8 && kReadB && kG == 0 && kD > 0) WAIT_VMCNT(kD - 1);
    frag(0, 0);
#pragma unroll
    for (int j = 0; j < kNP; ++j) {
      // counted wait: slot j's register piece and DMA piece were issued in this slot one stage ago (register piece first);
      // everything issued since may still fly.  The DMA piece of slot j + 1 must have landed before its fragment read below.
      if (kIss == 8) {
        if (j < kG) WAIT_VMCNT_FOR(kG - 1 + ((j + 1 < kD) ? (kD > 1 ? kD - 2 : 0) : (j < kD ? kD - 1 : kD)), u[j < kG ? j : 0]);
        else if (j + 1 < kD) WAIT_VMCNT(kG + (kD > 1 ? kD - 2 : 0));
      } else if (j == 0) {
        WAIT_VMCNT(kPieces < 63 ? kPieces : 63);          // (vmcnt has 6 bits)
      }
      if (j + 1 < kNP) frag(j + 1, (j + 1) & 1);
#pragma unroll
      for (int m = 0; m < kMf / 2; ++m) {
        const float b0 = (kG > 0 && j < kG) ? u[j < kG ? j : 0][2 * m] : bf[j & 1][m].x;
        const float b1 = (kG > 0 && j < kG) ? u[j < kG ? j : 0][2 * m + 1] : bf[j & 1][m].y;
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j & 1][m].x, b0, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j & 1][m].y, b1, acc[j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kG > 0 && j < kG) GLOAD16(u[j], voff, sg + (j >> 3) * 8192, ((j & 7) - 4) * 1024);
      if (kD > 0 && issuer) {
#pragma unroll
        for (int q = 0; q < (kPieces + kNP - 1) / kNP; ++q) {
          const int pc = j + q * kNP;
          if (pc < kPieces) GLDS16(voff, sd + (pc >> 3) * 8192, (uint32_t)(priv + half + 4096), ((pc & 7) - 4) * 1024);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  WAIT_VMCNT(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int p = 0; p < kNP; ++p) r += acc[p][0] + acc[p][15];
  if (kG > 0) for (int j = 0; j < kG; ++j) r += u[j][0];
  if (r == 123.456f) out[0] = r;
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7)) cyc[wave ? 1 : 0] = t1 - t0;
}

template <class K>
void run(const char* name, K kern, const float* src, float* d, unsigned long long* dc, int kD, int kG, int kMf, int stream) {
  const int stages = stream ? 200 : 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 151552);
  hipMemset(dc, 0, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 151552, 0, src, d, dc, stages, stream);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 151552, 0, src, d, dc, stages, stream);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c[2]; hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
  const double per = (double)(c[0] > c[1] ? c[0] : c[1]) / stages;
  const double bytes = 8.0 * 1024 * (kD + kG);
  printf("%-58s %6.0f cyc/stage (matrix %4d)  %5.1f KB/stage/CU = %5.1f B/clk/CU  wall %7.3f ms -> %.2f GHz\n", name, per, 2 * kNP * kMf * 64,
         bytes / 1024, bytes / per, ms, per * stages / (ms * 1e6));
  fflush(stdout);
}

#define RUN(label, D, G, MF, RA, RB, ISS, STREAM) run(label, fill<D, G, MF, RA, RB, ISS>, src, d, dc, D, G, MF, STREAM)

int main() {
  float* d; unsigned long long* dc; float* src;
  const size_t src_bytes = (size_t)256 * 200 * 8 * 16384 + (1 << 20);      // the streaming runs: 256 workgroups x 200 stages x 128 KB
  if (hipMalloc(&src, src_bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(src, 0, src_bytes);
  hipMalloc(&d, 4); hipMalloc(&dc, 16);
  for (int rep = 0; rep < 2; ++rep) {
    printf("== pass %d\n", rep);
    RUN("none, 4-ch stage", 0, 0, 2, false, false, 8, 0);
    RUN("none + A,B fragment reads (18 ds_read_b64)", 0, 0, 2, true, true, 8, 0);
    RUN("dma 3 pieces/wave (L2)", 3, 0, 2, false, false, 8, 0);
    RUN("dma 5 pieces/wave (L2)", 5, 0, 2, false, false, 8, 0);
    RUN("dma 8 pieces/wave (L2)", 8, 0, 2, false, false, 8, 0);
    RUN("dma 8 pieces/wave (L2) + A,B fragment reads = product", 8, 0, 2, true, true, 8, 0);
    RUN("dma 8 pieces/wave (stream)", 8, 0, 2, false, false, 8, 1);
    RUN("dma 3 pieces/wave (stream)", 3, 0, 2, false, false, 8, 1);
    RUN("dma 64 pieces/stage issued by 4 waves (L2)", 8, 0, 2, false, false, 4, 0);
    RUN("dma 64 pieces/stage issued by 2 waves (L2)", 8, 0, 2, false, false, 2, 0);
    RUN("dma 64 pieces/stage issued by 1 wave (L2)", 8, 0, 2, false, false, 1, 0);
    RUN("vgpr 3 pieces/wave (L2)", 0, 3, 2, false, false, 8, 0);
    RUN("vgpr 5 pieces/wave (L2)", 0, 5, 2, false, false, 8, 0);
    RUN("vgpr 8 pieces/wave (L2)", 0, 8, 2, false, false, 8, 0);
    RUN("vgpr 8 pieces/wave (stream)", 0, 8, 2, false, false, 8, 2);
    RUN("mixed: dma 3 (V) + vgpr 5 (U), A reads  [4-ch design]", 3, 5, 2, true, false, 8, 0);
    RUN("mixed: dma 3 (V, stream) + vgpr 5 (U), A reads", 3, 5, 2, true, false, 8, 1);
    RUN("none, 8-ch stage", 0, 0, 4, false, false, 8, 0);
    RUN("mixed 8-ch: dma 5 (V) + vgpr 9 (U), A reads [8-ch design]", 5, 9, 4, true, false, 8, 0);
    RUN("mixed 8-ch: dma 5 (V, stream) + vgpr 9 (U), A reads", 5, 9, 4, true, false, 8, 1);
    RUN("vgpr 9 pieces/wave, 8-ch stage (L2)", 0, 9, 4, false, false, 8, 0);
  }
  return 0;
}
