// How fast does ONE wave per SIMD issue the matrix phase of conv_wino_kernel (8 positions x 8 v_mfma_f32_16x16x4_f32 with
// double-buffered ds_read_b64 fragments), and what changes with the fragment reads removed, with both waves of a SIMD
// in their matrix phase at once, or with v_mfma_f32_32x32x2_f32?   Cycles per MFMA by s_memtime (wave 0 of block 0).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/wino_mfma_phase.hip -o /tmp/wino_mfma_phase && /tmp/wino_mfma_phase
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode bit 0: fragments from LDS (else registers); bit 1: all 8 waves compute (else waves 0-3 only)
template <int kMode>
__global__ __launch_bounds__(512, 2) void k16(float* out, unsigned long long* cyc, int stages) {
  extern __shared__ float4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 16384; i += 512) smem[i] = (float)((i * 7) & 15) * 0.125f;
  __syncthreads();
  if (!(kMode & 2) && wave >= 4) return;
  const int li = lane & 15, lq = lane >> 4, swz = (lq & 1) << 4;
  const int ph = wave >> 2, tb0 = ((wave >> 1) & 1) * 32, cb0 = (wave & 1) * 32;
  int aoff[2], boff[2];
  for (int a = 0; a < 2; ++a) aoff[a] = ph * 0 + (lq * 64 + ((tb0 + a * 16 + li) ^ swz)) * 2;
  for (int b = 0; b < 2; ++b) boff[b] = 8192 + (lq * 64 + ((cb0 + b * 16 + li) ^ swz)) * 2;
  f32x4 acc[8][2][2];
  for (int p = 0; p < 8; ++p) for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[p][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float2 af[2][2], bf[2][2];
  const float2 cst = make_float2(lane * 1e-3f, 1.0f + lane * 1e-4f);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int st = 0; st < stages; ++st) {
    const float* Vc = smem, *Uc = smem;
    if (kMode & 1) {
      for (int a = 0; a < 2; ++a) af[0][a] = *reinterpret_cast<const float2*>(Vc + aoff[a]);
      for (int b = 0; b < 2; ++b) bf[0][b] = *reinterpret_cast<const float2*>(Uc + boff[b]);
    } else { af[0][0] = af[0][1] = bf[0][0] = bf[0][1] = cst; af[1][0] = af[1][1] = bf[1][0] = bf[1][1] = cst; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int cur = ps & 1;
      if ((kMode & 1) && ps + 1 < 8) {
#pragma unroll
        for (int a = 0; a < 2; ++a) af[cur ^ 1][a] = *reinterpret_cast<const float2*>(Vc + (ps + 1) * 512 + aoff[a]);
#pragma unroll
        for (int b = 0; b < 2; ++b) bf[cur ^ 1][b] = *reinterpret_cast<const float2*>(Uc + (ps + 1) * 512 + boff[b]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[ps][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][a].x, bf[cur][b].x, acc[ps][a][b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[ps][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][a].y, bf[cur][b].y, acc[ps][a][b], 0, 0, 0);
      if ((kMode & 1) && ps + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int p = 0; p < 8; ++p) for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) r += acc[p][a][b][0] + acc[p][a][b][3];
  if (r == 123.456f) out[0] = r;
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) cyc[wave >> 2] = t1 - t0;
}

// the same FLOPs on v_mfma_f32_32x32x2_f32: a wave's 32 tiles x 32 couts x 8 positions = 8 accumulators of 16 registers;
// per position and channel pair ONE 32x32 block, k = 2 per MFMA -> 4 MFMAs (64 cycles each) per position and stage
template <int kMode>
__global__ __launch_bounds__(512, 2) void k32(float* out, unsigned long long* cyc, int stages) {
  extern __shared__ float4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 16384; i += 512) smem[i] = (float)((i * 7) & 15) * 0.125f;
  __syncthreads();
  if (!(kMode & 2) && wave >= 4) return;
  const int l32 = lane & 31, lh = lane >> 5;
  // fragment: lane (i = lane % 32, k = lane / 32) reads channel k of pairs ... as ds_read_b128: 4 consecutive channels? use
  // b64 = channel pair (2 lh', 2 lh' + 1)... here: two b64 reads per operand per position cover 8 channels x 32 rows
  const int aoff = ((wave >> 1) & 1) * 64 + l32 * 2 + lh * 128, boff = 8192 + (wave & 1) * 64 + l32 * 2 + lh * 128;
  f32x16 acc[8];
  for (int p = 0; p < 8; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  float2 af[2][2], bf[2][2];
  const float2 cst = make_float2(lane * 1e-3f, 1.0f + lane * 1e-4f);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int st = 0; st < stages; ++st) {
    if (kMode & 1) {
      for (int h = 0; h < 2; ++h) { af[0][h] = *reinterpret_cast<const float2*>(smem + aoff + h * 256); bf[0][h] = *reinterpret_cast<const float2*>(smem + boff + h * 256); }
    } else { af[0][0] = af[0][1] = bf[0][0] = bf[0][1] = cst; af[1][0] = af[1][1] = bf[1][0] = bf[1][1] = cst; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int cur = ps & 1;
      if ((kMode & 1) && ps + 1 < 8) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          af[cur ^ 1][h] = *reinterpret_cast<const float2*>(smem + (ps + 1) * 512 + aoff + h * 256);
          bf[cur ^ 1][h] = *reinterpret_cast<const float2*>(smem + (ps + 1) * 512 + boff + h * 256);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        acc[ps] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][h].x, bf[cur][h].x, acc[ps], 0, 0, 0);
        acc[ps] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][h].y, bf[cur][h].y, acc[ps], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int p = 0; p < 8; ++p) r += acc[p][0] + acc[p][15];
  if (r == 123.456f) out[0] = r;
  if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) cyc[wave >> 2] = t1 - t0;
}

template <class K>
void run(const char* name, K kern, float* d, unsigned long long* dc, int mfma_per_stage, int cyc_per_mfma) {
  const int stages = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipMemset(dc, 0, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 65536, 0, d, dc, stages);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 65536, 0, d, dc, stages);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c[2]; hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
  printf("%-58s %8.3f ms   wave0 %6.1f cycles/MFMA (ideal %d)   wave4 %6.1f\n", name, ms, (double)c[0] / stages / mfma_per_stage, cyc_per_mfma,
         (double)c[1] / stages / mfma_per_stage);
}

int main() {
  float* d; unsigned long long* dc; hipMalloc(&d, 4); hipMalloc(&dc, 16);
  run("16x16x4, register operands, 1 wave/SIMD", k16<0>, d, dc, 64, 32);
  run("16x16x4, LDS fragments (conv_wino stage), 1 wave/SIMD", k16<1>, d, dc, 64, 32);
  run("16x16x4, register operands, 2 waves/SIMD", k16<2>, d, dc, 64, 32);
  run("16x16x4, LDS fragments, 2 waves/SIMD (both in matrix phase)", k16<3>, d, dc, 64, 32);
  run("32x32x2, register operands, 1 wave/SIMD", k32<0>, d, dc, 32, 64);
  run("32x32x2, LDS fragments, 1 wave/SIMD", k32<1>, d, dc, 32, 64);
  run("32x32x2, LDS fragments, 2 waves/SIMD", k32<3>, d, dc, 32, 64);
  return 0;
}
