// Does fp32 MFMA work of one wave overlap with VALU work of ANOTHER wave on the same SIMD (gfx950)?
// 8 waves per workgroup = 2 per SIMD.  Waves 0-3 run a chain-free stream of v_mfma_f32_16x16x4_f32, waves 4-7 a stream of
// VALU instructions (fma / exp / packed fma).  Times: MFMA alone, VALU alone, both.  If "both" ~ max(a, b) the pipes
// overlap; if ~ a + b they share the issue/execution resource.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_valu_overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int kValuKind>
__global__ __launch_bounds__(512) void k(float* out, int n_mfma, int n_valu, int mode) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      f32x4 acc[8];
      for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
      for (int it = 0; it < n_mfma; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
      for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
  } else if (mode & 2) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const float c = 0.999f, d = 1e-3f;
    for (int it = 0; it < n_valu; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (kValuKind == 0) v[i] = fmaf(v[i], c, d);
        else if (kValuKind == 1) v[i] = __expf(v[i]) * 1e-3f;
        else if ((i & 1) == 0) {             // float2 fma -> v_pk_fma_f32: two elements per instruction
          f32x2 t = {v[i], v[i + 1]};
          t = t * (f32x2){c, c} + (f32x2){d, d};
          v[i] = t.x; v[i + 1] = t.y;
        }
      }
    }
    for (int i = 0; i < 8; ++i) r += v[i];
  }
  if (r == 123.456f) out[0] = r;
}

template <int K>
float run(float* d, int nm, int nv, int mode) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<K>, dim3(256 * 4), dim3(512), 0, 0, d, nm, nv, mode);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<K>, dim3(256 * 4), dim3(512), 0, 0, d, nm, nv, mode);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  float* d; hipMalloc(&d, 4);
  const int nm = 4000;                    // 32000 MFMAs per wave x 32 cycles = 1.02 M cycles
  const char* names[3] = {"v_fma_f32", "v_exp_f32 + v_mul", "v_pk_fma_f32"};
  for (int kind = 0; kind < 3; ++kind) {
    const int nv = kind == 1 ? 6000 : 30000;
    float a, b, c;
    if (kind == 0) { a = run<0>(d, nm, nv, 1); b = run<0>(d, nm, nv, 2); c = run<0>(d, nm, nv, 3); }
    else if (kind == 1) { a = run<1>(d, nm, nv, 1); b = run<1>(d, nm, nv, 2); c = run<1>(d, nm, nv, 3); }
    else { a = run<2>(d, nm, nv, 1); b = run<2>(d, nm, nv, 2); c = run<2>(d, nm, nv, 3); }
    printf("%-20s mfma alone %.3f ms   valu alone %.3f ms   both %.3f ms   (max %.3f, sum %.3f)\n", names[kind], a, b, c,
           a > b ? a : b, a + b);
  }
  return 0;
}
