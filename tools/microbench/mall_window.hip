// Does a producer -> consumer pair of launches keep its intermediate in the 256 MiB Infinity Cache?  (round 5)
//
// The two-kernel F(4x4,3x3) convolution writes the transformed input V (2.25 x the layer input: 302 MB at 128 ch @32x32, batch 256)
// with wino4_xform_vq_kernel and reads it back with conv_wino4r_kernel; the pass runs at ~4.5 TB/s of HBM traffic and is ~21 % of the
// 3x3 class.  If a V window of S bytes that is written and then read by the NEXT launch is served on-die, cutting a layer into
// batch chunks (pass(chunk) -> matrix kernel(chunk) on the same V window) takes V off the HBM.  This benchmark measures that premise:
//   reuse   write S bytes, read the same S bytes, again and again on the SAME window
//   cold    the same pair of launches, but every repetition on a fresh window of a 3 GiB pool (nothing can be resident)
//   pass    reads S / 2.25 bytes of a (fresh) input and writes S bytes (the transform pass's traffic shape), window reused / cold
// for S = 16 .. 768 MB; 16-byte lanes, grid-stride, 2048 workgroups of 256 threads.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mall_window.hip -o tools/microbench/mall_window && tools/microbench/mall_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void write_k(float4* dst, size_t n4, float v) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
}
__global__ __launch_bounds__(256) void read_k(const float4* src, size_t n4, float* sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { const float4 a = src[i]; acc += (a.x + a.y) + (a.z + a.w); }
  if (acc == 123.456f) *sink = acc;
}
// reads n4_in float4s, writes 9 float4s for every 4 read (2.25 x)
__global__ __launch_bounds__(256) void pass_k(const float4* src, size_t n4_in, float4* dst) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t groups = n4_in / 4;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
    float4 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = src[(size_t)k * groups + g];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float4 u = a[k & 3], w = a[(k + 1) & 3];
      dst[(size_t)k * groups + g] = make_float4(u.x + w.x, u.y - w.y, u.z + w.z, u.w - w.w);
    }
  }
}

int main() {
  const size_t MB = 1u << 20;
  const size_t pool_bytes = 3072 * MB, in_bytes = 1024 * MB;
  char *pool, *inp; float* sink;
  CK(hipMalloc(&pool, pool_bytes)); CK(hipMalloc(&inp, in_bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(pool, 0, pool_bytes)); CK(hipMemset(inp, 0, in_bytes));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  const int grid = 2048, reps = 12;
  const size_t sizes[] = {16, 32, 64, 96, 128, 160, 192, 256, 384, 768};
  for (int pass = 0; pass < 2; ++pass) {
    printf("== pass %d\n", pass);
    for (size_t S_mb : sizes) {
      const size_t S = S_mb * MB, n4 = S / 16;
      for (int cold = 0; cold < 2; ++cold) {
        double tw = 0, tr = 0;
        size_t off = 0;
        for (int r = 0; r < reps + 2; ++r) {
          if (cold) { off += S; if (off + S > pool_bytes) off = 0; }
          float4* win = reinterpret_cast<float4*>(pool + off);
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, win, n4, (float)r);
          CK(hipEventRecord(e1));
          hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, win, n4, sink);
          CK(hipEventRecord(e2));
          CK(hipEventSynchronize(e2));
          float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
          if (r >= 2) { tw += a; tr += b; }
        }
        printf("S = %4zu MB  %-5s  write %7.1f us = %5.2f TB/s   read %7.1f us = %5.2f TB/s\n", S_mb, cold ? "cold" : "reuse",
               tw / reps * 1e3, S / (tw / reps * 1e-3) / 1e12, tr / reps * 1e3, S / (tr / reps * 1e-3) / 1e12);
      }
      // the transform pass's shape: input S / 2.25 (always a fresh input window), output window reused / cold, then the read
      const size_t n4_in = (n4 / 9) * 4, Sin = n4_in * 16;
      for (int cold = 0; cold < 2; ++cold) {
        double tp = 0, tr = 0;
        size_t off = 0, ioff = 0;
        for (int r = 0; r < reps + 2; ++r) {
          if (cold) { off += S; if (off + S > pool_bytes) off = 0; }
          ioff += Sin; if (ioff + Sin > in_bytes) ioff = 0;
          float4* win = reinterpret_cast<float4*>(pool + off);
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(pass_k, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const float4*>(inp + ioff), n4_in, win);
          CK(hipEventRecord(e1));
          hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, win, (n4 / 9) * 9, sink);
          CK(hipEventRecord(e2));
          CK(hipEventSynchronize(e2));
          float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
          if (r >= 2) { tp += a; tr += b; }
        }
        printf("S = %4zu MB  %-5s  pass  %7.1f us = %5.2f TB/s (in + out)   read %7.1f us = %5.2f TB/s\n", S_mb, cold ? "cold" : "reuse",
               tp / reps * 1e3, (S + Sin) / (tp / reps * 1e-3) / 1e12, tr / reps * 1e3, S / (tr / reps * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
