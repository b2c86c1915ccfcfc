// Which of two co-resident workgroups of a CU is "the second one"?  (conv_wino4r.hip, half-size workgroups: the second workgroup
// of every CU starts half a tile late so that one's epilogue runs under the other's MFMAs.)  512 workgroups of 256 threads with
// 78 KB of LDS each = exactly two per CU; every workgroup records HW_ID, XCC_ID, LDS_ALLOC and its start time.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/wg_slots.hip -o tools/microbench/wg_slots && tools/microbench/wg_slots
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned long long* out, int spin) {
  extern __shared__ float4 smem4[];
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);        // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);      // HW_REG_XCC_ID
    const unsigned lds = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);       // HW_REG_LDS_ALLOC
    out[blockIdx.x * 4 + 0] = hw; out[blockIdx.x * 4 + 1] = xcc; out[blockIdx.x * 4 + 2] = lds;
    out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(16);
  if (spin < 0) reinterpret_cast<float*>(smem4)[threadIdx.x] = 1.f;
}
int main() {
  const int n = 1024;
  unsigned long long* d; hipMalloc(&d, n * 32); hipMemset(d, 0, n * 32);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 80000);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 78336, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(n * 4);
    hipMemcpy(h.data(), d, n * 32, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (int i = 0; i < n; ++i) if (h[i * 4 + 3] < tmin) tmin = h[i * 4 + 3];
    printf("== pass %d: bid  hw_id     xcc   lds_alloc  start  | wave_id simd cu sh se\n", rep);
    std::map<unsigned long long, std::vector<int>> by_cu;
    for (int i = 0; i < n; ++i) {
      const unsigned hw = (unsigned)h[i * 4], xcc = (unsigned)h[i * 4 + 1] & 0xf;
      const unsigned long long key = ((unsigned long long)xcc << 32) | (hw & 0xff00 & ~0xffu) | ((hw >> 8) & 0xff) << 0;
      by_cu[((unsigned long long)xcc << 16) | ((hw >> 8) & 0xffff)].push_back(i);
      if (i < 80 || (i >= 256 && i < 272) || (i >= 512 && i < 528))
        printf("%4d  %08x  %04x  %08x  %8llu | %2u %u %2u %u %u\n", i, hw, (unsigned)h[i * 4 + 1], (unsigned)h[i * 4 + 2], h[i * 4 + 3] - tmin,
               hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7);
      (void)key;
    }
    printf("distinct (xcc, hw_id[23:8]) keys: %zu\n", by_cu.size());
    int shown = 0;
    for (auto& kv : by_cu) {
      if (shown++ >= 12) break;
      printf(" key %llx:", kv.first);
      for (int i : kv.second) printf("  bid %d (wave_id %u, lds %08x, t %llu)", i, (unsigned)h[i * 4] & 15, (unsigned)h[i * 4 + 2], h[i * 4 + 3] - tmin);
      printf("\n");
    }
  }
  return 0;
}
