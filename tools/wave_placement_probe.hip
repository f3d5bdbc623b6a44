// Where do the wavefronts of 3-wavefront workgroups land? 1024 workgroups of 192 threads with 40 KB of LDS each (the decoder's shape:
// four resident per CU), every wavefront records HW_REG_HW_ID; the host tallies wavefronts per SIMD of every CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/wave_placement_probe.hip -o tools/wave_placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>

__global__ void probe(uint32_t* out, int spin)
{
  extern __shared__ unsigned char smem[];
  smem[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  uint32_t id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {
  }
  if ((threadIdx.x & 63) == 0) {
    out[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)]     = id;
    out[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 1] = xcc;
  }
  out[0] += smem[(threadIdx.x + 1) % blockDim.x] == 255 ? 1 : 0; // keeps the LDS alive
}

int main(int argc, char** argv)
{
  const int threads = argc > 1 ? atoi(argv[1]) : 192, lds = argc > 2 ? atoi(argv[2]) : 40768, wgs = argc > 3 ? atoi(argv[3]) : 1024;
  const int waves = threads / 64;
  uint32_t* d;
  hipMalloc(&d, (size_t)wgs * waves * 8);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(probe, dim3(wgs), dim3(threads), lds, 0, d, 2000000);
  hipDeviceSynchronize();
  std::vector<uint32_t> h((size_t)wgs * waves * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx950 may differ: print raw too)
  std::map<uint32_t, std::vector<int>> per_cu; // key: xcc, se, sh, cu
  for (int w = 0; w < wgs * waves; ++w) {
    const uint32_t id = h[2 * w], xcc = h[2 * w + 1] & 0xf;
    const uint32_t simd = (id >> 4) & 3, cu = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7;
    auto& v = per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu];
    v.resize(4);
    v[simd]++;
  }
  std::map<std::vector<int>, int> patterns;
  for (auto& kv : per_cu) {
    std::vector<int> s = kv.second;
    patterns[s]++;
  }
  printf("%d workgroups x %d wavefronts, %d B LDS: %zu distinct (xcc, se, sh, cu) keys\n", wgs, waves, lds, per_cu.size());
  for (auto& kv : patterns)
    printf("  wavefronts per SIMD [%d %d %d %d]: %d CUs\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
  printf("  first raw ids: %08x %08x %08x %08x %08x %08x\n", h[0], h[2], h[4], h[6], h[8], h[10]);
  return 0;
}
