// Does the LDS of gfx950 serve 2-byte accesses at odd byte addresses (ds_read_u16 / ds_write_b16), and at what cost?
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_unaligned_probe.hip -o tools/lds_unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(uint32_t* out, int off, int iters, unsigned long long* cyc)
{
  __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x)
    lds[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  // every lane reads two bytes at 2 * lane + off
  uint32_t addr = (uint32_t)(uintptr_t)lds + 2 * threadIdx.x + off;
  uint32_t v    = 0, acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    acc += v;
    addr ^= (v & 0); // keep the dependence
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = v;
  // unaligned write: store 0xBEEF-ish pattern at the same address, read back bytewise
  __syncthreads();
  uint32_t w = 0xA000u + threadIdx.x;
  asm volatile("ds_write_b16 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(w) : "memory");
  __syncthreads();
  out[256 + threadIdx.x] = (uint32_t)lds[2 * threadIdx.x + off] | ((uint32_t)lds[2 * threadIdx.x + off + 1] << 8);
  if (threadIdx.x == 0)
    cyc[0] = t1 - t0, out[1023] = acc;
}

int main()
{
  uint32_t* d;
  unsigned long long* c;
  hipMalloc(&d, 4096);
  hipMalloc(&c, 8);
  for (int off = 0; off < 4; ++off) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, off, 4096, c);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(1024);
    unsigned long long    cy;
    hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    int bad_r = 0, bad_w = 0;
    for (int l = 0; l < 64; ++l) {
      const int      a   = 2 * l + off;
      const uint32_t exp = (uint32_t)(uint8_t)(a * 7 + 3) | ((uint32_t)(uint8_t)((a + 1) * 7 + 3) << 8);
      bad_r += h[l] != exp;
      bad_w += h[256 + l] != (0xA000u + l);
    }
    printf("offset %d: read mismatches %d, write mismatches %d, %.1f cycles per dependent ds_read_u16\n", off, bad_r, bad_w, (double)cy / 4096);
  }
  return 0;
}
