// Probe: LDS pipeline cost (cycles per wave64 instruction per CU) of the narrow DS operations the LDPC decoder uses.
// build: hipcc -O2 --offload-arch=gfx950 tools/lds_probe.hip -o tools/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(192) k(int* out, int iters)
{
  extern __shared__ int8_t lds[];
  const unsigned           tid = threadIdx.x;
  for (unsigned i = tid; i < 39000; i += 192)
    lds[i] = (int8_t)i;
  __syncthreads();
  unsigned a   = (MODE == 2 || MODE == 5) ? tid * 4 : (MODE == 1 || MODE == 4) ? tid * 2 : tid; // consecutive lanes, no conflicts
  unsigned acc = 0, v0, v1, v2, v3;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0)
      asm volatile("ds_read_i8 %0, %4\n ds_read_i8 %1, %4 offset:256\n ds_read_i8 %2, %4 offset:512\n ds_read_i8 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a));
    if (MODE == 1)
      asm volatile("ds_read_u16 %0, %4\n ds_read_u16 %1, %4 offset:512\n ds_read_u16 %2, %4 offset:1024\n ds_read_u16 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a));
    if (MODE == 2)
      asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:1024\n ds_read_b32 %2, %4 offset:2048\n ds_read_b32 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a));
    if (MODE == 3) {
      v0 = v1 = v2 = v3 = it;
      asm volatile("ds_write_b8 %4, %0\n ds_write_b8 %4, %1 offset:256\n ds_write_b8_d16_hi %4, %2 offset:512\n ds_write_b8_d16_hi %4, %3 offset:768\n s_waitcnt lgkmcnt(0)"
                   :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(a) : "memory");
    }
    if (MODE == 4) {
      v0 = v1 = v2 = v3 = it;
      asm volatile("ds_write_b16 %4, %0\n ds_write_b16 %4, %1 offset:512\n ds_write_b16 %4, %2 offset:1024\n ds_write_b16 %4, %3 offset:1536\n s_waitcnt lgkmcnt(0)"
                   :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(a) : "memory");
    }
    if (MODE == 5) {
      v0 = v1 = v2 = v3 = it;
      asm volatile("ds_write_b32 %4, %0\n ds_write_b32 %4, %1 offset:1024\n ds_write_b32 %4, %2 offset:2048\n ds_write_b32 %4, %3 offset:3072\n s_waitcnt lgkmcnt(0)"
                   :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(a) : "memory");
    }
    if (MODE == 6) // unaligned u16 reads (odd byte address)
      asm volatile("ds_read_u16 %0, %4 offset:1\n ds_read_u16 %1, %4 offset:513\n ds_read_u16 %2, %4 offset:1025\n ds_read_u16 %3, %4 offset:1537\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a));
    acc += v0 + v1 + v2 + v3;
  }
  if (acc == 0x12345678)
    out[0] = acc;
  if (MODE == 6 && blockIdx.x == 0 && tid < 4)
    out[1 + tid] = v0; // lanes 0..3: bytes (1,2),(3,4).. of the pattern -> shows whether odd addresses are honoured
}

template <int MODE>
void run(const char* name, int* d)
{
  const int   iters = 20000, grid = 256 * 4;
  hipEvent_t  e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 39232);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(192), 39232, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(192), 39232, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per CU: 4 WGs x 3 waves x iters x 4 instructions
  const double instr_per_cu = 12.0 * iters * 4;
  printf("%-22s %.3f ms -> %.2f ns per wave64 instruction per CU (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / instr_per_cu,
         ms * 1e6 / instr_per_cu * 2.4);
}

int main()
{
  int* d;
  hipMalloc(&d, 64);
  hipMemset(d, 0, 64);
  run<0>("ds_read_i8", d);
  run<1>("ds_read_u16", d);
  run<2>("ds_read_b32", d);
  run<3>("ds_write_b8(+d16_hi)", d);
  run<4>("ds_write_b16", d);
  run<5>("ds_write_b32", d);
  run<6>("ds_read_u16 odd addr", d);
  int h[5];
  hipMemcpy(h, d, 20, hipMemcpyDeviceToHost);
  printf("odd-address u16 reads, lanes 0..3: %04x %04x %04x %04x (pattern byte i = i: expect 0201 0403 0605 0807)\n", h[1], h[2], h[3], h[4]);
  return 0;
}
