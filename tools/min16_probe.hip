// Does v_min_u16 / v_sub_u16 (VOP2) on gfx950 clear or keep the upper half of its destination register? (A 16-bit wrap of an LDS
// address is only usable as the address itself when the upper half is zero.)
// build: hipcc -O2 --offload-arch=gfx950 tools/min16_probe.hip -o tools/min16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o)
{
  unsigned a = 0x00010005u + threadIdx.x, b = 0xffff0003u, d = 0xdeadbeefu, e = 0xdeadbeefu;
  asm volatile("v_min_u16 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
  asm volatile("v_sub_u16 %0, %1, %2" : "+v"(e) : "v"(b), "v"(a));
  o[2 * threadIdx.x]     = d;
  o[2 * threadIdx.x + 1] = e;
}
int main()
{
  unsigned *d, h[8];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; ++i)
    printf("lane %d: v_min_u16(0x%04x, 0x0003) into 0xdeadbeef -> 0x%08x   v_sub_u16(0x0003, 0x%04x) into 0xdeadbeef -> 0x%08x\n", i, 5 + i, h[2 * i], 5 + i, h[2 * i + 1]);
  return 0;
}
