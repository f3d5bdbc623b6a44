// Probe: issue cost (cycles per wave64 instruction per SIMD) of the packed-16 VALU instructions the LDPC decoder uses.
// build: hipcc -O2 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEF_KERNEL(NAME, ASM)                                                                             \
  __global__ void __launch_bounds__(256) NAME(unsigned* out, int iters)                                  \
  {                                                                                                       \
    unsigned r[8], a = threadIdx.x * 2654435761u, b = threadIdx.x + 77u;                                  \
    for (int i = 0; i < 8; ++i)                                                                           \
      r[i] = a + i;                                                                                       \
    for (int it = 0; it < iters; ++it) {                                                                  \
      _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                       \
      {                                                                                                   \
        asm volatile(ASM "\n" : "+v"(r[0]) : "v"(a), "v"(b));                                             \
        asm volatile(ASM "\n" : "+v"(r[1]) : "v"(a), "v"(b));                                             \
        asm volatile(ASM "\n" : "+v"(r[2]) : "v"(a), "v"(b));                                             \
        asm volatile(ASM "\n" : "+v"(r[3]) : "v"(a), "v"(b));                                             \
        asm volatile(ASM "\n" : "+v"(r[4]) : "v"(a), "v"(b));                                             \
        asm volatile(ASM "\n" : "+v"(r[5]) : "v"(a), "v"(b));                                             \
        asm volatile(ASM "\n" : "+v"(r[6]) : "v"(a), "v"(b));                                             \
        asm volatile(ASM "\n" : "+v"(r[7]) : "v"(a), "v"(b));                                             \
      }                                                                                                   \
    }                                                                                                     \
    unsigned acc = 0;                                                                                     \
    for (int i = 0; i < 8; ++i)                                                                           \
      acc ^= r[i];                                                                                        \
    if (acc == 0x12345678)                                                                                \
      out[0] = acc;                                                                                       \
  }

DEF_KERNEL(k_add32, "v_add_u32 %0, %1, %0")
DEF_KERNEL(k_pk_add, "v_pk_add_u16 %0, %1, %0")
DEF_KERNEL(k_pk_sub, "v_pk_sub_i16 %0, %1, %0")
DEF_KERNEL(k_pk_min, "v_pk_min_i16 %0, %1, %0")
DEF_KERNEL(k_pk_max, "v_pk_max_i16 %0, %1, %0")
DEF_KERNEL(k_pk_mad, "v_pk_mad_u16 %0, %1, %2, %0")
DEF_KERNEL(k_pk_mul, "v_pk_mul_lo_u16 %0, %1, %0")
DEF_KERNEL(k_pk_ashr, "v_pk_ashrrev_i16 %0, 3, %0")
DEF_KERNEL(k_pk_lshl, "v_pk_lshlrev_b16 %0, 3, %0")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %1, %0, %2")
DEF_KERNEL(k_xor, "v_xor_b32 %0, %1, %0")
DEF_KERNEL(k_bfi, "v_bfi_b32 %0, %1, %2, %0")
DEF_KERNEL(k_min_u32, "v_min_u32 %0, %1, %0")
DEF_KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %1, %2, %0")
DEF_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %1, %0")
DEF_KERNEL(k_dep_pk, "v_pk_add_u16 %0, %0, %0\n v_pk_min_i16 %0, %0, %1") /* two dependent packed ops incl. hazard nops? none inserted in asm */

template <typename K>
void run(const char* name, K kern, unsigned* d, int per_asm = 1)
{
  const int  iters = 4000, grid = 256 * 3; // 3 WGs of 4 waves per CU -> 3 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = 3.0 * iters * 32 * per_asm;
  printf("%-16s %.3f ms -> %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e6 * 2.4 / instr_per_simd);
}

int main()
{
  unsigned* d;
  hipMalloc(&d, 64);
  run("v_add_u32", k_add32, d);
  run("v_pk_add_u16", k_pk_add, d);
  run("v_pk_sub_i16", k_pk_sub, d);
  run("v_pk_min_i16", k_pk_min, d);
  run("v_pk_max_i16", k_pk_max, d);
  run("v_pk_mad_u16", k_pk_mad, d);
  run("v_pk_mul_lo_u16", k_pk_mul, d);
  run("v_pk_ashrrev_i16", k_pk_ashr, d);
  run("v_pk_lshlrev_b16", k_pk_lshl, d);
  run("v_perm_b32", k_perm, d);
  run("v_xor_b32", k_xor, d);
  run("v_bfi_b32", k_bfi, d);
  run("v_min_u32", k_min_u32, d);
  run("v_mad_u32_u24", k_mad_u24, d);
  run("v_mul_lo_u32", k_mul_lo, d);
  run("dep pk pair", k_dep_pk, d, 2);
  return 0;
}
