// Probe: VALU issue cost on gfx950 in SHADER CYCLES (s_memtime), independent of any assumed clock: cycles a SIMD spends per wave64
// instruction when W waves per SIMD run the same independent-instruction stream, W = 1, 2, 3, 4, 8, for the instruction classes the
// LDPC decoder is made of (VOP3P packed 16-bit, v_perm_b32, plain VOP2 integer) and for the decoder's measured instruction mix.
// MI355X_MICROARCH.md:54,473 gives 4 cycles for a lone wave and 2 cycles per wave64 VALU instruction once >= 2 waves are resident;
// this program is the measurement bench.py's `roofline_valu` cites (profiles/r02_valu_probe.txt).
// build: hipcc -O2 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define BODY8(ASM)                                        \
  asm volatile(ASM "\n" : "+v"(r[0]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[1]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[2]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[3]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[4]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[5]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[6]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[7]) : "v"(a), "v"(b));

#define DEF_KERNEL(NAME, ...)                                                                      \
  __global__ void __launch_bounds__(256) NAME(unsigned long long* cyc, unsigned* sink, int iters) \
  {                                                                                                \
    unsigned r[8], a = threadIdx.x * 2654435761u, b = threadIdx.x + 77u;                           \
    for (int i = 0; i < 8; ++i)                                                                    \
      r[i] = a + i;                                                                                \
    __syncthreads();                                                                               \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                    \
    for (int it = 0; it < iters; ++it) {                                                           \
      __VA_ARGS__                                                                                  \
    }                                                                                              \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                    \
    unsigned acc = 0;                                                                              \
    for (int i = 0; i < 8; ++i)                                                                    \
      acc ^= r[i];                                                                                 \
    if (acc == 0x12345678)                                                                         \
      sink[0] = acc;                                                                               \
    if ((threadIdx.x & 63) == 0)                                                                   \
      cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;                            \
  }

// the same instruction as ONE dependent chain (every instruction reads the result of the one before it) and as two interleaved chains
#define DEP8(ASM)                                         \
  asm volatile(ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" ASM "\n" : "+v"(r[0]) : "v"(a), "v"(b));
#define DEP8x2(ASM)                                       \
  asm volatile(ASM "\n" : "+v"(r[0]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[1]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[0]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[1]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[0]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[1]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[0]) : "v"(a), "v"(b)); \
  asm volatile(ASM "\n" : "+v"(r[1]) : "v"(a), "v"(b));
#define DEPN(ASM, I) asm volatile(ASM "\n" : "+v"(r[I]) : "v"(a), "v"(b));
#define DEP12x3(ASM) DEPN(ASM, 0) DEPN(ASM, 1) DEPN(ASM, 2) DEPN(ASM, 0) DEPN(ASM, 1) DEPN(ASM, 2) DEPN(ASM, 0) DEPN(ASM, 1) DEPN(ASM, 2) DEPN(ASM, 0) DEPN(ASM, 1) DEPN(ASM, 2)
#define DEP8x4(ASM) DEPN(ASM, 0) DEPN(ASM, 1) DEPN(ASM, 2) DEPN(ASM, 3) DEPN(ASM, 0) DEPN(ASM, 1) DEPN(ASM, 2) DEPN(ASM, 3)
#define DEP12x6(ASM) DEPN(ASM, 0) DEPN(ASM, 1) DEPN(ASM, 2) DEPN(ASM, 3) DEPN(ASM, 4) DEPN(ASM, 5) DEPN(ASM, 0) DEPN(ASM, 1) DEPN(ASM, 2) DEPN(ASM, 3) DEPN(ASM, 4) DEPN(ASM, 5)
#define D120x3(ASM) DEP12x3(ASM) DEP12x3(ASM) DEP12x3(ASM) DEP12x3(ASM) DEP12x3(ASM) DEP12x3(ASM) DEP12x3(ASM) DEP12x3(ASM) DEP12x3(ASM) DEP12x3(ASM)
#define D120x6(ASM) DEP12x6(ASM) DEP12x6(ASM) DEP12x6(ASM) DEP12x6(ASM) DEP12x6(ASM) DEP12x6(ASM) DEP12x6(ASM) DEP12x6(ASM) DEP12x6(ASM) DEP12x6(ASM)
#define D128x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM) DEP8x4(ASM)
#define D128(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM) DEP8(ASM)
#define D128x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM) DEP8x2(ASM)
#define B32(ASM) BODY8(ASM) BODY8(ASM) BODY8(ASM) BODY8(ASM)
#define B128(ASM) B32(ASM) B32(ASM) B32(ASM) B32(ASM)
// 128 instructions per loop iteration in the single-instruction kernels (loop overhead < 3 %), 32 in the mix
DEF_KERNEL(k_add32, B128("v_add_u32 %0, %1, %0"))
DEF_KERNEL(k_add32_dep, D128("v_add_u32 %0, %1, %0"))
DEF_KERNEL(k_add32_dep2, D128x2("v_add_u32 %0, %1, %0"))
DEF_KERNEL(k_pk_add_dep, D128("v_pk_add_u16 %0, %1, %0"))
DEF_KERNEL(k_pk_add_dep2, D128x2("v_pk_add_u16 %0, %1, %0"))
DEF_KERNEL(k_pk_add_dep3, D120x3("v_pk_add_u16 %0, %1, %0"))
DEF_KERNEL(k_pk_add_dep4, D128x4("v_pk_add_u16 %0, %1, %0"))
DEF_KERNEL(k_pk_add_dep6, D120x6("v_pk_add_u16 %0, %1, %0"))
DEF_KERNEL(k_add32_dep3, D120x3("v_add_u32 %0, %1, %0"))
DEF_KERNEL(k_add32_dep4, D128x4("v_add_u32 %0, %1, %0"))
// operands from scalar registers (the decoder's shifts and column offsets are SGPRs) and the condition-code forms
DEF_KERNEL(k_add32_sgpr, B128("v_add_u32 %0, s20, %0"))
DEF_KERNEL(k_pk_add_sgpr, B128("v_pk_add_u16 %0, s20, %0"))
DEF_KERNEL(k_min_u32_sgpr, B128("v_min_u32 %0, s20, %0"))
DEF_KERNEL(k_cndmask_sgpr, B128("v_cndmask_b32_e64 %0, %1, %0, s[22:23]"))
DEF_KERNEL(k_cmp_cnd, B128("v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc"))
DEF_KERNEL(k_min_u16, B128("v_min_u16 %0, %1, %0"))
DEF_KERNEL(k_sub_u16, B128("v_sub_u16 %0, %1, %0"))
DEF_KERNEL(k_min_u32, B128("v_min_u32 %0, %1, %0"))
DEF_KERNEL(k_add32_lit, B128("v_add_u32 %0, 0x12345, %0"))
DEF_KERNEL(k_add32_e64, B128("v_add_u32_e64 %0, %1, %0"))
DEF_KERNEL(k_xor, B128("v_xor_b32 %0, %1, %0"))
DEF_KERNEL(k_mov, B128("v_mov_b32 %0, %1"))
DEF_KERNEL(k_addf, B128("v_add_f32 %0, %1, %0"))
DEF_KERNEL(k_fmac, B128("v_fmac_f32 %0, %1, %2"))
DEF_KERNEL(k_fma32, B128("v_fma_f32 %0, %1, %2, %0"))
DEF_KERNEL(k_add16, B128("v_add_u16 %0, %1, %0"))
DEF_KERNEL(k_min16, B128("v_min_i16 %0, %1, %0"))
DEF_KERNEL(k_min32, B128("v_min_i32 %0, %1, %0"))
DEF_KERNEL(k_lshl, B128("v_lshlrev_b32 %0, 3, %0"))
DEF_KERNEL(k_cndmask, B128("v_cndmask_b32 %0, %1, %0, vcc"))
DEF_KERNEL(k_add3, B128("v_add3_u32 %0, %1, %2, %0"))
DEF_KERNEL(k_and_or, B128("v_and_or_b32 %0, %1, %2, %0"))
DEF_KERNEL(k_lshl_add, B128("v_lshl_add_u32 %0, %1, 2, %0"))
DEF_KERNEL(k_med3_i32, B128("v_med3_i32 %0, %1, %2, %0"))
DEF_KERNEL(k_sdwa, B128("v_add_u32_sdwa %0, %1, %0 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:DWORD"))
DEF_KERNEL(k_dpp, B128("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf"))
DEF_KERNEL(k_pk_add, B128("v_pk_add_u16 %0, %1, %0"))
DEF_KERNEL(k_pk_min, B128("v_pk_min_i16 %0, %1, %0"))
DEF_KERNEL(k_pk_mad, B128("v_pk_mad_u16 %0, %1, %2, %0"))
DEF_KERNEL(k_perm, B128("v_perm_b32 %0, %1, %0, %2"))
DEF_KERNEL(k_bfi, B128("v_bfi_b32 %0, %1, %2, %0"))
DEF_KERNEL(k_bfe, B128("v_bfe_i32 %0, %0, 3, 8"))
DEF_KERNEL(k_med3_i16, B128("v_med3_i16 %0, %1, %2, %0"))
DEF_KERNEL(k_min3_i16, B128("v_min3_i16 %0, %1, %2, %0"))
DEF_KERNEL(k_sad_u8, B128("v_sad_u8 %0, %1, %2, %0"))
DEF_KERNEL(k_dot4, B128("v_dot4_i32_i8 %0, %1, %2, %0"))
// the packed decoder's mix per 32: 16 VOP3P (add/sub/min/max/mad), 4 v_perm_b32, 12 VOP2/VOP3 32-bit integer
DEF_KERNEL(k_mix, BODY8("v_pk_add_u16 %0, %1, %0") BODY8("v_pk_min_i16 %0, %1, %0")
           asm volatile("v_perm_b32 %0, %4, %0, %5\n v_perm_b32 %1, %4, %1, %5\n v_perm_b32 %2, %4, %2, %5\n v_perm_b32 %3, %4, %3, %5\n"
                        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(a), "v"(b));
           asm volatile("v_xor_b32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_and_b32 %2, %4, %2\n v_min_u32 %3, %4, %3\n"
                        : "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a));
           BODY8("v_add_u32 %0, %1, %0"))

template <typename K>
void run(const char* name, K kern, unsigned long long* d_cyc, unsigned* d_sink, int instr_per_iter)
{
  const int iters = instr_per_iter >= 120 ? 500 : 2000;
  printf("%-26s", name);
  for (int W : {1, 2, 3, 4, 8}) {
    // W workgroups of 4 waves per CU -> W waves per SIMD (256 CUs)
    const int grid = 256 * W;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_cyc, d_sink, 10);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_cyc, d_sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h)
      sum += (double)v;
    const double per_wave = sum / h.size();                         // cycles one wave needed for its stream
    const double per_simd = per_wave / ((double)iters * instr_per_iter) / W; // SIMD cycles per wave64 instruction with W waves sharing it
    printf("  W=%d: %5.2f", W, per_simd);
  }
  printf("   (SIMD cycles per wave64 instruction)\n");
}

int main()
{
  unsigned long long* d_cyc;
  unsigned*           d_sink;
  hipMalloc(&d_cyc, 256 * 8 * 4 * 8);
  hipMalloc(&d_sink, 64);
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  printf("device %s, %d CUs, clock %d MHz; W = waves per SIMD\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate / 1000);
  run("v_add_u32 e32 (VOP2)", k_add32, d_cyc, d_sink, 128);
  run("v_add_u32 e64 (VOP3)", k_add32_e64, d_cyc, d_sink, 128);
  run("v_xor_b32 (VOP2)", k_xor, d_cyc, d_sink, 128);
  run("v_mov_b32 (VOP1)", k_mov, d_cyc, d_sink, 128);
  run("v_add_f32 (VOP2)", k_addf, d_cyc, d_sink, 128);
  run("v_fmac_f32 (VOP2)", k_fmac, d_cyc, d_sink, 128);
  run("v_fma_f32 (VOP3)", k_fma32, d_cyc, d_sink, 128);
  run("v_add_u16 (VOP2)", k_add16, d_cyc, d_sink, 128);
  run("v_min_i16 (VOP2)", k_min16, d_cyc, d_sink, 128);
  run("v_min_i32 (VOP2)", k_min32, d_cyc, d_sink, 128);
  run("v_lshlrev_b32 (VOP2)", k_lshl, d_cyc, d_sink, 128);
  run("v_cndmask_b32 (VOP2)", k_cndmask, d_cyc, d_sink, 128);
  run("v_add3_u32 (VOP3)", k_add3, d_cyc, d_sink, 128);
  run("v_and_or_b32 (VOP3)", k_and_or, d_cyc, d_sink, 128);
  run("v_lshl_add_u32 (VOP3)", k_lshl_add, d_cyc, d_sink, 128);
  run("v_med3_i32 (VOP3)", k_med3_i32, d_cyc, d_sink, 128);
  run("v_add_u32_sdwa", k_sdwa, d_cyc, d_sink, 128);
  run("v_add_u32_dpp", k_dpp, d_cyc, d_sink, 128);
  run("v_pk_add_u16 (VOP3P)", k_pk_add, d_cyc, d_sink, 128);
  run("v_pk_min_i16 (VOP3P)", k_pk_min, d_cyc, d_sink, 128);
  run("v_pk_mad_u16 (VOP3P)", k_pk_mad, d_cyc, d_sink, 128);
  run("v_perm_b32 (VOP3)", k_perm, d_cyc, d_sink, 128);
  run("v_bfi_b32 (VOP3)", k_bfi, d_cyc, d_sink, 128);
  run("v_bfe_i32 (VOP3)", k_bfe, d_cyc, d_sink, 128);
  run("v_med3_i16 (VOP3)", k_med3_i16, d_cyc, d_sink, 128);
  run("v_min3_i16 (VOP3)", k_min3_i16, d_cyc, d_sink, 128);
  run("v_sad_u8 (VOP3)", k_sad_u8, d_cyc, d_sink, 128);
  run("v_dot4_i32_i8 (VOP3P)", k_dot4, d_cyc, d_sink, 128);
  run("decoder mix 16pk/4perm/12i32", k_mix, d_cyc, d_sink, 32);
  run("v_add_u32 v, SGPR, v", k_add32_sgpr, d_cyc, d_sink, 128);
  run("v_add_u32 v, literal, v", k_add32_lit, d_cyc, d_sink, 128);
  run("v_pk_add_u16 v, SGPR, v", k_pk_add_sgpr, d_cyc, d_sink, 128);
  run("v_min_u32 v, SGPR, v", k_min_u32_sgpr, d_cyc, d_sink, 128);
  run("v_cndmask_b32 e64, SGPR pair", k_cndmask_sgpr, d_cyc, d_sink, 128);
  run("v_cmp + v_cndmask (per PAIR)", k_cmp_cnd, d_cyc, d_sink, 128);
  run("v_min_u32 (VOP2)", k_min_u32, d_cyc, d_sink, 128);
  run("v_min_u16 (VOP2)", k_min_u16, d_cyc, d_sink, 128);
  run("v_sub_u16 (VOP2)", k_sub_u16, d_cyc, d_sink, 128);
  run("v_add_u32, ONE dependent chain", k_add32_dep, d_cyc, d_sink, 128);
  run("v_add_u32, two chains", k_add32_dep2, d_cyc, d_sink, 128);
  run("v_pk_add_u16, ONE dep. chain", k_pk_add_dep, d_cyc, d_sink, 128);
  run("v_pk_add_u16, two chains", k_pk_add_dep2, d_cyc, d_sink, 128);
  run("v_pk_add_u16, three chains", k_pk_add_dep3, d_cyc, d_sink, 120);
  run("v_pk_add_u16, four chains", k_pk_add_dep4, d_cyc, d_sink, 128);
  run("v_pk_add_u16, six chains", k_pk_add_dep6, d_cyc, d_sink, 120);
  run("v_add_u32, three chains", k_add32_dep3, d_cyc, d_sink, 120);
  run("v_add_u32, four chains", k_add32_dep4, d_cyc, d_sink, 128);
  return 0;
}
