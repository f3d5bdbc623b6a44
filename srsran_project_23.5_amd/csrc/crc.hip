// Batched CRC calculator (srsran::crc_calculator, include/srsran/phy/upper/channel_coding/crc_calculator.h:45-67).
#include "crc_device.h"

#ifndef CRC_WIDE_BELOW
#define CRC_WIDE_BELOW 512
#endif
#ifndef CRC_NARROW_THREADS
#define CRC_NARROW_THREADS 512 // 1024 transport blocks of 40 kB: 57.9 us at 256, 49.0 at 512, 60.5 at 1024 threads
#endif
namespace {
__global__ void __launch_bounds__(1024)
crc_kernel(const miphy_crc_desc* __restrict__ descs, const miphy_graph_tables* __restrict__ tab, const uint8_t* __restrict__ data, uint32_t* __restrict__ out)
{
  __shared__ uint32_t red[16];
  __shared__ uint32_t tab8[256];
  const miphy_crc_desc d     = descs[blockIdx.x];
  const uint32_t       order = tab->crc_order[d.poly];
  const bool           byte_table = order >= 8 && d.nbits >= 64u * blockDim.x; // long messages: the table pays for its 256 x 8 steps
  if (byte_table) {
    crc_build_table8(tab8, tab->crc_poly[d.poly], order, threadIdx.x, blockDim.x);
    __syncthreads();
  }
  uint32_t part = crc_partial(tab, (int)d.poly, data, d.bit_offset, d.nbits, threadIdx.x, blockDim.x, byte_table ? tab8 : nullptr);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
    part ^= __shfl_xor(part, off);
  if ((threadIdx.x & 63) == 0)
    red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t r = 0;
    for (unsigned w = 0; w < blockDim.x / 64; ++w)
      r ^= red[w];
    out[blockIdx.x] = r;
  }
}
} // namespace

extern "C" int miphy_crc_batch(miphy_ctx*            ctx,
                               const miphy_crc_desc* descs,
                               int                   descs_on_device,
                               uint32_t              n,
                               const uint8_t*        data,
                               uint32_t*             checksums,
                               void*                 stream)
{
  MIPHY_REQUIRE(ctx && descs && data && checksums, "miphy_crc_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  if (!descs_on_device)
    for (uint32_t i = 0; i < n; ++i)
      MIPHY_REQUIRE(descs[i].poly <= MIPHY_CRC11, "crc: desc %u: invalid polynomial %u", i, descs[i].poly);
  hipStream_t s       = (hipStream_t)stream;
  const void* d_descs = nullptr;
  int         rc      = miphy_stage_descs(ctx, descs, descs_on_device, sizeof(miphy_crc_desc) * (size_t)n, s, &d_descs);
  if (rc)
    return rc;
  // Few, long messages (transport blocks): 1024 threads each; many short ones: 256 are plenty.
  hipLaunchKernelGGL(crc_kernel, dim3(n), dim3(n <= CRC_WIDE_BELOW ? 1024 : CRC_NARROW_THREADS), 0, s, (const miphy_crc_desc*)d_descs, ctx->d_tables, data, checksums);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
