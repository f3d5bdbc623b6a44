// Device-side parallel CRC (zero initial state, MSB-first, no reflection) shared by several kernels.
// Behaviour contract: lib/phy/upper/channel_coding/crc_calculator_lut_impl.cpp:33-153.
#pragma once
#include "miphy_internal.h"

__device__ __forceinline__ uint32_t crc_gf2_mulmod(uint32_t a, uint32_t b, uint32_t poly, uint32_t order)
{
  uint32_t       r   = 0;
  const uint32_t top = 1u << order;
  for (int k = (int)order - 1; k >= 0; --k) {
    r <<= 1;
    r ^= (r & top) ? poly : 0u;
    r ^= ((b >> k) & 1u) ? a : 0u;
  }
  return r;
}

// Reads 32 message bits starting at absolute bit position `bit` of an MSB-first packed buffer (first bit -> bit 31).
__device__ __forceinline__ uint32_t crc_load32(const uint8_t* __restrict__ data, uint64_t bit)
{
  const uint64_t byte = bit >> 3;
  const int      sh   = (int)(bit & 7);
  uint64_t       v    = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k)
    v = (v << 8) | data[byte + k];
  return (uint32_t)(v >> (8 - sh));
}

// x^(32*k) mod poly: one table entry for k < 320, one product of two entries up to k < 65536 (2 Mbit), square-and-multiply beyond.
__device__ __forceinline__ uint32_t crc_pow32(const miphy_graph_tables* tab, int p, uint32_t k, uint32_t poly, uint32_t order)
{
  if (k < 320)
    return tab->crc_pow32[p][k];
  if (k < 65536)
    return crc_gf2_mulmod(tab->crc_pow32[p][k & 255u], tab->crc_pow32_hi[p][k >> 8], poly, order);
  uint32_t r = 1;
  for (int b = 0; k != 0; ++b, k >>= 1)
    if (k & 1u)
      r = crc_gf2_mulmod(r, tab->crc_pow2[p][b], poly, order);
  return r;
}

// CRC of `nbits` bits starting at bit `bit0`, computed cooperatively by the `nthreads` threads of the caller's group
// (tid in [0,nthreads)). Returns this thread's partial remainder: XOR-reduce over the group gives the checksum.
// NOTE: reads up to 4 bytes past the last message byte (callers pad their buffers).
__device__ __forceinline__ uint32_t
crc_partial(const miphy_graph_tables* tab, int p, const uint8_t* __restrict__ data, uint64_t bit0, uint32_t nbits, int tid, int nthreads)
{
  const uint32_t poly = tab->crc_poly[p], order = tab->crc_order[p], top = 1u << order;
  const uint32_t nwords = (nbits + 31) >> 5;            // last word may be partial
  const uint32_t per    = (nwords + nthreads - 1) / nthreads;
  const uint32_t w0     = (uint32_t)tid * per;
  if (w0 >= nwords)
    return 0;
  const uint32_t w1  = min(w0 + per, nwords);
  uint32_t       reg = 0;
  uint32_t       bits_done_end = 0; // bits consumed up to the end of my run
  // Four words are fetched before any of them is consumed: one memory latency per four words instead of one per word.
  for (uint32_t wb = w0; wb < w1; wb += 4) {
    uint32_t v4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      v4[q] = (wb + q < w1) ? crc_load32(data, bit0 + 32ull * (wb + q)) : 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t w = wb + q;
      if (w < w1) {
        const uint32_t rem = nbits - 32 * w;
        const int      len = rem < 32 ? (int)rem : 32;
        const uint32_t v   = v4[q];
        for (int b = 0; b < len; ++b) {
          reg = (reg << 1) ^ (((v >> (31 - b)) & 1u) << order);
          reg ^= (reg & top) ? poly : 0u;
        }
        bits_done_end = 32 * w + len;
      }
    }
  }
  reg &= top - 1u;
  // Weight: x^(nbits - bits_done_end).
  const uint32_t after = nbits - bits_done_end;
  if (after) {
    reg = crc_gf2_mulmod(reg, crc_pow32(tab, p, after >> 5, poly, order), poly, order);
    for (uint32_t b = 0; b < (after & 31u); ++b) {
      reg <<= 1;
      reg ^= (reg & top) ? poly : 0u;
    }
  }
  return reg;
}
