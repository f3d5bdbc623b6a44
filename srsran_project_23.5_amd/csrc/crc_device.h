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

// Reads 32 message bits starting at absolute bit position `bit` of an MSB-first packed buffer (first bit -> bit 31): the two aligned
// dwords that hold bytes [bit / 8, bit / 8 + 5) -- the second one only when one of those bytes lies in it --, funnel-shifted and byte-swapped
// (five single-byte loads per word before).
__device__ __forceinline__ uint32_t crc_load32(const uint8_t* __restrict__ data, uint64_t bit)
{
  const uintptr_t a  = (uintptr_t)(data + (bit >> 3));
  const uint32_t  sh = (uint32_t)(bit & 7u), ab = (uint32_t)(a & 3u);
  const uint32_t* p4 = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
  const uint32_t  lo = p4[0];
  const uint32_t  hi = (ab != 0 || sh != 0) ? p4[1] : 0u;
  const uint32_t  w  = __builtin_bswap32(ab ? __builtin_amdgcn_alignbyte(hi, lo, ab) : lo);
  if (sh == 0)
    return w;
  const uint32_t b4 = (hi >> (8u * ab)) & 0xffu;
  return (w << sh) | (b4 >> (8u - sh));
}

// Byte table of a CRC polynomial of order >= 8 for crc_partial: tab8[b] = (b(x) x^order) mod P. 256 entries, filled by the caller's threads.
__device__ __forceinline__ void crc_build_table8(uint32_t* tab8, uint32_t poly, uint32_t order, int tid, int nthreads)
{
  const uint32_t top = 1u << order;
  for (int b = tid; b < 256; b += nthreads) {
    uint32_t reg = (uint32_t)b << (order - 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      reg <<= 1;
      reg ^= (reg & top) ? poly : 0u;
    }
    tab8[b] = reg & (top - 1u);
  }
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
crc_partial(const miphy_graph_tables* tab, int p, const uint8_t* __restrict__ data, uint64_t bit0, uint32_t nbits, int tid, int nthreads,
            const uint32_t* tab8 = nullptr /* crc_build_table8 of this polynomial (LDS): whole words then take four table steps instead of 32 bit steps */)
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
  // Sixteen words are fetched before any of them is consumed: one memory latency per sixteen words instead of one per word (a lane of the
  // transport-block checksum owns forty: ten round trips with groups of four were most of the kernel's 0.09 ms per 1024 transport blocks).
  constexpr int CRC_GROUP = 16;
  for (uint32_t wb = w0; wb < w1; wb += CRC_GROUP) {
    uint32_t v4[CRC_GROUP];
#pragma unroll
    for (int q = 0; q < CRC_GROUP; ++q)
      v4[q] = crc_load32(data, bit0 + 32ull * min(wb + q, w1 - 1u)); // (unconditional: a word behind the run repeats its last one and is not used)
#pragma unroll
    for (int q = 0; q < CRC_GROUP; ++q) {
      const uint32_t w = wb + q;
      if (w < w1) {
        const uint32_t rem = nbits - 32 * w;
        const int      len = rem < 32 ? (int)rem : 32;
        const uint32_t v   = v4[q];
        if (tab8 && len == 32) {
          reg &= top - 1u;
#pragma unroll
          for (int bi = 0; bi < 4; ++bi)
            reg = ((reg << 8) & (top - 1u)) ^ tab8[((reg >> (order - 8)) ^ (v >> (24 - 8 * bi))) & 0xffu];
        } else {
          for (int b = 0; b < len; ++b) {
            reg = (reg << 1) ^ (((v >> (31 - b)) & 1u) << order);
            reg ^= (reg & top) ? poly : 0u;
          }
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
