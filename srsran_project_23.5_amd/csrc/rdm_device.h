// Device-side pieces of the LDPC rate dematcher shared by rate_dematch_kernel (ldpc_ratematch.hip) and the LDPC decoder that
// dematches while it loads its codeblock (ldpc_decode_pk.hip).
// Behaviour contract: lib/phy/upper/channel_coding/ldpc/ldpc_rate_dematcher_impl.cpp:43-254.
#pragma once
#include "miphy_internal.h"

namespace {

struct rm_geom {
  int N, Ncb, F, f0, f1, L, k0, r0, Kq, E, mod;
};

__device__ __forceinline__ rm_geom make_geom(const miphy_ldpc_rdm_desc& d)
{
  rm_geom g;
  const int bgK = (d.bg == 1) ? 22 : 10, nshort = (d.bg == 1) ? 66 : 50;
  const int Z   = d.Z;
  g.N           = nshort * Z;
  g.Ncb         = (d.Nref > 0 && (int)d.Nref < g.N) ? (int)d.Nref : g.N;
  // TS 38.212 Table 5.4.2.1-2 (rate_matcher_impl.cpp:64-94): k0 = floor(k0num * Ncb / N) * Z.
  const int num = (d.bg == 1) ? ((d.rv == 0) ? 0 : (d.rv == 1) ? 17 : (d.rv == 2) ? 33 : 56)
                              : ((d.rv == 0) ? 0 : (d.rv == 1) ? 13 : (d.rv == 2) ? 25 : 43);
  g.k0          = (g.Ncb == g.N) ? num * Z : (int)(((long long)num * g.Ncb) / g.N) * Z; // (full buffer: no 64-bit division -- a few hundred instructions per lane)
  g.f1          = (bgK - 2) * Z;
  g.F           = d.nof_filler_bits;
  g.f0          = g.f1 - g.F;
  g.L           = g.Ncb - g.F;
  int k0p       = (g.k0 >= g.f0 && g.k0 < g.f1) ? g.f1 : g.k0;
  g.r0          = (k0p < g.f0) ? k0p : k0p - g.F;
  g.E           = (int)d.E;
  g.mod         = d.mod;
  g.Kq          = g.E / g.mod;
  return g;
}

struct image_access { // LDS image laid out like the output buffer (single-pass geometry only), see rate_dematch_kernel
  const int8_t* img;
  int           r0, f0, F, jbase, f1, gapcut;
  __device__ __forceinline__ int slot(int j) const { return j - jbase - ((j >= f1) ? gapcut : 0); }
  __device__ __forceinline__ int operator()(int i) const
  {
    const int r = r0 + i;
    return img[slot((r < f0) ? r : r + F)];
  }
};

// De-interleaving stage: input byte k = p * MOD + q is the LLR of rank index i = q * Kq + p, which goes to image slot
// rank + (rank >= f0 ? adj : 0) - jbase (single-pass geometry: see image_access). A lane takes 16 consecutive input bytes;
// for the power-of-two modulation orders these are 16 / MOD whole symbols, so that per bit plane q the slot is one add away
// from a per-plane scalar and the symbols of a lane land on consecutive bytes.
// 16 input bytes at byte offset 16 * v of a stream with ANY alignment: one dword-aligned 16-byte load plus the following dword,
// funnel-shifted by the (uniform) misalignment mb = address & 3. The caller keeps 16 * v + 20 - mb <= E.
struct __attribute__((packed, aligned(4))) rdm_u4 {
  uint32_t x, y, z, w;
};
__device__ __forceinline__ uint4 load_in16(const int8_t* __restrict__ in, int mb, int v)
{
  const uint32_t* p4 = reinterpret_cast<const uint32_t*>(in - mb) + 4 * v;
  const rdm_u4    a  = *reinterpret_cast<const rdm_u4*>(p4);
  if (mb == 0)
    return make_uint4(a.x, a.y, a.z, a.w);
  const uint32_t e = p4[4];
  return make_uint4(__builtin_amdgcn_alignbyte(a.y, a.x, mb), __builtin_amdgcn_alignbyte(a.z, a.y, mb), __builtin_amdgcn_alignbyte(a.w, a.z, mb),
                    __builtin_amdgcn_alignbyte(e, a.w, mb));
}

template <int MOD>
__device__ __forceinline__ void stage_vector(const image_access& img, int8_t* __restrict__ lds, const rm_geom& g, int adj, int v, const uint4& x)
{
  const uint32_t w[4] = {x.x, x.y, x.z, x.w};
  if (MOD == 6) {
    int p = (16 * v) / 6;
    int q = 16 * v - p * 6;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int r                                  = g.r0 + q * g.Kq + p;
      lds[r - img.jbase + ((r >= g.f0) ? adj : 0)] = (int8_t)(w[b >> 2] >> (8 * (b & 3)));
      if (++q == 6) {
        q = 0;
        ++p;
      }
    }
  } else {
    constexpr int SYM = 16 / MOD; // symbols per 16-byte vector
    const int     p0  = v * SYM;
#pragma unroll
    for (int q = 0; q < MOD; ++q) {
      const int r0q = g.r0 + q * g.Kq - img.jbase; // uniform
      const int lim = g.f0 - g.r0 - q * g.Kq;      // uniform: symbols p >= lim of this plane lie behind the filler gap
#pragma unroll
      for (int sy = 0; sy < SYM; ++sy) {
        const int p = p0 + sy;
        const int b = sy * MOD + q;
        lds[r0q + p + ((p >= lim) ? adj : 0)] = (int8_t)(w[b >> 2] >> (8 * (b & 3)));
      }
    }
  }
}

} // namespace
