// Modulation mapper shared by the PDSCH modulator (pdsch_mod.hip) and the EVM computation of the PUSCH demodulator (pusch_demod.hip).
// Behaviour contract: lib/phy/upper/channel_modulation/modulation_mapper_impl.cpp:31-146.
#pragma once
#include "miphy_internal.h"

// Constellation point of `mod` scrambled bits, TS 38.211 5.1 (b0 = first bit = most significant bit of the reference's table index).
__device__ __forceinline__ float2 map_symbol(int mod, const uint32_t bits /* bit t = t-th bit of the symbol */, unsigned sym_idx)
{
  if (mod == 1) {
    const float v = 0.70710678118654752440f;
    const float s = (bits & 1u) ? -v : v;
    return make_float2((sym_idx & 1u) ? -s : s, s);
  }
  int       lr = 0, li = 0;
  const int h  = mod >> 1;
  for (int j = h - 1; j >= 0; --j) {
    const int sr = 1 - 2 * (int)((bits >> (2 * j)) & 1u), si = 1 - 2 * (int)((bits >> (2 * j + 1)) & 1u);
    const int w  = 1 << (h - 1 - j);
    lr           = sr * (w - lr);
    li           = si * (w - li);
  }
  const float avg = (mod == 2) ? 2.f : (mod == 4) ? 10.f : (mod == 6) ? 42.f : 170.f;
  const float sc  = sqrtf(1.0f / avg); // constant folded per modulation: correctly rounded division and square root
  return make_float2((float)lr * sc, (float)li * sc);
}

