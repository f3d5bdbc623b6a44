// Host-side caches hanging off a context: FFT twiddle tables, OFDM symbol plans, polar tables.
#pragma once
#include "miphy_internal.h"
#include <map>
#include <tuple>
#include <vector>

struct ofdm_plan_dev {
  int   N, rg, window_offset, nsymb_sf;
  int   cp_len[56];   // per symbol of the subframe
  int   sym_off[56];  // offset of the symbol start (CP included) inside its slot
  float coef_re[56];  // phase compensation * scale
  float coef_im[56];
};

struct miphy_ctx_ext {
  std::map<uint32_t, float*>                                    twiddles; // N -> device exp(-2 pi i j / N), j < N
  std::map<std::pair<uint32_t, uint32_t>, float*>               ramps;    // (N, offset) -> device window ramp
  std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, float, double, int>, ofdm_plan_dev*> plans;
  std::vector<void*>                                            to_free;
};

// Returns the device twiddle table for size N (creates and caches it).
int miphy_get_twiddles(miphy_ctx* ctx, uint32_t N, const float** out);
