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

// Device-resident description of one polar code (polar_code_impl::set) plus the pruned SSC schedule.
struct polar_plan {
  uint32_t K, E, n, N, nPC;
  // device tables
  uint16_t* d_info_pos;  // K + nPC positions of the K-set, ascending
  uint8_t*  d_is_pc;     // K + nPC flags: 1 = parity-check position
  uint16_t* d_tx_src;    // E entries: rate-matched bit o = encoded bit tx_src[o]
  int32_t*  d_rx_first;  // N entries: first rate-matched index feeding codeword position q (-1: punctured -> 0, -2: shortened -> +inf)
  uint16_t* d_rx_fidx;   // E entries: index in the received sequence of bit-selection index k (identity unless ibil)
  uint32_t* d_sched;     // SSC schedule: op | stage << 4 | pos << 8
  uint8_t*  d_pi_il;     // K entries: CRC interleaver (pdcch)
  uint32_t  sched_len;
};

struct miphy_ctx_ext {
  std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, polar_plan> polar_plans;
  std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, uint8_t*>   polar_kset; // per-position K-set flags (SCL)
  std::map<uint32_t, float*>                                    twiddles; // N -> device exp(-2 pi i j / N), j < N
  std::map<std::pair<uint32_t, uint32_t>, float*>               ramps;    // (N, offset) -> device window ramp
  std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, float, double, int>, ofdm_plan_dev*> plans;
  std::vector<void*>                                            to_free;
  void*                                                         d_gold = nullptr; // Gold-sequence jump table (chest.hip)
  void*                                                         d_pi_il_max = nullptr; // polar interleaver pattern (polar.hip)
};

// Returns the device twiddle table for size N (creates and caches it).
int miphy_get_twiddles(miphy_ctx* ctx, uint32_t N, const float** out);
