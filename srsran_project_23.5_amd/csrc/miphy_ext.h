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

// Prepared PDSCH encode (sch.hip): segmentation and descriptor upload once; used by the PDSCH processor plan (pdsch_proc.hip).
struct miphy_pdsch_encode_prepared;
int  miphy_pdsch_encode_prepare(miphy_ctx* ctx, const miphy_pdsch_tb_desc* tbs, uint32_t n, miphy_pdsch_encode_prepared** out);
int  miphy_pdsch_encode_prepared_run(miphy_pdsch_encode_prepared* p, const uint8_t* tb_in, uint8_t* codeword_out, hipStream_t s);
void miphy_pdsch_encode_prepared_destroy(miphy_pdsch_encode_prepared* p);

// One codeblock of the transport-block level PDSCH encoder (pdsch_cb_encode.hip): assembly from the transport block, LDPC encoding and rate
// matching in one kernel.
struct miphy_pdsch_cb_desc {
  uint64_t tb_offset;       // packed transport-block bytes
  uint64_t cw_offset;       // byte offset of this codeblock's E rate-matched bits (one per byte) in the codeword array
  uint32_t tb_bit_offset;   // first transport-block bit of the codeblock
  uint32_t take_bits;       // transport-block bits copied
  uint32_t tb_index;        // index into the TB CRC array
  uint32_t E, Nref, out_len; // rate-matched length, limited buffer (0 = none), part of the circular buffer the rate matcher reads
  uint16_t nof_tb_crc_bits, zero_pad; // > 0 on the last codeblock: append the TB CRC and the zero padding
  uint16_t Z, nof_filler_bits;
  uint8_t  nof_cb_crc_bits, bg, rv, mod;
  uint32_t K;
};
// A codeblock the bit-packed kernel takes (pdsch_cb_encode.hip): lifting size a multiple of 32, byte-aligned pieces, selected bits within its LDS buffer.
__host__ __device__ inline bool miphy_pdsch_cb_packed_ok(const miphy_pdsch_cb_desc& d)
{
  return d.Z % 32 == 0 && d.tb_bit_offset % 8 == 0 && d.take_bits % 8 == 0 && d.zero_pad % 8 == 0 && d.nof_tb_crc_bits % 8 == 0 &&
         (d.nof_cb_crc_bits == 0 || d.nof_cb_crc_bits == 24) && d.E <= (1u << 16);
}
size_t miphy_pdsch_cb_encode_lds(uint32_t K, uint32_t Z, uint32_t out_len);
size_t miphy_pdsch_cb_encode_pk_lds(const miphy_pdsch_cb_desc& d);
int    miphy_pdsch_cb_encode_launch(miphy_ctx* ctx, const miphy_pdsch_cb_desc* d_descs, uint32_t ncb, size_t max_lds, const uint8_t* tb_in, const uint32_t* tb_crc,
                                    uint8_t* cw_out, hipStream_t s, uint32_t npacked = 0, size_t max_lds_pk = 0);
