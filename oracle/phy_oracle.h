/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar) of the reference hot path -- srsRAN_Project 23.5 upper-PHY channel coding,
 * OFDM and DM-RS channel estimation -- used as the parity checker for the HIP kernels.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library; the product path (libmiphy.so) never does.
 *
 * Parity pinning: every function is checked against (1) the reference itself, compiled in place into
 * oracle/_ref/libref_capi.so (tests/test_oracle_vs_ref.py, runs where /root/reference exists) and (2) the golden
 * fixtures in tests/golden/ that were generated from the reference by oracle/gen_golden.py (run everywhere).
 */
#ifndef PHY_ORACLE_H
#define PHY_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* CRC polynomial ids (order of crc_generator_poly, include/srsran/phy/upper/channel_coding/crc_calculator.h:31-38). */
enum { ORC_CRC24A = 0, ORC_CRC24B = 1, ORC_CRC24C = 2, ORC_CRC16 = 3, ORC_CRC11 = 4, ORC_CRC6 = 5 };

/* CRC of a bit sequence (one bit per byte; only bit 0 of each byte is used). */
uint32_t orc_crc_bits(int poly, const uint8_t* bits, unsigned nbits);
/* CRC of the first nbits of an MSB-first packed buffer. */
uint32_t orc_crc_packed(int poly, const uint8_t* bytes, unsigned nbits);

/* LDPC encoder. in: bg_K*Z bytes, one bit per byte, filler = 254. out: out_len <= N_short*Z bytes. */
int orc_ldpc_encode(int bg, int Z, const uint8_t* in, uint8_t* out, unsigned out_len);

/* LDPC decoder (AVX2 arithmetic rule of the reference).
 * Returns: iterations (>=1) when crc_poly >= 0 and the CRC matched after that iteration; 0 otherwise ("nullopt").
 * out_packed: bg_K*Z bits MSB-first. If all input LLRs are zero: out = all ones when crc_poly < 0, untouched
 * otherwise. soft_out (optional, may be NULL): final soft bits, N_full*Z bytes. */
int orc_ldpc_decode(int           bg,
                    int           Z,
                    const int8_t* llr,
                    unsigned      in_len,
                    unsigned      nof_filler,
                    int           crc_poly,
                    unsigned      max_iter,
                    uint8_t*      out_packed,
                    int8_t*       soft_out);

/* Rate matcher. in: N bytes (one bit/byte, filler 254 allowed). out: E bytes. mod = bits per symbol (1,2,4,6,8). */
int orc_ldpc_rate_match(int rv, int mod, unsigned Nref, unsigned nof_filler, const uint8_t* in, unsigned N, uint8_t* out, unsigned E);
/* Rate dematcher (AVX2 combine rule: saturating add clamped to +-120). out: N bytes, in/out. */
int orc_ldpc_rate_dematch(int rv, int mod, unsigned Nref, unsigned nof_filler, int new_data, const int8_t* in, unsigned E, int8_t* out, unsigned N);

/* Transport-block segmentation parameters (TS 38.212 5.2.2 + 5.4.2.1 as restated by the reference segmenter). */
typedef struct {
  unsigned tbs;            /* TB size, bits (no CRC) */
  unsigned nof_tb_crc_bits;
  unsigned nof_cbs;
  unsigned Z;
  unsigned K;              /* segment length = bg_K*Z */
  unsigned N;              /* full codeblock length (3K or 5K) */
  unsigned cb_info_bits;   /* information bits per segment (TB bits incl. TB CRC share, excl. CB CRC) */
  unsigned nof_cb_crc_bits;/* 24 when nof_cbs > 1, else 0 */
  unsigned nof_filler_bits;
  unsigned zero_pad;       /* zero padding bits in last CB */
  unsigned nof_short_segments;
  unsigned E[52];          /* rate-matched length per CB */
  unsigned cw_offset[52];
  unsigned crc_poly;       /* CRC checked by the decoder for each CB */
} orc_segmentation_t;
int orc_ldpc_segmentation(unsigned tbs, int bg, int mod, unsigned nof_layers, unsigned nof_ch_symbols, orc_segmentation_t* s);

/* PDSCH encoder: tb packed bytes -> codeword (one bit per byte), length nof_ch_symbols*mod. */
int orc_pdsch_encode(int bg, int rv, int mod, unsigned Nref, unsigned nof_layers, unsigned nof_ch_symbols,
                     const uint8_t* tb, unsigned tb_bytes, uint8_t* codeword);

/* PUSCH decoder: one (re)transmission. softbuf: nof_cbs * N int8 (in/out), cb_crc: nof_cbs flags (in/out),
 * cb_msgs: nof_cbs * ceil(K/8) bytes decoded messages (in/out). Returns tb_crc_ok; iters_minmax[2]. */
int orc_pusch_decode(int bg, int rv, int mod, unsigned Nref, unsigned nof_layers, unsigned nof_ch_symbols,
                     unsigned tb_bytes, int new_data, const int8_t* llrs, unsigned max_iter, int early_stop,
                     int8_t* softbuf, uint8_t* cb_crc, uint8_t* cb_msgs, uint8_t* tb_out, int* iters_minmax);


/* ------------------------------------------------------------------------------------------------ DFT / OFDM
 * Exact-arithmetic checkers: the transform itself is evaluated in double precision (the reference's generic radix-2 DFT
 * and FFTW are only specified up to rounding: tests/unittests/phy/generic_functions/dft_processor_test.cpp:40-42),
 * everything around it follows the reference's single-precision operation order. cf_t = interleaved float re,im. */
int orc_dft(unsigned N, int inverse, const float* in, float* out);
typedef struct {
  unsigned numerology, bw_rb, dft_size, window_offset;
  float    scale;
  double   center_freq_hz;
} orc_ofdm_cfg;
unsigned orc_ofdm_slot_size(const orc_ofdm_cfg* c, unsigned slot_index);
/* in: slot samples; grid_out: [14][bw_rb*12] cf_t */
int orc_ofdm_demod_slot(const orc_ofdm_cfg* c, unsigned slot_index, const float* in, float* grid_out);
int orc_ofdm_mod_slot(const orc_ofdm_cfg* c, unsigned slot_index, const float* grid_in, float* out);


/* ------------------------------------------------------------------------------------------------ DM-RS PUSCH estimator
 * dmrs_pusch_estimator_impl.cpp:71-212 + port_channel_estimator_average_impl.cpp:97-347 (no frequency hopping: the
 * PUSCH estimator of 23.5 never configures it). Gold sequence: TS 38.211 5.2.1 (pseudo_random_generator_impl.cpp).
 * grid_in: [nof_rx_ports][14][nof_prb_grid*12] cf_t; ce_out: [layer][port][first+nof][nof_prb_grid*12] cf_t (only the
 * allocated PRBs are written); scalars_out: per (port, layer) {rsrp, epre, noise_var, snr, time_alignment_s}. */
void orc_gold_sequence(unsigned c_init, unsigned offset, unsigned nbits, uint8_t* out);
int  orc_dmrs_pusch_estimate(unsigned numerology, unsigned slot_in_frame, int dmrs_type2, unsigned scrambling_id, int n_scid,
                             float scaling, const uint8_t* symbols_mask, const uint8_t* rb_mask, unsigned nof_prb_grid,
                             unsigned first_symbol, unsigned nof_symbols, unsigned nof_tx_layers, unsigned nof_rx_ports,
                             const float* grid_in, float* ce_out, float* scalars_out);


/* ------------------------------------------------------------------------------------------------ polar
 * polar_code_impl.cpp:325-490 (code construction), polar_allocator_impl.cpp:28-70, polar_encoder_impl.cpp:32-86,
 * polar_rate_matcher_impl.cpp:31-106, polar_interleaver_impl.cpp:27-56, polar_rate_dematcher_impl.cpp:29-118,
 * polar_decoder_impl.cpp:32-350 (SSC), polar_deallocator_impl.cpp:27-42, pdcch_encoder_impl.cpp:33-98. */
typedef struct {
  unsigned K, E, nMax, ibil;
  unsigned n, N, nPC, nWmPC;
  uint8_t  K_set[1024];  /* 1 = information (or parity-check) position */
  uint16_t PC_set[4];    /* nPC sorted positions, terminated by 1024 */
  uint16_t blk_interleaver[1024];
} orc_polar_code_t;
int orc_polar_code_set(orc_polar_code_t* c, unsigned K, unsigned E, unsigned nMax, int ibil);
int orc_polar_encode_chain(unsigned K, unsigned E, unsigned nMax, int ibil, const uint8_t* msg, uint8_t* out, uint8_t* allocated_out,
                           uint8_t* encoded_out);
int orc_ulsch_demultiplex(int mod, unsigned nof_layers, unsigned nof_prb, unsigned start_symbol, unsigned nof_symbols, unsigned G_rvd, int dmrs_type,
                          unsigned dmrs_symbols_mask, unsigned cdm_groups, unsigned G_ack, unsigned G_csi1, unsigned G_csi2, unsigned O_ack, unsigned O_csi1,
                          unsigned O_csi2, const int8_t* in, int8_t* sch, int8_t* ack, int8_t* csi1, int8_t* csi2, unsigned* nof_sch_llr,
                          uint16_t* placeholders, unsigned* nof_placeholders);
int orc_polar_sc_textbook(unsigned K, unsigned E, unsigned nMax, int ibil, const int8_t* llr, uint8_t* msg, int* zero_seen);
int orc_polar_decode_chain(unsigned K, unsigned E, unsigned nMax, int ibil, const int8_t* llr, uint8_t* msg, int8_t* dematched_out,
                           uint8_t* decoded_u_out);
void orc_polar_interleave(const uint8_t* in, uint8_t* out, unsigned K, int rx);
/* PDCCH: payload A bits (1 bit/byte) + rnti -> E rate-matched bits. */
int orc_pdcch_encode(const uint8_t* payload, unsigned A, unsigned rnti, unsigned E, uint8_t* out);
/* Successive-cancellation LIST decoder (list size L in {1,2,4,8}) -- NO reference counterpart (the reference only has the
 * list-size-1 SSC decoder above); this restates the algorithm of the HIP kernel so that the kernel can be checked bit for
 * bit: LLR-domain path metrics (PM += |llr| when the decision disagrees with the hard decision), min-sum f, saturating g with
 * the reference's LLR algebra, candidates ranked by (metric, 2*slot + flip), survivors renumbered by rank. Aligned all-frozen blocks
 * (rate-0 nodes) are processed at their own stage: penalty = sum of the negative stage LLRs, bits and partial sums zero.
 * crc_mode 0: best metric path, msg = K bits in K-set order. crc_mode 1 (PDCCH): candidates are de-interleaved (Pi_IL) and
 * checked with CRC24C over 24 leading ones + payload with the RNTI mask; crc_mode 2 (PBCH): CRC24C without ones / mask.
 * In modes 1/2 msg = the K de-interleaved bits of the selected path. Returns the chosen path's metric; *crc_ok. */
int orc_polar_scl_decode(unsigned K, unsigned E, unsigned nMax, int ibil, unsigned L, int crc_mode, unsigned rnti, const int8_t* llr,
                         uint8_t* msg, int* crc_ok);
/* PBCH (pbch_encoder_impl.cpp:41-190): payload 32 bytes (first 24 used) -> 864 rate-matched bits. */
int orc_pbch_encode(unsigned N_id, unsigned ssb_idx, unsigned L_max, int hrf, unsigned sfn, unsigned k_ssb, const uint8_t* payload, uint8_t* out);


/* ------------------------------------------------------------------------------------------------ PUSCH demodulator
 * Soft demapper alone: symbols cf_t, one noise variance per symbol, mod in {1 (pi/2-BPSK), 2, 4, 6, 8} -> nsym*mod int8 LLRs. */
void orc_demodulate_soft(int mod, unsigned nsym, const float* symbols, const float* noise_vars, int8_t* llr);
/* Whole demodulator for one transmit layer: RE extraction (DM-RS symbols keep the REs outside the CDM groups without data),
 * ZF/MRC equalisation over nof_rx_ports, soft demapping, descrambling (c_init = rnti*2^15 + n_id). grid: [port][14][nsc] cf_t,
 * ce: [port][ce_nof_symbols][nsc] cf_t (absolute symbol index), noise_var: estimator's value for port 0. Returns the number of
 * LLRs written. eq_out / nvar_out (optional): equalised symbols and post-equalisation noise variances. */
int orc_pusch_demodulate_ex(unsigned rnti, unsigned n_id, int mod, unsigned start_symbol, unsigned nof_symbols, const uint8_t* dmrs_symbols_mask,
                            int dmrs_type2, unsigned nof_cdm_groups_without_data, const uint8_t* rb_mask, unsigned nof_prb_grid,
                            unsigned nof_rx_ports, const float* grid, const float* ce, unsigned ce_nof_symbols, float noise_var, int8_t* llr_out,
                            float* eq_out, float* nvar_out, const uint16_t* placeholders, unsigned nof_ph, float* evm_out);
int orc_pusch_demodulate(unsigned rnti, unsigned n_id, int mod, unsigned start_symbol, unsigned nof_symbols, const uint8_t* dmrs_symbols_mask,
                         int dmrs_type2, unsigned nof_cdm_groups_without_data, const uint8_t* rb_mask, unsigned nof_prb_grid,
                         unsigned nof_rx_ports, const float* grid, const float* ce, unsigned ce_nof_symbols, float noise_var, int8_t* llr_out,
                         float* eq_out, float* nvar_out);

/* Zero-forcing equalizer on its own (channel_equalizer_zf_impl.cpp:123-162): one layer on 1..4 ports (equalize_zf_1xn.h:120-158, the
 * scalar path with the exact reciprocal) or two layers on two ports (equalize_zf_2x2.cpp:30-117). Layouts of the reference's tensors:
 * ch_symbols [port][nof_re], ch_estimates [layer][port][nof_re], eq_symbols / eq_noise_vars [layer][nof_re]; cf_t as float pairs.
 * Returns 0, or -1 for a topology the reference asserts on. */
int orc_channel_equalize(unsigned nof_re, unsigned nof_rx_ports, unsigned nof_tx_layers, const float* ch_symbols, const float* ch_estimates,
                         float noise_var, float tx_scaling, float* eq_symbols, float* eq_noise_vars);

/* ------------------------------------------------------------------------------------------------ PDSCH modulator + DM-RS
 * Modulation mapper: bits one per byte -> cf_t symbols (mod 1 = pi/2-BPSK, 2, 4, 6, 8). */
void orc_modulate(int mod, unsigned nsym, const uint8_t* bits, float* symbols);
/* pdsch_modulator::modulate. cw0/cw1: codewords, one bit per byte; mod[2]; prb_list: allocated PRBs in mapping order; reserved patterns:
 * res_prb_mask [nof_reserved][nof_prb_grid] bytes, res_re_mask (12 bits), res_symbols (14 bits); ports[layer] = grid port.
 * grid: [ports][14][nof_prb_grid*12] cf_t, only the mapped REs are written. Returns the number of REs per layer, or -1 when the
 * codeword lengths do not match the allocation. */
int orc_pdsch_modulate(unsigned rnti, unsigned n_id, float scaling, unsigned nof_layers, const int* mod, const uint8_t* cw0, unsigned nbits0,
                       const uint8_t* cw1, unsigned nbits1, unsigned start_symbol, unsigned nof_symbols, const uint8_t* dmrs_symbols_mask,
                       int dmrs_type2, unsigned nof_cdm_groups_without_data, unsigned bwp_start_rb, unsigned bwp_size_rb, const uint16_t* prb_list,
                       unsigned nof_prb, unsigned nof_reserved, const uint8_t* res_prb_mask, const uint16_t* res_re_mask,
                       const uint16_t* res_symbols, const uint8_t* ports, unsigned nof_prb_grid, float* grid);
/* dmrs_pdsch_processor::map. ports[p] = grid port of the p-th DM-RS port (1000 + p). */
int orc_dmrs_pdsch_map(unsigned slot_in_frame, unsigned reference_point_k_rb, int type2, unsigned scrambling_id, int n_scid, float amplitude,
                       const uint8_t* symbols_mask, const uint8_t* rb_mask, unsigned nof_prb_grid, unsigned nof_ports, const uint8_t* ports, float* grid);

/* ------------------------------------------------------------------------------------------------ PDCCH processor
 * pdcch_processor_impl::process after the CCE-to-PRB mapping (pdcch_processor_impl.cpp:65-117): encoder, scrambling + QPSK + scaling +
 * mapping (pdcch_modulator_impl.cpp:30-91), DM-RS (dmrs_pdcch_processor_impl.cpp:30-101). rb_mask: one byte per PRB of the grid.
 * grid: one port, [14][nof_prb_grid*12] cf_t; only the mapped REs are written. Returns the number of data REs or -1. */
int orc_pdcch_process(unsigned slot_in_frame, unsigned rnti, unsigned n_id_data, unsigned n_rnti, unsigned n_id_dmrs, unsigned reference_point_k_rb,
                      float data_power_offset_dB, float dmrs_power_offset_dB, const uint8_t* payload, unsigned A, unsigned aggregation_level,
                      unsigned start_symbol, unsigned duration, const uint8_t* rb_mask, unsigned nof_prb_grid, float* grid);

/* ------------------------------------------------------------------------------------------------ SS/PBCH block processor
 * ssb_processor_impl::process (ssb_processor_impl.cpp:30-106) after the position look-up: PBCH encoder, pbch_modulator (:28-113),
 * dmrs_pbch_processor (:28-100), pss_processor (:28-91), sss_processor (:28-119). grid: one port, [14][nof_prb_grid*12] cf_t. */
int orc_ssb_process(unsigned N_id, unsigned ssb_idx, unsigned L_max, int hrf, unsigned sfn, unsigned k_ssb, const uint8_t* payload, unsigned ssb_first_subcarrier,
                    unsigned ssb_first_symbol, float beta_pss_dB, unsigned nof_prb_grid, float* grid);

/* ------------------------------------------------------------------------------------------------ NZP-CSI-RS generator
 * nzp_csi_rs_generator_impl::map (nzp_csi_rs_generator_impl.cpp:163-221) with generate_sequence (:110-129), get_nof_skipped_elements (:69-108),
 * get_seq_len (:131-161) and apply_cdm (:223-296), given the per-port patterns of get_csi_rs_pattern(). grid: [max port + 1][14][nsc] cf_t. */
int orc_csi_rs_map(unsigned slot_in_frame, unsigned scrambling_id, float amplitude, unsigned start_rb, unsigned nof_rb, unsigned rb_begin, unsigned rb_end,
                   unsigned rb_stride, unsigned mapping_row, unsigned cdm, unsigned freq_density, unsigned nof_ports, const uint8_t* ports,
                   const uint16_t* re_mask, const uint16_t* symbol_mask, unsigned nof_prb_grid, float* grid);

/* ------------------------------------------------------------------------------------------------ Open Fronthaul IQ
 * compression: 0 = none (fixed point, iq_compression_none_impl.cpp:29-69), 1 = BFP (iq_compression_bfp_impl.cpp:28-143).
 * payload: per PRB, BFP [udCompParam][24 x data_width bits, big endian] = 1 + 3*data_width bytes, none the 3*data_width bytes only
 * (the U-plane section payload, ofh_uplane_message_builder_impl.cpp:145-152). simd_arithmetic: the production AVX2 / AVX-512 BFP
 * classes (data_width 9 multiplies by the rounded reciprocal 1 / (32767 / 2^e), iq_compression_bfp_avx2.cpp:92-132,
 * srsvec/conversion.cpp:101-130); 0: the generic class (division by 32767, iq_compression_bfp_impl.cpp:100-121). */
void orc_ofh_iq_decompress(int compression, const uint8_t* payload, unsigned nof_prb, unsigned data_width, int simd_arithmetic, float* out /* cf_t, nof_prb*12 */);
/* iq_compression_bfp_impl::compress (:47-98) / iq_compression_none_impl::compress (:29-51): quantisation through
 * srsvec::convert_round (round to nearest even with int16 saturation on the SIMD part of each converted span, std::round and a
 * wrapping cast on its tail, conversion.cpp:61-99; BFP converts the whole call at once, none PRB by PRB), [exponent, shift,] pack.
 * data_width 8..16 (narrower widths trip an assertion in the reference's packing, bit_buffer::insert). */
void orc_ofh_iq_compress(int compression, const float* in /* cf_t, nof_prb*12 */, unsigned nof_prb, unsigned data_width, float iq_scaling, uint8_t* payload);

#ifdef __cplusplus
}
#endif
#endif
