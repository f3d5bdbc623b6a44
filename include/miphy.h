/*
 * miphy.h -- C ABI of the MI355X-native 5G NR upper-PHY hot path (libmiphy.so).
 *
 * This is the drop-in boundary: plain C, POD arguments, caller-owned DEVICE pointers (HBM resident), an explicit
 * HIP stream passed as void*.  Every entry point is batched -- the per-call C++ adapters that implement the
 * reference's abstract classes (srsran_project_23.5_amd/adapters/) wrap these with a batch of one.
 *
 * Each function cites the reference interface it replaces (paths relative to the srsRAN_Project 23.5 tree).
 * Return value: 0 on success, negative MIPHY_E* on error (the reference aborts through srsran_assert on the same
 * precondition violations; C has no asserts, so the adapters turn a non-zero code into report_fatal_error).
 *
 * Data conventions equal the reference's:
 *   - LLR: int8, finite range [-120,120], +-127 = +-infinity (include/srsran/phy/upper/log_likelihood_ratio.h:46-240)
 *   - packed bits: MSB first in each byte (include/srsran/adt/bit_buffer.h:35-44)
 *   - unpacked bits: one bit per byte, filler bit = 254 (include/srsran/phy/upper/channel_coding/ldpc/ldpc.h:107)
 *   - cf_t: interleaved fp32 (re, im)
 *   - resource grid: [port][symbol][subcarrier], subcarrier fastest (lib/phy/support/resource_grid_impl.h:42-46)
 */
#ifndef MIPHY_H
#define MIPHY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPHY_VERSION 1

enum {
  MIPHY_OK        = 0,
  MIPHY_EINVAL    = -1, /* invalid argument (would be an srsran_assert in the reference) */
  MIPHY_EHIP      = -2, /* HIP runtime error, see miphy_last_error() */
  MIPHY_ENOMEM    = -3,
  MIPHY_EUNSUPP   = -4
};

/* CRC polynomials, same order as srsran::crc_generator_poly (include/srsran/phy/upper/channel_coding/crc_calculator.h:31-38). */
enum { MIPHY_CRC24A = 0, MIPHY_CRC24B = 1, MIPHY_CRC24C = 2, MIPHY_CRC16 = 3, MIPHY_CRC11 = 4, MIPHY_CRC_NONE = 255 };

typedef struct miphy_ctx miphy_ctx;

/* Creates a context on HIP device `device`: uploads the TS 38.212 graph tables, CRC tables and polar tables and
 * allocates the scratch workspace.  One context per host thread (the reference's processors are single-threaded
 * objects, one per worker: lib/phy/upper/uplink_processor_concurrent.h:41-54).
 * Execution model of every *_batch entry point: the work is ENQUEUED on `stream` and the call returns; nothing waits for the
 * stream. Descriptor arrays in host memory are copied when the call is made (pinned staging ring of the context: the array can be
 * reused at once; the device is synchronised only when the 8 MB ring wraps) -- or pass descriptors that already live on the device
 * (`*_on_device` = 1). Scratch workspaces belong to the context and are reused by its next call in stream order: use one stream
 * at a time per context. With device-resident descriptors and plans a sequence of calls can be captured in a HIP graph as it is
 * (bench.py replays the single-slot pipeline that way); host descriptors must not be captured: a replay would read a staging
 * slot that has been reused since. */
int         miphy_create(int device, miphy_ctx** ctx);
void        miphy_destroy(miphy_ctx* ctx);
const char* miphy_last_error(void);
int         miphy_version(void);

/* ------------------------------------------------------------------------------------------------------------------
 * LDPC decoder  --  replaces srsran::ldpc_decoder::decode
 *   include/srsran/phy/upper/channel_coding/ldpc/ldpc_decoder.h:73-74
 *   lib/phy/upper/channel_coding/ldpc/ldpc_decoder_impl.cpp:60-146 with the AVX2 arithmetic of ldpc_decoder_avx2.cpp.
 * One descriptor per codeblock.  `iters[i]` receives the value of the optional<unsigned> the reference returns:
 * the iteration count (>=1) when crc_poly != NONE and the CRC matched, 0 for nullopt.
 * The output holds bg_K*Z bits per codeblock, packed.  All-zero input: output all ones when crc_poly == NONE,
 * untouched otherwise (ldpc_decoder_impl.cpp:88-94).
 * Input LLR domain: [-127,127]; `in_len` should be a multiple of Z for bit-exact parity. */
typedef struct {
  uint8_t  bg;              /* 1 or 2 */
  uint8_t  crc_poly;        /* MIPHY_CRC*; MIPHY_CRC_NONE = no early stop (crc == nullptr) */
  uint16_t Z;               /* lifting size */
  uint16_t max_iter;        /* algorithm_conf.max_iterations (default 6) */
  uint16_t nof_filler_bits; /* cb_specific.nof_filler_bits */
  uint32_t in_len;          /* number of input LLRs (first one belongs to variable node 2) */
  uint32_t flags;           /* bit 0: check the CRC only after the last iteration (pusch_decoder_impl.cpp:105-118) */
  uint64_t llr_offset;      /* element offset of this codeblock's LLRs inside `llr` */
  uint64_t out_offset;      /* byte offset of this codeblock's packed message inside `out_bits` */
} miphy_ldpc_dec_desc;

/* Optional launch bounds for device-resident descriptors (which the host cannot inspect): the largest lifting size
 * and input length in the batch. NULL = worst case (Z = 384, full-length codeblocks: one workgroup per CU).
 * The library sizes its on-chip buffers for ceil((max_in_len + 2 max_Z) / max_Z) variable nodes. When the batch mixes lifting
 * sizes, a codeblock with a smaller Z reaches more nodes per LLR: pass max_in_len = (n_max - 2) * max_Z with
 * n_max = max over codeblocks of ceil((in_len + 2 Z) / Z). A bound that is too small corrupts the result.
 * Any mix of lifting sizes up to max_Z, odd ones included, is legal in one batch. Device-resident descriptors are decoded by ONE
 * launch sized for max_Z (a codeblock with a small Z then leaves most lanes of its workgroup idle); hand the descriptors over in host
 * memory, or use the transport-block level entry points, to have the batch sorted into per-lifting-size launch classes. */
typedef struct {
  uint32_t max_Z;
  uint32_t max_in_len;
} miphy_ldpc_dec_limits;

int miphy_ldpc_decode_batch(miphy_ctx*                 ctx,
                            const miphy_ldpc_dec_desc* descs, /* n descriptors; host memory unless descs_on_device */
                            int                        descs_on_device,
                            uint32_t                   n,
                            const int8_t*              llr,      /* device */
                            uint8_t*                   out_bits, /* device */
                            int32_t*                   iters,    /* device, n entries */
                            const miphy_ldpc_dec_limits* limits, /* may be NULL */
                            void*                      stream);
/* Prepared form of miphy_ldpc_decode_batch for batches whose geometry repeats (same descriptors slot after slot): the descriptors
 * are validated, sorted into launch classes and uploaded once; a run only launches (no host synchronisation, no staging). */
typedef struct miphy_ldpc_decode_plan miphy_ldpc_decode_plan;
int      miphy_ldpc_decode_plan_create(miphy_ctx* ctx, const miphy_ldpc_dec_desc* descs /* host */, uint32_t n, miphy_ldpc_decode_plan** out);
int      miphy_ldpc_decode_plan_run(miphy_ldpc_decode_plan* plan, const int8_t* llr, uint8_t* out_bits, int32_t* iters, void* stream);
uint32_t miphy_ldpc_decode_plan_nof_launches(const miphy_ldpc_decode_plan* plan); /* kernel launches per run = launch classes */
void     miphy_ldpc_decode_plan_destroy(miphy_ldpc_decode_plan* plan);

/* Test / A-B knob: 0 = automatic choice (host descriptors: sorted into launch classes by lifting size and code rate, one launch per
 * class; device descriptors: one launch), 1 = one-row-per-lane kernel, 2 = packed two-rows-per-lane kernel as one launch,
 * 3 = class-sorted launches, 4 = class-sorted launches with the geometry of a batch that fills the chip whatever its size (no latency form, messages in global
 * memory wherever that buys residency), 5 = class-sorted launches with the latency form of the packed kernel (two parts) on every class (A-B measurement only),
 * 6 = automatic choice with the latency form in two parts instead of four. Adding 0x100 keeps ALL messages of a GMSG launch in global memory (A-B of the
 * split between LDS and global memory). All kernels produce identical results. */
void miphy_debug_force_ldpc_kernel(int mode);
/* Which decoder kernels have been launched since the last reset (tests assert that a forced choice really ran): */
#define MIPHY_LDPC_KERNEL_SCALAR 1u /* one check row per lane */
#define MIPHY_LDPC_KERNEL_PACKED 2u /* two check rows per lane, one codeblock per workgroup */
#define MIPHY_LDPC_KERNEL_FUSED  4u /* ... that rate-dematches while it loads */
#define MIPHY_LDPC_KERNEL_GMSG   8u /* ... with check-to-variable messages (all, or the layers behind a boundary) in global memory */
#define MIPHY_LDPC_KERNEL_WAVE  16u /* several small codeblocks per wavefront (Z <= 64) */
#define MIPHY_LDPC_KERNEL_SPLIT 32u /* packed kernel in its latency form: four (or two) times the wavefronts per codeblock (launches of at most one codeblock per CU) */
#define MIPHY_LDPC_KERNEL_GMSG_PART 64u /* ... a GMSG launch that kept the first layers' messages in LDS (only the layers behind a boundary in global memory) */
unsigned miphy_debug_ldpc_kernels_used(int reset);
/* A-B knob: streams the launch classes of one call are spread over (1 = one after another on the caller's stream). */
void miphy_debug_set_ldpc_class_streams(int n);
/* Test hook: codeblocks of the PDSCH encoder launches since the last reset that the bit-packed kernel took (out[0]) and in total (out[1]). */
void miphy_debug_pdsch_cb_counts(unsigned out[2], int reset);

/* ------------------------------------------------------------------------------------------------------------------
 * LDPC rate dematcher  --  replaces srsran::ldpc_rate_dematcher::rate_dematch
 *   include/srsran/phy/upper/channel_coding/ldpc/ldpc_rate_dematcher.h:52-55
 *   lib/phy/upper/channel_coding/ldpc/ldpc_rate_dematcher_impl.cpp:43-254, AVX2 combine rule
 *   (ldpc_rate_dematcher_avx2_impl.cpp:45-58: saturating add clamped to +-120).
 * One descriptor per codeblock. The output is the HARQ soft buffer view of the FULL codeblock (N = 66Z / 50Z LLRs),
 * read-modify-written exactly like the reference does: with new_data the same positions are cleared / set to +inf
 * (fillers) / left untouched, otherwise the new LLRs are combined into the existing ones. */
typedef struct {
  uint8_t  bg;              /* 1 or 2 */
  uint8_t  rv;              /* redundancy version 0..3 */
  uint8_t  mod;             /* bits per symbol: 1, 2, 4, 6, 8 */
  uint8_t  new_data;        /* 1: first transmission (copy mode), 0: combine with the soft buffer */
  uint16_t Z;
  uint16_t nof_filler_bits;
  uint32_t Nref;            /* limited-buffer length, 0 = unlimited */
  uint32_t E;               /* rate-matched length of this codeblock (multiple of mod) */
  uint64_t in_offset;       /* element offset of the E input LLRs inside `llr_in` */
  uint64_t out_offset;      /* element offset of the N-LLR soft buffer inside `softbuf` */
} miphy_ldpc_rdm_desc;

/* Optional launch bound for device-resident descriptors: the largest E in the batch (sizes the LDS staging buffer;
 * NULL = assume up to 60 KiB, which limits residency). */
typedef struct {
  uint32_t max_E;
} miphy_ldpc_rdm_limits;

int miphy_ldpc_rate_dematch_batch(miphy_ctx*                   ctx,
                                  const miphy_ldpc_rdm_desc*   descs,
                                  int                          descs_on_device,
                                  uint32_t                     n,
                                  const int8_t*                llr_in,  /* device */
                                  int8_t*                      softbuf, /* device, in/out */
                                  const miphy_ldpc_rdm_limits* limits,  /* may be NULL */
                                  void*                        stream);

/* ------------------------------------------------------------------------------------------------------------------
 * LDPC rate matcher  --  replaces srsran::ldpc_rate_matcher::rate_match
 *   include/srsran/phy/upper/channel_coding/ldpc/ldpc_rate_matcher.h:46
 *   lib/phy/upper/channel_coding/ldpc/ldpc_rate_matcher_impl.cpp:42-182.
 * in: full codeblock, N = 66Z / 50Z bytes, one bit per byte (fillers = 254 are skipped positionally);
 * out: E bytes, one bit per byte. Uses the same descriptor as the dematcher (new_data ignored;
 * in_offset -> codeblock inside `cb_in`, out_offset -> E output bytes inside `out`). */
int miphy_ldpc_rate_match_batch(miphy_ctx*                 ctx,
                                const miphy_ldpc_rdm_desc* descs,
                                int                        descs_on_device,
                                uint32_t                   n,
                                const uint8_t*             cb_in, /* device */
                                uint8_t*                   out,   /* device */
                                void*                      stream);

/* ------------------------------------------------------------------------------------------------------------------
 * LDPC encoder  --  replaces srsran::ldpc_encoder::encode
 *   include/srsran/phy/upper/channel_coding/ldpc/ldpc_encoder.h:46-47
 *   lib/phy/upper/channel_coding/ldpc/ldpc_encoder_impl.cpp:44-81, ldpc_encoder_generic.cpp:30-223.
 * in: bg_K*Z bytes, one bit per byte (filler bits = 254, copied through to the output);
 * out: out_len <= N_short*Z bytes (codeblock without the first 2Z punctured bits). */
typedef struct {
  uint8_t  bg;
  uint8_t  reserved0;
  uint16_t Z;
  uint32_t out_len;
  uint64_t in_offset;  /* byte offset of the message inside `msg_in` */
  uint64_t out_offset; /* byte offset of the codeblock inside `cb_out` */
} miphy_ldpc_enc_desc;

int miphy_ldpc_encode_batch(miphy_ctx*                 ctx,
                            const miphy_ldpc_enc_desc* descs,
                            int                        descs_on_device,
                            uint32_t                   n,
                            const uint8_t*             msg_in, /* device */
                            uint8_t*                   cb_out, /* device */
                            void*                      stream);

/* ------------------------------------------------------------------------------------------------------------------
 * CRC calculator  --  replaces srsran::crc_calculator::calculate / calculate_bit / calculate_byte
 *   include/srsran/phy/upper/channel_coding/crc_calculator.h:45-67, lib/phy/upper/channel_coding/crc_calculator_lut_impl.cpp.
 * One descriptor per message; messages are MSB-first packed (bit_offset may be unaligned). */
typedef struct {
  uint64_t bit_offset; /* first bit of the message inside `data` */
  uint32_t nbits;
  uint32_t poly;       /* MIPHY_CRC* */
} miphy_crc_desc;

int miphy_crc_batch(miphy_ctx*            ctx,
                    const miphy_crc_desc* descs,
                    int                   descs_on_device,
                    uint32_t              n,
                    const uint8_t*        data,      /* device, packed */
                    uint32_t*             checksums, /* device, n entries */
                    void*                 stream);

/* ------------------------------------------------------------------------------------------------------------------
 * DFT processor  --  replaces srsran::dft_processor (get_input/run), batched
 *   include/srsran/phy/generic_functions/dft_processor.h:34-73, lib/phy/generic_functions/dft_processor_generic_impl.cpp.
 * Unnormalised; DIRECT uses exp(-j...), INVERSE exp(+j...). `n` transforms stored back to back (size cf_t each).
 * Supported sizes: every 2^a * 3^b <= 4096 in one LDS pass, and 4608 ... 49152 of the reference's list through a four-step
 * transform (all 18 sizes of dft_processor_generic_impl.cpp:193-210); anything else -> MIPHY_EUNSUPP. */
int miphy_dft_batch(miphy_ctx* ctx, uint32_t size, int inverse, uint32_t n, const float* in /* device cf_t */,
                    float* out /* device cf_t */, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * OFDM slot (de)modulator  --  replaces srsran::ofdm_slot_demodulator::demodulate / ofdm_slot_modulator::modulate
 *   include/srsran/phy/lower/modulation/ofdm_demodulator.h:33-102, lib/phy/lower/modulation/ofdm_demodulator_impl.cpp:34-170
 *   include/srsran/phy/lower/modulation/ofdm_modulator.h,          lib/phy/lower/modulation/ofdm_modulator_impl.cpp:55-138
 *   phase compensation: include/srsran/phy/lower/modulation/phase_compensation_lut.h:49-96 (TS 38.211 5.4).
 * The configuration struct has the fields of srsran::ofdm_demodulator_configuration / ofdm_modulator_configuration
 * (normal cyclic prefix). One job = one (slot, port): 14 symbols. */
typedef struct {
  uint32_t numerology;                /* mu: SCS = 15 kHz * 2^mu (0..2) */
  uint32_t bw_rb;                     /* resource grid width in PRB */
  uint32_t dft_size;
  uint32_t nof_samples_window_offset; /* demodulator only; 0 for the modulator */
  float    scale;
  float    reserved;
  double   center_freq_hz;
} miphy_ofdm_config;

typedef struct {
  uint64_t samples_offset; /* cf_t element offset of the first sample of the slot inside the time-domain buffer */
  uint64_t grid_offset;    /* cf_t element offset of this port's [14][bw_rb*12] grid inside the grid buffer */
  uint32_t slot_index;     /* slot index within the subframe */
  uint32_t grid_empty;     /* modulator: non-zero reproduces the empty-grid shortcut (output zeros) */
} miphy_ofdm_job;

int miphy_ofdm_demodulate_slots(miphy_ctx* ctx, const miphy_ofdm_config* cfg, const miphy_ofdm_job* jobs, int jobs_on_device,
                                uint32_t n, const float* samples /* device cf_t */, float* grid /* device cf_t */, void* stream);
int miphy_ofdm_modulate_slots(miphy_ctx* ctx, const miphy_ofdm_config* cfg, const miphy_ofdm_job* jobs, int jobs_on_device,
                              uint32_t n, const float* grid /* device cf_t */, float* samples /* device cf_t */, void* stream);

/* The same transforms one OFDM symbol at a time, for the symbol-granular interfaces of the lower PHY
 * (srsran::ofdm_symbol_demodulator / ofdm_symbol_modulator, ofdm_demodulator.h:55-74, ofdm_modulator.h:55-74; callers
 * lib/phy/lower/processors/uplink/puxch/puxch_processor_impl.cpp:64-76, downlink/pdxch/pdxch_processor_impl.cpp:76-82).
 * A job is one symbol: slot_index = symbol index within the SUBFRAME (0 .. 14 * 2^numerology - 1), samples_offset = cf_t offset of
 * the symbol's first sample (cyclic prefix included), grid_offset = cf_t offset of its row of bw_rb * 12 subcarriers. */
uint32_t miphy_ofdm_symbol_size(const miphy_ofdm_config* cfg, uint32_t symbol_index); /* cyclic prefix + dft_size samples; host only */
int miphy_ofdm_demodulate_symbols(miphy_ctx* ctx, const miphy_ofdm_config* cfg, const miphy_ofdm_job* jobs, int jobs_on_device, uint32_t n,
                                  const float* samples /* device cf_t */, float* grid /* device cf_t */, void* stream);
int miphy_ofdm_modulate_symbols(miphy_ctx* ctx, const miphy_ofdm_config* cfg, const miphy_ofdm_job* jobs, int jobs_on_device, uint32_t n,
                                const float* grid /* device cf_t */, float* samples /* device cf_t */, void* stream);
/* Number of samples of slot `slot_index` (ofdm_slot_demodulator::get_slot_size). Returns 0 on invalid configuration. */
uint32_t miphy_ofdm_slot_size(const miphy_ofdm_config* cfg, uint32_t slot_index);

/* ------------------------------------------------------------------------------------------------------------------
 * DM-RS based PUSCH channel estimator  --  replaces srsran::dmrs_pusch_estimator::estimate (which drives
 * srsran::port_channel_estimator::compute for every receive port)
 *   include/srsran/phy/upper/signal_processors/dmrs_pusch_estimator.h:39-85
 *   lib/phy/upper/signal_processors/dmrs_pusch_estimator_impl.cpp:71-212      (Gold-sequence pilots, layer weights)
 *   include/srsran/phy/upper/signal_processors/port_channel_estimator.h:45-107
 *   lib/phy/upper/signal_processors/port_channel_estimator_average_impl.cpp:97-347 (LS, averaging, RSRP/EPRE/noise/SNR,
 *   time alignment through a 4096-point IDFT, linear interpolation, broadcast to all symbols).
 * One job = one PUSCH allocation in one slot; the kernel runs one workgroup per (job, rx port, layer).
 * DM-RS type 1 only (the reference's estimator asserts out of bounds for type 2); no intra-slot frequency hopping
 * (dmrs_pusch_estimator_impl.cpp never configures it). */
typedef struct {
  uint32_t numerology;     /* slot.numerology() */
  uint32_t slot_in_frame;  /* slot.slot_index(): slot number within the radio frame */
  uint32_t scrambling_id;
  float    scaling;        /* beta_DMRS amplitude scaling */
  uint8_t  n_scid;
  uint8_t  nof_tx_layers;  /* 1..4 */
  uint8_t  nof_rx_ports;   /* 1..4, port p is grid port rx_ports[p] */
  uint8_t  first_symbol;
  uint8_t  nof_symbols;
  uint8_t  rx_ports[4];
  uint8_t  ce_compact;     /* 0: the estimate is copied to every symbol of the allocation like the reference does
                              (port_channel_estimator_average_impl.cpp:216-224); 1: one row per (layer, rx port) -- the copies are
                              identical, so a consumer that knows it (miphy_pusch_demodulate_batch with ce_compact) needs no more */
  uint8_t  reserved[2];
  uint16_t symbols_mask;   /* bit l = OFDM symbol l carries DM-RS */
  uint16_t grid_nof_prb;   /* width of the resource grid / rb_mask.size() */
  uint64_t rb_mask[5];     /* bit i = PRB i belongs to the allocation */
  uint64_t grid_offset;    /* cf_t offset of grid port 0: [port][14][grid_nof_prb*12] */
  uint64_t ce_offset;      /* cf_t offset of the estimate: [layer][rx port][first_symbol+nof_symbols][grid_nof_prb*12];
                              only the allocated PRBs of symbols [first_symbol, first_symbol+nof_symbols) are written.
                              With ce_compact: [layer][rx port][grid_nof_prb*12] */
  uint64_t scalars_offset; /* float offset: per (rx port, layer) {rsrp, epre, noise_var, snr, time_alignment_s} */
  /* port_channel_estimator::configuration beyond what dmrs_pusch_estimator_impl fills in (port_channel_estimator.h:45-107,
   * port_channel_estimator_average_impl.cpp:97-224): intra-slot frequency hopping and externally generated pilots. */
  uint64_t rb_mask2[5];    /* layer_dmrs_pattern::rb_mask2: allocation of the second hop (used when hop_symbol != 0) */
  uint64_t pilots_offset;  /* cf_t offset into `pilots` of miphy_port_channel_estimate_batch: dmrs_symbol_list
                              [layer][DM-RS symbol][pilot of the allocated PRBs]; ignored by miphy_dmrs_pusch_estimate_batch */
  uint8_t  hop_symbol;     /* layer_dmrs_pattern::hopping_symbol_index: first OFDM symbol of the second hop, 0 = no hopping;
                              honoured by miphy_port_channel_estimate_batch only (the reference's PUSCH estimator never hops) */
  uint8_t  re_odd_mask;    /* with external pilots: bit ly = layer ly has its DM-RS on the odd subcarriers (layer_dmrs_pattern::re_pattern of
                              DM-RS type 1); the PUSCH estimator derives it from the layer number instead */
  uint8_t  reserved2[6];
} miphy_pusch_chest_job;

/* The port estimator on its own: the same kernel with the pilots given by the caller (device, cf_t) instead of generated from
 * (slot, scrambling_id, n_scid) -- srsran::port_channel_estimator::compute (port_channel_estimator.h:102-106). slot_in_frame,
 * scrambling_id and n_scid of the jobs are ignored; hop_symbol / rb_mask2 select intra-slot frequency hopping (per-hop least squares,
 * interpolation and mapping, time alignment averaged over the hops, noise variance = epre / 1000 as the reference forces it with
 * hopping: port_channel_estimator_average_impl.cpp:118-138). */
/* jobs_on_device: 0 = host array (validated), 1 = device array; with a device array bits 8-11 / 12-15 may carry the largest nof_rx_ports /
 * nof_tx_layers of the batch (0 = unknown: workgroups for 4 x 4 are launched and the surplus ones exit at once). */
#define MIPHY_JOBS_ON_HOST 0
#define MIPHY_JOBS_ON_DEVICE 1
#define MIPHY_JOBS_ON_DEVICE_HINT(max_ports, max_layers) (1 | ((int)(max_ports) << 8) | ((int)(max_layers) << 12))
int miphy_port_channel_estimate_batch(miphy_ctx* ctx, const miphy_pusch_chest_job* jobs, int jobs_on_device, uint32_t n, const float* grid /* device cf_t */,
                                      const float* pilots /* device cf_t */, float* ce /* device cf_t */, float* scalars /* device */, void* stream);
int miphy_dmrs_pusch_estimate_batch(miphy_ctx* ctx, const miphy_pusch_chest_job* jobs, int jobs_on_device, uint32_t n,
                                    const float* grid /* device cf_t */, float* ce /* device cf_t */, float* scalars /* device */,
                                    void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * PUSCH demodulator  --  replaces srsran::pusch_demodulator::demodulate (the block between the channel estimator
 * and the decoder), i.e. channel_equalizer::equalize + demodulation_mapper::demodulate_soft + descrambling in one pass
 *   include/srsran/phy/upper/channel_processors/pusch_demodulator.h:40-108
 *   lib/phy/upper/channel_processors/pusch_demodulator_impl.cpp:31-152, pusch_demodulator_impl.h:74-172
 *   lib/phy/upper/equalization/channel_equalizer_zf_impl.cpp:123-162, equalize_zf_1xn.h:42-158
 *   lib/phy/upper/channel_modulation/demodulation_mapper_impl.cpp:34-106 and demodulation_mapper_q*.cpp
 * One transmit layer (the reference asserts the same, pusch_demodulator_impl.cpp:83), 1..4 receive ports, pi/2-BPSK .. 256QAM,
 * DM-RS type 1 or 2, no UCI placeholders, no EVM report. One job = one PUSCH transmission; the LLRs are the codeword in the order
 * the decoder expects (symbol-major, subcarrier ascending). The channel estimate and the noise variance are read where
 * miphy_dmrs_pusch_estimate_batch wrote them (layer 0 block; noise variance of rx port 0: scalars[scalars_offset + 2]). */
typedef struct {
  uint32_t rnti;
  uint32_t n_id;           /* scrambling identity, c_init = rnti * 2^15 + n_id */
  uint8_t  mod;            /* bits per symbol: 1 (pi/2-BPSK), 2, 4, 6, 8 */
  uint8_t  nof_rx_ports;   /* 1..4, port p is grid port rx_ports[p] */
  uint8_t  start_symbol;
  uint8_t  nof_symbols;
  uint8_t  dmrs_type;      /* 1 or 2 */
  uint8_t  nof_cdm_groups_without_data;
  uint8_t  ce_nof_symbols; /* symbols per (port) block of the channel estimate = first_symbol + nof_symbols of the estimator job */
  uint8_t  ce_compact;     /* 1: the channel estimate has one row per rx port ([rx port][grid_nof_prb*12], an estimator job with
                              ce_compact) used for every symbol; ce_nof_symbols is then ignored */
  uint8_t  rx_ports[4];
  uint16_t dmrs_symbols_mask; /* bit l = OFDM symbol l carries DM-RS */
  uint16_t grid_nof_prb;
  uint32_t nof_llr;        /* codeword length; must equal miphy_pusch_demod_nof_llr() */
  uint64_t rb_mask[5];
  uint64_t grid_offset;    /* cf_t offset of grid port 0: [port][14][grid_nof_prb*12] */
  uint64_t ce_offset;      /* cf_t offset of the channel estimate: [rx port][ce_nof_symbols][grid_nof_prb*12] */
  uint64_t scalars_offset; /* float offset of the estimator scalars of this transmission */
  uint64_t llr_offset;     /* int8 offset of the output codeword */
  /* UCI multiplexed on the PUSCH (pusch_demodulator::configuration::placeholders, pusch_demodulator_impl.cpp:99-152): indices, in
   * codeword order, of the resource elements that carry a repetition placeholder (ulsch_placeholder_list; from
   * miphy_ulsch_placeholders). In such an element bit 0 is descrambled normally, bit 1 with the chip of bit 0, the others not at all. */
  uint32_t placeholders_offset; /* uint16 offset into the `placeholders` array of miphy_pusch_demodulate_batch_ex */
  uint32_t nof_placeholders;    /* 0: none */
  uint64_t evm_offset;          /* float offset into `evm_sums` of 14 per-symbol sums of |hard-decided symbol - equalised symbol|^2
                                   (pusch_demodulator_impl.cpp:89-90, evm_calculator_generic_impl.cpp:31-47): EVM = sqrt(sum / data REs) */
} miphy_pusch_demod_job;

/* Number of LLRs the allocation of `job` produces (data REs x mod); 0 on an invalid job. Host function. */
uint32_t miphy_pusch_demod_nof_llr(const miphy_pusch_demod_job* job);
/* _ex: with the placeholder lists (device, may be NULL when no job has any) and the EVM sums (device, may be NULL: no EVM). */
int miphy_pusch_demodulate_batch_ex(miphy_ctx* ctx, const miphy_pusch_demod_job* jobs, int jobs_on_device, uint32_t n, const float* grid,
                                    const float* ce, const float* scalars, int8_t* llr, const uint16_t* placeholders, float* evm_sums, void* stream);
int miphy_pusch_demodulate_batch(miphy_ctx* ctx, const miphy_pusch_demod_job* jobs, int jobs_on_device, uint32_t n,
                                 const float* grid /* device cf_t */, const float* ce /* device cf_t */, const float* scalars /* device */,
                                 int8_t* llr_out /* device */, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Channel equalizer on its own  --  replaces srsran::channel_equalizer::equalize of the zero-forcing equalizer
 * (include/srsran/phy/upper/equalization/channel_equalizer.h:27-114, lib/phy/upper/equalization/channel_equalizer_zf_impl.cpp:123-162).
 * Topologies are the reference's: one transmit layer with 1..4 receive ports (equalize_zf_1xn.h:120-158) and two layers on two ports
 * (equalize_zf_2x2.cpp:30-117); anything else is MIPHY_EINVAL where the reference asserts. The noise variance is the one of the
 * first receive port, as the reference takes it (channel_equalizer_zf_impl.cpp:141-142). Abnormal elements (zero / non-finite
 * denominator, non-positive or non-finite noise variance) give symbol 0 and noise variance +inf.
 * Note: in 23.5 the PUSCH demodulator never calls the equalizer with two layers (pusch_demodulator_impl.h asserts one layer, "layer
 * demapping is not implemented"), so the 2 x 2 case only exists behind this entry point; miphy_pusch_demodulate_batch fuses the
 * 1 x N case. Tensors are the reference's re_measurement layouts, resource element fastest. */
typedef struct {
  uint32_t nof_re;
  uint8_t  nof_rx_ports;        /* 1..4 */
  uint8_t  nof_tx_layers;       /* 1, or 2 with nof_rx_ports == 2 */
  uint8_t  reserved[2];
  float    noise_var;           /* noise_var_estimates[0] */
  float    tx_scaling;          /* > 0 */
  uint64_t ch_symbols_offset;   /* cf_t offset: [rx port][nof_re] */
  uint64_t ch_estimates_offset; /* cf_t offset: [tx layer][rx port][nof_re] */
  uint64_t eq_symbols_offset;   /* cf_t offset: [tx layer][nof_re] */
  uint64_t eq_noise_vars_offset; /* float offset: [tx layer][nof_re] */
} miphy_equalizer_job;
int miphy_channel_equalize_batch(miphy_ctx* ctx, const miphy_equalizer_job* jobs, int jobs_on_device, uint32_t n, const float* ch_symbols /* device cf_t */,
                                 const float* ch_estimates /* device cf_t */, float* eq_symbols /* device cf_t */, float* eq_noise_vars /* device */,
                                 void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * PDSCH modulator and PDSCH DM-RS  --  replace srsran::pdsch_modulator::modulate and srsran::dmrs_pdsch_processor::map
 * (after the encoder, before the OFDM modulator)
 *   include/srsran/phy/upper/channel_processors/pdsch_modulator.h:40-99, lib/.../pdsch_modulator_impl.cpp:30-282
 *   lib/phy/upper/channel_modulation/modulation_mapper_impl.cpp:31-146
 *   include/srsran/phy/upper/signal_processors/dmrs_pdsch_processor.h:36-66, lib/.../dmrs_pdsch_processor_impl.cpp:30-169
 * Modulator: one codeword on one layer, PRBs mapped in ascending order (what 23.5 supports: its layer mapper and its
 * non-contiguous mapping path are broken, see csrc/pdsch_mod.hip); DM-RS pattern of the bandwidth part and up to four reserved RE
 * patterns are skipped. The codeword is one bit per byte (the layout miphy_pdsch_encode_batch writes). Only the mapped REs of
 * the grid are written. */
typedef struct {
  uint64_t prb_mask[5]; /* PRBs (grid numbering) the pattern applies to */
  uint16_t re_mask;     /* bit k = subcarrier k of each of these PRBs */
  uint16_t symbols;     /* bit l = OFDM symbol l */
  uint32_t pad;
} miphy_re_pattern;

typedef struct {
  uint32_t rnti;
  uint32_t n_id;
  float    scaling;        /* applied when std::isnormal(scaling), like the reference */
  uint8_t  mod;            /* bits per symbol: 1 (pi/2-BPSK), 2, 4, 6, 8 */
  uint8_t  port;           /* grid port of the (single) layer */
  uint8_t  start_symbol;
  uint8_t  nof_symbols;
  uint8_t  dmrs_type;      /* 1 or 2 */
  uint8_t  nof_cdm_groups_without_data;
  uint8_t  nof_reserved;   /* 0..4 */
  uint8_t  reserved0;
  uint16_t dmrs_symbols_mask;
  uint16_t grid_nof_prb;
  uint16_t bwp_start_rb;   /* the DM-RS pattern covers [bwp_start_rb, bwp_start_rb + bwp_size_rb) */
  uint16_t bwp_size_rb;
  uint32_t nof_bits;       /* codeword length; must equal miphy_pdsch_mod_nof_re() * mod */
  uint64_t rb_mask[5];     /* allocated PRBs, grid numbering (rb_allocation::get_prb_mask) */
  miphy_re_pattern reserved[4];
  uint64_t cw_offset;      /* byte offset of the codeword (one bit per byte) */
  uint64_t grid_offset;    /* cf_t offset of grid port 0: [port][14][grid_nof_prb*12] */
} miphy_pdsch_mod_job;

uint32_t miphy_pdsch_mod_nof_re(const miphy_pdsch_mod_job* job); /* host: data REs of the allocation; 0 on an invalid job */
int miphy_pdsch_modulate_batch(miphy_ctx* ctx, const miphy_pdsch_mod_job* jobs, int jobs_on_device, uint32_t n,
                               const uint8_t* codewords /* device */, float* grid /* device cf_t */, void* stream);

typedef struct {
  uint32_t slot_in_frame;         /* slot.slot_index() */
  uint32_t reference_point_k_rb;
  uint32_t scrambling_id;
  float    amplitude;             /* linear amplitude of the DM-RS (the sequence is +-amplitude/sqrt(2)) */
  uint8_t  dmrs_type;             /* 1 or 2 */
  uint8_t  n_scid;
  uint8_t  nof_ports;             /* DM-RS ports 1000 .. 1000 + nof_ports - 1 */
  uint8_t  reserved0;
  uint8_t  ports[12];             /* grid port of each DM-RS port */
  uint16_t symbols_mask;
  uint16_t grid_nof_prb;
  uint32_t pad;
  uint64_t rb_mask[5];
  uint64_t grid_offset;           /* cf_t offset of grid port 0 */
} miphy_dmrs_pdsch_job;

int miphy_dmrs_pdsch_map_batch(miphy_ctx* ctx, const miphy_dmrs_pdsch_job* jobs, int jobs_on_device, uint32_t n, float* grid /* device cf_t */,
                               void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Polar code chains  --  replace the chain of srsran::polar_code::set + polar_allocator::allocate + polar_encoder::encode
 * + polar_rate_matcher::rate_match (transmit) and polar_rate_dematcher::rate_dematch + polar_decoder::decode +
 * polar_deallocator::deallocate (receive), as wired in tests/unittests/phy/upper/channel_coding/polar/polar_chain_test.cpp:156-210
 *   include/srsran/phy/upper/channel_coding/polar/polar_{code,allocator,encoder,rate_matcher,rate_dematcher,decoder,deallocator}.h
 *   lib/phy/upper/channel_coding/polar/polar_code_impl.cpp:325-490 and the *_impl.cpp files next to it.
 * The decoder is the reference's simplified successive cancellation (list size 1) with rate-0 / rate-1 node pruning.
 * All codewords of one call share the code (K, E, nMax, ibil), like one polar_code object does.
 * msg: n x K bytes (one bit per byte); rm: n x E bytes / LLRs. Optional taps (may be NULL) expose the intermediate
 * results with the layouts of the single reference blocks: `allocated` / `encoded` / `decoded_u`: n x N bytes,
 * `dematched`: n x N LLRs. */
typedef struct {
  uint32_t K;    /* message length incl. CRC */
  uint32_t E;    /* rate-matched length */
  uint32_t nMax; /* 9 (downlink) or 10 (uplink) */
  uint32_t ibil; /* channel interleaver present (uplink) */
} miphy_polar_code;

/* Single blocks of the chains, for the block-level interfaces (one wavefront per codeword, n codewords back to back):
 *   ALLOCATE      polar_allocator::allocate        (polar_allocator.h:42-43)        in: n x K bits      out: n x N bits
 *   ENCODE        polar_encoder::encode            (polar_encoder.h:43)             in: n x 2^param     out: n x 2^param   (code may be NULL)
 *   RATE_MATCH    polar_rate_matcher::rate_match   (polar_rate_matcher.h:42-43)     in: n x N bits      out: n x E bits
 *   RATE_DEMATCH  polar_rate_dematcher::rate_dematch (polar_rate_dematcher.h:45-47) in: n x E LLRs      out: n x N LLRs
 *   DECODE        polar_decoder::decode            (polar_decoder.h:47-48)          in: n x N LLRs      out: n x N bits (u domain)
 *   DEALLOCATE    polar_deallocator::deallocate    (polar_deallocator.h:41-42)      in: n x N bits      out: n x K bits
 *   INTERLEAVE_TX / _RX  polar_interleaver::interleave (polar_interleaver.h:45-46)  in / out: n x param bits, param = K <= 164 (code may be NULL)
 * One bit per byte, LLRs int8, everything in device memory. */
enum {
  MIPHY_POLAR_OP_ALLOCATE = 0,
  MIPHY_POLAR_OP_ENCODE,
  MIPHY_POLAR_OP_RATE_MATCH,
  MIPHY_POLAR_OP_RATE_DEMATCH,
  MIPHY_POLAR_OP_DECODE,
  MIPHY_POLAR_OP_DEALLOCATE,
  MIPHY_POLAR_OP_INTERLEAVE_TX,
  MIPHY_POLAR_OP_INTERLEAVE_RX
};
/* Host-side query of the derived code parameters (polar_code::get_n / get_N / get_nPC). Returns 0 or MIPHY_EINVAL. */
int miphy_polar_code_info(const miphy_polar_code* code, uint32_t* n, uint32_t* N, uint32_t* nPC);

int miphy_polar_encode_batch(miphy_ctx* ctx, const miphy_polar_code* code, uint32_t n, const uint8_t* msg /* device */,
                             uint8_t* rm_out /* device */, uint8_t* allocated_tap, uint8_t* encoded_tap, void* stream);
int miphy_polar_block_batch(miphy_ctx* ctx, const miphy_polar_code* code, uint32_t op, uint32_t param, uint32_t n, const void* in /* device */,
                            void* out /* device */, void* stream);
int miphy_polar_decode_batch(miphy_ctx* ctx, const miphy_polar_code* code, uint32_t n, const int8_t* llr /* device */,
                             uint8_t* msg_out /* device */, int8_t* dematched_tap, uint8_t* decoded_u_tap, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * PDCCH encoder  --  replaces srsran::pdcch_encoder::encode
 *   include/srsran/phy/upper/channel_processors/pdcch_encoder.h:36-53, lib/phy/upper/channel_processors/pdcch_encoder_impl.cpp:33-98
 * (CRC24C over 24 leading ones + payload, RNTI scrambling of the last 16 CRC bits, CRC interleaver, polar chain with
 * nMax = 9). payload: n x A bytes (one bit per byte); rnti: n entries; out: n x E bytes. */
int miphy_pdcch_encode_batch(miphy_ctx* ctx, uint32_t A, uint32_t E, uint32_t n, const uint8_t* payload /* device */,
                             const uint16_t* rnti /* device */, uint8_t* out /* device */, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * PUSCH decoder (whole transport blocks)  --  replaces srsran::pusch_decoder::decode
 *   include/srsran/phy/upper/channel_processors/pusch_decoder.h:41-78
 *   lib/phy/upper/channel_processors/pusch_decoder_impl.cpp:121-225 (segment_rx -> per codeblock rate-dematch + LDPC decode
 *   with the per-codeblock CRC, skip of codeblocks already decoded in a previous transmission, TB assembly, TB CRC24A,
 *   reset of the codeblock CRC flags when the TB CRC fails), ldpc_segmenter_impl.cpp:253-334 for the segmentation.
 * The HARQ state the reference keeps in an rx_softbuffer (include/srsran/phy/upper/rx_softbuffer.h:42-72) lives in three
 * caller-owned DEVICE arrays: soft bits [codeblock][N], decoded codeblock messages [codeblock][ceil(K/8)] and CRC flags
 * [codeblock]; a transport block addresses its slice through `harq_cb_index` (index of its first codeblock).
 * The transport block bytes are written only when every codeblock CRC passed (like the reference). */
typedef struct {
  uint8_t  bg;                  /* segmenter_cfg.base_graph: 1 or 2 */
  uint8_t  rv;                  /* segmenter_cfg.rv */
  uint8_t  mod;                 /* bits per symbol of segmenter_cfg.mod */
  uint8_t  nof_layers;          /* segmenter_cfg.nof_layers */
  uint8_t  new_data;            /* configuration::new_data */
  uint8_t  use_early_stop;      /* configuration::use_early_stop */
  uint16_t nof_ldpc_iterations; /* configuration::nof_ldpc_iterations */
  uint32_t Nref;                /* segmenter_cfg.Nref */
  uint32_t nof_ch_symbols;      /* segmenter_cfg.nof_ch_symbols */
  uint32_t tb_bytes;            /* transport_block.size() */
  uint32_t harq_cb_index;       /* first codeblock slot of this TB in the HARQ arrays */
  uint64_t llr_offset;          /* element offset of the nof_ch_symbols*mod codeword LLRs inside `llrs` */
  uint64_t tb_offset;           /* byte offset of the transport block inside `tb_out` */
} miphy_pusch_tb_desc;

typedef struct {              /* srsran::pusch_decoder_result */
  int32_t  tb_crc_ok;
  uint32_t nof_codeblocks_total;
  uint32_t iters_min;           /* ldpc_decoder_stats over the codeblocks decoded in this call (0 when none) */
  uint32_t iters_max;
  float    iters_mean;
  uint32_t nof_decoded;         /* number of observations */
} miphy_pusch_result;

/* Segmentation parameters of a transport block (ldpc::compute_nof_codeblocks / compute_lifting_size /
 * compute_codeblock_size, include/srsran/phy/upper/channel_coding/ldpc/ldpc.h:128-207). Host only. */
typedef struct {
  uint32_t nof_cbs, Z, K, N, nof_filler_bits, nof_tb_crc_bits, nof_cb_crc_bits, cb_info_bits, zero_pad;
} miphy_sch_segmentation;
int miphy_sch_segmentation_info(uint32_t tb_bytes, uint32_t bg, miphy_sch_segmentation* out);

int miphy_pusch_decode_batch(miphy_ctx*                 ctx,
                             const miphy_pusch_tb_desc* tbs, /* host */
                             uint32_t                   n,
                             const int8_t*              llrs,          /* device */
                             int8_t*                    harq_softbits, /* device, in/out, stride 66*384 per codeblock slot */
                             uint8_t*                   harq_msgs,     /* device, in/out, stride 1056 B per codeblock slot */
                             uint8_t*                   harq_crc_ok,   /* device, in/out, 1 B per codeblock slot */
                             uint8_t*                   tb_out,        /* device */
                             miphy_pusch_result*        results,       /* device, n entries */
                             void*                      stream);

/* Prepared form of miphy_pusch_decode_batch for allocations that repeat (semi-static scheduling, benchmarks): _create runs the
 * segmentation (ldpc_segmenter_impl.cpp:253-334) and uploads the codeblock descriptors once; every _run is kernel launches only
 * (CRC-flag reset, rate dematching, LDPC decoding, transport-block assembly + TB CRC + result records) with no host
 * synchronisation, on the arrays passed to that run. A plan belongs to the context it was created on and must not run
 * concurrently with itself. */
typedef struct miphy_pusch_decode_plan miphy_pusch_decode_plan;
int  miphy_pusch_decode_plan_create(miphy_ctx* ctx, const miphy_pusch_tb_desc* tbs /* host */, uint32_t n, miphy_pusch_decode_plan** out);
int  miphy_pusch_decode_plan_run(miphy_pusch_decode_plan* plan,
                                 const int8_t*            llrs,          /* device */
                                 int8_t*                  harq_softbits, /* device, in/out */
                                 uint8_t*                 harq_msgs,     /* device, in/out */
                                 uint8_t*                 harq_crc_ok,   /* device, in/out */
                                 uint8_t*                 tb_out,        /* device */
                                 miphy_pusch_result*      results,       /* device, n entries */
                                 void*                    stream);
void miphy_pusch_decode_plan_destroy(miphy_pusch_decode_plan* plan);
/* Optional per-kernel timing of the runs of a plan (measurement aid, e.g. bench.py's roofline): _enable_timing makes every run
 * record HIP events on its stream around the rate dematcher, the LDPC decoder and the transport-block assembly (a ring of max_runs
 * runs; batches above 65535 codeblocks are refused); _read_timing waits for the recorded runs, returns their mean durations in
 * milliseconds {rate dematch, LDPC decode, TB assembly} and starts a new series. */
int  miphy_pusch_decode_plan_enable_timing(miphy_pusch_decode_plan* plan, uint32_t max_runs);
int  miphy_pusch_decode_plan_read_timing(miphy_pusch_decode_plan* plan, float ms_out[3], uint32_t* runs_out);
/* What the plan will launch: info[0] = codeblocks, info[1] = 1 when every codeblock is rate-dematched while the decoder loads it (one
 * kernel reads the rate-matched LLRs and writes the soft-buffer image; the "rate dematch" time of _read_timing is then only the HARQ
 * flag reset), 0 when the dematcher runs as its own launch, info[2] = largest number of variable nodes a codeblock can reach. */
int  miphy_pusch_decode_plan_info(const miphy_pusch_decode_plan* plan, uint32_t info[3]);
/* LDPC decoder launches per run: the codeblocks of the batch are sorted into launch classes (lifting size, base graph, reachable
 * layers, dematch-in-decoder or not), one launch each. A batch of identical allocations has one. */
uint32_t miphy_pusch_decode_plan_nof_launches(const miphy_pusch_decode_plan* plan);

/* ------------------------------------------------------------------------------------------------------------------
 * PUSCH processor  --  replaces srsran::pusch_processor::process for PDUs without UCI
 *   include/srsran/phy/upper/channel_processors/pusch_processor.h:84-162
 *   lib/phy/upper/channel_processors/pusch_processor_impl.cpp:108-330
 * One call = channel estimation + demodulation + transport-block decoding of n PDUs; channel estimates and LLRs never leave the
 * device. Restrictions of the reference apply (pusch_processor_impl.cpp:96-104): DM-RS type 1, two CDM groups without data, one
 * transmit layer. HARQ arrays and `results` as in miphy_pusch_decode_batch; scalars_out: per PDU [4 rx ports][5] floats
 * {rsrp, epre, noise_var, snr, time_alignment_s} of layer 0 (what channel_estimate::get_channel_state_information averages). */
typedef struct {
  uint32_t numerology;
  uint32_t slot_in_frame;
  uint32_t rnti;
  uint32_t n_id;
  uint32_t dmrs_scrambling_id;
  uint32_t Nref;                /* tbs_lbrm_bytes * 8 */
  uint32_t tb_bytes;
  uint32_t harq_cb_index;
  uint8_t  n_scid;
  uint8_t  mod;                 /* bits per symbol */
  uint8_t  nof_rx_ports;
  uint8_t  start_symbol;
  uint8_t  nof_symbols;
  uint8_t  bg;                  /* codeword.ldpc_base_graph */
  uint8_t  rv;
  uint8_t  new_data;
  uint8_t  rx_ports[4];
  uint8_t  use_early_stop;
  uint8_t  reserved0;
  uint16_t nof_ldpc_iterations;
  uint16_t dmrs_symbols_mask;
  uint16_t grid_nof_prb;
  uint32_t pad;
  uint64_t rb_mask[5];          /* freq_alloc.get_prb_mask(bwp_start_rb, bwp_size_rb) */
  uint64_t grid_offset;         /* cf_t offset of grid port 0: [port][14][grid_nof_prb*12] */
  uint64_t tb_offset;           /* byte offset of the transport block inside `tb_out` */
} miphy_pusch_pdu;

/* UCI multiplexed on the PUSCH of a PDU (pusch_processor::uci_description + the lengths ulsch_information derives from it,
 * include/srsran/ran/pusch/ulsch_info.h: the caller -- the adapter through the reference's own get_ulsch_information() -- supplies them). */
typedef struct {
  uint32_t nof_harq_ack_bits, nof_csi_part1_bits, nof_csi_part2_bits;             /* O: information bits (0: field absent) */
  uint32_t nof_enc_harq_ack_bits, nof_enc_csi_part1_bits, nof_enc_csi_part2_bits; /* G: encoded, rate-matched bits */
  uint32_t nof_harq_ack_rvd;                                                      /* G^HARQ-ACK_rvd */
  uint32_t has_codeword;                                                          /* 0: the PDU carries no transport block (UCI only) */
  uint64_t harq_ack_offset, csi_part1_offset, csi_part2_offset;                   /* int8 offsets of the three soft-bit streams in uci_llr_out */
} miphy_pusch_uci;

/* _ex: PDUs with multiplexed UCI and the EVM. `uci`: NULL or n entries (host). The soft bits of the UCI fields go to `uci_llr_out`
 * (device; decoding them stays with the caller: uci_decoder is not on this path), `evm_out` (device, n floats, may be NULL) receives
 * the error vector magnitude of every PDU (pusch_demodulator::demodulation_status::evm). PDUs without codeword are demodulated and
 * demultiplexed only; their result record is zero. */
int miphy_pusch_process_batch_ex(miphy_ctx* ctx, const miphy_pusch_pdu* pdus /* host */, const miphy_pusch_uci* uci /* host or NULL */, uint32_t n,
                                 const float* grid, int8_t* harq_softbits, uint8_t* harq_msgs, uint8_t* harq_crc_ok, uint8_t* tb_out,
                                 miphy_pusch_result* results, float* scalars_out, int8_t* uci_llr_out, float* evm_out, void* stream);
int miphy_pusch_process_batch(miphy_ctx* ctx, const miphy_pusch_pdu* pdus /* host */, uint32_t n, const float* grid /* device cf_t */,
                              int8_t* harq_softbits, uint8_t* harq_msgs, uint8_t* harq_crc_ok, uint8_t* tb_out /* device */,
                              miphy_pusch_result* results /* device, n */, float* scalars_out /* device, n x 20 */, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * UL-SCH demultiplexer  --  replaces srsran::ulsch_demultiplex::demultiplex / get_placeholders (UCI multiplexed on PUSCH)
 *   include/srsran/phy/upper/channel_processors/ulsch_demultiplex.h:38-110, ulsch_placeholder_list.h:35-105
 *   lib/phy/upper/channel_processors/ulsch_demultiplex_impl.cpp:74-453 (TS 38.212 6.2.7)
 * One job = one PUSCH codeword: the descrambled LLRs of the demodulator are split into the UL-SCH data stream (all-zero
 * elements where HARQ-ACK punctures reserved resource elements) and the HARQ-ACK / CSI part 1 / CSI part 2 streams. */
typedef struct {
  uint8_t  mod;                         /* bits per symbol */
  uint8_t  nof_layers;                  /* 1..4 */
  uint8_t  start_symbol;
  uint8_t  nof_symbols;
  uint8_t  dmrs_type;                   /* 1 or 2 */
  uint8_t  nof_cdm_groups_without_data;
  uint16_t dmrs_symbols_mask;           /* bit l = OFDM symbol l carries DM-RS */
  uint16_t nof_prb;                     /* PRBs allocated to the transmission */
  uint16_t reserved;
  uint32_t nof_harq_ack_rvd;            /* G^HARQ-ACK_rvd: bits reserved for HARQ-ACK (0 when more than two HARQ-ACK bits are sent) */
  uint32_t nof_enc_harq_ack_bits;       /* G^HARQ-ACK: encoded, rate-matched bits of each field = length of its output stream */
  uint32_t nof_enc_csi_part1_bits;
  uint32_t nof_enc_csi_part2_bits;
  uint32_t nof_harq_ack_bits;           /* O: information bits of each field (a field with exactly one bit has repetition placeholders) */
  uint32_t nof_csi_part1_bits;
  uint32_t nof_csi_part2_bits;
  uint64_t in_offset;                   /* int8 offset of the codeword LLRs */
  uint64_t sch_offset;                  /* int8 offsets of the four output streams */
  uint64_t harq_ack_offset;
  uint64_t csi_part1_offset;
  uint64_t csi_part2_offset;
} miphy_ulsch_demux_job;

/* Host: number of LLRs of the codeword (input) and of the UL-SCH data stream (output) of a job. MIPHY_EINVAL for an impossible
 * configuration (the reference asserts: fields that do not fit the allocation). */
int miphy_ulsch_demux_sizes(const miphy_ulsch_demux_job* job, uint32_t* nof_in_llr, uint32_t* nof_sch_llr);
/* Host: ulsch_demultiplex::get_placeholders -- indices (in the order of the input resource elements) of the elements that carry a
 * repetition placeholder; at most `cap` are written, *n is the full count. */
int miphy_ulsch_placeholders(const miphy_ulsch_demux_job* job, uint16_t* re_indices, uint32_t cap, uint32_t* n);
int miphy_ulsch_demultiplex_batch(miphy_ctx* ctx, const miphy_ulsch_demux_job* jobs /* host */, uint32_t n, const int8_t* llr_in /* device */,
                                  int8_t* sch_out, int8_t* harq_ack_out, int8_t* csi_part1_out, int8_t* csi_part2_out /* device */, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * PDSCH encoder (whole transport blocks)  --  replaces srsran::pdsch_encoder::encode
 *   include/srsran/phy/upper/channel_processors/pdsch_encoder.h, lib/phy/upper/channel_processors/pdsch_encoder_impl.cpp:28-65
 *   (segment_tx: TB CRC16/24A, CB CRC24B, zero padding, fillers -> LDPC encode -> rate match into the codeword),
 *   ldpc_segmenter_impl.cpp:89-234.
 * tb_in: packed transport blocks; codeword_out: one bit per byte, nof_ch_symbols*mod bytes per TB. */
typedef struct {
  uint8_t  bg;
  uint8_t  rv;
  uint8_t  mod;
  uint8_t  nof_layers;
  uint32_t Nref;
  uint32_t nof_ch_symbols;
  uint32_t tb_bytes;
  uint64_t tb_offset;       /* byte offset of the packed transport block inside `tb_in` */
  uint64_t codeword_offset; /* byte offset of the codeword inside `codeword_out` */
} miphy_pdsch_tb_desc;

int miphy_pdsch_encode_batch(miphy_ctx* ctx, const miphy_pdsch_tb_desc* tbs /* host */, uint32_t n, const uint8_t* tb_in /* device */,
                             uint8_t* codeword_out /* device */, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * PDSCH processor (whole PDUs)  --  replaces srsran::pdsch_processor::process
 *   include/srsran/phy/upper/channel_processors/pdsch_processor.h:64-165, lib/phy/upper/channel_processors/pdsch_processor_impl.cpp:110-305
 * Transport block -> resource grid in one call: the number of data REs of the allocation (reserved patterns and the DM-RS
 * pattern of the bandwidth part excluded, pdsch_processor_impl.cpp:198-220), pdsch_encoder (Nref = 8 * tbs_lbrm_bytes),
 * pdsch_modulator with scaling 10^(-ratio_pdsch_data_to_sss_dB / 20) and dmrs_pdsch_processor with amplitude
 * 10^(-ratio_pdsch_dmrs_to_sss_dB / 20) on the same port. The codewords stay on the device. Same restrictions as the
 * reference (pdsch_processor_impl.cpp:143-196): DM-RS type 1, one codeword on one layer, contiguous allocation. */
typedef struct {
  uint32_t slot_in_frame;              /* pdu.slot.slot_index() */
  uint32_t rnti;
  uint32_t n_id;                       /* data scrambling identity */
  uint32_t dmrs_scrambling_id;
  uint32_t tbs_lbrm_bytes;             /* 1 .. 66*384/8 */
  uint32_t tb_bytes;
  float    ratio_pdsch_dmrs_to_sss_dB;
  float    ratio_pdsch_data_to_sss_dB;
  uint8_t  bg;                         /* 1 or 2 */
  uint8_t  rv;
  uint8_t  mod;                        /* bits per symbol */
  uint8_t  port;                       /* grid port of the layer */
  uint8_t  start_symbol;
  uint8_t  nof_symbols;
  uint8_t  nof_cdm_groups_without_data;
  uint8_t  n_scid;
  uint8_t  ref_point_prb0;             /* 1: DM-RS reference point is the first PRB of the BWP (pdu_t::PRB0), 0: CRB0 */
  uint8_t  nof_reserved;               /* 0..4 */
  uint16_t dmrs_symbols_mask;
  uint16_t grid_nof_prb;
  uint16_t bwp_start_rb;
  uint16_t bwp_size_rb;
  uint16_t pad;
  uint64_t rb_mask[5];                 /* allocated PRBs, grid numbering */
  miphy_re_pattern reserved[4];
  uint64_t tb_offset;                  /* byte offset of the packed transport block inside `tb_in` */
  uint64_t grid_offset;                /* cf_t offset of grid port 0: [port][14][grid_nof_prb*12] */
} miphy_pdsch_pdu;

/* Data REs of the allocation (pdsch_processor_impl::compute_nof_data_re); 0 on an invalid PDU. Host function. */
uint32_t miphy_pdsch_pdu_nof_re(const miphy_pdsch_pdu* pdu);
int miphy_pdsch_process_batch(miphy_ctx* ctx, const miphy_pdsch_pdu* pdus /* host */, uint32_t n, const uint8_t* tb_in /* device */,
                              float* grid /* device cf_t; only the mapped REs are written */, void* stream);
/* Prepared form for allocations that repeat slot after slot: PDU validation, segmentation and descriptor uploads once, a run only
 * launches (TB CRC, codeblock preparation, LDPC encoder, rate matcher, modulator, DM-RS), no staging, no host synchronisation. */
typedef struct miphy_pdsch_process_plan miphy_pdsch_process_plan;
int  miphy_pdsch_process_plan_create(miphy_ctx* ctx, const miphy_pdsch_pdu* pdus /* host */, uint32_t n, miphy_pdsch_process_plan** out);
int  miphy_pdsch_process_plan_run(miphy_pdsch_process_plan* plan, const uint8_t* tb_in /* device */, float* grid /* device cf_t */, void* stream);
void miphy_pdsch_process_plan_destroy(miphy_pdsch_process_plan* plan);

/* ------------------------------------------------------------------------------------------------------------------
 * PDCCH processor (whole PDUs)  --  replaces srsran::pdcch_processor::process after the CCE-to-PRB mapping
 *   include/srsran/phy/upper/channel_processors/pdcch_processor.h:41-152, lib/phy/upper/channel_processors/pdcch_processor_impl.cpp:65-117,
 *   pdcch_modulator_impl.cpp:30-91, lib/phy/upper/signal_processors/dmrs_pdcch_processor_impl.cpp:30-101, dmrs_helper.h:44-96
 * DCI payload -> resource-grid REs in one call: pdcch_encoder (CRC24C with the RNTI mask, interleaver, polar code, E = 108 x
 * aggregation level), scrambling with c_init = (n_rnti << 16) + n_id_data, QPSK, scaling 10^(data_power_offset_dB / 20), mapping on
 * REs {0,2,3,4,6,7,8,10,11} of the PRBs of `rb_mask` over `duration` symbols (symbol by symbol, ascending subcarrier), and the DM-RS
 * of every symbol (c_init from slot, symbol and n_id_dmrs; three pilots per PRB on REs 1, 5, 9, sequence counted from the
 * reference point, amplitude 10^(dmrs_power_offset_dB / 20) / sqrt(2)). rb_mask is the result of the reference's CCE-to-PRB
 * mapping (pdcch_processor_impl::compute_rb_mask, host bookkeeping that stays with the caller). Normal cyclic prefix, one port. */
typedef struct {
  uint32_t slot_in_frame;         /* pdu.slot.slot_index() */
  uint32_t rnti;                  /* dci.rnti: CRC mask */
  uint32_t n_id_pdcch_data;
  uint32_t n_rnti;
  uint32_t n_id_pdcch_dmrs;
  uint32_t reference_point_k_rb;  /* bwp_start_rb for CORESET 0, else 0 (pdcch_processor_impl.cpp:98-99) */
  float    data_power_offset_dB;
  float    dmrs_power_offset_dB;
  uint16_t payload_size;          /* 12..128 bits */
  uint8_t  aggregation_level;     /* 1, 2, 4, 8 or 16 */
  uint8_t  start_symbol;          /* coreset.start_symbol_index */
  uint8_t  duration;              /* 1..3 */
  uint8_t  port;                  /* grid port */
  uint16_t grid_nof_prb;
  uint64_t rb_mask[5];            /* 6 x aggregation_level / duration PRBs */
  uint64_t payload_offset;        /* byte offset of the payload bits (one per byte) inside `payloads` */
  uint64_t grid_offset;           /* cf_t offset of grid port 0: [port][14][grid_nof_prb*12] */
  uint64_t work_offset;           /* set by the library */
} miphy_pdcch_pdu;

int miphy_pdcch_process_batch(miphy_ctx* ctx, const miphy_pdcch_pdu* pdus /* host */, uint32_t n, const uint8_t* payloads /* device */,
                              float* grid /* device cf_t; only the mapped REs are written */, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * PBCH encoder  --  replaces srsran::pbch_encoder::encode
 *   include/srsran/phy/upper/channel_processors/pbch_encoder.h:53-80, lib/phy/upper/channel_processors/pbch_encoder_impl.cpp:41-190
 * (payload interleaving G(j) with SFN / half-frame / SSB-index bits, Gold-sequence scrambling, CRC24C, CRC interleaver,
 * polar chain K = 56, E = 864). The 32-bit payload generation and scrambling are bit bookkeeping done on the host; CRC,
 * interleaving and the polar chain run on the device. msgs: host array; out: n x 864 bytes (one bit per byte), device. */
typedef struct {
  uint32_t N_id;        /* physical cell identifier */
  uint32_t ssb_idx;
  uint32_t L_max;       /* 4, 8 or 64 */
  uint32_t hrf;         /* half-frame flag */
  uint32_t sfn;
  uint32_t k_ssb;
  uint8_t  payload[32]; /* one bit per byte (only the first 24 are used, like the reference) */
} miphy_pbch_msg;

int miphy_pbch_encode_batch(miphy_ctx* ctx, const miphy_pbch_msg* msgs /* host */, uint32_t n, uint8_t* out /* device */, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * SS/PBCH block processor  --  replaces srsran::ssb_processor::process after the position look-up
 *   include/srsran/phy/upper/channel_processors/ssb_processor.h:41-80, lib/phy/upper/channel_processors/ssb_processor_impl.cpp:30-106,
 *   pbch_modulator_impl.cpp:28-113, lib/phy/upper/signal_processors/dmrs_pbch_processor_impl.cpp:28-100, pss_processor_impl.cpp:28-91,
 *   sss_processor_impl.cpp:28-119
 * One call: PBCH encoding (miphy_pbch_encode_batch), scrambling with the cell identity advanced by (ssb_idx & 7) * 864, QPSK on the
 * 432 PBCH REs, the 144 PBCH DM-RS (c_init from ssb_idx / half frame / cell identity, +-1/sqrt(2)), PSS (127 REs, amplitude
 * 10^(beta_pss / 20)) and SSS, written to every listed port. ssb_first_symbol / ssb_first_subcarrier are the reference's
 * ssb_get_l_first() % 14 and ssb_get_k_first() (include/srsran/ran/ssb_mapping.h), host bookkeeping that stays with the caller. */
typedef struct {
  miphy_pbch_msg msg;             /* N_id = physical cell identity, ssb_idx, L_max, hrf, sfn, k_ssb, payload */
  uint32_t ssb_first_subcarrier;
  uint32_t ssb_first_symbol;      /* 0..10 */
  float    beta_pss_dB;
  uint16_t grid_nof_prb;
  uint8_t  nof_ports;             /* 1..4 */
  uint8_t  ports[4];
  uint8_t  pad;
  uint64_t grid_offset;           /* cf_t offset of grid port 0: [port][14][grid_nof_prb*12] */
} miphy_ssb_pdu;

int miphy_ssb_process_batch(miphy_ctx* ctx, const miphy_ssb_pdu* pdus /* host */, uint32_t n, float* grid /* device cf_t; only the SSB REs are written */,
                            void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * NZP-CSI-RS generator  --  replaces srsran::nzp_csi_rs_generator::map after the pattern look-up
 *   include/srsran/phy/upper/signal_processors/nzp_csi_rs_generator.h:37-91, lib/phy/upper/signal_processors/nzp_csi_rs_generator_impl.cpp:34-296
 * Per port and OFDM symbol of its pattern: Gold sequence with c_init = (2^10 (14 n_slot + l + 1)(2 n_id + 1) + n_id) mod 2^31, the
 * elements below the first occupied PRB skipped (:69-108), QPSK of amplitude `amplitude` / sqrt(2), the CDM weights w_t[l'] w_f[k']
 * of the port's index in its CDM group (:34-57, 223-296), written to the REs of the port's pattern inside [start_rb, start_rb + nof_rb).
 * The per-port patterns (rb_begin / rb_end / rb_stride, RE mask, symbol mask) are the output of the reference's get_csi_rs_pattern()
 * (TS 38.211 Table 7.4.1.5.3-1 bookkeeping, include/srsran/ran/csi_rs/csi_rs_pattern.h), which stays with the caller. */
typedef struct {
  uint32_t slot_in_frame;
  uint32_t scrambling_id;
  float    amplitude;
  uint16_t start_rb;
  uint16_t nof_rb;
  uint16_t rb_begin;       /* csi_rs_pattern */
  uint16_t rb_end;
  uint16_t rb_stride;
  uint16_t grid_nof_prb;
  uint8_t  mapping_row;    /* csi_rs_mapping_table_row (only row 2 changes the sequence offset) */
  uint8_t  cdm;            /* csi_rs_cdm_type: 0 none, 1 FD-CDM2, 2 CDM4-FD2-TD2, 3 CDM8-FD2-TD4 */
  uint8_t  freq_density;   /* csi_rs_freq_density_type: 0 0.5 even PRBs, 1 0.5 odd PRBs, 2 one, 3 three */
  uint8_t  nof_ports;      /* 1..16 */
  uint8_t  ports[16];      /* grid port of each CSI-RS port */
  uint16_t re_mask[16];    /* csi_rs_pattern_port::re_mask, bit k = subcarrier k of the PRB */
  uint16_t symbol_mask[16];
  uint64_t grid_offset;    /* cf_t offset of grid port 0: [port][14][grid_nof_prb*12] */
} miphy_csi_rs_job;

int miphy_csi_rs_map_batch(miphy_ctx* ctx, const miphy_csi_rs_job* jobs, int jobs_on_device, uint32_t n, float* grid /* device cf_t */, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Polar successive-cancellation LIST decoder (list size 1, 2, 4 or 8), optionally CRC-aided  --  no counterpart in the
 * reference (its polar_decoder is the list-size-1 SSC decoder that miphy_polar_decode_batch reproduces bit for bit); this
 * is the SCL-8 path BASELINE.json's north_star / configs[3] ask for. Same rate-dematcher and LLR algebra as the reference;
 * path metric PM += |llr| whenever a decision disagrees with the hard decision; candidates ranked by (metric, 2*slot+flip).
 *   crc_mode 0: msg_out = K bits of the best-metric path in K-set order (the format of miphy_polar_decode_batch);
 *   crc_mode 1: PDCCH -- every surviving path is CRC de-interleaved (TS 38.212 5.3.1.1) and checked with CRC24C over 24
 *               leading ones + payload and the RNTI mask (pdcch_encoder_impl.cpp:33-59 inverted); the best-metric path that
 *               passes wins; msg_out = the K de-interleaved bits (payload then masked CRC), crc_ok_out = 1 on a pass;
 *   crc_mode 2: PBCH -- the same without leading ones / RNTI.
 * rnti may be NULL unless crc_mode == 1. Bit-exact against oracle/phy_oracle.c::orc_polar_scl_decode. */
int miphy_polar_decode_list_batch(miphy_ctx* ctx, const miphy_polar_code* code, uint32_t list_size, uint32_t crc_mode, uint32_t n,
                                  const int8_t* llr /* device, n x E */, const uint16_t* rnti /* device, n */,
                                  uint8_t* msg_out /* device, n x K */, uint8_t* crc_ok_out /* device, n */,
                                  int32_t* metric_out /* device, n; may be NULL */, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Open Fronthaul IQ (de)compression  --  replaces srsran::ofh::iq_decompressor::decompress and
 * srsran::ofh::iq_compressor::compress for the two methods the reference implements, compression_type::none (fixed point)
 * and compression_type::BFP (block floating point) (U-plane payloads to / from the resource grid in device memory)
 *   include/srsran/ofh/compression/iq_decompressor.h:35-52, iq_compressor.h:35-52, compressed_prb.h:36-80, compression_params.h:38-53,
 *   lib/ofh/compression/iq_compression_bfp_impl.cpp:28-143, iq_compression_bfp_avx2.cpp:31-132, iq_compression_none_impl.cpp:29-69,
 *   compressed_prb.cpp:31-79, quantizer.h:34-100, lib/srsvec/conversion.cpp:61-130
 * A job is one U-plane section: `nof_prb` consecutive PRB records as ofh_uplane_message_builder_impl.cpp:145-152 puts them on the
 * wire -- BFP: [udCompParam][24 x data_width bits, big endian] (1 + 3 * data_width bytes), none: the 3 * data_width bytes of
 * samples only -- and the nof_prb * 12 subcarriers of one (port, symbol) row of a resource grid they belong to.
 * Decompression: BFP sample = sign_extend(bits) * 2^(udCompParam & 15) / 32767; none sample = sign_extend(bits) / (2^(w-1) - 1).
 * `simd_arithmetic` != 0 reproduces the reference's production BFP classes ("avx2" / "avx512": data_width 9 multiplies by the
 * rounded reciprocal 1 / (32767 / 2^e)), 0 its generic class (a division for every width); both bit for bit. (For 9-bit samples
 * and exponents 0..7 the two forms give the same single-precision value, tests/test_oracle_golden.py checks that exhaustively;
 * the switch only matters for exponents a compliant RU does not send.)
 * Compression (data_width 8..16; the reference's packing asserts below that): quantisation through srsvec::convert_round exactly
 * (round to nearest even with saturation for the SIMD part of the converted span, round-half-away with a wrapping conversion for
 * its tail: BFP converts the whole job with scale 32767 * iq_scaling and 16-bit range, so the tail is the last 24 * nof_prb mod 16
 * values; none converts PRB by PRB with scale (2^(w-1) - 1) * iq_scaling, so the tail is the last 8 values of every PRB), then for
 * BFP the exponent from the largest magnitude of the PRB and an arithmetic shift, then packing of the low data_width bits. */
enum { MIPHY_OFH_COMPRESSION_NONE = 0, MIPHY_OFH_COMPRESSION_BFP = 1 }; /* values of srsran::ofh::compression_type */

typedef struct {
  uint64_t payload_offset; /* byte offset of the first PRB record */
  uint64_t grid_offset;    /* cf_t offset of the first subcarrier */
  uint32_t nof_prb;        /* 1..275 */
  uint16_t data_width;     /* 1..16 (compression: 8..16) */
  uint16_t compression;    /* MIPHY_OFH_COMPRESSION_*; other methods -> MIPHY_EUNSUPP (the reference aborts on them) */
} miphy_ofh_iq_job;

/* Bytes of one PRB record: 3 * data_width, plus the udCompParam byte for BFP. Host function. */
uint32_t miphy_ofh_iq_record_bytes(uint32_t compression, uint32_t data_width);

int miphy_ofh_iq_decompress_batch(miphy_ctx* ctx, const miphy_ofh_iq_job* jobs, int jobs_on_device, uint32_t n, const uint8_t* payload /* device */,
                                  float* grid /* device cf_t */, int simd_arithmetic, void* stream);
int miphy_ofh_iq_compress_batch(miphy_ctx* ctx, const miphy_ofh_iq_job* jobs, int jobs_on_device, uint32_t n, const float* grid /* device cf_t */,
                                float iq_scaling, uint8_t* payload /* device */, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Device-resident HARQ softbuffer pool  --  replaces srsran::rx_softbuffer_pool / rx_softbuffer
 *   include/srsran/phy/upper/rx_softbuffer_pool.h:31-96, include/srsran/phy/upper/rx_softbuffer.h:42-72,
 *   lib/phy/upper/rx_softbuffer_pool_impl.cpp:27-69, lib/phy/upper/rx_softbuffer_impl.h:33-258
 * The pool owns the three HARQ arrays miphy_pusch_decode_batch / miphy_pusch_process_batch work on (soft bits, decoded
 * messages, codeblock CRC flags, see above) and hands out softbuffers keyed by (rnti, harq process) with the reference's
 * reservation rules and state machine:
 *   reserve:  the first softbuffer whose last identifier equals (rnti, harq_id) -- in any state -- else the first available
 *             one; fails (buffer = -1, like an invalid unique_rx_softbuffer) when that softbuffer is locked or the pool-wide
 *             codeblock budget cannot cover nof_codeblocks. A reservation with the same number of codeblocks keeps the
 *             softbuffer (and its contents); a different number returns its codeblocks to the budget first.
 *   lock / unlock / release: reserved -> locked -> reserved | released; a released softbuffer can be reserved again and is
 *             freed by the next run_slot.
 *   run_slot: frees released softbuffers and reserved ones whose expiry slot (reservation slot + expire_timeout_slots,
 *             modulo nof_slots_wrap, compared like slot_point) is <= slot.
 * The pool never touches the contents. Every softbuffer owns a fixed extent of max_codeblocks_per_buffer codeblock slots
 * (first_cb = buffer * max_codeblocks_per_buffer: the `harq_cb_index` of the transport-block descriptors), so reservations
 * never fragment; max_nof_codeblocks is enforced as a budget exactly like the reference's codeblock pool. Device memory:
 * max_softbuffers * max_codeblocks_per_buffer * (66*384 + 1056 + 1) bytes. Thread safe (one mutex, like the reference).
 * ctx == NULL creates a bookkeeping-only pool without device arrays (miphy_harq_pool_arrays then fails). */
typedef struct miphy_harq_pool miphy_harq_pool;

typedef struct {
  uint32_t max_softbuffers;           /* rx_softbuffer_pool_config::max_softbuffers */
  uint32_t max_nof_codeblocks;        /* rx_softbuffer_pool_config::max_nof_codeblocks (budget over all softbuffers) */
  uint32_t expire_timeout_slots;      /* rx_softbuffer_pool_config::expire_timeout_slots */
  uint32_t nof_slots_wrap;            /* period of the slot counter: 10240 << numerology (slot_point.h:122) */
  uint32_t max_codeblocks_per_buffer; /* extent of one softbuffer; 0 = 52 (MAX_NOF_SEGMENTS, codeblock_metadata.h:88) */
} miphy_harq_pool_config;

enum { MIPHY_HARQ_AVAILABLE = 0, MIPHY_HARQ_RESERVED = 1, MIPHY_HARQ_LOCKED = 2, MIPHY_HARQ_RELEASED = 3 };

typedef struct {
  uint32_t state; /* MIPHY_HARQ_* */
  uint32_t rnti;
  uint32_t harq_id;
  uint32_t nof_codeblocks;
  uint32_t first_cb;
  uint32_t expire_slot;
} miphy_harq_buffer_info;

int  miphy_harq_pool_create(miphy_ctx* ctx, const miphy_harq_pool_config* cfg, miphy_harq_pool** out);
void miphy_harq_pool_destroy(miphy_harq_pool* pool);
/* *buffer = softbuffer index or -1 (no softbuffer: the caller drops the transmission, as the reference does). */
int miphy_harq_pool_reserve(miphy_harq_pool* pool, uint32_t slot, uint32_t rnti, uint32_t harq_id, uint32_t nof_codeblocks,
                            int32_t* buffer, uint32_t* first_cb);
int miphy_harq_pool_lock(miphy_harq_pool* pool, int32_t buffer);    /* MIPHY_EINVAL unless reserved (reference: assertion) */
int miphy_harq_pool_unlock(miphy_harq_pool* pool, int32_t buffer);  /* locked -> reserved, otherwise no effect */
int miphy_harq_pool_release(miphy_harq_pool* pool, int32_t buffer); /* MIPHY_EINVAL unless reserved or locked */
int miphy_harq_pool_run_slot(miphy_harq_pool* pool, uint32_t slot);
int miphy_harq_pool_info(miphy_harq_pool* pool, int32_t buffer, miphy_harq_buffer_info* out);
/* Remaining codeblock budget. */
int miphy_harq_pool_free_codeblocks(miphy_harq_pool* pool, uint32_t* out);
/* The device arrays to pass as harq_softbits / harq_msgs / harq_crc_ok. */
int miphy_harq_pool_arrays(miphy_harq_pool* pool, int8_t** softbits, uint8_t** msgs, uint8_t** crc_ok);

#ifdef __cplusplus
}
#endif
#endif /* MIPHY_H */
