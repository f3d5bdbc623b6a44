// TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or called from the product path.
//
// C API over the *reference's own classes* (srsRAN_Project 23.5, compiled in place from
// /root/reference by oracle/build_ref.sh into oracle/_ref/).  Every function instantiates the
// reference implementation through the reference's public factories / headers and forwards
// plain pointers, so Python (ctypes) can
//   * generate the golden fixtures under tests/golden/ (oracle/gen_golden.py),
//   * validate the C restatement in oracle/phy_oracle.c,
//   * serve as `cpu_baseline.kind = "reference"` in bench.py (AVX2 path, timed on host cores).
// Nothing here re-implements an algorithm.
#include "srsran/ofh/compression/compression_factory.h"
#include "srsran/phy/generic_functions/dft_processor.h"
#include "srsran/phy/upper/resource_grid_mapper.h"
#include "srsran/ran/csi_rs/csi_rs_pattern.h"
#include "srsran/ran/pdcch/cce_to_prb_mapping.h"
#include "srsran/ran/ssb_mapping.h"
#include "srsran/ran/precoding/precoding_codebooks.h"
#include "srsran/phy/upper/channel_coding/channel_coding_factories.h"
#include "srsran/phy/upper/channel_processors/channel_processor_factories.h"
#include "srsran/phy/upper/rx_softbuffer_pool.h"
#include "srsran/phy/upper/unique_rx_softbuffer.h"
#include "srsran/srsvec/bit.h"
#include "srsran/phy/lower/modulation/modulation_factories.h"
#include "srsran/phy/support/support_factories.h"
#include "srsran/phy/upper/channel_estimation.h"
#include "srsran/phy/upper/channel_modulation/channel_modulation_factories.h"
#include "srsran/phy/upper/equalization/equalization_factories.h"
#include "srsran/phy/upper/sequence_generators/sequence_generator_factories.h"
#include "srsran/phy/upper/signal_processors/signal_processor_factories.h"
#include "lib/phy/generic_functions/dft_processor_generic_impl.h"
#include "lib/scheduler/support/tbs_calculator.h"
#include "srsran/ran/sch_mcs.h"
#include "srsran/ran/ldpc_base_graph.h"
#include <atomic>
#include <chrono>
#include <pthread.h>
#include <sched.h>
#include <thread>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

// Internal reference headers (graph tables are not exported through include/).
#include "lib/phy/upper/channel_coding/ldpc/ldpc_graph_impl.h"
#include "lib/phy/upper/channel_coding/ldpc/ldpc_luts_impl.h"

using namespace srsran;

namespace {

const char* impl_name(int impl)
{
  switch (impl) {
    case 0:
      return "generic";
    case 1:
      return "avx2";
    case 2:
      return "avx512";
    default:
      return "auto";
  }
}

crc_generator_poly to_poly(int p)
{
  switch (p) {
    case 0:
      return crc_generator_poly::CRC24A;
    case 1:
      return crc_generator_poly::CRC24B;
    case 2:
      return crc_generator_poly::CRC24C;
    case 3:
      return crc_generator_poly::CRC16;
    case 4:
      return crc_generator_poly::CRC11;
    default:
      return crc_generator_poly::CRC6;
  }
}

codeblock_metadata make_meta(int bg, int Z, int rv, int mod, unsigned Nref, unsigned nof_filler, unsigned nof_crc_bits)
{
  codeblock_metadata m          = {};
  m.tb_common.base_graph        = (bg == 1) ? ldpc_base_graph_type::BG1 : ldpc_base_graph_type::BG2;
  m.tb_common.lifting_size      = static_cast<ldpc::lifting_size_t>(Z);
  m.tb_common.rv                = rv;
  m.tb_common.mod               = static_cast<modulation_scheme>(mod);
  m.tb_common.Nref              = Nref;
  m.cb_specific.nof_filler_bits = nof_filler;
  m.cb_specific.nof_crc_bits    = nof_crc_bits;
  return m;
}

// A dft_processor_factory (public abstract interface, include/srsran/phy/generic_functions/generic_functions_factories.h)
// that hands out the reference's own generic radix-2 DFT implementation. (The reference's factory translation unit
// unconditionally needs <fftw3.h>, which this image lacks, so that file is not compiled.)
class generic_dft_factory : public dft_processor_factory
{
public:
  std::unique_ptr<dft_processor> create(const dft_processor::configuration& config) override
  {
    auto p = std::make_unique<dft_processor_generic_impl>(config);
    if (!p->is_valid()) {
      return nullptr;
    }
    return p;
  }
};

double now_s()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

extern "C" {

int ref_capi_version()
{
  return 2;
}

// ---------------------------------------------------------------- LDPC graph tables (TS 38.212 Tables 5.3.2-2/-3)
// out[m*68+n] = shift (already reduced mod Z) or 0xffff.
void ref_ldpc_get_graph(int bg, int Z, uint16_t* out)
{
  ldpc::BG_matrix_t g = ldpc::get_graph((bg == 1) ? ldpc_base_graph_type::BG1 : ldpc_base_graph_type::BG2,
                                        static_cast<ldpc::lifting_size_t>(Z));
  for (unsigned m = 0; m != ldpc::MAX_BG_M; ++m) {
    for (unsigned n = 0; n != ldpc::MAX_BG_N_FULL; ++n) {
      out[m * ldpc::MAX_BG_N_FULL + n] = g[m][n];
    }
  }
}

// out[m*20+e] = variable node index or 0xffff.
void ref_ldpc_get_adjacency(int bg, uint16_t* out)
{
  const ldpc::BG_adjacency_matrix_t* a =
      ldpc::get_adjacency_matrix((bg == 1) ? ldpc_base_graph_type::BG1 : ldpc_base_graph_type::BG2);
  for (unsigned m = 0; m != ldpc::MAX_BG_M; ++m) {
    for (unsigned e = 0; e != ldpc::MAX_BG_CHECK_EDGES; ++e) {
      out[m * ldpc::MAX_BG_CHECK_EDGES + e] = (*a)[m][e];
    }
  }
}

int ref_ldpc_lifting_index(int Z)
{
  return ldpc::get_lifting_index(static_cast<ldpc::lifting_size_t>(Z));
}

// ---------------------------------------------------------------- CRC
// bits: one bit per byte. Returns checksum.
uint32_t ref_crc_bits(int poly, const uint8_t* bits, unsigned nbits, int impl_lut)
{
  auto f   = create_crc_calculator_factory_sw(impl_lut ? "lut" : "auto");
  auto crc = f->create(to_poly(poly));
  return crc->calculate_bit(span<const uint8_t>(bits, nbits));
}

// ---------------------------------------------------------------- LDPC encoder
// in: bg_K*Z bytes (one bit per byte, fillers = 254); out: out_len bytes.
int ref_ldpc_encode(int bg, int Z, const uint8_t* in, unsigned in_len, uint8_t* out, unsigned out_len, int impl)
{
  auto f = create_ldpc_encoder_factory_sw(impl_name(impl));
  if (!f) {
    return -1;
  }
  auto enc = f->create();
  if (!enc) {
    return -1;
  }
  codeblock_metadata m = make_meta(bg, Z, 0, 1, 0, 0, 24);
  enc->encode(span<uint8_t>(out, out_len), span<const uint8_t>(in, in_len), m.tb_common);
  return 0;
}

// ---------------------------------------------------------------- LDPC decoder
// llr: in_len int8; out_bits: packed MSB-first, bg_K*Z bits. crc_poly < 0 -> no CRC (nullptr).
// Returns number of iterations if the CRC matched (early stop), 0 if nullopt.
struct ref_ldpc_decoder_handle {
  std::unique_ptr<ldpc_decoder>   dec;
  std::unique_ptr<crc_calculator> crc[6];
};

void* ref_ldpc_decoder_create(int impl)
{
  auto f = create_ldpc_decoder_factory_sw(impl_name(impl));
  if (!f) {
    return nullptr;
  }
  auto h = new ref_ldpc_decoder_handle;
  h->dec = f->create();
  if (!h->dec) {
    delete h;
    return nullptr;
  }
  auto cf = create_crc_calculator_factory_sw("auto");
  for (int p = 0; p != 6; ++p) {
    h->crc[p] = cf->create(to_poly(p));
  }
  return h;
}

void ref_ldpc_decoder_destroy(void* hv)
{
  delete static_cast<ref_ldpc_decoder_handle*>(hv);
}

int ref_ldpc_decode(void*         hv,
                    int           bg,
                    int           Z,
                    const int8_t* llr,
                    unsigned      in_len,
                    unsigned      nof_filler,
                    unsigned      nof_crc_bits,
                    int           crc_poly,
                    unsigned      max_iter,
                    uint8_t*      out_packed)
{
  auto*                       h = static_cast<ref_ldpc_decoder_handle*>(hv);
  unsigned                    K = ((bg == 1) ? 22 : 10) * Z;
  ldpc_decoder::configuration cfg;
  cfg.block_conf                    = make_meta(bg, Z, 0, 1, 0, nof_filler, nof_crc_bits);
  cfg.algorithm_conf.max_iterations = max_iter;
  dynamic_bit_buffer out(K);
  optional<unsigned> r = h->dec->decode(out,
                                        span<const log_likelihood_ratio>(reinterpret_cast<const log_likelihood_ratio*>(llr), in_len),
                                        (crc_poly < 0) ? nullptr : h->crc[crc_poly].get(),
                                        cfg);
  std::memcpy(out_packed, out.get_buffer().data(), (K + 7) / 8);
  return r.has_value() ? static_cast<int>(r.value()) : 0;
}

// Times `reps` decodes of `n_cb` codeblocks (laid out back to back, in_len each); returns seconds.
double ref_ldpc_decode_time(void*         hv,
                            int           bg,
                            int           Z,
                            const int8_t* llr,
                            unsigned      in_len,
                            unsigned      n_cb,
                            unsigned      nof_filler,
                            int           crc_poly,
                            unsigned      max_iter,
                            unsigned      reps,
                            uint8_t*      out_packed)
{
  unsigned Kb = (((bg == 1) ? 22 : 10) * Z + 7) / 8;
  double   t0 = now_s();
  for (unsigned r = 0; r != reps; ++r) {
    for (unsigned i = 0; i != n_cb; ++i) {
      ref_ldpc_decode(hv, bg, Z, llr + size_t(i) * in_len, in_len, nof_filler, 24, crc_poly, max_iter, out_packed + size_t(i) * Kb);
    }
  }
  return now_s() - t0;
}

// ---------------------------------------------------------------- LDPC rate matcher / dematcher
int ref_ldpc_rate_match(int            bg,
                        int            Z,
                        int            rv,
                        int            mod,
                        unsigned       Nref,
                        unsigned       nof_filler,
                        const uint8_t* in,
                        unsigned       in_len,
                        uint8_t*       out,
                        unsigned       out_len)
{
  auto               rm = create_ldpc_rate_matcher_factory_sw()->create();
  codeblock_metadata m  = make_meta(bg, Z, rv, mod, Nref, nof_filler, 24);
  rm->rate_match(span<uint8_t>(out, out_len), span<const uint8_t>(in, in_len), m);
  return 0;
}

// out is in/out (soft buffer), length full_length.
int ref_ldpc_rate_dematch(int           bg,
                          int           Z,
                          int           rv,
                          int           mod,
                          unsigned      Nref,
                          unsigned      nof_filler,
                          int           new_data,
                          const int8_t* in,
                          unsigned      in_len,
                          int8_t*       out,
                          unsigned      out_len,
                          int           impl)
{
  auto f = create_ldpc_rate_dematcher_factory_sw(impl_name(impl));
  if (!f) {
    return -1;
  }
  auto rdm = f->create();
  if (!rdm) {
    return -1;
  }
  codeblock_metadata m = make_meta(bg, Z, rv, mod, Nref, nof_filler, 24);
  rdm->rate_dematch(span<log_likelihood_ratio>(reinterpret_cast<log_likelihood_ratio*>(out), out_len),
                    span<const log_likelihood_ratio>(reinterpret_cast<const log_likelihood_ratio*>(in), in_len),
                    new_data != 0,
                    m);
  return 0;
}

// ---------------------------------------------------------------- PDSCH encoder (segment + encode + rate match)
// tb: packed bytes; codeword: one bit per byte, nof_ch_symbols*Qm.
int ref_pdsch_encode(int            bg,
                     int            rv,
                     int            mod,
                     unsigned       Nref,
                     unsigned       nof_layers,
                     unsigned       nof_ch_symbols,
                     const uint8_t* tb,
                     unsigned       tb_bytes,
                     uint8_t*       codeword,
                     unsigned       cw_len,
                     int            impl)
{
  auto                                  crcf = create_crc_calculator_factory_sw("auto");
  pdsch_encoder_factory_sw_configuration c;
  c.encoder_factory      = create_ldpc_encoder_factory_sw(impl_name(impl));
  c.rate_matcher_factory = create_ldpc_rate_matcher_factory_sw();
  c.segmenter_factory    = create_ldpc_segmenter_tx_factory_sw(crcf);
  auto             enc   = create_pdsch_encoder_factory_sw(c)->create();
  segmenter_config cfg;
  cfg.base_graph     = (bg == 1) ? ldpc_base_graph_type::BG1 : ldpc_base_graph_type::BG2;
  cfg.rv             = rv;
  cfg.mod            = static_cast<modulation_scheme>(mod);
  cfg.Nref           = Nref;
  cfg.nof_layers     = nof_layers;
  cfg.nof_ch_symbols = nof_ch_symbols;
  enc->encode(span<uint8_t>(codeword, cw_len), span<const uint8_t>(tb, tb_bytes), cfg);
  return 0;
}

// ---------------------------------------------------------------- PUSCH decoder (segment + dematch + decode + CRC)
struct ref_pusch_decoder_handle {
  std::unique_ptr<pusch_decoder>      dec;
  std::unique_ptr<rx_softbuffer_pool> pool;
};

void* ref_pusch_decoder_create(int impl)
{
  auto                                   crcf = create_crc_calculator_factory_sw("auto");
  pusch_decoder_factory_sw_configuration c;
  c.crc_factory       = crcf;
  c.decoder_factory   = create_ldpc_decoder_factory_sw(impl_name(impl));
  c.dematcher_factory = create_ldpc_rate_dematcher_factory_sw(impl_name(impl));
  c.segmenter_factory = create_ldpc_segmenter_rx_factory_sw();
  if (!c.decoder_factory || !c.dematcher_factory) {
    return nullptr;
  }
  auto h = new ref_pusch_decoder_handle;
  h->dec = create_pusch_decoder_factory_sw(c)->create();
  rx_softbuffer_pool_config pc;
  pc.max_codeblock_size   = ldpc::MAX_CODEBLOCK_SIZE;
  pc.max_softbuffers      = 4;
  pc.max_nof_codeblocks   = 4 * 64;
  pc.expire_timeout_slots = 100000;
  h->pool                 = create_rx_softbuffer_pool(pc);
  return h;
}

void ref_pusch_decoder_destroy(void* hv)
{
  delete static_cast<ref_pusch_decoder_handle*>(hv);
}

// Decodes a sequence of `nof_tx` (re)transmissions of one TB (same HARQ process). llrs holds nof_tx codewords of
// cw_len each, rvs[nof_tx]. Outputs, per transmission: tb_crc_ok, nof_codeblocks, min/max iterations, and the TB bytes.
int ref_pusch_decode(void*          hv,
                     int            bg,
                     int            mod,
                     unsigned       Nref,
                     unsigned       nof_layers,
                     unsigned       nof_ch_symbols,
                     unsigned       tb_bytes,
                     unsigned       nof_tx,
                     const int*     rvs,
                     const int8_t*  llrs,
                     unsigned       cw_len,
                     unsigned       max_iter,
                     int            early_stop,
                     uint8_t*       tb_out,      // nof_tx * tb_bytes
                     int*           tb_crc_ok,   // nof_tx
                     int*           iters_minmax // nof_tx * 2
)
{
  auto*                    h = static_cast<ref_pusch_decoder_handle*>(hv);
  rx_softbuffer_identifier id;
  id.rnti          = 0x1234;
  id.harq_ack_id   = 3;
  unsigned nof_cbs = ldpc::compute_nof_codeblocks(units::bits(tb_bytes * 8),
                                                  (bg == 1) ? ldpc_base_graph_type::BG1 : ldpc_base_graph_type::BG2);
  slot_point slot(1, 0);
  for (unsigned t = 0; t != nof_tx; ++t) {
    unique_rx_softbuffer sb = h->pool->reserve_softbuffer(slot, id, nof_cbs);
    if (!sb.is_valid()) {
      return -1;
    }
    pusch_decoder::configuration cfg;
    cfg.segmenter_cfg.base_graph     = (bg == 1) ? ldpc_base_graph_type::BG1 : ldpc_base_graph_type::BG2;
    cfg.segmenter_cfg.rv             = rvs[t];
    cfg.segmenter_cfg.mod            = static_cast<modulation_scheme>(mod);
    cfg.segmenter_cfg.Nref           = Nref;
    cfg.segmenter_cfg.nof_layers     = nof_layers;
    cfg.segmenter_cfg.nof_ch_symbols = nof_ch_symbols;
    cfg.nof_ldpc_iterations          = max_iter;
    cfg.use_early_stop               = early_stop != 0;
    cfg.new_data                     = (t == 0);
    pusch_decoder_result res;
    std::vector<uint8_t> tb(tb_bytes, 0);
    h->dec->decode(tb,
                   res,
                   &sb.get(),
                   span<const log_likelihood_ratio>(reinterpret_cast<const log_likelihood_ratio*>(llrs) + size_t(t) * cw_len, cw_len),
                   cfg);
    std::memcpy(tb_out + size_t(t) * tb_bytes, tb.data(), tb_bytes);
    tb_crc_ok[t]            = res.tb_crc_ok ? 1 : 0;
    iters_minmax[2 * t]     = res.ldpc_decoder_stats.get_nof_observations() ? (int)res.ldpc_decoder_stats.get_min() : 0;
    iters_minmax[2 * t + 1] = res.ldpc_decoder_stats.get_nof_observations() ? (int)res.ldpc_decoder_stats.get_max() : 0;
  }
  // Release the buffer for the next call.
  h->pool->run_slot(slot + 200000);
  return (int)nof_cbs;
}

// ---------------------------------------------------------------- DFT (generic radix-2 implementation of the reference)
int ref_dft(unsigned size, int inverse, const float* in, float* out)
{
  generic_dft_factory          f;
  dft_processor::configuration c;
  c.size = size;
  c.dir  = inverse ? dft_processor::direction::INVERSE : dft_processor::direction::DIRECT;
  auto d = f.create(c);
  if (!d) {
    return -1;
  }
  std::memcpy(d->get_input().data(), in, sizeof(cf_t) * size);
  span<const cf_t> o = d->run();
  std::memcpy(out, o.data(), sizeof(cf_t) * size);
  return 0;
}

// ---------------------------------------------------------------- OFDM slot demodulator / modulator
// in: slot samples (get_slot_size) ; grid_out: [14][bw_rb*12] cf_t for one port.
int ref_ofdm_demod_slot(unsigned     numerology,
                        unsigned     bw_rb,
                        unsigned     dft_size,
                        unsigned     window_offset,
                        float        scale,
                        double       center_freq_hz,
                        unsigned     slot_index,
                        const float* in,
                        unsigned     nof_in_samples,
                        float*       grid_out)
{
  ofdm_factory_generic_configuration fc;
  fc.dft_factory = std::make_shared<generic_dft_factory>();
  auto                           f = create_ofdm_demodulator_factory_generic(fc);
  ofdm_demodulator_configuration c;
  c.numerology                = numerology;
  c.bw_rb                     = bw_rb;
  c.dft_size                  = dft_size;
  c.cp                        = cyclic_prefix::NORMAL;
  c.nof_samples_window_offset = window_offset;
  c.scale                     = scale;
  c.center_freq_hz            = center_freq_hz;
  auto d                      = f->create_ofdm_slot_demodulator(c);
  if (d->get_slot_size(slot_index) != nof_in_samples) {
    return -(int)d->get_slot_size(slot_index);
  }
  auto grid = create_resource_grid(1, 14, bw_rb * 12);
  d->demodulate(*grid, span<const cf_t>(reinterpret_cast<const cf_t*>(in), nof_in_samples), 0, slot_index);
  for (unsigned l = 0; l != 14; ++l) {
    grid->get(span<cf_t>(reinterpret_cast<cf_t*>(grid_out) + size_t(l) * bw_rb * 12, bw_rb * 12), 0, l, 0);
  }
  return 0;
}

int ref_ofdm_mod_slot(unsigned     numerology,
                      unsigned     bw_rb,
                      unsigned     dft_size,
                      float        scale,
                      double       center_freq_hz,
                      unsigned     slot_index,
                      const float* grid_in,
                      float*       out,
                      unsigned     nof_out_samples)
{
  ofdm_factory_generic_configuration fc;
  fc.dft_factory = std::make_shared<generic_dft_factory>();
  auto                         f = create_ofdm_modulator_factory_generic(fc);
  ofdm_modulator_configuration c;
  c.numerology     = numerology;
  c.bw_rb          = bw_rb;
  c.dft_size       = dft_size;
  c.cp             = cyclic_prefix::NORMAL;
  c.scale          = scale;
  c.center_freq_hz = center_freq_hz;
  auto m           = f->create_ofdm_slot_modulator(c);
  if (m->get_slot_size(slot_index) != nof_out_samples) {
    return -(int)m->get_slot_size(slot_index);
  }
  auto grid = create_resource_grid(1, 14, bw_rb * 12);
  for (unsigned l = 0; l != 14; ++l) {
    grid->put(0, l, 0, span<const cf_t>(reinterpret_cast<const cf_t*>(grid_in) + size_t(l) * bw_rb * 12, bw_rb * 12));
  }
  m->modulate(span<cf_t>(reinterpret_cast<cf_t*>(out), nof_out_samples), *grid, 0, slot_index);
  return 0;
}

// Times `reps` slot demodulations (single thread); returns seconds.
double ref_ofdm_demod_time(unsigned numerology, unsigned bw_rb, unsigned dft_size, unsigned window_offset, double center_freq_hz,
                           const float* in, unsigned nof_in_samples, unsigned reps)
{
  ofdm_factory_generic_configuration fc;
  fc.dft_factory = std::make_shared<generic_dft_factory>();
  auto                           f = create_ofdm_demodulator_factory_generic(fc);
  ofdm_demodulator_configuration c;
  c.numerology                = numerology;
  c.bw_rb                     = bw_rb;
  c.dft_size                  = dft_size;
  c.cp                        = cyclic_prefix::NORMAL;
  c.nof_samples_window_offset = window_offset;
  c.scale                     = 1.0F;
  c.center_freq_hz            = center_freq_hz;
  auto   d                    = f->create_ofdm_slot_demodulator(c);
  auto   grid                 = create_resource_grid(1, 14, bw_rb * 12);
  double t0                   = now_s();
  for (unsigned r = 0; r != reps; ++r) {
    d->demodulate(*grid, span<const cf_t>(reinterpret_cast<const cf_t*>(in), nof_in_samples), 0, 0);
  }
  return now_s() - t0;
}

// ---------------------------------------------------------------- DM-RS PUSCH channel estimator
// grid_in: [nof_rx_ports][14][nof_prb_grid*12] cf_t. rb_mask: nof_prb_grid bytes (0/1). symbols_mask: 14 bytes.
// ce_out: [nof_layers][nof_rx_ports][first+nof][nof_prb_grid*12] cf_t (path-major like channel_estimate).
// scalars_out: per (port, layer): rsrp, epre, noise_var, snr, time_alignment_s  (5 floats).
int ref_dmrs_pusch_estimate(unsigned       numerology,
                            unsigned       slot_index,
                            int            dmrs_type2,
                            unsigned       scrambling_id,
                            int            n_scid,
                            float          scaling,
                            const uint8_t* symbols_mask,
                            const uint8_t* rb_mask,
                            unsigned       nof_prb_grid,
                            unsigned       first_symbol,
                            unsigned       nof_symbols,
                            unsigned       nof_tx_layers,
                            unsigned       nof_rx_ports,
                            const float*   grid_in,
                            float*         ce_out,
                            float*         scalars_out)
{
  auto prg_f  = create_pseudo_random_generator_sw_factory();
  auto port_f = create_port_channel_estimator_factory_sw(std::make_shared<generic_dft_factory>());
  auto est    = create_dmrs_pusch_estimator_factory_sw(prg_f, port_f)->create();
  auto grid   = create_resource_grid(nof_rx_ports, 14, nof_prb_grid * 12);
  for (unsigned p = 0; p != nof_rx_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      grid->put(p, l, 0, span<const cf_t>(reinterpret_cast<const cf_t*>(grid_in) + (size_t(p) * 14 + l) * nof_prb_grid * 12, nof_prb_grid * 12));
    }
  }
  dmrs_pusch_estimator::configuration cfg;
  cfg.slot          = slot_point(numerology, slot_index);
  cfg.type          = dmrs_type2 ? dmrs_type::TYPE2 : dmrs_type::TYPE1;
  cfg.scrambling_id = scrambling_id;
  cfg.n_scid        = n_scid != 0;
  cfg.scaling       = scaling;
  cfg.c_prefix      = cyclic_prefix::NORMAL;
  cfg.symbols_mask  = bounded_bitset<MAX_NSYMB_PER_SLOT>(14);
  for (unsigned l = 0; l != 14; ++l) {
    if (symbols_mask[l]) {
      cfg.symbols_mask.set(l);
    }
  }
  cfg.rb_mask = bounded_bitset<MAX_RB>(nof_prb_grid);
  for (unsigned r = 0; r != nof_prb_grid; ++r) {
    if (rb_mask[r]) {
      cfg.rb_mask.set(r);
    }
  }
  cfg.first_symbol  = first_symbol;
  cfg.nof_symbols   = nof_symbols;
  cfg.nof_tx_layers = nof_tx_layers;
  for (unsigned p = 0; p != nof_rx_ports; ++p) {
    cfg.rx_ports.push_back(p);
  }
  channel_estimate ce;
  est->estimate(ce, *grid, cfg);
  unsigned nsymb = first_symbol + nof_symbols;
  unsigned nsc   = nof_prb_grid * 12;
  for (unsigned ly = 0; ly != nof_tx_layers; ++ly) {
    for (unsigned p = 0; p != nof_rx_ports; ++p) {
      for (unsigned l = 0; l != nsymb; ++l) {
        span<const cf_t> v = static_cast<const channel_estimate&>(ce).get_symbol_ch_estimate(l, p, ly);
        std::memcpy(ce_out + 2 * (((size_t(ly) * nof_rx_ports + p) * nsymb + l) * nsc), v.data(), sizeof(cf_t) * nsc);
      }
      float* sc = scalars_out + 5 * (size_t(p) * nof_tx_layers + ly);
      sc[0]     = ce.get_rsrp(p, ly);
      sc[1]     = ce.get_epre(p, ly);
      sc[2]     = ce.get_noise_variance(p, ly);
      sc[3]     = ce.get_snr(p, ly);
      sc[4]     = (float)ce.get_time_alignment(p, ly).to_seconds();
    }
  }
  return 0;
}

// ---------------------------------------------------------------- PUSCH demodulator
static modulation_scheme mod_from_bits(int mod)
{
  switch (mod) {
    case 1:
      return modulation_scheme::PI_2_BPSK;
    case 2:
      return modulation_scheme::QPSK;
    case 4:
      return modulation_scheme::QAM16;
    case 6:
      return modulation_scheme::QAM64;
    default:
      return modulation_scheme::QAM256;
  }
}

// Soft demapper alone (demodulation_mapper_impl.cpp:83-106).
int ref_demodulate_soft(int mod, unsigned nsym, const float* symbols, const float* noise_vars, int8_t* llr)
{
  auto demod = create_channel_modulation_sw_factory()->create_demodulation_mapper();
  std::vector<log_likelihood_ratio> out(size_t(nsym) * mod);
  demod->demodulate_soft(out,
                         span<const cf_t>(reinterpret_cast<const cf_t*>(symbols), nsym),
                         span<const float>(noise_vars, nsym),
                         mod_from_bits(mod));
  for (size_t i = 0; i != out.size(); ++i) {
    llr[i] = out[i].to_value_type();
  }
  return 0;
}

// Modulation mapper (TX side, modulation_mapper_impl.cpp) - only used to build stimuli. bits: one bit per byte.
int ref_modulate(int mod, unsigned nsym, const uint8_t* bits, float* symbols)
{
  auto               mapper = create_channel_modulation_sw_factory()->create_modulation_mapper();
  dynamic_bit_buffer packed(nsym * mod);
  for (unsigned i = 0; i != nsym * (unsigned)mod; ++i) {
    packed.insert(bits[i] & 1u, i, 1);
  }
  mapper->modulate(span<cf_t>(reinterpret_cast<cf_t*>(symbols), nsym), packed, mod_from_bits(mod));
  return 0;
}

// pusch_demodulator_impl::demodulate, one transmit layer. grid_in: [nof_rx_ports][14][nsc]; ce_in: [nof_rx_ports][ce_nof_symbols][nsc].
int ref_pusch_demodulate(unsigned       rnti,
                         unsigned       n_id,
                         int            mod,
                         unsigned       start_symbol,
                         unsigned       nof_symbols,
                         const uint8_t* dmrs_symbols_mask,
                         int            dmrs_type2,
                         unsigned       nof_cdm_groups_without_data,
                         const uint8_t* rb_mask,
                         unsigned       nof_prb_grid,
                         unsigned       nof_rx_ports,
                         const float*   grid_in,
                         const float*   ce_in,
                         unsigned       ce_nof_symbols,
                         float          noise_var,
                         int8_t*        llr_out,
                         unsigned       nof_llr)
{
  auto dem = create_pusch_demodulator_factory_sw(
                 create_channel_equalizer_factory_zf(), create_channel_modulation_sw_factory(), create_pseudo_random_generator_sw_factory())
                 ->create();
  unsigned nsc  = nof_prb_grid * 12;
  auto     grid = create_resource_grid(nof_rx_ports, 14, nsc);
  for (unsigned p = 0; p != nof_rx_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      grid->put(p, l, 0, span<const cf_t>(reinterpret_cast<const cf_t*>(grid_in) + (size_t(p) * 14 + l) * nsc, nsc));
    }
  }
  channel_estimate::channel_estimate_dimensions dims;
  dims.nof_prb       = nof_prb_grid;
  dims.nof_symbols   = ce_nof_symbols;
  dims.nof_rx_ports  = nof_rx_ports;
  dims.nof_tx_layers = 1;
  channel_estimate ce(dims);
  for (unsigned p = 0; p != nof_rx_ports; ++p) {
    for (unsigned l = 0; l != ce_nof_symbols; ++l) {
      span<cf_t> v = ce.get_symbol_ch_estimate(l, p, 0);
      std::memcpy(v.data(), ce_in + 2 * ((size_t(p) * ce_nof_symbols + l) * nsc), sizeof(cf_t) * nsc);
    }
    ce.set_noise_variance(noise_var, p, 0);
  }
  pusch_demodulator::configuration cfg;
  cfg.rnti    = rnti;
  cfg.rb_mask = bounded_bitset<MAX_RB>(nof_prb_grid);
  for (unsigned r = 0; r != nof_prb_grid; ++r) {
    if (rb_mask[r]) {
      cfg.rb_mask.set(r);
    }
  }
  cfg.modulation         = mod_from_bits(mod);
  cfg.start_symbol_index = start_symbol;
  cfg.nof_symbols        = nof_symbols;
  for (unsigned l = 0; l != 14; ++l) {
    cfg.dmrs_symb_pos[l] = dmrs_symbols_mask[l] != 0;
  }
  cfg.dmrs_config_type            = dmrs_type2 ? dmrs_type::TYPE2 : dmrs_type::TYPE1;
  cfg.nof_cdm_groups_without_data = nof_cdm_groups_without_data;
  cfg.n_id                        = n_id;
  cfg.nof_tx_layers               = 1;
  for (unsigned p = 0; p != nof_rx_ports; ++p) {
    cfg.rx_ports.push_back(p);
  }
  std::vector<log_likelihood_ratio> out(nof_llr);
  dem->demodulate(out, *grid, ce, cfg);
  for (unsigned i = 0; i != nof_llr; ++i) {
    llr_out[i] = out[i].to_value_type();
  }
  return 0;
}

// The same with repetition placeholders in the descrambler and the EVM of the demodulation status.
int ref_pusch_demodulate_ex(unsigned       rnti,
                         unsigned       n_id,
                         int            mod,
                         unsigned       start_symbol,
                         unsigned       nof_symbols,
                         const uint8_t* dmrs_symbols_mask,
                         int            dmrs_type2,
                         unsigned       nof_cdm_groups_without_data,
                         const uint8_t* rb_mask,
                         unsigned       nof_prb_grid,
                         unsigned       nof_rx_ports,
                         const float*   grid_in,
                         const float*   ce_in,
                         unsigned       ce_nof_symbols,
                         float          noise_var,
                         int8_t*        llr_out,
                         unsigned       nof_llr,
                         const uint16_t* placeholders,
                         unsigned        nof_placeholders,
                         float*          evm_out)
{
  auto dem = create_pusch_demodulator_factory_sw(
                 create_channel_equalizer_factory_zf(), create_channel_modulation_sw_factory(), create_pseudo_random_generator_sw_factory(), evm_out != nullptr)
                 ->create();
  unsigned nsc  = nof_prb_grid * 12;
  auto     grid = create_resource_grid(nof_rx_ports, 14, nsc);
  for (unsigned p = 0; p != nof_rx_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      grid->put(p, l, 0, span<const cf_t>(reinterpret_cast<const cf_t*>(grid_in) + (size_t(p) * 14 + l) * nsc, nsc));
    }
  }
  channel_estimate::channel_estimate_dimensions dims;
  dims.nof_prb       = nof_prb_grid;
  dims.nof_symbols   = ce_nof_symbols;
  dims.nof_rx_ports  = nof_rx_ports;
  dims.nof_tx_layers = 1;
  channel_estimate ce(dims);
  for (unsigned p = 0; p != nof_rx_ports; ++p) {
    for (unsigned l = 0; l != ce_nof_symbols; ++l) {
      span<cf_t> v = ce.get_symbol_ch_estimate(l, p, 0);
      std::memcpy(v.data(), ce_in + 2 * ((size_t(p) * ce_nof_symbols + l) * nsc), sizeof(cf_t) * nsc);
    }
    ce.set_noise_variance(noise_var, p, 0);
  }
  pusch_demodulator::configuration cfg;
  cfg.rnti    = rnti;
  cfg.rb_mask = bounded_bitset<MAX_RB>(nof_prb_grid);
  for (unsigned r = 0; r != nof_prb_grid; ++r) {
    if (rb_mask[r]) {
      cfg.rb_mask.set(r);
    }
  }
  cfg.modulation         = mod_from_bits(mod);
  cfg.start_symbol_index = start_symbol;
  cfg.nof_symbols        = nof_symbols;
  for (unsigned l = 0; l != 14; ++l) {
    cfg.dmrs_symb_pos[l] = dmrs_symbols_mask[l] != 0;
  }
  cfg.dmrs_config_type            = dmrs_type2 ? dmrs_type::TYPE2 : dmrs_type::TYPE1;
  cfg.nof_cdm_groups_without_data = nof_cdm_groups_without_data;
  cfg.n_id                        = n_id;
  cfg.nof_tx_layers               = 1;
  for (unsigned p = 0; p != nof_rx_ports; ++p) {
    cfg.rx_ports.push_back(p);
  }
  for (unsigned i = 0; i != nof_placeholders; ++i) {
    cfg.placeholders.push_back(placeholders[i]);
  }
  std::vector<log_likelihood_ratio> out(nof_llr);
  pusch_demodulator::demodulation_status st = dem->demodulate(out, *grid, ce, cfg);
  if (evm_out != nullptr) {
    *evm_out = st.evm.has_value() ? st.evm.value() : -1.0F;
  }
  for (unsigned i = 0; i != nof_llr; ++i) {
    llr_out[i] = out[i].to_value_type();
  }
  return 0;
}

// ---------------------------------------------------------------- PDSCH modulator + DM-RS PDSCH
// vrb_mask: [bwp_size] bytes (type-0 allocation relative to the BWP); interleaved: 0 = none, 1 = create_interleaved_other(L_i = 2).
// grid_out: [nof_grid_ports][14][nof_prb_grid*12] (zero where nothing was mapped); prb_list_out: the PRB indices in mapping order.
int ref_pdsch_modulate(unsigned        rnti,
                       unsigned        n_id,
                       float           scaling,
                       unsigned        nof_layers,
                       const int*      mod,
                       const uint8_t*  cw0,
                       unsigned        nbits0,
                       const uint8_t*  cw1,
                       unsigned        nbits1,
                       unsigned        start_symbol,
                       unsigned        nof_symbols,
                       const uint8_t*  dmrs_symbols_mask,
                       int             dmrs_type2,
                       unsigned        nof_cdm_groups_without_data,
                       unsigned        bwp_start_rb,
                       unsigned        bwp_size_rb,
                       const uint8_t*  vrb_mask,
                       int             interleaved,
                       unsigned        nof_reserved,
                       const uint8_t*  res_prb_mask,
                       const uint16_t* res_re_mask,
                       const uint16_t* res_symbols,
                       const uint8_t*  ports,
                       unsigned        nof_prb_grid,
                       unsigned        nof_grid_ports,
                       float*          grid_out,
                       uint16_t*       prb_list_out,
                       unsigned*       nof_prb_out)
{
  auto     m    = create_pdsch_modulator_factory_sw(create_channel_modulation_sw_factory(), create_pseudo_random_generator_sw_factory())->create();
  unsigned nsc  = nof_prb_grid * 12;
  auto     grid = create_resource_grid(nof_grid_ports, 14, nsc);
  grid->set_all_zero();
  pdsch_modulator::config_t cfg;
  cfg.rnti = rnti, cfg.bwp_size_rb = bwp_size_rb, cfg.bwp_start_rb = bwp_start_rb;
  cfg.modulation1 = mod_from_bits(mod[0]), cfg.modulation2 = mod_from_bits(mod[1]);
  bounded_bitset<MAX_RB> vrb(bwp_size_rb);
  for (unsigned r = 0; r != bwp_size_rb; ++r) {
    if (vrb_mask[r]) {
      vrb.set(r);
    }
  }
  optional<vrb_to_prb_mapper> mapper;
  if (interleaved) {
    mapper.emplace(vrb_to_prb_mapper::create_interleaved_other(bwp_start_rb, bwp_size_rb, 2));
  }
  cfg.freq_allocation    = rb_allocation::make_type0(vrb, mapper);
  cfg.start_symbol_index = start_symbol, cfg.nof_symbols = nof_symbols;
  cfg.dmrs_symb_pos      = symbol_slot_mask(14);
  for (unsigned l = 0; l != 14; ++l) {
    if (dmrs_symbols_mask[l]) {
      cfg.dmrs_symb_pos.set(l);
    }
  }
  cfg.dmrs_config_type            = dmrs_type2 ? dmrs_type::TYPE2 : dmrs_type::TYPE1;
  cfg.nof_cdm_groups_without_data = nof_cdm_groups_without_data;
  cfg.n_id = n_id, cfg.scaling = scaling, cfg.pmi = 0;
  for (unsigned r = 0; r != nof_reserved; ++r) {
    re_pattern pat;
    pat.prb_mask = bounded_bitset<MAX_RB>(nof_prb_grid);
    for (unsigned b = 0; b != nof_prb_grid; ++b) {
      if (res_prb_mask[size_t(r) * nof_prb_grid + b]) {
        pat.prb_mask.set(b);
      }
    }
    for (unsigned k = 0; k != 12; ++k) {
      pat.re_mask.set(k, (res_re_mask[r] >> k) & 1U);
    }
    pat.symbols = symbol_slot_mask(14);
    for (unsigned l = 0; l != 14; ++l) {
      pat.symbols.set(l, (res_symbols[r] >> l) & 1U);
    }
    cfg.reserved.merge(pat);
  }
  for (unsigned l = 0; l != nof_layers; ++l) {
    cfg.ports.push_back(ports[l]);
  }
  std::vector<dynamic_bit_buffer> packed;
  const uint8_t*                  cws[2] = {cw0, cw1};
  unsigned                        nb[2]  = {nbits0, nbits1};
  unsigned                        ncw    = (nof_layers >= 4) ? 2 : 1;
  for (unsigned q = 0; q != ncw; ++q) {
    packed.emplace_back(nb[q]);
    for (unsigned i = 0; i != nb[q]; ++i) {
      packed.back().insert(cws[q][i] & 1U, i, 1);
    }
  }
  std::vector<bit_buffer> views;
  for (auto& b : packed) {
    views.emplace_back(b);
  }
  m->modulate(*grid, views, cfg);
  for (unsigned p = 0; p != nof_grid_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      grid->get(span<cf_t>(reinterpret_cast<cf_t*>(grid_out) + (size_t(p) * 14 + l) * nsc, nsc), p, l, 0);
    }
  }
  // The contiguous mapping path (the only one that works in 23.5) walks rb_allocation::get_prb_mask in ascending order.
  auto     pm = cfg.freq_allocation.get_prb_mask(bwp_start_rb, bwp_size_rb);
  unsigned n  = 0;
  pm.for_each(0, pm.size(), [&](unsigned r) { prb_list_out[n++] = r; });
  *nof_prb_out = n;
  return 0;
}

// Allocated PRBs (rb_allocation::get_prb_mask, ascending) of a type-0 allocation inside a bandwidth part.
int ref_prb_indices(unsigned bwp_start_rb, unsigned bwp_size_rb, const uint8_t* vrb_mask, int interleaved, uint16_t* out, unsigned* n_out)
{
  bounded_bitset<MAX_RB> vrb(bwp_size_rb);
  for (unsigned r = 0; r != bwp_size_rb; ++r) {
    if (vrb_mask[r]) {
      vrb.set(r);
    }
  }
  optional<vrb_to_prb_mapper> mapper;
  if (interleaved) {
    mapper.emplace(vrb_to_prb_mapper::create_interleaved_other(bwp_start_rb, bwp_size_rb, 2));
  }
  auto     pm = rb_allocation::make_type0(vrb, mapper).get_prb_mask(bwp_start_rb, bwp_size_rb);
  unsigned n  = 0;
  pm.for_each(0, pm.size(), [&](unsigned r) { out[n++] = r; });
  *n_out = n;
  return 0;
}

int ref_dmrs_pdsch_map(unsigned       numerology,
                       unsigned       slot_index,
                       unsigned       reference_point_k_rb,
                       int            type2,
                       unsigned       scrambling_id,
                       int            n_scid,
                       float          amplitude,
                       const uint8_t* symbols_mask,
                       const uint8_t* rb_mask,
                       unsigned       nof_prb_grid,
                       unsigned       nof_ports,
                       const uint8_t* ports,
                       unsigned       nof_grid_ports,
                       float*         grid_out)
{
  auto     d    = create_dmrs_pdsch_processor_factory_sw(create_pseudo_random_generator_sw_factory())->create();
  unsigned nsc  = nof_prb_grid * 12;
  auto     grid = create_resource_grid(nof_grid_ports, 14, nsc);
  grid->set_all_zero();
  dmrs_pdsch_processor::config_t cfg;
  cfg.slot = slot_point(numerology, slot_index), cfg.reference_point_k_rb = reference_point_k_rb;
  cfg.type = type2 ? dmrs_type::TYPE2 : dmrs_type::TYPE1, cfg.scrambling_id = scrambling_id, cfg.n_scid = n_scid != 0, cfg.amplitude = amplitude;
  cfg.symbols_mask = symbol_slot_mask(14);
  for (unsigned l = 0; l != 14; ++l) {
    if (symbols_mask[l]) {
      cfg.symbols_mask.set(l);
    }
  }
  cfg.rb_mask = bounded_bitset<MAX_RB>(nof_prb_grid);
  for (unsigned r = 0; r != nof_prb_grid; ++r) {
    if (rb_mask[r]) {
      cfg.rb_mask.set(r);
    }
  }
  for (unsigned p = 0; p != nof_ports; ++p) {
    cfg.ports.push_back(ports[p]);
  }
  d->map(*grid, cfg);
  for (unsigned p = 0; p != nof_grid_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      grid->get(span<cf_t>(reinterpret_cast<cf_t*>(grid_out) + (size_t(p) * 14 + l) * nsc, nsc), p, l, 0);
    }
  }
  return 0;
}

// ---------------------------------------------------------------- polar
// Code construction: fills N, nPC, K_set (N bytes), PC_set (up to 3), F_set (N bytes), blk_interleaver (N uint16).
int ref_polar_code(unsigned K, unsigned E, unsigned nMax, int ibil, unsigned* n_out, unsigned* npc_out, uint8_t* k_set, uint16_t* pc_set,
                   uint16_t* blk_interleaver)
{
  auto f    = create_polar_factory_sw();
  auto code = f->create_code();
  code->set(K, E, nMax, ibil ? polar_code_ibil::present : polar_code_ibil::not_present);
  *n_out   = code->get_n();
  *npc_out = code->get_nPC();
  unsigned N = code->get_N();
  for (unsigned i = 0; i != N; ++i) {
    k_set[i] = code->get_K_set().test(i) ? 1 : 0;
  }
  span<const uint16_t> pc = code->get_PC_set();
  for (unsigned i = 0; i != pc.size(); ++i) {
    pc_set[i] = pc[i];
  }
  span<const uint16_t> bi = code->get_blk_interleaver();
  for (unsigned i = 0; i != N; ++i) {
    blk_interleaver[i] = bi[i];
  }
  return (int)pc.size();
}

// Mother-code reliability sequence for size 2^n (polar_code_impl.cpp get_mother_code).
// Tx chain of polar_chain_test.cpp:176-190: allocate -> encode -> rate match. msg: K bytes (1 bit/byte) -> out: E bytes.
int ref_polar_encode_chain(unsigned K, unsigned E, unsigned nMax, int ibil, const uint8_t* msg, uint8_t* out, uint8_t* allocated_out,
                           uint8_t* encoded_out)
{
  auto f    = create_polar_factory_sw();
  auto code = f->create_code();
  code->set(K, E, nMax, ibil ? polar_code_ibil::present : polar_code_ibil::not_present);
  unsigned             N = code->get_N();
  std::vector<uint8_t> alloc(N), enc(N);
  f->create_allocator()->allocate(alloc, span<const uint8_t>(msg, K), *code);
  f->create_encoder()->encode(enc, alloc, code->get_n());
  f->create_rate_matcher()->rate_match(span<uint8_t>(out, E), enc, *code);
  if (allocated_out) {
    std::memcpy(allocated_out, alloc.data(), N);
  }
  if (encoded_out) {
    std::memcpy(encoded_out, enc.data(), N);
  }
  return (int)N;
}

// Rx chain of polar_chain_test.cpp:197-206: rate dematch -> SSC decode -> deallocate. llr: E int8 -> msg: K bytes.
int ref_polar_decode_chain(unsigned K, unsigned E, unsigned nMax, int ibil, const int8_t* llr, uint8_t* msg, int8_t* dematched_out,
                           uint8_t* decoded_u_out)
{
  auto f    = create_polar_factory_sw();
  auto code = f->create_code();
  code->set(K, E, nMax, ibil ? polar_code_ibil::present : polar_code_ibil::not_present);
  unsigned                          N = code->get_N();
  std::vector<log_likelihood_ratio> dem(N);
  std::vector<uint8_t>              u(N);
  f->create_rate_dematcher()->rate_dematch(dem, span<const log_likelihood_ratio>(reinterpret_cast<const log_likelihood_ratio*>(llr), E), *code);
  f->create_decoder(nMax)->decode(u, dem, *code);
  f->create_deallocator()->deallocate(span<uint8_t>(msg, K), u, *code);
  if (dematched_out) {
    std::memcpy(dematched_out, dem.data(), N);
  }
  if (decoded_u_out) {
    std::memcpy(decoded_u_out, u.data(), N);
  }
  return (int)N;
}

// CRC interleaver (polar_interleaver, TS 38.212 5.3.1.1). dir: 0 = tx, 1 = rx.
int ref_polar_interleave(const uint8_t* in, uint8_t* out, unsigned K, int dir)
{
  auto f = create_polar_factory_sw();
  f->create_interleaver()->interleave(span<uint8_t>(out, K), span<const uint8_t>(in, K), dir ? polar_interleaver_direction::rx : polar_interleaver_direction::tx);
  return 0;
}

// ---------------------------------------------------------------- PDCCH encoder
int ref_pdcch_encode(const uint8_t* payload, unsigned A, unsigned rnti, unsigned E, uint8_t* out)
{
  auto                     enc = create_pdcch_encoder_factory_sw(create_crc_calculator_factory_sw("auto"), create_polar_factory_sw())->create();
  pdcch_encoder::config_t cfg;
  cfg.E    = E;
  cfg.rnti = rnti;
  enc->encode(span<uint8_t>(out, E), span<const uint8_t>(payload, A), cfg);
  return 0;
}

// ---------------------------------------------------------------- PBCH encoder
int ref_pbch_encode(unsigned N_id, unsigned ssb_idx, unsigned L_max, int hrf, unsigned sfn, unsigned k_ssb, const uint8_t* payload, uint8_t* out)
{
  auto enc = create_pbch_encoder_factory_sw(create_crc_calculator_factory_sw("auto"), create_pseudo_random_generator_sw_factory(), create_polar_factory_sw())->create();
  pbch_encoder::pbch_msg_t msg;
  msg.N_id    = N_id;
  msg.ssb_idx = ssb_idx;
  msg.L_max   = L_max;
  msg.hrf     = hrf != 0;
  msg.sfn     = sfn;
  msg.k_ssb   = k_ssb;
  for (unsigned i = 0; i != pbch_encoder::A; ++i) {
    msg.payload[i] = payload[i];
  }
  enc->encode(span<uint8_t>(out, pbch_encoder::E), msg);
  return 0;
}

// ---------------------------------------------------------------- rx_softbuffer_pool (reservation state machine)
// The unique_rx_softbuffer a reservation returns is kept under the caller's handle: reserve = reserve + lock, drop = the
// destructor (unlock), release = unique_rx_softbuffer::release. Returns the ordinal of the softbuffer object (order of first
// appearance) or -1 for an invalid softbuffer.
struct ref_pool_t {
  std::unique_ptr<rx_softbuffer_pool>         pool;
  std::map<int, unique_rx_softbuffer>         held;
  std::vector<const void*>                    seen;
  unsigned                                    numerology;
};

void* ref_pool_create(unsigned max_codeblock_size, unsigned max_softbuffers, unsigned max_nof_codeblocks, unsigned expire_timeout_slots, unsigned numerology)
{
  rx_softbuffer_pool_config cfg;
  cfg.max_codeblock_size   = max_codeblock_size;
  cfg.max_softbuffers      = max_softbuffers;
  cfg.max_nof_codeblocks   = max_nof_codeblocks;
  cfg.expire_timeout_slots = expire_timeout_slots;
  auto* p                  = new ref_pool_t;
  p->pool                  = create_rx_softbuffer_pool(cfg);
  p->numerology            = numerology;
  return p;
}

void ref_pool_destroy(void* h)
{
  auto* p = static_cast<ref_pool_t*>(h);
  p->held.clear();
  delete p;
}

int ref_pool_reserve(void* h, unsigned slot_count, unsigned rnti, unsigned harq_id, unsigned nof_codeblocks, int handle, unsigned* nof_codeblocks_out)
{
  auto* p = static_cast<ref_pool_t*>(h);
  p->held.erase(handle);
  rx_softbuffer_identifier id;
  id.rnti        = static_cast<uint16_t>(rnti);
  id.harq_ack_id = static_cast<uint8_t>(harq_id);
  unique_rx_softbuffer u = p->pool->reserve_softbuffer(slot_point(p->numerology, slot_count), id, nof_codeblocks);
  if (!u.is_valid()) {
    return -1;
  }
  const void* addr     = &u.get();
  *nof_codeblocks_out  = u.get().get_nof_codeblocks();
  int ordinal          = -1;
  for (size_t i = 0; i != p->seen.size(); ++i) {
    if (p->seen[i] == addr) {
      ordinal = static_cast<int>(i);
    }
  }
  if (ordinal < 0) {
    ordinal = static_cast<int>(p->seen.size());
    p->seen.push_back(addr);
  }
  p->held.emplace(handle, std::move(u));
  return ordinal;
}

void ref_pool_drop(void* h, int handle)
{
  static_cast<ref_pool_t*>(h)->held.erase(handle);
}

void ref_pool_release(void* h, int handle)
{
  auto* p  = static_cast<ref_pool_t*>(h);
  auto  it = p->held.find(handle);
  if (it != p->held.end()) {
    it->second.release();
    p->held.erase(it);
  }
}

void ref_pool_run_slot(void* h, unsigned slot_count)
{
  auto* p = static_cast<ref_pool_t*>(h);
  p->pool->run_slot(slot_point(p->numerology, slot_count));
}

// ---------------------------------------------------------------- Open Fronthaul BFP (de)compression
// payload: per PRB [udCompParam (BFP only)][3 * data_width bytes], the layout ofh_uplane_message_builder_impl.cpp:145-152 serialises.
int ref_ofh_iq_decompress(int compression, const char* impl, const uint8_t* payload, unsigned nof_prb, unsigned data_width, float* out)
{
  const ofh::compression_type      type = compression == 1 ? ofh::compression_type::BFP : ofh::compression_type::none;
  const unsigned                   hdr  = compression == 1 ? 1 : 0;
  auto                             dec  = ofh::create_iq_decompressor(type, impl);
  std::vector<ofh::compressed_prb> prbs(nof_prb);
  for (unsigned p = 0; p != nof_prb; ++p) {
    const uint8_t* rec = payload + static_cast<size_t>(p) * (hdr + 3 * data_width);
    prbs[p].set_compression_param(hdr ? rec[0] : 0);
    std::memcpy(prbs[p].get_buffer().data(), rec + hdr, 3 * data_width);
    prbs[p].set_stored_size(3 * data_width);
  }
  ofh::ru_compression_params params;
  params.type       = type;
  params.data_width = data_width;
  dec->decompress(span<cf_t>(reinterpret_cast<cf_t*>(out), nof_prb * 12), prbs, params);
  return 0;
}

int ref_ofh_iq_compress(int compression, const char* impl, const float* in, unsigned nof_prb, unsigned data_width, float iq_scaling, uint8_t* payload)
{
  const ofh::compression_type      type = compression == 1 ? ofh::compression_type::BFP : ofh::compression_type::none;
  const unsigned                   hdr  = compression == 1 ? 1 : 0;
  auto                             enc  = ofh::create_iq_compressor(type, iq_scaling, impl);
  std::vector<ofh::compressed_prb> prbs(nof_prb);
  ofh::ru_compression_params       params;
  params.type       = type;
  params.data_width = data_width;
  enc->compress(prbs, span<const cf_t>(reinterpret_cast<const cf_t*>(in), nof_prb * 12), params);
  for (unsigned p = 0; p != nof_prb; ++p) {
    uint8_t* rec = payload + static_cast<size_t>(p) * (hdr + 3 * data_width);
    if (hdr) {
      rec[0] = prbs[p].get_compression_param();
    }
    span<const uint8_t> d = prbs[p].get_packed_data();
    if (d.size() != 3 * data_width) {
      return -1;
    }
    std::memcpy(rec + hdr, d.data(), d.size());
  }
  return 0;
}

// ---------------------------------------------------------------- PDCCH processor
// mapping: 0 CORESET0, 1 non-interleaved, 2 interleaved. freq_resources: one byte per group of 6 PRBs (up to 45). grid: port 0,
// [14][nof_prb_grid*12]; rb_mask_out: the PRBs the reference's CCE-to-PRB mapping selected (what the device path takes as input).
int ref_pdcch_process(int mapping, unsigned bwp_start, unsigned bwp_size, unsigned start_symbol, unsigned duration, const uint8_t* freq_resources,
                      unsigned nof_freq_resources, unsigned reg_bundle_size, unsigned interleaver_size, unsigned shift_index, unsigned numerology,
                      unsigned slot_index, unsigned rnti, unsigned n_id_dmrs, unsigned n_id_data, unsigned n_rnti, unsigned cce_index,
                      unsigned aggregation_level, float dmrs_dB, float data_dB, const uint8_t* payload, unsigned A, unsigned nof_prb_grid, float* grid,
                      uint8_t* rb_mask_out)
{
  auto prg  = create_pseudo_random_generator_sw_factory();
  auto proc = create_pdcch_processor_factory_sw(create_pdcch_encoder_factory_sw(create_crc_calculator_factory_sw("auto"), create_polar_factory_sw()),
                                                create_pdcch_modulator_factory_sw(create_channel_modulation_sw_factory(), prg),
                                                create_dmrs_pdcch_processor_factory_sw(prg))
                  ->create();
  pdcch_processor::pdu_t pdu;
  pdu.slot = slot_point(numerology, slot_index), pdu.cp = cyclic_prefix::NORMAL;
  pdu.coreset.bwp_size_rb = bwp_size, pdu.coreset.bwp_start_rb = bwp_start, pdu.coreset.start_symbol_index = start_symbol, pdu.coreset.duration = duration;
  pdu.coreset.frequency_resources = freq_resource_bitmap(nof_freq_resources);
  for (unsigned i = 0; i != nof_freq_resources; ++i) {
    if (freq_resources[i]) {
      pdu.coreset.frequency_resources.set(i, true);
    }
  }
  pdu.coreset.cce_to_reg_mapping = mapping == 0   ? pdcch_processor::cce_to_reg_mapping_type::CORESET0
                                   : mapping == 1 ? pdcch_processor::cce_to_reg_mapping_type::NON_INTERLEAVED
                                                  : pdcch_processor::cce_to_reg_mapping_type::INTERLEAVED;
  pdu.coreset.reg_bundle_size = reg_bundle_size, pdu.coreset.interleaver_size = interleaver_size, pdu.coreset.shift_index = shift_index;
  pdu.dci.rnti = rnti, pdu.dci.n_id_pdcch_dmrs = n_id_dmrs, pdu.dci.n_id_pdcch_data = n_id_data, pdu.dci.n_rnti = n_rnti, pdu.dci.cce_index = cce_index;
  pdu.dci.aggregation_level = aggregation_level, pdu.dci.dmrs_power_offset_dB = dmrs_dB, pdu.dci.data_power_offset_dB = data_dB;
  for (unsigned i = 0; i != A; ++i) {
    pdu.dci.payload.push_back(payload[i]);
  }
  pdu.dci.precoding = make_single_port();
  // the same CCE-to-PRB mapping the processor applies (pdcch_processor_impl::compute_rb_mask)
  prb_index_list prbs;
  switch (pdu.coreset.cce_to_reg_mapping) {
    case pdcch_processor::cce_to_reg_mapping_type::CORESET0:
      prbs = cce_to_prb_mapping_coreset0(bwp_start, bwp_size, duration, shift_index, aggregation_level, cce_index);
      break;
    case pdcch_processor::cce_to_reg_mapping_type::NON_INTERLEAVED:
      prbs = cce_to_prb_mapping_non_interleaved(bwp_start, pdu.coreset.frequency_resources, duration, aggregation_level, cce_index);
      break;
    default:
      prbs = cce_to_prb_mapping_interleaved(bwp_start, pdu.coreset.frequency_resources, duration, reg_bundle_size, interleaver_size, shift_index, aggregation_level,
                                            cce_index);
      break;
  }
  std::memset(rb_mask_out, 0, nof_prb_grid);
  for (uint16_t r : prbs) {
    if (r >= nof_prb_grid) {
      return -1;
    }
    rb_mask_out[r] = 1;
  }
  auto g = create_resource_grid(1, 14, nof_prb_grid * 12);
  g->set_all_zero();
  resource_grid_mapper mapper(*g);
  proc->process(mapper, pdu);
  for (unsigned l = 0; l != 14; ++l) {
    g->get(span<cf_t>(reinterpret_cast<cf_t*>(grid) + static_cast<size_t>(l) * nof_prb_grid * 12, nof_prb_grid * 12), 0, l, 0);
  }
  return 0;
}

// ---------------------------------------------------------------- SS/PBCH block processor
// pattern_case 0..4 = A..E; returns the first symbol (in the slot) and first subcarrier the reference derived (ssb_mapping.h), which are
// the inputs of the device path. grid: port 0, [14][nof_prb_grid*12].
int ref_ssb_process(unsigned numerology, unsigned sfn, unsigned slot_in_frame, unsigned N_id, float beta_pss, unsigned ssb_idx, unsigned L_max,
                    unsigned common_scs_khz, unsigned subcarrier_offset, unsigned offset_to_pointA, int pattern_case, const uint8_t* payload,
                    unsigned nof_prb_grid, float* grid, unsigned* l_start_out, unsigned* k_start_out)
{
  ssb_processor_factory_sw_configuration cfg;
  auto                                   prg = create_pseudo_random_generator_sw_factory();
  cfg.encoder_factory   = create_pbch_encoder_factory_sw(create_crc_calculator_factory_sw("auto"), prg, create_polar_factory_sw());
  cfg.modulator_factory = create_pbch_modulator_factory_sw(create_channel_modulation_sw_factory(), prg);
  cfg.dmrs_factory      = create_dmrs_pbch_processor_factory_sw(prg);
  cfg.pss_factory       = create_pss_processor_factory_sw();
  cfg.sss_factory       = create_sss_processor_factory_sw();
  auto proc             = create_ssb_processor_factory_sw(cfg)->create();
  ssb_processor::pdu_t pdu;
  pdu.slot = slot_point(numerology, sfn, slot_in_frame), pdu.phys_cell_id = N_id, pdu.beta_pss = beta_pss, pdu.ssb_idx = ssb_idx, pdu.L_max = L_max;
  pdu.common_scs        = common_scs_khz == 15 ? subcarrier_spacing::kHz15 : (common_scs_khz == 30 ? subcarrier_spacing::kHz30 : subcarrier_spacing::kHz60);
  pdu.subcarrier_offset = subcarrier_offset, pdu.offset_to_pointA = offset_to_pointA;
  pdu.pattern_case      = static_cast<ssb_pattern_case>(pattern_case);
  for (unsigned i = 0; i != 32; ++i) {
    pdu.bch_payload[i] = payload[i];
  }
  pdu.ports.push_back(0);
  const unsigned l_in_burst = ssb_get_l_first(pdu.pattern_case, ssb_idx);
  if (l_in_burst / 14 != pdu.slot.hrf_slot_index()) {
    return -2; // the slot does not carry this block
  }
  *l_start_out = l_in_burst % 14;
  *k_start_out = ssb_get_k_first(to_frequency_range(pdu.pattern_case), to_subcarrier_spacing(pdu.pattern_case), pdu.common_scs, pdu.offset_to_pointA, pdu.subcarrier_offset);
  if (*k_start_out + 240 > nof_prb_grid * 12) {
    return -1;
  }
  auto g = create_resource_grid(1, 14, nof_prb_grid * 12);
  g->set_all_zero();
  proc->process(*g, pdu);
  for (unsigned l = 0; l != 14; ++l) {
    g->get(span<cf_t>(reinterpret_cast<cf_t*>(grid) + static_cast<size_t>(l) * nof_prb_grid * 12, nof_prb_grid * 12), 0, l, 0);
  }
  return 0;
}

// ---------------------------------------------------------------- NZP-CSI-RS generator
// Runs the reference generator and returns the per-port patterns its get_csi_rs_pattern() produced (the inputs of the device path).
// grid: [nof_ports][14][nof_prb_grid*12], port i of the configuration on grid port i.
int ref_csi_rs_map(unsigned numerology, unsigned slot_index, unsigned start_rb, unsigned nof_rb, unsigned row, const unsigned* k_ref, unsigned nof_k_ref,
                   unsigned l0, unsigned l1, unsigned cdm, unsigned density, unsigned scrambling_id, float amplitude, unsigned nof_ports, unsigned nof_prb_grid,
                   float* grid, unsigned* rb_begin_end_stride, uint16_t* re_mask, uint16_t* symbol_mask)
{
  auto                           gen = create_nzp_csi_rs_generator_factory_sw(create_pseudo_random_generator_sw_factory())->create();
  nzp_csi_rs_generator::config_t cfg;
  cfg.slot = slot_point(numerology, slot_index), cfg.cp = cyclic_prefix::NORMAL, cfg.start_rb = start_rb, cfg.nof_rb = nof_rb, cfg.csi_rs_mapping_table_row = row;
  for (unsigned i = 0; i != nof_k_ref; ++i) {
    cfg.freq_allocation_ref_idx.push_back(k_ref[i]);
  }
  cfg.symbol_l0 = l0, cfg.symbol_l1 = l1, cfg.cdm = static_cast<csi_rs_cdm_type>(cdm), cfg.freq_density = static_cast<csi_rs_freq_density_type>(density);
  cfg.scrambling_id = scrambling_id, cfg.amplitude = amplitude, cfg.pmi = 0;
  for (unsigned i = 0; i != nof_ports; ++i) {
    cfg.ports.push_back(i);
  }
  csi_rs_pattern_configuration pc;
  pc.start_rb = start_rb, pc.nof_rb = nof_rb, pc.csi_rs_mapping_table_row = row, pc.freq_allocation_ref_idx = cfg.freq_allocation_ref_idx;
  pc.symbol_l0 = l0, pc.symbol_l1 = l1, pc.cdm = cfg.cdm, pc.freq_density = cfg.freq_density, pc.nof_ports = nof_ports;
  csi_rs_pattern pat     = get_csi_rs_pattern(pc);
  rb_begin_end_stride[0] = pat.rb_begin, rb_begin_end_stride[1] = pat.rb_end, rb_begin_end_stride[2] = pat.rb_stride;
  for (unsigned p = 0; p != nof_ports; ++p) {
    uint16_t rm = 0, sm = 0;
    for (unsigned k = 0; k != 12; ++k) {
      rm |= static_cast<uint16_t>(pat.prb_patterns[p].re_mask.test(k) ? (1U << k) : 0U);
    }
    for (unsigned l = 0; l != 14; ++l) {
      sm |= static_cast<uint16_t>(pat.prb_patterns[p].symbol_mask.test(l) ? (1U << l) : 0U);
    }
    re_mask[p] = rm, symbol_mask[p] = sm;
  }
  auto g = create_resource_grid(nof_ports, 14, nof_prb_grid * 12);
  g->set_all_zero();
  gen->map(*g, cfg);
  for (unsigned p = 0; p != nof_ports; ++p) {
    for (unsigned l = 0; l != 14; ++l) {
      g->get(span<cf_t>(reinterpret_cast<cf_t*>(grid) + (static_cast<size_t>(p) * 14 + l) * nof_prb_grid * 12, nof_prb_grid * 12), p, l, 0);
    }
  }
  return 0;
}

// ---------------------------------------------------------------- CPU baseline: the whole receive chain, driven like the reference benchmark
// T worker threads pinned 1:1 to the CPUs in `cpus`, one instance of every block per thread (ofdm_slot_demodulator over the
// generic DFT, pusch_processor from the software factories with the AVX2 LDPC decoder / rate dematcher, an rx_softbuffer_pool), each
// looping over the same `nslots` input slots until `seconds` have passed -- the shape of
// tests/benchmarks/phy/upper/channel_processors/pusch_processor_benchmark.cpp:576-632 (thread_process / pinned workers).
// with_ofdm = 1: time-domain samples -> transport block (what bench.py's `value` covers); 0: the grids are demodulated once
// outside the timed region and only pusch_processor::process is timed; 2: only the pusch_decoder (dematch + decode) on LLRs.
// samples: [nslots][slot_samples] cf_t. Slot s is slot-in-frame s (its subframe slot index is s % 2). Returns elapsed seconds;
// slots_done[t] / tb_ok[t] per thread.
namespace {
struct chain_notifier : public pusch_processor_result_notifier {
  bool ok = false;
  void on_csi(const channel_state_information&) override {}
  void on_uci(const pusch_processor_result_control&) override {}
  void on_sch(const pusch_processor_result_data& d) override { ok = d.data.tb_crc_ok; }
};
} // namespace

double ref_pusch_chain_bench(unsigned     nthreads,
                             const int*   cpus,
                             double       seconds,
                             int          with_ofdm,
                             const float* samples,
                             unsigned     nslots,
                             unsigned     slot_samples,
                             unsigned     nof_prb,
                             int          mod,
                             unsigned     tbs_bits,
                             unsigned     rnti,
                             unsigned     n_id,
                             unsigned     dmrs_scrambling_id,
                             unsigned     dft_size,
                             unsigned     window_offset,
                             float        ofdm_scale,
                             double       center_freq_hz,
                             unsigned     max_iter,
                             int          early_stop,
                             uint64_t*    slots_done,
                             uint64_t*    tb_ok)
{
  const unsigned           nsc = nof_prb * 12;
  std::atomic<int>         ready{0};
  std::atomic<bool>        go{false}, stop{false};
  std::vector<std::thread> workers;
  for (unsigned t = 0; t != nthreads; ++t) {
    slots_done[t] = 0, tb_ok[t] = 0;
    workers.emplace_back([&, t]() {
      if (cpus != nullptr && cpus[t] >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(cpus[t], &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
      }
      auto                               prg = create_pseudo_random_generator_sw_factory();
      ofdm_factory_generic_configuration fc;
      fc.dft_factory = std::make_shared<generic_dft_factory>();
      ofdm_demodulator_configuration oc;
      oc.numerology = 1, oc.bw_rb = nof_prb, oc.dft_size = dft_size, oc.cp = cyclic_prefix::NORMAL;
      oc.nof_samples_window_offset = window_offset, oc.scale = ofdm_scale, oc.center_freq_hz = center_freq_hz;
      auto ofdm = create_ofdm_demodulator_factory_generic(fc)->create_ofdm_slot_demodulator(oc);
      pusch_decoder_factory_sw_configuration dc;
      dc.crc_factory       = create_crc_calculator_factory_sw("auto");
      dc.decoder_factory   = create_ldpc_decoder_factory_sw("avx2");
      dc.dematcher_factory = create_ldpc_rate_dematcher_factory_sw("avx2");
      dc.segmenter_factory = create_ldpc_segmenter_rx_factory_sw();
      uci_decoder_factory_sw_configuration uc;
      uc.decoder_factory = create_short_block_detector_factory_sw();
      pusch_processor_factory_sw_configuration pc;
      pc.estimator_factory =
          create_dmrs_pusch_estimator_factory_sw(prg, create_port_channel_estimator_factory_sw(std::make_shared<generic_dft_factory>()));
      pc.demodulator_factory = create_pusch_demodulator_factory_sw(create_channel_equalizer_factory_zf(), create_channel_modulation_sw_factory(), prg);
      pc.demux_factory       = create_ulsch_demultiplex_factory_sw();
      pc.decoder_factory     = create_pusch_decoder_factory_sw(dc);
      pc.uci_dec_factory     = create_uci_decoder_factory_sw(uc);
      pc.ch_estimate_dimensions.nof_prb = MAX_RB, pc.ch_estimate_dimensions.nof_symbols = MAX_NSYMB_PER_SLOT;
      pc.ch_estimate_dimensions.nof_rx_ports = 1, pc.ch_estimate_dimensions.nof_tx_layers = 1;
      pc.dec_nof_iterations = max_iter, pc.dec_enable_early_stop = early_stop != 0;
      auto proc = create_pusch_processor_factory_sw(pc)->create();
      rx_softbuffer_pool_config spc;
      spc.max_codeblock_size = ldpc::MAX_CODEBLOCK_SIZE, spc.max_softbuffers = 2, spc.max_nof_codeblocks = 128, spc.expire_timeout_slots = 100000;
      auto                 pool = create_rx_softbuffer_pool(spc);
      const unsigned       nof_cbs = ldpc::compute_nof_codeblocks(units::bits(tbs_bits), ldpc_base_graph_type::BG1);
      std::vector<uint8_t> tb(tbs_bits / 8);
      // Grids: demodulated per iteration (with_ofdm == 1) or once up front.
      std::vector<std::unique_ptr<resource_grid>> grids;
      for (unsigned s = 0; s != nslots; ++s) {
        grids.push_back(create_resource_grid(1, 14, nsc));
        ofdm->demodulate(*grids[s], span<const cf_t>(reinterpret_cast<const cf_t*>(samples) + size_t(s) * slot_samples, slot_samples), 0, s % 2);
      }
      symbol_slot_mask dm(14);
      dm.set(2);
      ready.fetch_add(1);
      while (!go.load()) {
        std::this_thread::yield();
      }
      unsigned k = t;
      while (!stop.load(std::memory_order_relaxed)) {
        const unsigned s = k % nslots;
        if (with_ofdm == 1) {
          ofdm->demodulate(*grids[s], span<const cf_t>(reinterpret_cast<const cf_t*>(samples) + size_t(s) * slot_samples, slot_samples), 0, s % 2);
        }
        pusch_processor::pdu_t pdu;
        pdu.slot = slot_point(1, s), pdu.rnti = rnti, pdu.bwp_size_rb = nof_prb, pdu.bwp_start_rb = 0, pdu.cp = cyclic_prefix::NORMAL;
        pdu.mcs_descr.modulation = mod_from_bits(mod), pdu.mcs_descr.target_code_rate = 0.9F;
        pdu.codeword.emplace();
        pdu.codeword.value().rv = 0, pdu.codeword.value().ldpc_base_graph = ldpc_base_graph_type::BG1, pdu.codeword.value().new_data = true;
        pdu.uci = {};
        pdu.uci.alpha_scaling = 1.0F, pdu.uci.beta_offset_harq_ack = 20.0F, pdu.uci.beta_offset_csi_part1 = 6.25F, pdu.uci.beta_offset_csi_part2 = 6.25F;
        pdu.n_id = n_id, pdu.nof_tx_layers = 1;
        pdu.rx_ports.push_back(0);
        pdu.dmrs_symbol_mask = dm, pdu.dmrs = dmrs_type::TYPE1, pdu.scrambling_id = dmrs_scrambling_id, pdu.n_scid = false;
        pdu.nof_cdm_groups_without_data = 2;
        pdu.freq_alloc         = rb_allocation::make_type1(0, nof_prb);
        pdu.start_symbol_index = 0, pdu.nof_symbols = 14, pdu.tbs_lbrm_bytes = ldpc::MAX_CODEBLOCK_SIZE / 8;
        rx_softbuffer_identifier id;
        id.rnti = static_cast<uint16_t>(rnti), id.harq_ack_id = 0;
        unique_rx_softbuffer sb = pool->reserve_softbuffer(slot_point(1, s), id, nof_cbs);
        chain_notifier       n;
        proc->process(tb, sb.get(), n, *grids[s], pdu);
        sb.release();
        ++slots_done[t];
        tb_ok[t] += n.ok ? 1 : 0;
        ++k;
      }
    });
  }
  while (ready.load() != (int)nthreads) {
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  const double t0 = now_s();
  go.store(true);
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  stop.store(true);
  for (auto& w : workers) {
    w.join();
  }
  return now_s() - t0;
}

// Decoder-only leg (pusch_decoder: rate dematcher + LDPC decoder + CRCs), same threading. llrs: [nslots][cw_len].
double ref_pusch_decoder_bench(unsigned      nthreads,
                               const int*    cpus,
                               double        seconds,
                               const int8_t* llrs,
                               unsigned      nslots,
                               unsigned      cw_len,
                               int           mod,
                               unsigned      nof_ch_symbols,
                               unsigned      tbs_bits,
                               unsigned      max_iter,
                               int           early_stop,
                               uint64_t*     slots_done,
                               uint64_t*     tb_ok)
{
  std::atomic<int>         ready{0};
  std::atomic<bool>        go{false}, stop{false};
  std::vector<std::thread> workers;
  for (unsigned t = 0; t != nthreads; ++t) {
    slots_done[t] = 0, tb_ok[t] = 0;
    workers.emplace_back([&, t]() {
      if (cpus != nullptr && cpus[t] >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(cpus[t], &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
      }
      void* h   = ref_pusch_decoder_create(1);
      int   rv0 = 0, ok = 0, mm[2];
      std::vector<uint8_t> tb(tbs_bits / 8);
      ready.fetch_add(1);
      while (!go.load()) {
        std::this_thread::yield();
      }
      unsigned k = t;
      while (!stop.load(std::memory_order_relaxed)) {
        ref_pusch_decode(h, 1, mod, 0, 1, nof_ch_symbols, tbs_bits / 8, 1, &rv0, llrs + size_t(k % nslots) * cw_len, cw_len, max_iter, early_stop, tb.data(), &ok, mm);
        ++slots_done[t];
        tb_ok[t] += ok;
        ++k;
      }
      ref_pusch_decoder_destroy(h);
    });
  }
  while (ready.load() != (int)nthreads) {
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  const double t0 = now_s();
  go.store(true);
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  stop.store(true);
  for (auto& w : workers) {
    w.join();
  }
  return now_s() - t0;
}

// ---------------------------------------------------------------- TBS / base graph of an allocation (the reference's own calculators)
// tbs_calculator_calculate (lib/scheduler/support/tbs_calculator.cpp, TS 38.214 5.1.3.2) and get_ldpc_base_graph (sch_mcs / ldpc helpers).
unsigned ref_tbs_calculate(unsigned nof_symb_sh, unsigned nof_dmrs_prb, unsigned nof_oh_prb, int mod_bits, float rate_x1024, unsigned nof_layers, unsigned n_prb)
{
  tbs_calculator_configuration c;
  c.nof_symb_sh = nof_symb_sh, c.nof_dmrs_prb = nof_dmrs_prb, c.nof_oh_prb = nof_oh_prb;
  c.mcs_descr.modulation = mod_from_bits(mod_bits), c.mcs_descr.target_code_rate = rate_x1024;
  c.nof_layers = nof_layers, c.tb_scaling_field = 0, c.n_prb = n_prb;
  return tbs_calculator_calculate(c);
}

int ref_ldpc_base_graph(float rate_x1024, unsigned tbs_bits)
{
  return get_ldpc_base_graph(rate_x1024 / 1024.0F, units::bits(tbs_bits)) == ldpc_base_graph_type::BG1 ? 1 : 2;
}

// ---------------------------------------------------------------- receive chain of a slot that carries SEVERAL PUSCH PDUs
// ofdm_slot_demodulator once per slot, then pusch_processor::process per PDU (what upper_phy_rx_symbol_handler_impl does with the
// PDUs of a slot), one processor instance per pinned thread like pusch_processor_benchmark. pdus: npdus records of 8 unsigned
// {rb_start, nof_prb, mod bits, TBS bits, base graph, rnti, n_id, target code rate x1024}. isa: 1 = avx2, 2 = avx512 decoder / dematcher.
double ref_pusch_chain_bench_multi(unsigned        nthreads,
                                   const int*      cpus,
                                   double          seconds,
                                   int             with_ofdm,
                                   const float*    samples,
                                   unsigned        nslots,
                                   unsigned        slot_samples,
                                   unsigned        grid_prb,
                                   unsigned        npdus,
                                   const unsigned* pdus,
                                   unsigned        dmrs_scrambling_id,
                                   unsigned        dft_size,
                                   unsigned        window_offset,
                                   float           ofdm_scale,
                                   double          center_freq_hz,
                                   unsigned        max_iter,
                                   int             early_stop,
                                   int             isa,
                                   uint64_t*       slots_done,
                                   uint64_t*       tb_ok)
{
  const unsigned           nsc = grid_prb * 12;
  std::atomic<int>         ready{0};
  std::atomic<bool>        go{false}, stop{false};
  std::vector<std::thread> workers;
  for (unsigned t = 0; t != nthreads; ++t) {
    slots_done[t] = 0, tb_ok[t] = 0;
    workers.emplace_back([&, t]() {
      if (cpus != nullptr && cpus[t] >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(cpus[t], &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
      }
      auto                               prg = create_pseudo_random_generator_sw_factory();
      ofdm_factory_generic_configuration fc;
      fc.dft_factory = std::make_shared<generic_dft_factory>();
      ofdm_demodulator_configuration oc;
      oc.numerology = 1, oc.bw_rb = grid_prb, oc.dft_size = dft_size, oc.cp = cyclic_prefix::NORMAL;
      oc.nof_samples_window_offset = window_offset, oc.scale = ofdm_scale, oc.center_freq_hz = center_freq_hz;
      auto ofdm = create_ofdm_demodulator_factory_generic(fc)->create_ofdm_slot_demodulator(oc);
      pusch_decoder_factory_sw_configuration dc;
      dc.crc_factory       = create_crc_calculator_factory_sw("auto");
      dc.decoder_factory   = create_ldpc_decoder_factory_sw(impl_name(isa));
      dc.dematcher_factory = create_ldpc_rate_dematcher_factory_sw(impl_name(isa));
      dc.segmenter_factory = create_ldpc_segmenter_rx_factory_sw();
      uci_decoder_factory_sw_configuration uc;
      uc.decoder_factory = create_short_block_detector_factory_sw();
      pusch_processor_factory_sw_configuration pc;
      pc.estimator_factory =
          create_dmrs_pusch_estimator_factory_sw(prg, create_port_channel_estimator_factory_sw(std::make_shared<generic_dft_factory>()));
      pc.demodulator_factory = create_pusch_demodulator_factory_sw(create_channel_equalizer_factory_zf(), create_channel_modulation_sw_factory(), prg);
      pc.demux_factory       = create_ulsch_demultiplex_factory_sw();
      pc.decoder_factory     = create_pusch_decoder_factory_sw(dc);
      pc.uci_dec_factory     = create_uci_decoder_factory_sw(uc);
      pc.ch_estimate_dimensions.nof_prb = MAX_RB, pc.ch_estimate_dimensions.nof_symbols = MAX_NSYMB_PER_SLOT;
      pc.ch_estimate_dimensions.nof_rx_ports = 1, pc.ch_estimate_dimensions.nof_tx_layers = 1;
      pc.dec_nof_iterations = max_iter, pc.dec_enable_early_stop = early_stop != 0;
      auto proc = create_pusch_processor_factory_sw(pc)->create();
      rx_softbuffer_pool_config spc;
      spc.max_codeblock_size = ldpc::MAX_CODEBLOCK_SIZE, spc.max_softbuffers = 2 * npdus, spc.max_nof_codeblocks = 256, spc.expire_timeout_slots = 100000;
      auto                 pool = create_rx_softbuffer_pool(spc);
      std::vector<uint8_t> tb(1277992 / 8);
      std::vector<std::unique_ptr<resource_grid>> grids;
      for (unsigned s = 0; s != nslots; ++s) {
        grids.push_back(create_resource_grid(1, 14, nsc));
        ofdm->demodulate(*grids[s], span<const cf_t>(reinterpret_cast<const cf_t*>(samples) + size_t(s) * slot_samples, slot_samples), 0, s % 2);
      }
      symbol_slot_mask dm(14);
      dm.set(2);
      ready.fetch_add(1);
      while (!go.load()) {
        std::this_thread::yield();
      }
      unsigned k = t;
      while (!stop.load(std::memory_order_relaxed)) {
        const unsigned s = k % nslots;
        if (with_ofdm == 1) {
          ofdm->demodulate(*grids[s], span<const cf_t>(reinterpret_cast<const cf_t*>(samples) + size_t(s) * slot_samples, slot_samples), 0, s % 2);
        }
        for (unsigned u = 0; u != npdus; ++u) {
          const unsigned*        q = pdus + 8 * u;
          pusch_processor::pdu_t pdu;
          pdu.slot = slot_point(1, s), pdu.rnti = q[5], pdu.bwp_size_rb = grid_prb, pdu.bwp_start_rb = 0, pdu.cp = cyclic_prefix::NORMAL;
          pdu.mcs_descr.modulation = mod_from_bits(q[2]), pdu.mcs_descr.target_code_rate = static_cast<float>(q[7]); // R x 1024 (sch_mcs.h)
          pdu.codeword.emplace();
          pdu.codeword.value().rv = 0, pdu.codeword.value().new_data = true;
          pdu.codeword.value().ldpc_base_graph = q[4] == 1 ? ldpc_base_graph_type::BG1 : ldpc_base_graph_type::BG2;
          pdu.uci = {};
          pdu.uci.alpha_scaling = 1.0F, pdu.uci.beta_offset_harq_ack = 20.0F, pdu.uci.beta_offset_csi_part1 = 6.25F, pdu.uci.beta_offset_csi_part2 = 6.25F;
          pdu.n_id = q[6], pdu.nof_tx_layers = 1;
          pdu.rx_ports.push_back(0);
          pdu.dmrs_symbol_mask = dm, pdu.dmrs = dmrs_type::TYPE1, pdu.scrambling_id = dmrs_scrambling_id, pdu.n_scid = false;
          pdu.nof_cdm_groups_without_data = 2;
          pdu.freq_alloc         = rb_allocation::make_type1(q[0], q[1]);
          pdu.start_symbol_index = 0, pdu.nof_symbols = 14, pdu.tbs_lbrm_bytes = ldpc::MAX_CODEBLOCK_SIZE / 8;
          rx_softbuffer_identifier id;
          id.rnti = static_cast<uint16_t>(q[5]), id.harq_ack_id = 0;
          const unsigned nof_cbs = ldpc::compute_nof_codeblocks(units::bits(q[3]), pdu.codeword.value().ldpc_base_graph);
          unique_rx_softbuffer sb = pool->reserve_softbuffer(slot_point(1, s), id, nof_cbs);
          chain_notifier       n;
          proc->process(span<uint8_t>(tb.data(), q[3] / 8), sb.get(), n, *grids[s], pdu);
          sb.release();
          tb_ok[t] += n.ok ? 1 : 0;
        }
        ++slots_done[t];
        ++k;
      }
    });
  }
  while (ready.load() != (int)nthreads) {
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  const double t0 = now_s();
  go.store(true);
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  stop.store(true);
  for (auto& w : workers) {
    w.join();
  }
  return now_s() - t0;
}

// Decoder-only leg with the instruction set of the decoder / dematcher as a parameter (1 = avx2, 2 = avx512: the reference's
// "avx512" classes, ldpc_decoder_avx512.cpp), otherwise ref_pusch_decoder_bench.
double ref_pusch_decoder_bench_isa(unsigned nthreads, const int* cpus, double seconds, const int8_t* llrs, unsigned nslots, unsigned cw_len, int mod,
                                   unsigned nof_ch_symbols, unsigned tbs_bits, unsigned max_iter, int early_stop, int isa, uint64_t* slots_done,
                                   uint64_t* tb_ok)
{
  std::atomic<int>         ready{0};
  std::atomic<bool>        go{false}, stop{false};
  std::atomic<int>         failed{0};
  std::vector<std::thread> workers;
  for (unsigned t = 0; t != nthreads; ++t) {
    slots_done[t] = 0, tb_ok[t] = 0;
    workers.emplace_back([&, t]() {
      if (cpus != nullptr && cpus[t] >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(cpus[t], &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
      }
      void* h = ref_pusch_decoder_create(isa);
      if (h == nullptr) {
        failed.fetch_add(1);
        ready.fetch_add(1);
        return;
      }
      int                  rv0 = 0, ok = 0, mm[2];
      std::vector<uint8_t> tb(tbs_bits / 8);
      ready.fetch_add(1);
      while (!go.load()) {
        std::this_thread::yield();
      }
      unsigned k = t;
      while (!stop.load(std::memory_order_relaxed)) {
        ref_pusch_decode(h, 1, mod, 0, 1, nof_ch_symbols, tbs_bits / 8, 1, &rv0, llrs + size_t(k % nslots) * cw_len, cw_len, max_iter, early_stop, tb.data(), &ok, mm);
        ++slots_done[t];
        tb_ok[t] += ok;
        ++k;
      }
      ref_pusch_decoder_destroy(h);
    });
  }
  while (ready.load() != (int)nthreads) {
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  const double t0 = now_s();
  go.store(true);
  std::this_thread::sleep_for(std::chrono::duration<double>(failed.load() ? 0.0 : seconds));
  stop.store(true);
  for (auto& w : workers) {
    w.join();
  }
  return failed.load() ? -1.0 : now_s() - t0;
}

// ---------------------------------------------------------------- transmit chain: pdsch_processor + ofdm_slot_modulator
// pdsch_processor::process of one full-band PDU (pdsch_processor_benchmark.cpp:416-520 style: one processor per pinned thread) followed,
// with_ofdm = 1, by ofdm_slot_modulator::modulate of the slot's grid. tbs: nslots transport blocks of tbs_bits / 8 bytes.
double ref_pdsch_chain_bench(unsigned nthreads, const int* cpus, double seconds, int with_ofdm, const uint8_t* tbs, unsigned nslots, unsigned nof_prb, int mod,
                             unsigned tbs_bits, unsigned rnti, unsigned n_id, unsigned dmrs_scrambling_id, unsigned dft_size, float ofdm_scale,
                             double center_freq_hz, uint64_t* slots_done, float* checksum)
{
  const unsigned           nsc = nof_prb * 12;
  std::atomic<int>         ready{0};
  std::atomic<bool>        go{false}, stop{false};
  std::vector<std::thread> workers;
  for (unsigned t = 0; t != nthreads; ++t) {
    slots_done[t] = 0;
    workers.emplace_back([&, t]() {
      if (cpus != nullptr && cpus[t] >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(cpus[t], &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
      }
      auto                                   crcf = create_crc_calculator_factory_sw("auto");
      auto                                   prg  = create_pseudo_random_generator_sw_factory();
      pdsch_encoder_factory_sw_configuration ec;
      ec.encoder_factory      = create_ldpc_encoder_factory_sw("avx2");
      ec.rate_matcher_factory = create_ldpc_rate_matcher_factory_sw();
      ec.segmenter_factory    = create_ldpc_segmenter_tx_factory_sw(crcf);
      auto proc = create_pdsch_processor_factory_sw(create_pdsch_encoder_factory_sw(ec),
                                                    create_pdsch_modulator_factory_sw(create_channel_modulation_sw_factory(), prg),
                                                    create_dmrs_pdsch_processor_factory_sw(prg))
                      ->create();
      ofdm_factory_generic_configuration fc;
      fc.dft_factory = std::make_shared<generic_dft_factory>();
      ofdm_modulator_configuration oc;
      oc.numerology = 1, oc.bw_rb = nof_prb, oc.dft_size = dft_size, oc.cp = cyclic_prefix::NORMAL, oc.scale = ofdm_scale, oc.center_freq_hz = center_freq_hz;
      auto              ofdm = create_ofdm_modulator_factory_generic(fc)->create_ofdm_slot_modulator(oc);
      auto              grid = create_resource_grid(1, 14, nsc);
      std::vector<cf_t> out(ofdm->get_slot_size(0));
      symbol_slot_mask  dm(14);
      dm.set(2);
      ready.fetch_add(1);
      while (!go.load()) {
        std::this_thread::yield();
      }
      unsigned k = t;
      while (!stop.load(std::memory_order_relaxed)) {
        const unsigned         s = k % nslots;
        pdsch_processor::pdu_t pdu;
        pdu.slot = slot_point(1, s % 20), pdu.rnti = rnti, pdu.bwp_size_rb = nof_prb, pdu.bwp_start_rb = 0, pdu.cp = cyclic_prefix::NORMAL;
        pdu.codewords.push_back(pdsch_processor::codeword_description{mod_from_bits(mod), 0});
        pdu.n_id = n_id;
        pdu.ports.push_back(0);
        pdu.ref_point = pdsch_processor::pdu_t::CRB0, pdu.dmrs_symbol_mask = dm, pdu.dmrs = dmrs_type::TYPE1, pdu.scrambling_id = dmrs_scrambling_id, pdu.n_scid = false;
        pdu.nof_cdm_groups_without_data = 2, pdu.freq_alloc = rb_allocation::make_type1(0, nof_prb), pdu.start_symbol_index = 0, pdu.nof_symbols = 14;
        pdu.ldpc_base_graph = ldpc_base_graph_type::BG1, pdu.tbs_lbrm_bytes = ldpc::MAX_CODEBLOCK_SIZE / 8;
        pdu.ratio_pdsch_dmrs_to_sss_dB = -3.0F, pdu.ratio_pdsch_data_to_sss_dB = 0.0F;
        static_vector<span<const uint8_t>, pdsch_processor::MAX_NOF_TRANSPORT_BLOCKS> data;
        data.emplace_back(span<const uint8_t>(tbs + size_t(s) * (tbs_bits / 8), tbs_bits / 8));
        proc->process(*grid, data, pdu);
        if (with_ofdm == 1) {
          ofdm->modulate(out, *grid, 0, s % 2);
        }
        ++slots_done[t];
        ++k;
      }
      if (t == 0 && checksum != nullptr) {
        float a = 0.0F;
        for (const cf_t& v : out) {
          a += std::abs(v);
        }
        *checksum = a;
      }
    });
  }
  while (ready.load() != (int)nthreads) {
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  const double t0 = now_s();
  go.store(true);
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  stop.store(true);
  for (auto& w : workers) {
    w.join();
  }
  return now_s() - t0;
}

// ---------------------------------------------------------------- polar: PDCCH encode and SSC decode chain (polar_chain_test.cpp:156-210)
// Per codeword: pdcch_encoder::encode (CRC24C + RNTI mask, interleaver, polar encoder, rate matcher) when stage = 0; rate dematcher +
// SSC decoder + deallocator on the given LLRs when stage = 1. payloads: ncw x A bits, llrs: ncw x E. Objects created once per thread.
double ref_polar_chain_bench(unsigned nthreads, const int* cpus, double seconds, int stage, unsigned A, unsigned E, const uint8_t* payloads, const int8_t* llrs,
                             unsigned ncw, uint64_t* done)
{
  std::atomic<int>         ready{0};
  std::atomic<bool>        go{false}, stop{false};
  std::vector<std::thread> workers;
  const unsigned           K = A + 24;
  for (unsigned t = 0; t != nthreads; ++t) {
    done[t] = 0;
    workers.emplace_back([&, t]() {
      if (cpus != nullptr && cpus[t] >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(cpus[t], &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
      }
      auto enc  = create_pdcch_encoder_factory_sw(create_crc_calculator_factory_sw("auto"), create_polar_factory_sw())->create();
      auto f    = create_polar_factory_sw();
      auto code = f->create_code();
      code->set(K, E, 9, polar_code_ibil::not_present);
      const unsigned                    N   = code->get_N();
      auto                              rdm = f->create_rate_dematcher();
      auto                              dec = f->create_decoder(9);
      auto                              dea = f->create_deallocator();
      std::vector<log_likelihood_ratio> dem(N);
      std::vector<uint8_t>              u(N), msg(K), out(E);
      ready.fetch_add(1);
      while (!go.load()) {
        std::this_thread::yield();
      }
      unsigned k = t;
      while (!stop.load(std::memory_order_relaxed)) {
        const unsigned c = k % ncw;
        if (stage == 0) {
          pdcch_encoder::config_t cfg;
          cfg.E = E, cfg.rnti = 0x1234 + c;
          enc->encode(out, span<const uint8_t>(payloads + size_t(c) * A, A), cfg);
        } else {
          rdm->rate_dematch(dem, span<const log_likelihood_ratio>(reinterpret_cast<const log_likelihood_ratio*>(llrs) + size_t(c) * E, E), *code);
          dec->decode(u, dem, *code);
          dea->deallocate(msg, u, *code);
        }
        ++done[t];
        ++k;
      }
    });
  }
  while (ready.load() != (int)nthreads) {
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  const double t0 = now_s();
  go.store(true);
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  stop.store(true);
  for (auto& w : workers) {
    w.join();
  }
  return now_s() - t0;
}

// ---------------------------------------------------------------- channel equalizer on its own
// create_channel_equalizer_factory_zf()->create()->equalize() on tensors of the reference's own types.
int ref_channel_equalize(unsigned nof_re, unsigned nof_rx_ports, unsigned nof_tx_layers, const float* ch_symbols, const float* ch_estimates, float noise_var,
                         float tx_scaling, float* eq_symbols, float* eq_noise_vars)
{
  auto                                                                    eq = create_channel_equalizer_factory_zf()->create();
  dynamic_tensor<2, cf_t, channel_equalizer::re_list::dims>                y({nof_re, nof_rx_ports}), z({nof_re, nof_tx_layers});
  dynamic_tensor<2, float, channel_equalizer::re_list::dims>        nv({nof_re, nof_tx_layers});
  dynamic_tensor<3, cf_t, channel_equalizer::ch_est_list::dims>            h({nof_re, nof_rx_ports, nof_tx_layers});
  span<cf_t>                                                               yv = y.get_data(), hv = h.get_data();
  std::memcpy(yv.data(), ch_symbols, sizeof(cf_t) * yv.size());
  std::memcpy(hv.data(), ch_estimates, sizeof(cf_t) * hv.size());
  std::vector<float> nvars(nof_rx_ports, noise_var);
  eq->equalize(z, nv, y, h, nvars, tx_scaling);
  std::memcpy(eq_symbols, z.get_data().data(), sizeof(cf_t) * z.get_data().size());
  std::memcpy(eq_noise_vars, nv.get_data().data(), sizeof(float) * nv.get_data().size());
  return 0;
}

// ---------------------------------------------------------------- UL-SCH demultiplexer (UCI on PUSCH)
// ulsch_demultiplex_impl through its factory: demultiplex() and get_placeholders(). Stream lengths: n_sch / G_ack / G_csi1 / G_csi2 LLRs.
int ref_ulsch_demultiplex(int mod, unsigned nof_layers, unsigned nof_prb, unsigned start_symbol, unsigned nof_symbols, unsigned G_rvd, int dmrs_type2,
                          unsigned dmrs_symbols_mask, unsigned cdm_groups, unsigned G_ack, unsigned G_csi1, unsigned G_csi2, unsigned O_ack, unsigned O_csi1,
                          unsigned O_csi2, const int8_t* in, unsigned n_in, int8_t* sch, unsigned n_sch, int8_t* ack, int8_t* csi1, int8_t* csi2,
                          uint16_t* placeholders, unsigned* nof_placeholders)
{
  auto                             dm = create_ulsch_demultiplex_factory_sw()->create();
  ulsch_demultiplex::configuration cfg;
  cfg.modulation = mod_from_bits(mod), cfg.nof_layers = nof_layers, cfg.nof_prb = nof_prb, cfg.start_symbol_index = start_symbol, cfg.nof_symbols = nof_symbols;
  cfg.nof_harq_ack_rvd = G_rvd, cfg.dmrs = dmrs_type2 ? dmrs_type::TYPE2 : dmrs_type::TYPE1, cfg.nof_cdm_groups_without_data = cdm_groups;
  cfg.dmrs_symbol_mask = symbol_slot_mask(14);
  for (unsigned l = 0; l != 14; ++l) {
    if ((dmrs_symbols_mask >> l) & 1U) {
      cfg.dmrs_symbol_mask.set(l);
    }
  }
  std::vector<log_likelihood_ratio> vin(n_in), vsch(n_sch), vack(G_ack), vc1(G_csi1), vc2(G_csi2);
  for (unsigned i = 0; i != n_in; ++i) {
    vin[i] = log_likelihood_ratio(in[i]);
  }
  dm->demultiplex(vsch, vack, vc1, vc2, vin, cfg);
  for (unsigned i = 0; i != n_sch; ++i) {
    sch[i] = vsch[i].to_value_type();
  }
  for (unsigned i = 0; i != G_ack; ++i) {
    ack[i] = vack[i].to_value_type();
  }
  for (unsigned i = 0; i != G_csi1; ++i) {
    csi1[i] = vc1[i].to_value_type();
  }
  for (unsigned i = 0; i != G_csi2; ++i) {
    csi2[i] = vc2[i].to_value_type();
  }
  ulsch_demultiplex::message_information mi;
  mi.nof_harq_ack_bits = O_ack, mi.nof_enc_harq_ack_bits = G_ack, mi.nof_csi_part1_bits = O_csi1, mi.nof_enc_csi_part1_bits = G_csi1;
  mi.nof_csi_part2_bits = O_csi2, mi.nof_enc_csi_part2_bits = G_csi2;
  ulsch_placeholder_list pl = dm->get_placeholders(mi, cfg);
  unsigned               k  = 0;
  // the list only exposes its entries as bit positions: bit = bits_per_symbol * (layers * re + layer) + 1
  pl.for_each(cfg.modulation, 1, [&](unsigned y_bit, unsigned) { placeholders[k++] = static_cast<uint16_t>((y_bit - 1) / mod); });
  *nof_placeholders = k;
  return 0;
}

} // extern "C"
