// PUSCH processor: channel estimation, demodulation and transport-block decoding of a batch of PUSCH PDUs in one
// call, everything between the resource grid and the transport block stays on the device.
// Behaviour contract: lib/phy/upper/channel_processors/pusch_processor_impl.cpp:108-330 for PDUs without UCI (the reference
// itself restricts the rest: DM-RS type 1, two CDM groups without data, one layer: pusch_processor_impl.cpp:96-104,331-345).
// This is host-side composition of miphy_dmrs_pusch_estimate_batch, miphy_pusch_demodulate_batch and miphy_pusch_decode_batch
// with the parameters the reference derives (DM-RS scaling from the SCH-to-DM-RS power ratio, codeword length from the allocation).
#include "miphy_ext.h"
#include <cmath>
#include <vector>

extern "C" int miphy_pusch_process_batch(miphy_ctx* ctx, const miphy_pusch_pdu* pdus, uint32_t n, const float* grid, int8_t* harq_softbits,
                                         uint8_t* harq_msgs, uint8_t* harq_crc_ok, uint8_t* tb_out, miphy_pusch_result* results, float* scalars_out,
                                         void* stream)
{
  MIPHY_REQUIRE(ctx && pdus && grid && harq_softbits && harq_msgs && harq_crc_ok && tb_out && results && scalars_out,
                "miphy_pusch_process_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  hipStream_t                        s = (hipStream_t)stream;
  std::vector<miphy_pusch_chest_job> cj(n);
  std::vector<miphy_pusch_demod_job> dj(n);
  std::vector<miphy_pusch_tb_desc>   tb(n);
  size_t                             ce_elems = 0, llr_bytes = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const miphy_pusch_pdu& p = pdus[i];
    MIPHY_REQUIRE(p.nof_rx_ports >= 1 && p.nof_rx_ports <= 4, "pusch_process: PDU %u: invalid number of receive ports", i);
    MIPHY_REQUIRE(p.nof_symbols >= 1 && p.start_symbol + p.nof_symbols <= 14, "pusch_process: PDU %u: invalid time allocation", i);
    MIPHY_REQUIRE(p.dmrs_symbols_mask != 0, "pusch_process: PDU %u: no DM-RS symbol", i);
    MIPHY_REQUIRE(p.grid_nof_prb >= 1 && p.grid_nof_prb <= 275, "pusch_process: PDU %u: invalid grid width", i);
    const uint32_t nsc = p.grid_nof_prb * 12u, nsym_ce = (uint32_t)p.start_symbol + p.nof_symbols;
    miphy_pusch_chest_job& c = cj[i];
    c                        = {};
    c.numerology = p.numerology, c.slot_in_frame = p.slot_in_frame, c.scrambling_id = p.dmrs_scrambling_id;
    // pusch_processor_impl.cpp:150: amplitude of the DM-RS relative to the data, two CDM groups without data -> +3 dB
    c.scaling = powf(10.0f, 3.0f / 20.0f);
    c.n_scid = p.n_scid, c.nof_tx_layers = 1, c.nof_rx_ports = p.nof_rx_ports, c.first_symbol = p.start_symbol, c.nof_symbols = p.nof_symbols;
    c.symbols_mask = p.dmrs_symbols_mask, c.grid_nof_prb = p.grid_nof_prb;
    c.ce_compact   = 1; // the estimate only feeds the demodulator below: one row per port instead of a copy per symbol
    miphy_pusch_demod_job& d = dj[i];
    d                        = {};
    d.rnti = p.rnti, d.n_id = p.n_id, d.mod = p.mod, d.nof_rx_ports = p.nof_rx_ports, d.start_symbol = p.start_symbol, d.nof_symbols = p.nof_symbols;
    d.dmrs_type = 1, d.nof_cdm_groups_without_data = 2, d.ce_nof_symbols = (uint8_t)nsym_ce, d.ce_compact = 1;
    d.dmrs_symbols_mask = p.dmrs_symbols_mask, d.grid_nof_prb = p.grid_nof_prb;
    for (int k = 0; k < 4; ++k)
      c.rx_ports[k] = p.rx_ports[k], d.rx_ports[k] = p.rx_ports[k];
    for (int k = 0; k < 5; ++k)
      c.rb_mask[k] = p.rb_mask[k], d.rb_mask[k] = p.rb_mask[k];
    c.grid_offset = d.grid_offset = p.grid_offset;
    c.ce_offset = d.ce_offset = ce_elems;
    c.scalars_offset = d.scalars_offset = (uint64_t)i * 20; // [4 ports][5] floats per PDU, layer 0
    d.llr_offset                       = llr_bytes;
    d.nof_llr                          = miphy_pusch_demod_nof_llr(&d);
    MIPHY_REQUIRE(d.nof_llr > 0, "pusch_process: PDU %u: empty allocation", i);
    miphy_pusch_tb_desc& t = tb[i];
    t                      = {};
    t.bg = p.bg, t.rv = p.rv, t.mod = p.mod, t.nof_layers = 1, t.new_data = p.new_data, t.use_early_stop = p.use_early_stop;
    t.nof_ldpc_iterations = p.nof_ldpc_iterations, t.Nref = p.Nref, t.nof_ch_symbols = d.nof_llr / p.mod, t.tb_bytes = p.tb_bytes;
    t.harq_cb_index = p.harq_cb_index, t.llr_offset = llr_bytes, t.tb_offset = p.tb_offset;
    ce_elems += (size_t)p.nof_rx_ports * nsc;
    llr_bytes += (d.nof_llr + 15u) & ~15u;
  }
  // [channel estimates cf_t | LLRs] in a workspace of the context; the estimator scalars go straight to the caller's array.
  void* work = nullptr;
  int   rc   = miphy_get_workspace(ctx, ce_elems * 8 + llr_bytes + 64, s, &work, 1);
  if (rc)
    return rc;
  float*  d_ce  = static_cast<float*>(work);
  int8_t* d_llr = reinterpret_cast<int8_t*>(work) + ce_elems * 8;
  if ((rc = miphy_dmrs_pusch_estimate_batch(ctx, cj.data(), 0, n, grid, d_ce, scalars_out, s)))
    return rc;
  if ((rc = miphy_pusch_demodulate_batch(ctx, dj.data(), 0, n, grid, d_ce, scalars_out, d_llr, s)))
    return rc;
  return miphy_pusch_decode_batch(ctx, tb.data(), n, d_llr, harq_softbits, harq_msgs, harq_crc_ok, tb_out, results, s);
}
