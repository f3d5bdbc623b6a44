// PDSCH processor: transport blocks to resource-grid REs in one call, the codewords stay on the device.
// Behaviour contract: lib/phy/upper/channel_processors/pdsch_processor_impl.cpp:110-305 (process = encode + modulate + put_dmrs,
// with the restrictions of assert_pdu :143-196). Host-side composition of miphy_pdsch_encode_batch, miphy_pdsch_modulate_batch and
// miphy_dmrs_pdsch_map_batch with the parameters the reference derives from the PDU.
#include "miphy_ext.h"
#include <cmath>
#include <vector>

namespace {
// pdsch_processor_impl::modulate (:256-276): the modulator configuration of a PDU.
void mod_job_of(const miphy_pdsch_pdu& p, miphy_pdsch_mod_job& m)
{
  m      = {};
  m.rnti = p.rnti, m.n_id = p.n_id, m.mod = p.mod, m.port = p.port, m.start_symbol = p.start_symbol, m.nof_symbols = p.nof_symbols;
  m.scaling   = powf(10.0f, -p.ratio_pdsch_data_to_sss_dB / 20.0f); // convert_dB_to_amplitude(-ratio)
  m.dmrs_type = 1, m.nof_cdm_groups_without_data = p.nof_cdm_groups_without_data, m.nof_reserved = p.nof_reserved;
  m.dmrs_symbols_mask = p.dmrs_symbols_mask, m.grid_nof_prb = p.grid_nof_prb, m.bwp_start_rb = p.bwp_start_rb, m.bwp_size_rb = p.bwp_size_rb;
  for (int k = 0; k < 5; ++k)
    m.rb_mask[k] = p.rb_mask[k];
  for (int k = 0; k < 4; ++k)
    m.reserved[k] = p.reserved[k];
  m.grid_offset = p.grid_offset;
}
} // namespace

extern "C" uint32_t miphy_pdsch_pdu_nof_re(const miphy_pdsch_pdu* pdu)
{
  if (!pdu || pdu->nof_reserved > 4)
    return 0;
  miphy_pdsch_mod_job m;
  mod_job_of(*pdu, m);
  return miphy_pdsch_mod_nof_re(&m);
}

extern "C" int miphy_pdsch_process_batch(miphy_ctx* ctx, const miphy_pdsch_pdu* pdus, uint32_t n, const uint8_t* tb_in, float* grid, void* stream)
{
  MIPHY_REQUIRE(ctx && pdus && tb_in && grid, "miphy_pdsch_process_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  hipStream_t                       s = (hipStream_t)stream;
  std::vector<miphy_pdsch_tb_desc>  tb(n);
  std::vector<miphy_pdsch_mod_job>  mj(n);
  std::vector<miphy_dmrs_pdsch_job> dj(n);
  size_t                            cw_bytes = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const miphy_pdsch_pdu& p = pdus[i];
    // assert_pdu (:143-196)
    MIPHY_REQUIRE(p.dmrs_symbols_mask != 0 && p.dmrs_symbols_mask < (1u << 14), "pdsch_process: PDU %u: invalid DM-RS symbol mask", i);
    MIPHY_REQUIRE(p.nof_symbols >= 1 && p.start_symbol + p.nof_symbols <= 14, "pdsch_process: PDU %u: the time allocation exceeds the slot", i);
    const unsigned first_dmrs = __builtin_ctz(p.dmrs_symbols_mask), last_dmrs = 31 - __builtin_clz((unsigned)p.dmrs_symbols_mask);
    MIPHY_REQUIRE(first_dmrs >= p.start_symbol && last_dmrs < (unsigned)p.start_symbol + p.nof_symbols,
                  "pdsch_process: PDU %u: DM-RS symbols outside the time allocation", i);
    MIPHY_REQUIRE(p.nof_cdm_groups_without_data >= 1 && p.nof_cdm_groups_without_data <= 2, "pdsch_process: PDU %u: invalid number of CDM groups without data", i);
    MIPHY_REQUIRE(p.tbs_lbrm_bytes > 0 && p.tbs_lbrm_bytes <= 66 * 384 / 8, "pdsch_process: PDU %u: invalid LBRM size (%u bytes)", i, p.tbs_lbrm_bytes);
    MIPHY_REQUIRE(p.nof_reserved <= 4, "pdsch_process: PDU %u: at most 4 reserved RE patterns", i);
    MIPHY_REQUIRE(p.bg == 1 || p.bg == 2, "pdsch_process: PDU %u: invalid base graph", i);
    miphy_pdsch_mod_job& m = mj[i];
    mod_job_of(p, m);
    const uint32_t nre = miphy_pdsch_mod_nof_re(&m);
    MIPHY_REQUIRE(nre > 0, "pdsch_process: PDU %u: invalid or empty allocation", i);
    m.nof_bits  = nre * p.mod;
    m.cw_offset = cw_bytes;
    miphy_pdsch_tb_desc& t = tb[i];
    t                      = {};
    t.bg = p.bg, t.rv = p.rv, t.mod = p.mod, t.nof_layers = 1, t.Nref = p.tbs_lbrm_bytes * 8, t.nof_ch_symbols = nre, t.tb_bytes = p.tb_bytes;
    t.tb_offset = p.tb_offset, t.codeword_offset = cw_bytes;
    miphy_dmrs_pdsch_job& d = dj[i]; // put_dmrs (:278-305)
    d                       = {};
    d.slot_in_frame = p.slot_in_frame, d.reference_point_k_rb = p.ref_point_prb0 ? p.bwp_start_rb : 0, d.scrambling_id = p.dmrs_scrambling_id;
    d.amplitude = powf(10.0f, -p.ratio_pdsch_dmrs_to_sss_dB / 20.0f);
    d.dmrs_type = 1, d.n_scid = p.n_scid, d.nof_ports = 1, d.ports[0] = p.port, d.symbols_mask = p.dmrs_symbols_mask, d.grid_nof_prb = p.grid_nof_prb;
    for (int k = 0; k < 5; ++k)
      d.rb_mask[k] = p.rb_mask[k];
    d.grid_offset = p.grid_offset;
    cw_bytes += ((size_t)m.nof_bits + 15u) & ~(size_t)15u;
  }
  void* work = nullptr; // codewords (one bit per byte) in a workspace of the context
  int   rc   = miphy_get_workspace(ctx, cw_bytes + 64, s, &work, 2);
  if (rc)
    return rc;
  uint8_t* d_cw = static_cast<uint8_t*>(work);
  if ((rc = miphy_pdsch_encode_batch(ctx, tb.data(), n, tb_in, d_cw, s)))
    return rc;
  if ((rc = miphy_pdsch_modulate_batch(ctx, mj.data(), 0, n, d_cw, grid, s)))
    return rc;
  return miphy_dmrs_pdsch_map_batch(ctx, dj.data(), 0, n, grid, s);
}
