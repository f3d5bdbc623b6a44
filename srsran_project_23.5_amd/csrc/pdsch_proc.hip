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

namespace {
// What pdsch_processor_impl::process derives from a batch of PDUs: the encoder's transport-block records, the modulator jobs and the DM-RS jobs
// (assert_pdu :143-196, modulate :256-276, put_dmrs :278-305); cw_bytes = bytes of the codeword area (one bit per byte).
int derive_pdsch_jobs(const miphy_pdsch_pdu* pdus, uint32_t n, std::vector<miphy_pdsch_tb_desc>& tb, std::vector<miphy_pdsch_mod_job>& mj,
                      std::vector<miphy_dmrs_pdsch_job>& dj, size_t& cw_bytes)
{
  tb.resize(n), mj.resize(n), dj.resize(n);
  cw_bytes = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const miphy_pdsch_pdu& p = pdus[i];
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
    miphy_dmrs_pdsch_job& d = dj[i];
    d                       = {};
    d.slot_in_frame = p.slot_in_frame, d.reference_point_k_rb = p.ref_point_prb0 ? p.bwp_start_rb : 0, d.scrambling_id = p.dmrs_scrambling_id;
    d.amplitude = powf(10.0f, -p.ratio_pdsch_dmrs_to_sss_dB / 20.0f);
    d.dmrs_type = 1, d.n_scid = p.n_scid, d.nof_ports = 1, d.ports[0] = p.port, d.symbols_mask = p.dmrs_symbols_mask, d.grid_nof_prb = p.grid_nof_prb;
    for (int k = 0; k < 5; ++k)
      d.rb_mask[k] = p.rb_mask[k];
    d.grid_offset = p.grid_offset;
    cw_bytes += ((size_t)m.nof_bits + 15u) & ~(size_t)15u;
  }
  return MIPHY_OK;
}
} // namespace

extern "C" int miphy_pdsch_process_batch(miphy_ctx* ctx, const miphy_pdsch_pdu* pdus, uint32_t n, const uint8_t* tb_in, float* grid, void* stream)
{
  MIPHY_REQUIRE(ctx && pdus && tb_in && grid, "miphy_pdsch_process_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  hipStream_t                       s = (hipStream_t)stream;
  std::vector<miphy_pdsch_tb_desc>  tb;
  std::vector<miphy_pdsch_mod_job>  mj;
  std::vector<miphy_dmrs_pdsch_job> dj;
  size_t                            cw_bytes = 0;
  int                               rc = derive_pdsch_jobs(pdus, n, tb, mj, dj, cw_bytes);
  if (rc)
    return rc;
  void* work = nullptr; // codewords (one bit per byte) in a workspace of the context
  if ((rc = miphy_get_workspace(ctx, cw_bytes + 64, s, &work, 2)))
    return rc;
  uint8_t* d_cw = static_cast<uint8_t*>(work);
  if ((rc = miphy_pdsch_encode_batch(ctx, tb.data(), n, tb_in, d_cw, s)))
    return rc;
  if ((rc = miphy_pdsch_modulate_batch(ctx, mj.data(), 0, n, d_cw, grid, s)))
    return rc;
  return miphy_dmrs_pdsch_map_batch(ctx, dj.data(), 0, n, grid, s);
}

// ---- prepared form: PDU validation, segmentation and every descriptor upload happen once; a run is the seven launches of the chain, nothing
// staged, no host synchronisation -- for allocations that repeat slot after slot (and for batches whose descriptors exceed the staging ring,
// where the per-call form has to wait for its upload).
struct miphy_pdsch_process_plan {
  miphy_ctx*                   ctx;
  uint32_t                     n;
  miphy_pdsch_encode_prepared* enc;
  void*                        d_buf; // [modulator jobs | DM-RS jobs | codewords]
  const miphy_pdsch_mod_job*   d_mj;
  const miphy_dmrs_pdsch_job*  d_dj;
  uint8_t*                     d_cw;
};

extern "C" int miphy_pdsch_process_plan_create(miphy_ctx* ctx, const miphy_pdsch_pdu* pdus, uint32_t n, miphy_pdsch_process_plan** out)
{
  MIPHY_REQUIRE(ctx && pdus && out && n > 0, "miphy_pdsch_process_plan_create: null argument or empty batch");
  MIPHY_REQUIRE(n <= 65535, "pdsch_process: at most 65535 PDUs per plan");
  std::vector<miphy_pdsch_tb_desc>  tb;
  std::vector<miphy_pdsch_mod_job>  mj;
  std::vector<miphy_dmrs_pdsch_job> dj;
  size_t                            cw_bytes = 0;
  int                               rc = derive_pdsch_jobs(pdus, n, tb, mj, dj, cw_bytes);
  if (rc)
    return rc;
  for (uint32_t i = 0; i < n; ++i) { // what the modulator and DM-RS entry points check on host jobs (device jobs are the caller's promise)
    const miphy_pdsch_pdu& q = pdus[i];
    MIPHY_REQUIRE(q.mod == 1 || q.mod == 2 || q.mod == 4 || q.mod == 6 || q.mod == 8, "pdsch_process: PDU %u: invalid modulation order %u", i, q.mod);
    MIPHY_REQUIRE(q.grid_nof_prb >= 1 && q.grid_nof_prb <= 275 && q.bwp_start_rb + q.bwp_size_rb <= 275, "pdsch_process: PDU %u: invalid grid / BWP", i);
    MIPHY_REQUIRE(q.n_id < 1024 && q.rnti < 65536, "pdsch_process: PDU %u: invalid scrambling identifiers", i);
    MIPHY_REQUIRE(q.port < 16, "pdsch_process: PDU %u: invalid port", i);
    MIPHY_REQUIRE(!q.ref_point_prb0 || q.bwp_start_rb < q.grid_nof_prb, "pdsch_process: PDU %u: reference point outside the grid", i);
  }
  auto* p = new miphy_pdsch_process_plan();
  p->ctx = ctx, p->n = n, p->enc = nullptr, p->d_buf = nullptr;
  if ((rc = miphy_pdsch_encode_prepare(ctx, tb.data(), n, &p->enc))) {
    delete p;
    return rc;
  }
  const size_t b0 = ((size_t)n * sizeof(miphy_pdsch_mod_job) + 15) & ~(size_t)15, b1 = ((size_t)n * sizeof(miphy_dmrs_pdsch_job) + 15) & ~(size_t)15;
  hipError_t   e  = hipMalloc(&p->d_buf, b0 + b1 + cw_bytes + 64);
  if (e == hipSuccess)
    e = hipMemcpy(p->d_buf, mj.data(), (size_t)n * sizeof(miphy_pdsch_mod_job), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = hipMemcpy((uint8_t*)p->d_buf + b0, dj.data(), (size_t)n * sizeof(miphy_dmrs_pdsch_job), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    miphy_set_error("miphy_pdsch_process_plan_create: %s", hipGetErrorString(e));
    if (p->d_buf)
      (void)hipFree(p->d_buf);
    miphy_pdsch_encode_prepared_destroy(p->enc);
    delete p;
    return MIPHY_EHIP;
  }
  p->d_mj = reinterpret_cast<const miphy_pdsch_mod_job*>(p->d_buf);
  p->d_dj = reinterpret_cast<const miphy_dmrs_pdsch_job*>((uint8_t*)p->d_buf + b0);
  p->d_cw = (uint8_t*)p->d_buf + b0 + b1;
  *out    = p;
  return MIPHY_OK;
}

extern "C" int miphy_pdsch_process_plan_run(miphy_pdsch_process_plan* p, const uint8_t* tb_in, float* grid, void* stream)
{
  MIPHY_REQUIRE(p && tb_in && grid, "miphy_pdsch_process_plan_run: null argument");
  hipStream_t s = (hipStream_t)stream;
  int         rc;
  if ((rc = miphy_pdsch_encode_prepared_run(p->enc, tb_in, p->d_cw, s)))
    return rc;
  if ((rc = miphy_pdsch_modulate_batch(p->ctx, p->d_mj, 1, p->n, p->d_cw, grid, s)))
    return rc;
  return miphy_dmrs_pdsch_map_batch(p->ctx, p->d_dj, 1, p->n, grid, s);
}

extern "C" void miphy_pdsch_process_plan_destroy(miphy_pdsch_process_plan* p)
{
  if (!p)
    return;
  if (p->d_buf)
    (void)hipFree(p->d_buf);
  miphy_pdsch_encode_prepared_destroy(p->enc);
  delete p;
}
