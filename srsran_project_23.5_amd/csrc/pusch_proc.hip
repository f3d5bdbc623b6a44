// PUSCH processor: channel estimation, demodulation and transport-block decoding of a batch of PUSCH PDUs in one
// call, everything between the resource grid and the transport block stays on the device.
// Behaviour contract: lib/phy/upper/channel_processors/pusch_processor_impl.cpp:108-330 for PDUs without UCI (the reference
// itself restricts the rest: DM-RS type 1, two CDM groups without data, one layer: pusch_processor_impl.cpp:96-104,331-345).
// This is host-side composition of miphy_dmrs_pusch_estimate_batch, miphy_pusch_demodulate_batch and miphy_pusch_decode_batch
// with the parameters the reference derives (DM-RS scaling from the SCH-to-DM-RS power ratio, codeword length from the allocation).
#include "miphy_ext.h"
#include <cmath>
#include <vector>

namespace {
// EVM of every PDU from the per-symbol sums of the demodulator: sqrt(sum / data REs) (evm_calculator_generic_impl.cpp:45-46).
__global__ void evm_finish_kernel(const float* __restrict__ sums, const uint32_t* __restrict__ nof_re, float* __restrict__ evm, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  float acc = 0.f;
  for (int l = 0; l < 14; ++l)
    acc += sums[(size_t)i * 14 + l];
  evm[i] = nof_re[i] ? sqrtf(acc / (float)nof_re[i]) : 0.f;
}
__global__ void zero_results_kernel(miphy_pusch_result* __restrict__ r, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    r[i] = miphy_pusch_result{};
}
} // namespace

extern "C" int miphy_pusch_process_batch(miphy_ctx* ctx, const miphy_pusch_pdu* pdus, uint32_t n, const float* grid, int8_t* harq_softbits,
                                         uint8_t* harq_msgs, uint8_t* harq_crc_ok, uint8_t* tb_out, miphy_pusch_result* results, float* scalars_out,
                                         void* stream)
{
  return miphy_pusch_process_batch_ex(ctx, pdus, nullptr, n, grid, harq_softbits, harq_msgs, harq_crc_ok, tb_out, results, scalars_out, nullptr, nullptr,
                                      stream);
}

extern "C" int miphy_pusch_process_batch_ex(miphy_ctx* ctx, const miphy_pusch_pdu* pdus, const miphy_pusch_uci* uci, uint32_t n, const float* grid,
                                            int8_t* harq_softbits, uint8_t* harq_msgs, uint8_t* harq_crc_ok, uint8_t* tb_out, miphy_pusch_result* results,
                                            float* scalars_out, int8_t* uci_llr_out, float* evm_out, void* stream)
{
  MIPHY_REQUIRE(ctx && pdus && grid && harq_softbits && harq_msgs && harq_crc_ok && tb_out && results && scalars_out,
                "miphy_pusch_process_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  hipStream_t                        s = (hipStream_t)stream;
  std::vector<miphy_pusch_chest_job> cj(n);
  std::vector<miphy_pusch_demod_job> dj(n);
  std::vector<miphy_pusch_tb_desc>   tb;
  std::vector<miphy_ulsch_demux_job> xj;        // PDUs with multiplexed UCI
  std::vector<uint16_t>              ph;        // their repetition placeholders, back to back
  std::vector<uint32_t>              nof_re(n); // data REs per PDU (EVM)
  std::vector<uint32_t>              no_tb;     // PDUs without a transport block
  std::vector<uint32_t>              tb_index;  // PDU of every transport-block descriptor
  size_t                             ce_elems = 0, llr_bytes = 0, sch_bytes = 0;
  bool                               any_uci  = false;
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
    d.evm_offset = (uint64_t)i * 14;
    nof_re[i]    = d.nof_llr / p.mod;
    uint64_t sch_llr_offset = llr_bytes; // where the decoder reads the UL-SCH soft bits: the codeword itself unless UCI is multiplexed
    uint32_t nof_sch_llr    = d.nof_llr;
    const miphy_pusch_uci* u = uci ? &uci[i] : nullptr;
    const bool has_uci = u && (u->nof_harq_ack_bits || u->nof_csi_part1_bits || u->nof_csi_part2_bits);
    if (has_uci) {
      MIPHY_REQUIRE(uci_llr_out, "pusch_process: PDU %u carries UCI but uci_llr_out is NULL", i);
      any_uci = true;
      miphy_ulsch_demux_job x = {};
      uint32_t              nprb = 0;
      for (unsigned r = 0; r < p.grid_nof_prb; ++r)
        nprb += (uint32_t)((p.rb_mask[r >> 6] >> (r & 63)) & 1ull);
      x.mod = p.mod, x.nof_layers = 1, x.start_symbol = p.start_symbol, x.nof_symbols = p.nof_symbols, x.dmrs_type = 1, x.nof_cdm_groups_without_data = 2;
      x.dmrs_symbols_mask = p.dmrs_symbols_mask, x.nof_prb = (uint16_t)nprb, x.nof_harq_ack_rvd = u->nof_harq_ack_rvd;
      x.nof_enc_harq_ack_bits = u->nof_enc_harq_ack_bits, x.nof_enc_csi_part1_bits = u->nof_enc_csi_part1_bits;
      x.nof_enc_csi_part2_bits = u->nof_enc_csi_part2_bits;
      x.nof_harq_ack_bits = u->nof_harq_ack_bits, x.nof_csi_part1_bits = u->nof_csi_part1_bits, x.nof_csi_part2_bits = u->nof_csi_part2_bits;
      uint32_t nin = 0;
      int      rc  = miphy_ulsch_demux_sizes(&x, &nin, &nof_sch_llr);
      if (rc)
        return rc;
      MIPHY_REQUIRE(nin == d.nof_llr, "pusch_process: PDU %u: the UL-SCH multiplexing covers %u soft bits, the allocation holds %u", i, nin, d.nof_llr);
      x.in_offset = llr_bytes, x.sch_offset = sch_bytes, x.harq_ack_offset = u->harq_ack_offset, x.csi_part1_offset = u->csi_part1_offset;
      x.csi_part2_offset = u->csi_part2_offset;
      xj.push_back(x);
      // repetition placeholders of the descrambler
      uint32_t nph = 0;
      if ((rc = miphy_ulsch_placeholders(&x, nullptr, 0, &nph)))
        return rc;
      d.placeholders_offset = (uint32_t)ph.size(), d.nof_placeholders = nph;
      ph.resize(ph.size() + nph);
      if (nph && (rc = miphy_ulsch_placeholders(&x, ph.data() + d.placeholders_offset, nph, &nph)))
        return rc;
      sch_llr_offset = sch_bytes; // relative to the SCH region, rebased below
      sch_bytes += (nof_sch_llr + 15u) & ~15u;
    }
    // include/miphy.h: has_codeword = 0 means the PDU carries no transport block -- whatever its UCI fields say. A PDU with neither is
    // not a PUSCH transmission.
    MIPHY_REQUIRE(!u || u->has_codeword || has_uci, "pusch_process: PDU %u: neither a codeword nor UCI", i);
    if (!u || u->has_codeword) {
      miphy_pusch_tb_desc t = {};
      t.bg = p.bg, t.rv = p.rv, t.mod = p.mod, t.nof_layers = 1, t.new_data = p.new_data, t.use_early_stop = p.use_early_stop;
      t.nof_ldpc_iterations = p.nof_ldpc_iterations, t.Nref = p.Nref, t.nof_ch_symbols = nof_sch_llr / p.mod, t.tb_bytes = p.tb_bytes;
      t.harq_cb_index = p.harq_cb_index, t.tb_offset = p.tb_offset;
      t.llr_offset = has_uci ? (uint64_t)1 << 63 | sch_llr_offset : sch_llr_offset; // bit 63: offset inside the SCH region (resolved below)
      tb.push_back(t);
      tb_index.push_back(i);
    } else {
      no_tb.push_back(i);
    }
    ce_elems += (size_t)p.nof_rx_ports * nsc;
    llr_bytes += (d.nof_llr + 15u) & ~15u;
  }
  for (auto& t : tb) // SCH region behind the codeword LLRs
    if (t.llr_offset >> 63)
      t.llr_offset = (t.llr_offset & ~((uint64_t)1 << 63)) + llr_bytes;
  for (auto& x : xj)
    x.sch_offset += llr_bytes;
  // [channel estimates cf_t | codeword LLRs | UL-SCH LLRs of the PDUs with UCI | EVM sums | placeholders | data REs] in a workspace of the
  // context; the estimator scalars go straight to the caller's array.
  const size_t evm_bytes = evm_out ? (size_t)n * 14 * 4 : 0, ph_bytes = (ph.size() * 2 + 15) & ~(size_t)15, nre_bytes = evm_out ? (size_t)n * 4 : 0;
  void*        work      = nullptr;
  int          rc        = miphy_get_workspace(ctx, ce_elems * 8 + llr_bytes + sch_bytes + evm_bytes + ph_bytes + nre_bytes + 256, s, &work, 1);
  if (rc)
    return rc;
  float*    d_ce  = static_cast<float*>(work);
  int8_t*   d_llr = reinterpret_cast<int8_t*>(work) + ce_elems * 8;
  float*    d_evm = reinterpret_cast<float*>(d_llr + llr_bytes + sch_bytes);
  uint16_t* d_ph  = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(d_evm) + evm_bytes);
  uint32_t* d_nre = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(d_ph) + ph_bytes);
  if (!ph.empty() && (rc = miphy_upload(ctx, d_ph, ph.data(), ph.size() * 2, s)))
    return rc;
  if (evm_out && (rc = miphy_upload(ctx, d_nre, nof_re.data(), (size_t)n * 4, s)))
    return rc;
  if ((rc = miphy_dmrs_pusch_estimate_batch(ctx, cj.data(), 0, n, grid, d_ce, scalars_out, s)))
    return rc;
  if ((rc = miphy_pusch_demodulate_batch_ex(ctx, dj.data(), 0, n, grid, d_ce, scalars_out, d_llr, ph.empty() ? nullptr : d_ph, evm_out ? d_evm : nullptr, s)))
    return rc;
  if (evm_out) {
    hipLaunchKernelGGL(evm_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_evm, d_nre, evm_out, n);
    MIPHY_HIP_CHECK(hipGetLastError());
  }
  if (any_uci && (rc = miphy_ulsch_demultiplex_batch(ctx, xj.data(), (uint32_t)xj.size(), d_llr, d_llr, uci_llr_out, uci_llr_out, uci_llr_out, s)))
    return rc;
  if (tb.size() == n) // the common case: every PDU has a transport block, results in PDU order
    return miphy_pusch_decode_batch(ctx, tb.data(), n, d_llr, harq_softbits, harq_msgs, harq_crc_ok, tb_out, results, s);
  // Some PDUs carry UCI only: decode the others into a compact result array, then put the records in PDU order.
  hipLaunchKernelGGL(zero_results_kernel, dim3((n + 255) / 256), dim3(256), 0, s, results, n);
  MIPHY_HIP_CHECK(hipGetLastError());
  if (tb.empty())
    return MIPHY_OK;
  void* rws = nullptr;
  if ((rc = miphy_get_workspace(ctx, tb.size() * sizeof(miphy_pusch_result), s, &rws, 2)))
    return rc;
  if ((rc = miphy_pusch_decode_batch(ctx, tb.data(), (uint32_t)tb.size(), d_llr, harq_softbits, harq_msgs, harq_crc_ok, tb_out, (miphy_pusch_result*)rws, s)))
    return rc;
  for (size_t k = 0; k < tb.size(); ++k)
    MIPHY_HIP_CHECK(hipMemcpyAsync(results + tb_index[k], (miphy_pusch_result*)rws + k, sizeof(miphy_pusch_result), hipMemcpyDeviceToDevice, s));
  return MIPHY_OK;
}
