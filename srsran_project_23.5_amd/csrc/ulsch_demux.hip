// UL-SCH demultiplexer: splits the codeword LLRs of a PUSCH transmission with multiplexed UCI into the UL-SCH data stream and the
// HARQ-ACK / CSI part 1 / CSI part 2 streams. Behaviour contract: ulsch_demultiplex_impl.cpp:74-453 (see ulsch_device.h for how
// the serial scan of the reference becomes a closed form per resource element).
#include "miphy_ext.h"
#include "ulsch_device.h"
#include <vector>

// The scan over the OFDM symbols of ulsch_demultiplex_impl.cpp:126-201: per symbol the stride and count of every field and the
// stream positions at which the symbol starts. Returns MIPHY_EINVAL where the reference would assert.
int ulsch_plan_symbols(const miphy_ulsch_demux_job& j, ulsch_plan& out)
{
  MIPHY_REQUIRE(j.mod == 1 || j.mod == 2 || j.mod == 4 || j.mod == 6 || j.mod == 8, "ulsch_demultiplex: invalid modulation order %u", j.mod);
  MIPHY_REQUIRE(j.nof_layers >= 1 && j.nof_layers <= 4, "ulsch_demultiplex: invalid number of layers");
  MIPHY_REQUIRE(j.nof_symbols >= 1 && j.start_symbol + j.nof_symbols <= 14, "ulsch_demultiplex: invalid time allocation");
  MIPHY_REQUIRE(j.nof_prb >= 1 && j.nof_prb <= 275, "ulsch_demultiplex: invalid number of PRBs");
  MIPHY_REQUIRE(j.dmrs_type == 1 || j.dmrs_type == 2, "ulsch_demultiplex: invalid DM-RS type");
  MIPHY_REQUIRE(j.nof_cdm_groups_without_data >= 1 && j.nof_cdm_groups_without_data <= (j.dmrs_type == 1 ? 2 : 3),
                "ulsch_demultiplex: invalid number of CDM groups without data");
  const uint32_t mask = j.dmrs_symbols_mask & 0x3fffu;
  MIPHY_REQUIRE(mask != 0, "ulsch_demultiplex: no DM-RS symbol");
  out               = {};
  out.nof_symbols   = j.nof_symbols;
  out.bits_per_re   = (uint32_t)j.mod * j.nof_layers;
  const uint32_t bpr = out.bits_per_re;
  // l1: first symbol without DM-RS after the first DM-RS symbol; l1_csi: first symbol without DM-RS (:31-53)
  uint32_t first_dmrs = 0;
  while (!((mask >> first_dmrs) & 1u))
    ++first_dmrs;
  uint32_t l1 = first_dmrs;
  while (l1 < 14 && ((mask >> l1) & 1u))
    ++l1;
  MIPHY_REQUIRE(l1 < 14, "ulsch_demultiplex: no symbol without DM-RS behind the first DM-RS symbol");
  uint32_t l1_csi = 0;
  while ((mask >> l1_csi) & 1u)
    ++l1_csi;
  const uint32_t re_dmrs_prb = j.nof_cdm_groups_without_data * (j.dmrs_type == 1 ? 6u : 4u);
  const uint32_t nof_re_dmrs = (12u - re_dmrs_prb) * j.nof_prb;
  const uint32_t G_rvd = j.nof_harq_ack_rvd, G_ack = j.nof_enc_harq_ack_bits, G_c1 = j.nof_enc_csi_part1_bits, G_c2 = j.nof_enc_csi_part2_bits;
  uint32_t m_rvd = 0, m_ack = 0, m_c1 = 0, m_c2 = 0;
  uint32_t off_in = 0, off_sch = 0, off_ack = 0, off_c1 = 0, off_c2 = 0;
  for (uint32_t k = 0; k < j.nof_symbols; ++k) {
    const uint32_t     l = j.start_symbol + k;
    ulsch_symbol_plan& p = out.sym[k];
    p.off_in = off_in, p.off_sch = off_sch, p.off_ack = off_ack, p.off_csi1 = off_c1, p.off_csi2 = off_c2;
    p.flags = (G_rvd != 0) ? 1u : 0u;
    if ((mask >> l) & 1u) { // only UL-SCH data next to the DM-RS
      p.nof_re = (uint16_t)nof_re_dmrs;
      off_in += nof_re_dmrs, off_sch += nof_re_dmrs;
      continue;
    }
    const uint32_t M = j.nof_prb * 12u; // no PT-RS
    uint32_t       M_uci = M, M_rvd = 0;
    uint32_t       ack_d = 0, ack_n = 0, rvd_d = 0, rvd_n = 0, c1_d = 0, c1_n = 0, c2_d = 0, c2_n = 0;
    auto           spread = [bpr](uint32_t remaining_bits, uint32_t avail_re, uint32_t& d, uint32_t& n) {
      // all of the available elements, or every d-th of them when fewer bits are left than they can carry
      d = 1, n = avail_re;
      if (remaining_bits < avail_re * bpr) {
        d = (avail_re * bpr) / remaining_bits;
        n = ulsch_ceil_div(remaining_bits, bpr);
      }
    };
    if (l >= l1) {
      const uint32_t rvd_left = G_rvd - m_rvd, ack_left = G_ack - m_ack;
      if (G_rvd != 0 && rvd_left != 0) {
        spread(rvd_left, M_uci, rvd_d, rvd_n);
        M_rvd = rvd_n;
        if (ack_left != 0)
          spread(ack_left, M_rvd, ack_d, ack_n);
      } else if (ack_left != 0) {
        spread(ack_left, M_uci, ack_d, ack_n);
        M_uci -= ack_n;
      }
    }
    if (l >= l1_csi) {
      const uint32_t c1_left = G_c1 - m_c1, c2_left = G_c2 - m_c2;
      if (M_uci > M_rvd && c1_left != 0) {
        spread(c1_left, M_uci - M_rvd, c1_d, c1_n);
        M_uci -= c1_n;
      }
      if (M_uci > 0 && c2_left != 0) {
        spread(c2_left, M_uci, c2_d, c2_n);
        M_uci -= c2_n;
      }
    }
    m_rvd += rvd_n * bpr, m_ack += ack_n * bpr, m_c1 += c1_n * bpr, m_c2 += c2_n * bpr;
    // the symbol plan stores strides and counts in 16 bits: a stride beyond that (a field of fewer bits than one element holds, on a
    // wide allocation) must be refused, not truncated
    MIPHY_REQUIRE(M <= 0xffffu && rvd_d <= 0xffffu && ack_d <= 0xffffu && c1_d <= 0xffffu && c2_d <= 0xffffu,
                  "ulsch_demultiplex: symbol %u: stride out of range (%u reserved, %u HARQ-ACK, %u CSI-1, %u CSI-2, %u elements): field lengths must be multiples of the bits per element",
                  l, rvd_d, ack_d, c1_d, c2_d, M);
    p.nof_re = (uint16_t)M;
    p.rvd_d = (uint16_t)rvd_d, p.rvd_cnt = (uint16_t)rvd_n, p.ack_d = (uint16_t)ack_d, p.ack_cnt = (uint16_t)ack_n;
    p.csi1_d = (uint16_t)c1_d, p.csi1_cnt = (uint16_t)c1_n, p.csi2_d = (uint16_t)c2_d, p.csi2_cnt = (uint16_t)c2_n;
    // stream lengths of the symbol: every subcarrier consumes one input element; HARQ-ACK on reserved elements does not take the
    // element away from SCH / CSI part 2 (they receive zeros there)
    const bool     on_rvd = G_rvd != 0;
    const uint32_t sch_n  = M - (on_rvd ? 0u : ack_n) - c1_n - c2_n;
    off_in += M, off_sch += sch_n, off_ack += ack_n, off_c1 += c1_n, off_c2 += c2_n;
  }
  MIPHY_REQUIRE(m_rvd == G_rvd && m_ack == G_ack && m_c1 == G_c1 && m_c2 == G_c2,
                "ulsch_demultiplex: the UCI fields do not fit the allocation (%u/%u reserved, %u/%u HARQ-ACK, %u/%u CSI-1, %u/%u CSI-2 bits placed)", m_rvd, G_rvd,
                m_ack, G_ack, m_c1, G_c1, m_c2, G_c2);
  out.nof_in_re = off_in, out.nof_sch_re = off_sch, out.nof_ack_re = off_ack, out.nof_csi1_re = off_c1, out.nof_csi2_re = off_c2;
  out.one_bit_fields = (j.mod >= 2) ? ((j.nof_harq_ack_bits == 1 ? 1u : 0u) | (j.nof_csi_part1_bits == 1 ? 2u : 0u) | (j.nof_csi_part2_bits == 1 ? 4u : 0u)) : 0u;
  return MIPHY_OK;
}

namespace {

struct demux_dev_job {
  uint64_t in_offset, sch_offset, ack_offset, csi1_offset, csi2_offset;
};

__global__ void __launch_bounds__(256) ulsch_demux_kernel(const ulsch_plan* __restrict__ plans, const demux_dev_job* __restrict__ jobs,
                                                          const int8_t* __restrict__ in, int8_t* __restrict__ sch, int8_t* __restrict__ ack,
                                                          int8_t* __restrict__ csi1, int8_t* __restrict__ csi2)
{
  const ulsch_plan&  pl = plans[blockIdx.x];
  const unsigned     k  = blockIdx.y;
  if (k >= pl.nof_symbols)
    return;
  const ulsch_symbol_plan p   = pl.sym[k];
  const demux_dev_job     j   = jobs[blockIdx.x];
  const uint32_t          bpr = pl.bits_per_re;
  const int8_t*           src = in + j.in_offset + (size_t)p.off_in * bpr;
  int8_t*                 dst[4] = {sch + j.sch_offset + (size_t)p.off_sch * bpr, ack + j.ack_offset + (size_t)p.off_ack * bpr,
                                    csi1 + j.csi1_offset + (size_t)p.off_csi1 * bpr, csi2 + j.csi2_offset + (size_t)p.off_csi2 * bpr};
  for (uint32_t i = threadIdx.x; i < p.nof_re; i += blockDim.x) {
    const ulsch_re_class c = ulsch_classify(p, i);
    int8_t*              o = (c.cls == ULSCH_SCH ? dst[0] : c.cls == ULSCH_ACK ? dst[1] : c.cls == ULSCH_CSI1 ? dst[2] : dst[3]) + (size_t)c.rank * bpr;
    for (uint32_t b = 0; b < bpr; ++b)
      o[b] = src[(size_t)i * bpr + b];
    if (c.punctured) {
      int8_t* z = (c.zero_cls == ULSCH_CSI2 ? dst[3] : dst[0]) + (size_t)c.zero_rank * bpr;
      for (uint32_t b = 0; b < bpr; ++b)
        z[b] = 0;
    }
  }
}

} // namespace

extern "C" int miphy_ulsch_demux_sizes(const miphy_ulsch_demux_job* job, uint32_t* nof_in_llr, uint32_t* nof_sch_llr)
{
  MIPHY_REQUIRE(job && nof_in_llr && nof_sch_llr, "miphy_ulsch_demux_sizes: null argument");
  ulsch_plan pl;
  int        rc = ulsch_plan_symbols(*job, pl);
  if (rc)
    return rc;
  *nof_in_llr  = pl.nof_in_re * pl.bits_per_re;
  *nof_sch_llr = pl.nof_sch_re * pl.bits_per_re;
  return MIPHY_OK;
}

extern "C" int miphy_ulsch_placeholders(const miphy_ulsch_demux_job* job, uint16_t* re_indices, uint32_t cap, uint32_t* n)
{
  MIPHY_REQUIRE(job && n && (re_indices || cap == 0), "miphy_ulsch_placeholders: null argument");
  ulsch_plan pl;
  int        rc = ulsch_plan_symbols(*job, pl);
  if (rc)
    return rc;
  uint32_t cnt = 0;
  if (pl.one_bit_fields) // ulsch_demultiplex_impl.cpp:393-453: the elements of every field that carries exactly one bit, in input order
    for (uint32_t k = 0; k < pl.nof_symbols; ++k)
      for (uint32_t i = 0; i < pl.sym[k].nof_re; ++i) {
        const ulsch_re_class c = ulsch_classify(pl.sym[k], i);
        if (c.cls != ULSCH_SCH && ((pl.one_bit_fields >> (c.cls - 1)) & 1u)) {
          if (cnt < cap)
            re_indices[cnt] = (uint16_t)(pl.sym[k].off_in + i);
          ++cnt;
        }
      }
  *n = cnt;
  return MIPHY_OK;
}

extern "C" int miphy_ulsch_demultiplex_batch(miphy_ctx* ctx, const miphy_ulsch_demux_job* jobs, uint32_t n, const int8_t* llr_in, int8_t* sch_out,
                                             int8_t* harq_ack_out, int8_t* csi_part1_out, int8_t* csi_part2_out, void* stream)
{
  MIPHY_REQUIRE(ctx && jobs && llr_in && sch_out && harq_ack_out && csi_part1_out && csi_part2_out, "miphy_ulsch_demultiplex_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "ulsch_demultiplex: at most 65535 jobs per call");
  std::vector<ulsch_plan>    plans(n);
  std::vector<demux_dev_job> dj(n);
  for (uint32_t i = 0; i < n; ++i) {
    int rc = ulsch_plan_symbols(jobs[i], plans[i]);
    if (rc)
      return rc;
    MIPHY_REQUIRE(plans[i].nof_ack_re * plans[i].bits_per_re == jobs[i].nof_enc_harq_ack_bits, "ulsch_demultiplex: job %u: HARQ-ACK length is not a whole number of elements", i);
    dj[i] = {jobs[i].in_offset, jobs[i].sch_offset, jobs[i].harq_ack_offset, jobs[i].csi_part1_offset, jobs[i].csi_part2_offset};
  }
  hipStream_t  s     = (hipStream_t)stream;
  const size_t bytes = n * (sizeof(ulsch_plan) + sizeof(demux_dev_job)) + 64;
  void*        ws    = nullptr;
  int          rc    = miphy_get_workspace(ctx, bytes, s, &ws, 0);
  if (rc)
    return rc;
  auto* d_plans = static_cast<ulsch_plan*>(ws);
  auto* d_jobs  = reinterpret_cast<demux_dev_job*>(d_plans + n);
  if ((rc = miphy_upload(ctx, d_plans, plans.data(), n * sizeof(ulsch_plan), s)) || (rc = miphy_upload(ctx, d_jobs, dj.data(), n * sizeof(demux_dev_job), s)))
    return rc; // (the host vectors go out of scope: the bytes travel through the pinned ring)
  hipLaunchKernelGGL(ulsch_demux_kernel, dim3(n, 14), dim3(256), 0, s, d_plans, d_jobs, llr_in, sch_out, harq_ack_out, csi_part1_out, csi_part2_out);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
