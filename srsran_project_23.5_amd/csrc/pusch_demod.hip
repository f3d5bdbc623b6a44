// PUSCH demodulator: resource-element extraction, ZF / MRC equalisation over the receive ports, soft demapping to int8 LLRs and
// descrambling, fused in one pass over the resource grid -- one workgroup per (transmission, OFDM symbol).
//
// Behaviour contract:
//   lib/phy/upper/channel_processors/pusch_demodulator_impl.cpp:31-152, pusch_demodulator_impl.h:74-172 (which REs, in which order)
//   lib/phy/upper/equalization/channel_equalizer_zf_impl.cpp:123-162, equalize_zf_1xn.h:42-158 (one transmit layer)
//   lib/phy/upper/channel_modulation/demodulation_mapper_impl.cpp:34-106, demodulation_mapper_{qpsk,qam16,qam64,qam256}.cpp and
//   avx2_helpers.h:103-236 (interval functions, quantisation: scale by 120/range, clip, round to nearest even)
//   descrambling: c_init = rnti * 2^15 + n_id, TS 38.211 6.3.1.1 (no UCI placeholders).
// The equalised symbols and noise variances never leave registers; HBM sees the grid and the channel estimate once and the LLRs
// once. The descrambling sequence of the symbol is produced in LDS by jumping the two LFSRs to the symbol's first bit
// (gold_device.h). Floating point: single IEEE operations in a fixed order (no contraction), exact division,
// so the LLRs are reproducible bit for bit by a scalar CPU restatement; against the reference AVX2 build they are within one step.
#define NR_DEMOD_TABLE_ATTR __device__
#include "gold_device.h"
#include "miphy_ext.h"
#include "tables/nr_demod_tables.h"
#include <cmath>

namespace {

constexpr int MAX_SYM_WORDS = 832; // 275 PRB * 12 RE * 8 bit / 32 = 825 words (+ 1 guard word)

__device__ __forceinline__ int demod_quantize(float v, float scale)
{
#pragma clang fp contract(off)
  float s = v * scale;
  s       = (s > 120.0f) ? 120.0f : s;
  s       = (s < -120.0f) ? -120.0f : s;
  const float r = rintf(s);
  return (r <= 120.0f && r >= -120.0f) ? (int)r : 0; // NaN -> 0
}

__device__ __forceinline__ float demod_interval(float x, float rcp_noise, float rcp_width, int count, const float* slope, const float* intercept)
{
#pragma clang fp contract(off)
  int idx = (int)floorf(x * rcp_width) + count / 2;
  idx     = idx < 0 ? 0 : (idx > count - 1 ? count - 1 : idx);
  float t = slope[idx] * x;
  t       = t + intercept[idx];
  return t * rcp_noise;
}

// One equalised symbol -> MOD LLRs, packed one per byte (LSB = first bit) into 64 bits.
template <int MOD>
__device__ __forceinline__ uint64_t demod_symbol(float re, float im, float nvar, unsigned sym_idx)
{
#pragma clang fp contract(off)
  int l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (MOD == 1) {
    if (nvar > 0.f) {
      const float a = (sym_idx & 1u) ? im : re, b = (sym_idx & 1u) ? -re : im;
      const float v = NR_DEMOD_QPSK_GAIN * (a + b) / nvar;
      float       c = v;
      if (fabsf(v) > 24.f)
        c = copysignf(24.f, v);
      l[0] = (int)roundf(c / 24.f * 120.f);
    }
  } else {
    const float rcp  = (nvar > 0.f) ? 1.0f / nvar : 0.0f;
    const float x[2] = {re, im};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      if (MOD == 2) {
        l[d] = demod_quantize((NR_DEMOD_QPSK_GAIN * x[d]) * rcp, 120.0f / 24.f);
      } else if (MOD == 4) {
        const float first  = NR_DEMOD_QAM16_GAIN * x[d];
        const float second = 2.0f * first - copysignf(0.8f, x[d]);
        const float l01    = (fabsf(x[d]) > NR_DEMOD_QAM16_THRESHOLD) ? second : first;
        const float l23    = 0.8f - fabsf(first);
        l[d]               = demod_quantize(l01 * rcp, 120.0f / 20.f);
        l[2 + d]           = demod_quantize(l23 * rcp, 120.0f / 20.f);
      } else if (MOD == 6) {
        l[d]     = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM64_B0_RCP_WIDTH, NR_DEMOD_QAM64_B0_COUNT, NR_DEMOD_QAM64_B0_SLOPE, NR_DEMOD_QAM64_B0_INTERCEPT), 6.0f);
        l[2 + d] = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM64_B1_RCP_WIDTH, NR_DEMOD_QAM64_B1_COUNT, NR_DEMOD_QAM64_B1_SLOPE, NR_DEMOD_QAM64_B1_INTERCEPT), 6.0f);
        l[4 + d] = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM64_B2_RCP_WIDTH, NR_DEMOD_QAM64_B2_COUNT, NR_DEMOD_QAM64_B2_SLOPE, NR_DEMOD_QAM64_B2_INTERCEPT), 6.0f);
      } else {
        l[d]     = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM256_B0_RCP_WIDTH, NR_DEMOD_QAM256_B0_COUNT, NR_DEMOD_QAM256_B0_SLOPE, NR_DEMOD_QAM256_B0_INTERCEPT), 6.0f);
        l[2 + d] = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM256_B1_RCP_WIDTH, NR_DEMOD_QAM256_B1_COUNT, NR_DEMOD_QAM256_B1_SLOPE, NR_DEMOD_QAM256_B1_INTERCEPT), 6.0f);
        l[4 + d] = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM256_B2_RCP_WIDTH, NR_DEMOD_QAM256_B2_COUNT, NR_DEMOD_QAM256_B2_SLOPE, NR_DEMOD_QAM256_B2_INTERCEPT), 6.0f);
        l[6 + d] = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM256_B3_RCP_WIDTH, NR_DEMOD_QAM256_B3_COUNT, NR_DEMOD_QAM256_B3_SLOPE, NR_DEMOD_QAM256_B3_INTERCEPT), 6.0f);
      }
    }
  }
  uint64_t out = 0;
#pragma unroll
  for (int b = 0; b < MOD; ++b)
    out |= (uint64_t)(uint8_t)(int8_t)l[b] << (8 * b);
  return out;
}

// 12-bit mask of the REs of a PRB that carry DM-RS (dmrs_mapping.h:76-92).
__device__ __forceinline__ unsigned dmrs_prb_mask(int type, unsigned cdm)
{
  unsigned m = 0;
  for (unsigned k = 0; k < 12; ++k)
    m |= ((type == 1) ? ((k % 2) < cdm) : ((k % 6) < 2 * cdm)) ? (1u << k) : 0u;
  return m;
}

template <int MOD>
__device__ __forceinline__ void demod_body(const miphy_pusch_demod_job& job, const uint16_t* prb_of, const uint8_t* pos, int npp, int n_re, int prefix,
                                           int sy, const uint32_t* cw, const float2* __restrict__ grid, const float2* __restrict__ ce, float noise_var,
                                           int8_t* __restrict__ llr, int tid, int nt)
{
#pragma clang fp contract(off)
  const int     nsc = job.grid_nof_prb * 12;
  const float2* g   = grid + job.grid_offset;
  const float2* h   = ce + job.ce_offset;
  int8_t*       o   = llr + job.llr_offset + (size_t)prefix * MOD;
  const bool    aligned = ((uintptr_t)o % (MOD == 8 ? 8 : MOD == 4 ? 4 : MOD == 1 ? 1 : 2)) == 0;
  for (int r = tid; r < n_re; r += nt) {
    const int prb = prb_of[r / npp], k = r - (r / npp) * npp;
    const int sc  = prb * 12 + pos[k];
    // equalize_zf_1xn.h:120-158
    float ch_mod_sq = 0.f, acc_re = 0.f, acc_im = 0.f;
    for (int p = 0; p < job.nof_rx_ports; ++p) {
      const float2 y = g[((size_t)job.rx_ports[p] * 14 + sy) * nsc + sc];
      const float2 c = h[((size_t)p * job.ce_nof_symbols + sy) * nsc + sc];
      const float  t = c.x * c.x, u = c.y * c.y;
      ch_mod_sq      = ch_mod_sq + (t + u);
      const float a = y.x * c.x, b = y.y * c.y, cc = y.y * c.x, d = y.x * c.y;
      acc_re        = acc_re + (a + b);
      acc_im        = acc_im + (cc - d);
    }
    const float d_pinv = 1.0f * ch_mod_sq;
    const float rcpd   = 1.0f / d_pinv;
    const float v      = rcpd * (noise_var / 1.0f);
    float       z_re = 0.f, z_im = 0.f, nv = INFINITY;
    if (d_pinv > 0.f && d_pinv < INFINITY && v > 0.f && v < INFINITY) {
      z_re = acc_re * rcpd;
      z_im = acc_im * rcpd;
      nv   = v;
    }
    uint64_t w = demod_symbol<MOD>(z_re, z_im, nv, (unsigned)(prefix + r));
    // descramble: bit b of this RE is sequence bit (prefix + r) * MOD + b; cw holds the bits of this OFDM symbol from r = 0
    const int      bi   = r * MOD;
    const uint64_t two  = (uint64_t)cw[bi >> 5] | ((uint64_t)cw[(bi >> 5) + 1] << 32);
    const uint32_t bits = (uint32_t)(two >> (bi & 31));
#pragma unroll
    for (int b = 0; b < MOD; ++b) {
      if ((bits >> b) & 1u) {
        const uint64_t m = 0xffull << (8 * b);
        const uint64_t n = (uint64_t)(uint8_t)(-(int8_t)(w >> (8 * b))) << (8 * b);
        w                = (w & ~m) | n;
      }
    }
    int8_t* q = o + (size_t)r * MOD;
    if (aligned) {
      if (MOD == 8)
        *reinterpret_cast<uint2*>(q) = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
      else if (MOD == 6) {
        uint16_t* q2 = reinterpret_cast<uint16_t*>(q);
        q2[0] = (uint16_t)w, q2[1] = (uint16_t)(w >> 16), q2[2] = (uint16_t)(w >> 32);
      } else if (MOD == 4)
        *reinterpret_cast<uint32_t*>(q) = (uint32_t)w;
      else if (MOD == 2)
        *reinterpret_cast<uint16_t*>(q) = (uint16_t)w;
      else
        q[0] = (int8_t)w;
    } else {
#pragma unroll
      for (int b = 0; b < MOD; ++b)
        q[b] = (int8_t)(w >> (8 * b));
    }
  }
}

__global__ void __launch_bounds__(256) pusch_demod_kernel(const miphy_pusch_demod_job* __restrict__ jobs, const gold_tables* __restrict__ gt,
                                                          const float2* __restrict__ grid, const float2* __restrict__ ce,
                                                          const float* __restrict__ scalars, int8_t* __restrict__ llr)
{
  __shared__ uint32_t w1[MAX_SYM_WORDS], w2[MAX_SYM_WORDS];
  __shared__ uint16_t prb_of[276];
  __shared__ uint8_t  pos[12];
  __shared__ int      nprb_s;
  const miphy_pusch_demod_job job = jobs[blockIdx.x];
  const int                   sy  = blockIdx.y;
  const int                   tid = threadIdx.x, nt = blockDim.x;
  if (sy < job.start_symbol || sy >= job.start_symbol + job.nof_symbols)
    return;
  const unsigned dmask   = dmrs_prb_mask(job.dmrs_type, job.nof_cdm_groups_without_data);
  const int      per_dm  = 12 - __popc(dmask);
  const bool     is_dmrs = (job.dmrs_symbols_mask >> sy) & 1;
  const int      npp     = is_dmrs ? per_dm : 12; // data REs per PRB in this symbol
  if (npp == 0)
    return;
  const int nprb_grid = job.grid_nof_prb;
  for (int r = tid; r < nprb_grid; r += nt) {
    const int wd = r >> 6, bt = r & 63;
    if ((job.rb_mask[wd] >> bt) & 1ull) {
      int idx = __popcll(job.rb_mask[wd] & ((1ull << bt) - 1ull));
      for (int w = 0; w < wd; ++w)
        idx += __popcll(job.rb_mask[w]);
      prb_of[idx] = (uint16_t)r;
    }
  }
  if (tid == 0) {
    int c = 0;
    for (int w = 0; w < 5; ++w)
      c += __popcll(w * 64 < nprb_grid ? (job.rb_mask[w] & ((nprb_grid - w * 64 >= 64) ? ~0ull : ((1ull << (nprb_grid - w * 64)) - 1ull))) : 0ull);
    nprb_s = c;
    int k  = 0;
    for (int q = 0; q < 12; ++q)
      if (!is_dmrs || !((dmask >> q) & 1u))
        pos[k++] = (uint8_t)q;
  }
  __syncthreads();
  const int nprb = nprb_s;
  int       prefix = 0; // data REs of the transmission before this symbol
  for (int s = job.start_symbol; s < sy; ++s)
    prefix += nprb * (((job.dmrs_symbols_mask >> s) & 1) ? per_dm : 12);
  const int n_re   = nprb * npp;
  const int mod    = job.mod;
  const int nwords = ((n_re * mod + 31) >> 5) + 1; // + 1: the 64-bit window of the last RE
  gold_long_block(*gt, (job.rnti << 15) + job.n_id, (uint32_t)prefix * (uint32_t)mod, nwords, w1, w2, w1, tid, nt);
  const float noise_var = scalars[job.scalars_offset + 2];
  switch (mod) {
    case 8:
      demod_body<8>(job, prb_of, pos, npp, n_re, prefix, sy, w1, grid, ce, noise_var, llr, tid, nt);
      break;
    case 6:
      demod_body<6>(job, prb_of, pos, npp, n_re, prefix, sy, w1, grid, ce, noise_var, llr, tid, nt);
      break;
    case 4:
      demod_body<4>(job, prb_of, pos, npp, n_re, prefix, sy, w1, grid, ce, noise_var, llr, tid, nt);
      break;
    case 2:
      demod_body<2>(job, prb_of, pos, npp, n_re, prefix, sy, w1, grid, ce, noise_var, llr, tid, nt);
      break;
    default:
      demod_body<1>(job, prb_of, pos, npp, n_re, prefix, sy, w1, grid, ce, noise_var, llr, tid, nt);
      break;
  }
}

} // namespace

extern "C" uint32_t miphy_pusch_demod_nof_llr(const miphy_pusch_demod_job* j)
{
  if (!j || (j->dmrs_type != 1 && j->dmrs_type != 2) || j->grid_nof_prb > 275)
    return 0;
  unsigned dm = 0;
  for (unsigned k = 0; k < 12; ++k)
    dm += (j->dmrs_type == 1) ? ((k % 2) < j->nof_cdm_groups_without_data) : ((k % 6) < 2u * j->nof_cdm_groups_without_data);
  unsigned nprb = 0;
  for (unsigned r = 0; r < j->grid_nof_prb; ++r)
    nprb += (unsigned)((j->rb_mask[r >> 6] >> (r & 63)) & 1ull);
  unsigned n = 0;
  for (unsigned s = j->start_symbol; s < (unsigned)j->start_symbol + j->nof_symbols && s < 14; ++s)
    n += nprb * (((j->dmrs_symbols_mask >> s) & 1) ? 12 - dm : 12);
  return n * j->mod;
}

extern "C" int miphy_pusch_demodulate_batch(miphy_ctx* ctx, const miphy_pusch_demod_job* jobs, int jobs_on_device, uint32_t n, const float* grid,
                                            const float* ce, const float* scalars, int8_t* llr, void* stream)
{
  MIPHY_REQUIRE(ctx && jobs && grid && ce && scalars && llr, "miphy_pusch_demodulate_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "pusch_demodulate: batch too large (max 65535 transmissions per call)");
  if (!jobs_on_device) {
    for (uint32_t i = 0; i < n; ++i) {
      const miphy_pusch_demod_job& j = jobs[i];
      MIPHY_REQUIRE(j.mod == 1 || j.mod == 2 || j.mod == 4 || j.mod == 6 || j.mod == 8, "pusch_demodulate: job %u: invalid modulation order %u", i, j.mod);
      MIPHY_REQUIRE(j.nof_rx_ports >= 1 && j.nof_rx_ports <= 4, "pusch_demodulate: job %u: invalid number of receive ports", i);
      MIPHY_REQUIRE(j.nof_symbols >= 1 && j.start_symbol + j.nof_symbols <= 14, "pusch_demodulate: job %u: invalid time allocation", i);
      MIPHY_REQUIRE(j.ce_nof_symbols >= j.start_symbol + j.nof_symbols && j.ce_nof_symbols <= 14, "pusch_demodulate: job %u: channel estimate too short", i);
      MIPHY_REQUIRE(j.dmrs_type == 1 || j.dmrs_type == 2, "pusch_demodulate: job %u: invalid DM-RS type", i);
      MIPHY_REQUIRE(j.nof_cdm_groups_without_data >= 1 && j.nof_cdm_groups_without_data <= (j.dmrs_type == 1 ? 2 : 3),
                    "pusch_demodulate: job %u: invalid number of CDM groups without data", i);
      MIPHY_REQUIRE(j.grid_nof_prb >= 1 && j.grid_nof_prb <= 275, "pusch_demodulate: job %u: invalid grid width", i);
      MIPHY_REQUIRE(j.n_id < 1024, "pusch_demodulate: job %u: invalid scrambling identifier", i);
      MIPHY_REQUIRE(j.rnti < 65536, "pusch_demodulate: job %u: invalid RNTI", i);
      // pusch_demodulator_impl.cpp:76-80: the codeword length must match the number of data REs
      MIPHY_REQUIRE(j.nof_llr == miphy_pusch_demod_nof_llr(&j), "pusch_demodulate: job %u: %u LLRs requested, the allocation holds %u", i, j.nof_llr,
                    miphy_pusch_demod_nof_llr(&j));
    }
  }
  hipStream_t        s  = (hipStream_t)stream;
  const gold_tables* gt = nullptr;
  int                rc = miphy_get_gold_tables(ctx, &gt);
  if (rc)
    return rc;
  const void* d_jobs = nullptr;
  rc                 = miphy_stage_descs(ctx, jobs, jobs_on_device, sizeof(miphy_pusch_demod_job) * (size_t)n, s, &d_jobs);
  if (rc)
    return rc;
  hipLaunchKernelGGL(pusch_demod_kernel, dim3(n, 14), dim3(256), 0, s, (const miphy_pusch_demod_job*)d_jobs, gt, (const float2*)grid, (const float2*)ce,
                     scalars, llr);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
