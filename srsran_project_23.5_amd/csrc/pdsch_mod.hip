// PDSCH modulator and PDSCH DM-RS mapping: the transmit-side counterpart of pusch_demod.hip.
//
// Behaviour contract:
//   lib/phy/upper/channel_processors/pdsch_modulator_impl.cpp:30-282 (scrambling c_init = rnti * 2^15 + q * 2^14 + n_id, modulation,
//   optional scaling, mapping to the allocated PRBs in ascending order skipping the DM-RS pattern of the bandwidth part and the
//   reserved RE patterns), lib/phy/upper/channel_modulation/modulation_mapper_impl.cpp:31-146 (constellation = integer level *
//   sqrtf(1 / average power)), lib/phy/upper/signal_processors/dmrs_pdsch_processor_impl.cpp:30-169 + dmrs_helper.h:44-96.
// One transmit layer / one codeword and contiguous (ascending) PRB mapping: the 23.5 reference cannot do more -- its layer mapper
// indexes out of bounds for more than one layer (pdsch_modulator_impl.cpp:80-103) and its non-contiguous mapping path writes a
// single PRB (map_to_prb_other). One workgroup per (transmission, OFDM symbol); every resource element is written once, with the
// exact single-precision value of the reference (level * scale [* scaling]).
#include "gold_device.h"
#include "mod_device.h"
#include "miphy_ext.h"
#include <cmath>
#include <type_traits>

namespace {


__device__ __forceinline__ unsigned dmrs_prb_mask(int type, unsigned cdm)
{
  unsigned m = 0;
  for (unsigned k = 0; k < 12; ++k)
    m |= ((type == 1) ? ((k % 2) < cdm) : ((k % 6) < 2 * cdm)) ? (1u << k) : 0u;
  return m;
}

// Allocated-PRB list (compact, ascending) from a 275-bit mask held in LDS; returns the number of allocated PRBs through *count.
__device__ __forceinline__ void build_prb_list(const uint64_t* rbm, int nprb_grid, int first_rb, uint16_t* prb_of, int* count, int tid, int nt)
{
  for (int r = tid; r < nprb_grid; r += nt) {
    const int      wd = r >> 6, bt = r & 63;
    const uint64_t m  = rbm[wd];
    if (r >= first_rb && ((m >> bt) & 1ull)) {
      int idx = __popcll(m & ((1ull << bt) - 1ull));
      for (int w = 0; w < wd; ++w)
        idx += __popcll(rbm[w]);
      prb_of[idx] = (uint16_t)r;
    }
  }
  if (tid == 0) {
    int c = 0;
    for (int w = 0; w < 5; ++w)
      c += __popcll(w * 64 < nprb_grid ? (rbm[w] & ((nprb_grid - w * 64 >= 64) ? ~0ull : ((1ull << (nprb_grid - w * 64)) - 1ull))) : 0ull);
    *count = c;
  }
}

// Scrambling sequence c(0 .. nof_bits - 1) of every transmission, one workgroup each (c_init = rnti * 2^15 + n_id, TS 38.211 7.3.1.1): x1 from
// the table, x2 from its linear basis and the doubling word recurrence (gold_device.h) -- as the PUSCH demodulator does. The OFDM symbols
// of a transmission use consecutive pieces of it; before, every (transmission, symbol) workgroup jumped both LFSRs to its first bit and ran
// the recurrences on one wavefront while three waited (0.40 ms per 1024 slots of 273 PRB: the longest kernel of the transmit chain).
constexpr int PDSCH_SEQ_STRIDE = GOLD_X1_WORDS;
__global__ void __launch_bounds__(512) pdsch_seq_kernel(const miphy_pdsch_mod_job* __restrict__ jobs, const gold_tables* __restrict__ gt, uint32_t* __restrict__ seq,
                                                        int* __restrict__ prefix_out)
{
  __shared__ uint32_t w[PDSCH_SEQ_STRIDE];
  __shared__ uint64_t rbm[5], resm[4][5];
  __shared__ int      cnt[14];
  const miphy_pdsch_mod_job* __restrict__ jp = jobs + blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  // ---- data elements of the transmission before every OFDM symbol (prefix[job][14]): each (transmission, symbol) workgroup of the
  // modulator used to recount all earlier symbols itself
  {
    const unsigned dmask     = dmrs_prb_mask(jp->dmrs_type, jp->nof_cdm_groups_without_data);
    const unsigned dmrs_syms = jp->dmrs_symbols_mask;
    const int      nprb_grid = jp->grid_nof_prb, nres = jp->nof_reserved;
    const int      bwp0 = jp->bwp_start_rb, bwp1 = bwp0 + jp->bwp_size_rb;
    const int      s0 = jp->start_symbol, s1 = s0 + jp->nof_symbols;
    if (tid < 5)
      rbm[tid] = jp->rb_mask[tid];
    if (tid >= 32 && tid < 32 + 5 * nres)
      resm[(tid - 32) / 5][(tid - 32) % 5] = jp->reserved[(tid - 32) / 5].prb_mask[(tid - 32) % 5];
    if (tid >= 64 && tid < 78)
      cnt[tid - 64] = 0;
    __syncthreads();
    for (int q = tid; q < 14 * nprb_grid; q += nt) {
      const int s = q / nprb_grid, rb = q - s * nprb_grid;
      if (s < s0 || s >= s1 || !((rbm[rb >> 6] >> (rb & 63)) & 1ull))
        continue;
      unsigned ex = (((dmrs_syms >> s) & 1u) && rb >= bwp0 && rb < bwp1) ? dmask : 0u;
      for (int r = 0; r < nres; ++r)
        if (((jp->reserved[r].symbols >> s) & 1u) && ((resm[r][rb >> 6] >> (rb & 63)) & 1ull))
          ex |= jp->reserved[r].re_mask;
      atomicAdd(&cnt[s], __popc(~ex & 0xfffu));
    }
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int s = 0; s < 14; ++s) {
        prefix_out[blockIdx.x * 14 + s] = run;
        run += cnt[s];
      }
    }
  }
  int       nwords = (int)((jp->nof_bits + 31u) >> 5) + 2; // + the 64-bit window of the last resource element
  nwords           = nwords > PDSCH_SEQ_STRIDE ? PDSCH_SEQ_STRIDE : nwords;
  gold_x2_sequence(*gt, (jp->rnti << 15) + jp->n_id, nwords, w, tid, nt);
  uint32_t* o = seq + (size_t)blockIdx.x * PDSCH_SEQ_STRIDE;
  for (int i = tid; i < nwords; i += nt)
    o[i] = w[i] ^ gt->x1_seq[i];
}

__global__ void __launch_bounds__(256) pdsch_mod_kernel(const miphy_pdsch_mod_job* __restrict__ jobs, const gold_tables* __restrict__ gt,
                                                        const uint8_t* __restrict__ cw_base, float2* __restrict__ grid, const uint32_t* __restrict__ seq_base,
                                                        const int* __restrict__ prefix_base)
{
  __shared__ uint16_t prb_of[276];
  __shared__ uint16_t keep_of[276];  // per allocated PRB: 12-bit mask of the REs that carry data in this symbol
  __shared__ uint16_t off_of[276];   // per allocated PRB: index of its first data RE within the symbol
  __shared__ uint64_t rbm[5], resm[4][5];
  __shared__ int      nprb_s, red[8];
  const miphy_pdsch_mod_job* __restrict__ jp = jobs + blockIdx.x;
  const int sy  = blockIdx.y;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int start_symbol = jp->start_symbol, nof_symbols = jp->nof_symbols;
  if (sy < start_symbol || sy >= start_symbol + nof_symbols)
    return;
  const unsigned dmask     = dmrs_prb_mask(jp->dmrs_type, jp->nof_cdm_groups_without_data);
  const unsigned dmrs_syms = jp->dmrs_symbols_mask;
  const int      nprb_grid = jp->grid_nof_prb, nres = jp->nof_reserved;
  const int      bwp0 = jp->bwp_start_rb, bwp1 = bwp0 + jp->bwp_size_rb;
  if (tid < 5)
    rbm[tid] = jp->rb_mask[tid];
  if (tid >= 32 && tid < 32 + 5 * nres)
    resm[(tid - 32) / 5][(tid - 32) % 5] = jp->reserved[(tid - 32) / 5].prb_mask[(tid - 32) % 5];
  __syncthreads();
  build_prb_list(rbm, nprb_grid, 0, prb_of, &nprb_s, tid, nt);
  __syncthreads();
  const int nprb = nprb_s;
  // reserved patterns: re_mask / symbols per pattern (uniform registers)
  unsigned res_re[4] = {0, 0, 0, 0}, res_sy[4] = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (r < nres) {
      res_re[r] = jp->reserved[r].re_mask;
      res_sy[r] = jp->reserved[r].symbols;
    }
  // data REs per PRB of this symbol; the elements of the transmission in earlier symbols were counted by pdsch_seq_kernel
  const int prefix = prefix_base[blockIdx.x * 14 + sy];
  {
    const int s = sy;
    for (int i = tid; i < nprb; i += nt) {
      const int rb = prb_of[i];
      unsigned  ex = (((dmrs_syms >> s) & 1u) && rb >= bwp0 && rb < bwp1) ? dmask : 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < nres && ((res_sy[r] >> s) & 1u) && ((resm[r][rb >> 6] >> (rb & 63)) & 1ull))
          ex |= res_re[r];
      keep_of[i] = (uint16_t)(~ex & 0xfffu);
    }
  }
  __syncthreads();
  // exclusive scan of the per-PRB counts of this symbol (<= 275 entries: one wavefront, 5 entries per lane)
  if (tid < 64) {
    int c[5], sum = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int i = tid * 5 + q;
      c[q]        = (i < nprb) ? __popc((unsigned)keep_of[i]) : 0;
      sum += c[q];
    }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(inc, o);
      inc += (tid >= o) ? v : 0;
    }
    int run = inc - sum;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int i = tid * 5 + q;
      if (i < nprb)
        off_of[i] = (uint16_t)run;
      run += c[q];
    }
    if (tid == 63)
      red[4] = inc;
  }
  __syncthreads();
  const int n_re = red[4];
  if (n_re == 0)
    return;
  const int       mod = jp->mod;
  const uint32_t* seq = seq_base + (size_t)blockIdx.x * PDSCH_SEQ_STRIDE;
  const float    scaling = jp->scaling;
  const bool     scale   = isnormal(scaling);
  const uint8_t* cw      = cw_base + jp->cw_offset;
  float2*        g       = grid + jp->grid_offset + ((size_t)jp->port * 14 + sy) * (nprb_grid * 12);
  // The common shapes (16QAM / 256QAM, codeword aligned to its element size) in two sweeps: every request of the thread's resource elements --
  // the element's bytes as one wide load and the two words of its scrambling window -- is issued before the first element is mapped, with
  // unconditional loads (an element that carries no data reads element 0 of the symbol and is not stored). The one-element-at-a-time loop
  // below paid one memory round trip per element: thirteen per thread on 273 PRBs.
  auto fast = [&](auto MODC) {
    constexpr int MOD = decltype(MODC)::value;
    constexpr int IT = (275 * 12 + 255) / 256;
    uint32_t      lo[IT], hi[IT], q0[IT], q1[IT];
    int           dst[IT]; // grid column of the element, -1: none
    uint32_t      sh[IT], dd[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int      idx  = tid + it * 256;
      const bool     in   = idx < nprb * 12;
      const int      i    = in ? idx / 12 : 0, k = in ? idx - i * 12 : 0;
      const unsigned keep = keep_of[i];
      const bool     has  = in && ((keep >> k) & 1u);
      const int      j    = has ? off_of[i] + __popc(keep & ((1u << k) - 1u)) : 0;
      const uint32_t d    = (uint32_t)prefix + (uint32_t)j;
      const uint32_t bi   = d * (uint32_t)MOD;
      q0[it] = seq[bi >> 5], q1[it] = seq[(bi >> 5) + 1];
      sh[it] = bi & 31u, dd[it] = d;
      if (MOD == 8) {
        const uint2 v = *reinterpret_cast<const uint2*>(cw + (size_t)d * 8);
        lo[it] = v.x, hi[it] = v.y;
      } else {
        lo[it] = *reinterpret_cast<const uint32_t*>(cw + (size_t)d * 4), hi[it] = 0;
      }
      dst[it] = has ? (int)prb_of[i] * 12 + k : -1;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const uint32_t cb  = (uint32_t)((((uint64_t)q1[it] << 32) | q0[it]) >> sh[it]);
      const uint32_t raw = (((lo[it] & 0x01010101u) * 0x01020408u) >> 24) | ((((hi[it] & 0x01010101u) * 0x01020408u) >> 24) << 4);
      const uint32_t nat = (raw ^ cb) & ((1u << MOD) - 1u);
      float2         x   = map_symbol(MOD, nat, dd[it]);
      if (scale) {
        x.x = x.x * scaling;
        x.y = x.y * scaling;
      }
      if (dst[it] >= 0)
        g[dst[it]] = x;
    }
  };
  if (mod == 8 && (((uintptr_t)(cw + (size_t)prefix * 8)) & 7u) == 0) {
    fast(std::integral_constant<int, 8>{});
    return;
  }
  if (mod == 4 && (((uintptr_t)(cw + (size_t)prefix * 4)) & 3u) == 0) {
    fast(std::integral_constant<int, 4>{});
    return;
  }
  for (int idx = tid; idx < nprb * 12; idx += nt) {
    const int      i = idx / 12, k = idx - i * 12;
    const unsigned keep = keep_of[i];
    if (!((keep >> k) & 1u))
      continue;
    const int      j  = off_of[i] + __popc(keep & ((1u << k) - 1u)); // RE index within the symbol
    const size_t   d  = (size_t)prefix + j;                           // symbol index within the codeword
    const uint32_t bi = (uint32_t)d * (uint32_t)mod;                  // its first bit: position in the transmission's scrambling sequence
    const uint64_t two = (uint64_t)seq[bi >> 5] | ((uint64_t)seq[(bi >> 5) + 1] << 32);
    const uint32_t cb  = (uint32_t)(two >> (bi & 31)); // scrambling bits of this RE, bit t = t-th bit
    // the symbol's bits (one per byte), b0 first, gathered into bit t = t-th bit: one wide load where the run of bytes is aligned, then
    // bit 0 of every byte through a multiplication ((x & 0x01010101) * 0x01020408 puts bytes 0..3 into bits 24..27)
    const uint8_t* cp  = cw + d * mod;
    uint32_t       raw = 0;
    if (mod == 8 && (((uintptr_t)cp) & 7u) == 0) {
      const uint2 v = *reinterpret_cast<const uint2*>(cp);
      raw           = (((v.x & 0x01010101u) * 0x01020408u) >> 24) | ((((v.y & 0x01010101u) * 0x01020408u) >> 24) << 4);
    } else if (mod == 4 && (((uintptr_t)cp) & 3u) == 0) {
      raw = ((*reinterpret_cast<const uint32_t*>(cp) & 0x01010101u) * 0x01020408u) >> 24;
    } else if (mod == 2 && (((uintptr_t)cp) & 1u) == 0) {
      const uint32_t v = *reinterpret_cast<const uint16_t*>(cp);
      raw              = (v & 1u) | ((v >> 7) & 2u);
    } else if (mod == 6 && (((uintptr_t)cp) & 1u) == 0) {
      const uint16_t* c2 = reinterpret_cast<const uint16_t*>(cp);
      const uint32_t  v0 = c2[0], v1 = c2[1], v2 = c2[2];
      raw = (v0 & 1u) | ((v0 >> 7) & 2u) | ((v1 & 1u) << 2) | ((v1 >> 5) & 8u) | ((v2 & 1u) << 4) | ((v2 >> 3) & 32u);
    } else {
      for (int t = 0; t < mod; ++t)
        raw |= (uint32_t)(cp[t] & 1u) << t;
    }
    const uint32_t nat = (raw ^ cb) & ((1u << mod) - 1u); // scrambled bits, bit t = t-th bit of the symbol (b0 first)
    float2 x = map_symbol(mod, nat, (unsigned)d);
    if (scale) {
      x.x = x.x * scaling;
      x.y = x.y * scaling;
    }
    g[prb_of[i] * 12 + k] = x;
  }
}

__global__ void __launch_bounds__(256) dmrs_pdsch_kernel(const miphy_dmrs_pdsch_job* __restrict__ jobs, const gold_tables* __restrict__ gt,
                                                         float2* __restrict__ grid)
{
  __shared__ uint32_t w1[128], w2[128];
  __shared__ uint16_t prb_of[276];
  __shared__ uint64_t rbm[5];
  __shared__ int      nprb_s;
  const miphy_dmrs_pdsch_job* __restrict__ jp = jobs + blockIdx.x;
  const int      sy   = blockIdx.y;
  const int      tid  = threadIdx.x, nt = blockDim.x;
  const unsigned syms = jp->symbols_mask;
  if (!((syms >> sy) & 1u))
    return;
  const int nprb_grid = jp->grid_nof_prb, ref = jp->reference_point_k_rb;
  const bool type2 = jp->dmrs_type == 2;
  const int  npr   = type2 ? 4 : 6;
  if (tid < 5) { // PRBs below the reference point are never generated (dmrs_helper.h:53)
    uint64_t m = jp->rb_mask[tid];
    for (int b = 0; b < 64; ++b)
      if (tid * 64 + b < ref)
        m &= ~(1ull << b);
    rbm[tid] = m;
  }
  __syncthreads();
  build_prb_list(rbm, nprb_grid, 0, prb_of, &nprb_s, tid, nt);
  const uint64_t t      = ((uint64_t)(14u * jp->slot_in_frame + (uint32_t)sy + 1u) * (2ull * jp->scrambling_id + 1ull)) % (1ull << 31);
  const uint32_t c_init = (uint32_t)((t * (1ull << 17) + (2ull * jp->scrambling_id + (jp->n_scid ? 1u : 0u))) % (1ull << 31));
  const int      nbits  = 2 * npr * (nprb_grid - ref);
  __syncthreads();
  gold_long_block(*gt, c_init, 0, ((nbits + 31) >> 5) + 1, w1, w2, w1, tid, nt);
  const int   nprb    = nprb_s;
  const float amp     = (float)(0.70710678118654752440 * (double)jp->amplitude);
  const int   l_prime = (sy != 0 && ((syms >> (sy - 1)) & 1u)) ? 1 : 0;
  const int   nports  = jp->nof_ports;
  for (int idx = tid; idx < nprb * npr; idx += nt) {
    const int i = idx / npr, q = idx - i * npr;
    const int rb = prb_of[i];
    const int gI = (rb - ref) * npr + q;                 // position in the sequence, counted from the reference point
    const int k  = !type2 ? 2 * q : (q < 2 ? q : 4 + q); // type 1: 0,2,..,10 ; type 2: 0,1,6,7
    const float re0 = ((w1[(2 * gI) >> 5] >> ((2 * gI) & 31)) & 1u) ? -amp : amp;
    const float im0 = ((w1[(2 * gI + 1) >> 5] >> ((2 * gI + 1) & 31)) & 1u) ? -amp : amp;
    for (int p = 0; p < nports; ++p) {
      const int   delta = !type2 ? (p >> 1) & 1 : 2 * ((p >> 1) % 3);
      const float wf1   = (p & 1) ? -1.f : 1.f;
      const float wt    = (l_prime && p >= (type2 ? 6 : 4)) ? -1.f : 1.f;
      const float w     = wt * ((idx & 1) ? wf1 : 1.f); // idx = index in the generated sequence (allocated PRBs only)
      grid[jp->grid_offset + ((size_t)jp->ports[p] * 14 + sy) * (nprb_grid * 12) + rb * 12 + k + delta] = make_float2(re0 * w, im0 * w);
    }
  }
}

// PDCCH: QPSK mapping of the encoded, scrambled bits and the DM-RS of one (PDU, symbol of the CORESET) per workgroup
// (pdcch_modulator_impl.cpp:30-91, dmrs_pdcch_processor_impl.cpp:30-101).
__global__ void __launch_bounds__(256) pdcch_map_kernel(const miphy_pdcch_pdu* __restrict__ pdus, const gold_tables* __restrict__ gt,
                                                        const uint8_t* __restrict__ enc_base, float2* __restrict__ grid)
{
  __shared__ uint32_t w1[128], w2[128];
  __shared__ uint16_t prb_of[276];
  __shared__ uint64_t rbm[5];
  __shared__ int      nprb_s;
  const miphy_pdcch_pdu* __restrict__ pp = pdus + blockIdx.x;
  const int s = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
  if (s >= pp->duration)
    return;
  const int sy = pp->start_symbol + s, nprb_grid = pp->grid_nof_prb, ref = pp->reference_point_k_rb;
  if (tid < 5)
    rbm[tid] = pp->rb_mask[tid];
  __syncthreads();
  build_prb_list(rbm, nprb_grid, 0, prb_of, &nprb_s, tid, nt);
  __syncthreads();
  const int nprb = nprb_s, R = 9 * nprb, prefix = s * R;
  float2*   g    = grid + pp->grid_offset + ((size_t)pp->port * 14 + sy) * (nprb_grid * 12);
  // data: scrambling slice of this symbol, c_init = (n_rnti << 16) + n_id (mod 2^31)
  gold_long_block(*gt, ((pp->n_rnti << 16) + pp->n_id_pdcch_data) & 0x7fffffffu, 2u * (uint32_t)prefix, ((2 * R + 31) >> 5) + 1, w1, w2, w1, tid, nt);
  const float    scaling = powf(10.0f, pp->data_power_offset_dB / 20.0f); // convert_dB_to_amplitude
  const bool     scale   = isnormal(scaling);
  const uint8_t* enc     = enc_base + pp->work_offset;
  for (int idx = tid; idx < R; idx += nt) {
    const int      i = idx / 9, q = idx - i * 9;
    const int      k = q + (q + 2) / 3;                       // REs 0,2,3,4,6,7,8,10,11: the DM-RS sit on 1, 5, 9
    const int      d = prefix + idx;                          // QPSK symbol index within the PDU
    const uint32_t c0 = (w1[(2 * idx) >> 5] >> ((2 * idx) & 31)) & 1u, c1 = (w1[(2 * idx + 1) >> 5] >> ((2 * idx + 1) & 31)) & 1u;
    const uint32_t nat = ((uint32_t)(enc[2 * d] & 1u) ^ c0) | (((uint32_t)(enc[2 * d + 1] & 1u) ^ c1) << 1);
    float2 x = map_symbol(2, nat, (unsigned)d);
    if (scale) {
      x.x = x.x * scaling;
      x.y = x.y * scaling;
    }
    g[prb_of[i] * 12 + k] = x;
  }
  __syncthreads();
  // DM-RS of this symbol (normal cyclic prefix: 14 symbols per slot)
  const uint64_t t      = ((uint64_t)(14u * pp->slot_in_frame + (uint32_t)sy + 1u) * (2ull * pp->n_id_pdcch_dmrs + 1ull)) % (1ull << 31);
  const uint32_t c_init = (uint32_t)((t * (1ull << 17) + 2ull * pp->n_id_pdcch_dmrs) % (1ull << 31));
  const int      nbits  = 2 * 3 * (nprb_grid - ref);
  gold_long_block(*gt, c_init, 0, ((nbits + 31) >> 5) + 1, w1, w2, w1, tid, nt);
  const float amp = (float)(0.70710678118654752440 * (double)powf(10.0f, pp->dmrs_power_offset_dB / 20.0f));
  for (int idx = tid; idx < 3 * nprb; idx += nt) {
    const int i = idx / 3, q = idx - 3 * i, rb = prb_of[i];
    if (rb < ref)
      continue; // never generated (dmrs_helper.h:58)
    const int   gI = (rb - ref) * 3 + q;
    const float re = ((w1[(2 * gI) >> 5] >> ((2 * gI) & 31)) & 1u) ? -amp : amp;
    const float im = ((w1[(2 * gI + 1) >> 5] >> ((2 * gI + 1) & 31)) & 1u) ? -amp : amp;
    g[rb * 12 + 1 + 4 * q] = make_float2(re, im);
  }
}

// SS/PBCH block: PBCH symbols, PBCH DM-RS, PSS and SSS of one block per workgroup (pbch_modulator_impl.cpp:28-113,
// dmrs_pbch_processor_impl.cpp:28-100, pss_processor_impl.cpp:28-91, sss_processor_impl.cpp:28-119).
__global__ void __launch_bounds__(256) ssb_map_kernel(const miphy_ssb_pdu* __restrict__ pdus, const gold_tables* __restrict__ gt,
                                                      const uint8_t* __restrict__ enc_base, float2* __restrict__ grid)
{
#pragma clang fp contract(off)
  __shared__ uint32_t w1[64], w2[64];
  __shared__ uint8_t  xp[134], x0[134], x1[134]; // m-sequences of PSS / SSS
  const miphy_ssb_pdu* __restrict__ pp = pdus + blockIdx.x;
  const int      tid = threadIdx.x, nt = blockDim.x;
  const unsigned N_id = pp->msg.N_id, ssb_idx = pp->msg.ssb_idx, v = N_id % 4;
  const unsigned l0 = pp->ssb_first_symbol, k0 = pp->ssb_first_subcarrier, nsc = pp->grid_nof_prb * 12u;
  if (tid < 3) {
    uint8_t* x = tid == 0 ? xp : (tid == 1 ? x0 : x1);
    for (int i = 0; i < 7; ++i)
      x[i] = 0;
    if (tid == 0)
      x[6] = 1, x[5] = 1, x[4] = 1, x[3] = 0, x[2] = 1, x[1] = 1, x[0] = 0; // pss_processor_impl.cpp:32-38
    else
      x[0] = 1;                                                             // sss_processor_impl.cpp:30-37, 51-58
    for (int i = 0; i < 127; ++i)
      x[i + 7] = (uint8_t)((x[i + (tid == 2 ? 1 : 4)] + x[i]) & 1u);
  }
  // PBCH: scrambling sequence of the cell, advanced by (ssb_idx & 7) * 864 (pbch_modulator_impl.cpp:28-38)
  gold_long_block(*gt, N_id, (ssb_idx & 7u) * 864u, 28, w1, w2, w1, tid, nt);
  const uint8_t* enc = enc_base + (size_t)blockIdx.x * 864u;
  // position of the idx-th PBCH symbol / DM-RS: symbol 1 (240 subcarriers), symbol 2 lower (48) and upper (48) part, symbol 3 (240)
  auto place = [&](int idx, int per4, unsigned& l, unsigned& k) { // per4 = 3 data REs or 1 DM-RS per group of four subcarriers
    const int n1 = 60 * per4, n2 = 12 * per4;
    int       g, r, base, ls;
    if (idx < n1)
      ls = 1, base = 0, g = idx / per4, r = idx - g * per4;
    else if (idx < n1 + n2)
      ls = 2, base = 0, g = (idx - n1) / per4, r = (idx - n1) - g * per4;
    else if (idx < n1 + 2 * n2)
      ls = 2, base = 192, g = (idx - n1 - n2) / per4, r = (idx - n1 - n2) - g * per4;
    else
      ls = 3, base = 0, g = (idx - n1 - 2 * n2) / per4, r = (idx - n1 - 2 * n2) - g * per4;
    const int pos = per4 == 1 ? (int)v : r + (r >= (int)v ? 1 : 0); // DM-RS on k % 4 == v, data on the other three
    l = l0 + ls, k = k0 + base + 4 * g + pos;
  };
  for (int idx = tid; idx < 432; idx += nt) {
    const uint32_t c0  = (w1[(2 * idx) >> 5] >> ((2 * idx) & 31)) & 1u, c1 = (w1[(2 * idx + 1) >> 5] >> ((2 * idx + 1) & 31)) & 1u;
    const uint32_t nat = ((uint32_t)(enc[2 * idx] & 1u) ^ c0) | (((uint32_t)(enc[2 * idx + 1] & 1u) ^ c1) << 1);
    const float2   x   = map_symbol(2, nat, (unsigned)idx);
    unsigned       l, k;
    place(idx, 3, l, k);
    for (int p = 0; p < pp->nof_ports; ++p)
      grid[pp->grid_offset + ((size_t)pp->ports[p] * 14 + l) * nsc + k] = x;
  }
  __syncthreads();
  // DM-RS for PBCH (dmrs_pbch_processor_impl.cpp:28-47)
  uint32_t i_ssb = (ssb_idx & 3u) + 4u * (pp->msg.hrf ? 1u : 0u);
  if (pp->msg.L_max == 8 || pp->msg.L_max == 64)
    i_ssb = ssb_idx & 7u;
  const uint32_t c_init = (((i_ssb + 1u) * ((N_id / 4u) + 1u)) << 11) + ((i_ssb + 1u) << 6) + (N_id % 4u);
  gold_long_block(*gt, c_init, 0, 10, w1, w2, w1, tid, nt);
  const float a = (float)0.70710678118654752440; // prg->generate(sequence, M_SQRT1_2)
  for (int idx = tid; idx < 144; idx += nt) {
    const float re = ((w1[(2 * idx) >> 5] >> ((2 * idx) & 31)) & 1u) ? -a : a;
    const float im = ((w1[(2 * idx + 1) >> 5] >> ((2 * idx + 1) & 31)) & 1u) ? -a : a;
    unsigned    l, k;
    place(idx, 1, l, k);
    for (int p = 0; p < pp->nof_ports; ++p)
      grid[pp->grid_offset + ((size_t)pp->ports[p] * 14 + l) * nsc + k] = make_float2(re, im);
  }
  // PSS (symbol 0) and SSS (symbol 2), subcarriers 56..182 of the block
  const unsigned nid1 = N_id / 3u, nid2 = N_id % 3u;
  const unsigned m = (43u * nid2) % 127u, m0 = 15u * (nid1 / 112u) + 5u * nid2, m1 = nid1 % 112u;
  const float    amp_pss = powf(10.0f, pp->beta_pss_dB / 20.0f); // convert_dB_to_amplitude(beta_pss)
  for (int n = tid; n < 127; n += nt) {
    const float dp = 1.0f - 2.0f * (float)xp[(n + m) % 127u];
    // srsvec::sc_prod(cf_t, float): both components are multiplied
    const float2 pss = make_float2(dp * amp_pss, 0.0f * amp_pss);
    const float  d0v = 1.0f - 2.0f * (float)x0[(n + m0) % 127u], d1v = 1.0f - 2.0f * (float)x1[(n + m1) % 127u];
    // sc_prod by amplitude 1, then the complex product with d1 (real-valued factors: the imaginary part is a signed zero)
    const float ar = d0v * 1.0f, ai = 0.0f * 1.0f, br = d1v, bi = 0.0f;
    const float2 sss = make_float2(ar * br - ai * bi, ar * bi + ai * br);
    for (int p = 0; p < pp->nof_ports; ++p) {
      float2* g = grid + pp->grid_offset + (size_t)pp->ports[p] * 14 * nsc + k0 + 56 + n;
      g[(size_t)(l0 + 0) * nsc] = pss;
      g[(size_t)(l0 + 2) * nsc] = sss;
    }
  }
}

// NZP-CSI-RS: one (job, port, OFDM symbol) per workgroup (nzp_csi_rs_generator_impl.cpp:34-296).
__global__ void __launch_bounds__(256) csi_rs_kernel(const miphy_csi_rs_job* __restrict__ jobs, const gold_tables* __restrict__ gt, float2* __restrict__ grid)
{
#pragma clang fp contract(off)
  __shared__ uint32_t w1[64], w2[64];
  const miphy_csi_rs_job* __restrict__ jp = jobs + blockIdx.x;
  const int port = blockIdx.y, l = blockIdx.z, tid = threadIdx.x, nt = blockDim.x;
  if (port >= jp->nof_ports || !((jp->symbol_mask[port] >> l) & 1u))
    return;
  const unsigned start_rb = jp->start_rb, nof_rb = jp->nof_rb, dens = jp->freq_density, cdm = jp->cdm;
  // sequence length of one symbol (:131-161) and the elements skipped below the first occupied PRB (:69-108)
  unsigned seq_len = nof_rb, first_prb = start_rb, adv;
  if (dens <= 1) {
    seq_len /= 2;
    if ((nof_rb & 1u) && (((start_rb & 1u) != 0) == (dens == 1)))
      ++seq_len;
    first_prb = dens == 0 ? start_rb + (start_rb & 1u) : start_rb + (1u - (start_rb & 1u));
    adv       = jp->mapping_row == 2 ? first_prb / 2 : first_prb;
  } else if (dens == 3) {
    seq_len *= 3;
    adv = 3 * first_prb;
  } else {
    adv = jp->mapping_row == 2 ? first_prb : 2 * first_prb;
  }
  if (cdm != 0)
    seq_len *= 2;
  const uint64_t t      = ((uint64_t)1024u * (14u * jp->slot_in_frame + (uint32_t)l + 1u) * (2ull * jp->scrambling_id + 1ull) + jp->scrambling_id) % (1ull << 31);
  gold_long_block(*gt, (uint32_t)t, 2u * adv, ((2 * (int)seq_len + 31) >> 5) + 1, w1, w2, w1, tid, nt);
  const float    amp   = (float)(0.70710678118654752440 * (double)jp->amplitude);
  const unsigned gsize = cdm == 0 ? 1u : (cdm == 1 ? 2u : (cdm == 2 ? 4u : 8u));
  const unsigned cidx  = (unsigned)port % gsize;                                  // index inside the CDM group
  const unsigned lidx  = (unsigned)__popc(jp->symbol_mask[port] & ((1u << l) - 1u)); // l' = position of this symbol in the port's pattern
  // w_f = {+1, (-1)^cidx}; w_t[l'] from the Hadamard rows of the tables (:34-57)
  const float wf1 = (cidx & 1u) ? -1.0f : 1.0f;
  float       wt  = 1.0f;
  if (cdm >= 2) {
    const unsigned row = cidx >> 1; // 0..1 (TD2) or 0..3 (TD4): rows {++++, +-+-, ++--, +--+}
    const unsigned neg = row == 0 ? 0x0u : (row == 1 ? 0xau : (row == 2 ? 0xcu : 0x6u));
    wt                 = ((neg >> lidx) & 1u) ? -1.0f : 1.0f;
  }
  const unsigned re_mask = jp->re_mask[port], n_re = (unsigned)__popc(re_mask);
  const unsigned stride = jp->rb_stride ? jp->rb_stride : 1u, nsc = jp->grid_nof_prb * 12u;
  float2*        g      = grid + jp->grid_offset + ((size_t)jp->ports[port] * 14 + l) * nsc;
  // pattern PRBs inside [start_rb, start_rb + nof_rb), ascending; element k of the sequence goes to the k-th set RE
  unsigned j0 = 0; // first pattern PRB index inside the window
  if (start_rb > jp->rb_begin)
    j0 = (start_rb - jp->rb_begin + stride - 1u) / stride;
  for (unsigned k = tid; k < seq_len; k += nt) {
    const unsigned j = k / n_re, r = k - j * n_re;
    const unsigned rb = jp->rb_begin + (j0 + j) * stride;
    if (rb >= jp->rb_end || rb >= start_rb + nof_rb)
      continue;
    unsigned m = re_mask; // r-th set bit
    for (unsigned q = 0; q < r; ++q)
      m &= m - 1u;
    const unsigned sc = (unsigned)__ffs((int)m) - 1u;
    float re = ((w1[(2 * k) >> 5] >> ((2 * k) & 31)) & 1u) ? -amp : amp;
    float im = ((w1[(2 * k + 1) >> 5] >> ((2 * k + 1) & 31)) & 1u) ? -amp : amp;
    if (cdm == 1) {
      const float w = (k & 1u) ? wf1 : 1.0f; // table.w_f[k'] * seq
      re = w * re, im = w * im;
    } else if (cdm >= 2) {
      const float w = wt * ((k & 1u) ? wf1 : 1.0f); // (w_t[l'] * w_f[k']) * seq
      re = w * re, im = w * im;
    }
    g[rb * 12u + sc] = make_float2(re, im);
  }
}

uint32_t host_nof_re(const miphy_pdsch_mod_job& j)
{
  unsigned dm = 0;
  for (unsigned k = 0; k < 12; ++k)
    dm |= ((j.dmrs_type == 1) ? ((k % 2) < j.nof_cdm_groups_without_data) : ((k % 6) < 2u * j.nof_cdm_groups_without_data)) ? (1u << k) : 0u;
  uint32_t n = 0;
  for (unsigned s = j.start_symbol; s < (unsigned)j.start_symbol + j.nof_symbols && s < 14; ++s)
    for (unsigned rb = 0; rb < j.grid_nof_prb; ++rb) {
      if (!((j.rb_mask[rb >> 6] >> (rb & 63)) & 1ull))
        continue;
      unsigned ex = (((j.dmrs_symbols_mask >> s) & 1u) && rb >= j.bwp_start_rb && rb < (unsigned)j.bwp_start_rb + j.bwp_size_rb) ? dm : 0u;
      for (unsigned r = 0; r < j.nof_reserved && r < 4; ++r)
        if (((j.reserved[r].symbols >> s) & 1u) && ((j.reserved[r].prb_mask[rb >> 6] >> (rb & 63)) & 1ull))
          ex |= j.reserved[r].re_mask;
      n += (uint32_t)__builtin_popcount(~ex & 0xfffu);
    }
  return n;
}

} // namespace

extern "C" uint32_t miphy_pdsch_mod_nof_re(const miphy_pdsch_mod_job* j)
{
  if (!j || (j->dmrs_type != 1 && j->dmrs_type != 2) || j->grid_nof_prb > 275 || j->nof_reserved > 4)
    return 0;
  return host_nof_re(*j);
}

extern "C" int miphy_pdsch_modulate_batch(miphy_ctx* ctx, const miphy_pdsch_mod_job* jobs, int jobs_on_device, uint32_t n, const uint8_t* codewords,
                                          float* grid, void* stream)
{
  MIPHY_REQUIRE(ctx && jobs && codewords && grid, "miphy_pdsch_modulate_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "pdsch_modulate: batch too large (max 65535 transmissions per call)");
  if (!jobs_on_device) {
    for (uint32_t i = 0; i < n; ++i) {
      const miphy_pdsch_mod_job& j = jobs[i];
      MIPHY_REQUIRE(j.mod == 1 || j.mod == 2 || j.mod == 4 || j.mod == 6 || j.mod == 8, "pdsch_modulate: job %u: invalid modulation order %u", i, j.mod);
      MIPHY_REQUIRE(j.nof_symbols >= 1 && j.start_symbol + j.nof_symbols <= 14, "pdsch_modulate: job %u: invalid time allocation", i);
      MIPHY_REQUIRE(j.dmrs_type == 1 || j.dmrs_type == 2, "pdsch_modulate: job %u: invalid DM-RS type", i);
      MIPHY_REQUIRE(j.nof_cdm_groups_without_data >= 1 && j.nof_cdm_groups_without_data <= (j.dmrs_type == 1 ? 2 : 3),
                    "pdsch_modulate: job %u: invalid number of CDM groups without data", i);
      MIPHY_REQUIRE(j.grid_nof_prb >= 1 && j.grid_nof_prb <= 275 && j.bwp_start_rb + j.bwp_size_rb <= 275, "pdsch_modulate: job %u: invalid grid / BWP", i);
      MIPHY_REQUIRE(j.nof_reserved <= 4, "pdsch_modulate: job %u: at most 4 reserved RE patterns (re_pattern_list::MAX_RE_PATTERN)", i);
      MIPHY_REQUIRE(j.n_id < 1024 && j.rnti < 65536, "pdsch_modulate: job %u: invalid scrambling identifiers", i);
      MIPHY_REQUIRE(j.port < 16, "pdsch_modulate: job %u: invalid port", i);
      // pdsch_modulator_impl.cpp:158-160: every element of the layer must be mapped
      MIPHY_REQUIRE(j.nof_bits == host_nof_re(j) * j.mod, "pdsch_modulate: job %u: codeword of %u bits, the allocation holds %u", i, j.nof_bits,
                    host_nof_re(j) * j.mod);
    }
  }
  hipStream_t        s  = (hipStream_t)stream;
  const gold_tables* gt = nullptr;
  int                rc = miphy_get_gold_tables(ctx, &gt);
  if (rc)
    return rc;
  const void* d_jobs = nullptr;
  rc                 = miphy_stage_descs(ctx, jobs, jobs_on_device, sizeof(miphy_pdsch_mod_job) * (size_t)n, s, &d_jobs);
  if (rc)
    return rc;
  void* seq = nullptr; // scrambling sequences of the transmissions (the workspace the PUSCH demodulator keeps its sequences in), then their 14 symbol prefixes
  const size_t seq_bytes = (size_t)n * PDSCH_SEQ_STRIDE * sizeof(uint32_t);
  if ((rc = miphy_get_workspace(ctx, seq_bytes + (size_t)n * 14 * sizeof(int), s, &seq, 4)))
    return rc;
  int* prefix = reinterpret_cast<int*>(static_cast<uint8_t*>(seq) + seq_bytes);
  hipLaunchKernelGGL(pdsch_seq_kernel, dim3(n), dim3(512), 0, s, (const miphy_pdsch_mod_job*)d_jobs, gt, (uint32_t*)seq, prefix);
  hipLaunchKernelGGL(pdsch_mod_kernel, dim3(n, 14), dim3(256), 0, s, (const miphy_pdsch_mod_job*)d_jobs, gt, codewords, (float2*)grid, (const uint32_t*)seq,
                     (const int*)prefix);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

extern "C" int miphy_dmrs_pdsch_map_batch(miphy_ctx* ctx, const miphy_dmrs_pdsch_job* jobs, int jobs_on_device, uint32_t n, float* grid, void* stream)
{
  MIPHY_REQUIRE(ctx && jobs && grid, "miphy_dmrs_pdsch_map_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "dmrs_pdsch_map: batch too large (max 65535 transmissions per call)");
  if (!jobs_on_device) {
    for (uint32_t i = 0; i < n; ++i) {
      const miphy_dmrs_pdsch_job& j = jobs[i];
      MIPHY_REQUIRE(j.dmrs_type == 1 || j.dmrs_type == 2, "dmrs_pdsch_map: job %u: invalid DM-RS type", i);
      MIPHY_REQUIRE(j.nof_ports >= 1 && j.nof_ports <= (j.dmrs_type == 1 ? 8 : 12), "dmrs_pdsch_map: job %u: invalid number of ports", i);
      MIPHY_REQUIRE(j.grid_nof_prb >= 1 && j.grid_nof_prb <= 275 && j.reference_point_k_rb < j.grid_nof_prb, "dmrs_pdsch_map: job %u: invalid grid", i);
      MIPHY_REQUIRE(j.symbols_mask < (1u << 14), "dmrs_pdsch_map: job %u: invalid symbol mask", i);
    }
  }
  hipStream_t        s  = (hipStream_t)stream;
  const gold_tables* gt = nullptr;
  int                rc = miphy_get_gold_tables(ctx, &gt);
  if (rc)
    return rc;
  const void* d_jobs = nullptr;
  rc                 = miphy_stage_descs(ctx, jobs, jobs_on_device, sizeof(miphy_dmrs_pdsch_job) * (size_t)n, s, &d_jobs);
  if (rc)
    return rc;
  hipLaunchKernelGGL(dmrs_pdsch_kernel, dim3(n, 14), dim3(256), 0, s, (const miphy_dmrs_pdsch_job*)d_jobs, gt, (float2*)grid);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

// ---------------------------------------------------------------------------------------------------- PDCCH processor
extern "C" int miphy_pdcch_process_batch(miphy_ctx* ctx, const miphy_pdcch_pdu* pdus, uint32_t n, const uint8_t* payloads, float* grid, void* stream)
{
  MIPHY_REQUIRE(ctx && pdus && payloads && grid, "miphy_pdcch_process_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "pdcch_process: at most 65535 PDUs per call");
  hipStream_t                  s = (hipStream_t)stream;
  std::vector<miphy_pdcch_pdu> p(pdus, pdus + n);
  size_t                       enc_bytes = 0;
  for (uint32_t i = 0; i < n; ++i) {
    miphy_pdcch_pdu& q  = p[i];
    const unsigned   al = q.aggregation_level;
    MIPHY_REQUIRE(al == 1 || al == 2 || al == 4 || al == 8 || al == 16, "pdcch_process: PDU %u: invalid aggregation level %u", i, al);
    MIPHY_REQUIRE(q.duration >= 1 && q.duration <= 3 && q.start_symbol + q.duration <= 14, "pdcch_process: PDU %u: invalid CORESET duration", i);
    MIPHY_REQUIRE(q.payload_size >= 12 && q.payload_size <= 128, "pdcch_process: PDU %u: payload size %u out of range (12..128)", i, (unsigned)q.payload_size);
    MIPHY_REQUIRE(q.grid_nof_prb >= 1 && q.grid_nof_prb <= 275 && q.reference_point_k_rb < q.grid_nof_prb, "pdcch_process: PDU %u: invalid grid", i);
    MIPHY_REQUIRE(q.rnti <= 0xffff && q.n_rnti <= 0xffff && q.n_id_pdcch_data <= 0xffff && q.n_id_pdcch_dmrs <= 0xffff, "pdcch_process: PDU %u: identifier out of range", i);
    unsigned nprb = 0;
    for (unsigned r = 0; r < q.grid_nof_prb; ++r)
      nprb += (unsigned)((q.rb_mask[r >> 6] >> (r & 63)) & 1ull);
    // E = aggregation level x 6 REG x 9 RE x 2 bits must fill the PRBs of the mask over the CORESET symbols
    MIPHY_REQUIRE(nprb * q.duration == 6u * al, "pdcch_process: PDU %u: %u PRBs x %u symbols do not match aggregation level %u", i, nprb, (unsigned)q.duration, al);
    q.work_offset = enc_bytes;
    enc_bytes += (108u * al + 15u) & ~15u;
  }
  void* work = nullptr;
  int   rc   = miphy_get_workspace(ctx, enc_bytes + 64, s, &work, 2);
  if (rc)
    return rc;
  const void* d_pdus = nullptr;
  if ((rc = miphy_stage_descs(ctx, p.data(), 0, sizeof(miphy_pdcch_pdu) * (size_t)n, s, &d_pdus)))
    return rc;
  const gold_tables* gt = nullptr;
  if ((rc = miphy_get_gold_tables(ctx, &gt)))
    return rc;
  uint8_t* d_enc = static_cast<uint8_t*>(work);
  for (uint32_t i = 0; i < n; ++i) { // a handful of PDUs per slot: one small launch each (the code depends on payload size and level)
    const miphy_pdcch_pdu* dq = static_cast<const miphy_pdcch_pdu*>(d_pdus) + i;
    if ((rc = miphy_pdcch_encode_batch(ctx, p[i].payload_size, 108u * p[i].aggregation_level, 1, payloads + p[i].payload_offset,
                                       reinterpret_cast<const uint16_t*>(&dq->rnti), d_enc + p[i].work_offset, s)))
      return rc;
  }
  hipLaunchKernelGGL(pdcch_map_kernel, dim3(n, 3), dim3(256), 0, s, (const miphy_pdcch_pdu*)d_pdus, gt, d_enc, (float2*)grid);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

// ---------------------------------------------------------------------------------------------------- SS/PBCH block processor
extern "C" int miphy_ssb_process_batch(miphy_ctx* ctx, const miphy_ssb_pdu* pdus, uint32_t n, float* grid, void* stream)
{
  MIPHY_REQUIRE(ctx && pdus && grid, "miphy_ssb_process_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "ssb_process: at most 65535 blocks per call");
  hipStream_t                 s = (hipStream_t)stream;
  std::vector<miphy_pbch_msg> msgs(n);
  for (uint32_t i = 0; i < n; ++i) {
    const miphy_ssb_pdu& q = pdus[i];
    MIPHY_REQUIRE(q.msg.N_id < 1008, "ssb_process: block %u: invalid physical cell identity %u", i, q.msg.N_id);
    MIPHY_REQUIRE(q.msg.L_max == 4 || q.msg.L_max == 8 || q.msg.L_max == 64, "ssb_process: block %u: invalid L_max %u", i, q.msg.L_max);
    MIPHY_REQUIRE(q.nof_ports >= 1 && q.nof_ports <= 4, "ssb_process: block %u: invalid number of ports", i);
    MIPHY_REQUIRE(q.grid_nof_prb >= 20 && q.grid_nof_prb <= 275 && q.ssb_first_subcarrier + 240u <= q.grid_nof_prb * 12u, "ssb_process: block %u: the block does not fit the grid",
                  i);
    MIPHY_REQUIRE(q.ssb_first_symbol + 4u <= 14u, "ssb_process: block %u: the block does not fit the slot", i);
    msgs[i] = q.msg;
  }
  void* work = nullptr; // encoded PBCH bits, 864 per block
  int   rc   = miphy_get_workspace(ctx, (size_t)n * 864 + 64, s, &work, 2);
  if (rc)
    return rc;
  if ((rc = miphy_pbch_encode_batch(ctx, msgs.data(), n, static_cast<uint8_t*>(work), s)))
    return rc;
  const void* d_pdus = nullptr;
  if ((rc = miphy_stage_descs(ctx, pdus, 0, sizeof(miphy_ssb_pdu) * (size_t)n, s, &d_pdus)))
    return rc;
  const gold_tables* gt = nullptr;
  if ((rc = miphy_get_gold_tables(ctx, &gt)))
    return rc;
  hipLaunchKernelGGL(ssb_map_kernel, dim3(n), dim3(256), 0, s, (const miphy_ssb_pdu*)d_pdus, gt, static_cast<const uint8_t*>(work), (float2*)grid);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

// ---------------------------------------------------------------------------------------------------- NZP-CSI-RS generator
extern "C" int miphy_csi_rs_map_batch(miphy_ctx* ctx, const miphy_csi_rs_job* jobs, int jobs_on_device, uint32_t n, float* grid, void* stream)
{
  MIPHY_REQUIRE(ctx && jobs && grid, "miphy_csi_rs_map_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "csi_rs_map: at most 65535 jobs per call");
  if (!jobs_on_device)
    for (uint32_t i = 0; i < n; ++i) {
      const miphy_csi_rs_job& j = jobs[i];
      MIPHY_REQUIRE(j.nof_ports >= 1 && j.nof_ports <= 16, "csi_rs_map: job %u: invalid number of ports %u", i, (unsigned)j.nof_ports);
      MIPHY_REQUIRE(j.cdm <= 3 && j.freq_density <= 3, "csi_rs_map: job %u: invalid CDM type or density", i);
      MIPHY_REQUIRE(j.grid_nof_prb >= 1 && j.grid_nof_prb <= 275 && (unsigned)j.start_rb + j.nof_rb <= j.grid_nof_prb && j.rb_end <= j.grid_nof_prb && j.nof_rb >= 1,
                    "csi_rs_map: job %u: the PRB range exceeds the grid", i);
      MIPHY_REQUIRE(j.rb_stride >= 1 && j.rb_begin <= j.rb_end, "csi_rs_map: job %u: invalid PRB pattern", i);
      const unsigned per_symbol = (j.freq_density == 3 ? 3u : 1u) * (j.cdm ? 2u : 1u) * j.nof_rb;
      MIPHY_REQUIRE(per_symbol <= 1000, "csi_rs_map: job %u: sequence too long", i);
      for (unsigned p = 0; p < j.nof_ports; ++p)
        MIPHY_REQUIRE(j.re_mask[p] != 0 && j.re_mask[p] < (1u << 12) && j.symbol_mask[p] != 0 && j.symbol_mask[p] < (1u << 14), "csi_rs_map: job %u: port %u: invalid pattern", i, p);
    }
  hipStream_t s      = (hipStream_t)stream;
  const void* d_jobs = nullptr;
  int         rc     = miphy_stage_descs(ctx, jobs, jobs_on_device, sizeof(miphy_csi_rs_job) * (size_t)n, s, &d_jobs);
  if (rc)
    return rc;
  const gold_tables* gt = nullptr;
  if ((rc = miphy_get_gold_tables(ctx, &gt)))
    return rc;
  hipLaunchKernelGGL(csi_rs_kernel, dim3(n, 16, 14), dim3(256), 0, s, (const miphy_csi_rs_job*)d_jobs, gt, (float2*)grid);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
