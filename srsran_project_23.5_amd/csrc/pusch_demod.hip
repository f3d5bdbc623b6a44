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
#include "mod_device.h"
#include "miphy_ext.h"
#include "tables/nr_demod_tables.h"
#include <cmath>

namespace {

// Quantisation of avx2_helpers.h:103-157: scale, clip to +-120, round to nearest even, NaN -> 0.
__device__ __forceinline__ int demod_quantize(float v, float scale)
{
#pragma clang fp contract(off)
  float s = v * scale;
  s       = (s > 120.0f) ? 120.0f : s;
  s       = (s < -120.0f) ? -120.0f : s;
  const float r = rintf(s);
  return (r <= 120.0f && r >= -120.0f) ? (int)r : 0; // NaN -> 0
}
// Same for a value that is known not to be NaN (the caller checks the inputs once per resource element).
__device__ __forceinline__ int demod_quantize_fast(float v, float scale)
{
#pragma clang fp contract(off)
  return (int)rintf(__builtin_amdgcn_fmed3f(v * scale, -120.0f, 120.0f));
}

// LDS copy of the interval tables of one modulation: level k (bits 2k, 2k+1) at tab[16 k ...], {slope, intercept} pairs. In two steps:
// the lanes request their table entry from memory (demod_table_entry) as soon as the modulation is known, and put it into LDS
// (demod_table_store) after the work that does not need it -- the request is one of the dependent memory round trips at the head of every
// workgroup (arguments -> descriptor -> tables -> samples), this takes it off the chain.
struct demod_tab_entry {
  int    idx; // position in tab[], -1: this lane holds none
  float2 v;
};
__device__ __forceinline__ demod_tab_entry demod_table_entry(int mod, int tid)
{
  demod_tab_entry e;
  e.idx = -1, e.v = make_float2(0.f, 0.f);
  if (mod == 6) {
    if (tid < 8)
      e.idx = tid, e.v = make_float2(NR_DEMOD_QAM64_B0_SLOPE[tid], NR_DEMOD_QAM64_B0_INTERCEPT[tid]);
    else if (tid < 16)
      e.idx = 16 + tid - 8, e.v = make_float2(NR_DEMOD_QAM64_B1_SLOPE[tid - 8], NR_DEMOD_QAM64_B1_INTERCEPT[tid - 8]);
    else if (tid < 20)
      e.idx = 32 + tid - 16, e.v = make_float2(NR_DEMOD_QAM64_B2_SLOPE[tid - 16], NR_DEMOD_QAM64_B2_INTERCEPT[tid - 16]);
  } else if (mod == 8) {
    if (tid < 16)
      e.idx = tid, e.v = make_float2(NR_DEMOD_QAM256_B0_SLOPE[tid], NR_DEMOD_QAM256_B0_INTERCEPT[tid]);
    else if (tid < 32)
      e.idx = tid, e.v = make_float2(NR_DEMOD_QAM256_B1_SLOPE[tid - 16], NR_DEMOD_QAM256_B1_INTERCEPT[tid - 16]);
    else if (tid < 48)
      e.idx = tid, e.v = make_float2(NR_DEMOD_QAM256_B2_SLOPE[tid - 32], NR_DEMOD_QAM256_B2_INTERCEPT[tid - 32]);
    else if (tid < 56)
      e.idx = tid, e.v = make_float2(NR_DEMOD_QAM256_B3_SLOPE[tid - 48], NR_DEMOD_QAM256_B3_INTERCEPT[tid - 48]);
  }
  return e;
}
__device__ __forceinline__ void demod_table_store(float2* tab, const demod_tab_entry& e)
{
  if (e.idx >= 0)
    tab[e.idx] = e.v;
}

__device__ __forceinline__ int demod_interval_idx(float x, float rcp_width, int count)
{
#pragma clang fp contract(off)
  const int idx = (int)floorf(x * rcp_width) + count / 2;
  return idx < 0 ? 0 : (idx > count - 1 ? count - 1 : idx);
}
__device__ __forceinline__ float demod_interval_at(float x, float rcp_noise, float2 si)
{
#pragma clang fp contract(off)
  float t = si.x * x;
  t       = t + si.y;
  return t * rcp_noise;
}

// One equalised symbol -> MOD LLRs in l[]. FAST: no NaN can appear (finite symbol, finite reciprocal noise), which allows the
// one-instruction clamp; otherwise the exact NaN-preserving sequence of the reference is used.
template <int MOD, bool FAST>
__device__ __forceinline__ void demod_symbol(float re, float im, float nvar, unsigned sym_idx, const float2* tab, int* l)
{
#pragma clang fp contract(off)
#define Q(v, sc) (FAST ? demod_quantize_fast((v), (sc)) : demod_quantize((v), (sc)))
  if (MOD == 1) {
    l[0] = 0;
    if (nvar > 0.f) {
      const float a = (sym_idx & 1u) ? im : re, b = (sym_idx & 1u) ? -re : im;
      const float v = NR_DEMOD_QPSK_GAIN * (a + b) / nvar;
      float       c = v;
      if (fabsf(v) > 24.f)
        c = copysignf(24.f, v);
      l[0] = (int)roundf(c / 24.f * 120.f);
    }
    return;
  }
  const float rcp  = (nvar > 0.f) ? 1.0f / nvar : 0.0f;
  const float x[2] = {re, im};
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    if (MOD == 2) {
      l[d] = Q((NR_DEMOD_QPSK_GAIN * x[d]) * rcp, 120.0f / 24.f);
    } else if (MOD == 4) {
      const float first  = NR_DEMOD_QAM16_GAIN * x[d];
      const float second = 2.0f * first - copysignf(0.8f, x[d]);
      const float l01    = (fabsf(x[d]) > NR_DEMOD_QAM16_THRESHOLD) ? second : first;
      const float l23    = 0.8f - fabsf(first);
      l[d]               = Q(l01 * rcp, 6.0f);
      l[2 + d]           = Q(l23 * rcp, 6.0f);
    } else if (MOD == 6) {
      const int i0 = demod_interval_idx(x[d], NR_DEMOD_QAM64_B0_RCP_WIDTH, 8); // bits 0-3 share the grid of 8 intervals
      const int i2 = demod_interval_idx(x[d], NR_DEMOD_QAM64_B2_RCP_WIDTH, 4);
      l[d]         = Q(demod_interval_at(x[d], rcp, tab[i0]), 6.0f);
      l[2 + d]     = Q(demod_interval_at(x[d], rcp, tab[16 + i0]), 6.0f);
      l[4 + d]     = Q(demod_interval_at(x[d], rcp, tab[32 + i2]), 6.0f);
    } else {
      const int i0 = demod_interval_idx(x[d], NR_DEMOD_QAM256_B0_RCP_WIDTH, 16); // bits 0-5 share the grid of 16 intervals
      const int i3 = demod_interval_idx(x[d], NR_DEMOD_QAM256_B3_RCP_WIDTH, 8);
      l[d]         = Q(demod_interval_at(x[d], rcp, tab[i0]), 6.0f);
      l[2 + d]     = Q(demod_interval_at(x[d], rcp, tab[16 + i0]), 6.0f);
      l[4 + d]     = Q(demod_interval_at(x[d], rcp, tab[32 + i0]), 6.0f);
      l[6 + d]     = Q(demod_interval_at(x[d], rcp, tab[48 + i3]), 6.0f);
    }
  }
#undef Q
}

// The common case in one pass: finite symbol and noise (FAST above), no EVM, no placeholders, MOD >= 2. Descrambling is folded into
// the quantisation scale (chip b set: scale -> -scale; clip and round-to-nearest-even are symmetric, so the LLR is negated exactly),
// and rounding + conversion into adding 1.5 * 2^23: the sum is rounded to nearest even at unit spacing and its low mantissa byte is
// the two's complement int8 of the result. Returns the MOD bytes packed, LLR b in byte b. Same values as demod_symbol<MOD, true>
// followed by the sign flip.
__device__ __forceinline__ uint32_t demod_qbyte(float v, float scale, uint32_t chips, int b)
{
#pragma clang fp contract(off)
  const float sc = __uint_as_float(((chips << (31 - b)) & 0x80000000u) | __float_as_uint(scale));
  return __float_as_uint(__builtin_amdgcn_fmed3f(v * sc, -120.0f, 120.0f) + 12582912.0f);
}
__device__ __forceinline__ uint32_t pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
  const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x0c0c0400u), cd = __builtin_amdgcn_perm(d, c, 0x0c0c0400u);
  return __builtin_amdgcn_perm(cd, ab, 0x05040100u);
}
template <int MOD>
__device__ __forceinline__ uint64_t demod_symbol_packed(float re, float im, float nvar, const float2* tab, uint32_t chips)
{
#pragma clang fp contract(off)
  const float rcp  = (nvar > 0.f) ? 1.0f / nvar : 0.0f;
  const float x[2] = {re, im};
  uint32_t    q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    if (MOD == 2) {
      q[d] = demod_qbyte((NR_DEMOD_QPSK_GAIN * x[d]) * rcp, 120.0f / 24.f, chips, d);
    } else if (MOD == 4) {
      const float first  = NR_DEMOD_QAM16_GAIN * x[d];
      const float second = 2.0f * first - copysignf(0.8f, x[d]);
      const float l01    = (fabsf(x[d]) > NR_DEMOD_QAM16_THRESHOLD) ? second : first;
      const float l23    = 0.8f - fabsf(first);
      q[d]               = demod_qbyte(l01 * rcp, 6.0f, chips, d);
      q[2 + d]           = demod_qbyte(l23 * rcp, 6.0f, chips, 2 + d);
    } else if (MOD == 6) {
      const int i0 = demod_interval_idx(x[d], NR_DEMOD_QAM64_B0_RCP_WIDTH, 8);
      const int i2 = demod_interval_idx(x[d], NR_DEMOD_QAM64_B2_RCP_WIDTH, 4);
      q[d]         = demod_qbyte(demod_interval_at(x[d], rcp, tab[i0]), 6.0f, chips, d);
      q[2 + d]     = demod_qbyte(demod_interval_at(x[d], rcp, tab[16 + i0]), 6.0f, chips, 2 + d);
      q[4 + d]     = demod_qbyte(demod_interval_at(x[d], rcp, tab[32 + i2]), 6.0f, chips, 4 + d);
    } else {
      const int i0 = demod_interval_idx(x[d], NR_DEMOD_QAM256_B0_RCP_WIDTH, 16);
      const int i3 = demod_interval_idx(x[d], NR_DEMOD_QAM256_B3_RCP_WIDTH, 8);
      q[d]         = demod_qbyte(demod_interval_at(x[d], rcp, tab[i0]), 6.0f, chips, d);
      q[2 + d]     = demod_qbyte(demod_interval_at(x[d], rcp, tab[16 + i0]), 6.0f, chips, 2 + d);
      q[4 + d]     = demod_qbyte(demod_interval_at(x[d], rcp, tab[32 + i0]), 6.0f, chips, 4 + d);
      q[6 + d]     = demod_qbyte(demod_interval_at(x[d], rcp, tab[48 + i3]), 6.0f, chips, 6 + d);
    }
  }
  const uint32_t lo = pack4(q[0], q[1], q[2], q[3]);
  const uint32_t hi = MOD > 4 ? pack4(q[4], q[5], q[6], q[7]) : 0u;
  return (uint64_t)lo | ((uint64_t)hi << 32);
}

// 12-bit mask of the REs of a PRB that carry DM-RS (dmrs_mapping.h:76-92).
__device__ __forceinline__ unsigned dmrs_prb_mask(int type, unsigned cdm)
{
  unsigned m = 0;
  for (unsigned k = 0; k < 12; ++k)
    m |= ((type == 1) ? ((k % 2) < cdm) : ((k % 6) < 2 * cdm)) ? (1u << k) : 0u;
  return m;
}

struct demod_args {
  int           nsc, nports, ce_nof_symbols;
  uint32_t      rxp; // rx_ports[0..3], one byte each
  const float2* grid;
  const float2* ce;
  int8_t*       llr;
  float         noise_var;
  const uint16_t* ph;  // repetition placeholders of this transmission (sorted RE indices), nph of them
  int             nph;
  float*          evm_part; // per (OFDM symbol, chunk) partial sums of |hard-decided symbol - equalised symbol|^2 of this transmission, or nullptr
  const uint32_t* seq;      // the transmission's scrambling sequence (pusch_scrambling_kernel)
  int             start_symbol, end_symbol, per_dm, nprb;
  int             prefix0; // data elements of the transmission before start_symbol (a workgroup may own a part of the symbols)
  unsigned        dmrs_syms;
};

constexpr int DEMOD_THREADS = 256;
constexpr int DEMOD_MAX_CHUNKS = (275 * 12 + DEMOD_THREADS - 1) / DEMOD_THREADS; // 13

// One thread = one allocated subcarrier, walked over the OFDM symbols of the allocation: with the compact estimate (one row per port,
// valid for every symbol) the channel coefficients, 1 / sum |h|^2, the post-equalisation noise variance and its reciprocal are
// computed ONCE and stay in registers -- per resource element only the received sample is loaded (the next symbol's is requested
// before the current one is worked on). The values are those of the per-element computation (same operations, same order).
// active: the thread owns a subcarrier; a_idx: its index among the allocated subcarriers (= its rank among the data elements of a
// symbol without DM-RS); r_dm: its rank among the data elements of a DM-RS symbol, -1 when it carries DM-RS there.
// The common case of demod_columns (below) with every memory request of the thread in flight at once: compact estimate, NP receive
// ports known at compile time, no EVM, no placeholders, MOD >= 2. The walk over the OFDM symbols is unrolled; the received samples of
// ALL symbols and their descrambling words are requested before the first one is worked on (one memory latency per thread instead of
// one per symbol: with a lookahead of one symbol the wavefronts of a CU spent 54 % of their cycles waiting), and are consumed in
// request order (the counter waits of the in-order returns fall out of the unrolled code). Same operations in the same order as the
// general walk, so the LLRs are identical. A descrambling window of MOD bits never straddles a word unless MOD == 6: one word then.
template <int MOD, int NP>
__device__ __forceinline__ void demod_columns_deep(const demod_args& a, bool active, int sc, int a_idx, int r_dm, const float2* tab)
{
#pragma clang fp contract(off)
  constexpr int NS  = 14;
  const int     nsc = a.nsc;
  const float2* gp[NP];
  float2        h[NP];
  // Every request is unconditional (a branch around a load makes the compiler wait for ALL outstanding loads where the paths join):
  // symbols behind the allocation repeat its last one, a lane without a subcarrier uses subcarrier sc = 0 and element 0, a DM-RS element
  // reads the word of element 0 of its symbol; what they compute is never stored.
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    gp[p] = a.grid + (size_t)((a.rxp >> (8 * p)) & 0xffu) * 14 * nsc + sc;
    h[p]  = a.ce[(size_t)p * nsc + sc];
  }
  const int a_ld = active ? a_idx : 0, r_ld = (active && r_dm >= 0) ? r_dm : 0;
  float2    y[NS][NP];
  uint32_t  w0[NS], w1[NS];
  {
    int prefix = a.prefix0;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int  sy  = min(a.start_symbol + k, a.end_symbol - 1);
      const bool dm  = (a.dmrs_syms >> sy) & 1;
#pragma unroll
      for (int p = 0; p < NP; ++p)
        y[k][p] = gp[p][(size_t)sy * nsc];
      const uint32_t bo = (uint32_t)(prefix + (dm ? r_ld : a_ld)) * (uint32_t)MOD;
      w0[k]             = a.seq[bo >> 5];
      w1[k]             = (MOD == 6) ? a.seq[(bo >> 5) + 1] : 0u;
      if (a.start_symbol + k < a.end_symbol - 1) // uniform (scalar arithmetic)
        prefix += a.nprb * (dm ? a.per_dm : 12);
    }
  }
  // equalize_zf_1xn.h:120-158, the part that only depends on the estimate (as in demod_columns)
  float ch_mod_sq = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const float t = h[p].x * h[p].x, u = h[p].y * h[p].y;
    ch_mod_sq     = ch_mod_sq + (t + u);
  }
  const float d_pinv  = 1.0f * ch_mod_sq;
  const float rcpd    = 1.0f / d_pinv;
  const float vv      = rcpd * (a.noise_var / 1.0f);
  const float nv      = (d_pinv > 0.f && d_pinv < INFINITY && vv > 0.f && vv < INFINITY) ? vv : INFINITY;
  const float rcp_chk = (nv > 0.f) ? 1.0f / nv : 0.0f;
  int         prefix  = a.prefix0;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int  sy = a.start_symbol + k;
    const bool dm = (a.dmrs_syms >> min(sy, 13)) & 1;
    const int  r  = dm ? (a.per_dm ? r_dm : -1) : a_idx;
    float acc_re = 0.f, acc_im = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const float2 c  = h[p];
      const float  aa = y[k][p].x * c.x, b = y[k][p].y * c.y, cc = y[k][p].y * c.x, d = y[k][p].x * c.y;
      acc_re          = acc_re + (aa + b);
      acc_im          = acc_im + (cc - d);
    }
    float z_re = 0.f, z_im = 0.f;
    if (nv < INFINITY) {
      z_re = acc_re * rcpd;
      z_im = acc_im * rcpd;
    }
    const bool     fast    = fabsf(z_re) < INFINITY && fabsf(z_im) < INFINITY && rcp_chk < INFINITY;
    const unsigned re      = (unsigned)(prefix + max(r, 0));
    int8_t*        q       = a.llr + (size_t)re * MOD;
    const bool     aligned = ((uintptr_t)(a.llr + (size_t)prefix * MOD) % (MOD == 8 ? 8 : MOD == 4 ? 4 : 2)) == 0;
    const uint32_t sh      = (re * (uint32_t)MOD) & 31u;
    const uint32_t bits    = (MOD == 6) ? (uint32_t)((((uint64_t)w1[k] << 32) | w0[k]) >> sh) : (w0[k] >> sh);
    const uint64_t w       = demod_symbol_packed<MOD>(z_re, z_im, nv, tab, bits); // (garbage where !fast: replaced below)
    const bool     has     = active && r >= 0 && sy < a.end_symbol;
    if (has) {
      if (fast && aligned) {
        if (MOD == 8)
          *reinterpret_cast<uint2*>(q) = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
        else if (MOD == 6) {
          uint16_t* q2 = reinterpret_cast<uint16_t*>(q);
          q2[0] = (uint16_t)w, q2[1] = (uint16_t)(w >> 16), q2[2] = (uint16_t)(w >> 32);
        } else if (MOD == 4)
          *reinterpret_cast<uint32_t*>(q) = (uint32_t)w;
        else
          *reinterpret_cast<uint16_t*>(q) = (uint16_t)w;
      } else {
        int l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (fast)
          demod_symbol<MOD, true>(z_re, z_im, nv, re, tab, l);
        else
          demod_symbol<MOD, false>(z_re, z_im, nv, re, tab, l);
#pragma unroll
        for (int b = 0; b < MOD; ++b) {
          const int m = -(int)((bits >> b) & 1u);
          q[b]        = (int8_t)((l[b] ^ m) - m);
        }
      }
    }
    if (sy < a.end_symbol)
      prefix += a.nprb * (dm ? a.per_dm : 12);
  }
}

template <int MOD>
__device__ __forceinline__ void demod_columns(const demod_args& a, bool active, int sc, int a_idx, int r_dm, const float2* tab, float* evm_red, int tid)
{
#pragma clang fp contract(off)
#ifndef DEMOD_NO_DEEP
  if (MOD >= 2 && a.ce_nof_symbols == 1 && !a.evm_part && !a.nph && a.nports == 1) { // uniform
    demod_columns_deep<(MOD >= 2 ? MOD : 2), 1>(a, active, sc, a_idx, r_dm, tab);
    return;
  }
#endif
  const int   nsc = a.nsc, nports = a.nports;
  const float noise_var = a.noise_var;
  const bool  compact   = a.ce_nof_symbols == 1;
  const float2* gp[4];
  const float2* hp[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    gp[p] = a.grid + (size_t)((a.rxp >> (8 * p)) & 0xffu) * 14 * nsc + sc;
    hp[p] = a.ce + (size_t)p * a.ce_nof_symbols * nsc + sc;
  }
  float2 h[4], yn[4];
  float  ch_mod_sq = 0.f, rcpd = 0.f, nv = INFINITY, rcp_chk = 0.f;
  auto   channel = [&]() { // equalize_zf_1xn.h:120-158, the part that only depends on the estimate
    ch_mod_sq = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (p < nports) {
        const float t = h[p].x * h[p].x, u = h[p].y * h[p].y;
        ch_mod_sq     = ch_mod_sq + (t + u);
      }
    const float d_pinv = 1.0f * ch_mod_sq;
    rcpd               = 1.0f / d_pinv;
    const float v      = rcpd * (noise_var / 1.0f);
    nv                 = (d_pinv > 0.f && d_pinv < INFINITY && v > 0.f && v < INFINITY) ? v : INFINITY;
    rcp_chk            = (nv > 0.f) ? 1.0f / nv : 0.0f;
  };
#pragma unroll
  for (int p = 0; p < 4; ++p)
    h[p] = yn[p] = make_float2(0.f, 0.f);
  // Where symbol sy finds this thread's element in the codeword (rank r among the data elements of the symbol, -1: none).
  auto rank_in = [&](int sy) { return (((a.dmrs_syms >> sy) & 1) ? (a.per_dm ? r_dm : -1) : a_idx); };
  auto fetch   = [&](int sy, int prefix, uint64_t& two) { // samples of symbol sy and the 64-bit window of its descrambling chips
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (p < nports)
        yn[p] = gp[p][(size_t)sy * nsc];
    const int r = rank_in(sy);
    if (r >= 0) {
      const uint32_t bo = (uint32_t)(prefix + r) * (uint32_t)MOD;
      two               = (uint64_t)a.seq[bo >> 5] | ((uint64_t)a.seq[(bo >> 5) + 1] << 32);
    }
  };
  uint64_t two_n = 0;
  if (active) {
    if (compact) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (p < nports)
          h[p] = hp[p][0];
    }
    fetch(a.start_symbol, a.prefix0, two_n);
  }
  if (compact)
    channel();
  int prefix = a.prefix0; // data elements of the transmission before the current symbol
  for (int sy = a.start_symbol; sy < a.end_symbol; ++sy) {
    const bool is_dmrs = (a.dmrs_syms >> sy) & 1;
    const int  npp     = is_dmrs ? a.per_dm : 12;
    float2     y[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
      y[p] = yn[p];
    const uint64_t two = two_n;
    if (active && sy + 1 < a.end_symbol)
      fetch(sy + 1, prefix + a.nprb * npp, two_n);
    const int  r   = rank_in(sy);
    const bool has = active && r >= 0;
    float      evm_acc = 0.f;
    if (has) {
      if (!compact) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (p < nports)
            h[p] = hp[p][(size_t)sy * nsc];
        channel();
      }
      float acc_re = 0.f, acc_im = 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (p < nports) {
          const float2 c = h[p];
          const float  aa = y[p].x * c.x, b = y[p].y * c.y, cc = y[p].y * c.x, d = y[p].x * c.y;
          acc_re         = acc_re + (aa + b);
          acc_im         = acc_im + (cc - d);
        }
      }
      float z_re = 0.f, z_im = 0.f;
      if (nv < INFINITY) {
        z_re = acc_re * rcpd;
        z_im = acc_im * rcpd;
      }
      // NaNs can only come from a non-finite symbol or an infinite reciprocal noise variance (0 * inf): rare, exact path.
      const bool     fast = fabsf(z_re) < INFINITY && fabsf(z_im) < INFINITY && rcp_chk < INFINITY;
      const unsigned re   = (unsigned)(prefix + r); // index of the element in the codeword
      int8_t*        q    = a.llr + (size_t)re * MOD;
      const bool     aligned = ((uintptr_t)(a.llr + (size_t)prefix * MOD) % (MOD == 8 ? 8 : MOD == 4 ? 4 : MOD == 1 ? 1 : 2)) == 0;
      // descrambling chips: bit b of this element is sequence bit re * MOD + b (window requested one symbol ahead)
      uint32_t bits = (uint32_t)(two >> ((re * (uint32_t)MOD) & 31u));
      if (MOD >= 2 && fast && aligned && !a.evm_part && !a.nph) { // the common case: quantise, descramble and pack in one go
        const uint64_t w = demod_symbol_packed<MOD>(z_re, z_im, nv, tab, bits);
        if (MOD == 8)
          *reinterpret_cast<uint2*>(q) = make_uint2((uint32_t)w, (uint32_t)(w >> 32));
        else if (MOD == 6) {
          uint16_t* q2 = reinterpret_cast<uint16_t*>(q);
          q2[0] = (uint16_t)w, q2[1] = (uint16_t)(w >> 16), q2[2] = (uint16_t)(w >> 32);
        } else if (MOD == 4)
          *reinterpret_cast<uint32_t*>(q) = (uint32_t)w;
        else
          *reinterpret_cast<uint16_t*>(q) = (uint16_t)w;
      } else {
        int l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (fast)
          demod_symbol<MOD, true>(z_re, z_im, nv, re, tab, l);
        else
          demod_symbol<MOD, false>(z_re, z_im, nv, re, tab, l);
        if (a.evm_part) { // evm_calculator_generic_impl.cpp:31-47 on the soft bits BEFORE descrambling: hard decision, modulation, error power
          uint32_t hb = 0;
#pragma unroll
          for (int b = 0; b < MOD; ++b)
            hb |= (uint32_t)(l[b] <= 0) << b;
          const float2 id = map_symbol(MOD, hb, re);
          const float  er = id.x - z_re, ei = id.y - z_im;
          evm_acc         = er * er + ei * ei;
        }
        if (a.nph) { // binary search of this element in the placeholder list (pusch_demodulator_impl.cpp:117-149)
          int  lo = 0, hi = a.nph - 1;
          bool found = false;
          while (lo <= hi) {
            const int      mid = (lo + hi) >> 1;
            const unsigned v   = a.ph[mid];
            if (v == re) {
              found = true;
              break;
            }
            if (v < re)
              lo = mid + 1;
            else
              hi = mid - 1;
          }
          if (found)
            bits = (bits & 1u) ? 3u : 0u; // y repeats the chip of bit 0, the x placeholders behind it are not scrambled
        }
#pragma unroll
        for (int b = 0; b < MOD; ++b) {
          const int m = -(int)((bits >> b) & 1u); // 0 / -1
          l[b]        = (l[b] ^ m) - m;
        }
        uint64_t w = 0;
#pragma unroll
        for (int b = 0; b < MOD; ++b)
          w |= (uint64_t)(uint8_t)(int8_t)l[b] << (8 * b);
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
    if (a.evm_part && npp != 0) { // deterministic order: wavefront shuffle tree, then the four wavefronts in sequence (uniform branch)
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
        evm_acc += __shfl_xor(evm_acc, off);
      __syncthreads();
      if ((tid & 63) == 0)
        evm_red[tid >> 6] = evm_acc;
      __syncthreads();
      if (tid == 0)
        a.evm_part[sy * DEMOD_MAX_CHUNKS + blockIdx.y] = ((evm_red[0] + evm_red[1]) + evm_red[2]) + evm_red[3];
    }
    prefix += a.nprb * npp;
  }
}

// Scrambling sequence of every transmission, one workgroup each: c(0 .. nof_llr - 1) from c_init = rnti * 2^15 + n_id, x1 from the
// table, x2 by the doubling word recurrence (gold_device.h). The OFDM symbols of a transmission use consecutive pieces of it, so no
// jump-ahead is needed; inside the demodulator the generation was a serial chain that held three of four wavefronts of every
// (transmission, symbol) workgroup at a barrier (0.13 of 0.37 ms per 1024 slots). seq: [transmission][SEQ_STRIDE] words.
constexpr int SEQ_STRIDE = GOLD_X1_WORDS;
#ifndef SCR_THREADS
#define SCR_THREADS 512
#endif
__global__ void __launch_bounds__(SCR_THREADS) pusch_scrambling_kernel(const miphy_pusch_demod_job* __restrict__ jobs, const gold_tables* __restrict__ gt,
                                                                uint32_t* __restrict__ seq)
{
  __shared__ uint32_t w[SEQ_STRIDE];
  const miphy_pusch_demod_job* __restrict__ jp = jobs + blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  int       nwords = (int)((jp->nof_llr + 31u) >> 5) + 2; // + the 64-bit window of the last resource element
  nwords           = nwords > SEQ_STRIDE ? SEQ_STRIDE : nwords;
  gold_x2_sequence(*gt, (jp->rnti << 15) + jp->n_id, nwords, w, tid, nt);
  uint32_t* o = seq + (size_t)blockIdx.x * SEQ_STRIDE;
  for (int i = tid; i < nwords; i += nt)
    o[i] = w[i] ^ gt->x1_seq[i];
}

// The descriptor as 30 dwords in scalar registers: its sub-dword fields read one by one become VECTOR byte / short loads with a wait
// each (there are no scalar sub-dword loads on gfx950) -- several dependent memory round trips at the head of every workgroup.
// Layout of miphy_pusch_demod_job (include/miphy.h; sizeof = 120 is checked by tests/test_cabi.py).
struct demod_job_words {
  uint32_t w[30];
  __device__ __forceinline__ int      mod() const { return (int)(w[2] & 0xffu); }
  __device__ __forceinline__ int      nof_rx_ports() const { return (int)((w[2] >> 8) & 0xffu); }
  __device__ __forceinline__ int      start_symbol() const { return (int)((w[2] >> 16) & 0xffu); }
  __device__ __forceinline__ int      nof_symbols() const { return (int)(w[2] >> 24); }
  __device__ __forceinline__ int      dmrs_type() const { return (int)(w[3] & 0xffu); }
  __device__ __forceinline__ unsigned cdm_groups() const { return (w[3] >> 8) & 0xffu; }
  __device__ __forceinline__ int      ce_nof_symbols() const { return (int)((w[3] >> 16) & 0xffu); }
  __device__ __forceinline__ bool     ce_compact() const { return (w[3] >> 24) != 0; }
  __device__ __forceinline__ uint32_t rx_ports() const { return w[4]; }
  __device__ __forceinline__ unsigned dmrs_symbols_mask() const { return w[5] & 0xffffu; }
  __device__ __forceinline__ int      grid_nof_prb() const { return (int)(w[5] >> 16); }
  __device__ __forceinline__ uint64_t rb_mask(int k) const { return (uint64_t)w[8 + 2 * k] | ((uint64_t)w[9 + 2 * k] << 32); }
  __device__ __forceinline__ uint64_t grid_offset() const { return (uint64_t)w[18] | ((uint64_t)w[19] << 32); }
  __device__ __forceinline__ uint64_t ce_offset() const { return (uint64_t)w[20] | ((uint64_t)w[21] << 32); }
  __device__ __forceinline__ uint64_t scalars_offset() const { return (uint64_t)w[22] | ((uint64_t)w[23] << 32); }
  __device__ __forceinline__ uint64_t llr_offset() const { return (uint64_t)w[24] | ((uint64_t)w[25] << 32); }
  __device__ __forceinline__ uint32_t placeholders_offset() const { return w[26]; }
  __device__ __forceinline__ uint32_t nof_placeholders() const { return w[27]; }
  __device__ __forceinline__ uint64_t evm_offset() const { return (uint64_t)w[28] | ((uint64_t)w[29] << 32); }
};
static_assert(sizeof(miphy_pusch_demod_job) == 120 && offsetof(miphy_pusch_demod_job, rb_mask) == 32 && offsetof(miphy_pusch_demod_job, grid_offset) == 72 &&
                  offsetof(miphy_pusch_demod_job, placeholders_offset) == 104 && offsetof(miphy_pusch_demod_job, evm_offset) == 112 &&
                  offsetof(miphy_pusch_demod_job, rx_ports) == 16 && offsetof(miphy_pusch_demod_job, dmrs_symbols_mask) == 20,
              "demod_job_words follows the layout of miphy_pusch_demod_job");
__device__ __forceinline__ demod_job_words demod_load_job(const miphy_pusch_demod_job* __restrict__ jp)
{
  const uint32_t* __restrict__ s = reinterpret_cast<const uint32_t*>(jp);
  demod_job_words j;
#pragma unroll
  for (int i = 0; i < 30; ++i)
    j.w[i] = s[i];
  return j;
}

// Allocated PRBs of a job: count and, for the lanes of a workgroup, the compact list prb_of[rank] (LDS). Uniform result.
__device__ __forceinline__ int demod_prb_list(const demod_job_words& j, uint64_t* rbm, uint16_t* prb_of, int* nprb_s, int tid, int nt)
{
  const int nprb_grid = j.grid_nof_prb();
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k)
      rbm[k] = j.rb_mask(k);
  }
  __syncthreads();
  for (int r = tid; r < nprb_grid; r += nt) {
    const int      wd = r >> 6, bt = r & 63;
    const uint64_t m  = rbm[wd];
    if ((m >> bt) & 1ull) {
      int idx = __popcll(m & ((1ull << bt) - 1ull));
      for (int w = 0; w < wd; ++w)
        idx += __popcll(rbm[w]);
      prb_of[idx] = (uint16_t)r;
    }
  }
  int c = 0; // uniform: scalar popcounts
#pragma unroll
  for (int w = 0; w < 5; ++w) {
    const int left = nprb_grid - 64 * w;
    if (left > 0)
      c += __popcll(j.rb_mask(w) & (left >= 64 ? ~0ull : ((1ull << left) - 1ull)));
  }
  (void)nprb_s;
  __syncthreads();
  return c;
}

// grid (transmissions, chunks of 256 allocated subcarriers, parts of the OFDM symbols): a batch that fills the chip walks all symbols of a
// chunk in one workgroup (the channel row is set up once); a small one (a single slot: 13 chunks) is cut along the symbols as well, which
// shortens the dependent walk of every thread.
#ifndef DEMOD_MIN_WAVES
#define DEMOD_MIN_WAVES 6 // 80 registers (four of them spilled outside the walk): 0.207 against 0.214 ms per 1024 slots at five
#endif
__global__ void __launch_bounds__(DEMOD_THREADS, DEMOD_MIN_WAVES) pusch_demod_kernel(const miphy_pusch_demod_job* __restrict__ jobs, const float2* __restrict__ grid,
                                                                    const float2* __restrict__ ce, const float* __restrict__ scalars,
                                                                    int8_t* __restrict__ llr, const uint16_t* __restrict__ placeholders,
                                                                    float* __restrict__ evm_part, const uint32_t* __restrict__ seq)
{
  __shared__ uint16_t prb_of[276];
  __shared__ int      nprb_s;
  __shared__ float2   tab[64];
  __shared__ uint64_t rbm[5];
  __shared__ float    evm_red[4];
  const demod_job_words j    = demod_load_job(jobs + blockIdx.x);
  const int             tid  = threadIdx.x;
  { // A chunk behind the allocation leaves before it has requested or built anything (the grid is sized for the widest allocation of the
    // batch: in a slot shared by eight UEs most workgroups of the narrow ones are such chunks). Scalar popcounts of the descriptor words.
    int c = 0;
#pragma unroll
    for (int w = 0; w < 5; ++w) {
      const int left = j.grid_nof_prb() - 64 * w;
      if (left > 0)
        c += __popcll(j.rb_mask(w) & (left >= 64 ? ~0ull : ((1ull << left) - 1ull)));
    }
    if ((int)blockIdx.y * DEMOD_THREADS >= c * 12)
      return; // uniform
  }
  const demod_tab_entry te   = demod_table_entry(j.mod(), tid);            // requested now, stored behind the PRB list
  const float           noise_var_early = scalars[j.scalars_offset() + 2]; // likewise
  const int             nprb = demod_prb_list(j, rbm, prb_of, &nprb_s, tid, DEMOD_THREADS);
  demod_table_store(tab, te); // (made visible by the barrier in front of the walk; a workgroup that leaves below needs no table)
  if ((int)blockIdx.y * DEMOD_THREADS >= nprb * 12)
    return; // uniform
  const unsigned dmask = dmrs_prb_mask(j.dmrs_type(), j.cdm_groups());
  demod_args     a;
  a.per_dm         = 12 - __popc(dmask);
  a.dmrs_syms      = j.dmrs_symbols_mask();
  a.start_symbol   = j.start_symbol();
  a.end_symbol     = a.start_symbol + j.nof_symbols();
  a.nprb           = nprb;
  a.prefix0        = 0;
  if (gridDim.z > 1) { // uniform
    const int per = (a.end_symbol - a.start_symbol + (int)gridDim.z - 1) / (int)gridDim.z;
    const int f = a.start_symbol + (int)blockIdx.z * per, l = min(a.end_symbol, f + per);
    if (f >= l)
      return;
    for (int sy = a.start_symbol; sy < f; ++sy)
      a.prefix0 += nprb * (((a.dmrs_syms >> sy) & 1u) ? a.per_dm : 12);
    a.start_symbol = f;
    a.end_symbol   = l;
  }
  a.nsc            = j.grid_nof_prb() * 12;
  a.nports         = j.nof_rx_ports();
  a.ce_nof_symbols = j.ce_compact() ? 1 : j.ce_nof_symbols(); // compact estimate: one row per port, valid for every symbol
  a.rxp            = j.rx_ports();
  a.grid           = grid + j.grid_offset();
  a.ce             = ce + j.ce_offset();
  a.llr            = llr + j.llr_offset();
  a.noise_var      = noise_var_early;
  a.nph            = placeholders ? (int)j.nof_placeholders() : 0;
  a.ph             = placeholders ? placeholders + j.placeholders_offset() : nullptr;
  a.evm_part       = evm_part ? evm_part + (size_t)blockIdx.x * 14 * DEMOD_MAX_CHUNKS : nullptr;
  a.seq            = seq + (size_t)blockIdx.x * SEQ_STRIDE;
  // this thread's subcarrier
  const int  a_idx  = (int)blockIdx.y * DEMOD_THREADS + tid;
  const bool active = a_idx < nprb * 12;
  const int  pr     = a_idx / 12, k = a_idx - 12 * pr;
  const int  sc     = active ? (int)prb_of[pr] * 12 + k : 0;
  const int  r_dm   = ((dmask >> k) & 1u) ? -1 : pr * a.per_dm + __popc(~dmask & ((1u << k) - 1u));
  const int  mod    = j.mod();
  __syncthreads(); // interval tables in LDS
  switch (mod) {
    case 8:
      demod_columns<8>(a, active, sc, a_idx, r_dm, tab, evm_red, tid);
      break;
    case 6:
      demod_columns<6>(a, active, sc, a_idx, r_dm, tab, evm_red, tid);
      break;
    case 4:
      demod_columns<4>(a, active, sc, a_idx, r_dm, tab, evm_red, tid);
      break;
    case 2:
      demod_columns<2>(a, active, sc, a_idx, r_dm, tab, evm_red, tid);
      break;
    default:
      demod_columns<1>(a, active, sc, a_idx, r_dm, tab, evm_red, tid);
      break;
  }
}

// Per-symbol EVM sums of a transmission from the per-chunk partials, chunks added in order (deterministic); symbols without data: 0.
__global__ void __launch_bounds__(64) pusch_evm_reduce_kernel(const miphy_pusch_demod_job* __restrict__ jobs, const float* __restrict__ evm_part,
                                                              float* __restrict__ evm_sums)
{
  const miphy_pusch_demod_job* __restrict__ jp = jobs + blockIdx.x;
  const int sy = threadIdx.x;
  if (sy >= 14)
    return;
  int nprb = 0;
  for (int w = 0; w < 5; ++w) {
    const int left = (int)jp->grid_nof_prb - 64 * w;
    if (left > 0)
      nprb += __popcll(jp->rb_mask[w] & (left >= 64 ? ~0ull : ((1ull << left) - 1ull)));
  }
  const unsigned dmask = dmrs_prb_mask(jp->dmrs_type, jp->nof_cdm_groups_without_data);
  const int      npp   = ((jp->dmrs_symbols_mask >> sy) & 1) ? 12 - __popc(dmask) : 12;
  float          s     = 0.f;
  if (sy >= jp->start_symbol && sy < jp->start_symbol + jp->nof_symbols && npp != 0) {
    const int    nch = (nprb * 12 + DEMOD_THREADS - 1) / DEMOD_THREADS;
    const float* p   = evm_part + ((size_t)blockIdx.x * 14 + sy) * DEMOD_MAX_CHUNKS;
    for (int c = 0; c < nch; ++c)
      s += p[c];
  }
  evm_sums[jp->evm_offset + sy] = s;
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
  return miphy_pusch_demodulate_batch_ex(ctx, jobs, jobs_on_device, n, grid, ce, scalars, llr, nullptr, nullptr, stream);
}

extern "C" int miphy_pusch_demodulate_batch_ex(miphy_ctx* ctx, const miphy_pusch_demod_job* jobs, int jobs_on_device, uint32_t n, const float* grid,
                                               const float* ce, const float* scalars, int8_t* llr, const uint16_t* placeholders, float* evm_sums,
                                               void* stream)
{
  MIPHY_REQUIRE(ctx && jobs && grid && ce && scalars && llr, "miphy_pusch_demodulate_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "pusch_demodulate: batch too large (max 65535 transmissions per call)");
  uint32_t max_chunks = DEMOD_MAX_CHUNKS; // device-resident jobs: the widest grid
  if (!jobs_on_device) {
    uint32_t max_prb = 1;
    for (uint32_t i = 0; i < n; ++i) {
      uint32_t np = 0;
      for (unsigned r = 0; r < jobs[i].grid_nof_prb && r < 275; ++r)
        np += (uint32_t)((jobs[i].rb_mask[r >> 6] >> (r & 63)) & 1ull);
      max_prb = np > max_prb ? np : max_prb;
    }
    max_chunks = (max_prb * 12 + DEMOD_THREADS - 1) / DEMOD_THREADS;
    for (uint32_t i = 0; i < n; ++i) {
      const miphy_pusch_demod_job& j = jobs[i];
      MIPHY_REQUIRE(j.mod == 1 || j.mod == 2 || j.mod == 4 || j.mod == 6 || j.mod == 8, "pusch_demodulate: job %u: invalid modulation order %u", i, j.mod);
      MIPHY_REQUIRE(j.nof_rx_ports >= 1 && j.nof_rx_ports <= 4, "pusch_demodulate: job %u: invalid number of receive ports", i);
      MIPHY_REQUIRE(j.nof_symbols >= 1 && j.start_symbol + j.nof_symbols <= 14, "pusch_demodulate: job %u: invalid time allocation", i);
      MIPHY_REQUIRE(j.ce_compact || (j.ce_nof_symbols >= j.start_symbol + j.nof_symbols && j.ce_nof_symbols <= 14),
                    "pusch_demodulate: job %u: channel estimate too short", i);
      MIPHY_REQUIRE(j.dmrs_type == 1 || j.dmrs_type == 2, "pusch_demodulate: job %u: invalid DM-RS type", i);
      MIPHY_REQUIRE(j.nof_cdm_groups_without_data >= 1 && j.nof_cdm_groups_without_data <= (j.dmrs_type == 1 ? 2 : 3),
                    "pusch_demodulate: job %u: invalid number of CDM groups without data", i);
      MIPHY_REQUIRE(j.grid_nof_prb >= 1 && j.grid_nof_prb <= 275, "pusch_demodulate: job %u: invalid grid width", i);
      MIPHY_REQUIRE(j.n_id < 1024, "pusch_demodulate: job %u: invalid scrambling identifier", i);
      MIPHY_REQUIRE(j.rnti < 65536, "pusch_demodulate: job %u: invalid RNTI", i);
      // pusch_demodulator_impl.cpp:76-80: the codeword length must match the number of data REs
      MIPHY_REQUIRE(j.nof_llr == miphy_pusch_demod_nof_llr(&j), "pusch_demodulate: job %u: %u LLRs requested, the allocation holds %u", i, j.nof_llr,
                    miphy_pusch_demod_nof_llr(&j));
      MIPHY_REQUIRE(j.nof_placeholders == 0 || (placeholders && j.mod >= 2), "pusch_demodulate: job %u: placeholders need the list and a modulation order of at least 2", i);
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
  void*        work      = nullptr;
  const size_t seq_bytes = (size_t)n * SEQ_STRIDE * sizeof(uint32_t);
  const size_t evm_bytes = evm_sums ? (size_t)n * 14 * DEMOD_MAX_CHUNKS * sizeof(float) : 0;
  if ((rc = miphy_get_workspace(ctx, seq_bytes + evm_bytes, s, &work, 4)))
    return rc;
  uint32_t* seq      = static_cast<uint32_t*>(work);
  float*    evm_part = evm_sums ? reinterpret_cast<float*>(static_cast<uint8_t*>(work) + seq_bytes) : nullptr;
  hipLaunchKernelGGL(pusch_scrambling_kernel, dim3(n), dim3(SCR_THREADS), 0, s, (const miphy_pusch_demod_job*)d_jobs, gt, seq);
  const uint32_t sym_parts = std::max(1u, std::min(4u, (uint32_t)ctx->num_cus / (n * max_chunks)));
  hipLaunchKernelGGL(pusch_demod_kernel, dim3(n, max_chunks, sym_parts), dim3(DEMOD_THREADS), 0, s, (const miphy_pusch_demod_job*)d_jobs, (const float2*)grid,
                     (const float2*)ce, scalars, llr, placeholders, evm_part, (const uint32_t*)seq);
  if (evm_sums)
    hipLaunchKernelGGL(pusch_evm_reduce_kernel, dim3(n), dim3(64), 0, s, (const miphy_pusch_demod_job*)d_jobs, (const float*)evm_part, evm_sums);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
