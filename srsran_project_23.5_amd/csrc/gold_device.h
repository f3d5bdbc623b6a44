// Gold sequence (TS 38.211 5.2.1) helpers shared by the DM-RS estimator and the PUSCH demodulator.
#pragma once
#include "miphy_internal.h"

// Gold sequence of TS 38.211 5.2.1 produced 28 bits per step: x(n+31+k) only depends on x(n+k), x(n+3+k) (x1) or
// x(n+k..n+3+k) (x2) for k <= 27, so a 31-bit window yields the next 28 bits with shifts and XORs.
__device__ __forceinline__ uint32_t x1_step28(uint32_t s)
{ // s: bits n..n+30 ; returns bits n+31..n+58 in [27:0]
  return ((s >> 3) ^ s) & 0x0fffffffu;
}
__device__ __forceinline__ uint32_t x2_step28(uint32_t s)
{
  return ((s >> 3) ^ (s >> 2) ^ (s >> 1) ^ s) & 0x0fffffffu;
}

// State of both LFSRs after the Nc = 1600 warm-up, as 31-bit windows. x1 starts from a constant, so its state is a
// constant; x2's state is linear in c_init: the XOR of one precomputed column per set bit of c_init.
struct gold_jump {
  uint32_t x1_1600;
  uint32_t x2_col[31];
};

__host__ __device__ inline void gold_jump_init(gold_jump& g)
{
  auto adv = [](uint32_t s1, bool is_x2) {
    for (int i = 0; i < 1600; ++i) {
      const uint32_t b = is_x2 ? (((s1 >> 3) ^ (s1 >> 2) ^ (s1 >> 1) ^ s1) & 1u) : (((s1 >> 3) ^ s1) & 1u);
      s1               = (s1 >> 1) | (b << 30);
    }
    return s1;
  };
  g.x1_1600 = adv(1u, false);
  for (int k = 0; k < 31; ++k)
    g.x2_col[k] = adv(1u << k, true);
}

// Head of one LFSR sequence: its first `head` (<= 31) 32-bit words from the 31-bit state window `s` (28 bits per step).
__device__ __forceinline__ void lfsr_head(uint32_t s, bool is_x2, int head, uint32_t* w)
{
  uint64_t acc  = 0;
  int      have = 0, k = 0;
  while (k < head) {
    acc |= (uint64_t)(s & 0x0fffffffu) << have;
    have += 28;
    if (have >= 32) {
      w[k++] = (uint32_t)acc;
      acc >>= 32;
      have -= 32;
    }
    const uint32_t n = is_x2 ? x2_step28(s) : x1_step28(s);
    s                = ((s >> 28) | (n << 3)) & 0x7fffffffu;
  }
}


// Jump-ahead: the 31-bit window s (bit i = x(n+i)) advances by one position through a linear map M; pow[k] holds the columns
// of M^(2^k), so any offset costs one matrix-vector product (31 conditional XORs) per set bit of the offset.
constexpr int GOLD_POW = 26; // offsets below 2^26
constexpr int GOLD_BASIS_WORDS = 104;
constexpr int GOLD_X1_WORDS = 11584; // x1 of c(0 .. 370687): the longest PUSCH codeword (275 PRB x 12 x 14 symbols x 8 bits) and a window word
struct gold_tables {
  gold_jump j; // state after the Nc = 1600 warm-up (first member: the estimator only needs this part)
  uint32_t  x1_pow[GOLD_POW][31];
  uint32_t  x2_pow[GOLD_POW][31];
  uint32_t  x1_seq[GOLD_X1_WORDS]; // x1 does not depend on c_init: its contribution to c(n), bit-packed LSB first from n = 0
  // x2 is LINEAR in c_init: word k of its contribution to c(n) is the XOR, over the set bits b of c_init, of x2_basis[b][k] (the sequence
  // of c_init = 2^b). The first GOLD_BASIS_WORDS words of any sequence therefore cost 31 independent loads and XORs per word, all words
  // side by side -- no serial LFSR walk (the 31-word head by one lane was the longest single step of the DM-RS estimator and of the
  // scrambling-sequence kernel). 104 words = 3 328 bits: a whole DM-RS symbol of 275 PRBs.
  uint32_t  x2_basis[31][GOLD_BASIS_WORDS];
};

__host__ __device__ inline uint32_t gold_mat_apply(const uint32_t* cols, uint32_t v)
{
  uint32_t r = 0;
  for (int i = 0; i < 31; ++i)
    r ^= ((v >> i) & 1u) ? cols[i] : 0u;
  return r;
}

inline void gold_tables_init(gold_tables& t)
{
  gold_jump_init(t.j);
  for (int i = 0; i < 31; ++i) {
    const uint32_t e = 1u << i;
    t.x1_pow[0][i]   = (e >> 1) | ((((e >> 3) ^ e) & 1u) << 30);
    t.x2_pow[0][i]   = (e >> 1) | ((((e >> 3) ^ (e >> 2) ^ (e >> 1) ^ e) & 1u) << 30);
  }
  for (int k = 1; k < GOLD_POW; ++k)
    for (int i = 0; i < 31; ++i) {
      t.x1_pow[k][i] = gold_mat_apply(t.x1_pow[k - 1], t.x1_pow[k - 1][i]);
      t.x2_pow[k][i] = gold_mat_apply(t.x2_pow[k - 1], t.x2_pow[k - 1][i]);
    }
  uint32_t s1 = t.j.x1_1600; // bit i = x1(1600 + n + i)
  for (int wd = 0; wd < GOLD_X1_WORDS; ++wd) {
    uint32_t v = 0;
    for (int b = 0; b < 32; ++b) {
      v |= (s1 & 1u) << b;
      s1 = (s1 >> 1) | ((((s1 >> 3) ^ s1) & 1u) << 30);
    }
    t.x1_seq[wd] = v;
  }
  for (int bit = 0; bit < 31; ++bit) {
    uint32_t s2 = t.j.x2_col[bit]; // window of x2 at position 1600 for c_init = 2^bit
    for (int wd = 0; wd < GOLD_BASIS_WORDS; ++wd) {
      uint32_t v = 0;
      for (int b = 0; b < 32; ++b) {
        v |= (s2 & 1u) << b;
        s2 = (s2 >> 1) | ((((s2 >> 3) ^ (s2 >> 2) ^ (s2 >> 1) ^ s2) & 1u) << 30);
      }
      t.x2_basis[bit][wd] = v;
    }
  }
}

// Word k (< GOLD_BASIS_WORDS) of the x2 half of c(n) for c_init: 31 independent loads (consecutive lanes, consecutive words).
__device__ __forceinline__ uint32_t gold_x2_word(const gold_tables& t, uint32_t c_init, int k)
{
  uint32_t v = 0;
#pragma unroll 8
  for (int b = 0; b < 31; ++b) {
    const uint32_t col = t.x2_basis[b][k];
    v ^= ((c_init >> b) & 1u) ? col : 0u;
  }
  return v;
}

// The x2 half of c(0 .. 32 nwords - 1) in LDS, by a whole workgroup (nt >= 128 threads): the first GOLD_BASIS_WORDS words from the basis
// (gold_x2_word); then the word recurrence w[i] = w[i-28] ^ w[i-29] ^ w[i-30] ^ w[i-31] (the bit recurrence raised to the 32nd power) in
// its squares w[i] = w[i-28d] ^ w[i-29d] ^ w[i-30d] ^ w[i-31d], d = 2^k, which need 31 d words of history and yield 28 d independent
// words per step -- the history doubles with every step (a step larger than the workgroup is a short loop), so a sequence of n words
// takes about log2(n / 104) + 3 barriers.
__device__ __forceinline__ void gold_x2_sequence(const gold_tables& t, uint32_t c_init, int nwords, uint32_t* w, int tid, int nt)
{
  const int head = min(nwords, GOLD_BASIS_WORDS);
  for (int i = tid; i < head; i += nt)
    w[i] = gold_x2_word(t, c_init, i);
  __syncthreads();
  int have = head, k = 0;
  while ((62 << k) <= have) // 31 * 2d words of history allow the square d -> 2d
    ++k;
  while (have < nwords) {
    const int d = 1 << k, step = 28 * d, end = min(have + step, nwords);
    for (int i = have + tid; i < end; i += nt)
      w[i] = w[i - 28 * d] ^ w[i - 29 * d] ^ w[i - 30 * d] ^ w[i - 31 * d];
    __syncthreads();
    have += step;
    while ((62 << k) <= have)
      ++k;
  }
}

// 31-bit windows of x1 / x2 at sequence position `offset` of c(n) (i.e. LFSR position 1600 + offset).
__device__ __forceinline__ uint32_t gold_state(const gold_tables& t, bool is_x2, uint32_t c_init, uint32_t offset)
{
  uint32_t st = t.j.x1_1600;
  if (is_x2) {
    st = 0;
    for (int k = 0; k < 31; ++k)
      st ^= ((c_init >> k) & 1u) ? t.j.x2_col[k] : 0u;
  }
  for (int k = 0; k < GOLD_POW; ++k)
    if ((offset >> k) & 1u)
      st = gold_mat_apply(is_x2 ? t.x2_pow[k] : t.x1_pow[k], st);
  return st;
}

// c(offset .. offset + nbits - 1), bit-packed LSB first into out[0 .. nwords), for ONE long sequence. All of it runs on the first
// wavefront, x1 on lanes 0-31 and x2 on lanes 32-63, without workgroup barriers: the jump to `offset` is a chain of matrix-vector
// products in which lane i contributes column i and a 5-step XOR butterfly adds them up; lane 0 of each half produces the 31-word
// head; then the word recurrences (see chest.hip) give 28 words per step and LFSR on lanes 0-27 of each half (LDS accesses of one
// wavefront are executed in order). The other wavefronts only join at the final barrier.
// w1 / w2: LDS scratch of nwords words each; out may alias w1.
__device__ __forceinline__ void gold_long_block(const gold_tables& t, uint32_t c_init, uint32_t offset, int nwords, uint32_t* w1, uint32_t* w2,
                                                uint32_t* out, int tid, int nt)
{
  const int head = nwords < 31 ? nwords : 31;
  if (tid < 64) {
    const bool is_x2 = tid >= 32;
    const int  lane  = tid & 31;
    auto       xor32 = [](uint32_t v) {
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1)
        v ^= __shfl_xor(v, o); // stays inside the 32-lane half
      return v;
    };
    uint32_t st = t.j.x1_1600;
    if (is_x2)
      st = xor32((lane < 31 && ((c_init >> lane) & 1u)) ? t.j.x2_col[lane] : 0u);
    // all columns this lane may need are fetched up front (independent loads: one memory latency instead of one per set bit)
    uint32_t cols[GOLD_POW];
#pragma unroll
    for (int k = 0; k < GOLD_POW; ++k)
      cols[k] = (lane < 31 && ((offset >> k) & 1u)) ? (is_x2 ? t.x2_pow[k][lane] : t.x1_pow[k][lane]) : 0u;
#pragma unroll
    for (int k = 0; k < GOLD_POW; ++k)
      if ((offset >> k) & 1u) // uniform
        st = xor32(((st >> lane) & 1u) ? cols[k] : 0u);
    volatile uint32_t* w = is_x2 ? w2 : w1;
    if (lane == 0)
      lfsr_head(st, is_x2, head, is_x2 ? w2 : w1);
    __builtin_amdgcn_wave_barrier();
    for (int i0 = 31; i0 < nwords; i0 += 28) {
      const int i = i0 + lane;
      if (lane < 28 && i < nwords)
        w[i] = is_x2 ? (w[i - 28] ^ w[i - 29] ^ w[i - 30] ^ w[i - 31]) : (w[i - 28] ^ w[i - 31]);
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int i = tid; i < nwords; i += nt)
    out[i] = w1[i] ^ w2[i];
  __syncthreads();
}

// Device copy of the tables, created on first use and cached in the context.
int miphy_get_gold_tables(miphy_ctx* ctx, const gold_tables** out);
