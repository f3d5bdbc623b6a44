// Device code shared by the packed LDPC decoder kernels (ldpc_decode_pk.hip: one codeblock per workgroup; ldpc_decode_pkw.hip:
// several small codeblocks per wavefront). Arithmetic contract: ldpc_decoder_impl.cpp:60-146 + ldpc_decoder_avx2.cpp:66-243,
// avx2_support.h:65-106 of the reference; every kernel built from these functions is bit-identical to it.
#pragma once
#include "miphy_internal.h"

namespace {

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x2 as_s2(uint32_t x)
{
  return __builtin_bit_cast(s16x2, x);
}
__device__ __forceinline__ uint32_t as_u(s16x2 x)
{
  return __builtin_bit_cast(uint32_t, x);
}
__device__ __forceinline__ s16x2 splat(int c)
{
  return s16x2{(short)c, (short)c};
}
__device__ __forceinline__ s16x2 pk_min(s16x2 a, s16x2 b)
{
  return __builtin_elementwise_min(a, b);
}
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b)
{
  return __builtin_elementwise_max(a, b);
}
__device__ __forceinline__ s16x2 pk_ashr15(s16x2 a)
{
  return a >> splat(15);
}
// two sign-extended bytes -> one register with two int16 (bytes 1:0 of each source)
__device__ __forceinline__ s16x2 pk_pair(int lo, int hi)
{
  return as_s2(__builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u));
}
// Message dword {A1, A0, B1, B0} (bytes 0..3) -> edge 0 = (sext A0, sext B0), edge 1 = (sext A1, sext B1).
// v_perm_b32 selectors 8 / 9 replicate the sign of byte 1 / 3 of the low source (10 / 11: of the high source).
__device__ __forceinline__ s16x2 c2v_even(uint32_t w)
{
  return as_s2(__builtin_amdgcn_perm(w, w, 0x09030801u));
}
__device__ __forceinline__ s16x2 c2v_odd(uint32_t w)
{
  const uint32_t h = w << 8; // bytes {0, A1, A0, B1}: A1 -> byte 1, B1 -> byte 3
  return as_s2(__builtin_amdgcn_perm(h, h, 0x09030801u));
}
__device__ __forceinline__ uint32_t c2v_pack(s16x2 c0, s16x2 c1)
{
  // {S0 = c0 -> bytes 4..7, S1 = c1 -> bytes 0..3}: out = {c1.A, c0.A, c1.B, c0.B}
  return __builtin_amdgcn_perm(as_u(c0), as_u(c1), 0x06020400u);
}

// Finer stamps inside a layer of the latency form (tools/ldpc_phase_probe.py --inner; debug build with -DLDPC_PK_PROFILE2 only).
#ifdef LDPC_PK_PROFILE2
__device__ unsigned long long g_ldpc_prof2[8];
#define P2_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define P2_ADD(slot, a, b)                                                 \
  do {                                                                     \
    if (threadIdx.x == 0)                                                  \
      atomicAdd(&g_ldpc_prof2[slot], (unsigned long long)((b) - (a)));     \
  } while (0)
#else
#define P2_T(var)
#define P2_ADD(slot, a, b)
#endif

// Wavefront priorities of the packed decoders (s_setprio, 0 = lowest). The SIMD's arbiter otherwise favours the oldest ready wavefront; a
// wavefront that is about to REQUEST the soft bits and messages of a layer should go first (its LDS round trip then runs under the other
// wavefronts' arithmetic), the long first phase of the row update last, the second phase (stores, then the barrier the codeblock's other
// wavefronts wait at) and the phases around the layer loop in between. Measured on the headline launch (38 912 codeblocks, same box):
// no priorities 3.72 ms, request 1 alone 3.64, request 2 + second phase 1 3.46-3.52, request 3 + second phase 2 3.47, the same + outside 2
// 3.43 (the values below), outside 3 3.45, second phase 3 3.52, first phase 1 3.48. -D overrides keep the A-B of tools/ab_bench.py possible.
#ifndef LDPC_PK_SETPRIO
#define LDPC_PK_SETPRIO 3
#endif
#ifndef LDPC_PK_SETPRIO1
#define LDPC_PK_SETPRIO1 0
#endif
#ifndef LDPC_PK_SETPRIO2
#define LDPC_PK_SETPRIO2 2
#endif
#ifndef LDPC_PK_SETPRIO_OUT
#define LDPC_PK_SETPRIO_OUT 2
#endif

constexpr int LLR_MAX = 120;
constexpr int LLR_INF = 127;
constexpr int INF_MUL = 255; // an infinite soft bit (|s| > 120) becomes a message of magnitude >= 255 + 24

// `base` = LDS byte offset of the codeblock's soft bits (0 where a workgroup holds one codeblock: the term then folds away).
// PARTS > 1 (latency form of the packed kernel: PARTS times the wavefronts per codeblock, each part of the workgroup owns a share of
// the edges of a layer): D is the number of edges of THIS part; between the two phases the parts exchange their partial {min1, min2,
// sign parity} through LDS (`xch` = this lane's column of the exchange area [part][3][xs], `part` = this part) and merge them -- the
// two smallest magnitudes of a union are min(a1, b1) and min(max(a1, b1), min(a2, b2)), an associative rule. The function then
// contains a workgroup barrier: EVERY thread calls it, `active` = the lane owns rows (an idle lane computes on soft bits it may read
// but stores nothing).
struct pk_no_hook {
  __device__ __forceinline__ void operator()() const {}
};
// `mid` is called once between the two phases (the latency form issues the scalar loads of the next layer's edges there: behind the last LDS
// read of the layer, so that they do not turn the partial lgkmcnt waits on the in-order LDS returns into waits for everything).
template <int D, bool FIRST, int PARTS = 1, typename MID = pk_no_hook>
__device__ __forceinline__ void update_rows_pk(int8_t* __restrict__ soft,
                                               uint32_t* __restrict__ c2v, // this lane's message dword of edges 0,1 of the layer
                                               const uint32_t* __restrict__ edges, // {shift, column*Z} per edge
                                               int l,
                                               int H,
                                               int Z,
                                               uint32_t base = 0,
                                               bool active = true,
                                               uint32_t* xch = nullptr,
                                               int part = 0,
                                               int xs = 0,
                                               MID mid = MID())
{
  constexpr bool SPLIT = PARTS > 1;
  s16x2    v2c[D], mabs[D];
  uint32_t adrA[D], adrB[D];
  int      rawA[D], rawB[D];
  uint32_t cw[(D + 1) / 2];
  P2_T(q0);
  __builtin_amdgcn_s_setprio(LDPC_PK_SETPRIO); // the address arithmetic and the LDS requests of a layer
  // Stage A: every address of the layer, then every LDS read of the layer in one go (2 soft bits per edge + the old
  // messages): the latency of the LDS pipe is paid once per layer instead of once per group of edges.
  // Both rows' addresses with packed 16-bit arithmetic: {l, l + H} + shift, wrap at Z by the unsigned minimum of p and p - Z, + column
  // offset -- four packed instructions for the two rows, then one mask and one shift to split the pair (LDS addresses stay below 2^16).
  {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 X  = {(unsigned short)l, (unsigned short)(l + H)};
    const u16x2 Zs = {(unsigned short)Z, (unsigned short)Z};
#pragma unroll
    for (int j = 0; j < D; ++j) {
#ifdef LDPC_PK_EMU_NOSLOAD // timing-only (WRONG results): no scalar loads of the edge table
      const unsigned short sh = (unsigned short)(7 * j + (l & 1)), co = (unsigned short)(j * Z);
#else
      const unsigned short sh = (unsigned short)edges[2 * j], co = (unsigned short)edges[2 * j + 1];
#endif
      const u16x2 T = X + u16x2{sh, sh};
      const u16x2 R = __builtin_elementwise_min(T, (u16x2)(T - Zs));
      const uint32_t P = __builtin_bit_cast(uint32_t, (u16x2)(R + u16x2{co, co}));
      adrA[j] = (P & 0xffffu) + base;
      adrB[j] = (P >> 16) + base;
    }
  }
  P2_T(q1);
#pragma unroll
  for (int j = 0; j < D; ++j) {
    rawA[j] = soft[adrA[j]];
    rawB[j] = soft[adrB[j]];
  }
  if (!FIRST) {
#pragma unroll
    for (int jj = 0; jj < (D + 1) / 2; ++jj)
      cw[jj] = c2v[64 * jj];
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(LDPC_PK_SETPRIO1);
  __builtin_amdgcn_sched_barrier(0);
  P2_T(q2);
  s16x2    mag1 = splat(LLR_MAX), mag2 = splat(LLR_MAX);
  uint32_t spx  = 0;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const s16x2 s = pk_pair(rawA[j], rawB[j]);
    // |s| > 120: infinite soft bit -> "infinite" message with the same sign: d != 0 only then, and 255 * d dominates.
    const s16x2 sc = pk_min(pk_max(s, splat(-LLR_MAX)), splat(LLR_MAX));
    const s16x2 d  = s - sc;
    s16x2       t  = sc;
    if (!FIRST) {
      const uint32_t w = cw[j >> 1];
      const s16x2    c = (j & 1) ? c2v_odd(w) : c2v_even(w);
      t                = pk_min(pk_max(sc - c, splat(-LLR_MAX)), splat(LLR_MAX));
    }
    const s16x2 v = d * splat(INF_MUL) + t;
    v2c[j]        = v;
    spx ^= as_u(v);
    const s16x2 av   = pk_max(v, -v);
    mabs[j]          = av;
    const s16x2 help = pk_max(mag1, av);
    mag1             = pk_min(mag1, av);
    mag2             = pk_min(mag2, help);
  }
  P2_T(q3);
#ifdef LDPC_PK_PROFILE2
  unsigned long long q4 = q3;
#endif
  if (SPLIT) {
    uint32_t* xw = xch + 3 * part * xs;
    xw[0] = as_u(mag1), xw[xs] = as_u(mag2), xw[2 * xs] = spx;
#ifndef LDPC_PK_EMU_NOXBAR // timing-only (WRONG results): no barrier in the exchange
    __syncthreads();
#endif
#ifdef LDPC_PK_PROFILE2
    q4 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int o = 1; o < PARTS; ++o) {
      const int       other = (part + o) & (PARTS - 1); // the merge is commutative: any order gives the same two minima
      const uint32_t* xr    = xch + 3 * other * xs;
      const s16x2     o1 = as_s2(xr[0]), o2 = as_s2(xr[xs]);
      spx ^= xr[2 * xs];
      mag2 = pk_min(pk_max(mag1, o1), pk_min(mag2, o2));
      mag1 = pk_min(mag1, o1);
    }
  }
  P2_T(q5);
  mid();
  __builtin_amdgcn_s_setprio(LDPC_PK_SETPRIO2); // the second phase: messages, soft-bit stores, then the layer barrier
  // Scaling by 0.8 = floor(x * 52428 / 65536), per row (avx2_support.h:65-106).
  const uint32_t s1A = ((uint32_t)(uint16_t)mag1.x * 52428u) >> 16, s1B = ((uint32_t)(uint16_t)mag1.y * 52428u) >> 16;
  const uint32_t s2A = ((uint32_t)(uint16_t)mag2.x * 52428u) >> 16, s2B = ((uint32_t)(uint16_t)mag2.y * 52428u) >> 16;
  // The product of ALL signs is folded into the two candidate magnitudes once per layer; an edge then only applies its own sign.
  const s16x2 pm  = pk_ashr15(as_s2(spx));
  const s16x2 s2u = as_s2(s2A | (s2B << 16));
  const s16x2 s1u = as_s2(s1A | (s1B << 16));
  const s16x2 s2p = as_s2(as_u(s2u) ^ as_u(pm)) - pm;
  const s16x2 dsp = (as_s2(as_u(s1u) ^ as_u(pm)) - pm) - s2p; // +-(min1 - min2), scaled
  s16x2 cprev = splat(0);
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const s16x2 v = v2c[j];
    // x = 0 where |v| == min1 (this edge provided the minimum, or ties it: then min1 == min2), 1 elsewhere
    const s16x2 x   = pk_min(mabs[j] - mag1, splat(1));
    const s16x2 mag = x * dsp + s2p;
    const s16x2 m   = pk_ashr15(v); // -1 where this edge's own message is negative
    const s16x2 c   = as_s2(as_u(mag) ^ as_u(m)) - m;
    const uint32_t r = as_u(pk_min(pk_max(c + v, splat(-LLR_INF)), splat(LLR_INF)));
    if (!SPLIT || active) {
      if (j & 1)
        c2v[64 * (j >> 1)] = c2v_pack(cprev, c);
      else if (j == D - 1)
        c2v[64 * (j >> 1)] = c2v_pack(c, c);
      soft[adrA[j]] = (int8_t)r;
      soft[adrB[j]] = (int8_t)(r >> 16);
    }
    cprev = c;
  }
  __builtin_amdgcn_s_setprio(LDPC_PK_SETPRIO1);
  P2_T(q6);
  P2_ADD(0, q0, q1); // scalar edge loads + address arithmetic
  P2_ADD(1, q1, q2); // LDS reads issued and returned
  P2_ADD(2, q2, q3); // phase 1
  P2_ADD(3, q3, q4); // exchange: stores + barrier
  P2_ADD(4, q4, q5); // exchange: loads + merge
  P2_ADD(5, q5, q6); // scaling + phase 2 + stores
  P2_ADD(6, q0, q6);
}

// A part without edges in this layer (a low-degree layer split four ways): it contributes the neutral partial result and takes part in
// the exchange barrier.
template <int PARTS, typename MID>
__device__ __forceinline__ void update_rows_pk_none(uint32_t* xch, int part, int xs, MID mid)
{
  uint32_t* xw = xch + 3 * part * xs;
  xw[0] = as_u(splat(LLR_MAX)), xw[xs] = as_u(splat(LLR_MAX)), xw[2 * xs] = 0u;
  __syncthreads();
  mid();
}

// The latency form: this part's edges of a layer of degree d. The ceil(d / 2) message pairs of the layer are dealt to the parts in
// contiguous runs (part i takes pairs [P i / PARTS, P (i + 1) / PARTS)), so every split falls on a message pair and the message layout
// is that of the throughput form. A part's share is at most PK_PART_EDGES edges (degree <= 19: 10 of two parts, 6 of four).
// Its {shift, column} words are fetched ONE LAYER AHEAD as scalar values (load_part_edges): with one codeblock per CU nothing hides the
// scalar-load round trip at the head of a layer (0.14 us per layer visit, measured by replacing the table with constants).
template <int PARTS>
struct part_edges {
  static constexpr int EMAX = PARTS == 4 ? 6 : 10;
  uint32_t             w[2 * EMAX]; // {shift, column * Z} of this part's edges (words behind its share belong to the next edges: unused)
  int                  da;          // edges of this part in the layer
  int                  p0;          // first message pair of this part in the layer
};

template <int PARTS>
__device__ __forceinline__ void load_part_edges(part_edges<PARTS>& pe, const uint32_t* __restrict__ edges_g, uint32_t li, int part)
{
  const int d = (int)((li >> 10) & 0x3fu), P = (d + 1) >> 1;
  const int p0 = (P * part) / PARTS, p1 = (P * (part + 1)) / PARTS;
  pe.p0 = p0, pe.da = min(2 * p1, d) - 2 * p0;
  const uint32_t* e = edges_g + 2 * ((int)(li & 0x3ffu) + 2 * p0); // (reads past a layer's or the table's last edge stay inside miphy_graph_tables)
#pragma unroll
  for (int k = 0; k < 2 * part_edges<PARTS>::EMAX; ++k)
    pe.w[k] = e[k];
}

template <bool FIRST, int PARTS, typename MID>
__device__ __forceinline__ void update_rows_pk_split(const part_edges<PARTS>& pe, int part, int8_t* soft, uint32_t* c2v, int l, int H, int Z, bool active,
                                                     uint32_t* xch, int xs, MID mid)
{
  c2v += 64 * pe.p0;
  switch (pe.da) {
#define PK_SPLIT_CASE(N)                                                                            \
  case N:                                                                                           \
    if (N <= part_edges<PARTS>::EMAX)                                                               \
      update_rows_pk<(N <= part_edges<PARTS>::EMAX ? N : 1), FIRST, PARTS, MID>(soft, c2v, pe.w, l, H, Z, 0, active, xch, part, xs, mid); \
    break;
    PK_SPLIT_CASE(10)
    PK_SPLIT_CASE(9)
    PK_SPLIT_CASE(6)
    PK_SPLIT_CASE(5)
    PK_SPLIT_CASE(4)
    PK_SPLIT_CASE(3)
    PK_SPLIT_CASE(2)
    PK_SPLIT_CASE(1)
    default:
      update_rows_pk_none<PARTS, MID>(xch, part, xs, mid);
      break;
#undef PK_SPLIT_CASE
  }
}

template <bool FIRST>
__device__ __forceinline__ void update_rows_pk_any(int d, int8_t* soft, uint32_t* c2v, const uint32_t* edges, int l, int H, int Z, uint32_t base = 0)
{
  switch (d) {
    case 19:
      update_rows_pk<19, FIRST>(soft, c2v, edges, l, H, Z, base);
      break;
    case 10:
      update_rows_pk<10, FIRST>(soft, c2v, edges, l, H, Z, base);
      break;
    case 9:
      update_rows_pk<9, FIRST>(soft, c2v, edges, l, H, Z, base);
      break;
    case 8:
      update_rows_pk<8, FIRST>(soft, c2v, edges, l, H, Z, base);
      break;
    case 7:
      update_rows_pk<7, FIRST>(soft, c2v, edges, l, H, Z, base);
      break;
    case 6:
      update_rows_pk<6, FIRST>(soft, c2v, edges, l, H, Z, base);
      break;
    case 5:
      update_rows_pk<5, FIRST>(soft, c2v, edges, l, H, Z, base);
      break;
    case 4:
      update_rows_pk<4, FIRST>(soft, c2v, edges, l, H, Z, base);
      break;
    default:
      update_rows_pk<3, FIRST>(soft, c2v, edges, l, H, Z, base);
      break;
  }
}

__device__ __forceinline__ uint32_t gf2_mulmod(uint32_t a, uint32_t b, uint32_t poly, uint32_t order)
{
  uint32_t       r   = 0;
  const uint32_t top = 1u << order;
  for (int k = (int)order - 1; k >= 0; --k) {
    r <<= 1;
    r ^= (r & top) ? poly : 0u;
    r ^= ((b >> k) & 1u) ? a : 0u;
  }
  return r;
}

__device__ __forceinline__ uint32_t hard_word(const int8_t* soft, int t, int K)
{
  const uint32_t* p = reinterpret_cast<const uint32_t*>(soft) + 8 * t;
  uint32_t        w = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const uint32_t x = p[q];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int v = (int8_t)(x >> (8 * b));
      w |= (uint32_t)(v <= 0) << (31 - (4 * q + b));
    }
  }
  const int rem = K - 32 * t;
  if (rem < 32)
    w &= (rem <= 0) ? 0u : (0xffffffffu << (32 - rem));
  return w;
}

// Hard-decision flags of the 32 soft bits of group t in the layout of crc_zmask: bit (q + 8 b) = (soft[32 t + 4 q + b] <= 0).
// Per dword of four soft bytes: bit 7 of a byte of ((x & 0x7f..) + 0x7f..) says "low seven bits non-zero"; the byte is <= 0 when
// its sign bit is set or that bit is clear.
__device__ __forceinline__ uint32_t hard_flags(const int8_t* soft, int t)
{
  const uint4* p  = reinterpret_cast<const uint4*>(soft) + 2 * t;
  const uint4  lo = p[0], hi = p[1];
  const uint32_t x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  uint32_t       w    = 0;
#pragma unroll
  for (int q = 7; q >= 0; --q) {
    const uint32_t nz  = (x[q] & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    const uint32_t le0 = (x[q] | ~nz) & 0x80808080u;
    w                  = (w << 1) | (le0 >> 7);
  }
  return w;
}

} // namespace
