// LDPC decoder, packed variant: every lane owns TWO lifted check rows (l and l + ceil(Z/2)) and processes them with the
// packed 16-bit VALU instructions of CDNA (v_pk_add/sub/min/max/ashr/lshl_i16), halving both the instruction count per
// codeblock and the number of wavefronts a codeblock occupies. Same arithmetic contract as ldpc_decode.hip (reference:
// ldpc_decoder_impl.cpp:60-146 + ldpc_decoder_avx2.cpp:66-243, avx2_support.h:65-106); results are bit-identical.
//
// Per (layer, lane) state, all fields duplicated for the two rows in the low / high half-word:
//   Wm  : [7:0] scaled min1, [15:8] scaled min2  |  same for row B in [31:16]
//   Wi0 : bit j (and 16 + j) = edge j is the argmin edge, j < 16 ; Wi1: edges 16.. (degree 19 rows only)
//   Ws0 : bit j (and 16 + j) = sign of the outgoing c2v message of edge j ; Ws1: edges 16..
// A mask for edge j is obtained for both rows at once with two packed shifts (lshl by 15-j, ashr by 15).
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
// Both half-words: bit `bit` -> all-ones / all-zeros mask.
// Hides how a lane mask was produced: otherwise LLVM rewrites "mask & x" into per-half compare + select, which has no
// packed form and costs 4-5 instructions instead of one.
__device__ __forceinline__ uint32_t opaque(uint32_t m)
{
  asm("" : "+v"(m));
  return m;
}
__device__ __forceinline__ uint32_t pk_bit_mask(uint32_t w, int bit)
{
  return opaque(as_u(pk_ashr15(as_s2(w) << splat(15 - bit))));
}

constexpr int LLR_MAX = 120;
constexpr int LLR_INF = 127;
constexpr int INF_INT = 255;

// 0/1 per half-word from bit `bit` (and 16 + bit) of w.
__device__ __forceinline__ uint32_t pk_bit01(uint32_t w, int bit)
{
  return (w >> bit) & 0x00010001u;
}
// a * b + c per half-word (low 16 bits).
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c)
{
  return as_u(as_s2(a) * as_s2(b) + as_s2(c));
}

template <int D, bool FIRST>
__device__ __forceinline__ void update_rows_pk(int8_t* __restrict__ soft,
                                               uint32_t&       Wm,
                                               uint32_t&       Wi0,
                                               uint32_t&       Wi1,
                                               uint32_t&       Ws0,
                                               uint32_t&       Ws1,
                                               const uint32_t* __restrict__ edges,
                                               int l,
                                               int H,
                                               int Z)
{
  uint32_t v2c[D];
  uint32_t adrA[D], adrB[D];
  s16x2    mag1 = splat(LLR_MAX), mag2 = splat(LLR_MAX);
  uint32_t spx  = 0;
  const uint32_t m1p = Wm & 0x00ff00ffu;
  const uint32_t dm  = as_u(as_s2((Wm >> 8) & 0x00ff00ffu) - as_s2(m1p));
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const uint32_t e  = edges[j];
    uint32_t       pA = (uint32_t)l + (e >> 16);
    pA                = min(pA, pA - (uint32_t)Z);
    uint32_t pB       = pA + (uint32_t)H;
    pB                = min(pB, pB - (uint32_t)Z);
    const uint32_t base = e & 0xffffu;
    const uint32_t aA = base + pA, aB = base + pB;
    adrA[j]           = aA;
    adrB[j]           = aB;
    // two sign-extended bytes -> one register with two int16 (bytes 1:0 of each source)
    const s16x2 s = as_s2(__builtin_amdgcn_perm((uint32_t)(int)soft[aB], (uint32_t)(int)soft[aA], 0x05040100u));
    s16x2       v;
    if (FIRST) {
      v = s;
    } else {
      const uint32_t im01 = (j < 16) ? pk_bit01(Wi0, j) : pk_bit01(Wi1, j - 16);
      const uint32_t mag  = pk_mad(im01, dm, m1p);                 // argmin edge ? min2 : min1
      const uint32_t s01  = (j < 16) ? pk_bit01(Ws0, j) : pk_bit01(Ws1, j - 16);
      const uint32_t f    = pk_mad(s01, as_u(splat(-2)), as_u(splat(1))); // +1 / -1
      const s16x2    c    = as_s2(mag) * as_s2(f);
      v                   = pk_min(pk_max(s - c, splat(-LLR_MAX)), splat(LLR_MAX));
    }
    // |s| > 120: infinite soft bit -> infinite message with the same sign (carried as +-INF_INT, see ldpc_decode.hip).
    const s16x2    as_   = pk_max(s, -s);
    const uint32_t infm  = opaque(as_u(pk_ashr15(splat(LLR_MAX) - as_)));
    const uint32_t vinf  = as_u(pk_ashr15(s)) ^ as_u(splat(INF_INT));
    const uint32_t vu    = (as_u(v) & ~infm) | (vinf & infm);
    v2c[j]               = vu;
    spx ^= vu;
    const s16x2 vv   = as_s2(vu);
    const s16x2 av   = pk_max(vv, -vv);
    const s16x2 help = pk_max(mag1, av);
    mag1             = pk_min(mag1, av);
    mag2             = pk_min(mag2, help);
  }
  // Scaling by 0.8 = floor(x * 52428 / 65536), per row (avx2_support.h:65-106).
  const uint32_t s1A = ((uint32_t)(uint16_t)mag1.x * 52428u) >> 16, s1B = ((uint32_t)(uint16_t)mag1.y * 52428u) >> 16;
  const uint32_t s2A = ((uint32_t)(uint16_t)mag2.x * 52428u) >> 16, s2B = ((uint32_t)(uint16_t)mag2.y * 52428u) >> 16;
  const uint32_t s1p = s1A | (s1B << 16);
  const uint32_t ds  = as_u(as_s2(s2A | (s2B << 16)) - as_s2(s1p));
  const uint32_t spm = spx & 0x80008000u;
  uint32_t       ni0 = 0, ni1 = 0, ns0 = 0, ns1 = 0;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const s16x2    v    = as_s2(v2c[j]);
    const s16x2    av   = pk_max(v, -v);
    // 1 where |v| == min1 (ties make min1 == min2): (|v| - min1 - 1) is negative only then
    const uint32_t im01 = (as_u((av - mag1) - splat(1)) >> 15) & 0x00010001u;
    const uint32_t mag  = pk_mad(im01, ds, s1p);
    const uint32_t s01  = ((v2c[j] ^ spm) >> 15) & 0x00010001u; // sign of the product of all other messages
    const uint32_t f    = pk_mad(s01, as_u(splat(-2)), as_u(splat(1)));
    const s16x2    c    = as_s2(mag) * as_s2(f);
    if (j < 16) {
      ni0 |= im01 << j;
      ns0 |= s01 << j;
    } else {
      ni1 |= im01 << (j - 16);
      ns1 |= s01 << (j - 16);
    }
    const uint32_t r = as_u(pk_min(pk_max(c + v, splat(-LLR_INF)), splat(LLR_INF)));
    soft[adrA[j]]    = (int8_t)r;
    soft[adrB[j]]    = (int8_t)(r >> 16);
  }
  Wm  = s1A | (s2A << 8) | (s1B << 16) | (s2B << 24);
  Wi0 = ni0;
  Ws0 = ns0;
  if (D > 16) {
    Wi1 = ni1;
    Ws1 = ns1;
  }
}

template <bool FIRST>
__device__ __forceinline__ void update_rows_pk_any(int d, int8_t* soft, uint32_t& Wm, uint32_t& Wi0, uint32_t& Wi1, uint32_t& Ws0, uint32_t& Ws1,
                                                   const uint32_t* edges, int l, int H, int Z)
{
  switch (d) {
    case 19:
      update_rows_pk<19, FIRST>(soft, Wm, Wi0, Wi1, Ws0, Ws1, edges, l, H, Z);
      break;
    case 10:
      update_rows_pk<10, FIRST>(soft, Wm, Wi0, Wi1, Ws0, Ws1, edges, l, H, Z);
      break;
    case 9:
      update_rows_pk<9, FIRST>(soft, Wm, Wi0, Wi1, Ws0, Ws1, edges, l, H, Z);
      break;
    case 8:
      update_rows_pk<8, FIRST>(soft, Wm, Wi0, Wi1, Ws0, Ws1, edges, l, H, Z);
      break;
    case 7:
      update_rows_pk<7, FIRST>(soft, Wm, Wi0, Wi1, Ws0, Ws1, edges, l, H, Z);
      break;
    case 6:
      update_rows_pk<6, FIRST>(soft, Wm, Wi0, Wi1, Ws0, Ws1, edges, l, H, Z);
      break;
    case 5:
      update_rows_pk<5, FIRST>(soft, Wm, Wi0, Wi1, Ws0, Ws1, edges, l, H, Z);
      break;
    case 4:
      update_rows_pk<4, FIRST>(soft, Wm, Wi0, Wi1, Ws0, Ws1, edges, l, H, Z);
      break;
    default:
      update_rows_pk<3, FIRST>(soft, Wm, Wi0, Wi1, Ws0, Ws1, edges, l, H, Z);
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

// CRC over the first L hard bits, words strided over the block's threads.
__device__ __forceinline__ uint32_t block_crc(const int8_t* soft, const miphy_graph_tables* __restrict__ tab, int crc_id, uint32_t poly,
                                              uint32_t order, int K, int L, uint32_t* red, int tid, int nt)
{
  const int      nfull = L >> 5, rbits = L & 31, nwords = (L + 31) >> 5;
  const uint32_t top   = 1u << order;
  uint32_t       part  = 0;
  for (int t = tid; t < nwords; t += nt) {
    const uint32_t w   = hard_word(soft, t, K);
    const int      len = min(32, L - 32 * t);
    uint32_t       reg = 0;
    for (int b = 0; b < len; ++b) {
      reg = (reg << 1) ^ (((w >> (31 - b)) & 1u) << order);
      reg ^= (reg & top) ? poly : 0u;
    }
    reg &= top - 1u;
    if (t < nfull) {
      reg = gf2_mulmod(reg, tab->crc_pow32[crc_id][nfull - 1 - t], poly, order);
      for (int b = 0; b < rbits; ++b) {
        reg <<= 1;
        reg ^= (reg & top) ? poly : 0u;
      }
    }
    part ^= reg;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
    part ^= __shfl_xor(part, off);
  if ((tid & 63) == 0)
    red[2 + (tid >> 6)] = part;
  __syncthreads();
  uint32_t crc = 0;
  for (int w = 0; w < (nt >> 6); ++w)
    crc ^= red[2 + w];
  __syncthreads();
  return crc;
}

#ifndef LDPC_PK_MIN_WAVES
#define LDPC_PK_MIN_WAVES 4
#endif
__global__ void __launch_bounds__(192, LDPC_PK_MIN_WAVES)
ldpc_decode_pk_kernel(const miphy_ldpc_dec_desc* __restrict__ descs,
                      const miphy_graph_tables* __restrict__ tab,
                      const int8_t* __restrict__ llr_base,
                      uint8_t* __restrict__ out_base,
                      int32_t* __restrict__ iters_out,
                      int max_nodes,
                      const uint32_t* __restrict__ harq_slot,
                      uint8_t* __restrict__ harq_crc_ok)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const miphy_ldpc_dec_desc dsc = descs[blockIdx.x];
  const int                 tid = threadIdx.x;
  const int                 nt  = blockDim.x;
  const int                 Z   = dsc.Z;
  const int                 H   = (Z + 1) >> 1;
  const int                 bgi = (dsc.bg == 1) ? 0 : 1;
  const int                 bgK = bgi ? 10 : 22;
  const int                 bgM = bgi ? 42 : 46;
  const int                 K   = bgK * Z;
  const int                 zp  = tab->z_pos[Z];

  // Soft bits live in a STATIC LDS array so that its base folds into the DS instructions' offset field (a dynamic base
  // costs one v_add per access); the state arrays stay dynamic (sized by the reachable layers).
  __shared__ __attribute__((aligned(16))) int8_t soft[68 * MIPHY_MAX_Z];
  const int lay_alloc  = min(bgM, max(4, max_nodes - bgK));
  const int soft_bytes = ((bgK + lay_alloc) * Z + 15) & ~15;
  // State arrays: 3 words per (layer, lane) + 2 extra words for the (at most 4) degree > 16 layers.
  uint32_t* st  = reinterpret_cast<uint32_t*>(smem);
  uint32_t* stx = st + 3 * lay_alloc * H;
  uint32_t* red = stx + 2 * 4 * H;

  if (harq_crc_ok && harq_crc_ok[harq_slot[blockIdx.x]]) {
    if (tid == 0)
      iters_out[blockIdx.x] = -1;
    return;
  }
  const int8_t* llr    = llr_base + dsc.llr_offset;
  uint8_t*      out    = out_base + dsc.out_offset;
  const int     in_len = (int)dsc.in_len;

  if (tid < 16)
    red[tid] = 0;
  for (int k = tid; k < 2 * Z; k += nt)
    soft[k] = 0;
  for (int k = 2 * Z + in_len + tid; k < soft_bytes; k += nt)
    soft[k] = 0;
  __syncthreads();
  int last = 0;
  if ((((uintptr_t)llr | (uintptr_t)(2 * Z)) & 15) == 0) {
    const uint4* src = reinterpret_cast<const uint4*>(llr);
    uint4*       dst = reinterpret_cast<uint4*>(soft + 2 * Z);
    const int    nq  = in_len >> 4;
    for (int q = tid; q < nq; q += nt) {
      const uint4 v = src[q];
      dst[q]        = v;
      int hi = -1;
      hi     = v.x ? 3 - (__clz((int)v.x) >> 3) : hi;
      hi     = v.y ? 7 - (__clz((int)v.y) >> 3) : hi;
      hi     = v.z ? 11 - (__clz((int)v.z) >> 3) : hi;
      hi     = v.w ? 15 - (__clz((int)v.w) >> 3) : hi;
      last   = (hi >= 0) ? 16 * q + hi + 1 : last;
    }
    for (int k = (nq << 4) + tid; k < in_len; k += nt) {
      const int8_t v  = llr[k];
      soft[2 * Z + k] = v;
      last            = (v != 0) ? k + 1 : last;
    }
  } else {
    for (int k = tid; k < in_len; k += nt) {
      const int8_t v  = llr[k];
      soft[2 * Z + k] = v;
      last            = (v != 0) ? k + 1 : last;
    }
  }
  atomicMax(reinterpret_cast<int*>(&red[0]), last);
  __syncthreads();
  last = (int)red[0];

  const bool use_crc = dsc.crc_poly != MIPHY_CRC_NONE;
  const int  kwords  = (K + 31) >> 5;
  if (last == 0) {
    if (!use_crc) {
      for (int b = tid; b < (K + 7) / 8; b += nt) {
        const int rem = K - 8 * b;
        out[b]        = (rem >= 8) ? 0xff : (uint8_t)(0xff << (8 - rem));
      }
    }
    if (tid == 0)
      iters_out[blockIdx.x] = 0;
    return;
  }
  int cb_len = max(last + 2 * Z, K + 4 * Z);
  cb_len     = ((cb_len + Z - 1) / Z) * Z;
  const int nof_layers = cb_len / Z - bgK;

  const uint32_t* edges_g   = tab->edge[bgi][zp];
  const uint16_t* row_start = tab->row_start[bgi];
  uint32_t        poly = 0, order = 0;
  int             L = 0;
  if (use_crc) {
    poly  = tab->crc_poly[dsc.crc_poly];
    order = tab->crc_order[dsc.crc_poly];
    L     = K - dsc.nof_filler_bits;
  }
  const bool final_only = use_crc && (dsc.flags & 1u);

  int       result_iters = 0;
  const int max_iter     = dsc.max_iter;
  for (int it = 0; it < max_iter; ++it) {
    for (int m = 0; m < nof_layers; ++m) {
      const int       e0    = row_start[m];
      const int       d     = row_start[m + 1] - e0;
      const uint32_t* edges = edges_g + e0;
      if (tid < H) {
        uint32_t  Wm = 0, Wi0 = 0, Wi1 = 0, Ws0 = 0, Ws1 = 0;
        uint32_t* sp = st + (3 * m) * H + tid;
        if (it == 0) {
          update_rows_pk_any<true>(d, soft, Wm, Wi0, Wi1, Ws0, Ws1, edges, tid, H, Z);
        } else {
          Wm  = sp[0];
          Wi0 = sp[H];
          Ws0 = sp[2 * H];
          if (d > 16) {
            Wi1 = stx[(2 * m) * H + tid];
            Ws1 = stx[(2 * m + 1) * H + tid];
          }
          update_rows_pk_any<false>(d, soft, Wm, Wi0, Wi1, Ws0, Ws1, edges, tid, H, Z);
        }
        sp[0]     = Wm;
        sp[H]     = Wi0;
        sp[2 * H] = Ws0;
        if (d > 16) {
          stx[(2 * m) * H + tid]     = Wi1;
          stx[(2 * m + 1) * H + tid] = Ws1;
        }
      }
      __syncthreads();
    }
    if (use_crc && !final_only) {
      if (block_crc(soft, tab, dsc.crc_poly, poly, order, K, L, red, tid, nt) == 0) {
        result_iters = it + 1;
        break;
      }
    }
  }
  if (final_only)
    result_iters = (block_crc(soft, tab, dsc.crc_poly, poly, order, K, L, red, tid, nt) == 0) ? max_iter : 0;

  for (int t = tid; t < kwords; t += nt) {
    const uint32_t w      = hard_word(soft, t, K);
    const int      nbytes = min(4, (K - 32 * t + 7) / 8);
    for (int q = 0; q < nbytes; ++q)
      out[4 * t + q] = (uint8_t)(w >> (24 - 8 * q));
  }
  if (tid == 0) {
    iters_out[blockIdx.x] = result_iters;
    if (harq_crc_ok && result_iters > 0)
      harq_crc_ok[harq_slot[blockIdx.x]] = 1;
  }
}

} // namespace

// LDS bytes the packed kernel needs for a given geometry (Zt >= Z of every codeblock, lay = layer bound, bgK).
size_t miphy_ldpc_pk_lds_bytes(int bgK, int lay, size_t Zt)
{
  const size_t H = (Zt + 1) / 2;
  (void)bgK; // the soft-bit array is static
  return (size_t)(3 * lay + 8) * H * 4 + 64;
}

int miphy_ldpc_pk_launch(const miphy_ldpc_dec_desc* d_descs, const miphy_graph_tables* tab, uint32_t n, int threads, size_t lds, const int8_t* llr,
                         uint8_t* out_bits, int32_t* iters, int nodes_all, const uint32_t* harq_slot, uint8_t* harq_crc_ok, hipStream_t s)
{
  static thread_local size_t lds_set = 0;
  if (lds > lds_set) {
    MIPHY_HIP_CHECK(hipFuncSetAttribute((const void*)ldpc_decode_pk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set = lds;
  }
  hipLaunchKernelGGL(ldpc_decode_pk_kernel, dim3(n), dim3(threads), lds, s, d_descs, tab, llr, out_bits, iters, nodes_all, harq_slot, harq_crc_ok);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
