// LDPC decoder -- layered normalised min-sum on int8 LLRs, one workgroup per codeblock.
//
// Behaviour contract: srsran::ldpc_decoder_impl::decode (lib/phy/upper/channel_coding/ldpc/ldpc_decoder_impl.cpp:60-146)
// with the arithmetic of the AVX2 hooks (ldpc_decoder_avx2.cpp:66-243, avx2_support.h:65-106).
//
// MI355X mapping (not a translation of the CPU data flow):
//   * thread i of the workgroup owns lifted check row i of every layer; a cyclic shift is an LDS address rotation,
//     so no data is ever moved to "rotate" a node;
//   * the soft bits of the whole codeblock (<= 68*384 B) live in LDS for the lifetime of the decode;
//   * check-to-variable messages are NOT stored: a check row's messages are fully determined by
//     (scaled min1, scaled min2, argmin, per-edge sign), which is packed in one 32-bit word per (layer,row)
//     (two words for the four degree-19 rows of BG1) -- exact, not an approximation;
//   * the v2c values of a row stay in registers between the min search and the soft-bit update (the per-degree
//     template makes every index static);
//   * hard decision + CRC run in-kernel: each lane reduces one 32-bit word of the message to a partial remainder,
//     multiplies it by x^(32k) mod P and the partial remainders are XOR-reduced with wavefront shuffles.
#include "miphy_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int LLR_MAX = 120;
constexpr int LLR_INF = 127;

// State word layout: [6:0] scaled min1, [13:7] scaled min2, [18:14] argmin edge, [31:19] c2v sign of edges 0..12.
// Second word (degree > 13 only): c2v sign of edges 13...
//
// Inside a row update an infinite LLR (|x| > 120, i.e. +-127 in memory) is carried as +-INF_INT so that the
// promotion rules of the reference (ldpc_decoder_avx2.cpp:85-105,205-243: "infinity is sticky", "|sum| > 120 becomes
// infinity") collapse into one clamp: c2v magnitudes are <= 95, so INF_INT + c2v always stays beyond +-120.
constexpr int INF_INT = 255;

template <int D, bool FIRST>
__device__ __forceinline__ void
update_row(int8_t* __restrict__ soft, uint32_t& w0, uint32_t& w1, const uint32_t* __restrict__ edges, int i, int Z)
{
  int v2c[D];
  int addr[D];
  int mag1 = LLR_MAX, mag2 = LLR_MAX; // running min / second min of |v2c|
  int spx  = 0;                       // XOR of all v2c values: bit 31 = sign product
  const int      old_m1  = w0 & 127;
  const int      old_m2  = (w0 >> 7) & 127;
  const int      old_arg = (w0 >> 14) & 31;
  const uint32_t old_sgn = (w0 >> 19) | (D > 13 ? (w1 << 13) : 0u);
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const uint32_t e   = edges[j];
    uint32_t       pos = (uint32_t)i + (e >> 16);
    pos                = min(pos, pos - (uint32_t)Z); // (i + shift) mod Z
    const int a        = (int)((e & 0xffffu) + pos);
    addr[j]            = a;
    const int s        = soft[a];
    int       v;
    if (FIRST) {
      v = s; // first visit of the layer: plain copy (ldpc_decoder_impl.cpp:181-185)
    } else {
      const int mag   = (old_arg == j) ? old_m2 : old_m1;
      const int smask = (int)__builtin_amdgcn_sbfe((int)old_sgn, j, 1); // 0 or -1
      const int c     = (mag ^ smask) - smask;
      v               = min(max(s - c, -LLR_MAX), LLR_MAX); // ldpc_decoder_avx2.cpp:85-92
    }
    // |s| > 120 <=> infinite soft bit: the message is infinite with the same sign (avx2.cpp:94-104).
    const bool inf = (uint32_t)(s + LLR_MAX) > (uint32_t)(2 * LLR_MAX);
    v              = inf ? ((s >> 31) ^ INF_INT) : v;
    v2c[j]         = v;
    spx ^= v;
    const int av   = max(v, -v);
    const int help = max(mag1, av); // strict "<" tie rule is value-equivalent: ties make min1 == min2
    mag1           = min(mag1, av);
    mag2           = min(mag2, help);
  }
  // Scaling by 0.8: floor(x * 52428 / 65536) (avx2_support.h:65-106).
  const int s1    = (mag1 * 52428) >> 16;
  const int s2    = (mag2 * 52428) >> 16;
  const int spm   = spx & (int)0x80000000;
  int       arg   = 0;
  uint32_t  cs    = 0;
#pragma unroll
  for (int j = D - 1; j >= 0; --j) {
    const int  v     = v2c[j];
    const bool ismin = max(v, -v) == mag1;
    const int  mag   = ismin ? s2 : s1;
    arg              = ismin ? j : arg;
    const int smask  = (v ^ spm) >> 31; // sign of the product of all other messages
    const int c      = (mag ^ smask) - smask;
    cs |= (uint32_t)(smask & 1) << j;
    const int r   = min(max(c + v, -LLR_INF), LLR_INF); // avx2.cpp:205-243 in the +-INF_INT encoding
    soft[addr[j]] = (int8_t)r;
  }
  w0 = (uint32_t)s1 | ((uint32_t)s2 << 7) | ((uint32_t)arg << 14) | (cs << 19);
  if (D > 13)
    w1 = cs >> 13;
}

template <bool FIRST>
__device__ __forceinline__ void update_row_any(int            d,
                                               int8_t*        soft,
                                               uint32_t&      w0,
                                               uint32_t&      w1,
                                               const uint32_t* edges,
                                               int            i,
                                               int            Z)
{
  switch (d) {
    case 19:
      update_row<19, FIRST>(soft, w0, w1, edges, i, Z);
      break;
    case 10:
      update_row<10, FIRST>(soft, w0, w1, edges, i, Z);
      break;
    case 9:
      update_row<9, FIRST>(soft, w0, w1, edges, i, Z);
      break;
    case 8:
      update_row<8, FIRST>(soft, w0, w1, edges, i, Z);
      break;
    case 7:
      update_row<7, FIRST>(soft, w0, w1, edges, i, Z);
      break;
    case 6:
      update_row<6, FIRST>(soft, w0, w1, edges, i, Z);
      break;
    case 5:
      update_row<5, FIRST>(soft, w0, w1, edges, i, Z);
      break;
    case 4:
      update_row<4, FIRST>(soft, w0, w1, edges, i, Z);
      break;
    default:
      update_row<3, FIRST>(soft, w0, w1, edges, i, Z);
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

// Hard decision of 32 consecutive soft bits starting at soft[32*t]; returns the big-endian numeric value (first bit in
// bit 31). Positions >= K are masked to 0. bit = (llr <= 0), log_likelihood_ratio.h:86.
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


// CRC of the first L hard bits of the message held in `soft` (every thread of the block must call). One 32-bit word of the
// message per lane: partial remainder, weight x^(32*(nfull-1-t)+rbits) mod P, XOR reduction over the block.
__device__ __forceinline__ uint32_t block_crc(const int8_t* soft, const miphy_graph_tables* __restrict__ tab, int crc_id, uint32_t poly,
                                              uint32_t order, int K, int L, uint32_t* red, int tid, int nt)
{
  const int kwords = (K + 31) >> 5, nfull = L >> 5, rbits = L & 31;
  uint32_t  part   = 0;
  if (tid < kwords && 32 * tid < L) {
    const uint32_t w   = hard_word(soft, tid, K);
    const int      len = min(32, L - 32 * tid);
    const uint32_t top = 1u << order;
    uint32_t       reg = 0;
    for (int b = 0; b < len; ++b) {
      reg = (reg << 1) ^ (((w >> (31 - b)) & 1u) << order);
      reg ^= (reg & top) ? poly : 0u;
    }
    reg &= top - 1u;
    if (tid < nfull) {
      reg = gf2_mulmod(reg, tab->crc_pow32[crc_id][nfull - 1 - tid], poly, order);
      for (int b = 0; b < rbits; ++b) {
        reg <<= 1;
        reg ^= (reg & top) ? poly : 0u;
      }
    }
    part = reg;
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

#ifndef LDPC_MIN_WAVES
#define LDPC_MIN_WAVES 1
#endif
__global__ void __launch_bounds__(MIPHY_MAX_Z, LDPC_MIN_WAVES)
ldpc_decode_kernel(const miphy_ldpc_dec_desc* __restrict__ descs,
                   const miphy_graph_tables* __restrict__ tab,
                   const int8_t* __restrict__ llr_base,
                   uint8_t* __restrict__ out_base,
                   int32_t* __restrict__ iters_out,
                   int max_nodes, // host bound on ceil((in_len + 2Z) / Z) over the batch
                   const uint32_t* __restrict__ harq_slot, // optional: per-descriptor codeblock slot in harq_crc_ok
                   uint8_t* __restrict__ harq_crc_ok,      // optional: skip codeblocks already decoded, flag new successes
                   const uint32_t* __restrict__ cb_order)     // optional: workgroup b decodes codeblock order[b] of the arrays
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t            cbx = cb_order ? cb_order[blockIdx.x] : blockIdx.x;
  const miphy_ldpc_dec_desc dsc = descs[cbx];
  const int                 tid = threadIdx.x;
  const int                 nt  = blockDim.x;
  const int                 Z   = dsc.Z;
  const int                 bgi = (dsc.bg == 1) ? 0 : 1;
  const int                 bgK = bgi ? 10 : 22;
  const int                 bgM = bgi ? 42 : 46;
  const int                 K   = bgK * Z;
  const int                 zp  = tab->z_pos[Z];

  int8_t*   soft = reinterpret_cast<int8_t*>(smem);
  const int lay_alloc  = min(bgM, max(4, max_nodes - bgK));
  const int soft_bytes = ((bgK + lay_alloc) * Z + 15) & ~15;
  // Check-row state is only allocated for the layers this launch can reach (host-side bound from in_len).
  uint32_t* st0  = reinterpret_cast<uint32_t*>(smem + soft_bytes);
  uint32_t* st1  = st0 + lay_alloc * Z;
  uint32_t* red  = st1 + 4 * Z; // 16 words of scratch

  const int8_t* llr = llr_base + dsc.llr_offset;
  uint8_t*      out = out_base + dsc.out_offset;
  const int     in_len = (int)dsc.in_len;

  if (harq_crc_ok && harq_crc_ok[harq_slot[cbx]]) { // pusch_decoder_impl.cpp:184: CRC already OK, keep the message
    if (tid == 0)
      iters_out[cbx] = -1;
    return;
  }
  if (tid < 16)
    red[tid] = 0;
  // Stage LLRs into LDS (variable nodes 0,1 are punctured -> 0) and find the last non-zero input.
  for (int k = tid; k < 2 * Z; k += nt)
    soft[k] = 0;
  for (int k = 2 * Z + in_len + tid; k < soft_bytes; k += nt)
    soft[k] = 0;
  __syncthreads();
  int last = 0;
  if ((((uintptr_t)llr | (uintptr_t)(2 * Z)) & 15) == 0) {
    // 16-byte coalesced path (the common case: Z multiple of 8, 16-byte aligned codeblock buffers).
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

  if (last == 0) { // ldpc_decoder_impl.cpp:88-94
    if (!use_crc) {
      for (int b = tid; b < (K + 7) / 8; b += nt) {
        const int rem = K - 8 * b;
        out[b]        = (rem >= 8) ? 0xff : (uint8_t)(0xff << (8 - rem));
      }
    }
    if (tid == 0)
      iters_out[cbx] = 0;
    return;
  }

  // ldpc_decoder_impl.cpp:101-114
  int cb_len = max(last + 2 * Z, K + 4 * Z);
  cb_len     = ((cb_len + Z - 1) / Z) * Z;
  const int nof_layers = cb_len / Z - bgK;

  const uint32_t* edges_g   = tab->edge[bgi][zp];
  const uint16_t* row_start = tab->row_start[bgi];

  // CRC constants.
  uint32_t poly = 0, order = 0;
  int      L = 0;
  if (use_crc) {
    poly  = tab->crc_poly[dsc.crc_poly];
    order = tab->crc_order[dsc.crc_poly];
    L     = K - dsc.nof_filler_bits; // ldpc_decoder_impl.cpp:55
  }
  const bool final_only = use_crc && (dsc.flags & 1u);

  int result_iters = 0;
  const int max_iter = dsc.max_iter;
  for (int it = 0; it < max_iter; ++it) {
    for (int m = 0; m < nof_layers; ++m) {
      const int       e0    = row_start[m];
      const int       d     = row_start[m + 1] - e0;
      const uint32_t* edges = edges_g + e0;
      if (tid < Z) {
        uint32_t w0 = 0, w1 = 0;
        if (it == 0) {
          update_row_any<true>(d, soft, w0, w1, edges, tid, Z);
        } else {
          w0 = st0[m * Z + tid];
          if (d > 13)
            w1 = st1[m * Z + tid];
          update_row_any<false>(d, soft, w0, w1, edges, tid, Z);
        }
        st0[m * Z + tid] = w0;
        if (d > 13)
          st1[m * Z + tid] = w1;
      }
      __syncthreads();
    }
    if (use_crc && !final_only) { // ldpc_decoder_impl.cpp:126-133
      if (block_crc(soft, tab, dsc.crc_poly, poly, order, K, L, red, tid, nt) == 0) {
        result_iters = it + 1;
        break;
      }
    }
  }
  if (final_only) // pusch_decoder_impl.cpp:105-118: decode without early stop, then check the CRC once
    result_iters = (block_crc(soft, tab, dsc.crc_poly, poly, order, K, L, red, tid, nt) == 0) ? max_iter : 0;

  // Final hard bits (identical to what the reference leaves in `output`: the bits of the last iteration run).
  if (tid < kwords) {
    const uint32_t w = hard_word(soft, tid, K);
    const int      nbytes = min(4, (K - 32 * tid + 7) / 8);
    for (int q = 0; q < nbytes; ++q)
      out[4 * tid + q] = (uint8_t)(w >> (24 - 8 * q));
  }
  if (tid == 0) {
    iters_out[cbx] = result_iters;
    if (harq_crc_ok && result_iters > 0)
      harq_crc_ok[harq_slot[cbx]] = 1;
  }
}

} // namespace


#ifndef LDPC_HYBRID_LDS_MARGIN
#define LDPC_HYBRID_LDS_MARGIN 2048
#endif
static bool     g_hybrid_msgs  = true; // A-B: miphy_debug_force_ldpc_kernel(mode | 0x100) = all messages of a GMSG launch in global memory
static int      g_force_kernel = 0; // 0 auto, 1 one-row-per-lane kernel, 2 packed kernel as ONE launch, 3 class-sorted launches (miphy_debug_force_ldpc_kernel)
static unsigned g_kernels_used = 0; // MIPHY_LDPC_KERNEL_* of every decoder launch since the last reset (miphy_debug_ldpc_kernels_used)

extern "C" void miphy_debug_force_ldpc_kernel(int mode)
{
  g_force_kernel = mode & 0xff;
  g_hybrid_msgs  = !(mode & 0x100);
}

bool miphy_ldpc_scalar_forced()
{
  return g_force_kernel == 1;
}

static int g_class_streams = 1 + MIPHY_NOF_SIDE_STREAMS; // streams the launch classes of a call are spread over (miphy_debug_set_ldpc_class_streams)

extern "C" void miphy_debug_set_ldpc_class_streams(int n)
{
  g_class_streams = n < 1 ? 1 : (n > 1 + MIPHY_NOF_SIDE_STREAMS ? 1 + MIPHY_NOF_SIDE_STREAMS : n);
}

extern "C" unsigned miphy_debug_ldpc_kernels_used(int reset)
{
  const unsigned m = g_kernels_used;
  if (reset)
    g_kernels_used = 0;
  return m;
}

// pusch_decoder_impl.cpp:146-149: the codeblock CRC flags of a new transmission start cleared.
__global__ void harq_flags_reset_kernel(const uint32_t* __restrict__ slots, uint32_t n, uint8_t* __restrict__ harq_crc_ok)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    harq_crc_ok[slots[i]] = 0;
}

int miphy_ldpc_flags_reset(const uint32_t* d_slots, uint32_t n, uint8_t* harq_crc_ok, hipStream_t s)
{
  if (!n || !d_slots || !harq_crc_ok)
    return MIPHY_OK;
  hipLaunchKernelGGL(harq_flags_reset_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_slots, n, harq_crc_ok);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

// ---- class-sorted launches ------------------------------------------------------------------------------------------------------
// The reference decoder scales its work with the lifting size of each codeblock (ldpc_decoder_avx2.cpp:59-64). A launch has ONE
// workgroup size and ONE LDS size, so a batch that mixes lifting sizes and code rates is sorted on the host into classes that share
// both, and every class gets a launch of its own:
//   * kind 0: Z <= 64 -- the wave kernel (ldpc_decode_pkw.hip), several codeblocks of one (base graph, Z) per wavefront;
//   * kind 1..3: the packed kernel with 1, 2 or 3 wavefronts per codeblock (Z <= 128 / 256 / 384);
//   * per base graph and per bucket of reachable layers (LDS per codeblock, hence codeblocks per CU, follows the layers);
//   * codeblocks the decoder can rate-dematch while it loads (first transmissions, see sch.hip) apart from those it cannot.
static int layer_bucket(int lay)
{
  static const int bound[6] = {4, 6, 10, 16, 26, 46};
  int              b        = 0;
  while (lay > bound[b])
    ++b;
  return b;
}

void miphy_ldpc_build_classes(const miphy_ldpc_dec_desc* descs, uint32_t n, const uint8_t* fusable, miphy_ldpc_classes& C)
{
  struct item {
    uint64_t key;
    uint32_t idx;
  };
  std::vector<item> it(n);
  for (uint32_t i = 0; i < n; ++i) {
    const miphy_ldpc_dec_desc& d   = descs[i];
    const int                  bgi = d.bg == 1 ? 0 : 1, bgK = bgi ? 10 : 22, bgM = bgi ? 42 : 46;
    const int                  nodes = (int)((d.in_len + 2u * d.Z + d.Z - 1u) / d.Z);
    const int                  lay   = std::min(bgM, std::max(4, nodes - bgK));
    const int                  H     = (d.Z + 1) / 2;
    const int                  kind  = d.Z <= 64 ? 0 : (H + 63) / 64;
    const bool                 fus   = fusable && fusable[i] && kind != 0;
    // class: fused | kind | base graph | layer bucket; below it the fields a bundle of the wave kernel must share; then the layer bound
    uint64_t key = ((uint64_t)(fus ? 1 : 0) << 63) | ((uint64_t)kind << 61) | ((uint64_t)bgi << 60) | ((uint64_t)layer_bucket(lay) << 57);
    if (kind == 0)
      key |= ((uint64_t)d.Z << 40) | ((uint64_t)(d.crc_poly & 0xff) << 32) | ((uint64_t)(d.flags & 1u) << 31) | ((uint64_t)d.max_iter << 8);
    key |= (uint64_t)lay; // 6 bits, read back below
    it[i].key = key, it[i].idx = i;
  }
  std::stable_sort(it.begin(), it.end(), [](const item& a, const item& b) { return (a.key >> 8) < (b.key >> 8); });
  C.order.resize(n);
  C.bundles.clear();
  C.classes.clear();
  C.nof_unfused = 0;
  C.identity    = true;
  const uint64_t class_mask = ~(uint64_t)0 << 57;
  for (uint32_t p = 0; p < n;) {
    uint32_t q = p;
    while (q < n && (it[q].key & class_mask) == (it[p].key & class_mask))
      ++q;
    miphy_ldpc_class c = {};
    c.fused            = (uint8_t)(it[p].key >> 63);
    c.kind             = (uint8_t)((it[p].key >> 61) & 3);
    c.bgi              = (uint8_t)((it[p].key >> 60) & 1);
    c.first = p, c.count = q - p;
    c.bundle_first = (uint32_t)C.bundles.size() / 2;
    const int bgK  = c.bgi ? 10 : 22;
    for (uint32_t k = p; k < q; ++k) {
      const miphy_ldpc_dec_desc& d = descs[it[k].idx];
      c.lay                        = std::max<uint8_t>(c.lay, (uint8_t)(it[k].key & 63));
      c.max_Z                      = std::max<uint16_t>(c.max_Z, d.Z);
      C.order[k]                   = it[k].idx;
      C.identity &= it[k].idx == k;
    }
    if (!c.fused)
      C.nof_unfused += c.count;
    if (c.kind == 0) {
      for (uint32_t k = p; k < q;) { // bundles: runs of identical (Z, CRC, mode, iterations), G codeblocks at most
        const miphy_ldpc_dec_desc& d = descs[it[k].idx];
        const uint32_t             G = 64u / ((d.Z + 1u) / 2u);
        uint32_t                   e = k;
        while (e < q && e - k < G && ((it[e].key ^ it[k].key) >> 8) == 0)
          ++e;
        C.bundles.push_back(k);
        C.bundles.push_back(e - k);
        const uint32_t sstride = (((uint32_t)(bgK + c.lay) * d.Z) + 15u) & ~15u;
        c.soft_total           = std::max(c.soft_total, G * sstride);
        k                      = e;
      }
    }
    c.bundle_count = (uint32_t)C.bundles.size() / 2 - c.bundle_first;
    C.classes.push_back(c);
    p = q;
  }
}

int miphy_ldpc_decode_classes_launch(miphy_ctx* ctx, const miphy_ldpc_dec_desc* d_descs, const miphy_ldpc_classes& C, const uint32_t* d_order,
                                     const uint32_t* d_bundles, const int8_t* llr, uint8_t* out_bits, int32_t* iters, const uint32_t* harq_slot,
                                     uint8_t* harq_crc_ok, hipStream_t s, const miphy_ldpc_rdm_desc* d_rdm, const int8_t* rm_in, bool allow_fuse)
{
  const size_t nc = C.classes.size();
  if (nc == 0)
    return MIPHY_OK;
  // Geometry of every class first: the launches of one call run side by side, so each needs message scratch of its own.
  struct geom {
    bool   fuse, gm;
    int    split; // parts of the latency form (0: throughput form)
    int    threads, pairs, lds_pairs;
    size_t lds, gmsg_bytes, gmsg_off;
    int    stream; // 0 = the caller's, 1 .. = side streams
  };
  std::vector<geom> g(nc);
  size_t            gmsg_total = 0;
  for (size_t i = 0; i < nc; ++i) {
    const miphy_ldpc_class& c = C.classes[i];
    geom&                   q = g[i];
    q = geom{};
    const int bgK = c.bgi ? 10 : 22;
    if (g_force_kernel == 1)
      continue;
    if (c.kind == 0) {
      q.gmsg_bytes = miphy_ldpc_pkw_gmsg_bytes(ctx, c.bundle_count, c.bgi, c.lay, c.soft_total, g_force_kernel == 4);
    } else {
      q.threads = 64 * c.kind;
      q.pairs   = ctx->h_tables->pair_start[c.bgi][c.lay];
      q.fuse    = c.fused && allow_fuse && d_rdm;
      const size_t lds_l = miphy_ldpc_pk_lds_bytes(bgK, c.lay, c.max_Z, q.pairs), lds_g = miphy_ldpc_pk_lds_bytes(bgK, c.lay, c.max_Z, 0);
      // messages in LDS while that keeps as many codeblocks resident per CU as the registers allow; otherwise in global memory
      auto per_cu = [&](size_t lds) { return std::max(1, std::min((int)((size_t)160 * 1024 / lds), miphy_ldpc_pk_waves_per_cu(q.fuse) / (int)c.kind)); };
      // (and only where the class has more codeblocks than stay resident with the messages in LDS: otherwise the global round trip per
      // layer visit buys nothing)
      q.gm         = per_cu(lds_g) > per_cu(lds_l) && (g_force_kernel == 4 || c.count > (uint32_t)(ctx->num_cus * per_cu(lds_l)));
      q.lds        = q.gm ? lds_g : lds_l;
      // Only as many layers' messages leave LDS as that residency needs: the first layers keep theirs (a lane's messages are private to it,
      // so the split is free), the global round trip and its L2 traffic are paid for the rest.
      if (q.gm && g_force_kernel != 4 && g_hybrid_msgs) {
        for (int k = c.lay - 1; k > 0; --k) {
          const int    pk_ = ctx->h_tables->pair_start[c.bgi][k];
          const size_t l_  = miphy_ldpc_pk_lds_bytes(bgK, c.lay, c.max_Z, pk_);
          if (per_cu(l_ + LDPC_HYBRID_LDS_MARGIN) == per_cu(lds_g)) { // (margin: a CU filled to the last byte of the sum held one workgroup fewer -- allocation granularity)
            q.lds_pairs = pk_, q.lds = l_;
            break;
          }
        }
      }
      // Latency form where the class cannot fill the chip anyway (at most one codeblock per CU): twice the wavefronts per codeblock,
      // messages in LDS (residency is no concern then).
      // (four parts while the codeblocks of the class still find a CU each and the workgroup stays within 1024 threads; forced mode 6: two)
      const int    parts = (g_force_kernel == 5 || g_force_kernel == 6) ? 2 : 4;
      const size_t lds_s = miphy_ldpc_pk_lds_bytes(bgK, c.lay, c.max_Z, q.pairs, parts);
      q.split            = (g_force_kernel != 4 && (c.count <= (uint32_t)ctx->num_cus || g_force_kernel == 5) && lds_s <= (size_t)160 * 1024) ? parts : 0;
      if (q.split)
        q.gm = false, q.lds = lds_s;
      q.gmsg_bytes = miphy_ldpc_pk_gmsg_bytes(ctx, c.count, q.threads, q.lds, q.fuse, q.gm ? q.pairs - q.lds_pairs : 0);
    }
    q.gmsg_off = gmsg_total;
    gmsg_total += (q.gmsg_bytes + 255) & ~(size_t)255;
  }
  uint8_t* gmsg_base = nullptr;
  int      rc;
  if (gmsg_total) {
    void* w = nullptr;
    if ((rc = miphy_get_workspace(ctx, gmsg_total, s, &w, 3)))
      return rc;
    gmsg_base = (uint8_t*)w;
  }
  // Classes to streams. A class whose whole grid is a fraction of the chip (at most four wavefronts per CU: a few hundred small
  // codeblocks) is a latency chain -- one after another such classes each cost their full latency with the chip idle, next to a
  // large class they cost nothing: they go to the side streams, round robin. The large classes stay on the caller's stream one
  // after another: two chip-filling persistent grids side by side was measured slower and erratic (4.6 ms in sequence, 4.1 to 6.7
  // side by side on the mixed slot of bench.py), each holds the registers and LDS the other was tuned to have.
  int nstreams = 1;
  if (nc > 1 && g_class_streams > 1) {
    int small = 0;
    for (size_t i = 0; i < nc; ++i) {
      const miphy_ldpc_class& c = C.classes[i];
      const uint64_t waves = c.kind == 0 ? c.bundle_count : (uint64_t)c.count * c.kind * (g[i].split ? g[i].split : 1);
      if (waves <= (uint64_t)ctx->num_cus * 4)
        g[i].stream = 1 + (small++ % (g_class_streams - 1));
    }
    if (small == (int)nc) // nothing large: the first small class takes the caller's stream
      g[0].stream = 0;
    if (small > 0)
      nstreams = g_class_streams;
  }
  if (nstreams > 1) {
    if ((rc = miphy_side_streams(ctx)))
      return rc;
    MIPHY_HIP_CHECK(hipEventRecord((hipEvent_t)ctx->ev_fork, s));
    for (int k = 0; k < MIPHY_NOF_SIDE_STREAMS; ++k)
      MIPHY_HIP_CHECK(hipStreamWaitEvent((hipStream_t)ctx->side_stream[k], (hipEvent_t)ctx->ev_fork, 0));
  }
  for (size_t k2 = 0; k2 < 2 * nc; ++k2) { // the side-stream classes first: they start while the large ones are being enqueued
    const size_t i = k2 % nc;
    if ((g[i].stream != 0) != (k2 < nc))
      continue;
    const miphy_ldpc_class& c  = C.classes[i];
    const geom&             q  = g[i];
    hipStream_t             st = q.stream == 0 ? s : (hipStream_t)ctx->side_stream[q.stream - 1];
    void*                   gb = q.gmsg_bytes ? gmsg_base + q.gmsg_off : nullptr;
    if (g_force_kernel == 1) { // A-B knob: the one-row-per-lane kernel on every class (the caller has dematched: allow_fuse is false then)
      const int    bgK = c.bgi ? 10 : 22, threads = ((c.max_Z + 63) / 64) * 64;
      const size_t lds = ((((size_t)bgK + c.lay) * threads + 15) & ~(size_t)15) + (size_t)(c.lay + 4) * threads * 4 + 64;
      if (lds > 48 * 1024)
        MIPHY_HIP_CHECK(hipFuncSetAttribute((const void*)ldpc_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(ldpc_decode_kernel, dim3(c.count), dim3(threads), lds, st, d_descs, ctx->d_tables, llr, out_bits, iters, bgK + c.lay, harq_slot,
                         harq_crc_ok, d_order + c.first);
      MIPHY_HIP_CHECK(hipGetLastError());
      g_kernels_used |= MIPHY_LDPC_KERNEL_SCALAR;
      continue;
    }
    if (c.kind == 0) {
      int gm = 0;
      if ((rc = miphy_ldpc_pkw_launch(ctx, d_descs, d_order, d_bundles + 2 * (size_t)c.bundle_first, c.bundle_count, c.bgi, c.lay, c.soft_total, llr, out_bits,
                                      iters, harq_slot, harq_crc_ok, st, &gm, gb, g_force_kernel == 4)))
        return rc;
      g_kernels_used |= MIPHY_LDPC_KERNEL_WAVE | (gm ? MIPHY_LDPC_KERNEL_GMSG : 0u);
      continue;
    }
    const int bgK = c.bgi ? 10 : 22;
    if ((rc = miphy_ldpc_pk_launch(ctx, d_descs, c.count, q.threads, q.lds, llr, out_bits, iters, bgK + c.lay, harq_slot, harq_crc_ok, st,
                                   q.fuse ? d_rdm : nullptr, q.fuse ? rm_in : nullptr, q.gm ? q.pairs - q.lds_pairs : 0,
                                   (C.identity && nc == 1) ? nullptr : d_order + c.first, gb, q.split, q.gm ? q.lds_pairs : 0)))
      return rc;
    g_kernels_used |= MIPHY_LDPC_KERNEL_PACKED | (q.fuse ? MIPHY_LDPC_KERNEL_FUSED : 0u) | (q.gm ? MIPHY_LDPC_KERNEL_GMSG : 0u) |
                      (q.split ? MIPHY_LDPC_KERNEL_SPLIT : 0u) | ((q.gm && q.lds_pairs > 0) ? MIPHY_LDPC_KERNEL_GMSG_PART : 0u);
  }
  if (nstreams > 1) {
    for (int k = 0; k < MIPHY_NOF_SIDE_STREAMS; ++k) {
      MIPHY_HIP_CHECK(hipEventRecord((hipEvent_t)ctx->ev_join[k], (hipStream_t)ctx->side_stream[k]));
      MIPHY_HIP_CHECK(hipStreamWaitEvent(s, (hipEvent_t)ctx->ev_join[k], 0));
    }
  }
  return MIPHY_OK;
}

// ---- one launch for the whole batch (device-resident descriptors, which the host cannot sort; forced kernels of the A-B knob) -------
int miphy_ldpc_decode_launch(miphy_ctx*                   ctx,
                             const miphy_ldpc_dec_desc*   descs,
                             int                          descs_on_device,
                             uint32_t                     n,
                             const int8_t*                llr,
                             uint8_t*                     out_bits,
                             int32_t*                     iters,
                             const miphy_ldpc_dec_limits* limits,
                             const uint32_t*              harq_slot,
                             uint8_t*                     harq_crc_ok,
                             void*                        stream,
                             const miphy_ldpc_rdm_desc*   fuse_rdm,
                             const int8_t*                fuse_in,
                             const miphy_ldpc_rdm_limits* fuse_rlim,
                             int                          bg_mask,
                             const uint32_t*              reset_slots,
                             uint32_t                     nof_reset_slots)
{
  MIPHY_REQUIRE(ctx && descs && llr && out_bits && iters, "ldpc_decode: null argument");
  if (n == 0)
    return MIPHY_OK;
  hipStream_t s = (hipStream_t)stream;
  // Launch geometry: threads from the largest Z, LDS from the largest number of layers any codeblock can reach
  // (nof_layers <= ceil((in_len + 2Z)/Z) - bg_K, ldpc_decoder_impl.cpp:101-114). Host descriptors are validated here
  // (the reference asserts the same conditions, ldpc_decoder_impl.cpp:66-84); for device descriptors the caller
  // vouches for validity and may pass `limits` (worst case assumed otherwise).
  int    max_threads = 64;
  int    max_nodes[2] = {0, 0}; // per base graph: largest ceil((in_len + 2Z) / Z)
  auto   account      = [&](unsigned bg, unsigned Z, unsigned in_len) {
    const int nodes   = (int)((in_len + 2 * Z + Z - 1) / Z);
    const int threads = ((Z + 63) / 64) * 64;
    max_threads       = threads > max_threads ? threads : max_threads;
    max_nodes[bg - 1] = nodes > max_nodes[bg - 1] ? nodes : max_nodes[bg - 1];
  };
  if (!descs_on_device) {
    for (uint32_t i = 0; i < n; ++i) {
      const miphy_ldpc_dec_desc& d = descs[i];
      MIPHY_REQUIRE(d.bg == 1 || d.bg == 2, "ldpc_decode: desc %u: invalid base graph %u", i, d.bg);
      MIPHY_REQUIRE(d.Z <= MIPHY_MAX_Z && ctx->h_tables->z_pos[d.Z] != 0xffff, "ldpc_decode: desc %u: invalid lifting size %u", i, d.Z);
      const unsigned bgK = (d.bg == 1) ? 22 : 10, nshort = (d.bg == 1) ? 66 : 50;
      MIPHY_REQUIRE(d.in_len >= (bgK + 2) * d.Z && d.in_len <= nshort * d.Z, "ldpc_decode: desc %u: input length %u out of range", i, d.in_len);
      MIPHY_REQUIRE(d.max_iter > 0, "ldpc_decode: desc %u: max_iter must be > 0", i);
      MIPHY_REQUIRE(d.crc_poly == MIPHY_CRC_NONE || d.crc_poly <= MIPHY_CRC11, "ldpc_decode: desc %u: invalid CRC", i);
      MIPHY_REQUIRE(d.nof_filler_bits < bgK * d.Z, "ldpc_decode: desc %u: invalid number of filler bits", i);
      account(d.bg, d.Z, d.in_len);
    }
    if (g_force_kernel == 0 || g_force_kernel >= 3) {
      // host descriptors: sorted into launch classes (nothing is dematched by the decoder on this path: fuse_rdm comes with device
      // descriptors only)
      miphy_ldpc_classes C;
      miphy_ldpc_build_classes(descs, n, nullptr, C);
      const void *d_descs = nullptr, *d_order = nullptr, *d_bundles = nullptr;
      int         rc;
      if ((rc = miphy_stage_descs(ctx, descs, 0, sizeof(miphy_ldpc_dec_desc) * (size_t)n, s, &d_descs)))
        return rc;
      if ((rc = miphy_stage_descs(ctx, C.order.data(), 0, sizeof(uint32_t) * (size_t)n, s, &d_order)))
        return rc;
      if (!C.bundles.empty() && (rc = miphy_stage_descs(ctx, C.bundles.data(), 0, sizeof(uint32_t) * C.bundles.size(), s, &d_bundles)))
        return rc;
      if ((rc = miphy_ldpc_flags_reset(reset_slots, nof_reset_slots, harq_crc_ok, s)))
        return rc;
      return miphy_ldpc_decode_classes_launch(ctx, (const miphy_ldpc_dec_desc*)d_descs, C, (const uint32_t*)d_order, (const uint32_t*)d_bundles, llr, out_bits,
                                              iters, harq_slot, harq_crc_ok, s, nullptr, nullptr, false);
    }
  } else if (limits) {
    MIPHY_REQUIRE(limits->max_Z >= 2 && limits->max_Z <= MIPHY_MAX_Z, "ldpc_decode: limits: invalid max_Z");
    // the base graph of device descriptors is not visible here: size for those the caller names (bg_mask, default both)
    if (bg_mask & 1)
      account(1, limits->max_Z, limits->max_in_len);
    if (bg_mask & 2)
      account(2, limits->max_Z, limits->max_in_len);
  } else {
    account(1, MIPHY_MAX_Z, 66 * MIPHY_MAX_Z);
    account(2, MIPHY_MAX_Z, 50 * MIPHY_MAX_Z);
  }
  const int nodes_all = max_nodes[0] > max_nodes[1] ? max_nodes[0] : max_nodes[1];
  size_t    max_lds   = 0;
  for (int b = 0; b < 2; ++b) {
    if (!max_nodes[b])
      continue;
    // The kernel sizes its arrays from the batch-wide node bound (same formula as below).
    const int    bgK = b ? 10 : 22, bgM = b ? 42 : 46;
    int          lay = nodes_all - bgK;
    lay              = lay < 4 ? 4 : (lay > bgM ? bgM : lay);
    const size_t Zt  = (size_t)max_threads; // >= max Z
    const size_t lds = (((bgK + lay) * Zt + 15) & ~(size_t)15) + (size_t)(lay + 4) * Zt * 4 + 64;
    max_lds          = lds > max_lds ? lds : max_lds;
  }
  const void* d_descs = nullptr;
  int         rc      = miphy_stage_descs(ctx, descs, descs_on_device, sizeof(miphy_ldpc_dec_desc) * (size_t)n, s, &d_descs);
  if (rc)
    return rc;
  // Kernel choice. The packed kernel (two check rows per lane, explicit messages in LDS) executes ~1.6x fewer instructions per
  // row but needs more LDS per codeblock; at low code rates / mid lifting sizes that leaves one small workgroup per CU, and the
  // one-row-per-lane kernel (compressed messages, twice the wavefronts) wins. Both are scored by the check rows a CU holds in
  // flight (workgroups per CU limited by LDS, wavefront slots and registers), the packed one weighted by its instruction advantage;
  // measured crossovers: tools/ldpc_rate_sweep.py. miphy_debug_force_ldpc_kernel() overrides (parity tests run both kernels).
  // Any lifting size is legal in the packed kernel (an odd one folds its unpaired last row onto itself, ldpc_decode_pk.hip).
  const bool         pk_ok = !descs_on_device || limits;
  const int          pk_threads = ((max_threads / 2 + 63) / 64) * 64; // max_threads >= max Z, a multiple of 64
  size_t             pk_lds     = 0, pk_lds_g = 0;                    // with the messages in LDS / in global memory
  int                pk_pairs   = 0;
  if (pk_ok) {
    for (int b = 0; b < 2; ++b) {
      if (!max_nodes[b])
        continue;
      const int bgK = b ? 10 : 22, bgM = b ? 42 : 46;
      int       lay = nodes_all - bgK;
      lay           = lay < 4 ? 4 : (lay > bgM ? bgM : lay);
      const int    pairs = ctx->h_tables->pair_start[b][lay];
      const size_t l     = miphy_ldpc_pk_lds_bytes(bgK, lay, (size_t)max_threads, pairs);
      const size_t lg    = miphy_ldpc_pk_lds_bytes(bgK, lay, (size_t)max_threads, 0);
      pk_lds             = l > pk_lds ? l : pk_lds;
      pk_lds_g           = lg > pk_lds_g ? lg : pk_lds_g;
      pk_pairs           = pairs > pk_pairs ? pairs : pk_pairs;
    }
  }
  // Messages in LDS while that keeps as many codeblocks resident per CU as the registers allow (3 wavefronts per SIMD); otherwise
  // (more than ~6 layers at Z = 384) in global memory, where they cost an L2 round trip per layer visit but leave room for four
  // codeblocks per CU: measured 2.0x at rate 1/3, tools/ldpc_rate_sweep.py.
  const int  pk_waves   = pk_threads / 64;
  const bool will_fuse  = fuse_rdm && ((uintptr_t)llr & 15) == 0;
  auto       pk_per_cu  = [&](size_t lds) { return std::max(1, std::min((int)((size_t)160 * 1024 / (lds ? lds : 1)), miphy_ldpc_pk_waves_per_cu(will_fuse) / pk_waves)); };
  const bool pk_gmsg    = pk_ok && pk_per_cu(pk_lds_g) > pk_per_cu(pk_lds) && (g_force_kernel >= 2 || n > (uint32_t)(ctx->num_cus * pk_per_cu(pk_lds)));
  if (pk_gmsg)
    pk_lds = pk_lds_g;
  auto rows_in_flight = [](size_t lds, int threads, int waves_per_simd_by_regs, int rows_per_lane) {
    const int waves = threads / 64;
    int       wgs   = (int)((size_t)160 * 1024 / (lds ? lds : 1));
    wgs             = std::min(wgs, 32 / waves);                          // 8 wavefront slots per SIMD
    wgs             = std::min(wgs, 4 * waves_per_simd_by_regs / waves);  // register file
    wgs             = std::max(wgs, 1);
    return wgs * threads * rows_per_lane;
  };
  bool use_pk = pk_ok && 1.6 * rows_in_flight(pk_lds, pk_threads, 4, 2) >= 1.0 * rows_in_flight(max_lds, max_threads, 8, 1);
  if (g_force_kernel == 1)
    use_pk = false;
  if (g_force_kernel >= 2)
    use_pk = pk_ok;
  // The fused form needs 16-byte aligned soft buffers (its write-back is vectorised) and lifting sizes that are multiples of 16 (the
  // caller vouches for that when it passes fuse_rdm); otherwise the dematcher runs on its own.
  const bool fuse = fuse_rdm && use_pk && ((uintptr_t)llr & 15) == 0;
  if (!fuse && (rc = miphy_ldpc_flags_reset(reset_slots, nof_reset_slots, harq_crc_ok, s)))
    return rc;
  if (fuse_rdm && !fuse) {
    if ((rc = miphy_ldpc_rate_dematch_batch(ctx, fuse_rdm, 1, n, fuse_in, const_cast<int8_t*>(llr), fuse_rlim, s)))
      return rc;
  }
  if (use_pk) {
    g_kernels_used |= MIPHY_LDPC_KERNEL_PACKED | (fuse ? MIPHY_LDPC_KERNEL_FUSED : 0u) | (pk_gmsg ? MIPHY_LDPC_KERNEL_GMSG : 0u);
    return miphy_ldpc_pk_launch(ctx, (const miphy_ldpc_dec_desc*)d_descs, n, pk_threads, pk_lds, llr, out_bits, iters, nodes_all, harq_slot,
                                harq_crc_ok, s, fuse ? fuse_rdm : nullptr, fuse ? fuse_in : nullptr, pk_gmsg ? pk_pairs : 0);
  }
  // Above the default 64 KB of dynamic LDS the limit has to be raised; it is a per-device attribute of the kernel, so it is set on
  // every such launch (a cache per thread would be wrong for a thread that drives several devices).
  if (max_lds > 48 * 1024) {
    MIPHY_HIP_CHECK(hipFuncSetAttribute((const void*)ldpc_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds));
  }
  g_kernels_used |= MIPHY_LDPC_KERNEL_SCALAR;
  hipLaunchKernelGGL(ldpc_decode_kernel, dim3(n), dim3(max_threads), max_lds, s, (const miphy_ldpc_dec_desc*)d_descs, ctx->d_tables, llr, out_bits, iters, nodes_all, harq_slot, harq_crc_ok, (const uint32_t*)nullptr);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

extern "C" int miphy_ldpc_decode_batch(miphy_ctx*                   ctx,
                                       const miphy_ldpc_dec_desc*   descs,
                                       int                          descs_on_device,
                                       uint32_t                     n,
                                       const int8_t*                llr,
                                       uint8_t*                     out_bits,
                                       int32_t*                     iters,
                                       const miphy_ldpc_dec_limits* limits,
                                       void*                        stream)
{
  return miphy_ldpc_decode_launch(ctx, descs, descs_on_device, n, llr, out_bits, iters, limits, nullptr, nullptr, stream);
}

// ---- prepared form: descriptors validated, sorted into launch classes and uploaded once; every run is launches only -------------------
struct miphy_ldpc_decode_plan {
  miphy_ctx*         ctx;
  uint32_t           n;
  miphy_ldpc_classes cls; // classes only (order / bundles live in d_buf)
  void*              d_buf;
  const miphy_ldpc_dec_desc* d_descs;
  const uint32_t*    d_order;
  const uint32_t*    d_bundles;
};

extern "C" int miphy_ldpc_decode_plan_create(miphy_ctx* ctx, const miphy_ldpc_dec_desc* descs, uint32_t n, miphy_ldpc_decode_plan** out)
{
  MIPHY_REQUIRE(ctx && descs && out && n > 0, "miphy_ldpc_decode_plan_create: null argument or empty batch");
  for (uint32_t i = 0; i < n; ++i) {
    const miphy_ldpc_dec_desc& d = descs[i];
    MIPHY_REQUIRE(d.bg == 1 || d.bg == 2, "ldpc_decode: desc %u: invalid base graph %u", i, d.bg);
    MIPHY_REQUIRE(d.Z <= MIPHY_MAX_Z && ctx->h_tables->z_pos[d.Z] != 0xffff, "ldpc_decode: desc %u: invalid lifting size %u", i, d.Z);
    const unsigned bgK = (d.bg == 1) ? 22 : 10, nshort = (d.bg == 1) ? 66 : 50;
    MIPHY_REQUIRE(d.in_len >= (bgK + 2) * d.Z && d.in_len <= nshort * d.Z, "ldpc_decode: desc %u: input length %u out of range", i, d.in_len);
    MIPHY_REQUIRE(d.max_iter > 0, "ldpc_decode: desc %u: max_iter must be > 0", i);
    MIPHY_REQUIRE(d.crc_poly == MIPHY_CRC_NONE || d.crc_poly <= MIPHY_CRC11, "ldpc_decode: desc %u: invalid CRC", i);
    MIPHY_REQUIRE(d.nof_filler_bits < bgK * d.Z, "ldpc_decode: desc %u: invalid number of filler bits", i);
  }
  auto* p = new miphy_ldpc_decode_plan();
  p->ctx = ctx, p->n = n, p->d_buf = nullptr;
  miphy_ldpc_build_classes(descs, n, nullptr, p->cls);
  const size_t b0 = sizeof(miphy_ldpc_dec_desc) * (size_t)n, b1 = 4 * (size_t)n, b2 = 4 * p->cls.bundles.size();
  std::vector<uint8_t> host(b0 + b1 + b2);
  memcpy(host.data(), descs, b0);
  memcpy(host.data() + b0, p->cls.order.data(), b1);
  if (b2)
    memcpy(host.data() + b0 + b1, p->cls.bundles.data(), b2);
  hipError_t e = hipMalloc(&p->d_buf, host.size());
  if (e == hipSuccess)
    e = hipMemcpy(p->d_buf, host.data(), host.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    miphy_set_error("miphy_ldpc_decode_plan_create: %s", hipGetErrorString(e));
    if (p->d_buf)
      (void)hipFree(p->d_buf);
    delete p;
    return MIPHY_EHIP;
  }
  p->d_descs   = reinterpret_cast<const miphy_ldpc_dec_desc*>(p->d_buf);
  p->d_order   = reinterpret_cast<const uint32_t*>((uint8_t*)p->d_buf + b0);
  p->d_bundles = reinterpret_cast<const uint32_t*>((uint8_t*)p->d_buf + b0 + b1);
  std::vector<uint32_t>().swap(p->cls.order);
  std::vector<uint32_t>().swap(p->cls.bundles);
  *out = p;
  return MIPHY_OK;
}

extern "C" int miphy_ldpc_decode_plan_run(miphy_ldpc_decode_plan* p, const int8_t* llr, uint8_t* out_bits, int32_t* iters, void* stream)
{
  MIPHY_REQUIRE(p && llr && out_bits && iters, "miphy_ldpc_decode_plan_run: null argument");
  return miphy_ldpc_decode_classes_launch(p->ctx, p->d_descs, p->cls, p->d_order, p->d_bundles, llr, out_bits, iters, nullptr, nullptr, (hipStream_t)stream,
                                          nullptr, nullptr, false);
}

extern "C" uint32_t miphy_ldpc_decode_plan_nof_launches(const miphy_ldpc_decode_plan* p)
{
  return p ? (uint32_t)p->cls.classes.size() : 0u;
}

extern "C" void miphy_ldpc_decode_plan_destroy(miphy_ldpc_decode_plan* p)
{
  if (!p)
    return;
  if (p->d_buf)
    (void)hipFree(p->d_buf);
  delete p;
}
