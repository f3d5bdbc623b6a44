// LDPC decoder for small lifting sizes (Z <= 64): SEVERAL codeblocks per wavefront.
//
// The packed kernel of ldpc_decode_pk.hip gives every codeblock a workgroup of its own, and a lane two lifted check rows; a codeblock
// with Z = 15 then keeps 8 of 64 lanes busy and still pays a workgroup's load / CRC / output phases and barriers. The reference scales
// its work with the lifting size (ldpc_decoder_avx2.cpp:59-64: node_size_avx2 = ceil(Z / 32) vectors per node). Here a wavefront takes a
// BUNDLE of G = floor(64 / H) codeblocks of one (base graph, lifting size), H = ceil(Z / 2): lanes [g H, (g + 1) H) own the row pairs
// (l, l + H) of codeblock g, every codeblock has its own soft bits in LDS (the group's byte offset is one more term of the address
// sum the row update computes anyway) and its own message dwords (a lane's messages are private to it, so the layout
// [edge pair][64 lanes] of the packed kernel serves all groups at once). The workgroup IS the wavefront: no s_barrier anywhere, the
// layers of a codeblock are ordered by the in-order LDS pipe of the wave, and the groups advance through the layers in lockstep (a
// group with fewer layers, an earlier CRC match or all-zero input idles under the execution mask).
//
// Same arithmetic as the other decoders (ldpc_pk_device.h; reference: ldpc_decoder_impl.cpp:60-146 + ldpc_decoder_avx2.cpp:66-243);
// bundles are formed on the host from codeblocks that agree in base graph, lifting size, CRC polynomial, CRC mode and iteration
// limit (miphy_ldpc_build_classes, ldpc_decode.hip), everything else (input length, fillers, buffers, HARQ state) is per group.
#include "miphy_internal.h"
#include "ldpc_pk_device.h"
#include <algorithm>

namespace {

// Is the checksum of the first L hard bits of this lane's codeblock zero? Mask / popcount form (crc_zmask), words strided over the
// H lanes of the group, partial parities combined through the group's word of LDS. Every lane of the wave must call.
__device__ __forceinline__ bool group_crc_is_zero(const int8_t* softg, const miphy_graph_tables* __restrict__ tab, int zi, int order, int L,
                                                  uint32_t* redg, int l, int H, bool member)
{
  if (member) {
    const int nw = (L + 31) >> 5;
    uint32_t  acc[24];
#pragma unroll
    for (int k = 0; k < 24; ++k)
      acc[k] = 0;
    for (int t = l; t < nw; t += H) {
      uint32_t  w   = hard_flags(softg, t);
      const int rem = L - 32 * t;
      if (rem < 32) { // last word: positions 4 q + b >= rem are not message bits
        uint32_t valid = 0;
        for (int q = 0; q < 8; ++q) {
          const int      nb = min(4, max(0, rem - 4 * q));
          const uint32_t lo = (nb >= 4) ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
          valid |= (0x01010101u & lo) << q;
        }
        w &= valid;
      }
      const uint4* m = reinterpret_cast<const uint4*>(tab->crc_zmask[zi][nw - 1 - t]);
#pragma unroll
      for (int g4 = 0; g4 < 6; ++g4) {
        const uint4 mk = m[g4];
        acc[4 * g4 + 0] += __builtin_popcount(w & mk.x);
        acc[4 * g4 + 1] += __builtin_popcount(w & mk.y);
        acc[4 * g4 + 2] += __builtin_popcount(w & mk.z);
        acc[4 * g4 + 3] += __builtin_popcount(w & mk.w);
      }
    }
    uint32_t par = 0;
#pragma unroll
    for (int k = 0; k < 24; ++k)
      par |= (acc[k] & 1u) << k;
    par &= (1u << order) - 1u;
    if (par)
      atomicXor(redg, par);
  }
  __syncthreads();
  const uint32_t crc = member ? *redg : 1u;
  __syncthreads();
  if (member && l == 0)
    *redg = 0;
  __syncthreads();
  return crc == 0;
}

// The same by division, for polynomials without a mask table (CRC24C, CRC11): partial remainder of each 32-bit word times its
// position weight.
__device__ __forceinline__ bool group_crc_is_zero_div(const int8_t* softg, const miphy_graph_tables* __restrict__ tab, int crc_id, uint32_t poly,
                                                      uint32_t order, int K, int L, uint32_t* redg, int l, int H, bool member)
{
  if (member) {
    const int      nfull = L >> 5, rbits = L & 31, nwords = (L + 31) >> 5;
    const uint32_t top   = 1u << order;
    uint32_t       part  = 0;
    for (int t = l; t < nwords; t += H) {
      const uint32_t w   = hard_word(softg, t, K);
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
    if (part)
      atomicXor(redg, part);
  }
  __syncthreads();
  const uint32_t crc = member ? *redg : 1u;
  __syncthreads();
  if (member && l == 0)
    *redg = 0;
  __syncthreads();
  return crc == 0;
}

// One wavefront per workgroup (so __syncthreads() is a wait for the wave's own LDS operations, no s_barrier is emitted), persistent:
// the grid is what the chip holds, bundles beyond it are drawn from the launch's queue counter (ticket scheme of the packed kernel).
template <bool GMSG>
__global__ void __launch_bounds__(64, 4)
ldpc_decode_pkw_kernel(const miphy_ldpc_dec_desc* __restrict__ descs,
                       const miphy_graph_tables* __restrict__ tab,
                       const int8_t* __restrict__ llr_base,
                       uint8_t* __restrict__ out_base,
                       int32_t* __restrict__ iters_out,
                       int lay_alloc,  // layers any codeblock of the launch can reach (sizes soft bits and messages)
                       int soft_total, // LDS bytes reserved for the soft bits of a bundle (largest G * group stride of the launch)
                       const uint32_t* __restrict__ harq_slot,
                       uint8_t* __restrict__ harq_crc_ok,
                       const uint32_t* __restrict__ order,   // codeblock indices
                       const uint32_t* __restrict__ bundles, // per bundle: first position in `order`, number of codeblocks
                       uint32_t nof_bundles,
                       uint32_t* __restrict__ queue,
                       uint32_t* __restrict__ gmsg,
                       int gmsg_pairs)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  int8_t*   soft = reinterpret_cast<int8_t*>(smem);
  for (uint32_t b = blockIdx.x; b < nof_bundles;) {
    // next bundle of this wave (the ticket that draws the last bundle clears the counter for the next launch)
    uint32_t next = 0;
    if (lane == 0) {
      const uint32_t ticket = atomicAdd(queue, 1u);
      if (ticket == nof_bundles - 1u)
        *queue = 0u;
      next = gridDim.x + ticket;
    }
    next = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);

    const uint32_t            first = bundles[2 * b], cnt = bundles[2 * b + 1];
    const miphy_ldpc_dec_desc d0    = load_words(descs + order[first]); // what the bundle has in common
    const int                 Z     = d0.Z;
    const int                 H     = (Z + 1) >> 1;
    const int                 bgi   = (d0.bg == 1) ? 0 : 1;
    const int                 bgK   = bgi ? 10 : 22;
    const int                 bgM   = bgi ? 42 : 46;
    const int                 K     = bgK * Z;
    const int                 zp    = tab->z_pos[Z];
    const int                 g     = lane / H; // group = codeblock of the bundle
    const int                 l     = lane - g * H;
    const bool                member = g < (int)cnt;
    const int                 sstride = (((bgK + lay_alloc) * Z) + 15) & ~15;
    const uint32_t            base    = member ? (uint32_t)(g * sstride) : 0u;
    const int                 pairs_all = tab->pair_start[bgi][lay_alloc];
    uint32_t* msg0 = GMSG ? gmsg + (size_t)blockIdx.x * ((size_t)gmsg_pairs * 64) + lane : reinterpret_cast<uint32_t*>(smem + soft_total) + lane;
    uint32_t* red  = reinterpret_cast<uint32_t*>(smem + soft_total) + (GMSG ? 0 : pairs_all * 64); // [0,64): last non-zero input, [64,128): CRC

    const uint32_t             cbi    = order[first + (member ? g : 0)];
    const miphy_ldpc_dec_desc* dp     = descs + cbi;
    const int                  in_len = (int)dp->in_len;
    const int                  nf     = dp->nof_filler_bits;
    const int8_t*              llr    = llr_base + dp->llr_offset;
    uint8_t*                   out    = out_base + dp->out_offset;

    __syncthreads(); // the previous bundle's accesses to soft[] / red[] are done
    red[lane]      = 0;
    red[64 + lane] = 0;
    bool decode = member;
    if (member && harq_crc_ok && harq_crc_ok[harq_slot[cbi]]) { // pusch_decoder_impl.cpp:184: CRC already OK, keep the message
      decode = false;
      if (l == 0)
        iters_out[cbi] = -1;
    }
    if (member) { // clear the group's soft bits: punctured nodes, everything behind the input
      uint4* z4 = reinterpret_cast<uint4*>(soft + base);
      for (int k = l; k < (sstride >> 4); k += H)
        z4[k] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    int last = 0;
    if (decode) {
      // the group's lanes copy their codeblock as aligned dwords (any alignment of the input) and note the last non-zero soft bit
      const int       mb  = (int)((uintptr_t)llr & 3);
      const uint32_t* p4  = reinterpret_cast<const uint32_t*>(llr - mb);
      const int       ndw = (in_len + mb + 3) >> 2;
      int8_t*         dst = soft + base + 2 * Z - mb; // byte b of dword q goes to dst[4 q + b]
#pragma unroll 4
      for (int q = l; q < ndw; q += H) {
        const uint32_t w = p4[q];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
          const int pos = 4 * q + bb - mb;
          if (pos >= 0 && pos < in_len) {
            const int8_t v  = (int8_t)(w >> (8 * bb));
            dst[4 * q + bb] = v;
            last            = v ? pos + 1 : last;
          }
        }
      }
      atomicMax(reinterpret_cast<int*>(&red[g]), last);
    }
    __syncthreads();
    last = member ? (int)red[g] : 0;

    const bool use_crc    = d0.crc_poly != MIPHY_CRC_NONE;
    const bool final_only = use_crc && (d0.flags & 1u);
    const int  max_iter   = d0.max_iter;
    const int  kwords     = (K + 31) >> 5;
    if (decode && last == 0) { // ldpc_decoder_impl.cpp:88-94
      if (!use_crc) {
        for (int bb = l; bb < (K + 7) / 8; bb += H) {
          const int rem = K - 8 * bb;
          out[bb]       = (rem >= 8) ? 0xff : (uint8_t)(0xff << (8 - rem));
        }
      }
      if (l == 0)
        iters_out[cbi] = 0;
      decode = false;
    }
    // ldpc_decoder_impl.cpp:101-114
    int nlay = 0;
    if (decode) {
      int cb_len = max(last + 2 * Z, K + 4 * Z);
      cb_len     = ((cb_len + Z - 1) / Z) * Z;
      nlay       = cb_len / Z - bgK;
    }
    int nlay_max = nlay;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
      nlay_max = max(nlay_max, __shfl_xor(nlay_max, off));
    nlay_max = __builtin_amdgcn_readfirstlane(nlay_max);

    const uint32_t* edges_g = tab->edge_sb[bgi][zp];
    uint32_t        lay_info = 0; // per layer (one per lane): first edge, degree, first message pair
    if (lane < bgM) {
      const uint32_t e0 = tab->row_start[bgi][lane];
      lay_info          = e0 | (((uint32_t)tab->row_start[bgi][lane + 1] - e0) << 10) | ((uint32_t)tab->pair_start[bgi][lane] << 16);
    }
    uint32_t poly = 0, order_c = 0;
    int      L = 0;
    if (use_crc) {
      poly    = tab->crc_poly[d0.crc_poly];
      order_c = tab->crc_order[d0.crc_poly];
      L       = K - nf;
    }
    const int zi = use_crc ? miphy_crc_zmask_index(d0.crc_poly) : -1;

    // Odd lifting size: the last lane of a group would own rows H - 1 and Z (= row 0 again); with a zero distance to its second row
    // both halves of its packed registers carry row H - 1 (same reads, same values, same stores).
    int Zv = Z, Hv = (l + H < Z) ? H : 0;
    asm volatile("" : "+v"(Zv), "+v"(Hv));
    int  result_iters = 0;
    bool run          = decode;
    for (int it = 0; it < max_iter; ++it) {
      if (!__any(run))
        break;
      for (int m = 0; m < nlay_max; ++m) {
        const uint32_t  li    = (uint32_t)__builtin_amdgcn_readlane((int)lay_info, m);
        const int       e0    = (int)(li & 0x3ffu);
        const int       d     = (int)((li >> 10) & 0x3fu);
        const uint32_t* edges = edges_g + 2 * e0;
        if (run && m < nlay) {
          uint32_t* cl = msg0 + 64 * (li >> 16);
          if (it == 0)
            update_rows_pk_any<true>(d, soft, cl, edges, l, Hv, Zv, base);
          else
            update_rows_pk_any<false>(d, soft, cl, edges, l, Hv, Zv, base);
        }
        __syncthreads();
      }
      if (use_crc && !final_only) { // ldpc_decoder_impl.cpp:126-133
        const bool ok = zi >= 0 ? group_crc_is_zero(soft + base, tab, zi, (int)order_c, L, &red[64 + g], l, H, run)
                                : group_crc_is_zero_div(soft + base, tab, d0.crc_poly, poly, order_c, K, L, &red[64 + g], l, H, run);
        if (run && ok) {
          result_iters = it + 1;
          run          = false;
        }
      }
    }
    if (final_only) { // pusch_decoder_impl.cpp:105-118
      const bool ok = zi >= 0 ? group_crc_is_zero(soft + base, tab, zi, (int)order_c, L, &red[64 + g], l, H, decode)
                              : group_crc_is_zero_div(soft + base, tab, d0.crc_poly, poly, order_c, K, L, &red[64 + g], l, H, decode);
      result_iters  = ok ? max_iter : 0;
    }
    if (decode) {
      const bool out_aligned = ((uintptr_t)out & 3u) == 0;
      for (int t = l; t < kwords; t += H) {
        const uint32_t w      = hard_word(soft + base, t, K);
        const int      nbytes = min(4, (K - 32 * t + 7) / 8);
        if (nbytes == 4 && out_aligned) {
          reinterpret_cast<uint32_t*>(out)[t] = __builtin_bswap32(w);
        } else {
          for (int qq = 0; qq < nbytes; ++qq)
            out[4 * t + qq] = (uint8_t)(w >> (24 - 8 * qq));
        }
      }
      if (l == 0) {
        iters_out[cbi] = result_iters;
        if (harq_crc_ok && result_iters > 0)
          harq_crc_ok[harq_slot[cbi]] = 1;
      }
    }
    b = next;
  }
}

} // namespace

// LDS of one bundle: soft bits of G groups (largest G * stride over the lifting sizes of the launch), the message dwords unless they
// live in global memory, 128 reduction words.
size_t miphy_ldpc_pkw_lds_bytes(size_t soft_total, int pairs_all)
{
  return soft_total + (size_t)pairs_all * 256 + 512;
}

namespace {
struct pkw_geometry {
  bool     gm;
  size_t   lds;
  uint32_t grid;
  int      pairs;
};
pkw_geometry pkw_geom(const miphy_ctx* ctx, uint32_t nof_bundles, int bgi, int lay, size_t soft_total, bool throughput_form)
{
  pkw_geometry g;
  g.pairs            = ctx->h_tables->pair_start[bgi][lay];
  const size_t lds_l = miphy_ldpc_pkw_lds_bytes(soft_total, g.pairs), lds_g = miphy_ldpc_pkw_lds_bytes(soft_total, 0);
  // Wavefronts per CU: LDS and the 16 the register budget of the kernel allows. The messages move to global memory (one coalesced
  // dword per lane, edge pair and layer visit, re-read out of L2 an iteration later) where LDS would leave fewer than two wavefronts
  // per SIMD and the move buys residency.
  auto per_cu = [](size_t lds) { return std::max(1, std::min((int)((size_t)160 * 1024 / lds), 16)); };
  // ... and only where the launch has more bundles than stay resident with the messages in LDS: a launch that fits the chip anyway is a
  // latency chain, and a global round trip per layer visit would sit on it.
  g.gm        = per_cu(lds_l) < 8 && per_cu(lds_g) > per_cu(lds_l) && (throughput_form || nof_bundles > (uint32_t)(ctx->num_cus * per_cu(lds_l)));
  g.lds       = g.gm ? lds_g : lds_l;
  g.grid      = std::min<uint32_t>(nof_bundles, (uint32_t)(ctx->num_cus * per_cu(g.lds)));
  return g;
}
} // namespace

size_t miphy_ldpc_pkw_gmsg_bytes(const miphy_ctx* ctx, uint32_t nof_bundles, int bgi, int lay, size_t soft_total, bool throughput_form)
{
  if (nof_bundles == 0)
    return 0;
  const pkw_geometry g = pkw_geom(ctx, nof_bundles, bgi, lay, soft_total, throughput_form);
  return g.gm ? (size_t)g.grid * (size_t)g.pairs * 256 : 0;
}

int miphy_ldpc_pkw_launch(miphy_ctx* ctx, const miphy_ldpc_dec_desc* d_descs, const uint32_t* d_order, const uint32_t* d_bundles, uint32_t nof_bundles,
                          int bgi, int lay, size_t soft_total, const int8_t* llr, uint8_t* out_bits, int32_t* iters, const uint32_t* harq_slot,
                          uint8_t* harq_crc_ok, hipStream_t s, int* used_gmsg, void* gmsg_buf, bool throughput_form)
{
  if (nof_bundles == 0)
    return MIPHY_OK;
  const pkw_geometry g = pkw_geom(ctx, nof_bundles, bgi, lay, soft_total, throughput_form);
  if (used_gmsg)
    *used_gmsg = g.gm ? 1 : 0;
  const void* kern = g.gm ? (const void*)ldpc_decode_pkw_kernel<true> : (const void*)ldpc_decode_pkw_kernel<false>;
  if (g.lds > 48 * 1024)
    MIPHY_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));
  uint32_t* queue = nullptr;
  int       rc    = miphy_next_queue_counter(ctx, &queue);
  if (rc)
    return rc;
  void* gmsg = gmsg_buf;
  if (g.gm && !gmsg && (rc = miphy_get_workspace(ctx, (size_t)g.grid * (size_t)g.pairs * 256, s, &gmsg, 3)))
    return rc;
  if (g.gm)
    hipLaunchKernelGGL((ldpc_decode_pkw_kernel<true>), dim3(g.grid), dim3(64), g.lds, s, d_descs, ctx->d_tables, llr, out_bits, iters, lay, (int)soft_total,
                       harq_slot, harq_crc_ok, d_order, d_bundles, nof_bundles, queue, (uint32_t*)gmsg, g.pairs);
  else
    hipLaunchKernelGGL((ldpc_decode_pkw_kernel<false>), dim3(g.grid), dim3(64), g.lds, s, d_descs, ctx->d_tables, llr, out_bits, iters, lay, (int)soft_total,
                       harq_slot, harq_crc_ok, d_order, d_bundles, nof_bundles, queue, (uint32_t*)nullptr, 0);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
