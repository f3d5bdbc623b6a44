// PDSCH codeblock chain in ONE kernel: segmentation of the transport block into this codeblock (TB bits, TB CRC and zero padding on the
// last one, CRC24B, fillers), LDPC encoding and rate matching, one workgroup per codeblock with the message, the core parity nodes and the
// needed part of the circular buffer in LDS. HBM sees the packed transport-block bits once (K/8 bytes per codeblock) and the rate-matched
// codeword once (E bytes, one bit per byte as the modulator reads it) -- the three-kernel chain it replaces in the transport-block level
// encoder wrote and re-read the unpacked message (K bytes) and the encoded codeblock (up to N bytes) through HBM on the way.
//
// Behaviour contract (identical outputs to pdsch_cb_prepare_kernel -> ldpc_encode_kernel -> rate_match_kernel, which stay behind
// miphy_ldpc_encode_batch / miphy_ldpc_rate_match_batch for codeblock-level callers):
//   ldpc_segmenter_impl.cpp:150-220, pdsch_encoder_impl.cpp:46-50 (codeblock assembly),
//   ldpc_encoder_impl.cpp:44-81, ldpc_encoder_generic.cpp:30-223 (systematic accumulation, closed-form core parity, extension rows),
//   ldpc_rate_matcher_impl.cpp:42-182 (bit selection from k0 around the fillers, bit interleaving).
#include "crc_device.h"
#include "miphy_ext.h"
#include "rdm_device.h"
#include <type_traits>

namespace {

constexpr int CBE_THREADS = MIPHY_MAX_Z; // lane l owns bit l of every lifted node

__global__ void __launch_bounds__(CBE_THREADS)
pdsch_cb_encode_kernel(const miphy_pdsch_cb_desc* __restrict__ descs, const miphy_graph_tables* __restrict__ tab, const uint8_t* __restrict__ tb_in,
                       const uint32_t* __restrict__ tb_crc, uint8_t* __restrict__ cw_out, int skip_packed)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t       red[CBE_THREADS / 64];
  const miphy_pdsch_cb_desc d   = descs[blockIdx.x];
  if (skip_packed && miphy_pdsch_cb_packed_ok(d))
    return; // the packed kernel of the same call encodes this codeblock
  const int                 tid = threadIdx.x, nt = blockDim.x;
  const int                 Z   = d.Z;
  const int                 bgi = (d.bg == 1) ? 0 : 1;
  const int                 bgK = bgi ? 10 : 22;
  const int                 K   = bgK * Z;
  const int                 zp  = tab->z_pos[Z];
  const int                 ils = tab->i_ls[Z];
  const int                 out_len = (int)d.out_len;
  uint8_t*                  msg = smem;                       // K bytes, one bit each (fillers 254)
  uint8_t*                  aux = msg + ((K + 15) & ~15);     // 4 Z
  uint8_t*                  par = aux + 4 * Z;                // 4 Z
  uint8_t*                  cb  = par + ((4 * Z + 15) & ~15); // out_len bytes: the codeblock as the rate matcher addresses it
  // the base graph's edges {shift << 16 | column * Z} and row starts in LDS: the row loops below then read them with one broadcast LDS load
  // per edge instead of a dependent scalar load from memory per edge (a serial chain of ~100 load latencies per codeblock)
  uint32_t* ledge  = reinterpret_cast<uint32_t*>(cb + ((out_len + 15) & ~15));
  uint16_t* lstart = reinterpret_cast<uint16_t*>(ledge + MIPHY_MAX_EDGES);
  {
    const uint32_t* eg = tab->edge[bgi][zp];
    const int       ne = bgi ? MIPHY_BG2_EDGES : MIPHY_BG1_EDGES;
    for (int e = tid; e < ne; e += nt)
      ledge[e] = eg[e];
    if (tid < 48)
      lstart[tid] = tab->row_start[bgi][tid];
  }

  // ---- the codeblock's message (pdsch_cb_prepare_kernel)
  const uint8_t* tb   = tb_in + d.tb_offset;
  uint32_t       used = d.take_bits;
  for (uint32_t i = tid; i < d.take_bits; i += nt) {
    const uint32_t bit = d.tb_bit_offset + i;
    msg[i]             = (tb[bit >> 3] >> (7 - (bit & 7))) & 1u;
  }
  if (d.nof_tb_crc_bits) {
    const uint32_t crc = tb_crc[d.tb_index];
    for (uint32_t i = tid; i < (uint32_t)d.nof_tb_crc_bits + d.zero_pad; i += nt)
      msg[used + i] = (i < d.nof_tb_crc_bits) ? (uint8_t)((crc >> (d.nof_tb_crc_bits - 1 - i)) & 1u) : 0;
    used += d.nof_tb_crc_bits + d.zero_pad;
  }
  __syncthreads();
  if (d.nof_cb_crc_bits) {
    // CRC24B over the `used` unpacked bits: lane t reduces its run of bits, weights with x^(bits that follow)
    const uint32_t poly = tab->crc_poly[MIPHY_CRC24B], order = 24, top = 1u << 24;
    const uint32_t per  = (used + nt - 1) / nt;
    const uint32_t b0   = tid * per;
    uint32_t       reg  = 0;
    if (b0 < used) {
      const uint32_t b1 = min(b0 + per, used);
      for (uint32_t i = b0; i < b1; ++i) {
        reg = (reg << 1) ^ ((uint32_t)msg[i] << order);
        reg ^= (reg & top) ? poly : 0u;
      }
      reg &= top - 1u;
      const uint32_t after = used - b1;
      if (after) {
        reg = crc_gf2_mulmod(reg, crc_pow32(tab, MIPHY_CRC24B, after >> 5, poly, order), poly, order);
        for (uint32_t b = 0; b < (after & 31u); ++b) {
          reg <<= 1;
          reg ^= (reg & top) ? poly : 0u;
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
      reg ^= __shfl_xor(reg, off);
    if ((tid & 63) == 0)
      red[tid >> 6] = reg;
    __syncthreads();
    uint32_t crc = 0;
    for (int w = 0; w < (nt >> 6); ++w)
      crc ^= red[w];
    if (tid < 24)
      msg[used + tid] = (uint8_t)((crc >> (23 - tid)) & 1u);
    used += 24;
  }
  for (uint32_t i = used + tid; i < (uint32_t)K; i += nt)
    msg[i] = 254; // ldpc::FILLER_BIT
  __syncthreads();

  // ---- LDPC encoder (ldpc_encode_kernel), output into the LDS codeblock
  int cb_len = max(out_len + 2 * Z, K + 4 * Z);
  cb_len     = ((cb_len + Z - 1) / Z) * Z;
  const int nof_layers = cb_len / Z - bgK;
  for (int k = 2 * Z + tid; k < K; k += nt) // systematic part, shortened by 2 Z, verbatim (fillers stay 254)
    if (k - 2 * Z < out_len)
      cb[k - 2 * Z] = msg[k];
  const uint32_t* edges_g   = ledge;
  const uint16_t* row_start = lstart;
  const int       l         = tid;
  const int       hr_base   = bgK * Z;
  if (l < Z) {
    for (int m = 0; m < 4; ++m) {
      uint32_t acc = 0;
      for (int e = row_start[m]; e < row_start[m + 1]; ++e) {
        const uint32_t ed  = edges_g[e];
        const int      col = (int)(ed & 0xffffu);
        if (col >= hr_base)
          continue;
        int pos = l + (int)(ed >> 16);
        pos     = (pos >= Z) ? pos - Z : pos;
        acc ^= msg[col + pos];
      }
      aux[m * Z + l] = (uint8_t)(acc & 1u);
    }
  }
  __syncthreads();
  if (l < Z) {
    int i = l;
    if (bgi == 0 && ils == 6) {
      i = (l - 105) % Z;
      i = (i < 0) ? i + Z : i;
    } else if (bgi == 1 && ils != 3 && ils != 7) {
      i = (l == 0) ? Z - 1 : l - 1;
    }
    par[l] = aux[i] ^ aux[Z + i] ^ aux[2 * Z + i] ^ aux[3 * Z + i];
  }
  __syncthreads();
  if (l < Z) {
    const int     ln = (l + 1 == Z) ? 0 : l + 1;
    const uint8_t a0 = aux[l], a1 = aux[Z + l], a2 = aux[2 * Z + l], a3 = aux[3 * Z + l];
    uint8_t       p1, p2, p3;
    if (bgi == 0) {
      const uint8_t p0x = (ils == 6) ? par[l] : par[ln];
      p1 = a0 ^ p0x;
      p3 = a3 ^ p0x;
      p2 = a2 ^ p3;
    } else {
      const uint8_t p0x = (ils == 3 || ils == 7) ? par[ln] : par[l];
      p1 = a0 ^ p0x;
      p2 = a1 ^ p1;
      p3 = a3 ^ p0x;
    }
    par[Z + l]     = p1;
    par[2 * Z + l] = p2;
    par[3 * Z + l] = p3;
    (void)a1;
    (void)a2;
  }
  __syncthreads();
  if (l < Z) {
    for (int k = 0; k < 4; ++k) {
      const int o = (bgK + k - 2) * Z + l;
      if (o < out_len)
        cb[o] = par[k * Z + l];
    }
    for (int m = 4; m < nof_layers; ++m) { // extension rows: information part + the core parity nodes
      const int o = (bgK + m - 2) * Z + l;
      if (o >= out_len)
        break;
      uint32_t acc = 0;
      for (int e = row_start[m]; e < row_start[m + 1]; ++e) {
        const uint32_t ed  = edges_g[e];
        const int      col = (int)(ed & 0xffffu);
        int            pos = l + (int)(ed >> 16);
        pos                = (pos >= Z) ? pos - Z : pos;
        if (col < hr_base)
          acc ^= msg[col + pos];
        else if (col < hr_base + 4 * Z)
          acc ^= par[col - hr_base + pos];
      }
      cb[o] = (uint8_t)(acc & 1u);
    }
  }
  __syncthreads();

  // ---- rate matcher (rate_match_kernel): bit selection from k0 around the fillers, interleaving out[i * mod + j] = sel[j * Kq + i]
  miphy_ldpc_rdm_desc r = {};
  r.bg = d.bg, r.rv = d.rv, r.mod = d.mod, r.Z = d.Z, r.nof_filler_bits = d.nof_filler_bits, r.Nref = d.Nref, r.E = d.E;
  const rm_geom g   = make_geom(r);
  uint8_t*      out = cw_out + d.cw_offset;
  // source of output bit o: o = i * mod + j -> selected bit j * Kq + i, rank (r0 + that) mod L in the buffer without its fillers. The
  // divisions are by the modulation order (a constant per instantiation) and a wrap that takes at most a few subtractions unless the
  // codeblock is repeated many times over.
  const bool few_wraps = (long long)g.r0 + g.E <= 4ll * g.L;
  auto       emit      = [&](auto MODC) {
    constexpr int MOD = decltype(MODC)::value;
    auto          src = [&](int o) {
      const int i = o / MOD, jj = o - i * MOD;
      int       rank = g.r0 + g.Kq * jj + i;
      if (few_wraps) {
        rank = (rank >= g.L) ? rank - g.L : rank;
        rank = (rank >= g.L) ? rank - g.L : rank;
        rank = (rank >= g.L) ? rank - g.L : rank;
      } else {
        rank %= g.L;
      }
      return (int)cb[(rank < g.f0) ? rank : rank + g.F];
    };
    if ((((uintptr_t)out) & 3u) == 0) { // four consecutive output bits per lane, one dword store
      const int nq = g.E >> 2;
      for (int q = tid; q < nq; q += nt) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          w |= (uint32_t)src(4 * q + b) << (8 * b);
        reinterpret_cast<uint32_t*>(out)[q] = w;
      }
      for (int o = (nq << 2) + tid; o < g.E; o += nt)
        out[o] = (uint8_t)src(o);
    } else {
      for (int o = tid; o < g.E; o += nt)
        out[o] = (uint8_t)src(o);
    }
  };
  switch (g.mod) {
    case 8:
      emit(std::integral_constant<int, 8>{});
      break;
    case 6:
      emit(std::integral_constant<int, 6>{});
      break;
    case 4:
      emit(std::integral_constant<int, 4>{});
      break;
    case 2:
      emit(std::integral_constant<int, 2>{});
      break;
    default:
      emit(std::integral_constant<int, 1>{});
      break;
  }
}


// ---- the same chain on PACKED bits, one WAVEFRONT per codeblock (lifting sizes that are multiples of 32: every large transport block) ------------
// The kernel above spends one lane per bit. Here a lifted node is W = Z / 32 words, bit i of a vector in word i >> 5 at position 31 - (i & 31)
// (the order of the transport-block bytes), and a cyclic shift of a node is a word rotation plus one funnel shift per word:
//   out word k = bits [32 k + s, 32 k + s + 32) mod Z of the input = alignbit(in[(k + s / 32) mod W], in[(k + s / 32 + 1) mod W], 32 - s % 32).
// A row of the base graph costs one such word per (edge, output word) instead of Z single-bit reads: the four core rows of base graph 1
// are 76 x 12 word operations per codeblock. Steps (all in LDS, the wavefront's DS operations execute in order, the barriers are those of a
// one-wavefront workgroup): message words straight from the transport-block bytes (byte-swapped dwords), TB CRC / padding on the last
// codeblock, CRC24B by a byte table built in LDS (each lane a run of words, one weight multiplication per lane), core rows, closed-form
// core parity (rotations by 0 / 1 / Z - 1 / Z - 105), extension rows, then the rate matcher in two moves: the selected bits S[t] =
// buffer[(r0 + t) mod L around the fillers] as packed words (a funnel shift per word, the few words that straddle the filler gap or the
// wrap bit by bit), and the interleaver out[i mod + j] = S[j Kq + i] -- for a lane that stores dword q, q + 64, ... the four source bits
// sit at FIXED bit positions of consecutive words of S (256 output bytes later = 256 / mod elements later = a whole number of words), so a
// bit costs one LDS read and one bit-field extract. Same outputs as the kernel above (tests/test_sch_gpu.py, test_pdsch_proc_gpu.py).
constexpr int      CBP_THREADS = 64;

__device__ __forceinline__ uint32_t cbp_alignbit(uint32_t hi, uint32_t lo, uint32_t sh)
{
  return __builtin_amdgcn_alignbit(hi, lo, sh);
}
// word k of the node `n` (W words) rotated so that output bit l = input bit (l + s) mod Z
__device__ __forceinline__ uint32_t cbp_rotw(const uint32_t* n, int W, int k, uint32_t s)
{
  int i0 = k + (int)(s >> 5);
  i0     = (i0 >= W) ? i0 - W : i0;
  int i1 = i0 + 1;
  i1     = (i1 == W) ? 0 : i1;
  const uint32_t a = n[i0], b = n[i1], r = s & 31u;
  return r ? cbp_alignbit(a, b, 32u - r) : a;
}
// bits [p, p + 32) of a packed array (reads word (p >> 5) + 1 as well)
__device__ __forceinline__ uint32_t cbp_get32(const uint32_t* a, uint32_t p)
{
  const uint32_t w = p >> 5, r = p & 31u;
  const uint32_t x = a[w], y = a[w + 1];
  return r ? cbp_alignbit(x, y, 32u - r) : x;
}
__device__ __forceinline__ uint32_t cbp_bit(const uint32_t* a, uint32_t p)
{
  return (a[p >> 5] >> (31u - (p & 31u))) & 1u;
}

__global__ void __launch_bounds__(CBP_THREADS)
pdsch_cb_encode_pk_kernel(const miphy_pdsch_cb_desc* __restrict__ descs, const miphy_graph_tables* __restrict__ tab, const uint8_t* __restrict__ tb_in,
                          const uint32_t* __restrict__ tb_crc, uint8_t* __restrict__ cw_out)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const miphy_pdsch_cb_desc d = descs[blockIdx.x]; // (as scalar dwords -- load_words -- the kernel measured 200 us against 162 us per 38 912 codeblocks)
  if (!miphy_pdsch_cb_packed_ok(d))
    return; // the one-lane-per-bit kernel takes this codeblock
  const int lane = threadIdx.x;
  const int Z = d.Z, W = Z >> 5;
  const int bgi = (d.bg == 1) ? 0 : 1, bgK = bgi ? 10 : 22;
  const int K = bgK * Z, KW = bgK * W;
  const int zp = tab->z_pos[Z], ils = tab->i_ls[Z];
  const int out_len = (int)d.out_len;
  int       cb_len  = max(out_len + 2 * Z, K + 4 * Z);
  cb_len            = ((cb_len + Z - 1) / Z) * Z;
  const int nof_layers = cb_len / Z - bgK;
  const int NW         = (bgK + nof_layers) * W;
  const int SW         = ((int)d.E + 31) >> 5;
  uint32_t* node   = reinterpret_cast<uint32_t*>(smem);       // nodes 0 .. bgK + nof_layers - 1, W words each (+ 2 words read by cbp_get32)
  uint32_t* S      = node + NW + 2;                            // selected bits (+ 2); before that: scratch of the core rows
  uint32_t* ledge  = S + max(SW, 6 * W) + 2;
  uint16_t* lstart = reinterpret_cast<uint16_t*>(ledge + MIPHY_MAX_EDGES);
  uint32_t* crct   = reinterpret_cast<uint32_t*>(lstart + 48);
  const uint32_t poly = tab->crc_poly[MIPHY_CRC24B];
  // Every memory request of the head is issued before the first result is used (base-graph edges, row starts, the message dwords of the
  // transport block, its checksum): a wavefront per codeblock has nobody to hide a chain of dependent round trips behind.
  {
    const uint32_t* eg = tab->edge[bgi][zp];
    const int       ne = bgi ? MIPHY_BG2_EDGES : MIPHY_BG1_EDGES;
    constexpr int   EU = (MIPHY_MAX_EDGES + CBP_THREADS - 1) / CBP_THREADS, MU = (22 * 12 + CBP_THREADS - 1) / CBP_THREADS;
    uint32_t        ev[EU], mlo[MU], mhi[MU];
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int e = lane + u * CBP_THREADS;
      ev[u]       = (e < ne) ? eg[e] : 0u;
    }
    const uint32_t rs = (lane < 48) ? tab->row_start[bgi][lane] : 0u;
    const uint8_t* tb = tb_in + d.tb_offset;
    const uint32_t B00 = d.tb_bit_offset >> 3, Bend = (d.tb_bit_offset + d.take_bits) >> 3;
#pragma unroll
    for (int u = 0; u < MU; ++u) {
      const int      k  = lane + u * CBP_THREADS;
      const uint32_t B0 = B00 + 4u * (uint32_t)k;
      mlo[u] = mhi[u] = 0u;
      if (k < KW && B0 < Bend) {
        const uintptr_t a  = (uintptr_t)(tb + B0);
        const uint32_t* p4 = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
        const uint32_t  sh = (uint32_t)(a & 3u);
        mlo[u]             = p4[0];
        if (sh != 0 && B0 + (4u - sh) < Bend)
          mhi[u] = p4[1];
      }
    }
    if (d.nof_cb_crc_bits) // byte table of CRC24B: (b(x) x^24) mod P
      for (int b = lane; b < 256; b += CBP_THREADS) {
        uint32_t reg = (uint32_t)b << 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          reg <<= 1;
          reg ^= (reg & (1u << 24)) ? poly : 0u;
        }
        crct[b] = reg & 0xffffffu;
      }
    for (int i = KW + lane; i < NW + 2; i += CBP_THREADS)
      node[i] = 0u;
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int e = lane + u * CBP_THREADS;
      if (e < ne)
        ledge[e] = ev[u];
    }
    if (lane < 48)
      lstart[lane] = (uint16_t)rs;
    // ---- message words from the transport-block bytes (byte-swapped: bit i of the codeblock at position 31 - i % 32 of word i / 32)
#pragma unroll
    for (int u = 0; u < MU; ++u) {
      const int k = lane + u * CBP_THREADS;
      if (k < KW) {
        const uint32_t B0 = B00 + 4u * (uint32_t)k;
        uint32_t       v  = 0;
        if (B0 < Bend) {
          const uint32_t sh = (uint32_t)((uintptr_t)(tb + B0) & 3u);
          v                 = __builtin_bswap32(sh ? __builtin_amdgcn_alignbyte(mhi[u], mlo[u], sh) : mlo[u]);
          const uint32_t nb = Bend - B0;
          if (nb < 4)
            v &= ~(0xffffffffu >> (8u * nb));
        }
        node[k] = v;
      }
    }
  }
  __syncthreads();
  auto put_bits = [&](uint32_t pos_bits, uint32_t value, int nbits) { // lane 0: `value` (nbits, a multiple of 8) at a byte-aligned position, over zeros
    for (int b = 0; b < nbits / 8; ++b) {
      const uint32_t byte = (value >> (nbits - 8 - 8 * b)) & 0xffu, pos = (pos_bits >> 3) + b;
      node[pos >> 2] |= byte << (24u - 8u * (pos & 3u));
    }
  };
  uint32_t used = d.take_bits;
  if (d.nof_tb_crc_bits) {
    if (lane == 0)
      put_bits(used, tb_crc[d.tb_index], d.nof_tb_crc_bits);
    used += d.nof_tb_crc_bits + d.zero_pad;
    __syncthreads();
  }
  if (d.nof_cb_crc_bits) {
    const uint32_t nbytes = used >> 3, nwords = (nbytes + 3) >> 2;
    const uint32_t perw   = (nwords + CBP_THREADS - 1) / CBP_THREADS;
    const uint32_t w0     = (uint32_t)lane * perw;
    uint32_t       reg    = 0;
    if (w0 < nwords) {
      const uint32_t w1 = min(w0 + perw, nwords);
      for (uint32_t w = w0; w < w1; ++w) {
        const uint32_t x = node[w];
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
          if (4 * w + bi < nbytes)
            reg = ((reg << 8) & 0xffffffu) ^ crct[((reg >> 16) ^ (x >> (24 - 8 * bi))) & 0xffu];
      }
      const uint32_t after = 8u * (nbytes - min(4u * w1, nbytes));
      if (after) {
        reg = crc_gf2_mulmod(reg, crc_pow32(tab, MIPHY_CRC24B, after >> 5, poly, 24), poly, 24);
        for (uint32_t b = 0; b < (after & 31u); ++b) {
          reg <<= 1;
          reg ^= (reg & (1u << 24)) ? poly : 0u;
        }
      }
      reg &= 0xffffffu;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
      reg ^= __shfl_xor(reg, off);
    if (lane == 0)
      put_bits(used, reg, 24);
    used += 24;
    __syncthreads();
  }
  // ---- core rows (information part), closed-form core parity (ldpc_encoder_generic.cpp:100-223)
  uint32_t* aux = S;          // [4][W]
  uint32_t* s0  = S + 4 * W;  // [W]
  const uint32_t invW = (65536u + (uint32_t)W - 1u) / (uint32_t)W; // x / W = (x * invW) >> 16 for the x < 600 divided below (exact: W <= 12)
  if (lane < 4 * W) {
    const int m = (int)(((uint32_t)lane * invW) >> 16), k = lane - m * W;
    uint32_t  acc = 0;
    const int e1  = lstart[m + 1];
#pragma unroll 4
    for (int e = lstart[m]; e < e1; ++e) {
      const uint32_t ed = ledge[e];
      const int      cw = (int)((ed & 0xffffu) >> 5);
      acc ^= (cw < KW) ? cbp_rotw(node + cw, W, k, ed >> 16) : 0u; // (a parity column reads a zeroed node: branch-free)
    }
    aux[lane] = acc;
  }
  __syncthreads();
  if (lane < W)
    s0[lane] = aux[lane] ^ aux[W + lane] ^ aux[2 * W + lane] ^ aux[3 * W + lane];
  __syncthreads();
  uint32_t* par = node + KW; // nodes bgK .. bgK + 3
  if (lane < W) {
    uint32_t sp = 0; // par0 bit l = s0 bit i(l): i = (l - 105) mod Z / l - 1 / l
    if (bgi == 0 && ils == 6)
      sp = (uint32_t)((Z - (105 % Z)) % Z);
    else if (bgi == 1 && ils != 3 && ils != 7)
      sp = (uint32_t)(Z - 1);
    par[lane] = cbp_rotw(s0, W, lane, sp);
  }
  __syncthreads();
  if (lane < W) {
    const uint32_t a0 = aux[lane], a1 = aux[W + lane], a2 = aux[2 * W + lane], a3 = aux[3 * W + lane];
    uint32_t       p1, p2, p3;
    if (bgi == 0) {
      const uint32_t p0x = cbp_rotw(par, W, lane, (ils == 6) ? 0u : 1u);
      p1 = a0 ^ p0x, p3 = a3 ^ p0x, p2 = a2 ^ p3;
    } else {
      const uint32_t p0x = cbp_rotw(par, W, lane, (ils == 3 || ils == 7) ? 1u : 0u);
      p1 = a0 ^ p0x, p2 = a1 ^ p1, p3 = a3 ^ p0x;
    }
    par[W + lane] = p1, par[2 * W + lane] = p2, par[3 * W + lane] = p3;
  }
  __syncthreads();
  // ---- extension rows the rate matcher will read: information part + core parity nodes
  {
    int mend = 4;
    while (mend < nof_layers && (bgK + mend - 2) * Z < out_len)
      ++mend;
    const int CW4 = (bgK + 4) * W;
    for (int idx = lane; idx < (mend - 4) * W; idx += CBP_THREADS) {
      const int m = 4 + (int)(((uint32_t)idx * invW) >> 16), k = idx - (m - 4) * W;
      uint32_t  acc = 0;
      for (int e = lstart[m]; e < lstart[m + 1]; ++e) {
        const uint32_t ed = ledge[e];
        const int      cw = (int)((ed & 0xffffu) >> 5);
        if (cw < CW4)
          acc ^= cbp_rotw(node + cw, W, k, ed >> 16);
      }
      node[(bgK + m) * W + k] = acc;
    }
  }
  __syncthreads();
  // ---- rate matcher, first move: the selected bits
  miphy_ldpc_rdm_desc r = {};
  r.bg = d.bg, r.rv = d.rv, r.mod = d.mod, r.Z = d.Z, r.nof_filler_bits = d.nof_filler_bits, r.Nref = d.Nref, r.E = d.E;
  const rm_geom  g  = make_geom(r);
  const uint32_t Z2 = 2u * (uint32_t)Z;
  uint32_t rho = ((uint32_t)g.r0 + 32u * (uint32_t)lane) % (uint32_t)g.L; // rank of this lane's first selected bit, advanced by 2048 per round
  for (int k = lane; k < SW; k += CBP_THREADS) {
    uint32_t v;
    if ((int)rho + 32 <= g.L && ((int)rho + 32 <= g.f0 || (int)rho >= g.f0)) {
      v = cbp_get32(node, (((int)rho < g.f0) ? rho : rho + (uint32_t)g.F) + Z2);
    } else {
      v = 0;
      uint32_t rr = rho;
      for (int b = 0; b < 32; ++b) {
        const uint32_t idx = ((int)rr < g.f0) ? rr : rr + (uint32_t)g.F;
        v |= cbp_bit(node, idx + Z2) << (31 - b);
        rr = ((int)rr + 1 >= g.L) ? 0u : rr + 1u;
      }
    }
    S[k] = v;
    rho += 32u * CBP_THREADS;
    while (rho >= (uint32_t)g.L)
      rho -= (uint32_t)g.L;
  }
  if (lane < 2)
    S[SW + lane] = 0u;
  __syncthreads();
  // ---- second move: the interleaver, one bit per output byte
  uint8_t*  out = cw_out + d.cw_offset;
  const int E = g.E, Kq = g.Kq, mod = g.mod;
  auto      src = [&](int o) -> uint32_t { // output bit o = i * mod + j <- selected bit j * Kq + i
    const int i = o / mod, j = o - i * mod;
    return cbp_bit(S, (uint32_t)(j * Kq + i));
  };
  // (the modulation order stays a run-time value here: one instantiation of the loop per order measured 195 us against 162 us per 38 912
  // codeblocks of the headline size)
  if ((((uintptr_t)out) & 3u) == 0) {
    const int nq = E >> 2;
    if (mod == 8 || mod == 4 || mod == 2 || mod == 1) {
      const int       lg  = (mod == 8) ? 3 : (mod == 4) ? 2 : (mod == 2) ? 1 : 0; // the orders of this path are powers of two: shifts, not divisions
      const int       adv = 8 >> lg; // words of S per 64 dwords of output (256 bytes = 256 / mod elements later)
      uint32_t        sb[4];
      const uint32_t* sp[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int      o0 = 4 * lane + b, i0 = o0 >> lg, j = o0 - (i0 << lg);
        const uint32_t t0 = (uint32_t)(j * Kq + i0);
        sp[b] = S + (t0 >> 5), sb[b] = 31u - (t0 & 31u);
      }
      // four rounds at a time: sixteen independent LDS reads in flight per lane
      uint32_t* o32 = reinterpret_cast<uint32_t*>(out);
      int       q   = lane;
      for (; q + 3 * CBP_THREADS < nq; q += 4 * CBP_THREADS) {
        uint32_t x[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            x[u][b] = sp[b][u * adv];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          uint32_t w = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b)
            w |= ((x[u][b] >> sb[b]) & 1u) << (8 * b);
          o32[q + u * CBP_THREADS] = w;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
          sp[b] += 4 * adv;
      }
      for (; q < nq; q += CBP_THREADS) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          w |= ((sp[b][0] >> sb[b]) & 1u) << (8 * b);
        o32[q] = w;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          sp[b] += adv;
      }
    } else {
      for (int q = lane; q < nq; q += CBP_THREADS) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          w |= src(4 * q + b) << (8 * b);
        reinterpret_cast<uint32_t*>(out)[q] = w;
      }
    }
    for (int o = (nq << 2) + lane; o < E; o += CBP_THREADS)
      out[o] = (uint8_t)src(o);
  } else {
    for (int o = lane; o < E; o += CBP_THREADS)
      out[o] = (uint8_t)src(o);
  }
}

} // namespace

// d_descs: device descriptors; max_lds: largest dynamic LDS any codeblock of the launch needs (miphy_pdsch_cb_encode_lds).
size_t miphy_pdsch_cb_encode_lds(uint32_t K, uint32_t Z, uint32_t out_len)
{
  return ((K + 15) & ~(size_t)15) + 4 * (size_t)Z + ((4 * (size_t)Z + 15) & ~(size_t)15) + ((out_len + 15) & ~(size_t)15) + MIPHY_MAX_EDGES * 4 + 48 * 2 + 16;
}

// LDS of the packed kernel for one codeblock (0: the codeblock is not eligible, miphy_pdsch_cb_packed_ok).
size_t miphy_pdsch_cb_encode_pk_lds(const miphy_pdsch_cb_desc& d)
{
  if (!miphy_pdsch_cb_packed_ok(d))
    return 0;
  const uint32_t Z = d.Z, W = Z / 32, bgK = d.bg == 1 ? 22 : 10, K = bgK * Z;
  uint32_t       cb_len = std::max(d.out_len + 2 * Z, K + 4 * Z);
  cb_len                = ((cb_len + Z - 1) / Z) * Z;
  const uint32_t NW = (cb_len / Z) * W, SW = (d.E + 31) / 32;
  return 4 * ((size_t)NW + 2 + std::max(SW, 6 * W) + 2 + MIPHY_MAX_EDGES + 256) + 48 * 2 + 16;
}

// npacked of the ncb codeblocks go to the packed kernel (the builder counts them); both kernels run over all descriptors and leave the
// other kernel's codeblocks alone.
int miphy_pdsch_cb_encode_launch(miphy_ctx* ctx, const miphy_pdsch_cb_desc* d_descs, uint32_t ncb, size_t max_lds, const uint8_t* tb_in, const uint32_t* tb_crc,
                                 uint8_t* cw_out, hipStream_t s, uint32_t npacked, size_t max_lds_pk)
{
  if (ncb == 0)
    return MIPHY_OK;
  if (npacked > 0) {
    if (max_lds_pk > 48 * 1024)
      MIPHY_HIP_CHECK(hipFuncSetAttribute((const void*)pdsch_cb_encode_pk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_pk));
    hipLaunchKernelGGL(pdsch_cb_encode_pk_kernel, dim3(ncb), dim3(CBP_THREADS), max_lds_pk, s, d_descs, ctx->d_tables, tb_in, tb_crc, cw_out);
  }
  if (npacked < ncb) {
    if (max_lds > 48 * 1024)
      MIPHY_HIP_CHECK(hipFuncSetAttribute((const void*)pdsch_cb_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds));
    hipLaunchKernelGGL(pdsch_cb_encode_kernel, dim3(ncb), dim3(CBE_THREADS), max_lds, s, d_descs, ctx->d_tables, tb_in, tb_crc, cw_out, npacked > 0 ? 1 : 0);
  }
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
