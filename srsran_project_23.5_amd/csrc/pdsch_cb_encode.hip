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
                       const uint32_t* __restrict__ tb_crc, uint8_t* __restrict__ cw_out)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t       red[CBE_THREADS / 64];
  const miphy_pdsch_cb_desc d   = descs[blockIdx.x];
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

} // namespace

// d_descs: device descriptors; max_lds: largest dynamic LDS any codeblock of the launch needs (miphy_pdsch_cb_encode_lds).
size_t miphy_pdsch_cb_encode_lds(uint32_t K, uint32_t Z, uint32_t out_len)
{
  return ((K + 15) & ~(size_t)15) + 4 * (size_t)Z + ((4 * (size_t)Z + 15) & ~(size_t)15) + ((out_len + 15) & ~(size_t)15) + MIPHY_MAX_EDGES * 4 + 48 * 2 + 16;
}

int miphy_pdsch_cb_encode_launch(miphy_ctx* ctx, const miphy_pdsch_cb_desc* d_descs, uint32_t ncb, size_t max_lds, const uint8_t* tb_in, const uint32_t* tb_crc,
                                 uint8_t* cw_out, hipStream_t s)
{
  if (ncb == 0)
    return MIPHY_OK;
  if (max_lds > 48 * 1024)
    MIPHY_HIP_CHECK(hipFuncSetAttribute((const void*)pdsch_cb_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds));
  hipLaunchKernelGGL(pdsch_cb_encode_kernel, dim3(ncb), dim3(CBE_THREADS), max_lds, s, d_descs, ctx->d_tables, tb_in, tb_crc, cw_out);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
