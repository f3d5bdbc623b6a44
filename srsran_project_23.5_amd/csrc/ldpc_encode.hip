// LDPC encoder -- one workgroup per codeblock, lane l owns bit l of every lifted node.
//
// Behaviour contract: srsran::ldpc_encoder_impl::encode (lib/phy/upper/channel_coding/ldpc/ldpc_encoder_impl.cpp:44-81)
// and ldpc_encoder_generic.cpp:30-223 (systematic accumulation, four closed-form high-rate solutions, extension rows).
// The message and the four core parity nodes are staged in LDS; cyclic shifts are LDS address rotations.
#include "miphy_internal.h"

namespace {

__global__ void __launch_bounds__(MIPHY_MAX_Z)
ldpc_encode_kernel(const miphy_ldpc_enc_desc* __restrict__ descs,
                   const miphy_graph_tables* __restrict__ tab,
                   const uint8_t* __restrict__ in_base,
                   uint8_t* __restrict__ out_base)
{
  __shared__ uint8_t msg[22 * MIPHY_MAX_Z];
  __shared__ uint8_t aux[4 * MIPHY_MAX_Z];
  __shared__ uint8_t par[4 * MIPHY_MAX_Z];
  const miphy_ldpc_enc_desc d   = descs[blockIdx.x];
  const int                 tid = threadIdx.x, nt = blockDim.x;
  const int                 Z   = d.Z;
  const int                 bgi = (d.bg == 1) ? 0 : 1;
  const int                 bgK = bgi ? 10 : 22;
  const int                 K   = bgK * Z;
  const int                 zp  = tab->z_pos[Z];
  const int                 ils = tab->i_ls[Z];
  const uint8_t*            in  = in_base + d.in_offset;
  uint8_t*                  out = out_base + d.out_offset;
  const int                 out_len = (int)d.out_len;

  // ldpc_encoder_impl.cpp:61-70
  int cb_len = max(out_len + 2 * Z, K + 4 * Z);
  cb_len     = ((cb_len + Z - 1) / Z) * Z;
  const int nof_layers = cb_len / Z - bgK;

  for (int k = tid; k < K; k += nt) {
    const uint8_t v = in[k];
    msg[k]          = v;
    // systematic part, shortened by 2Z, copied verbatim (fillers stay 254): generic.cpp:87,113-119
    if (k >= 2 * Z && k - 2 * Z < out_len)
      out[k - 2 * Z] = v;
  }
  __syncthreads();

  const uint32_t* edges_g   = tab->edge[bgi][zp];
  const uint16_t* row_start = tab->row_start[bgi];
  const int       l         = tid;
  const int       hr_base   = bgK * Z; // LDS-offset base of the first parity node in the edge table's column*Z units

  // Core rows: aux[m][l] = XOR of rotated information nodes (generic.cpp:56-88).
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
  // First parity node (generic.cpp:121-223).
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
        out[o] = par[k * Z + l];
    }
    // Extension rows (generic.cpp:90-111): information part + the <=4 core parity nodes.
    for (int m = 4; m < nof_layers; ++m) {
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
      out[o] = (uint8_t)(acc & 1u);
    }
  }
}

} // namespace

extern "C" int miphy_ldpc_encode_batch(miphy_ctx*                 ctx,
                                       const miphy_ldpc_enc_desc* descs,
                                       int                        descs_on_device,
                                       uint32_t                   n,
                                       const uint8_t*             msg_in,
                                       uint8_t*                   cb_out,
                                       void*                      stream)
{
  MIPHY_REQUIRE(ctx && descs && msg_in && cb_out, "miphy_ldpc_encode_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  int threads = MIPHY_MAX_Z;
  if (!descs_on_device) {
    threads = 64;
    for (uint32_t i = 0; i < n; ++i) {
      const miphy_ldpc_enc_desc& d = descs[i];
      MIPHY_REQUIRE(d.bg == 1 || d.bg == 2, "ldpc_encode: desc %u: invalid base graph", i);
      MIPHY_REQUIRE(d.Z <= MIPHY_MAX_Z && ctx->h_tables->z_pos[d.Z] != 0xffff, "ldpc_encode: desc %u: invalid lifting size %u", i, d.Z);
      const unsigned nshort = (d.bg == 1) ? 66 : 50;
      MIPHY_REQUIRE(d.out_len <= nshort * d.Z, "ldpc_encode: desc %u: output size %u exceeds %u", i, d.out_len, nshort * d.Z);
      const int t = ((d.Z + 63) / 64) * 64;
      threads     = t > threads ? t : threads;
    }
  }
  hipStream_t s       = (hipStream_t)stream;
  const void* d_descs = nullptr;
  int         rc      = miphy_stage_descs(ctx, descs, descs_on_device, sizeof(miphy_ldpc_enc_desc) * (size_t)n, s, &d_descs);
  if (rc)
    return rc;
  hipLaunchKernelGGL(ldpc_encode_kernel, dim3(n), dim3(threads), 0, s, (const miphy_ldpc_enc_desc*)d_descs, ctx->d_tables, msg_in, cb_out);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
