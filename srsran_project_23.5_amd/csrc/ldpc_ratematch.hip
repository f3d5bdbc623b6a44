// LDPC rate matcher / rate dematcher -- gather formulation (every output element computes where its input lives).
//
// Behaviour contract:
//   rate_match   : lib/phy/upper/channel_coding/ldpc/ldpc_rate_matcher_impl.cpp:42-182
//   rate_dematch : lib/phy/upper/channel_coding/ldpc/ldpc_rate_dematcher_impl.cpp:43-254 with the AVX2 combine rule
//                  (ldpc_rate_dematcher_avx2_impl.cpp:45-58).
// The CPU code walks the circular buffer chunk by chunk; here each thread owns output positions and derives, in closed
// form, the (de)interleaver index, the circular-buffer rank (fillers skipped positionally) and the wrap-around passes,
// so both directions are pure HBM-bound gathers with coalesced writes.
#include "miphy_internal.h"

namespace {

struct rm_geom {
  int N, Ncb, F, f0, f1, L, k0, r0, Kq, E, mod;
};

__device__ __forceinline__ rm_geom make_geom(const miphy_ldpc_rdm_desc& d)
{
  rm_geom g;
  const int bgK = (d.bg == 1) ? 22 : 10, nshort = (d.bg == 1) ? 66 : 50;
  const int Z   = d.Z;
  g.N           = nshort * Z;
  g.Ncb         = (d.Nref > 0 && (int)d.Nref < g.N) ? (int)d.Nref : g.N;
  // TS 38.212 Table 5.4.2.1-2 (rate_matcher_impl.cpp:64-94): k0 = floor(k0num * Ncb / N) * Z.
  const int num = (d.bg == 1) ? ((d.rv == 0) ? 0 : (d.rv == 1) ? 17 : (d.rv == 2) ? 33 : 56)
                              : ((d.rv == 0) ? 0 : (d.rv == 1) ? 13 : (d.rv == 2) ? 25 : 43);
  g.k0          = (int)(((long long)num * g.Ncb) / g.N) * Z;
  g.f1          = (bgK - 2) * Z;
  g.F           = d.nof_filler_bits;
  g.f0          = g.f1 - g.F;
  g.L           = g.Ncb - g.F;
  int k0p       = (g.k0 >= g.f0 && g.k0 < g.f1) ? g.f1 : g.k0;
  g.r0          = (k0p < g.f0) ? k0p : k0p - g.F;
  g.E           = (int)d.E;
  g.mod         = d.mod;
  g.Kq          = g.E / g.mod;
  return g;
}

// AVX2 combine: adds_epi8 then clamp to +-120.
__device__ __forceinline__ int combine(int a, int b)
{
  int s = a + b;
  return min(max(s, -120), 120);
}

__global__ void __launch_bounds__(256)
rate_dematch_kernel(const miphy_ldpc_rdm_desc* __restrict__ descs, const int8_t* __restrict__ in_base, int8_t* __restrict__ out_base)
{
  const miphy_ldpc_rdm_desc d = descs[blockIdx.y];
  const rm_geom             g = make_geom(d);
  const int8_t*             in  = in_base + d.in_offset;
  int8_t*                   out = out_base + d.out_offset;
  const bool                nd  = d.new_data != 0;

  // Pass-0 bookkeeping for the copy mode (rate_dematcher_impl.cpp:125-198, restated in closed form).
  const int  cap0    = g.L - g.r0; // elements the first pass can take before wrapping
  const bool wrapped = g.E > cap0;
  int        idx_end;
  const int  k0p = (g.k0 >= g.f0 && g.k0 < g.f1) ? g.f1 : g.k0;
  if (k0p < g.f0) {
    if (g.E <= g.f0 - k0p) {
      idx_end = g.f1 % g.Ncb;
    } else {
      int rem = g.E - (g.f0 - k0p);
      int n   = min(g.Ncb - g.f1, rem);
      idx_end = (g.f1 + n) % g.Ncb;
    }
  } else {
    int n   = min(g.Ncb - k0p, g.E);
    idx_end = (k0p + n) % g.Ncb;
  }
  const bool tail_on    = nd && !wrapped && idx_end != 0;
  const int  tail_start = g.N - (g.Ncb - idx_end);

  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < g.N; j += gridDim.x * blockDim.x) {
    const bool in_buf = j < g.Ncb;
    const bool filler = j >= g.f0 && j < g.f1;
    int        acc    = 0;
    bool       write  = false;
    int        i_next = g.E; // first input index still to combine
    if (nd) {
      if (filler) {
        out[j] = 127;
        continue;
      }
      bool has0 = false;
      if (in_buf) {
        const int r = (j < g.f0) ? j : j - g.F;
        int       i0 = r - g.r0;
        if (i0 >= 0) {
          has0 = i0 < g.E;
          if (has0) {
            const int q = i0 / g.Kq;
            acc         = in[(i0 - q * g.Kq) * g.mod + q];
            write       = true;
            i_next      = i0 + g.L;
          }
        } else {
          i_next = i0 + g.L;
        }
      }
      if (!has0) {
        const bool zeroed = (j < g.f0) && ((k0p < g.f0) ? (j < k0p) : true);
        const bool tail   = tail_on && j >= tail_start;
        if (zeroed || tail) {
          acc   = 0;
          write = true;
        } else {
          acc = out[j];
        }
      }
    } else {
      if (!in_buf || filler)
        continue;
      const int r  = (j < g.f0) ? j : j - g.F;
      int       i0 = r - g.r0;
      i0           = (i0 < 0) ? i0 + g.L : i0;
      i_next       = i0;
      if (i_next < g.E)
        acc = out[j];
    }
    for (int i = i_next; i < g.E; i += g.L) {
      const int q = i / g.Kq;
      acc         = combine(acc, in[(i - q * g.Kq) * g.mod + q]);
      write       = true;
    }
    if (write)
      out[j] = (int8_t)acc;
  }
}

__global__ void __launch_bounds__(256)
rate_match_kernel(const miphy_ldpc_rdm_desc* __restrict__ descs, const uint8_t* __restrict__ in_base, uint8_t* __restrict__ out_base)
{
  const miphy_ldpc_rdm_desc d = descs[blockIdx.y];
  const rm_geom             g = make_geom(d);
  const uint8_t*            in  = in_base + d.in_offset;
  uint8_t*                  out = out_base + d.out_offset;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < g.E; o += gridDim.x * blockDim.x) {
    const int i    = o / g.mod; // interleaver: out[i*mod + j] = sel[j*Kq + i]  (rate_matcher_impl.cpp:152-182)
    const int jj   = o - i * g.mod;
    const int x    = g.Kq * jj + i;
    const int rank = (g.r0 + x) % g.L;
    const int pos  = (rank < g.f0) ? rank : rank + g.F;
    out[o]         = in[pos];
  }
}

int check_descs(miphy_ctx* ctx, const miphy_ldpc_rdm_desc* descs, uint32_t n, const char* who)
{
  for (uint32_t i = 0; i < n; ++i) {
    const miphy_ldpc_rdm_desc& d = descs[i];
    MIPHY_REQUIRE(d.bg == 1 || d.bg == 2, "%s: desc %u: invalid base graph", who, i);
    MIPHY_REQUIRE(d.Z <= MIPHY_MAX_Z && ctx->h_tables->z_pos[d.Z] != 0xffff, "%s: desc %u: invalid lifting size %u", who, i, d.Z);
    MIPHY_REQUIRE(d.rv <= 3, "%s: desc %u: RV should be an integer between 0 and 3", who, i);
    MIPHY_REQUIRE(d.mod == 1 || d.mod == 2 || d.mod == 4 || d.mod == 6 || d.mod == 8, "%s: desc %u: invalid modulation order %u", who, i, d.mod);
    MIPHY_REQUIRE(d.E > 0 && d.E % d.mod == 0, "%s: desc %u: length %u is not a multiple of the modulation order", who, i, d.E);
    const unsigned bgK = (d.bg == 1) ? 22 : 10, nshort = (d.bg == 1) ? 66 : 50;
    MIPHY_REQUIRE(d.E <= 8448u * 35u, "%s: desc %u: rate-matched length %u too large", who, i, d.E);
    MIPHY_REQUIRE(d.nof_filler_bits < (bgK - 2) * d.Z, "%s: desc %u: invalid number of filler bits", who, i);
    MIPHY_REQUIRE(d.Nref == 0 || d.Nref > (bgK - 2) * d.Z, "%s: desc %u: Nref %u does not cover the systematic bits", who, i, d.Nref);
    MIPHY_REQUIRE(d.Nref <= 66u * 384u, "%s: desc %u: Nref too large", who, i);
    (void)nshort;
  }
  return MIPHY_OK;
}

} // namespace

extern "C" int miphy_ldpc_rate_dematch_batch(miphy_ctx*                 ctx,
                                             const miphy_ldpc_rdm_desc* descs,
                                             int                        descs_on_device,
                                             uint32_t                   n,
                                             const int8_t*              llr_in,
                                             int8_t*                    softbuf,
                                             void*                      stream)
{
  MIPHY_REQUIRE(ctx && descs && llr_in && softbuf, "miphy_ldpc_rate_dematch_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "rate_dematch: batch too large (max 65535 codeblocks per call)");
  if (!descs_on_device) {
    int rc = check_descs(ctx, descs, n, "rate_dematch");
    if (rc)
      return rc;
  }
  hipStream_t s       = (hipStream_t)stream;
  const void* d_descs = nullptr;
  int         rc      = miphy_stage_descs(ctx, descs, descs_on_device, sizeof(miphy_ldpc_rdm_desc) * (size_t)n, s, &d_descs);
  if (rc)
    return rc;
  hipLaunchKernelGGL(rate_dematch_kernel, dim3(25, n), dim3(256), 0, s, (const miphy_ldpc_rdm_desc*)d_descs, llr_in, softbuf);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

extern "C" int miphy_ldpc_rate_match_batch(miphy_ctx*                 ctx,
                                           const miphy_ldpc_rdm_desc* descs,
                                           int                        descs_on_device,
                                           uint32_t                   n,
                                           const uint8_t*             cb_in,
                                           uint8_t*                   out,
                                           void*                      stream)
{
  MIPHY_REQUIRE(ctx && descs && cb_in && out, "miphy_ldpc_rate_match_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "rate_match: batch too large (max 65535 codeblocks per call)");
  if (!descs_on_device) {
    int rc = check_descs(ctx, descs, n, "rate_match");
    if (rc)
      return rc;
  }
  hipStream_t s       = (hipStream_t)stream;
  const void* d_descs = nullptr;
  int         rc      = miphy_stage_descs(ctx, descs, descs_on_device, sizeof(miphy_ldpc_rdm_desc) * (size_t)n, s, &d_descs);
  if (rc)
    return rc;
  hipLaunchKernelGGL(rate_match_kernel, dim3(16, n), dim3(256), 0, s, (const miphy_ldpc_rdm_desc*)d_descs, cb_in, out);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
