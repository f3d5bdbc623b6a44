// Zero-forcing channel equalizer as a block of its own. Behaviour contract: channel_equalizer_zf_impl.cpp:123-162 (dispatch and
// the choice of the noise variance), equalize_zf_1xn.h:120-158 (one layer, scalar path: exact reciprocal) and equalize_zf_2x2.cpp:30-117.
// HBM-bound streaming kernel: per resource element 2 * ports * (1 + layers) floats in, 3 * layers floats out, every operand read once
// with consecutive lanes on consecutive elements; grid = (chunks of 1024 elements, jobs).
#include "miphy_internal.h"
#include <math.h>

namespace {

constexpr int EQ_THREADS = 256, EQ_RE_PER_BLOCK = 1024;

__device__ __forceinline__ bool is_normal(float x)
{
  const float a = fabsf(x);
  return a >= 1.17549435e-38f && a < INFINITY; // std::isnormal: neither zero, subnormal, infinite nor NaN
}

__global__ __launch_bounds__(EQ_THREADS) void equalize_kernel(const miphy_equalizer_job* __restrict__ jobs, const float2* __restrict__ ch_symbols,
                                                              const float2* __restrict__ ch_estimates, float2* __restrict__ eq_symbols,
                                                              float* __restrict__ eq_noise_vars)
{
#pragma clang fp contract(off)
  const miphy_equalizer_job j = jobs[blockIdx.y];
  const uint32_t nre = j.nof_re;
  const uint32_t lo  = blockIdx.x * EQ_RE_PER_BLOCK;
  if (lo >= nre)
    return;
  const uint32_t hi = min(nre, lo + EQ_RE_PER_BLOCK);
  const float2*  y  = ch_symbols + j.ch_symbols_offset;
  const float2*  h  = ch_estimates + j.ch_estimates_offset;
  float2*        z  = eq_symbols + j.eq_symbols_offset;
  float*         nv = eq_noise_vars + j.eq_noise_vars_offset;
  const float    noise_var = j.noise_var, tx = j.tx_scaling;
  const bool     nv_ok = is_normal(noise_var) && noise_var > 0.f;
  if (j.nof_tx_layers == 1) {
    const int np = j.nof_rx_ports;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += EQ_THREADS) {
      float ch_mod_sq = 0.f, acc_re = 0.f, acc_im = 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (p < np) {
          const float2 r = y[(size_t)p * nre + i], c = h[(size_t)p * nre + i];
          ch_mod_sq      = ch_mod_sq + (c.x * c.x + c.y * c.y);
          acc_re         = acc_re + (r.x * c.x + r.y * c.y); // re_in * conj(ch_est)
          acc_im         = acc_im + (r.y * c.x - r.x * c.y);
        }
      }
      const float d_pinv = tx * ch_mod_sq;
      float2      o      = make_float2(0.f, 0.f);
      float       v      = INFINITY;
      if (is_normal(d_pinv) && nv_ok) {
        const float rcp = 1.0f / d_pinv;
        o               = make_float2(acc_re * rcp, acc_im * rcp);
        v               = rcp * (noise_var / tx);
      }
      z[i]  = o;
      nv[i] = v;
    }
    return;
  }
  // two layers on two ports
  const float2 *y0 = y, *y1 = y + nre;
  const float2 *h00 = h, *h10 = h + nre, *h01 = h + 2 * (size_t)nre, *h11 = h + 3 * (size_t)nre; // h<port><layer>
  for (uint32_t i = lo + threadIdx.x; i < hi; i += EQ_THREADS) {
    const float2 a = h00[i], b = h01[i], c = h10[i], d = h11[i], r0 = y0[i], r1 = y1[i];
    const float  n0 = (a.x * a.x + a.y * a.y) + (c.x * c.x + c.y * c.y); // squared norm of the layer-0 column
    const float  n1 = (b.x * b.x + b.y * b.y) + (d.x * d.x + d.y * d.y);
    // xi = conj(a) b + conj(c) d
    const float xr = (a.x * b.x + a.y * b.y) + (c.x * d.x + c.y * d.y);
    const float xi = (a.x * b.y - a.y * b.x) + (c.x * d.y - c.y * d.x);
    const float xm = xr * xr + xi * xi;
    // matched inputs: conj(h_p0_l) r0 + conj(h_p1_l) r1
    const float m0r = (a.x * r0.x + a.y * r0.y) + (c.x * r1.x + c.y * r1.y), m0i = (a.x * r0.y - a.y * r0.x) + (c.x * r1.y - c.y * r1.x);
    const float m1r = (b.x * r0.x + b.y * r0.y) + (d.x * r1.x + d.y * r1.y), m1i = (b.x * r0.y - b.y * r0.x) + (d.x * r1.y - d.y * r1.x);
    const float d_pinv  = tx * ((n0 * n1) - xm);
    const float d_nvars = tx * d_pinv;
    float2      o0 = make_float2(0.f, 0.f), o1 = o0;
    float       v0 = INFINITY, v1 = INFINITY;
    if (is_normal(d_pinv) && nv_ok) {
      const float rp = 1.0f / d_pinv, rn = 1.0f / d_nvars;
      // (n1 m0 - xi m1) / d ; (n0 m1 - conj(xi) m0) / d
      o0 = make_float2((n1 * m0r - (xr * m1r - xi * m1i)) * rp, (n1 * m0i - (xr * m1i + xi * m1r)) * rp);
      o1 = make_float2((n0 * m1r - (xr * m0r + xi * m0i)) * rp, (n0 * m1i - (xr * m0i - xi * m0r)) * rp);
      v0 = noise_var * n1 * rn;
      v1 = noise_var * n0 * rn;
    }
    z[i]                = o0;
    z[(size_t)nre + i]  = o1;
    nv[i]               = v0;
    nv[(size_t)nre + i] = v1;
  }
}

} // namespace

extern "C" int miphy_channel_equalize_batch(miphy_ctx* ctx, const miphy_equalizer_job* jobs, int jobs_on_device, uint32_t n, const float* ch_symbols,
                                            const float* ch_estimates, float* eq_symbols, float* eq_noise_vars, void* stream)
{
  MIPHY_REQUIRE(ctx && jobs && ch_symbols && ch_estimates && eq_symbols && eq_noise_vars, "miphy_channel_equalize_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "channel_equalize: at most 65535 jobs per call");
  uint32_t max_re = 275 * 12 * 14; // device-resident jobs: the largest slot
  if (!jobs_on_device) {
    max_re = 0;
    for (uint32_t i = 0; i < n; ++i) {
      const miphy_equalizer_job& j = jobs[i];
      MIPHY_REQUIRE((j.nof_tx_layers == 1 && j.nof_rx_ports >= 1 && j.nof_rx_ports <= 4) || (j.nof_tx_layers == 2 && j.nof_rx_ports == 2),
                    "Invalid channel spatial topology: %u Rx ports, %u Tx layers.", j.nof_rx_ports, j.nof_tx_layers);
      MIPHY_REQUIRE(j.tx_scaling > 0.f, "Tx scaling factor must be positive.");
      max_re = j.nof_re > max_re ? j.nof_re : max_re;
    }
    if (max_re == 0)
      return MIPHY_OK;
  }
  hipStream_t s      = (hipStream_t)stream;
  const void* d_jobs = nullptr;
  int         rc     = miphy_stage_descs(ctx, jobs, jobs_on_device, sizeof(miphy_equalizer_job) * (size_t)n, s, &d_jobs);
  if (rc)
    return rc;
  hipLaunchKernelGGL(equalize_kernel, dim3((max_re + EQ_RE_PER_BLOCK - 1) / EQ_RE_PER_BLOCK, n), dim3(EQ_THREADS), 0, s,
                     (const miphy_equalizer_job*)d_jobs, (const float2*)ch_symbols, (const float2*)ch_estimates, (float2*)eq_symbols, eq_noise_vars);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
