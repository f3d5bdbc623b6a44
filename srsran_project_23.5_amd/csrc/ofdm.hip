// Batched DFT and OFDM slot (de)modulation: one workgroup per OFDM symbol, FFT in LDS, CP / window-offset handling fused
// into the load and the TS 38.211 5.4 phase compensation, window ramp and fftshift mapping fused into the store.
//
// Behaviour contract: lib/phy/lower/modulation/ofdm_demodulator_impl.cpp:93-138, ofdm_modulator_impl.cpp:55-99,
// include/srsran/phy/lower/modulation/phase_compensation_lut.h:49-96, include/srsran/ran/cyclic_prefix.h:96-107.
#include "fft_device.h"
#include <algorithm>
#include <cstdlib>
#include "miphy_ext.h"
#include <cmath>
#include <complex>

#ifndef OFDM_MOD_WAVES
#define OFDM_MOD_WAVES 8
#endif
namespace {

template <bool INV>
__global__ void __launch_bounds__(512) dft_kernel(const float2* __restrict__ in, float2* __restrict__ out, const cplx* __restrict__ tw, int N)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cplx*         x   = reinterpret_cast<cplx*>(smem);
  const float2* src = in + (size_t)blockIdx.x * N;
  float2*       dst = out + (size_t)blockIdx.x * N;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    float2 v   = src[i];
    x[fpad(i)] = {v.x, v.y};
  }
  __syncthreads();
  fft_lds<INV>(x, N, tw, threadIdx.x, blockDim.x);
  for (int i = threadIdx.x; i < N; i += blockDim.x)
    dst[i] = make_float2(x[fpad(i)].x, x[fpad(i)].y);
}

// ---- sizes above 4096 (up to 49152 = 192 x 256): four-step FFT through HBM. N = N1 * N2, n = N2*n1 + n2, k = k1 + N1*k2:
//   X[k1 + N1 k2] = sum_n2 W_N^(n2 k1) W_N2^(n2 k2) [ sum_n1 x[N2 n1 + n2] W_N1^(n1 k1) ]
// Step 1: N1-point transforms down the columns (16 adjacent columns per workgroup so that every row segment is a 128-byte
// access), times the twiddle W_N^(n2 k1), stored as A[k1][n2]. Step 2: N2-point transforms along the rows of A, 16 rows per
// workgroup, scattered to X[k1 + N1 k2] in 128-byte segments.
constexpr int FS_TILE = 16;

template <bool INV>
__global__ void __launch_bounds__(256) dft_fs_step1_kernel(const float2* __restrict__ in,
                                                           float2* __restrict__ tmp,
                                                           const cplx* __restrict__ tw1,
                                                           const cplx* __restrict__ twN,
                                                           int N1,
                                                           int N2)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cplx*         x      = reinterpret_cast<cplx*>(smem);
  const int     stride = (int)(fft_lds_bytes(N1) / 8);
  const size_t  N      = (size_t)N1 * N2;
  const float2* src    = in + (size_t)blockIdx.y * N;
  float2*       dst    = tmp + (size_t)blockIdx.y * N;
  const int     c0     = blockIdx.x * FS_TILE;
  for (int i = threadIdx.x; i < FS_TILE * N1; i += blockDim.x) {
    const int n1 = i / FS_TILE, c = i % FS_TILE;
    const float2 v = src[(size_t)N2 * n1 + c0 + c];
    x[c * stride + fpad(n1)] = {v.x, v.y};
  }
  __syncthreads();
  for (int c = 0; c < FS_TILE; ++c)
    fft_lds<INV>(x + c * stride, N1, tw1, threadIdx.x, blockDim.x);
  for (int i = threadIdx.x; i < FS_TILE * N1; i += blockDim.x) {
    const int k1 = i / FS_TILE, c = i % FS_TILE;
    const int n2 = c0 + c;
    const cplx w = cconj_if<INV>(twN[(size_t)n2 * k1 % N]);
    const cplx v = cmul(x[c * stride + fpad(k1)], w);
    dst[(size_t)k1 * N2 + n2] = make_float2(v.x, v.y);
  }
}

template <bool INV>
__global__ void __launch_bounds__(256)
dft_fs_step2_kernel(const float2* __restrict__ tmp, float2* __restrict__ out, const cplx* __restrict__ tw2, int N1, int N2)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cplx*         x      = reinterpret_cast<cplx*>(smem);
  const int     stride = (int)(fft_lds_bytes(N2) / 8);
  const size_t  N      = (size_t)N1 * N2;
  const float2* src    = tmp + (size_t)blockIdx.y * N;
  float2*       dst    = out + (size_t)blockIdx.y * N;
  const int     r0     = blockIdx.x * FS_TILE;
  for (int i = threadIdx.x; i < FS_TILE * N2; i += blockDim.x) {
    const int r = i / N2, n2 = i % N2;
    const float2 v = src[(size_t)(r0 + r) * N2 + n2];
    x[r * stride + fpad(n2)] = {v.x, v.y};
  }
  __syncthreads();
  for (int r = 0; r < FS_TILE; ++r)
    fft_lds<INV>(x + r * stride, N2, tw2, threadIdx.x, blockDim.x);
  for (int i = threadIdx.x; i < FS_TILE * N2; i += blockDim.x) {
    const int k2 = i / FS_TILE, r = i % FS_TILE;
    const cplx v = x[r * stride + fpad(k2)];
    dst[(size_t)N1 * k2 + r0 + r] = make_float2(v.x, v.y);
  }
}

// Factorisation (both factors multiples of the 16-wide tile) used for the large sizes of the reference's list (dft_processor_generic_impl.cpp:193-210).
bool four_step_factors(uint32_t N, uint32_t& N1, uint32_t& N2)
{
  switch (N) {
    case 4608:
      N1 = 48, N2 = 96;
      return true;
    case 6144:
      N1 = 64, N2 = 96;
      return true;
    case 9216:
      N1 = 96, N2 = 96;
      return true;
    case 12288:
      N1 = 96, N2 = 128;
      return true;
    case 18432:
      N1 = 96, N2 = 192;
      return true;
    case 24576:
      N1 = 128, N2 = 192;
      return true;
    case 36864:
      N1 = 192, N2 = 192;
      return true;
    case 49152:
      N1 = 192, N2 = 256;
      return true;
    default:
      return false;
  }
}

// Each workgroup loops over (slot, symbol) pairs with stride gridDim.x; the launcher starts as many workgroups as the chip
// holds at once. (Keeping the next symbol's samples in flight in registers while transforming the current one was measured:
// the 16 extra registers cost a resident workgroup per CU and the time stayed the same, so the loads are plain.)
template <bool WIDE, int NCT>
__device__ __forceinline__ void ofdm_demod_body(const miphy_ofdm_job* __restrict__ jobs,
                                                const ofdm_plan_dev* __restrict__ plan,
                                                const cplx* __restrict__ tw,
                                                const cplx* __restrict__ ramp,
                                                const float2* __restrict__ samples,
                                                float2* __restrict__ grid,
                                                int total,
                                                int per) // 14: jobs are slots; 1: jobs are single symbols (slot_index = symbol in the subframe)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cplx*     x  = reinterpret_cast<cplx*>(smem);
  const int N  = NCT != 0 ? NCT : plan->N, rg = plan->rg, half = rg / 2;
  const int nt = NCT != 0 ? NCT / 8 : (int)blockDim.x;
  auto pad = [](int i) { return NCT == 4096 ? fpad_skew(i) : fpad(i); };
  auto symbol = [&](int idx) {
    const miphy_ofdm_job& job = jobs[idx / per];
    const int             l   = (per == 1) ? 0 : idx % 14;
    const int             sym = (per == 1) ? (int)job.slot_index : (int)job.slot_index * 14 + l;
    // FFT window: starts `window_offset` samples before the end of the cyclic prefix (demodulator_impl.cpp:115).
    const float2* src = samples + job.samples_offset + ((per == 1) ? 0 : plan->sym_off[sym]) + plan->cp_len[sym] - plan->window_offset;
    if (NCT == 4096) {
      // Register to register: lane t loads samples t + 512 k (consecutive lanes, consecutive samples) straight into its first
      // butterfly and stores bins t + 512 k straight from its last one -- the staging sweeps of the LDS buffer before the first and
      // after the last pass, and their barriers, are gone (ten LDS sweeps and ten barriers per symbol before, six and five now).
      const int t = threadIdx.x;
      cplx      a[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float2 v = src[t + 512 * k];
        a[k]           = {v.x, v.y};
      }
      fft4096_regs<false, true>(x, tw, t, a);
      const cplx coef = {plan->coef_re[sym], plan->coef_im[sym]};
      float2*    dst  = grid + job.grid_offset + (size_t)l * rg;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int bin = t + 512 * k;
        const int sc  = (bin < half) ? bin + half : bin - (4096 - half); // demodulator_impl.cpp:131-137, inverted
        if (bin < half || bin >= 4096 - half) {
          cplx v = cmul(a[k], coef); // sc_prod(dft_output, phase * scale)
          if (ramp)
            v = cmul(v, ramp[bin]);  // window-offset phase ramp (:60-76,127-129)
          dst[sc] = make_float2(v.x, v.y);
        }
      }
      return;
    }
    if ((((uintptr_t)src) & 15) == 0) { // 16-byte loads: two samples per lane
      const float4* src4 = reinterpret_cast<const float4*>(src);
      for (int i = threadIdx.x; i < N / 2; i += nt) {
        const float4 v     = src4[i];
        x[pad(2 * i)]     = {v.x, v.y};
        x[pad(2 * i + 1)] = {v.z, v.w};
      }
    } else {
      for (int i = threadIdx.x; i < N; i += nt) {
        float2 v  = src[i];
        x[pad(i)] = {v.x, v.y};
      }
    }
    __syncthreads();
    if (NCT == 4096)
      fft4096_lds<false, true>(x, tw, threadIdx.x);
    else
      fft_lds_w<false, WIDE>(x, N, tw, threadIdx.x, nt);
    const cplx coef = {plan->coef_re[sym], plan->coef_im[sym]};
    float2*    dst  = grid + job.grid_offset + (size_t)l * rg;
    for (int k = threadIdx.x; k < rg; k += nt) {
      const int bin = (k < half) ? N - half + k : k - half; // demodulator_impl.cpp:131-137
      cplx      v   = cmul(x[pad(bin)], coef);               // sc_prod(dft_output, phase * scale)
      if (ramp)
        v = cmul(v, ramp[bin]);                              // window-offset phase ramp (:60-76,127-129)
      dst[k] = make_float2(v.x, v.y);
    }
  };
  if (NCT != 0) { // one symbol per workgroup, straight-line code: inside a loop over symbols the compiler keeps the addresses
                  // of all four passes live across iterations and spills (252-308 bytes of scratch per lane were measured)
    if ((int)blockIdx.x < total)
      symbol(blockIdx.x);
    return;
  }
  for (int idx = blockIdx.x; idx < total; idx += gridDim.x) {
    symbol(idx);
    __syncthreads(); // x is rewritten by the next symbol
  }
}

// One radix-8 butterfly per thread: 64 registers, so that four 4096-point transforms (LDS-bound) are resident per CU.
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8)))
ofdm_demod_wide_kernel(const miphy_ofdm_job* __restrict__ jobs, const ofdm_plan_dev* __restrict__ plan, const cplx* __restrict__ tw, const cplx* __restrict__ ramp,
                       const float2* __restrict__ samples, float2* __restrict__ grid, int total, int per)
{
  ofdm_demod_body<true, 0>(jobs, plan, tw, ramp, samples, grid, total, per);
}
// The 4096-point symbol of the 100 MHz / 30 kHz carrier on 512 threads: compile-time strides (fft4096_lds).
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8)))
ofdm_demod_4096_kernel(const miphy_ofdm_job* __restrict__ jobs, const ofdm_plan_dev* __restrict__ plan, const cplx* __restrict__ tw, const cplx* __restrict__ ramp,
                       const float2* __restrict__ samples, float2* __restrict__ grid, int total, int per)
{
  ofdm_demod_body<true, 4096>(jobs, plan, tw, ramp, samples, grid, total, per);
}
__global__ void __launch_bounds__(512)
ofdm_demod_kernel(const miphy_ofdm_job* __restrict__ jobs, const ofdm_plan_dev* __restrict__ plan, const cplx* __restrict__ tw, const cplx* __restrict__ ramp,
                  const float2* __restrict__ samples, float2* __restrict__ grid, int total, int per)
{
  ofdm_demod_body<false, 0>(jobs, plan, tw, ramp, samples, grid, total, per);
}

template <bool WIDE, int NCT>
__device__ __forceinline__ void ofdm_mod_body(const miphy_ofdm_job* __restrict__ jobs,
                                              const ofdm_plan_dev* __restrict__ plan,
                                              const cplx* __restrict__ tw,
                                              const float2* __restrict__ grid,
                                              float2* __restrict__ samples,
                                              int per) // 14: jobs are slots (grid x = symbol); 1: jobs are single symbols
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cplx*                x   = reinterpret_cast<cplx*>(smem);
  const miphy_ofdm_job job = jobs[blockIdx.y];
  const int            l   = blockIdx.x;
  const int            N = plan->N, rg = plan->rg;
  const int            sym = (per == 1) ? (int)job.slot_index : (int)job.slot_index * 14 + l;
  const int            cp  = plan->cp_len[sym];
  float2*              dst = samples + job.samples_offset + ((per == 1) ? 0 : plan->sym_off[sym]);
  if (job.grid_empty) { // modulator_impl.cpp:77-80
    for (int i = threadIdx.x; i < N + cp; i += blockDim.x)
      dst[i] = make_float2(0.f, 0.f);
    return;
  }
  const float2* src  = grid + job.grid_offset + (size_t)l * rg;
  const int     half = rg / 2;
  if (NCT == 4096) {
    // Register to register (see the demodulator): lane t gathers bins t + 512 k from the grid row, transforms, and stores samples
    // t + 512 k (and their copies in the cyclic prefix) straight from its last butterfly.
    const int t = threadIdx.x;
    cplx      a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      // bins [0, rg/2) <- upper half of the grid row, bins [N - rg/2, N) <- lower half, the rest zero (:82-86); branch-free: an
      // unused bin loads element 0 and discards it (per-element branches made the compiler carry copies of a[] through scratch)
      const int    i  = t + 512 * k;
      const bool   lo = i < half, hi = i >= 4096 - half;
      const float2 g  = src[lo ? half + i : (hi ? i - (4096 - half) : 0)];
      a[k]            = (lo || hi) ? cplx{g.x, g.y} : cplx{0.f, 0.f};
    }
    fft4096_regs<true, true>(x, tw, t, a);
    const cplx coef = {plan->coef_re[sym], plan->coef_im[sym]};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int    j = t + 512 * k;
      const cplx   v = cmul(a[k], coef);
      const float2 o = make_float2(v.x, v.y);
      dst[cp + j]    = o;
      if (k == 7 && j >= 4096 - cp) // cyclic prefix = copy of the tail (:98); shorter than 512 samples: only the last octet reaches it
        dst[j - (4096 - cp)] = o;
    }
    return;
  }
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    // bins [0, rg/2) <- upper half of the grid, bins [N - rg/2, N) <- lower half, the rest stays zero (:82-86)
    cplx v = {0.f, 0.f};
    if (i < half) {
      float2 g = src[half + i];
      v        = {g.x, g.y};
    } else if (i >= N - half) {
      float2 g = src[i - (N - half)];
      v        = {g.x, g.y};
    }
    x[NCT == 4096 ? fpad_skew(i) : fpad(i)] = v;
  }
  __syncthreads();
  if (NCT == 4096)
    fft4096_lds<true, true>(x, tw, threadIdx.x);
  else
    fft_lds_w<true, WIDE>(x, N, tw, threadIdx.x, blockDim.x);
  const cplx coef = {plan->coef_re[sym], plan->coef_im[sym]};
  for (int i = threadIdx.x; i < N + cp; i += blockDim.x) {
    const int j = (i < cp) ? N - cp + i : i - cp; // cyclic prefix = copy of the tail (:98)
    cplx      v = cmul(x[NCT == 4096 ? fpad_skew(j) : fpad(j)], coef);
    dst[i]      = make_float2(v.x, v.y);
  }
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8)))
ofdm_mod_wide_kernel(const miphy_ofdm_job* __restrict__ jobs, const ofdm_plan_dev* __restrict__ plan, const cplx* __restrict__ tw, const float2* __restrict__ grid,
                     float2* __restrict__ samples, int per)
{
  ofdm_mod_body<true, 0>(jobs, plan, tw, grid, samples, per);
}
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(OFDM_MOD_WAVES, 8)))
ofdm_mod_4096_kernel(const miphy_ofdm_job* __restrict__ jobs, const ofdm_plan_dev* __restrict__ plan, const cplx* __restrict__ tw, const float2* __restrict__ grid,
                     float2* __restrict__ samples, int per)
{
  ofdm_mod_body<true, 4096>(jobs, plan, tw, grid, samples, per);
}
__global__ void __launch_bounds__(512)
ofdm_mod_kernel(const miphy_ofdm_job* __restrict__ jobs, const ofdm_plan_dev* __restrict__ plan, const cplx* __restrict__ tw, const float2* __restrict__ grid,
                float2* __restrict__ samples, int per)
{
  ofdm_mod_body<false, 0>(jobs, plan, tw, grid, samples, per);
}

bool size_supported(uint32_t N)
{
  if (N < 8 || N > 4096)
    return false;
  uint32_t n = N;
  while (n % 2 == 0)
    n /= 2;
  while (n % 3 == 0)
    n /= 3;
  return n == 1;
}

int threads_for(uint32_t N)
{
  // N / 16 threads are the minimum the passes need (fft_device.h); N / 8 puts one radix-8 butterfly on every thread, which
  // halves the latency of a pass and doubles the wavefronts a CU holds per LDS-resident transform.
  int nt = (int)(N / 8);
  nt                     = ((nt + 63) / 64) * 64;
  return nt < 64 ? 64 : (nt > 512 ? 512 : nt);
}

// cyclic_prefix::get_length (normal CP) in samples: (144 >> mu) (+16 for symbol 0 and 7*2^mu) kappa units.
int cp_samples(uint32_t mu, uint32_t sym_sf, uint32_t dft_size)
{
  uint32_t units = 144u >> mu;
  if (sym_sf == 0 || sym_sf == 7u * (1u << mu))
    units += 16;
  // to_samples: units * srate / (15000 * 2048) with srate = 15000 * 2^mu * dft_size
  return (int)((uint64_t)units * (1u << mu) * dft_size / 2048u);
}

int get_plan(miphy_ctx* ctx, const miphy_ofdm_config* c, int is_tx, ofdm_plan_dev** out)
{
  auto key = std::make_tuple(c->numerology, c->bw_rb, c->dft_size, c->nof_samples_window_offset, c->scale, c->center_freq_hz, is_tx);
  auto it  = ctx->ext->plans.find(key);
  if (it != ctx->ext->plans.end()) {
    *out = it->second;
    return MIPHY_OK;
  }
  ofdm_plan_dev h;
  h.N             = (int)c->dft_size;
  h.rg            = (int)c->bw_rb * 12;
  h.window_offset = (int)c->nof_samples_window_offset;
  const int nslots = 1 << c->numerology;
  h.nsymb_sf       = 14 * nslots;
  const double srate = 15000.0 * (double)(1u << c->numerology) * (double)c->dft_size;
  // phase_compensation_lut.h:55-82: cumulative start time of each symbol in the subframe, in double precision.
  unsigned symbol_offset = 0;
  for (int sym = 0; sym < h.nsymb_sf; ++sym) {
    if (sym % 14 == 0) {
      // start of a slot: offsets inside the slot restart
    }
    const int cp   = cp_samples(c->numerology, (uint32_t)sym, c->dft_size);
    h.cp_len[sym]  = cp;
    symbol_offset += (unsigned)cp;
    const double start_time_s = (double)symbol_offset / srate;
    const double phase        = (is_tx ? -1.0 : 1.0) * 2.0 * M_PI * c->center_freq_hz * start_time_s;
    std::complex<float> pc    = static_cast<std::complex<float>>(std::exp(std::complex<double>(0.0, 1.0) * phase));
    std::complex<float> coef  = pc * c->scale; // "phase_compensation * scale" in float (demodulator_impl.cpp:124)
    h.coef_re[sym]            = coef.real();
    h.coef_im[sym]            = coef.imag();
    symbol_offset += c->dft_size;
  }
  for (int slot = 0; slot < nslots; ++slot) {
    int off = 0;
    for (int l = 0; l < 14; ++l) {
      h.sym_off[slot * 14 + l] = off;
      off += h.cp_len[slot * 14 + l] + h.N;
    }
  }
  ofdm_plan_dev* d = nullptr;
  MIPHY_HIP_CHECK(hipMalloc((void**)&d, sizeof(h)));
  MIPHY_HIP_CHECK(hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice));
  ctx->ext->plans[key] = d;
  ctx->ext->to_free.push_back(d);
  *out = d;
  return MIPHY_OK;
}

int get_ramp(miphy_ctx* ctx, uint32_t N, uint32_t offset, const float** out)
{
  auto key = std::make_pair(N, offset);
  auto it  = ctx->ext->ramps.find(key);
  if (it != ctx->ext->ramps.end()) {
    *out = it->second;
    return MIPHY_OK;
  }
  // demodulator_impl.cpp:70-75, in single precision like the reference.
  std::vector<std::complex<float>> r(N);
  const std::complex<float> omega = std::complex<float>(0.f, 1.f) * static_cast<float>(offset) * static_cast<float>(2.0 * M_PI) / static_cast<float>(N);
  for (uint32_t i = 0; i < N; ++i)
    r[i] = std::exp(omega * static_cast<float>(i));
  float* d = nullptr;
  MIPHY_HIP_CHECK(hipMalloc((void**)&d, sizeof(float) * 2 * N));
  MIPHY_HIP_CHECK(hipMemcpy(d, r.data(), sizeof(float) * 2 * N, hipMemcpyHostToDevice));
  ctx->ext->ramps[key] = d;
  ctx->ext->to_free.push_back(d);
  *out = d;
  return MIPHY_OK;
}

int check_cfg(const miphy_ofdm_config* c, const char* who)
{
  MIPHY_REQUIRE(c->numerology <= 2, "%s: numerology %u not supported (0..2)", who, c->numerology);
  MIPHY_REQUIRE(size_supported(c->dft_size), "%s: DFT size %u not supported (2^a*3^b <= 4096)", who, c->dft_size);
  MIPHY_REQUIRE(c->dft_size > c->bw_rb * 12, "%s: the DFT size (%u) must be greater than the resource grid size (%u)", who, c->dft_size, c->bw_rb * 12);
  MIPHY_REQUIRE(std::isnormal(c->scale), "%s: invalid scaling factor", who);
  MIPHY_REQUIRE(c->dft_size % 128 == 0, "%s: DFT size %u gives a non-integer cyclic prefix", who, c->dft_size);
  MIPHY_REQUIRE(c->nof_samples_window_offset < (144 * c->dft_size) / 2048, "%s: the DFT window offset (%u) must be lower than %u", who,
                c->nof_samples_window_offset, (144 * c->dft_size) / 2048);
  return MIPHY_OK;
}

} // namespace

int miphy_get_twiddles(miphy_ctx* ctx, uint32_t N, const float** out)
{
  auto it = ctx->ext->twiddles.find(N);
  if (it != ctx->ext->twiddles.end()) {
    *out = it->second;
    return MIPHY_OK;
  }
  std::vector<float> w(2 * (size_t)N);
  for (uint32_t j = 0; j < N; ++j) {
    const double a = -2.0 * M_PI * (double)j / (double)N;
    w[2 * j]       = (float)std::cos(a);
    w[2 * j + 1]   = (float)std::sin(a);
  }
  float* d = nullptr;
  MIPHY_HIP_CHECK(hipMalloc((void**)&d, sizeof(float) * 2 * N));
  MIPHY_HIP_CHECK(hipMemcpy(d, w.data(), sizeof(float) * 2 * N, hipMemcpyHostToDevice));
  ctx->ext->twiddles[N] = d;
  ctx->ext->to_free.push_back(d);
  *out = d;
  return MIPHY_OK;
}

extern "C" uint32_t miphy_ofdm_slot_size(const miphy_ofdm_config* c, uint32_t slot_index)
{
  if (!c || c->numerology > 2 || slot_index >= (1u << c->numerology) || c->dft_size % 128)
    return 0;
  uint32_t n = 0;
  for (uint32_t l = 0; l < 14; ++l)
    n += (uint32_t)cp_samples(c->numerology, slot_index * 14 + l, c->dft_size) + c->dft_size;
  return n;
}

extern "C" int miphy_dft_batch(miphy_ctx* ctx, uint32_t size, int inverse, uint32_t n, const float* in, float* out, void* stream)
{
  MIPHY_REQUIRE(ctx && in && out, "miphy_dft_batch: null argument");
  hipStream_t s = (hipStream_t)stream;
  uint32_t    N1 = 0, N2 = 0;
  if (!size_supported(size) && four_step_factors(size, N1, N2)) {
    if (n == 0)
      return MIPHY_OK;
    MIPHY_REQUIRE(n <= 65535, "miphy_dft_batch: at most 65535 transforms of size %u per call", size);
    const float *tw1 = nullptr, *tw2 = nullptr, *twN = nullptr;
    int          rc;
    if ((rc = miphy_get_twiddles(ctx, N1, &tw1)) || (rc = miphy_get_twiddles(ctx, N2, &tw2)) || (rc = miphy_get_twiddles(ctx, size, &twN)))
      return rc;
    void* tmp = nullptr;
    if ((rc = miphy_get_workspace(ctx, (size_t)n * size * 8, s, &tmp)))
      return rc;
    const size_t lds1 = FS_TILE * fft_lds_bytes(N1), lds2 = FS_TILE * fft_lds_bytes(N2);
    if (inverse) {
      hipLaunchKernelGGL(dft_fs_step1_kernel<true>, dim3(N2 / FS_TILE, n), dim3(256), lds1, s, (const float2*)in, (float2*)tmp, (const cplx*)tw1,
                         (const cplx*)twN, (int)N1, (int)N2);
      hipLaunchKernelGGL(dft_fs_step2_kernel<true>, dim3(N1 / FS_TILE, n), dim3(256), lds2, s, (const float2*)tmp, (float2*)out, (const cplx*)tw2, (int)N1,
                         (int)N2);
    } else {
      hipLaunchKernelGGL(dft_fs_step1_kernel<false>, dim3(N2 / FS_TILE, n), dim3(256), lds1, s, (const float2*)in, (float2*)tmp, (const cplx*)tw1,
                         (const cplx*)twN, (int)N1, (int)N2);
      hipLaunchKernelGGL(dft_fs_step2_kernel<false>, dim3(N1 / FS_TILE, n), dim3(256), lds2, s, (const float2*)tmp, (float2*)out, (const cplx*)tw2, (int)N1,
                         (int)N2);
    }
    MIPHY_HIP_CHECK(hipGetLastError());
    return MIPHY_OK;
  }
  if (!size_supported(size)) {
    miphy_set_error("miphy_dft_batch: size %u not supported (2^a*3^b <= 4096, or one of 4608 ... 49152)", size);
    return MIPHY_EUNSUPP;
  }
  if (n == 0)
    return MIPHY_OK;
  const float* tw = nullptr;
  int          rc = miphy_get_twiddles(ctx, size, &tw);
  if (rc)
    return rc;
  if (inverse)
    hipLaunchKernelGGL(dft_kernel<true>, dim3(n), dim3(threads_for(size)), fft_lds_bytes(size), s, (const float2*)in, (float2*)out, (const cplx*)tw, (int)size);
  else
    hipLaunchKernelGGL(dft_kernel<false>, dim3(n), dim3(threads_for(size)), fft_lds_bytes(size), s, (const float2*)in, (float2*)out, (const cplx*)tw, (int)size);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

namespace {

// Shared by the slot and the symbol entry points: `per` = 14 (a job is a slot of 14 symbols) or 1 (a job is one OFDM symbol, its
// slot_index field holding the symbol index within the subframe, samples_offset / grid_offset pointing at the symbol itself).
int ofdm_demodulate(miphy_ctx* ctx, const miphy_ofdm_config* cfg, const miphy_ofdm_job* jobs, int jobs_on_device, uint32_t n, const float* samples, float* grid,
                    void* stream, int per, const char* what)
{
  int rc = check_cfg(cfg, what);
  if (rc)
    return rc;
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "%s: at most 65535 jobs per call", what);
  const uint32_t lim = (per == 14) ? (1u << cfg->numerology) : (14u << cfg->numerology);
  if (!jobs_on_device)
    for (uint32_t i = 0; i < n; ++i)
      MIPHY_REQUIRE(jobs[i].slot_index < lim, "%s: job %u: %s index %u out of range", what, i, per == 14 ? "slot" : "symbol", jobs[i].slot_index);
  ofdm_plan_dev* plan = nullptr;
  const float *  tw = nullptr, *ramp = nullptr;
  if ((rc = get_plan(ctx, cfg, 0, &plan)) || (rc = miphy_get_twiddles(ctx, cfg->dft_size, &tw)))
    return rc;
  if (cfg->nof_samples_window_offset && (rc = get_ramp(ctx, cfg->dft_size, cfg->nof_samples_window_offset, &ramp)))
    return rc;
  hipStream_t s      = (hipStream_t)stream;
  const void* d_jobs = nullptr;
  if ((rc = miphy_stage_descs(ctx, jobs, jobs_on_device, sizeof(miphy_ofdm_job) * (size_t)n, s, &d_jobs)))
    return rc;
  const int    nt    = threads_for(cfg->dft_size);
  const bool   wide  = cfg->dft_size <= 8u * (uint32_t)nt;
  const int    total = per * (int)n;
  const size_t lds   = fft_lds_bytes(cfg->dft_size);
  // As many workgroups as the chip holds at once (LDS-bound), each looping over its share of the symbols.
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)(160 * 1024) / (lds + 512)));
  const int nwg    = std::min(total, ctx->num_cus * per_cu);
  if (wide && cfg->dft_size == 4096 && nt == 512)
    hipLaunchKernelGGL(ofdm_demod_4096_kernel, dim3(total), dim3(nt), lds, s, (const miphy_ofdm_job*)d_jobs, plan, (const cplx*)tw, (const cplx*)ramp,
                       (const float2*)samples, (float2*)grid, total, per);
  else if (wide)
    hipLaunchKernelGGL(ofdm_demod_wide_kernel, dim3(nwg), dim3(nt), lds, s, (const miphy_ofdm_job*)d_jobs, plan, (const cplx*)tw, (const cplx*)ramp,
                       (const float2*)samples, (float2*)grid, total, per);
  else
    hipLaunchKernelGGL(ofdm_demod_kernel, dim3(nwg), dim3(nt), lds, s, (const miphy_ofdm_job*)d_jobs, plan, (const cplx*)tw, (const cplx*)ramp,
                       (const float2*)samples, (float2*)grid, total, per);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

int ofdm_modulate(miphy_ctx* ctx, const miphy_ofdm_config* cfg, const miphy_ofdm_job* jobs, int jobs_on_device, uint32_t n, const float* grid, float* samples,
                  void* stream, int per, const char* what)
{
  int rc = check_cfg(cfg, what);
  if (rc)
    return rc;
  MIPHY_REQUIRE(cfg->nof_samples_window_offset == 0, "%s: window offset applies to the demodulator only", what);
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "%s: at most 65535 jobs per call", what);
  const uint32_t lim = (per == 14) ? (1u << cfg->numerology) : (14u << cfg->numerology);
  if (!jobs_on_device)
    for (uint32_t i = 0; i < n; ++i)
      MIPHY_REQUIRE(jobs[i].slot_index < lim, "%s: job %u: %s index %u out of range", what, i, per == 14 ? "slot" : "symbol", jobs[i].slot_index);
  ofdm_plan_dev* plan = nullptr;
  const float*   tw   = nullptr;
  if ((rc = get_plan(ctx, cfg, 1, &plan)) || (rc = miphy_get_twiddles(ctx, cfg->dft_size, &tw)))
    return rc;
  hipStream_t s      = (hipStream_t)stream;
  const void* d_jobs = nullptr;
  if ((rc = miphy_stage_descs(ctx, jobs, jobs_on_device, sizeof(miphy_ofdm_job) * (size_t)n, s, &d_jobs)))
    return rc;
  const int  nt = threads_for(cfg->dft_size);
  const dim3 g(per, n);
  if (cfg->dft_size == 4096 && nt == 512)
    hipLaunchKernelGGL(ofdm_mod_4096_kernel, g, dim3(nt), fft_lds_bytes(cfg->dft_size), s, (const miphy_ofdm_job*)d_jobs, plan, (const cplx*)tw,
                       (const float2*)grid, (float2*)samples, per);
  else if (cfg->dft_size <= 8u * (uint32_t)nt)
    hipLaunchKernelGGL(ofdm_mod_wide_kernel, g, dim3(nt), fft_lds_bytes(cfg->dft_size), s, (const miphy_ofdm_job*)d_jobs, plan, (const cplx*)tw,
                       (const float2*)grid, (float2*)samples, per);
  else
    hipLaunchKernelGGL(ofdm_mod_kernel, g, dim3(nt), fft_lds_bytes(cfg->dft_size), s, (const miphy_ofdm_job*)d_jobs, plan, (const cplx*)tw,
                       (const float2*)grid, (float2*)samples, per);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

} // namespace

extern "C" int miphy_ofdm_demodulate_slots(miphy_ctx* ctx, const miphy_ofdm_config* cfg, const miphy_ofdm_job* jobs, int jobs_on_device, uint32_t n,
                                           const float* samples, float* grid, void* stream)
{
  MIPHY_REQUIRE(ctx && cfg && jobs && samples && grid, "miphy_ofdm_demodulate_slots: null argument");
  return ofdm_demodulate(ctx, cfg, jobs, jobs_on_device, n, samples, grid, stream, 14, "ofdm_demodulate");
}

extern "C" int miphy_ofdm_modulate_slots(miphy_ctx* ctx, const miphy_ofdm_config* cfg, const miphy_ofdm_job* jobs, int jobs_on_device, uint32_t n,
                                         const float* grid, float* samples, void* stream)
{
  MIPHY_REQUIRE(ctx && cfg && jobs && samples && grid, "miphy_ofdm_modulate_slots: null argument");
  return ofdm_modulate(ctx, cfg, jobs, jobs_on_device, n, grid, samples, stream, 14, "ofdm_modulate");
}

extern "C" int miphy_ofdm_demodulate_symbols(miphy_ctx* ctx, const miphy_ofdm_config* cfg, const miphy_ofdm_job* jobs, int jobs_on_device, uint32_t n,
                                             const float* samples, float* grid, void* stream)
{
  MIPHY_REQUIRE(ctx && cfg && jobs && samples && grid, "miphy_ofdm_demodulate_symbols: null argument");
  return ofdm_demodulate(ctx, cfg, jobs, jobs_on_device, n, samples, grid, stream, 1, "ofdm_demodulate_symbols");
}

extern "C" int miphy_ofdm_modulate_symbols(miphy_ctx* ctx, const miphy_ofdm_config* cfg, const miphy_ofdm_job* jobs, int jobs_on_device, uint32_t n,
                                           const float* grid, float* samples, void* stream)
{
  MIPHY_REQUIRE(ctx && cfg && jobs && samples && grid, "miphy_ofdm_modulate_symbols: null argument");
  return ofdm_modulate(ctx, cfg, jobs, jobs_on_device, n, grid, samples, stream, 1, "ofdm_modulate_symbols");
}

extern "C" uint32_t miphy_ofdm_symbol_size(const miphy_ofdm_config* c, uint32_t symbol_index)
{
  // cyclic prefix of symbol `symbol_index` of the subframe + DFT size (cyclic_prefix.h:96-107, ofdm_demodulator.h:64)
  if (!c || c->numerology > 4 || symbol_index >= (14u << c->numerology) || c->dft_size == 0)
    return 0;
  const uint32_t units = (144u >> c->numerology) + ((symbol_index == 0 || symbol_index == (7u << c->numerology)) ? 16u : 0u);
  return (units << c->numerology) * c->dft_size / 2048u + c->dft_size; // kappa units of 1 / (15 kHz * 2048) at a rate of dft_size * 15 kHz * 2^mu
}
