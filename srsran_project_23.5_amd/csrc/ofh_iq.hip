// Open Fronthaul IQ (de)compression -- block floating point and uncompressed fixed point -- between U-plane section payloads and
// resource-grid rows, both in device memory. Behaviour contract: lib/ofh/compression/iq_compression_bfp_impl.cpp:28-143 (generic
// class), iq_compression_bfp_avx2.cpp:31-132 (production arithmetic), iq_compression_none_impl.cpp:29-69, compressed_prb.cpp:31-79
// (bit packing), quantizer.h:34-100, lib/srsvec/conversion.cpp:61-130 (quantisation rounding). HBM-bound byte work:
// (1 +) 3w bytes against 96 bytes per PRB.
#include "miphy_internal.h"

#pragma clang fp contract(off)

namespace {
constexpr int   BFP_THREADS = 64; // one wavefront per workgroup: a 273-PRB section is 12.8 (decompression) / 13 (compression) wavefronts, so
                                  // the ragged last workgroup wastes 2 % of the lanes instead of 20 % with 256 threads
constexpr float Q_GAIN      = 32767.0f; // quantizer(Q_BIT_WIDTH = 16).gain

// ------------------------------------------------------------------------------------------------ decompression
// One thread per group of four REs: 8 samples = 8w bits = w whole bytes at byte 1 + g*w of the PRB record, 32 bytes of output.
__global__ void __launch_bounds__(BFP_THREADS)
ofh_iq_decompress_kernel(const miphy_ofh_iq_job* __restrict__ jobs, const uint8_t* __restrict__ payload, float* __restrict__ grid, int simd_arithmetic)
{
  const miphy_ofh_iq_job& job  = jobs[blockIdx.y];
  const unsigned           unit = blockIdx.x * BFP_THREADS + threadIdx.x; // (prb, group)
  const unsigned           w    = job.data_width;
  if (unit >= min(job.nof_prb, 275u) * 3u || w < 1u || w > 16u || job.compression > MIPHY_OFH_COMPRESSION_BFP) // device-resident jobs are not validated on the host
    return;
  const unsigned prb = unit / 3u, g = unit - prb * 3u;
  const bool     bfp = job.compression == MIPHY_OFH_COMPRESSION_BFP;
  const unsigned hdr = bfp ? 1u : 0u;
  const uint8_t* rec = payload + job.payload_offset + (size_t)prb * (hdr + 3u * w);
  const unsigned e   = bfp ? (rec[0] & 15u) : 0u;
  // The w bytes of the group as a big-endian bit string in four 32-bit words (w <= 16): aligned dword loads + byte alignment.
  const uint8_t*  src   = rec + hdr + g * w;
  const uintptr_t a     = reinterpret_cast<uintptr_t>(src);
  const uint32_t* al    = reinterpret_cast<const uint32_t*>(a & ~uintptr_t(3));
  const unsigned  mis   = (unsigned)(a & 3u);
  const unsigned  ndw   = (mis + w + 3u) >> 2; // dwords that hold bytes of this group (never reads a dword without one)
  uint32_t        raw[5];
#pragma unroll
  for (int k = 0; k < 5; ++k)
    raw[k] = (unsigned)k < ndw ? al[k] : 0u;
  uint32_t be[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    be[k] = __builtin_bswap32(__builtin_amdgcn_alignbyte(raw[k + 1], raw[k], mis));
  float out[8];
  const int   scaler_i = (int)(int16_t)(1 << e); // `int16_t scaler = 1 << exponent` (:107): -32768 for the exponent 15 of 1-bit samples
  const float scaler   = (float)scaler_i;
  // srsvec::convert(int16 -> float): gain = 1 / (32767 / scaler), then a product (iq_compression_bfp_avx2.cpp:126-129)
  const float recip = 1.0f / (Q_GAIN / scaler);
  const bool  mul   = bfp && simd_arithmetic && w == 9u;
  // iq_compression_none_impl::decompress (:53-68): quantizer(data_width).to_float -> division by 2^(w-1) - 1
  const float gain  = bfp ? Q_GAIN : (float)(1 << (w - 1u)) - 1.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned pos = (unsigned)i * w, k = pos >> 5, sh = pos & 31u;
    const uint64_t two = ((uint64_t)be[k] << 32) | (k + 1 < 4 ? be[k + 1] : 0u);
    const uint32_t v   = (uint32_t)(two >> (64u - sh - w)) & ((1u << w) - 1u);
    const int      s   = ((int)(v << (32u - w))) >> (32u - w); // quantizer::sign_extend
    out[i]             = mul ? (float)s * recip : (float)(s * scaler_i) / gain;
  }
  float4* dst = reinterpret_cast<float4*>(grid + 2 * (job.grid_offset + (size_t)prb * 12u + g * 4u));
  if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
    typedef float vec4 __attribute__((ext_vector_type(4)));
    vec4* d4 = reinterpret_cast<vec4*>(dst); // streamed: the grid is consumed by a later kernel, not by this one
    __builtin_nontemporal_store(vec4{out[0], out[1], out[2], out[3]}, d4);
    __builtin_nontemporal_store(vec4{out[4], out[5], out[6], out[7]}, d4 + 1);
  } else {
    float2* d2 = reinterpret_cast<float2*>(dst);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      d2[i] = make_float2(out[2 * i], out[2 * i + 1]);
  }
}

// ------------------------------------------------------------------------------------------------ compression
// srsran_simd_convert_2f_s_round: _mm256_round_ps(nearest even) + cvtps_epi32 (integer indefinite when out of range) + packs
__device__ __forceinline__ int quantize_simd(float a)
{
  const float r = __builtin_rintf(a);
  const int   v = (r > -2147483904.0f && r < 2147483648.0f) ? (int)r : (int)0x80000000;
  return v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
}
// gen_conversion_helper<true>: static_cast<int16_t>(std::round(a)) as x86-64 executes it (32-bit conversion, low half)
__device__ __forceinline__ int quantize_tail(float a)
{
  const float r = __builtin_roundf(a);
  const int   v = (r > -2147483904.0f && r < 2147483648.0f) ? (int)r : (int)0x80000000;
  return (int)(int16_t)(uint16_t)(v & 0xffff);
}

// Three lanes per PRB, four REs = eight samples = one w-byte group each (21 PRBs per wavefront, lane 63 idles): 32-byte reads,
// min / max of the PRB over the three lanes by shuffles, and every lane packs the group it already holds.
constexpr unsigned BFP_PRB_PER_WAVE = 21;

__global__ void __launch_bounds__(BFP_THREADS)
ofh_iq_compress_kernel(const miphy_ofh_iq_job* __restrict__ jobs, const float* __restrict__ grid, float iq_scaling, uint8_t* __restrict__ payload)
{
  const miphy_ofh_iq_job& job  = jobs[blockIdx.y];
  const unsigned           lane = threadIdx.x, w = job.data_width;
  const unsigned           prb = blockIdx.x * BFP_PRB_PER_WAVE + lane / 3u, g = lane % 3u;
  const unsigned           nprb = min(job.nof_prb, 275u);
  const bool               live = lane < 3u * BFP_PRB_PER_WAVE && prb < nprb && w >= 8u && w <= 16u && job.compression <= MIPHY_OFH_COMPRESSION_BFP; // also guards device-resident jobs
  const bool               bfp  = job.compression == MIPHY_OFH_COMPRESSION_BFP;
  // quantizer::to_fixed_point(span): scale = gain * in_scale. BFP quantises the whole job to 16 bits in one conversion, the
  // uncompressed format PRB by PRB to data_width bits (iq_compression_none_impl.cpp:36-49): that decides where the scalar tail
  // of srsvec::convert_round falls.
  const float    scale    = (bfp ? Q_GAIN : (float)(1 << (w - 1u)) - 1.0f) * iq_scaling;
  const unsigned simd_len = bfp ? ((24u * nprb) >> 4) << 4 : 16u;
  int                      q[8];
  int                      mx = -32768, mn = 32767;
  if (live) {
    const float* src = grid + 2 * (job.grid_offset + (size_t)prb * 12u + g * 4u);
    float        v[8];
    if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
      const float4 a = reinterpret_cast<const float4*>(src)[0], b = reinterpret_cast<const float4*>(src)[1];
      v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float2 c = reinterpret_cast<const float2*>(src)[i];
        v[2 * i] = c.x, v[2 * i + 1] = c.y;
      }
    }
    const unsigned e0 = (bfp ? prb * 24u : 0u) + g * 8u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      q[i] = e0 + (unsigned)i < simd_len ? quantize_simd(v[i] * scale) : quantize_tail(v[i] * scale);
      mx = max(mx, q[i]), mn = min(mn, q[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      q[i] = 0;
  }
  // min / max of the PRB over its three lanes (lane 63 reads itself)
  const int base = (int)min(lane - g, 60u);
  mx = max(max(__shfl(mx, base), __shfl(mx, base + 1)), __shfl(mx, base + 2));
  mn = min(min(__shfl(mn, base), __shfl(mn, base + 1)), __shfl(mn, base + 2));
  if (!live)
    return;
  // iq_compression_bfp_impl::compress_prb_generic (:61-64) and determine_exponent (:28-41)
  const int      a = abs(mx), b = abs(mn) - 1;
  const unsigned max_abs   = (unsigned)(a > b ? a : b) & 0xffffu;
  const unsigned max_shift = 16u - w;
  unsigned       lz        = max_shift;
  if (max_abs > 0 && max_shift > 0)
    lz = (unsigned)__clz((int)max_abs) - 16u - 1u;
  const int      raw      = (int)min(max_shift, lz);
  const int      ex       = (int)max_shift - raw;
  const unsigned exponent = bfp && ex > 0 ? (unsigned)ex : 0u;
  const unsigned hdr      = bfp ? 1u : 0u;
  uint8_t*       rec      = payload + job.payload_offset + (size_t)prb * (hdr + 3u * w);
  if (bfp && g == 0u)
    rec[0] = (uint8_t)exponent;
  // 8 samples x w bits -> w bytes, MSB first (compressed_prb::pack_compressed_data): the low w bits of the shifted samples
  const unsigned    mask = (1u << w) - 1u;
  unsigned __int128 bits = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    bits = (bits << w) | ((unsigned)(q[i] >> exponent) & mask);
  uint8_t* dst = rec + hdr + g * w;
  // the first eight bytes in one (unaligned) 64-bit store, the remaining w - 8 one by one
  const uint64_t head = __builtin_bswap64((uint64_t)(bits >> (8u * (w - 8u))));
  __builtin_memcpy(dst, &head, 8);
  for (unsigned k = 8; k < w; ++k)
    dst[k] = (uint8_t)(bits >> (8u * (w - 1u - k)));
}

int check_jobs(const miphy_ofh_iq_job* jobs, uint32_t n, uint32_t min_width, uint32_t* max_prb)
{
  *max_prb = 0;
  for (uint32_t i = 0; i < n; ++i) {
    MIPHY_REQUIRE(jobs[i].nof_prb >= 1 && jobs[i].nof_prb <= 275, "ofh_iq: job %u: %u PRBs out of range", i, jobs[i].nof_prb);
    MIPHY_REQUIRE(jobs[i].data_width >= min_width && jobs[i].data_width <= 16, "ofh_iq: job %u: data width %u not supported (%u..16)", i,
                  (unsigned)jobs[i].data_width, min_width);
    if (jobs[i].compression != MIPHY_OFH_COMPRESSION_NONE && jobs[i].compression != MIPHY_OFH_COMPRESSION_BFP) {
      miphy_set_error("ofh_iq: job %u: compression method %u is not implemented (the reference implements none and BFP)", i, (unsigned)jobs[i].compression);
      return MIPHY_EUNSUPP;
    }
    *max_prb = jobs[i].nof_prb > *max_prb ? jobs[i].nof_prb : *max_prb;
  }
  return MIPHY_OK;
}
} // namespace

extern "C" uint32_t miphy_ofh_iq_record_bytes(uint32_t compression, uint32_t data_width)
{
  return (compression == MIPHY_OFH_COMPRESSION_BFP ? 1u : 0u) + 3u * data_width;
}

extern "C" int miphy_ofh_iq_decompress_batch(miphy_ctx* ctx, const miphy_ofh_iq_job* jobs, int jobs_on_device, uint32_t n, const uint8_t* payload,
                                              float* grid, int simd_arithmetic, void* stream)
{
  MIPHY_REQUIRE(ctx && jobs && payload && grid, "miphy_ofh_iq_decompress_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "ofh_iq_decompress: at most 65535 jobs per call");
  uint32_t max_prb = 275; // device-resident jobs: the grid covers the largest section, the kernel bounds each job itself
  int      rc;
  if (!jobs_on_device && (rc = check_jobs(jobs, n, 1, &max_prb)))
    return rc;
  hipStream_t s      = (hipStream_t)stream;
  const void* d_jobs = nullptr;
  if ((rc = miphy_stage_descs(ctx, jobs, jobs_on_device, sizeof(miphy_ofh_iq_job) * (size_t)n, s, &d_jobs)))
    return rc;
  hipLaunchKernelGGL(ofh_iq_decompress_kernel, dim3((max_prb * 3 + BFP_THREADS - 1) / BFP_THREADS, n), dim3(BFP_THREADS), 0, s,
                     (const miphy_ofh_iq_job*)d_jobs, payload, grid, simd_arithmetic);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

extern "C" int miphy_ofh_iq_compress_batch(miphy_ctx* ctx, const miphy_ofh_iq_job* jobs, int jobs_on_device, uint32_t n, const float* grid,
                                            float iq_scaling, uint8_t* payload, void* stream)
{
  MIPHY_REQUIRE(ctx && jobs && payload && grid, "miphy_ofh_iq_compress_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  MIPHY_REQUIRE(n <= 65535, "ofh_iq_compress: at most 65535 jobs per call");
  uint32_t max_prb = 275;
  int      rc;
  if (!jobs_on_device && (rc = check_jobs(jobs, n, 8, &max_prb)))
    return rc;
  hipStream_t s      = (hipStream_t)stream;
  const void* d_jobs = nullptr;
  if ((rc = miphy_stage_descs(ctx, jobs, jobs_on_device, sizeof(miphy_ofh_iq_job) * (size_t)n, s, &d_jobs)))
    return rc;
  hipLaunchKernelGGL(ofh_iq_compress_kernel, dim3((max_prb + BFP_PRB_PER_WAVE - 1) / BFP_PRB_PER_WAVE, n), dim3(BFP_THREADS), 0, s,
                     (const miphy_ofh_iq_job*)d_jobs, grid, iq_scaling, payload);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
