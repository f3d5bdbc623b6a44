// In-LDS Stockham FFT (autosort, decimation in frequency) for N = 2^a * 3^b <= 4096, one workgroup per transform.
// Unnormalised, sign -1 for DIRECT and +1 for INVERSE like srsran::dft_processor
// (include/srsran/phy/generic_functions/dft_processor.h:34-73, lib/phy/generic_functions/dft_processor_generic_impl.cpp:191).
//
// One LDS buffer of N float2: every pass loads its radix-R inputs into registers, barriers, then scatters the outputs
// (two barriers per pass, half the LDS of a ping-pong scheme -> more transforms resident per CU).
#pragma once
#include "miphy_internal.h"

// Two floats as a vector type: additions, subtractions and the complex product below compile to the packed fp32 instructions
// of CDNA3/4 (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32), one instruction for both components.
typedef float cplx __attribute__((ext_vector_type(2)));

// LDS index padding: one extra element after every 8. The radix-8 scatter writes elements 8*t + k from lane t, i.e. with a
// 64-byte stride that would put 32 lanes on two banks; with the pad the stride becomes 72 bytes and a 16-lane group covers
// all banks (measured before: SQ_LDS_BANK_CONFLICT = 52 % of the LDS cycles of the OFDM kernel).
__device__ __forceinline__ int fpad(int i)
{
  return i + (i >> 3);
}
// LDS bytes needed for an N-point buffer.
__host__ __device__ constexpr size_t fft_lds_bytes(size_t N)
{
  return (N + (N >> 3) + 16) * 8; // + the skew of fpad_skew
}
// Padding of the 4096-point OFDM kernels: fpad plus one element after every 512. The radix-8 reads of a pass are 512 elements
// apart (t + 512 k); with fpad alone that is 576 elements = 18 x 256 bytes, i.e. the two addresses of a paired 64-bit read
// (ds_read2st64_b64) fall on the same banks -- measured as LDS bank-conflict cycles 1.6 x the LDS instruction cycles.
__device__ __forceinline__ int fpad_skew(int i)
{
  return i + (i >> 3) + (i >> 9);
}
__device__ __forceinline__ cplx cmul(cplx a, cplx b)
{
  const cplx t = a.yy * cplx{-b.y, b.x};
  return a.xx * b + t;
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b)
{
  return a + b;
}
__device__ __forceinline__ cplx csub(cplx a, cplx b)
{
  return a - b;
}
// multiply by -i (DIRECT) or +i (INVERSE)
template <bool INV>
__device__ __forceinline__ cplx cmul_mi(cplx a)
{
  return INV ? cplx{-a.y, a.x} : cplx{a.y, -a.x};
}
template <bool INV>
__device__ __forceinline__ cplx cconj_if(cplx a)
{
  return INV ? cplx{a.x, -a.y} : a;
}

template <bool INV>
__device__ __forceinline__ void dft2(cplx& a, cplx& b)
{
  cplx t = csub(a, b);
  a      = cadd(a, b);
  b      = t;
}

template <bool INV>
__device__ __forceinline__ void dft4(cplx& a0, cplx& a1, cplx& a2, cplx& a3)
{
  cplx s02 = cadd(a0, a2), d02 = csub(a0, a2);
  cplx s13 = cadd(a1, a3), d13 = cmul_mi<INV>(csub(a1, a3));
  a0 = cadd(s02, s13);
  a2 = csub(s02, s13);
  a1 = cadd(d02, d13);
  a3 = csub(d02, d13);
}

template <bool INV>
__device__ __forceinline__ void dft3(cplx& a0, cplx& a1, cplx& a2)
{
  // w = exp(-+ 2 pi i / 3) = -1/2 -+ i sqrt(3)/2
  const float hs = 0.86602540378443864676f;
  cplx        s  = cadd(a1, a2);
  cplx        d  = csub(a1, a2);
  cplx        m  = {a0.x - 0.5f * s.x, a0.y - 0.5f * s.y};
  cplx        r  = cmul_mi<INV>(cplx{hs * d.x, hs * d.y}); // -+ i * hs * (a1 - a2)
  a0             = cadd(a0, s);
  a1             = cadd(m, r);
  a2             = csub(m, r);
}

template <bool INV>
__device__ __forceinline__ void dft8(cplx* a)
{
  // Two radix-4 on even/odd + twiddles exp(-+ 2 pi i k / 8).
  dft4<INV>(a[0], a[2], a[4], a[6]);
  dft4<INV>(a[1], a[3], a[5], a[7]);
  const float r = 0.70710678118654752440f;
  cplx        t1 = INV ? cplx{r * (a[3].x - a[3].y), r * (a[3].x + a[3].y)} : cplx{r * (a[3].x + a[3].y), r * (a[3].y - a[3].x)};
  cplx        t2 = cmul_mi<INV>(a[5]);
  cplx        t3 = INV ? cplx{-r * (a[7].x + a[7].y), r * (a[7].x - a[7].y)} : cplx{r * (a[7].y - a[7].x), -r * (a[7].x + a[7].y)};
  // outputs: X[k] = E[k] + w^k O[k], X[k+4] = E[k] - w^k O[k], with E = (a0,a2,a4,a6), O = (a1,a3,a5,a7)
  cplx e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6], o0 = a[1];
  a[0] = cadd(e0, o0);
  a[4] = csub(e0, o0);
  a[1] = cadd(e1, t1);
  a[5] = csub(e1, t1);
  a[2] = cadd(e2, t2);
  a[6] = csub(e2, t2);
  a[3] = cadd(e3, t3);
  a[7] = csub(e3, t3);
}

// One Stockham pass of radix R over the N-point buffer `x` (LDS). n = current sub-transform length, s = stride (product of
// the radices already applied). tw = exp(-2 pi i j / N) table in global memory. All threads of the block must call.
// WIDE: the block has at least N / 8 threads (one radix-8 butterfly per thread); otherwise at least N / 16. The register
// arrays are sized for the case, which decides how many transforms a CU holds.
template <int R, bool INV, bool WIDE>
__device__ __forceinline__ void fft_pass(cplx* x, int N, int n, int s, const cplx* __restrict__ tw, int tid, int nt)
{
  const int m  = n / R;
  const int nb = N / R; // butterflies
  // Butterflies per thread: nb <= MAXB * nt.
  constexpr int MAXB = ((R == 8) ? 2 : (R == 4) ? 4 : (R == 2) ? 8 : 6) / (WIDE ? 2 : 1);
  cplx          a[MAXB][R];
  cplx          w1s[MAXB];
  int           ps[MAXB], qs[MAXB];
  // s is a power of two in every radix-8/4/2 pass (those run first): shift instead of a division.
  const bool pow2  = (s & (s - 1)) == 0;
  const int  shift = 31 - __clz(s);
#pragma unroll
  for (int c = 0; c < MAXB; ++c) {
    const int t = tid + c * nt;
    if (t < nb) {
      const int p = pow2 ? (t >> shift) : (t / s), q = t - p * s;
      ps[c] = p, qs[c] = q;
      if (m > 1)
        w1s[c] = tw[p * s]; // issued before the LDS reads and the barrier: its latency hides behind them
#pragma unroll
      for (int k = 0; k < R; ++k)
        a[c][k] = x[fpad(q + s * (p + m * k))];
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < MAXB; ++c) {
    const int t = tid + c * nt;
    if (t < nb) {
      const int p = ps[c], q = qs[c];
      cplx*     v = a[c];
      if (R == 2)
        dft2<INV>(v[0], v[1]);
      else if (R == 3)
        dft3<INV>(v[0], v[1], v[2]);
      else if (R == 4)
        dft4<INV>(v[0], v[1], v[2], v[3]);
      else
        dft8<INV>(v);
      if (m > 1) { // twiddle exp(-+ 2 pi i p k / n) = W_N^(p k s)
        const cplx w1 = cconj_if<INV>(w1s[c]);
        cplx       w  = w1;
#pragma unroll
        for (int k = 1; k < R; ++k) {
          v[k] = cmul(v[k], w);
          if (k + 1 < R)
            w = cmul(w, w1);
        }
      }
#pragma unroll
      for (int k = 0; k < R; ++k)
        x[fpad(q + s * (R * p + k))] = v[k];
    }
  }
  __syncthreads();
}

// Full transform of the N points in LDS buffer x (natural order in, natural order out).
template <bool INV, bool WIDE>
__device__ __forceinline__ void fft_lds_w(cplx* x, int N, const cplx* __restrict__ tw, int tid, int nt)
{
  int n = N, s = 1;
  while (n % 8 == 0) {
    fft_pass<8, INV, WIDE>(x, N, n, s, tw, tid, nt);
    n /= 8;
    s *= 8;
  }
  while (n % 4 == 0) {
    fft_pass<4, INV, WIDE>(x, N, n, s, tw, tid, nt);
    n /= 4;
    s *= 4;
  }
  while (n % 2 == 0) {
    fft_pass<2, INV, WIDE>(x, N, n, s, tw, tid, nt);
    n /= 2;
    s *= 2;
  }
  while (n % 3 == 0) {
    fft_pass<3, INV, WIDE>(x, N, n, s, tw, tid, nt);
    n /= 3;
    s *= 3;
  }
}
template <bool INV>
__device__ __forceinline__ void fft_lds(cplx* x, int N, const cplx* __restrict__ tw, int tid, int nt)
{
  if (N <= 8 * nt)
    fft_lds_w<INV, true>(x, N, tw, tid, nt);
  else
    fft_lds_w<INV, false>(x, N, tw, tid, nt);
}

// N = 4096 = 8^4 on 512 threads (the 100 MHz / 30 kHz symbol): one radix-8 butterfly per thread and pass with every stride a
// compile-time constant, so that the padded LDS addresses of a pass are one base plus immediates (reads: fpad(t) + 576 k; writes:
// 9 t + k, then fpad(q) + 9 S p + (9 S / 8) k). Same butterflies, twiddles and order of operations as fft_lds: identical results.
// IN_REGS: the butterfly's inputs x[t + 512 k] are already in a[] (the first pass of a transform whose input the caller loads from
// global memory in exactly that pattern -- consecutive lanes, consecutive elements); OUT_REGS: the outputs stay in a[] (the last pass
// writes natural order X[t + 512 k], which the caller stores to global memory). Either way the LDS sweep and its barrier are skipped.
template <bool INV, int S, bool SKEW, bool IN_REGS = false, bool OUT_REGS = false>
__device__ __forceinline__ void fft4096_pass(cplx* x, const cplx* __restrict__ tw, int t, cplx* a)
{
  constexpr int M  = 512 / S;            // sub-transform length / 8
  constexpr int RS = SKEW ? 577 : 576;   // read stride: fpad(_skew)(t + 512 k) = fpad(t) + RS k
  const int     p = t / S, q = t % S;
  cplx          w1 = {1.f, 0.f};
  if (M > 1)
    w1 = tw[p * S];
  if (!IN_REGS) {
    const cplx* xr = x + fpad(t);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      a[k] = xr[RS * k];
    if (!OUT_REGS)
      __syncthreads(); // every lane has read before any lane overwrites
  }
  dft8<INV>(a);
  if (M > 1) {
    w1     = cconj_if<INV>(w1);
    cplx w = w1;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      a[k] = cmul(a[k], w);
      if (k + 1 < 8)
        w = cmul(w, w1);
    }
  }
  if (OUT_REGS)
    return;
  // write index q + S (8 p + k), padded: S = 1: 9 t + k [+ t / 64]; S = 8: q + 72 p + 9 k [+ p / 8]; S = 64: fpad(q) + 576 p + 72 k
  // [+ p]; S = 512: fpad(q) + 576 k [+ k]
  constexpr int WS = (S == 1) ? 1 : (S == 512 ? RS : 9 * S / 8);
  int           wb = (S == 1) ? 9 * t : fpad(q) + 9 * S * p;
  if (SKEW)
    wb += (S == 1) ? (t >> 6) : (S == 8 ? (p >> 3) : (S == 64 ? p : 0));
  cplx* xw = x + wb;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    xw[k * WS] = a[k];
  __syncthreads();
}

template <bool INV, bool SKEW = false>
__device__ __forceinline__ void fft4096_lds(cplx* x, const cplx* __restrict__ tw, int t)
{
  cplx a[8];
  fft4096_pass<INV, 1, SKEW>(x, tw, t, a);
  fft4096_pass<INV, 8, SKEW>(x, tw, t, a);
  fft4096_pass<INV, 64, SKEW>(x, tw, t, a);
  fft4096_pass<INV, 512, SKEW>(x, tw, t, a);
}

// The same transform from registers to registers: lane t brings x[t + 512 k] in a[k] and takes X[t + 512 k] away in a[k]. Six LDS sweeps
// and five barriers instead of ten and ten (the LDS buffer must not be in use by the workgroup when it is called). Same butterflies,
// twiddles and order of operations: identical results.
template <bool INV, bool SKEW = false>
__device__ __forceinline__ void fft4096_regs(cplx* x, const cplx* __restrict__ tw, int t, cplx* a)
{
  fft4096_pass<INV, 1, SKEW, true, false>(x, tw, t, a);
  fft4096_pass<INV, 8, SKEW>(x, tw, t, a);
  fft4096_pass<INV, 64, SKEW>(x, tw, t, a);
  fft4096_pass<INV, 512, SKEW, false, true>(x, tw, t, a);
}
