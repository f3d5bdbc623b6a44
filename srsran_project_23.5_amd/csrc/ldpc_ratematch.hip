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
#include "rdm_device.h"
#include <cstdlib>

namespace {

// AVX2 combine: adds_epi8 then clamp to +-120.
__device__ __forceinline__ int combine(int a, int b)
{
  int s = a + b;
  return min(max(s, -120), 120);
}

// One workgroup per codeblock. The E input LLRs are staged once into LDS with coalesced 16-byte loads (when they fit), every
// thread then produces 16 consecutive output positions and writes them with one 16-byte store; the (rare) positions the
// reference leaves untouched fall back to byte stores.
constexpr int RDM_LDS_BYTES = 60 * 1024;
constexpr int RDM_SLOW_CAP  = 16;

struct rdm_ctx {
  rm_geom g;
  bool    nd, wrapped, tail_on;
  int     k0p, tail_start;
};

// Where the LLR of rank index i (position in the de-interleaved sequence, 0 <= i < E) lives.
struct raw_access { // interleaved input as received: in[p * mod + q] with i = q * Kq + p
  const int8_t* in;
  int           Kq, mod;
  __device__ __forceinline__ int operator()(int i) const
  {
    const int q = i / Kq;
    return in[(i - q * Kq) * mod + q];
  }
};
struct lin_access { // de-interleaved copy in LDS: rank index i at byte i
  const int8_t* a;
  __device__ __forceinline__ int operator()(int i) const { return a[i]; }
};
// Sixteen consecutive bytes of an LDS array at ANY byte offset: five aligned dwords, funnel-shifted (the array has four bytes of slack behind it).
__device__ __forceinline__ void lds_load16_unaligned(const int8_t* a, int i, uint32_t (&w)[4])
{
  const uint32_t* p  = reinterpret_cast<const uint32_t*>(a + (i & ~3));
  const uint32_t  sh = (uint32_t)(i & 3);
  uint32_t        d[5];
#pragma unroll
  for (int k = 0; k < 5; ++k)
    d[k] = p[k];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    w[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
}
__device__ __forceinline__ uint32_t combine4(uint32_t o, uint32_t x)
{
  uint32_t r = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b)
    r |= (uint32_t)(combine((int)(int8_t)(o >> (8 * b)), (int)(int8_t)(x >> (8 * b))) & 0xff) << (8 * b);
  return r;
}
// Value of output position j. Returns false when the reference leaves the position untouched.
template <typename IN>
__device__ __forceinline__ bool rdm_value(const rdm_ctx& c, const IN& in, const int8_t* __restrict__ out, int j, int& result)
{
  const rm_geom& g      = c.g;
  const bool     in_buf = j < g.Ncb;
  const bool     filler = j >= g.f0 && j < g.f1;
  int            acc    = 0;
  bool           write  = false;
  int            i_next = g.E;
  if (c.nd) {
    if (filler) {
      result = 127;
      return true;
    }
    bool has0 = false;
    if (in_buf) {
      const int r  = (j < g.f0) ? j : j - g.F;
      const int i0 = r - g.r0;
      if (i0 >= 0) {
        has0 = i0 < g.E;
        if (has0) {
          acc         = in(i0);
          write       = true;
          i_next      = i0 + g.L;
        }
      } else {
        i_next = i0 + g.L;
      }
    }
    if (!has0) {
      const bool zeroed = (j < g.f0) && ((c.k0p < g.f0) ? (j < c.k0p) : true);
      const bool tail   = c.tail_on && j >= c.tail_start;
      if (zeroed || tail) {
        acc   = 0;
        write = true;
      } else {
        acc = out[j];
      }
    }
  } else {
    if (!in_buf || filler)
      return false;
    const int r  = (j < g.f0) ? j : j - g.F;
    int       i0 = r - g.r0;
    i0           = (i0 < 0) ? i0 + g.L : i0;
    i_next       = i0;
    if (i_next < g.E)
      acc = out[j];
  }
  for (int i = i_next; i < g.E; i += g.L) {
    acc   = combine(acc, in(i));
    write = true;
  }
  result = acc;
  return write;
}

constexpr int RDM_PRE = 4; // input vectors per lane that are in flight while the constant regions are being written
template <int MOD>
__device__ __forceinline__ void stage_image(const image_access& img, const int8_t* __restrict__ in, int8_t* __restrict__ lds, int nq, const rm_geom& g,
                                            int tid, int nt, int mb, const uint4 (&pre)[RDM_PRE])
{
  const int adj = g.F - img.gapcut;
#pragma unroll
  for (int k = 0; k < RDM_PRE; ++k)
    if (tid + k * nt < nq)
      stage_vector<MOD>(img, lds, g, adj, tid + k * nt, pre[k]);
  for (int v = tid + RDM_PRE * nt; v < nq; v += nt)
    stage_vector<MOD>(img, lds, g, adj, v, load_in16(in, mb, v));
  for (int k = (nq << 4) + tid; k < g.E; k += nt) {
    const int p = k / MOD, q = k - p * MOD;
    const int r = g.r0 + q * g.Kq + p;
    lds[r - img.jbase + ((r >= g.f0) ? adj : 0)] = in[k];
  }
}

__global__ void __launch_bounds__(256)
rate_dematch_kernel(const miphy_ldpc_rdm_desc* __restrict__ descs, const int8_t* __restrict__ in_base, int8_t* __restrict__ out_base, int lds_bytes)
{
  extern __shared__ __attribute__((aligned(16))) int8_t lds_in[];
  __shared__ int            slow_list[RDM_SLOW_CAP];
  __shared__ int            slow_n;
  const miphy_ldpc_rdm_desc d = descs[blockIdx.x];
  rdm_ctx                   c;
  c.g                         = make_geom(d);
  const rm_geom& g            = c.g;
  const int8_t*  in           = in_base + d.in_offset;
  int8_t*        out          = out_base + d.out_offset;
  c.nd                        = d.new_data != 0;
  // Pass-0 bookkeeping for the copy mode (rate_dematcher_impl.cpp:125-198, restated in closed form).
  const int cap0 = g.L - g.r0; // elements the first pass can take before wrapping
  c.wrapped      = g.E > cap0;
  int idx_end;
  c.k0p = (g.k0 >= g.f0 && g.k0 < g.f1) ? g.f1 : g.k0;
  if (c.k0p < g.f0) {
    if (g.E <= g.f0 - c.k0p) {
      idx_end = g.f1 % g.Ncb;
    } else {
      int rem = g.E - (g.f0 - c.k0p);
      int n   = min(g.Ncb - g.f1, rem);
      idx_end = (g.f1 + n) % g.Ncb;
    }
  } else {
    int n   = min(g.Ncb - c.k0p, g.E);
    idx_end = (c.k0p + n) % g.Ncb;
  }
  c.tail_on    = c.nd && !c.wrapped && idx_end != 0;
  c.tail_start = g.N - (g.Ncb - idx_end);

  const int  tid    = threadIdx.x, nt = blockDim.x;
  const bool single = g.E <= cap0; // E does not wrap around the circular buffer: the overwhelmingly common case
  // Staging. Single-pass geometry: the input is DE-INTERLEAVED while it is staged, straight into an LDS image of the output
  // buffer (slot(j) keeps j mod 16, the filler gap is cut down to its length mod 16), so that the output phase is a copy of
  // aligned 16-byte LDS vectors -- byte gathers out of an interleaved LDS copy put all 64 lanes on one bank.
  // Otherwise (wrap-around combining) the input is staged as received and gathered.
  const int    j_first = (g.r0 < g.f0) ? g.r0 : g.r0 + g.F;
  image_access img;
  img.img    = lds_in;
  img.r0     = g.r0;
  img.f0     = g.f0;
  img.F      = g.F;
  img.f1     = g.f1;
  img.jbase  = j_first & ~15;
  img.gapcut = (j_first < g.f0) ? (g.F & ~15) : 0;
  const bool use_img = single && g.E + 64 <= lds_bytes;
  const bool use_lds = use_img || g.E <= lds_bytes;
  const bool in_vec  = (((uintptr_t)in) & 15) == 0;
  // Region boundaries of the single-pass output (see the output phase below).
  const int r_end = g.r0 + g.E;
  int       B[10];
  B[0] = 0;
  B[1] = (c.k0p < g.f0) ? c.k0p : g.f0;
  B[2] = (g.r0 < g.f0) ? g.r0 : g.f0;
  B[3] = (g.r0 < g.f0) ? min(g.f0, r_end) : g.f0;
  B[4] = g.f0;
  B[5] = g.f1;
  B[6] = max(g.f1, g.r0 + g.F);
  B[7] = max(B[6], r_end + g.F);
  B[8] = c.tail_on ? max(B[7], c.tail_start) : g.N;
  B[9] = g.N;
  const bool fast_out = use_img && (((uintptr_t)out) & 15) == 0;
  // Wrap-around geometry (the E bits run past the end of the circular buffer: retransmissions at redundancy versions 2 and 3, a first
  // transmission longer than the buffer). The input is de-interleaved into a LINEAR LDS array (rank index i at byte i); the 16 positions
  // of an output vector that lies inside the buffer, clear of the filler gap, of the wrap point and of the ends of a pass then take 16
  // consecutive bytes of that array per pass (at a byte offset of any alignment), combined in pass order as the reference's chunks are
  // (ldpc_rate_dematcher_impl.cpp:125-198); the few other vectors run the byte-wise rule. Before: byte-wise for every position, with a
  // division per input element (5 x the time of the single-pass geometry).
  if (!single && g.E + 64 <= lds_bytes) {
    rm_geom      g2 = g;
    image_access im2 = img;
    g2.r0 = 0, g2.f0 = 0x3fffffff, g2.F = 0; // stage_image with these: LDS byte q * Kq + p <- input byte p * mod + q
    im2.jbase = 0, im2.gapcut = 0;
    const int mb = (int)(((uintptr_t)in) & 3);
    const int nq = (mb == 0) ? (g.E >> 4) : ((g.E >= 4) ? ((g.E - 4) >> 4) : 0);
    uint4     pre[RDM_PRE];
#pragma unroll
    for (int k = 0; k < RDM_PRE; ++k)
      pre[k] = (tid + k * nt < nq) ? load_in16(in, mb, tid + k * nt) : make_uint4(0, 0, 0, 0);
    switch (g.mod) {
      case 8:
        stage_image<8>(im2, in, lds_in, nq, g2, tid, nt, mb, pre);
        break;
      case 6:
        stage_image<6>(im2, in, lds_in, nq, g2, tid, nt, mb, pre);
        break;
      case 4:
        stage_image<4>(im2, in, lds_in, nq, g2, tid, nt, mb, pre);
        break;
      case 2:
        stage_image<2>(im2, in, lds_in, nq, g2, tid, nt, mb, pre);
        break;
      default:
        stage_image<1>(im2, in, lds_in, nq, g2, tid, nt, mb, pre);
        break;
    }
    __syncthreads();
    lin_access lin;
    lin.a              = lds_in;
    const bool vec_out = (((uintptr_t)out) & 15) == 0;
    const int  nvec    = (g.N + 15) >> 4;
    for (int v = tid; v < nvec; v += nt) {
      const int j0 = v << 4, j1 = j0 + 15;
      if (vec_out && j1 < g.Ncb && (j1 < g.f0 || j0 >= g.f1)) {
        const int ra = (j0 < g.f0) ? j0 : j0 - g.F;
        int       i0 = ra - g.r0;
        const bool before = i0 < 0; // ranks in front of the starting point: no first pass (copy mode then starts from zero or from the old value)
        i0 += before ? g.L : 0;
        if (i0 + 15 < g.L && !(c.nd && before)) {
          if (i0 >= g.E) {
            if (!c.nd)
              continue; // no input maps here: untouched
          } else {
            bool partial = false;
            for (int i = i0; i < g.E; i += g.L)
              partial |= i + 15 >= g.E;
            if (!partial) {
              uint32_t acc[4];
              int      i = i0;
              if (c.nd) {
                lds_load16_unaligned(lds_in, i, acc);
                i += g.L;
              } else {
                const uint4 old = *reinterpret_cast<const uint4*>(out + j0);
                acc[0] = old.x, acc[1] = old.y, acc[2] = old.z, acc[3] = old.w;
              }
              for (; i < g.E; i += g.L) {
                uint32_t x[4];
                lds_load16_unaligned(lds_in, i, x);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  acc[k] = combine4(acc[k], x[k]);
              }
              *reinterpret_cast<uint4*>(out + j0) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
              continue;
            }
          }
        }
      }
      uint32_t w[4] = {0, 0, 0, 0};
      uint32_t keep = 0;
#pragma unroll 4
      for (int b = 0; b < 16; ++b) {
        const int j  = j0 + b;
        int       r  = 0;
        bool      wr = false;
        if (j < g.N)
          wr = rdm_value(c, lin, out, j, r);
        keep |= (wr ? 0u : 1u) << b;
        w[b >> 2] |= (uint32_t)(r & 0xff) << (8 * (b & 3));
      }
      if (keep == 0 && vec_out && j0 + 16 <= g.N) {
        *reinterpret_cast<uint4*>(out + j0) = make_uint4(w[0], w[1], w[2], w[3]);
      } else {
        for (int b = 0; b < 16; ++b)
          if (!((keep >> b) & 1u) && j0 + b < g.N)
            out[j0 + b] = (int8_t)(w[b >> 2] >> (8 * (b & 3)));
      }
    }
    return;
  }
  if (use_img) {
    const int mb = (int)(((uintptr_t)in) & 3);
    const int nq = (mb == 0) ? (g.E >> 4) : ((g.E >= 4) ? ((g.E - 4) >> 4) : 0);
    uint4     pre[RDM_PRE];
#pragma unroll
    for (int k = 0; k < RDM_PRE; ++k)
      pre[k] = (tid + k * nt < nq) ? load_in16(in, mb, tid + k * nt) : make_uint4(0, 0, 0, 0);
    // While the input is in flight: the regions that do not depend on it (cleared ranges and fillers, copy mode only).
    if (fast_out && c.nd) {
      uint4* dst = reinterpret_cast<uint4*>(out);
#pragma unroll
      for (int k = 0; k <= 8; k += 4) {
        const int      vs = (B[k] + 15) >> 4, ve = B[k + 1] >> 4;
        const uint32_t f  = (k == 4) ? 0x7f7f7f7fu : 0u;
        // streaming stores: most of this is the never-transmitted tail of the circular buffer, nobody reads it back soon
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 fv = {f, f, f, f};
        for (int v = vs + tid; v < ve; v += nt)
          __builtin_nontemporal_store(fv, reinterpret_cast<u32x4*>(dst) + v);
      }
    }
    switch (g.mod) {
      case 8:
        stage_image<8>(img, in, lds_in, nq, g, tid, nt, mb, pre);
        break;
      case 6:
        stage_image<6>(img, in, lds_in, nq, g, tid, nt, mb, pre);
        break;
      case 4:
        stage_image<4>(img, in, lds_in, nq, g, tid, nt, mb, pre);
        break;
      case 2:
        stage_image<2>(img, in, lds_in, nq, g, tid, nt, mb, pre);
        break;
      default:
        stage_image<1>(img, in, lds_in, nq, g, tid, nt, mb, pre);
        break;
    }
    __syncthreads();
  } else if (use_lds) {
    if (in_vec) {
      const uint4* src = reinterpret_cast<const uint4*>(in);
      uint4*       dst = reinterpret_cast<uint4*>(lds_in);
      const int    nq  = g.E >> 4;
      for (int q = tid; q < nq; q += nt)
        dst[q] = src[q];
      for (int k = (nq << 4) + tid; k < g.E; k += nt)
        lds_in[k] = in[k];
    } else {
      for (int k = tid; k < g.E; k += nt)
        lds_in[k] = in[k];
    }
    __syncthreads();
  }
  if (fast_out) {
    // Output phase, single-pass geometry. The buffer is a fixed sequence of (possibly empty) intervals, in this order:
    //   [0,z) cleared | [z,sA) untouched | [sA,eA) data | [eA,f0) untouched | [f0,f1) fillers | [f1,sB) untouched |
    //   [sB,eB) data | [eB,t) untouched | [t,N) cleared                     (rdm_value above, restated per region)
    // (the cleared ranges and the fillers were written above, under the input loads). In combining mode nothing is cleared and
    // the fillers stay. Whole 16-byte vectors inside one interval are produced by a
    // loop without per-vector classification; the (at most 9) vectors that straddle a boundary run the byte-wise rule on
    // 16 lanes side by side.
    if (tid == 0) {
      int n = 0, last = -1;
#pragma unroll
      for (int k = 1; k <= 9; ++k) {
        const int bp = B[k];
        if ((bp & 15) != 0 && bp <= g.N && (bp >> 4) != last) {
          last           = bp >> 4;
          slow_list[n++] = last;
        }
      }
      slow_n = n;
    }
#pragma unroll
    for (int k = 2; k <= 6; k += 4) {
      const int vs = (B[k] + 15) >> 4, ve = B[k + 1] >> 4;
      if (ve <= vs)
        continue;
      uint4* dst = reinterpret_cast<uint4*>(out);
      // data: the image keeps j mod 16, behind the filler gap it is shifted by gapcut (a multiple of 16)
      const uint4* srcv = reinterpret_cast<const uint4*>(lds_in + (vs << 4) - img.jbase - ((k == 6) ? img.gapcut : 0)) - vs;
      if (c.nd) {
        for (int v = vs + tid; v < ve; v += nt)
          dst[v] = srcv[v];
      } else {
        for (int v = vs + tid; v < ve; v += nt) {
          const uint4    x = srcv[v], old = dst[v];
          const uint32_t xw[4] = {x.x, x.y, x.z, x.w}, ow[4] = {old.x, old.y, old.z, old.w};
          uint32_t       w[4];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            uint32_t r = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b)
              r |= (uint32_t)(combine((int)(int8_t)(ow[kk] >> (8 * b)), (int)(int8_t)(xw[kk] >> (8 * b))) & 0xff) << (8 * b);
            w[kk] = r;
          }
          dst[v] = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
    }
    __syncthreads();
    const int n = slow_n;
    for (int idx = tid; idx < 16 * n; idx += nt) {
      const int j = (slow_list[idx >> 4] << 4) + (idx & 15);
      int       r = 0;
      if (j < g.N && rdm_value(c, img, out, j, r))
        out[j] = (int8_t)r;
    }
    return;
  }
  raw_access raw;
  raw.in  = use_lds ? lds_in : in;
  raw.Kq  = g.Kq;
  raw.mod = g.mod;

  const bool vec_out = (((uintptr_t)out) & 15) == 0;
  const int  nvec    = (g.N + 15) >> 4;
  // A 16-byte output vector that lies entirely inside the data interval (ranks [r0, r0 + E)), or entirely outside every
  // written / cleared / combined region, takes a fast path.
  for (int v = tid; v < nvec; v += nt) {
    const int j0 = v << 4;
    if (single && vec_out && j0 + 16 <= g.N) {
      const int  j1      = j0 + 15;
      const bool no_fill = (j1 < g.f0) || (j0 >= g.f1);
      if (no_fill && j1 < g.Ncb) {
        const int ra = (j0 < g.f0) ? j0 : j0 - g.F, rb = ra + 15;
        if (ra >= g.r0 && rb < r_end) { // all data
          uint32_t w[4];
          if (use_img) {
            const uint4 x = *reinterpret_cast<const uint4*>(lds_in + img.slot(j0));
            w[0] = x.x, w[1] = x.y, w[2] = x.z, w[3] = x.w;
          } else {
            int i = ra - g.r0;
            int q = i / g.Kq;
            int p = i - q * g.Kq;
            w[0] = w[1] = w[2] = w[3] = 0;
#pragma unroll
            for (int b = 0; b < 16; ++b) {
              const int x = raw.in[p * g.mod + q];
              w[b >> 2] |= (uint32_t)(x & 0xff) << (8 * (b & 3));
              ++p;
              if (p == g.Kq) {
                p = 0;
                ++q;
              }
            }
          }
          if (!c.nd) {
            const uint4    old   = *reinterpret_cast<const uint4*>(out + j0);
            const uint32_t ow[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              uint32_t r = 0;
#pragma unroll
              for (int b = 0; b < 4; ++b)
                r |= (uint32_t)(combine((int)(int8_t)(ow[k] >> (8 * b)), (int)(int8_t)(w[k] >> (8 * b))) & 0xff) << (8 * b);
              w[k] = r;
            }
          }
          *reinterpret_cast<uint4*>(out + j0) = make_uint4(w[0], w[1], w[2], w[3]);
          continue;
        }
        if (rb < g.r0 || ra >= r_end) { // no input maps here
          if (!c.nd)
            continue; // untouched
          const bool zeroed_all = (j1 < g.f0) && ((c.k0p < g.f0) ? (j1 < c.k0p) : true);
          const bool tail_all   = c.tail_on && j0 >= c.tail_start;
          if (zeroed_all || tail_all) {
            *reinterpret_cast<uint4*>(out + j0) = make_uint4(0, 0, 0, 0);
            continue;
          }
        }
      } else if (j0 >= g.Ncb && (!c.nd || (c.tail_on && j0 >= c.tail_start) || !c.tail_on)) {
        // beyond the circular buffer: only the tail clear can touch it
        if (c.nd && c.tail_on && j0 >= c.tail_start)
          *reinterpret_cast<uint4*>(out + j0) = make_uint4(0, 0, 0, 0);
        if (!(c.nd && c.tail_on && j0 < c.tail_start))
          continue;
      }
    }
    uint32_t w[4] = {0, 0, 0, 0};
    uint32_t keep = 0;
#pragma unroll 4
    for (int b = 0; b < 16; ++b) {
      const int j = j0 + b;
      int       r = 0;
      bool      wr = false;
      if (j < g.N)
        wr = use_img ? rdm_value(c, img, out, j, r) : rdm_value(c, raw, out, j, r);
      keep |= (wr ? 0u : 1u) << b;
      w[b >> 2] |= (uint32_t)(r & 0xff) << (8 * (b & 3));
    }
    if (keep == 0 && vec_out && j0 + 16 <= g.N) {
      *reinterpret_cast<uint4*>(out + j0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
      for (int b = 0; b < 16; ++b)
        if (!((keep >> b) & 1u) && j0 + b < g.N)
          out[j0 + b] = (int8_t)(w[b >> 2] >> (8 * (b & 3)));
    }
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
                                             const miphy_ldpc_rdm_limits* limits,
                                             void*                      stream)
{
  MIPHY_REQUIRE(ctx && descs && llr_in && softbuf, "miphy_ldpc_rate_dematch_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
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
  // LDS staging buffer: the largest rate-matched length in the batch (known from host descriptors or `limits`), capped.
  uint32_t max_E = RDM_LDS_BYTES;
  if (!descs_on_device) {
    max_E = 0;
    for (uint32_t i = 0; i < n; ++i)
      max_E = descs[i].E > max_E ? descs[i].E : max_E;
  } else if (limits) {
    max_E = limits->max_E;
  }
  // + 64: the LDS image of the single-pass path keeps the output alignment (see rate_dematch_kernel)
  const int lds_bytes = (max_E > (uint32_t)RDM_LDS_BYTES) ? 0 : (int)((max_E + 64u + 15u) & ~15u);
  hipLaunchKernelGGL(rate_dematch_kernel, dim3(n), dim3(256), lds_bytes, s, (const miphy_ldpc_rdm_desc*)d_descs, llr_in, softbuf, lds_bytes);
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
