// LDPC decoder, packed variant: every lane owns TWO lifted check rows (l and l + Z/2) and processes them with the packed
// 16-bit VALU instructions of CDNA (v_pk_add/sub/min/max/mad), halving both the instruction count per codeblock and the
// number of wavefronts a codeblock occupies. Same arithmetic contract as ldpc_decode.hip (reference:
// ldpc_decoder_impl.cpp:60-146 + ldpc_decoder_avx2.cpp:66-243, avx2_support.h:65-106); results are bit-identical.
//
// Check-to-variable messages are kept EXPLICITLY in LDS, one int8 per (edge, row), no VALU work to rebuild a message from
// compressed min/argmin/sign state (which dominated an earlier formulation). A lane packs the messages of two consecutive
// edges into one dword {A(j+1), A(j), B(j+1), B(j)} (bytes 0..3): edge j sits in the odd bytes, where v_perm_b32 can
// sign-extend it to two int16 in ONE instruction; edge j+1 needs one extra shift. Per wave the dwords of an edge pair are
// 64 consecutive words: conflict-free, immediate offsets, and only one LDS read and one LDS write per TWO edges (measured on
// gfx950: a DS write costs ~5 cycles of the CU's LDS pipe whatever its width, a read ~2.8 -- tools/lds_probe.hip).
#include "miphy_internal.h"
#include "rdm_device.h"
#include "ldpc_pk_device.h"
#include <algorithm>
#include <cstdlib>

namespace {

// Is the checksum of the first L hard bits zero? Mask / popcount form (crc_zmask in miphy_internal.h): no bit-serial division, no
// position weights. zi = table index of the polynomial, order = its degree.
__device__ __forceinline__ bool block_crc_is_zero(const int8_t* soft, const miphy_graph_tables* __restrict__ tab, int zi, int order, int L,
                                                  uint32_t* red, int tid, int nt)
{
  const int nw = (L + 31) >> 5;
  uint32_t  acc[24];
#pragma unroll
  for (int k = 0; k < 24; ++k)
    acc[k] = 0;
  for (int t = tid; t < nw; t += nt) {
    uint32_t  w   = hard_flags(soft, t);
    const int rem = L - 32 * t;
    if (rem < 32) { // last word: positions 4 q + b >= rem are not message bits
      uint32_t valid = 0;
      for (int q = 0; q < 8; ++q) {
        const int      nb = min(4, max(0, rem - 4 * q));                      // message bits among the four of dword q
        const uint32_t lo = (nb >= 4) ? 0xffffffffu : ((1u << (8 * nb)) - 1u); // their byte lanes
        valid |= (0x01010101u & lo) << q;
      }
      w &= valid;
    }
    const uint4* m = reinterpret_cast<const uint4*>(tab->crc_zmask[zi][nw - 1 - t]);
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      const uint4 mk = m[g];
      acc[4 * g + 0] += __builtin_popcount(w & mk.x);
      acc[4 * g + 1] += __builtin_popcount(w & mk.y);
      acc[4 * g + 2] += __builtin_popcount(w & mk.z);
      acc[4 * g + 3] += __builtin_popcount(w & mk.w);
    }
  }
  uint32_t par = 0;
#pragma unroll
  for (int k = 0; k < 24; ++k)
    par |= (acc[k] & 1u) << k;
  par &= (1u << order) - 1u;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
    par ^= __shfl_xor(par, off);
  if ((tid & 63) == 0)
    red[2 + (tid >> 6)] = par;
  __syncthreads();
  uint32_t crc = 0;
  for (int w = 0; w < (nt >> 6); ++w)
    crc ^= red[2 + w];
  __syncthreads();
  return crc == 0;
}

// CRC over the first L hard bits, words strided over the block's threads.
__device__ __forceinline__ uint32_t block_crc(const int8_t* soft, const miphy_graph_tables* __restrict__ tab, int crc_id, uint32_t poly,
                                              uint32_t order, int K, int L, uint32_t* red, int tid, int nt)
{
  const int      nfull = L >> 5, rbits = L & 31, nwords = (L + 31) >> 5;
  const uint32_t top   = 1u << order;
  uint32_t       part  = 0;
  for (int t = tid; t < nwords; t += nt) {
    const uint32_t w   = hard_word(soft, t, K);
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
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
    part ^= __shfl_xor(part, off);
  if ((tid & 63) == 0)
    red[2 + (tid >> 6)] = part;
  __syncthreads();
  uint32_t crc = 0;
  for (int w = 0; w < (nt >> 6); ++w)
    crc ^= red[2 + w];
  __syncthreads();
  return crc;
}

// Phase timing for tools/ldpc_phase_probe.py (debug build with -DLDPC_PK_PROFILE only; never compiled into libmiphy.so).
#ifdef LDPC_PK_PROFILE
__device__ unsigned long long g_ldpc_prof[2048 * 8]; // per workgroup: 7 phase sums + codeblock count (no atomics: one writer per row)
#define PROF_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PROF_ADD(slot, a, b) prof_acc[slot] += (unsigned long long)((b) - (a))
#define PROF_COUNT()                                              \
  do {                                                            \
    if (tid == 0 && blockIdx.x < 2048) {                          \
      for (int q = 0; q < 7; ++q)                                 \
        g_ldpc_prof[blockIdx.x * 8 + q] += prof_acc[q];           \
      g_ldpc_prof[blockIdx.x * 8 + 7] += 1;                       \
    }                                                             \
    for (int q = 0; q < 7; ++q)                                   \
      prof_acc[q] = 0;                                            \
  } while (0)
#else
#define PROF_T(var)
#define PROF_ADD(slot, a, b)
#define PROF_COUNT()
#endif

// Wavefronts per SIMD the register allocation is held to: 4 (128 registers, five 3-wavefront codeblocks per CU with the messages in
// global memory) measured 11 % faster for the plain decoder. The variant that dematches while loading runs at 3: its load phase is a
// function of its own (pk_fused_load) and the prefetch registers are reassigned unconditionally, which leaves the kernel at 160
// registers without a single spill (3.89 ms per 38 912 codeblocks; 3.95 with the load phase inlined and 20 spills); at 4 wavefronts the
// launcher moves the messages to global memory for a fifth codeblock per CU, and that traffic next to the soft-buffer image stream
// costs more than the occupancy gives (4.63 ms).
#define LDPC_PK_MIN_WAVES_PLAIN 4
#ifndef LDPC_PK_MIN_WAVES_SPLIT
#define LDPC_PK_MIN_WAVES_SPLIT 3 // the latency form: its launches never fill a CU, the register budget is free
#endif
#ifndef LDPC_PK_MIN_WAVES_FUSED
#define LDPC_PK_MIN_WAVES_FUSED 3
#endif
// Raw form of load_in16(): the dword-aligned 16 bytes and the following dword of input vector v, not yet funnel-shifted, so that a
// prefetch keeps them in flight (the shift happens where the vector is consumed).
struct raw_in16 {
  uint32_t a0, a1, a2, a3, e;
};
__device__ __forceinline__ raw_in16 load_in16_raw(const int8_t* __restrict__ in, int mb, int v)
{
  const uint32_t* p4 = reinterpret_cast<const uint32_t*>(in - mb) + 4 * v;
  const rdm_u4    a  = *reinterpret_cast<const rdm_u4*>(p4);
  raw_in16        r;
  r.a0 = a.x, r.a1 = a.y, r.a2 = a.z, r.a3 = a.w;
  r.e  = (mb != 0) ? p4[4] : 0u;
  return r;
}
__device__ __forceinline__ uint4 shift_in16(const raw_in16& r, int mb)
{
  if (mb == 0)
    return make_uint4(r.a0, r.a1, r.a2, r.a3);
  return make_uint4(__builtin_amdgcn_alignbyte(r.a1, r.a0, mb), __builtin_amdgcn_alignbyte(r.a2, r.a1, mb), __builtin_amdgcn_alignbyte(r.a3, r.a2, mb),
                    __builtin_amdgcn_alignbyte(r.e, r.a3, mb));
}
constexpr int PK_PRE = 3; // rate-matched input vectors per lane fetched one codeblock ahead (3 x 192 lanes x 16 B = 9216 B)

// FUSED = the codeblock's first transmission is rate-dematched while it is loaded (ldpc_rate_dematcher_impl.cpp:43-254 for
// new data, redundancy version 0, full circular buffer, E + fillers <= N): `llr_base` is then the HARQ soft-buffer array, which
// receives the dematched codeblock exactly as the reference's dematcher leaves it (data, fillers at +127, zeros behind), and the
// rate-matched LLRs come from `rm_in_base` through the descriptors `rdm` (same index as `descs`). The input of the NEXT codeblock
// of the workgroup is fetched into registers while the current one decodes.
// GMSG = the check-to-variable messages live in global memory instead of LDS (`gmsg`: per resident workgroup
// [wave][edge pair][64 lanes] dwords, `gmsg_pairs` pairs per wave): a lane only ever touches its own messages, one coalesced dword
// per two edges and layer visit, re-read an iteration later out of L2 / Infinity Cache. Low-rate codeblocks (many layers) then
// need LDS for their soft bits only, so that four of them stay resident per CU instead of one.
// The load phase of the variant that dematches while loading, as a function of its own (never inlined): its staging needs far more
// registers than the layer loop, and inlined it decides the register allocation of the whole kernel; out of line the kernel holds
// LDPC_PK_MIN_WAVES_FUSED wavefronts per SIMD with a layer loop free of scratch accesses, and only this phase pays for its spills.
// A plain pointer argument of a function that is not inlined is a generic one (flat_load: both wait counters, aperture test); the
// rate-matched input is global memory, and a parameter that says so gives global_load.
typedef __attribute__((address_space(1))) const int8_t* global_ci8;
__device__ __attribute__((noinline)) void pk_fused_load(int8_t* __restrict__ soft, global_ci8 in_global, int E, int F, int mod, int Z, int bgK, int soft_bytes,
                                                        raw_in16 p0, raw_in16 p1, raw_in16 p2, int tid, int nt)
{
  static_assert(PK_PRE == 3, "three prefetched vectors are passed by value");
  const int8_t* __restrict__ in = (const int8_t*)in_global;
    // ---- rate dematching into the LDS image soft[2Z + j] = buffer position j
    // (the lane index is made opaque here: the compiler would otherwise hoist every per-lane staging address out of the
    // codeblock loop and keep dozens of them alive across the decoder)
    int stid = tid;
    asm volatile("" : "+v"(stid));
    const int                 mb  = (int)(((uintptr_t)in) & 3);
    const int                 nq  = (mb == 0) ? (E >> 4) : ((E >= 4) ? ((E - 4) >> 4) : 0);
    int8_t*                   img = soft + 2 * Z;
    rm_geom                   g;
    g.r0 = 0, g.f1 = (bgK - 2) * Z, g.f0 = g.f1 - F, g.F = F, g.E = E, g.mod = mod, g.Kq = E / mod;
    image_access ia;
    ia.jbase = 0;
    {
      uint4* z4 = reinterpret_cast<uint4*>(soft);
      for (int k = stid; k < (2 * Z) >> 4; k += nt) // the two punctured nodes (Z is a multiple of 16 for every Z >= 128)
        z4[k] = make_uint4(0, 0, 0, 0);
      for (int k = g.f0 + stid; k < g.f1; k += nt)
        img[k] = 127; // fillers
      for (int k = 2 * Z + E + F + stid; k < soft_bytes; k += nt)
        soft[k] = 0; // never transmitted
    }
    switch (mod) {
#define PK_STAGE(MOD)                                                                         \
  case MOD:                                                                                   \
    _Pragma("unroll") for (int k = 0; k < PK_PRE; ++k) if (stid + k * nt < nq)                 \
        stage_vector<MOD>(ia, img, g, F, stid + k * nt, shift_in16(k == 0 ? p0 : (k == 1 ? p1 : p2), mb));               \
    for (int v = stid + PK_PRE * nt; v < nq; v += nt)                                          \
      stage_vector<MOD>(ia, img, g, F, v, load_in16(in, mb, v));                              \
    break;
      PK_STAGE(8)
      PK_STAGE(6)
      PK_STAGE(4)
      PK_STAGE(2)
      default:
        PK_STAGE(1)
#undef PK_STAGE
    }
    for (int k = (nq << 4) + stid; k < E; k += nt) {
      const int p = k / (int)mod, q = k - p * (int)mod;
      const int r = q * g.Kq + p;
      img[r + ((r >= g.f0) ? F : 0)] = in[k];
    }
    __syncthreads();
}

// The image goes to the HARQ soft buffer (what the reference's dematcher leaves there) while the last non-zero soft bit is found
// (ldpc_decoder_impl.cpp:86-99). Inlined into the kernel: a function has to wait for its stores before it returns (25 KB per
// codeblock on their way to HBM), here they drain under the first layers. Returns this lane's candidate for the position behind the
// last non-zero soft bit.
__device__ __forceinline__ int pk_fused_image_out(const int8_t* __restrict__ soft, int8_t* __restrict__ llr_img, int Z, int bgi, int in_len, int tid, int nt)
{
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  int          last = 0;
  u32x4*       dst  = reinterpret_cast<u32x4*>(llr_img);
  const uint4* src  = reinterpret_cast<const uint4*>(soft + 2 * Z);
  const int    nimg = in_len >> 4, nall = (((bgi ? 50 : 66) * Z) >> 4);
  for (int q = tid; q < nimg; q += nt) {
    const uint4 v = src[q];
    const u32x4 o = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(o, dst + q);
    int hi = -1;
    hi     = v.x ? 3 - (__clz((int)v.x) >> 3) : hi;
    hi     = v.y ? 7 - (__clz((int)v.y) >> 3) : hi;
    hi     = v.z ? 11 - (__clz((int)v.z) >> 3) : hi;
    hi     = v.w ? 15 - (__clz((int)v.w) >> 3) : hi;
    last   = (hi >= 0) ? 16 * q + hi + 1 : last;
  }
  const u32x4 zero = {0, 0, 0, 0};
  for (int q = nimg + tid; q < nall; q += nt)
    __builtin_nontemporal_store(zero, dst + q);
  return last;
}

// PARTS > 1 = the latency form for launches that leave most of the chip idle (a single slot is 38 codeblocks on 256 CUs): PARTS times the
// wavefronts per codeblock, the PARTS parts of the workgroup share the edges of every layer (update_rows_pk, PARTS) -- about 1 / PARTS of the
// instructions per wavefront and layer for one more barrier, same results, same LDS image (messages always in LDS).
template <bool FUSED, bool GMSG, int PARTS = 1>
__global__ void __launch_bounds__(192 * PARTS, PARTS > 1 ? LDPC_PK_MIN_WAVES_SPLIT : (FUSED ? LDPC_PK_MIN_WAVES_FUSED : LDPC_PK_MIN_WAVES_PLAIN))
ldpc_decode_pk_kernel(const miphy_ldpc_dec_desc* __restrict__ descs,
                      const miphy_graph_tables* __restrict__ tab,
                      const int8_t* __restrict__ llr_base,
                      uint8_t* __restrict__ out_base,
                      int32_t* __restrict__ iters_out,
                      int max_nodes,
                      const uint32_t* __restrict__ harq_slot,
                      uint8_t* __restrict__ harq_crc_ok,
                      uint32_t n,
                      uint32_t* __restrict__ queue,
                      const miphy_ldpc_rdm_desc* __restrict__ rdm,
                      const int8_t* __restrict__ rm_in_base,
                      uint32_t* __restrict__ gmsg,
                      int gmsg_pairs,
                      int lds_pairs, // GMSG: the first lds_pairs message dwords of a lane stay in LDS (a layer boundary), the other gmsg_pairs are global
                      const uint32_t* __restrict__ cb_order) // optional: the launch decodes codeblocks order[0 .. n) of the arrays (one class of a sorted batch)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int nt  = blockDim.x;
  constexpr bool SPLIT = PARTS > 1;
  static_assert(PARTS == 1 || PARTS == 2 || PARTS == 4, "parts of the latency form");
  static_assert(!(SPLIT && GMSG), "the latency form keeps its messages in LDS");
  const int nth  = nt / PARTS;                   // threads that own rows: the whole workgroup, or each part of it
  // (wave-uniform by construction -- nth is a multiple of 64 --, and said so: the part selects the edges of a layer, which must stay scalar loads)
  const int part = SPLIT ? __builtin_amdgcn_readfirstlane((tid >= nth) + (PARTS > 2 ? (tid >= 2 * nth) + (tid >= 3 * nth) : 0)) : 0;
  const int lr   = tid - part * nth;             // row pair (lr, lr + H) of this lane
  raw_in16  pre[PK_PRE];
  if (FUSED) {
    if (blockIdx.x < n) {
      const miphy_ldpc_rdm_desc r0 = load_words(rdm + (cb_order ? cb_order[blockIdx.x] : blockIdx.x));
      const int8_t*             in = rm_in_base + r0.in_offset;
      const int                 mb = (int)(((uintptr_t)in) & 3), E = (int)r0.E;
      const int                 nq = (mb == 0) ? (E >> 4) : ((E >= 4) ? ((E - 4) >> 4) : 0);
#pragma unroll
      for (int k = 0; k < PK_PRE; ++k) {
        pre[k] = raw_in16{0, 0, 0, 0, 0}; // always assigned: an old value must not stay live across the decode of a codeblock
        if (tid + k * nt < nq)
          pre[k] = load_in16_raw(in, mb, tid + k * nt);
      }
    }
  }
  // Persistent workgroups: the grid is what the chip holds at once; a workgroup decodes codeblock blockIdx.x first and then takes
  // codeblocks gridDim.x, gridDim.x + 1, ... from the launch's queue counter until the batch is exhausted (every wave reaches the
  // exit: the counter only grows). No workgroup launch / LDS allocation between codeblocks, and heterogeneous batches balance.
#ifdef LDPC_PK_PROFILE
  unsigned long long prof_acc[7] = {};
#endif
  for (uint32_t q = blockIdx.x; q < n;) {
  __builtin_amdgcn_s_setprio(LDPC_PK_SETPRIO_OUT); // the phases around the layer loop (load / dematch, CRC, output): ldpc_pk_device.h
  PROF_T(p_start);
  const uint32_t cb = cb_order ? cb_order[q] : q;
  const miphy_ldpc_dec_desc dsc = load_words(descs + cb);
  const int                 Z   = dsc.Z;
  const int                 H   = (Z + 1) >> 1;
  const int                 bgi = (dsc.bg == 1) ? 0 : 1;
  const int                 bgK = bgi ? 10 : 22;
  const int                 bgM = bgi ? 42 : 46;
  const int                 K   = bgK * Z;
  const int                 zp  = tab->z_pos[Z];

  // Soft bits come FIRST in the dynamic LDS block: the kernel has no static LDS, so their base is the constant 0 and folds
  // into the DS instructions (a run-time base would cost one v_add per access). Then the check-to-variable messages,
  // [wave][edge pair][64 lanes] dwords, then a few reduction words. Everything is sized by the layers the batch can reach, so
  // a high-rate batch (4 layers, 76 edges) packs 4 codeblocks = 12 waves per CU.
  int8_t*   soft       = reinterpret_cast<int8_t*>(smem);
  const int lay_alloc  = min(bgM, max(4, max_nodes - bgK));
  const int soft_bytes = ((bgK + lay_alloc) * Z + 15) & ~15;
  const int pairs_all  = tab->pair_start[bgi][lay_alloc];
  const int pairs_lds  = GMSG ? lds_pairs : pairs_all;
  uint32_t* c2v_lane   = reinterpret_cast<uint32_t*>(smem + soft_bytes) + (lr >> 6) * (pairs_lds * 64) + (lr & 63);
  uint32_t* c2v_glob   = GMSG ? gmsg + ((size_t)blockIdx.x * (nt >> 6) + (tid >> 6)) * ((size_t)gmsg_pairs * 64) + (tid & 63) - 64 * (size_t)pairs_lds : nullptr;
  uint32_t* red        = reinterpret_cast<uint32_t*>(smem + soft_bytes) + (nth >> 6) * (pairs_lds * 64);
  // latency form: exchange slots behind the reduction words, [part][3][nth] dwords
  uint32_t* xch = red + 16 + lr;

  // Next codeblock of this workgroup (taken now so that the queue round trip is off the critical path).
  __syncthreads(); // the previous codeblock's readers of red[] / soft[] are done
  if (tid == 0) {
    // One ticket per codeblock processed: the launch draws exactly n tickets (0 .. n - 1), so the workgroup that draws the last one
    // knows nobody else will touch the counter and leaves it at zero for the next launch (no clearing memset per launch, and a
    // captured launch can be replayed).
    const uint32_t ticket = atomicAdd(queue, 1u);
    if (ticket == n - 1u)
      *queue = 0u;
    red[15] = gridDim.x + ticket;
  }
  if (!FUSED && harq_crc_ok && harq_crc_ok[harq_slot[cb]]) {
    if (tid == 0)
      iters_out[cb] = -1;
    __syncthreads();
    q = red[15];
    continue;
  }
  const int8_t* llr    = llr_base + dsc.llr_offset;
  uint8_t*      out    = out_base + dsc.out_offset;
  const int     in_len = (int)dsc.in_len;

  if (tid < 15)
    red[tid] = 0;
  int last = 0;
  // (re)fills pre[] with the first input vectors of codeblock c
  auto prefetch = [&](uint32_t c) {
#pragma unroll
    for (int k = 0; k < PK_PRE; ++k)
      pre[k] = raw_in16{0, 0, 0, 0, 0};
    if (c < n) {
      const miphy_ldpc_rdm_desc rn  = load_words(rdm + (cb_order ? cb_order[c] : c));
      const int8_t*             inn = rm_in_base + rn.in_offset;
      const int                 mbn = (int)(((uintptr_t)inn) & 3), En = (int)rn.E;
      const int                 nqn = (mbn == 0) ? (En >> 4) : ((En >= 4) ? ((En - 4) >> 4) : 0);
#pragma unroll
      for (int k = 0; k < PK_PRE; ++k) {
        pre[k] = raw_in16{0, 0, 0, 0, 0};
        if (tid + k * nt < nqn)
          pre[k] = load_in16_raw(inn, mbn, tid + k * nt);
      }
    }
  };
  if (FUSED) {
    const miphy_ldpc_rdm_desc rd = load_words(rdm + cb);
    pk_fused_load(soft, (global_ci8)(rm_in_base + rd.in_offset), (int)rd.E, (int)dsc.nof_filler_bits, (int)rd.mod, Z, bgK, soft_bytes, pre[0], pre[1], pre[2], tid, nt);
    last = pk_fused_image_out(soft, const_cast<int8_t*>(llr), Z, bgi, in_len, tid, nt);
  } else {
  for (int k = tid; k < 2 * Z; k += nt)
    soft[k] = 0;
  for (int k = 2 * Z + in_len + tid; k < soft_bytes; k += nt)
    soft[k] = 0;
  __syncthreads();
  if ((((uintptr_t)llr | (uintptr_t)(2 * Z)) & 15) == 0) {
    const uint4* src = reinterpret_cast<const uint4*>(llr);
    uint4*       dst = reinterpret_cast<uint4*>(soft + 2 * Z);
    const int    nq  = in_len >> 4;
    for (int q = tid; q < nq; q += nt) {
      const uint4 v = src[q];
      dst[q]        = v;
      int hi = -1;
      hi     = v.x ? 3 - (__clz((int)v.x) >> 3) : hi;
      hi     = v.y ? 7 - (__clz((int)v.y) >> 3) : hi;
      hi     = v.z ? 11 - (__clz((int)v.z) >> 3) : hi;
      hi     = v.w ? 15 - (__clz((int)v.w) >> 3) : hi;
      last   = (hi >= 0) ? 16 * q + hi + 1 : last;
    }
    for (int k = (nq << 4) + tid; k < in_len; k += nt) {
      const int8_t v  = llr[k];
      soft[2 * Z + k] = v;
      last            = (v != 0) ? k + 1 : last;
    }
  } else {
    for (int k = tid; k < in_len; k += nt) {
      const int8_t v  = llr[k];
      soft[2 * Z + k] = v;
      last            = (v != 0) ? k + 1 : last;
    }
  }
  }
  atomicMax(reinterpret_cast<int*>(&red[0]), last);
  __syncthreads();
  last = (int)red[0];

  const bool use_crc = dsc.crc_poly != MIPHY_CRC_NONE;
  const int  kwords  = (K + 31) >> 5;
  if (last == 0) {
    if (!use_crc) {
      for (int b = tid; b < (K + 7) / 8; b += nt) {
        const int rem = K - 8 * b;
        out[b]        = (rem >= 8) ? 0xff : (uint8_t)(0xff << (8 - rem));
      }
    }
    if (tid == 0) {
      iters_out[cb] = 0;
      if (FUSED && harq_crc_ok)
        harq_crc_ok[harq_slot[cb]] = 0;
    }
    if (FUSED)
      prefetch(red[15]);
    q = red[15];
    continue;
  }
  int cb_len = max(last + 2 * Z, K + 4 * Z);
  cb_len     = ((cb_len + Z - 1) / Z) * Z;
  const int nof_layers = cb_len / Z - bgK;

  const uint32_t* edges_g   = tab->edge_sb[bgi][zp];
  // Per-layer constants of the base graph, one layer per lane (loaded once per codeblock): first edge (bits 0-9), degree (10-15),
  // first message pair (16-31). Inside the layer loop they come out with v_readlane instead of three dependent 16-bit loads at the
  // head of every layer.
  uint32_t lay_info = 0;
  {
    const int ml = tid & 63;
    if (ml < bgM) {
      const uint32_t e0 = tab->row_start[bgi][ml];
      lay_info          = e0 | (((uint32_t)tab->row_start[bgi][ml + 1] - e0) << 10) | ((uint32_t)tab->pair_start[bgi][ml] << 16);
    }
  }
  uint32_t        poly = 0, order = 0;
  int             L = 0;
  if (use_crc) {
    poly  = tab->crc_poly[dsc.crc_poly];
    order = tab->crc_order[dsc.crc_poly];
    L     = K - dsc.nof_filler_bits;
  }
  const bool final_only = use_crc && (dsc.flags & 1u);
  // mask form of the CRC zero test where the polynomial has a table (every polynomial a codeblock carries)
  const int zi = (use_crc && L <= 32 * MIPHY_CRC_ZMASK_WORDS) ? miphy_crc_zmask_index(dsc.crc_poly) : -1;

  int       result_iters = 0;
  const int max_iter     = dsc.max_iter;
  PROF_T(p_loaded);
  PROF_ADD(0, p_start, p_loaded);
  // The two wrap constants of the address computations live in VECTOR registers: a VOP2 add / sub with a scalar operand issues at
  // the rate of the three-operand encodings (2.88 instead of 1.93 cycles at three waves per SIMD, tools/valu_probe), and three of
  // the eight address instructions per edge only have Z or Z/2 as their second operand.
  // Odd lifting size: the last lane would own rows H - 1 and Z (= row 0 again, a lane's second row is l + H). Its distance to the
  // second row is set to zero instead: both halves of its packed registers then carry row H - 1, read the same soft bits, compute
  // the same values and store them to the same addresses -- no row is visited twice and no instruction is added.
  int Zv = Z, Hv = (lr + H < Z) ? H : 0;
#ifndef LDPC_PK_SCALAR_WRAP
  asm volatile("" : "+v"(Zv), "+v"(Hv));
#endif
  __builtin_amdgcn_s_setprio(LDPC_PK_SETPRIO1);
  part_edges<PARTS> pe; // latency form: this part's edges of the layer about to run, fetched while the layer before it runs
  if (SPLIT)
    load_part_edges<PARTS>(pe, edges_g, (uint32_t)__builtin_amdgcn_readlane((int)lay_info, 0), part);
  for (int it = 0; it < max_iter; ++it) {
    for (int m = 0; m < nof_layers; ++m) {
      const uint32_t  li    = (uint32_t)__builtin_amdgcn_readlane((int)lay_info, m);
      const int       e0    = (int)(li & 0x3ffu);
      const int       d     = (int)((li >> 10) & 0x3fu);
      const uint32_t* edges = edges_g + 2 * e0;
      PROF_T(p_l0);
      if (SPLIT) { // every thread: the function holds the barrier of the exchange
        part_edges<PARTS> pn;
        const uint32_t    lin  = (uint32_t)__builtin_amdgcn_readlane((int)lay_info, (m + 1 < nof_layers) ? m + 1 : 0);
        auto              next = [&]() { load_part_edges<PARTS>(pn, edges_g, lin, part); };
        uint32_t* cl = c2v_lane + 64 * (li >> 16);
        if (it == 0)
          update_rows_pk_split<true, PARTS>(pe, part, soft, cl, lr, Hv, Zv, lr < H, xch, nth, next);
        else
          update_rows_pk_split<false, PARTS>(pe, part, soft, cl, lr, Hv, Zv, lr < H, xch, nth, next);
        pe = pn;
      } else if (tid < H) {
        if (GMSG && (int)(li >> 16) >= pairs_lds) { // this layer's messages live in global memory (separate code: the address space is part of the instruction)
          uint32_t* cl = c2v_glob + 64 * (li >> 16);
          if (it == 0)
            update_rows_pk_any<true>(d, soft, cl, edges, tid, Hv, Zv);
          else
            update_rows_pk_any<false>(d, soft, cl, edges, tid, Hv, Zv);
        } else {
          uint32_t* cl = c2v_lane + 64 * (li >> 16);
          if (it == 0)
            update_rows_pk_any<true>(d, soft, cl, edges, tid, Hv, Zv);
          else
            update_rows_pk_any<false>(d, soft, cl, edges, tid, Hv, Zv);
        }
      }
      PROF_T(p_l1);
      __syncthreads();
      PROF_T(p_l2);
      PROF_ADD(1, p_l0, p_l1);
      PROF_ADD(2, p_l1, p_l2);
    }
    if (use_crc && !final_only) {
      if (zi >= 0 ? block_crc_is_zero(soft, tab, zi, (int)order, L, red, tid, nt) : block_crc(soft, tab, dsc.crc_poly, poly, order, K, L, red, tid, nt) == 0) {
        result_iters = it + 1;
        break;
      }
    }
  }
  PROF_T(p_dec);
  __builtin_amdgcn_s_setprio(LDPC_PK_SETPRIO_OUT);
  if (FUSED) {
    // The input of this workgroup's next codeblock is requested now: the loads fly while the CRC and the hard decision run, and
    // the registers that hold them are not live inside the layer loop.
    prefetch(red[15]);
  }
  if (final_only)
    result_iters = (zi >= 0 ? block_crc_is_zero(soft, tab, zi, (int)order, L, red, tid, nt) : block_crc(soft, tab, dsc.crc_poly, poly, order, K, L, red, tid, nt) == 0)
                       ? max_iter
                       : 0;
  PROF_T(p_crc);
  PROF_ADD(3, p_dec, p_crc);

  const bool out_aligned = ((uintptr_t)out & 3u) == 0;
  for (int t = tid; t < kwords; t += nt) {
    const uint32_t w      = hard_word(soft, t, K);
    const int      nbytes = min(4, (K - 32 * t + 7) / 8);
    if (nbytes == 4 && out_aligned) {
      reinterpret_cast<uint32_t*>(out)[t] = __builtin_bswap32(w); // MSB-first bytes
    } else {
      for (int q = 0; q < nbytes; ++q)
        out[4 * t + q] = (uint8_t)(w >> (24 - 8 * q));
    }
  }
  if (tid == 0) {
    iters_out[cb] = result_iters;
    // A codeblock that is dematched here is a first transmission: its flag is written either way, which is the reset the caller
    // would otherwise launch before the decoder (pusch_decoder_impl.cpp:146-149).
    if (harq_crc_ok && (FUSED || result_iters > 0))
      harq_crc_ok[harq_slot[cb]] = result_iters > 0;
  }
  PROF_T(p_end);
  PROF_ADD(4, p_crc, p_end);
  PROF_ADD(5, p_start, p_end);
  PROF_COUNT();
  q = red[15];
  } // codeblock loop
}

} // namespace

#ifdef LDPC_PK_PROFILE
extern "C" int miphy_debug_ldpc_profile(unsigned long long out[8], int reset)
{
  static unsigned long long h[2048 * 8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ldpc_prof), sizeof(h)) != hipSuccess)
    return -1;
  for (int q = 0; q < 8; ++q) {
    out[q] = 0;
    for (int b = 0; b < 2048; ++b)
      out[q] += h[b * 8 + q];
  }
  if (reset) {
    for (auto& v : h)
      v = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ldpc_prof), h, sizeof(h)) != hipSuccess)
      return -1;
  }
  return 0;
}
#endif

#ifdef LDPC_PK_PROFILE2
extern "C" int miphy_debug_ldpc_profile2(unsigned long long out[8], int reset)
{
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ldpc_prof2), 8 * sizeof(unsigned long long)) != hipSuccess)
    return -1;
  if (reset) {
    unsigned long long z[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ldpc_prof2), z, sizeof(z)) != hipSuccess)
      return -1;
  }
  return 0;
}
#endif

// LDS bytes the packed kernel needs for a given geometry (Zt >= Z of every codeblock, lay = layer bound, pairs_all = message
// dwords per lane of those layers).
int miphy_ldpc_pk_waves_per_cu(bool fused, int parts)
{
#ifdef LDPC_PK_REPORT_WAVES_FUSED // A-B: the register allocation of one occupancy run at another
  if (fused && parts <= 1)
    return 4 * LDPC_PK_REPORT_WAVES_FUSED;
#endif
  return 4 * (parts > 1 ? LDPC_PK_MIN_WAVES_SPLIT : (fused ? LDPC_PK_MIN_WAVES_FUSED : LDPC_PK_MIN_WAVES_PLAIN)); // what __launch_bounds__ of the kernel guarantees per CU
}

size_t miphy_ldpc_pk_lds_bytes(int bgK, int lay, size_t Zt, int pairs_all, int parts)
{
  const size_t waves = ((Zt + 1) / 2 + 63) / 64;
  return ((((size_t)bgK + lay) * Zt + 15) & ~(size_t)15) + waves * (size_t)pairs_all * 256 + 64 + (parts > 1 ? (size_t)parts * 3 * 64 * waves * 4 : 0);
}

uint32_t miphy_ldpc_pk_grid(const miphy_ctx* ctx, uint32_t n, int threads, size_t lds, bool fused, int parts)
{
  // (threads = those of the launch: the latency form passes twice the row-owning threads)
  // Resident workgroups per CU: LDS, the wavefronts per CU the register budget of the kernel allows (__launch_bounds__), 32 slots.
  const int waves  = threads / 64;
  int       per_cu = (int)((size_t)160 * 1024 / lds);
  per_cu           = std::min(per_cu, miphy_ldpc_pk_waves_per_cu(fused, parts) / waves);
  per_cu           = std::max(per_cu, 1);
  return std::min<uint32_t>(n, (uint32_t)(ctx->num_cus * per_cu));
}

size_t miphy_ldpc_pk_gmsg_bytes(const miphy_ctx* ctx, uint32_t n, int threads, size_t lds, bool fused, int gmsg_pairs)
{
  return gmsg_pairs > 0 ? (size_t)miphy_ldpc_pk_grid(ctx, n, threads, lds, fused) * (threads / 64) * (size_t)gmsg_pairs * 256 : 0;
}

int miphy_ldpc_pk_launch(miphy_ctx* ctx, const miphy_ldpc_dec_desc* d_descs, uint32_t n, int threads, size_t lds, const int8_t* llr,
                         uint8_t* out_bits, int32_t* iters, int nodes_all, const uint32_t* harq_slot, uint8_t* harq_crc_ok, hipStream_t s,
                         const miphy_ldpc_rdm_desc* d_rdm, const int8_t* rm_in, int gmsg_pairs, const uint32_t* d_order, void* gmsg_buf, int parts, int lds_pairs)
{
  const bool fused = d_rdm != nullptr, gm = gmsg_pairs > 0;
  const bool split = parts > 1;
  MIPHY_REQUIRE(parts <= 1 || parts == 2 || parts == 4, "ldpc_decode: the latency form has two or four parts");
  MIPHY_REQUIRE(!(split && gm), "ldpc_decode: the latency form keeps its messages in LDS");
  if (split)
    threads *= parts; // `threads` = the row-owning threads of a codeblock; `lds` already holds the exchange slots
  MIPHY_REQUIRE(threads <= 1024, "ldpc_decode: workgroup of %d threads", threads);
  const void* kern = parts == 4 ? (fused ? (const void*)ldpc_decode_pk_kernel<true, false, 4> : (const void*)ldpc_decode_pk_kernel<false, false, 4>)
                     : split    ? (fused ? (const void*)ldpc_decode_pk_kernel<true, false, 2> : (const void*)ldpc_decode_pk_kernel<false, false, 2>)
                           : fused ? (gm ? (const void*)ldpc_decode_pk_kernel<true, true> : (const void*)ldpc_decode_pk_kernel<true, false>)
                                   : (gm ? (const void*)ldpc_decode_pk_kernel<false, true> : (const void*)ldpc_decode_pk_kernel<false, false>);
  // Above the default 64 KB of dynamic LDS the limit has to be raised; it is a per-device attribute of the kernel, so it is set on
  // every such launch (a cache per thread would be wrong for a thread that drives several devices).
  if (lds > 48 * 1024) {
    MIPHY_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  const uint32_t grid  = miphy_ldpc_pk_grid(ctx, n, threads, lds, fused, parts);
  uint32_t*      queue = nullptr;
  int            rc    = miphy_next_queue_counter(ctx, &queue);
  if (rc)
    return rc;
  void* gmsg = gmsg_buf;
  if (gm && !gmsg && (rc = miphy_get_workspace(ctx, miphy_ldpc_pk_gmsg_bytes(ctx, n, threads, lds, fused, gmsg_pairs), s, &gmsg, 3)))
    return rc;
#define PK_LAUNCH(F, G, ...)                                                                                                                         \
  hipLaunchKernelGGL((ldpc_decode_pk_kernel<F, G, ##__VA_ARGS__>), dim3(grid), dim3(threads), lds, s, d_descs, ctx->d_tables, llr, out_bits, iters, nodes_all, \
                     harq_slot, harq_crc_ok, n, queue, d_rdm, rm_in, (uint32_t*)gmsg, gmsg_pairs, lds_pairs, d_order)
  if (parts == 4 && fused)
    PK_LAUNCH(true, false, 4);
  else if (parts == 4)
    PK_LAUNCH(false, false, 4);
  else if (split && fused)
    PK_LAUNCH(true, false, 2);
  else if (split)
    PK_LAUNCH(false, false, 2);
  else if (fused && gm)
    PK_LAUNCH(true, true);
  else if (fused)
    PK_LAUNCH(true, false);
  else if (gm)
    PK_LAUNCH(false, true);
  else
    PK_LAUNCH(false, false);
#undef PK_LAUNCH
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
