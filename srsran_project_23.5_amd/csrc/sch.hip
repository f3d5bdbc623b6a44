// Transport-block level shared-channel entry points: whole pusch_decoder::decode / pdsch_encoder::encode calls in one C call.
//
// Behaviour contract: lib/phy/upper/channel_processors/pusch_decoder_impl.cpp:121-225, pdsch_encoder_impl.cpp:28-65,
// lib/phy/upper/channel_coding/ldpc/ldpc_segmenter_impl.cpp:57-334, include/srsran/phy/upper/channel_coding/ldpc/ldpc.h:128-207.
// The segmentation arithmetic runs on the host (it is bookkeeping) and turns every transport block into
// codeblock descriptors for the batched kernels; two small kernels do the bit plumbing around them (TB -> codeblock
// messages with CRCs and fillers; codeblock messages -> TB with the TB CRC and the result record).
#include "crc_device.h"
#include "miphy_ext.h"
#include "tables/nr_ldpc_tables.h"
#include <cstring>
#include <vector>

namespace {

constexpr uint32_t HARQ_CB_STRIDE  = 66 * 384; // soft bits per codeblock slot
constexpr uint32_t HARQ_MSG_STRIDE = 1056;     // packed message bytes per codeblock slot

struct seg_t {
  uint32_t tbs, nof_tb_crc_bits, nof_cbs, Z, K, N, cb_info_bits, nof_cb_crc_bits, nof_filler_bits, zero_pad, crc_poly;
};

// ldpc.h:128-207 + ldpc_segmenter_impl.cpp:104-141
int segmentation(uint32_t tb_bytes, uint32_t bg, seg_t& s)
{
  MIPHY_REQUIRE(bg == 1 || bg == 2, "sch: invalid base graph %u", bg);
  MIPHY_REQUIRE(tb_bytes > 0 && (uint64_t)tb_bytes * 8 + 24 <= 1277992, "sch: transport block size %u bytes out of range", tb_bytes);
  s.tbs             = tb_bytes * 8;
  s.nof_tb_crc_bits = (s.tbs <= 3824) ? 16 : 24;
  const uint32_t B   = s.tbs + s.nof_tb_crc_bits;
  const uint32_t Kcb = (bg == 1) ? 8448 : 3840;
  s.nof_cbs          = (B <= Kcb) ? 1 : (B + (Kcb - 24) - 1) / (Kcb - 24);
  MIPHY_REQUIRE(s.nof_cbs <= 52, "sch: %u codeblocks exceed MAX_NOF_SEGMENTS", s.nof_cbs);
  const uint32_t Bp = B + ((s.nof_cbs > 1) ? 24 * s.nof_cbs : 0);
  uint32_t       Kb = 22;
  if (bg == 2)
    Kb = (B > 640) ? 10 : (B > 560) ? 9 : (B > 192) ? 8 : 6;
  s.Z = 0;
  for (int i = 0; i < NR_LDPC_NOF_LIFTING_SIZES; ++i)
    if ((uint32_t)NR_LDPC_LIFTING_SIZES[i] * s.nof_cbs * Kb >= Bp) {
      s.Z = NR_LDPC_LIFTING_SIZES[i];
      break;
    }
  MIPHY_REQUIRE(s.Z != 0, "sch: no lifting size fits the transport block");
  s.K               = ((bg == 1) ? 22 : 10) * s.Z;
  s.N               = s.K * ((bg == 1) ? 3 : 5);
  s.nof_cb_crc_bits = (s.nof_cbs > 1) ? 24 : 0;
  s.cb_info_bits    = (Bp + s.nof_cbs - 1) / s.nof_cbs - s.nof_cb_crc_bits;
  s.zero_pad        = (s.cb_info_bits + s.nof_cb_crc_bits) * s.nof_cbs - Bp;
  s.nof_filler_bits = s.K - s.cb_info_bits - s.nof_cb_crc_bits;
  // pusch_decoder_impl.cpp:44-55
  s.crc_poly = (s.nof_cbs > 1) ? MIPHY_CRC24B : ((s.tbs > 3824) ? MIPHY_CRC24A : MIPHY_CRC16);
  return MIPHY_OK;
}

// ldpc_segmenter_impl.cpp:57-67
uint32_t rm_length(const seg_t& s, uint32_t i_seg, uint32_t mod, uint32_t nof_layers, uint32_t nof_ch_symbols)
{
  const uint32_t sym_layer = nof_ch_symbols / nof_layers;
  const uint32_t nshort    = s.nof_cbs - (sym_layer % s.nof_cbs);
  const uint32_t t         = (i_seg < nshort) ? sym_layer / s.nof_cbs : (sym_layer + s.nof_cbs - 1) / s.nof_cbs;
  return t * nof_layers * mod;
}

struct tb_asm_desc { // per transport block, consumed by pusch_tb_part_kernel / pusch_tb_finish_kernel
  uint32_t first_desc;  // first codeblock descriptor / iters entry of this TB
  uint32_t nof_cbs;
  uint32_t harq_cb_index;
  uint32_t nof_data_bits; // data bits per codeblock (msg_length - crc - filler)
  uint32_t tb_and_crc_bits;
  uint32_t tb_bytes;
  uint32_t max_iter;
  uint32_t pad;
  uint64_t tb_offset;
};

// Transport-block assembly (pusch_decoder_impl.cpp:198-222): one workgroup per transport block, one WAVEFRONT per codeblock
// (round robin). When every codeblock passed its CRC, a wavefront copies the data bits of its codeblock to their place in the
// transport block and computes their part of the TB checksum by masks and popcounts (crc_zmask_packed24a: the remainder of the
// codeblock's bits as if the message ended with them), which one lane then weights with the bits that follow in TB + CRC24A.
// The XOR of the parts is zero exactly when CRC24A passes (all parts carry the same invertible factor x^24, see below).
// Then: decoder statistics, result record, CRC flags reset on a TB CRC failure.
// The mask / popcount form of the checksum parts applies to multi-codeblock transport blocks with byte-aligned payloads.
__host__ __device__ inline bool tb_mask_crc(const tb_asm_desc& d)
{
  return d.nof_cbs > 1 && d.nof_data_bits <= 32u * MIPHY_CRC_ZMASK_WORDS && (d.nof_data_bits & 7u) == 0;
}

__global__ void __launch_bounds__(1024) pusch_tb_assemble_kernel(const tb_asm_desc* __restrict__ descs,
                                                                const miphy_graph_tables* __restrict__ tab,
                                                                const int32_t* __restrict__ iters,
                                                                const uint8_t* __restrict__ harq_msgs,
                                                                uint8_t* __restrict__ harq_crc_ok,
                                                                uint8_t* __restrict__ tb_out,
                                                                miphy_pusch_result* __restrict__ results,
                                                                const uint32_t* __restrict__ cbw)
{
  __shared__ uint32_t cbpar[64]; // checksum part of every codeblock (at most 52), before its weight
  __shared__ int      all_ok;
  // The checksum masks of the word positions of one codeblock (rows counted from the END of the message, so a shorter last
  // codeblock uses the leading rows of the same copy): fetched once per transport block instead of 96 bytes per word and codeblock
  // out of L2 -- that stream was 1 GB per launch of 1024 transport blocks and half of this kernel's time.
  __shared__ __attribute__((aligned(16))) uint32_t zmask[MIPHY_CRC_ZMASK_WORDS * 24];
  const tb_asm_desc d    = descs[blockIdx.x];
  const int         tid  = threadIdx.x;
  const int         lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  if (tid < 64) {
    // one lane per codeblock (at most 52), reduced across the first wavefront
    int      ok = 1;
    uint32_t mn = 0xffffffffu, mx = 0, cnt = 0, sum = 0;
    if ((uint32_t)tid < d.nof_cbs) {
      ok           = harq_crc_ok[d.harq_cb_index + tid] != 0;
      const int it = iters[d.first_desc + tid];
      if (it >= 0) { // decoded in this call: stats.update(iterations or max) (pusch_decoder_impl.cpp:188-194)
        const uint32_t v = it > 0 ? (uint32_t)it : d.max_iter;
        mn = v, mx = v, sum = v, cnt = 1;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      ok &= __shfl_xor(ok, off);
      mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
      sum += (uint32_t)__shfl_xor((int)sum, off);
      cnt += (uint32_t)__shfl_xor((int)cnt, off);
    }
    if (tid == 0) {
      all_ok = ok;
      miphy_pusch_result r;
      r.tb_crc_ok            = 0;
      r.nof_codeblocks_total = d.nof_cbs;
      r.iters_min            = cnt ? mn : 0;
      r.iters_max            = mx;
      r.iters_mean           = cnt ? (float)sum / (float)cnt : 0.f;
      r.nof_decoded          = cnt;
      results[blockIdx.x]    = r;
    }
  }
  if (tid < 64)
    cbpar[tid] = 0;
  __syncthreads();
  if (!all_ok)
    return; // nothing is copied, flags stay as they are (multiple codeblocks) / tb_crc_ok = false (single codeblock)
  const uint32_t tb_bits = d.tb_bytes * 8;
  const bool     mask_crc = tb_mask_crc(d);
  if (mask_crc) {
    const uint32_t nrows = (d.nof_data_bits + 31) >> 5;
    const uint4*   src   = reinterpret_cast<const uint4*>(tab->crc_zmask_packed24a[0]);
    uint4*         dst   = reinterpret_cast<uint4*>(zmask);
    for (uint32_t q = tid; q < nrows * 6; q += blockDim.x)
      dst[q] = src[q];
    __syncthreads();
  }
  for (uint32_t c = wave; c < d.nof_cbs; c += nwaves) {
    const uint8_t* msg   = harq_msgs + (size_t)(d.harq_cb_index + c) * HARQ_MSG_STRIDE;
    const uint32_t bit0  = c * d.nof_data_bits; // first TB(+CRC) bit of this codeblock
    const uint32_t nbits = min(d.nof_data_bits, d.tb_and_crc_bits - min(bit0, d.tb_and_crc_bits));
    // ---- copy: whole bytes (codeblock payloads are byte aligned for every TS 38.214 transport block size; a single codeblock
    // starts at bit 0, only its last byte can be partial and is masked)
    const uint32_t o0 = bit0 >> 3;
    const uint32_t e0 = (min(bit0 + nbits, tb_bits) + 7) >> 3; // end byte in the transport block
    // (tb_bits is a whole number of bytes, so no byte is partial.) The destination has any alignment, the source slot is dword
    // aligned: head bytes up to the destination's dword boundary, then aligned dword stores fed by a funnel shift of two source
    // dwords (the slot is HARQ_MSG_STRIDE = 1056 bytes, the dword behind the last payload byte is inside it), then tail bytes.
    // All source words of the codeblock are requested before the first store (a slot is 264 dwords: at most five per lane), and the
    // checksum below works on the same registers.
    constexpr int   KW   = (HARQ_MSG_STRIDE / 4 + 63) / 64;
    const uint32_t* s32  = reinterpret_cast<const uint32_t*>(msg);
    const uint32_t  nw   = (nbits + 31) >> 5; // words the checksum covers
    uint32_t        lo[KW], hi[KW];
    uint8_t*        dst  = tb_out + d.tb_offset + o0;
    const uint32_t  nb   = e0 > o0 ? e0 - o0 : 0;
    const uint32_t  head = min(nb, (uint32_t)((4u - (uint32_t)((uintptr_t)dst & 3u)) & 3u));
    const uint32_t  ndw  = (nb - head) >> 2;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      const uint32_t i = (uint32_t)lane + 64u * k;
      lo[k]            = (i < ndw || i < nw) ? s32[i] : 0u;
      hi[k]            = (i < ndw && head) ? s32[i + 1] : 0u;
    }
    {
      uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
      if ((uint32_t)lane < head)
        dst[lane] = msg[lane];
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        const uint32_t i = (uint32_t)lane + 64u * k;
        if (i < ndw)
          d32[i] = head ? __builtin_amdgcn_alignbyte(hi[k], lo[k], head) : lo[k];
      }
      for (uint32_t b = head + 4 * ndw + lane; b < nb; b += 64)
        dst[b] = msg[b];
    }
    // ---- checksum part
    if (mask_crc) {
      // (the mask table assumes the message padded with zeros to whole words: folded into cbw)
      uint32_t acc[24];
#pragma unroll
      for (int k = 0; k < 24; ++k)
        acc[k] = 0;
#pragma unroll
      for (int k = 0; k < KW; ++k) {
        const uint32_t t = (uint32_t)lane + 64u * k;
        if (t < nw) {
          uint32_t w = lo[k];
          if (32 * t + 32 > nbits) { // last word: keep the first nbits - 32 t message bits = the leading bytes (byte aligned)
            const uint32_t nbl = (nbits - 32 * t) >> 3;
            w &= (nbl >= 4) ? 0xffffffffu : ((1u << (8 * nbl)) - 1u);
          }
          const uint4* m = reinterpret_cast<const uint4*>(zmask + (nw - 1 - t) * 24);
#pragma unroll
          for (int g = 0; g < 6; ++g) {
            const uint4 mk = m[g];
            acc[4 * g + 0] += __builtin_popcount(w & mk.x);
            acc[4 * g + 1] += __builtin_popcount(w & mk.y);
            acc[4 * g + 2] += __builtin_popcount(w & mk.z);
            acc[4 * g + 3] += __builtin_popcount(w & mk.w);
          }
        }
      }
      uint32_t par = 0;
#pragma unroll
      for (int k = 0; k < 24; ++k)
        par |= (acc[k] & 1u) << k;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
        par ^= __shfl_xor(par, off);
      // par = D(x) x^(r + 24) mod P; its weight x^(after + 24 - r) (cbw, from the host) brings it to the common factor x^24 below.
      if (lane == 0)
        cbpar[c] = par;
    } else if (d.nof_cbs > 1) {
      uint32_t par = crc_partial(tab, MIPHY_CRC24A, msg, 0, nbits, lane, 64);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
        par ^= __shfl_xor(par, off);
      if (lane == 0)
        cbpar[c] = par; // weight x^(after + 24): the same common factor as the mask form
    }
  }
  __syncthreads();
  if (tid < 64) { // one lane per codeblock: part x weight, XOR over the codeblocks (all parts carry the invertible factor x^24)
    uint32_t v = 0;
    if ((uint32_t)tid < d.nof_cbs && d.nof_cbs > 1)
      v = crc_gf2_mulmod(cbpar[tid], cbw[d.first_desc + tid], tab->crc_poly[MIPHY_CRC24A], 24);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
      v ^= __shfl_xor(v, off);
    if (tid == 0)
      cbpar[0] = v;
  }
  __syncthreads();
  if (tid == 0) {
    const uint32_t rem = cbpar[0];
    const int tb_ok               = (d.nof_cbs == 1) || (rem == 0);
    results[blockIdx.x].tb_crc_ok = tb_ok;
    all_ok                        = tb_ok;
  }
  __syncthreads();
  if (!all_ok) // pusch_decoder_impl.cpp:218-220: at least one codeblock is a false negative, reset all of them
    for (uint32_t c = tid; c < d.nof_cbs; c += blockDim.x)
      harq_crc_ok[d.harq_cb_index + c] = 0;
}

template <typename T>
T* stage_vec(uint8_t*& h, uint8_t*& d, const std::vector<T>& v, size_t& off)
{
  off              = (off + 15) & ~(size_t)15;
  T* dev           = reinterpret_cast<T*>(d + off);
  if (!v.empty())
    std::memcpy(h + off, v.data(), v.size() * sizeof(T));
  off += v.size() * sizeof(T);
  return dev;
}

} // namespace

extern "C" int miphy_sch_segmentation_info(uint32_t tb_bytes, uint32_t bg, miphy_sch_segmentation* out)
{
  if (!out) {
    miphy_set_error("miphy_sch_segmentation_info: null argument");
    return MIPHY_EINVAL;
  }
  seg_t s;
  int   rc = segmentation(tb_bytes, bg, s);
  if (rc)
    return rc;
  out->nof_cbs = s.nof_cbs, out->Z = s.Z, out->K = s.K, out->N = s.N, out->nof_filler_bits = s.nof_filler_bits;
  out->nof_tb_crc_bits = s.nof_tb_crc_bits, out->nof_cb_crc_bits = s.nof_cb_crc_bits, out->cb_info_bits = s.cb_info_bits;
  out->zero_pad = s.zero_pad;
  return MIPHY_OK;
}

namespace {

// Host-side product of the segmentation of a batch of transport blocks: everything the three launches of a PUSCH decode need.
struct pusch_decode_build {
  std::vector<miphy_ldpc_rdm_desc> rdm;
  std::vector<miphy_ldpc_dec_desc> dec;
  std::vector<uint32_t>            slots, reset_slots;
  std::vector<tb_asm_desc>         asmd;
  std::vector<uint32_t>            cb_tb; // transport block of each codeblock
  std::vector<uint32_t>            cbw;   // per codeblock: x^sh mod CRC24A, the weight of its checksum part in the transport block's (tb assembly)
  uint32_t                         max_Z = 2, max_nodes = 0, max_E = 0;
  int                              bg_mask   = 0;     // bit 0 / 1: base graph 1 / 2 occurs (the decoder sizes its LDS per base graph)
  std::vector<uint8_t>             fusable;           // per codeblock: it can be rate-dematched by the decoder while it loads
  // Launches: chunks of at most 65535 codeblocks (whole transport blocks), each sorted into the decoder's launch classes
  // (miphy_ldpc_build_classes: by lifting size, base graph, reachable layers, dematch-while-loading or not).
  struct chunk {
    uint32_t           t0, t1, c0, c1;
    miphy_ldpc_classes cls;
    uint32_t           order_off, bundle_off, rdm_nf_off; // positions in the concatenated `order` / `bundles` / `rdm_nf` arrays
  };
  std::vector<chunk>               chunks;
  std::vector<uint32_t>            order, bundles;    // all chunks, indices relative to the chunk's first codeblock
  std::vector<miphy_ldpc_rdm_desc> rdm_nf;            // dematcher descriptors of the codeblocks of classes that are not fused, chunk after chunk
  bool                             all_fused = true;  // every class of every chunk dematches in the decoder
};

// Device-side view of a staged build (pointers into one buffer).
struct pusch_decode_dev {
  const miphy_ldpc_rdm_desc* rdm;
  const miphy_ldpc_rdm_desc* rdm_nf;
  const uint32_t*            order;
  const uint32_t*            bundles;
  const miphy_ldpc_dec_desc* dec;
  const uint32_t*            slots;
  const uint32_t*            reset;
  const tb_asm_desc*         asmd;
  const uint32_t*            cb_tb;
  const uint32_t*            cbw;
  int32_t*                   iters;
  uint32_t*                  part; // per codeblock: its part of the TB checksum
  size_t                     staged; // bytes to copy host -> device
  size_t                     total;  // bytes of the whole buffer
};

// x^e mod the CRC24A polynomial (host, square and multiply).
static uint32_t crc24a_mulmod(uint32_t a, uint32_t b)
{
  uint32_t r = 0;
  for (int k = 23; k >= 0; --k) {
    r <<= 1;
    r ^= (r & (1u << 24)) ? 0x1864CFBu : 0u;
    r ^= ((b >> k) & 1u) ? a : 0u;
  }
  return r;
}
static uint32_t crc24a_x_pow(uint64_t e)
{
  uint32_t r = 1, base = 2;
  for (; e; e >>= 1) {
    if (e & 1u)
      r = crc24a_mulmod(r, base);
    base = crc24a_mulmod(base, base);
  }
  return r;
}

int build_pusch_decode(const miphy_pusch_tb_desc* tbs, uint32_t n, pusch_decode_build& b)
{
  b.asmd.resize(n);
  for (uint32_t t = 0; t < n; ++t) {
    const miphy_pusch_tb_desc& d = tbs[t];
    seg_t                      sg;
    int                        rc = segmentation(d.tb_bytes, d.bg, sg);
    if (rc)
      return rc;
    MIPHY_REQUIRE(d.rv <= 3, "pusch_decode: TB %u: invalid redundancy version", t);
    MIPHY_REQUIRE(d.nof_layers >= 1 && d.nof_layers <= 4, "pusch_decode: TB %u: invalid number of layers", t);
    MIPHY_REQUIRE(d.mod == 1 || d.mod == 2 || d.mod == 4 || d.mod == 6 || d.mod == 8, "pusch_decode: TB %u: invalid modulation", t);
    MIPHY_REQUIRE(d.nof_ch_symbols % d.nof_layers == 0, "pusch_decode: TB %u: channel symbols not a multiple of the layers", t);
    MIPHY_REQUIRE(d.nof_ldpc_iterations > 0, "pusch_decode: TB %u: nof_ldpc_iterations must be > 0", t);
    MIPHY_REQUIRE((sg.cb_info_bits % 8) == 0 || sg.nof_cbs == 1,
                  "pusch_decode: TB %u: TBS %u is not a TS 38.214 transport block size (codeblock payloads are not byte aligned)", t, sg.tbs);
    const uint32_t bgK = (d.bg == 1) ? 22 : 10;
    tb_asm_desc&   a   = b.asmd[t];
    a.first_desc       = (uint32_t)b.dec.size();
    a.nof_cbs          = sg.nof_cbs;
    a.harq_cb_index    = d.harq_cb_index;
    a.nof_data_bits    = sg.K - ((sg.nof_cbs == 1) ? sg.nof_tb_crc_bits : 24) - sg.nof_filler_bits; // get_cblk_bit_breakdown
    a.tb_and_crc_bits  = sg.tbs + ((sg.nof_cbs > 1) ? 24 : 0);
    a.tb_bytes         = d.tb_bytes;
    a.max_iter         = d.nof_ldpc_iterations;
    a.tb_offset        = d.tb_offset;
    uint32_t cw_off = 0, tb_nodes = 0;
    const bool mask_crc = tb_mask_crc(a);
    for (uint32_t c = 0; c < sg.nof_cbs; ++c) {
      { // weight of this codeblock's checksum part (pusch_tb_assemble_kernel): every part is brought to the common factor x^24
        const uint32_t bit0  = c * a.nof_data_bits;
        const uint32_t nbits = std::min(a.nof_data_bits, a.tb_and_crc_bits - std::min(bit0, a.tb_and_crc_bits));
        const uint32_t after = a.tb_and_crc_bits - (bit0 + nbits);
        const uint32_t r     = mask_crc ? 32 * ((nbits + 31) >> 5) - nbits : 0; // zero bits the mask table assumes behind the message
        b.cbw.push_back(crc24a_x_pow((uint64_t)after + 24 - r));
      }
      const uint32_t      E    = rm_length(sg, c, d.mod, d.nof_layers, d.nof_ch_symbols);
      const uint32_t      slot = d.harq_cb_index + c;
      miphy_ldpc_rdm_desc r    = {};
      r.bg = d.bg, r.rv = d.rv, r.mod = d.mod, r.new_data = d.new_data ? 1 : 0, r.Z = (uint16_t)sg.Z;
      r.nof_filler_bits = (uint16_t)sg.nof_filler_bits, r.Nref = d.Nref, r.E = E;
      r.in_offset = d.llr_offset + cw_off, r.out_offset = (uint64_t)slot * HARQ_CB_STRIDE;
      MIPHY_REQUIRE(E > 0, "pusch_decode: TB %u: empty codeblock", t);
      b.max_E = E > b.max_E ? E : b.max_E;
      b.rdm.push_back(r);
      miphy_ldpc_dec_desc q = {};
      q.bg = d.bg, q.crc_poly = (uint8_t)sg.crc_poly, q.Z = (uint16_t)sg.Z, q.max_iter = d.nof_ldpc_iterations;
      q.nof_filler_bits = (uint16_t)sg.nof_filler_bits;
      // The reference hands the full-length soft buffer to the decoder (pusch_decoder_impl.cpp:177,186), which trims it at the
      // last non-zero soft bit. For a first transmission with rv 0 everything behind the E rate-matched bits (+ fillers) is zero
      // by construction, so the bound is known here: the decoder then sizes its LDS for the layers that can be reached.
      q.in_len = sg.N;
      {
        // (not with a limited buffer: there the dematcher leaves part of the tail untouched, and stale soft bits of an earlier
        // use of the HARQ slot are visible to the reference decoder)
        const bool full_buffer = !(d.Nref > 0 && d.Nref < sg.N);
        if (d.new_data && d.rv == 0 && full_buffer && E + sg.nof_filler_bits <= sg.N) {
          const uint32_t need = ((E + sg.nof_filler_bits + sg.Z - 1) / sg.Z) * sg.Z;
          const uint32_t lo   = (bgK + 2) * sg.Z;
          q.in_len            = need < lo ? lo : (need > sg.N ? sg.N : need);
        }
      }
      {
        const uint32_t nodes = (q.in_len + 2 * sg.Z + sg.Z - 1) / sg.Z; // variable nodes this codeblock can reach
        tb_nodes             = nodes > tb_nodes ? nodes : tb_nodes;
      }
      // Dematching while the decoder loads: first transmission at redundancy version 0 into the full circular buffer, the E bits
      // neither wrap around it nor stop inside the systematic part, soft-buffer slots 16-byte aligned (HARQ_CB_STRIDE is).
      b.fusable.push_back(d.new_data && d.rv == 0 && !(d.Nref > 0 && d.Nref < sg.N) && E + sg.nof_filler_bits <= sg.N &&
                          E >= (bgK - 2) * sg.Z - sg.nof_filler_bits && (sg.Z % 16) == 0);
      b.bg_mask |= 1 << (d.bg - 1);
      q.flags           = d.use_early_stop ? 0u : 1u;
      q.llr_offset = (uint64_t)slot * HARQ_CB_STRIDE, q.out_offset = (uint64_t)slot * HARQ_MSG_STRIDE;
      b.dec.push_back(q);
      b.cb_tb.push_back(t);
      b.slots.push_back(slot);
      if (d.new_data)
        b.reset_slots.push_back(slot);
      cw_off += E;
    }
    MIPHY_REQUIRE(cw_off == d.nof_ch_symbols * d.mod, "pusch_decode: TB %u: codeblock lengths (%u) do not add up to the codeword (%u)", t, cw_off,
                  d.nof_ch_symbols * d.mod);
    b.max_Z     = sg.Z > b.max_Z ? sg.Z : b.max_Z;
    b.max_nodes = tb_nodes > b.max_nodes ? tb_nodes : b.max_nodes;
  }
  // chunks of whole transport blocks, each with its launch classes
  for (uint32_t t0 = 0; t0 < n;) {
    pusch_decode_build::chunk ch;
    ch.t0 = t0, ch.t1 = t0, ch.c0 = b.asmd[t0].first_desc, ch.c1 = ch.c0;
    while (ch.t1 < n && ch.c1 + b.asmd[ch.t1].nof_cbs - ch.c0 <= 65535) {
      ch.c1 += b.asmd[ch.t1].nof_cbs;
      ++ch.t1;
    }
    MIPHY_REQUIRE(ch.t1 > ch.t0, "pusch_decode: TB %u has more than 65535 codeblocks", t0);
    miphy_ldpc_build_classes(b.dec.data() + ch.c0, ch.c1 - ch.c0, b.fusable.data() + ch.c0, ch.cls);
    ch.order_off = (uint32_t)b.order.size(), ch.bundle_off = (uint32_t)b.bundles.size(), ch.rdm_nf_off = (uint32_t)b.rdm_nf.size();
    b.order.insert(b.order.end(), ch.cls.order.begin(), ch.cls.order.end());
    b.bundles.insert(b.bundles.end(), ch.cls.bundles.begin(), ch.cls.bundles.end());
    for (uint32_t k = 0; k < ch.cls.nof_unfused; ++k)
      b.rdm_nf.push_back(b.rdm[ch.c0 + ch.cls.order[k]]);
    b.all_fused &= ch.cls.nof_unfused == 0;
    std::vector<uint32_t>().swap(ch.cls.order); // kept in b.order
    std::vector<uint32_t>().swap(ch.cls.bundles);
    t0 = ch.t1;
    b.chunks.push_back(std::move(ch));
  }
  return MIPHY_OK;
}

size_t pusch_decode_bytes(const pusch_decode_build& b)
{
  return 64 + (b.rdm.size() + b.rdm_nf.size()) * sizeof(b.rdm[0]) + b.dec.size() * sizeof(b.dec[0]) + (b.slots.size() + b.reset_slots.size()) * 4 +
         b.asmd.size() * sizeof(b.asmd[0]) + (b.cb_tb.size() + b.cbw.size() + b.order.size() + b.bundles.size()) * 4 + b.dec.size() * 8 + 16 * 12;
}

// Lays the build out in a host image `h` of the device buffer `dv` (same offsets) and returns the device pointers.
pusch_decode_dev layout_pusch_decode(const pusch_decode_build& b, uint8_t* h, uint8_t* dv)
{
  pusch_decode_dev v;
  size_t           off = 0;
  v.rdm     = stage_vec(h, dv, b.rdm, off);
  v.rdm_nf  = stage_vec(h, dv, b.rdm_nf, off);
  v.order   = stage_vec(h, dv, b.order, off);
  v.bundles = stage_vec(h, dv, b.bundles, off);
  v.dec     = stage_vec(h, dv, b.dec, off);
  v.slots   = stage_vec(h, dv, b.slots, off);
  v.reset   = stage_vec(h, dv, b.reset_slots, off);
  v.asmd    = stage_vec(h, dv, b.asmd, off);
  v.cb_tb   = stage_vec(h, dv, b.cb_tb, off);
  v.cbw     = stage_vec(h, dv, b.cbw, off);
  v.staged  = off;
  off       = (off + 15) & ~(size_t)15;
  v.iters   = reinterpret_cast<int32_t*>(dv + off);
  off += b.dec.size() * 4;
  off   = (off + 15) & ~(size_t)15;
  v.part  = reinterpret_cast<uint32_t*>(dv + off);
  v.total = off + b.dec.size() * 4;
  return v;
}

// The launches of one PUSCH decode on staged descriptors: CRC-flag reset of new transmissions, rate dematching into the HARQ
// soft buffers, LDPC decoding (codeblocks already decoded are skipped), transport-block assembly + TB CRC + result records.
// Launches of at most 65535 codeblocks each; transport blocks are never split across launches.
int launch_pusch_decode(miphy_ctx* ctx, const pusch_decode_build& b, const pusch_decode_dev& v, uint32_t n, const int8_t* llrs, int8_t* harq_softbits,
                        uint8_t* harq_msgs, uint8_t* harq_crc_ok, uint8_t* tb_out, miphy_pusch_result* results, hipStream_t s,
                        hipEvent_t* ev = nullptr /* optional: 4 events around the dematch launches / the decoder launches / the assembly */)
{
  int rc;
  miphy_ldpc_rdm_limits rlim = {b.max_E};
  // Dematching inside the decoder needs 16-byte aligned soft buffers (its write-back is vectorised; HARQ_CB_STRIDE keeps the slots so)
  const bool allow_fuse = ((uintptr_t)harq_softbits & 15) == 0 && !miphy_ldpc_scalar_forced();
  if (ev)
    MIPHY_HIP_CHECK(hipEventRecord(ev[0], s));
  // the CRC flags of the new transmissions of the call (a decoder that dematches itself writes the flags of its codeblocks either way)
  if (!(b.all_fused && allow_fuse) && (rc = miphy_ldpc_flags_reset(v.reset, (uint32_t)b.reset_slots.size(), harq_crc_ok, s)))
    return rc;
  for (const pusch_decode_build::chunk& ch : b.chunks) { // rate dematching as launches of its own where the decoder does not do it
    if (!allow_fuse)
      rc = miphy_ldpc_rate_dematch_batch(ctx, v.rdm + ch.c0, 1, ch.c1 - ch.c0, llrs, harq_softbits, &rlim, s);
    else
      rc = ch.cls.nof_unfused ? miphy_ldpc_rate_dematch_batch(ctx, v.rdm_nf + ch.rdm_nf_off, 1, ch.cls.nof_unfused, llrs, harq_softbits, &rlim, s) : MIPHY_OK;
    if (rc)
      return rc;
  }
  if (ev)
    MIPHY_HIP_CHECK(hipEventRecord(ev[1], s));
  for (const pusch_decode_build::chunk& ch : b.chunks) {
    if ((rc = miphy_ldpc_decode_classes_launch(ctx, v.dec + ch.c0, ch.cls, v.order + ch.order_off, v.bundles + ch.bundle_off, harq_softbits, harq_msgs,
                                               v.iters + ch.c0, v.slots + ch.c0, harq_crc_ok, s, v.rdm + ch.c0, llrs, allow_fuse)))
      return rc;
  }
  if (ev)
    MIPHY_HIP_CHECK(hipEventRecord(ev[2], s));
  #ifndef TBA_THREADS
#define TBA_THREADS 512
#endif
  // One workgroup per transport block, one wavefront per codeblock in turns: eight wavefronts when the batch fills the chip (measured
  // best at 1024 transport blocks), sixteen when it does not (a single slot: three turns of its 38 codeblocks instead of five).
  const unsigned tba_threads = n * 4u <= (unsigned)ctx->num_cus ? 1024u : (unsigned)TBA_THREADS;
  hipLaunchKernelGGL(pusch_tb_assemble_kernel, dim3(n), dim3(tba_threads), 0, s, v.asmd, ctx->d_tables, v.iters, harq_msgs, harq_crc_ok, tb_out, results, v.cbw);
  if (ev)
    MIPHY_HIP_CHECK(hipEventRecord(ev[3], s));
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

} // namespace

extern "C" int miphy_pusch_decode_batch(miphy_ctx*                 ctx,
                                        const miphy_pusch_tb_desc* tbs,
                                        uint32_t                   n,
                                        const int8_t*              llrs,
                                        int8_t*                    harq_softbits,
                                        uint8_t*                   harq_msgs,
                                        uint8_t*                   harq_crc_ok,
                                        uint8_t*                   tb_out,
                                        miphy_pusch_result*        results,
                                        void*                      stream)
{
  MIPHY_REQUIRE(ctx && tbs && llrs && harq_softbits && harq_msgs && harq_crc_ok && tb_out && results, "miphy_pusch_decode_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  hipStream_t        s = (hipStream_t)stream;
  pusch_decode_build b;
  int                rc = build_pusch_decode(tbs, n, b);
  if (rc)
    return rc;
  const size_t         bytes = pusch_decode_bytes(b);
  std::vector<uint8_t> host(bytes);
  void*                wsv = nullptr;
  if ((rc = miphy_get_workspace(ctx, bytes, s, &wsv)))
    return rc;
  const pusch_decode_dev v = layout_pusch_decode(b, host.data(), (uint8_t*)wsv);
  if ((rc = miphy_upload(ctx, wsv, host.data(), v.staged, s))) // through the pinned ring: the stream is not waited for
    return rc;
  return launch_pusch_decode(ctx, b, v, n, llrs, harq_softbits, harq_msgs, harq_crc_ok, tb_out, results, s);
}

// ---- prepared form: the segmentation and the descriptor upload happen once, every run is launches only (no host
// synchronisation), for allocations that repeat slot after slot.
struct miphy_pusch_decode_plan {
  miphy_ctx*         ctx;
  uint32_t           n;
  pusch_decode_build b; // host side kept for the launch limits (the descriptor vectors are released after staging)
  pusch_decode_dev   v;
  void*              d_buf;
  uint32_t           ncb;
  std::vector<hipEvent_t> events; // timing: 4 per run, ring of events.size() / 4 runs
  uint32_t           timed_runs;
};

extern "C" int miphy_pusch_decode_plan_create(miphy_ctx* ctx, const miphy_pusch_tb_desc* tbs, uint32_t n, miphy_pusch_decode_plan** out)
{
  MIPHY_REQUIRE(ctx && tbs && out && n > 0, "miphy_pusch_decode_plan_create: null argument or empty batch");
  auto* p = new miphy_pusch_decode_plan();
  p->ctx = ctx, p->n = n, p->d_buf = nullptr, p->timed_runs = 0;
  int rc = build_pusch_decode(tbs, n, p->b);
  p->ncb = (uint32_t)p->b.dec.size();
  if (rc) {
    delete p;
    return rc;
  }
  const size_t         bytes = pusch_decode_bytes(p->b);
  std::vector<uint8_t> host(bytes);
  hipError_t           e = hipMalloc(&p->d_buf, bytes);
  if (e != hipSuccess) {
    miphy_set_error("miphy_pusch_decode_plan_create: hipMalloc(%zu) -> %s", bytes, hipGetErrorString(e));
    delete p;
    return MIPHY_EHIP;
  }
  p->v = layout_pusch_decode(p->b, host.data(), (uint8_t*)p->d_buf);
  e    = hipMemcpy(p->d_buf, host.data(), p->v.staged, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    miphy_set_error("miphy_pusch_decode_plan_create: hipMemcpy -> %s", hipGetErrorString(e));
    (void)hipFree(p->d_buf);
    delete p;
    return MIPHY_EHIP;
  }
  // only sizes and limits are needed from here on
  std::vector<miphy_ldpc_rdm_desc>().swap(p->b.rdm);
  std::vector<miphy_ldpc_rdm_desc>().swap(p->b.rdm_nf);
  std::vector<miphy_ldpc_dec_desc>().swap(p->b.dec);
  std::vector<uint32_t>().swap(p->b.slots);
  std::vector<uint32_t>().swap(p->b.cb_tb);
  std::vector<uint32_t>().swap(p->b.order);
  std::vector<uint32_t>().swap(p->b.bundles);
  std::vector<uint8_t>().swap(p->b.fusable);
  *out = p;
  return MIPHY_OK;
}

extern "C" int miphy_pusch_decode_plan_run(miphy_pusch_decode_plan* p,
                                           const int8_t*            llrs,
                                           int8_t*                  harq_softbits,
                                           uint8_t*                 harq_msgs,
                                           uint8_t*                 harq_crc_ok,
                                           uint8_t*                 tb_out,
                                           miphy_pusch_result*      results,
                                           void*                    stream)
{
  MIPHY_REQUIRE(p && llrs && harq_softbits && harq_msgs && harq_crc_ok && tb_out && results, "miphy_pusch_decode_plan_run: null argument");
  hipEvent_t* ev = nullptr;
  if (!p->events.empty()) {
    const uint32_t cap = (uint32_t)p->events.size() / 4;
    ev                 = &p->events[(size_t)(p->timed_runs % cap) * 4];
    ++p->timed_runs;
  }
  return launch_pusch_decode(p->ctx, p->b, p->v, p->n, llrs, harq_softbits, harq_msgs, harq_crc_ok, tb_out, results, (hipStream_t)stream, ev);
}

extern "C" int miphy_pusch_decode_plan_enable_timing(miphy_pusch_decode_plan* p, uint32_t max_runs)
{
  MIPHY_REQUIRE(p && max_runs > 0 && max_runs <= 4096, "miphy_pusch_decode_plan_enable_timing: invalid argument");
  for (hipEvent_t e : p->events)
    (void)hipEventDestroy(e);
  p->events.assign((size_t)max_runs * 4, nullptr);
  for (hipEvent_t& e : p->events)
    MIPHY_HIP_CHECK(hipEventCreate(&e));
  p->timed_runs = 0;
  return MIPHY_OK;
}

extern "C" int miphy_pusch_decode_plan_read_timing(miphy_pusch_decode_plan* p, float ms_out[3], uint32_t* runs_out)
{
  MIPHY_REQUIRE(p && ms_out && runs_out, "miphy_pusch_decode_plan_read_timing: null argument");
  ms_out[0] = ms_out[1] = ms_out[2] = 0.f;
  const uint32_t cap  = (uint32_t)p->events.size() / 4;
  const uint32_t runs = p->timed_runs < cap ? p->timed_runs : cap;
  for (uint32_t r = 0; r < runs; ++r) {
    hipEvent_t* ev = &p->events[(size_t)r * 4];
    MIPHY_HIP_CHECK(hipEventSynchronize(ev[3]));
    for (int k = 0; k < 3; ++k) {
      float ms = 0.f;
      MIPHY_HIP_CHECK(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
      ms_out[k] += ms;
    }
  }
  for (int k = 0; k < 3 && runs; ++k)
    ms_out[k] /= (float)runs;
  *runs_out     = runs;
  p->timed_runs = 0;
  return MIPHY_OK;
}

extern "C" int miphy_pusch_decode_plan_info(const miphy_pusch_decode_plan* p, uint32_t info[3])
{
  MIPHY_REQUIRE(p && info, "miphy_pusch_decode_plan_info: null argument");
  info[0] = p->ncb;
  info[1] = p->b.all_fused ? 1u : 0u;
  info[2] = p->b.max_nodes;
  return MIPHY_OK;
}

extern "C" uint32_t miphy_pusch_decode_plan_nof_launches(const miphy_pusch_decode_plan* p)
{
  uint32_t n = 0;
  if (p)
    for (const auto& ch : p->b.chunks)
      n += (uint32_t)ch.cls.classes.size();
  return n;
}

extern "C" void miphy_pusch_decode_plan_destroy(miphy_pusch_decode_plan* p)
{
  if (!p)
    return;
  if (p->d_buf)
    (void)hipFree(p->d_buf);
  for (hipEvent_t e : p->events)
    (void)hipEventDestroy(e);
  delete p;
}

namespace {
// Host-side product of the segmentation of a batch of downlink transport blocks: what the four launches of a PDSCH encode need.
struct pdsch_encode_build {
  std::vector<miphy_crc_desc>      crcd;
  std::vector<miphy_pdsch_cb_desc> cbs; // one record per codeblock for the fused kernel (pdsch_cb_encode.hip)
  size_t                           max_lds = 0, max_lds_pk = 0; // of the one-lane-per-bit kernel's codeblocks / of the packed kernel's
  uint32_t                         npacked = 0;
  uint32_t                         n = 0, ncb = 0;
};
struct pdsch_encode_dev {
  const miphy_crc_desc*      crcd;
  const miphy_pdsch_cb_desc* cbs;
  uint32_t*                  tbcrc;
  size_t                     staged, total;
};

int build_pdsch_encode(const miphy_pdsch_tb_desc* tbs, uint32_t n, pdsch_encode_build& b)
{
  b.n = n;
  b.crcd.resize(n);
  for (uint32_t t = 0; t < n; ++t) {
    const miphy_pdsch_tb_desc& d = tbs[t];
    seg_t                      sg;
    int                        rc = segmentation(d.tb_bytes, d.bg, sg);
    if (rc)
      return rc;
    MIPHY_REQUIRE(d.rv <= 3, "pdsch_encode: TB %u: invalid redundancy version", t);
    MIPHY_REQUIRE(d.nof_layers >= 1 && d.nof_layers <= 4, "pdsch_encode: TB %u: invalid number of layers", t);
    MIPHY_REQUIRE(d.mod == 1 || d.mod == 2 || d.mod == 4 || d.mod == 6 || d.mod == 8, "pdsch_encode: TB %u: invalid modulation", t);
    MIPHY_REQUIRE(d.nof_ch_symbols % d.nof_layers == 0, "pdsch_encode: TB %u: channel symbols not a multiple of the layers", t);
    b.crcd[t].bit_offset = d.tb_offset * 8;
    b.crcd[t].nbits      = sg.tbs;
    b.crcd[t].poly       = (sg.nof_tb_crc_bits == 16) ? MIPHY_CRC16 : MIPHY_CRC24A;
    uint32_t tb_bit = 0, cw_off = 0;
    for (uint32_t c = 0; c < sg.nof_cbs; ++c) {
      const bool          last = (c == sg.nof_cbs - 1);
      miphy_pdsch_cb_desc p    = {};
      p.tb_offset = d.tb_offset, p.tb_bit_offset = tb_bit, p.tb_index = t;
      p.take_bits       = sg.cb_info_bits - (last ? sg.nof_tb_crc_bits + sg.zero_pad : 0);
      p.nof_tb_crc_bits = (uint16_t)(last ? sg.nof_tb_crc_bits : 0);
      p.zero_pad        = (uint16_t)(last ? sg.zero_pad : 0);
      p.nof_cb_crc_bits = (uint8_t)sg.nof_cb_crc_bits;
      p.K               = sg.K;
      tb_bit += p.take_bits;
      const uint32_t E = rm_length(sg, c, d.mod, d.nof_layers, d.nof_ch_symbols);
      MIPHY_REQUIRE(E > 0, "pdsch_encode: TB %u: empty codeblock", t);
      p.bg = d.bg, p.rv = d.rv, p.mod = d.mod, p.Z = (uint16_t)sg.Z, p.nof_filler_bits = (uint16_t)sg.nof_filler_bits, p.Nref = d.Nref, p.E = E;
      p.cw_offset = d.codeword_offset + cw_off;
      {
        // Only the part of the circular buffer the rate matcher will read is encoded: from k0 (TS 38.212 Table 5.4.2.1-2) over E
        // bits plus the fillers it may skip, or the whole buffer when that wraps. A high-rate codeblock needs 2 of the 42 extension
        // nodes of base graph 1.
        const uint32_t Ncb = (d.Nref > 0 && d.Nref < sg.N) ? d.Nref : sg.N;
        const uint32_t num = (d.bg == 1) ? ((d.rv == 0) ? 0u : (d.rv == 1) ? 17u : (d.rv == 2) ? 33u : 56u)
                                         : ((d.rv == 0) ? 0u : (d.rv == 1) ? 13u : (d.rv == 2) ? 25u : 43u);
        const uint32_t k0  = (uint32_t)(((uint64_t)num * Ncb) / sg.N) * sg.Z;
        const uint64_t end = (uint64_t)k0 + E + sg.nof_filler_bits;
        p.out_len          = (end >= Ncb) ? Ncb : (uint32_t)end;
      }
      if (const size_t lp = miphy_pdsch_cb_encode_pk_lds(p)) // the bit-packed kernel takes it
        b.max_lds_pk = std::max(b.max_lds_pk, lp), ++b.npacked;
      else
        b.max_lds = std::max(b.max_lds, miphy_pdsch_cb_encode_lds(sg.K, sg.Z, p.out_len));
      b.cbs.push_back(p);
      cw_off += E;
    }
    MIPHY_REQUIRE(cw_off == d.nof_ch_symbols * d.mod, "pdsch_encode: TB %u: codeblock lengths (%u) do not add up to the codeword (%u)", t, cw_off,
                  d.nof_ch_symbols * d.mod);
  }
  b.ncb = (uint32_t)b.cbs.size();
  MIPHY_REQUIRE(b.ncb <= 65535, "pdsch_encode: %u codeblocks in one call (max 65535)", b.ncb);
  return MIPHY_OK;
}

size_t pdsch_encode_bytes(const pdsch_encode_build& b)
{
  return 128 + b.crcd.size() * sizeof(b.crcd[0]) + b.cbs.size() * sizeof(b.cbs[0]) + (size_t)b.n * 4 + 16 * 8;
}

// Lays the descriptors out in a host image `h` of the device buffer `dv` (same offsets); the TB checksums follow the staged part.
pdsch_encode_dev layout_pdsch_encode(const pdsch_encode_build& b, uint8_t* h, uint8_t* dv)
{
  pdsch_encode_dev v;
  size_t           off = 0;
  v.crcd   = stage_vec(h, dv, b.crcd, off);
  v.cbs    = stage_vec(h, dv, b.cbs, off);
  v.staged = off;
  off      = (off + 15) & ~(size_t)15;
  v.tbcrc  = reinterpret_cast<uint32_t*>(dv + off);
  v.total  = off + (size_t)b.n * 4;
  return v;
}

// TB CRC of every transport block, then ONE kernel per codeblock: assembly (TB bits, TB CRC, padding, CRC24B, fillers), LDPC encoder and
// rate matcher with the codeblock in LDS (pdsch_cb_encode.hip).
unsigned g_pdsch_cb_counts[2] = {0, 0}; // codeblocks encoded by the bit-packed kernel / in total (miphy_debug_pdsch_cb_counts)

int launch_pdsch_encode(miphy_ctx* ctx, uint32_t n, uint32_t ncb, size_t max_lds, const pdsch_encode_dev& v, const uint8_t* tb_in, uint8_t* codeword_out,
                        hipStream_t s, uint32_t npacked, size_t max_lds_pk)
{
  int rc;
  g_pdsch_cb_counts[0] += npacked, g_pdsch_cb_counts[1] += ncb;
  if ((rc = miphy_crc_batch(ctx, v.crcd, 1, n, tb_in, v.tbcrc, s)))
    return rc;
  return miphy_pdsch_cb_encode_launch(ctx, v.cbs, ncb, max_lds, tb_in, v.tbcrc, codeword_out, s, npacked, max_lds_pk);
}
} // namespace

// Test hook: how many codeblocks of the PDSCH encoder launches since the last reset went to the bit-packed kernel, and how many in total.
extern "C" void miphy_debug_pdsch_cb_counts(unsigned out[2], int reset)
{
  out[0] = g_pdsch_cb_counts[0], out[1] = g_pdsch_cb_counts[1];
  if (reset)
    g_pdsch_cb_counts[0] = g_pdsch_cb_counts[1] = 0;
}

extern "C" int miphy_pdsch_encode_batch(miphy_ctx* ctx, const miphy_pdsch_tb_desc* tbs, uint32_t n, const uint8_t* tb_in, uint8_t* codeword_out, void* stream)
{
  MIPHY_REQUIRE(ctx && tbs && tb_in && codeword_out, "miphy_pdsch_encode_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  hipStream_t        s = (hipStream_t)stream;
  pdsch_encode_build b;
  int                rc = build_pdsch_encode(tbs, n, b);
  if (rc)
    return rc;
  const size_t bytes = pdsch_encode_bytes(b);
  void*        wsv   = nullptr;
  if ((rc = miphy_get_workspace(ctx, bytes, s, &wsv)))
    return rc;
  std::vector<uint8_t>   host(bytes);
  const pdsch_encode_dev v = layout_pdsch_encode(b, host.data(), (uint8_t*)wsv);
  if ((rc = miphy_upload(ctx, wsv, host.data(), v.staged, s)))
    return rc;
  return launch_pdsch_encode(ctx, n, b.ncb, b.max_lds, v, tb_in, codeword_out, s, b.npacked, b.max_lds_pk);
}

// ---- prepared form (internal: the PDSCH processor plan of pdsch_proc.hip builds on it): segmentation and descriptor upload once, a run
// is launches only.
struct miphy_pdsch_encode_prepared {
  miphy_ctx*       ctx;
  uint32_t         n, ncb;
  size_t           max_lds, max_lds_pk;
  uint32_t         npacked;
  pdsch_encode_dev v;
  void*            d_buf;
};

int miphy_pdsch_encode_prepare(miphy_ctx* ctx, const miphy_pdsch_tb_desc* tbs, uint32_t n, miphy_pdsch_encode_prepared** out)
{
  MIPHY_REQUIRE(ctx && tbs && out && n > 0, "miphy_pdsch_encode_prepare: null argument or empty batch");
  pdsch_encode_build b;
  int                rc = build_pdsch_encode(tbs, n, b);
  if (rc)
    return rc;
  const size_t bytes = pdsch_encode_bytes(b);
  auto*        p     = new miphy_pdsch_encode_prepared();
  p->ctx = ctx, p->n = n, p->ncb = b.ncb, p->max_lds = b.max_lds, p->max_lds_pk = b.max_lds_pk, p->npacked = b.npacked, p->d_buf = nullptr;
  std::vector<uint8_t> host(bytes);
  hipError_t           e = hipMalloc(&p->d_buf, bytes);
  if (e == hipSuccess) {
    p->v = layout_pdsch_encode(b, host.data(), (uint8_t*)p->d_buf);
    e    = hipMemcpy(p->d_buf, host.data(), p->v.staged, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    miphy_set_error("miphy_pdsch_encode_prepare: %s", hipGetErrorString(e));
    if (p->d_buf)
      (void)hipFree(p->d_buf);
    delete p;
    return MIPHY_EHIP;
  }
  *out = p;
  return MIPHY_OK;
}

int miphy_pdsch_encode_prepared_run(miphy_pdsch_encode_prepared* p, const uint8_t* tb_in, uint8_t* codeword_out, hipStream_t s)
{
  return launch_pdsch_encode(p->ctx, p->n, p->ncb, p->max_lds, p->v, tb_in, codeword_out, s, p->npacked, p->max_lds_pk);
}

void miphy_pdsch_encode_prepared_destroy(miphy_pdsch_encode_prepared* p)
{
  if (!p)
    return;
  if (p->d_buf)
    (void)hipFree(p->d_buf);
  delete p;
}
