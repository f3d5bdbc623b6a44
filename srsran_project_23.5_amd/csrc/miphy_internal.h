// Internal definitions shared by the HIP translation units of libmiphy.so (gfx950 only).
#pragma once
#include "../../include/miphy.h"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define MIPHY_MAX_Z 384
#define MIPHY_NOF_Z 51
#define MIPHY_BG1_EDGES 316
#define MIPHY_BG2_EDGES 197
#define MIPHY_MAX_EDGES 316
#define MIPHY_CRC_ZMASK_WORDS 264 // 8448 bits: the largest codeblock
#define MIPHY_NOF_SIDE_STREAMS 3

// Per-(base graph, lifting size) edge table entry: low 16 bits = column*Z (LDS byte offset of the variable node),
// high 16 bits = cyclic shift (already reduced modulo Z).
struct miphy_graph_tables {
  uint32_t edge[2][MIPHY_NOF_Z][MIPHY_MAX_EDGES];
  uint32_t edge_sb[2][MIPHY_NOF_Z][2 * MIPHY_MAX_EDGES]; // the same, unpacked: {shift, column*Z} per edge (scalar operands as loaded)
  uint16_t row_start[2][48];
  uint16_t pair_start[2][48]; // prefix sum of ceil(degree / 2) over the layers (packed decoder message storage)
  uint16_t z_pos[MIPHY_MAX_Z + 1]; // position of Z in the list of lifting sizes, 0xffff if invalid
  uint8_t  i_ls[MIPHY_MAX_Z + 1];  // lifting-size set index
  // CRC: pow32[p][k] = x^(32k) mod poly_p for k in [0, 320); poly/order per id.
  uint32_t crc_pow32[5][320];
  uint32_t crc_pow2[5][24]; // x^(32 * 2^b) mod poly
  uint32_t crc_pow32_hi[5][256]; // x^(32 * 256 * k) mod poly: with crc_pow32[k & 255] covers messages up to 2 Mbit in one product
  uint32_t crc_poly[5];
  uint32_t crc_order[5];
  // Zero test of a codeblock CRC by masks (LDPC decoders): for the polynomials a codeblock can carry (index 0 = CRC24A, 1 = CRC24B,
  // 2 = CRC16) and word u counted from the END of the message padded with zeros to a multiple of 32 bits, crc_zmask[.][u][k] selects
  // the bits of that word whose weight x^(distance to the end + order) mod P has bit k set; bit k of the checksum of the padded
  // message is the parity of the sum over u of popcount(word & mask). (M(x) x^r mod P == 0 <=> M(x) mod P == 0, so the padding
  // does not change the verdict.) Word layout: bit (q + 8 b) of a word is message bit 4 q + b of its group of 32 (see hard_flags()).
  // ([k][u], coalesced across the lanes, was measured: 24 dword loads per lane instead of 6 x 16 bytes cost the decoder 1 %; the 25 KB
  // table is cache resident either way.)
  uint32_t crc_zmask[3][MIPHY_CRC_ZMASK_WORDS][24];
  // The same for CRC24A over PACKED message bytes read as little-endian dwords (transport-block assembly): bit 8 k + 7 - j of a
  // word is message bit 8 k + j of its group of 32.
  uint32_t crc_zmask_packed24a[MIPHY_CRC_ZMASK_WORDS][24];
};
// index into crc_zmask for a MIPHY_CRC_* id, -1 if the polynomial has no mask table
static inline __host__ __device__ int miphy_crc_zmask_index(int crc_id)
{
  return crc_id == 0 ? 0 : (crc_id == 1 ? 1 : (crc_id == 3 ? 2 : -1));
}

struct miphy_ctx_ext; // C++ side caches (twiddle tables, OFDM plans), see miphy_ext.h

struct miphy_ctx {
  int                  device;
  miphy_ctx_ext*       ext;
  miphy_graph_tables*  d_tables; // device copy
  miphy_graph_tables*  h_tables; // host copy
  void*                d_desc_staging; // descriptor staging: a RING of desc_staging_bytes on the device and the same in pinned host memory,
  size_t               desc_staging_bytes; // handed out front to back (miphy_stage_descs, miphy_upload); the device is synchronised only when
  void*                h_desc_staging; // pinned    // the ring wraps, so a call with host descriptors never waits for the stream
  size_t               staging_head;
  void*                ring_streams[4]; // streams that regions of the ring were staged for since its last wrap (the wrap waits for them)
  int                  nof_ring_streams; // 5 = more than four
  void*                d_work[5];      // scratch workspaces, grown on demand: [0] the transport-block level entry points, DFT, polar;
  size_t               work_bytes[5];  // [1] intermediate buffers of miphy_pusch_process_batch, [2] codewords of miphy_pdsch_process_batch,
                                       // [3] check-to-variable messages of the LDPC decoder when they do not stay in LDS,
                                       // [4] scrambling sequences of the PUSCH demodulator
  int                  num_cus; // compute units of the device (persistent-kernel grid sizing)
  uint32_t*            d_queue; // work-queue counters of the persistent kernels: a ring of MIPHY_NOF_QUEUE_COUNTERS words, one per launch
  uint32_t             queue_next;
  // Side streams of the class-sorted LDPC decoder launches (created on first use): the launch classes of one call are independent, and the
  // small ones are latency-bound chains that leave the chip idle -- they run next to the large ones, forked from / joined to the caller's
  // stream with events (a pattern a HIP graph capture records as is).
  void*                side_stream[MIPHY_NOF_SIDE_STREAMS];
  void*                ev_fork;
  void*                ev_join[MIPHY_NOF_SIDE_STREAMS];
};
#define MIPHY_NOF_QUEUE_COUNTERS 256

void miphy_set_error(const char* fmt, ...);

#define MIPHY_HIP_CHECK(expr)                                                                 \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      miphy_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));   \
      return MIPHY_EHIP;                                                                      \
    }                                                                                         \
  } while (0)

#define MIPHY_REQUIRE(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      miphy_set_error(__VA_ARGS__);       \
      return MIPHY_EINVAL;                \
    }                                     \
  } while (0)

#ifdef __HIPCC__
// A descriptor through dword loads: read field by field, its 8- and 16-bit members become vector loads with a wait each (gfx950 has no
// scalar sub-dword loads); as dwords the whole record arrives in scalar registers with one request and the fields are shifts.
template <class T>
__device__ __forceinline__ T load_words(const T* __restrict__ p)
{
  static_assert(sizeof(T) % 4 == 0, "dword multiple");
  uint32_t                     w[sizeof(T) / 4];
  const uint32_t* __restrict__ s = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i)
    w[i] = s[i];
  T r;
  __builtin_memcpy(&r, w, sizeof(T));
  return r;
}
#endif

// Ensures the descriptor array is on the device; returns the device pointer through *out. Host descriptors are copied into the
// context's staging ring (pinned host -> device, asynchronous on `s`): the call does not wait for the stream, and the caller's array
// can be reused as soon as it returns.
int miphy_stage_descs(miphy_ctx* ctx, const void* descs, int on_device, size_t bytes, hipStream_t s, const void** out);
// Asynchronous upload of a host buffer of any kind (pageable, a local vector) to device memory `dst`, ordered on `s`: the bytes travel
// through the pinned ring, so `src` can be released when the call returns and the stream is not waited for. (Buffers larger than half
// the ring are copied directly and the stream is synchronised, as every upload was before.)
int miphy_upload(miphy_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s);

// Decoder launch, ONE kernel launch for the whole batch: device-resident descriptors (which the host cannot sort) and the forced kernels
// of miphy_debug_force_ldpc_kernel. Host descriptors are sorted into launch classes instead (below) unless a kernel is forced.
// fuse_rdm / fuse_in / fuse_rlim (device descriptors with the same index as descs): every codeblock is a first transmission that
// can be rate-dematched while the decoder loads it (conditions in ldpc_decode_pk.hip; sch.hip checks them); `llr` is then the HARQ
// soft-buffer array the dematched codeblocks are written to. When the packed kernel is not the one selected, the rate dematcher runs
// as its own launch first -- the result is the same either way.
int miphy_ldpc_decode_launch(miphy_ctx* ctx, const miphy_ldpc_dec_desc* descs, int descs_on_device, uint32_t n, const int8_t* llr,
                             uint8_t* out_bits, int32_t* iters, const miphy_ldpc_dec_limits* limits, const uint32_t* harq_slot,
                             uint8_t* harq_crc_ok, void* stream, const miphy_ldpc_rdm_desc* fuse_rdm = nullptr,
                             const int8_t* fuse_in = nullptr, const miphy_ldpc_rdm_limits* fuse_rlim = nullptr,
                             int bg_mask = 3 /* device descriptors: bit 0 / 1 = base graph 1 / 2 occurs */,
                             const uint32_t* reset_slots = nullptr, uint32_t nof_reset_slots = 0 /* CRC flags to clear before decoding (device
                             array): new transmissions. Skipped when the decoder dematches itself -- it then writes every flag either way */);

#ifdef __cplusplus
#include <vector>
// Launch classes of a batch whose descriptors the host can see (ldpc_decode.hip): codeblocks sorted so that each class shares a
// workgroup size and an LDS size.
struct miphy_ldpc_class {
  uint8_t  kind;  // 0 = wave kernel (Z <= 64, several codeblocks per wavefront); 1..3 = packed kernel with that many wavefronts per codeblock
  uint8_t  bgi;   // base graph - 1
  uint8_t  fused; // every codeblock can be rate-dematched by the decoder while it loads
  uint8_t  lay;   // layers a codeblock of the class can reach at most
  uint16_t max_Z;
  uint32_t first, count;               // the class's range of `order`
  uint32_t bundle_first, bundle_count; // kind 0: its range of `bundles` (pairs of words)
  uint32_t soft_total;                 // kind 0: LDS bytes for the soft bits of a bundle
};
struct miphy_ldpc_classes {
  std::vector<uint32_t>         order;   // codeblock indices, class after class; classes that are not fused come first
  std::vector<uint32_t>         bundles; // wave kernel: {first position in order, count} per bundle
  std::vector<miphy_ldpc_class> classes;
  uint32_t                      nof_unfused = 0; // order[0 .. nof_unfused) need the rate dematcher as a launch of its own
  bool                          identity    = true;
};
// fusable: optional flag per codeblock ("the decoder may dematch it"); descriptors must be valid (validated by the caller).
void miphy_ldpc_build_classes(const miphy_ldpc_dec_desc* descs, uint32_t n, const uint8_t* fusable, miphy_ldpc_classes& C);
// One launch per class. d_order / d_bundles = device copies of C.order / C.bundles. d_rdm / rm_in: rate-dematcher descriptors (same
// index as d_descs) and rate-matched input for the fused classes; allow_fuse = false runs them unfused (the caller has dematched).
int miphy_ldpc_decode_classes_launch(miphy_ctx* ctx, const miphy_ldpc_dec_desc* d_descs, const miphy_ldpc_classes& C, const uint32_t* d_order,
                                     const uint32_t* d_bundles, const int8_t* llr, uint8_t* out_bits, int32_t* iters, const uint32_t* harq_slot,
                                     uint8_t* harq_crc_ok, hipStream_t s, const miphy_ldpc_rdm_desc* d_rdm, const int8_t* rm_in, bool allow_fuse);
bool miphy_ldpc_scalar_forced(); // miphy_debug_force_ldpc_kernel(1): nothing is dematched inside the decoder then
int miphy_ldpc_flags_reset(const uint32_t* d_slots, uint32_t n, uint8_t* harq_crc_ok, hipStream_t s);
// Wave kernel (ldpc_decode_pkw.hip).
int miphy_ldpc_pkw_launch(miphy_ctx* ctx, const miphy_ldpc_dec_desc* d_descs, const uint32_t* d_order, const uint32_t* d_bundles, uint32_t nof_bundles,
                          int bgi, int lay, size_t soft_total, const int8_t* llr, uint8_t* out_bits, int32_t* iters, const uint32_t* harq_slot,
                          uint8_t* harq_crc_ok, hipStream_t s, int* used_gmsg, void* gmsg_buf = nullptr,
                          bool throughput_form = false /* A-B knob: geometry of a launch that fills the chip, whatever its size */);
size_t miphy_ldpc_pkw_gmsg_bytes(const miphy_ctx* ctx, uint32_t nof_bundles, int bgi, int lay, size_t soft_total, bool throughput_form = false);
#endif

// Returns a device scratch buffer of at least `bytes` (reallocated, after a stream sync, when it has to grow).
int miphy_get_workspace(miphy_ctx* ctx, size_t bytes, hipStream_t s, void** out, int which = 0);

// Packed (two rows per lane) LDPC decoder kernel, ldpc_decode_pk.hip.
size_t miphy_ldpc_pk_lds_bytes(int bgK, int lay, size_t Zt, int pairs_all, int parts = 1); // pairs_all = 0: messages in global memory; parts = 2 / 4: + exchange slots of the latency form
int    miphy_ldpc_pk_waves_per_cu(bool fused, int parts = 1);
int    miphy_ldpc_pk_launch(miphy_ctx* ctx, const miphy_ldpc_dec_desc* d_descs, uint32_t n, int threads, size_t lds, const int8_t* llr,
                            uint8_t* out_bits, int32_t* iters, int nodes_all, const uint32_t* harq_slot, uint8_t* harq_crc_ok, hipStream_t s,
                            const miphy_ldpc_rdm_desc* d_rdm = nullptr, const int8_t* rm_in = nullptr, int gmsg_pairs = 0,
                            const uint32_t* d_order = nullptr /* the launch decodes codeblocks d_order[0 .. n) of the arrays */,
                            void* gmsg_buf = nullptr /* message scratch of miphy_ldpc_pk_gmsg_bytes() bytes; null: the context's workspace */,
                            int parts = 1 /* latency form: 2 or 4 times the wavefronts per codeblock (lds from miphy_ldpc_pk_lds_bytes(..., parts)) */,
                            int lds_pairs = 0 /* with gmsg_pairs > 0: message dwords per lane that stay in LDS in front of the global ones (a layer boundary of the base graph) */);
// Resident workgroups of such a launch and the bytes of global message scratch it needs (0 with the messages in LDS).
uint32_t miphy_ldpc_pk_grid(const miphy_ctx* ctx, uint32_t n, int threads, size_t lds, bool fused, int parts = 1);
size_t   miphy_ldpc_pk_gmsg_bytes(const miphy_ctx* ctx, uint32_t n, int threads, size_t lds, bool fused, int gmsg_pairs);
// The context's side streams and fork / join events, created on first use.
int miphy_side_streams(miphy_ctx* ctx);
// The next work-queue counter of the context (zero: every launch leaves its counter cleared).
int miphy_next_queue_counter(miphy_ctx* ctx, uint32_t** out);
