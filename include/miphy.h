/*
 * miphy.h -- C ABI of the MI355X-native 5G NR upper-PHY hot path (libmiphy.so).
 *
 * This is the drop-in boundary: plain C, POD arguments, caller-owned DEVICE pointers (HBM resident), an explicit
 * HIP stream passed as void*.  Every entry point is batched -- the per-call C++ adapters that implement the
 * reference's abstract classes (srsran_project_23.5_amd/adapters/) wrap these with a batch of one.
 *
 * Each function cites the reference interface it replaces (paths relative to the srsRAN_Project 23.5 tree).
 * Return value: 0 on success, negative MIPHY_E* on error (the reference aborts through srsran_assert on the same
 * precondition violations; C has no asserts, so the adapters turn a non-zero code into report_fatal_error).
 *
 * Data conventions equal the reference's:
 *   - LLR: int8, finite range [-120,120], +-127 = +-infinity (include/srsran/phy/upper/log_likelihood_ratio.h:46-240)
 *   - packed bits: MSB first in each byte (include/srsran/adt/bit_buffer.h:35-44)
 *   - unpacked bits: one bit per byte, filler bit = 254 (include/srsran/phy/upper/channel_coding/ldpc/ldpc.h:107)
 *   - cf_t: interleaved fp32 (re, im)
 *   - resource grid: [port][symbol][subcarrier], subcarrier fastest (lib/phy/support/resource_grid_impl.h:42-46)
 */
#ifndef MIPHY_H
#define MIPHY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPHY_VERSION 1

enum {
  MIPHY_OK        = 0,
  MIPHY_EINVAL    = -1, /* invalid argument (would be an srsran_assert in the reference) */
  MIPHY_EHIP      = -2, /* HIP runtime error, see miphy_last_error() */
  MIPHY_ENOMEM    = -3,
  MIPHY_EUNSUPP   = -4
};

/* CRC polynomials, same order as srsran::crc_generator_poly (include/srsran/phy/upper/channel_coding/crc_calculator.h:31-38). */
enum { MIPHY_CRC24A = 0, MIPHY_CRC24B = 1, MIPHY_CRC24C = 2, MIPHY_CRC16 = 3, MIPHY_CRC11 = 4, MIPHY_CRC_NONE = 255 };

typedef struct miphy_ctx miphy_ctx;

/* Creates a context on HIP device `device`: uploads the TS 38.212 graph tables, CRC tables and polar tables and
 * allocates the scratch workspace.  One context per host thread (the reference's processors are single-threaded
 * objects, one per worker: lib/phy/upper/uplink_processor_concurrent.h:41-54). */
int         miphy_create(int device, miphy_ctx** ctx);
void        miphy_destroy(miphy_ctx* ctx);
const char* miphy_last_error(void);
int         miphy_version(void);

/* ------------------------------------------------------------------------------------------------------------------
 * LDPC decoder  --  replaces srsran::ldpc_decoder::decode
 *   include/srsran/phy/upper/channel_coding/ldpc/ldpc_decoder.h:73-74
 *   lib/phy/upper/channel_coding/ldpc/ldpc_decoder_impl.cpp:60-146 with the AVX2 arithmetic of ldpc_decoder_avx2.cpp.
 * One descriptor per codeblock.  `iters[i]` receives the value of the optional<unsigned> the reference returns:
 * the iteration count (>=1) when crc_poly != NONE and the CRC matched, 0 for nullopt.
 * The output holds bg_K*Z bits per codeblock, packed.  All-zero input: output all ones when crc_poly == NONE,
 * untouched otherwise (ldpc_decoder_impl.cpp:88-94).
 * Input LLR domain: [-127,127]; `in_len` should be a multiple of Z for bit-exact parity. */
typedef struct {
  uint8_t  bg;              /* 1 or 2 */
  uint8_t  crc_poly;        /* MIPHY_CRC*; MIPHY_CRC_NONE = no early stop (crc == nullptr) */
  uint16_t Z;               /* lifting size */
  uint16_t max_iter;        /* algorithm_conf.max_iterations (default 6) */
  uint16_t nof_filler_bits; /* cb_specific.nof_filler_bits */
  uint32_t in_len;          /* number of input LLRs (first one belongs to variable node 2) */
  uint32_t reserved;
  uint64_t llr_offset;      /* element offset of this codeblock's LLRs inside `llr` */
  uint64_t out_offset;      /* byte offset of this codeblock's packed message inside `out_bits` */
} miphy_ldpc_dec_desc;

int miphy_ldpc_decode_batch(miphy_ctx*                 ctx,
                            const miphy_ldpc_dec_desc* descs, /* n descriptors; host memory unless descs_on_device */
                            int                        descs_on_device,
                            uint32_t                   n,
                            const int8_t*              llr,      /* device */
                            uint8_t*                   out_bits, /* device */
                            int32_t*                   iters,    /* device, n entries */
                            void*                      stream);

#ifdef __cplusplus
}
#endif
#endif /* MIPHY_H */
