// UL-SCH multiplexing of UCI on PUSCH (TS 38.212 6.2.7) in closed form, shared by the demultiplexer kernel (ulsch_demux.hip) and
// by the PUSCH demodulator (repetition placeholders of its descrambler, pusch_demod.hip).
// Behaviour contract: lib/phy/upper/channel_processors/ulsch_demultiplex_impl.cpp:74-291 -- a serial scan over the OFDM symbols and,
// inside a symbol, over the subcarriers, that hands every resource element to one of {SCH data, HARQ-ACK, CSI part 1, CSI part 2}.
// Here the scan over the symbols stays serial (host, 14 steps: ulsch_plan_symbols) and yields, per symbol, the stride and count
// of every field and where the symbol starts in every stream; the class of a subcarrier and its rank inside its class then follow
// from those numbers alone, so every resource element is classified independently on the device.
#pragma once
#include "miphy_internal.h"

// Per OFDM symbol of the allocation (index = symbol - start_symbol). Strides / counts are in resource elements.
struct ulsch_symbol_plan {
  uint16_t nof_re;   // resource elements of the symbol that carry UL-SCH or UCI (DM-RS symbols: the REs left of the DM-RS)
  uint16_t rvd_d, rvd_cnt;
  uint16_t ack_d, ack_cnt;
  uint16_t csi1_d, csi1_cnt;
  uint16_t csi2_d, csi2_cnt;
  uint16_t flags;    // bit 0: HARQ-ACK goes on RESERVED elements (G_ack_rvd != 0): it then punctures SCH / CSI part 2
  uint32_t off_in, off_sch, off_ack, off_csi1, off_csi2; // resource elements before this symbol in every stream
};

struct ulsch_plan {
  ulsch_symbol_plan sym[14];
  uint32_t          nof_symbols;
  uint32_t          bits_per_re;   // modulation order x layers
  uint32_t          nof_in_re, nof_sch_re, nof_ack_re, nof_csi1_re, nof_csi2_re;
  uint32_t          one_bit_fields; // bit 0 / 1 / 2: HARQ-ACK / CSI part 1 / CSI part 2 carries exactly one information bit (placeholders)
};

enum { ULSCH_SCH = 0, ULSCH_ACK = 1, ULSCH_CSI1 = 2, ULSCH_CSI2 = 3 };

struct ulsch_re_class {
  int      cls;      // stream of the element read from the input: ULSCH_*
  uint32_t rank;     // its position inside that stream, counted from the start of the symbol
  bool     punctured; // HARQ-ACK on a reserved element: SCH (or CSI part 2) additionally gets an all-zero element ...
  int      zero_cls;  // ... in this stream ...
  uint32_t zero_rank; // ... at this position
};

__host__ __device__ inline uint32_t ulsch_ceil_div(uint32_t a, uint32_t b)
{
  return (a + b - 1) / b;
}

// Elements of an arithmetic selection (every d-th candidate, cnt of them) among the first x candidates.
__host__ __device__ inline uint32_t ulsch_hits_below(uint32_t x, uint32_t d, uint32_t cnt)
{
  if (cnt == 0)
    return 0;
  const uint32_t h = ulsch_ceil_div(x, d);
  return h < cnt ? h : cnt;
}
__host__ __device__ inline bool ulsch_is_hit(uint32_t j, uint32_t d, uint32_t cnt)
{
  return cnt != 0 && (j % d) == 0 && (j / d) < cnt;
}

// Class of subcarrier i (0 <= i < nof_re) of a symbol. The order of the tests is the order of the reference's loop body
// (ulsch_demultiplex_impl.cpp:203-263); the candidate counters of that loop (i_ack, i_csi1, i_csi2) are the ranks below.
__host__ __device__ inline ulsch_re_class ulsch_classify(const ulsch_symbol_plan& p, uint32_t i)
{
  ulsch_re_class r;
  r.punctured = false, r.zero_cls = ULSCH_SCH, r.zero_rank = 0;
  const bool     on_rvd   = (p.flags & 1u) != 0;
  const bool     reserved = ulsch_is_hit(i, p.rvd_d ? p.rvd_d : 1, p.rvd_cnt);
  const uint32_t rvd_below = ulsch_hits_below(i, p.rvd_d ? p.rvd_d : 1, p.rvd_cnt);
  bool           ack = false;
  uint32_t       ack_rank = 0, acknr_below = 0; // acknr: HARQ-ACK elements that take the element for themselves (no reserved region)
  if (on_rvd) {
    if (reserved) {
      const uint32_t k = i / p.rvd_d;
      ack              = ulsch_is_hit(k, p.ack_d ? p.ack_d : 1, p.ack_cnt);
      ack_rank         = p.ack_d ? k / p.ack_d : 0;
    }
  } else {
    ack         = ulsch_is_hit(i, p.ack_d ? p.ack_d : 1, p.ack_cnt);
    ack_rank    = p.ack_d ? i / p.ack_d : 0;
    acknr_below = ulsch_hits_below(i, p.ack_d ? p.ack_d : 1, p.ack_cnt);
    if (ack) {
      r.cls = ULSCH_ACK, r.rank = ack_rank;
      return r;
    }
  }
  // CSI part 1 never uses reserved elements: its candidates are the elements that are neither reserved nor taken above.
  const uint32_t j1         = i - acknr_below - rvd_below;
  const uint32_t csi1_below = ulsch_hits_below(j1, p.csi1_d ? p.csi1_d : 1, p.csi1_cnt);
  const bool     csi1       = !reserved && ulsch_is_hit(j1, p.csi1_d ? p.csi1_d : 1, p.csi1_cnt);
  // CSI part 2: everything that is left (reserved elements included).
  const uint32_t j2         = i - acknr_below - csi1_below;
  const uint32_t csi2_below = ulsch_hits_below(j2, p.csi2_d ? p.csi2_d : 1, p.csi2_cnt);
  const bool     csi2       = !csi1 && ulsch_is_hit(j2, p.csi2_d ? p.csi2_d : 1, p.csi2_cnt);
  const uint32_t j3         = i - acknr_below - csi1_below - csi2_below; // SCH rank
  int      cls;
  uint32_t rank;
  if (csi1)
    cls = ULSCH_CSI1, rank = csi1_below;
  else if (csi2)
    cls = ULSCH_CSI2, rank = csi2_below;
  else
    cls = ULSCH_SCH, rank = j3;
  if (ack) { // reserved element carrying HARQ-ACK: the input goes to HARQ-ACK, the punctured stream gets zeros
    r.cls = ULSCH_ACK, r.rank = ack_rank;
    r.punctured = true, r.zero_cls = cls, r.zero_rank = rank;
    return r;
  }
  r.cls = cls, r.rank = rank;
  return r;
}
