/* TEST INFRASTRUCTURE ONLY -- see phy_oracle.h.
 *
 * Scalar C restatement of the reference algorithms (srsRAN_Project 23.5).  Every function cites the reference
 * file:line it follows.  Written from the reference's *behaviour*; the formulation (node-wise, byte arrays, no SIMD)
 * is deliberately different from the HIP kernels (row-wise, compressed check state) so that the two check each other.
 */
#include "phy_oracle.h"
#include "../srsran_project_23.5_amd/csrc/tables/nr_ldpc_tables.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LLR_MAX 120
#define LLR_INF 127
#define FILLER_BIT 254
#define MAX_Z 384

/* ------------------------------------------------------------------------------------------------ CRC
 * lib/phy/upper/channel_coding/crc_calculator_lut_impl.cpp:33-153: plain MSB-first polynomial division, zero initial
 * state, no reflection; `reversecrcbit` makes the result for a non-multiple-of-8 length equal to the CRC of exactly
 * the given bits, so a bit-serial division is an exact restatement.                                              */
static const uint32_t CRC_POLY[6]  = {0x1864CFB, 0x1800063, 0x1B2B117, 0x11021, 0xE21, 0x61};
static const unsigned CRC_ORDER[6] = {24, 24, 24, 16, 11, 6};

uint32_t orc_crc_bits(int poly, const uint8_t* bits, unsigned nbits)
{
  uint32_t p = CRC_POLY[poly], top = 1u << CRC_ORDER[poly], mask = top - 1, r = 0;
  for (unsigned i = 0; i < nbits; ++i) {
    r = (r << 1) ^ ((uint32_t)(bits[i] & 1) << CRC_ORDER[poly]);
    if (r & top)
      r ^= p;
  }
  /* the division above already includes the x^order shift (message bit injected at x^order). */
  return r & mask;
}

uint32_t orc_crc_packed(int poly, const uint8_t* bytes, unsigned nbits)
{
  uint32_t p = CRC_POLY[poly], top = 1u << CRC_ORDER[poly], mask = top - 1, r = 0;
  for (unsigned i = 0; i < nbits; ++i) {
    uint32_t b = (bytes[i >> 3] >> (7 - (i & 7))) & 1u;
    r          = (r << 1) ^ (b << CRC_ORDER[poly]);
    if (r & top)
      r ^= p;
  }
  return r & mask;
}

/* ------------------------------------------------------------------------------------------------ graph helpers */
typedef struct {
  int             bgK, bgM, N_full, N_short, E;
  const uint16_t* row_start;
  const uint8_t*  col;
  const uint16_t* shift; /* [E] for the lifting-size set */
  int             i_ls;
} graph_t;

static int lifting_set(int Z)
{
  static const int A[8] = {2, 3, 5, 7, 9, 11, 13, 15};
  if (Z < 2 || Z > 384)
    return -1;
  /* i_LS: Z = a * 2^j (TS 38.212 Table 5.3.2-1). Strip factors of two, keeping a in the table (2 itself is a=2). */
  for (int i = 7; i >= 0; --i) {
    int a = A[i];
    if (Z % a)
      continue;
    int q = Z / a;
    if ((q & (q - 1)) == 0)
      return i;
  }
  return -1;
}

static int get_graph(int bg, int Z, graph_t* g)
{
  int ils = lifting_set(Z);
  if (ils < 0)
    return -1;
  /* Z must be one of the 51 lifting sizes. */
  int ok = 0;
  for (int i = 0; i < 51; ++i)
    ok |= (NR_LDPC_LIFTING_SIZES[i] == Z);
  if (!ok)
    return -1;
  g->i_ls = ils;
  if (bg == 1) {
    g->bgK = 22, g->bgM = 46, g->N_full = 68, g->N_short = 66, g->E = NR_LDPC_BG1_NOF_EDGES;
    g->row_start = NR_LDPC_BG1_ROW_START, g->col = NR_LDPC_BG1_COL, g->shift = NR_LDPC_BG1_SHIFT[ils];
  } else {
    g->bgK = 10, g->bgM = 42, g->N_full = 52, g->N_short = 50, g->E = NR_LDPC_BG2_NOF_EDGES;
    g->row_start = NR_LDPC_BG2_ROW_START, g->col = NR_LDPC_BG2_COL, g->shift = NR_LDPC_BG2_SHIFT[ils];
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------ LDPC encoder
 * ldpc_encoder_impl.cpp:44-81 (lengths), ldpc_encoder_generic.cpp:56-223 (systematic accumulation, the four
 * high-rate closed forms, extension rows, output shortening by 2Z).                                              */
int orc_ldpc_encode(int bg, int Z, const uint8_t* in, uint8_t* out, unsigned out_len)
{
  graph_t g;
  if (get_graph(bg, Z, &g))
    return -1;
  unsigned K = g.bgK * Z;
  if (out_len > (unsigned)g.N_short * Z)
    return -2;
  unsigned cb_len = out_len + 2u * Z;
  if (cb_len < K + 4u * Z)
    cb_len = K + 4u * Z;
  if (cb_len % Z)
    cb_len = (cb_len / Z + 1) * Z;
  unsigned nof_layers = cb_len / Z - g.bgK;

  uint8_t* cb  = (uint8_t*)calloc((size_t)g.N_full * Z, 1);
  uint8_t* aux = (uint8_t*)calloc((size_t)g.bgM * Z, 1);
  memcpy(cb, in, K);
  /* aux[m][l] = XOR_k msg_k[(l + shift) % Z], fillers (254) count as 0 (generic.cpp:82). */
  for (int m = 0; m < g.bgM; ++m) {
    for (int e = g.row_start[m]; e < g.row_start[m + 1]; ++e) {
      int n = g.col[e];
      if (n >= g.bgK)
        continue;
      unsigned s = g.shift[e] % Z;
      for (unsigned l = 0; l < (unsigned)Z; ++l)
        aux[m * Z + l] ^= in[n * Z + (l + s) % Z] & 1u;
    }
  }
  uint8_t *p0 = cb + K, *p1 = p0 + Z, *p2 = p1 + Z, *p3 = p2 + Z;
  const uint8_t *a0 = aux, *a1 = aux + Z, *a2 = aux + 2 * Z, *a3 = aux + 3 * Z;
  for (int k = 0; k < Z; ++k) {
    if (bg == 1 && g.i_ls == 6) { /* generic.cpp:121-145 */
      int i = ((k - 105) % Z + Z) % Z;
      p0[k] = a0[i] ^ a1[i] ^ a2[i] ^ a3[i];
    } else if (bg == 2 && g.i_ls != 3 && g.i_ls != 7) { /* generic.cpp:197-223 */
      int i = ((k - 1) % Z + Z) % Z;
      p0[k] = a0[i] ^ a1[i] ^ a2[i] ^ a3[i];
    } else { /* generic.cpp:147-195 */
      p0[k] = a0[k] ^ a1[k] ^ a2[k] ^ a3[k];
    }
  }
  for (int k = 0; k < Z; ++k) {
    if (bg == 1 && g.i_ls == 6) {
      p1[k] = a0[k] ^ p0[k];
      p3[k] = a3[k] ^ p0[k];
      p2[k] = a2[k] ^ p3[k];
    } else if (bg == 1) {
      p1[k] = a0[k] ^ p0[(k + 1) % Z];
      p3[k] = a3[k] ^ p0[(k + 1) % Z];
      p2[k] = a2[k] ^ p3[k];
    } else if (g.i_ls == 3 || g.i_ls == 7) {
      p1[k] = a0[k] ^ p0[(k + 1) % Z];
      p2[k] = a1[k] ^ p1[k];
      p3[k] = a3[k] ^ p0[(k + 1) % Z];
    } else {
      p1[k] = a0[k] ^ p0[k];
      p2[k] = a1[k] ^ p1[k];
      p3[k] = a3[k] ^ p0[k];
    }
  }
  /* Extension rows (generic.cpp:90-111). */
  for (unsigned m = 4; m < nof_layers; ++m) {
    for (int i = 0; i < Z; ++i) {
      uint8_t t = aux[m * Z + i];
      for (int e = g.row_start[m]; e < g.row_start[m + 1]; ++e) {
        int n = g.col[e];
        if (n < g.bgK || n >= g.bgK + 4)
          continue;
        t ^= cb[n * Z + (i + g.shift[e] % Z) % Z];
      }
      cb[(g.bgK + m) * Z + i] = t;
    }
  }
  memcpy(out, cb + 2u * Z, out_len);
  free(cb);
  free(aux);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ LDPC decoder
 * ldpc_decoder_impl.cpp:60-297 (control flow, layer schedule) with the AVX2 arithmetic
 * (ldpc_decoder_avx2.cpp:66-243, avx2_support.h:65-106).                                                          */
static int sat8(int v)
{
  return v > 127 ? 127 : (v < -128 ? -128 : v);
}

/* ldpc_decoder_avx2.cpp:85-105 */
static int8_t v2c_rule(int8_t soft, int8_t c2v)
{
  int v = sat8((int)soft - (int)c2v);
  if (v > LLR_MAX)
    v = LLR_MAX;
  if (v < -LLR_MAX)
    v = -LLR_MAX;
  if (!(LLR_INF > soft))
    v = LLR_INF;
  if (!(soft > -LLR_INF))
    v = -LLR_INF;
  return (int8_t)v;
}

/* avx2_support.h:65-106 with sf = 0.8f: static_cast<uint16_t>(0.8f * 65536) = 52428; (byte * 52428) >> 16. */
static int8_t scale_rule(int8_t a)
{
  if (a > LLR_MAX || a < -LLR_MAX)
    return a;
  return (int8_t)((((unsigned)(uint8_t)a) * 52428u) >> 16);
}

/* ldpc_decoder_avx2.cpp:205-243 */
static int8_t soft_rule(int8_t c2v, int8_t v2c)
{
  int c_pinf = c2v > LLR_MAX, c_minf = c2v < -LLR_MAX, v_pinf = v2c > LLR_MAX, v_minf = v2c < -LLR_MAX;
  int s = sat8((int)c2v + (int)v2c);
  if (s > LLR_MAX || (c_pinf && !v_minf) || (v_pinf && !c_minf))
    s = LLR_INF;
  if (s < -LLR_MAX || (c_minf && !v_pinf) || (v_minf && !c_pinf))
    s = -LLR_INF;
  return (int8_t)s;
}

static void pack_hard_bits(uint8_t* out, const int8_t* soft, unsigned K)
{
  memset(out, 0, (K + 7) / 8);
  for (unsigned i = 0; i < K; ++i)
    if (soft[i] <= 0) /* log_likelihood_ratio.h:86 */
      out[i >> 3] |= (uint8_t)(0x80u >> (i & 7));
}

int orc_ldpc_decode(int           bg,
                    int           Z,
                    const int8_t* llr,
                    unsigned      in_len,
                    unsigned      nof_filler,
                    int           crc_poly,
                    unsigned      max_iter,
                    uint8_t*      out_packed,
                    int8_t*       soft_out)
{
  graph_t g;
  if (get_graph(bg, Z, &g))
    return -1;
  unsigned K = g.bgK * Z;
  if (in_len > (unsigned)g.N_short * Z || in_len < K + 2u * Z)
    return -2;
  /* impl.cpp:86-99: trim trailing zeros. */
  unsigned last = in_len;
  while (last > 0 && llr[last - 1] == 0)
    --last;
  if (last == 0) {
    if (crc_poly < 0) { /* bit_buffer::one(), bit_buffer.h:72-85 */
      memset(out_packed, 0xff, K / 8);
      if (K % 8)
        out_packed[K / 8] = (uint8_t)(0xff << (8 - K % 8));
    }
    return 0;
  }
  unsigned cb_len = last + 2u * Z;
  if (cb_len < K + 4u * Z)
    cb_len = K + 4u * Z;
  if (cb_len % Z)
    cb_len = (cb_len / Z + 1) * Z;
  unsigned nof_layers           = cb_len / Z - g.bgK;
  unsigned nof_significant_bits = K - nof_filler;

  size_t  NZ   = (size_t)g.N_full * Z;
  int8_t* soft = (int8_t*)calloc(NZ, 1);
  memcpy(soft + 2u * Z, llr, in_len); /* impl.cpp:148-173; tail beyond in_len defined as 0 */
  int8_t*  c2v   = (int8_t*)calloc((size_t)g.E * Z, 1); /* per edge, variable-index domain */
  uint8_t* init  = (uint8_t*)calloc(g.bgM, 1);
  int8_t*  v2c   = (int8_t*)malloc(20 * MAX_Z);
  int8_t*  rot   = (int8_t*)malloc(20 * MAX_Z);
  int      iters = 0;

  for (unsigned it = 0; it < max_iter; ++it) {
    for (unsigned m = 0; m < nof_layers; ++m) {
      int e0 = g.row_start[m], d = g.row_start[m + 1] - e0;
      /* variable-to-check (impl.cpp:175-200) + backward rotation (avx2.cpp:122) */
      for (int j = 0; j < d; ++j) {
        int      n = g.col[e0 + j];
        unsigned s = g.shift[e0 + j] % Z;
        for (int k = 0; k < Z; ++k)
          v2c[j * MAX_Z + k] = init[m] ? v2c_rule(soft[n * Z + k], c2v[(size_t)(e0 + j) * Z + k]) : soft[n * Z + k];
        for (int i = 0; i < Z; ++i)
          rot[j * MAX_Z + i] = v2c[j * MAX_Z + (i + s) % Z];
      }
      for (int i = 0; i < Z; ++i) {
        /* avx2.cpp:113-158 (init impl.cpp:239-244) */
        int8_t  min1 = LLR_MAX, min2 = LLR_MAX;
        int     idx = 0;
        uint8_t sp  = 0;
        for (int j = 0; j < d; ++j) {
          int8_t r = rot[j * MAX_Z + i];
          sp ^= (uint8_t)r;
          int8_t a    = (r == -128) ? (int8_t)-128 : (int8_t)abs(r);
          int    mask = min1 > a;
          int8_t help = mask ? min1 : a;
          if (mask) {
            min1 = a;
            idx  = j;
          }
          if (min2 > a)
            min2 = help;
        }
        /* avx2.cpp:160-203 */
        for (int j = 0; j < d; ++j) {
          int8_t  r   = rot[j * MAX_Z + i];
          int8_t  mag = scale_rule((idx == j) ? min2 : min1);
          uint8_t fs  = (uint8_t)r ^ sp;
          int8_t  c   = (fs & 0x80) ? (int8_t)(-mag) : mag;
          unsigned s  = g.shift[e0 + j] % Z;
          c2v[(size_t)(e0 + j) * Z + (i + s) % Z] = c;
        }
      }
      /* soft-bit update (impl.cpp:202-218) */
      for (int j = 0; j < d; ++j) {
        int n = g.col[e0 + j];
        for (int k = 0; k < Z; ++k)
          soft[n * Z + k] = soft_rule(c2v[(size_t)(e0 + j) * Z + k], v2c[j * MAX_Z + k]);
      }
      init[m] = 1;
    }
    if (crc_poly >= 0) { /* impl.cpp:126-133 */
      pack_hard_bits(out_packed, soft, K);
      if (orc_crc_packed(crc_poly, out_packed, nof_significant_bits) == 0) {
        iters = (int)it + 1;
        break;
      }
    }
  }
  if (crc_poly < 0)
    pack_hard_bits(out_packed, soft, K);
  if (soft_out)
    memcpy(soft_out, soft, NZ);
  free(soft);
  free(c2v);
  free(init);
  free(v2c);
  free(rot);
  return iters;
}

/* ------------------------------------------------------------------------------------------------ rate matching
 * ldpc_rate_matcher_impl.cpp:42-182.                                                                              */
static int rm_common(unsigned N, int rv, unsigned Nref, unsigned* Ncb, unsigned* k0, unsigned* sys_bits, unsigned* Z_out)
{
  static const double SF1[4] = {0, 17, 33, 56}, SF2[4] = {0, 13, 25, 43};
  const double*       sf;
  unsigned            nshort, bgK;
  if (N % 66 == 0) { /* BG1 tested first (impl.cpp:68-80) */
    sf = SF1, nshort = 66, bgK = 22;
  } else if (N % 50 == 0) {
    sf = SF2, nshort = 50, bgK = 10;
  } else
    return -1;
  unsigned Z = N / nshort;
  *Ncb       = (Nref > 0 && Nref < N) ? Nref : N;
  double tmp = (sf[rv] * (double)*Ncb) / (double)N;
  *k0        = (unsigned)((uint16_t)floor(tmp)) * Z;
  *sys_bits  = (bgK - 2) * Z;
  *Z_out     = Z;
  return 0;
}

int orc_ldpc_rate_match(int rv, int mod, unsigned Nref, unsigned nof_filler, const uint8_t* in, unsigned N, uint8_t* out, unsigned E)
{
  unsigned Ncb, k0, sys, Z;
  if (rm_common(N, rv, Nref, &Ncb, &k0, &sys, &Z) || mod < 1 || E % mod)
    return -1;
  uint8_t* sel = (uint8_t*)malloc(E ? E : 1);
  unsigned f0 = sys - nof_filler, f1 = sys;
  /* select_bits (impl.cpp:107-149): positional filler skip, wrap at Ncb. */
  unsigned idx = k0, o = 0;
  while (o < E) {
    if (idx >= f0 && idx < f1)
      idx = f1;
    if (idx >= Ncb) { /* the reference wraps with % after each chunk; a k0/filler jump never lands beyond Ncb unless Ncb <= f1 */
      idx %= Ncb;
      continue;
    }
    sel[o++] = in[idx];
    idx      = (idx + 1) % Ncb;
  }
  if (mod == 1) {
    memcpy(out, sel, E);
  } else { /* interleave (impl.cpp:152-182) */
    unsigned Kq = E / mod;
    for (unsigned i = 0, oi = 0; i < Kq; ++i)
      for (int j = 0; j < mod; ++j, ++oi)
        out[oi] = sel[Kq * j + i];
  }
  free(sel);
  return 0;
}

/* ldpc_rate_dematcher_impl.cpp:43-254 with the AVX2 combine (ldpc_rate_dematcher_avx2_impl.cpp:45-58). */
static int8_t combine_rule(int8_t a, int8_t b)
{
  int s = sat8((int)a + (int)b);
  if (s > LLR_MAX)
    s = LLR_MAX;
  if (s < -LLR_MAX)
    s = -LLR_MAX;
  return (int8_t)s;
}

int orc_ldpc_rate_dematch(int rv, int mod, unsigned Nref, unsigned nof_filler, int new_data, const int8_t* in, unsigned E, int8_t* out, unsigned N)
{
  unsigned Ncb, k0, sys, Z;
  if (rm_common(N, rv, Nref, &Ncb, &k0, &sys, &Z) || mod < 1 || E % mod)
    return -1;
  int8_t* d = (int8_t*)malloc(E ? E : 1);
  if (mod == 1) {
    memcpy(d, in, E);
  } else { /* deinterleave (impl.cpp:200-254) */
    unsigned Kq = E / mod;
    for (unsigned i = 0, ii = 0; i < Kq; ++i)
      for (int j = 0; j < mod; ++j, ++ii)
        d[Kq * j + i] = in[ii];
  }
  /* allot_llrs (impl.cpp:125-198), restated chunk by chunk so that exactly the positions the reference zeroes,
   * overwrites, combines or leaves untouched are treated the same way (including its quirks: with new_data only
   * [0,k0) / [0,info) and the tail after the last written position are cleared; with a limited buffer the tail
   * clear is taken from the end of the full-length block, impl.cpp:195-197). */
  unsigned info = sys - nof_filler;
  int      copy = new_data;
  unsigned idx = k0, pos = 0, rem = E;
  while (rem > 0) {
    if (idx < info) {
      unsigned n = info - idx < rem ? info - idx : rem;
      if (copy) {
        memset(out, 0, idx);
        memcpy(out + idx, d + pos, n);
      } else {
        for (unsigned i = 0; i < n; ++i)
          out[idx + i] = combine_rule(out[idx + i], d[pos + i]);
      }
      idx += n, pos += n, rem -= n;
    } else if (copy) {
      memset(out, 0, info);
    }
    if (copy)
      memset(out + info, LLR_INF, nof_filler);
    if (idx < sys)
      idx = sys;
    unsigned room = Ncb - idx;
    unsigned n    = room < rem ? room : rem;
    if (copy) {
      memcpy(out + idx, d + pos, n);
    } else {
      for (unsigned i = 0; i < n; ++i)
        out[idx + i] = combine_rule(out[idx + i], d[pos + i]);
    }
    idx = (idx + n) % Ncb, pos += n, rem -= n;
    if (rem > 0)
      copy = 0;
  }
  if (copy && idx != 0)
    memset(out + N - (Ncb - idx), 0, Ncb - idx);
  free(d);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ segmentation
 * ldpc.h:128-207, ldpc_segmenter_impl.cpp:57-67,89-334.                                                           */
int orc_ldpc_segmentation(unsigned tbs, int bg, int mod, unsigned nof_layers, unsigned nof_ch_symbols, orc_segmentation_t* s)
{
  memset(s, 0, sizeof(*s));
  s->tbs             = tbs;
  s->nof_tb_crc_bits = (tbs <= 3824) ? 16 : 24;
  unsigned B         = tbs + s->nof_tb_crc_bits;
  unsigned Kcb       = (bg == 1) ? 8448 : 3840;
  s->nof_cbs         = (B <= Kcb) ? 1 : (B + (Kcb - 24) - 1) / (Kcb - 24);
  if (s->nof_cbs > 52)
    return -1;
  unsigned Bp = B + ((s->nof_cbs > 1) ? 24 * s->nof_cbs : 0);
  unsigned Kb = 22;
  if (bg == 2)
    Kb = (B > 640) ? 10 : (B > 560) ? 9 : (B > 192) ? 8 : 6;
  s->Z = 0;
  for (int i = 0; i < 51; ++i) {
    if (NR_LDPC_LIFTING_SIZES[i] * s->nof_cbs * Kb >= Bp) {
      s->Z = NR_LDPC_LIFTING_SIZES[i];
      break;
    }
  }
  if (!s->Z)
    return -1;
  s->K               = ((bg == 1) ? 22 : 10) * s->Z;
  s->N               = s->K * ((bg == 1) ? 3 : 5);
  s->nof_cb_crc_bits = (s->nof_cbs > 1) ? 24 : 0;
  s->cb_info_bits    = (Bp + s->nof_cbs - 1) / s->nof_cbs - s->nof_cb_crc_bits;
  s->zero_pad        = (s->cb_info_bits + s->nof_cb_crc_bits) * s->nof_cbs - Bp;
  s->nof_filler_bits = s->K - s->cb_info_bits - s->nof_cb_crc_bits;
  unsigned sym_layer = nof_ch_symbols / nof_layers;
  s->nof_short_segments = s->nof_cbs - (sym_layer % s->nof_cbs);
  unsigned off          = 0;
  for (unsigned i = 0; i < s->nof_cbs; ++i) {
    unsigned t      = (i < s->nof_short_segments) ? sym_layer / s->nof_cbs : (sym_layer + s->nof_cbs - 1) / s->nof_cbs;
    s->E[i]         = t * nof_layers * mod;
    s->cw_offset[i] = off;
    off += s->E[i];
  }
  if (off != nof_ch_symbols * mod)
    return -2;
  /* pusch_decoder_impl.cpp:44-55 */
  s->crc_poly = (s->nof_cbs > 1) ? ORC_CRC24B : ((tbs > 3824) ? ORC_CRC24A : ORC_CRC16);
  return 0;
}

static unsigned get_bit(const uint8_t* p, unsigned i)
{
  return (p[i >> 3] >> (7 - (i & 7))) & 1u;
}
static void set_bit(uint8_t* p, unsigned i, unsigned b)
{
  if (b)
    p[i >> 3] |= (uint8_t)(0x80u >> (i & 7));
  else
    p[i >> 3] &= (uint8_t)~(0x80u >> (i & 7));
}

/* pdsch_encoder_impl.cpp:28-65 + ldpc_segmenter_impl.cpp:89-234. */
int orc_pdsch_encode(int bg, int rv, int mod, unsigned Nref, unsigned nof_layers, unsigned nof_ch_symbols,
                     const uint8_t* tb, unsigned tb_bytes, uint8_t* codeword)
{
  orc_segmentation_t s;
  if (orc_ldpc_segmentation(tb_bytes * 8, bg, mod, nof_layers, nof_ch_symbols, &s))
    return -1;
  uint32_t tb_crc = orc_crc_packed((s.nof_tb_crc_bits == 16) ? ORC_CRC16 : ORC_CRC24A, tb, s.tbs);
  uint8_t* msg    = (uint8_t*)malloc(s.K);
  uint8_t* cb     = (uint8_t*)malloc(s.N);
  unsigned tb_off = 0;
  for (unsigned c = 0; c < s.nof_cbs; ++c) {
    unsigned used = 0;
    unsigned take = s.cb_info_bits;
    int      last = (c == s.nof_cbs - 1);
    if (last)
      take -= s.nof_tb_crc_bits + s.zero_pad;
    for (unsigned i = 0; i < take; ++i)
      msg[used++] = (uint8_t)get_bit(tb, tb_off + i);
    tb_off += take;
    if (last) {
      for (unsigned i = 0; i < s.nof_tb_crc_bits; ++i)
        msg[used++] = (uint8_t)((tb_crc >> (s.nof_tb_crc_bits - 1 - i)) & 1u);
      for (unsigned i = 0; i < s.zero_pad; ++i)
        msg[used++] = 0;
    }
    if (s.nof_cb_crc_bits) {
      uint32_t crc = orc_crc_bits(ORC_CRC24B, msg, used);
      for (unsigned i = 0; i < 24; ++i)
        msg[used++] = (uint8_t)((crc >> (23 - i)) & 1u);
    }
    while (used < s.K)
      msg[used++] = FILLER_BIT; /* pdsch_encoder_impl.cpp:49-50 */
    orc_ldpc_encode(bg, (int)s.Z, msg, cb, s.N);
    orc_ldpc_rate_match(rv, mod, Nref, s.nof_filler_bits, cb, s.N, codeword + s.cw_offset[c], s.E[c]);
  }
  free(msg);
  free(cb);
  return (int)s.nof_cbs;
}

/* pusch_decoder_impl.cpp:121-225. */
int orc_pusch_decode(int bg, int rv, int mod, unsigned Nref, unsigned nof_layers, unsigned nof_ch_symbols,
                     unsigned tb_bytes, int new_data, const int8_t* llrs, unsigned max_iter, int early_stop,
                     int8_t* softbuf, uint8_t* cb_crc, uint8_t* cb_msgs, uint8_t* tb_out, int* iters_minmax)
{
  orc_segmentation_t s;
  unsigned           tbs = tb_bytes * 8;
  if (orc_ldpc_segmentation(tbs, bg, mod, nof_layers, nof_ch_symbols, &s))
    return -1;
  unsigned tb_and_crc = tbs + ((s.nof_cbs > 1) ? 24 : 0);
  uint8_t* tmp_tb     = (uint8_t*)calloc((tb_and_crc + 7) / 8 + 1, 1);
  unsigned msg_bytes  = (s.K + 7) / 8;
  if (new_data)
    memset(cb_crc, 0, s.nof_cbs);
  unsigned tb_off = 0;
  int      imin = 0, imax = 0, nobs = 0;
  for (unsigned c = 0; c < s.nof_cbs; ++c) {
    unsigned nof_data_bits = s.K - ((s.nof_cbs == 1) ? s.nof_tb_crc_bits : 24) - s.nof_filler_bits;
    unsigned free_bits     = tb_and_crc - tb_off;
    unsigned nof_new       = free_bits < nof_data_bits ? free_bits : nof_data_bits;
    int8_t*  cb            = softbuf + (size_t)c * s.N;
    uint8_t* msg           = cb_msgs + (size_t)c * msg_bytes;
    orc_ldpc_rate_dematch(rv, mod, Nref, s.nof_filler_bits, new_data, llrs + s.cw_offset[c], s.E[c], cb, s.N);
    if (!cb_crc[c]) {
      int it;
      if (early_stop) {
        it = orc_ldpc_decode(bg, (int)s.Z, cb, s.N, s.nof_filler_bits, (int)s.crc_poly, max_iter, msg, 0);
      } else {
        orc_ldpc_decode(bg, (int)s.Z, cb, s.N, s.nof_filler_bits, -1, max_iter, msg, 0);
        it = (orc_crc_packed((int)s.crc_poly, msg, s.K - s.nof_filler_bits) == 0) ? (int)max_iter : 0;
      }
      int upd = it ? it : (int)max_iter;
      if (it)
        cb_crc[c] = 1;
      if (!nobs || upd < imin)
        imin = upd;
      if (!nobs || upd > imax)
        imax = upd;
      ++nobs;
    }
    for (unsigned i = 0; i < nof_new; ++i)
      set_bit(tmp_tb, tb_off + i, get_bit(msg, i));
    tb_off += nof_new;
  }
  int ok = 0;
  if (s.nof_cbs == 1) {
    ok = cb_crc[0];
    if (ok)
      memcpy(tb_out, tmp_tb, tb_bytes);
  } else {
    int all = 1;
    for (unsigned c = 0; c < s.nof_cbs; ++c)
      all &= cb_crc[c];
    if (all) {
      memcpy(tb_out, tmp_tb, tb_bytes);
      if (orc_crc_packed(ORC_CRC24A, tmp_tb, tb_and_crc) == 0)
        ok = 1;
      else
        memset(cb_crc, 0, s.nof_cbs);
    }
  }
  iters_minmax[0] = imin;
  iters_minmax[1] = imax;
  free(tmp_tb);
  return ok;
}

/* ------------------------------------------------------------------------------------------------ DFT / OFDM
 * dft_processor contract: include/srsran/phy/generic_functions/dft_processor.h:34-73 (unnormalised, DIRECT = exp(-j..)). */
#include <complex.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static void dft_double(unsigned N, int inverse, const double complex* in, double complex* out)
{
  double complex* w = (double complex*)malloc(sizeof(double complex) * N);
  for (unsigned j = 0; j < N; ++j) {
    double a = (inverse ? 2.0 : -2.0) * M_PI * (double)j / (double)N;
    w[j]     = cos(a) + I * sin(a);
  }
  for (unsigned k = 0; k < N; ++k) {
    double complex acc = 0;
    unsigned       idx = 0;
    for (unsigned n = 0; n < N; ++n) {
      acc += in[n] * w[idx];
      idx += k;
      if (idx >= N)
        idx -= N;
    }
    out[k] = acc;
  }
  free(w);
}

int orc_dft(unsigned N, int inverse, const float* in, float* out)
{
  double complex* a = (double complex*)malloc(sizeof(double complex) * N);
  double complex* b = (double complex*)malloc(sizeof(double complex) * N);
  for (unsigned i = 0; i < N; ++i)
    a[i] = (double)in[2 * i] + I * (double)in[2 * i + 1];
  dft_double(N, inverse, a, b);
  for (unsigned i = 0; i < N; ++i) {
    out[2 * i]     = (float)creal(b[i]);
    out[2 * i + 1] = (float)cimag(b[i]);
  }
  free(a);
  free(b);
  return 0;
}

/* include/srsran/ran/cyclic_prefix.h:96-107 + phy_time_unit.h:98-108 (normal CP). */
static unsigned cp_samples(unsigned mu, unsigned sym_sf, unsigned dft_size)
{
  unsigned units = 144u >> mu;
  if (sym_sf == 0 || sym_sf == 7u * (1u << mu))
    units += 16;
  return (unsigned)((unsigned long long)units * (1u << mu) * dft_size / 2048u);
}

unsigned orc_ofdm_slot_size(const orc_ofdm_cfg* c, unsigned slot_index)
{
  unsigned n = 0;
  for (unsigned l = 0; l < 14; ++l)
    n += cp_samples(c->numerology, slot_index * 14 + l, c->dft_size) + c->dft_size;
  return n;
}

/* phase_compensation_lut.h:55-82 */
static float complex phase_coef(const orc_ofdm_cfg* c, unsigned sym_sf, int is_tx)
{
  double   srate = 15000.0 * (double)(1u << c->numerology) * (double)c->dft_size;
  unsigned off   = 0;
  for (unsigned s = 0; s <= sym_sf; ++s) {
    off += cp_samples(c->numerology, s, c->dft_size);
    if (s == sym_sf)
      break;
    off += c->dft_size;
  }
  double t     = (double)off / srate;
  double phase = (is_tx ? -1.0 : 1.0) * 2.0 * M_PI * c->center_freq_hz * t;
  double complex e = cexp(I * phase);
  return (float)creal(e) + I * (float)cimag(e);
}

static float complex cmulf_(float complex a, float complex b)
{ /* plain complex product, as the reference's srsvec::prod / sc_prod do (no inf/nan fix-ups) */
  float ar = crealf(a), ai = cimagf(a), br = crealf(b), bi = cimagf(b);
  return (ar * br - ai * bi) + I * (ar * bi + ai * br);
}

/* ofdm_demodulator_impl.cpp:93-138 */
int orc_ofdm_demod_slot(const orc_ofdm_cfg* c, unsigned slot_index, const float* in, float* grid_out)
{
  unsigned        N = c->dft_size, rg = c->bw_rb * 12;
  double complex* a = (double complex*)malloc(sizeof(double complex) * N);
  double complex* b = (double complex*)malloc(sizeof(double complex) * N);
  unsigned        pos = 0;
  for (unsigned l = 0; l < 14; ++l) {
    unsigned sym = slot_index * 14 + l;
    unsigned cp  = cp_samples(c->numerology, sym, N);
    const float* src = in + 2 * (size_t)(pos + cp - c->window_offset);
    for (unsigned i = 0; i < N; ++i)
      a[i] = (double)src[2 * i] + I * (double)src[2 * i + 1];
    dft_double(N, 0, a, b);
    float complex coef = phase_coef(c, sym, 0);
    coef = (crealf(coef) * c->scale) + I * (cimagf(coef) * c->scale);
    float complex omega = 0;
    if (c->window_offset) /* :70-72, single precision like the reference */
      omega = I * ((float)c->window_offset * (float)(2.0 * M_PI) / (float)N);
    for (unsigned k = 0; k < rg; ++k) {
      unsigned      bin = (k < rg / 2) ? N - rg / 2 + k : k - rg / 2;
      float complex x   = (float)creal(b[bin]) + I * (float)cimag(b[bin]);
      float complex v   = cmulf_(x, coef);
      if (c->window_offset)
        v = cmulf_(v, cexpf(omega * (float)bin));
      grid_out[2 * ((size_t)l * rg + k)]     = crealf(v);
      grid_out[2 * ((size_t)l * rg + k) + 1] = cimagf(v);
    }
    pos += cp + N;
  }
  free(a);
  free(b);
  return 0;
}

/* ofdm_modulator_impl.cpp:55-99 */
int orc_ofdm_mod_slot(const orc_ofdm_cfg* c, unsigned slot_index, const float* grid_in, float* out)
{
  unsigned        N = c->dft_size, rg = c->bw_rb * 12;
  double complex* a = (double complex*)malloc(sizeof(double complex) * N);
  double complex* b = (double complex*)malloc(sizeof(double complex) * N);
  unsigned        pos = 0;
  for (unsigned l = 0; l < 14; ++l) {
    unsigned sym = slot_index * 14 + l;
    unsigned cp  = cp_samples(c->numerology, sym, N);
    for (unsigned i = 0; i < N; ++i)
      a[i] = 0;
    const float* g = grid_in + 2 * (size_t)l * rg;
    for (unsigned k = 0; k < rg / 2; ++k) {
      a[N - rg / 2 + k] = (double)g[2 * k] + I * (double)g[2 * k + 1];
      a[k]              = (double)g[2 * (rg / 2 + k)] + I * (double)g[2 * (rg / 2 + k) + 1];
    }
    dft_double(N, 1, a, b);
    float complex coef = phase_coef(c, sym, 1);
    coef = (crealf(coef) * c->scale) + I * (cimagf(coef) * c->scale);
    float* dst = out + 2 * (size_t)pos;
    for (unsigned i = 0; i < N + cp; ++i) {
      unsigned      j = (i < cp) ? N - cp + i : i - cp;
      float complex x = (float)creal(b[j]) + I * (float)cimag(b[j]);
      float complex v = cmulf_(x, coef);
      dst[2 * i]      = crealf(v);
      dst[2 * i + 1]  = cimagf(v);
    }
    pos += cp + N;
  }
  free(a);
  free(b);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ Gold sequence
 * TS 38.211 5.2.1: c(n) = x1(n + 1600) ^ x2(n + 1600), x1(n+31) = x1(n+3)^x1(n), x2(n+31) = x2(n+3)^x2(n+2)^x2(n+1)^x2(n),
 * x1(0)=1, x2 = c_init (lib/phy/upper/sequence_generators/pseudo_random_generator_impl.cpp:43-83). */
void orc_gold_sequence(unsigned c_init, unsigned offset, unsigned nbits, uint8_t* out)
{
  unsigned total = 1600 + offset + nbits + 31;
  uint8_t* x1    = (uint8_t*)calloc(total + 31, 1);
  uint8_t* x2    = (uint8_t*)calloc(total + 31, 1);
  x1[0]          = 1;
  for (int i = 0; i < 31; ++i)
    x2[i] = (c_init >> i) & 1u;
  for (unsigned n = 0; n + 31 < total + 31; ++n) {
    x1[n + 31] = x1[n + 3] ^ x1[n];
    x2[n + 31] = x2[n + 3] ^ x2[n + 2] ^ x2[n + 1] ^ x2[n];
  }
  for (unsigned n = 0; n < nbits; ++n)
    out[n] = x1[n + offset + 1600] ^ x2[n + offset + 1600];
  free(x1);
  free(x2);
}

/* re_pattern / w_f / w_t of dmrs_pusch_estimator_impl.cpp:31-69 */
static void dmrs_port_params(int type2, unsigned layer, unsigned* re_idx /* pilots positions inside a PRB */, unsigned* nre, float* wf1, float* wt1)
{
  if (!type2) {
    unsigned delta = (layer / 2) % 2; /* ports 0,1,4,5 -> even; 2,3,6,7 -> odd */
    *nre           = 6;
    for (unsigned i = 0; i < 6; ++i)
      re_idx[i] = 2 * i + delta;
    *wf1 = (layer % 2) ? -1.f : 1.f;
    *wt1 = (layer >= 4) ? -1.f : 1.f;
  } else {
    unsigned delta = 2 * ((layer / 2) % 3);
    *nre           = 4;
    re_idx[0] = delta, re_idx[1] = delta + 1, re_idx[2] = delta + 6, re_idx[3] = delta + 7;
    *wf1 = (layer % 2) ? -1.f : 1.f;
    *wt1 = (layer >= 6) ? -1.f : 1.f;
  }
}

int orc_dmrs_pusch_estimate(unsigned numerology, unsigned slot_in_frame, int dmrs_type2, unsigned scrambling_id, int n_scid,
                            float scaling, const uint8_t* symbols_mask, const uint8_t* rb_mask, unsigned nof_prb_grid,
                            unsigned first_symbol, unsigned nof_symbols, unsigned nof_tx_layers, unsigned nof_rx_ports,
                            const float* grid_in, float* ce_out, float* scalars_out)
{
  const unsigned DFT_SIZE = 4096, HALF_CP = ((144 / 2) * DFT_SIZE) / 2048;
  unsigned       nsc = nof_prb_grid * 12, nsymb_out = first_symbol + nof_symbols;
  unsigned       per_rb = dmrs_type2 ? 4 : 6;
  unsigned       nprb = 0;
  for (unsigned r = 0; r < nof_prb_grid; ++r)
    nprb += rb_mask[r] ? 1 : 0;
  unsigned dmrs_syms[14], nds = 0;
  for (unsigned l = first_symbol; l < first_symbol + nof_symbols; ++l)
    if (symbols_mask[l])
      dmrs_syms[nds++] = l;
  if (!nprb || !nds)
    return -1;
  unsigned np = nprb * per_rb; /* pilots per symbol */
  /* Layer-0 pilots (dmrs_pusch_estimator_impl.cpp:112-172, dmrs_helper.h:45-96): QPSK from c(2i), c(2i+1), counted from PRB 0. */
  float complex* pil0 = (float complex*)malloc(sizeof(float complex) * np * nds);
  uint8_t*       c    = (uint8_t*)malloc(2 * per_rb * nof_prb_grid + 8);
  const float    amp  = (float)M_SQRT1_2;
  for (unsigned d = 0; d < nds; ++d) {
    unsigned long long t = ((unsigned long long)(14 * slot_in_frame + dmrs_syms[d] + 1) * (2ull * scrambling_id + 1)) % (1ull << 31);
    unsigned c_init = (unsigned)((t * (1ull << 17) + (2ull * scrambling_id + (n_scid ? 1 : 0))) % (1ull << 31));
    orc_gold_sequence(c_init, 0, 2 * per_rb * nof_prb_grid, c);
    unsigned i = 0;
    for (unsigned r = 0; r < nof_prb_grid; ++r) {
      if (!rb_mask[r])
        continue;
      for (unsigned q = 0; q < per_rb; ++q, ++i) {
        unsigned g = r * per_rb + q;
        pil0[d * np + i] = amp * (1.f - 2.f * c[2 * g]) + I * (amp * (1.f - 2.f * c[2 * g + 1]));
      }
    }
  }
  free(c);
  (void)numerology;
  float complex* pil  = (float complex*)malloc(sizeof(float complex) * np * nds);
  float complex* rx   = (float complex*)malloc(sizeof(float complex) * np * nds);
  float complex* lse  = (float complex*)malloc(sizeof(float complex) * np);
  float complex* ce   = (float complex*)malloc(sizeof(float complex) * nprb * 12);
  double complex* fa  = (double complex*)malloc(sizeof(double complex) * DFT_SIZE);
  double complex* fb  = (double complex*)malloc(sizeof(double complex) * DFT_SIZE);
  for (unsigned ly = 0; ly < nof_tx_layers; ++ly) {
    unsigned re_idx[6], nre;
    float    wf1, wt1;
    dmrs_port_params(dmrs_type2, ly, re_idx, &nre, &wf1, &wt1);
    /* layer weights (dmrs_pusch_estimator_impl.cpp:175-206) */
    for (unsigned d = 0; d < nds; ++d)
      for (unsigned i = 0; i < np; ++i) {
        float complex v = pil0[d * np + i];
        if (ly != 0) {
          if (wt1 != 1.f && (d % 2 == 1))
            v = v * wt1;
          if (wf1 != 1.f && (i % 2 == 1))
            v = v * wf1;
        }
        pil[d * np + i] = v;
      }
    for (unsigned p = 0; p < nof_rx_ports; ++p) {
      const float* g = grid_in + 2 * (size_t)p * 14 * nsc;
      /* extract pilots REs (port_channel_estimator_average_impl.cpp:227-269) */
      for (unsigned d = 0; d < nds; ++d) {
        unsigned i = 0;
        for (unsigned r = 0; r < nof_prb_grid; ++r) {
          if (!rb_mask[r])
            continue;
          for (unsigned q = 0; q < nre; ++q, ++i) {
            size_t k       = (size_t)dmrs_syms[d] * nsc + r * 12 + re_idx[q];
            rx[d * np + i] = g[2 * k] + I * g[2 * k + 1];
          }
        }
      }
      float epre = 0, rsrp = 0;
      for (unsigned i = 0; i < np; ++i)
        lse[i] = 0;
      for (unsigned d = 0; d < nds; ++d) {
        float e = 0;
        for (unsigned i = 0; i < np; ++i) {
          lse[i] += cmulf_(rx[d * np + i], conjf(pil[d * np + i]));
          e += crealf(rx[d * np + i]) * crealf(rx[d * np + i]) + cimagf(rx[d * np + i]) * cimagf(rx[d * np + i]);
        }
        epre += e;
      }
      {
        float e = 0;
        for (unsigned i = 0; i < np; ++i)
          e += crealf(lse[i]) * crealf(lse[i]) + cimagf(lse[i]) * cimagf(lse[i]);
        rsrp += e / (float)nds;
      }
      float total_scaling = 1.0f / ((float)nds * scaling);
      for (unsigned i = 0; i < np; ++i)
        lse[i] = lse[i] * total_scaling;
      /* noise (:271-310) */
      float noise = 0;
      {
        unsigned win = nre;
        for (unsigned d = 0; d < nds; ++d) {
          float pw = 0;
          for (unsigned b = 0; b < np; b += win) {
            float complex avg = 0;
            for (unsigned i = 0; i < win; ++i)
              avg += lse[b + i];
            avg = avg / (float)win;
            avg = cmulf_(avg, -scaling + 0.0f * I);
            for (unsigned i = 0; i < win; ++i) {
              float complex pr = cmulf_(avg, pil[d * np + b + i]) + rx[d * np + b + i];
              pw += crealf(pr) * crealf(pr) + cimagf(pr) * cimagf(pr);
            }
          }
          noise += pw / (float)np * (float)win;
        }
      }
      /* time alignment (:312-347) */
      float ta;
      {
        for (unsigned i = 0; i < DFT_SIZE; ++i)
          fa[i] = 0;
        unsigned i = 0;
        for (unsigned r = 0; r < nof_prb_grid; ++r) {
          if (!rb_mask[r])
            continue;
          for (unsigned q = 0; q < nre; ++q, ++i)
            fa[r * 12 + re_idx[q]] = (double)crealf(lse[i]) + I * (double)cimagf(lse[i]);
        }
        dft_double(DFT_SIZE, 1, fa, fb);
        unsigned id = 0, ia = 0;
        float    md = -1, ma = -1;
        for (unsigned k = 0; k < HALF_CP; ++k) {
          float vd = (float)(creal(fb[k]) * creal(fb[k]) + cimag(fb[k]) * cimag(fb[k]));
          if (vd > md)
            md = vd, id = k;
          unsigned kk = DFT_SIZE - HALF_CP + k;
          float    va = (float)(creal(fb[kk]) * creal(fb[kk]) + cimag(fb[kk]) * cimag(fb[kk]));
          if (va > ma)
            ma = va, ia = k;
        }
        ta = (md >= ma) ? (float)id : -(float)(HALF_CP - ia);
      }
      /* interpolation (interpolator_linear_impl.cpp:32-79): offset = first pilot RE, stride = distance to the second */
      {
        unsigned nout = nprb * 12, offset = re_idx[0], stride = re_idx[1] - re_idx[0];
        if (nout == np) {
          memcpy(ce, lse, sizeof(float complex) * np);
        } else if (np == 1) {
          for (unsigned k = 0; k < nout; ++k)
            ce[k] = lse[0];
        } else {
          unsigned io = offset, ii = 0;
          for (unsigned k = 0; k <= io; ++k)
            ce[k] = lse[0];
          for (unsigned next = io + stride; next < nout; next += stride) {
            float complex jump = (lse[ii + 1] - lse[ii]) / (float)stride;
            float complex val  = ce[io];
            for (unsigned k = 0; k < stride; ++k) {
              val += jump;
              ce[io + 1 + k] = val;
            }
            io = next;
            ++ii;
          }
          for (unsigned k = io + 1; k < nout; ++k)
            ce[k] = lse[ii];
        }
      }
      /* broadcast to every symbol of the allocation (:216-224) */
      for (unsigned l = first_symbol; l < first_symbol + nof_symbols; ++l) {
        float*   dst = ce_out + 2 * ((((size_t)ly * nof_rx_ports + p) * nsymb_out + l) * nsc);
        unsigned j   = 0;
        for (unsigned r = 0; r < nof_prb_grid; ++r) {
          if (!rb_mask[r])
            continue;
          for (unsigned k = 0; k < 12; ++k) {
            dst[2 * (r * 12 + k)]     = crealf(ce[j * 12 + k]);
            dst[2 * (r * 12 + k) + 1] = cimagf(ce[j * 12 + k]);
          }
          ++j;
        }
      }
      /* scalars (:118-144) */
      float ndp = (float)(np * nds);
      rsrp /= ndp;
      epre /= ndp;
      float scs_khz   = 15.0f * (float)(1u << numerology);
      float ta_s      = ta / ((float)DFT_SIZE * scs_khz * 1000.0F);
      float noise_var = noise / (float)(nre * nds - 1);
      if (nds < 3)
        noise_var = powf(10.0F, -30.0F / 10.0F) * epre;
      float datarp = rsrp / scaling / scaling;
      float snr    = (noise_var != 0) ? datarp / noise_var : 1000.f;
      float* sc    = scalars_out + 5 * ((size_t)p * nof_tx_layers + ly);
      sc[0] = rsrp, sc[1] = epre, sc[2] = noise_var, sc[3] = snr, sc[4] = ta_s;
    }
  }
  free(pil0), free(pil), free(rx), free(lse), free(ce), free(fa), free(fb);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ polar */
#include "../srsran_project_23.5_amd/csrc/tables/nr_polar_tables.h"

/* polar_code_impl.cpp:325-490 */
int orc_polar_code_set(orc_polar_code_t* c, unsigned K, unsigned E, unsigned nMax, int ibil)
{
  memset(c, 0, sizeof(*c));
  c->K = K, c->E = E, c->nMax = nMax, c->ibil = (unsigned)ibil;
  if (E > 8192)
    return -1;
  if (nMax == 9) {
    if (K < 36 || K > 164)
      return -1;
  } else if (nMax == 10) {
    if (K < 18 || (K > 25 && K < 31) || K > 1023)
      return -1;
  } else
    return -1;
  unsigned nPC = 0, nWmPC = 0;
  if (K <= 25) {
    nPC = 3;
    if (E > K + 189)
      nWmPC = 1;
  }
  if (!(K + nPC < E))
    return -1;
  unsigned e = 1;
  for (; e <= 13; ++e)
    if ((1u << e) >= E)
      break;
  unsigned n1 = ((8 * E <= 9 * (1u << (e - 1))) && (16 * K < 9 * E)) ? e - 1 : e;
  unsigned k  = 0;
  for (; k <= 10; ++k)
    if ((1u << k) >= K)
      break;
  unsigned n2 = k + 3;
  unsigned n  = n1 < n2 ? n1 : n2;
  if (nMax < n)
    n = nMax;
  if (n < 5)
    n = 5;
  unsigned N = 1u << n;
  if (!(K < N))
    return -1;
  c->n = n, c->N = N, c->nPC = nPC, c->nWmPC = nWmPC;
  /* tables for this n */
  uint16_t mother[1024];
  unsigned m = 0;
  for (unsigned i = 0; i < 1024; ++i)
    if (NR_POLAR_Q1024[i] < N)
      mother[m++] = NR_POLAR_Q1024[i];
  for (unsigned j = 0; j < N; ++j)
    c->blk_interleaver[j] = (uint16_t)(NR_POLAR_SUBBLOCK_P[32 * j / N] * (N / 32) + j % (N / 32));
  uint16_t        tmpK[1024];
  const uint16_t* Kset = mother + (N - (K + nPC));
  if (N > E) {
    unsigned T = 0, nF = N - E, N_th = 3 * N / 4;
    uint16_t F[1024];
    if (16 * K <= 7 * E) { /* puncturing */
      T = (E >= N_th) ? N_th - (E >> 1) - 1 : 9 * N / 16 - (E >> 2);
      for (unsigned i = 0; i < nF; ++i)
        F[i] = c->blk_interleaver[i];
    } else { /* shortening */
      for (unsigned i = 0; i < nF; ++i)
        F[i] = c->blk_interleaver[E + i];
    }
    unsigned o = 0; /* setdiff_stable (:294-323): drops x <= T and members of F */
    for (unsigned i = 0; i < N; ++i) {
      int flag = mother[i] <= T;
      for (unsigned j = 0; j < nF && !flag; ++j)
        flag = (mother[i] == F[j]);
      if (!flag)
        tmpK[o++] = mother[i];
    }
    Kset = tmpK + (o - K - nPC);
  }
  unsigned npc_lo = (nPC > nWmPC) ? nPC - nWmPC : 0;
  for (unsigned i = 0; i < npc_lo; ++i)
    c->PC_set[i] = Kset[i];
  if (nWmPC == 1)
    c->PC_set[nPC - 1] = (K <= 21) ? 252 : 248;
  for (unsigned i = 0; i < K + nPC; ++i)
    c->K_set[Kset[i]] = 1;
  for (unsigned i = 0; i + 1 < nPC; ++i) /* sort */
    for (unsigned j = i + 1; j < nPC; ++j)
      if (c->PC_set[j] < c->PC_set[i]) {
        uint16_t t = c->PC_set[i];
        c->PC_set[i] = c->PC_set[j], c->PC_set[j] = t;
      }
  c->PC_set[nPC] = 1024;
  return 0;
}

static void polar_transform(uint8_t* x, unsigned n)
{ /* polar_encoder_impl.cpp:32-52: out[i] ^= out[i + half] at every level (natural order, no bit reversal) */
  unsigned N = 1u << n;
  for (unsigned half = 1; half < N; half <<= 1)
    for (unsigned b = 0; b < N; b += 2 * half)
      for (unsigned i = 0; i < half; ++i)
        x[b + i] ^= x[b + i + half];
}

int orc_polar_encode_chain(unsigned K, unsigned E, unsigned nMax, int ibil, const uint8_t* msg, uint8_t* out, uint8_t* allocated_out,
                           uint8_t* encoded_out)
{
  orc_polar_code_t c;
  if (orc_polar_code_set(&c, K, E, nMax, ibil))
    return -1;
  unsigned N = c.N;
  uint8_t  u[1024];
  memset(u, 0, N);
  if (c.nPC == 0) { /* polar_allocator_impl.cpp:37-41 */
    unsigned i = 0;
    for (unsigned p = 0; p < N; ++p)
      if (c.K_set[p])
        u[p] = msg[i++];
  } else { /* :42-68 */
    unsigned y0 = 0, y1 = 0, y2 = 0, y3 = 0, y4 = 0, iPC = 0, iK = 0;
    for (unsigned p = 0; p < N; ++p) {
      unsigned t = y0;
      y0 = y1, y1 = y2, y2 = y3, y3 = y4, y4 = t;
      if (c.K_set[p]) {
        if (p == c.PC_set[iPC]) {
          ++iPC;
          u[p] = (uint8_t)y0;
        } else {
          u[p] = msg[iK];
          y0 ^= msg[iK];
          ++iK;
        }
      }
    }
  }
  if (allocated_out)
    memcpy(allocated_out, u, N);
  uint8_t d[1024];
  memcpy(d, u, N);
  polar_transform(d, c.n);
  if (encoded_out)
    memcpy(encoded_out, d, N);
  /* rate matching (polar_rate_matcher_impl.cpp:31-106) */
  uint8_t* y = (uint8_t*)malloc(8192 + 1024);
  for (unsigned j = 0; j < N; ++j)
    y[j] = d[c.blk_interleaver[j]];
  uint8_t* e = y;
  if (E >= N) {
    for (unsigned k2 = N; k2 < E; ++k2)
      y[k2] = y[k2 % N];
  } else if (16 * K <= 7 * E) {
    e = y + (N - E);
  }
  if (!ibil) {
    memcpy(out, e, E);
  } else {
    unsigned S = 1, T = 1;
    while (S < E) {
      ++T;
      S += T;
    }
    unsigned io = 0;
    for (unsigned r = 0; r < T; ++r) {
      unsigned ii = r;
      for (unsigned cc = 0; cc < T - r; ++cc) {
        if (ii < E) {
          out[io++] = e[ii];
          ii += T - cc;
        } else
          break;
      }
    }
  }
  free(y);
  return (int)N;
}

/* LLR algebra (log_likelihood_ratio.cpp:38-85, .h:208-216) */
static int8_t llr_add(int8_t a, int8_t b)
{ /* a + b: the reference evaluates `rhs += *this`, i.e. the special cases look at the right operand first */
  if (b == -a)
    return 0;
  if (b > LLR_MAX || b < -LLR_MAX)
    return b;
  if (a > LLR_MAX || a < -LLR_MAX)
    return a;
  int t = a + b;
  if (abs(t) > LLR_MAX)
    return (int8_t)(t > 0 ? LLR_MAX : -LLR_MAX);
  return (int8_t)t;
}
static int8_t llr_promotion_sum(int8_t a, int8_t b)
{
  if (a == -b)
    return 0;
  if (a > LLR_MAX || a < -LLR_MAX)
    return a;
  if (b > LLR_MAX || b < -LLR_MAX)
    return b;
  int t = a + b;
  if (abs(t) > LLR_MAX)
    return (int8_t)(t > 0 ? LLR_INF : -LLR_INF);
  return (int8_t)t;
}
static int8_t llr_soft_xor(int8_t x, int8_t y)
{
  int ax = abs(x), ay = abs(y), m = ax < ay ? ax : ay;
  return (int8_t)(((int)x * (int)y < 0) ? -m : m);
}

/* SSC decoder, recursive form of polar_decoder_impl.cpp:209-333. llr: 2^s values, u/est: outputs at offset pos. */
static void ssc_node(const uint8_t* frozen, const int8_t* llr, unsigned s, unsigned pos, uint8_t* u, uint8_t* est)
{
  unsigned size = 1u << s;
  int      any = 0, all = 1;
  for (unsigned i = 0; i < size; ++i) {
    any |= !frozen[pos + i];
    all &= !frozen[pos + i];
  }
  if (!any) /* rate 0 */
    return;
  if (all) { /* rate 1 (:227-257): hard decision, then re-encode the subtree to obtain the u-domain bits */
    for (unsigned i = 0; i < size; ++i)
      est[pos + i] = (uint8_t)(llr[i] <= 0);
    memcpy(u + pos, est + pos, size);
    if (s > 0)
      polar_transform(u + pos, s);
    return;
  }
  unsigned half = size / 2;
  int8_t   l0[512];
  for (unsigned i = 0; i < half; ++i)
    l0[i] = llr_soft_xor(llr[i], llr[i + half]);
  ssc_node(frozen, l0, s - 1, pos, u, est);
  for (unsigned i = 0; i < half; ++i) { /* g: y + x or y - x with the saturating operator (:32-61) */
    int8_t x = llr[i], y = llr[i + half];
    l0[i]    = est[pos + i] ? llr_add(y, (int8_t)-x) : llr_add(y, x);
  }
  ssc_node(frozen, l0, s - 1, pos + half, u, est);
  for (unsigned i = 0; i < half; ++i)
    est[pos + i] ^= est[pos + half + i];
}

static void polar_dematch(const orc_polar_code_t* c, const int8_t* llr, int8_t* d)
{ /* polar_rate_dematcher_impl.cpp:29-118 */
  unsigned N = c->N, E = c->E, K = c->K;
  int8_t*  buf = (int8_t*)calloc(8192 + 2048, 1);
  int8_t*  e   = buf + 1024;
  if (!c->ibil) {
    memcpy(e, llr, E);
  } else {
    unsigned S = 1, T = 1;
    while (S < E)
      S += ++T;
    unsigned io = 0;
    for (unsigned r = 0; r < T; ++r) {
      unsigned ii = r;
      for (unsigned cc = 0; cc < T - r; ++cc) {
        if (ii < E) {
          e[ii] = llr[io++];
          ii += T - cc;
        } else
          break;
      }
    }
  }
  int8_t* y = e;
  if (E >= N) {
    for (unsigned k2 = N; k2 < E; ++k2)
      y[k2 % N] = llr_promotion_sum(y[k2 % N], e[k2]);
  } else if (16 * K <= 7 * E) {
    y = e - (N - E);
    for (unsigned k2 = 0; k2 < N - E; ++k2)
      y[k2] = 0;
  } else {
    for (unsigned k2 = E; k2 < N; ++k2)
      y[k2] = LLR_INF;
  }
  for (unsigned j = 0; j < N; ++j)
    d[c->blk_interleaver[j]] = y[j];
  free(buf);
}

int orc_polar_decode_chain(unsigned K, unsigned E, unsigned nMax, int ibil, const int8_t* llr, uint8_t* msg, int8_t* dematched_out,
                           uint8_t* decoded_u_out)
{
  orc_polar_code_t c;
  if (orc_polar_code_set(&c, K, E, nMax, ibil))
    return -1;
  unsigned N = c.N;
  int8_t d[1024];
  polar_dematch(&c, llr, d);
  if (dematched_out)
    memcpy(dematched_out, d, N);
  uint8_t frozen[1024], u[1024], est[1024];
  for (unsigned i = 0; i < N; ++i)
    frozen[i] = !c.K_set[i];
  memset(u, 0, N);
  memset(est, 0, N);
  ssc_node(frozen, d, c.n, 0, u, est);
  if (decoded_u_out)
    memcpy(decoded_u_out, u, N);
  /* deallocation (polar_deallocator_impl.cpp:27-42) */
  unsigned iPC = 0, iK = 0;
  for (unsigned p = 0; p < N; ++p) {
    if (!c.K_set[p])
      continue;
    if (p == c.PC_set[iPC])
      ++iPC;
    else
      msg[iK++] = u[p];
  }
  return (int)N;
}

void orc_polar_interleave(const uint8_t* in, uint8_t* out, unsigned K, int rx)
{ /* polar_interleaver_impl.cpp:37-56 */
  unsigned k = 0;
  for (unsigned m = 0; m < NR_POLAR_K_MAX_IL; ++m) {
    if (NR_POLAR_PI_IL_MAX[m] >= NR_POLAR_K_MAX_IL - K) {
      unsigned pi_k = NR_POLAR_PI_IL_MAX[m] - (NR_POLAR_K_MAX_IL - K);
      if (!rx)
        out[k] = in[pi_k];
      else
        out[pi_k] = in[k];
      ++k;
    }
  }
}

int orc_pdcch_encode(const uint8_t* payload, unsigned A, unsigned rnti, unsigned E, uint8_t* out)
{ /* pdcch_encoder_impl.cpp:33-98 */
  unsigned K = A + 24;
  uint8_t  c[24 + 164 + 24], cp[164];
  memset(c, 1, 24);
  memcpy(c + 24, payload, A);
  uint32_t crc = orc_crc_bits(ORC_CRC24C, c, 24 + A);
  for (unsigned i = 0; i < 24; ++i)
    c[24 + A + i] = (uint8_t)((crc >> (23 - i)) & 1u);
  for (unsigned i = 0; i < 16; ++i)
    c[24 + A + 8 + i] ^= (uint8_t)((rnti >> (15 - i)) & 1u);
  orc_polar_interleave(c + 24, cp, K, 0);
  return orc_polar_encode_chain(K, E, 9, 0, cp, out, 0, 0) > 0 ? 0 : -1;
}

/* pbch_encoder_impl.cpp:41-190 */
int orc_pbch_encode(unsigned N_id, unsigned ssb_idx, unsigned L_max, int hrf, unsigned sfn, unsigned k_ssb, const uint8_t* payload, uint8_t* out)
{
  static const uint8_t G[32] = {16, 23, 18, 17, 8, 30, 10, 6, 24, 7, 0, 5, 3, 2, 1, 4, 9, 11, 12, 13, 14, 15, 19, 20, 21, 22, 25, 26, 27, 28, 29, 31};
  uint8_t  a[32] = {0}, ap[32], k[56], kp[56];
  unsigned j_sfn = 0, j_other = 14;
  for (unsigned i = 0; i < 24; ++i) {
    if (i >= 1 && i < 7)
      a[G[j_sfn++]] = payload[i];
    else
      a[G[j_other++]] = payload[i];
  }
  for (int b = 3; b >= 0; --b)
    a[G[j_sfn++]] = (uint8_t)((sfn >> b) & 1u);
  a[G[10]] = hrf ? 1 : 0;
  if (L_max == 64) {
    a[G[11]] = (ssb_idx >> 5) & 1u, a[G[12]] = (ssb_idx >> 4) & 1u, a[G[13]] = (ssb_idx >> 3) & 1u;
  } else {
    a[G[11]] = (k_ssb >> 4) & 1u, a[G[12]] = 0, a[G[13]] = 0;
  }
  unsigned M = (L_max == 64) ? 26 : 29;
  unsigned v = 2u * a[G[7]] + a[G[8]];
  uint8_t  c[32];
  orc_gold_sequence(N_id, M * v, 32, c);
  for (unsigned i = 0, j = 0; i < 32; ++i) {
    uint8_t s_i = c[j];
    int     ssb = (i == G[11] || i == G[12] || i == G[13]) && L_max == 64;
    if (ssb || i == G[10] || i == G[8] || i == G[7])
      s_i = 0;
    else
      ++j;
    ap[i] = a[i] ^ s_i;
  }
  memcpy(k, ap, 32);
  uint32_t crc = orc_crc_bits(ORC_CRC24C, ap, 32);
  for (unsigned i = 0; i < 24; ++i)
    k[32 + i] = (uint8_t)((crc >> (23 - i)) & 1u);
  orc_polar_interleave(k, kp, 56, 0);
  return orc_polar_encode_chain(56, 864, 9, 0, kp, out, 0, 0) > 0 ? 0 : -1;
}

/* ------------------------------------------------------------------------------------------------ SCL -- PARITY UNPINNED for L > 1 (the
 * reference has no list decoder). Written to the literature, not to the kernel: successive-cancellation LIST decoding (Tal, Vardy 2015) with
 * the LLR path metric (Balatsoukas-Stimming, Parizi, Burg 2015) in the reference's saturating int8 algebra:
 *   - at an information bit every live path forks into u = 0 and u = 1; the branch that agrees with the hard decision [llr <= 0] keeps its
 *     metric, the other one adds |llr|; at a frozen bit u = 0 and the metric grows by |llr| when llr < 0;
 *   - the L candidates with the smallest metrics survive, ties go to the lower candidate index (path p, hard-decision branch first);
 *   - at the end the surviving path with the smallest metric whose CRC checks is returned (crc_mode 0: the smallest metric).
 * L = 1 is plain successive cancellation and IS pinned: equal to orc_polar_sc_textbook below, and to the reference's SSC decoder except on
 * exact zero LLRs at information leaves (tests/test_polar_sc_pinning.py, tests/test_polar_gpu.py). The kernel (csrc/polar.hip) must
 * reproduce this function bit for bit. An aligned all-frozen block is handled at its own stage (the metric penalty is the sum of the negative
 * LLRs of that stage, identical to the leaf-by-leaf sum under the min-sum f and exact g). */
typedef struct {
  int8_t  llr[1024];  /* stage s at offset 2^s, size 2^s (s < n) */
  uint8_t bl[1024];   /* left partial sums of stage s at offset 2^s */
  uint8_t u[1024];
  int     pm;
} scl_path_t;

/* Textbook successive cancellation (Arikan 2009, LLR form with the min-sum f of the reference's algebra): the recursion over the
 * code tree with NO node pruning and NO list -- f = soft_xor on the upper branch, g = y +- x (saturating, infinities sticky:
 * log_likelihood_ratio.cpp:38-70) on the lower one, a leaf decides (llr <= 0) on an information / parity-check position and 0 on
 * a frozen one (log_likelihood_ratio.h:86). Written from the definition, independently of the kernels and of the reference's
 * simplified decoder: it is what list size 1 of the list decoder must reproduce, and it coincides with the reference's SSC
 * decoder (polar_decoder_impl.cpp:179-350, rate-1 nodes decided at once) whenever no information leaf sees an LLR of exactly
 * zero: a zero is a tie the two orders of evaluation break differently (SSC thresholds the node input, SC the leaf value).
 * *zero_seen reports such a tie. */
static void sc_textbook_rec(const int8_t* alpha, unsigned size, unsigned pos, const uint8_t* K_set, uint8_t* u, uint8_t* x, int* zero_seen)
{
  if (size == 1) {
    if (alpha[0] == 0 && K_set[pos])
      *zero_seen = 1; /* a tie at a decision (a zero anywhere in the input of a rate-1 node reaches one of its leaves as a zero) */
    u[pos] = K_set[pos] ? (uint8_t)(alpha[0] <= 0) : 0;
    x[0]   = u[pos];
    return;
  }
  const unsigned half = size / 2;
  int8_t         a[512];
  uint8_t        xl[512], xr[512];
  for (unsigned j = 0; j < half; ++j)
    a[j] = llr_soft_xor(alpha[j], alpha[j + half]);
  sc_textbook_rec(a, half, pos, K_set, u, xl, zero_seen);
  for (unsigned j = 0; j < half; ++j)
    a[j] = xl[j] ? llr_add(alpha[j + half], (int8_t)-alpha[j]) : llr_add(alpha[j + half], alpha[j]);
  sc_textbook_rec(a, half, pos + half, K_set, u, xr, zero_seen);
  for (unsigned j = 0; j < half; ++j) {
    x[j]        = xl[j] ^ xr[j];
    x[j + half] = xr[j];
  }
}

int orc_polar_sc_textbook(unsigned K, unsigned E, unsigned nMax, int ibil, const int8_t* llr, uint8_t* msg, int* zero_seen)
{
  orc_polar_code_t c;
  if (orc_polar_code_set(&c, K, E, nMax, ibil))
    return -1;
  int8_t  ch[1024];
  uint8_t u[1024], x[1024];
  polar_dematch(&c, llr, ch);
  *zero_seen = 0;
  sc_textbook_rec(ch, c.N, 0, c.K_set, u, x, zero_seen);
  unsigned iPC = 0, iK = 0;
  for (unsigned q = 0; q < c.N; ++q) { /* polar_deallocator_impl.cpp:27-42 */
    if (!c.K_set[q])
      continue;
    if (q == c.PC_set[iPC])
      ++iPC;
    else
      msg[iK++] = u[q];
  }
  return (int)c.N;
}

int orc_polar_scl_decode(unsigned K, unsigned E, unsigned nMax, int ibil, unsigned L, int crc_mode, unsigned rnti, const int8_t* llr,
                         uint8_t* msg, int* crc_ok)
{
  orc_polar_code_t c;
  if (orc_polar_code_set(&c, K, E, nMax, ibil) || (L != 1 && L != 2 && L != 4 && L != 8))
    return -1;
  unsigned N = c.N, n = c.n;
  int8_t   ch[1024];
  polar_dematch(&c, llr, ch);
  scl_path_t* P = (scl_path_t*)calloc(2 * L, sizeof(scl_path_t));
  scl_path_t* Q = P + L;
  unsigned    active = 1;
  for (unsigned i = 0; i < N;) {
    /* Largest aligned all-frozen block [i, i + 2^r) (a "rate-0 node"): it is processed at stage r in one step -- its path metric
     * penalty is the sum of the negative stage-r LLRs (identical to the leaf-by-leaf sum under min-sum f and exact g). */
    unsigned r = 0;
    if (!c.K_set[i]) {
      while (r < n && (i & ((2u << r) - 1u)) == 0) {
        int frozen = 1;
        for (unsigned j = 0; j < (2u << r) && frozen; ++j)
          frozen = !c.K_set[i + j];
        if (!frozen)
          break;
        ++r;
      }
    }
    const unsigned B = 1u << r;
    /* stage-r LLRs of every active path */
    for (unsigned p = 0; p < active; ++p) {
      scl_path_t* a = &P[p];
      unsigned    t = n;
      if (i != 0) {
        t = 0;
        while (!((i >> t) & 1u))
          ++t;
        const int8_t* up = (t + 1 == n) ? ch : a->llr + (2u << t);
        for (unsigned j = 0; j < (1u << t); ++j) {
          int8_t x = up[j], y = up[j + (1u << t)];
          a->llr[(1u << t) + j] = a->bl[(1u << t) + j] ? llr_add(y, (int8_t)-x) : llr_add(y, x);
        }
      }
      for (int s = (int)t - 1; s >= (int)r; --s) {
        const int8_t* up = ((unsigned)s + 1 == n) ? ch : a->llr + (2u << s);
        for (unsigned j = 0; j < (1u << s); ++j)
          a->llr[(1u << s) + j] = llr_soft_xor(up[j], up[j + (1u << s)]);
      }
    }
    if (!c.K_set[i]) {
      for (unsigned p = 0; p < active; ++p) {
        const int8_t* v = (r == n) ? ch : P[p].llr + B;
        for (unsigned j = 0; j < B; ++j) {
          P[p].u[i + j] = 0;
          if (v[j] < 0)
            P[p].pm += -v[j];
        }
      }
    } else {
      unsigned nc = 2 * active, keep = nc < L ? nc : L;
      int      met[16];
      uint8_t  bit[16];
      for (unsigned p = 0; p < active; ++p) {
        int l0 = P[p].llr[1], hard = l0 <= 0, al = l0 < 0 ? -l0 : l0;
        met[2 * p] = P[p].pm, bit[2 * p] = (uint8_t)hard;
        met[2 * p + 1] = P[p].pm + al, bit[2 * p + 1] = (uint8_t)!hard;
      }
      for (unsigned cnd = 0; cnd < nc; ++cnd) {
        unsigned rank = 0;
        for (unsigned o = 0; o < nc; ++o)
          rank += (met[o] < met[cnd]) || (met[o] == met[cnd] && o < cnd);
        if (rank < keep) {
          Q[rank]      = P[cnd >> 1];
          Q[rank].u[i] = bit[cnd];
          Q[rank].pm   = met[cnd];
        }
      }
      scl_path_t* tmp = P;
      P = Q, Q = tmp;
      active = keep;
    }
    /* partial sums of the finished block (stage r) */
    for (unsigned p = 0; p < active; ++p) {
      scl_path_t* a = &P[p];
      if (r == n)
        break;
      if (!((i >> r) & 1u)) {
        for (unsigned j = 0; j < B; ++j)
          a->bl[B + j] = a->u[i + j];
      } else {
        uint8_t  cur[1024];
        unsigned sz = B, s = r;
        for (unsigned j = 0; j < B; ++j)
          cur[j] = a->u[i + j];
        while (s < n && ((i >> s) & 1u)) {
          for (unsigned j = 0; j < sz; ++j) {
            cur[sz + j] = cur[j];
            cur[j] ^= a->bl[(1u << s) + j];
          }
          sz <<= 1, ++s;
        }
        if (s < n)
          memcpy(a->bl + (1u << s), cur, sz);
      }
    }
    i += B;
  }
  /* selection */
  int best = -1, best_ok = -1;
  uint8_t cand[8][1024];
  for (unsigned p = 0; p < active; ++p) {
    uint8_t  kb[1024];
    unsigned iPC = 0, iK = 0;
    for (unsigned q = 0; q < N; ++q) {
      if (!c.K_set[q])
        continue;
      if (q == c.PC_set[iPC])
        ++iPC;
      else
        kb[iK++] = P[p].u[q];
    }
    int ok = 0;
    if (crc_mode == 0) {
      memcpy(cand[p], kb, K);
    } else {
      orc_polar_interleave(kb, cand[p], K, 1);
      unsigned A = K - 24;
      uint8_t  tmpb[24 + 1024];
      unsigned ones = (crc_mode == 1) ? 24 : 0;
      memset(tmpb, 1, ones);
      memcpy(tmpb + ones, cand[p], A);
      uint32_t crc = orc_crc_bits(ORC_CRC24C, tmpb, ones + A);
      uint32_t rx  = 0;
      for (unsigned b = 0; b < 24; ++b)
        rx = (rx << 1) | cand[p][A + b];
      if (crc_mode == 1)
        rx ^= (rnti & 0xffffu);
      ok = (crc == rx);
    }
    if (best < 0 || P[p].pm < P[best].pm)
      best = (int)p;
    if (ok && (best_ok < 0 || P[p].pm < P[best_ok].pm))
      best_ok = (int)p;
  }
  int sel = (best_ok >= 0) ? best_ok : best;
  memcpy(msg, cand[sel], K);
  *crc_ok = best_ok >= 0;
  int pm = P[sel].pm;
  free(P < Q ? P : Q);
  return pm;
}

/* ================================================================================================= PUSCH demodulator
 * pusch_demodulator_impl.cpp:31-152 (RE extraction pusch_demodulator_impl.h:74-172, equalise, soft-demap, descramble),
 * channel_equalizer_zf_impl.cpp:123-162 + equalize_zf_1xn.h:42-158 (one transmit layer, any number of ports),
 * demodulation_mapper_impl.cpp:34-106 and demodulation_mapper_{qpsk,qam16,qam64,qam256}.cpp (AVX2 arithmetic: reciprocal noise by
 * an exact division, interval index = floor(x * (1/width)), (slope*x + intercept) * rcp_noise, quantisation = scale, clip to
 * +-120, round to nearest even: avx2_helpers.h:103-157,164-236), descrambling with c_init = rnti * 2^15 + n_id.
 * Floating point: every operation is a single correctly rounded IEEE operation in the order written here (no contraction:
 * this file is compiled for baseline x86-64 without FMA); the reference's AVX2 equaliser uses the approximate _mm256_rcp_ps and
 * its build may contract a*b+c, so reference LLRs can differ by one quantisation step (stated tolerance: +-1, see the tests). */
#include "../srsran_project_23.5_amd/csrc/tables/nr_demod_tables.h"

static int8_t demod_quantize(float v, float range_limit)
{
  float s = v * (120.0f / range_limit);
  if (s > 120.0f)
    s = 120.0f;
  if (s < -120.0f)
    s = -120.0f;
  float r = rintf(s); /* round to nearest even */
  if (!(r <= 120.0f && r >= -120.0f))
    return 0; /* NaN */
  return (int8_t)r;
}

static float demod_interval(float x, float rcp_noise, float rcp_width, int count, const float* slope, const float* intercept)
{
  int idx = (int)floorf(x * rcp_width) + count / 2;
  idx     = idx < 0 ? 0 : (idx > count - 1 ? count - 1 : idx);
  float t = slope[idx] * x;
  t       = t + intercept[idx];
  return t * rcp_noise;
}

/* One equalised symbol -> mod LLRs (before descrambling). sym_idx only matters for pi/2-BPSK. */
static void demod_symbol(int mod, float re, float im, float nvar, unsigned sym_idx, int8_t* out)
{
  if (mod == 1) { /* pi/2-BPSK, demodulation_mapper_impl.cpp:34-81 */
    float l = 0.f;
    if (nvar > 0) {
      float a = (sym_idx & 1u) ? im : re, b = (sym_idx & 1u) ? -re : im;
      l       = 2.0f * (float)M_SQRT2 * (a + b) / nvar;
    }
    /* scalar quantiser of the reference for this modulation (log_likelihood_ratio.cpp:87-96) */
    float c = l;
    if (fabsf(l) > 24.f)
      c = copysignf(24.f, l);
    out[0] = (nvar > 0) ? (int8_t)roundf(c / 24.f * 120.f) : 0;
    return;
  }
  float rcp = (nvar > 0) ? 1.0f / nvar : 0.0f;
  float x[2] = {re, im};
  if (mod == 2) {
    for (int d = 0; d < 2; ++d)
      out[d] = demod_quantize((NR_DEMOD_QPSK_GAIN * x[d]) * rcp, 24.f);
  } else if (mod == 4) {
    for (int d = 0; d < 2; ++d) {
      float first  = NR_DEMOD_QAM16_GAIN * x[d];
      float second = 2.0f * first - copysignf(0.8f, x[d]);
      float l01    = (fabsf(x[d]) > NR_DEMOD_QAM16_THRESHOLD) ? second : first;
      float l23    = 0.8f - fabsf(first);
      out[d]       = demod_quantize(l01 * rcp, 20.f);
      out[2 + d]   = demod_quantize(l23 * rcp, 20.f);
    }
  } else if (mod == 6) {
    for (int d = 0; d < 2; ++d) {
      out[d]     = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM64_B0_RCP_WIDTH, NR_DEMOD_QAM64_B0_COUNT, NR_DEMOD_QAM64_B0_SLOPE, NR_DEMOD_QAM64_B0_INTERCEPT), 20.f);
      out[2 + d] = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM64_B1_RCP_WIDTH, NR_DEMOD_QAM64_B1_COUNT, NR_DEMOD_QAM64_B1_SLOPE, NR_DEMOD_QAM64_B1_INTERCEPT), 20.f);
      out[4 + d] = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM64_B2_RCP_WIDTH, NR_DEMOD_QAM64_B2_COUNT, NR_DEMOD_QAM64_B2_SLOPE, NR_DEMOD_QAM64_B2_INTERCEPT), 20.f);
    }
  } else {
    for (int d = 0; d < 2; ++d) {
      out[d]     = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM256_B0_RCP_WIDTH, NR_DEMOD_QAM256_B0_COUNT, NR_DEMOD_QAM256_B0_SLOPE, NR_DEMOD_QAM256_B0_INTERCEPT), 20.f);
      out[2 + d] = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM256_B1_RCP_WIDTH, NR_DEMOD_QAM256_B1_COUNT, NR_DEMOD_QAM256_B1_SLOPE, NR_DEMOD_QAM256_B1_INTERCEPT), 20.f);
      out[4 + d] = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM256_B2_RCP_WIDTH, NR_DEMOD_QAM256_B2_COUNT, NR_DEMOD_QAM256_B2_SLOPE, NR_DEMOD_QAM256_B2_INTERCEPT), 20.f);
      out[6 + d] = demod_quantize(demod_interval(x[d], rcp, NR_DEMOD_QAM256_B3_RCP_WIDTH, NR_DEMOD_QAM256_B3_COUNT, NR_DEMOD_QAM256_B3_SLOPE, NR_DEMOD_QAM256_B3_INTERCEPT), 20.f);
    }
  }
}

void orc_demodulate_soft(int mod, unsigned nsym, const float* symbols, const float* noise_vars, int8_t* llr)
{
  for (unsigned i = 0; i < nsym; ++i)
    demod_symbol(mod, symbols[2 * i], symbols[2 * i + 1], noise_vars[i], i, llr + (size_t)i * (unsigned)mod);
}

/* 12-bit mask of the resource elements of a PRB that carry DM-RS (dmrs_mapping.h:76-92). */
static unsigned dmrs_prb_mask(int type2, unsigned nof_cdm_groups_without_data)
{
  unsigned m = 0;
  for (unsigned k = 0; k < 12; ++k) {
    unsigned set = !type2 ? ((k % 2) < nof_cdm_groups_without_data) : ((k % 6) < 2 * nof_cdm_groups_without_data);
    m |= set << k;
  }
  return m;
}

static void mod_symbol(int mod, const uint8_t* b, unsigned sym_idx, float* re, float* im);

/* placeholders: sorted resource-element indices carrying a repetition placeholder (ulsch_placeholder_list), nof_ph of them;
 * evm_out: error vector magnitude of pusch_demodulator_impl.cpp:89-90 (NULL: not computed). */
int orc_pusch_demodulate_ex(unsigned rnti, unsigned n_id, int mod, unsigned start_symbol, unsigned nof_symbols, const uint8_t* dmrs_symbols_mask,
                            int dmrs_type2, unsigned nof_cdm_groups_without_data, const uint8_t* rb_mask, unsigned nof_prb_grid,
                            unsigned nof_rx_ports, const float* grid, const float* ce, unsigned ce_nof_symbols, float noise_var, int8_t* llr_out,
                            float* eq_out, float* nvar_out, const uint16_t* placeholders, unsigned nof_ph, float* evm_out)
{
  float* eq_own = NULL;
  if (evm_out && !eq_out)
    eq_out = eq_own = (float*)malloc(sizeof(float) * 2 * (size_t)nof_prb_grid * 12 * 14);
  const unsigned nsc   = nof_prb_grid * 12;
  const unsigned dmask = dmrs_prb_mask(dmrs_type2, nof_cdm_groups_without_data);
  unsigned       n     = 0;
  for (unsigned sy = start_symbol; sy < start_symbol + nof_symbols; ++sy) {
    for (unsigned rb = 0; rb < nof_prb_grid; ++rb) {
      if (!rb_mask[rb])
        continue;
      for (unsigned k = 0; k < 12; ++k) {
        if (dmrs_symbols_mask[sy] && ((dmask >> k) & 1u))
          continue;
        const unsigned sc = rb * 12 + k;
        /* equalize_zf_1xn.h:120-158 */
        float ch_mod_sq = 0.f, acc_re = 0.f, acc_im = 0.f;
        for (unsigned p = 0; p < nof_rx_ports; ++p) {
          const float* y = grid + 2 * ((size_t)(p * 14 + sy) * nsc + sc);
          const float* h = ce + 2 * ((size_t)(p * ce_nof_symbols + sy) * nsc + sc);
          float        t = h[0] * h[0];
          float        u = h[1] * h[1];
          ch_mod_sq      = ch_mod_sq + (t + u);
          /* y * conj(h) */
          float a = y[0] * h[0], b = y[1] * h[1], c = y[1] * h[0], d = y[0] * h[1];
          acc_re  = acc_re + (a + b);
          acc_im  = acc_im + (c - d);
        }
        const float d_pinv = 1.0f * ch_mod_sq;
        float       z_re = 0.f, z_im = 0.f, nv = INFINITY;
        const float rcpd = 1.0f / d_pinv;
        const float v    = rcpd * (noise_var / 1.0f);
        if (d_pinv > 0.f && d_pinv < INFINITY && v > 0.f && v < INFINITY) {
          z_re = acc_re * rcpd;
          z_im = acc_im * rcpd;
          nv   = v;
        }
        if (eq_out) {
          eq_out[2 * n]     = z_re;
          eq_out[2 * n + 1] = z_im;
        }
        if (nvar_out)
          nvar_out[n] = nv;
        demod_symbol(mod, z_re, z_im, nv, n, llr_out + (size_t)n * (unsigned)mod);
        ++n;
      }
    }
  }
  const unsigned nbits = n * (unsigned)mod;
  if (evm_out) { /* evm_calculator_generic_impl.cpp:31-47: hard decision of the soft bits (before descrambling), modulation, error power */
    float acc = 0.f;
    for (unsigned i = 0; i < n; ++i) {
      uint8_t hb[8];
      float   re, im;
      for (int b = 0; b < mod; ++b)
        hb[b] = llr_out[(size_t)i * (unsigned)mod + (unsigned)b] <= 0;
      mod_symbol(mod, hb, i, &re, &im);
      const float er = re - eq_out[2 * i], ei = im - eq_out[2 * i + 1];
      acc += er * er + ei * ei;
    }
    *evm_out = n ? sqrtf(acc / (float)n) : 0.f;
  }
  free(eq_own);
  /* descrambling (pusch_demodulator_impl.cpp:99-152). In an element with a repetition placeholder the first soft bit takes its own
   * chip, the second one the SAME chip (y repeats the bit before it) and the x placeholders behind them are left as they are;
   * the sequence advances over all of them. */
  uint8_t* c = (uint8_t*)malloc(nbits ? nbits : 1);
  orc_gold_sequence((rnti << 15) + n_id, 0, nbits, c);
  unsigned ip = 0;
  for (unsigned re = 0; re < n; ++re) {
    while (ip < nof_ph && placeholders[ip] < re)
      ++ip;
    const int ph = mod >= 2 && ip < nof_ph && placeholders[ip] == re;
    for (int b = 0; b < mod; ++b) {
      const size_t i    = (size_t)re * (unsigned)mod + (unsigned)b;
      const int    chip = ph ? (b < 2 ? c[(size_t)re * (unsigned)mod] : 0) : c[i];
      if (chip)
        llr_out[i] = (int8_t)-llr_out[i];
    }
  }
  free(c);
  return (int)nbits;
}

int orc_pusch_demodulate(unsigned rnti, unsigned n_id, int mod, unsigned start_symbol, unsigned nof_symbols, const uint8_t* dmrs_symbols_mask,
                         int dmrs_type2, unsigned nof_cdm_groups_without_data, const uint8_t* rb_mask, unsigned nof_prb_grid,
                         unsigned nof_rx_ports, const float* grid, const float* ce, unsigned ce_nof_symbols, float noise_var, int8_t* llr_out,
                         float* eq_out, float* nvar_out)
{
  return orc_pusch_demodulate_ex(rnti, n_id, mod, start_symbol, nof_symbols, dmrs_symbols_mask, dmrs_type2, nof_cdm_groups_without_data, rb_mask,
                                 nof_prb_grid, nof_rx_ports, grid, ce, ce_nof_symbols, noise_var, llr_out, eq_out, nvar_out, NULL, 0, NULL);
}

/* ================================================================================================= PDSCH modulator + DM-RS
 * modulation_mapper_impl.cpp:31-146 (constellation = integer level * sqrtf(1 / average power), TS 38.211 5.1),
 * pdsch_modulator_impl.cpp:30-282 (scramble with c_init = rnti*2^15 + q*2^14 + n_id, modulate, optional scaling, codeword-to-layer
 * mapping with the reference's rule "two codewords from four layers on", mapping to the allocated PRBs in the given order skipping
 * the DM-RS pattern of the bandwidth part and the reserved RE patterns), dmrs_pdsch_processor_impl.cpp:30-169 + dmrs_helper.h:44-96. */
static float mod_scale(int mod)
{
  /* average power of the integer constellation: 2 (QPSK), 10, 42, 170 -> scaling = sqrt(1 / avg) in single precision */
  const float avg = (mod == 2) ? 2.f : (mod == 4) ? 10.f : (mod == 6) ? 42.f : 170.f;
  return sqrtf(1.0f / avg);
}

/* bits: one bit per byte, first bit = most significant. */
static void mod_symbol(int mod, const uint8_t* b, unsigned sym_idx, float* re, float* im)
{
  if (mod == 1) { /* pi/2-BPSK: even (s, s), odd (-s, s) with s = +-1/sqrt(2) */
    const float v = (float)M_SQRT1_2;
    const float s = (b[0] & 1u) ? -v : v;
    *re           = (sym_idx & 1u) ? -s : s;
    *im           = s;
    return;
  }
  int lr = 0, li = 0;
  const int h = mod / 2;
  /* x = (1-2b0)[2^(h-1) - (1-2b2)[2^(h-2) - ...]], real part from the even bits, imaginary part from the odd ones */
  for (int j = h - 1; j >= 0; --j) {
    const int sr = 1 - 2 * (int)(b[2 * j] & 1u), si = 1 - 2 * (int)(b[2 * j + 1] & 1u);
    const int w  = 1 << (h - 1 - j);
    lr           = sr * (w - lr);
    li           = si * (w - li);
  }
  const float sc = mod_scale(mod);
  *re            = (float)lr * sc;
  *im            = (float)li * sc;
}

void orc_modulate(int mod, unsigned nsym, const uint8_t* bits, float* symbols)
{
  for (unsigned i = 0; i < nsym; ++i)
    mod_symbol(mod, bits + (size_t)i * (unsigned)mod, i, symbols + 2 * i, symbols + 2 * i + 1);
}

int orc_pdsch_modulate(unsigned rnti, unsigned n_id, float scaling, unsigned nof_layers, const int* mod, const uint8_t* cw0, unsigned nbits0,
                       const uint8_t* cw1, unsigned nbits1, unsigned start_symbol, unsigned nof_symbols, const uint8_t* dmrs_symbols_mask,
                       int dmrs_type2, unsigned nof_cdm_groups_without_data, unsigned bwp_start_rb, unsigned bwp_size_rb, const uint16_t* prb_list,
                       unsigned nof_prb, unsigned nof_reserved, const uint8_t* res_prb_mask, const uint16_t* res_re_mask,
                       const uint16_t* res_symbols, const uint8_t* ports, unsigned nof_prb_grid, float* grid)
{
  const unsigned nsc      = nof_prb_grid * 12;
  const unsigned nof_cw   = (nof_layers >= 4) ? 2 : 1;
  const unsigned lcw[2]   = {nof_layers / nof_cw, nof_layers - nof_layers / nof_cw};
  const uint8_t* cw[2]    = {cw0, cw1};
  const unsigned nbits[2] = {nbits0, nbits1};
  const unsigned dmask    = dmrs_prb_mask(dmrs_type2, nof_cdm_groups_without_data);
  uint8_t*       c[2]     = {NULL, NULL};
  for (unsigned q = 0; q < nof_cw; ++q) {
    c[q] = (uint8_t*)malloc(nbits[q] ? nbits[q] : 1);
    orc_gold_sequence((rnti << 15) + (q << 14) + n_id, 0, nbits[q], c[q]);
  }
  unsigned i_re = 0;
  int      rc   = 0;
  for (unsigned sy = start_symbol; sy < start_symbol + nof_symbols; ++sy) {
    for (unsigned pi = 0; pi < nof_prb; ++pi) {
      const unsigned rb = prb_list[pi];
      for (unsigned k = 0; k < 12; ++k) {
        int excluded = dmrs_symbols_mask[sy] && ((dmask >> k) & 1u) && rb >= bwp_start_rb && rb < bwp_start_rb + bwp_size_rb;
        for (unsigned r = 0; r < nof_reserved; ++r)
          excluded |= res_prb_mask[(size_t)r * nof_prb_grid + rb] && ((res_re_mask[r] >> k) & 1u) && ((res_symbols[r] >> sy) & 1u);
        if (excluded)
          continue;
        for (unsigned ly = 0; ly < nof_layers; ++ly) {
          const unsigned q = (ly < lcw[0]) ? 0 : 1;
          const unsigned d = (q == 0) ? lcw[0] * i_re + ly : lcw[1] * i_re + (ly - lcw[0]);
          const unsigned m = (unsigned)mod[q];
          if ((size_t)(d + 1) * m > nbits[q]) {
            rc = -1;
            continue;
          }
          uint8_t b[8];
          for (unsigned t = 0; t < m; ++t)
            b[t] = (cw[q][(size_t)d * m + t] ^ c[q][(size_t)d * m + t]) & 1u;
          float re, im;
          mod_symbol((int)m, b, d, &re, &im);
          if (isnormal(scaling)) {
            re = re * scaling;
            im = im * scaling;
          }
          float* o = grid + 2 * (((size_t)ports[ly] * 14 + sy) * nsc + rb * 12 + k);
          o[0]     = re;
          o[1]     = im;
        }
        ++i_re;
      }
    }
  }
  for (unsigned q = 0; q < nof_cw; ++q) {
    if ((size_t)i_re * lcw[q] * (unsigned)mod[q] != nbits[q])
      rc = -1; /* pdsch_modulator_impl.cpp:222-225: every element of every layer must be mapped */
    free(c[q]);
  }
  return rc ? rc : (int)i_re;
}

int orc_dmrs_pdsch_map(unsigned slot_in_frame, unsigned reference_point_k_rb, int type2, unsigned scrambling_id, int n_scid, float amplitude,
                       const uint8_t* symbols_mask, const uint8_t* rb_mask, unsigned nof_prb_grid, unsigned nof_ports, const uint8_t* ports, float* grid)
{
  const unsigned nsc = nof_prb_grid * 12, npr = type2 ? 4 : 6;
  const float    amp = (float)(M_SQRT1_2 * (double)amplitude);
  uint8_t*       c   = (uint8_t*)malloc(2 * npr * nof_prb_grid + 1);
  for (unsigned sy = 0; sy < 14; ++sy) {
    if (!symbols_mask[sy])
      continue;
    const unsigned long long t = ((unsigned long long)(14 * slot_in_frame + sy + 1) * (2ull * scrambling_id + 1)) % (1ull << 31);
    const unsigned c_init      = (unsigned)((t * (1ull << 17) + (2ull * scrambling_id + (n_scid ? 1 : 0))) % (1ull << 31));
    orc_gold_sequence(c_init, 0, 2 * npr * nof_prb_grid, c);
    const unsigned l_prime = (sy != 0 && symbols_mask[sy - 1]) ? 1 : 0;
    for (unsigned p = 0; p < nof_ports; ++p) {
      const unsigned delta = !type2 ? (p / 2) % 2 : 2 * ((p / 2) % 3);
      const float    wf1   = (p % 2) ? -1.f : 1.f;
      const float    wt    = (l_prime && p >= (type2 ? 6u : 4u)) ? -1.f : 1.f;
      unsigned       i     = 0; /* index in the generated sequence (allocated PRBs only) */
      for (unsigned rb = reference_point_k_rb; rb < nof_prb_grid; ++rb) {
        if (!rb_mask[rb])
          continue;
        for (unsigned q = 0; q < npr; ++q, ++i) {
          const unsigned g  = (rb - reference_point_k_rb) * npr + q; /* sequence position counted from the reference point */
          const unsigned k  = !type2 ? 2 * q : (q < 2 ? q : 4 + q);     /* type 1: 0,2,..,10 ; type 2: 0,1,6,7 */
          float          re = c[2 * g] ? -amp : amp, im = c[2 * g + 1] ? -amp : amp;
          const float    w  = wt * ((i & 1u) ? wf1 : 1.f);
          re                = re * w;
          im                = im * w;
          float* o          = grid + 2 * (((size_t)ports[p] * 14 + sy) * nsc + rb * 12 + k + delta);
          o[0]              = re;
          o[1]              = im;
        }
      }
    }
  }
  free(c);
  return 0;
}

/* ================================================================================================ Open Fronthaul BFP compression
 * lib/ofh/compression/iq_compression_bfp_impl.cpp, compressed_prb.cpp, quantizer.h; include/srsran/ofh/compression/compression_params.h
 * (Q_BIT_WIDTH = MAX_IQ_WIDTH = 16). */
static unsigned ofh_extract_bits(const uint8_t* data, unsigned pos, unsigned length) /* compressed_prb::extract_bits (:63-79), MSB first */
{
  unsigned v = 0;
  for (unsigned i = 0; i < length; ++i, ++pos)
    v = (v << 1) | ((data[pos >> 3] >> (7 - (pos & 7))) & 1u);
  return v;
}

void orc_ofh_iq_decompress(int compression, const uint8_t* payload, unsigned nof_prb, unsigned w, int simd_arithmetic, float* out)
{
  const int      bfp  = compression == 1;
  const unsigned hdr  = bfp ? 1u : 0u;
  const float    gain = bfp ? 32767.0f : (float)(1 << (w - 1)) - 1.0f; /* quantizer(bit_width): (1 << (bit_width - 1)) - 1 */
  for (unsigned p = 0; p < nof_prb; ++p) {
    const uint8_t* rec      = payload + (size_t)p * (hdr + 3 * w);
    const unsigned exponent = bfp ? rec[0] : 0;
    const int16_t  scaler   = (int16_t)(1 << exponent);
    for (unsigned i = 0; i < 24; ++i) {
      /* quantizer::sign_extend (quantizer.h:88-92) */
      int16_t v = (int16_t)ofh_extract_bits(rec + hdr, i * w, w);
      v         = (int16_t)((int16_t)(v << (16 - w)) >> (16 - w));
      float f;
      if (bfp && simd_arithmetic && w == 9) { /* quantizer::to_float(span) -> srsvec::convert(int16 -> float) */
        const float scale = gain / scaler;
        const float g     = 1.0f / scale;
        f                 = (float)v * g;
      } else { /* quantizer::to_float(int) (quantizer.h:70) */
        f = (float)((int)v * (int)scaler) / gain;
      }
      out[(size_t)p * 24 + i] = f;
    }
  }
}

static unsigned ofh_determine_exponent(unsigned x, unsigned w) /* iq_compression_bfp_impl::determine_exponent (:28-41), x < 65536 */
{
  const unsigned max_shift = 16 - w;
  unsigned       lz        = max_shift;
  if (x > 0 && max_shift > 0) {
    unsigned clz16 = 0;
    while (!((x << clz16) & 0x8000u))
      ++clz16;
    lz = clz16 - 1;
  }
  const int raw = (int)(max_shift < lz ? max_shift : lz);
  const int e   = (int)(16 - w) - raw;
  return e > 0 ? (unsigned)e : 0;
}

void orc_ofh_iq_compress(int compression, const float* in, unsigned nof_prb, unsigned w, float iq_scaling, uint8_t* payload)
{
  const int      bfp   = compression == 1;
  const unsigned hdr   = bfp ? 1u : 0u;
  const float    scale = (bfp ? 32767.0f : (float)(1 << (w - 1)) - 1.0f) * iq_scaling; /* quantizer::to_fixed_point(span): gain * in_scale */
  const unsigned len   = 24 * nof_prb;
  int16_t*       q     = (int16_t*)malloc(sizeof(int16_t) * (len ? len : 1));
  for (unsigned i = 0; i < len; ++i) {
    /* srsvec::convert_round over the whole call (BFP) or over one PRB at a time (none): SIMD part, then scalar tail */
    const unsigned pos = bfp ? i : i % 24, span_len = bfp ? len : 24, simd_len = (span_len / 16) * 16;
    const float    a   = in[i] * scale;
    if (pos < simd_len) { /* _mm256_round_ps(nearest) + cvtps_epi32 + packs_epi32 */
      const float r = rintf(a);
      long        v;
      if (!(r > -2147483904.0f && r < 2147483648.0f))
        v = -2147483647L - 1; /* cvtps_epi32 integer indefinite (also NaN) */
      else
        v = (long)r;
      q[i] = (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
    } else { /* static_cast<int16_t>(std::round(a)): what x86-64 makes of it -- cvttss2si to 32 bits (integer indefinite when out
                of range), then the low 16 bits, so out-of-range values wrap here while they saturate in the SIMD part */
      const float r = roundf(a);
      const long  v = (r > -2147483904.0f && r < 2147483648.0f) ? (long)r : -2147483647L - 1;
      q[i]          = (int16_t)(uint16_t)((unsigned long)v & 0xffffu);
    }
  }
  for (unsigned p = 0; p < nof_prb; ++p) {
    const int16_t* x        = q + p * 24;
    unsigned       exponent = 0;
    if (bfp) {
      int mx = x[0], mn = x[0];
      for (unsigned i = 1; i < 24; ++i) {
        mx = x[i] > mx ? x[i] : mx;
        mn = x[i] < mn ? x[i] : mn;
      }
      const int      a = abs(mx), b = abs(mn) - 1;
      const unsigned max_abs = (unsigned)(a > b ? a : b);
      exponent               = ofh_determine_exponent(max_abs & 0xffffu, w);
    }
    uint8_t* rec = payload + (size_t)p * (hdr + 3 * w);
    memset(rec, 0, hdr + 3 * w);
    if (bfp)
      rec[0] = (uint8_t)exponent;
    for (unsigned i = 0; i < 24; ++i) { /* compressed_prb::pack_compressed_data (:31-61): the low w bits of each sample, MSB first */
      const unsigned v = (unsigned)(uint16_t)(int16_t)(x[i] >> exponent);
      for (unsigned b2 = 0; b2 < w; ++b2) {
        const unsigned pos = i * w + b2;
        rec[hdr + (pos >> 3)] |= (uint8_t)(((v >> (w - 1 - b2)) & 1u) << (7 - (pos & 7)));
      }
    }
  }
  free(q);
}

/* ================================================================================================ PDCCH processor */
int orc_pdcch_process(unsigned slot_in_frame, unsigned rnti, unsigned n_id_data, unsigned n_rnti, unsigned n_id_dmrs, unsigned reference_point_k_rb,
                      float data_power_offset_dB, float dmrs_power_offset_dB, const uint8_t* payload, unsigned A, unsigned aggregation_level,
                      unsigned start_symbol, unsigned duration, const uint8_t* rb_mask, unsigned nof_prb_grid, float* grid)
{
  static const unsigned data_re[9] = {0, 2, 3, 4, 6, 7, 8, 10, 11};
  const unsigned        E = 108 * aggregation_level, nsc = nof_prb_grid * 12;
  unsigned              nprb = 0;
  for (unsigned rb = 0; rb < nof_prb_grid; ++rb)
    nprb += rb_mask[rb] ? 1 : 0;
  if (nprb * 9 * duration * 2 != E)
    return -1;
  uint8_t* enc = (uint8_t*)malloc(E);
  uint8_t* c   = (uint8_t*)malloc(E > 6 * nof_prb_grid ? E : 6 * nof_prb_grid);
  float*   sym = (float*)malloc(sizeof(float) * E);
  if (orc_pdcch_encode(payload, A, rnti, E, enc) < 0) {
    free(enc), free(c), free(sym);
    return -1;
  }
  /* pdcch_modulator_impl::scramble (:30-40), modulate (:42-56), map (:58-73) */
  orc_gold_sequence(((n_rnti << 16) + n_id_data) % (1u << 31), 0, E, c);
  for (unsigned i = 0; i < E; ++i)
    enc[i] = (enc[i] ^ c[i]) & 1u;
  orc_modulate(2, E / 2, enc, sym);
  const float scaling = powf(10.0f, data_power_offset_dB / 20.0f); /* convert_dB_to_amplitude (math_utils.h:101-104) */
  if (isnormal(scaling))
    for (unsigned i = 0; i < E; ++i)
      sym[i] = sym[i] * scaling;
  unsigned i = 0;
  for (unsigned sy = start_symbol; sy < start_symbol + duration; ++sy)
    for (unsigned rb = 0; rb < nof_prb_grid; ++rb)
      if (rb_mask[rb])
        for (unsigned q = 0; q < 9; ++q, ++i) {
          float* o = grid + 2 * ((size_t)sy * nsc + rb * 12 + data_re[q]);
          o[0] = sym[2 * i], o[1] = sym[2 * i + 1];
        }
  /* dmrs_pdcch_processor_impl (:30-101) with dmrs_sequence_generate (dmrs_helper.h:44-96): three pilots per PRB, counted from the reference point */
  const float amp = (float)(M_SQRT1_2 * (double)powf(10.0f, dmrs_power_offset_dB / 20.0f));
  for (unsigned sy = start_symbol; sy < start_symbol + duration; ++sy) {
    const unsigned long long t = ((unsigned long long)(14 * slot_in_frame + sy + 1) * (2ull * n_id_dmrs + 1)) % (1ull << 31);
    const unsigned c_init      = (unsigned)((t * (1ull << 17) + 2ull * n_id_dmrs) % (1ull << 31));
    orc_gold_sequence(c_init, 0, 6 * nof_prb_grid, c);
    for (unsigned rb = reference_point_k_rb; rb < nof_prb_grid; ++rb) {
      if (!rb_mask[rb])
        continue;
      for (unsigned q = 0; q < 3; ++q) {
        const unsigned g = (rb - reference_point_k_rb) * 3 + q;
        float*         o = grid + 2 * ((size_t)sy * nsc + rb * 12 + 1 + 4 * q);
        o[0] = c[2 * g] ? -amp : amp, o[1] = c[2 * g + 1] ? -amp : amp;
      }
    }
  }
  free(enc), free(c), free(sym);
  return (int)(E / 2);
}

/* ================================================================================================ SS/PBCH block processor */
int orc_ssb_process(unsigned N_id, unsigned ssb_idx, unsigned L_max, int hrf, unsigned sfn, unsigned k_ssb, const uint8_t* payload, unsigned k0, unsigned l0,
                    float beta_pss_dB, unsigned nof_prb_grid, float* grid)
{
  const unsigned nsc = nof_prb_grid * 12, v = N_id % 4;
  uint8_t        enc[864], c[864];
  float          sym[864];
  if (k0 + 240 > nsc || l0 + 4 > 14)
    return -1;
  orc_pbch_encode(N_id, ssb_idx, L_max, hrf, sfn, k_ssb, payload, enc);
  /* pbch_modulator_impl::scramble (:28-38), modulate (:40-48), map (:50-94) */
  orc_gold_sequence(N_id, (ssb_idx & 7u) * 864u, 864, c);
  for (unsigned i = 0; i < 864; ++i)
    enc[i] = (enc[i] ^ c[i]) & 1u;
  orc_modulate(2, 432, enc, sym);
  unsigned cnt = 0;
  for (unsigned part = 0; part < 4; ++part) { /* symbol 1, symbol 2 lower, symbol 2 upper, symbol 3 */
    const unsigned l = l0 + (part == 0 ? 1 : (part == 3 ? 3 : 2)), ka = (part == 2) ? 192 : 0, kb = (part == 1) ? 48 : 240;
    for (unsigned k = ka; k < kb; ++k)
      if (k % 4 != v) {
        float* o = grid + 2 * ((size_t)l * nsc + k0 + k);
        o[0] = sym[2 * cnt], o[1] = sym[2 * cnt + 1];
        ++cnt;
      }
  }
  /* dmrs_pbch_processor_impl::c_init (:28-39), generation, mapping */
  unsigned i_ssb = (ssb_idx & 3u) + 4u * (hrf ? 1u : 0u);
  if (L_max == 8 || L_max == 64)
    i_ssb = ssb_idx & 7u;
  const unsigned c_init = (((i_ssb + 1u) * ((N_id / 4u) + 1u)) << 11) + ((i_ssb + 1u) << 6) + (N_id % 4u);
  orc_gold_sequence(c_init, 0, 288, c);
  const float a = (float)M_SQRT1_2;
  cnt           = 0;
  for (unsigned part = 0; part < 4; ++part) {
    const unsigned l = l0 + (part == 0 ? 1 : (part == 3 ? 3 : 2)), ka = (part == 2) ? 192 : 0, kb = (part == 1) ? 48 : 240;
    for (unsigned k = ka + v; k < kb; k += 4) {
      float* o = grid + 2 * ((size_t)l * nsc + k0 + k);
      o[0] = c[2 * cnt] ? -a : a, o[1] = c[2 * cnt + 1] ? -a : a;
      ++cnt;
    }
  }
  /* pss_processor_impl (:28-68) and sss_processor_impl (:28-103): m-sequences, cyclic shifts, products */
  uint8_t xp[134] = {0, 1, 1, 0, 1, 1, 1}, x0[134] = {1}, x1[134] = {1};
  for (unsigned i = 0; i < 127; ++i) {
    xp[i + 7] = (uint8_t)((xp[i + 4] + xp[i]) % 2);
    x0[i + 7] = (uint8_t)((x0[i + 4] + x0[i]) % 2);
    x1[i + 7] = (uint8_t)((x1[i + 1] + x1[i]) % 2);
  }
  const unsigned nid1 = N_id / 3, nid2 = N_id % 3, m = (43 * nid2) % 127, m0 = 15 * (nid1 / 112) + 5 * nid2, m1 = nid1 % 112;
  const float    amp = powf(10.0f, beta_pss_dB / 20.0f);
  for (unsigned n = 0; n < 127; ++n) {
    const float dp = 1.0f - 2.0f * (float)xp[(n + m) % 127];
    float*      o  = grid + 2 * ((size_t)l0 * nsc + k0 + 56 + n);
    o[0] = dp * amp, o[1] = 0.0f * amp;
    const float ar = (1.0f - 2.0f * (float)x0[(n + m0) % 127]) * 1.0f, ai = 0.0f * 1.0f, br = 1.0f - 2.0f * (float)x1[(n + m1) % 127], bi = 0.0f;
    o    = grid + 2 * ((size_t)(l0 + 2) * nsc + k0 + 56 + n);
    o[0] = ar * br - ai * bi, o[1] = ar * bi + ai * br;
  }
  return 0;
}

/* ================================================================================================ NZP-CSI-RS generator */
int orc_csi_rs_map(unsigned slot_in_frame, unsigned scrambling_id, float amplitude, unsigned start_rb, unsigned nof_rb, unsigned rb_begin, unsigned rb_end,
                   unsigned rb_stride, unsigned mapping_row, unsigned cdm, unsigned freq_density, unsigned nof_ports, const uint8_t* ports,
                   const uint16_t* re_mask, const uint16_t* symbol_mask, unsigned nof_prb_grid, float* grid)
{
  static const float w_f[2][2] = {{1.f, 1.f}, {1.f, -1.f}};
  static const float w_t[4][4] = {{1.f, 1.f, 1.f, 1.f}, {1.f, -1.f, 1.f, -1.f}, {1.f, 1.f, -1.f, -1.f}, {1.f, -1.f, -1.f, 1.f}};
  const unsigned     nsc = nof_prb_grid * 12;
  /* get_seq_len (:131-161) */
  unsigned seq_len = nof_rb;
  if (freq_density <= 1) {
    seq_len /= 2;
    if (nof_rb % 2 != 0) {
      if (start_rb % 2 != 0)
        seq_len += (freq_density == 1);
      else
        seq_len += (freq_density == 0);
    }
  } else if (freq_density == 3) {
    seq_len *= 3;
  }
  if (cdm != 0)
    seq_len *= 2;
  /* get_nof_skipped_elements (:69-108) */
  unsigned first_prb = start_rb, adv;
  if (freq_density == 0)
    first_prb = start_rb + start_rb % 2;
  else if (freq_density == 1)
    first_prb = start_rb + (1 - start_rb % 2);
  if (freq_density == 3)
    adv = 3 * first_prb;
  else if (freq_density == 2)
    adv = (mapping_row == 2) ? first_prb : 2 * first_prb;
  else
    adv = (mapping_row == 2) ? first_prb / 2 : first_prb;
  const unsigned gsize = cdm == 0 ? 1 : (cdm == 1 ? 2 : (cdm == 2 ? 4 : 8));
  const float    amp   = (float)(M_SQRT1_2 * (double)amplitude);
  uint8_t*       c     = (uint8_t*)malloc(2 * seq_len + 2);
  float*         seq   = (float*)malloc(sizeof(float) * (2 * seq_len + 2));
  for (unsigned p = 0; p < nof_ports; ++p) {
    const unsigned cidx = p % gsize;
    unsigned       lidx = 0;
    for (unsigned l = 0; l < 14; ++l) {
      if (!((symbol_mask[p] >> l) & 1u))
        continue;
      const unsigned long long t = (1024ull * (14 * slot_in_frame + l + 1) * (2ull * scrambling_id + 1) + scrambling_id) % (1ull << 31);
      orc_gold_sequence((unsigned)t, 2 * adv, 2 * seq_len, c);
      for (unsigned k = 0; k < seq_len; ++k) {
        float re = c[2 * k] ? -amp : amp, im = c[2 * k + 1] ? -amp : amp;
        if (cdm == 1) {
          const float w = w_f[cidx & 1][k & 1];
          re = w * re, im = w * im;
        } else if (cdm >= 2) {
          const float w = w_t[cidx >> 1][lidx] * w_f[cidx & 1][k & 1];
          re = w * re, im = w * im;
        }
        seq[2 * k] = re, seq[2 * k + 1] = im;
      }
      ++lidx;
      /* grid.put(port, l, start_rb * NRE, mask_csi, sequence): ascending subcarriers of the pattern inside the PRB window */
      unsigned k = 0;
      for (unsigned rb = rb_begin; rb < rb_end; rb += rb_stride) {
        if (rb < start_rb || rb >= start_rb + nof_rb)
          continue;
        for (unsigned sc = 0; sc < 12; ++sc)
          if ((re_mask[p] >> sc) & 1u) {
            if (k >= seq_len) {
              free(c), free(seq);
              return -1;
            }
            float* o = grid + 2 * (((size_t)ports[p] * 14 + l) * nsc + rb * 12 + sc);
            o[0] = seq[2 * k], o[1] = seq[2 * k + 1];
            ++k;
          }
      }
      if (k != seq_len) {
        free(c), free(seq);
        return -1;
      }
    }
  }
  free(c), free(seq);
  return 0;
}

/* ------------------------------------------------------------------------------------------------ UL-SCH demultiplexing (UCI on PUSCH)
 * ulsch_demultiplex_impl.cpp:74-453 (TS 38.212 6.2.7) as the serial scan it is: symbol by symbol, subcarrier by subcarrier, every
 * resource element goes to SCH data, HARQ-ACK, CSI part 1 or CSI part 2; HARQ-ACK on RESERVED elements (G_rvd != 0) punctures SCH /
 * CSI part 2, which receive an all-zero element there. One pass serves both users of the reference: demultiplexing (in != NULL)
 * and the list of repetition placeholders (elements of a field that carries exactly one information bit, in input order).
 * Returns the number of input LLRs, or -1 where the reference asserts. */
int orc_ulsch_demultiplex(int mod, unsigned nof_layers, unsigned nof_prb, unsigned start_symbol, unsigned nof_symbols, unsigned G_rvd, int dmrs_type,
                          unsigned dmrs_symbols_mask, unsigned cdm_groups, unsigned G_ack, unsigned G_csi1, unsigned G_csi2, unsigned O_ack, unsigned O_csi1,
                          unsigned O_csi2, const int8_t* in, int8_t* sch, int8_t* ack, int8_t* csi1, int8_t* csi2, unsigned* nof_sch_llr,
                          uint16_t* placeholders, unsigned* nof_placeholders)
{
  const unsigned bpr  = (unsigned)mod * nof_layers;
  const unsigned mask = dmrs_symbols_mask & 0x3fffu;
  if (!mask || nof_symbols == 0 || start_symbol + nof_symbols > 14)
    return -1;
  unsigned first_dmrs = 0, l1, l1_csi = 0;
  while (!((mask >> first_dmrs) & 1u))
    ++first_dmrs;
  for (l1 = first_dmrs; l1 < 14 && ((mask >> l1) & 1u); ++l1) {
  }
  if (l1 >= 14)
    return -1;
  while ((mask >> l1_csi) & 1u)
    ++l1_csi;
  const unsigned re_dmrs = (12u - cdm_groups * (dmrs_type == 1 ? 6u : 4u)) * nof_prb;
  const int      want_ph = mod >= 2 && (O_ack == 1 || O_csi1 == 1 || O_csi2 == 1);
  unsigned       m_rvd = 0, m_ack = 0, m_c1 = 0, m_c2 = 0;
  unsigned       n_in = 0, n_sch = 0, n_ack = 0, n_c1 = 0, n_c2 = 0, n_ph = 0; /* consumed / produced resource elements */
#define ORC_TAKE(dst, cnt)                                   \
  do {                                                       \
    if (in && (dst))                                         \
      memcpy((dst) + (size_t)(cnt)*bpr, in + (size_t)n_in * bpr, bpr); \
    ++(cnt), ++n_in;                                         \
  } while (0)
#define ORC_ZERO(dst, cnt)                      \
  do {                                          \
    if (in && (dst))                            \
      memset((dst) + (size_t)(cnt)*bpr, 0, bpr); \
    ++(cnt);                                    \
  } while (0)
  for (unsigned l = start_symbol; l < start_symbol + nof_symbols; ++l) {
    if ((mask >> l) & 1u) {
      for (unsigned i = 0; i < re_dmrs; ++i)
        ORC_TAKE(sch, n_sch);
      continue;
    }
    const unsigned M = nof_prb * 12u;
    unsigned       M_uci = M, M_rvd = 0;
    unsigned       ack_d = 0, ack_n = 0, rvd_d = 0, rvd_n = 0, c1_d = 0, c1_n = 0, c2_d = 0, c2_n = 0;
    if (l >= l1) {
      const unsigned rvd_left = G_rvd - m_rvd, ack_left = G_ack - m_ack;
      if (G_rvd != 0 && rvd_left != 0) {
        rvd_d = 1, rvd_n = M_uci;
        if (rvd_left < M_uci * bpr)
          rvd_d = (M_uci * bpr) / rvd_left, rvd_n = (rvd_left + bpr - 1) / bpr;
        M_rvd = rvd_n;
        if (ack_left != 0) {
          ack_d = 1, ack_n = M_rvd;
          if (ack_left < M_rvd * bpr)
            ack_d = (M_rvd * bpr) / ack_left, ack_n = (ack_left + bpr - 1) / bpr;
        }
      } else if (ack_left != 0) {
        ack_d = 1, ack_n = M_uci;
        if (ack_left < M_uci * bpr)
          ack_d = (M_uci * bpr) / ack_left, ack_n = (ack_left + bpr - 1) / bpr;
        M_uci -= ack_n;
      }
    }
    if (l >= l1_csi) {
      const unsigned c1_left = G_csi1 - m_c1, c2_left = G_csi2 - m_c2;
      if (M_uci > M_rvd && c1_left != 0) {
        c1_d = 1, c1_n = M_uci - M_rvd;
        if (c1_left < (M_uci - M_rvd) * bpr)
          c1_d = ((M_uci - M_rvd) * bpr) / c1_left, c1_n = (c1_left + bpr - 1) / bpr;
        M_uci -= c1_n;
      }
      if (M_uci > 0 && c2_left != 0) {
        c2_d = 1, c2_n = M_uci;
        if (c2_left < M_uci * bpr)
          c2_d = (M_uci * bpr) / c2_left, c2_n = (c2_left + bpr - 1) / bpr;
        M_uci -= c2_n;
      }
    }
    unsigned sch_left = M_uci;
    m_rvd += rvd_n * bpr, m_ack += ack_n * bpr, m_c1 += c1_n * bpr, m_c2 += c2_n * bpr;
    for (unsigned i = 0, i_c1 = 0, i_c2 = 0, i_ack = 0; i < M; ++i) {
      const int reserved = (rvd_n != 0) && (i % rvd_d == 0);
      int       zero     = 0;
      if (reserved)
        --rvd_n;
      if (G_rvd != 0) {
        if (reserved && ack_n != 0 && ((i_ack++) % ack_d == 0)) {
          if (want_ph && O_ack == 1)
            placeholders[n_ph++] = (uint16_t)n_in;
          ORC_TAKE(ack, n_ack);
          --ack_n;
          zero = 1;
        }
      } else if (ack_n != 0 && ((i_ack++) % ack_d == 0)) {
        if (want_ph && O_ack == 1)
          placeholders[n_ph++] = (uint16_t)n_in;
        ORC_TAKE(ack, n_ack);
        --ack_n;
        continue;
      }
      if (!reserved && c1_n != 0 && ((i_c1++) % c1_d == 0)) {
        if (want_ph && O_csi1 == 1)
          placeholders[n_ph++] = (uint16_t)n_in;
        ORC_TAKE(csi1, n_c1);
        --c1_n;
        continue;
      }
      if (c2_n != 0 && ((i_c2++) % c2_d == 0)) {
        if (zero) {
          ORC_ZERO(csi2, n_c2);
        } else {
          if (want_ph && O_csi2 == 1)
            placeholders[n_ph++] = (uint16_t)n_in;
          ORC_TAKE(csi2, n_c2);
        }
        --c2_n;
        continue;
      }
      if (sch_left == 0)
        return -1;
      if (zero)
        ORC_ZERO(sch, n_sch);
      else
        ORC_TAKE(sch, n_sch);
      --sch_left;
    }
    if (ack_n || c1_n || c2_n || sch_left)
      return -1;
  }
#undef ORC_TAKE
#undef ORC_ZERO
  if (m_rvd != G_rvd || m_ack != G_ack || m_c1 != G_csi1 || m_c2 != G_csi2)
    return -1;
  if (nof_sch_llr)
    *nof_sch_llr = n_sch * bpr;
  if (nof_placeholders)
    *nof_placeholders = n_ph;
  return (int)(n_in * bpr);
}

/* ------------------------------------------------------------------------------------------------ zero-forcing equalizer (stand-alone block)
 * channel_equalizer_zf_impl.cpp:123-162 picks the kernel; both kernels below follow the reference's scalar statements one by one
 * (std::complex products written out: (a+bi)(c+di) = (ac-bd) + (ad+bc)i, no fused multiply-add: this file is built with -ffp-contract=off). */
static int orc_isnormal_f(float x)
{
  return isnormal(x);
}
int orc_channel_equalize(unsigned nof_re, unsigned nof_rx_ports, unsigned nof_tx_layers, const float* ch_symbols, const float* ch_estimates,
                         float noise_var, float tx_scaling, float* eq_symbols, float* eq_noise_vars)
{
  const int nv_ok = orc_isnormal_f(noise_var) && noise_var > 0.0f;
  if (nof_tx_layers == 1 && nof_rx_ports >= 1 && nof_rx_ports <= 4) { /* equalize_zf_1xn.h:120-158 */
    for (unsigned i = 0; i != nof_re; ++i) {
      float ch_mod_sq = 0.0f, re = 0.0f, im = 0.0f;
      for (unsigned p = 0; p != nof_rx_ports; ++p) {
        const float* y = ch_symbols + 2 * ((size_t)p * nof_re + i);
        const float* c = ch_estimates + 2 * ((size_t)p * nof_re + i);
        ch_mod_sq += c[0] * c[0] + c[1] * c[1];
        re += y[0] * c[0] - y[1] * (-c[1]); /* re_in * conj(ch_est) */
        im += y[0] * (-c[1]) + y[1] * c[0];
      }
      eq_symbols[2 * i] = 0.0f, eq_symbols[2 * i + 1] = 0.0f;
      eq_noise_vars[i] = INFINITY;
      const float d_pinv = tx_scaling * ch_mod_sq;
      if (orc_isnormal_f(d_pinv) && nv_ok) {
        const float rcp      = 1.0f / d_pinv;
        eq_symbols[2 * i]     = re * rcp;
        eq_symbols[2 * i + 1] = im * rcp;
        eq_noise_vars[i]      = rcp * (noise_var / tx_scaling);
      }
    }
    return 0;
  }
  if (nof_tx_layers == 2 && nof_rx_ports == 2) { /* equalize_zf_2x2.cpp:56-116 */
    const size_t n = nof_re;
    for (unsigned i = 0; i != nof_re; ++i) {
      const float *a = ch_estimates + 2 * i, *c = ch_estimates + 2 * (n + i);           /* layer 0: port 0, port 1 */
      const float *b = ch_estimates + 2 * (2 * n + i), *d = ch_estimates + 2 * (3 * n + i); /* layer 1: port 0, port 1 */
      const float *r0 = ch_symbols + 2 * i, *r1 = ch_symbols + 2 * (n + i);
      const float n0 = (a[0] * a[0] + a[1] * a[1]) + (c[0] * c[0] + c[1] * c[1]);
      const float n1 = (b[0] * b[0] + b[1] * b[1]) + (d[0] * d[0] + d[1] * d[1]);
      /* xi = conj(a) b + conj(c) d */
      const float xr = (a[0] * b[0] - (-a[1]) * b[1]) + (c[0] * d[0] - (-c[1]) * d[1]);
      const float xi = (a[0] * b[1] + (-a[1]) * b[0]) + (c[0] * d[1] + (-c[1]) * d[0]);
      const float xm = xr * xr + xi * xi;
      const float m0r = (a[0] * r0[0] - (-a[1]) * r0[1]) + (c[0] * r1[0] - (-c[1]) * r1[1]);
      const float m0i = (a[0] * r0[1] + (-a[1]) * r0[0]) + (c[0] * r1[1] + (-c[1]) * r1[0]);
      const float m1r = (b[0] * r0[0] - (-b[1]) * r0[1]) + (d[0] * r1[0] - (-d[1]) * r1[1]);
      const float m1i = (b[0] * r0[1] + (-b[1]) * r0[0]) + (d[0] * r1[1] + (-d[1]) * r1[0]);
      const float d_pinv  = tx_scaling * ((n0 * n1) - xm);
      const float d_nvars = tx_scaling * d_pinv;
      float*      o0 = eq_symbols + 2 * i;
      float*      o1 = eq_symbols + 2 * (n + i);
      o0[0] = o0[1] = o1[0] = o1[1] = 0.0f;
      eq_noise_vars[i] = eq_noise_vars[n + i] = INFINITY;
      if (orc_isnormal_f(d_pinv) && nv_ok) {
        const float rp = 1.0f / d_pinv, rn = 1.0f / d_nvars;
        o0[0] = (n1 * m0r - (xr * m1r - xi * m1i)) * rp;
        o0[1] = (n1 * m0i - (xr * m1i + xi * m1r)) * rp;
        o1[0] = (n0 * m1r - (xr * m0r - (-xi) * m0i)) * rp; /* conj(xi) * m0 */
        o1[1] = (n0 * m1i - (xr * m0i + (-xi) * m0r)) * rp;
        eq_noise_vars[i]     = noise_var * n1 * rn;
        eq_noise_vars[n + i] = noise_var * n0 * rn;
      }
    }
    return 0;
  }
  return -1;
}
