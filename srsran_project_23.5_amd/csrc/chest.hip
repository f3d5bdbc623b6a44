// DM-RS based PUSCH channel estimator: one workgroup per (allocation, rx port, layer).
//
// Behaviour contract: lib/phy/upper/signal_processors/dmrs_pusch_estimator_impl.cpp:71-212 and
// port_channel_estimator_average_impl.cpp:97-347 (LS at the pilots, average over DM-RS symbols, RSRP / EPRE / noise /
// SNR, time alignment from the peak of a zero-padded 4096-point IDFT, linear interpolation, copy to all symbols).
// Everything stays on chip: Gold-sequence pilots are generated into LDS (28 bits per LFSR step), the LS estimates and the
// IDFT buffer live in LDS, HBM sees the DM-RS resource elements once and the estimate once.
#include "fft_device.h"
#include "gold_device.h"
#include "miphy_ext.h"
#include <cmath>
#include <cstdlib>

namespace {

constexpr int CE_DFT = 4096;                       // port_channel_estimator_average_impl::DFT_SIZE
constexpr int HALF_CP = ((144 / 2) * CE_DFT) / 2048; // 144
constexpr int MAX_PILOTS = 275 * 6;

// Gold sequences c(0..nbits-1) of `nseq` <= 4 initial values (one per DM-RS symbol), bit-packed LSB-first, 104 words apart in `out`:
// every word on a lane of its own from the tables (x1 is a constant sequence, x2 is linear in c_init: gold_device.h) -- 31 independent
// loads per lane instead of the serial 31-word LFSR head and three recurrence steps.
__device__ __forceinline__ void gold_bits_block(const gold_tables& gt, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int nseq, int nbits, uint32_t* out,
                                                int tid, int nt)
{
  const int nwords = (nbits + 31) >> 5; // <= GOLD_BASIS_WORDS (3 300 bits)
  for (int k = tid; k < nseq * nwords; k += nt) {
    const int      q  = k / nwords, i = k - q * nwords;
    const uint32_t ci = q == 0 ? c0 : (q == 1 ? c1 : (q == 2 ? c2 : c3));
    out[q * 104 + i]  = gt.x1_seq[i] ^ gold_x2_word(gt, ci, i);
  }
  __syncthreads();
}

__device__ __forceinline__ float block_sum(float v, float* red, int tid)
{
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
    v += __shfl_xor(v, off);
  __syncthreads();
  if ((tid & 63) == 0)
    red[tid >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (unsigned w = 1; w < blockDim.x / 64; ++w)
    r += red[w];
  return r;
}


// GENERAL = false: the PUSCH DM-RS estimator proper (pilots generated from the job, one hop: dmrs_pusch_estimator_impl.cpp never
// configures hopping); true: pilots from the caller and intra-slot frequency hopping (the port estimator on its own).
template <bool GENERAL>
__global__ void __launch_bounds__(512, GENERAL ? 2 : 4) chest_kernel(const miphy_pusch_chest_job* __restrict__ jobs,
                                                    const gold_jump* __restrict__ gj,
                                                    const cplx* __restrict__ tw,
                                                    const float2* __restrict__ grid,
                                                    float2* __restrict__ ce_out,
                                                    float* __restrict__ scalars,
                                                    const float2* __restrict__ ext_pilots_arg,
                                                    int ports_dim) // blockIdx.y = layer * ports_dim + port
{
  const float2* ext_pilots = GENERAL ? ext_pilots_arg : nullptr;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cplx*     fbuf   = reinterpret_cast<cplx*>(smem);                  // 4096 cplx: IDFT buffer, later interpolated response
  cplx*     lse    = fbuf + fft_lds_bytes(CE_DFT) / 8;                // MAX_PILOTS
  uint32_t* cbits  = reinterpret_cast<uint32_t*>(lse + MAX_PILOTS);   // 4 symbols x 104 words
  uint32_t* gtmp   = cbits + 4 * 104;                                 // Gold scratch: 4 symbols x 2 x 104 words
  uint16_t* prb_of = reinterpret_cast<uint16_t*>(gtmp + 4 * 208);     // allocated PRB list (<= 275)
  float*    red    = reinterpret_cast<float*>(prb_of + 276);          // reductions

  // Read field by field (uniform loads); the arrays go through LDS / a packed word: a private copy of the descriptor indexed at
  // run time would live in scratch memory.
  // static fields from a dword copy in scalar registers (load_words), the run-time indexed arrays from the record in memory
  const miphy_pusch_chest_job& jmem = jobs[blockIdx.x];
  const miphy_pusch_chest_job  job  = load_words(jobs + blockIdx.x);
  __shared__ uint64_t rbm[5];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int port = blockIdx.y % ports_dim, layer = blockIdx.y / ports_dim;
  // gridDim.z == 2 (launches that leave most of the chip idle -- a single slot): the time-alignment step -- the 4096-point IDFT and its peak
  // search, which only the side-band scalar sc[4] needs -- runs in a workgroup of its own (z = 1: LS estimates, IDFT, peak) next to the one
  // that estimates, interpolates and stores (z = 0): the demodulator behind this kernel waits for the shorter of the two chains + sc[2].
  const bool ta_only = gridDim.z == 2 && blockIdx.z == 1, do_ta = gridDim.z == 1 || ta_only, do_main = !ta_only;
  if (port >= job.nof_rx_ports || layer >= job.nof_tx_layers)
    return;
  const int nprb_grid = job.grid_nof_prb, nsc = nprb_grid * 12;
  const int first = job.first_symbol, nsymb_out = first + job.nof_symbols;
  const int hop_symbol = (GENERAL && job.hop_symbol > first && job.hop_symbol < nsymb_out) ? job.hop_symbol : 0;
  const int nhops      = hop_symbol ? 2 : 1;
  // DM-RS symbols of the whole allocation (the pilots of the second hop follow those of the first in an external pilot list)
  int nds_all = 0;
  for (int l = first; l < nsymb_out; ++l)
    nds_all += (job.symbols_mask >> l) & 1;
  const int   delta = ext_pilots ? ((job.re_odd_mask >> layer) & 1) : ((layer >> 1) & 1); // RE pattern: even subcarriers for ports 0,1 ; odd for 2,3
  const float wf1 = (layer & 1) ? -1.f : 1.f;       // frequency weight on odd pilots (layers > 0)
  const float amp = 0.70710678118654752440f;
  const float2* g = grid + job.grid_offset + (size_t)(((uint32_t)job.rx_ports[0] | ((uint32_t)job.rx_ports[1] << 8) | ((uint32_t)job.rx_ports[2] << 16) | ((uint32_t)job.rx_ports[3] << 24)) >> (8 * port) & 0xffu) * 14 * nsc;
  const float   beta = job.scaling;
  const bool    compact = job.ce_compact != 0;
  float2*       dst0    = ce_out + job.ce_offset + ((size_t)(layer * job.nof_rx_ports + port) * (compact ? 1 : nsymb_out)) * nsc;
  // accumulated over the hops (port_channel_estimator_average_impl.cpp:110-124)
  float epre_tot = 0.f, rsrp_tot = 0.f, noise_tot = 0.f, ta_tot = 0.f;
  int   np_sym = 0; // pilots per DM-RS symbol (the same in both hops)
  for (int hop = 0; hop < nhops; ++hop) {
  const int h_first = (hop == 1) ? hop_symbol : first, h_last = (hop == 0 && hop_symbol) ? hop_symbol : nsymb_out;
  __syncthreads();
  if (tid < 5)
    rbm[tid] = (hop == 1) ? jmem.rb_mask2[tid] : jmem.rb_mask[tid];
  __syncthreads();
  // DM-RS symbol list inside the hop.
  int dsym[4], nds = 0;
  for (int l = h_first; l < h_last; ++l)
    if ((job.symbols_mask >> l) & 1) {
      if (nds < 4)
        dsym[nds] = l;
      ++nds;
    }
  const int hop_offset = (hop == 1) ? nds_all - nds : 0; // compute_layer_hop: pilots.size().nof_symbols - nof_dmrs_symbols
  // Allocated PRB list (compact): PRB r goes to slot popcount(mask bits below r).
  for (int r = tid; r < nprb_grid; r += nt) {
    const int wd = r >> 6, bt = r & 63;
    if ((rbm[wd] >> bt) & 1ull) {
      int idx = __popcll(rbm[wd] & ((1ull << bt) - 1ull));
      for (int w = 0; w < wd; ++w)
        idx += __popcll(rbm[w]);
      prb_of[idx] = (uint16_t)r;
    }
  }
  int nprb = 0; // every thread: the same popcounts
  for (int w = 0; w < 5; ++w)
    nprb += __popcll(w * 64 < nprb_grid ? (rbm[w] & ((nprb_grid - w * 64 >= 64) ? ~0ull : ((1ull << (nprb_grid - w * 64)) - 1ull))) : 0ull);
  const int np = nprb * 6;
  np_sym       = np;
  __syncthreads(); // prb_of[]
  if (do_ta) // the IDFT buffer of the time-alignment step, cleared here: nothing waits for it
    for (int i = tid; i < (int)(fft_lds_bytes(CE_DFT) / 8); i += nt)
      fbuf[i] = {0.f, 0.f};
  // The received DM-RS elements of this thread's pilots (at most four per thread and DM-RS symbol: 1 650 pilots on 512 threads) are
  // requested NOW, before the pilot sequences are put together: the two memory round trips overlap, and both the LS phase and the noise
  // phase then work from registers (they used to fetch the same elements one after the other).
  constexpr int XK = 4;
  float2        xq[XK][4];
#pragma unroll
  for (int kk = 0; kk < XK; ++kk) {
    const int i = tid + kk * nt;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      xq[kk][d] = make_float2(0.f, 0.f);
      if (i < np && d < nds) {
        const int r = prb_of[i / 6], q = i % 6;
        xq[kk][d]   = g[(size_t)dsym[d] * nsc + r * 12 + 2 * q + delta];
      }
    }
  }
  if (!ext_pilots) {
    // Gold sequences of the DM-RS symbols (dmrs_pusch_estimator_impl.cpp:158-162).
    uint32_t c_init[4] = {0, 0, 0, 0};
    for (int q = 0; q < nds && q < 4; ++q) {
      const uint64_t t = ((uint64_t)(14u * job.slot_in_frame + (uint32_t)dsym[q] + 1u) * (2ull * job.scrambling_id + 1ull)) % (1ull << 31);
      c_init[q]        = (uint32_t)((t * (1ull << 17) + (2ull * job.scrambling_id + (job.n_scid ? 1u : 0u))) % (1ull << 31));
    }
    gold_bits_block(*reinterpret_cast<const gold_tables*>(gj), c_init[0], c_init[1], c_init[2], c_init[3], min(nds, 4), 12 * nprb_grid, cbits, tid, nt);
  } else {
    __syncthreads();
  }
  // pilot of DM-RS symbol d (of this hop), pilot i of the allocation: generated from the Gold sequence counted from PRB 0
  // (dmrs_helper.h:45-96) with the layer's frequency weight, or read from the caller's list
  const float2* xp = ext_pilots ? ext_pilots + job.pilots_offset + ((size_t)layer * nds_all + hop_offset) * np : nullptr;
  auto pilot = [&](int d, int i, int gp) -> cplx {
    if (xp) {
      const float2 v = xp[(size_t)d * np + i];
      return cplx{v.x, v.y};
    }
    const uint32_t* cb = cbits + d * 104;
    const int       b0 = 2 * gp;
    const float     pr = amp * (1.f - 2.f * (float)((cb[b0 >> 5] >> (b0 & 31)) & 1u));
    const float     pi = amp * (1.f - 2.f * (float)((cb[(b0 + 1) >> 5] >> ((b0 + 1) & 31)) & 1u));
    const float     w  = (layer != 0 && (i & 1)) ? wf1 : 1.f;
    return cplx{pr * w, pi * w};
  };

  // ---- LS estimate, EPRE (port_channel_estimator_average_impl.cpp:180-201)
  float epre_acc = 0.f, rsrp_acc = 0.f;
#pragma unroll
  for (int kk = 0; kk < XK; ++kk) {
    const int i = tid + kk * nt;
    if (i >= np)
      continue;
    const int r = prb_of[i / 6], q = i % 6;
    const int gp = r * 6 + q;                      // pilot index counted from PRB 0
    cplx      acc = {0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (d >= nds)
        continue;
      const cplx   p = pilot(d, i, gp);
      const float2 x = xq[kk][d];
      acc.x += x.x * p.x + x.y * p.y; // rx * conj(pilot)
      acc.y += x.y * p.x - x.x * p.y;
      epre_acc += x.x * x.x + x.y * x.y;
    }
    rsrp_acc += acc.x * acc.x + acc.y * acc.y;
    const float ts = 1.0f / ((float)nds * beta);
    lse[i]         = {acc.x * ts, acc.y * ts};
  }
  epre_tot += block_sum(epre_acc, red, tid);
  rsrp_tot += block_sum(rsrp_acc, red, tid) / (float)nds;
  __syncthreads();

  // ---- noise (:271-310): per-PRB mean of the estimates, predicted observation, residual power (symbol group 0 only)
  float noise_acc = 0.f;
#pragma unroll
  for (int kk = 0; kk < XK; ++kk) {
    const int i = tid + kk * nt;
    if (i >= (do_main ? np : 0))
      continue;
    const int b = (i / 6) * 6;
    cplx      avg = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 6; ++k)
      avg = cadd(avg, lse[b + k]);
    avg = {avg.x / 6.f * -beta, avg.y / 6.f * -beta};
    const int r = prb_of[i / 6], q = i % 6, gp = r * 6 + q;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (d >= nds)
        continue;
      const cplx   pred = cmul(avg, pilot(d, i, gp));
      const float2 x    = xq[kk][d];
      const float  er = pred.x + x.x, ei = pred.y + x.y;
      noise_acc += er * er + ei * ei;
    }
  }
  noise_tot += block_sum(noise_acc, red, tid) / (float)np * 6.f; // = sum over symbols of |residual|^2 / np * window (:300-309)

  // ---- time alignment (:312-347): zero-padded IDFT of the LS estimates at their RE positions
  if (do_ta) {
  // (fbuf was cleared at the head of the hop, under the memory requests; the barriers of the reductions above order it)
  for (int i = tid; i < np; i += nt)
    fbuf[fpad(prb_of[i / 6] * 12 + 2 * (i % 6) + delta)] = lse[i];
  __syncthreads();
  if (nt == 512)
    fft4096_lds<true>(fbuf, tw, tid); // CE_DFT = 4096 on 512 threads: compile-time strides
  else
    fft_lds<true>(fbuf, CE_DFT, tw, tid, nt);
  // arg-max of |.|^2 over the first / last HALF_CP taps (first occurrence wins, like std::max_element)
  float best = -1.f;
  int   bidx = 0x7fffffff;
  if (tid < HALF_CP) {
    const cplx v = fbuf[fpad(tid)];
    best = v.x * v.x + v.y * v.y;
    bidx = tid;
  } else if (tid < 2 * HALF_CP - 0 && tid >= HALF_CP && tid < 256) {
    const int  k = tid - HALF_CP; // 0..111 ; the remaining taps are handled below
    const cplx v = fbuf[fpad(CE_DFT - HALF_CP + k)];
    best = v.x * v.x + v.y * v.y;
    bidx = HALF_CP + k;
  }
  // taps HALF_CP-? : 2*HALF_CP = 288 > 256 threads -> second round for the rest of the "advance" window
  float best2 = -1.f;
  int   bidx2 = 0x7fffffff;
  if (tid + 256 < 2 * HALF_CP) {
    const int  k = tid + 256 - HALF_CP;
    const cplx v = fbuf[fpad(CE_DFT - HALF_CP + k)];
    best2 = v.x * v.x + v.y * v.y;
    bidx2 = HALF_CP + k;
  }
  // Reduce separately for delay (idx < HALF_CP) and advance (idx >= HALF_CP) windows using LDS atomics on a 64-bit key:
  // key = (value bits << 32) | (0xffffffff - idx) so that the max picks the largest value, then the smallest index.
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(red);
  __syncthreads();
  if (tid < 2)
    keys[tid] = 0ull;
  __syncthreads();
  if (bidx != 0x7fffffff)
    atomicMax(&keys[bidx >= HALF_CP ? 1 : 0], ((unsigned long long)__float_as_uint(best) << 32) | (0xffffffffu - (unsigned)bidx));
  if (bidx2 != 0x7fffffff)
    atomicMax(&keys[1], ((unsigned long long)__float_as_uint(best2) << 32) | (0xffffffffu - (unsigned)bidx2));
  __syncthreads();
  const float md = __uint_as_float((unsigned)(keys[0] >> 32)), ma = __uint_as_float((unsigned)(keys[1] >> 32));
  const int   id = (int)(0xffffffffu - (unsigned)(keys[0] & 0xffffffffu));
  const int   ia = (int)(0xffffffffu - (unsigned)(keys[1] & 0xffffffffu)) - HALF_CP;
  ta_tot += (md >= ma) ? (float)id : -(float)(HALF_CP - ia);
  __syncthreads();
  } // do_ta

  // ---- linear interpolation over the concatenated allocated PRBs (interpolator_linear_impl.cpp:58-78; offset = delta,
  // stride 2, edges held) written straight to every OFDM symbol of the hop (:216-224).
  const int nout = do_main ? nprb * 12 : 0;
  for (int k = tid; k < nout; k += nt) {
    cplx v;
    const int kk = k - delta;
    if (kk <= 0) {
      v = lse[0];
    } else {
      const int i = kk >> 1;
      if (i >= np - 1) {
        v = lse[np - 1];
      } else if (kk & 1) {
        v = {lse[i].x + (lse[i + 1].x - lse[i].x) * 0.5f, lse[i].y + (lse[i + 1].y - lse[i].y) * 0.5f};
      } else {
        v = lse[i];
      }
    }
    const int    r   = prb_of[k / 12];
    const size_t col = (size_t)r * 12 + (k % 12);
    if (compact) {
      dst0[col] = make_float2(v.x, v.y);
    } else {
      for (int l = h_first; l < h_last; ++l)
        dst0[(size_t)l * nsc + col] = make_float2(v.x, v.y);
    }
  }
  } // hops

  // ---- side-band scalars (:118-144)
  if (tid == 0) {
    const float ndp  = (float)(np_sym * nds_all);
    const float rsrp = rsrp_tot / ndp;
    const float epre = epre_tot / ndp;
    // noise_var = sum over hops and symbols of the residual energy / (window * all DM-RS symbols - 1)
    float noise_var = noise_tot / (float)(6 * nds_all - 1);
    if (nds_all < 3 || nhops == 2)
      noise_var = 0.001f * epre; // convert_dB_to_power(-30) * epre
    const float datarp = rsrp / beta / beta;
    const float snr    = (noise_var != 0.f) ? datarp / noise_var : 1000.f;
    const float scs_khz = 15.f * (float)(1u << job.numerology);
    float*      sc      = scalars + job.scalars_offset + 5 * ((size_t)port * job.nof_tx_layers + layer);
    if (do_main) {
      sc[0] = rsrp;
      sc[1] = epre;
      sc[2] = noise_var;
      sc[3] = snr;
    }
    if (do_ta)
      sc[4] = (nhops == 2 ? ta_tot / 2.0f : ta_tot) / ((float)CE_DFT * scs_khz * 1000.0f);
  }
}

} // namespace

static int chest_launch(miphy_ctx* ctx, const miphy_pusch_chest_job* jobs, int jobs_on_device, uint32_t n, const float* grid, const float* pilots, float* ce,
                        float* scalars, void* stream, const char* what)
{
  MIPHY_REQUIRE(ctx && jobs && grid && ce && scalars, "%s: null argument", what);
  if (n == 0)
    return MIPHY_OK;
  // Device-resident jobs cannot be inspected here: bits 8-11 / 12-15 of jobs_on_device may carry the largest nof_rx_ports / nof_tx_layers
  // of the batch (0 = unknown: the grid is sized for 4 x 4 and the surplus workgroups exit at once).
  const unsigned hint_ports = ((unsigned)jobs_on_device >> 8) & 0xfu, hint_layers = ((unsigned)jobs_on_device >> 12) & 0xfu;
  jobs_on_device &= 0xff;
  MIPHY_REQUIRE(hint_ports <= 4 && hint_layers <= 4, "%s: invalid port / layer hint", what);
  unsigned max_ports = hint_ports ? hint_ports : 4, max_layers = hint_layers ? hint_layers : 4;
  if (!jobs_on_device) {
    max_ports = max_layers = 1;
    for (uint32_t i = 0; i < n; ++i) {
      const miphy_pusch_chest_job& j = jobs[i];
      MIPHY_REQUIRE(j.numerology <= 4, "pusch_chest: job %u: invalid numerology", i);
      MIPHY_REQUIRE(j.nof_tx_layers >= 1 && j.nof_tx_layers <= 4, "pusch_chest: job %u: invalid number of layers %u", i, j.nof_tx_layers);
      MIPHY_REQUIRE(j.nof_rx_ports >= 1 && j.nof_rx_ports <= 4, "pusch_chest: job %u: invalid number of rx ports %u", i, j.nof_rx_ports);
      MIPHY_REQUIRE(j.grid_nof_prb >= 1 && j.grid_nof_prb <= 275, "pusch_chest: job %u: invalid grid width", i);
      MIPHY_REQUIRE(j.first_symbol + j.nof_symbols <= 14 && j.nof_symbols > 0, "pusch_chest: job %u: invalid symbol range", i);
      MIPHY_REQUIRE(j.scaling > 0, "pusch_chest: job %u: the DM-RS to data scaling factor should be a positive number", i);
      unsigned nds = 0, nprb = 0;
      for (unsigned l = j.first_symbol; l < (unsigned)j.first_symbol + j.nof_symbols; ++l)
        nds += (j.symbols_mask >> l) & 1;
      for (unsigned r = 0; r < j.grid_nof_prb; ++r)
        nprb += (unsigned)((j.rb_mask[r >> 6] >> (r & 63)) & 1ull);
      MIPHY_REQUIRE(nds >= 1 && nds <= 4, "pusch_chest: job %u: %u DM-RS symbols (1..4 supported)", i, nds);
      MIPHY_REQUIRE(pilots || j.hop_symbol == 0, "pusch_chest: job %u: intra-slot frequency hopping is served by miphy_port_channel_estimate_batch", i);
      if (j.hop_symbol != 0) { // intra-slot frequency hopping (port_channel_estimator_average_impl.cpp:118-124,153-163)
        MIPHY_REQUIRE(j.hop_symbol > j.first_symbol && j.hop_symbol < j.first_symbol + j.nof_symbols, "pusch_chest: job %u: hop symbol outside the allocation", i);
        MIPHY_REQUIRE(!j.ce_compact, "pusch_chest: job %u: the compact estimate holds one hop only", i);
        unsigned d0 = 0, d1 = 0, nprb2 = 0;
        for (unsigned l = j.first_symbol; l < (unsigned)j.first_symbol + j.nof_symbols; ++l)
          (l < j.hop_symbol ? d0 : d1) += (j.symbols_mask >> l) & 1;
        for (unsigned r = 0; r < j.grid_nof_prb; ++r)
          nprb2 += (unsigned)((j.rb_mask2[r >> 6] >> (r & 63)) & 1ull);
        MIPHY_REQUIRE(d0 >= 1 && d1 >= 1, "pusch_chest: job %u: every hop needs a DM-RS symbol", i);
        MIPHY_REQUIRE(nprb2 == nprb, "pusch_chest: job %u: the hops have different numbers of PRBs", i);
      }
      MIPHY_REQUIRE(nprb >= 1, "pusch_chest: job %u: empty allocation", i);
      max_ports  = j.nof_rx_ports > max_ports ? j.nof_rx_ports : max_ports;
      max_layers = j.nof_tx_layers > max_layers ? j.nof_tx_layers : max_layers;
    }
  }
  const float* tw = nullptr;
  int          rc = miphy_get_twiddles(ctx, CE_DFT, &tw);
  if (rc)
    return rc;
  hipStream_t s      = (hipStream_t)stream;
  const void* d_jobs = nullptr;
  if ((rc = miphy_stage_descs(ctx, jobs, jobs_on_device, sizeof(miphy_pusch_chest_job) * (size_t)n, s, &d_jobs)))
    return rc;
  const size_t lds = fft_lds_bytes(CE_DFT) + (size_t)MAX_PILOTS * 8 + 4 * 104 * 4 + 4 * 208 * 4 + 276 * 2 + 64 + 64;
  if (!ctx->ext->d_gold) {
    int rc = miphy_get_gold_tables(ctx, nullptr);
    if (rc)
      return rc;
  }
  // One workgroup of 512 threads per (job, port, layer); two -- the time-alignment chain on its own -- where the launch leaves most of the
  // chip idle anyway. (256 threads are slower for the 4096-point IDFT.)
  const int cthreads = 512;
  const int ngrp     = ((uint64_t)n * max_ports * max_layers * 2 <= (uint64_t)ctx->num_cus) ? 2 : 1;
  if (pilots)
    hipLaunchKernelGGL(chest_kernel<true>, dim3(n, max_ports * max_layers, ngrp), dim3(cthreads), lds, s, (const miphy_pusch_chest_job*)d_jobs,
                       (const gold_jump*)ctx->ext->d_gold, (const cplx*)tw, (const float2*)grid, (float2*)ce, scalars, (const float2*)pilots, (int)max_ports);
  else
    hipLaunchKernelGGL(chest_kernel<false>, dim3(n, max_ports * max_layers, ngrp), dim3(cthreads), lds, s, (const miphy_pusch_chest_job*)d_jobs,
                       (const gold_jump*)ctx->ext->d_gold, (const cplx*)tw, (const float2*)grid, (float2*)ce, scalars, (const float2*)nullptr, (int)max_ports);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

extern "C" int miphy_dmrs_pusch_estimate_batch(miphy_ctx* ctx, const miphy_pusch_chest_job* jobs, int jobs_on_device, uint32_t n, const float* grid, float* ce,
                                               float* scalars, void* stream)
{
  return chest_launch(ctx, jobs, jobs_on_device, n, grid, nullptr, ce, scalars, stream, "miphy_dmrs_pusch_estimate_batch");
}

extern "C" int miphy_port_channel_estimate_batch(miphy_ctx* ctx, const miphy_pusch_chest_job* jobs, int jobs_on_device, uint32_t n, const float* grid,
                                                 const float* pilots, float* ce, float* scalars, void* stream)
{
  MIPHY_REQUIRE(pilots, "miphy_port_channel_estimate_batch: null pilots");
  return chest_launch(ctx, jobs, jobs_on_device, n, grid, pilots, ce, scalars, stream, "miphy_port_channel_estimate_batch");
}
