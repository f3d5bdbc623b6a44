// Polar coding chains (PDCCH / PBCH / UCI): code construction on the host, one wavefront per codeword on the device.
//
// Behaviour contract (all under lib/phy/upper/channel_coding/polar/): polar_code_impl.cpp:325-490 (sets, n, nPC),
// polar_allocator_impl.cpp:28-70, polar_encoder_impl.cpp:32-86, polar_rate_matcher_impl.cpp:31-106,
// polar_rate_dematcher_impl.cpp:29-118, polar_decoder_impl.cpp:32-350 (simplified successive cancellation),
// polar_deallocator_impl.cpp:27-42, polar_interleaver_impl.cpp:27-56; PDCCH: channel_processors/pdcch_encoder_impl.cpp:33-98.
//
// MI355X mapping: the decoding tree only depends on the frozen set, so the host flattens the recursion into a pruned
// schedule (f / g / rate-1 / combine ops); a 64-lane wavefront executes it with all LLR stages, partial sums and
// decisions resident in LDS (2N + 2N bytes per codeword). Rate matching and its inverse are single gathers through
// host-composed index tables (sub-block interleaver o bit selection o channel interleaver).
#include "crc_device.h"
#include "miphy_ext.h"
#include "tables/nr_polar_tables.h"
#include <algorithm>
#include <vector>

namespace {

enum { OP_F = 1, OP_G = 2, OP_R1 = 3, OP_COMB = 4 };

struct host_code {
  uint32_t              K, E, n, N, nPC, nWmPC;
  std::vector<uint8_t>  k_set;
  std::vector<uint16_t> pc_set;
  std::vector<uint16_t> blk;
};

// polar_code_impl.cpp:325-490
int build_code(const miphy_polar_code* c, host_code& h)
{
  const uint32_t K = c->K, E = c->E, nMax = c->nMax;
  MIPHY_REQUIRE(E <= 8192, "polar: E = %u exceeds EMAX", E);
  if (nMax == 9) {
    MIPHY_REQUIRE(!(K < 36 || K > 164), "polar: codeblock length (K=%u) not supported for downlink transmission, choose 165 > K > 35", K);
  } else if (nMax == 10) {
    MIPHY_REQUIRE(!(K < 18 || (K > 25 && K < 31) || K > 1023), "polar: codeblock length (K=%u) not supported for uplink transmission", K);
  } else {
    MIPHY_REQUIRE(false, "polar: nMax not supported, choose 9 for downlink and 10 for uplink transmissions");
  }
  uint32_t nPC = 0, nWmPC = 0;
  if (K <= 25) {
    nPC = 3;
    if (E > K + 189)
      nWmPC = 1;
  }
  MIPHY_REQUIRE(K + nPC < E, "polar: rate-matched codeword length (E=%u) not supported, choose E > K + nPC", E);
  uint32_t e = 1;
  while (e <= 13 && (1u << e) < E)
    ++e;
  const uint32_t n1 = ((8 * E <= 9 * (1u << (e - 1))) && (16 * K < 9 * E)) ? e - 1 : e;
  uint32_t       k  = 0;
  while (k <= 10 && (1u << k) < K)
    ++k;
  uint32_t n = std::min(std::min(n1, k + 3), nMax);
  n          = std::max(n, 5u);
  const uint32_t N = 1u << n;
  MIPHY_REQUIRE(K < N, "polar: codeblock length (K=%u) not supported, choose K < N", K);
  h.K = K, h.E = E, h.n = n, h.N = N, h.nPC = nPC, h.nWmPC = nWmPC;
  std::vector<uint16_t> mother;
  for (uint32_t i = 0; i < 1024; ++i)
    if (NR_POLAR_Q1024[i] < N)
      mother.push_back(NR_POLAR_Q1024[i]);
  h.blk.resize(N);
  for (uint32_t j = 0; j < N; ++j)
    h.blk[j] = (uint16_t)(NR_POLAR_SUBBLOCK_P[32 * j / N] * (N / 32) + j % (N / 32));
  std::vector<uint16_t> cand(mother);
  if (N > E) {
    std::vector<uint8_t> drop(N, 0);
    uint32_t             T = 0;
    if (16 * K <= 7 * E) { // puncturing
      const uint32_t N_th = 3 * N / 4;
      T                   = (E >= N_th) ? N_th - (E >> 1) - 1 : 9 * N / 16 - (E >> 2);
      for (uint32_t i = 0; i < N - E; ++i)
        drop[h.blk[i]] = 1;
    } else { // shortening
      for (uint32_t i = E; i < N; ++i)
        drop[h.blk[i]] = 1;
    }
    cand.clear();
    for (uint16_t q : mother)
      if (!(q <= T) && !drop[q]) // setdiff_stable: also drops every index <= T (T = 0 when shortening)
        cand.push_back(q);
  }
  MIPHY_REQUIRE(cand.size() >= K + nPC, "polar: not enough reliable positions");
  const uint16_t* Kset = cand.data() + (cand.size() - K - nPC);
  h.pc_set.clear();
  for (uint32_t i = 0; i < ((nPC > nWmPC) ? nPC - nWmPC : 0); ++i)
    h.pc_set.push_back(Kset[i]);
  if (nWmPC == 1)
    h.pc_set.push_back((K <= 21) ? 252 : 248);
  std::sort(h.pc_set.begin(), h.pc_set.end());
  h.k_set.assign(N, 0);
  for (uint32_t i = 0; i < K + nPC; ++i)
    h.k_set[Kset[i]] = 1;
  return MIPHY_OK;
}

void emit(std::vector<uint32_t>& s, uint32_t op, uint32_t stage, uint32_t pos)
{
  s.push_back(op | (stage << 4) | (pos << 8));
}

// Flattens polar_decoder_impl.cpp:209-333 (rate_0_node / rate_1_node / rate_r_node) into a list of vector operations.
void build_schedule(const std::vector<uint8_t>& k_set, uint32_t s, uint32_t pos, std::vector<uint32_t>& out)
{
  const uint32_t size = 1u << s;
  bool           any = false, all = true;
  for (uint32_t i = 0; i < size; ++i) {
    any |= k_set[pos + i] != 0;
    all &= k_set[pos + i] != 0;
  }
  if (!any)
    return;
  if (all) {
    emit(out, OP_R1, s, pos);
    return;
  }
  emit(out, OP_F, s, pos);
  build_schedule(k_set, s - 1, pos, out);
  emit(out, OP_G, s, pos);
  build_schedule(k_set, s - 1, pos + size / 2, out);
  emit(out, OP_COMB, s, pos);
}

template <typename T>
int upload(miphy_ctx* ctx, const std::vector<T>& v, T** d)
{
  const size_t bytes = std::max<size_t>(sizeof(T), v.size() * sizeof(T));
  MIPHY_HIP_CHECK(hipMalloc((void**)d, bytes));
  if (!v.empty())
    MIPHY_HIP_CHECK(hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  ctx->ext->to_free.push_back(*d);
  return MIPHY_OK;
}

int get_plan(miphy_ctx* ctx, const miphy_polar_code* c, const polar_plan** out)
{
  auto key = std::make_tuple(c->K, c->E, c->nMax, c->ibil ? 1u : 0u);
  auto it  = ctx->ext->polar_plans.find(key);
  if (it != ctx->ext->polar_plans.end()) {
    *out = &it->second;
    return MIPHY_OK;
  }
  host_code h;
  int       rc = build_code(c, h);
  if (rc)
    return rc;
  const uint32_t N = h.N, E = h.E, K = h.K;
  polar_plan     p = {};
  p.K = K, p.E = E, p.n = h.n, p.N = N, p.nPC = h.nPC;
  std::vector<uint16_t> info_pos;
  std::vector<uint8_t>  is_pc;
  for (uint32_t q = 0; q < N; ++q)
    if (h.k_set[q]) {
      info_pos.push_back((uint16_t)q);
      is_pc.push_back(std::find(h.pc_set.begin(), h.pc_set.end(), (uint16_t)q) != h.pc_set.end());
    }
  // Channel interleaver (polar_rate_matcher_impl.cpp:62-88): f[io] = e[ii].
  std::vector<uint16_t> perm(E);
  if (c->ibil) {
    uint32_t S = 1, T = 1;
    while (S < E) {
      ++T;
      S += T;
    }
    uint32_t io = 0;
    for (uint32_t r = 0; r < T; ++r) {
      uint32_t ii = r;
      for (uint32_t cc = 0; cc < T - r; ++cc) {
        if (ii < E) {
          perm[io++] = (uint16_t)ii;
          ii += T - cc;
        } else
          break;
      }
    }
  } else {
    for (uint32_t i = 0; i < E; ++i)
      perm[i] = (uint16_t)i;
  }
  // Bit selection (polar_rate_matcher_impl.cpp:43-60): e[k] = y[sel(k)], y[j] = d[blk[j]].
  const bool punct = (E < N) && (16 * K <= 7 * E);
  auto       sel   = [&](uint32_t k) { return (E >= N) ? k % N : (punct ? k + (N - E) : k); };
  std::vector<uint16_t> tx_src(E), rx_fidx(E);
  for (uint32_t o = 0; o < E; ++o) {
    tx_src[o]        = h.blk[sel(perm[o])];
    rx_fidx[perm[o]] = (uint16_t)o;
  }
  // Inverse (polar_rate_dematcher_impl.cpp:43-68): for codeword position q = blk[j], y[j] comes from e[j'] (+ repetitions).
  std::vector<int32_t> rx_first(N);
  for (uint32_t j = 0; j < N; ++j) {
    int32_t first;
    if (E >= N)
      first = (int32_t)j;
    else if (punct)
      first = (j < N - E) ? -1 : (int32_t)(j - (N - E));
    else
      first = (j < E) ? (int32_t)j : -2;
    rx_first[h.blk[j]] = first;
  }
  std::vector<uint32_t> sched;
  build_schedule(h.k_set, h.n, 0, sched);
  std::vector<uint8_t> pi_il;
  for (uint32_t m = 0; m < NR_POLAR_K_MAX_IL; ++m)
    if (K <= NR_POLAR_K_MAX_IL && NR_POLAR_PI_IL_MAX[m] >= NR_POLAR_K_MAX_IL - K)
      pi_il.push_back((uint8_t)(NR_POLAR_PI_IL_MAX[m] - (NR_POLAR_K_MAX_IL - K)));
  p.sched_len = (uint32_t)sched.size();
  if ((rc = upload(ctx, info_pos, &p.d_info_pos)) || (rc = upload(ctx, is_pc, &p.d_is_pc)) || (rc = upload(ctx, tx_src, &p.d_tx_src)) ||
      (rc = upload(ctx, rx_first, &p.d_rx_first)) || (rc = upload(ctx, rx_fidx, &p.d_rx_fidx)) || (rc = upload(ctx, sched, &p.d_sched)) ||
      (rc = upload(ctx, pi_il, &p.d_pi_il)))
    return rc;
  auto ins = ctx->ext->polar_plans.emplace(key, p);
  *out     = &ins.first->second;
  return MIPHY_OK;
}

// ---------------------------------------------------------------------------------------------------- device side
__device__ __forceinline__ void polar_transform_lds(uint8_t* x, int n, int lane)
{ // polar_encoder_impl.cpp:32-52: x[i] ^= x[i + half] for every level, natural order.
  const int N = 1 << n;
  for (int half = 1; half < N; half <<= 1) {
    for (int t = lane; t < N / 2; t += 64) {
      const int b = ((t / half) * 2 * half) + (t % half);
      x[b] ^= x[b + half];
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void polar_tx_chain(const polar_plan& p, uint8_t* u, const uint8_t* msg_lds, uint8_t* __restrict__ out,
                                               uint8_t* __restrict__ alloc_tap, uint8_t* __restrict__ enc_tap, int lane)
{
  const int N = (int)p.N, KP = (int)(p.K + p.nPC);
  for (int i = lane; i < N; i += 64)
    u[i] = 0;
  __syncthreads();
  if (p.nPC == 0) { // polar_allocator_impl.cpp:37-41
    for (int i = lane; i < KP; i += 64)
      u[p.d_info_pos[i]] = msg_lds[i];
  } else if (lane == 0) { // :42-68, five-stage cyclic shift register, inherently serial and only used for K <= 25
    unsigned y[5] = {0, 0, 0, 0, 0};
    int      iK = 0, ip = 0;
    for (int q = 0; q < N; ++q) {
      const unsigned t = y[0];
      y[0] = y[1], y[1] = y[2], y[2] = y[3], y[3] = y[4], y[4] = t;
      if (ip < KP && p.d_info_pos[ip] == q) {
        if (p.d_is_pc[ip]) {
          u[q] = (uint8_t)y[0];
        } else {
          u[q] = msg_lds[iK];
          y[0] ^= msg_lds[iK];
          ++iK;
        }
        ++ip;
      }
    }
  }
  __syncthreads();
  if (alloc_tap)
    for (int i = lane; i < N; i += 64)
      alloc_tap[i] = u[i];
  polar_transform_lds(u, (int)p.n, lane);
  if (enc_tap)
    for (int i = lane; i < N; i += 64)
      enc_tap[i] = u[i];
  for (int o = lane; o < (int)p.E; o += 64) // sub-block interleaver, bit selection, channel interleaver: one gather
    out[o] = u[p.d_tx_src[o]];
}

__global__ void __launch_bounds__(64) polar_encode_kernel(polar_plan p, const uint8_t* __restrict__ msg, uint8_t* __restrict__ out,
                                                          uint8_t* __restrict__ alloc_tap, uint8_t* __restrict__ enc_tap)
{
  __shared__ uint8_t u[1024];
  __shared__ uint8_t m[1024];
  const int          lane = threadIdx.x;
  const size_t       cw   = blockIdx.x;
  for (int i = lane; i < (int)p.K; i += 64)
    m[i] = msg[cw * p.K + i];
  __syncthreads();
  polar_tx_chain(p, u, m, out + cw * p.E, alloc_tap ? alloc_tap + cw * p.N : nullptr, enc_tap ? enc_tap + cw * p.N : nullptr, lane);
}

// PDCCH (ones = 24 leading ones, RNTI mask) and PBCH (ones = 0, rnti == nullptr) share the CRC24C + interleave + polar chain.
__global__ void __launch_bounds__(64) pdcch_encode_kernel(polar_plan p, uint32_t A, const uint8_t* __restrict__ payload,
                                                          const uint16_t* __restrict__ rnti, uint8_t* __restrict__ out, int ones)
{
  __shared__ uint8_t u[1024];
  __shared__ uint8_t c[24 + 164];
  __shared__ uint8_t cp[164];
  const int          lane = threadIdx.x;
  const size_t       cw   = blockIdx.x;
  const int          K    = (int)p.K;
  // pdcch_encoder_impl.cpp:33-59: 24 leading ones, payload, CRC24C, RNTI mask on the last 16 parity bits.
  for (int i = lane; i < 24 + (int)A; i += 64)
    c[i] = (i < 24) ? 1 : payload[cw * A + (i - 24)];
  __syncthreads();
  if (lane == 0) {
    uint32_t reg = 0;
    for (int i = 24 - ones; i < 24 + (int)A; ++i) {
      reg = (reg << 1) ^ ((uint32_t)(c[i] & 1u) << 24);
      reg ^= (reg & 0x1000000u) ? 0x1B2B117u : 0u;
    }
    const uint32_t r = rnti ? rnti[cw] : 0u;
    for (int i = 0; i < 24; ++i) {
      uint32_t b = (reg >> (23 - i)) & 1u;
      if (i >= 8)
        b ^= (r >> (15 - (i - 8))) & 1u;
      c[24 + A + i] = (uint8_t)b;
    }
  }
  __syncthreads();
  for (int k = lane; k < K; k += 64) // CRC interleaver (polar_interleaver_impl.cpp:37-56), tx direction
    cp[k] = c[24 + p.d_pi_il[k]];
  __syncthreads();
  polar_tx_chain(p, u, cp, out + cw * p.E, nullptr, nullptr, lane);
}

// LLR algebra of log_likelihood_ratio.cpp:38-85 / .h:208-216.
#ifndef POLAR_SSC_CW
#define POLAR_SSC_CW 4
#endif
__device__ __forceinline__ int llr_add(int a, int b)
{ // a + b (special cases inspect the right operand first, like `rhs += *this`)
  if (b == -a)
    return 0;
  if (b > 120 || b < -120)
    return b;
  if (a > 120 || a < -120)
    return a;
  return min(max(a + b, -120), 120);
}
__device__ __forceinline__ int llr_promotion_sum(int a, int b)
{
  if (a == -b)
    return 0;
  if (a > 120 || a < -120)
    return a;
  if (b > 120 || b < -120)
    return b;
  const int t = a + b;
  return (t > 120) ? 127 : ((t < -120) ? -127 : t);
}
__device__ __forceinline__ int llr_soft_xor(int x, int y)
{
  const int m = min(abs(x), abs(y));
  return (x * y < 0) ? -m : m;
}

// C codewords per wavefront, W = 64 / C lanes each. Every codeword of a batch has the same code, hence the same pruned schedule: the codewords of a
// wavefront run it in lockstep, each on its own LDS slice. One codeword per wavefront kept 23 % of the lanes busy (SQ_THREAD_CYCLES_VALU, profiles/r03):
// below stage 6 an operation is narrower than the wavefront, and the pruned schedule spends most of its operations there; with C codewords the narrow
// operations fill C times the lanes and every operation's fixed cost (schedule word, barrier) is shared by C codewords.
template <int C>
__global__ void __launch_bounds__(64) polar_decode_kernel(polar_plan p, uint32_t ncw, const int8_t* __restrict__ llr_in, uint8_t* __restrict__ msg_out,
                                                          int8_t* __restrict__ dem_tap, uint8_t* __restrict__ u_tap)
{
  constexpr int W = 64 / C;
  __shared__ int8_t  L_all[C][2048]; // stage s buffer at offset 2^s (size 2^s)
  __shared__ uint8_t est_all[C][1024];
  __shared__ uint8_t u_all[C][1024];
  const int          c = threadIdx.x / W, lane = threadIdx.x % W;
  const size_t       cw = (size_t)blockIdx.x * C + c;
  const bool         live = cw < ncw;
  int8_t*            L    = L_all[c];
  uint8_t*           est  = est_all[c];
  uint8_t*           u    = u_all[c];
  const int          N = (int)p.N, n = (int)p.n, E = (int)p.E;
  const int8_t*      f = llr_in + (live ? cw : 0) * p.E;
  // Rate dematching (polar_rate_dematcher_impl.cpp:29-118) as a gather: repetitions are accumulated in order.
  for (int q = lane; q < N; q += W) {
    const int first = p.d_rx_first[q];
    int       v;
    if (first == -1) {
      v = 0;
    } else if (first == -2) {
      v = 127;
    } else {
      v = f[p.d_rx_fidx[first]];
      for (int k = first + N; k < E; k += N)
        v = llr_promotion_sum(v, f[p.d_rx_fidx[k]]);
    }
    L[N + q] = (int8_t)v;
    est[q]   = 0;
    u[q]     = 0;
    if (dem_tap && live)
      dem_tap[cw * N + q] = (int8_t)v;
  }
  __syncthreads();
  for (uint32_t k = 0; k < p.sched_len; ++k) {
    const uint32_t op    = p.d_sched[k];
    const int      type  = op & 15, s = (op >> 4) & 15, pos = (int)(op >> 8);
    const int      size  = 1 << s, half = size >> 1;
    int8_t*        ls    = L + size;
    int8_t*        lc    = L + half;
    if (type == OP_F) {
      for (int i = lane; i < half; i += W)
        lc[i] = (int8_t)llr_soft_xor(ls[i], ls[i + half]);
    } else if (type == OP_G) {
      for (int i = lane; i < half; i += W) {
        const int x = ls[i], y = ls[i + half];
        lc[i]       = (int8_t)(est[pos + i] ? llr_add(y, -x) : llr_add(y, x));
      }
    } else if (type == OP_R1) {
      for (int i = lane; i < size; i += W) {
        const uint8_t b = ls[i] <= 0;
        est[pos + i]    = b;
        u[pos + i]      = b;
      }
      __syncthreads();
      for (int h = 1; h < size; h <<= 1) { // re-encode the subtree (polar_decoder_impl.cpp:243-248)
        for (int t = lane; t < half; t += W) {
          const int b = ((t / h) * 2 * h) + (t % h);
          u[pos + b] ^= u[pos + b + h];
        }
        __syncthreads();
      }
    } else { // OP_COMB
      for (int i = lane; i < half; i += W)
        est[pos + i] ^= est[pos + half + i];
    }
    __syncthreads();
  }
  if (!live)
    return;
  if (u_tap)
    for (int i = lane; i < N; i += W)
      u_tap[cw * N + i] = u[i];
  // Deallocation (polar_deallocator_impl.cpp:27-42): K-set positions that are not parity checks, ascending.
  if (p.nPC == 0) {
    for (int i = lane; i < (int)p.K; i += W)
      msg_out[cw * p.K + i] = u[p.d_info_pos[i]];
  } else if (lane == 0) {
    int iK = 0;
    for (int i = 0; i < (int)(p.K + p.nPC); ++i)
      if (!p.d_is_pc[i])
        msg_out[cw * p.K + iK++] = u[p.d_info_pos[i]];
  }
  (void)n;
}


// ---- single blocks of the chain, for the block-level interfaces of the reference (polar_allocator / polar_encoder /
// polar_rate_matcher / polar_rate_dematcher / polar_decoder / polar_deallocator / polar_interleaver): the same device code as the
// chains above, entered and left at one stage. One wavefront per codeword.
__global__ void __launch_bounds__(64) polar_block_kernel(polar_plan p, int op, int param, const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                         const uint8_t* __restrict__ pi_il_max)
{
  __shared__ int8_t  L[2048];
  __shared__ uint8_t est[1024];
  __shared__ uint8_t u[1024];
  const int          lane = threadIdx.x;
  const size_t       cw   = blockIdx.x;
  const int          N = (int)p.N, E = (int)p.E, K = (int)p.K, KP = (int)(p.K + p.nPC);
  switch (op) {
    case MIPHY_POLAR_OP_ALLOCATE: { // polar_allocator_impl.cpp:28-70
      uint8_t* m = est;
      for (int i = lane; i < K; i += 64)
        m[i] = in[cw * K + i];
      for (int i = lane; i < N; i += 64)
        u[i] = 0;
      __syncthreads();
      if (p.nPC == 0) {
        for (int i = lane; i < KP; i += 64)
          u[p.d_info_pos[i]] = m[i];
      } else if (lane == 0) {
        unsigned y[5] = {0, 0, 0, 0, 0};
        int      iK = 0, ip = 0;
        for (int q = 0; q < N; ++q) {
          const unsigned t = y[0];
          y[0] = y[1], y[1] = y[2], y[2] = y[3], y[3] = y[4], y[4] = t;
          if (ip < KP && p.d_info_pos[ip] == q) {
            if (p.d_is_pc[ip]) {
              u[q] = (uint8_t)y[0];
            } else {
              u[q] = m[iK];
              y[0] ^= m[iK];
              ++iK;
            }
            ++ip;
          }
        }
      }
      __syncthreads();
      for (int i = lane; i < N; i += 64)
        out[cw * N + i] = u[i];
      break;
    }
    case MIPHY_POLAR_OP_ENCODE: { // polar_encoder_impl.cpp:32-86, param = log2 of the code size
      const int NN = 1 << param;
      for (int i = lane; i < NN; i += 64)
        u[i] = in[cw * NN + i];
      __syncthreads();
      polar_transform_lds(u, param, lane);
      for (int i = lane; i < NN; i += 64)
        out[cw * NN + i] = u[i];
      break;
    }
    case MIPHY_POLAR_OP_RATE_MATCH: // polar_rate_matcher_impl.cpp:31-106 as one gather
      for (int o = lane; o < E; o += 64)
        out[cw * E + o] = in[cw * N + p.d_tx_src[o]];
      break;
    case MIPHY_POLAR_OP_RATE_DEMATCH: { // polar_rate_dematcher_impl.cpp:29-118
      const int8_t* f = reinterpret_cast<const int8_t*>(in) + cw * E;
      for (int q = lane; q < N; q += 64) {
        const int first = p.d_rx_first[q];
        int       v;
        if (first == -1) {
          v = 0;
        } else if (first == -2) {
          v = 127;
        } else {
          v = f[p.d_rx_fidx[first]];
          for (int k = first + N; k < E; k += N)
            v = llr_promotion_sum(v, f[p.d_rx_fidx[k]]);
        }
        reinterpret_cast<int8_t*>(out)[cw * N + q] = (int8_t)v;
      }
      break;
    }
    case MIPHY_POLAR_OP_DECODE: { // polar_decoder_impl.cpp:179-350 (simplified successive cancellation), output in the u domain
      for (int q = lane; q < N; q += 64) {
        L[N + q] = reinterpret_cast<const int8_t*>(in)[cw * N + q];
        est[q]   = 0;
        u[q]     = 0;
      }
      __syncthreads();
      for (uint32_t k = 0; k < p.sched_len; ++k) {
        const uint32_t op2  = p.d_sched[k];
        const int      type = op2 & 15, s = (op2 >> 4) & 15, pos = (int)(op2 >> 8);
        const int      size = 1 << s, half = size >> 1;
        int8_t*        ls   = L + size;
        int8_t*        lc   = L + half;
        if (type == OP_F) {
          for (int i = lane; i < half; i += 64)
            lc[i] = (int8_t)llr_soft_xor(ls[i], ls[i + half]);
        } else if (type == OP_G) {
          for (int i = lane; i < half; i += 64) {
            const int x = ls[i], y = ls[i + half];
            lc[i]       = (int8_t)(est[pos + i] ? llr_add(y, -x) : llr_add(y, x));
          }
        } else if (type == OP_R1) {
          for (int i = lane; i < size; i += 64) {
            const uint8_t b = ls[i] <= 0;
            est[pos + i]    = b;
            u[pos + i]      = b;
          }
          __syncthreads();
          for (int h = 1; h < size; h <<= 1) {
            for (int t = lane; t < half; t += 64) {
              const int b = ((t / h) * 2 * h) + (t % h);
              u[pos + b] ^= u[pos + b + h];
            }
            __syncthreads();
          }
        } else {
          for (int i = lane; i < half; i += 64)
            est[pos + i] ^= est[pos + half + i];
        }
        __syncthreads();
      }
      for (int i = lane; i < N; i += 64)
        out[cw * N + i] = u[i];
      break;
    }
    case MIPHY_POLAR_OP_DEALLOCATE: // polar_deallocator_impl.cpp:27-42
      if (p.nPC == 0) {
        for (int i = lane; i < K; i += 64)
          out[cw * K + i] = in[cw * N + p.d_info_pos[i]];
      } else if (lane == 0) {
        int iK = 0;
        for (int i = 0; i < KP; ++i)
          if (!p.d_is_pc[i])
            out[cw * K + iK++] = in[cw * N + p.d_info_pos[i]];
      }
      break;
    default: { // MIPHY_POLAR_OP_INTERLEAVE_TX / _RX: polar_interleaver_impl.cpp:27-56, param = K
      const int KK = param;
      if (lane == 0) { // K <= 164: the selection of the pattern entries is a serial scan
        int k = 0;
        for (int m = 0; m < (int)NR_POLAR_K_MAX_IL; ++m)
          if ((int)pi_il_max[m] >= (int)NR_POLAR_K_MAX_IL - KK) {
            const int pi_k = (int)pi_il_max[m] - ((int)NR_POLAR_K_MAX_IL - KK);
            if (op == MIPHY_POLAR_OP_INTERLEAVE_TX)
              out[cw * KK + k] = in[cw * KK + pi_k];
            else
              out[cw * KK + pi_k] = in[cw * KK + k];
            ++k;
          }
      }
      break;
    }
  }
}

} // namespace

extern "C" int miphy_polar_code_info(const miphy_polar_code* code, uint32_t* n, uint32_t* N, uint32_t* nPC)
{
  if (!code) {
    miphy_set_error("miphy_polar_code_info: null argument");
    return MIPHY_EINVAL;
  }
  host_code h;
  int       rc = build_code(code, h);
  if (rc)
    return rc;
  if (n)
    *n = h.n;
  if (N)
    *N = h.N;
  if (nPC)
    *nPC = h.nPC;
  return MIPHY_OK;
}

extern "C" int miphy_polar_encode_batch(miphy_ctx*              ctx,
                                        const miphy_polar_code* code,
                                        uint32_t                n,
                                        const uint8_t*          msg,
                                        uint8_t*                rm_out,
                                        uint8_t*                allocated_tap,
                                        uint8_t*                encoded_tap,
                                        void*                   stream)
{
  MIPHY_REQUIRE(ctx && code && msg && rm_out, "miphy_polar_encode_batch: null argument");
  const polar_plan* p  = nullptr;
  int               rc = get_plan(ctx, code, &p);
  if (rc || n == 0)
    return rc;
  hipLaunchKernelGGL(polar_encode_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, *p, msg, rm_out, allocated_tap, encoded_tap);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

extern "C" int miphy_polar_decode_batch(miphy_ctx*              ctx,
                                        const miphy_polar_code* code,
                                        uint32_t                n,
                                        const int8_t*           llr,
                                        uint8_t*                msg_out,
                                        int8_t*                 dematched_tap,
                                        uint8_t*                decoded_u_tap,
                                        void*                   stream)
{
  MIPHY_REQUIRE(ctx && code && llr && msg_out, "miphy_polar_decode_batch: null argument");
  const polar_plan* p  = nullptr;
  int               rc = get_plan(ctx, code, &p);
  if (rc || n == 0)
    return rc;
  // Codewords per wavefront: four (16 lanes each) once the batch fills the chip several times over, fewer for small batches (a lone codeword
  // keeps the whole wavefront: its wide stages finish in fewer steps).
  const uint32_t per_chip = (uint32_t)ctx->num_cus * 8u;
  if (n >= 4u * per_chip)
    hipLaunchKernelGGL((polar_decode_kernel<POLAR_SSC_CW>), dim3((n + POLAR_SSC_CW - 1) / POLAR_SSC_CW), dim3(64), 0, (hipStream_t)stream, *p, n, llr, msg_out, dematched_tap, decoded_u_tap);
  else if (n >= 2u * per_chip)
    hipLaunchKernelGGL((polar_decode_kernel<2>), dim3((n + 1) / 2), dim3(64), 0, (hipStream_t)stream, *p, n, llr, msg_out, dematched_tap, decoded_u_tap);
  else
    hipLaunchKernelGGL((polar_decode_kernel<1>), dim3(n), dim3(64), 0, (hipStream_t)stream, *p, n, llr, msg_out, dematched_tap, decoded_u_tap);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

extern "C" int miphy_polar_block_batch(miphy_ctx* ctx, const miphy_polar_code* code, uint32_t op, uint32_t param, uint32_t n, const void* in, void* out,
                                       void* stream)
{
  MIPHY_REQUIRE(ctx && in && out, "miphy_polar_block_batch: null argument");
  MIPHY_REQUIRE(op <= MIPHY_POLAR_OP_INTERLEAVE_RX, "miphy_polar_block_batch: invalid operation %u", op);
  polar_plan        none = {};
  const polar_plan* p    = &none;
  const bool        needs_code = !(op == MIPHY_POLAR_OP_ENCODE || op == MIPHY_POLAR_OP_INTERLEAVE_TX || op == MIPHY_POLAR_OP_INTERLEAVE_RX);
  if (needs_code) {
    MIPHY_REQUIRE(code, "miphy_polar_block_batch: operation %u needs the code", op);
    int rc = get_plan(ctx, code, &p);
    if (rc)
      return rc;
  } else if (op == MIPHY_POLAR_OP_ENCODE) {
    MIPHY_REQUIRE(param >= 5 && param <= 10, "miphy_polar_block_batch: code size 2^%u out of range", param); // polar_code.h:52-61
  } else {
    MIPHY_REQUIRE(param >= 1 && param <= NR_POLAR_K_MAX_IL, "miphy_polar_block_batch: interleaver length %u out of range (K_MAX_IL = 164)", param);
  }
  if (n == 0)
    return MIPHY_OK;
  // the interleaver pattern (TS 38.212 Table 5.3.1.1-1) lives with the device tables of the context
  static_assert(sizeof(NR_POLAR_PI_IL_MAX[0]) == 1, "pattern entries are bytes");
  auto& ext = *ctx->ext;
  if (!ext.d_pi_il_max) {
    MIPHY_HIP_CHECK(hipMalloc(&ext.d_pi_il_max, NR_POLAR_K_MAX_IL));
    MIPHY_HIP_CHECK(hipMemcpy(ext.d_pi_il_max, NR_POLAR_PI_IL_MAX, NR_POLAR_K_MAX_IL, hipMemcpyHostToDevice));
    ext.to_free.push_back(ext.d_pi_il_max);
  }
  hipLaunchKernelGGL(polar_block_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, *p, (int)op, (int)param, (const uint8_t*)in, (uint8_t*)out,
                     (const uint8_t*)ext.d_pi_il_max);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

extern "C" int miphy_pdcch_encode_batch(miphy_ctx*      ctx,
                                        uint32_t        A,
                                        uint32_t        E,
                                        uint32_t        n,
                                        const uint8_t*  payload,
                                        const uint16_t* rnti,
                                        uint8_t*        out,
                                        void*           stream)
{
  MIPHY_REQUIRE(ctx && payload && rnti && out, "miphy_pdcch_encode_batch: null argument");
  MIPHY_REQUIRE(A >= 12 && A <= 128, "pdcch_encode: payload size %u out of range (12..MAX_DCI_PAYLOAD_SIZE = 128)", A);
  miphy_polar_code  code = {A + 24, E, 9, 0};
  const polar_plan* p    = nullptr;
  int               rc   = get_plan(ctx, &code, &p);
  if (rc || n == 0)
    return rc;
  hipLaunchKernelGGL(pdcch_encode_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, *p, A, payload, rnti, out, 24);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

// ---------------------------------------------------------------------------------------------------- SCL list decoder
namespace {

// One wavefront per codeword. LDS: channel LLRs [N] + two banks (current / scratch) of L paths x {llr[N], bl[N], u[N]}.
//   llr: stage s node LLRs at offset 2^s (s < n), bl: left-child partial sums of stage s at offset 2^s, u: decisions.
// All L paths advance together: a stage of size 2^s over `active` paths is one flat loop over active * 2^s lanes' worth of
// elements; forking ranks the 2 * active candidates with wavefront shuffles and copies the survivors bank to bank.
__global__ void __launch_bounds__(64) polar_scl_kernel(polar_plan p, int L, int crc_mode, const int8_t* __restrict__ llr_in,
                                                       const uint16_t* __restrict__ rnti, const uint8_t* __restrict__ k_set,
                                                       uint8_t* __restrict__ msg_out, uint8_t* __restrict__ crc_ok_out,
                                                       int32_t* __restrict__ metric_out)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int    lane = threadIdx.x;
  const size_t cw   = blockIdx.x;
  const int    N = (int)p.N, n = (int)p.n, E = (int)p.E, K = (int)p.K;
  int8_t*      ch     = reinterpret_cast<int8_t*>(smem);
  const int    PSZ    = 3 * N; // bytes per path
  uint8_t*     bankA  = smem + N;
  uint8_t*     bankB  = bankA + (size_t)L * PSZ;
  int*         pm     = reinterpret_cast<int*>(bankB + (size_t)L * PSZ); // [L]
  int*         sel    = pm + 8;                                           // parent[8], bit[8], pm[8]
  uint8_t*     kset   = reinterpret_cast<uint8_t*>(sel + 24);             // [N] information-set flags: one LDS read per leaf
                                                                          // instead of a dependent global load (~1.5 us each)
  const int8_t* f_in  = llr_in + cw * p.E;
  for (int q = lane; q < 2 * N; q += 64)
    kset[q] = k_set[q]; // [0, N): information-set flags, [N, 2N): rate-0 block exponents
  // Rate dematching (same gather as polar_decode_kernel).
  for (int q = lane; q < N; q += 64) {
    const int first = p.d_rx_first[q];
    int       v;
    if (first == -1) {
      v = 0;
    } else if (first == -2) {
      v = 127;
    } else {
      v = f_in[p.d_rx_fidx[first]];
      for (int k = first + N; k < E; k += N)
        v = llr_promotion_sum(v, f_in[p.d_rx_fidx[k]]);
    }
    ch[q] = (int8_t)v;
  }
  if (lane < 8)
    pm[lane] = 0;
  __syncthreads();
  uint8_t* P = bankA;
  uint8_t* Q = bankB;
  int      active = 1;
  for (int i = 0; i < N;) {
    // An aligned all-frozen block [i, i + 2^r) (rate-0 node) is processed at stage r in one step: its penalty is the sum of the
    // negative stage-r LLRs, its bits and partial sums are zero.
    const int r = kset[N + i], B = 1 << r;
    // ---- stage-r LLRs
    int t = n;
    if (i != 0) {
      t = __ffs(i) - 1;
      const int sz = 1 << t;
      for (int idx = lane; idx < active * sz; idx += 64) {
        const int     q = idx >> t, j = idx & (sz - 1);
        uint8_t*      a  = P + q * PSZ;
        const int8_t* up = (t + 1 == n) ? ch : reinterpret_cast<int8_t*>(a) + 2 * sz;
        const int     x = up[j], y = up[j + sz];
        reinterpret_cast<int8_t*>(a)[sz + j] = (int8_t)(a[N + sz + j] ? llr_add(y, -x) : llr_add(y, x));
      }
      __syncthreads();
    }
    for (int s = t - 1; s >= r; --s) {
      const int sz = 1 << s;
      for (int idx = lane; idx < active * sz; idx += 64) {
        const int     q = idx >> s, j = idx & (sz - 1);
        int8_t*       a  = reinterpret_cast<int8_t*>(P + q * PSZ);
        const int8_t* up = (s + 1 == n) ? ch : a + 2 * sz;
        a[sz + j]        = (int8_t)llr_soft_xor(up[j], up[j + sz]);
      }
      __syncthreads();
    }
    // ---- decision
    if (!kset[i]) {
      for (int idx = lane; idx < active * B; idx += 64) {
        const int q = idx >> r, j = idx & (B - 1);
        const int v = (r == n) ? ch[j] : reinterpret_cast<int8_t*>(P + q * PSZ)[B + j];
        P[q * PSZ + 2 * N + i + j] = 0;
        if (v < 0)
          atomicAdd(&pm[q], -v);
      }
      __syncthreads();
    } else {
      const int nc = 2 * active, keep = min(nc, L);
      int       met = 0x7fffffff, bit = 0;
      if (lane < nc) {
        const int q = lane >> 1, l0 = reinterpret_cast<int8_t*>(P + q * PSZ)[1];
        const int hard = l0 <= 0, al = abs(l0);
        met = pm[q] + ((lane & 1) ? al : 0);
        bit = (lane & 1) ? !hard : hard;
      }
      int rank = 0;
      for (int o = 0; o < nc; ++o) {
        const int mo = __shfl(met, o);
        rank += (mo < met) || (mo == met && o < lane);
      }
      __syncthreads(); // pm[] has been read by everyone
      if (lane < nc && rank < keep) {
        sel[rank]      = lane >> 1;
        sel[8 + rank]  = bit;
        pm[rank]       = met;
      }
      __syncthreads();
      // survivors: bank P (parent) -> bank Q (slot), 16 bytes per lane per step
      const int vec_per_path = PSZ >> 4;
      for (int idx = lane; idx < keep * vec_per_path; idx += 64) {
        const int r = idx / vec_per_path, v = idx - r * vec_per_path;
        reinterpret_cast<uint4*>(Q + r * PSZ)[v] = reinterpret_cast<const uint4*>(P + sel[r] * PSZ)[v];
      }
      __syncthreads();
      if (lane < keep)
        Q[lane * PSZ + 2 * N + i] = (uint8_t)sel[8 + lane];
      uint8_t* tmp = P;
      P            = Q;
      Q            = tmp;
      active       = keep;
      __syncthreads();
    }
    // ---- partial sums of the finished block at stage r (bank Q's u area serves as the per-path working vector)
    if (!((i >> r) & 1)) {
      for (int idx = lane; idx < active * B; idx += 64) {
        const int q = idx >> r, j = idx & (B - 1);
        P[q * PSZ + N + B + j] = P[q * PSZ + 2 * N + i + j];
      }
    } else {
      for (int idx = lane; idx < active * B; idx += 64) {
        const int q = idx >> r, j = idx & (B - 1);
        Q[q * PSZ + 2 * N + j] = P[q * PSZ + 2 * N + i + j];
      }
      __syncthreads();
      int sz = B, s = r;
      while (s < n && ((i >> s) & 1)) {
        for (int idx = lane; idx < active * sz; idx += 64) {
          const int q = idx >> s, j = idx & (sz - 1);
          uint8_t*  cur = Q + q * PSZ + 2 * N;
          const uint8_t c0 = cur[j];
          cur[sz + j]      = c0;
          cur[j]           = c0 ^ P[q * PSZ + N + sz + j];
        }
        __syncthreads();
        sz <<= 1;
        ++s;
      }
      if (s < n) {
        for (int idx = lane; idx < active * sz; idx += 64) {
          const int q = idx >> s, j = idx & (sz - 1);
          P[q * PSZ + N + sz + j] = Q[q * PSZ + 2 * N + j];
        }
      }
    }
    __syncthreads();
    i += B;
  }
  // ---- selection: extract the K bits of every path (into bank Q), optional de-interleave + CRC, best metric wins
  int      my_pm = 0x7fffffff, my_ok = 0;
  uint8_t* cand  = Q + (lane < 8 ? lane : 0) * PSZ;     // K bits at cand[0..K), K-set order at cand[N..N+K)
  if (lane < active) {
    const uint8_t* u  = P + lane * PSZ + 2 * N;
    int            iK = 0;
    for (int i = 0; i < (int)(p.K + p.nPC); ++i)
      if (!p.d_is_pc[i])
        cand[N + iK++] = u[p.d_info_pos[i]];
    my_pm = pm[lane];
    if (crc_mode == 0) {
      for (int k = 0; k < K; ++k)
        cand[k] = cand[N + k];
    } else {
      for (int k = 0; k < K; ++k) // polar_interleaver, rx direction: out[pi(k)] = in[k]
        cand[p.d_pi_il[k]] = cand[N + k];
      const int A    = K - 24;
      uint32_t  reg  = 0;
      const int ones = (crc_mode == 1) ? 24 : 0;
      for (int b = 0; b < ones + A; ++b) {
        const uint32_t bitv = (b < ones) ? 1u : cand[b - ones];
        reg                 = (reg << 1) ^ (bitv << 24);
        reg ^= (reg & 0x1000000u) ? 0x1B2B117u : 0u;
      }
      uint32_t rx = 0;
      for (int b = 0; b < 24; ++b)
        rx = (rx << 1) | cand[A + b];
      if (crc_mode == 1)
        rx ^= (uint32_t)rnti[cw];
      my_ok = (reg & 0xffffffu) == rx;
    }
  }
  // winner: smallest metric among CRC passes (if any), else smallest metric; ties -> lowest slot
  const unsigned long long okmask = __ballot(my_ok != 0);
  const bool               any_ok = okmask != 0ull;
  int                      key    = (lane < active && (!any_ok || my_ok)) ? my_pm : 0x7fffffff;
  int                      best   = key, best_lane = lane;
#pragma unroll
  for (int off = 4; off >= 1; off >>= 1) {
    const int ok = __shfl_xor(best, off), ol = __shfl_xor(best_lane, off);
    if (ok < best || (ok == best && ol < best_lane)) {
      best      = ok;
      best_lane = ol;
    }
  }
  best_lane = __shfl(best_lane, 0);
  best      = __shfl(best, 0);
  __syncthreads();
  const uint8_t* win = Q + best_lane * PSZ;
  for (int k = lane; k < K; k += 64)
    msg_out[cw * p.K + k] = win[k];
  if (lane == 0) {
    crc_ok_out[cw] = any_ok ? 1 : 0;
    if (metric_out)
      metric_out[cw] = best;
  }
}

} // namespace

extern "C" int miphy_polar_decode_list_batch(miphy_ctx*              ctx,
                                             const miphy_polar_code* code,
                                             uint32_t                list_size,
                                             uint32_t                crc_mode,
                                             uint32_t                n,
                                             const int8_t*           llr,
                                             const uint16_t*         rnti,
                                             uint8_t*                msg_out,
                                             uint8_t*                crc_ok_out,
                                             int32_t*                metric_out,
                                             void*                   stream)
{
  MIPHY_REQUIRE(ctx && code && llr && msg_out && crc_ok_out, "miphy_polar_decode_list_batch: null argument");
  MIPHY_REQUIRE(list_size == 1 || list_size == 2 || list_size == 4 || list_size == 8, "polar_decode_list: list size %u not in {1,2,4,8}", list_size);
  MIPHY_REQUIRE(crc_mode <= 2, "polar_decode_list: invalid crc_mode %u", crc_mode);
  MIPHY_REQUIRE(crc_mode != 1 || rnti, "polar_decode_list: PDCCH CRC mode needs the RNTI array");
  MIPHY_REQUIRE(crc_mode == 0 || (code->K > 24 && code->K <= 164), "polar_decode_list: CRC-aided modes need 24 < K <= 164");
  const polar_plan* p  = nullptr;
  int               rc = get_plan(ctx, code, &p);
  if (rc || n == 0)
    return rc;
  // K-set membership per position (frozen / information), uploaded once per plan.
  auto key = std::make_tuple(code->K, code->E, code->nMax, code->ibil ? 1u : 0u);
  auto it  = ctx->ext->polar_kset.find(key);
  if (it == ctx->ext->polar_kset.end()) {
    host_code h;
    if ((rc = build_code(code, h)))
      return rc;
    // followed by, per position, the exponent r of the largest aligned all-frozen block [i, i + 2^r) that starts there
    std::vector<uint8_t> tab(h.k_set.begin(), h.k_set.end());
    const uint32_t       Np = (uint32_t)h.k_set.size();
    uint32_t             np = 0;
    while ((1u << np) < Np)
      ++np;
    tab.resize(2 * Np, 0);
    for (uint32_t i = 0; i < Np; ++i) {
      uint32_t r = 0;
      if (!h.k_set[i]) {
        while (r < np && (i & ((2u << r) - 1u)) == 0) {
          bool frozen = true;
          for (uint32_t j = 0; j < (2u << r) && frozen; ++j)
            frozen = !h.k_set[i + j];
          if (!frozen)
            break;
          ++r;
        }
      }
      tab[Np + i] = (uint8_t)r;
    }
    uint8_t* d = nullptr;
    if ((rc = upload(ctx, tab, &d)))
      return rc;
    it = ctx->ext->polar_kset.emplace(key, d).first;
  }
  const size_t lds = (size_t)p->N + 2 * (size_t)list_size * 3 * p->N + 8 * 4 + 24 * 4 + 2 * p->N + 64;
  // Above the default 64 KB of dynamic LDS the limit has to be raised; it is a per-device attribute of the kernel, so it is set on
  // every such launch (a cache per thread would be wrong for a thread that drives several devices).
  if (lds > 48 * 1024) {
    MIPHY_HIP_CHECK(hipFuncSetAttribute((const void*)polar_scl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL(polar_scl_kernel, dim3(n), dim3(64), lds, (hipStream_t)stream, *p, (int)list_size, (int)crc_mode, llr, rnti, it->second, msg_out,
                     crc_ok_out, metric_out);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}

namespace {
// TS 38.211 5.2.1 Gold sequence bit c(n) on the host (PBCH scrambling needs at most a few hundred bits).
void host_gold(uint32_t c_init, uint32_t offset, uint32_t nbits, uint8_t* out)
{
  const uint32_t       total = 1600 + offset + nbits + 31;
  std::vector<uint8_t> x1(total + 31, 0), x2(total + 31, 0);
  x1[0] = 1;
  for (int i = 0; i < 31; ++i)
    x2[i] = (c_init >> i) & 1u;
  for (uint32_t n = 0; n + 31 < total + 31; ++n) {
    x1[n + 31] = x1[n + 3] ^ x1[n];
    x2[n + 31] = x2[n + 3] ^ x2[n + 2] ^ x2[n + 1] ^ x2[n];
  }
  for (uint32_t n = 0; n < nbits; ++n)
    out[n] = x1[n + offset + 1600] ^ x2[n + offset + 1600];
}
} // namespace

extern "C" int miphy_pbch_encode_batch(miphy_ctx* ctx, const miphy_pbch_msg* msgs, uint32_t n, uint8_t* out, void* stream)
{
  MIPHY_REQUIRE(ctx && msgs && out, "miphy_pbch_encode_batch: null argument");
  if (n == 0)
    return MIPHY_OK;
  // TS 38.212 Table 7.1.1-1, payload interleaver pattern G(j) (pbch_encoder_impl.cpp:33-35).
  static const uint8_t G[32] = {16, 23, 18, 17, 8, 30, 10, 6, 24, 7, 0, 5, 3, 2, 1, 4, 9, 11, 12, 13, 14, 15, 19, 20, 21, 22, 25, 26, 27, 28, 29, 31};
  std::vector<uint8_t> ap((size_t)n * 32);
  for (uint32_t m = 0; m < n; ++m) {
    const miphy_pbch_msg& msg = msgs[m];
    MIPHY_REQUIRE(msg.N_id <= 1007, "pbch_encode: message %u: invalid N_id %u", m, msg.N_id);
    MIPHY_REQUIRE(msg.L_max == 4 || msg.L_max == 8 || msg.L_max == 64, "pbch_encode: message %u: invalid L_max %u", m, msg.L_max);
    uint8_t a[32] = {0};
    // payload_generate (pbch_encoder_impl.cpp:42-86)
    uint32_t j_sfn = 0, j_other = 14;
    for (uint32_t i = 0; i < 24; ++i) {
      if (i >= 1 && i < 7)
        a[G[j_sfn++]] = msg.payload[i] & 1u;
      else
        a[G[j_other++]] = msg.payload[i] & 1u;
    }
    a[G[j_sfn++]] = (msg.sfn >> 3) & 1u;
    a[G[j_sfn++]] = (msg.sfn >> 2) & 1u;
    a[G[j_sfn++]] = (msg.sfn >> 1) & 1u;
    a[G[j_sfn++]] = (msg.sfn >> 0) & 1u;
    a[G[10]]      = msg.hrf ? 1 : 0;
    if (msg.L_max == 64) {
      a[G[11]] = (msg.ssb_idx >> 5) & 1u;
      a[G[12]] = (msg.ssb_idx >> 4) & 1u;
      a[G[13]] = (msg.ssb_idx >> 3) & 1u;
    } else {
      a[G[11]] = (msg.k_ssb >> 4) & 1u;
      a[G[12]] = 0;
      a[G[13]] = 0;
    }
    // scramble (pbch_encoder_impl.cpp:88-129)
    const uint32_t M = (msg.L_max == 64) ? 32 - 6 : 32 - 3;
    const uint32_t v = 2 * a[G[7]] + a[G[8]]; // 3rd and 2nd LSB of the SFN
    uint8_t        c[32];
    host_gold(msg.N_id, M * v, 32, c);
    uint32_t j = 0;
    for (uint32_t i = 0; i < 32; ++i) {
      uint8_t    s_i        = c[j];
      const bool is_ssb_idx = (i == G[11] || i == G[12] || i == G[13]) && msg.L_max == 64;
      if (is_ssb_idx || i == G[10] || i == G[8] || i == G[7])
        s_i = 0;
      else
        ++j;
      ap[(size_t)m * 32 + i] = a[i] ^ s_i;
    }
  }
  miphy_polar_code  code = {56, 864, 9, 0};
  const polar_plan* p    = nullptr;
  int               rc   = get_plan(ctx, &code, &p);
  if (rc)
    return rc;
  hipStream_t s  = (hipStream_t)stream;
  void*       ws = nullptr;
  if ((rc = miphy_get_workspace(ctx, ap.size(), s, &ws)))
    return rc;
  if ((rc = miphy_upload(ctx, ws, ap.data(), ap.size(), s))) // `ap` is a local buffer
    return rc;
  hipLaunchKernelGGL(pdcch_encode_kernel, dim3(n), dim3(64), 0, s, *p, 32u, (const uint8_t*)ws, (const uint16_t*)nullptr, out, 0);
  MIPHY_HIP_CHECK(hipGetLastError());
  return MIPHY_OK;
}
