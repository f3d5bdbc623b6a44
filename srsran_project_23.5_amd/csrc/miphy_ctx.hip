// Context: device tables (TS 38.212 graphs, CRC constants), descriptor staging, error reporting.
#include "miphy_internal.h"
#include "miphy_ext.h"
#include "tables/nr_ldpc_tables.h"
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <new>

static thread_local char g_err[512] = "";

void miphy_set_error(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* miphy_last_error(void)
{
  return g_err;
}

extern "C" int miphy_version(void)
{
  return MIPHY_VERSION;
}

static int lifting_set(int Z)
{
  static const int A[8] = {2, 3, 5, 7, 9, 11, 13, 15};
  for (int i = 7; i >= 0; --i) {
    if (Z % A[i])
      continue;
    int q = Z / A[i];
    if ((q & (q - 1)) == 0)
      return i;
  }
  return -1;
}

static uint32_t gf2_mulmod(uint32_t a, uint32_t b, uint32_t poly, unsigned order)
{
  // a*b mod poly over GF(2), operands < 2^order.
  uint32_t r = 0, top = 1u << order;
  for (int i = (int)order - 1; i >= 0; --i) {
    r <<= 1;
    if (r & top)
      r ^= poly;
    if ((b >> i) & 1u)
      r ^= a;
  }
  return r;
}

static void build_tables(miphy_graph_tables* t)
{
  memset(t, 0, sizeof(*t));
  for (int z = 0; z <= MIPHY_MAX_Z; ++z) {
    t->z_pos[z] = 0xffff;
    t->i_ls[z]  = 0xff;
  }
  for (int p = 0; p < MIPHY_NOF_Z; ++p) {
    int Z        = NR_LDPC_LIFTING_SIZES[p];
    t->z_pos[Z]  = (uint16_t)p;
    int ils      = lifting_set(Z);
    t->i_ls[Z]   = (uint8_t)ils;
    for (int e = 0; e < NR_LDPC_BG1_NOF_EDGES; ++e)
      t->edge[0][p][e] = (uint32_t)(NR_LDPC_BG1_COL[e] * Z) | ((uint32_t)(NR_LDPC_BG1_SHIFT[ils][e] % Z) << 16);
    for (int e = 0; e < NR_LDPC_BG2_NOF_EDGES; ++e)
      t->edge[1][p][e] = (uint32_t)(NR_LDPC_BG2_COL[e] * Z) | ((uint32_t)(NR_LDPC_BG2_SHIFT[ils][e] % Z) << 16);
  }
  for (int b = 0; b < 2; ++b)
    for (int p = 0; p < MIPHY_NOF_Z; ++p)
      for (int e = 0; e < (b ? NR_LDPC_BG2_NOF_EDGES : NR_LDPC_BG1_NOF_EDGES); ++e) {
        t->edge_sb[b][p][2 * e]     = t->edge[b][p][e] >> 16;
        t->edge_sb[b][p][2 * e + 1] = t->edge[b][p][e] & 0xffffu;
      }
  for (int m = 0; m <= NR_LDPC_BG1_M; ++m)
    t->row_start[0][m] = NR_LDPC_BG1_ROW_START[m];
  for (int m = 0; m <= NR_LDPC_BG2_M; ++m)
    t->row_start[1][m] = NR_LDPC_BG2_ROW_START[m];
  for (int b = 0; b < 2; ++b) {
    t->pair_start[b][0] = 0;
    for (int m = 0; m < (b ? NR_LDPC_BG2_M : NR_LDPC_BG1_M); ++m)
      t->pair_start[b][m + 1] = (uint16_t)(t->pair_start[b][m] + (t->row_start[b][m + 1] - t->row_start[b][m] + 1) / 2);
  }
  static const uint32_t POLY[5]  = {0x1864CFB, 0x1800063, 0x1B2B117, 0x11021, 0xE21};
  static const uint32_t ORDER[5] = {24, 24, 24, 16, 11};
  for (int p = 0; p < 5; ++p) {
    t->crc_poly[p]  = POLY[p];
    t->crc_order[p] = ORDER[p];
    // x^32 mod poly
    uint32_t x32 = 1, top = 1u << ORDER[p];
    for (int i = 0; i < 32; ++i) {
      x32 <<= 1;
      if (x32 & top)
        x32 ^= POLY[p];
    }
    uint32_t v = 1;
    for (int k = 0; k < 320; ++k) {
      t->crc_pow32[p][k] = v;
      v                  = gf2_mulmod(v, x32, POLY[p], ORDER[p]);
    }
    {
      const uint32_t x8192 = gf2_mulmod(t->crc_pow32[p][255], x32, POLY[p], ORDER[p]); // x^(32*256)
      uint32_t       h     = 1;
      for (int k = 0; k < 256; ++k) {
        t->crc_pow32_hi[p][k] = h;
        h                     = gf2_mulmod(h, x8192, POLY[p], ORDER[p]);
      }
    }
    v = x32;
    for (int b = 0; b < 24; ++b) {
      t->crc_pow2[p][b] = v;
      v                 = gf2_mulmod(v, v, POLY[p], ORDER[p]);
    }
    const int zi = miphy_crc_zmask_index(p);
    if (zi >= 0) {
      // weight of the bit at distance j from the end of the message: x^(j + order) mod P
      uint32_t c = 1;
      for (unsigned i = 0; i < ORDER[p]; ++i) {
        c <<= 1;
        if (c & top)
          c ^= POLY[p];
      }
      for (int u = 0; u < MIPHY_CRC_ZMASK_WORDS; ++u)
        for (int jl = 0; jl < 32; ++jl) { // j = 32 u + jl
          const int i_local = 31 - jl, q = i_local >> 2, b = i_local & 3;
          for (unsigned k = 0; k < ORDER[p]; ++k)
            if ((c >> k) & 1u) {
              t->crc_zmask[zi][u][k] |= 1u << (q + 8 * b);
              if (p == 0)
                t->crc_zmask_packed24a[u][k] |= 1u << (8 * (i_local >> 3) + 7 - (i_local & 7));
            }
          c <<= 1;
          if (c & top)
            c ^= POLY[p];
        }
    }
  }
}

extern "C" int miphy_create(int device, miphy_ctx** out)
{
  if (!out) {
    miphy_set_error("miphy_create: null out pointer");
    return MIPHY_EINVAL;
  }
  *out = nullptr;
  MIPHY_HIP_CHECK(hipSetDevice(device));
  miphy_ctx* c = new (std::nothrow) miphy_ctx();
  if (!c)
    return MIPHY_ENOMEM;
  c->device   = device;
  c->num_cus  = 256;
  (void)hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, device);
  c->ext      = new miphy_ctx_ext();
  c->h_tables = (miphy_graph_tables*)malloc(sizeof(miphy_graph_tables));
  build_tables(c->h_tables);
  MIPHY_HIP_CHECK(hipMalloc((void**)&c->d_tables, sizeof(miphy_graph_tables)));
  MIPHY_HIP_CHECK(hipMemcpy(c->d_tables, c->h_tables, sizeof(miphy_graph_tables), hipMemcpyHostToDevice));
  c->desc_staging_bytes = 8u << 20;
  c->staging_head       = 0;
  MIPHY_HIP_CHECK(hipMalloc(&c->d_desc_staging, c->desc_staging_bytes));
  MIPHY_HIP_CHECK(hipHostMalloc(&c->h_desc_staging, c->desc_staging_bytes, hipHostMallocDefault));
  MIPHY_HIP_CHECK(hipMalloc((void**)&c->d_queue, MIPHY_NOF_QUEUE_COUNTERS * sizeof(uint32_t)));
  MIPHY_HIP_CHECK(hipMemset(c->d_queue, 0, MIPHY_NOF_QUEUE_COUNTERS * sizeof(uint32_t)));
  c->queue_next = 0;
  *out = c;
  return MIPHY_OK;
}

extern "C" void miphy_destroy(miphy_ctx* c)
{
  if (!c)
    return;
  (void)hipSetDevice(c->device);
  (void)hipFree(c->d_tables);
  for (void* p : c->ext->to_free)
    (void)hipFree(p);
  delete c->ext;
  for (int k = 0; k < MIPHY_NOF_SIDE_STREAMS; ++k) {
    if (c->side_stream[k])
      (void)hipStreamDestroy((hipStream_t)c->side_stream[k]);
    if (c->ev_join[k])
      (void)hipEventDestroy((hipEvent_t)c->ev_join[k]);
  }
  if (c->ev_fork)
    (void)hipEventDestroy((hipEvent_t)c->ev_fork);
  (void)hipFree(c->d_desc_staging);
  (void)hipFree(c->d_queue);
  for (void* w : c->d_work)
    (void)hipFree(w);
  (void)hipHostFree(c->h_desc_staging);
  free(c->h_tables);
  delete c;
}

int miphy_side_streams(miphy_ctx* ctx)
{
  if (ctx->ev_fork)
    return MIPHY_OK;
  for (int k = 0; k < MIPHY_NOF_SIDE_STREAMS; ++k) {
    hipStream_t st = nullptr;
    hipEvent_t  e  = nullptr;
    MIPHY_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    MIPHY_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->side_stream[k] = st, ctx->ev_join[k] = e;
  }
  hipEvent_t f = nullptr;
  MIPHY_HIP_CHECK(hipEventCreateWithFlags(&f, hipEventDisableTiming));
  ctx->ev_fork = f;
  return MIPHY_OK;
}

int miphy_next_queue_counter(miphy_ctx* ctx, uint32_t** out)
{
  // A launch owns its counter until MIPHY_NOF_QUEUE_COUNTERS later launches of this context have been enqueued. The counters are
  // zero when the context is created and every launch leaves its counter at zero (the workgroup that draws the last ticket).
  *out = ctx->d_queue + (ctx->queue_next++ % MIPHY_NOF_QUEUE_COUNTERS);
  return MIPHY_OK;
}

// Front-to-back allocation in the staging ring. Regions handed out since the last wrap are never overwritten; on a wrap every
// consumer of the old regions must be done. The consumers run on the streams the regions were staged for, so the wrap waits for
// THOSE streams (up to four are remembered between wraps; more than that falls back to the device-wide wait) -- not for the other
// contexts and cells that share the device -- once per desc_staging_bytes of descriptors, not once per call. A stream that is being
// captured cannot be waited for: the wrap then fails with MIPHY_EUNSUPP (capture with device-resident descriptors or a prepared plan).
static int staging_take(miphy_ctx* ctx, size_t bytes, hipStream_t s, size_t* off)
{
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (ctx->staging_head + need > ctx->desc_staging_bytes) {
    if (ctx->nof_ring_streams > 4) {
      MIPHY_HIP_CHECK(hipDeviceSynchronize());
    } else {
      for (int i = 0; i < ctx->nof_ring_streams; ++i) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (ctx->ring_streams[i] && hipStreamIsCapturing((hipStream_t)ctx->ring_streams[i], &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
          miphy_set_error("descriptor staging ring wrapped while a stream that consumes it is being captured");
          return MIPHY_EUNSUPP;
        }
        MIPHY_HIP_CHECK(hipStreamSynchronize((hipStream_t)ctx->ring_streams[i]));
      }
    }
    ctx->staging_head     = 0;
    ctx->nof_ring_streams = 0;
  }
  int k = 0;
  while (k < ctx->nof_ring_streams && k < 4 && ctx->ring_streams[k] != (void*)s)
    ++k;
  if (k == ctx->nof_ring_streams || k == 4) {
    if (k < 4)
      ctx->ring_streams[k] = (void*)s;
    ctx->nof_ring_streams = k < 4 ? k + 1 : 5;
  }
  *off = ctx->staging_head;
  ctx->staging_head += need;
  return MIPHY_OK;
}

int miphy_stage_descs(miphy_ctx* ctx, const void* descs, int on_device, size_t bytes, hipStream_t s, const void** out)
{
  if (on_device) {
    *out = descs;
    return MIPHY_OK;
  }
  MIPHY_REQUIRE(bytes <= ctx->desc_staging_bytes, "descriptor batch too large (%zu bytes > %zu)", bytes, ctx->desc_staging_bytes);
  size_t off = 0;
  int    rc  = staging_take(ctx, bytes, s, &off);
  if (rc)
    return rc;
  uint8_t* h = static_cast<uint8_t*>(ctx->h_desc_staging) + off;
  uint8_t* d = static_cast<uint8_t*>(ctx->d_desc_staging) + off;
  memcpy(h, descs, bytes);
  MIPHY_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
  *out = d;
  return MIPHY_OK;
}

int miphy_upload(miphy_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s)
{
  if (bytes == 0)
    return MIPHY_OK;
  if (bytes > ctx->desc_staging_bytes / 2) {
    MIPHY_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    MIPHY_HIP_CHECK(hipStreamSynchronize(s)); // `src` may be pageable or about to go out of scope
    return MIPHY_OK;
  }
  size_t off = 0;
  int    rc  = staging_take(ctx, bytes, s, &off);
  if (rc)
    return rc;
  uint8_t* h = static_cast<uint8_t*>(ctx->h_desc_staging) + off;
  memcpy(h, src, bytes);
  MIPHY_HIP_CHECK(hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, s));
  return MIPHY_OK;
}

int miphy_get_workspace(miphy_ctx* ctx, size_t bytes, hipStream_t s, void** out, int which)
{
  if (bytes > ctx->work_bytes[which]) {
    MIPHY_HIP_CHECK(hipStreamSynchronize(s));
    if (ctx->d_work[which])
      MIPHY_HIP_CHECK(hipFree(ctx->d_work[which]));
    ctx->d_work[which]     = nullptr;
    ctx->work_bytes[which] = 0;
    size_t want            = bytes + bytes / 4 + (1u << 20);
    MIPHY_HIP_CHECK(hipMalloc(&ctx->d_work[which], want));
    ctx->work_bytes[which] = want;
  }
  *out = ctx->d_work[which];
  return MIPHY_OK;
}

#include "gold_device.h"
#include "miphy_ext.h"
int miphy_get_gold_tables(miphy_ctx* ctx, const gold_tables** out)
{
  if (!ctx->ext->d_gold) {
    gold_tables* t = new gold_tables;
    gold_tables_init(*t);
    hipError_t e = hipMalloc(&ctx->ext->d_gold, sizeof(gold_tables));
    if (e == hipSuccess)
      e = hipMemcpy(ctx->ext->d_gold, t, sizeof(gold_tables), hipMemcpyHostToDevice);
    delete t;
    if (e != hipSuccess) {
      miphy_set_error("gold tables: %s", hipGetErrorString(e));
      ctx->ext->d_gold = nullptr;
      return MIPHY_EHIP;
    }
    ctx->ext->to_free.push_back(ctx->ext->d_gold);
  }
  if (out)
    *out = (const gold_tables*)ctx->ext->d_gold;
  return MIPHY_OK;
}
