// Device-resident HARQ softbuffer pool: the reservation rules and state machine of srsran::rx_softbuffer_pool_impl
// (lib/phy/upper/rx_softbuffer_pool_impl.cpp:27-69) and rx_softbuffer_impl (lib/phy/upper/rx_softbuffer_impl.h:33-258) over
// the HARQ arrays the transport-block kernels work on. Host bookkeeping only; the arrays are never touched here.
#include "miphy_internal.h"
#include <mutex>
#include <new>
#include <vector>

namespace {
constexpr size_t CB_SOFT_STRIDE = 66 * 384;
constexpr size_t CB_MSG_STRIDE  = 1056;

struct softbuffer {
  uint32_t state          = MIPHY_HARQ_AVAILABLE;
  uint32_t rnti           = 0; // rx_softbuffer_identifier{} of a fresh rx_softbuffer_impl
  uint32_t harq_id        = 0;
  uint32_t expire_slot    = 0;
  uint32_t nof_codeblocks = 0;
};
} // namespace

struct miphy_harq_pool {
  miphy_harq_pool_config  cfg;
  std::mutex              mutex;
  std::vector<softbuffer> buffers;
  uint32_t                free_cbs = 0;
  int                     device   = -1;
  int8_t*                 d_soft   = nullptr;
  uint8_t*                d_msgs   = nullptr;
  uint8_t*                d_crc    = nullptr;

  // slot_point::operator<= (include/srsran/ran/slot_point.h:163-174) on counters modulo cfg.nof_slots_wrap.
  bool slot_le(uint32_t a, uint32_t b) const
  {
    if (a == b)
      return true;
    const int v = (int)b - (int)a, w = (int)cfg.nof_slots_wrap;
    return v > 0 ? v < w / 2 : v < -w / 2;
  }
  // rx_softbuffer_impl::free
  void free_buffer(softbuffer& b)
  {
    free_cbs += b.nof_codeblocks;
    b.nof_codeblocks = 0;
    b.state          = MIPHY_HARQ_AVAILABLE;
  }
  // rx_softbuffer_impl::reserve
  bool reserve(softbuffer& b, uint32_t rnti, uint32_t harq_id, uint32_t expire, uint32_t n)
  {
    if (b.state == MIPHY_HARQ_LOCKED)
      return false;
    b.rnti = rnti, b.harq_id = harq_id, b.expire_slot = expire;
    if (n == b.nof_codeblocks) {
      b.state = MIPHY_HARQ_RESERVED;
      return true;
    }
    free_buffer(b);
    if (n > free_cbs)
      return false; // the reference reserves one by one, fails on the first miss and frees what it took
    free_cbs -= n;
    b.nof_codeblocks = n;
    b.state          = MIPHY_HARQ_RESERVED;
    return true;
  }
};

extern "C" int miphy_harq_pool_create(miphy_ctx* ctx, const miphy_harq_pool_config* cfg, miphy_harq_pool** out)
{
  MIPHY_REQUIRE(cfg && out, "harq_pool_create: null argument");
  *out = nullptr;
  MIPHY_REQUIRE(cfg->max_softbuffers > 0 && cfg->max_softbuffers < (1u << 24), "harq_pool_create: max_softbuffers %u out of range", cfg->max_softbuffers);
  MIPHY_REQUIRE(cfg->nof_slots_wrap >= 2 && cfg->nof_slots_wrap <= (10240u << 4), "harq_pool_create: nof_slots_wrap %u out of range", cfg->nof_slots_wrap);
  MIPHY_REQUIRE(cfg->expire_timeout_slots < cfg->nof_slots_wrap / 2, "harq_pool_create: expire_timeout_slots %u must be below half the slot period",
                cfg->expire_timeout_slots);
  miphy_harq_pool* p = new (std::nothrow) miphy_harq_pool();
  if (!p)
    return MIPHY_ENOMEM;
  p->cfg = *cfg;
  if (p->cfg.max_codeblocks_per_buffer == 0)
    p->cfg.max_codeblocks_per_buffer = 52;
  p->buffers.resize(cfg->max_softbuffers);
  p->free_cbs = cfg->max_nof_codeblocks;
  if (ctx) {
    p->device        = ctx->device;
    const size_t ncb = (size_t)p->cfg.max_softbuffers * p->cfg.max_codeblocks_per_buffer;
    hipError_t   e   = hipSetDevice(ctx->device);
    if (e == hipSuccess)
      e = hipMalloc((void**)&p->d_soft, ncb * CB_SOFT_STRIDE);
    if (e == hipSuccess)
      e = hipMalloc((void**)&p->d_msgs, ncb * CB_MSG_STRIDE);
    if (e == hipSuccess)
      e = hipMalloc((void**)&p->d_crc, ncb);
    // A softbuffer's contents are defined from its first new-data transmission on; zeroes make diagnostics readable.
    if (e == hipSuccess)
      e = hipMemset(p->d_soft, 0, ncb * CB_SOFT_STRIDE);
    if (e == hipSuccess)
      e = hipMemset(p->d_msgs, 0, ncb * CB_MSG_STRIDE);
    if (e == hipSuccess)
      e = hipMemset(p->d_crc, 0, ncb);
    if (e != hipSuccess) {
      miphy_set_error("harq_pool_create: %zu codeblock slots: %s", ncb, hipGetErrorString(e));
      miphy_harq_pool_destroy(p);
      return e == hipErrorOutOfMemory ? MIPHY_ENOMEM : MIPHY_EHIP;
    }
  }
  *out = p;
  return MIPHY_OK;
}

extern "C" void miphy_harq_pool_destroy(miphy_harq_pool* p)
{
  if (!p)
    return;
  if (p->device >= 0)
    (void)hipSetDevice(p->device);
  (void)hipFree(p->d_soft);
  (void)hipFree(p->d_msgs);
  (void)hipFree(p->d_crc);
  delete p;
}

extern "C" int miphy_harq_pool_reserve(miphy_harq_pool* p, uint32_t slot, uint32_t rnti, uint32_t harq_id, uint32_t nof_codeblocks, int32_t* buffer,
                                       uint32_t* first_cb)
{
  MIPHY_REQUIRE(p && buffer, "harq_pool_reserve: null argument");
  *buffer = -1;
  if (first_cb)
    *first_cb = 0;
  MIPHY_REQUIRE(slot < p->cfg.nof_slots_wrap, "harq_pool_reserve: slot %u outside the period %u", slot, p->cfg.nof_slots_wrap);
  MIPHY_REQUIRE(rnti <= 0xffff && harq_id <= 0xff, "harq_pool_reserve: identifier (%u, %u) out of range", rnti, harq_id);
  MIPHY_REQUIRE(nof_codeblocks <= p->cfg.max_codeblocks_per_buffer, "harq_pool_reserve: %u codeblocks exceed the softbuffer extent %u", nof_codeblocks,
                p->cfg.max_codeblocks_per_buffer);
  std::lock_guard<std::mutex> lock(p->mutex);
  const uint32_t              expire = (slot + p->cfg.expire_timeout_slots) % p->cfg.nof_slots_wrap;
  int                         pick   = -1;
  for (size_t i = 0; i != p->buffers.size() && pick < 0; ++i) // same identifier, whatever the state (match_id)
    if (p->buffers[i].rnti == rnti && p->buffers[i].harq_id == harq_id)
      pick = (int)i;
  for (size_t i = 0; i != p->buffers.size() && pick < 0; ++i) // else the first available one (is_available)
    if (p->buffers[i].state == MIPHY_HARQ_AVAILABLE || p->buffers[i].state == MIPHY_HARQ_RELEASED)
      pick = (int)i;
  if (pick < 0 || !p->reserve(p->buffers[pick], rnti, harq_id, expire, nof_codeblocks))
    return MIPHY_OK;
  *buffer = pick;
  if (first_cb)
    *first_cb = (uint32_t)pick * p->cfg.max_codeblocks_per_buffer;
  return MIPHY_OK;
}

#define HARQ_BUFFER_ARG(name)                                                                                       \
  MIPHY_REQUIRE(p, name ": null pool");                                                                             \
  MIPHY_REQUIRE(buffer >= 0 && (size_t)buffer < p->buffers.size(), name ": softbuffer %d out of range", (int)buffer); \
  std::lock_guard<std::mutex> lock(p->mutex);                                                                       \
  softbuffer&                 b = p->buffers[buffer]

extern "C" int miphy_harq_pool_lock(miphy_harq_pool* p, int32_t buffer)
{
  HARQ_BUFFER_ARG("harq_pool_lock");
  MIPHY_REQUIRE(b.state == MIPHY_HARQ_RESERVED, "harq_pool_lock: softbuffer %d is not reserved (state %u)", (int)buffer, b.state);
  b.state = MIPHY_HARQ_LOCKED;
  return MIPHY_OK;
}

extern "C" int miphy_harq_pool_unlock(miphy_harq_pool* p, int32_t buffer)
{
  HARQ_BUFFER_ARG("harq_pool_unlock");
  if (b.state == MIPHY_HARQ_LOCKED)
    b.state = MIPHY_HARQ_RESERVED;
  return MIPHY_OK;
}

extern "C" int miphy_harq_pool_release(miphy_harq_pool* p, int32_t buffer)
{
  HARQ_BUFFER_ARG("harq_pool_release");
  MIPHY_REQUIRE(b.state == MIPHY_HARQ_RESERVED || b.state == MIPHY_HARQ_LOCKED, "harq_pool_release: softbuffer %d is neither reserved nor locked (state %u)",
                (int)buffer, b.state);
  b.state = MIPHY_HARQ_RELEASED;
  return MIPHY_OK;
}

extern "C" int miphy_harq_pool_run_slot(miphy_harq_pool* p, uint32_t slot)
{
  MIPHY_REQUIRE(p, "harq_pool_run_slot: null pool");
  MIPHY_REQUIRE(slot < p->cfg.nof_slots_wrap, "harq_pool_run_slot: slot %u outside the period %u", slot, p->cfg.nof_slots_wrap);
  std::lock_guard<std::mutex> lock(p->mutex);
  for (softbuffer& b : p->buffers) { // rx_softbuffer_impl::run_slot
    const bool released = b.state == MIPHY_HARQ_RELEASED;
    const bool expired  = b.state == MIPHY_HARQ_RESERVED && p->slot_le(b.expire_slot, slot);
    if (released || expired)
      p->free_buffer(b);
  }
  return MIPHY_OK;
}

extern "C" int miphy_harq_pool_info(miphy_harq_pool* p, int32_t buffer, miphy_harq_buffer_info* out)
{
  MIPHY_REQUIRE(out, "harq_pool_info: null argument");
  HARQ_BUFFER_ARG("harq_pool_info");
  out->state = b.state, out->rnti = b.rnti, out->harq_id = b.harq_id, out->nof_codeblocks = b.nof_codeblocks;
  out->first_cb = (uint32_t)buffer * p->cfg.max_codeblocks_per_buffer, out->expire_slot = b.expire_slot;
  return MIPHY_OK;
}

extern "C" int miphy_harq_pool_free_codeblocks(miphy_harq_pool* p, uint32_t* out)
{
  MIPHY_REQUIRE(p && out, "harq_pool_free_codeblocks: null argument");
  std::lock_guard<std::mutex> lock(p->mutex);
  *out = p->free_cbs;
  return MIPHY_OK;
}

extern "C" int miphy_harq_pool_arrays(miphy_harq_pool* p, int8_t** softbits, uint8_t** msgs, uint8_t** crc_ok)
{
  MIPHY_REQUIRE(p && softbits && msgs && crc_ok, "harq_pool_arrays: null argument");
  MIPHY_REQUIRE(p->d_soft, "harq_pool_arrays: bookkeeping-only pool (created without a context)");
  *softbits = p->d_soft, *msgs = p->d_msgs, *crc_ok = p->d_crc;
  return MIPHY_OK;
}
