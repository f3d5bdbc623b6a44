// Host-side adapters: implement srsRAN_Project 23.5's own abstract interfaces on top of the miphy C ABI, so that the
// reference's factories / processors can use the MI355X path as a drop-in.
//
// Each class derives from the reference interface it replaces and forwards ONE call as a batch of one: spans in, H2D copy,
// kernel(s), D2H copy, spans out -- same argument meaning, same results, `report_fatal_error` where the reference would
// assert. They are deliberately thin: throughput comes from the batched C ABI (include/miphy.h), which the slot-level
// callers should use directly; the adapters exist so that every existing call site keeps compiling and keeps its tests.
//
// This header needs the reference's include/ directory and <hip/hip_runtime_api.h>; it is header-only and is compiled by
// whoever integrates it (the reference's build, or oracle/build_ref.sh for the drop-in test).
#pragma once

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include "miphy.h"
#include "srsran/ofh/compression/iq_compressor.h"
#include "srsran/ofh/compression/iq_decompressor.h"
#include "srsran/phy/generic_functions/generic_functions_factories.h"
#include "srsran/phy/lower/modulation/modulation_factories.h"
#include "srsran/phy/support/resource_grid.h"
#include "srsran/phy/upper/channel_coding/channel_coding_factories.h"
#include "srsran/phy/upper/channel_estimation.h"
#include "srsran/phy/support/resource_grid_context.h"
#include "srsran/phy/upper/channel_processors/channel_processor_factories.h"
#include "srsran/phy/upper/downlink_processor.h"
#include "srsran/phy/upper/equalization/equalization_factories.h"
#include "srsran/phy/upper/resource_grid_mapper.h"
#include "srsran/phy/upper/upper_phy_rg_gateway.h"
#include "srsran/ran/csi_rs/csi_rs_pattern.h"
#include "srsran/ran/pdcch/cce_to_prb_mapping.h"
#include "srsran/ran/pusch/ulsch_info.h"
#include "srsran/ran/ssb_mapping.h"
#include "srsran/ran/precoding/precoding_codebooks.h"
#include "srsran/phy/upper/rx_softbuffer.h"
#include "srsran/phy/upper/rx_softbuffer_pool.h"
#include "srsran/phy/upper/unique_rx_softbuffer.h"
#include "srsran/phy/upper/uplink_processor.h"
#include "srsran/phy/upper/upper_phy_rx_results_notifier.h"
#include "srsran/phy/upper/signal_processors/nzp_csi_rs_generator.h"
#include "srsran/phy/upper/signal_processors/port_channel_estimator.h"
#include "srsran/phy/upper/signal_processors/signal_processor_factories.h"
#include "srsran/support/error_handling.h"
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstring>
#include <hip/hip_runtime_api.h>
#include <memory>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace miphy {

/// Precondition that also holds in release builds (srsran_assert compiles out): report_fatal_error when it fails.
template <typename... Args>
inline void require(bool condition, const char* fmtstr, Args&&... args)
{
  if (!condition) {
    srsran::report_fatal_error(fmtstr, std::forward<Args>(args)...);
  }
}

// ---------------------------------------------------------------------------------------------------------------- context
/// Owns a miphy context, a stream and a growable pair of device buffers used as staging by the per-call adapters.
class context
{
public:
  explicit context(int device_ = 0) : device(device_)
  {
    check(miphy_create(device, &ctx), "miphy_create"); // makes `device` the calling thread's current device
    hip(hipStreamCreate(&stream), "hipStreamCreate");
  }
  /// Makes the context's device the current one of the calling thread. A new thread starts on device 0: every thread that
  /// drives this context (allocations in buf(), the library's workspaces, launches) has to call this first.
  void bind_thread() const { hip(hipSetDevice(device), "hipSetDevice"); }
  ~context()
  {
    for (auto& b : bufs) {
      (void)hipFree(b.p);
    }
    (void)hipStreamDestroy(stream);
    miphy_destroy(ctx);
  }
  context(const context&)            = delete;
  context& operator=(const context&) = delete;

  static void check(int rc, const char* what)
  {
    if (rc != MIPHY_OK) {
      srsran::report_fatal_error("{} failed ({}): {}", what, rc, miphy_last_error());
    }
  }
  static void hip(hipError_t e, const char* what)
  {
    if (e != hipSuccess) {
      srsran::report_fatal_error("{} failed: {}", what, hipGetErrorString(e));
    }
  }
  /// Device scratch buffer number \c i of at least \c bytes bytes.
  void* buf(unsigned i, size_t bytes)
  {
    if (bufs.size() <= i) {
      bufs.resize(i + 1);
    }
    if (bufs[i].n < bytes) {
      hip(hipStreamSynchronize(stream), "sync");
      (void)hipFree(bufs[i].p);
      bufs[i].n = bytes + bytes / 2 + 4096;
      hip(hipMalloc(&bufs[i].p, bufs[i].n), "hipMalloc");
    }
    return bufs[i].p;
  }
  void h2d(void* d, const void* h, size_t n) { hip(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, stream), "h2d"); }
  void d2h(void* h, const void* d, size_t n) { hip(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, stream), "d2h"); }
  void sync() { hip(hipStreamSynchronize(stream), "sync"); }

  miphy_ctx*  ctx    = nullptr;
  hipStream_t stream = nullptr;
  const int   device;

private:
  struct dbuf {
    void*  p = nullptr;
    size_t n = 0;
  };
  std::vector<dbuf> bufs;
};

inline uint8_t to_miphy_crc(srsran::crc_generator_poly p)
{
  switch (p) {
    case srsran::crc_generator_poly::CRC24A:
      return MIPHY_CRC24A;
    case srsran::crc_generator_poly::CRC24B:
      return MIPHY_CRC24B;
    case srsran::crc_generator_poly::CRC24C:
      return MIPHY_CRC24C;
    case srsran::crc_generator_poly::CRC16:
      return MIPHY_CRC16;
    case srsran::crc_generator_poly::CRC11:
      return MIPHY_CRC11;
    default:
      srsran::report_fatal_error("CRC polynomial not supported by the HIP path");
  }
}

inline uint8_t bg_id(srsran::ldpc_base_graph_type bg)
{
  return bg == srsran::ldpc_base_graph_type::BG1 ? 1 : 2;
}

// ---------------------------------------------------------------------------------------------------------------- LDPC
/// srsran::ldpc_decoder over miphy_ldpc_decode_batch (include/srsran/phy/upper/channel_coding/ldpc/ldpc_decoder.h:73-74).
class ldpc_decoder_hip : public srsran::ldpc_decoder
{
public:
  explicit ldpc_decoder_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  srsran::optional<unsigned> decode(srsran::bit_buffer&                              output,
                                    srsran::span<const srsran::log_likelihood_ratio> input,
                                    srsran::crc_calculator*                          crc,
                                    const configuration&                             cfg) override
  {
    miphy_ldpc_dec_desc d = {};
    d.bg                  = bg_id(cfg.block_conf.tb_common.base_graph);
    d.Z                   = static_cast<uint16_t>(cfg.block_conf.tb_common.lifting_size);
    d.crc_poly            = crc ? to_miphy_crc(crc->get_generator_poly()) : MIPHY_CRC_NONE;
    d.max_iter            = cfg.algorithm_conf.max_iterations;
    d.nof_filler_bits     = cfg.block_conf.cb_specific.nof_filler_bits;
    d.in_len              = input.size();
    size_t   nbytes       = (output.size() + 7) / 8;
    auto*    d_llr        = static_cast<int8_t*>(c->buf(0, input.size()));
    auto*    d_out        = static_cast<uint8_t*>(c->buf(1, nbytes + 16));
    int32_t* d_it         = reinterpret_cast<int32_t*>(d_out + ((nbytes + 7) / 8) * 8);
    c->h2d(d_llr, input.data(), input.size());
    c->h2d(d_out, output.get_buffer().data(), nbytes); // all-zero input with a CRC leaves the output untouched
    context::check(miphy_ldpc_decode_batch(c->ctx, &d, 0, 1, d_llr, d_out, d_it, nullptr, c->stream), "ldpc_decode");
    int32_t it = 0;
    c->d2h(output.get_buffer().data(), d_out, nbytes);
    c->d2h(&it, d_it, sizeof(it));
    c->sync();
    if (it > 0) {
      return static_cast<unsigned>(it);
    }
    return srsran::nullopt;
  }

private:
  std::shared_ptr<context> c;
};

/// srsran::ldpc_encoder over miphy_ldpc_encode_batch (ldpc_encoder.h:46-47).
class ldpc_encoder_hip : public srsran::ldpc_encoder
{
public:
  explicit ldpc_encoder_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void encode(srsran::span<uint8_t>                                 output,
              srsran::span<const uint8_t>                           input,
              const srsran::codeblock_metadata::tb_common_metadata& cfg) override
  {
    miphy_ldpc_enc_desc d = {};
    d.bg                  = bg_id(cfg.base_graph);
    d.Z                   = static_cast<uint16_t>(cfg.lifting_size);
    d.out_len             = output.size();
    auto* d_in            = static_cast<uint8_t*>(c->buf(0, input.size()));
    auto* d_out           = static_cast<uint8_t*>(c->buf(1, output.size()));
    c->h2d(d_in, input.data(), input.size());
    context::check(miphy_ldpc_encode_batch(c->ctx, &d, 0, 1, d_in, d_out, c->stream), "ldpc_encode");
    c->d2h(output.data(), d_out, output.size());
    c->sync();
  }

private:
  std::shared_ptr<context> c;
};

inline miphy_ldpc_rdm_desc make_rdm_desc(const srsran::codeblock_metadata& cfg, unsigned block_length, unsigned E, bool new_data)
{
  miphy_ldpc_rdm_desc d = {};
  // The reference infers the base graph from the block length (ldpc_rate_matcher_impl.cpp:68-80).
  d.bg = (block_length % 66 == 0) ? 1 : 2;
  d.Z  = static_cast<uint16_t>(block_length / (d.bg == 1 ? 66 : 50));
  d.rv = cfg.tb_common.rv;
  d.mod             = srsran::get_bits_per_symbol(cfg.tb_common.mod);
  d.new_data        = new_data;
  d.nof_filler_bits = cfg.cb_specific.nof_filler_bits;
  d.Nref            = cfg.tb_common.Nref;
  d.E               = E;
  return d;
}

/// srsran::ldpc_rate_matcher over miphy_ldpc_rate_match_batch (ldpc_rate_matcher.h:46).
class ldpc_rate_matcher_hip : public srsran::ldpc_rate_matcher
{
public:
  explicit ldpc_rate_matcher_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void rate_match(srsran::span<uint8_t> output, srsran::span<const uint8_t> input, const srsran::codeblock_metadata& cfg) override
  {
    miphy_ldpc_rdm_desc d = make_rdm_desc(cfg, input.size(), output.size(), true);
    auto* d_in  = static_cast<uint8_t*>(c->buf(0, input.size()));
    auto* d_out = static_cast<uint8_t*>(c->buf(1, output.size()));
    c->h2d(d_in, input.data(), input.size());
    context::check(miphy_ldpc_rate_match_batch(c->ctx, &d, 0, 1, d_in, d_out, c->stream), "ldpc_rate_match");
    c->d2h(output.data(), d_out, output.size());
    c->sync();
  }

private:
  std::shared_ptr<context> c;
};

/// srsran::ldpc_rate_dematcher over miphy_ldpc_rate_dematch_batch (ldpc_rate_dematcher.h:52-55).
class ldpc_rate_dematcher_hip : public srsran::ldpc_rate_dematcher
{
public:
  explicit ldpc_rate_dematcher_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void rate_dematch(srsran::span<srsran::log_likelihood_ratio>       output,
                    srsran::span<const srsran::log_likelihood_ratio> input,
                    bool                                             new_data,
                    const srsran::codeblock_metadata&                cfg) override
  {
    miphy_ldpc_rdm_desc d = make_rdm_desc(cfg, output.size(), input.size(), new_data);
    auto* d_in  = static_cast<int8_t*>(c->buf(0, input.size()));
    auto* d_out = static_cast<int8_t*>(c->buf(1, output.size()));
    c->h2d(d_in, input.data(), input.size());
    c->h2d(d_out, output.data(), output.size()); // in/out: the soft buffer content is combined or partially kept
    context::check(miphy_ldpc_rate_dematch_batch(c->ctx, &d, 0, 1, d_in, d_out, nullptr, c->stream), "ldpc_rate_dematch");
    c->d2h(output.data(), d_out, output.size());
    c->sync();
  }

private:
  std::shared_ptr<context> c;
};

// ---------------------------------------------------------------------------------------------------------------- SCH
/// srsran::pdsch_encoder over miphy_pdsch_encode_batch (pdsch_encoder.h:54).
class pdsch_encoder_hip : public srsran::pdsch_encoder
{
public:
  explicit pdsch_encoder_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void encode(srsran::span<uint8_t> codeword, srsran::span<const uint8_t> transport_block, const srsran::segmenter_config& cfg) override
  {
    miphy_pdsch_tb_desc d = {};
    d.bg                  = bg_id(cfg.base_graph);
    d.rv                  = cfg.rv;
    d.mod                 = srsran::get_bits_per_symbol(cfg.mod);
    d.nof_layers          = cfg.nof_layers;
    d.Nref                = cfg.Nref;
    d.nof_ch_symbols      = cfg.nof_ch_symbols;
    d.tb_bytes            = transport_block.size();
    auto* d_tb            = static_cast<uint8_t*>(c->buf(0, transport_block.size() + 16));
    auto* d_cw            = static_cast<uint8_t*>(c->buf(1, codeword.size()));
    c->h2d(d_tb, transport_block.data(), transport_block.size());
    context::check(miphy_pdsch_encode_batch(c->ctx, &d, 1, d_tb, d_cw, c->stream), "pdsch_encode");
    c->d2h(codeword.data(), d_cw, codeword.size());
    c->sync();
  }

private:
  std::shared_ptr<context> c;
};

// ---------------------------------------------------------------------------------------------------------------- HARQ softbuffer pool
/// A softbuffer of the device-resident pool (miphy_harq_pool). pusch_decoder_hip and pusch_processor_hip recognise it and work
/// on the device arrays in place: no HARQ state crosses PCIe. The host accessors of srsran::rx_softbuffer still work, so that
/// a CPU block can be handed the same softbuffer: they download the codeblock into a host mirror, and everything handed out
/// is written back to the device before the next device use and when the softbuffer is unlocked or released.
class rx_softbuffer_hip : public srsran::unique_rx_softbuffer::softbuffer
{
public:
  static constexpr size_t CBS = 66 * 384, MSG = 1056;

  rx_softbuffer_hip(std::shared_ptr<context> c, miphy_harq_pool* pool, int32_t index) : c(std::move(c)), pool(pool), index(index)
  {
    context::check(miphy_harq_pool_arrays(pool, &d_soft, &d_msgs, &d_crc), "harq_pool_arrays");
  }
  unsigned get_nof_codeblocks() const override { return info().nof_codeblocks; }
  void     reset_codeblocks_crc() override
  {
    const miphy_harq_buffer_info i = info();
    crc_out                        = false;
    context::hip(hipMemsetAsync(d_crc + i.first_cb, 0, i.nof_codeblocks, c->stream), "memset");
    c->sync();
  }
  srsran::span<bool> get_codeblocks_crc() override
  {
    const miphy_harq_buffer_info i = info();
    if (!crc_out) {
      crc_bytes.resize(i.nof_codeblocks);
      c->d2h(crc_bytes.data(), d_crc + i.first_cb, i.nof_codeblocks);
      c->sync();
      crc_mirror.reset(new bool[i.nof_codeblocks + 1]);
      for (unsigned k = 0; k != i.nof_codeblocks; ++k) {
        crc_mirror[k] = crc_bytes[k] != 0;
      }
      crc_out = true;
    }
    return srsran::span<bool>(crc_mirror.get(), i.nof_codeblocks);
  }
  srsran::span<srsran::log_likelihood_ratio> get_codeblock_soft_bits(unsigned codeblock_id, unsigned codeblock_size) override
  {
    const miphy_harq_buffer_info i = info();
    srsran_assert(codeblock_id < i.nof_codeblocks, "Codeblock index ({}) is out of range ({}).", codeblock_id, i.nof_codeblocks);
    srsran_assert(codeblock_size <= CBS, "Codeblock size {} exceeds maximum size {}.", codeblock_size, CBS);
    mirror& m = mirror_of(codeblock_id);
    if (!m.soft_out) {
      m.soft.resize(CBS);
      c->d2h(m.soft.data(), d_soft + (i.first_cb + codeblock_id) * CBS, CBS);
      c->sync();
      m.soft_out = true;
    }
    return srsran::span<srsran::log_likelihood_ratio>(m.soft).first(codeblock_size);
  }
  srsran::bit_buffer get_codeblock_data_bits(unsigned codeblock_id, unsigned data_size) override
  {
    const miphy_harq_buffer_info i = info();
    srsran_assert(codeblock_id < i.nof_codeblocks, "Codeblock index ({}) is out of range ({}).", codeblock_id, i.nof_codeblocks);
    srsran_assert(data_size <= MSG * 8, "Codeblock data size {} exceeds maximum size {}.", data_size, MSG * 8);
    mirror& m = mirror_of(codeblock_id);
    if (!m.msg_out) {
      m.msg.resize(MSG * 8);
      c->d2h(m.msg.get_buffer().data(), d_msgs + (i.first_cb + codeblock_id) * MSG, MSG);
      c->sync();
      m.msg_out = true;
    }
    return m.msg.first(data_size);
  }
  void lock() override { context::check(miphy_harq_pool_lock(pool, index), "harq_pool_lock"); }
  void unlock() override
  {
    flush();
    context::check(miphy_harq_pool_unlock(pool, index), "harq_pool_unlock");
  }
  void release() override
  {
    flush();
    context::check(miphy_harq_pool_release(pool, index), "harq_pool_release");
  }

  /// Writes back whatever the host accessors handed out since the last flush (a CPU block may have modified it).
  void flush()
  {
    const miphy_harq_buffer_info i    = info();
    bool                         any  = false;
    for (size_t k = 0; k != mirrors.size(); ++k) {
      if (!mirrors[k]) {
        continue;
      }
      mirror& m = *mirrors[k];
      if (m.soft_out && k < i.nof_codeblocks) {
        c->h2d(d_soft + (i.first_cb + k) * CBS, m.soft.data(), CBS);
        any = true;
      }
      if (m.msg_out && k < i.nof_codeblocks) {
        c->h2d(d_msgs + (i.first_cb + k) * MSG, m.msg.get_buffer().data(), MSG);
        any = true;
      }
      m.soft_out = m.msg_out = false;
    }
    if (crc_out) {
      for (size_t k = 0; k != crc_bytes.size(); ++k) {
        crc_bytes[k] = crc_mirror[k] ? 1 : 0;
      }
      c->h2d(d_crc + i.first_cb, crc_bytes.data(), std::min<size_t>(crc_bytes.size(), i.nof_codeblocks));
      any     = true;
      crc_out = false;
    }
    if (any) {
      c->sync();
    }
  }
  uint32_t first_cb() const { return info().first_cb; }
  int8_t*  softbits() const { return d_soft; }
  uint8_t* msgs() const { return d_msgs; }
  uint8_t* crc_ok() const { return d_crc; }

private:
  struct mirror {
    std::vector<srsran::log_likelihood_ratio> soft;
    srsran::dynamic_bit_buffer                msg;
    bool                                      soft_out = false, msg_out = false;
  };
  miphy_harq_buffer_info info() const
  {
    miphy_harq_buffer_info i;
    context::check(miphy_harq_pool_info(pool, index, &i), "harq_pool_info");
    return i;
  }
  // The views handed out must stay valid while other codeblocks are visited: every mirror is its own allocation.
  mirror& mirror_of(unsigned cb)
  {
    if (mirrors.size() <= cb) {
      mirrors.resize(cb + 1);
    }
    if (!mirrors[cb]) {
      mirrors[cb] = std::make_unique<mirror>();
    }
    return *mirrors[cb];
  }

  std::shared_ptr<context> c;
  miphy_harq_pool*         pool;
  int32_t                  index;
  int8_t*                  d_soft = nullptr;
  uint8_t *                d_msgs = nullptr, *d_crc = nullptr;
  std::vector<std::unique_ptr<mirror>> mirrors;
  std::vector<uint8_t>     crc_bytes;
  std::unique_ptr<bool[]>  crc_mirror;
  bool                     crc_out = false;
};

/// srsran::rx_softbuffer_pool over miphy_harq_pool (rx_softbuffer_pool.h:52-80): same reservation rules and life cycle as
/// create_rx_softbuffer_pool(config), with the codeblocks in device memory. max_codeblock_size is fixed by the device
/// layout (66 * 384 soft bits per codeblock). The slot period needs the numerology, which the reference's configuration does
/// not carry: it is taken from the first slot_point the pool sees unless given here.
class rx_softbuffer_pool_hip : public srsran::rx_softbuffer_pool
{
public:
  rx_softbuffer_pool_hip(std::shared_ptr<context> c, const srsran::rx_softbuffer_pool_config& config) : c(std::move(c)), config(config)
  {
    srsran_assert(config.max_codeblock_size <= rx_softbuffer_hip::CBS, "Codeblocks of the device pool hold {} soft bits.", rx_softbuffer_hip::CBS);
  }
  ~rx_softbuffer_pool_hip() override
  {
    buffers.clear();
    miphy_harq_pool_destroy(pool);
  }
  srsran::unique_rx_softbuffer
  reserve_softbuffer(const srsran::slot_point& slot, const srsran::rx_softbuffer_identifier& id, unsigned nof_codeblocks) override
  {
    std::lock_guard<std::mutex> lock(mutex);
    create(slot);
    int32_t  b     = -1;
    uint32_t first = 0;
    context::check(miphy_harq_pool_reserve(pool, slot.to_uint(), id.rnti, id.harq_ack_id, nof_codeblocks, &b, &first), "harq_pool_reserve");
    if (b < 0) {
      return srsran::unique_rx_softbuffer();
    }
    return srsran::unique_rx_softbuffer(*buffers[b]); // locks, like the reference
  }
  void run_slot(const srsran::slot_point& slot) override
  {
    std::lock_guard<std::mutex> lock(mutex);
    create(slot);
    context::check(miphy_harq_pool_run_slot(pool, slot.to_uint()), "harq_pool_run_slot");
  }

private:
  void create(const srsran::slot_point& slot)
  {
    if (pool != nullptr) {
      return;
    }
    miphy_harq_pool_config cfg = {};
    cfg.max_softbuffers = config.max_softbuffers, cfg.max_nof_codeblocks = config.max_nof_codeblocks;
    cfg.expire_timeout_slots = config.expire_timeout_slots, cfg.nof_slots_wrap = slot.nof_slots_per_system_frame();
    cfg.max_codeblocks_per_buffer = srsran::MAX_NOF_SEGMENTS;
    context::check(miphy_harq_pool_create(c->ctx, &cfg, &pool), "harq_pool_create");
    for (unsigned i = 0; i != config.max_softbuffers; ++i) {
      buffers.emplace_back(std::make_unique<rx_softbuffer_hip>(c, pool, static_cast<int32_t>(i)));
    }
  }

  std::shared_ptr<context>                        c;
  srsran::rx_softbuffer_pool_config               config;
  std::mutex                                      mutex;
  miphy_harq_pool*                                pool = nullptr;
  std::vector<std::unique_ptr<rx_softbuffer_hip>> buffers;
};

/// Replaces srsran::create_rx_softbuffer_pool(config).
inline std::unique_ptr<srsran::rx_softbuffer_pool> create_rx_softbuffer_pool_hip(std::shared_ptr<context> c, const srsran::rx_softbuffer_pool_config& config)
{
  return std::make_unique<rx_softbuffer_pool_hip>(std::move(c), config);
}

/// srsran::pusch_decoder over miphy_pusch_decode_batch (pusch_decoder.h:74-78). With a softbuffer of rx_softbuffer_pool_hip
/// the HARQ state is used in place in device memory; with any other rx_softbuffer (the reference's CPU pool) it is uploaded
/// before and downloaded after the call.
class pusch_decoder_hip : public srsran::pusch_decoder
{
public:
  explicit pusch_decoder_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void decode(srsran::span<uint8_t>                            transport_block,
              srsran::pusch_decoder_result&                    stats,
              srsran::rx_softbuffer*                           soft_codeword,
              srsran::span<const srsran::log_likelihood_ratio> llrs,
              const configuration&                             cfg) override
  {
    miphy_sch_segmentation sg;
    context::check(miphy_sch_segmentation_info(transport_block.size(), bg_id(cfg.segmenter_cfg.base_graph), &sg), "segmentation");
    srsran_assert(sg.nof_cbs == soft_codeword->get_nof_codeblocks(), "Wrong number of codeblocks.");
    const size_t CBS = 66 * 384, MSG = 1056;
    auto*        resident = dynamic_cast<rx_softbuffer_hip*>(soft_codeword);
    auto*        d_llr  = static_cast<int8_t*>(c->buf(0, llrs.size()));
    auto*        d_misc = static_cast<uint8_t*>(c->buf(3, 64 + sizeof(miphy_pusch_result) + transport_block.size() + 64));
    auto*        d_res  = reinterpret_cast<miphy_pusch_result*>(d_misc + 64);
    uint8_t*     d_tb   = d_misc + 64 + 64;
    int8_t*      d_soft = nullptr;
    uint8_t *    d_msg = nullptr, *d_crc = nullptr;
    std::vector<uint8_t> crc_h(sg.nof_cbs);
    if (resident != nullptr) {
      resident->flush();
      d_soft = resident->softbits(), d_msg = resident->msgs(), d_crc = resident->crc_ok();
    } else {
      d_soft = static_cast<int8_t*>(c->buf(1, sg.nof_cbs * CBS));
      d_msg  = static_cast<uint8_t*>(c->buf(2, sg.nof_cbs * MSG));
      d_crc  = d_misc;
      srsran::span<bool> crcs = soft_codeword->get_codeblocks_crc();
      for (unsigned i = 0; i != sg.nof_cbs; ++i) {
        crc_h[i] = crcs[i] ? 1 : 0;
        auto sb  = soft_codeword->get_codeblock_soft_bits(i, sg.N);
        c->h2d(d_soft + i * CBS, sb.data(), sg.N);
        auto mb = soft_codeword->get_codeblock_data_bits(i, sg.K);
        c->h2d(d_msg + i * MSG, mb.get_buffer().data(), (sg.K + 7) / 8);
      }
      c->h2d(d_crc, crc_h.data(), sg.nof_cbs);
    }
    c->h2d(d_llr, llrs.data(), llrs.size());
    c->h2d(d_tb, transport_block.data(), transport_block.size());
    miphy_pusch_tb_desc d = {};
    d.bg                  = bg_id(cfg.segmenter_cfg.base_graph);
    d.rv                  = cfg.segmenter_cfg.rv;
    d.mod                 = srsran::get_bits_per_symbol(cfg.segmenter_cfg.mod);
    d.nof_layers          = cfg.segmenter_cfg.nof_layers;
    d.new_data            = cfg.new_data;
    d.use_early_stop      = cfg.use_early_stop;
    d.nof_ldpc_iterations = cfg.nof_ldpc_iterations;
    d.Nref                = cfg.segmenter_cfg.Nref;
    d.nof_ch_symbols      = cfg.segmenter_cfg.nof_ch_symbols;
    d.tb_bytes            = transport_block.size();
    d.harq_cb_index       = resident != nullptr ? resident->first_cb() : 0;
    context::check(miphy_pusch_decode_batch(c->ctx, &d, 1, d_llr, d_soft, d_msg, d_crc, d_tb, d_res, c->stream), "pusch_decode");
    miphy_pusch_result r;
    c->d2h(&r, d_res, sizeof(r));
    c->d2h(transport_block.data(), d_tb, transport_block.size());
    if (resident == nullptr) {
      c->d2h(crc_h.data(), d_crc, sg.nof_cbs);
      for (unsigned i = 0; i != sg.nof_cbs; ++i) {
        auto sb = soft_codeword->get_codeblock_soft_bits(i, sg.N);
        c->d2h(sb.data(), d_soft + i * CBS, sg.N);
        auto mb = soft_codeword->get_codeblock_data_bits(i, sg.K);
        c->d2h(mb.get_buffer().data(), d_msg + i * MSG, (sg.K + 7) / 8);
      }
    }
    c->sync();
    if (resident == nullptr) {
      srsran::span<bool> crcs = soft_codeword->get_codeblocks_crc();
      for (unsigned i = 0; i != sg.nof_cbs; ++i) {
        crcs[i] = crc_h[i] != 0;
      }
    }
    stats.tb_crc_ok            = r.tb_crc_ok != 0;
    stats.nof_codeblocks_total = r.nof_codeblocks_total;
    stats.ldpc_decoder_stats.reset();
    // min / max / count are reproduced exactly; the running mean is rebuilt from them and the device-side mean.
    if (r.nof_decoded > 0) {
      stats.ldpc_decoder_stats.update(r.iters_min);
      for (unsigned i = 1; i + 1 < r.nof_decoded; ++i) {
        stats.ldpc_decoder_stats.update(static_cast<unsigned>(r.iters_mean + 0.5F));
      }
      if (r.nof_decoded > 1) {
        stats.ldpc_decoder_stats.update(r.iters_max);
      }
    }
  }

private:
  std::shared_ptr<context> c;
};

// ---------------------------------------------------------------------------------------------------------------- DFT
/// srsran::dft_processor over miphy_dft_batch (include/srsran/phy/generic_functions/dft_processor.h:34-73). Like the
/// reference implementations it owns its input and output buffers and hands out views.
class dft_processor_hip : public srsran::dft_processor
{
public:
  dft_processor_hip(std::shared_ptr<context> c, const configuration& cfg) : c(std::move(c)), dir(cfg.dir), in(cfg.size), out(cfg.size) {}
  direction                        get_direction() const override { return dir; }
  unsigned                         get_size() const override { return in.size(); }
  srsran::span<srsran::cf_t>       get_input() override { return in; }
  srsran::span<const srsran::cf_t> run() override
  {
    const size_t bytes = in.size() * sizeof(srsran::cf_t);
    auto*        d_in  = static_cast<float*>(c->buf(0, bytes));
    auto*        d_out = static_cast<float*>(c->buf(1, bytes));
    c->h2d(d_in, in.data(), bytes);
    context::check(miphy_dft_batch(c->ctx, in.size(), dir == direction::INVERSE, 1, d_in, d_out, c->stream), "dft");
    c->d2h(out.data(), d_out, bytes);
    c->sync();
    return out;
  }

private:
  std::shared_ptr<context>  c;
  direction                 dir;
  std::vector<srsran::cf_t> in, out;
};

class dft_processor_factory_hip : public srsran::dft_processor_factory
{
public:
  explicit dft_processor_factory_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  std::unique_ptr<srsran::dft_processor> create(const srsran::dft_processor::configuration& cfg) override
  {
    return std::make_unique<dft_processor_hip>(c, cfg);
  }

private:
  std::shared_ptr<context> c;
};

// ---------------------------------------------------------------------------------------------------------------- OFDM
/// srsran::ofdm_slot_demodulator over miphy_ofdm_demodulate_slots (ofdm_demodulator.h:91-101).
class ofdm_slot_demodulator_hip : public srsran::ofdm_slot_demodulator
{
public:
  ofdm_slot_demodulator_hip(std::shared_ptr<context> c, const srsran::ofdm_demodulator_configuration& cfg_) : c(std::move(c))
  {
    cfg.numerology                = cfg_.numerology;
    cfg.bw_rb                     = cfg_.bw_rb;
    cfg.dft_size                  = cfg_.dft_size;
    cfg.nof_samples_window_offset = cfg_.nof_samples_window_offset;
    cfg.scale                     = cfg_.scale;
    cfg.center_freq_hz            = cfg_.center_freq_hz;
    srsran_assert(cfg_.cp == srsran::cyclic_prefix::NORMAL, "Only normal cyclic prefix is supported by the HIP path.");
  }
  unsigned get_slot_size(unsigned slot_index) const override { return miphy_ofdm_slot_size(&cfg, slot_index); }
  void demodulate(srsran::resource_grid_writer& grid, srsran::span<const srsran::cf_t> input, unsigned port_index, unsigned slot_index) override
  {
    const unsigned rg   = cfg.bw_rb * 12;
    auto*          d_in = static_cast<float*>(c->buf(0, input.size() * sizeof(srsran::cf_t)));
    auto*          d_g  = static_cast<float*>(c->buf(1, 14 * rg * sizeof(srsran::cf_t)));
    c->h2d(d_in, input.data(), input.size() * sizeof(srsran::cf_t));
    miphy_ofdm_job job = {0, 0, slot_index, 0};
    context::check(miphy_ofdm_demodulate_slots(c->ctx, &cfg, &job, 0, 1, d_in, d_g, c->stream), "ofdm_demodulate");
    host.resize(14 * rg);
    c->d2h(host.data(), d_g, host.size() * sizeof(srsran::cf_t));
    c->sync();
    for (unsigned l = 0; l != 14; ++l) {
      grid.put(port_index, l, 0, srsran::span<const srsran::cf_t>(host.data() + l * rg, rg));
    }
  }

private:
  std::shared_ptr<context>  c;
  miphy_ofdm_config         cfg = {};
  std::vector<srsran::cf_t> host;
};

/// srsran::ofdm_slot_modulator over miphy_ofdm_modulate_slots (ofdm_modulator.h:89-101).
class ofdm_slot_modulator_hip : public srsran::ofdm_slot_modulator
{
public:
  ofdm_slot_modulator_hip(std::shared_ptr<context> c, const srsran::ofdm_modulator_configuration& cfg_) : c(std::move(c))
  {
    cfg.numerology     = cfg_.numerology;
    cfg.bw_rb          = cfg_.bw_rb;
    cfg.dft_size       = cfg_.dft_size;
    cfg.scale          = cfg_.scale;
    cfg.center_freq_hz = cfg_.center_freq_hz;
    srsran_assert(cfg_.cp == srsran::cyclic_prefix::NORMAL, "Only normal cyclic prefix is supported by the HIP path.");
  }
  unsigned get_slot_size(unsigned slot_index) const override { return miphy_ofdm_slot_size(&cfg, slot_index); }
  void modulate(srsran::span<srsran::cf_t> output, const srsran::resource_grid_reader& grid, unsigned port_index, unsigned slot_index) override
  {
    const unsigned rg = cfg.bw_rb * 12;
    host.resize(14 * rg);
    miphy_ofdm_job job = {0, 0, slot_index, grid.is_empty(port_index) ? 1U : 0U};
    if (!job.grid_empty) {
      for (unsigned l = 0; l != 14; ++l) {
        grid.get(srsran::span<srsran::cf_t>(host.data() + l * rg, rg), port_index, l, 0);
      }
    }
    auto* d_g   = static_cast<float*>(c->buf(0, host.size() * sizeof(srsran::cf_t)));
    auto* d_out = static_cast<float*>(c->buf(1, output.size() * sizeof(srsran::cf_t)));
    c->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    context::check(miphy_ofdm_modulate_slots(c->ctx, &cfg, &job, 0, 1, d_g, d_out, c->stream), "ofdm_modulate");
    c->d2h(output.data(), d_out, output.size() * sizeof(srsran::cf_t));
    c->sync();
  }

private:
  std::shared_ptr<context>  c;
  miphy_ofdm_config         cfg = {};
  std::vector<srsran::cf_t> host;
};

/// srsran::ofdm_symbol_demodulator over miphy_ofdm_demodulate_symbols (ofdm_demodulator.h:55-74): what the lower PHY calls once per
/// received OFDM symbol and port (lib/phy/lower/processors/uplink/puxch/puxch_processor_impl.cpp:64-76). One symbol per call:
/// samples up, one transform, one row of subcarriers back into the grid before the call returns, as the interface requires.
class ofdm_symbol_demodulator_hip : public srsran::ofdm_symbol_demodulator
{
public:
  ofdm_symbol_demodulator_hip(std::shared_ptr<context> c, const srsran::ofdm_demodulator_configuration& cfg_) : c(std::move(c))
  {
    cfg.numerology                = cfg_.numerology;
    cfg.bw_rb                     = cfg_.bw_rb;
    cfg.dft_size                  = cfg_.dft_size;
    cfg.nof_samples_window_offset = cfg_.nof_samples_window_offset;
    cfg.scale                     = cfg_.scale;
    cfg.center_freq_hz            = cfg_.center_freq_hz;
    require(cfg_.cp == srsran::cyclic_prefix::NORMAL, "Only normal cyclic prefix is supported by the HIP path.");
  }
  unsigned get_symbol_size(unsigned symbol_index) const override { return miphy_ofdm_symbol_size(&cfg, symbol_index); }
  void demodulate(srsran::resource_grid_writer& grid, srsran::span<const srsran::cf_t> input, unsigned port_index, unsigned symbol_index) override
  {
    require(input.size() == get_symbol_size(symbol_index), "The input size is not consistent with the symbol size.");
    const unsigned rg   = cfg.bw_rb * 12;
    auto*          d_in = static_cast<float*>(c->buf(0, input.size() * sizeof(srsran::cf_t)));
    auto*          d_g  = static_cast<float*>(c->buf(1, rg * sizeof(srsran::cf_t)));
    c->h2d(d_in, input.data(), input.size() * sizeof(srsran::cf_t));
    miphy_ofdm_job job = {0, 0, symbol_index, 0};
    context::check(miphy_ofdm_demodulate_symbols(c->ctx, &cfg, &job, 0, 1, d_in, d_g, c->stream), "ofdm_demodulate_symbols");
    host.resize(rg);
    c->d2h(host.data(), d_g, host.size() * sizeof(srsran::cf_t));
    c->sync();
    grid.put(port_index, symbol_index % 14, 0, host); // the grid holds one slot (ofdm_demodulator_impl.cpp:133,137)
  }

private:
  std::shared_ptr<context>  c;
  miphy_ofdm_config         cfg = {};
  std::vector<srsran::cf_t> host;
};

/// srsran::ofdm_symbol_modulator over miphy_ofdm_modulate_symbols (ofdm_modulator.h:55-74; caller
/// lib/phy/lower/processors/downlink/pdxch/pdxch_processor_impl.cpp:76-82).
class ofdm_symbol_modulator_hip : public srsran::ofdm_symbol_modulator
{
public:
  ofdm_symbol_modulator_hip(std::shared_ptr<context> c, const srsran::ofdm_modulator_configuration& cfg_) : c(std::move(c))
  {
    cfg.numerology     = cfg_.numerology;
    cfg.bw_rb          = cfg_.bw_rb;
    cfg.dft_size       = cfg_.dft_size;
    cfg.scale          = cfg_.scale;
    cfg.center_freq_hz = cfg_.center_freq_hz;
    require(cfg_.cp == srsran::cyclic_prefix::NORMAL, "Only normal cyclic prefix is supported by the HIP path.");
  }
  unsigned get_symbol_size(unsigned symbol_index) const override { return miphy_ofdm_symbol_size(&cfg, symbol_index); }
  void modulate(srsran::span<srsran::cf_t> output, const srsran::resource_grid_reader& grid, unsigned port_index, unsigned symbol_index) override
  {
    require(output.size() == get_symbol_size(symbol_index), "The output size is not consistent with the symbol size.");
    const unsigned rg = cfg.bw_rb * 12;
    host.resize(rg);
    miphy_ofdm_job job = {0, 0, symbol_index, grid.is_empty(port_index) ? 1U : 0U};
    if (!job.grid_empty) {
      grid.get(host, port_index, symbol_index % 14, 0);
    }
    auto* d_g   = static_cast<float*>(c->buf(0, host.size() * sizeof(srsran::cf_t)));
    auto* d_out = static_cast<float*>(c->buf(1, output.size() * sizeof(srsran::cf_t)));
    c->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    context::check(miphy_ofdm_modulate_symbols(c->ctx, &cfg, &job, 0, 1, d_g, d_out, c->stream), "ofdm_modulate_symbols");
    c->d2h(output.data(), d_out, output.size() * sizeof(srsran::cf_t));
    c->sync();
  }

private:
  std::shared_ptr<context>  c;
  miphy_ofdm_config         cfg = {};
  std::vector<srsran::cf_t> host;
};

// ---------------------------------------------------------------------------------------------------------------- estimator
/// srsran::dmrs_pusch_estimator over miphy_dmrs_pusch_estimate_batch (dmrs_pusch_estimator.h:84).
class dmrs_pusch_estimator_hip : public srsran::dmrs_pusch_estimator
{
public:
  explicit dmrs_pusch_estimator_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void estimate(srsran::channel_estimate& estimate, const srsran::resource_grid_reader& grid, const configuration& config) override
  {
    srsran_assert(config.type == srsran::dmrs_type::TYPE1, "Only DM-RS type 1 is supported.");
    const unsigned nprb = config.rb_mask.size(), nsc = nprb * 12;
    const unsigned nports = config.rx_ports.size(), nl = config.nof_tx_layers, nsymb = config.first_symbol + config.nof_symbols;
    estimate.resize({nprb, nsymb, nports, nl});
    miphy_pusch_chest_job j = {};
    j.numerology            = config.slot.numerology();
    j.slot_in_frame         = config.slot.slot_index();
    j.scrambling_id         = config.scrambling_id;
    j.scaling               = config.scaling;
    j.n_scid                = config.n_scid;
    j.nof_tx_layers         = nl;
    j.nof_rx_ports          = nports;
    j.first_symbol          = config.first_symbol;
    j.nof_symbols           = config.nof_symbols;
    j.grid_nof_prb          = nprb;
    for (unsigned p = 0; p != nports; ++p) {
      j.rx_ports[p] = p; // the staging grid below is already ordered by rx_ports
    }
    for (unsigned l = 0; l != 14; ++l) {
      if (l < config.symbols_mask.size() && config.symbols_mask.test(l)) {
        j.symbols_mask |= static_cast<uint16_t>(1U << l);
      }
    }
    config.rb_mask.for_each(0, nprb, [&j](unsigned r) { j.rb_mask[r >> 6] |= 1ULL << (r & 63); });
    host.resize(static_cast<size_t>(nports) * 14 * nsc);
    for (unsigned p = 0; p != nports; ++p) {
      for (unsigned l = 0; l != 14; ++l) {
        grid.get(srsran::span<srsran::cf_t>(host.data() + (static_cast<size_t>(p) * 14 + l) * nsc, nsc), config.rx_ports[p], l, 0);
      }
    }
    const size_t ce_n = static_cast<size_t>(nl) * nports * nsymb * nsc;
    auto*        d_g  = static_cast<float*>(c->buf(0, host.size() * sizeof(srsran::cf_t)));
    auto*        d_ce = static_cast<float*>(c->buf(1, ce_n * sizeof(srsran::cf_t)));
    auto*        d_sc = static_cast<float*>(c->buf(2, nports * nl * 5 * sizeof(float)));
    c->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    ce_host.resize(ce_n);
    // Unallocated PRBs keep whatever the channel_estimate object holds (the reference only writes the allocation).
    for (unsigned ly = 0; ly != nl; ++ly) {
      for (unsigned p = 0; p != nports; ++p) {
        for (unsigned l = 0; l != nsymb; ++l) {
          auto v = estimate.get_symbol_ch_estimate(l, p, ly);
          std::memcpy(ce_host.data() + ((static_cast<size_t>(ly) * nports + p) * nsymb + l) * nsc, v.data(), nsc * sizeof(srsran::cf_t));
        }
      }
    }
    c->h2d(d_ce, ce_host.data(), ce_n * sizeof(srsran::cf_t));
    context::check(miphy_dmrs_pusch_estimate_batch(c->ctx, &j, 0, 1, d_g, d_ce, d_sc, c->stream), "dmrs_pusch_estimate");
    std::vector<float> sc(nports * nl * 5);
    c->d2h(ce_host.data(), d_ce, ce_n * sizeof(srsran::cf_t));
    c->d2h(sc.data(), d_sc, sc.size() * sizeof(float));
    c->sync();
    for (unsigned ly = 0; ly != nl; ++ly) {
      for (unsigned p = 0; p != nports; ++p) {
        for (unsigned l = 0; l != nsymb; ++l) {
          auto v = estimate.get_symbol_ch_estimate(l, p, ly);
          std::memcpy(v.data(), ce_host.data() + ((static_cast<size_t>(ly) * nports + p) * nsymb + l) * nsc, nsc * sizeof(srsran::cf_t));
        }
        const float* s = sc.data() + 5 * (static_cast<size_t>(p) * nl + ly);
        estimate.set_rsrp(s[0], p, ly);
        estimate.set_epre(s[1], p, ly);
        estimate.set_noise_variance(s[2], p, ly);
        estimate.set_snr(s[3], p, ly);
        estimate.set_time_alignment(srsran::phy_time_unit::from_seconds(s[4]), p, ly);
      }
    }
  }

private:
  std::shared_ptr<context>  c;
  std::vector<srsran::cf_t> host, ce_host;
};

// ---------------------------------------------------------------------------------------------------------------- PUSCH demodulator
/// srsran::pusch_demodulator over miphy_pusch_demodulate_batch (pusch_demodulator.h:105-108): equaliser, soft demapper and
/// descrambler of pusch_demodulator_impl in one device pass. No UCI placeholders, no EVM report (status.evm stays empty).
class pusch_demodulator_hip : public srsran::pusch_demodulator
{
public:
  explicit pusch_demodulator_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  demodulation_status demodulate(srsran::span<srsran::log_likelihood_ratio> codeword,
                                 const srsran::resource_grid_reader&        grid,
                                 const srsran::channel_estimate&            estimates,
                                 const configuration&                       config) override
  {
    srsran_assert(config.nof_tx_layers == 1, "Only a single transmit layer is supported.");
    bool has_placeholders = false;
    config.placeholders.for_each(config.modulation, config.nof_tx_layers, [&has_placeholders](unsigned, unsigned) { has_placeholders = true; });
    if (has_placeholders) {
      srsran::report_fatal_error("pusch_demodulator_hip: UCI placeholders are not supported.");
    }
    const unsigned nprb = config.rb_mask.size(), nsc = nprb * 12, nports = config.rx_ports.size();
    const unsigned nsymb = config.start_symbol_index + config.nof_symbols;
    miphy_pusch_demod_job j = {};
    j.rnti                  = config.rnti;
    j.n_id                  = config.n_id;
    j.mod                   = srsran::get_bits_per_symbol(config.modulation);
    j.nof_rx_ports          = nports;
    j.start_symbol          = config.start_symbol_index;
    j.nof_symbols           = config.nof_symbols;
    j.dmrs_type             = (config.dmrs_config_type == srsran::dmrs_type::TYPE1) ? 1 : 2;
    j.nof_cdm_groups_without_data = config.nof_cdm_groups_without_data;
    j.ce_nof_symbols        = nsymb;
    j.grid_nof_prb          = nprb;
    for (unsigned p = 0; p != nports; ++p) {
      j.rx_ports[p] = p; // the staging grid below is already ordered by rx_ports
    }
    for (unsigned l = 0; l != 14; ++l) {
      if (config.dmrs_symb_pos[l]) {
        j.dmrs_symbols_mask |= static_cast<uint16_t>(1U << l);
      }
    }
    config.rb_mask.for_each(0, nprb, [&j](unsigned r) { j.rb_mask[r >> 6] |= 1ULL << (r & 63); });
    j.nof_llr = codeword.size(); // validated against the allocation by the library, like pusch_demodulator_impl.cpp:76-80
    host.resize(static_cast<size_t>(nports) * 14 * nsc);
    for (unsigned p = 0; p != nports; ++p) {
      for (unsigned l = 0; l != 14; ++l) {
        grid.get(srsran::span<srsran::cf_t>(host.data() + (static_cast<size_t>(p) * 14 + l) * nsc, nsc), config.rx_ports[p], l, 0);
      }
    }
    ce_host.resize(static_cast<size_t>(nports) * nsymb * nsc);
    for (unsigned p = 0; p != nports; ++p) {
      for (unsigned l = 0; l != nsymb; ++l) {
        auto v = estimates.get_symbol_ch_estimate(l, p, 0);
        std::memcpy(ce_host.data() + (static_cast<size_t>(p) * nsymb + l) * nsc, v.data(), nsc * sizeof(srsran::cf_t));
      }
    }
    float sc[5] = {0, 0, estimates.get_noise_variance(0, 0), 0, 0};
    auto* d_g   = static_cast<float*>(c->buf(0, host.size() * sizeof(srsran::cf_t)));
    auto* d_ce  = static_cast<float*>(c->buf(1, ce_host.size() * sizeof(srsran::cf_t)));
    auto* d_sc  = static_cast<float*>(c->buf(2, sizeof(sc)));
    auto* d_llr = static_cast<int8_t*>(c->buf(3, codeword.size() + 16));
    c->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    c->h2d(d_ce, ce_host.data(), ce_host.size() * sizeof(srsran::cf_t));
    c->h2d(d_sc, sc, sizeof(sc));
    context::check(miphy_pusch_demodulate_batch(c->ctx, &j, 0, 1, d_g, d_ce, d_sc, d_llr, c->stream), "pusch_demodulate");
    static_assert(sizeof(srsran::log_likelihood_ratio) == 1, "LLRs are int8");
    c->d2h(codeword.data(), d_llr, codeword.size());
    c->sync();
    return {};
  }

private:
  std::shared_ptr<context>  c;
  std::vector<srsran::cf_t> host, ce_host;
};

// ---------------------------------------------------------------------------------------------------------------- PUSCH processor
/// srsran::pusch_processor over miphy_pusch_process_batch (pusch_processor.h:158-162): estimation, demodulation and decoding in one
/// device pass for PDUs that carry a codeword and no UCI; the resource grid goes to the device once, the transport block, the
/// HARQ buffer of the rx_softbuffer and the channel state information come back.
class pusch_processor_hip : public srsran::pusch_processor
{
public:
  /// \c uci_dec: the reference's UCI decoder (short block / polar decoding of the demultiplexed soft bits stays a CPU block);
  /// without it PDUs with multiplexed UCI are refused. \c enable_evm as in create_pusch_demodulator_factory_sw.
  pusch_processor_hip(std::shared_ptr<context> c, unsigned nof_iterations, bool early_stop, std::unique_ptr<srsran::uci_decoder> uci_dec_ = nullptr,
                      bool enable_evm_ = false) :
    c(std::move(c)), dec_nof_iterations(nof_iterations), dec_enable_early_stop(early_stop), uci_dec(std::move(uci_dec_)), enable_evm(enable_evm_)
  {
  }
  void process(srsran::span<uint8_t>                    data,
               srsran::rx_softbuffer&                   softbuffer,
               srsran::pusch_processor_result_notifier& notifier,
               const srsran::resource_grid_reader&      grid,
               const pdu_t&                             pdu) override
  {
    const bool has_uci = pdu.uci.nof_harq_ack != 0 || pdu.uci.nof_csi_part1 != 0 || pdu.uci.nof_csi_part2 != 0;
    require(!has_uci || uci_dec != nullptr, "pusch_processor_hip: a PDU with multiplexed UCI needs the UCI decoder (see the factory).");
    require(has_uci || pdu.codeword.has_value(), "pusch_processor_hip: the PDU carries neither a codeword nor UCI.");
    require(pdu.dmrs == srsran::dmrs_type::TYPE1 && pdu.nof_cdm_groups_without_data == 2 && pdu.nof_tx_layers == 1,
            "Only DM-RS type 1, two CDM groups without data and one layer are supported.");
    const srsran::bounded_bitset<srsran::MAX_RB> rb_mask = pdu.freq_alloc.get_prb_mask(pdu.bwp_start_rb, pdu.bwp_size_rb);
    const unsigned nprb = rb_mask.size(), nsc = nprb * 12, nports = pdu.rx_ports.size();
    miphy_pusch_pdu p = {};
    p.numerology = pdu.slot.numerology(), p.slot_in_frame = pdu.slot.slot_index(), p.rnti = pdu.rnti, p.n_id = pdu.n_id;
    p.dmrs_scrambling_id = pdu.scrambling_id, p.Nref = pdu.tbs_lbrm_bytes * 8, p.tb_bytes = data.size(), p.harq_cb_index = 0;
    p.n_scid = pdu.n_scid, p.mod = srsran::get_bits_per_symbol(pdu.mcs_descr.modulation), p.nof_rx_ports = nports;
    p.start_symbol = pdu.start_symbol_index, p.nof_symbols = pdu.nof_symbols;
    if (pdu.codeword.has_value()) {
      p.bg = bg_id(pdu.codeword.value().ldpc_base_graph), p.rv = pdu.codeword.value().rv, p.new_data = pdu.codeword.value().new_data;
    } else {
      p.bg = 1, p.tb_bytes = 1; // unused: no transport block is decoded
    }
    p.use_early_stop = dec_enable_early_stop, p.nof_ldpc_iterations = dec_nof_iterations, p.grid_nof_prb = nprb;
    for (unsigned i = 0; i != nports; ++i) {
      p.rx_ports[i] = i; // the staging grid is ordered by rx_ports
    }
    for (unsigned l = 0; l != 14 && l != pdu.dmrs_symbol_mask.size(); ++l) {
      if (pdu.dmrs_symbol_mask.test(l)) {
        p.dmrs_symbols_mask |= static_cast<uint16_t>(1U << l);
      }
    }
    rb_mask.for_each(0, nprb, [&p](unsigned r) { p.rb_mask[r >> 6] |= 1ULL << (r & 63); });
    // UCI lengths through the reference's own get_ulsch_information (pusch_processor_impl.cpp:152-170)
    miphy_pusch_uci u = {};
    u.has_codeword    = pdu.codeword.has_value() ? 1U : 0U;
    if (has_uci) {
      srsran::ulsch_configuration uc;
      uc.tbs = srsran::units::bytes(data.size()).to_bits(), uc.mcs_descr = pdu.mcs_descr;
      uc.nof_harq_ack_bits = srsran::units::bits(pdu.uci.nof_harq_ack), uc.nof_csi_part1_bits = srsran::units::bits(pdu.uci.nof_csi_part1);
      uc.nof_csi_part2_bits = srsran::units::bits(pdu.uci.nof_csi_part2), uc.alpha_scaling = pdu.uci.alpha_scaling;
      uc.beta_offset_harq_ack = pdu.uci.beta_offset_harq_ack, uc.beta_offset_csi_part1 = pdu.uci.beta_offset_csi_part1;
      uc.beta_offset_csi_part2 = pdu.uci.beta_offset_csi_part2, uc.nof_rb = pdu.freq_alloc.get_nof_rb();
      uc.start_symbol_index = pdu.start_symbol_index, uc.nof_symbols = pdu.nof_symbols, uc.dmrs_type = srsran::dmrs_config_type::type1;
      uc.dmrs_symbol_mask = pdu.dmrs_symbol_mask, uc.nof_cdm_groups_without_data = pdu.nof_cdm_groups_without_data, uc.nof_layers = pdu.nof_tx_layers;
      const srsran::ulsch_information info = srsran::get_ulsch_information(uc);
      u.nof_harq_ack_bits = pdu.uci.nof_harq_ack, u.nof_csi_part1_bits = pdu.uci.nof_csi_part1, u.nof_csi_part2_bits = pdu.uci.nof_csi_part2;
      u.nof_enc_harq_ack_bits = info.nof_harq_ack_bits.value(), u.nof_enc_csi_part1_bits = info.nof_csi_part1_bits.value();
      u.nof_enc_csi_part2_bits = info.nof_csi_part2_bits.value(), u.nof_harq_ack_rvd = info.nof_harq_ack_rvd.value();
      u.harq_ack_offset = 0, u.csi_part1_offset = u.nof_enc_harq_ack_bits, u.csi_part2_offset = u.csi_part1_offset + u.nof_enc_csi_part1_bits;
    }
    const size_t nof_uci_llr = static_cast<size_t>(u.nof_enc_harq_ack_bits) + u.nof_enc_csi_part1_bits + u.nof_enc_csi_part2_bits;
    miphy_sch_segmentation sg = {};
    if (pdu.codeword.has_value()) {
      context::check(miphy_sch_segmentation_info(data.size(), p.bg, &sg), "segmentation");
      require(sg.nof_cbs == softbuffer.get_nof_codeblocks(), "Wrong number of codeblocks.");
    }
    const size_t CBS = 66 * 384, MSG = 1056;
    host.resize(static_cast<size_t>(nports) * 14 * nsc);
    for (unsigned i = 0; i != nports; ++i) {
      for (unsigned l = 0; l != 14; ++l) {
        grid.get(srsran::span<srsran::cf_t>(host.data() + (static_cast<size_t>(i) * 14 + l) * nsc, nsc), pdu.rx_ports[i], l, 0);
      }
    }
    auto*    resident = dynamic_cast<rx_softbuffer_hip*>(&softbuffer);
    auto*    d_g    = static_cast<float*>(c->buf(0, host.size() * sizeof(srsran::cf_t)));
    auto*    d_misc = static_cast<uint8_t*>(c->buf(3, 64 + 64 + 128 + data.size() + 64));
    auto*    d_uci  = static_cast<int8_t*>(c->buf(4, nof_uci_llr + 64));
    auto*    d_evm  = reinterpret_cast<float*>(d_misc + 56); // [0,52) codeblock CRC flags of a host softbuffer, [56,60) EVM, [64,..) result
    auto*    d_res  = reinterpret_cast<miphy_pusch_result*>(d_misc + 64);
    auto*    d_sc   = reinterpret_cast<float*>(d_misc + 128);
    uint8_t* d_tb   = d_misc + 256;
    int8_t*  d_soft = nullptr;
    uint8_t *d_msg = nullptr, *d_crc = nullptr;
    std::vector<uint8_t> crc_h(sg.nof_cbs);
    if (resident != nullptr) { // softbuffer of rx_softbuffer_pool_hip: the HARQ state is used in place
      resident->flush();
      d_soft = resident->softbits(), d_msg = resident->msgs(), d_crc = resident->crc_ok();
      p.harq_cb_index = resident->first_cb();
    } else {
      d_soft = static_cast<int8_t*>(c->buf(1, sg.nof_cbs * CBS));
      d_msg  = static_cast<uint8_t*>(c->buf(2, sg.nof_cbs * MSG));
      d_crc  = d_misc;
      srsran::span<bool> crcs = softbuffer.get_codeblocks_crc();
      for (unsigned i = 0; i != sg.nof_cbs; ++i) {
        crc_h[i] = crcs[i] ? 1 : 0;
        auto sb  = softbuffer.get_codeblock_soft_bits(i, sg.N);
        c->h2d(d_soft + i * CBS, sb.data(), sg.N);
        auto mb = softbuffer.get_codeblock_data_bits(i, sg.K);
        c->h2d(d_msg + i * MSG, mb.get_buffer().data(), (sg.K + 7) / 8);
      }
      c->h2d(d_crc, crc_h.data(), sg.nof_cbs);
    }
    c->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    c->h2d(d_tb, data.data(), data.size());
    context::check(miphy_pusch_process_batch_ex(c->ctx, &p, &u, 1, d_g, d_soft, d_msg, d_crc, d_tb, d_res, d_sc, d_uci, enable_evm ? d_evm : nullptr, c->stream),
                   "pusch_process");
    miphy_pusch_result r;
    float              sc[20], evm = 0.F;
    uci_llr.resize(nof_uci_llr);
    if (nof_uci_llr) {
      c->d2h(uci_llr.data(), d_uci, nof_uci_llr);
    }
    if (enable_evm) {
      c->d2h(&evm, d_evm, sizeof(evm));
    }
    c->d2h(&r, d_res, sizeof(r));
    c->d2h(sc, d_sc, sizeof(sc));
    c->d2h(data.data(), d_tb, data.size());
    if (resident == nullptr) {
      c->d2h(crc_h.data(), d_crc, sg.nof_cbs);
      for (unsigned i = 0; i != sg.nof_cbs; ++i) {
        auto sb = softbuffer.get_codeblock_soft_bits(i, sg.N);
        c->d2h(sb.data(), d_soft + i * CBS, sg.N);
        auto mb = softbuffer.get_codeblock_data_bits(i, sg.K);
        c->d2h(mb.get_buffer().data(), d_msg + i * MSG, (sg.K + 7) / 8);
      }
    }
    c->sync();
    if (resident == nullptr) {
      srsran::span<bool> crcs = softbuffer.get_codeblocks_crc();
      for (unsigned i = 0; i != sg.nof_cbs; ++i) {
        crcs[i] = crc_h[i] != 0;
      }
    }
    // channel_estimate::get_channel_state_information (channel_estimation.h:211-232): linear averages over the receive ports, then dB
    srsran::channel_state_information csi = {};
    float                             epre = 0, rsrp = 0, snr = 0;
    double                            ta   = 0;
    for (unsigned i = 0; i != nports; ++i) {
      rsrp += sc[5 * i + 0], epre += sc[5 * i + 1], snr += sc[5 * i + 3], ta += sc[5 * i + 4];
    }
    csi.epre_dB        = srsran::convert_power_to_dB(epre / static_cast<float>(nports));
    csi.rsrp_dB        = srsran::convert_power_to_dB(rsrp / static_cast<float>(nports));
    csi.sinr_dB        = srsran::convert_power_to_dB(snr / static_cast<float>(nports));
    csi.time_alignment = srsran::phy_time_unit::from_seconds(ta / nports);
    if (enable_evm) {
      csi.sinr_dB = -20 * log10f(evm) - 3.7F; // pusch_processor_impl.cpp:237-241
    }
    notifier.on_csi(csi);
    if (has_uci) { // pusch_processor_impl.cpp:264-286: the three fields through the reference's UCI decoder
      srsran::uci_decoder::configuration ucfg;
      ucfg.modulation = pdu.mcs_descr.modulation;
      srsran::pusch_processor_result_control ru;
      if (enable_evm) {
        ru.evm.emplace(evm);
      }
      auto field = [&](size_t off, unsigned G, unsigned nof_bits) {
        srsran::pusch_uci_field f;
        if (nof_bits == 0) {
          f.payload.clear();
          f.status = srsran::uci_status::unknown;
          return f;
        }
        f.payload.resize(nof_bits);
        f.status = uci_dec->decode(f.payload, srsran::span<const srsran::log_likelihood_ratio>(uci_llr.data() + off, G), ucfg);
        return f;
      };
      ru.harq_ack  = field(u.harq_ack_offset, u.nof_enc_harq_ack_bits, pdu.uci.nof_harq_ack);
      ru.csi_part1 = field(u.csi_part1_offset, u.nof_enc_csi_part1_bits, pdu.uci.nof_csi_part1);
      ru.csi_part2 = field(u.csi_part2_offset, u.nof_enc_csi_part2_bits, pdu.uci.nof_csi_part2);
      notifier.on_uci(ru);
    }
    if (!pdu.codeword.has_value()) {
      return;
    }
    srsran::pusch_processor_result_data result;
    if (enable_evm) {
      result.evm.emplace(evm);
    }
    result.data.tb_crc_ok            = r.tb_crc_ok != 0;
    result.data.nof_codeblocks_total = r.nof_codeblocks_total;
    result.data.ldpc_decoder_stats.reset();
    if (r.nof_decoded > 0) {
      result.data.ldpc_decoder_stats.update(r.iters_min);
      for (unsigned i = 1; i + 1 < r.nof_decoded; ++i) {
        result.data.ldpc_decoder_stats.update(static_cast<unsigned>(r.iters_mean + 0.5F));
      }
      if (r.nof_decoded > 1) {
        result.data.ldpc_decoder_stats.update(r.iters_max);
      }
    }
    notifier.on_sch(result);
  }

private:
  std::shared_ptr<context>  c;
  unsigned                  dec_nof_iterations;
  bool                      dec_enable_early_stop;
  std::unique_ptr<srsran::uci_decoder>      uci_dec;
  bool                                      enable_evm;
  std::vector<srsran::cf_t>                 host;
  std::vector<srsran::log_likelihood_ratio> uci_llr;
};

/// Replaces create_pusch_processor_factory_sw(config) (channel_processor_factories.h): only the decoder settings of the
/// configuration are needed, the sub-block factories are not.
class pusch_processor_factory_hip : public srsran::pusch_processor_factory
{
public:
  /// \c uci_dec_factory: the reference's create_uci_decoder_factory_sw(...) for PDUs with multiplexed UCI (nullptr: such PDUs are
  /// rejected by the validator); \c enable_evm as in create_pusch_demodulator_factory_sw.
  pusch_processor_factory_hip(std::shared_ptr<context>                     c,
                              unsigned                                     nof_iterations,
                              bool                                         early_stop,
                              std::shared_ptr<srsran::uci_decoder_factory> uci_dec_factory = nullptr,
                              bool                                         enable_evm      = false) :
    c(std::move(c)), nof_iterations(nof_iterations), early_stop(early_stop), uci_dec_factory(std::move(uci_dec_factory)), enable_evm(enable_evm)
  {
  }
  std::unique_ptr<srsran::pusch_processor> create() override
  {
    return std::make_unique<pusch_processor_hip>(c, nof_iterations, early_stop, uci_dec_factory ? uci_dec_factory->create() : nullptr, enable_evm);
  }
  std::unique_ptr<srsran::pusch_pdu_validator> create_validator() override; // defined at the end of this header

private:
  std::shared_ptr<context>                     c;
  unsigned                                     nof_iterations;
  bool                                         early_stop;
  std::shared_ptr<srsran::uci_decoder_factory> uci_dec_factory;
  bool                                         enable_evm;
};

// ---------------------------------------------------------------------------------------------------------------- uplink processor
/// srsran::uplink_processor (uplink_processor.h:52-98, lib/phy/upper/uplink_processor_impl.cpp:143-214) that batches the PUSCH
/// PDUs of a slot into one device submission: process_pusch() only queues the PDU, flush() uploads the resource grid once,
/// runs miphy_pusch_process_batch over all queued PDUs and then notifies the results in the order the PDUs were queued, with
/// the notifications of uplink_processor_impl (channel state information, decoder result, payload view only when the CRC
/// passed, softbuffer released when the CRC passed). PRACH detection and PUCCH processing are delegated to the CPU blocks given
/// at construction. The softbuffers must come from rx_softbuffer_pool_hip (HARQ state resident on the device); a PDU with any
/// other softbuffer is processed at once through the per-PDU path.
///
/// The interface has no end-of-slot call: the caller invokes flush() after its loop over the PDUs of the slot
/// (upper_phy_rx_symbol_handler_impl.cpp:97-107); a PDU of another slot, a full queue and the destructor flush as well.
class uplink_processor_hip : public srsran::uplink_processor
{
public:
  /// \c linger_us > 0 (default): asynchronous delivery. process_pusch() returns at once; a delivery thread submits the PDUs of a slot
  /// as one device batch as soon as no further PDU has arrived for \c linger_us microseconds (or the batch is full / a PDU of another
  /// slot arrives) and calls the notifier from that thread -- the reference notifies from its executor threads as well. No end-of-slot
  /// call is needed, so the unmodified caller (upper_phy_rx_symbol_handler_impl.cpp:92-105) works. \c linger_us == 0: synchronous
  /// mode, results are delivered by flush() (or by the next PDU of another slot / the destructor).
  /// \c uci_dec_factory / \c enable_evm: for PDUs with multiplexed UCI, which take the per-PDU path (pusch_processor_hip).
  uplink_processor_hip(std::shared_ptr<context>                     c,
                       std::unique_ptr<srsran::prach_detector>      prach,
                       std::unique_ptr<srsran::pucch_processor>     pucch,
                       unsigned                                     grid_nof_ports,
                       unsigned                                     grid_nof_prb,
                       unsigned                                     dec_nof_iterations,
                       bool                                         dec_enable_early_stop,
                       unsigned                                     max_batch       = 64,
                       unsigned                                     linger_us       = 100,
                       std::shared_ptr<srsran::uci_decoder_factory> uci_dec_factory = nullptr,
                       bool                                         enable_evm      = false) :
    c(std::move(c)),
    prach(std::move(prach)),
    pucch(std::move(pucch)),
    single(this->c, dec_nof_iterations, dec_enable_early_stop, uci_dec_factory ? uci_dec_factory->create() : nullptr, enable_evm),
    nports(grid_nof_ports),
    nprb(grid_nof_prb),
    nof_iterations(dec_nof_iterations),
    early_stop(dec_enable_early_stop),
    max_batch(max_batch),
    linger(std::chrono::microseconds(linger_us))
  {
    if (linger_us != 0) {
      wc     = std::make_shared<context>(this->c->device); // the delivery thread drives the caller's device through its own context
      this->c->bind_thread();                              // (creating it left wc's device current, which is the same one)
      worker = std::thread([this]() { deliver_loop(); });
    }
  }
  ~uplink_processor_hip() override
  {
    flush();
    if (worker.joinable()) {
      {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
      }
      cv.notify_all();
      worker.join();
    }
  }

  void process_prach(srsran::upper_phy_rx_results_notifier& notifier, const srsran::prach_buffer& buffer, const srsran::prach_buffer_context& context) override
  {
    srsran_assert(prach, "A PRACH detector is required.");
    srsran::ul_prach_results r;
    r.context = context;
    srsran::prach_detector::configuration cfg;
    cfg.root_sequence_index = context.root_sequence_index, cfg.format = context.format, cfg.restricted_set = context.restricted_set;
    cfg.zero_correlation_zone = context.zero_correlation_zone, cfg.start_preamble_index = context.start_preamble_index;
    cfg.nof_preamble_indices = context.nof_preamble_indices, cfg.ra_scs = srsran::to_ra_subcarrier_spacing(context.pusch_scs);
    r.result = prach->detect(buffer, cfg);
    notifier.on_new_prach_results(r);
  }

  void process_pucch(srsran::upper_phy_rx_results_notifier& notifier, const srsran::resource_grid_reader& grid, const pucch_pdu& pdu) override
  {
    srsran_assert(pucch, "A PUCCH processor is required.");
    srsran::ul_pucch_results r;
    r.context = pdu.context;
    switch (pdu.context.format) {
      case srsran::pucch_format::FORMAT_0:
        r.processor_result = pucch->process(grid, pdu.format0);
        break;
      case srsran::pucch_format::FORMAT_1:
        r.processor_result = pucch->process(grid, pdu.format1);
        break;
      case srsran::pucch_format::FORMAT_2:
        r.processor_result = pucch->process(grid, pdu.format2);
        break;
      case srsran::pucch_format::FORMAT_3:
        r.processor_result = pucch->process(grid, pdu.format3);
        break;
      default:
        r.processor_result = pucch->process(grid, pdu.format4);
        break;
    }
    notifier.on_new_pucch_results(r);
  }

  void process_pusch(srsran::span<uint8_t>                  data,
                     srsran::unique_rx_softbuffer           softbuffer,
                     srsran::upper_phy_rx_results_notifier& notifier,
                     const srsran::resource_grid_reader&    grid,
                     const pusch_pdu&                       pdu) override
  {
    auto*      resident = dynamic_cast<rx_softbuffer_hip*>(&softbuffer.get());
    const bool has_uci  = pdu.pdu.uci.nof_harq_ack != 0 || pdu.pdu.uci.nof_csi_part1 != 0 || pdu.pdu.uci.nof_csi_part2 != 0;
    if (resident == nullptr || has_uci || !pdu.pdu.codeword.has_value()) {
      // not a device softbuffer, or UCI to decode on the host: per-PDU path at once, like uplink_processor_impl::process_pusch
      notifier_adaptor n(notifier, pdu, data);
      single.process(data, softbuffer.get(), n, grid, pdu.pdu);
      if (n.tb_crc_ok) {
        softbuffer.release();
      }
      return;
    }
    std::unique_lock<std::mutex> lk(mu);
    if (!open.entries.empty() && (open.entries.front().pdu.pdu.slot != pdu.pdu.slot || open.grid != &grid || open.entries.size() >= max_batch)) {
      close_open_batch(lk);
    }
    if (open.entries.empty()) {
      // The samples of the slot are copied NOW: the resource-grid pool may hand the grid to another slot before the batch is
      // submitted, and a late submission must be slow rather than wrong.
      const unsigned nsc = nprb * 12;
      open.grid          = &grid;
      open.samples.resize(static_cast<size_t>(nports) * 14 * nsc);
      for (unsigned p = 0; p != nports; ++p) {
        for (unsigned l = 0; l != 14; ++l) {
          grid.get(srsran::span<srsran::cf_t>(open.samples.data() + (static_cast<size_t>(p) * 14 + l) * nsc, nsc), p, l, 0);
        }
      }
    }
    open.entries.emplace_back(entry{data, std::move(softbuffer), resident, &notifier, pdu});
    last_push = std::chrono::steady_clock::now();
    lk.unlock();
    cv.notify_all();
  }

  /// Submits what is queued and returns when every result has been delivered (asynchronous mode: waits for the delivery thread).
  void flush()
  {
    std::unique_lock<std::mutex> lk(mu);
    if (!open.entries.empty()) {
      close_open_batch(lk);
    }
    if (worker.joinable()) {
      cv.notify_all();
      cv_done.wait(lk, [this]() { return ready.empty() && !busy; });
    }
  }

private:
  struct entry;
  struct batch;
  // Moves the open batch to the ready list (asynchronous mode) or runs it in place (synchronous mode). Called with the lock held.
  void close_open_batch(std::unique_lock<std::mutex>& lk)
  {
    batch b = std::move(open);
    open    = batch{};
    if (worker.joinable()) {
      ready.emplace_back(std::move(b));
      return;
    }
    lk.unlock();
    run_batch(b, *c);
    lk.lock();
  }
  void deliver_loop()
  {
    wc->bind_thread(); // a new thread starts on device 0
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      if (ready.empty() && !open.entries.empty()) {
        // wait for the linger time of the open batch, then close it
        if (!cv.wait_until(lk, last_push + linger, [this]() { return stop || !ready.empty() || std::chrono::steady_clock::now() >= last_push + linger; })) {
          continue;
        }
        if (ready.empty() && !open.entries.empty() && std::chrono::steady_clock::now() >= last_push + linger) {
          ready.emplace_back(std::move(open));
          open = batch{};
        }
      } else if (ready.empty()) {
        if (stop) {
          return;
        }
        cv.wait(lk, [this]() { return stop || !ready.empty() || !open.entries.empty(); });
        continue;
      }
      while (!ready.empty()) {
        batch b = std::move(ready.front());
        ready.pop_front();
        busy = true;
        lk.unlock();
        run_batch(b, *wc);
        lk.lock();
        busy = false;
      }
      cv_done.notify_all();
    }
  }
  void run_batch(batch& bt, context& dc)
  {
    std::vector<entry>& queue = bt.entries;
    if (queue.empty()) {
      return;
    }
    const unsigned n = queue.size();
    std::vector<srsran::cf_t>& host = bt.samples;
    std::vector<miphy_pusch_pdu> pdus(n);
    size_t                       tb_bytes = 0;
    int8_t*                      d_soft   = queue.front().resident->softbits();
    uint8_t *                    d_msg = queue.front().resident->msgs(), *d_crc = queue.front().resident->crc_ok();
    for (unsigned i = 0; i != n; ++i) {
      entry&                                  e   = queue[i];
      const srsran::pusch_processor::pdu_t&   pdu = e.pdu.pdu;
      require(e.resident->softbits() == d_soft, "All softbuffers of a batch must belong to the same device pool.");
      require(pdu.dmrs == srsran::dmrs_type::TYPE1 && pdu.nof_cdm_groups_without_data == 2 && pdu.nof_tx_layers == 1,
              "Only DM-RS type 1, two CDM groups without data and one layer are supported.");
      const srsran::bounded_bitset<srsran::MAX_RB> rb_mask = pdu.freq_alloc.get_prb_mask(pdu.bwp_start_rb, pdu.bwp_size_rb);
      require(rb_mask.size() <= nprb, "The allocation exceeds the resource grid.");
      e.resident->flush();
      miphy_pusch_pdu& p = pdus[i];
      p                  = {};
      p.numerology = pdu.slot.numerology(), p.slot_in_frame = pdu.slot.slot_index(), p.rnti = pdu.rnti, p.n_id = pdu.n_id;
      p.dmrs_scrambling_id = pdu.scrambling_id, p.Nref = pdu.tbs_lbrm_bytes * 8, p.tb_bytes = e.data.size();
      p.harq_cb_index = e.resident->first_cb();
      p.n_scid = pdu.n_scid, p.mod = srsran::get_bits_per_symbol(pdu.mcs_descr.modulation), p.nof_rx_ports = pdu.rx_ports.size();
      p.start_symbol = pdu.start_symbol_index, p.nof_symbols = pdu.nof_symbols;
      p.bg = bg_id(pdu.codeword.value().ldpc_base_graph), p.rv = pdu.codeword.value().rv, p.new_data = pdu.codeword.value().new_data;
      p.use_early_stop = early_stop, p.nof_ldpc_iterations = nof_iterations, p.grid_nof_prb = nprb;
      for (unsigned k = 0; k != pdu.rx_ports.size(); ++k) {
        require(pdu.rx_ports[k] < nports, "Receive port outside the resource grid.");
        p.rx_ports[k] = pdu.rx_ports[k];
      }
      for (unsigned l = 0; l != 14 && l != pdu.dmrs_symbol_mask.size(); ++l) {
        if (pdu.dmrs_symbol_mask.test(l)) {
          p.dmrs_symbols_mask |= static_cast<uint16_t>(1U << l);
        }
      }
      rb_mask.for_each(0, rb_mask.size(), [&p](unsigned r) { p.rb_mask[r >> 6] |= 1ULL << (r & 63); });
      p.grid_offset = 0, p.tb_offset = tb_bytes;
      tb_bytes += (e.data.size() + 15) & ~static_cast<size_t>(15);
    }
    auto*    d_g    = static_cast<float*>(dc.buf(0, host.size() * sizeof(srsran::cf_t)));
    auto*    d_tb   = static_cast<uint8_t*>(dc.buf(1, tb_bytes + 16));
    auto*    d_misc = static_cast<uint8_t*>(dc.buf(2, static_cast<size_t>(n) * (sizeof(miphy_pusch_result) + 80) + 64));
    auto*    d_res  = reinterpret_cast<miphy_pusch_result*>(d_misc);
    auto*    d_sc   = reinterpret_cast<float*>(d_misc + ((static_cast<size_t>(n) * sizeof(miphy_pusch_result) + 63) & ~static_cast<size_t>(63)));
    dc.h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    context::check(miphy_pusch_process_batch(dc.ctx, pdus.data(), n, d_g, d_soft, d_msg, d_crc, d_tb, d_res, d_sc, dc.stream), "pusch_process");
    std::vector<miphy_pusch_result> res(n);
    std::vector<float>              sc(static_cast<size_t>(n) * 20);
    std::vector<uint8_t>            tbs(tb_bytes);
    dc.d2h(res.data(), d_res, n * sizeof(miphy_pusch_result));
    dc.d2h(sc.data(), d_sc, sc.size() * sizeof(float));
    dc.d2h(tbs.data(), d_tb, tb_bytes);
    dc.sync();
    for (unsigned i = 0; i != n; ++i) {
      entry&                      e = queue[i];
      const miphy_pusch_result&   r = res[i];
      srsran::ul_pusch_results_data out;
      out.rnti = srsran::to_rnti(e.pdu.pdu.rnti), out.slot = e.pdu.pdu.slot, out.harq_id = e.pdu.harq_id;
      out.csi = csi_of(&sc[static_cast<size_t>(i) * 20], e.pdu.pdu.rx_ports.size());
      fill_decoder_result(out.decoder_result, r);
      if (r.tb_crc_ok != 0) {
        std::memcpy(e.data.data(), &tbs[pdus[i].tb_offset], e.data.size());
        out.payload = e.data;
      }
      e.notifier->on_new_pusch_results_data(out);
      if (r.tb_crc_ok != 0) {
        e.softbuffer.release();
      }
    }
    queue.clear(); // the remaining unique_rx_softbuffers unlock here
  }

  /// channel_estimate::get_channel_state_information (channel_estimation.h:211-232) from the estimator scalars of a PDU.
  static srsran::channel_state_information csi_of(const float* sc, unsigned nof_ports)
  {
    srsran::channel_state_information csi = {};
    float                             epre = 0, rsrp = 0, snr = 0;
    double                            ta   = 0;
    for (unsigned i = 0; i != nof_ports; ++i) {
      rsrp += sc[5 * i + 0], epre += sc[5 * i + 1], snr += sc[5 * i + 3], ta += sc[5 * i + 4];
    }
    csi.epre_dB        = srsran::convert_power_to_dB(epre / static_cast<float>(nof_ports));
    csi.rsrp_dB        = srsran::convert_power_to_dB(rsrp / static_cast<float>(nof_ports));
    csi.sinr_dB        = srsran::convert_power_to_dB(snr / static_cast<float>(nof_ports));
    csi.time_alignment = srsran::phy_time_unit::from_seconds(ta / nof_ports);
    return csi;
  }
  static void fill_decoder_result(srsran::pusch_decoder_result& out, const miphy_pusch_result& r)
  {
    out.tb_crc_ok            = r.tb_crc_ok != 0;
    out.nof_codeblocks_total = r.nof_codeblocks_total;
    out.ldpc_decoder_stats.reset();
    if (r.nof_decoded > 0) {
      out.ldpc_decoder_stats.update(r.iters_min);
      for (unsigned i = 1; i + 1 < r.nof_decoded; ++i) {
        out.ldpc_decoder_stats.update(static_cast<unsigned>(r.iters_mean + 0.5F));
      }
      if (r.nof_decoded > 1) {
        out.ldpc_decoder_stats.update(r.iters_max);
      }
    }
  }

private:
  /// pusch_processor_result_notifier_adaptor of uplink_processor_impl.cpp:41-105 for the per-PDU path.
  class notifier_adaptor : public srsran::pusch_processor_result_notifier
  {
  public:
    notifier_adaptor(srsran::upper_phy_rx_results_notifier& n, const pusch_pdu& pdu, srsran::span<const uint8_t> payload) : n(n), pdu(pdu), payload(payload) {}
    void on_csi(const srsran::channel_state_information& v) override { csi = v; }
    void on_uci(const srsran::pusch_processor_result_control& /**/) override {}
    void on_sch(const srsran::pusch_processor_result_data& sch) override
    {
      srsran::ul_pusch_results_data out;
      out.rnti = srsran::to_rnti(pdu.pdu.rnti), out.slot = pdu.pdu.slot, out.csi = csi, out.harq_id = pdu.harq_id;
      out.decoder_result = sch.data;
      out.payload        = sch.data.tb_crc_ok ? payload : srsran::span<const uint8_t>();
      n.on_new_pusch_results_data(out);
      tb_crc_ok = sch.data.tb_crc_ok;
    }
    bool tb_crc_ok = false;

  private:
    srsran::upper_phy_rx_results_notifier& n;
    const pusch_pdu&                       pdu;
    srsran::span<const uint8_t>            payload;
    srsran::channel_state_information      csi = {};
  };
  struct entry {
    srsran::span<uint8_t>                  data;
    srsran::unique_rx_softbuffer           softbuffer;
    rx_softbuffer_hip*                     resident;
    srsran::upper_phy_rx_results_notifier* notifier;
    pusch_pdu                              pdu;
  };

  struct batch {
    std::vector<entry>                  entries;
    const srsran::resource_grid_reader* grid = nullptr;
    std::vector<srsran::cf_t>           samples; // the slot's resource grid, copied when the batch was opened
  };

  std::shared_ptr<context>                 c, wc;
  std::unique_ptr<srsran::prach_detector>  prach;
  std::unique_ptr<srsran::pucch_processor> pucch;
  pusch_processor_hip                      single;
  unsigned                                 nports, nprb, nof_iterations;
  bool                                     early_stop;
  unsigned                                 max_batch;
  std::chrono::microseconds                linger;
  std::mutex                               mu;
  std::condition_variable                  cv, cv_done;
  std::thread                              worker;
  batch                                    open;
  std::deque<batch>                        ready;
  std::chrono::steady_clock::time_point    last_push;
  bool                                     stop = false, busy = false;
};

// ---------------------------------------------------------------------------------------------------------------- PDSCH modulator / DM-RS
/// The reserved-RE list only exposes per-symbol masks (re_pattern_list::get_exclusion_mask): turn them back into at most four
/// (PRB set, RE mask, symbol set) rectangles for the C ABI. Returns their number.
inline uint8_t reserved_rectangles(const srsran::re_pattern_list& reserved, unsigned nprb, miphy_re_pattern* out)
{
  if (reserved.get_nof_entries() == 0) {
    return 0;
  }
  const unsigned nsc = nprb * 12;
  struct rect {
    uint16_t                re;
    std::array<uint64_t, 5> prbs;
    uint16_t                symbols;
  };
  std::vector<rect> rects;
  for (unsigned l = 0; l != 14; ++l) {
    srsran::bounded_bitset<srsran::MAX_RB * srsran::NRE> msk(nsc);
    msk.fill(0, nsc, true);
    reserved.get_exclusion_mask(msk, l);
    std::vector<rect> here;
    for (unsigned r = 0; r != nprb; ++r) {
      uint16_t v = 0;
      for (unsigned k = 0; k != 12; ++k) {
        v |= static_cast<uint16_t>(msk.test(r * 12 + k) ? 0U : (1U << k));
      }
      if (v == 0) {
        continue;
      }
      auto it = std::find_if(here.begin(), here.end(), [v](const rect& x) { return x.re == v; });
      if (it == here.end()) {
        here.push_back(rect{v, {}, static_cast<uint16_t>(1U << l)});
        it = here.end() - 1;
      }
      it->prbs[r >> 6] |= 1ULL << (r & 63);
    }
    for (const rect& h : here) {
      auto it = std::find_if(rects.begin(), rects.end(), [&h](const rect& x) { return x.re == h.re && x.prbs == h.prbs; });
      if (it == rects.end()) {
        rects.push_back(h);
      } else {
        it->symbols |= h.symbols;
      }
    }
  }
  if (rects.size() > 4) {
    srsran::report_fatal_error("The reserved RE patterns need {} rectangles, at most 4 are supported by the HIP path.", rects.size());
  }
  uint8_t n = 0;
  for (const rect& x : rects) {
    miphy_re_pattern& o = out[n++];
    std::copy(x.prbs.begin(), x.prbs.end(), o.prb_mask);
    o.re_mask = x.re, o.symbols = x.symbols;
  }
  return n;
}

/// Writes the REs a kernel produced into a resource grid: `host` is one port of a staging grid [14][nsc] that was filled with
/// NaN before the kernel ran, so that everything that is not NaN was written.
inline void put_written_res(srsran::resource_grid_writer& grid, unsigned port, unsigned nsc, const srsran::cf_t* host)
{
  std::unique_ptr<bool[]>   mask(new bool[nsc]);
  std::vector<srsran::cf_t> vals;
  for (unsigned l = 0; l != 14; ++l) {
    vals.clear();
    for (unsigned k = 0; k != nsc; ++k) {
      const srsran::cf_t v = host[static_cast<size_t>(l) * nsc + k];
      mask[k]              = !std::isnan(v.real());
      if (mask[k]) {
        vals.push_back(v);
      }
    }
    if (!vals.empty()) {
      grid.put(port, l, 0, srsran::span<const bool>(mask.get(), nsc), vals);
    }
  }
}

/// srsran::pdsch_modulator over miphy_pdsch_modulate_batch (pdsch_modulator.h:98). One codeword on one layer, contiguous
/// allocation -- the configurations the 23.5 software modulator handles correctly.
class pdsch_modulator_hip : public srsran::pdsch_modulator
{
public:
  explicit pdsch_modulator_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void modulate(srsran::resource_grid_writer& grid, srsran::span<const srsran::bit_buffer> codewords, const config_t& config) override
  {
    srsran_assert(config.ports.size() == 1 && codewords.size() == 1, "Only one layer / one codeword is supported.");
    srsran_assert(config.freq_allocation.is_contiguous(), "Only contiguous allocations are supported.");
    const srsran::bounded_bitset<srsran::MAX_RB> prb = config.freq_allocation.get_prb_mask(config.bwp_start_rb, config.bwp_size_rb);
    const unsigned                               nprb = prb.size(), nsc = nprb * 12;
    miphy_pdsch_mod_job j = {};
    j.rnti = config.rnti, j.n_id = config.n_id, j.scaling = config.scaling;
    j.mod  = srsran::get_bits_per_symbol(config.modulation1);
    j.port = 0; // the staging grid has a single port
    j.start_symbol = config.start_symbol_index, j.nof_symbols = config.nof_symbols;
    j.dmrs_type    = (config.dmrs_config_type == srsran::dmrs_type::TYPE1) ? 1 : 2;
    j.nof_cdm_groups_without_data = config.nof_cdm_groups_without_data;
    j.grid_nof_prb = nprb, j.bwp_start_rb = config.bwp_start_rb, j.bwp_size_rb = config.bwp_size_rb;
    for (unsigned l = 0; l != 14 && l != config.dmrs_symb_pos.size(); ++l) {
      if (config.dmrs_symb_pos.test(l)) {
        j.dmrs_symbols_mask |= static_cast<uint16_t>(1U << l);
      }
    }
    prb.for_each(0, nprb, [&j](unsigned r) { j.rb_mask[r >> 6] |= 1ULL << (r & 63); });
    j.nof_reserved = reserved_rectangles(config.reserved, nprb, j.reserved);
    const srsran::bit_buffer& cw = codewords[0];
    j.nof_bits                   = cw.size();
    bits.resize(cw.size());
    for (unsigned i = 0; i != cw.size(); ++i) {
      bits[i] = cw.extract<uint8_t>(i, 1);
    }
    host.assign(static_cast<size_t>(14) * nsc, srsran::cf_t(NAN, NAN)); // NaN marks "not written by the kernel"
    auto* d_cw = static_cast<uint8_t*>(c->buf(0, bits.size() + 16));
    auto* d_g  = static_cast<float*>(c->buf(1, host.size() * sizeof(srsran::cf_t)));
    c->h2d(d_cw, bits.data(), bits.size());
    c->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    context::check(miphy_pdsch_modulate_batch(c->ctx, &j, 0, 1, d_cw, d_g, c->stream), "pdsch_modulate");
    c->d2h(host.data(), d_g, host.size() * sizeof(srsran::cf_t));
    c->sync();
    put_written_res(grid, config.ports[0], nsc, host.data());
  }

private:
  std::shared_ptr<context>  c;
  std::vector<uint8_t>      bits;
  std::vector<srsran::cf_t> host;
};

/// srsran::dmrs_pdsch_processor over miphy_dmrs_pdsch_map_batch (dmrs_pdsch_processor.h:65).
class dmrs_pdsch_processor_hip : public srsran::dmrs_pdsch_processor
{
public:
  explicit dmrs_pdsch_processor_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void map(srsran::resource_grid_writer& grid, const config_t& config) override
  {
    const unsigned nprb = config.rb_mask.size(), nsc = nprb * 12, nports = config.ports.size();
    miphy_dmrs_pdsch_job j = {};
    j.slot_in_frame = config.slot.slot_index(), j.reference_point_k_rb = config.reference_point_k_rb, j.scrambling_id = config.scrambling_id;
    j.amplitude = config.amplitude, j.dmrs_type = (config.type == srsran::dmrs_type::TYPE1) ? 1 : 2, j.n_scid = config.n_scid;
    j.nof_ports = nports, j.grid_nof_prb = nprb;
    for (unsigned p = 0; p != nports; ++p) {
      j.ports[p] = p; // staging grid indexed by DM-RS port
    }
    for (unsigned l = 0; l != 14 && l != config.symbols_mask.size(); ++l) {
      if (config.symbols_mask.test(l)) {
        j.symbols_mask |= static_cast<uint16_t>(1U << l);
      }
    }
    config.rb_mask.for_each(0, nprb, [&j](unsigned r) { j.rb_mask[r >> 6] |= 1ULL << (r & 63); });
    host.assign(static_cast<size_t>(nports) * 14 * nsc, srsran::cf_t(NAN, NAN));
    auto* d_g = static_cast<float*>(c->buf(0, host.size() * sizeof(srsran::cf_t)));
    c->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    context::check(miphy_dmrs_pdsch_map_batch(c->ctx, &j, 0, 1, d_g, c->stream), "dmrs_pdsch_map");
    c->d2h(host.data(), d_g, host.size() * sizeof(srsran::cf_t));
    c->sync();
    std::unique_ptr<bool[]>   m(new bool[nsc]);
    std::vector<srsran::cf_t> vals;
    for (unsigned p = 0; p != nports; ++p) {
      for (unsigned l = 0; l != 14; ++l) {
        vals.clear();
        for (unsigned k = 0; k != nsc; ++k) {
          const srsran::cf_t v = host[(static_cast<size_t>(p) * 14 + l) * nsc + k];
          m[k]                 = !std::isnan(v.real());
          if (m[k]) {
            vals.push_back(v);
          }
        }
        if (!vals.empty()) {
          grid.put(config.ports[p], l, 0, srsran::span<const bool>(m.get(), nsc), vals);
        }
      }
    }
  }

private:
  std::shared_ptr<context>  c;
  std::vector<srsran::cf_t> host;
};

// ---------------------------------------------------------------------------------------------------------------- PDSCH processor
/// srsran::pdsch_processor over miphy_pdsch_process_batch (pdsch_processor.h:164-166): encoding, modulation and DM-RS generation
/// in one device pass; the transport block goes up, the written resource elements come back.
class pdsch_processor_hip : public srsran::pdsch_processor
{
public:
  explicit pdsch_processor_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void process(srsran::resource_grid_writer&                                                        grid,
               srsran::static_vector<srsran::span<const uint8_t>, srsran::pdsch_processor::MAX_NOF_TRANSPORT_BLOCKS> data,
               const pdu_t&                                                                         pdu) override
  {
    srsran_assert(pdu.ports.size() == 1 && pdu.codewords.size() == 1 && data.size() == 1, "Only one layer / one codeword is supported.");
    srsran_assert(pdu.dmrs == srsran::dmrs_type::TYPE1, "Only DM-RS Type 1 is currently supported.");
    srsran_assert(pdu.freq_alloc.is_contiguous(), "Only contiguous allocation is currently supported.");
    const srsran::bounded_bitset<srsran::MAX_RB> prb  = pdu.freq_alloc.get_prb_mask(pdu.bwp_start_rb, pdu.bwp_size_rb);
    const unsigned                               nprb = prb.size(), nsc = nprb * 12;
    miphy_pdsch_pdu p = {};
    p.slot_in_frame = pdu.slot.slot_index(), p.rnti = pdu.rnti, p.n_id = pdu.n_id, p.dmrs_scrambling_id = pdu.scrambling_id;
    p.tbs_lbrm_bytes = pdu.tbs_lbrm_bytes, p.tb_bytes = data[0].size();
    p.ratio_pdsch_dmrs_to_sss_dB = pdu.ratio_pdsch_dmrs_to_sss_dB, p.ratio_pdsch_data_to_sss_dB = pdu.ratio_pdsch_data_to_sss_dB;
    p.bg = bg_id(pdu.ldpc_base_graph), p.rv = pdu.codewords[0].rv, p.mod = srsran::get_bits_per_symbol(pdu.codewords[0].modulation);
    p.port = 0; // the staging grid has a single port
    p.start_symbol = pdu.start_symbol_index, p.nof_symbols = pdu.nof_symbols, p.nof_cdm_groups_without_data = pdu.nof_cdm_groups_without_data;
    p.n_scid = pdu.n_scid ? 1 : 0, p.ref_point_prb0 = (pdu.ref_point == pdu_t::PRB0) ? 1 : 0;
    p.grid_nof_prb = nprb, p.bwp_start_rb = pdu.bwp_start_rb, p.bwp_size_rb = pdu.bwp_size_rb;
    for (unsigned l = 0; l != 14 && l != pdu.dmrs_symbol_mask.size(); ++l) {
      if (pdu.dmrs_symbol_mask.test(l)) {
        p.dmrs_symbols_mask |= static_cast<uint16_t>(1U << l);
      }
    }
    prb.for_each(0, nprb, [&p](unsigned r) { p.rb_mask[r >> 6] |= 1ULL << (r & 63); });
    p.nof_reserved = reserved_rectangles(pdu.reserved, nprb, p.reserved);
    host.assign(static_cast<size_t>(14) * nsc, srsran::cf_t(NAN, NAN)); // NaN marks "not written by the kernels"
    auto* d_tb = static_cast<uint8_t*>(c->buf(0, data[0].size() + 16));
    auto* d_g  = static_cast<float*>(c->buf(1, host.size() * sizeof(srsran::cf_t)));
    c->h2d(d_tb, data[0].data(), data[0].size());
    c->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    context::check(miphy_pdsch_process_batch(c->ctx, &p, 1, d_tb, d_g, c->stream), "pdsch_process");
    c->d2h(host.data(), d_g, host.size() * sizeof(srsran::cf_t));
    c->sync();
    put_written_res(grid, pdu.ports[0], nsc, host.data());
  }

private:
  std::shared_ptr<context>  c;
  std::vector<srsran::cf_t> host;
};

/// Replaces create_pdsch_processor_factory_sw(encoder, modulator, dmrs) (channel_processor_factories.h:153-156).
class pdsch_processor_factory_hip : public srsran::pdsch_processor_factory
{
public:
  explicit pdsch_processor_factory_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  std::unique_ptr<srsran::pdsch_processor>     create() override { return std::make_unique<pdsch_processor_hip>(c); }
  std::unique_ptr<srsran::pdsch_pdu_validator> create_validator() override; // defined at the end of this header

private:
  std::shared_ptr<context> c;
};

// ---------------------------------------------------------------------------------------------------------------- PDCCH processor
/// srsran::pdcch_processor over miphy_pdcch_process_batch (pdcch_processor.h:151): the CCE-to-PRB mapping is the reference's own host
/// function (pdcch_processor_impl::compute_rb_mask, lib/ran/pdcch/cce_to_prb_mapping.cpp), everything after it -- encoding, scrambling,
/// modulation, mapping and DM-RS -- runs on the device; the written REs come back through the resource_grid_mapper's grid.
class pdcch_processor_hip : public srsran::pdcch_processor
{
public:
  explicit pdcch_processor_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void process(srsran::resource_grid_mapper& mapper, const pdu_t& pdu) override
  {
    const coreset_description& coreset = pdu.coreset;
    const dci_description&     dci     = pdu.dci;
    srsran_assert(coreset.duration > 0 && coreset.duration <= srsran::pdcch_constants::MAX_CORESET_DURATION, "Invalid CORESET duration ({})", coreset.duration);
    srsran::prb_index_list prbs;
    switch (coreset.cce_to_reg_mapping) {
      case cce_to_reg_mapping_type::CORESET0:
        prbs = srsran::cce_to_prb_mapping_coreset0(coreset.bwp_start_rb, coreset.bwp_size_rb, coreset.duration, coreset.shift_index, dci.aggregation_level, dci.cce_index);
        break;
      case cce_to_reg_mapping_type::NON_INTERLEAVED:
        prbs = srsran::cce_to_prb_mapping_non_interleaved(coreset.bwp_start_rb, coreset.frequency_resources, coreset.duration, dci.aggregation_level, dci.cce_index);
        break;
      default:
        prbs = srsran::cce_to_prb_mapping_interleaved(coreset.bwp_start_rb, coreset.frequency_resources, coreset.duration, coreset.reg_bundle_size, coreset.interleaver_size,
                                                      coreset.shift_index, dci.aggregation_level, dci.cce_index);
        break;
    }
    const unsigned  nprb = coreset.bwp_start_rb + coreset.bwp_size_rb, nsc = nprb * 12;
    miphy_pdcch_pdu p    = {};
    p.slot_in_frame = pdu.slot.slot_index(), p.rnti = dci.rnti, p.n_id_pdcch_data = dci.n_id_pdcch_data, p.n_rnti = dci.n_rnti;
    p.n_id_pdcch_dmrs = dci.n_id_pdcch_dmrs;
    p.reference_point_k_rb = coreset.cce_to_reg_mapping == cce_to_reg_mapping_type::CORESET0 ? coreset.bwp_start_rb : 0;
    p.data_power_offset_dB = dci.data_power_offset_dB, p.dmrs_power_offset_dB = dci.dmrs_power_offset_dB;
    p.payload_size = dci.payload.size(), p.aggregation_level = dci.aggregation_level, p.start_symbol = coreset.start_symbol_index;
    p.duration = coreset.duration, p.port = 0, p.grid_nof_prb = nprb;
    for (uint16_t r : prbs) {
      p.rb_mask[r >> 6] |= 1ULL << (r & 63);
    }
    host.assign(static_cast<size_t>(14) * nsc, srsran::cf_t(NAN, NAN)); // NaN marks "not written by the kernels"
    auto* d_pl = static_cast<uint8_t*>(c->buf(0, dci.payload.size() + 16));
    auto* d_g  = static_cast<float*>(c->buf(1, host.size() * sizeof(srsran::cf_t)));
    c->h2d(d_pl, dci.payload.data(), dci.payload.size());
    c->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    context::check(miphy_pdcch_process_batch(c->ctx, &p, 1, d_pl, d_g, c->stream), "pdcch_process");
    c->d2h(host.data(), d_g, host.size() * sizeof(srsran::cf_t));
    c->sync();
    // Data REs {0,2,3,4,6,7,8,10,11} and DM-RS REs {1,5,9}: every RE of the allocated PRBs over the CORESET symbols was written, so
    // one pattern (all twelve REs) hands them to the mapper in its own order: symbol by symbol, ascending subcarrier.
    srsran::re_pattern pattern;
    pattern.prb_mask = srsran::bounded_bitset<srsran::MAX_RB>(nprb);
    for (uint16_t r : prbs) {
      pattern.prb_mask.set(r, true);
    }
    pattern.symbols.fill(coreset.start_symbol_index, coreset.start_symbol_index + coreset.duration);
    pattern.re_mask = ~srsran::re_prb_mask();
    srsran::dynamic_re_buffer res(1, prbs.size() * 12 * coreset.duration);
    srsran::span<srsran::cf_t> out = res.get_slice(0);
    unsigned                   n   = 0;
    for (unsigned l = coreset.start_symbol_index; l != coreset.start_symbol_index + coreset.duration; ++l) {
      for (unsigned r = 0; r != nprb; ++r) {
        if (pattern.prb_mask.test(r)) {
          for (unsigned k = 0; k != 12; ++k) {
            out[n++] = host[static_cast<size_t>(l) * nsc + r * 12 + k];
          }
        }
      }
    }
    srsran::re_pattern_list patterns;
    patterns.merge(pattern);
    mapper.map(res, patterns, srsran::make_single_port());
  }

private:
  std::shared_ptr<context>  c;
  std::vector<srsran::cf_t> host;
};

/// Replaces create_pdcch_processor_factory_sw(encoder, modulator, dmrs) (channel_processor_factories.h:113-116).
class pdcch_processor_factory_hip : public srsran::pdcch_processor_factory
{
public:
  explicit pdcch_processor_factory_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  std::unique_ptr<srsran::pdcch_processor>     create() override { return std::make_unique<pdcch_processor_hip>(c); }
  std::unique_ptr<srsran::pdcch_pdu_validator> create_validator() override; // defined at the end of this header

private:
  std::shared_ptr<context> c;
};

// ---------------------------------------------------------------------------------------------------------------- SS/PBCH block processor
/// srsran::ssb_processor over miphy_ssb_process_batch (ssb_processor.h:80): the block position is the reference's own look-up
/// (ssb_get_l_first / ssb_get_k_first, ssb_mapping.h), everything else -- PBCH encoding and modulation, its DM-RS, PSS and SSS -- is one
/// device pass; the 4 x 240 REs of the block come back and are put on every port of the PDU.
class ssb_processor_hip : public srsran::ssb_processor
{
public:
  explicit ssb_processor_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void process(srsran::resource_grid_writer& grid, const pdu_t& pdu) override
  {
    const unsigned l_in_burst = srsran::ssb_get_l_first(pdu.pattern_case, pdu.ssb_idx);
    const unsigned l_start    = l_in_burst % srsran::get_nsymb_per_slot(srsran::cyclic_prefix::NORMAL);
    const unsigned k_start    = srsran::ssb_get_k_first(srsran::to_frequency_range(pdu.pattern_case), srsran::to_subcarrier_spacing(pdu.pattern_case), pdu.common_scs,
                                                     pdu.offset_to_pointA, pdu.subcarrier_offset);
    srsran_assert((l_in_burst / srsran::get_nsymb_per_slot(srsran::cyclic_prefix::NORMAL)) == pdu.slot.hrf_slot_index(), "Invalid slot index ({}) for SSB index {}",
                  pdu.slot.hrf_slot_index(), l_in_burst);
    // staging grid: one port, just wide enough for the block
    const unsigned nprb = (k_start + 240 + 11) / 12, nsc = nprb * 12;
    miphy_ssb_pdu  p    = {};
    p.msg.N_id = pdu.phys_cell_id, p.msg.ssb_idx = pdu.ssb_idx, p.msg.L_max = pdu.L_max, p.msg.hrf = pdu.slot.is_odd_hrf() ? 1 : 0, p.msg.sfn = pdu.slot.sfn();
    p.msg.k_ssb = pdu.subcarrier_offset.to_uint();
    for (unsigned i = 0; i != 32 && i != pdu.bch_payload.size(); ++i) {
      p.msg.payload[i] = pdu.bch_payload[i];
    }
    p.ssb_first_subcarrier = k_start, p.ssb_first_symbol = l_start, p.beta_pss_dB = pdu.beta_pss, p.grid_nof_prb = nprb < 20 ? 20 : nprb;
    p.nof_ports = 1, p.ports[0] = 0;
    const unsigned nsc_g = p.grid_nof_prb * 12;
    (void)nsc;
    host.assign(static_cast<size_t>(14) * nsc_g, srsran::cf_t(NAN, NAN));
    auto* d_g = static_cast<float*>(c->buf(0, host.size() * sizeof(srsran::cf_t)));
    c->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    context::check(miphy_ssb_process_batch(c->ctx, &p, 1, d_g, c->stream), "ssb_process");
    c->d2h(host.data(), d_g, host.size() * sizeof(srsran::cf_t));
    c->sync();
    for (unsigned port : pdu.ports) {
      put_written_res(grid, port, nsc_g, host.data());
    }
  }

private:
  std::shared_ptr<context>  c;
  std::vector<srsran::cf_t> host;
};

/// Replaces create_ssb_processor_factory_sw(config) (channel_processor_factories.h:293-301).
class ssb_processor_factory_hip : public srsran::ssb_processor_factory
{
public:
  explicit ssb_processor_factory_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  std::unique_ptr<srsran::ssb_processor>     create() override { return std::make_unique<ssb_processor_hip>(c); }
  std::unique_ptr<srsran::ssb_pdu_validator> create_validator() override; // defined at the end of this header

private:
  std::shared_ptr<context> c;
};

// ---------------------------------------------------------------------------------------------------------------- NZP-CSI-RS generator
/// srsran::nzp_csi_rs_generator over miphy_csi_rs_map_batch (nzp_csi_rs_generator.h:91): the per-port patterns are the reference's own
/// get_csi_rs_pattern() (TS 38.211 Table 7.4.1.5.3-1 bookkeeping), sequence generation, CDM weights and RE mapping run on the device.
class nzp_csi_rs_generator_hip : public srsran::nzp_csi_rs_generator
{
public:
  explicit nzp_csi_rs_generator_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void map(srsran::resource_grid_writer& grid, const config_t& config) override
  {
    const unsigned nof_ports = config.ports.size();
    srsran_assert(config.pmi == 0, "Precoding is not currently supported");
    srsran_assert(nof_ports >= 1 && nof_ports <= 16, "Invalid number of ports.");
    srsran::csi_rs_pattern_configuration pc;
    pc.start_rb = config.start_rb, pc.nof_rb = config.nof_rb, pc.csi_rs_mapping_table_row = config.csi_rs_mapping_table_row;
    pc.freq_allocation_ref_idx = config.freq_allocation_ref_idx, pc.symbol_l0 = config.symbol_l0, pc.symbol_l1 = config.symbol_l1;
    pc.cdm = config.cdm, pc.freq_density = config.freq_density, pc.nof_ports = nof_ports;
    const srsran::csi_rs_pattern pat = srsran::get_csi_rs_pattern(pc);
    const unsigned               nprb = config.start_rb + config.nof_rb, nsc = nprb * 12;
    miphy_csi_rs_job             j    = {};
    j.slot_in_frame = config.slot.slot_index(), j.scrambling_id = config.scrambling_id, j.amplitude = config.amplitude;
    j.start_rb = config.start_rb, j.nof_rb = config.nof_rb, j.rb_begin = pat.rb_begin, j.rb_end = std::min<unsigned>(pat.rb_end, nprb), j.rb_stride = pat.rb_stride;
    j.grid_nof_prb = nprb, j.mapping_row = config.csi_rs_mapping_table_row, j.cdm = static_cast<uint8_t>(config.cdm);
    j.freq_density = static_cast<uint8_t>(config.freq_density), j.nof_ports = nof_ports;
    for (unsigned p = 0; p != nof_ports; ++p) {
      j.ports[p] = p; // staging grid indexed by CSI-RS port
      for (unsigned k = 0; k != 12; ++k) {
        j.re_mask[p] |= static_cast<uint16_t>(pat.prb_patterns[p].re_mask.test(k) ? (1U << k) : 0U);
      }
      for (unsigned l = 0; l != 14; ++l) {
        j.symbol_mask[p] |= static_cast<uint16_t>(pat.prb_patterns[p].symbol_mask.test(l) ? (1U << l) : 0U);
      }
    }
    host.assign(static_cast<size_t>(nof_ports) * 14 * nsc, srsran::cf_t(NAN, NAN)); // NaN marks "not written by the kernel"
    auto* d_g = static_cast<float*>(c->buf(0, host.size() * sizeof(srsran::cf_t)));
    c->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    context::check(miphy_csi_rs_map_batch(c->ctx, &j, 0, 1, d_g, c->stream), "csi_rs_map");
    c->d2h(host.data(), d_g, host.size() * sizeof(srsran::cf_t));
    c->sync();
    for (unsigned p = 0; p != nof_ports; ++p) {
      put_written_res(grid, config.ports[p], nsc, host.data() + static_cast<size_t>(p) * 14 * nsc);
    }
  }

private:
  std::shared_ptr<context>  c;
  std::vector<srsran::cf_t> host;
};

/// Replaces create_nzp_csi_rs_generator_factory_sw(prg) (signal_processor_factories.h:81-82).
class nzp_csi_rs_generator_factory_hip : public srsran::nzp_csi_rs_generator_factory
{
public:
  explicit nzp_csi_rs_generator_factory_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  std::unique_ptr<srsran::nzp_csi_rs_generator>               create() override { return std::make_unique<nzp_csi_rs_generator_hip>(c); }
  std::unique_ptr<srsran::nzp_csi_rs_configuration_validator> create_validator() override; // defined at the end of this header

private:
  std::shared_ptr<context> c;
};

// ---------------------------------------------------------------------------------------------------------------- downlink processor
/// srsran::downlink_processor (downlink_processor.h:45-111, lib/phy/upper/downlink_processor_single_executor_impl.cpp:54-199) that
/// batches the PDSCH PDUs of a slot: process_pdsch() queues, finish_processing_pdus() -- the reference's own end-of-slot call --
/// runs miphy_pdsch_process_batch over all of them (transport blocks up once, written REs down once), puts the REs into the grid
/// and sends it through the gateway. PDCCH, SSB and CSI-RS go to the CPU processors given at construction, at once. Same
/// observable behaviour as the reference processor with its executor: nothing happens without a configured grid, the grid is zeroed
/// on configuration and sent exactly once, after finish_processing_pdus(). Like the reference (downlink_processor_single_executor_impl
/// hands every PDU to an executor and sends the grid from there when the last one is done) the call returns at once: a completion
/// thread owned by the processor runs the batch, writes the resource elements and calls the gateway; is_reserved() stays true until
/// the grid has been sent, which is what the processor pool looks at before it hands the processor out again.
class downlink_processor_hip : public srsran::downlink_processor
{
public:
  downlink_processor_hip(std::shared_ptr<context>                      c,
                         srsran::upper_phy_rg_gateway&                 gateway,
                         std::unique_ptr<srsran::pdcch_processor>      pdcch,
                         std::unique_ptr<srsran::ssb_processor>        ssb,
                         std::unique_ptr<srsran::nzp_csi_rs_generator> csi_rs,
                         unsigned                                      grid_nof_ports,
                         unsigned                                      grid_nof_prb) :
    c(std::move(c)), gateway(gateway), pdcch(std::move(pdcch)), ssb(std::move(ssb)), csi_rs(std::move(csi_rs)), nports(grid_nof_ports), nprb(grid_nof_prb)
  {
    wc     = std::make_shared<context>(this->c->device); // the completion thread drives the caller's device through its own context and stream
    worker = std::thread([this] { completion_loop(); });
  }

  void process_pdcch(const srsran::pdcch_processor::pdu_t& pdu) override
  {
    if (grid == nullptr) {
      return;
    }
    srsran_assert(pdcch, "A PDCCH processor is required.");
    srsran::resource_grid_mapper mapper(*grid);
    pdcch->process(mapper, pdu);
  }
  void process_ssb(const srsran::ssb_processor::pdu_t& pdu) override
  {
    if (grid == nullptr) {
      return;
    }
    srsran_assert(ssb, "An SSB processor is required.");
    ssb->process(*grid, pdu);
  }
  void process_nzp_csi_rs(const srsran::nzp_csi_rs_generator::config_t& config) override
  {
    if (grid == nullptr) {
      return;
    }
    srsran_assert(csi_rs, "A CSI-RS generator is required.");
    csi_rs->map(*grid, config);
  }
  void process_pdsch(const srsran::static_vector<srsran::span<const uint8_t>, srsran::pdsch_processor::MAX_NOF_TRANSPORT_BLOCKS>& data,
                     const srsran::pdsch_processor::pdu_t&                                                                        pdu) override
  {
    if (grid == nullptr) {
      return;
    }
    // The PDU validator of the factory rejects these up front (pdsch_pdu_validator_hip); a caller that skips it fails loudly here
    // also in release builds.
    require(pdu.ports.size() == 1 && pdu.codewords.size() == 1 && data.size() == 1, "Only one layer / one codeword is supported.");
    require(pdu.dmrs == srsran::dmrs_type::TYPE1, "Only DM-RS Type 1 is currently supported.");
    require(pdu.freq_alloc.is_contiguous(), "Only contiguous allocation is currently supported.");
    require(pdu.ports[0] < nports, "Transmit port outside the resource grid.");
    queue.push_back(entry{data[0], pdu});
  }
  void configure_resource_grid(const srsran::resource_grid_context& context, srsran::resource_grid& grid_) override
  {
    require(!reserved.load(std::memory_order_acquire) && queue.empty(), "Reusing downlink processor that it is still processing PDUs.");
    rg_context = context;
    grid       = &grid_;
    grid->set_all_zero();
    reserved.store(true, std::memory_order_release);
  }
  void finish_processing_pdus() override
  {
    if (grid == nullptr) {
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mtx);
      pending = true; // the completion thread takes the queue, the grid and the context from here
    }
    cv.notify_all();
  }
  /// True from configure_resource_grid() until the completion thread has sent the grid.
  bool is_reserved() const override { return reserved.load(std::memory_order_acquire); }
  /// Blocks until the grid of the last finish_processing_pdus() has been sent (tests, shutdown).
  void wait_sent()
  {
    std::unique_lock<std::mutex> lk(mtx);
    cv.wait(lk, [this] { return !pending; });
  }
  ~downlink_processor_hip() override
  {
    {
      std::lock_guard<std::mutex> lk(mtx);
      stop = true;
    }
    cv.notify_all();
    if (worker.joinable()) {
      worker.join();
    }
  }

private:
  void completion_loop()
  {
    wc->bind_thread(); // a new thread starts on device 0
    std::unique_lock<std::mutex> lk(mtx);
    for (;;) {
      cv.wait(lk, [this] { return pending || stop; });
      if (pending) {
        lk.unlock();
        run_pdsch_batch();
        gateway.send(rg_context, *grid);
        lk.lock();
        // Everything of this slot is cleared BEFORE the reservation is released: once is_reserved() reads false the pool may hand the
        // processor out again, and the next slot's configure_resource_grid() / finish_processing_pdus() set `grid` and `pending` anew --
        // nothing written here may come after that.
        grid    = nullptr;
        pending = false;
        reserved.store(false, std::memory_order_release);
        cv.notify_all();
        continue;
      }
      if (stop) {
        return;
      }
    }
  }
  void run_pdsch_batch()
  {
    if (queue.empty()) {
      return;
    }
    const unsigned               n = queue.size(), nsc = nprb * 12;
    std::vector<miphy_pdsch_pdu> pdus(n);
    size_t                       tb_bytes = 0;
    for (unsigned i = 0; i != n; ++i) {
      const srsran::pdsch_processor::pdu_t&        pdu = queue[i].pdu;
      const srsran::bounded_bitset<srsran::MAX_RB> prb = pdu.freq_alloc.get_prb_mask(pdu.bwp_start_rb, pdu.bwp_size_rb);
      srsran_assert(prb.size() <= nprb, "The allocation exceeds the resource grid.");
      miphy_pdsch_pdu& p = pdus[i];
      p                  = {};
      p.slot_in_frame = pdu.slot.slot_index(), p.rnti = pdu.rnti, p.n_id = pdu.n_id, p.dmrs_scrambling_id = pdu.scrambling_id;
      p.tbs_lbrm_bytes = pdu.tbs_lbrm_bytes, p.tb_bytes = queue[i].data.size();
      p.ratio_pdsch_dmrs_to_sss_dB = pdu.ratio_pdsch_dmrs_to_sss_dB, p.ratio_pdsch_data_to_sss_dB = pdu.ratio_pdsch_data_to_sss_dB;
      p.bg = bg_id(pdu.ldpc_base_graph), p.rv = pdu.codewords[0].rv, p.mod = srsran::get_bits_per_symbol(pdu.codewords[0].modulation);
      p.port = pdu.ports[0];
      p.start_symbol = pdu.start_symbol_index, p.nof_symbols = pdu.nof_symbols, p.nof_cdm_groups_without_data = pdu.nof_cdm_groups_without_data;
      p.n_scid = pdu.n_scid ? 1 : 0, p.ref_point_prb0 = (pdu.ref_point == srsran::pdsch_processor::pdu_t::PRB0) ? 1 : 0;
      p.grid_nof_prb = nprb, p.bwp_start_rb = pdu.bwp_start_rb, p.bwp_size_rb = pdu.bwp_size_rb;
      for (unsigned l = 0; l != 14 && l != pdu.dmrs_symbol_mask.size(); ++l) {
        if (pdu.dmrs_symbol_mask.test(l)) {
          p.dmrs_symbols_mask |= static_cast<uint16_t>(1U << l);
        }
      }
      prb.for_each(0, prb.size(), [&p](unsigned r) { p.rb_mask[r >> 6] |= 1ULL << (r & 63); });
      p.nof_reserved = reserved_rectangles(pdu.reserved, prb.size(), p.reserved);
      p.tb_offset = tb_bytes, p.grid_offset = 0;
      tb_bytes += (queue[i].data.size() + 15) & ~static_cast<size_t>(15);
    }
    tbs.assign(tb_bytes + 16, 0);
    for (unsigned i = 0; i != n; ++i) {
      std::memcpy(&tbs[pdus[i].tb_offset], queue[i].data.data(), queue[i].data.size());
    }
    host.assign(static_cast<size_t>(nports) * 14 * nsc, srsran::cf_t(NAN, NAN)); // NaN marks "not written by the kernels"
    auto* d_tb = static_cast<uint8_t*>(wc->buf(0, tbs.size()));
    auto* d_g  = static_cast<float*>(wc->buf(1, host.size() * sizeof(srsran::cf_t)));
    wc->h2d(d_tb, tbs.data(), tbs.size());
    wc->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    context::check(miphy_pdsch_process_batch(wc->ctx, pdus.data(), n, d_tb, d_g, wc->stream), "pdsch_process");
    wc->d2h(host.data(), d_g, host.size() * sizeof(srsran::cf_t));
    wc->sync();
    for (unsigned p = 0; p != nports; ++p) {
      put_written_res(*grid, p, nsc, host.data() + static_cast<size_t>(p) * 14 * nsc);
    }
    queue.clear();
  }

  struct entry {
    srsran::span<const uint8_t>    data;
    srsran::pdsch_processor::pdu_t pdu;
  };
  std::shared_ptr<context>                      c;
  srsran::upper_phy_rg_gateway&                 gateway;
  std::unique_ptr<srsran::pdcch_processor>      pdcch;
  std::unique_ptr<srsran::ssb_processor>        ssb;
  std::unique_ptr<srsran::nzp_csi_rs_generator> csi_rs;
  unsigned                                      nports, nprb;
  srsran::resource_grid_context                 rg_context = {};
  srsran::resource_grid*                        grid       = nullptr;
  std::vector<entry>                            queue;
  std::vector<uint8_t>                          tbs;
  std::vector<srsran::cf_t>                     host;
  std::atomic<bool>                             reserved{false};
  std::mutex                                    mtx;
  std::condition_variable                       cv;
  bool                                          pending = false, stop = false;
  std::shared_ptr<context>                      wc;
  std::thread                                   worker; // started at the end of the constructor
};

/// A downlink processor whose four PDU types all run on the device: PDSCH batched per slot, PDCCH / SSB / CSI-RS through the
/// device processors above (replaces the pool entry downlink_processor_single_executor_factory::create builds, upper_phy_factories.cpp:269-300).
inline std::unique_ptr<srsran::downlink_processor>
create_downlink_processor_hip(std::shared_ptr<context> c, srsran::upper_phy_rg_gateway& gateway, unsigned grid_nof_ports, unsigned grid_nof_prb)
{
  return std::make_unique<downlink_processor_hip>(c, gateway, std::make_unique<pdcch_processor_hip>(c), std::make_unique<ssb_processor_hip>(c),
                                                  std::make_unique<nzp_csi_rs_generator_hip>(c), grid_nof_ports, grid_nof_prb);
}

// ---------------------------------------------------------------------------------------------------------------- Open Fronthaul IQ compression
/// srsran::ofh::iq_decompressor / iq_compressor for compression_type::BFP and compression_type::none over miphy_ofh_iq_*_batch
/// (iq_decompressor.h:49-50, iq_compressor.h:49-50): what iq_{de}compressor_selector dispatches to for the two methods the reference
/// implements. The compressed_prb objects are laid out as the U-plane section payload ([udCompParam][packed IQ] per PRB for BFP,
/// the packed IQ alone for the uncompressed format) on the way to and from the device.
class iq_compression_hip : public srsran::ofh::iq_compressor, public srsran::ofh::iq_decompressor
{
public:
  explicit iq_compression_hip(std::shared_ptr<context> c, float iq_scaling = 1.0F) : c(std::move(c)), iq_scaling(iq_scaling) {}
  void compress(srsran::span<srsran::ofh::compressed_prb> output, srsran::span<const srsran::cf_t> input, const srsran::ofh::ru_compression_params& params) override
  {
    srsran_assert(input.size() == output.size() * 12, "Wrong number of input samples.");
    const uint16_t comp = method(params);
    const unsigned w = params.data_width, hdr = comp == MIPHY_OFH_COMPRESSION_BFP ? 1 : 0, rec = hdr + 3 * w, nprb = output.size();
    if (nprb == 0) {
      return;
    }
    auto* d_x = static_cast<float*>(c->buf(0, input.size() * sizeof(srsran::cf_t)));
    auto* d_p = static_cast<uint8_t*>(c->buf(1, static_cast<size_t>(nprb) * rec));
    bytes.resize(static_cast<size_t>(nprb) * rec);
    c->h2d(d_x, input.data(), input.size() * sizeof(srsran::cf_t));
    srsran_assert(nprb <= 275, "At most 275 PRBs per call.");
    miphy_ofh_iq_job job{0, 0, nprb, static_cast<uint16_t>(w), comp};
    context::check(miphy_ofh_iq_compress_batch(c->ctx, &job, 0, 1, d_x, iq_scaling, d_p, c->stream), "ofh_iq_compress");
    c->d2h(bytes.data(), d_p, bytes.size());
    c->sync();
    for (unsigned p = 0; p != nprb; ++p) {
      if (hdr != 0) {
        output[p].set_compression_param(bytes[static_cast<size_t>(p) * rec]);
      }
      std::memcpy(output[p].get_buffer().data(), &bytes[static_cast<size_t>(p) * rec + hdr], 3 * w);
      output[p].set_stored_size(3 * w);
    }
  }
  void decompress(srsran::span<srsran::cf_t> output, srsran::span<const srsran::ofh::compressed_prb> input, const srsran::ofh::ru_compression_params& params) override
  {
    srsran_assert(output.size() == input.size() * 12, "Wrong number of output samples.");
    const uint16_t comp = method(params);
    const unsigned w = params.data_width, hdr = comp == MIPHY_OFH_COMPRESSION_BFP ? 1 : 0, rec = hdr + 3 * w, nprb = input.size();
    if (nprb == 0) {
      return;
    }
    srsran_assert(nprb <= 275, "At most 275 PRBs per call.");
    bytes.resize(static_cast<size_t>(nprb) * rec);
    for (unsigned p = 0; p != nprb; ++p) {
      if (hdr != 0) {
        bytes[static_cast<size_t>(p) * rec] = input[p].get_compression_param();
      }
      // get_packed_data() depends on set_stored_size(), which the U-plane decoder does not call: take the bytes themselves
      std::memcpy(&bytes[static_cast<size_t>(p) * rec + hdr], const_cast<srsran::ofh::compressed_prb&>(input[p]).get_buffer().data(), 3 * w);
    }
    auto* d_p = static_cast<uint8_t*>(c->buf(0, bytes.size()));
    auto* d_x = static_cast<float*>(c->buf(1, output.size() * sizeof(srsran::cf_t)));
    c->h2d(d_p, bytes.data(), bytes.size());
    miphy_ofh_iq_job job{0, 0, nprb, static_cast<uint16_t>(w), comp};
    context::check(miphy_ofh_iq_decompress_batch(c->ctx, &job, 0, 1, d_p, d_x, 1, c->stream), "ofh_iq_decompress");
    c->d2h(output.data(), d_x, output.size() * sizeof(srsran::cf_t));
    c->sync();
  }

private:
  static uint16_t method(const srsran::ofh::ru_compression_params& params)
  {
    if (params.type == srsran::ofh::compression_type::BFP) {
      return MIPHY_OFH_COMPRESSION_BFP;
    }
    if (params.type == srsran::ofh::compression_type::none) {
      return MIPHY_OFH_COMPRESSION_NONE;
    }
    srsran::report_fatal_error("Compression of {} type is not implemented", srsran::ofh::to_string(params.type));
  }

  std::shared_ptr<context> c;
  float                    iq_scaling;
  std::vector<uint8_t>     bytes;
};

/// Replace create_iq_compressor(type, iq_scaling, impl) and create_iq_decompressor(type, impl) for type = none | BFP, or the
/// selectors built from them (compression_factory.h:36-50).
inline std::unique_ptr<srsran::ofh::iq_compressor> create_iq_compressor_hip(std::shared_ptr<context> c, float iq_scaling = 1.0F)
{
  return std::make_unique<iq_compression_hip>(std::move(c), iq_scaling);
}
inline std::unique_ptr<srsran::ofh::iq_decompressor> create_iq_decompressor_hip(std::shared_ptr<context> c)
{
  return std::make_unique<iq_compression_hip>(std::move(c));
}

// ---------------------------------------------------------------------------------------------------------------- PDCCH
/// srsran::pdcch_encoder over miphy_pdcch_encode_batch (pdcch_encoder.h:53).
class pdcch_encoder_hip : public srsran::pdcch_encoder
{
public:
  explicit pdcch_encoder_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void encode(srsran::span<uint8_t> encoded, srsran::span<const uint8_t> data, const config_t& config) override
  {
    srsran_assert(encoded.size() >= config.E, "Output data vector is too small to store encoded bits");
    auto*    d_in  = static_cast<uint8_t*>(c->buf(0, data.size() + 16));
    auto*    d_out = static_cast<uint8_t*>(c->buf(1, config.E));
    uint16_t rnti  = static_cast<uint16_t>(config.rnti);
    auto*    d_rn  = reinterpret_cast<uint16_t*>(d_in + ((data.size() + 7) / 8) * 8);
    c->h2d(d_in, data.data(), data.size());
    c->h2d(d_rn, &rnti, sizeof(rnti));
    context::check(miphy_pdcch_encode_batch(c->ctx, data.size(), config.E, 1, d_in, d_rn, d_out, c->stream), "pdcch_encode");
    c->d2h(encoded.data(), d_out, config.E);
    c->sync();
  }

private:
  std::shared_ptr<context> c;
};

// ---------------------------------------------------------------------------------------------------------------- factories
#define MIPHY_SIMPLE_FACTORY(NAME, BASE, PRODUCT, IMPL)                                        \
  class NAME : public srsran::BASE                                                             \
  {                                                                                            \
  public:                                                                                      \
    explicit NAME(std::shared_ptr<context> c) : c(std::move(c)) {}                             \
    std::unique_ptr<srsran::PRODUCT> create() override { return std::make_unique<IMPL>(c); }   \
                                                                                               \
  private:                                                                                     \
    std::shared_ptr<context> c;                                                                \
  };

MIPHY_SIMPLE_FACTORY(ldpc_decoder_factory_hip, ldpc_decoder_factory, ldpc_decoder, ldpc_decoder_hip)
MIPHY_SIMPLE_FACTORY(ldpc_encoder_factory_hip, ldpc_encoder_factory, ldpc_encoder, ldpc_encoder_hip)
MIPHY_SIMPLE_FACTORY(ldpc_rate_matcher_factory_hip, ldpc_rate_matcher_factory, ldpc_rate_matcher, ldpc_rate_matcher_hip)
MIPHY_SIMPLE_FACTORY(ldpc_rate_dematcher_factory_hip, ldpc_rate_dematcher_factory, ldpc_rate_dematcher, ldpc_rate_dematcher_hip)
MIPHY_SIMPLE_FACTORY(pdsch_encoder_factory_hip, pdsch_encoder_factory, pdsch_encoder, pdsch_encoder_hip)
MIPHY_SIMPLE_FACTORY(pusch_decoder_factory_hip, pusch_decoder_factory, pusch_decoder, pusch_decoder_hip)
MIPHY_SIMPLE_FACTORY(dmrs_pusch_estimator_factory_hip, dmrs_pusch_estimator_factory, dmrs_pusch_estimator, dmrs_pusch_estimator_hip)
MIPHY_SIMPLE_FACTORY(pdcch_encoder_factory_hip, pdcch_encoder_factory, pdcch_encoder, pdcch_encoder_hip)
MIPHY_SIMPLE_FACTORY(pusch_demodulator_factory_hip, pusch_demodulator_factory, pusch_demodulator, pusch_demodulator_hip)
MIPHY_SIMPLE_FACTORY(pdsch_modulator_factory_hip, pdsch_modulator_factory, pdsch_modulator, pdsch_modulator_hip)
MIPHY_SIMPLE_FACTORY(dmrs_pdsch_processor_factory_hip, dmrs_pdsch_processor_factory, dmrs_pdsch_processor, dmrs_pdsch_processor_hip)
#undef MIPHY_SIMPLE_FACTORY

/// The string-selected factory functions of the reference (channel_coding_factories.cpp:86-180) gain a "hip" case that
/// returns these.
inline std::shared_ptr<srsran::ldpc_decoder_factory> create_ldpc_decoder_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<ldpc_decoder_factory_hip>(std::move(c));
}
inline std::shared_ptr<srsran::ldpc_encoder_factory> create_ldpc_encoder_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<ldpc_encoder_factory_hip>(std::move(c));
}
inline std::shared_ptr<srsran::ldpc_rate_matcher_factory> create_ldpc_rate_matcher_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<ldpc_rate_matcher_factory_hip>(std::move(c));
}
inline std::shared_ptr<srsran::ldpc_rate_dematcher_factory> create_ldpc_rate_dematcher_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<ldpc_rate_dematcher_factory_hip>(std::move(c));
}
inline std::shared_ptr<srsran::pdsch_encoder_factory> create_pdsch_encoder_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<pdsch_encoder_factory_hip>(std::move(c));
}
inline std::shared_ptr<srsran::pusch_decoder_factory> create_pusch_decoder_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<pusch_decoder_factory_hip>(std::move(c));
}
inline std::shared_ptr<srsran::dmrs_pusch_estimator_factory> create_dmrs_pusch_estimator_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<dmrs_pusch_estimator_factory_hip>(std::move(c));
}
inline std::shared_ptr<srsran::pdcch_encoder_factory> create_pdcch_encoder_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<pdcch_encoder_factory_hip>(std::move(c));
}
/// Replace create_pdsch_modulator_factory_sw(modulation, prg) (channel_processor_factories.h:140-142) and
/// create_dmrs_pdsch_processor_factory_sw(prg) (signal_processor_factories.h:46-47).
inline std::shared_ptr<srsran::pdsch_modulator_factory> create_pdsch_modulator_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<pdsch_modulator_factory_hip>(std::move(c));
}
inline std::shared_ptr<srsran::dmrs_pdsch_processor_factory> create_dmrs_pdsch_processor_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<dmrs_pdsch_processor_factory_hip>(std::move(c));
}
/// Replaces create_pusch_demodulator_factory_sw(equalizer, demodulation, prg) (channel_processor_factories.h:256-259).
inline std::shared_ptr<srsran::pusch_demodulator_factory> create_pusch_demodulator_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<pusch_demodulator_factory_hip>(std::move(c));
}

/// OFDM factories take a configuration per product (modulation_factories.h:34-76).
class ofdm_demodulator_factory_hip : public srsran::ofdm_demodulator_factory
{
public:
  explicit ofdm_demodulator_factory_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  std::unique_ptr<srsran::ofdm_symbol_demodulator> create_ofdm_symbol_demodulator(const srsran::ofdm_demodulator_configuration& cfg) override
  {
    return std::make_unique<ofdm_symbol_demodulator_hip>(c, cfg);
  }
  std::unique_ptr<srsran::ofdm_slot_demodulator> create_ofdm_slot_demodulator(const srsran::ofdm_demodulator_configuration& cfg) override
  {
    return std::make_unique<ofdm_slot_demodulator_hip>(c, cfg);
  }

private:
  std::shared_ptr<context> c;
};

class ofdm_modulator_factory_hip : public srsran::ofdm_modulator_factory
{
public:
  explicit ofdm_modulator_factory_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  std::unique_ptr<srsran::ofdm_symbol_modulator> create_ofdm_symbol_modulator(const srsran::ofdm_modulator_configuration& cfg) override
  {
    return std::make_unique<ofdm_symbol_modulator_hip>(c, cfg);
  }
  std::unique_ptr<srsran::ofdm_slot_modulator>   create_ofdm_slot_modulator(const srsran::ofdm_modulator_configuration& cfg) override
  {
    return std::make_unique<ofdm_slot_modulator_hip>(c, cfg);
  }

private:
  std::shared_ptr<context> c;
};

// ---------------------------------------------------------------------------------------------------------------- polar blocks
/// The block-level polar interfaces over miphy_polar_block_batch (include/srsran/phy/upper/channel_coding/polar/*.h), a batch of one
/// per call, so that the reference's own chains (pdcch_encoder_impl, pbch_encoder_impl, uci decoders, polar_chain_test.cpp:156-210)
/// run on the device block by block. polar_code stays the reference's host class (bookkeeping of sets and tables).
namespace detail {
inline miphy_polar_code to_miphy_code(const srsran::polar_code& code)
{
  miphy_polar_code mc = {code.get_K(), code.get_E(), 10, code.get_ibil() == srsran::polar_code_ibil::present ? 1U : 0U};
  // nMax is not kept by polar_code: it is 10 unless that would give a larger mother code than the one the code object holds
  uint32_t n = 0, N = 0, npc = 0;
  if (miphy_polar_code_info(&mc, &n, &N, &npc) != MIPHY_OK || n != code.get_n()) {
    mc.nMax = 9;
    context::check(miphy_polar_code_info(&mc, &n, &N, &npc), "polar_code_info");
    require(n == code.get_n(), "The polar code object is not a TS 38.212 code (n = {}).", code.get_n());
  }
  return mc;
}
inline void polar_block(context& c, const miphy_polar_code* code, uint32_t op, uint32_t param, void* out, size_t out_bytes, const void* in, size_t in_bytes)
{
  void* d_in  = c.buf(0, in_bytes);
  void* d_out = c.buf(1, out_bytes);
  c.h2d(d_in, in, in_bytes);
  context::check(miphy_polar_block_batch(c.ctx, code, op, param, 1, d_in, d_out, c.stream), "polar_block");
  c.d2h(out, d_out, out_bytes);
  c.sync();
}
} // namespace detail

class polar_allocator_hip : public srsran::polar_allocator
{
public:
  explicit polar_allocator_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void allocate(srsran::span<uint8_t> input_encoder, srsran::span<const uint8_t> message, const srsran::polar_code& code) override
  {
    require(input_encoder.size() == code.get_N() && message.size() == code.get_K(), "polar_allocator: wrong sizes");
    const miphy_polar_code mc = detail::to_miphy_code(code);
    detail::polar_block(*c, &mc, MIPHY_POLAR_OP_ALLOCATE, 0, input_encoder.data(), input_encoder.size(), message.data(), message.size());
  }

private:
  std::shared_ptr<context> c;
};

class polar_deallocator_hip : public srsran::polar_deallocator
{
public:
  explicit polar_deallocator_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void deallocate(srsran::span<uint8_t> message, srsran::span<const uint8_t> output_decoder, const srsran::polar_code& code) override
  {
    require(output_decoder.size() == code.get_N() && message.size() == code.get_K(), "polar_deallocator: wrong sizes");
    const miphy_polar_code mc = detail::to_miphy_code(code);
    detail::polar_block(*c, &mc, MIPHY_POLAR_OP_DEALLOCATE, 0, message.data(), message.size(), output_decoder.data(), output_decoder.size());
  }

private:
  std::shared_ptr<context> c;
};

class polar_encoder_hip : public srsran::polar_encoder
{
public:
  explicit polar_encoder_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void encode(srsran::span<uint8_t> output, srsran::span<const uint8_t> input, unsigned code_size_log) override
  {
    require(output.size() == (1U << code_size_log) && input.size() == output.size(), "polar_encoder: wrong sizes");
    detail::polar_block(*c, nullptr, MIPHY_POLAR_OP_ENCODE, code_size_log, output.data(), output.size(), input.data(), input.size());
  }

private:
  std::shared_ptr<context> c;
};

class polar_decoder_hip : public srsran::polar_decoder
{
public:
  polar_decoder_hip(std::shared_ptr<context> c, unsigned nMax) : c(std::move(c)), nMax(nMax) {}
  void decode(srsran::span<uint8_t> data_decoded, srsran::span<const srsran::log_likelihood_ratio> input_llr, const srsran::polar_code& code) override
  {
    require(code.get_n() <= nMax, "polar_decoder: code size 2^{} exceeds the decoder's 2^{}", code.get_n(), nMax);
    require(data_decoded.size() == code.get_N() && input_llr.size() == code.get_N(), "polar_decoder: wrong sizes");
    const miphy_polar_code mc = detail::to_miphy_code(code);
    detail::polar_block(*c, &mc, MIPHY_POLAR_OP_DECODE, 0, data_decoded.data(), data_decoded.size(), input_llr.data(), input_llr.size());
  }

private:
  std::shared_ptr<context> c;
  unsigned                 nMax;
};

class polar_rate_matcher_hip : public srsran::polar_rate_matcher
{
public:
  explicit polar_rate_matcher_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void rate_match(srsran::span<uint8_t> output, srsran::span<const uint8_t> input, const srsran::polar_code& code) override
  {
    require(output.size() == code.get_E() && input.size() == code.get_N(), "polar_rate_matcher: wrong sizes");
    const miphy_polar_code mc = detail::to_miphy_code(code);
    detail::polar_block(*c, &mc, MIPHY_POLAR_OP_RATE_MATCH, 0, output.data(), output.size(), input.data(), input.size());
  }

private:
  std::shared_ptr<context> c;
};

class polar_rate_dematcher_hip : public srsran::polar_rate_dematcher
{
public:
  explicit polar_rate_dematcher_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void rate_dematch(srsran::span<srsran::log_likelihood_ratio>       output,
                    srsran::span<const srsran::log_likelihood_ratio> input,
                    const srsran::polar_code&                        code) override
  {
    require(input.size() == code.get_E() && output.size() == code.get_N(), "polar_rate_dematcher: wrong sizes");
    const miphy_polar_code mc = detail::to_miphy_code(code);
    detail::polar_block(*c, &mc, MIPHY_POLAR_OP_RATE_DEMATCH, 0, output.data(), output.size(), input.data(), input.size());
  }

private:
  std::shared_ptr<context> c;
};

class polar_interleaver_hip : public srsran::polar_interleaver
{
public:
  explicit polar_interleaver_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void interleave(srsran::span<uint8_t> out, srsran::span<const uint8_t> in, srsran::polar_interleaver_direction direction) override
  {
    require(in.size() == out.size() && in.size() >= 1 && in.size() <= 164, "polar_interleaver: wrong sizes");
    detail::polar_block(*c, nullptr, direction == srsran::polar_interleaver_direction::tx ? MIPHY_POLAR_OP_INTERLEAVE_TX : MIPHY_POLAR_OP_INTERLEAVE_RX,
                        static_cast<uint32_t>(in.size()), out.data(), out.size(), in.data(), in.size());
  }

private:
  std::shared_ptr<context> c;
};

/// Replaces create_polar_factory_sw() (channel_coding_factories.h:107-121).
class polar_factory_hip : public srsran::polar_factory
{
public:
  explicit polar_factory_hip(std::shared_ptr<context> c) : c(std::move(c)), host(srsran::create_polar_factory_sw()) {}
  std::unique_ptr<srsran::polar_allocator>      create_allocator() override { return std::make_unique<polar_allocator_hip>(c); }
  std::unique_ptr<srsran::polar_code>           create_code() override { return host->create_code(); }
  std::unique_ptr<srsran::polar_deallocator>    create_deallocator() override { return std::make_unique<polar_deallocator_hip>(c); }
  std::unique_ptr<srsran::polar_decoder>        create_decoder(unsigned code_size_log) override { return std::make_unique<polar_decoder_hip>(c, code_size_log); }
  std::unique_ptr<srsran::polar_encoder>        create_encoder() override { return std::make_unique<polar_encoder_hip>(c); }
  std::unique_ptr<srsran::polar_interleaver>    create_interleaver() override { return std::make_unique<polar_interleaver_hip>(c); }
  std::unique_ptr<srsran::polar_rate_dematcher> create_rate_dematcher() override { return std::make_unique<polar_rate_dematcher_hip>(c); }
  std::unique_ptr<srsran::polar_rate_matcher>   create_rate_matcher() override { return std::make_unique<polar_rate_matcher_hip>(c); }

private:
  std::shared_ptr<context>               c;
  std::shared_ptr<srsran::polar_factory> host;
};
inline std::shared_ptr<srsran::polar_factory> create_polar_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<polar_factory_hip>(std::move(c));
}

// ---------------------------------------------------------------------------------------------------------------- CRC calculator
/// srsran::crc_calculator over miphy_crc_batch (crc_calculator.h:45-67); with its factory it fills the hardware seam
/// downlink_processor_factory_hw_config::crc_calc_factory (upper_phy_factories.h:151-161).
class crc_calculator_hip : public srsran::crc_calculator
{
public:
  crc_calculator_hip(std::shared_ptr<context> c, srsran::crc_generator_poly poly) : c(std::move(c)), poly(poly) {}
  srsran::crc_calculator_checksum_t calculate_byte(srsran::span<const uint8_t> data) override { return run(data.data(), data.size(), data.size() * 8); }
  srsran::crc_calculator_checksum_t calculate_bit(srsran::span<const uint8_t> data) override
  {
    packed.assign((data.size() + 7) / 8, 0);
    for (size_t i = 0; i != data.size(); ++i) {
      packed[i >> 3] |= static_cast<uint8_t>((data[i] & 1U) << (7 - (i & 7)));
    }
    return run(packed.data(), packed.size(), data.size());
  }
  srsran::crc_calculator_checksum_t calculate(const srsran::bit_buffer& data) override
  {
    const srsran::span<const uint8_t> bytes = data.get_buffer();
    return run(bytes.data(), bytes.size(), data.size());
  }
  srsran::crc_generator_poly get_generator_poly() const override { return poly; }

private:
  srsran::crc_calculator_checksum_t run(const uint8_t* bytes, size_t nbytes, size_t nbits)
  {
    if (nbits == 0) {
      return 0;
    }
    auto* d_in  = static_cast<uint8_t*>(c->buf(0, nbytes + 8)); // the device CRC reads whole words: keep a zero pad behind the message
    auto* d_out = static_cast<uint32_t*>(c->buf(1, 16));
    context::hip(hipMemsetAsync(d_in + nbytes, 0, 8, c->stream), "memset");
    c->h2d(d_in, bytes, nbytes);
    miphy_crc_desc d = {0, static_cast<uint32_t>(nbits), to_miphy_crc(poly)};
    context::check(miphy_crc_batch(c->ctx, &d, 0, 1, d_in, d_out, c->stream), "crc");
    uint32_t v = 0;
    c->d2h(&v, d_out, sizeof(v));
    c->sync();
    return v;
  }
  std::shared_ptr<context>   c;
  srsran::crc_generator_poly poly;
  std::vector<uint8_t>       packed;
};

class crc_calculator_factory_hip : public srsran::crc_calculator_factory
{
public:
  explicit crc_calculator_factory_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  std::unique_ptr<srsran::crc_calculator> create(srsran::crc_generator_poly poly) override
  {
    if (poly == srsran::crc_generator_poly::CRC6) {
      return nullptr; // UCI-only polynomial, not on this path (factories return nullptr for what they do not support)
    }
    return std::make_unique<crc_calculator_hip>(c, poly);
  }

private:
  std::shared_ptr<context> c;
};
inline std::shared_ptr<srsran::crc_calculator_factory> create_crc_calculator_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<crc_calculator_factory_hip>(std::move(c));
}

// ---------------------------------------------------------------------------------------------------------------- port channel estimator
/// srsran::port_channel_estimator over miphy_port_channel_estimate_batch (port_channel_estimator.h:102-106): the pilots come from the
/// caller (PUSCH and PUCCH estimators build them), one receive port per call, all layers, intra-slot frequency hopping included
/// (port_channel_estimator_average_impl.cpp:97-224). DM-RS on every second subcarrier (type 1 / PUCCH formats 1-4 comb); the layers
/// share symbols, PRBs and hop, as in every caller of the reference.
class port_channel_estimator_hip : public srsran::port_channel_estimator
{
public:
  explicit port_channel_estimator_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void compute(srsran::channel_estimate&           estimate,
               const srsran::resource_grid_reader& grid,
               unsigned                            port,
               const srsran::dmrs_symbol_list&     pilots,
               const configuration&                cfg) override
  {
    const unsigned nl = cfg.dmrs_pattern.size();
    require(nl >= 1 && nl <= 4, "port_channel_estimator_hip: 1 to 4 layers.");
    const layer_dmrs_pattern& pt   = cfg.dmrs_pattern[0];
    const unsigned            nprb = pt.rb_mask.size(), nsc = nprb * 12;
    miphy_pusch_chest_job     j    = {};
    j.numerology = srsran::to_numerology_value(cfg.scs), j.scaling = cfg.scaling, j.nof_tx_layers = nl, j.nof_rx_ports = 1;
    j.first_symbol = cfg.first_symbol, j.nof_symbols = cfg.nof_symbols, j.grid_nof_prb = nprb;
    for (unsigned ly = 0; ly != nl; ++ly) {
      const layer_dmrs_pattern& q = cfg.dmrs_pattern[ly];
      require(q.symbols == pt.symbols && q.rb_mask == pt.rb_mask && q.rb_mask2 == pt.rb_mask2 && q.hopping_symbol_index == pt.hopping_symbol_index,
              "port_channel_estimator_hip: the layers must share symbols, PRBs and hop.");
      uint16_t re = 0;
      for (unsigned k = 0; k != 12; ++k) {
        re |= static_cast<uint16_t>(q.re_pattern.test(k) ? (1U << k) : 0U);
      }
      require(re == 0x555 || re == 0xaaa, "port_channel_estimator_hip: DM-RS on every second subcarrier only.");
      j.re_odd_mask |= static_cast<uint8_t>((re == 0xaaa) ? (1U << ly) : 0U);
    }
    for (unsigned l = 0; l != 14 && l != pt.symbols.size(); ++l) {
      j.symbols_mask |= static_cast<uint16_t>(pt.symbols.test(l) ? (1U << l) : 0U);
    }
    pt.rb_mask.for_each(0, nprb, [&j](unsigned r) { j.rb_mask[r >> 6] |= 1ULL << (r & 63); });
    if (pt.hopping_symbol_index.has_value()) {
      j.hop_symbol = pt.hopping_symbol_index.value();
      pt.rb_mask2.for_each(0, pt.rb_mask2.size(), [&j](unsigned r) { j.rb_mask2[r >> 6] |= 1ULL << (r & 63); });
    }
    const unsigned nsymb = cfg.first_symbol + cfg.nof_symbols;
    host.resize(static_cast<size_t>(14) * nsc);
    for (unsigned l = 0; l != 14; ++l) {
      grid.get(srsran::span<srsran::cf_t>(host.data() + static_cast<size_t>(l) * nsc, nsc), port, l, 0);
    }
    const srsran::re_measurement_dimensions pd = pilots.size();
    pil.resize(static_cast<size_t>(nl) * pd.nof_symbols * pd.nof_subc);
    for (unsigned ly = 0; ly != nl; ++ly) {
      for (unsigned d = 0; d != pd.nof_symbols; ++d) {
        srsran::span<const srsran::cf_t> v = pilots.get_symbol(d, ly);
        std::copy(v.begin(), v.end(), pil.begin() + (static_cast<size_t>(ly) * pd.nof_symbols + d) * pd.nof_subc);
      }
    }
    auto* d_g  = static_cast<float*>(c->buf(0, host.size() * sizeof(srsran::cf_t)));
    auto* d_p  = static_cast<float*>(c->buf(1, pil.size() * sizeof(srsran::cf_t)));
    auto* d_ce = static_cast<float*>(c->buf(2, static_cast<size_t>(nl) * nsymb * nsc * sizeof(srsran::cf_t)));
    auto* d_sc = static_cast<float*>(c->buf(3, 5 * 4 * sizeof(float) + 64));
    c->h2d(d_g, host.data(), host.size() * sizeof(srsran::cf_t));
    c->h2d(d_p, pil.data(), pil.size() * sizeof(srsran::cf_t));
    context::check(miphy_port_channel_estimate_batch(c->ctx, &j, 0, 1, d_g, d_p, d_ce, d_sc, c->stream), "port_channel_estimate");
    ce.resize(static_cast<size_t>(nl) * nsymb * nsc);
    float sc[20];
    c->d2h(ce.data(), d_ce, ce.size() * sizeof(srsran::cf_t));
    c->d2h(sc, d_sc, sizeof(float) * 5 * nl);
    c->sync();
    for (unsigned ly = 0; ly != nl; ++ly) {
      for (unsigned l = cfg.first_symbol; l != nsymb; ++l) {
        // only the PRBs of the hop the symbol belongs to carry an estimate (port_channel_estimator_average_impl.cpp:216-224)
        const srsran::bounded_bitset<srsran::MAX_RB>& m = (pt.hopping_symbol_index.has_value() && l >= pt.hopping_symbol_index.value()) ? pt.rb_mask2 : pt.rb_mask;
        srsran::span<srsran::cf_t>                    o = estimate.get_symbol_ch_estimate(l, port, ly);
        const srsran::cf_t*                           v = ce.data() + (static_cast<size_t>(ly) * nsymb + l) * nsc;
        m.for_each(0, m.size(), [&](unsigned r) { std::copy(v + r * 12, v + r * 12 + 12, o.begin() + r * 12); });
      }
      estimate.set_rsrp(sc[5 * ly + 0], port, ly);
      estimate.set_epre(sc[5 * ly + 1], port, ly);
      estimate.set_noise_variance(sc[5 * ly + 2], port, ly);
      estimate.set_snr(sc[5 * ly + 3], port, ly);
      estimate.set_time_alignment(srsran::phy_time_unit::from_seconds(sc[5 * ly + 4]), port, ly);
    }
  }

private:
  std::shared_ptr<context>  c;
  std::vector<srsran::cf_t> host, pil, ce;
};

class port_channel_estimator_factory_hip : public srsran::port_channel_estimator_factory
{
public:
  explicit port_channel_estimator_factory_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  std::unique_ptr<srsran::port_channel_estimator> create() override { return std::make_unique<port_channel_estimator_hip>(c); }

private:
  std::shared_ptr<context> c;
};
inline std::shared_ptr<srsran::port_channel_estimator_factory> create_port_channel_estimator_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<port_channel_estimator_factory_hip>(std::move(c));
}

// ---------------------------------------------------------------------------------------------------------------- channel equalizer
/// srsran::channel_equalizer over miphy_channel_equalize_batch (channel_equalizer.h:93-100, channel_equalizer_zf_impl.cpp:123-162):
/// one transmit layer on up to four ports or two layers on two ports. Inside pusch_processor_hip the one-layer case runs fused with
/// the soft demapper; this class is the block for whoever builds a demodulator of their own around the reference interfaces.
class channel_equalizer_hip : public srsran::channel_equalizer
{
public:
  explicit channel_equalizer_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  void equalize(re_list&                 eq_symbols,
                noise_var_list&          eq_noise_vars,
                const re_list&           ch_symbols,
                const ch_est_list&       ch_estimates,
                srsran::span<const float> noise_var_estimates,
                float                    tx_scaling) override
  {
    const unsigned nre = ch_symbols.get_dimension_size(re_list::dims::re);
    const unsigned npt = ch_estimates.get_dimension_size(ch_est_list::dims::rx_port);
    const unsigned nl  = ch_estimates.get_dimension_size(ch_est_list::dims::tx_layer);
    // the checks of channel_equalizer_zf_impl.cpp:27-90
    require(ch_symbols.get_dimension_size(re_list::dims::slice) == npt, "Number of Rx ports does not match: ch_symbols vs ch_estimates.");
    require(ch_estimates.get_dimension_size(ch_est_list::dims::re) == nre, "Number of channel estimates does not match the number of Rx symbols.");
    require(eq_symbols.get_dimension_size(re_list::dims::re) == nre && eq_symbols.get_dimension_size(re_list::dims::slice) == nl,
            "Equalized symbols do not match the channel dimensions.");
    require(eq_noise_vars.get_dimension_size(re_list::dims::re) == nre && eq_noise_vars.get_dimension_size(re_list::dims::slice) == nl,
            "Post-equalization noise variances do not match the channel dimensions.");
    require(noise_var_estimates.size() == npt, "Number of noise variance estimates does not match the number of Rx ports.");
    require(tx_scaling > 0, "Tx scaling factor must be positive.");
    require((nl == 1 && npt >= 1 && npt <= 4) || (nl == 2 && npt == 2), "Invalid channel spatial topology.");
    if (nre == 0) {
      return;
    }
    miphy_equalizer_job j = {};
    j.nof_re = nre, j.nof_rx_ports = static_cast<uint8_t>(npt), j.nof_tx_layers = static_cast<uint8_t>(nl);
    j.noise_var = noise_var_estimates[0], j.tx_scaling = tx_scaling;
    const size_t ny = static_cast<size_t>(npt) * nre, nh = ny * nl, nz = static_cast<size_t>(nl) * nre;
    auto*        d_y = static_cast<float*>(c->buf(0, ny * sizeof(srsran::cf_t)));
    auto*        d_h = static_cast<float*>(c->buf(1, nh * sizeof(srsran::cf_t)));
    auto*        d_z = static_cast<float*>(c->buf(2, nz * sizeof(srsran::cf_t)));
    auto*        d_v = static_cast<float*>(c->buf(3, nz * sizeof(float)));
    c->h2d(d_y, ch_symbols.get_view<2>({}).data(), ny * sizeof(srsran::cf_t));
    c->h2d(d_h, ch_estimates.get_view<3>({}).data(), nh * sizeof(srsran::cf_t));
    context::check(miphy_channel_equalize_batch(c->ctx, &j, 0, 1, d_y, d_h, d_z, d_v, c->stream), "channel_equalize");
    c->d2h(eq_symbols.get_view<2>({}).data(), d_z, nz * sizeof(srsran::cf_t));
    c->d2h(eq_noise_vars.get_view<2>({}).data(), d_v, nz * sizeof(float));
    c->sync();
  }

private:
  std::shared_ptr<context> c;
};

class channel_equalizer_factory_hip : public srsran::channel_equalizer_factory
{
public:
  explicit channel_equalizer_factory_hip(std::shared_ptr<context> c) : c(std::move(c)) {}
  std::unique_ptr<srsran::channel_equalizer> create() override { return std::make_unique<channel_equalizer_hip>(c); }

private:
  std::shared_ptr<context> c;
};
inline std::shared_ptr<srsran::channel_equalizer_factory> create_channel_equalizer_factory_hip(std::shared_ptr<context> c)
{
  return std::make_shared<channel_equalizer_factory_hip>(std::move(c));
}

// ---------------------------------------------------------------------------------------------------------------- PDU validators
/// The validators the upper PHY asks the processor factories for (upper_phy_factories.cpp:96-99,329-334) and calls per PDU
/// (upper_phy_pdu_validators.h:71-74): the reference's own checks (its validator, built from the reference factory over the HIP
/// blocks) AND the restrictions of the device path, so that the MAC gets a clean rejection instead of an assertion.
class pusch_pdu_validator_hip : public srsran::pusch_pdu_validator
{
public:
  pusch_pdu_validator_hip(std::unique_ptr<srsran::pusch_pdu_validator> ref, bool uci_supported) : ref(std::move(ref)), uci_supported(uci_supported) {}
  bool is_valid(const srsran::pusch_processor::pdu_t& pdu) const override
  {
    if (!ref->is_valid(pdu)) {
      return false;
    }
    const bool has_uci = pdu.uci.nof_harq_ack != 0 || pdu.uci.nof_csi_part1 != 0 || pdu.uci.nof_csi_part2 != 0;
    // device path: multiplexed UCI only with a UCI decoder behind it, a codeword or UCI present, one layer, at most four receive
    // ports, normal cyclic prefix
    return (pdu.codeword.has_value() || has_uci) && (!has_uci || uci_supported) && pdu.nof_tx_layers == 1 && pdu.rx_ports.size() >= 1 &&
           pdu.rx_ports.size() <= 4 && pdu.cp == srsran::cyclic_prefix::NORMAL && pdu.mcs_descr.modulation != srsran::modulation_scheme::BPSK;
  }

private:
  std::unique_ptr<srsran::pusch_pdu_validator> ref;
  bool                                         uci_supported;
};

class pdsch_pdu_validator_hip : public srsran::pdsch_pdu_validator
{
public:
  explicit pdsch_pdu_validator_hip(std::unique_ptr<srsran::pdsch_pdu_validator> ref) : ref(std::move(ref)) {}
  bool is_valid(const srsran::pdsch_processor::pdu_t& pdu) const override
  {
    if (!ref->is_valid(pdu)) {
      return false;
    }
    // device path: one codeword on one layer, DM-RS type 1, contiguous (non-interleaved) allocation, at most four reserved patterns
    return pdu.codewords.size() == 1 && pdu.ports.size() == 1 && pdu.dmrs == srsran::dmrs_type::TYPE1 && pdu.freq_alloc.is_contiguous() &&
           pdu.cp == srsran::cyclic_prefix::NORMAL && pdu.reserved.get_nof_entries() <= 4;
  }

private:
  std::unique_ptr<srsran::pdsch_pdu_validator> ref;
};

inline std::unique_ptr<srsran::pusch_pdu_validator> pusch_processor_factory_hip::create_validator()
{
  srsran::uci_decoder_factory_sw_configuration uc;
  uc.decoder_factory = srsran::create_short_block_detector_factory_sw();
  srsran::pusch_processor_factory_sw_configuration pc;
  pc.estimator_factory                    = std::make_shared<dmrs_pusch_estimator_factory_hip>(c);
  pc.demodulator_factory                  = std::make_shared<pusch_demodulator_factory_hip>(c);
  pc.demux_factory                        = srsran::create_ulsch_demultiplex_factory_sw();
  pc.decoder_factory                      = std::make_shared<pusch_decoder_factory_hip>(c);
  pc.uci_dec_factory                      = srsran::create_uci_decoder_factory_sw(uc);
  pc.ch_estimate_dimensions.nof_prb       = srsran::MAX_RB;
  pc.ch_estimate_dimensions.nof_symbols   = srsran::MAX_NSYMB_PER_SLOT;
  pc.ch_estimate_dimensions.nof_rx_ports  = 4;
  pc.ch_estimate_dimensions.nof_tx_layers = 1;
  pc.dec_nof_iterations                   = nof_iterations;
  pc.dec_enable_early_stop                = early_stop;
  return std::make_unique<pusch_pdu_validator_hip>(srsran::create_pusch_processor_factory_sw(pc)->create_validator(), uci_dec_factory != nullptr);
}

inline std::unique_ptr<srsran::pdsch_pdu_validator> pdsch_processor_factory_hip::create_validator()
{
  auto ref = srsran::create_pdsch_processor_factory_sw(std::make_shared<pdsch_encoder_factory_hip>(c), std::make_shared<pdsch_modulator_factory_hip>(c),
                                                       std::make_shared<dmrs_pdsch_processor_factory_hip>(c));
  return std::make_unique<pdsch_pdu_validator_hip>(ref->create_validator());
}

// The PDCCH / SSB / CSI-RS device paths take everything the reference processors take: the reference's validators apply unchanged.
inline std::unique_ptr<srsran::pdcch_pdu_validator> pdcch_processor_factory_hip::create_validator()
{
  auto prg = srsran::create_pseudo_random_generator_sw_factory();
  auto ref = srsran::create_pdcch_processor_factory_sw(std::make_shared<pdcch_encoder_factory_hip>(c),
                                                       srsran::create_pdcch_modulator_factory_sw(srsran::create_channel_modulation_sw_factory(), prg),
                                                       srsran::create_dmrs_pdcch_processor_factory_sw(prg));
  return ref->create_validator();
}

inline std::unique_ptr<srsran::ssb_pdu_validator> ssb_processor_factory_hip::create_validator()
{
  auto                                       prg = srsran::create_pseudo_random_generator_sw_factory();
  srsran::ssb_processor_factory_sw_configuration sc;
  sc.encoder_factory   = srsran::create_pbch_encoder_factory_sw(srsran::create_crc_calculator_factory_sw("auto"), prg, srsran::create_polar_factory_sw());
  sc.modulator_factory = srsran::create_pbch_modulator_factory_sw(srsran::create_channel_modulation_sw_factory(), prg);
  sc.dmrs_factory      = srsran::create_dmrs_pbch_processor_factory_sw(prg);
  sc.pss_factory       = srsran::create_pss_processor_factory_sw();
  sc.sss_factory       = srsran::create_sss_processor_factory_sw();
  return srsran::create_ssb_processor_factory_sw(sc)->create_validator();
}

inline std::unique_ptr<srsran::nzp_csi_rs_configuration_validator> nzp_csi_rs_generator_factory_hip::create_validator()
{
  return srsran::create_nzp_csi_rs_generator_factory_sw(srsran::create_pseudo_random_generator_sw_factory())->create_validator();
}

// --------------------------------------------------------------------------------------------------- multi-GPU placement
/// Device placement of a node with several GPUs: the path shards by cell / transport block with no cross-unit
/// dependency, exactly like the reference runs one processor instance per worker (lib/phy/upper/uplink_processor_concurrent.h:41-54).
/// One context per device; a CELL is pinned to one device -- `cell % devices` -- so that everything stateful of its UEs (the HARQ
/// soft buffers of rx_softbuffer_pool_hip, which a retransmission must find where the first transmission left them) lives on
/// that device and no PHY data ever crosses xGMI. One HARQ pool per device, created on first use with the configuration given here.
/// A thread that drives a cell calls bind_thread(cell) once (a new thread starts on device 0).
class device_placement
{
public:
  /// \param nof_devices Devices to use, -1 = all visible ones.
  explicit device_placement(const srsran::rx_softbuffer_pool_config& pool_config_, int nof_devices = -1) : pool_config(pool_config_)
  {
    int n = 0;
    context::hip(hipGetDeviceCount(&n), "hipGetDeviceCount");
    if (nof_devices > 0 && nof_devices < n) {
      n = nof_devices;
    }
    require(n > 0, "No GPU visible.");
    for (int d = 0; d != n; ++d) {
      contexts.emplace_back(std::make_shared<context>(d));
    }
    pools.resize(contexts.size());
  }
  unsigned nof_devices() const { return contexts.size(); }
  /// Device index a cell is pinned to.
  unsigned device_of_cell(unsigned cell_index) const { return cell_index % contexts.size(); }
  /// The context every block of that cell is created with (create_*_factory_hip(placement.context_of_cell(cell))).
  std::shared_ptr<context> context_of_cell(unsigned cell_index) const { return contexts[device_of_cell(cell_index)]; }
  /// The HARQ pool of the cell's device (shared by the cells pinned to it).
  srsran::rx_softbuffer_pool& softbuffer_pool_of_cell(unsigned cell_index)
  {
    const unsigned              d = device_of_cell(cell_index);
    std::lock_guard<std::mutex> lk(mutex);
    if (!pools[d]) {
      contexts[d]->bind_thread();
      pools[d] = create_rx_softbuffer_pool_hip(contexts[d], pool_config);
    }
    return *pools[d];
  }
  /// Makes the cell's device the current one of the calling thread.
  void bind_thread(unsigned cell_index) const { contexts[device_of_cell(cell_index)]->bind_thread(); }

private:
  srsran::rx_softbuffer_pool_config                         pool_config;
  std::vector<std::shared_ptr<context>>                     contexts;
  std::vector<std::unique_ptr<srsran::rx_softbuffer_pool>>  pools;
  std::mutex                                                mutex;
};

} // namespace miphy
