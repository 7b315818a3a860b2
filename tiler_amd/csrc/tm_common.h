// tm_common.h -- shared host-side helpers of libtilemotion (error plumbing, device buffers, tables).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tilemotion.h"

namespace tmx {

void set_error(const char *fmt, ...);
const char *get_error();

#define TM_HIP(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      tmx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);        \
      return (e_ == hipErrorOutOfMemory) ? TM_E_NOMEM : TM_E_HIP;                                      \
    }                                                                                                  \
  } while (0)

#define TM_CHECK(cond, code, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      tmx::set_error(__VA_ARGS__);        \
      return (code);                     \
    }                                    \
  } while (0)

#define TM_TRY(expr)          \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != TM_OK) return rc_; \
  } while (0)

// Fails loudly when no gfx950 device can run the kernels (there is no CPU path).
int require_device();

// The library's environment switches (the table in include/tilemotion.h documents them).  They are sampled at the API boundary -- tm_create,
// tm_run, every tm_stage_* and fine-seam entry (knobs_reload, through require_device) -- into the calling thread's set; nothing inside a step
// calls getenv.
struct Knobs {
  bool knn_debug = false, knn_noprune = false, topk_brute = false, no_query_groups = false, dither_own_keys = false, dither_no_dedup = false,
       dither_literal = false, dedup_plain = false, dedup_sort = false, dedup_degrade_hash = false, dedup_full_order = false, motion_valu = false, pp_debug = false,
       comm_force_dist = false, features_plain = false, km_launches = false, kmodes_binwise = false, kmodes_fast_always = false, pp_sharded = false, window_dcts_by_tile = false, km_resident_fail = false, features_by_tile = false, motion_pack_separate = false, motion_force_flag = false;
  int topk_estimate = -1;  // TM_TOPK_ESTIMATE: the k-nearest search's first thresholds from a sample of the database: -1 by size, 0 never, 1 whenever possible
  double epu_table_gib = 6.0, comm_timeout_s = 120.0;
  long long knn_arena_entries = 0, dedup_radix_min = 1ll << 20;
};
const Knobs &knobs();
void knobs_reload();

// Device memory pool: hipMalloc / hipFree of the pipeline's multi-GB temporaries cost ~16 ms per 720p clip (and hipFree
// synchronises the device), so freed blocks are kept per host thread and handed out again.  A thread drives its encoder on
// one stream, so reuse is stream-ordered; blocks never migrate between threads.  pool_trim() returns everything to the
// driver (called when an encoder is destroyed, and automatically when an allocation fails).
int pool_alloc(void **p, size_t *bytes_inout);
void pool_free(void *p, size_t bytes);
void pool_trim();

// RAII device buffer
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) pool_free(p, bytes);
    p = nullptr;
    bytes = 0;
  }
  int alloc(size_t n) {
    if (n <= bytes && p) return TM_OK;
    release();
    if (n == 0) n = 16;
    size_t got = n;
    const int rc = pool_alloc(&p, &got);
    if (rc != TM_OK) { p = nullptr; return rc; }
    bytes = got;
    return TM_OK;
  }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Small read-backs (counts, flags, a handful of sums) through page-locked memory: a device-to-host copy into pageable memory -- a stack
// variable -- takes the runtime's staging path and costs 27 us behind a small kernel where the same copy into pinned memory costs 15
// (tools/micro/readback.hip; a step holds 50-100 of them).  Scoped: the destinations must outlive the object; copies queued and not
// waited for are dropped.  The staging area is the calling thread's (64 KB; larger copies go straight to their destination).
struct HostRead {
  explicit HostRead(hipStream_t s);
  ~HostRead();
  HostRead(const HostRead &) = delete;
  HostRead &operator=(const HostRead &) = delete;
  int get(void *dst, const void *src, size_t bytes);  // queue
  int wait();                                         // synchronise the stream, then fill the destinations
 private:
  struct Item { void *dst; size_t off, bytes; };
  hipStream_t stream;
  Item items[8];
  int n = 0;
};

// sixteen page-locked 32-bit words of the calling thread (nullptr if none are to be had): for flags a host loop polls behind events
// without draining the stream
int *pinned_words();

// Constant tables of the reference (utils.pas:47-109) + LUTs of InitLuts (tilingencoder.pas:1683-1727),
// uploaded once per device.
struct DeviceTables {
  float *dct_lut_f32[2] = {nullptr, nullptr};   // FDCTLut[special][4096]
  double *dct_lut_f64[2] = {nullptr, nullptr};  // FDCTLutDouble
  double *dct_cos_f64[2] = {nullptr, nullptr};  // the LUT's two cosine factors, cos((x + 0.5) u pi / div) as [u][x] (the int16 feature kernel's fast path)
  double *weights = nullptr;                    // cDCTWeights [3][8][8]
  float *srgb_lut = nullptr;                    // inverse sRGB of c/255, as Single
  uint8_t *snake = nullptr;                     // cDCTSnake
  uint8_t *dither_map = nullptr;                // cDitheringMap
};
int get_tables(const DeviceTables **out);

extern const uint8_t kDitheringMap[64];
extern const uint8_t kDCTSnake[64];
extern const double kDCTWeights[3][8][8];

inline bool mode_special(int mode) { return mode == TM_PVS_SPE_DCT || mode == TM_PVS_WEIGHTED_SPE_DCT; }
inline bool mode_weighted(int mode) { return mode == TM_PVS_WEIGHTED_DCT || mode == TM_PVS_WEIGHTED_SPE_DCT; }

}  // namespace tmx
