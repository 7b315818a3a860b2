// tm_tables.hip -- constant tables of the reference + LUT construction, error state, device check.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "tm_common.h"

namespace tmx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char *get_error() { return g_err; }

// ---- small read-backs through page-locked memory (HostRead, tm_common.h) ---------------------------------------------
namespace {
struct PinnedArea {
  uint8_t *p = nullptr;
  size_t used = 0;
  int depth = 0;  // live HostRead objects of this thread: the area is handed out bump-wise and starts over when the last one goes
  bool failed = false;
  static constexpr size_t CAP = 64 * 1024;  // (never freed: a thread's 64 KB, and thread exit may come after the runtime has gone)
};
thread_local PinnedArea t_pin;
}  // namespace

int *pinned_words() {
  static thread_local int *w = nullptr;
  static thread_local bool tried = false;
  if (!w && !tried) {
    tried = true;
    void *q = nullptr;
    if (hipHostMalloc(&q, 64, hipHostMallocPortable) == hipSuccess) w = (int *)q;
    else (void)hipGetLastError();
  }
  return w;
}

HostRead::HostRead(hipStream_t s) : stream(s) { t_pin.depth++; }
HostRead::~HostRead() {
  // copies queued and never waited for (an error return between get() and wait()) may still be in flight into the thread's area: they
  // must land before the area is handed out again
  if (n > 0) (void)hipStreamSynchronize(stream);
  if (--t_pin.depth == 0) t_pin.used = 0;
}
int HostRead::get(void *dst, const void *src, size_t bytes) {
  if (bytes == 0) return TM_OK;
  if (!t_pin.p && !t_pin.failed) {
    void *q = nullptr;
    if (hipHostMalloc(&q, PinnedArea::CAP, hipHostMallocPortable) == hipSuccess) t_pin.p = (uint8_t *)q;
    else { (void)hipGetLastError(); t_pin.failed = true; }  // no page-locked memory to be had: the plain copy is still correct
  }
  const size_t off = (t_pin.used + 15) & ~(size_t)15;
  if (!t_pin.p || n >= 8 || off + bytes > PinnedArea::CAP) {  // straight to the destination (stream-ordered like the others)
    TM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
    return TM_OK;
  }
  TM_HIP(hipMemcpyAsync(t_pin.p + off, src, bytes, hipMemcpyDeviceToHost, stream));
  t_pin.used = off + bytes;
  items[n++] = Item{dst, off, bytes};
  return TM_OK;
}
int HostRead::wait() {
  TM_HIP(hipStreamSynchronize(stream));
  for (int i = 0; i < n; i++) memcpy(items[i].dst, t_pin.p + items[i].off, items[i].bytes);
  n = 0;
  return TM_OK;
}

// ---- device memory pool (see tm_common.h) --------------------------------------------------------------------------
namespace {
struct PoolBlock { void *p; size_t bytes; int device; };
struct Pool {
  std::vector<PoolBlock> blocks;
  size_t held = 0;
  ~Pool() { for (auto &b : blocks) (void)hipFree(b.p); }
};
thread_local Pool t_pool;
size_t pool_cap() {
  static const size_t cap = [] {
    const char *e = getenv("TM_POOL_GIB");
    return (size_t)(e ? atof(e) : 96.0) * ((size_t)1 << 30);
  }();
  return cap;
}
}  // namespace

void pool_trim() {
  for (auto &b : t_pool.blocks) (void)hipFree(b.p);
  t_pool.blocks.clear();
  t_pool.held = 0;
}

int pool_alloc(void **p, size_t *bytes) {
  // size classes: multiples of 1/8 of the next power of two below the size (at most 12.5 % slack), 256-byte floor
  size_t n = std::max<size_t>(*bytes, 256);
  size_t step = 256;
  while (step * 16 <= n) step <<= 1;
  n = (n + step - 1) / step * step;
  int dev = 0;
  (void)hipGetDevice(&dev);
  int best = -1;
  for (int i = 0; i < (int)t_pool.blocks.size(); i++) {
    const PoolBlock &b = t_pool.blocks[i];
    if (b.device == dev && b.bytes >= n && b.bytes <= n + n / 4 && (best < 0 || b.bytes < t_pool.blocks[best].bytes)) best = i;
  }
  if (best >= 0) {
    *p = t_pool.blocks[best].p;
    *bytes = t_pool.blocks[best].bytes;
    t_pool.held -= t_pool.blocks[best].bytes;
    t_pool.blocks.erase(t_pool.blocks.begin() + best);
    return TM_OK;
  }
  hipError_t e = hipMalloc(p, n);
  if (e != hipSuccess) {  // give the pooled memory back and try once more
    (void)hipGetLastError();
    pool_trim();
    e = hipMalloc(p, n);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *p = nullptr;
    set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
    return TM_E_NOMEM;
  }
  *bytes = n;
  return TM_OK;
}

void pool_free(void *p, size_t bytes) {
  if (!p) return;
  if (t_pool.held + bytes > pool_cap()) { (void)hipFree(p); return; }
  int dev = 0;
  (void)hipGetDevice(&dev);
  t_pool.blocks.push_back({p, bytes, dev});
  t_pool.held += bytes;
}

namespace { thread_local Knobs t_knobs; }
const Knobs &knobs() { return t_knobs; }
void knobs_reload() {
  auto on = [](const char *name) { const char *v = getenv(name); return v != nullptr && !(v[0] == '0' && v[1] == 0); };  // set, and not "0"
  Knobs k;
  k.knn_debug = on("TM_KNN_DEBUG"); k.knn_noprune = on("TM_KNN_NOPRUNE"); k.topk_brute = on("TM_TOPK_BRUTE"); k.no_query_groups = on("TM_NO_QUERY_GROUPS");
  k.dither_own_keys = on("TM_DITHER_OWN_KEYS"); k.dither_no_dedup = on("TM_DITHER_NO_DEDUP"); k.dither_literal = on("TM_DITHER_LITERAL");
  k.dedup_plain = on("TM_DEDUP_PLAIN"); k.dedup_sort = on("TM_DEDUP_SORT"); k.dedup_degrade_hash = on("TM_DEDUP_DEGRADE_HASH"); k.dedup_full_order = on("TM_DEDUP_FULL_ORDER");
  k.motion_valu = on("TM_MOTION_VALU"); k.pp_debug = on("TM_PP_DEBUG"); k.comm_force_dist = on("TM_COMM_FORCE_DIST"); k.features_plain = on("TM_FEATURES_PLAIN"); k.km_launches = on("TM_KM_LAUNCHES"); k.kmodes_binwise = on("TM_KMODES_BINWISE"); k.kmodes_fast_always = on("TM_KMODES_FAST_ALWAYS"); k.pp_sharded = on("TM_PP_SHARDED"); k.window_dcts_by_tile = on("TM_WINDOW_DCTS_BY_TILE"); k.km_resident_fail = on("TM_KM_RESIDENT_FAIL"); k.features_by_tile = on("TM_FEATURES_BY_TILE"); k.motion_pack_separate = on("TM_MOTION_PACK_SEPARATE"); k.motion_force_flag = on("TM_MOTION_FORCE_FLAG");
  if (const char *v = getenv("TM_TOPK_ESTIMATE")) k.topk_estimate = (v[0] == '0' && v[1] == 0) ? 0 : 1;
  if (const char *v = getenv("TM_EPU_TABLE_GIB")) k.epu_table_gib = atof(v);
  if (const char *v = getenv("TM_COMM_TIMEOUT_S")) k.comm_timeout_s = std::max(1.0, atof(v));
  if (const char *v = getenv("TM_DEDUP_RADIX_MIN")) k.dedup_radix_min = std::max(0ll, atoll(v));
  if (const char *v = getenv("TM_KNN_ARENA_ENTRIES")) k.knn_arena_entries = std::max(0ll, atoll(v));
  t_knobs = k;
}

int require_device() {
  knobs_reload();  // every compute entry point of the C ABI comes through here first
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("no HIP device available (%s); libtilemotion has no CPU path", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    return TM_E_NODEVICE;
  }
  return TM_OK;
}

// cDitheringMap, utils.pas:47-56
const uint8_t kDitheringMap[64] = {0,  48, 12, 60, 3,  51, 15, 63, 32, 16, 44, 28, 35, 19, 47, 31, 8,  56, 4,  52, 11, 59,
                                   7,  55, 40, 24, 36, 20, 43, 27, 39, 23, 2,  50, 14, 62, 1,  49, 13, 61, 34, 18, 46, 30,
                                   33, 17, 45, 29, 10, 58, 6,  54, 9,  57, 5,  53, 42, 26, 38, 22, 41, 25, 37, 21};

// cDCTSnake, utils.pas:59-68 (raster v*8+u -> zig-zag rank)
const uint8_t kDCTSnake[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
                               41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
                               46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// cDCTWeights, utils.pas:72-97 (Daala PSNR-HVS CSF, Y/U/V)
const double kDCTWeights[3][8][8] = {
    {{1.6193873005, 2.2901594831, 2.08509755623, 1.48366094411, 1.00227514334, 0.678296995242, 0.466224900598, 0.3265091542},
     {2.2901594831, 1.94321815382, 2.04793073064, 1.68731108984, 1.2305666963, 0.868920337363, 0.61280991668, 0.436405793551},
     {2.08509755623, 2.04793073064, 1.34329019223, 1.09205635862, 0.875748795257, 0.670882927016, 0.501731932449, 0.372504254596},
     {1.48366094411, 1.68731108984, 1.09205635862, 0.772819797575, 0.605636379554, 0.48309405692, 0.380429446972, 0.295774038565},
     {1.00227514334, 1.2305666963, 0.875748795257, 0.605636379554, 0.448996256676, 0.352889268808, 0.283006984131, 0.226951348204},
     {0.678296995242, 0.868920337363, 0.670882927016, 0.48309405692, 0.352889268808, 0.27032073436, 0.215017739696, 0.17408067321},
     {0.466224900598, 0.61280991668, 0.501731932449, 0.380429446972, 0.283006984131, 0.215017739696, 0.168869545842, 0.136153931001},
     {0.3265091542, 0.436405793551, 0.372504254596, 0.295774038565, 0.226951348204, 0.17408067321, 0.136153931001, 0.109083846276}},
    {{1.91113096927, 2.46074210438, 1.18284184739, 1.14982565193, 1.05017074788, 0.898018824055, 0.74725392039, 0.615105596242},
     {2.46074210438, 1.58529308355, 1.21363250036, 1.38190029285, 1.33100189972, 1.17428548929, 0.996404342439, 0.830890433625},
     {1.18284184739, 1.21363250036, 0.978712413627, 1.02624506078, 1.03145147362, 0.960060382087, 0.849823426169, 0.731221236837},
     {1.14982565193, 1.38190029285, 1.02624506078, 0.861317501629, 0.801821139099, 0.751437590932, 0.685398513368, 0.608694761374},
     {1.05017074788, 1.33100189972, 1.03145147362, 0.801821139099, 0.676555426187, 0.605503172737, 0.55002013668, 0.495804539034},
     {0.898018824055, 1.17428548929, 0.960060382087, 0.751437590932, 0.605503172737, 0.514674450957, 0.454353482512, 0.407050308965},
     {0.74725392039, 0.996404342439, 0.849823426169, 0.685398513368, 0.55002013668, 0.454353482512, 0.389234902883, 0.342353999733},
     {0.615105596242, 0.830890433625, 0.731221236837, 0.608694761374, 0.495804539034, 0.407050308965, 0.342353999733, 0.295530605237}},
    {{2.03871978502, 2.62502345193, 1.26180942886, 1.11019789803, 1.01397751469, 0.867069376285, 0.721500455585, 0.593906509971},
     {2.62502345193, 1.69112867013, 1.17180569821, 1.3342742857, 1.28513006198, 1.13381474809, 0.962064122248, 0.802254508198},
     {1.26180942886, 1.17180569821, 0.944981930573, 0.990876405848, 0.995903384143, 0.926972725286, 0.820534991409, 0.706020324706},
     {1.11019789803, 1.3342742857, 0.990876405848, 0.831632933426, 0.77418706195, 0.725539939514, 0.661776842059, 0.587716619023},
     {1.01397751469, 1.28513006198, 0.995903384143, 0.77418706195, 0.653238524286, 0.584635025748, 0.531064164893, 0.478717061273},
     {0.867069376285, 1.13381474809, 0.926972725286, 0.725539939514, 0.584635025748, 0.496936637883, 0.438694579826, 0.393021669543},
     {0.721500455585, 0.962064122248, 0.820534991409, 0.661776842059, 0.531064164893, 0.438694579826, 0.375820256136, 0.330555063063},
     {0.593906509971, 0.802254508198, 0.706020324706, 0.587716619023, 0.478717061273, 0.393021669543, 0.330555063063, 0.285345396658}}};

// cDCTUVRatio, utils.pas:100-109 (array of TFloat: the sqrt(0.5) entries are Singles)
static float uv_ratio(int v, int u) {
  if (v == 0 && u == 0) return 0.5f;
  if (v == 0 || u == 0) return static_cast<float>(std::sqrt(0.5));
  return 1.0f;
}

static std::mutex g_tab_mu;
static DeviceTables g_tab[16];
static bool g_tab_ready[16] = {false};

template <class T> static int upload(T **dst, const T *src, size_t n) {
  TM_HIP(hipMalloc(reinterpret_cast<void **>(dst), n * sizeof(T)));
  TM_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return TM_OK;
}

int get_tables(const DeviceTables **out) {
  TM_TRY(require_device());
  int dev = 0;
  TM_HIP(hipGetDevice(&dev));
  TM_CHECK(dev >= 0 && dev < 16, TM_E_INVAL, "device ordinal %d out of range", dev);
  std::lock_guard<std::mutex> lk(g_tab_mu);
  if (!g_tab_ready[dev]) {
    DeviceTables &t = g_tab[dev];
    // InitLuts, tilingencoder.pas:1703-1714: index ((v*8+u)*8+y)*8+x, evaluated in double, stored also as Single
    std::vector<double> l64(4096);
    std::vector<float> l32(4096);
    for (int special = 0; special < 2; special++) {
      const double div = special ? 16.0 : 8.0;
      int i = 0;
      for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++)
          for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++, i++) {
              double val = std::cos((x + 0.5) * u * M_PI / div) * std::cos((y + 0.5) * v * M_PI / div) * (double)uv_ratio(v, u);
              l64[i] = val;
              l32[i] = static_cast<float>(val);
            }
      TM_TRY(upload(&t.dct_lut_f64[special], l64.data(), 4096));
      TM_TRY(upload(&t.dct_lut_f32[special], l32.data(), 4096));
      double cs[64];
      for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++) cs[u * 8 + x] = std::cos((x + 0.5) * u * M_PI / div);
      TM_TRY(upload(&t.dct_cos_f64[special], cs, 64));
    }
    TM_TRY(upload(&t.weights, &kDCTWeights[0][0][0], 192));
    float srgb[256];  // utils.pas:378-384: r := ir/255.0 (Single); gamma expansion; stored back into a Single
    for (int c = 0; c < 256; c++) {
      float r = static_cast<float>(c / 255.0);
      srgb[c] = ((double)r > 0.04045) ? static_cast<float>(std::pow(((double)r + 0.055) / 1.055, 2.4))
                                      : static_cast<float>((double)r / 12.92);
    }
    TM_TRY(upload(&t.srgb_lut, srgb, 256));
    TM_TRY(upload(&t.snake, kDCTSnake, 64));
    TM_TRY(upload(&t.dither_map, kDitheringMap, 64));
    g_tab_ready[dev] = true;
  }
  *out = &g_tab[dev];
  return TM_OK;
}

}  // namespace tmx
