// tm_encoder.hip -- coarse seam: the TTilingEncoder object behind tm_create/tm_run (include/tilemotion.h).
//
// Mirrors TTilingEncoder (tilingencoder.pas:308-568): settings with the reference's clamps (2919-3047) and INI keys
// (3745-3770), Run(step) walking esLoad..esSave (5529-5554), read-only Tiles/Frames/Palettes views.  Everything a
// step computes stays resident in HBM between steps; only small control data (correlations, digit plans, palette
// colours, counts) crosses to the host.  Every step of Run(esAll) is built (motion prediction, the extended-palette
// re-rank, OptimizePalettes, the .gtm writer and reader included); what stays outside the path is listed in DESIGN.md "Scope".
#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <fstream>
#include <map>
#include <sstream>

#include <rccl/rccl.h>

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {

// ---- small device helpers ------------------------------------------------------------------------------------
__global__ void k_gather_rows16(const uint4 *__restrict__ src, const int32_t *__restrict__ idx, int64_t n, int vec_per_row,
                                uint4 *__restrict__ dst) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * vec_per_row; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / vec_per_row;
    const int v = (int)(e - r * vec_per_row);
    dst[e] = src[(int64_t)idx[r] * vec_per_row + v];
  }
}
template <class T> __global__ void k_gather(const T *__restrict__ src, const int32_t *__restrict__ idx, int64_t n, T *__restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}
__global__ void k_clip_index(int32_t *__restrict__ idx, int64_t n, int32_t limit) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (idx[i] >= limit) idx[i] = -1;
}
__global__ void k_tilemap_from_subset(const int32_t *__restrict__ keep, const int32_t *__restrict__ pos, const int32_t *__restrict__ sub_remap,
                                      int64_t n, int32_t *__restrict__ tm_tile) {  // TransferTiles: TMI^.TileIdx := tIdx / -1 (4079-4083), then the remaps
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    tm_tile[i] = keep[i] ? sub_remap[pos[i]] : -1;
}
// use counts.  Neighbouring items often name the same tile (flat areas: one tile can own a tenth of the clip, and its counter then
// serialises every atomic of the launch), so a wave adds a RUN of equal indices with one atomic: heads of runs by comparing with the lane
// before, run lengths off the ballot of heads.
__global__ __launch_bounds__(256) void k_histogram(const int32_t *__restrict__ idx, int64_t n, uint32_t *__restrict__ hist) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x; i0 < n; i0 += stride) {  // (workgroup-uniform bound: every lane reaches the ballot)
    const int64_t i = i0 + threadIdx.x;
    const int v = i < n ? idx[i] : -1;
    const int prev = __shfl_up(v, 1);
    const bool head = lane == 0 || v != prev;
    const unsigned long long heads = __ballot(head);
    if (head && v >= 0) {
      const unsigned long long rest = lane == 63 ? 0ull : heads >> (lane + 1);
      const int len = rest ? __ffsll((long long)rest) : 64 - lane;
      atomicAdd(&hist[v], (uint32_t)len);
    }
  }
}
// The same into one copy of the histogram PER XCD: a workgroup adds to the copy of the XCD it runs on (the id is read from the hardware;
// placement only decides which copy, any copy is right) and k_hist_fold adds the eight copies up.  Every XCD has its own L2: an atomic on
// a word that all eight keep adding to travels between them every time, while a word only one XCD touches stays in that XCD's L2 --
// 0.49 -> 0.2 ms for the 4.3 M tile-map items of the bench clip (memset of the copies and the fold included); agent scope or workgroup
// scope measured the same, so the scope stays the one the memory model asks for.
__global__ __launch_bounds__(256) void k_histogram_xcd(const int32_t *__restrict__ idx, int64_t n, uint32_t *__restrict__ hist8, int64_t bins) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
  uint32_t *hist = hist8 + (int64_t)(xcc & 7u) * bins;
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x; i0 < n; i0 += stride) {
    const int64_t i = i0 + threadIdx.x;
    const int v = i < n ? idx[i] : -1;
    const int prev = __shfl_up(v, 1);
    const bool head = lane == 0 || v != prev;
    const unsigned long long heads = __ballot(head);
    if (head && v >= 0) {
      const unsigned long long rest = lane == 63 ? 0ull : heads >> (lane + 1);
      const int len = rest ? __ffsll((long long)rest) : 64 - lane;
      atomicAdd(&hist[v], (uint32_t)len);
    }
  }
}
__global__ void k_hist_fold(const uint32_t *__restrict__ hist8, int64_t bins, uint32_t *__restrict__ hist) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < bins; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t s = 0;
#pragma unroll
    for (int x = 0; x < 8; x++) s += hist8[x * bins + i];
    hist[i] = s;
  }
}
__global__ void k_lookup(const int32_t *__restrict__ idx, int64_t n, const int32_t *__restrict__ table, int32_t *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = idx[i] >= 0 ? table[idx[i]] : -1;
}
__global__ void k_lookup_inplace(int32_t *__restrict__ idx, int64_t n, const int32_t *__restrict__ table) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (idx[i] >= 0) idx[i] = table[idx[i]];
}
// sharded Reduce: one record per locally distinct tile = 64 pixel dwords + use count + mirror flags
__global__ void k_pack_unique(const uint32_t *__restrict__ tiles, const uint8_t *__restrict__ flags, const int32_t *__restrict__ order,
                              const uint32_t *__restrict__ use, int64_t n, uint32_t *__restrict__ rec) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * 66; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / 66;
    const int v = (int)(e - r * 66);
    rec[e] = v < 64 ? tiles[(int64_t)order[r] * 64 + v] : v == 64 ? use[r] : (uint32_t)flags[order[r]];
  }
}
__global__ void k_unpack_unique(const uint32_t *__restrict__ rec, int64_t n, uint32_t *__restrict__ tiles, uint32_t *__restrict__ use, uint8_t *__restrict__ flags) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * 66; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / 66;
    const int v = (int)(e - r * 66);
    if (v < 64) tiles[r * 64 + v] = rec[e]; else if (v == 64) use[r] = rec[e]; else flags[r] = (uint8_t)rec[e];
  }
}

// a frame tile's global index through the candidates: its local distinct tile travelled (in_s) as candidate number cand_pos[.] of this
// process, which the exact dedup of all candidates mapped to cand_remap[.]; anything else is beyond the tile budget
__global__ void k_compose_remap_cand(const int32_t *__restrict__ local_remap, int64_t n, const uint32_t *__restrict__ in_s, const int32_t *__restrict__ cand_pos,
                                     const int32_t *__restrict__ cand_remap, int32_t cand_off, int32_t limit, int32_t *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t l = local_remap[i];
    int32_t g = -1;
    if (in_s[l]) g = cand_remap[cand_off + cand_pos[l]];
    out[i] = g >= 0 && g < limit ? g : -1;
  }
}
static inline int gridn(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 16)); }

static int equal_quality_tile_count(double tc) {  // EqualQualityTileCount, utils.pas:1038-1041 (TFloat argument)
  const float f = (float)tc;
  return (int)llrint(std::sqrt((double)f) * std::log2(1 + (double)f));
}


struct Settings {
  std::string InputFileName, OutputFileName;
  int StartFrame = 0, FrameCount = 0;
  double Scaling = 1.0;
  int MotionPredictRadius = 32;
  bool GlobalTilingUseTargetPSNR = false;
  double GlobalTilingTargetPSNR = 20.0, GlobalTilingQualityBasedTileCount = 7.0;
  int GlobalTilingTileCount = 0;
  int PaletteSize = 16, PaletteCount = 1024;
  int DitheringMode = TM_PVS_WEIGHTED_SPE_DCT;
  bool DitheringUseThomasKnoll = true;
  int DitheringYliluoma2MixedColors = 4;
  bool FrameTilingExtendedPaletteUsage = true;
  int MaxThreadCount = 1;
  double ShotTransMaxSecondsPerKF = 15.0, ShotTransMinSecondsPerKF = 1.0, ShotTransCorrelLoThres = 0.8;
};

}  // namespace tmx

using namespace tmx;

struct tm_encoder {
  Settings s;
  int device = 0;
  hipStream_t stream = nullptr;
  tm_progress_cb cb = nullptr;
  void *cb_user = nullptr;
  // video (ReframeUI, tilingencoder.pas:2631-2638)
  int width = 0, height = 0, tm_w = 0, tm_h = 0, nframes = 0;
  double fps = 24.0;
  bool auto_tile_count = true;
  // device state
  DevBuf frames_owned;
  const void *frames = nullptr;  // [nframes][height][width] RGB32
  const void *frames_host = nullptr;  // the same in HOST memory (tm_set_frames_host): Load copies it over in chunks beside its own kernel
  hipStream_t copy_stream = nullptr;
  // Clips that come from host memory land in one of two device buffers: the one the last Load read, and the one a prefetch
  // (tm_prefetch_frames_host) is filling for the next Load while this clip's later steps run.
  struct HostClip {
    DevBuf buf;
    const void *host = nullptr;       // the host clip it holds (or is being filled with)
    std::vector<hipEvent_t> events;   // one per chunk, recorded on the copy stream
    int chunk = 0, nchunks = 0;
    bool pending = false;             // filled (or being filled) by a prefetch that no Load has adopted yet
    uint64_t seq = 0;                 // order of the prefetches
  } hclip[2];
  int hclip_cur = -1;                 // the buffer `frames` points into, if any
  uint64_t hclip_seq = 0;
  // Load's inter-frame correlation is a chain of additions per frame (0.86 ms at 720p x 300) that nothing before the key frames' first
  // use waits for: it runs on a stream of its own beside Reduce, and its host tail (square roots, FindKeyFrames) is taken when somebody
  // asks (load_tail): a later step, a getter, the next Load
  hipStream_t stream_aux = nullptr;
  hipEvent_t ev_tiles = nullptr;
  DevBuf dcorrel;
  bool load_tail_pending = false;
  double kf_lo_thres = 0, kf_min_s = 0, kf_max_s = 0, kf_fps = 0;  // ShotTrans* and the frame rate at the time of that Load
  DevBuf ftiles, fflags, flab;   // frame tiles (canonical), mirror flags, Lab means
  DevBuf gtiles, gflags, guse, gpal_idx, gpal_px, palettes_dev;  // global tiles
  DevBuf tm_tile, tm_pal, tm_err;  // tile map, frame-major: TileIdx, PalIdx, error behind PSNR (KNN or motion)
  DevBuf pm_err, tm_px, tm_py, tm_pred;  // motion prediction: PredictMotion's best error, PredictedX/Y (int8), IsPredicted (uint8)
  bool has_pm = false;                   // PredictMotion ran with a radius > 0: Reduce and Reconstruct take their motion branches
  double reduce_threshold = 0;           // last PSNR threshold SolveTileCount evaluated
  int reduce_probes = 0;
  int64_t q = 0, t = 0;
  bool has_pal_px = false, reconstructed = false;
  bool gtiles_have_rgb = false;  // false after ReloadGTM until Reduce has run again
  // host state
  std::vector<float> correl;
  std::vector<int32_t> kf_start;
  std::vector<int32_t> palettes_host;
  std::vector<uint8_t> h_fflags;
  double stage_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int shard_first = 0, shard_count = -1;  // query frames this process matches in Reconstruct (multi-GPU: one shard per rank)
  DevBuf pair_keys;          // the distinct pixel keys PreparePalettes' quantisation sorted out, for Dither (valid while pair_keys_n > 0:
  int64_t pair_keys_n = 0;   // every step that rewrites the global tiles zeroes it)
  int64_t dither_pairs = 0;  // distinct (palette, colour) pairs the last Dither planned (0: every pixel on its own)
  int dither_rank = 0, dither_world = 1;  // tiles this process dithers: [t * rank / world, t * (rank + 1) / world)
  // one process per GPU (tm_set_collective): the steps shard their work over `world` processes and merge through the host's collectives
  tm_collective_cb coll_cb = nullptr;
  void *coll_user = nullptr;
  bool coll_stream_ordered = false;  // the callback enqueues on e->stream (tm_set_collective_mode): no drain before, no wait after
  Collectives co;
  bool load_sharded = false;     // Load only filled the frame tiles of this process's frames (and of the frame before them)
  int load_first = 0, load_count = 0;
  // the native communicator (tm_comm_init): RCCL linked into the library, the collectives queued on the encoder's stream
  int64_t coll_calls[4] = {0, 0, 0, 0}, coll_bytes = 0;  // per kind, and the bytes this process put through them (tm_get_collective_stats)
  void coll_count(int kind, int64_t count) {
    coll_calls[kind]++;
    coll_bytes += kind == TM_COLL_ALLGATHER_BYTES ? count * co.world : count * (kind == TM_COLL_ALLREDUCE_SUM_I64 ? 8 : 4);
  }
  ncclComm_t comm = nullptr;
  bool force_dist = false;  // a one-rank communicator walks the sharded paths too (TM_COMM_FORCE_DIST=1: tests on a one-GPU box)
  bool dist() const { return (coll_cb != nullptr || comm != nullptr) && (co.world > 1 || force_dist); }
  // Query features of Reconstruct's first chunk, computed AHEAD on a second (non-blocking) stream: they depend on the frame tiles only.
  // Launched when PreparePalettes hands over to the host (OptimizePalettes' 2-5 ms search, then Dither's start), the one stretch where
  // the GPU idles; launched earlier they only trade time with the k-means kernels (measured: +3.8 ms there for -3.7 ms here).
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_qf = nullptr;
  DevBuf qf_pre;
  DevBuf qf_colmm;  // the prefetched distinct rows' column ranges (the feature kernel keeps them; Reconstruct's search reads them)
  int qf_f0 = -1, qf_nf = 0, qf_epu = -1;
  bool qf_valid = false;
  // Reduce's exact grouping of the frame tiles (motion prediction off, one process): group of every tile-map item and the first item of
  // every group.  Items of one group have the same pixels, hence the same features and the same nearest database row: Reconstruct
  // searches once per GROUP (3.2 of 4.3 million on the bench clip) and hands the answer to the group's items.
  DevBuf q_group, q_rep;
  int64_t q_groups = 0;
  bool qf_distinct = false;  // the prefetched features are the groups' (not a frame range's)
  void drop_prefetch() {  // never frees under a running kernel
    if (stream2) (void)hipStreamSynchronize(stream2);
    qf_valid = false;
    qf_pre.release();
  }
  ~tm_encoder() {
    drop_prefetch();
    if (comm) { (void)hipStreamSynchronize(stream); (void)ncclCommAbort(comm); }  // (abort = destroy without the collective handshake: no peer is waited for)
    if (ev_qf) (void)hipEventDestroy(ev_qf);
    if (stream2) (void)hipStreamDestroy(stream2);
    if (stream_aux) { (void)hipStreamSynchronize(stream_aux); (void)hipStreamDestroy(stream_aux); }
    if (ev_tiles) (void)hipEventDestroy(ev_tiles);
    if (copy_stream) (void)hipStreamSynchronize(copy_stream);
    for (HostClip &c : hclip)
      for (hipEvent_t ev : c.events) (void)hipEventDestroy(ev);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
  }
  double knn_ms = 0;   // device time of the distance kernel, summed over launches of the last Reconstruct
  int64_t knn_pairs = 0;
  int knn_launches = 0, knn_kbytes = 0;
  double knn_split_ms[3] = {0, 0, 0};  // seeds / lists / consume kernels of those launches
  int64_t knn_split_pairs[3] = {0, 0, 0};
  KmeansRunStats km_stats;  // of the last PreparePalettes (single process: the sharded path runs its own loops)
  int64_t knn_db_rows = 0;  // distinct database rows actually searched
  int64_t knn_queries = 0;  // queries of the last Reconstruct's searches (distinct frame tiles when Reduce's groups are used)
  int steps_done = 0;  // bit per step

  int64_t tm_size() const { return (int64_t)tm_w * tm_h; }
};

static void progress(tm_encoder *e, int step, int pos, int max) {
  if (e->cb) e->cb(e->cb_user, step, pos, max, 0);
}

// ---- settings ------------------------------------------------------------------------------------------------
static int clampi(int64_t v, int lo, int hi) { return (int)std::min<int64_t>(hi, std::max<int64_t>(lo, v)); }

static void recompute_auto_tile_count(tm_encoder *e) {  // SetGlobalTilingQualityBasedTileCount, tilingencoder.pas:2937-2948
  const int64_t raw = (int64_t)e->nframes * e->tm_size();
  const int eqtc = equal_quality_tile_count((double)raw);
  e->s.GlobalTilingTileCount = (int)std::min<int64_t>(llrint(e->s.GlobalTilingQualityBasedTileCount * eqtc), raw);
}

static int set_number(tm_encoder *e, const std::string &k, double v, bool is_int_like) {
  Settings &s = e->s;
  (void)is_int_like;
  if (k == "StartFrame") s.StartFrame = std::max(0, (int)v);
  else if (k == "FrameCount") s.FrameCount = std::max(0, (int)v);
  else if (k == "Scaling") s.Scaling = std::max(0.01, v);
  else if (k == "MotionPredictRadius") s.MotionPredictRadius = clampi((int64_t)v, 0, 128);  // 3046 clamps to 1..128; 0 = motion prediction off (build extension: the reference code paths for <= 0 exist, 1972, 1496)
  else if (k == "GlobalTilingUseTargetPSNR") s.GlobalTilingUseTargetPSNR = v != 0;
  else if (k == "GlobalTilingTargetPSNR") s.GlobalTilingTargetPSNR = std::min(10 * std::log(255 * 255 / 0.5) / std::log(10.0), std::max(0.0, v));
  else if (k == "GlobalTilingQualityBasedTileCount") { s.GlobalTilingQualityBasedTileCount = v; e->auto_tile_count = true; if (e->nframes) recompute_auto_tile_count(e); }
  else if (k == "GlobalTilingTileCount") {  // SetGlobalTilingTileCount, 2974-2986: has priority over the quality-based value
    const int64_t raw = (int64_t)e->nframes * e->tm_size();
    s.GlobalTilingTileCount = raw ? clampi((int64_t)v, 0, (int)std::min<int64_t>(raw, INT32_MAX)) : std::max(0, (int)v);
    e->auto_tile_count = s.GlobalTilingTileCount <= 0;
  }
  else if (k == "PaletteSize") s.PaletteSize = clampi((int64_t)v, 2, 64);
  else if (k == "PaletteCount") s.PaletteCount = clampi((int64_t)v, 1, 65536);
  else if (k == "DitheringMode") s.DitheringMode = clampi((int64_t)v, 0, 4);
  else if (k == "DitheringUseThomasKnoll") s.DitheringUseThomasKnoll = v != 0;
  else if (k == "DitheringYliluoma2MixedColors") s.DitheringYliluoma2MixedColors = clampi((int64_t)v, 1, 16);
  else if (k == "FrameTilingExtendedPaletteUsage") s.FrameTilingExtendedPaletteUsage = v != 0;
  else if (k == "MaxThreadCount") s.MaxThreadCount = std::max(1, (int)v);
  else if (k == "ShotTransMaxSecondsPerKF") s.ShotTransMaxSecondsPerKF = std::max(0.0, v);
  else if (k == "ShotTransMinSecondsPerKF") s.ShotTransMinSecondsPerKF = std::max(0.0, v);
  else if (k == "ShotTransCorrelLoThres") s.ShotTransCorrelLoThres = std::min(1.0, std::max(-1.0, v));
  else { set_error("unknown setting '%s'", k.c_str()); return TM_E_INVAL; }
  return TM_OK;
}

static int get_number(tm_encoder *e, const std::string &k, double *v) {
  const Settings &s = e->s;
  if (k == "StartFrame") *v = s.StartFrame;
  else if (k == "FrameCount") *v = s.FrameCount;
  else if (k == "Scaling") *v = s.Scaling;
  else if (k == "MotionPredictRadius") *v = s.MotionPredictRadius;
  else if (k == "GlobalTilingUseTargetPSNR") *v = s.GlobalTilingUseTargetPSNR;
  else if (k == "GlobalTilingTargetPSNR") *v = s.GlobalTilingTargetPSNR;
  else if (k == "GlobalTilingQualityBasedTileCount") *v = s.GlobalTilingQualityBasedTileCount;
  else if (k == "GlobalTilingTileCount") *v = s.GlobalTilingTileCount;
  else if (k == "PaletteSize") *v = s.PaletteSize;
  else if (k == "PaletteCount") *v = s.PaletteCount;
  else if (k == "DitheringMode") *v = s.DitheringMode;
  else if (k == "DitheringUseThomasKnoll") *v = s.DitheringUseThomasKnoll;
  else if (k == "DitheringYliluoma2MixedColors") *v = s.DitheringYliluoma2MixedColors;
  else if (k == "FrameTilingExtendedPaletteUsage") *v = s.FrameTilingExtendedPaletteUsage;
  else if (k == "MaxThreadCount") *v = s.MaxThreadCount;
  else if (k == "ShotTransMaxSecondsPerKF") *v = s.ShotTransMaxSecondsPerKF;
  else if (k == "ShotTransMinSecondsPerKF") *v = s.ShotTransMinSecondsPerKF;
  else if (k == "ShotTransCorrelLoThres") *v = s.ShotTransCorrelLoThres;
  else { set_error("unknown setting '%s'", k.c_str()); return TM_E_INVAL; }
  return TM_OK;
}

// ---- steps ---------------------------------------------------------------------------------------------------
static int need(tm_encoder *e, int step_bit, const char *what) {
  TM_CHECK(e->steps_done & (1 << step_bit), TM_E_INVAL, "step order: %s has not been run", what);
  return TM_OK;
}

static int coll_run(tm_encoder *e, int kind, void *buf, void *recv, int64_t count) {
  if (!e->coll_stream_ordered) TM_HIP(hipStreamSynchronize(e->stream));  // everything queued so far is done before the host's collective touches the buffers
  e->coll_count(kind, count);
  const int rc = e->coll_cb(e->coll_user, kind, buf, recv, count);
  TM_CHECK(rc == 0, TM_E_HIP, "the host's collective callback failed (kind %d, code %d)", kind, rc);
  return TM_OK;
}
#define TM_NCCL(call)                                                                                         \
  do {                                                                                                        \
    const ncclResult_t r_ = (call);                                                                           \
    if (r_ != ncclSuccess) { set_error("%s failed: %s", #call, ncclGetErrorString(r_)); return TM_E_HIP; }  \
  } while (0)
// The library's communicator is non-blocking (tm_comm_init), so a call on it may answer ncclInProgress: the state is then polled until
// it settles, for at most TM_COMM_TIMEOUT_S seconds (default 120).
static ncclResult_t nccl_settle(ncclComm_t comm, ncclResult_t r) {
  if (r != ncclInProgress) return r;
  const double limit = knobs().comm_timeout_s;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    ncclResult_t st = ncclSuccess;
    const ncclResult_t q = ncclCommGetAsyncError(comm, &st);
    if (q != ncclSuccess) return q;
    if (st != ncclInProgress) return st;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return ncclSystemError;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}
#define TM_NCCLC(comm, call)                                                                                  \
  do {                                                                                                        \
    const ncclResult_t r_ = nccl_settle((comm), (call));                                                      \
    if (r_ != ncclSuccess) { set_error("%s failed: %s", #call, ncclGetErrorString(r_)); return TM_E_HIP; }  \
  } while (0)

// the four collective kinds on the encoder's stream through the library's own communicator: nothing drains the stream before and
// nothing waits after -- the RCCL kernel is ordered between what the step queued before and what it queues next
static void bind_native_collectives(tm_encoder *e) {
  e->co.allreduce_sum_i32 = [e](void *b, int64_t n) -> int {
    e->coll_count(TM_COLL_ALLREDUCE_SUM_I32, n);
    TM_NCCLC(e->comm, ncclAllReduce(b, b, (size_t)n, ncclInt32, ncclSum, e->comm, e->stream));
    return (int)TM_OK;
  };
  e->co.allreduce_max_i32 = [e](void *b, int64_t n) -> int {
    e->coll_count(TM_COLL_ALLREDUCE_MAX_I32, n);
    TM_NCCLC(e->comm, ncclAllReduce(b, b, (size_t)n, ncclInt32, ncclMax, e->comm, e->stream));
    return (int)TM_OK;
  };
  e->co.allreduce_sum_i64 = [e](void *b, int64_t n) -> int {
    e->coll_count(TM_COLL_ALLREDUCE_SUM_I64, n);
    TM_NCCLC(e->comm, ncclAllReduce(b, b, (size_t)n, ncclInt64, ncclSum, e->comm, e->stream));
    return (int)TM_OK;
  };
  e->co.allgather = [e](const void *snd, void *rcv, int64_t bytes) -> int {
    e->coll_count(TM_COLL_ALLGATHER_BYTES, bytes);
    TM_NCCLC(e->comm, ncclAllGather(snd, rcv, (size_t)bytes, ncclInt8, e->comm, e->stream));
    return (int)TM_OK;
  };
}

static void bind_collectives(tm_encoder *e) {
  e->co.allreduce_sum_i32 = [e](void *b, int64_t n) { return coll_run(e, TM_COLL_ALLREDUCE_SUM_I32, b, nullptr, n); };
  e->co.allreduce_max_i32 = [e](void *b, int64_t n) { return coll_run(e, TM_COLL_ALLREDUCE_MAX_I32, b, nullptr, n); };
  e->co.allreduce_sum_i64 = [e](void *b, int64_t n) { return coll_run(e, TM_COLL_ALLREDUCE_SUM_I64, b, nullptr, n); };
  e->co.allgather = [e](const void *snd, void *rcv, int64_t bytes) { return coll_run(e, TM_COLL_ALLGATHER_BYTES, const_cast<void *>(snd), rcv, bytes); };
}
// this process's share [lo, hi) of n items, contiguous, earlier processes take the remainder (the same rule as tiler_amd.distributed.frame_shard)
static void share_of(int64_t n, int rank, int world, int64_t *lo, int64_t *hi) {
  const int64_t base = n / world, rem = n % world;
  *lo = rank * base + std::min<int64_t>(rank, rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
}
// all-gather of per-process pieces of different sizes: send `count` items of `item` bytes, receive everyone's into `out` (in rank
// order, contiguous); counts[r] comes back on the host
static int gather_var(tm_encoder *e, const void *send, int64_t count, int item, DevBuf &out, std::vector<int64_t> *counts) {
  const int W = e->co.world;
  DevBuf dcnt, dall, pad, recv;
  TM_TRY(dcnt.alloc(8)); TM_TRY(dall.alloc((size_t)W * 8));
  TM_HIP(hipMemcpyAsync(dcnt.p, &count, 8, hipMemcpyHostToDevice, e->stream));
  TM_TRY(e->co.allgather(dcnt.p, dall.p, 8));
  counts->assign((size_t)W, 0);
  {
    HostRead hr_(e->stream);
    TM_TRY(hr_.get(counts->data(), dall.p, (size_t)W * 8));
    TM_TRY(hr_.wait());
  }
  int64_t mx = 0, total = 0;
  for (int64_t c : *counts) { mx = std::max(mx, c); total += c; }
  TM_TRY(out.alloc((size_t)std::max<int64_t>(total, 1) * item));
  if (mx == 0) return TM_OK;
  const size_t chunk = (((size_t)mx * item + 15) / 16) * 16;
  TM_TRY(pad.alloc(chunk)); TM_TRY(recv.alloc(chunk * W));
  if (count > 0) TM_HIP(hipMemcpyAsync(pad.p, send, (size_t)count * item, hipMemcpyDeviceToDevice, e->stream));
  TM_TRY(e->co.allgather(pad.p, recv.p, (int64_t)chunk));
  int64_t off = 0;
  for (int r = 0; r < W; r++) {
    if ((*counts)[r] > 0)
      TM_HIP(hipMemcpyAsync(out.as<uint8_t>() + (size_t)off * item, recv.as<uint8_t>() + chunk * r, (size_t)(*counts)[r] * item, hipMemcpyDeviceToDevice, e->stream));
    off += (*counts)[r];
  }
  TM_HIP(hipStreamSynchronize(e->stream));
  return TM_OK;
}

// Steps that read the frame tiles / the global tiles' RGB pixels: ReloadGTM brings neither (the stream holds palette indices
// only, HasRGBPixels = False at tilingencoder.pas:4937), so after a reload these steps need Load (and Reduce) to have run again.
static int need_frame_tiles(tm_encoder *e, const char *step) {
  TM_CHECK(e->ftiles.p != nullptr && e->fflags.p != nullptr && e->flab.p != nullptr && e->q > 0, TM_E_INVAL,
           "step order: %s needs the frame tiles, which are not in memory (run Load first; ReloadGTM does not bring them)", step);
  return TM_OK;
}
static int need_global_rgb(tm_encoder *e, const char *step) {
  TM_CHECK(e->gtiles_have_rgb && e->gtiles.p != nullptr, TM_E_INVAL,
           "step order: %s needs the global tiles' RGB pixels (run Reduce first; a reloaded .gtm holds palette indices only)", step);
  return TM_OK;
}

// the host tail of Load -- PearsonCorrelation's last lines (2221-2227) and FindKeyFrames (3373-3411) -- once the sums are there
static int load_tail(tm_encoder *e) {
  if (!e->load_tail_pending) return TM_OK;
  std::vector<float> sums((size_t)e->nframes * 3);
  hipStream_t st = e->stream_aux ? e->stream_aux : e->stream;
  {
    HostRead hr_(st);
    TM_TRY(hr_.get(sums.data(), e->dcorrel.p, sums.size() * 4));
    TM_TRY(hr_.wait());
  }
  e->load_tail_pending = false;  // only now: a failed read-back leaves the tail to the next caller instead of stale key frames
  e->correl.assign(e->nframes, 0.0f);
  for (int f = 1; f < e->nframes; f++) {  // tail of PearsonCorrelation (2221-2227) in host IEEE arithmetic
    const float denx = std::sqrt(sums[f * 3 + 1]), deny = std::sqrt(sums[f * 3 + 2]);
    const float den = denx * deny;
    e->correl[f] = den != 0.0f ? sums[f * 3] / den : 1.0f;
  }
  // FindKeyFrames, automatic mode (3373-3411)
  e->kf_start.clear();
  int64_t last = INT32_MIN;
  for (int f = 0; f < e->nframes; f++) {
    bool kf = f == 0;
    // (the settings as they stood when Load ran: the reference finds its key frames inside Load, 1741-1840)
    if (!kf && (double)e->correl[f] < e->kf_lo_thres) kf = true;
    if (!kf && (double)(f - last) >= e->kf_max_s * e->kf_fps) kf = true;
    if ((double)(f - last) < e->kf_min_s * e->kf_fps) kf = false;
    if (kf) { e->kf_start.push_back(f); last = f; }
  }
  return TM_OK;
}

// the chunked upload of a host clip into device buffer `slot`, queued on the copy stream with one event per chunk
static int queue_host_clip(tm_encoder *e, int slot, const void *host) {
  tm_encoder::HostClip &hc = e->hclip[slot];
  const size_t fbytes = (size_t)e->width * e->height * 4;
  TM_TRY(hc.buf.alloc(fbytes * e->nframes));
  if (!e->copy_stream) TM_HIP(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  constexpr size_t chunk_mb = 48;  // (4-48 MB measured alike)
  hc.chunk = (int)std::max<size_t>(1, (chunk_mb << 20) / fbytes);
  hc.nchunks = (e->nframes + hc.chunk - 1) / hc.chunk;
  while ((int)hc.events.size() < hc.nchunks) {
    hipEvent_t ev;
    TM_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hc.events.push_back(ev);
  }
  for (int c = 0; c < hc.nchunks; c++) {
    const int f0 = c * hc.chunk, nf = std::min(hc.chunk, e->nframes - f0);
    TM_HIP(hipMemcpyAsync(hc.buf.as<uint8_t>() + fbytes * f0, (const uint8_t *)host + fbytes * f0, fbytes * nf, hipMemcpyHostToDevice, e->copy_stream));
    TM_HIP(hipEventRecord(hc.events[c], e->copy_stream));
  }
  hc.host = host;
  hc.pending = true;
  hc.seq = ++e->hclip_seq;
  return TM_OK;
}

static int step_load(tm_encoder *e) {  // Load, tilingencoder.pas:1741-1841 (decode excluded: frames are pushed in)
  TM_TRY(load_tail(e));  // (a correlation still running reads the Lab means this Load is about to replace)
  e->drop_prefetch();  // features of the previous frame tiles
  e->q_groups = 0;
  e->load_sharded = false;
  TM_CHECK(e->nframes > 0 && e->width > 0, TM_E_INVAL, "tm_set_video has not been called");
  TM_CHECK(e->frames != nullptr || e->frames_host != nullptr, TM_E_INVAL, "no frames: call tm_push_frame_rgb32 / tm_set_frames_device / tm_set_frames_host first");
  e->q = (int64_t)e->nframes * e->tm_size();
  TM_CHECK(e->q < (1ll << 31), TM_E_UNSUPPORTED, "%lld tile-map items: the index arrays are 32-bit (TileIdx is an Integer, tilingencoder.pas:179)", (long long)e->q);
  TM_TRY(e->ftiles.alloc((size_t)e->q * 256));
  TM_TRY(e->fflags.alloc((size_t)e->q + 4));  // (+4: merged as 32-bit words)
  TM_TRY(e->flab.alloc((size_t)e->q * 12));
  if (e->frames_host) {
    // The clip sits in host memory: chunks of frames cross PCIe on a copy stream while the Load kernel works on the chunk before
    // (pinned memory makes the copies asynchronous; pageable memory still works, serialised by the runtime).  A clip that
    // tm_prefetch_frames_host already queued is adopted instead: its copies ran beside the previous clip's steps.
    const size_t fbytes = (size_t)e->width * e->height * 4;
    const int64_t per = e->tm_size();
    int slot = -1;
    for (int i = 0; i < 2; i++)
      if (e->hclip[i].pending && e->hclip[i].host == e->frames_host && (slot < 0 || e->hclip[i].seq < e->hclip[slot].seq)) slot = i;
    if (slot < 0) {
      // no prefetch of this clip: any buffer that holds no waiting clip will do (the last Load's own included: its clip is being
      // replaced); with two other clips waiting the older one is dropped (the copy stream orders the new copies behind its own)
      for (int i = 0; i < 2; i++)
        if (!e->hclip[i].pending && (slot < 0 || i != e->hclip_cur)) slot = i;
      if (slot < 0) slot = e->hclip[0].seq < e->hclip[1].seq ? 0 : 1;
      TM_HIP(hipStreamSynchronize(e->stream));       // the destination may have been handed out by the pool a moment ago
      TM_TRY(queue_host_clip(e, slot, e->frames_host));
    }
    tm_encoder::HostClip &hc = e->hclip[slot];
    // chunks that have already arrived (a prefetched clip: all of them, as a rule) go through ONE launch; the rest follow chunk by chunk
    int arrived = 0;
    while (arrived < hc.nchunks && hipEventQuery(hc.events[arrived]) == hipSuccess) arrived++;
    (void)hipGetLastError();  // (hipErrorNotReady of the first chunk still in flight is not an error)
    if (arrived > 0) {
      const int nf = std::min(arrived * hc.chunk, e->nframes);
      TM_HIP(hipStreamWaitEvent(e->stream, hc.events[arrived - 1], 0));
      TM_TRY(launch_load(hc.buf.p, nf, e->width, e->height, e->tm_w, e->tm_h, e->ftiles.p, e->fflags.p, e->flab.p, e->stream));
    }
    for (int c = arrived; c < hc.nchunks; c++) {
      const int f0 = c * hc.chunk, nf = std::min(hc.chunk, e->nframes - f0);
      TM_HIP(hipStreamWaitEvent(e->stream, hc.events[c], 0));
      TM_TRY(launch_load(hc.buf.as<uint8_t>() + fbytes * f0, nf, e->width, e->height, e->tm_w, e->tm_h, e->ftiles.as<uint8_t>() + (int64_t)f0 * per * 256,
                         e->fflags.as<uint8_t>() + (int64_t)f0 * per, e->flab.as<uint8_t>() + (int64_t)f0 * per * 12, e->stream));
    }
    // From here on the encoder reads its own device copy: the host clip is no longer borrowed once this Load has returned (it
    // synchronises below), and a later Run(esLoad) without new frames reads the copy again.
    hc.pending = false;
    e->hclip_cur = slot;
    e->frames = hc.buf.p;
    e->frames_host = nullptr;
  } else if (e->dist() && e->s.MotionPredictRadius <= 0) {
    // One process per GPU, motion prediction off: every process loads its own frames (frames are independent, 1293-1411) plus the
    // one before them, whose Lab means the first correlation needs.  The mirror flags (read back with every tile map) and the
    // correlation sums are merged; the frame tiles stay where they are -- Reduce and Reconstruct only need a process's own.
    const size_t fbytes = (size_t)e->width * e->height * 4;
    const int64_t per1 = e->tm_size();
    int64_t f0, f1;
    share_of(e->nframes, e->co.rank, e->co.world, &f0, &f1);
    const int64_t lo = std::max<int64_t>(f0 - 1, 0);
    TM_HIP(hipMemsetAsync(e->fflags.p, 0, (size_t)e->q, e->stream));
    if (f1 > f0)
      TM_TRY(launch_load((const uint8_t *)e->frames + fbytes * lo, (int)(f1 - lo), e->width, e->height, e->tm_w, e->tm_h, e->ftiles.as<uint8_t>() + lo * per1 * 256,
                         e->fflags.as<uint8_t>() + lo * per1, e->flab.as<uint8_t>() + lo * per1 * 12, e->stream));
    if (lo < f0) TM_HIP(hipMemsetAsync(e->fflags.as<uint8_t>() + lo * per1, 0, (size_t)per1, e->stream));  // the neighbour's flags are its owner's to report
    e->load_sharded = true; e->load_first = (int)f0; e->load_count = (int)(f1 - f0);
  } else
  TM_TRY(launch_load(e->frames, e->nframes, e->width, e->height, e->tm_w, e->tm_h, e->ftiles.p, e->fflags.p, e->flab.p, e->stream));
  progress(e, TM_STEP_LOAD, 1, 3);
  // inter-frame correlation: one GPU thread per frame runs the reference's sequential Single sums (order matters)
  const int per = (int)e->tm_size() * 3;
  DevBuf &dcorrel = e->dcorrel;
  TM_TRY(dcorrel.alloc((size_t)e->nframes * 12));
  e->h_fflags.clear();  // fetched lazily by tm_get_tilemap
  if (e->load_sharded) {
    const int64_t f0 = e->load_first, f1 = f0 + e->load_count, lo = std::max<int64_t>(f0 - 1, 0);
    TM_HIP(hipMemsetAsync(dcorrel.p, 0, (size_t)e->nframes * 12, e->stream));
    // block b of the launch correlates frame lo + b with the one before it (block 0 has none): frames f0 .. f1-1 (frame 0 has no sum)
    if (f1 > f0) TM_TRY(launch_pearson(e->flab.as<uint8_t>() + lo * per * 4, (int)(f1 - lo), per, dcorrel.as<uint8_t>() + lo * 12, e->stream));
    if (lo < f0) TM_HIP(hipMemsetAsync(dcorrel.as<uint8_t>() + lo * 12, 0, 12, e->stream));
    TM_TRY(e->co.allreduce_sum_i32(dcorrel.p, (int64_t)e->nframes * 3));  // owner holds the float, everyone else +0.0: exact
    TM_TRY(e->co.allreduce_sum_i32(e->fflags.p, (e->q + 3) / 4));
    TM_HIP(hipStreamSynchronize(e->stream));
    hipStream_t keep = e->stream_aux;
    e->stream_aux = nullptr;  // (the sums sit behind the encoder's own stream here)
    e->load_tail_pending = true;
    e->kf_lo_thres = e->s.ShotTransCorrelLoThres; e->kf_min_s = e->s.ShotTransMinSecondsPerKF; e->kf_max_s = e->s.ShotTransMaxSecondsPerKF; e->kf_fps = e->fps;
    const int rc = load_tail(e);
    e->stream_aux = keep;
    TM_TRY(rc);
  } else {
    if (!e->stream_aux) TM_HIP(hipStreamCreateWithFlags(&e->stream_aux, hipStreamNonBlocking));
    if (!e->ev_tiles) TM_HIP(hipEventCreateWithFlags(&e->ev_tiles, hipEventDisableTiming));
    TM_HIP(hipEventRecord(e->ev_tiles, e->stream));
    TM_HIP(hipStreamWaitEvent(e->stream_aux, e->ev_tiles, 0));
    TM_TRY(launch_pearson(e->flab.p, e->nframes, per, dcorrel.p, e->stream_aux));
    e->load_tail_pending = true;
    e->kf_lo_thres = e->s.ShotTransCorrelLoThres; e->kf_min_s = e->s.ShotTransMinSecondsPerKF; e->kf_max_s = e->s.ShotTransMaxSecondsPerKF; e->kf_fps = e->fps;
  }
  progress(e, TM_STEP_LOAD, 2, 3);
  if (e->auto_tile_count || e->s.GlobalTilingTileCount <= 0) recompute_auto_tile_count(e);
  // tile map starts empty (InitFrames, 2661-2686)
  TM_TRY(e->tm_tile.alloc((size_t)e->q * 4));
  TM_TRY(e->tm_pal.alloc((size_t)e->q * 4));
  TM_TRY(e->tm_err.alloc((size_t)e->q * 4));
  TM_HIP(hipMemsetAsync(e->tm_tile.p, 0xff, (size_t)e->q * 4, e->stream));
  TM_HIP(hipMemsetAsync(e->tm_pal.p, 0xff, (size_t)e->q * 4, e->stream));
  TM_HIP(hipMemsetAsync(e->tm_err.p, 0xff, (size_t)e->q * 4, e->stream));
  e->t = 0;
  e->has_pal_px = e->reconstructed = e->has_pm = false;
  TM_HIP(hipStreamSynchronize(e->stream));  // Run(esLoad) is blocking for everything but the correlation above (and the stage times stay the stages')
  progress(e, TM_STEP_LOAD, 3, 3);
  return TM_OK;
}

static int step_predict_motion(tm_encoder *e) {
  // PredictMotion, tilingencoder.pas:1964-1991: frame 0 is searched in frame 1, frame f >= 1 in the SOURCE pixels of
  // frame f-1 (the front buffer is drawn from the un-mirrored frame tiles, 1255-1260), so frames are independent.
  TM_TRY(need(e, TM_STEP_LOAD, "Load"));
  e->has_pm = false;
  if (e->s.MotionPredictRadius <= 0) return TM_OK;  // 1972
  TM_TRY(need_frame_tiles(e, "PredictMotion"));
  TM_CHECK(!e->load_sharded, TM_E_INVAL, "PredictMotion: Load ran with motion prediction off and only brought this process's frames; run Load again");
  const int64_t per = e->tm_size();
  const int sw = e->tm_w * 8, sh = e->tm_h * 8;
  const int64_t nwin = (int64_t)(sw - 7) * (sh - 7);
  TM_TRY(e->pm_err.alloc((size_t)e->q * 4));
  TM_TRY(e->tm_px.alloc((size_t)e->q + 4));  // (+4: merged as 32-bit words)
  TM_TRY(e->tm_py.alloc((size_t)e->q + 4));
  TM_TRY(e->tm_pred.alloc((size_t)e->q + 4));
  TM_HIP(hipMemsetAsync(e->tm_pred.p, 0, (size_t)e->q, e->stream));
  const int sf = std::max(0, std::min(e->shard_first, e->nframes));
  const int sn = e->shard_count < 0 ? e->nframes - sf : std::max(0, std::min(e->shard_count, e->nframes - sf));
  if (sf > 0 || sn < e->nframes) {  // frames of other shards stay 0: the host merges shards with all-reduce(SUM)
    TM_HIP(hipMemsetAsync(e->pm_err.p, 0, (size_t)e->q * 4, e->stream));
    TM_HIP(hipMemsetAsync(e->tm_px.p, 0, (size_t)e->q, e->stream));
    TM_HIP(hipMemsetAsync(e->tm_py.p, 0, (size_t)e->q, e->stream));
  }
  DevBuf screen, win, cur;
  TM_TRY(screen.alloc((size_t)sw * sh * 4));
  TM_TRY(win.alloc((size_t)nwin * 384));
  TM_TRY(cur.alloc((size_t)per * 384));
  for (int f = sf; f < sf + sn; f++) {
    const int src = f >= 1 ? f - 1 : (e->nframes > 1 ? 1 : -1);
    if (src >= 0) TM_TRY(launch_tiles_to_screen(e->ftiles.as<uint8_t>() + (int64_t)src * per * 256, e->fflags.as<uint8_t>() + (int64_t)src * per, e->tm_w, e->tm_h, screen.p, e->stream));
    else TM_HIP(hipMemsetAsync(screen.p, 0, (size_t)sw * sh * 4, e->stream));  // a single frame is searched in a black buffer
    const int64_t off = (int64_t)f * per;
    TM_TRY(launch_features_rgb(e->ftiles.as<uint8_t>() + off * 256, per, e->fflags.as<uint8_t>() + off, TM_PVS_WEIGHTED_DCT, 0, cur.p, e->stream));
    TM_TRY(launch_motion_search_fb(cur.p, e->tm_w, e->tm_h, screen.p, win.p, e->s.MotionPredictRadius, e->pm_err.as<uint32_t>() + off,
                                   e->tm_px.as<int8_t>() + off, e->tm_py.as<int8_t>() + off, e->stream));
    if ((f & 15) == 15) progress(e, TM_STEP_PREDICT_MOTION, f, e->nframes);
  }
  if (e->dist()) {  // owner holds the value, everyone else 0
    TM_TRY(e->co.allreduce_sum_i32(e->pm_err.p, e->q));
    TM_TRY(e->co.allreduce_sum_i32(e->tm_px.p, (e->q + 3) / 4));
    TM_TRY(e->co.allreduce_sum_i32(e->tm_py.p, (e->q + 3) / 4));
  }
  TM_HIP(hipStreamSynchronize(e->stream));
  e->has_pm = true;
  e->reconstructed = false;
  progress(e, TM_STEP_PREDICT_MOTION, e->nframes, e->nframes);
  return TM_OK;
}

static int step_reduce_motion(tm_encoder *e) {
  TM_TRY(load_tail(e));
  // Reduce with motion prediction (1909-1926): SolveTileCount searches the PSNR threshold above which a tile-map item
  // stays predicted (4014-4046); the items below it are transferred (4048-4103), made unique and ordered (4038, 1923).
  // The search runs on per-group maxima of the prediction error (a group = one distinct tile content): PSNR is a
  // non-increasing function of the error, so "some member has PSNR <= x" is "the group's largest error exceeds the
  // largest error still predicted at x".  The state kept is the last probe's, as in the reference.
  const int64_t per = e->tm_size();
  DevBuf remap, order, use, kfmask, keep, sel, pos;
  TM_TRY(remap.alloc((size_t)e->q * 4)); TM_TRY(order.alloc((size_t)e->q * 4)); TM_TRY(use.alloc((size_t)e->q * 4));
  int64_t ngroups = 0;
  TM_TRY(run_dedup(e->ftiles.p, e->q, 256, nullptr, remap.p, order.p, use.p, &ngroups, e->stream));
  std::vector<uint8_t> hk((size_t)e->nframes, 0);
  for (int32_t k : e->kf_start) hk[(size_t)k] = 1;
  TM_TRY(kfmask.alloc(hk.size()));
  TM_HIP(hipMemcpyAsync(kfmask.p, hk.data(), hk.size(), hipMemcpyHostToDevice, e->stream));
  TM_TRY(keep.alloc((size_t)e->q * 4)); TM_TRY(sel.alloc((size_t)e->q * 4)); TM_TRY(pos.alloc((size_t)e->q * 4));
  const double target = e->s.GlobalTilingTileCount > 0 ? (double)e->s.GlobalTilingTileCount : (double)ngroups;
  TM_TRY(solve_tile_count(remap.p, ngroups, e->pm_err.p, kfmask.p, (int)per, e->q, target, e->tm_pred.p, keep.p, &e->reduce_threshold,
                          &e->reduce_probes, e->stream));
  progress(e, TM_STEP_REDUCE, 1, 2);
  int64_t nkeep = 0;
  TM_TRY(compact_kept(keep.p, e->q, sel.p, pos.p, &nkeep, e->stream));
  TM_CHECK(nkeep > 0, TM_E_INVAL, "Reduce: every tile is predicted, no global tile left");
  DevBuf sub, sremap, sorder, suse;
  TM_TRY(sub.alloc((size_t)nkeep * 256)); TM_TRY(sremap.alloc((size_t)nkeep * 4)); TM_TRY(sorder.alloc((size_t)nkeep * 4)); TM_TRY(suse.alloc((size_t)nkeep * 4));
  hipLaunchKernelGGL(k_gather_rows16, dim3(gridn(nkeep * 16)), dim3(256), 0, e->stream, e->ftiles.as<uint4>(), sel.as<int32_t>(), nkeep, 16, sub.as<uint4>());
  int64_t nu = 0;
  TM_TRY(run_dedup(sub.p, nkeep, 256, nullptr, sremap.p, sorder.p, suse.p, &nu, e->stream));
  e->t = nu;
  e->pair_keys_n = 0;
  TM_TRY(e->gtiles.alloc((size_t)e->t * 256));
  TM_TRY(e->gflags.alloc((size_t)std::max<int64_t>(e->t, 1)));
  TM_TRY(e->guse.alloc((size_t)e->t * 4));
  DevBuf gsrc;  // global tile -> frame tile index
  TM_TRY(gsrc.alloc((size_t)e->t * 4));
  hipLaunchKernelGGL(k_gather<int32_t>, dim3(gridn(e->t)), dim3(256), 0, e->stream, sel.as<int32_t>(), sorder.as<int32_t>(), e->t, gsrc.as<int32_t>());
  hipLaunchKernelGGL(k_gather_rows16, dim3(gridn(e->t * 16)), dim3(256), 0, e->stream, e->ftiles.as<uint4>(), gsrc.as<int32_t>(), e->t, 16,
                     e->gtiles.as<uint4>());
  hipLaunchKernelGGL(k_gather<uint8_t>, dim3(gridn(e->t)), dim3(256), 0, e->stream, e->fflags.as<uint8_t>(), gsrc.as<int32_t>(), e->t,
                     e->gflags.as<uint8_t>());
  TM_HIP(hipMemcpyAsync(e->guse.p, suse.p, (size_t)e->t * 4, hipMemcpyDeviceToDevice, e->stream));
  hipLaunchKernelGGL(k_tilemap_from_subset, dim3(gridn(e->q)), dim3(256), 0, e->stream, keep.as<int32_t>(), pos.as<int32_t>(), sremap.as<int32_t>(),
                     e->q, e->tm_tile.as<int32_t>());
  TM_HIP(hipGetLastError());
  TM_HIP(hipStreamSynchronize(e->stream));
  e->has_pal_px = e->reconstructed = false;
  progress(e, TM_STEP_REDUCE, 2, 2);
  return TM_OK;
}

static int step_reduce(tm_encoder *e) {
  // Reduce, tilingencoder.pas:1909-1926 = SolveTileCount (4043) + ReindexTiles(True).  After PredictMotion the threshold
  // search of step_reduce_motion runs.  With motion prediction switched off (MotionPredictRadius = 0, the benchmark's headline
  // configuration) no tile-map item is predicted, so TransferTiles (4048) moves every frame tile; MakeTilesUnique(True) +
  // ReindexTiles(True) are exact; the tile budget is then met by keeping the first GlobalTilingTileCount tiles of
  // that order (most used first), see DESIGN.md "Scope".
  TM_TRY(need(e, TM_STEP_LOAD, "Load"));
  TM_TRY(need_frame_tiles(e, "Reduce"));
  e->gtiles_have_rgb = true;
  e->q_groups = 0;
  e->drop_prefetch();
  if (e->has_pm) return step_reduce_motion(e);
  if (e->load_sharded) {
    // One process per GPU: exact dedup of this process's own frame tiles first, then of the union of every process's distinct
    // tiles (all-gathered: tile, use count, mirror flags of its first occurrence).  Processes own increasing frame ranges and the
    // union is laid out in process order, so "first occurrence" and the final order (use count descending, content ascending)
    // are those of the single-process run.
    const int64_t per = e->tm_size(), f0 = e->load_first, nloc = (int64_t)e->load_count * per;
    DevBuf lremap, lorder, luse, rec, urec, utiles, uuse, uflags, gremap, gorder, guse2;
    int64_t lnu = 0;
    TM_TRY(lremap.alloc((size_t)std::max<int64_t>(nloc, 1) * 4)); TM_TRY(lorder.alloc((size_t)std::max<int64_t>(nloc, 1) * 4)); TM_TRY(luse.alloc((size_t)std::max<int64_t>(nloc, 1) * 4));
    if (nloc > 0) TM_TRY(run_dedup(e->ftiles.as<uint8_t>() + f0 * per * 256, nloc, 256, nullptr, lremap.p, lorder.p, luse.p, &lnu, e->stream));
    // What travels: only the tiles that can be among the first GlobalTilingTileCount of the merged order, chosen on 16-byte keys every
    // process exchanges first (tm_dedup.hip, "Reduce over several processes"; gathering every distinct tile of every process, as the
    // first two rounds did, moved 857 MB on the bench clip).
    const int64_t budget = e->s.GlobalTilingTileCount > 0 ? (int64_t)e->s.GlobalTilingTileCount : 0;  // 0: no budget, everything stays
    DevBuf lkeys, allkeys, in_s, sel, spos, sidx, suse;
    int64_t nsel = lnu, key_off = 0;
    {
      TM_TRY(lkeys.alloc((size_t)std::max<int64_t>(lnu, 1) * 16));
      TM_TRY(reduce_make_keys(e->ftiles.as<uint8_t>() + f0 * per * 256, lorder.p, luse.p, lnu, 256, lkeys.p, e->stream));
      std::vector<int64_t> kcounts;
      TM_TRY(gather_var(e, lkeys.p, lnu, 16, allkeys, &kcounts));
      int64_t ntot = 0;
      for (int r = 0; r < e->co.world; r++) { if (r < e->co.rank) key_off += kcounts[r]; ntot += kcounts[r]; }
      TM_CHECK(ntot > 0 && ntot < (1ll << 31), TM_E_INVAL, "Reduce: %lld distinct tiles over all processes", (long long)ntot);
      TM_TRY(in_s.alloc((size_t)ntot * 4));
      TM_TRY(reduce_select_candidates(allkeys.p, ntot, budget, in_s.p, e->stream));
      TM_TRY(sel.alloc((size_t)std::max<int64_t>(lnu, 1) * 4)); TM_TRY(spos.alloc((size_t)std::max<int64_t>(lnu, 1) * 4));
      nsel = 0;
      if (lnu > 0) TM_TRY(compact_kept(in_s.as<uint32_t>() + key_off, lnu, sel.p, spos.p, &nsel, e->stream));
      TM_TRY(sidx.alloc((size_t)std::max<int64_t>(nsel, 1) * 4)); TM_TRY(suse.alloc((size_t)std::max<int64_t>(nsel, 1) * 4));
      if (nsel > 0) {
        hipLaunchKernelGGL(k_gather<int32_t>, dim3(gridn(nsel)), dim3(256), 0, e->stream, lorder.as<int32_t>(), sel.as<int32_t>(), nsel, sidx.as<int32_t>());
        hipLaunchKernelGGL(k_gather<uint32_t>, dim3(gridn(nsel)), dim3(256), 0, e->stream, luse.as<uint32_t>(), sel.as<int32_t>(), nsel, suse.as<uint32_t>());
      }
    }
    TM_TRY(rec.alloc((size_t)std::max<int64_t>(nsel, 1) * 264));
    if (nsel > 0)
      hipLaunchKernelGGL(k_pack_unique, dim3(gridn(nsel * 66)), dim3(256), 0, e->stream, e->ftiles.as<uint32_t>() + f0 * per * 64, e->fflags.as<uint8_t>() + f0 * per,
                         sidx.as<int32_t>(), suse.as<uint32_t>(), nsel, rec.as<uint32_t>());
    TM_HIP(hipGetLastError());
    std::vector<int64_t> counts;
    TM_TRY(gather_var(e, rec.p, nsel, 264, urec, &counts));
    int64_t nun = 0, my_off = 0;
    for (int r = 0; r < e->co.world; r++) { if (r < e->co.rank) my_off += counts[r]; nun += counts[r]; }
    TM_CHECK(nun > 0 && nun < (1ll << 31), TM_E_INVAL, "Reduce: %lld distinct tiles over all processes", (long long)nun);
    TM_TRY(utiles.alloc((size_t)nun * 256)); TM_TRY(uuse.alloc((size_t)nun * 4)); TM_TRY(uflags.alloc((size_t)nun));
    hipLaunchKernelGGL(k_unpack_unique, dim3(gridn(nun * 66)), dim3(256), 0, e->stream, urec.as<uint32_t>(), nun, utiles.as<uint32_t>(), uuse.as<uint32_t>(), uflags.as<uint8_t>());
    TM_HIP(hipGetLastError());
    TM_TRY(gremap.alloc((size_t)nun * 4)); TM_TRY(gorder.alloc((size_t)nun * 4)); TM_TRY(guse2.alloc((size_t)nun * 4));
    int64_t nu = 0;
    TM_TRY(run_dedup(utiles.p, nun, 256, uuse.p, gremap.p, gorder.p, guse2.p, &nu, e->stream));
    progress(e, TM_STEP_REDUCE, 1, 2);
    const int64_t target = e->s.GlobalTilingTileCount > 0 ? e->s.GlobalTilingTileCount : nu;
    e->t = std::min<int64_t>(nu, target);
    e->pair_keys_n = 0;
  TM_TRY(e->gtiles.alloc((size_t)e->t * 256));
    TM_TRY(e->gflags.alloc((size_t)std::max<int64_t>(e->t, 1)));
    TM_TRY(e->guse.alloc((size_t)e->t * 4));
    hipLaunchKernelGGL(k_gather_rows16, dim3(gridn(e->t * 16)), dim3(256), 0, e->stream, utiles.as<uint4>(), gorder.as<int32_t>(), e->t, 16, e->gtiles.as<uint4>());
    hipLaunchKernelGGL(k_gather<uint8_t>, dim3(gridn(e->t)), dim3(256), 0, e->stream, uflags.as<uint8_t>(), gorder.as<int32_t>(), e->t, e->gflags.as<uint8_t>());
    TM_HIP(hipMemcpyAsync(e->guse.p, guse2.p, (size_t)e->t * 4, hipMemcpyDeviceToDevice, e->stream));
    // tile map of this process's frames (TransferTiles: TileIdx := the tile's index, 4079-4083); the other frames' items are their owners'
    TM_HIP(hipMemsetAsync(e->tm_tile.p, 0xff, (size_t)e->q * 4, e->stream));
    if (nloc > 0)
      hipLaunchKernelGGL(k_compose_remap_cand, dim3(gridn(nloc)), dim3(256), 0, e->stream, lremap.as<int32_t>(), nloc, in_s.as<uint32_t>() + key_off, spos.as<int32_t>(),
                         gremap.as<int32_t>(), (int32_t)my_off, (int32_t)e->t, e->tm_tile.as<int32_t>() + f0 * per);
    TM_HIP(hipGetLastError());
    TM_HIP(hipStreamSynchronize(e->stream));
    e->has_pal_px = e->reconstructed = false;
    progress(e, TM_STEP_REDUCE, 2, 2);
    return TM_OK;
  }
  DevBuf remap, order, use;
  TM_TRY(remap.alloc((size_t)e->q * 4));
  TM_TRY(order.alloc((size_t)e->q * 4));
  TM_TRY(use.alloc((size_t)e->q * 4));
  int64_t nu = 0;
  // (only the first GlobalTilingTileCount tiles of the order stay: the rows behind them are counted and numbered, not ordered)
  TM_TRY(run_dedup(e->ftiles.p, e->q, 256, nullptr, remap.p, order.p, use.p, &nu, e->stream, e->s.GlobalTilingTileCount > 0 ? (int64_t)e->s.GlobalTilingTileCount : 0));
  progress(e, TM_STEP_REDUCE, 1, 2);
  int64_t target = e->s.GlobalTilingTileCount > 0 ? e->s.GlobalTilingTileCount : nu;
  e->t = std::min<int64_t>(nu, target);
  e->pair_keys_n = 0;
  TM_TRY(e->gtiles.alloc((size_t)e->t * 256));
  TM_TRY(e->gflags.alloc((size_t)std::max<int64_t>(e->t, 1)));
  TM_TRY(e->guse.alloc((size_t)e->t * 4));
  hipLaunchKernelGGL(k_gather_rows16, dim3(gridn(e->t * 16)), dim3(256), 0, e->stream, e->ftiles.as<uint4>(), order.as<int32_t>(), e->t, 16,
                     e->gtiles.as<uint4>());
  hipLaunchKernelGGL(k_gather<uint8_t>, dim3(gridn(e->t)), dim3(256), 0, e->stream, e->fflags.as<uint8_t>(), order.as<int32_t>(), e->t,
                     e->gflags.as<uint8_t>());
  TM_HIP(hipMemcpyAsync(e->guse.p, use.p, (size_t)e->t * 4, hipMemcpyDeviceToDevice, e->stream));
  TM_HIP(hipMemcpyAsync(e->tm_tile.p, remap.p, (size_t)e->q * 4, hipMemcpyDeviceToDevice, e->stream));
  hipLaunchKernelGGL(k_clip_index, dim3(gridn(e->q)), dim3(256), 0, e->stream, e->tm_tile.as<int32_t>(), e->q, (int32_t)e->t);
  TM_HIP(hipGetLastError());
  TM_HIP(hipStreamSynchronize(e->stream));
  if (!knobs().no_query_groups) {  // kept for Reconstruct: one search per distinct frame tile
    e->q_group = std::move(remap);
    e->q_rep = std::move(order);
    e->q_groups = nu;
  }
  e->has_pal_px = e->reconstructed = false;
  progress(e, TM_STEP_REDUCE, 2, 2);
  return TM_OK;
}

// frames per chunk of Reconstruct's query features (bounded scratch for long / 4K clips: streaming through HBM)
static int recon_chunk_frames(const tm_encoder *e, int sn, bool epu) {
  const int64_t per = e->tm_size(), budget = epu ? ((int64_t)2 << 30) : ((int64_t)8 << 30);
  return (int)std::max<int64_t>(1, std::min<int64_t>(std::max(sn, 1), budget / (per * 384)));
}

// may Reconstruct search once per distinct frame tile?  (the k = 1 search of the whole clip in one process, rows within one chunk)
static bool query_groups_usable(const tm_encoder *e, int sf, int sn, bool epu) {
  // (the extended-palette search keeps 64 candidates per query: 512 more bytes a row)
  return e->q_groups > 0 && !e->dist() && sf == 0 && sn == e->nframes && e->q_groups * (epu ? 384 + 512 : 384) <= ((int64_t)8 << 30);
}

static int prefetch_query_features(tm_encoder *e) {
  const int sf = std::max(0, std::min(e->shard_first, e->nframes));
  const int sn = e->shard_count < 0 ? e->nframes - sf : std::max(0, std::min(e->shard_count, e->nframes - sf));
  if (sn <= 0) return TM_OK;
  const bool epu = e->s.FrameTilingExtendedPaletteUsage;
  const int nf = std::min(recon_chunk_frames(e, sn, epu), sn);
  const int64_t per = e->tm_size();
  e->drop_prefetch();
  const bool distinct = query_groups_usable(e, sf, sn, epu);
  if (!e->stream2) {  // lowest priority: the small dependent kernels of PreparePalettes must not queue behind this one's workgroups
    int lo = 0, hi = 0;
    TM_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    TM_HIP(hipStreamCreateWithPriority(&e->stream2, hipStreamNonBlocking, lo));
  }
  if (!e->ev_qf) TM_HIP(hipEventCreateWithFlags(&e->ev_qf, hipEventDisableTiming));
  TM_TRY(e->qf_pre.alloc((size_t)(distinct ? e->q_groups : (int64_t)nf * per) * 384));
  TM_HIP(hipStreamSynchronize(e->stream));  // the pool handed out memory that work on the main stream may just have released
  if (distinct) {
    TM_TRY(e->qf_colmm.alloc(384 * 4));
    TM_HIP(hipMemsetAsync(e->qf_colmm.p, 0x7f, 192 * 4, e->stream2));                           // 0x7f7f7f7f: above any int16
    TM_HIP(hipMemsetAsync(e->qf_colmm.as<uint8_t>() + 192 * 4, 0x80, 192 * 4, e->stream2));     // 0x80808080: below any int16
    TM_TRY(launch_features_rgb_rows(e->ftiles.p, e->q_rep.p, e->q_groups, TM_PVS_WEIGHTED_DCT, 0, e->qf_pre.p, e->stream2, e->qf_colmm.p));
  }
  else
    TM_TRY(launch_features_rgb(e->ftiles.as<uint8_t>() + (int64_t)sf * per * 256, (int64_t)nf * per, nullptr, TM_PVS_WEIGHTED_DCT, 0, e->qf_pre.p, e->stream2));
  TM_HIP(hipEventRecord(e->ev_qf, e->stream2));
  e->qf_f0 = sf; e->qf_nf = nf; e->qf_epu = epu ? 1 : 0;
  e->qf_distinct = distinct;
  e->qf_valid = true;
  return TM_OK;
}

// the chunk [f0, f0 + nf) of query features: the prefetched buffer when it is that chunk (the main stream then waits for it), else computed now
static int query_features(tm_encoder *e, int f0, int nf, bool epu, DevBuf &qf, void **out) {
  const int64_t per = e->tm_size();
  if (e->qf_valid && !e->qf_distinct && e->qf_f0 == f0 && e->qf_nf == nf && e->qf_epu == (epu ? 1 : 0)) {
    TM_HIP(hipStreamWaitEvent(e->stream, e->ev_qf, 0));
    *out = e->qf_pre.p;
    return TM_OK;
  }
  TM_TRY(qf.alloc((size_t)nf * per * 384));
  *out = qf.p;
  return launch_features_rgb(e->ftiles.as<uint8_t>() + (int64_t)f0 * per * 256, (int64_t)nf * per, nullptr, TM_PVS_WEIGHTED_DCT, 0, qf.p, e->stream);
}

static int step_prepare_palettes(tm_encoder *e) {  // PreparePalettes, tilingencoder.pas:1843-1871
  TM_TRY(need(e, TM_STEP_REDUCE, "Reduce"));
  TM_TRY(need_global_rgb(e, "PreparePalettes"));
  TM_CHECK(e->t > 0, TM_E_INVAL, "no global tiles");
  const bool dbg = knobs().pp_debug;  // wall time of the sub-steps (adds stream synchronisations)
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!dbg) return;
    (void)hipStreamSynchronize(e->stream);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[tm_pp] %-28s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  DevBuf feat;
  e->pair_keys_n = 0;
  TM_TRY(e->gpal_idx.alloc((size_t)e->t * 4));
  TM_TRY(e->palettes_dev.alloc((size_t)e->s.PaletteCount * e->s.PaletteSize * 4));
  if (e->dist()) {
    // One process per GPU.  Tile -> palette: every process holds the clustering features of its own share of the global tiles; the
    // farthest-first picks are settled by an all-gather of one candidate per process and the Lloyd iterations by an all-reduce
    // of the exact integer sums (run_palettize_dist), then the palette indices of all shares are all-gathered.  Palette colours:
    // the palettes are independent tasks (one thread per palette in the reference, 1864): process r quantises the palettes
    // p = r (mod world), an all-reduce(SUM) assembles the set.
    if (!knobs().pp_sharded && palettize_resident(e->t, e->s.PaletteCount)) {
      // Up to 16 palettes and half a million tiles the single-GPU clustering is ONE resident launch of a few milliseconds (k_h_resident), and every
      // process holds all the global tiles: each runs it whole.  Sharded, a Lloyd iteration is an all-reduce of 25 KB -- 300 latency-bound
      // collectives on the bench clip, more than the whole clustering takes here -- plus two all-gathers per seeding pick; replicated there is none.
      // Integer sums and a fixed seed: every process ends with the same palettes.  (TM_PP_SHARDED=1 keeps the data-parallel form for A/B and tests.)
      TM_TRY(feat.alloc((size_t)e->t * 192 * 4));
      TM_TRY(launch_features_cluster(e->gtiles.p, e->t, e->s.DitheringMode, feat.p, e->stream));
      lap("cluster features (all tiles, every process)");
      TM_TRY(run_palettize(feat.p, e->guse.p, e->t, e->s.PaletteCount, 300, e->gpal_idx.p, e->stream));
      lap("tile -> palette (192-D, replicated)");
    } else {
    int64_t t0, t1;
    share_of(e->t, e->co.rank, e->co.world, &t0, &t1);
    const int64_t nl = t1 - t0;
    TM_TRY(feat.alloc((size_t)std::max<int64_t>(nl, 1) * 192 * 4));
    if (nl > 0) TM_TRY(launch_features_cluster(e->gtiles.as<uint8_t>() + t0 * 256, nl, e->s.DitheringMode, feat.p, e->stream));
    lap("cluster features (own share)");
    DevBuf lidx, all;
    TM_TRY(lidx.alloc((size_t)std::max<int64_t>(nl, 1) * 4));
    TM_TRY(run_palettize_dist(feat.p, e->guse.as<uint8_t>() + t0 * 4, nl, t0, e->s.PaletteCount, 300, lidx.p, e->co, e->stream));
    std::vector<int64_t> counts;
    TM_TRY(gather_var(e, lidx.p, nl, 4, all, &counts));
    TM_HIP(hipMemcpyAsync(e->gpal_idx.p, all.p, (size_t)e->t * 4, hipMemcpyDeviceToDevice, e->stream));
    lap("tile -> palette (192-D, data-parallel)");
    }
    progress(e, TM_STEP_PREPARE_PALETTES, 1, 3);
    TM_TRY(run_quantize_palettes_part(e->gtiles.p, e->gpal_idx.p, e->t, e->s.PaletteCount, e->s.PaletteSize, 300, e->palettes_dev.p, e->co.rank, e->co.world, e->stream));
    TM_TRY(e->co.allreduce_sum_i32(e->palettes_dev.p, (int64_t)e->s.PaletteCount * e->s.PaletteSize));
    lap("palette colours (3-D, own palettes)");
  } else {
  TM_TRY(feat.alloc((size_t)e->t * 192 * 4));
  TM_TRY(launch_features_cluster(e->gtiles.p, e->t, e->s.DitheringMode, feat.p, e->stream));
  lap("cluster features");
  TM_TRY(run_palettize(feat.p, e->guse.p, e->t, e->s.PaletteCount, 300, e->gpal_idx.p, e->stream));
  lap("tile -> palette (192-D)");
  progress(e, TM_STEP_PREPARE_PALETTES, 1, 3);
  TM_TRY(run_quantize_palettes(e->gtiles.p, e->gpal_idx.p, e->t, e->s.PaletteCount, e->s.PaletteSize, 300, e->palettes_dev.p, e->stream, &e->pair_keys, &e->pair_keys_n));
  e->km_stats = kmeans_run_stats();
  lap("palette colours (3-D)");
  }
  e->palettes_host.resize((size_t)e->s.PaletteCount * e->s.PaletteSize);
  {
    HostRead hr_(e->stream);
    TM_TRY(hr_.get(e->palettes_host.data(), e->palettes_dev.p, e->palettes_host.size() * 4));
    TM_TRY(hr_.wait());
  }
  progress(e, TM_STEP_PREPARE_PALETTES, 2, 3);
  TM_TRY(prefetch_query_features(e));  // the GPU has nothing to do while the host searches: Reconstruct's query features run now
  lap("prefetch launch");
  // OptimizePalettes (4309-4432): slot permutation by Powell on the host (P x PaletteSize colours)
  TM_TRY(optimize_palettes_host(e->palettes_host, e->s.PaletteCount, e->s.PaletteSize, nullptr));
  lap("OptimizePalettes (host)");
  TM_HIP(hipMemcpyAsync(e->palettes_dev.p, e->palettes_host.data(), e->palettes_host.size() * 4, hipMemcpyHostToDevice, e->stream));
  TM_HIP(hipStreamSynchronize(e->stream));
  progress(e, TM_STEP_PREPARE_PALETTES, 3, 3);
  return TM_OK;
}

static int step_dither(tm_encoder *e) {  // Dither, tilingencoder.pas:1873-1907
  TM_TRY(need(e, TM_STEP_PREPARE_PALETTES, "PreparePalettes"));
  TM_TRY(need_global_rgb(e, "Dither"));
  TM_TRY(e->gpal_px.alloc((size_t)e->t * 64));
  const int64_t t0 = e->t * e->dither_rank / e->dither_world, t1 = e->t * (e->dither_rank + 1) / e->dither_world;
  if (e->dither_world > 1) TM_HIP(hipMemsetAsync(e->gpal_px.p, 0, (size_t)e->t * 64, e->stream));  // other shards' tiles: 0, merged with SUM
  e->dither_pairs = 0;
  if (t1 > t0)
    TM_TRY(launch_dither(e->gtiles.as<uint8_t>() + t0 * 256, e->gflags.as<uint8_t>() + t0, e->gpal_idx.as<uint8_t>() + t0 * 4, t1 - t0, e->palettes_dev.p,
                         e->s.PaletteCount, e->s.PaletteSize, e->s.DitheringUseThomasKnoll ? 1 : 0, e->s.DitheringYliluoma2MixedColors,
                         e->gpal_px.as<uint8_t>() + t0 * 64, e->stream, &e->dither_pairs, e->pair_keys_n > 0 && !knobs().dither_own_keys ? e->pair_keys.p : nullptr,
                         e->pair_keys_n));
  if (e->dist() && e->dither_world > 1) TM_TRY(e->co.allreduce_sum_i32(e->gpal_px.p, e->t * 16));  // 64 bytes per tile = 16 words; other shares hold 0
  TM_HIP(hipStreamSynchronize(e->stream));
  e->has_pal_px = true;
  progress(e, TM_STEP_DITHER, 2, 2);
  return TM_OK;
}

static int step_reconstruct(tm_encoder *e) {
  // Reconstruct, tilingencoder.pas:1928-1962: PrepareReconstruct (4566) builds the int16 database of all global
  // tiles; TFrame.Reconstruct.DoXY (1464-1659) matches every frame tile.  The nearest-neighbour part does not depend on the
  // previous reconstructed frame, so all frames go in one batch; the motion branch (below) then walks the frames in order.
  TM_TRY(need(e, TM_STEP_DITHER, "Dither"));
  TM_TRY(need_frame_tiles(e, "Reconstruct"));
  TM_TRY(load_tail(e));
  DevBuf db, qf;
  TM_TRY(db.alloc((size_t)e->t * 384));
  if (e->dist()) {  // PrepareReconstruct (4566-4613) per share of the global tiles, then the all-gather of the int16 rows (T x 384 bytes in all)
    int64_t t0, t1;
    share_of(e->t, e->co.rank, e->co.world, &t0, &t1);
    DevBuf part, all;
    TM_TRY(part.alloc((size_t)std::max<int64_t>(t1 - t0, 1) * 384));
    if (t1 > t0)
      TM_TRY(launch_features_pal(e->gpal_px.as<uint8_t>() + t0 * 64, e->gpal_idx.as<uint8_t>() + t0 * 4, t1 - t0, e->palettes_dev.p, e->s.PaletteSize, TM_PVS_WEIGHTED_DCT, part.p, e->stream));
    std::vector<int64_t> counts;
    TM_TRY(gather_var(e, part.p, t1 - t0, 384, all, &counts));
    TM_HIP(hipMemcpyAsync(db.p, all.p, (size_t)e->t * 384, hipMemcpyDeviceToDevice, e->stream));
  } else
  TM_TRY(launch_features_pal(e->gpal_px.p, e->gpal_idx.p, e->t, e->palettes_dev.p, e->s.PaletteSize, TM_PVS_WEIGHTED_DCT, db.p, e->stream));
  // Many dithered tiles are byte-identical (Reindex merges them later, MakeTilesUnique(False) at 2014).  Under the
  // lowest-index tie rule the nearest neighbour among ALL rows is the nearest among the DISTINCT rows taken in order of
  // their first occurrence, so only those are searched; indices are mapped back afterwards.
  const int64_t per = e->tm_size();
  const int sf = std::max(0, std::min(e->shard_first, e->nframes));
  const int sn = e->shard_count < 0 ? e->nframes - sf : std::max(0, std::min(e->shard_count, e->nframes - sf));
  TM_CHECK(!e->load_sharded || (sf >= e->load_first && sf + sn <= e->load_first + e->load_count), TM_E_INVAL,
           "Reconstruct: frames [%d, %d) are not the ones this process loaded ([%d, %d))", sf, sf + sn, e->load_first, e->load_first + e->load_count);
  if (sf > 0 || sn < e->nframes) {  // frames of other shards: TileIdx / PalIdx -1 (merged with MAX), error 0 (merged with SUM: an error is any 32-bit pattern)
    TM_HIP(hipMemsetAsync(e->tm_tile.p, 0xff, (size_t)e->q * 4, e->stream));
    TM_HIP(hipMemsetAsync(e->tm_err.p, 0, (size_t)e->q * 4, e->stream));
    TM_HIP(hipMemsetAsync(e->tm_pal.p, 0xff, (size_t)e->q * 4, e->stream));
  }
  e->knn_ms = 0; e->knn_pairs = 0; e->knn_launches = 0; e->knn_db_rows = 0; e->knn_queries = 0;
  for (double &v : e->knn_split_ms) v = 0;
  e->knn_split_pairs[0] = e->knn_split_pairs[1] = e->knn_split_pairs[2] = 0;
  const bool epu = e->s.FrameTilingExtendedPaletteUsage;
  if (epu) {
    // FrameTilingExtendedPaletteUsage (1559-1610): the 64 nearest rows of the whole database (duplicates included, as
    // ann_kdtree_short_search_multi sees them), then every unique tile x every unique palette of that list, scored against a
    // table of all (tile, palette) feature vectors
    DevBuf table, idx64, err64;
    const int npal = e->s.PaletteCount;
    // the table of every tile under every palette while it fits (T x P x 384 bytes: 2 GB at 16 palettes); with the reference's default
    // of 1024 palettes it would be tens of terabytes, and the re-rank builds just the rows its queries name instead
    const double table_gib = knobs().epu_table_gib;
    const bool use_table = (double)e->t * npal * 384.0 <= table_gib * 1073741824.0;
    if (use_table) {
      TM_TRY(table.alloc((size_t)e->t * npal * 384));
      TM_TRY(launch_features_table(e->gpal_px.p, e->t, e->palettes_dev.p, npal, e->s.PaletteSize, table.p, e->stream));
    }
    progress(e, TM_STEP_RECONSTRUCT, 1, 2);
    const int chunk_frames = recon_chunk_frames(e, sn, true);
    const bool groups = query_groups_usable(e, sf, sn, true);
    TM_TRY(idx64.alloc((size_t)(groups ? e->q_groups : chunk_frames * per) * 64 * 4));
    TM_TRY(err64.alloc((size_t)(groups ? e->q_groups : chunk_frames * per) * 64 * 4));
    // the scan runs over the DISTINCT rows; every result is expanded to all its duplicates (they count, as
    // ann_kdtree_short_search_multi sees them) from member lists
    DevBuf d_remap, d_order, d_use, ddb, g_off, g_members;
    TM_TRY(d_remap.alloc((size_t)e->t * 4)); TM_TRY(d_order.alloc((size_t)e->t * 4)); TM_TRY(d_use.alloc((size_t)(e->t + 1) * 4));
    int64_t nd = 0;
    TM_TRY(run_dedup_ex(db.p, e->t, 384, nullptr, d_remap.p, d_order.p, d_use.p, &nd, 1, e->stream));
    TM_TRY(ddb.alloc((size_t)nd * 384));
    hipLaunchKernelGGL(k_gather_rows16, dim3(gridn(nd * 24)), dim3(256), 0, e->stream, db.as<uint4>(), d_order.as<int32_t>(), nd, 24, ddb.as<uint4>());
    TM_HIP(hipGetLastError());
    TM_TRY(g_off.alloc((size_t)(nd + 1) * 4)); TM_TRY(g_members.alloc((size_t)e->t * 4));
    TM_TRY(build_groups(d_remap.p, e->t, d_use.p, nd, g_off.p, g_members.p, e->stream));
    e->knn_db_rows = nd;
    tm_knn_index_impl *ix = nullptr;
    TM_TRY(knn_index_create(ddb.p, nd, e->stream, &ix));
    int rc = TM_OK;
    if (groups) {
      // one query per DISTINCT frame tile (Reduce's groups): the 64 candidates and the re-rank are functions of the query's features alone
      const int64_t ng = e->q_groups;
      DevBuf gt, gp, ge;
      TM_TRY(gt.alloc((size_t)ng * 4)); TM_TRY(gp.alloc((size_t)ng * 4)); TM_TRY(ge.alloc((size_t)ng * 4));
      void *qfp = nullptr;
      if (e->qf_valid && e->qf_distinct) {
        TM_HIP(hipStreamWaitEvent(e->stream, e->ev_qf, 0));
        qfp = e->qf_pre.p;
      } else {
        TM_TRY(qf.alloc((size_t)ng * 384));
        qfp = qf.p;
        TM_TRY(launch_features_rgb_rows(e->ftiles.p, e->q_rep.p, ng, TM_PVS_WEIGHTED_DCT, 0, qf.p, e->stream));
      }
      e->knn_queries += ng;
      rc = knobs().topk_brute ? launch_knn_topk(qfp, ng, db.p, e->t, 64, idx64.p, err64.p, e->stream)
                                   : knn_index_search_topk(ix, qfp, ng, 64, idx64.p, err64.p, e->stream, g_off.p, g_members.p, db.p, e->t);
      if (rc == TM_OK)
        rc = use_table ? launch_epu_rerank(qfp, ng, idx64.p, 64, e->gpal_idx.p, e->t, npal, table.p, gt.as<int32_t>(), gp.as<int32_t>(), ge.as<uint32_t>(), e->stream)
                       : launch_epu_rerank_ondemand(qfp, ng, idx64.p, 64, e->gpal_idx.p, e->t, e->gpal_px.p, e->palettes_dev.p, npal, e->s.PaletteSize,
                                                    gt.as<int32_t>(), gp.as<int32_t>(), ge.as<uint32_t>(), e->stream);
      if (rc == TM_OK) {
        hipLaunchKernelGGL(k_lookup, dim3(gridn(e->q)), dim3(256), 0, e->stream, e->q_group.as<int32_t>(), e->q, gt.as<int32_t>(), e->tm_tile.as<int32_t>());
        hipLaunchKernelGGL(k_lookup, dim3(gridn(e->q)), dim3(256), 0, e->stream, e->q_group.as<int32_t>(), e->q, gp.as<int32_t>(), e->tm_pal.as<int32_t>());
        hipLaunchKernelGGL(k_lookup, dim3(gridn(e->q)), dim3(256), 0, e->stream, e->q_group.as<int32_t>(), e->q, ge.as<int32_t>(), e->tm_err.as<int32_t>());
        TM_HIP(hipGetLastError());
        TM_HIP(hipStreamSynchronize(e->stream));  // gt / gp / ge die with this scope
      }
    } else
    for (int f0 = sf; rc == TM_OK && f0 < sf + sn; f0 += chunk_frames) {
      const int nf = std::min(chunk_frames, sf + sn - f0);
      const int64_t n = (int64_t)nf * per, off = (int64_t)f0 * per;
      void *qfp = nullptr;
      rc = query_features(e, f0, nf, true, qf, &qfp);
      e->knn_queries += n;
      if (rc == TM_OK)
        rc = knobs().topk_brute ? launch_knn_topk(qfp, n, db.p, e->t, 64, idx64.p, err64.p, e->stream)  // debugging aid: VALU brute force over all rows
                                     : knn_index_search_topk(ix, qfp, n, 64, idx64.p, err64.p, e->stream, g_off.p, g_members.p, db.p, e->t);
      if (rc == TM_OK)
        rc = use_table ? launch_epu_rerank(qfp, n, idx64.p, 64, e->gpal_idx.p, e->t, npal, table.p, e->tm_tile.as<int32_t>() + off,
                                           e->tm_pal.as<int32_t>() + off, e->tm_err.as<uint32_t>() + off, e->stream)
                       : launch_epu_rerank_ondemand(qfp, n, idx64.p, 64, e->gpal_idx.p, e->t, e->gpal_px.p, e->palettes_dev.p, npal, e->s.PaletteSize,
                                                    e->tm_tile.as<int32_t>() + off, e->tm_pal.as<int32_t>() + off, e->tm_err.as<uint32_t>() + off, e->stream);
    }
    knn_index_destroy(ix);
    TM_TRY(rc);
    TM_HIP(hipStreamSynchronize(e->stream));
  } else {
  DevBuf u_remap, u_order, u_use, udb;
  TM_TRY(u_remap.alloc((size_t)e->t * 4)); TM_TRY(u_order.alloc((size_t)e->t * 4)); TM_TRY(u_use.alloc((size_t)e->t * 4));
  int64_t nu = 0;
  TM_TRY(run_dedup_ex(db.p, e->t, 384, nullptr, u_remap.p, u_order.p, u_use.p, &nu, 1, e->stream));
  TM_TRY(udb.alloc((size_t)nu * 384));
  hipLaunchKernelGGL(k_gather_rows16, dim3(gridn(nu * 24)), dim3(256), 0, e->stream, db.as<uint4>(), u_order.as<int32_t>(), nu, 24,
                     udb.as<uint4>());
  TM_HIP(hipGetLastError());
  e->knn_db_rows = nu;
  tm_knn_index_impl *ix = nullptr;
  TM_TRY(knn_index_create(udb.p, nu, e->stream, &ix));
  progress(e, TM_STEP_RECONSTRUCT, 1, 2);
  int rc = TM_OK;
  if (query_groups_usable(e, sf, sn, false)) {
    // one query per DISTINCT frame tile (Reduce's groups); the items of a group take its answer
    const int64_t ng = e->q_groups;
    DevBuf gt, ge;
    TM_TRY(gt.alloc((size_t)ng * 4)); TM_TRY(ge.alloc((size_t)ng * 4));
    void *qfp = nullptr;
    const void *qmm = nullptr;
    if (e->qf_valid && e->qf_distinct) {
      TM_HIP(hipStreamWaitEvent(e->stream, e->ev_qf, 0));
      qfp = e->qf_pre.p;
      qmm = e->qf_colmm.p;
    } else {
      TM_TRY(qf.alloc((size_t)ng * 384));
      qfp = qf.p;
      TM_TRY(launch_features_rgb_rows(e->ftiles.p, e->q_rep.p, ng, TM_PVS_WEIGHTED_DCT, 0, qf.p, e->stream));
    }
    rc = knn_index_search(ix, qfp, ng, gt.p, ge.p, e->stream, qmm);
    e->knn_queries += ng;
    if (rc == TM_OK) {
      double ms = 0; int kb = 0; int64_t pairs = 0;
      knn_index_stats(ix, &ms, &kb, &pairs);
      e->knn_ms += ms; e->knn_pairs += pairs; e->knn_launches++; e->knn_kbytes = kb;
      {
        double sm[3]; int64_t sp[3];
        knn_index_kernel_split(ix, sm, sp);
        for (int i_ = 0; i_ < 3; i_++) e->knn_split_ms[i_] += sm[i_];
        e->knn_split_pairs[0] += sp[0]; e->knn_split_pairs[1] += sp[1]; e->knn_split_pairs[2] += sp[2];
      }
      hipLaunchKernelGGL(k_lookup, dim3(gridn(e->q)), dim3(256), 0, e->stream, e->q_group.as<int32_t>(), e->q, gt.as<int32_t>(), e->tm_tile.as<int32_t>());
      hipLaunchKernelGGL(k_lookup, dim3(gridn(e->q)), dim3(256), 0, e->stream, e->q_group.as<int32_t>(), e->q, ge.as<int32_t>(), e->tm_err.as<int32_t>());
      TM_HIP(hipGetLastError());
      TM_HIP(hipStreamSynchronize(e->stream));  // gt / ge die with this scope
    }
  } else {
  // query features in frame chunks (bounded scratch for long / 4K clips: streaming through HBM)
  const int chunk_frames = recon_chunk_frames(e, sn, false);
  for (int f0 = sf; rc == TM_OK && f0 < sf + sn; f0 += chunk_frames) {
    const int nf = std::min(chunk_frames, sf + sn - f0);
    const int64_t n = (int64_t)nf * per, off = (int64_t)f0 * per;
    void *qfp = nullptr;
    rc = query_features(e, f0, nf, false, qf, &qfp);
    if (rc == TM_OK) rc = knn_index_search(ix, qfp, n, e->tm_tile.as<int32_t>() + off, e->tm_err.as<uint32_t>() + off, e->stream);
    e->knn_queries += n;
    if (rc == TM_OK) {
      double ms = 0; int kb = 0; int64_t pairs = 0;
      knn_index_stats(ix, &ms, &kb, &pairs);
      e->knn_ms += ms; e->knn_pairs += pairs; e->knn_launches++; e->knn_kbytes = kb;
      {
        double sm[3]; int64_t sp[3];
        knn_index_kernel_split(ix, sm, sp);
        for (int i_ = 0; i_ < 3; i_++) e->knn_split_ms[i_] += sm[i_];
        e->knn_split_pairs[0] += sp[0]; e->knn_split_pairs[1] += sp[1]; e->knn_split_pairs[2] += sp[2];
      }
    }
  }
  }
  knn_index_destroy(ix);
  TM_TRY(rc);
  hipLaunchKernelGGL(k_lookup_inplace, dim3(gridn(e->q)), dim3(256), 0, e->stream, e->tm_tile.as<int32_t>(), e->q, u_order.as<int32_t>());
  hipLaunchKernelGGL(k_lookup, dim3(gridn(e->q)), dim3(256), 0, e->stream, e->tm_tile.as<int32_t>(), e->q, e->gpal_idx.as<int32_t>(),
                     e->tm_pal.as<int32_t>());  // TMI^.PalIdx := FTiles[TileIdx]^.PalIdx_Initial (1551)
  TM_HIP(hipGetLastError());
  }
  if (e->has_pm) {
    // motion branch (1496-1532, 1612-1654): frames in order, each searched in the previous RECONSTRUCTED frame; a key
    // frame's first frame has no motion candidate, so key-frame groups are independent chains.
    const int sw = e->tm_w * 8, sh = e->tm_h * 8;
    const int64_t nwin = (int64_t)(sw - 7) * (sh - 7);
    DevBuf fb[2], win, cur, mp;
    TM_TRY(fb[0].alloc((size_t)sw * sh * 4)); TM_TRY(fb[1].alloc((size_t)sw * sh * 4));
    TM_TRY(win.alloc((size_t)nwin * 384)); TM_TRY(cur.alloc((size_t)per * 384)); TM_TRY(mp.alloc((size_t)per * 4));
    TM_HIP(hipMemsetAsync(fb[0].p, 0, (size_t)sw * sh * 4, e->stream));
    TM_HIP(hipMemsetAsync(fb[1].p, 0, (size_t)sw * sh * 4, e->stream));
    std::vector<uint8_t> is_kf((size_t)e->nframes, 0);
    for (int32_t k : e->kf_start) is_kf[(size_t)k] = 1;
    TM_CHECK(sn == 0 || is_kf[(size_t)sf], TM_E_INVAL, "Reconstruct with motion prediction: a shard must start on a key frame (frame %d does not)", sf);
    if (sf > 0 || sn < e->nframes) {  // other shards' frames: zeros, so the host merges shards with all-reduce(SUM) on these arrays
      const int64_t a = (int64_t)sf * per, b = (int64_t)(sf + sn) * per;
      TM_HIP(hipMemsetAsync(e->tm_px.p, 0, (size_t)a, e->stream)); TM_HIP(hipMemsetAsync(e->tm_py.p, 0, (size_t)a, e->stream));
      TM_HIP(hipMemsetAsync(e->tm_px.as<uint8_t>() + b, 0, (size_t)(e->q - b), e->stream));
      TM_HIP(hipMemsetAsync(e->tm_py.as<uint8_t>() + b, 0, (size_t)(e->q - b), e->stream));
      TM_HIP(hipMemsetAsync(e->tm_pred.p, 0, (size_t)e->q, e->stream));
    }
    int cb = 0;
    for (int f = sf; f < sf + sn; f++) {
      const int64_t off = (int64_t)f * per;
      const bool search = !is_kf[(size_t)f];  // (Index <> PKeyFrame.StartFrame) and (ARadius >= 0), 1496
      if (search) {
        TM_TRY(launch_features_rgb(e->ftiles.as<uint8_t>() + off * 256, per, e->fflags.as<uint8_t>() + off, TM_PVS_WEIGHTED_DCT, 0, cur.p, e->stream));
        TM_TRY(launch_motion_search_fb(cur.p, e->tm_w, e->tm_h, fb[cb].p, win.p, e->s.MotionPredictRadius, mp.p, e->tm_px.as<int8_t>() + off,
                                       e->tm_py.as<int8_t>() + off, e->stream));
      }
      TM_TRY(launch_recon_decide(e->tm_w, (int)per, epu ? 1 : 0, search ? mp.p : nullptr, e->fflags.as<uint8_t>() + off, e->gpal_idx.p, e->gpal_px.p,
                                 e->palettes_dev.p, e->s.PaletteSize, fb[cb].p, fb[cb ^ 1].p, e->tm_tile.as<int32_t>() + off,
                                 e->tm_pal.as<int32_t>() + off, e->tm_err.as<uint32_t>() + off, e->tm_px.as<int8_t>() + off,
                                 e->tm_py.as<int8_t>() + off, e->tm_pred.as<uint8_t>() + off, e->stream));
      cb ^= 1;
    }
  }
  if (e->dist()) {  // merge the shards' items: TileIdx (and the re-rank's PalIdx) by MAX (others hold -1), the error and the motion results by SUM (others hold 0)
    TM_TRY(e->co.allreduce_max_i32(e->tm_tile.p, e->q));
    TM_TRY(e->co.allreduce_sum_i32(e->tm_err.p, e->q));
    if (epu) TM_TRY(e->co.allreduce_max_i32(e->tm_pal.p, e->q));
    if (e->has_pm) {
      TM_TRY(e->co.allreduce_sum_i32(e->tm_pred.p, (e->q + 3) / 4));
      TM_TRY(e->co.allreduce_sum_i32(e->tm_px.p, (e->q + 3) / 4));
      TM_TRY(e->co.allreduce_sum_i32(e->tm_py.p, (e->q + 3) / 4));
    }
    if (!epu) {  // TMI^.PalIdx := FTiles[TileIdx]^.PalIdx_Initial for every item (1551)
      hipLaunchKernelGGL(k_lookup, dim3(gridn(e->q)), dim3(256), 0, e->stream, e->tm_tile.as<int32_t>(), e->q, e->gpal_idx.as<int32_t>(), e->tm_pal.as<int32_t>());
      TM_HIP(hipGetLastError());
    }
  }
  TM_HIP(hipStreamSynchronize(e->stream));
  e->drop_prefetch();  // consumed (or not this chunk's): the buffer goes back to the pool now that both streams are idle
  e->reconstructed = true;
  progress(e, TM_STEP_RECONSTRUCT, 2, 2);
  return TM_OK;
}

static int step_reindex(tm_encoder *e) {  // Reindex, tilingencoder.pas:1993-2038
  TM_TRY(need(e, TM_STEP_RECONSTRUCT, "Reconstruct"));
  DevBuf hist, remap, order, use;
  TM_TRY(hist.alloc((size_t)e->t * 4));
  TM_TRY(remap.alloc((size_t)e->t * 4));
  TM_TRY(order.alloc((size_t)e->t * 4));
  TM_TRY(use.alloc((size_t)e->t * 4));
  // UseCount recount from the tile maps (2018-2031); MakeTilesUnique(False) merges by palette-index content and
  // sums the counts of merged tiles -- same totals as counting after the merge remap
  {  // one histogram copy per XCD, folded afterwards ("Atomics across XCDs")
    DevBuf h8;
    TM_TRY(h8.alloc((size_t)e->t * 4 * 8));
    TM_HIP(hipMemsetAsync(h8.p, 0, (size_t)e->t * 4 * 8, e->stream));
    hipLaunchKernelGGL(k_histogram_xcd, dim3(gridn(e->q)), dim3(256), 0, e->stream, e->tm_tile.as<int32_t>(), e->q, h8.as<uint32_t>(), (int64_t)e->t);
    hipLaunchKernelGGL(k_hist_fold, dim3(gridn(e->t)), dim3(256), 0, e->stream, h8.as<uint32_t>(), (int64_t)e->t, hist.as<uint32_t>());
    // (h8 goes back to the pool with this scope; what takes it next is queued on this stream behind the fold)
  }
  int64_t nu = 0;
  TM_TRY(run_dedup(e->gpal_px.p, e->t, 64, hist.p, remap.p, order.p, use.p, &nu, e->stream));
  progress(e, TM_STEP_REINDEX, 2, 3);
  DevBuf ntiles, nflags, npal_idx, npal_px, ntm;
  TM_TRY(ntiles.alloc((size_t)nu * 256)); TM_TRY(nflags.alloc((size_t)std::max<int64_t>(nu, 1))); TM_TRY(npal_idx.alloc((size_t)nu * 4));
  TM_TRY(npal_px.alloc((size_t)nu * 64)); TM_TRY(ntm.alloc((size_t)e->q * 4));
  hipLaunchKernelGGL(k_gather_rows16, dim3(gridn(nu * 16)), dim3(256), 0, e->stream, e->gtiles.as<uint4>(), order.as<int32_t>(), nu, 16, ntiles.as<uint4>());
  hipLaunchKernelGGL(k_gather_rows16, dim3(gridn(nu * 4)), dim3(256), 0, e->stream, e->gpal_px.as<uint4>(), order.as<int32_t>(), nu, 4, npal_px.as<uint4>());
  hipLaunchKernelGGL(k_gather<uint8_t>, dim3(gridn(nu)), dim3(256), 0, e->stream, e->gflags.as<uint8_t>(), order.as<int32_t>(), nu, nflags.as<uint8_t>());
  hipLaunchKernelGGL(k_gather<int32_t>, dim3(gridn(nu)), dim3(256), 0, e->stream, e->gpal_idx.as<int32_t>(), order.as<int32_t>(), nu, npal_idx.as<int32_t>());
  hipLaunchKernelGGL(k_lookup, dim3(gridn(e->q)), dim3(256), 0, e->stream, e->tm_tile.as<int32_t>(), e->q, remap.as<int32_t>(), ntm.as<int32_t>());
  TM_HIP(hipGetLastError());
  TM_HIP(hipStreamSynchronize(e->stream));
  e->pair_keys_n = 0;
  e->gtiles = std::move(ntiles); e->gflags = std::move(nflags); e->gpal_idx = std::move(npal_idx); e->gpal_px = std::move(npal_px);
  e->tm_tile = std::move(ntm);
  e->guse = std::move(use);
  e->t = nu;
  progress(e, TM_STEP_REINDEX, 3, 3);
  return TM_OK;
}

static std::string settings_text(const Settings &s) {  // GetSettings -> SaveSettings, tilingencoder.pas:2255, 3738-3775 (TMemIniFile layout)
  // Sections in the order of their first write, keys in write order inside a section (the ShotTrans* keys are written last but belong
  // to [Load]), a blank line after every section but the last, WriteBool as 0/1, WriteFloat in the shortest form -- and the line
  // ends of the Win64 build that is the reference (CR LF): the text of the reference's own demo streams, line for line, for every key
  // this snapshot still writes (tests/test_gtm.py::test_settings_text_matches_the_demo_streams_line_for_line).
  char buf[2048];
  auto flt = [](double v) { char b[64]; snprintf(b, sizeof(b), "%.15g", v); return std::string(b); };
  snprintf(buf, sizeof(buf),
           "[Load]\nInputFileName=%s\nOutputFileName=%s\nStartFrame=%d\nFrameCount=%d\nScaling=%s\nShotTransMaxSecondsPerKF=%s\n"
           "ShotTransMinSecondsPerKF=%s\nShotTransCorrelLoThres=%s\n\n[MotionPredict]\nMotionPredictRadius=%d\n\n"
           "[GlobalTiling]\nGlobalTilingUseTargetPSNR=%d\nGlobalTilingTargetPSNR=%s\nGlobalTilingQualityBasedTileCount=%s\n"
           "GlobalTilingTileCount=%d\n\n[Dither]\nPaletteSize=%d\nPaletteCount=%d\nDitheringMode=%d\nDitheringUseThomasKnoll=%d\n"
           "DitheringYliluoma2MixedColors=%d\n\n[FrameTiling]\nFrameTilingExtendedPaletteUsage=%d\n\n[Misc]\nMaxThreadCount=%d\n",
           s.InputFileName.c_str(), s.OutputFileName.c_str(), s.StartFrame, s.FrameCount, flt(s.Scaling).c_str(),
           flt(s.ShotTransMaxSecondsPerKF).c_str(), flt(s.ShotTransMinSecondsPerKF).c_str(), flt(s.ShotTransCorrelLoThres).c_str(),
           s.MotionPredictRadius, (int)s.GlobalTilingUseTargetPSNR, flt(s.GlobalTilingTargetPSNR).c_str(),
           flt(s.GlobalTilingQualityBasedTileCount).c_str(), s.GlobalTilingTileCount, s.PaletteSize, s.PaletteCount, s.DitheringMode,
           (int)s.DitheringUseThomasKnoll, s.DitheringYliluoma2MixedColors, (int)s.FrameTilingExtendedPaletteUsage, s.MaxThreadCount);
  std::string out;
  for (const char *c = buf; *c; c++) { if (*c == '\n') out += '\r'; out += *c; }
  return out;
}

static int save_to(tm_encoder *e, const char *path) {  // Save, tilingencoder.pas:2040-2058 -> SaveStream, 5177
  TM_TRY(need(e, TM_STEP_REINDEX, "Reindex"));
  TM_CHECK(path && *path, TM_E_INVAL, "Save: no output file name");
  TM_TRY(load_tail(e));
  TM_HIP(hipSetDevice(e->device));
  GtmInput in;
  in.tm_w = e->tm_w; in.tm_h = e->tm_h; in.nframes = e->nframes; in.fps = e->fps;
  in.kf_start = e->kf_start;
  std::vector<uint8_t> pal_px((size_t)e->t * 64);
  in.use.resize((size_t)e->t);
  if (e->t) {
    TM_HIP(hipMemcpy(pal_px.data(), e->gpal_px.p, pal_px.size(), hipMemcpyDeviceToHost));
    TM_HIP(hipMemcpy(in.use.data(), e->guse.p, (size_t)e->t * 4, hipMemcpyDeviceToHost));
  }
  in.pal_px = pal_px.data();
  in.palettes = e->palettes_host.data();
  in.pal_count = e->s.PaletteCount; in.pal_size = e->s.PaletteSize;
  std::vector<tm_tilemap_item> tmi((size_t)e->q);
  for (int f = 0; f < e->nframes; f++) TM_TRY(tm_get_tilemap(e, f, tmi.data() + (size_t)f * e->tm_size()));
  in.tilemap = tmi.data();
  in.settings = settings_text(e->s);
  return write_gtm(path, in);
}

// ---- GenerateY4M / GeneratePNGs (tilingencoder.pas:2126-2199, 2075-2124): the frames as Render (3455-3640) draws them with the
// constructor's defaults (FRenderPredicted, FRenderMirrored, FRenderOutputDithered on, no gamma: 5505-5507).  Host code: export tooling.
namespace {
struct FrameRenderer {
  tm_encoder *e;
  int sw, sh;
  std::vector<uint8_t> pal_px, fflags;
  std::vector<uint32_t> front, back, in_tiles;  // 0x00BBGGRR
  int init(bool input) {
    sw = e->tm_w * 8; sh = e->tm_h * 8;
    front.assign((size_t)sw * sh, 0); back.assign((size_t)sw * sh, 0);
    if (input) {
      TM_CHECK(e->ftiles.p && !e->load_sharded, TM_E_INVAL, "input frames: the frame tiles are not in memory (run Load)");
      in_tiles.resize((size_t)e->tm_size() * 64);
      fflags.resize((size_t)e->q);
      TM_HIP(hipMemcpy(fflags.data(), e->fflags.p, (size_t)e->q, hipMemcpyDeviceToHost));
    } else {
      TM_CHECK(e->has_pal_px && (e->steps_done & (1 << TM_STEP_RECONSTRUCT)), TM_E_INVAL, "output frames: Reconstruct (or ReloadGTM) has not been run");
      pal_px.resize((size_t)std::max<int64_t>(e->t, 1) * 64);
      if (e->t) TM_HIP(hipMemcpy(pal_px.data(), e->gpal_px.p, (size_t)e->t * 64, hipMemcpyDeviceToHost));
    }
    return TM_OK;
  }
  int render(int f, bool input) {  // -> front
    const int64_t per = e->tm_size();
    if (input) {  // "Input" tab (3537-3570): the frame's tiles back in their original orientation
      TM_HIP(hipMemcpy(in_tiles.data(), e->ftiles.as<uint8_t>() + (int64_t)f * per * 256, (size_t)per * 256, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < per; i++) {
        const int sx = (int)(i % e->tm_w), sy = (int)(i / e->tm_w), fl = fflags[(size_t)(f * per + i)];
        for (int ty = 0; ty < 8; ty++)
          for (int tx = 0; tx < 8; tx++)
            front[(size_t)(sy * 8 + ty) * sw + sx * 8 + tx] = in_tiles[(size_t)i * 64 + ((fl & 2) ? 7 - ty : ty) * 8 + ((fl & 1) ? 7 - tx : tx)] & 0xffffffu;
      }
      return TM_OK;
    }
    std::vector<tm_tilemap_item> tmi((size_t)per);
    TM_TRY(tm_get_tilemap(e, f, tmi.data()));
    std::swap(front, back);  // FRenderBackBuffer := the previous output frame (3575-3576)
    std::fill(front.begin(), front.end(), 0u);
    for (int64_t i = 0; i < per; i++) {
      const tm_tilemap_item &it = tmi[(size_t)i];
      const int sx = (int)(i % e->tm_w), sy = (int)(i / e->tm_w);
      if (it.Flags & 4) {  // predicted: 8 x 8 pixels of the back buffer at the predicted offset (3595-3606)
        for (int ty = 0; ty < 8; ty++)
          for (int tx = 0; tx < 8; tx++) {
            const int by = std::min(std::max(sy * 8 + it.PredictedY + ty, 0), sh - 1), bx = std::min(std::max(sx * 8 + it.PredictedX + tx, 0), sw - 1);
            front[(size_t)(sy * 8 + ty) * sw + sx * 8 + tx] = back[(size_t)by * sw + bx];
          }
      } else if (it.TileIdx >= 0 && it.TileIdx < e->t && it.PalIdx >= 0 && it.PalIdx < e->s.PaletteCount) {
        const uint8_t *px = &pal_px[(size_t)it.TileIdx * 64];
        const int32_t *pal = &e->palettes_host[(size_t)it.PalIdx * e->s.PaletteSize];
        for (int ty = 0; ty < 8; ty++)
          for (int tx = 0; tx < 8; tx++)
            front[(size_t)(sy * 8 + ty) * sw + sx * 8 + tx] = (uint32_t)pal[px[((it.Flags & 2) ? 7 - ty : ty) * 8 + ((it.Flags & 1) ? 7 - tx : tx)]] & 0xffffffu;
      }
    }
    return TM_OK;
  }
};

uint32_t crc32_of(const uint8_t *p, size_t n, uint32_t crc) {
  static uint32_t table[256];
  static bool made = false;
  if (!made) {
    for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
    made = true;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return ~crc;
}
void png_chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &data) {
  auto be32 = [&](uint32_t v) { out.push_back(v >> 24); out.push_back(v >> 16); out.push_back(v >> 8); out.push_back(v); };
  be32((uint32_t)data.size());
  const size_t at = out.size();
  out.insert(out.end(), type, type + 4);
  out.insert(out.end(), data.begin(), data.end());
  be32(crc32_of(out.data() + at, out.size() - at, 0));
}
// 24-bit RGB PNG (pf24bit, 2086); the image data travels in stored deflate blocks: valid for every decoder, no codec dependency
int write_png(const std::string &path, const std::vector<uint32_t> &img, int w, int h) {
  std::vector<uint8_t> raw((size_t)h * (1 + (size_t)w * 3));
  for (int y = 0; y < h; y++) {
    uint8_t *row = &raw[(size_t)y * (1 + (size_t)w * 3)];
    row[0] = 0;  // filter: none
    for (int x = 0; x < w; x++) { const uint32_t c = img[(size_t)y * w + x]; row[1 + x * 3] = c & 0xff; row[2 + x * 3] = (c >> 8) & 0xff; row[3 + x * 3] = (c >> 16) & 0xff; }
  }
  std::vector<uint8_t> z = {0x78, 0x01};
  uint32_t a = 1, b = 0;
  for (uint8_t v : raw) { a = (a + v) % 65521u; b = (b + a) % 65521u; }
  for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
    const size_t n = std::min<size_t>(65535, raw.size() - off);
    z.push_back(off + n >= raw.size() ? 1 : 0);
    z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
    z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
    if (raw.empty()) break;
  }
  z.push_back(b >> 8); z.push_back(b); z.push_back(a >> 8); z.push_back(a);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<uint8_t> ihdr = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 2, 0, 0, 0};
  png_chunk(out, "IHDR", ihdr);
  png_chunk(out, "IDAT", z);
  png_chunk(out, "IEND", {});
  std::ofstream f(path, std::ios::binary);
  TM_CHECK(f.good(), TM_E_IO, "cannot write %s", path.c_str());
  f.write((const char *)out.data(), (std::streamsize)out.size());
  return TM_OK;
}
std::string strip_ext(const std::string &p) {  // ChangeFileExt(name, '')
  const size_t dot = p.find_last_of('.'), sep = p.find_last_of("/\\");
  return (dot != std::string::npos && (sep == std::string::npos || dot > sep)) ? p.substr(0, dot) : p;
}
}  // namespace

static int generate_y4m(tm_encoder *e, const char *path, bool input) {  // GenerateY4M, tilingencoder.pas:2126-2199
  TM_CHECK(path && *path, TM_E_INVAL, "GenerateY4M: no file name");
  TM_HIP(hipSetDevice(e->device));
  FrameRenderer r{e};
  TM_TRY(r.init(input));
  std::ofstream f(path, std::ios::binary);
  TM_CHECK(f.good(), TM_E_IO, "cannot write %s", path);
  char hdr[128];
  snprintf(hdr, sizeof(hdr), "YUV4MPEG2 W%d H%d F%lld:1000000 Ip C444\n", r.sw, r.sh, (long long)std::nearbyint(e->fps * 1000000.0));  // 2146
  f << hdr;
  const size_t plane = (size_t)r.sw * r.sh;
  std::vector<uint8_t> yuv(plane * 3);
  auto rnd = [](float v, float add) { const long long q = (long long)std::nearbyint((double)(v + add)); return (uint8_t)std::min<long long>(255, std::max<long long>(0, q)); };
  for (int fr = 0; fr < e->nframes; fr++) {
    TM_TRY(r.render(fr, input));
    f << "FRAME \n";  // (with the space, 2161)
    for (size_t i = 0; i < plane; i++) {
      const uint32_t c = r.front[i];
      const int rr = c & 0xff, gg = (c >> 8) & 0xff, bb = (c >> 16) & 0xff;
      // RGBToYUV, utils.pas:478-490: the decimal constants are doubles, every right-hand side narrows to Single once
      const float yy = (float)(rr * (299.0 / 1000) + gg * (587.0 / 1000) + bb * (114.0 / 1000));
      const float uu = (float)(((double)bb - (double)yy) * 0.492), vv = (float)(((double)rr - (double)yy) * 0.877);
      yuv[i] = rnd(yy, 0.0f);
      yuv[plane + i] = rnd(uu, 128.0f);      // uf - Low(ShortInt)
      yuv[2 * plane + i] = rnd(vv, 128.0f);
    }
    f.write((const char *)yuv.data(), (std::streamsize)yuv.size());
    if ((fr & 15) == 15) progress(e, TM_STEP_SAVE, fr, e->nframes);
  }
  TM_CHECK(f.good(), TM_E_IO, "write to %s failed", path);
  return TM_OK;
}

static int generate_pngs(tm_encoder *e, bool input) {  // GeneratePNGs, tilingencoder.pas:2075-2124
  TM_CHECK(!e->s.OutputFileName.empty(), TM_E_INVAL, "GeneratePNGs: OutputFileName is not set");
  TM_HIP(hipSetDevice(e->device));
  FrameRenderer r{e};
  TM_TRY(r.init(input));
  const std::string base = strip_ext(e->s.OutputFileName);
  {
    std::ofstream pf(base + ".txt");  // the palettes, one colour per line: IntToHex($ff000000 or PaletteRGB, 8) (2101-2104)
    TM_CHECK(pf.good(), TM_E_IO, "cannot write %s.txt", base.c_str());
    char line[16];
    for (int32_t c : e->palettes_host) { snprintf(line, sizeof(line), "%08X", 0xff000000u | (uint32_t)c); pf << line << "\n"; }
  }
  for (int fr = 0; fr < e->nframes; fr++) {
    TM_TRY(r.render(fr, input));
    char name[32];
    snprintf(name, sizeof(name), "_%04d.png", fr);
    TM_TRY(write_png(base + name, r.front, r.sw, r.sh));
  }
  return TM_OK;
}

static int run_step(tm_encoder *e, int step) {
  TM_HIP(hipSetDevice(e->device));
  const auto t0 = std::chrono::steady_clock::now();
  int rc = TM_OK;
  switch (step) {
    case TM_STEP_LOAD: rc = step_load(e); break;
    case TM_STEP_PREDICT_MOTION: rc = step_predict_motion(e); break;
    case TM_STEP_REDUCE: rc = step_reduce(e); break;
    case TM_STEP_PREPARE_PALETTES: rc = step_prepare_palettes(e); break;
    case TM_STEP_DITHER: rc = step_dither(e); break;
    case TM_STEP_RECONSTRUCT: rc = step_reconstruct(e); break;
    case TM_STEP_REINDEX: rc = step_reindex(e); break;
    case TM_STEP_SAVE: rc = save_to(e, e->s.OutputFileName.c_str()); break;
    default: set_error("bad step %d", step); rc = TM_E_INVAL;
  }
  if (rc == TM_OK) {
    e->stage_ms[step] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    e->steps_done |= 1 << step;
    for (int later = step + 1; later < 8; later++) e->steps_done &= ~(1 << later);  // later state is stale now
  }
  return rc;
}

extern "C" {

tm_encoder *tm_create(void) {
  if (require_device() != TM_OK) return nullptr;
  tm_encoder *e = new tm_encoder();
  if (hipGetDevice(&e->device) != hipSuccess) e->device = 0;
  return e;
}

void tm_destroy(tm_encoder *e) {
  delete e;
  pool_trim();  // the calling thread's cached device blocks go back to the driver with the encoder
}

int tm_set_device(tm_encoder *e, int device) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  TM_HIP(hipSetDevice(device));
  e->device = device;
  return TM_OK;
}

int tm_load_default_settings(tm_encoder *e) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  e->s = Settings();
  e->auto_tile_count = true;
  return TM_OK;
}

static int load_settings_from(tm_encoder *e, std::istream &in) {  // LoadSettings, tilingencoder.pas:3777-3815
  tm_load_default_settings(e);
  std::map<std::string, std::string> kv;
  std::string line;
  while (std::getline(in, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
    if (line.empty() || line[0] == '[' || line[0] == ';') continue;  // key names are unique across sections
    const size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    kv[line.substr(0, eq)] = line.substr(eq + 1);
  }
  // GlobalTilingTileCount after GlobalTilingQualityBasedTileCount: it has priority (3796)
  static const char *order[] = {"StartFrame", "FrameCount", "Scaling", "MotionPredictRadius", "GlobalTilingUseTargetPSNR",
                                "GlobalTilingTargetPSNR", "GlobalTilingQualityBasedTileCount", "GlobalTilingTileCount", "PaletteSize",
                                "PaletteCount", "DitheringMode", "DitheringUseThomasKnoll", "DitheringYliluoma2MixedColors",
                                "FrameTilingExtendedPaletteUsage", "MaxThreadCount", "ShotTransMaxSecondsPerKF",
                                "ShotTransMinSecondsPerKF", "ShotTransCorrelLoThres"};
  for (const char *k : order) {
    auto it = kv.find(k);
    if (it == kv.end()) continue;
    std::string v = it->second;
    std::replace(v.begin(), v.end(), ',', '.');
    TM_TRY(set_number(e, k, atof(v.c_str()), false));
  }
  if (kv.count("InputFileName")) e->s.InputFileName = kv["InputFileName"];
  if (kv.count("OutputFileName")) e->s.OutputFileName = kv["OutputFileName"];
  return TM_OK;
}

int tm_load_settings_ini(tm_encoder *e, const char *path) {
  TM_CHECK(e && path, TM_E_INVAL, "null argument");
  std::ifstream in(path);
  TM_CHECK(in.good(), TM_E_IO, "cannot open %s", path);
  return load_settings_from(e, in);
}

int tm_save_settings_ini(tm_encoder *e, const char *path) {  // SaveSettings, tilingencoder.pas:3738-3775
  TM_CHECK(e && path, TM_E_INVAL, "null argument");
  std::ofstream out(path, std::ios::binary);
  TM_CHECK(out.good(), TM_E_IO, "cannot create %s", path);
  const std::string t = settings_text(e->s);
  out.write(t.data(), (std::streamsize)t.size());
  TM_CHECK(out.good(), TM_E_IO, "cannot write %s", path);
  return TM_OK;
}

// LoadSettings then SaveSettings on text, no device and no encoder needed: what an INI text becomes once the setters have clamped it
// (the embedded settings of a .gtm, tilingencoder.pas:5331-5335, are this text)
int tm_settings_text_host(const char *ini_text, char *out, int64_t cap, int64_t *out_len) {
  TM_CHECK(ini_text && out_len, TM_E_INVAL, "null argument");
  tm_encoder *e = new tm_encoder();  // host state only: nothing here touches a device
  std::istringstream in(ini_text);
  const int rc = load_settings_from(e, in);
  std::string t;
  if (rc == TM_OK) t = settings_text(e->s);
  delete e;
  TM_TRY(rc);
  *out_len = (int64_t)t.size();
  if (out && cap > 0) {
    const size_t n = std::min<size_t>(t.size(), (size_t)cap - 1);
    memcpy(out, t.data(), n);
    out[n] = 0;
  }
  return TM_OK;
}

int tm_set_int(tm_encoder *e, const char *key, int64_t v) { TM_CHECK(e && key, TM_E_INVAL, "null argument"); return set_number(e, key, (double)v, true); }
int tm_set_float(tm_encoder *e, const char *key, double v) { TM_CHECK(e && key, TM_E_INVAL, "null argument"); return set_number(e, key, v, false); }
int tm_set_bool(tm_encoder *e, const char *key, int v) { TM_CHECK(e && key, TM_E_INVAL, "null argument"); return set_number(e, key, v ? 1 : 0, true); }
int tm_set_str(tm_encoder *e, const char *key, const char *v) {
  TM_CHECK(e && key && v, TM_E_INVAL, "null argument");
  if (!strcmp(key, "InputFileName")) e->s.InputFileName = v;
  else if (!strcmp(key, "OutputFileName")) e->s.OutputFileName = v;
  else { set_error("unknown string setting '%s'", key); return TM_E_INVAL; }
  return TM_OK;
}
int tm_get_int(tm_encoder *e, const char *key, int64_t *v) {
  TM_CHECK(e && key && v, TM_E_INVAL, "null argument");
  double d;
  TM_TRY(get_number(e, key, &d));
  *v = (int64_t)llrint(d);
  return TM_OK;
}
int tm_get_float(tm_encoder *e, const char *key, double *v) { TM_CHECK(e && key && v, TM_E_INVAL, "null argument"); return get_number(e, key, v); }

int tm_set_progress_cb(tm_encoder *e, tm_progress_cb cb, void *user) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  e->cb = cb;
  e->cb_user = user;
  return TM_OK;
}

int tm_set_video(tm_encoder *e, int width, int height, double fps, int frame_count) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  TM_CHECK(width > 0 && height > 0 && frame_count > 0 && fps >= 0, TM_E_INVAL, "bad video geometry %dx%d x%d", width, height, frame_count);
  TM_HIP(hipSetDevice(e->device));
  e->width = width; e->height = height; e->fps = fps; e->nframes = frame_count;
  e->tm_w = (width - 1) / 8 + 1;   // ReframeUI((DstWidth - 1) div cTileWidth + 1, ...), tilingencoder.pas:1776
  e->tm_h = (height - 1) / 8 + 1;
  e->frames = nullptr;
  e->frames_host = nullptr;
  e->frames_owned.release();
  if (e->copy_stream) TM_HIP(hipStreamSynchronize(e->copy_stream));  // a prefetch of the old geometry may still be running
  if (e->stream_aux) TM_HIP(hipStreamSynchronize(e->stream_aux));    // and so may the old clip's correlation
  e->load_tail_pending = false;
  for (tm_encoder::HostClip &c : e->hclip) { c.buf.release(); c.pending = false; c.host = nullptr; }
  e->hclip_cur = -1;
  e->steps_done = 0;
  if (e->auto_tile_count) recompute_auto_tile_count(e);
  return TM_OK;
}

int tm_push_frame_rgb32(tm_encoder *e, int index, const uint32_t *pixels, int stride_px) {
  TM_CHECK(e && pixels, TM_E_INVAL, "null argument");
  TM_CHECK(e->nframes > 0, TM_E_INVAL, "tm_set_video has not been called");
  TM_CHECK(index >= 0 && index < e->nframes && stride_px >= e->width, TM_E_INVAL, "bad frame index/stride");
  TM_HIP(hipSetDevice(e->device));
  const size_t fbytes = (size_t)e->width * e->height * 4;
  e->frames_host = nullptr;
  e->hclip_cur = -1;
  if (!e->frames_owned.p || e->frames != e->frames_owned.p) {
    TM_TRY(e->frames_owned.alloc(fbytes * e->nframes));
    TM_HIP(hipMemsetAsync(e->frames_owned.p, 0, fbytes * e->nframes, e->stream));
    e->frames = e->frames_owned.p;
  }
  TM_HIP(hipMemcpy2DAsync(e->frames_owned.as<uint8_t>() + fbytes * index, (size_t)e->width * 4, pixels, (size_t)stride_px * 4,
                          (size_t)e->width * 4, e->height, hipMemcpyHostToDevice, e->stream));
  TM_HIP(hipStreamSynchronize(e->stream));  // caller's buffer is only valid during the call (extern.pas:883-892)
  return TM_OK;
}

int tm_set_frames_device(tm_encoder *e, const void *dev_frames) {
  TM_CHECK(e && dev_frames, TM_E_INVAL, "null argument");
  TM_CHECK(e->nframes > 0, TM_E_INVAL, "tm_set_video has not been called");
  e->frames_owned.release();
  e->frames = dev_frames;
  e->frames_host = nullptr;
  e->hclip_cur = -1;
  return TM_OK;
}

int tm_set_frames_host(tm_encoder *e, const uint32_t *host_frames) {
  TM_CHECK(e && host_frames, TM_E_INVAL, "null argument");
  TM_CHECK(e->nframes > 0, TM_E_INVAL, "tm_set_video has not been called");
  e->frames_host = host_frames;
  e->frames = nullptr;
  e->hclip_cur = -1;
  return TM_OK;
}

int tm_prefetch_frames_host(tm_encoder *e, const uint32_t *host_frames) {
  TM_CHECK(e && host_frames, TM_E_INVAL, "null argument");
  TM_CHECK(e->nframes > 0, TM_E_INVAL, "tm_set_video has not been called");
  TM_HIP(hipSetDevice(e->device));
  // the buffer to fill: one that neither holds a clip waiting for its Load nor the clip the steps in flight may still read from --
  // unless both are taken, in which case the clip of the last Load gives way (its steps have returned: tm_run blocks)
  int slot = -1;
  for (int i = 0; i < 2; i++)
    if (!e->hclip[i].pending && i != e->hclip_cur) slot = i;
  if (slot < 0 && e->hclip_cur >= 0 && !e->hclip[e->hclip_cur].pending) {
    slot = e->hclip_cur;
    e->hclip_cur = -1;
    if (e->frames == e->hclip[slot].buf.p) e->frames = nullptr;  // a Load without new frames would read a clip being overwritten
  }
  TM_CHECK(slot >= 0, TM_E_INVAL, "tm_prefetch_frames_host: two clips are already waiting for their Load");
  TM_HIP(hipStreamSynchronize(e->stream));  // nothing queued on the encoder's stream still reads (or the pool just handed out) that buffer
  return queue_host_clip(e, slot, host_frames);
}

int tm_run(tm_encoder *e, int step) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  knobs_reload();  // the environment switches are sampled here, once per Run; the steps read the sampled set
  if (step == TM_STEP_ALL) {  // Run(esAll): every step in order (5535-5553); Save only once an output name is set
    for (int s = TM_STEP_LOAD; s <= TM_STEP_REINDEX; s++) TM_TRY(run_step(e, s));
    if (!e->s.OutputFileName.empty()) TM_TRY(run_step(e, TM_STEP_SAVE));
    return TM_OK;
  }
  return run_step(e, step);
}

int tm_get_counts(tm_encoder *e, int64_t *tiles, int *frames, int *palettes, int *tm_w, int *tm_h, int *keyframes) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  if (tiles) *tiles = e->t;
  if (frames) *frames = e->nframes;
  if (palettes) *palettes = e->palettes_host.empty() ? 0 : e->s.PaletteCount;
  if (tm_w) *tm_w = e->tm_w;
  if (tm_h) *tm_h = e->tm_h;
  if (keyframes) { TM_TRY(load_tail(e)); *keyframes = (int)e->kf_start.size(); }
  return TM_OK;
}

int tm_get_tiles(tm_encoder *e, int64_t first, int64_t count, tm_tile_hdr *hdrs, uint8_t *pal_px, uint32_t *rgb_px) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  TM_CHECK(first >= 0 && count >= 0 && first + count <= e->t, TM_E_INVAL, "tile range [%lld,+%lld) outside 0..%lld", (long long)first,
           (long long)count, (long long)e->t);
  if (count == 0) return TM_OK;
  TM_HIP(hipSetDevice(e->device));
  if (rgb_px) TM_HIP(hipMemcpy(rgb_px, e->gtiles.as<uint8_t>() + first * 256, (size_t)count * 256, hipMemcpyDeviceToHost));
  if (pal_px) {
    if (e->has_pal_px) TM_HIP(hipMemcpy(pal_px, e->gpal_px.as<uint8_t>() + first * 64, (size_t)count * 64, hipMemcpyDeviceToHost));
    else memset(pal_px, 0, (size_t)count * 64);
  }
  if (hdrs) {
    std::vector<uint32_t> use(count);
    std::vector<int32_t> pi(count, -1);
    std::vector<uint8_t> fl(count);
    TM_HIP(hipMemcpy(use.data(), e->guse.as<uint8_t>() + first * 4, (size_t)count * 4, hipMemcpyDeviceToHost));
    TM_HIP(hipMemcpy(fl.data(), e->gflags.as<uint8_t>() + first, (size_t)count, hipMemcpyDeviceToHost));
    if (e->gpal_idx.p && (e->steps_done & (1 << TM_STEP_PREPARE_PALETTES)))
      TM_HIP(hipMemcpy(pi.data(), e->gpal_idx.as<uint8_t>() + first * 4, (size_t)count * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < count; i++) {
      hdrs[i].UseCount = use[i];
      hdrs[i].TmpIndex = (int32_t)(first + i);
      hdrs[i].MergeIndex = -1;
      hdrs[i].PalIdx_Initial = pi[i];
      hdrs[i].Flags = 1u | 2u | 4u | ((fl[i] & 1) ? 8u : 0u) | ((fl[i] & 2) ? 16u : 0u);  // Active, HasRGB, HasPal, H/V mirror
    }
  }
  return TM_OK;
}

int tm_get_tile(tm_encoder *e, int64_t i, tm_tile_hdr *hdr, uint8_t pal_px[64], uint32_t rgb_px[64]) {
  return tm_get_tiles(e, i, 1, hdr, pal_px, rgb_px);
}

// The tile maps as the reference's consumers read them (Frames[i].TileMap, tilingencoder.pas:178-184, 509-512): the packed 18-byte items are
// put together on the device -- TMI^.PSNR := EuclideanToPSNR(knnErr | mpErr) (1619 / 1644; after PredictMotion alone: of its best error, 1250)
// in double precision there, compared with a tolerance like every PSNR -- and cross PCIe in ONE copy.
__global__ __launch_bounds__(256) void k_pack_tilemap(const int32_t *__restrict__ ti, const int32_t *__restrict__ pi, const uint32_t *__restrict__ er,
                                                      const int8_t *__restrict__ px, const int8_t *__restrict__ py, const uint8_t *__restrict__ pr,
                                                      const uint8_t *__restrict__ ff, int with_psnr, int64_t n, uint32_t *__restrict__ out) {
  __shared__ uint32_t s_items[256 * 18 / 4 + 2];
  const int64_t base = (int64_t)blockIdx.x * 256, i = base + threadIdx.x;
  if (i < n) {
    uint8_t *o = reinterpret_cast<uint8_t *>(s_items) + threadIdx.x * 18;
    const int32_t t = ti[i], p = pi[i];
    float ps = 0.0f;
    if (with_psnr) {  // EuclideanToPSNR, utils.pas:1074-1078
      const float r = (float)((double)er[i] * (1.0 / 192));
      const float m = r > 0.5f ? r : 0.5f;
      ps = (float)(10 * log10(255 * 255 / (double)m));
    }
    const uint8_t f = ff[i];
    const uint32_t fl = (f & 1 ? 1u : 0u) | (f & 2 ? 2u : 0u) | ((pr && pr[i]) ? 4u : 0u);
    memcpy(o, &t, 4); memcpy(o + 4, &p, 4);
    o[8] = (uint8_t)(px ? px[i] : 0); o[9] = (uint8_t)(py ? py[i] : 0);
    memcpy(o + 10, &ps, 4); memcpy(o + 14, &fl, 4);
  }
  __syncthreads();
  const int64_t cnt = min((int64_t)256, n - base);
  const int words = (int)((cnt * 18 + 3) / 4);  // 256 items = 1152 whole words; the last workgroup's tail word is padded inside the staging buffer
  uint32_t *dst = out + base * 18 / 4;          // base * 18 is a multiple of 4
  for (int w = threadIdx.x; w < words; w += 256) dst[w] = s_items[w];
}

int tm_get_tilemaps(tm_encoder *e, int first_frame, int frame_count, tm_tilemap_item *items) {
  TM_CHECK(e && items, TM_E_INVAL, "null argument");
  TM_CHECK((e->steps_done & 1) && first_frame >= 0 && frame_count >= 0 && first_frame + frame_count <= e->nframes, TM_E_INVAL,
           "frame range [%d,+%d) outside 0..%d (or no Load yet)", first_frame, frame_count, e->nframes);
  if (frame_count == 0) return TM_OK;
  TM_HIP(hipSetDevice(e->device));
  const int64_t per = e->tm_size(), off = (int64_t)first_frame * per, n = per * frame_count;
  DevBuf pack;
  TM_TRY(pack.alloc((size_t)n * 18 + 8));
  const bool pm = e->has_pm;
  const uint32_t *er = (pm && !e->reconstructed) ? e->pm_err.as<uint32_t>() : e->tm_err.as<uint32_t>();
  hipLaunchKernelGGL(k_pack_tilemap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, e->tm_tile.as<int32_t>() + off, e->tm_pal.as<int32_t>() + off,
                     er + off, pm ? e->tm_px.as<int8_t>() + off : nullptr, pm ? e->tm_py.as<int8_t>() + off : nullptr,
                     pm ? e->tm_pred.as<uint8_t>() + off : nullptr, e->fflags.as<uint8_t>() + off, (e->reconstructed || pm) ? 1 : 0, n, pack.as<uint32_t>());
  TM_HIP(hipGetLastError());
  TM_HIP(hipMemcpyAsync(items, pack.p, (size_t)n * 18, hipMemcpyDeviceToHost, e->stream));  // page-locked destination: one DMA at PCIe rate
  TM_HIP(hipStreamSynchronize(e->stream));
  return TM_OK;
}

int tm_get_tilemap(tm_encoder *e, int frame, tm_tilemap_item *items) { return tm_get_tilemaps(e, frame, 1, items); }

int tm_get_palette(tm_encoder *e, int i, int32_t *rgb) {
  TM_CHECK(e && rgb, TM_E_INVAL, "null argument");
  TM_CHECK(!e->palettes_host.empty() && i >= 0 && i < e->s.PaletteCount, TM_E_INVAL, "bad palette %d", i);
  memcpy(rgb, &e->palettes_host[(size_t)i * e->s.PaletteSize], (size_t)e->s.PaletteSize * 4);
  return TM_OK;
}

int tm_get_keyframes(tm_encoder *e, int32_t *start_frames) {
  TM_CHECK(e && start_frames, TM_E_INVAL, "null argument");
  TM_TRY(load_tail(e));
  memcpy(start_frames, e->kf_start.data(), e->kf_start.size() * 4);
  return TM_OK;
}

int tm_get_frame_correlations(tm_encoder *e, float *correl) {
  TM_CHECK(e && correl, TM_E_INVAL, "null argument");
  TM_TRY(load_tail(e));
  memcpy(correl, e->correl.data(), e->correl.size() * 4);
  return TM_OK;
}

int tm_get_psnr(tm_encoder *e, double *per_keyframe, double *global_mean) {  // TKeyFrame.LogPSNR, tilingencoder.pas:1006-1028
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  TM_TRY(load_tail(e));
  TM_CHECK(e->reconstructed && e->tm_err.p && !e->kf_start.empty(), TM_E_INVAL, "PSNR: Reconstruct has not been run");
  TM_HIP(hipSetDevice(e->device));
  // ReconstructPSNRCml = sum of the items' PSNR (Single, 1619 / 1644) in a Double (1657); the reference adds in thread order, here
  // in item order.  Per key frame: / (TileMapSize x FrameCount) (1014); all: / (TileMapSize x frames) (1024).
  std::vector<uint32_t> er((size_t)e->q);
  TM_HIP(hipMemcpy(er.data(), e->tm_err.p, (size_t)e->q * 4, hipMemcpyDeviceToHost));
  const int64_t per = e->tm_size();
  double all = 0;
  for (size_t k = 0; k < e->kf_start.size(); k++) {
    const int64_t f0 = e->kf_start[k], f1 = k + 1 < e->kf_start.size() ? e->kf_start[k + 1] : e->nframes;
    double cml = 0;
    for (int64_t i = f0 * per; i < f1 * per; i++) cml += (double)euclidean_to_psnr(er[(size_t)i]);
    if (per_keyframe) per_keyframe[k] = f1 > f0 ? cml / (double)(per * (f1 - f0)) : 0.0;
    all += cml;
  }
  if (global_mean) *global_mean = all / (double)(per * e->nframes);
  return TM_OK;
}

int tm_get_stage_ms(tm_encoder *e, double ms[8]) {
  TM_CHECK(e && ms, TM_E_INVAL, "null argument");
  memcpy(ms, e->stage_ms, sizeof(e->stage_ms));
  return TM_OK;
}

int tm_set_query_shard(tm_encoder *e, int first_frame, int frame_count) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  TM_CHECK(first_frame >= 0, TM_E_INVAL, "bad shard");
  if (first_frame != e->shard_first || frame_count != e->shard_count) e->qf_valid = false;  // prefetched for the old range (freed with the next Load / Reconstruct)
  e->shard_first = first_frame;
  e->shard_count = frame_count;
  return TM_OK;
}

void *tm_get_stream(tm_encoder *e) { return e ? (void *)e->stream : nullptr; }

int tm_set_collective_mode(tm_encoder *e, int stream_ordered) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  e->coll_stream_ordered = stream_ordered != 0;
  return TM_OK;
}

int tm_comm_unique_id(uint8_t id[TM_COMM_ID_BYTES]) {
  TM_CHECK(id, TM_E_INVAL, "null argument");
  static_assert(sizeof(ncclUniqueId) == TM_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId u;
  TM_NCCL(ncclGetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return TM_OK;
}

int tm_comm_init(tm_encoder *e, const uint8_t id[TM_COMM_ID_BYTES], int rank, int world) {
  TM_CHECK(e && id, TM_E_INVAL, "null argument");
  TM_CHECK(world >= 1 && rank >= 0 && rank < world, TM_E_INVAL, "bad process %d of %d", rank, world);
  TM_CHECK(e->comm == nullptr, TM_E_INVAL, "tm_comm_init: this encoder already has a communicator (tm_comm_destroy first)");
  knobs_reload();
  TM_HIP(hipSetDevice(e->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  {
    // Non-blocking: a rank that never arrives (it failed before this call) must end in an error here, not in a wait without end
    // (TM_COMM_TIMEOUT_S seconds, default 120).  Later calls on the communicator go through TM_NCCL, which waits out ncclInProgress.
    ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
    cfg.blocking = 0;
    ncclResult_t r = ncclCommInitRankConfig(&e->comm, world, u, rank, &cfg);
    const double limit = knobs().comm_timeout_s;
    const auto t0 = std::chrono::steady_clock::now();
    while (r == ncclInProgress || (r == ncclSuccess && e->comm)) {
      ncclResult_t st = ncclSuccess;
      const ncclResult_t q = ncclCommGetAsyncError(e->comm, &st);
      if (q != ncclSuccess) { r = q; break; }
      if (st != ncclInProgress) { r = st; break; }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) { r = ncclSystemError; set_error("tm_comm_init: not every one of the %d processes arrived within %.0f s", world, limit); break; }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (r != ncclSuccess) {
      const std::string why = std::string(get_error());
      if (e->comm) { (void)ncclCommAbort(e->comm); e->comm = nullptr; }
      if (why.find("tm_comm_init: not every") == std::string::npos) set_error("ncclCommInitRankConfig failed: %s", ncclGetErrorString(r));
      return TM_E_HIP;
    }
  }
  e->coll_cb = nullptr;
  e->coll_user = nullptr;
  e->coll_stream_ordered = true;
  e->co.rank = rank;
  e->co.world = world;
  e->force_dist = knobs().comm_force_dist;
  bind_native_collectives(e);
  e->dither_rank = rank;
  e->dither_world = world;
  e->qf_valid = false;
  return TM_OK;
}

int tm_get_collective_stats(tm_encoder *e, int64_t calls[4], int64_t *bytes, int reset) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  if (calls) memcpy(calls, e->coll_calls, sizeof(e->coll_calls));
  if (bytes) *bytes = e->coll_bytes;
  if (reset) { memset(e->coll_calls, 0, sizeof(e->coll_calls)); e->coll_bytes = 0; }
  return TM_OK;
}

int tm_comm_destroy(tm_encoder *e) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  if (!e->comm) return TM_OK;
  TM_HIP(hipStreamSynchronize(e->stream));
  TM_NCCLC(e->comm, ncclCommFinalize(e->comm));  // (non-blocking communicator: flush what it still holds, then free it)
  TM_NCCL(ncclCommDestroy(e->comm));
  e->comm = nullptr;
  e->coll_stream_ordered = false;  // a callback installed afterwards gets the default contract: stream drained before, result in place after
  e->force_dist = false;
  e->co = Collectives();
  e->dither_rank = 0;
  e->dither_world = 1;
  e->qf_valid = false;
  return TM_OK;
}

int tm_set_collective(tm_encoder *e, int rank, int world, tm_collective_cb cb, void *user) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  TM_CHECK(world >= 1 && rank >= 0 && rank < world && (cb != nullptr || world == 1), TM_E_INVAL, "bad process %d of %d", rank, world);
  TM_CHECK(e->comm == nullptr, TM_E_INVAL, "tm_set_collective: the encoder has a native communicator (tm_comm_destroy first)");
  e->coll_stream_ordered = false;  // mode 0 until tm_set_collective_mode says otherwise
  e->coll_cb = world > 1 ? cb : nullptr;
  e->coll_user = user;
  e->co.rank = rank;
  e->co.world = world;
  bind_collectives(e);
  e->dither_rank = rank;
  e->dither_world = world;
  e->qf_valid = false;
  return TM_OK;
}

int tm_set_dither_shard(tm_encoder *e, int rank, int world) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  TM_CHECK(world >= 1 && rank >= 0 && rank < world, TM_E_INVAL, "bad dither shard %d of %d", rank, world);
  e->dither_rank = rank;
  e->dither_world = world;
  return TM_OK;
}

int tm_get_device_array(tm_encoder *e, int which, void **ptr, int64_t *count) {
  TM_CHECK(e && ptr && count, TM_E_INVAL, "null argument");
  switch (which) {
    case TM_ARRAY_TILEMAP_TILE: *ptr = e->tm_tile.p; *count = e->q; break;
    case TM_ARRAY_TILEMAP_ERR: *ptr = e->tm_err.p; *count = e->q; break;
    case TM_ARRAY_TILEMAP_PAL: *ptr = e->tm_pal.p; *count = e->q; break;
    case TM_ARRAY_TILEMAP_PRED: *ptr = e->has_pm ? e->tm_pred.p : nullptr; *count = e->has_pm ? e->q : 0; break;
    case TM_ARRAY_TILEMAP_PX: *ptr = e->has_pm ? e->tm_px.p : nullptr; *count = e->has_pm ? e->q : 0; break;
    case TM_ARRAY_TILEMAP_PY: *ptr = e->has_pm ? e->tm_py.p : nullptr; *count = e->has_pm ? e->q : 0; break;
    case TM_ARRAY_PM_ERR: *ptr = e->has_pm ? e->pm_err.p : nullptr; *count = e->has_pm ? e->q : 0; break;
    case TM_ARRAY_TILE_PALPX: *ptr = e->has_pal_px ? e->gpal_px.p : nullptr; *count = e->has_pal_px ? e->t * 64 : 0; break;
    default: set_error("bad array id %d", which); return TM_E_INVAL;
  }
  return TM_OK;
}

int tm_sync_tilemap(tm_encoder *e) {  // after shards were merged: TMI^.PalIdx := FTiles[TileIdx]^.PalIdx_Initial for every item
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  TM_TRY(need(e, TM_STEP_RECONSTRUCT, "Reconstruct"));
  if (e->s.FrameTilingExtendedPaletteUsage) return TM_OK;  // the item's palette is the re-rank's choice: merged like TileIdx (array 2)
  TM_HIP(hipSetDevice(e->device));
  hipLaunchKernelGGL(k_lookup, dim3(gridn(e->q)), dim3(256), 0, e->stream, e->tm_tile.as<int32_t>(), e->q, e->gpal_idx.as<int32_t>(),
                     e->tm_pal.as<int32_t>());
  TM_HIP(hipGetLastError());
  TM_HIP(hipStreamSynchronize(e->stream));
  return TM_OK;
}

int64_t tm_get_knn_queries(tm_encoder *e) { return e ? e->knn_queries : 0; }
int64_t tm_get_dither_pairs(tm_encoder *e) { return e ? e->dither_pairs : 0; }

int tm_get_kmeans_iters(tm_encoder *e, int *tile_iters, int64_t *tile_points, int *pixel_iters, int64_t *pixel_colours, int64_t *pixels, int64_t *pixel_colour_iters) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  if (tile_iters) *tile_iters = e->km_stats.tile_iters;
  if (tile_points) *tile_points = e->km_stats.tile_points;
  if (pixel_iters) *pixel_iters = e->km_stats.pixel_iters;
  if (pixel_colours) *pixel_colours = e->km_stats.pixel_colours;
  if (pixels) *pixels = e->km_stats.pixels;
  if (pixel_colour_iters) *pixel_colour_iters = e->km_stats.pixel_colour_iters;
  return TM_OK;
}

int tm_get_knn_kernel_split(tm_encoder *e, double ms[3], int64_t pairs[3]) {
  TM_CHECK(e && ms && pairs, TM_E_INVAL, "null argument");
  for (int i = 0; i < 3; i++) ms[i] = e->knn_split_ms[i];
  pairs[0] = e->knn_split_pairs[0]; pairs[1] = e->knn_split_pairs[1]; pairs[2] = e->knn_split_pairs[2];
  return TM_OK;
}

int tm_get_knn_stats(tm_encoder *e, double *kernel_ms, int64_t *pairs, int *launches, int *k_bytes, int64_t *db_rows) {
  if (e && db_rows) *db_rows = e->knn_db_rows;
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  if (kernel_ms) *kernel_ms = e->knn_ms;
  if (pairs) *pairs = e->knn_pairs;
  if (launches) *launches = e->knn_launches;
  if (k_bytes) *k_bytes = e->knn_kbytes;
  return TM_OK;
}

int tm_reload_gtm(tm_encoder *e, const char *path) {  // ReloadGTM, tilingencoder.pas:2059 -> LoadStream, 4880-5175
  TM_CHECK(e && path, TM_E_INVAL, "null argument");
  TM_CHECK(e->nframes > 0 && e->width > 0, TM_E_INVAL, "tm_set_video has not been called");
  GtmLoaded g;
  TM_TRY(read_gtm(path, &g));
  // "Mismatch between GTM and loaded video!" (5021-5032)
  TM_CHECK(g.header_frames < 0 || (g.header_frames == e->nframes && g.header_w == e->tm_w * 8 && g.header_h == e->tm_h * 8), TM_E_INVAL,
           "mismatch between GTM (%d frames, %dx%d) and loaded video (%d frames, %dx%d)", g.header_frames, g.header_w, g.header_h, e->nframes,
           e->tm_w * 8, e->tm_h * 8);
  TM_CHECK(g.nframes == e->nframes && g.tm_w == e->tm_w && g.tm_h == e->tm_h, TM_E_INVAL, "GTM stream does not match the loaded video");
  TM_HIP(hipSetDevice(e->device));
  const int64_t q = (int64_t)e->nframes * e->tm_size(), T = (int64_t)g.use.size();
  e->q = q; e->t = T; e->fps = g.fps;
  e->s.PaletteSize = g.pal_size; e->s.PaletteCount = std::max(1, g.pal_count);
  TM_TRY(load_tail(e));  // (not after these lines: a pending tail would put Load's key frames over the stream's)
  e->kf_start = g.kf_start;
  e->correl.assign((size_t)e->nframes, 0.0f);
  e->palettes_host.assign(g.palettes.begin(), g.palettes.end());
  e->palettes_host.resize((size_t)e->s.PaletteCount * e->s.PaletteSize, 0);
  TM_TRY(e->palettes_dev.alloc(e->palettes_host.size() * 4));
  TM_HIP(hipMemcpy(e->palettes_dev.p, e->palettes_host.data(), e->palettes_host.size() * 4, hipMemcpyHostToDevice));
  e->pair_keys_n = 0;
  TM_TRY(e->gtiles.alloc((size_t)std::max<int64_t>(T, 1) * 256)); TM_TRY(e->gpal_px.alloc((size_t)std::max<int64_t>(T, 1) * 64));
  TM_TRY(e->gflags.alloc((size_t)std::max<int64_t>(T, 1))); TM_TRY(e->guse.alloc((size_t)std::max<int64_t>(T, 1) * 4));
  TM_TRY(e->gpal_idx.alloc((size_t)std::max<int64_t>(T, 1) * 4));
  TM_HIP(hipMemset(e->gtiles.p, 0, (size_t)std::max<int64_t>(T, 1) * 256));  // the stream carries no RGB pixels (HasRGBPixels = False, 4937)
  TM_HIP(hipMemset(e->gflags.p, 0, (size_t)std::max<int64_t>(T, 1)));
  TM_HIP(hipMemset(e->gpal_idx.p, 0xff, (size_t)std::max<int64_t>(T, 1) * 4));
  if (T) {
    TM_HIP(hipMemcpy(e->gpal_px.p, g.pal_px.data(), (size_t)T * 64, hipMemcpyHostToDevice));
    TM_HIP(hipMemcpy(e->guse.p, g.use.data(), (size_t)T * 4, hipMemcpyHostToDevice));
  }
  std::vector<int32_t> ti((size_t)q), pi((size_t)q);
  std::vector<uint32_t> er((size_t)q, 0xffffffffu);
  std::vector<int8_t> px((size_t)q), py((size_t)q);
  std::vector<uint8_t> pr((size_t)q);
  e->h_fflags.assign((size_t)q, 0);
  for (int64_t i = 0; i < q; i++) {
    const tm_tilemap_item &it = g.tilemap[(size_t)i];
    ti[(size_t)i] = it.TileIdx; pi[(size_t)i] = it.PalIdx; px[(size_t)i] = it.PredictedX; py[(size_t)i] = it.PredictedY;
    pr[(size_t)i] = (it.Flags & 4) ? 1 : 0;
    e->h_fflags[(size_t)i] = (uint8_t)(it.Flags & 3);
  }
  TM_TRY(e->tm_tile.alloc((size_t)q * 4)); TM_TRY(e->tm_pal.alloc((size_t)q * 4)); TM_TRY(e->tm_err.alloc((size_t)q * 4));
  TM_TRY(e->tm_px.alloc((size_t)q)); TM_TRY(e->tm_py.alloc((size_t)q)); TM_TRY(e->tm_pred.alloc((size_t)q)); TM_TRY(e->pm_err.alloc((size_t)q * 4));
  TM_TRY(e->fflags.alloc((size_t)q));
  TM_HIP(hipMemcpy(e->tm_tile.p, ti.data(), (size_t)q * 4, hipMemcpyHostToDevice));
  TM_HIP(hipMemcpy(e->tm_pal.p, pi.data(), (size_t)q * 4, hipMemcpyHostToDevice));
  TM_HIP(hipMemcpy(e->tm_err.p, er.data(), (size_t)q * 4, hipMemcpyHostToDevice));
  TM_HIP(hipMemcpy(e->pm_err.p, er.data(), (size_t)q * 4, hipMemcpyHostToDevice));
  TM_HIP(hipMemcpy(e->tm_px.p, px.data(), (size_t)q, hipMemcpyHostToDevice));
  TM_HIP(hipMemcpy(e->tm_py.p, py.data(), (size_t)q, hipMemcpyHostToDevice));
  TM_HIP(hipMemcpy(e->tm_pred.p, pr.data(), (size_t)q, hipMemcpyHostToDevice));
  TM_HIP(hipMemcpy(e->fflags.p, e->h_fflags.data(), (size_t)q, hipMemcpyHostToDevice));
  e->has_pm = true;
  e->has_pal_px = true;
  e->reconstructed = false;  // PSNR is not in the stream
  e->gtiles_have_rgb = false;
  e->drop_prefetch();
  // every step's product the stream holds is in place: Save, Reindex and the read-back views work.  Steps that compute from the frame
  // tiles or from RGB pixels check for them (need_frame_tiles / need_global_rgb) and ask for Load / Reduce when they are missing.
  e->steps_done = 0xff;
  return TM_OK;
}

int tm_generate_y4m(tm_encoder *e, const char *path, int input) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  return generate_y4m(e, path, input != 0);
}

int tm_generate_pngs(tm_encoder *e, int input) {
  TM_CHECK(e, TM_E_INVAL, "null encoder");
  return generate_pngs(e, input != 0);
}

int tm_save_gtm(tm_encoder *e, const char *path) {
  TM_CHECK(e && path, TM_E_INVAL, "null argument");
  return save_to(e, path);
}

}  // extern "C"
