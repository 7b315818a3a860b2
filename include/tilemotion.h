/*
 * tilemotion.h -- C ABI of libtilemotion.so: the MI355X (gfx950) implementation of the TileMotion
 * encoder's per-frame tile pipeline (gligli/tiler: tilingencoder.pas / utils.pas / extern.pas).
 *
 * Plain pointers and sizes only; every function returns 0 on success or a negative TM_E_* code, never
 * throws or aborts across the boundary.  tm_last_error() gives the message of the calling thread's last
 * failure.  There is NO CPU fallback: without a usable HIP device every compute entry point fails with
 * TM_E_NODEVICE.
 *
 * Three layers, each replacing a reference seam (file:line in the reference tree):
 *   1. Coarse seam  tm_encoder_*  == TTilingEncoder's public surface (tilingencoder.pas:486-568):
 *      Create/Destroy, settings properties (3745-3770, clamps 2919-3047), frame callback contract of
 *      TFFMPEGFrameCallback (extern.pas:149), Run(step) with TEncoderStep values (tilingencoder.pas:18),
 *      read-only Tiles/Frames/Palettes views (509-512).
 *   2. Stage seam   tm_stage_*    == the per-step workers behind Run, on DEVICE pointers, so a host that
 *      owns several processes/GPUs (bench.py, tiler_amd.distributed) can put RCCL collectives between them.
 *   3. Fine seam    ann_kdtree_* / yakmo_* / bico_*  == the DLL imports of extern.pas:178-223, same
 *      per-call semantics, plus *_batch twins (per-call GPU use is latency bound; kept for compatibility).
 *
 * Several GPUs: ONE PROCESS PER GPU, and only that.  The surveyed boundary sketched a tm_set_device_mask for one process driving 1/2/4/8
 * devices; it is not built and will not be: an encoder is bound to one device (tm_set_device) and one host thread, N encoders become one
 * job through tm_comm_init (RCCL inside the library) or tm_set_collective (the host's own communicator), and the reference's single
 * control thread (tiler.lpr:64-70) starts N copies of itself with a rank each.  One process per device is what
 * RCCL and the driver's launcher (torch.distributed.run) assume, it keeps a fault or an out-of-memory on one device from taking the
 * other seven encoders with it, and the steps' host tails (OptimizePalettes, the key-frame logic, LZMA) run N times in parallel
 * instead of queueing on one thread.
 *
 * Environment switches.  All are optional; they are sampled at the API boundary (tm_create, tm_run, every tm_stage_* and fine-seam entry)
 * and never read inside a step.  Set and not "0" = on.
 *   TM_KNN_DEBUG            one line per search on stderr: the three kernels' times, pairs evaluated, list sizes, matrix instructions
 *   TM_KNN_NOPRUNE          the nearest-neighbour scan evaluates every (query, row) pair (bench.py's dense diagnostic launch)
 *   TM_KNN_ARENA_ENTRIES=<n> first size of the scan's tile-list arena (tests: a tiny one, so that a search is repeated with the counted size)
 *   TM_TOPK_BRUTE           the k-nearest search by the VALU brute force (tests compare the pruned scan with it)
 *   TM_EPU_TABLE_GIB=<x>    above this size the (tile, palette) feature table is not built, the pairs asked for are (default 6)
 *   TM_NO_QUERY_GROUPS      Reconstruct searches once per tile-map item instead of once per distinct frame tile (tests)
 *   TM_DITHER_OWN_KEYS      Dither collects its (palette, colour) pairs itself instead of taking PreparePalettes' keys (tests)
 *   TM_DITHER_NO_DEDUP      Dither plans every pixel on its own (tests); TM_DITHER_LITERAL: every tile through the literal-sort kernel
 *   TM_DEDUP_PLAIN          exact dedup by the comparator sort alone; TM_DEDUP_SORT: equal rows grouped by the radix sort of their hashes
 *                           (the front end of rounds 1-4) instead of the hash table; TM_DEDUP_RADIX_MIN=n: the distinct rows go into content
 *                           order by a radix sort of their 8-byte prefixes (whole rows compared only within short runs of equal prefixes;
 *                           the comparator merge sort where a run is long) from n rows on (default 2^20; tests: 1); TM_DEDUP_DEGRADE_HASH: a 2-bit hash, so that every group
 *                           collides (tests); TM_DEDUP_FULL_ORDER: the whole order, not only the rows that can survive the budget (tests)
 *   TM_MOTION_VALU          the motion search's VALU kernel only (tests drive both kernels)
 *   TM_FEATURES_PLAIN       the int16 DCT features sum every coefficient in the reference's order (no separable first look; the tests
 *                           compare the two forms)
 *   TM_KM_LAUNCHES          the tile -> palette k-means runs its skipping iterations as three launches each instead of one resident launch
 *                           for all of them (tests compare the two)
 *   TM_KMODES_BINWISE       every k-modes iteration bin by bin (two launches per 960 points) -- without the leg that scores all remaining points
 *                           at once and walks the bins in one launch for as long as no mode changes (tests compare the two);
 *                           TM_KMODES_FAST_ALWAYS: that leg is tried in every iteration after the first, also behind an iteration that moved many
 *                           points (tests: its stops and the hand-over to the bin-by-bin launches in the middle of an iteration)
 *   TM_FEATURES_BY_TILE     the int16 features of RGB tiles by the tile-at-a-time kernel (k_features_i16<0>) instead of eight tiles a wave
 *                           (k_features_tiles8; tests compare the two)
 *   TM_MOTION_PACK_SEPARATE the encoder's motion search as three launches per frame (int16 window features, their packing, the search) instead of
 *                           window features made in the search's layout at once (A/B runs, tests)
 *   TM_MOTION_FORCE_FLAG    treat every frame as beyond the matrix search's exact range: the fallback (int16 windows + VALU search) runs (tests)
 *   TM_TOPK_ESTIMATE        0: the k-nearest search never takes its first thresholds from a sample of the database (the curve window's bound instead);
 *                           1: whenever the database has rows enough for a sample (default: many queries against >= 16 384 rows)
 *   TM_KM_RESIDENT_FAIL     the resident launch of the tile k-means is treated as if its barrier had given up: the clustering is repeated from its
 *                           seeds through the launches (tests: the fallback's result must be the same)
 *   TM_WINDOW_DCTS_BY_TILE  the sliding-window features of motion prediction a window at a time (k_features_i16<2>) instead of by strips that share
 *                           the colour conversion and the row transforms between windows (tests compare the two)
 *   TM_PP_SHARDED           several processes: the tile -> palette clustering stays data-parallel (an all-reduce per Lloyd iteration) even where
 *                           every process could run it whole in one resident launch (tests, A/B)
 *   TM_PP_DEBUG             PreparePalettes prints its sub-steps' wall times (adds synchronisations)
 *   TM_COMM_FORCE_DIST      a one-process communicator still walks the sharded code paths (tests on a one-GPU box)
 *   TM_COMM_TIMEOUT_S=<s>   how long tm_comm_init (and a collective of the library's own communicator) waits for the other processes (120)
 *   TM_POOL_GIB=<x>         cap of the device-memory pool a thread keeps (96); TM_HOST_THREADS=<n>: OptimizePalettes' helper threads
 *                           (both read once per process)
 */
#ifndef TILEMOTION_H
#define TILEMOTION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TM_API __attribute__((visibility("default")))

enum {
  TM_OK = 0,
  TM_E_INVAL = -1,     /* bad argument / bad state (the reference would Assert) */
  TM_E_NODEVICE = -2,  /* no HIP device, or kernels for gfx950 cannot run here */
  TM_E_HIP = -3,       /* a HIP runtime call failed */
  TM_E_NOMEM = -4,
  TM_E_IO = -5,
  TM_E_UNSUPPORTED = -6 /* reference feature outside the hot path */
};

/* TEncoderStep, tilingencoder.pas:18 */
enum { TM_STEP_ALL = -1, TM_STEP_LOAD = 0, TM_STEP_PREDICT_MOTION = 1, TM_STEP_REDUCE = 2, TM_STEP_PREPARE_PALETTES = 3,
       TM_STEP_DITHER = 4, TM_STEP_RECONSTRUCT = 5, TM_STEP_REINDEX = 6, TM_STEP_SAVE = 7 };

/* TPsyVisMode, tilingencoder.pas:21 */
enum { TM_PVS_DCT = 0, TM_PVS_WEIGHTED_DCT = 1, TM_PVS_WAVELETS = 2, TM_PVS_SPE_DCT = 3, TM_PVS_WEIGHTED_SPE_DCT = 4 };

#define TM_NULL_COLOR ((int32_t)0xffff00ff) /* cDitheringNullColor, utils.pas:45 */

/* TTile header, packed, 20 bytes (tilingencoder.pas:116-121).  Flags bits: 0 Active, 1 HasRGBPixels,
 * 2 HasPalPixels, 3 HMirror_Initial, 4 VMirror_Initial (FPC set, 4 bytes). */
#pragma pack(push, 1)
typedef struct { uint32_t UseCount; int32_t TmpIndex; int32_t MergeIndex; int32_t PalIdx_Initial; uint32_t Flags; } tm_tile_hdr;
/* TTileMapItem, packed, 18 bytes (tilingencoder.pas:178-184).  Flags bits: 0 HMirror, 1 VMirror, 2 Predicted. */
typedef struct { int32_t TileIdx; int32_t PalIdx; int8_t PredictedX; int8_t PredictedY; float PSNR; uint32_t Flags; } tm_tilemap_item;
#pragma pack(pop)

TM_API const char *tm_last_error(void);
TM_API int tm_device_count(void);           /* usable gfx950 devices, 0 if none */
TM_API const char *tm_version(void);
/* The peaks a roofline divides by, measured on this device: a bare loop of the int8 MFMA the KNN kernel issues
 * (two waves per SIMD, operands in registers; about `seconds_hint` seconds) in TOP/s, and a stream triad a = b + s c over three
 * arrays of `bytes_per_array` in GB/s.  Diagnostics for bench.py; nothing in the product path calls them. */
TM_API int tm_probe_mfma_i8(double seconds_hint, double *tops);
TM_API int tm_probe_hbm_triad(int64_t bytes_per_array, double *gb_per_s);

/* ======================================================================================= coarse seam */
typedef struct tm_encoder tm_encoder;
typedef void (*tm_progress_cb)(void *user, int step, int position, int max, int hourglass); /* OnProgress, :304 */

TM_API tm_encoder *tm_create(void);                 /* TTilingEncoder.Create, :5484; NULL on failure */
TM_API void tm_destroy(tm_encoder *);               /* Destroy, :5516 */
TM_API int tm_set_device(tm_encoder *, int device); /* which HIP device this encoder (process) drives */
/* Settings: keys are the INI names of SaveSettings (:3745-3770); setters clamp like :2919-3047. */
TM_API int tm_load_default_settings(tm_encoder *);  /* LoadDefaultSettings, :3817-3845 */
TM_API int tm_load_settings_ini(tm_encoder *, const char *path); /* LoadSettings, :3777 */
TM_API int tm_save_settings_ini(tm_encoder *, const char *path); /* SaveSettings, :3738-3775: the text a .gtm embeds (:5331-5335), CR LF line ends */
/* LoadSettings followed by SaveSettings on text (host only, no device): `ini_text` through the setters' clamps, back as the INI text
 * SaveSettings would write; *out_len = its length, `out` (may be NULL) receives up to cap - 1 bytes and a terminator. */
TM_API int tm_settings_text_host(const char *ini_text, char *out, int64_t cap, int64_t *out_len);
TM_API int tm_set_int(tm_encoder *, const char *key, int64_t v);
TM_API int tm_set_float(tm_encoder *, const char *key, double v);
TM_API int tm_set_bool(tm_encoder *, const char *key, int v);
TM_API int tm_set_str(tm_encoder *, const char *key, const char *v);
TM_API int tm_get_int(tm_encoder *, const char *key, int64_t *v);
TM_API int tm_get_float(tm_encoder *, const char *key, double *v);
TM_API int tm_set_progress_cb(tm_encoder *, tm_progress_cb cb, void *user);
/* Video: what FFMPEG_Open + ReframeUI + InitFrames establish (:1772-1776, :2631, :2661). */
TM_API int tm_set_video(tm_encoder *, int width, int height, double fps, int frame_count);
/* One decoded frame, AV_PIX_FMT_RGB32 (uint32 0xAARRGGBB), as TFFMPEGFrameCallback hands it (extern.pas:149);
 * read during the call only.  stride_px = pixels per row. */
TM_API int tm_push_frame_rgb32(tm_encoder *, int index, const uint32_t *pixels, int stride_px);
/* Same, but the frames already sit in device memory as [frame_count][height][width] uint32 (bench path). */
TM_API int tm_set_frames_device(tm_encoder *, const void *dev_frames);
/* Same, with the whole clip in HOST memory as [frame_count][height][width] uint32 (the batch form of the frame callback for a
 * host that holds the decoded clip).  The next Load moves it across PCIe in chunks, each chunk's copy running beside the Load
 * kernel of the chunk before; page-locked memory makes the copies asynchronous.  The clip is BORROWED until that Load has
 * returned; from then on the encoder reads its own device copy (a later Run(esLoad) without new frames reads that copy), and
 * the host may free or reuse the memory. */
TM_API int tm_set_frames_host(tm_encoder *, const uint32_t *host_frames);
/* Start moving the clip the NEXT Load will read while the current clip's steps still run (a second device buffer, the copy
 * stream): call it before tm_run of the current clip, then tm_set_frames_host with the same pointer before the next one -- that
 * Load adopts the copies instead of issuing its own, so back-to-back clips pay PCIe beside the compute, not before it.
 * Borrowed until the adopting Load has returned.  At most one clip can wait beside the one in flight (TM_E_INVAL otherwise);
 * with both buffers taken the clip of the LAST Load gives way, after which a Load without new frames fails ("no frames"). */
TM_API int tm_prefetch_frames_host(tm_encoder *, const uint32_t *host_frames);
TM_API int tm_run(tm_encoder *, int step);          /* Run(AStep), :5529-5554; blocking */
/* read-back views (copy-out) */
TM_API int tm_get_counts(tm_encoder *, int64_t *tiles, int *frames, int *palettes, int *tm_w, int *tm_h, int *keyframes);
TM_API int tm_get_tile(tm_encoder *, int64_t i, tm_tile_hdr *hdr, uint8_t pal_px[64], uint32_t rgb_px[64]);
TM_API int tm_get_tiles(tm_encoder *, int64_t first, int64_t count, tm_tile_hdr *hdrs, uint8_t *pal_px, uint32_t *rgb_px);
TM_API int tm_get_tilemap(tm_encoder *, int frame, tm_tilemap_item *items /* tm_w*tm_h */);
/* Frames[first_frame .. first_frame+frame_count-1].TileMap in one call (tilingencoder.pas:178-184, 509-512): the packed 18-byte items are
 * put together on the device and cross PCIe in one copy (at the link's rate when `items` is page-locked memory); the per-frame form above is
 * this with frame_count = 1. */
TM_API int tm_get_tilemaps(tm_encoder *, int first_frame, int frame_count, tm_tilemap_item *items /* frame_count*tm_w*tm_h */);
TM_API int tm_get_palette(tm_encoder *, int i, int32_t *rgb /* PaletteSize */);
TM_API int tm_get_keyframes(tm_encoder *, int32_t *start_frames /* keyframes */);
TM_API int tm_get_frame_correlations(tm_encoder *, float *correl /* frames */);
/* TKeyFrame.LogPSNR (:1006-1028): mean "PSNR-HVS (by tile)" of every key frame (the items' PSNR summed in a Double, divided by tile-map
 * size x frames of the key frame) and of the whole clip; what the reference prints after Reconstruct.  Either pointer may be NULL. */
TM_API int tm_get_psnr(tm_encoder *, double *per_keyframe /* keyframes */, double *global_mean);
TM_API int tm_get_stage_ms(tm_encoder *, double ms[8]); /* wall ms of the last run of each step (ProgressRedraw, :3925) */
TM_API int tm_save_gtm(tm_encoder *, const char *path);  /* Save, :2040 -> SaveStream, :5177 */
/* The same writer on HOST arrays (no device needed): tiles in their final (Reindex) order with use counts, palettes
 * [pal_count][pal_size], tile maps [nframes][tm_h*tm_w].  What SaveStream reads from FTiles/FPalettes/FFrames. */
TM_API int tm_write_gtm_host(const char *path, int tm_w, int tm_h, int nframes, double fps, const int32_t *kf_start, int nkf,
                             const uint8_t *pal_px, const uint32_t *use, int64_t ntiles, const int32_t *palettes, int pal_count,
                             int pal_size, const tm_tilemap_item *tilemap, const char *settings_text);
/* LZCompress, extern.pas:420-439 (LZMA-alone: lc 8, lp 0, pb 2, 4 MiB dictionary, unknown size, end marker), host
 * buffers.  *out_n = compressed size; TM_E_INVAL (with *out_n set) when cap is too small. */
TM_API int tm_lz_compress_host(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_n);
/* LZDecompress, extern.pas:441-458: one stream; *out_n = decoded size (set even when cap is too small -> TM_E_INVAL),
 * *consumed (optional) = bytes of src the stream occupied, so that the next key frame's stream can follow. */
TM_API int tm_lz_decompress_host(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_n, size_t *consumed);
/* GenerateY4M (:2126-2199) and GeneratePNGs (:2075-2124): the frames as Render draws them with the constructor's defaults
 * (predicted items copied from the previous output frame, tiles through the item's palette and mirrors; :3573-3640, :5505-5507), or the
 * source frames (input != 0).  Y4M: 'YUV4MPEG2 W.. H.. F..:1000000 Ip C444', full-resolution Y, U + 128, V + 128 planes from RGBToYUV
 * (utils.pas:478-490), rounded and clamped.  PNGs: <OutputFileName without extension>_NNNN.png (24-bit RGB) + <...>.txt with the
 * palettes, one 'FFBBGGRR' line per colour.  Host code (export tooling). */
TM_API int tm_generate_y4m(tm_encoder *, const char *path, int input);
TM_API int tm_generate_pngs(tm_encoder *, int input);
/* ReloadGTM, :2059 -> LoadStream, :4880-5175: replaces the encoder's tiles (palette indices only), palettes, tile maps and key
 * frames with the file's; the video set with tm_set_video must match the file's header (:5021-5032) or TM_E_INVAL comes back.
 * Afterwards the read-back views and tm_save_gtm work on the loaded state. */
TM_API int tm_reload_gtm(tm_encoder *, const char *path);
/* Multi-GPU (one process per GPU): this process matches only frames [first, first+count) in Reconstruct (frames are
 * independent in the KNN branch, DoXY :1464); the host then merges the per-frame results of all processes with an
 * all-reduce(MAX) over the arrays below (other shards hold -1) and calls tm_sync_tilemap before Reindex.
 * PredictMotion honours the same range (its frames are independent, :1982-1985); with motion prediction on, the range
 * given for Reconstruct must start on a key frame (frames chain inside a key frame, :1496). */
enum { TM_ARRAY_TILEMAP_TILE = 0, TM_ARRAY_TILEMAP_ERR = 1, TM_ARRAY_TILEMAP_PAL = 2,
       /* with motion prediction: uint8 IsPredicted, int8 PredictedX, int8 PredictedY (1 byte per item; other shards hold 0: merge with SUM) */
       TM_ARRAY_TILEMAP_PRED = 3, TM_ARRAY_TILEMAP_PX = 4, TM_ARRAY_TILEMAP_PY = 5,
       TM_ARRAY_PM_ERR = 6 /* uint32 best error of PredictMotion per item; other shards hold 0 */,
       TM_ARRAY_TILE_PALPX = 7 /* uint8 [tiles][64] dithered palette indices (DitherTile's output, :2690-2724); tiles of other
                                  dither shards hold 0: merge with SUM, as bytes or as count/4 32-bit words (count is a multiple of 64) */ };
TM_API int tm_set_query_shard(tm_encoder *, int first_frame, int frame_count /* <0: to the end */);
/* One process per GPU with the merges INSIDE the steps: the host hands the encoder its rank, the number of processes and a
 * callback that performs a collective over them (the host's own communicator -- RCCL through torch.distributed in bench.py, gloo in the
 * CPU tests; the library ALSO links RCCL and can carry the collectives itself, tm_comm_init below, which bench.py takes with
 * TM_BENCH_NATIVE=1).  The callback runs on the caller's thread with the encoder's stream idle; it returns 0 once the result is in place.
 *   kind: TM_COLL_ALLREDUCE_SUM_I32 / _MAX_I32 / _SUM_I64: `count` elements in `dev_buf`, in place;
 *         TM_COLL_ALLGATHER_BYTES: `count` bytes from `dev_buf` of every process into `dev_recv` (world x count bytes, rank order).
 * With it set, Run(step) shards by itself: Load by frame (motion prediction off), Reduce as a local exact dedup + an all-gather of
 * the distinct tiles + a dedup of the union, PreparePalettes as data-parallel Lloyd (all-reduce of the integer sums per
 * iteration) and palette-parallel colour quantisation, Dither by global tile, Reconstruct with the database rows built per share
 * and all-gathered and the query frames of tm_set_query_shard.  Every process ends each step with the same global tiles,
 * palettes and merged tile maps as a single-process run. */
enum { TM_COLL_ALLREDUCE_SUM_I32 = 0, TM_COLL_ALLREDUCE_MAX_I32 = 1, TM_COLL_ALLREDUCE_SUM_I64 = 2, TM_COLL_ALLGATHER_BYTES = 3 };
typedef int (*tm_collective_cb)(void *user, int kind, void *dev_buf, void *dev_recv, int64_t count);
TM_API int tm_set_collective(tm_encoder *, int rank, int world, tm_collective_cb cb, void *user);
/* The HIP stream (hipStream_t) every step of this encoder is queued on, and the callback's contract with it.  Mode 0 (default): the
 * library drains the stream before each callback and the callback returns with the result in place (any communicator, any stream).
 * Mode 1, stream-ordered: the callback ENQUEUES the collective on that stream (RCCL: ncclAllReduce(..., stream), or a
 * torch.distributed call made with the stream current) and returns at once; the library neither drains before nor waits after --
 * the ~80 collectives of a step (one per Lloyd iteration among them) then cost no host round trip each. */
TM_API void *tm_get_stream(tm_encoder *);
TM_API int tm_set_collective_mode(tm_encoder *, int stream_ordered);
/* The native form of the above -- what a FreePascal host needs for N GPUs and nothing else: RCCL is linked into the library, one
 * process per GPU.  One process obtains an id (tm_comm_unique_id = ncclGetUniqueId), hands its 128 bytes to the others by any
 * means (a file, an environment variable, a pipe), and every process calls tm_comm_init(enc, id, rank, world) on an encoder whose
 * device has been chosen (tm_set_device).  From then on Run(step) shards and merges as described for tm_set_collective, with the
 * four collective kinds issued as ncclAllReduce / ncclAllGather on the encoder's own stream: no callback, no host round trip.
 * tm_comm_init is collective (ncclCommInitRank: it returns once all `world` processes have called it).  Sits where the
 * reference's Run (tilingencoder.pas:5529-5554) sits: the host's code above it does not change with the number of GPUs. */
#define TM_COMM_ID_BYTES 128
TM_API int tm_comm_unique_id(uint8_t id[TM_COMM_ID_BYTES]);
TM_API int tm_comm_init(tm_encoder *, const uint8_t id[TM_COMM_ID_BYTES], int rank, int world);
TM_API int tm_comm_destroy(tm_encoder *);
/* Collectives this process has issued since the last reset, by kind (index = TM_COLL_*), and the bytes it put through them
 * (all-reduce: the buffer; all-gather: world x the piece) -- whichever of the two paths above carries them. */
TM_API int tm_get_collective_stats(tm_encoder *, int64_t calls[4], int64_t *bytes, int reset);
/* Dither (DoDither :1873-1907, one independent DitherTile per global tile): this process dithers tiles
 * [T * rank / world, T * (rank + 1) / world) only (T = global tiles after Reduce) and zeroes the rest; the host merges
 * TM_ARRAY_TILE_PALPX with an all-reduce(SUM) before Reconstruct.  (0, 1) = every tile (default). */
TM_API int tm_set_dither_shard(tm_encoder *, int rank, int world);
TM_API int tm_get_device_array(tm_encoder *, int which, void **dev_ptr, int64_t *count /* elements: int32 for 0-2 and 6, bytes for 3-5 and 7 */);
TM_API int tm_sync_tilemap(tm_encoder *);
/* device time (HIP events on the encoder's stream) of the KNN distance kernel over the last Reconstruct */
/* pairs = (query, distinct database row) pairs the kernel evaluated; db_rows = distinct rows searched (<= global tiles) */
TM_API int tm_get_knn_stats(tm_encoder *, double *kernel_ms, int64_t *pairs, int *launches, int *k_bytes, int64_t *db_rows);
/* the same launches kernel by kernel (the scan is three kernels): device ms of k_knn_seed / k_knn_lists / k_knn_consume,
 * the (query, row) pairs the seed and the consume kernel evaluated, and the 32x32x32 int8 matrix instructions the consume kernel issued for them
 * (a chain has 6 + HT + HQ + min(HT, HQ); products with an all-zero high-digit chunk are skipped); kernel_ms above is the sum of the three
 * times, pairs the sum of the two counts */
TM_API int tm_get_knn_kernel_split(tm_encoder *, double ms[3], int64_t pairs[3]);
/* queries of the last Reconstruct's searches: the DISTINCT frame tiles when Reduce's exact groups can be used (one process, motion
 * prediction off), every tile-map item otherwise */
TM_API int64_t tm_get_knn_queries(tm_encoder *);
/* the last Dither: the distinct (palette, colour) pairs it planned once each (pixels look their pair up), 0 when every pixel was planned on
 * its own (few duplicates, the Yliluoma ditherer, few tiles for the number of palettes) */
TM_API int64_t tm_get_dither_pairs(tm_encoder *);
/* What the last PreparePalettes ran through (single process): Lloyd iterations and points of the tile -> palette clustering
 * (DoPalettization, tilingencoder.pas:4105-4245; yakmo's cap is cYakmoMaxIterations = 300) and, for the colour quantisation
 * (QuantizeUsingYakmo, :4434-4532), the iterations of the slowest palette, the distinct colours clustered, the pixels they stand for and
 * the sum over the palettes of (distinct colours x iterations). */
TM_API int tm_get_kmeans_iters(tm_encoder *, int *tile_iters, int64_t *tile_points, int *pixel_iters, int64_t *pixel_colours, int64_t *pixels,
                               int64_t *pixel_colour_iters);

/* ======================================================================================= stage seam
 * All pointers are DEVICE pointers unless named host_*.  `stream` is a hipStream_t (NULL = default stream).
 * Tiles are [n][64] uint32 0x00BBGGRR in the reference's canonical (mirrored) orientation. */

/* A1+A2+A3: TFrame.LoadFromImage (:1293) + PrepareInterFrameData (:1329) + mirror canonicalisation (:1393-1411).
 * frames: [nframes][img_h][img_w] RGB32.  Outputs: tiles [nframes*tm_w*tm_h][64], flags u8 (bit0 H, bit1 V),
 * lab_means f32 [ntiles][3]. */
TM_API int tm_stage_load(const void *frames, int nframes, int img_w, int img_h, int tm_w, int tm_h,
                         void *tiles, void *flags, void *lab_means, void *stream);

/* RGBToLAB (utils.pas:374-410) of n colours 0x00RRGGBB -> float [n][3] (L, a, b): the colour conversion the load and feature kernels
 * share, as an operator of its own (the whole 24-bit domain is checked against the oracle through it). */
TM_API int tm_stage_rgb_to_lab(const void *rgb, int64_t n, void *out_lab, void *stream);

/* A4+A5: ConvertToCpnPixels (:3049) + ComputeCpnPixelsPsyVisFeatures (:3103) -> int16 [n][192].
 * mirror_flags may be NULL (no un-mirroring). */
TM_API int tm_stage_features_rgb(const void *tiles, int64_t n, const void *mirror_flags, int mode, int use_lab,
                                 void *out_i16, void *stream);
/* FromPal=True variant (PrepareReconstruct.DoPsyV, :4570-4583): pal_px u8 [n][64], pal_idx i32 [n],
 * palettes i32 [npal][pal_size]. */
TM_API int tm_stage_features_pal(const void *pal_px, const void *pal_idx, int64_t n, const void *palettes, int pal_size,
                                 int mode, void *out_i16, void *stream);
/* A6 as used by DoPalettization (:4126,:4160): double DCT with UseLAB, Round()ed to int32 [n][192]. */
TM_API int tm_stage_features_cluster(const void *tiles, int64_t n, int mode, void *out_i32, void *stream);

/* FrameTilingExtendedPaletteUsage (:1559-1610).
 * ann_kdtree_short_search_multi(k, eps 0) for a batch: the k nearest database rows of every query by true L2, ordered by
 * (distance, index); out_idx i32 [nq][k] (-1 pads a database smaller than k), out_err u32 [nq][k]. */
TM_API int tm_stage_knn_topk(const void *queries_i16, int64_t nq, const void *db_i16, int64_t nt, int k, void *out_idx, void *out_err,
                             void *stream);
/* The re-rank: every unique tile of knn_idx[q][] x every unique palette of those tiles (tile_pal_idx = PalIdx_Initial),
 * distance = CompareEuclideanDCTPtr_asm as written (utils.pas:559-725); first strict minimum in ascending (tile, palette)
 * order.  out_tile / out_pal i32 [nq], out_err u32 [nq].  Blocking (builds and frees the ntiles x npal feature table). */
TM_API int tm_stage_epu_rerank(const void *queries_i16, int64_t nq, const void *knn_idx, int k, const void *pal_px, const void *tile_pal_idx,
                               int64_t ntiles, const void *palettes, int npal, int pal_size, void *out_tile, void *out_pal, void *out_err,
                               void *stream);

/* Motion prediction (PredictMotion :1154-1282 and the redo in Reconstruct :1496-1532).
 * DoDCTs: pvsWeightedDCT features of every 8x8 window of a frame buffer [height][width] u32 0x00BBGGRR
 * -> int16 [(height-7)*(width-7)][192], row-major over window positions. */
TM_API int tm_stage_window_dcts(const void *frame_buffer, int width, int height, void *out_i16, void *stream);
/* DoXY search: cur = features of the frame's tiles in ORIGINAL orientation [tm_h*tm_w][192]; window_dcts of the
 * previous frame buffer (tm_w*8 x tm_h*8); radius = MotionPredictRadius (1..128, decremented inside like :1271).
 * Error = CompareEuclideanDCTPtr_asm as written (utils.pas:559-725, entry xmm7 = 0) + manhattan distance; first strict
 * minimum in raster order.  out_err u32, out_px / out_py int8 (PredictedX / PredictedY). */
TM_API int tm_stage_motion_search(const void *cur_i16, int tm_w, int tm_h, const void *window_dcts, int radius, void *out_err,
                                  void *out_px, void *out_py, void *stream);

/* A13+A14 (KNN branch): exact nearest neighbour of every query in the database, both int16 [.][192];
 * what ann_kdtree_short_search(eps=0) answers (:1547).  Ties: lowest database index.
 * out_idx i32 [nq], out_err u32 [nq].  Blocking (needs one 1.5 KB read-back for the digit plan). */
TM_API int tm_stage_knn(const void *queries_i16, int64_t nq, const void *db_i16, int64_t nt,
                        void *out_idx, void *out_err, void *stream);
/* Same with a prepared database (ann_kdtree_create analogue): pack once, search many query batches. */
typedef struct tm_knn_index tm_knn_index;
TM_API tm_knn_index *tm_knn_index_create(const void *db_i16, int64_t nt, void *stream);
TM_API void tm_knn_index_destroy(tm_knn_index *);
TM_API int tm_knn_index_search(tm_knn_index *, const void *queries_i16, int64_t nq, void *out_idx, void *out_err, void *stream);
/* measured device time (ms, HIP events on `stream`) of the distance kernel in the last search, and its MFMA K */
TM_API int tm_knn_index_last_stats(tm_knn_index *, double *kernel_ms, int *k_bytes, int64_t *pairs);
/* diagnostics (tests): the digit plan of the calling thread's last scan -- 32-column chunks that carry a high digit on the database / query
 * side, 0..6 each, and whether it was the k-nearest collection: together they name the instantiation of the scan's kernels that ran -- and
 * how often a scan of this process has been repeated with a larger tile-list arena (TM_KNN_ARENA_ENTRIES sets the first size) */
TM_API int tm_knn_last_plan(int *ht, int *hq, int *topk, int64_t *arena_retries);

/* A12: Dither (:1873) = PreparePlan (:2268) + DitherTile (:2688) for every tile.  tiles/flags as above,
 * pal_idx i32 [n], palettes i32 [npal][pal_size] -> pal_px u8 [n][64] (canonical orientation). */
TM_API int tm_stage_dither(const void *tiles, const void *flags, const void *pal_idx, int64_t n, const void *palettes,
                           int npal, int pal_size, int use_thomas_knoll, int y2_mixed_colors, void *out_pal_px, void *stream);

/* A8/A16: MakeTilesUnique (:4720) + ReindexTiles (:4626) on n rows of `row_bytes` (256: RGB dwords compared as
 * unsigned dwords; 64: palette indices compared as bytes).  use_in u32[n] or NULL (=1).
 * Outputs: remap i32 [n] (final index of each row's representative, -1 if its use count is 0),
 * order i32 [n] (first *n_unique entries: original index of the representative at each final position),
 * use_out u32 [n].  Blocking. */
TM_API int tm_stage_dedup(const void *rows, int64_t n, int row_bytes, const void *use_in,
                          void *remap, void *order, void *use_out, int64_t *host_n_unique, void *stream);

/* A9/A10: the build's deterministic k-means (farthest-first init, exact integer sums); pts i32 [n][d],
 * weights u32 [n] or NULL.  assign i32 [n], centroids f64 [k][d] (device).  Returns live centroid count in *host_k. */
TM_API int tm_stage_kmeans(const void *pts_i32, const void *weights, int64_t n, int d, int k, int max_iter,
                           void *assign, void *centroids, int *host_k, int *host_iters, void *stream);
/* The same Lloyd iterations from the caller's own initial centres: host_init_idx[k] = indices of the points to start from (-1: none)
 * instead of the farthest-first picks -- the seam an experiment with another seeding rule (k-means++ and the like) goes through;
 * yakmo's own k-means++ draws cannot be reproduced. */
TM_API int tm_stage_kmeans_seeded(const void *pts_i32, const void *weights, int64_t n, int d, int k, const int64_t *host_init_idx, int max_iter,
                                  void *assign, void *centroids, int *host_k, int *host_iters, void *stream);
/* QuantizeUsingYakmo + DoQuantization (:4434-4564) for every palette at once: pixels of tiles grouped by pal_idx. */
TM_API int tm_stage_quantize_palettes(const void *tiles, const void *pal_idx, int64_t n, int npal, int pal_size, int max_iter,
                                      void *out_palettes, void *stream);
/* DoPalettization (:4105-4245): cluster features -> PalIdx_Initial ranked by tile count. */
TM_API int tm_stage_palettize(const void *feat_i32, const void *use, int64_t n, int npal, int max_iter, void *out_pal_idx,
                              void *stream);

/* A17: TKModes.ComputeKModes (kmodes.pas:923-1094; unreachable in the reference snapshot, named by the north star): k-modes on rows
 * of cKModesFeatureCount = 80 bytes (kmodes.pas:15; the reference's asm hard-codes 80, :338-342), dissimilarity = sum |a-b| + 2048 per
 * differing byte (:248-259), farthest-first initialisation from a starting point (:694-772), Huang's online mode update (:774-803),
 * empty-cluster repair with the LCG of :88-92 seeded $42381337 (:933), stop on cost non-decrease with three graces (:1040-1049).
 * HOST pointers, like the Pascal arrays: rows [n][80] with values < num_modalities, labels int32 [n] (0-based, as the code returns
 * them), centroids [num_clusters][80].  num_init <= 0: one run from point -num_init; > 0: that many runs from spread starting points
 * (:952-964), the cheapest kept.  max_iter < 0: no limit (aMaxIter = -1).  An upload and a read-back round tm_stage_kmodes_dev. */
TM_API int tm_stage_kmodes(const uint8_t *host_rows, int64_t n, int num_clusters, int num_init, int num_modalities, int max_iter, int32_t *host_labels,
                           uint8_t *host_centroids, uint64_t *host_cost, int *host_iters, void *stream);

/* The same on DEVICE pointers (rows [n][80], labels [n], centroids [num_clusters][80]); cost, the best run's iteration count and the
 * points x iterations gone through (all runs) come back to the host.  Blocking. */
TM_API int tm_stage_kmodes_dev(const uint8_t *dev_rows, int64_t n, int num_clusters, int num_init, int num_modalities, int max_iter, int32_t *dev_labels,
                               uint8_t *dev_centroids, uint64_t *host_cost, int *host_iters, int64_t *host_point_iters, void *stream);

/* A17, the other half: dl3quant (dlquant/quantizer.c:437-455; imported at extern.pas:196, never called, DLL not shipped): Dennis Lee's
 * DL3 quantiser -- histogram at lookup_bpc bits per channel (build_table3, :486-518), greedy merging of the pair of least error
 * (reduce_table3, :583-648; calc_err, :520-541), the entries' rounded means as the palette (set_palette3, :650-664).  DEVICE pointers:
 * npixels x (R, G, B) bytes in, [3][quant_to] planar bytes out (like userpal), *out_colors = colours left (<= quant_to).  Blocking.
 * Parity unpinned (no reference output exists); checked against the oracle's restatement. */
TM_API int tm_stage_dl3quant(const uint8_t *dev_rgb, int64_t npixels, int quant_to, int lookup_bpc, uint8_t *dev_palette, int *out_colors, void *stream);

/* A11: OptimizePalettes (:4309-4432), host arithmetic on HOST memory (P x PaletteSize colours); in place. */
TM_API int tm_optimize_palettes_host(int32_t *palettes, int pal_count, int pal_size, int *sweeps);

/* ======================================================================================= fine seam
 * extern.pas:182-185 (ANN_short.dll), :198-203 (yakmo.dll), :218-223 (BICO.dll).  Host pointers. */
/* ANN.dll (extern.pas:178-180; call sites tilingencoder.pas:4128, 4183-4187): double coordinates, any dimension.  These are the
 * DLL's own export names.  ANN_short.dll exports the SAME names for its int16 build (the `_short` suffix exists on the Pascal
 * side only: `external 'ANN_short.dll' name 'ann_kdtree_create'`, extern.pas:182-185); one shared object cannot export one
 * name twice, so the int16 set is exported as ann_kdtree_short_* and the import unit names those.
 * Exact nearest row (eps is ignored: the reference passes 0), squared distance in *err, ties -> lowest index. */
typedef struct tm_annd tm_annd;
TM_API tm_annd *ann_kdtree_create(double **rows, int n, int dd, int bs, int split);
TM_API void ann_kdtree_destroy(tm_annd *);
TM_API int ann_kdtree_search(tm_annd *, const double *q, double eps, double *err);
TM_API int ann_kdtree_search_batch(tm_annd *, const double *queries, int nq, int32_t *idxs, double *errs);
typedef struct tm_ann tm_ann;
TM_API tm_ann *ann_kdtree_short_create(int16_t **rows, int n, int dd, int bs, int split);
TM_API void ann_kdtree_short_destroy(tm_ann *);
TM_API int ann_kdtree_short_search(tm_ann *, const int16_t *q, uint32_t eps, uint32_t *err);
TM_API void ann_kdtree_short_search_multi(tm_ann *, int32_t *idxs, uint32_t *errs, int cnt, const int16_t *q, uint32_t eps);
TM_API int ann_kdtree_short_search_batch(tm_ann *, const int16_t *queries, int nq, int32_t *idxs, uint32_t *errs);

/* yakmo.dll (extern.pas:198-203).  The clustering is the build's deterministic k-means: yakmo's
 * k-means++ RNG is not recoverable.  Inputs are rounded to int32; 3- or 192-column data only. */
typedef struct tm_yakmo tm_yakmo;
TM_API tm_yakmo *yakmo_create(uint32_t k, uint32_t restart_count, int max_iter, int init_type, int init_seed, int do_normalize, int is_verbose);
TM_API void yakmo_destroy(tm_yakmo *);
TM_API void yakmo_set_num_threads(int num_threads);
TM_API void yakmo_load_train_data(tm_yakmo *, uint32_t row_count, uint32_t col_count, double **dataset);
TM_API void yakmo_train_on_data(tm_yakmo *, int32_t *point_to_cluster);
TM_API void yakmo_get_centroids(tm_yakmo *, double **centroids);

/* BICO.dll (extern.pas:218-223): the "coreset" is the build's weighted k-means with `coresetsize` centres. */
typedef struct tm_bico tm_bico;
TM_API tm_bico *bico_create(int64_t dimension, int64_t npoints, int64_t k, int64_t nrandproj, int64_t coresetsize, int random_seed);
TM_API void bico_destroy(tm_bico *);
TM_API void bico_set_num_threads(int num_threads);
TM_API void bico_set_rebuild_properties(tm_bico *, uint32_t interval, double initial, double grow);
TM_API void bico_insert_line(tm_bico *, const double *line, double weight);
TM_API int64_t bico_get_results(tm_bico *, double *centroids, double *weights);
/* dlquant_dll.dll (extern.pas:196; quantizer.h:20-21): HOST pointers as the import has them -- width x height RGB bytes in, the palette
 * planar into userpal[3][PALETTE_MAX = 65536]; returns 0 on success (1 on failure, message in tm_last_error).  = tm_stage_dl3quant
 * with an upload and a read-back around it. */
TM_API int dl3quant(unsigned char *inbuf, int width, int height, int quant_to, int lookup_bpc, unsigned char userpal[3][65536]);

#ifdef __cplusplus
}
#endif
#endif
