// tm_internal.h -- launcher prototypes shared between the kernel files, the stage ABI and the encoder.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "tm_common.h"

namespace tmx {

// tm_features.hip
int launch_load(const void *frames, int nframes, int img_w, int img_h, int tm_w, int tm_h, void *tiles, void *flags,
                void *lab_means, hipStream_t stream);
int launch_rgb_to_lab(const void *rgb, int64_t n, void *out, hipStream_t stream);
int launch_pearson(const void *lab, int nframes, int per, void *correl, hipStream_t stream);
int launch_features_rgb(const void *tiles, int64_t n, const void *mirror_flags, int mode, int use_lab, void *out, hipStream_t stream);
// colmm (optional): [384] ints on the device, mn[192] preset to INT_MAX and mx[192] to INT_MIN: the kernel folds the output columns' ranges in
int launch_features_rgb_rows(const void *tiles, const void *rows, int64_t n, int mode, int use_lab, void *out, hipStream_t stream, void *colmm = nullptr);
int launch_features_pal(const void *pal_px, const void *pal_idx, int64_t n, const void *palettes, int pal_size, int mode, void *out,
                        hipStream_t stream);
int launch_features_cluster(const void *tiles, int64_t n, int mode, void *out, hipStream_t stream);
int launch_window_dcts(const void *fb, int w, int h, void *out, hipStream_t stream, const int *only_if = nullptr);
// The windows' features straight into the matrix layout of the motion search (tm_motion.hip: what k_mo_pack_win makes of the int16 rows) -- the
// int16 rows are then never written.  Per block of 32 consecutive window positions of a row: [12 chunks][64 lanes][16 B] digits (low digits of the
// 160 plain coefficients + the 16 of b5 + b6, then the high digits), 32 norms, block 7 of the first half raw.  `flag` (device int, preset 0) is
// raised when a coefficient of the windows or of the `ntiles` tiles `cur` (their int16 features) lies beyond the range in which the matrix form is exact.
constexpr int MM_CH = 6;                                             // 32-wide chunks of the 160 plain coefficients + 16 of block 6 + 16 of padding
constexpr int MM_DIG = 2 * MM_CH * 1024, MM_NORM = MM_DIG, MM_QUIRK = MM_DIG + 128;
constexpr int MM_BLK_BYTES = MM_QUIRK + 32 * 16;                     // 12928
constexpr int MM_LIMIT6 = 10922;                                     // |coefficient| bound of blocks 5 and 6 under which a6 - b5 - b6 cannot saturate
constexpr int MM_LIMIT = 16383;                                      // |coefficient| bound under which no plain difference saturates
int launch_window_dcts_packed(const void *fb, int w, int h, const void *cur, int ntiles, void *packed, int *flag, hipStream_t stream);
// int16 features of the (tile, palette) pairs pairs[i] = tile << 32 | palette (the rows the extended-palette re-rank asks for)
int launch_features_pairs(const void *pal_px, const void *pairs, int64_t n, const void *palettes, int pal_size, void *out, hipStream_t stream);
int launch_features_table(const void *pal_px, int64_t ntiles, const void *palettes, int npal, int pal_size, void *out, hipStream_t stream);

// tm_epu.hip: FrameTilingExtendedPaletteUsage (tilingencoder.pas:1559-1610)
int launch_knn_topk(const void *queries, int64_t nq, const void *db, int64_t nt, int k, void *out_idx, void *out_err, hipStream_t stream);
int launch_epu_rerank_ondemand(const void *queries, int64_t nq, const void *knn_idx, int k, const void *tile_pal, int64_t ntiles, const void *pal_px,
                               const void *palettes, int npal, int pal_size, void *out_tile, void *out_pal, void *out_err, hipStream_t stream);
int launch_epu_rerank(const void *queries, int64_t nq, const void *knn_idx, int k, const void *tile_pal, int64_t ntiles, int npal,
                      const void *table, void *out_tile, void *out_pal, void *out_err, hipStream_t stream);

// tm_knn.hip
struct tm_knn_index_impl;
int knn_index_create(const void *db, int64_t nt, hipStream_t stream, tm_knn_index_impl **out);
void knn_index_destroy(tm_knn_index_impl *ix);
// query_colmm (optional): the queries' column ranges as launch_features_rgb_rows leaves them (device, [384]) -- saves the search its own pass
int knn_index_search(tm_knn_index_impl *ix, const void *queries, int64_t nq, void *out_idx, void *out_err, hipStream_t stream, const void *query_colmm = nullptr);
void knn_index_stats(tm_knn_index_impl *ix, double *ms, int *kbytes, int64_t *pairs);
void knn_last_plan(int *ht, int *hq, int *topk, long long *arena_retries);
// the last search's three kernels on their own: ms of seeds / lists / consume; pairs evaluated by seeds / consume, matrix instructions the consume kernel issued
void knn_index_kernel_split(tm_knn_index_impl *ix, double ms[3], int64_t pairs[3]);
// grp_off / grp_members (optional): the index was built over DISTINCT rows; results are expanded to the original rows (member lists
// as build_groups makes them), full_db = all rows, for the brute-force fallback
int knn_index_search_topk(tm_knn_index_impl *ix, const void *queries, int64_t nq, int k, void *out_idx, void *out_err, hipStream_t stream,
                          const void *grp_off = nullptr, const void *grp_members = nullptr, const void *full_db = nullptr, int64_t full_nt = 0);

// tm_dither.hip
int launch_dither(const void *tiles, const void *flags, const void *pal_idx, int64_t n, const void *palettes, int npal, int pal_size,
                  int use_tk, int y2_mixed, void *out_pal_px, hipStream_t stream, int64_t *pairs_planned = nullptr,
                  const void *pair_keys = nullptr, int64_t n_pair_keys = 0);
// pair_keys (optional): the distinct pixel keys palette << 24 | G << 16 | R << 8 | B of exactly these tiles (or a superset), as run_quantize_palettes
// leaves them -- saves Dither its own pass over the pixels

// tm_dedup.hip
int run_dedup(const void *rows, int64_t n, int row_bytes, const void *use_in, void *remap, void *order, void *use_out,
              int64_t *host_n_unique, hipStream_t stream, int64_t exact_first = 0);
int run_dedup_ex(const void *rows, int64_t n, int row_bytes, const void *use_in, void *remap, void *order, void *use_out,
                 int64_t *host_n_unique, int by_index, hipStream_t stream, int64_t exact_first = 0);

int build_groups(const void *remap, int64_t n, const void *counts, int64_t ngroups, void *off, void *members, hipStream_t stream);
int compact_kept(const void *keep, int64_t n, void *out_idx, void *pos, int64_t *host_count, hipStream_t stream);

// tm_motion.hip: motion prediction (tilingencoder.pas:1154-1282, 1496-1654) and Reduce's tile-count search (4014-4046)
int launch_motion_search(const void *cur, int tm_w, int tm_h, const void *win, int radius, void *best_err, void *px, void *py,
                         hipStream_t stream);
// the encoder's form: window features of the frame buffer `fb` (tm_w * 8 x tm_h * 8) and the search in one go -- the windows are made in the
// search's matrix layout at once (launch_window_dcts_packed); `win` ((tm_w*8-7) * (tm_h*8-7) * 384 B) is written only by the fallback (a frame
// whose coefficients leave the matrix form's exact range, or TM_MOTION_VALU=1) and by TM_MOTION_PACK_SEPARATE=1 (the two-pass form, for A/B runs)
int launch_motion_search_fb(const void *cur, int tm_w, int tm_h, const void *fb, void *win, int radius, void *best_err, void *px, void *py,
                            hipStream_t stream);
int launch_tiles_to_screen(const void *tiles, const void *flags, int tm_w, int tm_h, void *screen, hipStream_t stream);
int launch_recon_decide(int tm_w, int per, int pal_from_map, const void *mp_err, const void *fflags, const void *gpal_idx, const void *gpal_px,
                        const void *palettes, int pal_size, const void *back, void *front, void *tm_tile, void *tm_pal, void *tm_err,
                        const void *px, const void *py, void *pred, hipStream_t stream);
int solve_tile_count(const void *group, int64_t ngroups, const void *pm_err, const void *frame_is_kf, int per, int64_t q, double target,
                     void *pred, void *keep, double *x_out, int *probes_out, hipStream_t stream);
float euclidean_to_psnr(uint32_t e);

// One process per GPU: the collectives a step needs between its kernels, handed in by the host (tm_set_collective).  The calls
// are made on the caller's thread with the encoder's stream idle, and return with the result in place.
struct Collectives {
  int rank = 0, world = 1;
  std::function<int(void *buf, int64_t count)> allreduce_sum_i32, allreduce_max_i32, allreduce_sum_i64;
  std::function<int(const void *send, void *recv, int64_t bytes_per_rank)> allgather;  // recv: world x bytes_per_rank, rank order
};

// tm_kmeans.hip
// DoPalettization over `world` processes: every process holds the points of its own tile range (global index of the first:
// global_begin); the farthest-first picks are settled by an all-gather of one candidate per process, the Lloyd iterations by
// an all-reduce of the exact integer sums, so every process ends with the same centroids and with the assignment of its own
// range (out_pal_idx: n_local entries, already ranked by global tile count).
int run_palettize_dist(const void *feat_local, const void *use_local, int64_t n_local, int64_t global_begin, int npal, int max_iter,
                       void *out_pal_idx_local, const Collectives &co, hipStream_t stream);
int run_kmeans(const void *pts, const void *weights, int64_t n, int d, int k, int max_iter, void *assign, void *centroids, int *host_k,
               int *host_iters, hipStream_t stream);
// the same Lloyd iterations from the caller's own initial centres (k point indices, -1 = none) instead of the farthest-first picks
int run_kmeans_seeded(const void *pts, const void *weights, int64_t n, int d, int k, const int64_t *host_init_idx, int max_iter, void *assign, void *centroids,
                      int *host_k, int *host_iters, hipStream_t stream);
// keep_keys / keep_n (optional): the sorted distinct pixel keys palette << 24 | G << 16 | R << 8 | B the quantisation found, for Dither
struct DevBuf;
int run_quantize_palettes(const void *tiles, const void *pal_idx, int64_t n, int npal, int pal_size, int max_iter, void *out_palettes,
                          hipStream_t stream, DevBuf *keep_keys = nullptr, int64_t *keep_n = nullptr);
// the same for the palettes p with p % pal_world == pal_rank only (independent tasks, one thread per palette in the reference:
// tilingencoder.pas:1864); the other palettes' rows come back as zeros, so that an all-reduce(SUM) assembles the set
int run_quantize_palettes_part(const void *tiles, const void *pal_idx, int64_t n, int npal, int pal_size, int max_iter, void *out_palettes,
                               int pal_rank, int pal_world, hipStream_t stream, DevBuf *keep_keys = nullptr, int64_t *keep_n = nullptr);
int run_palettize(const void *feat, const void *use, int64_t n, int npal, int max_iter, void *out_pal_idx, hipStream_t stream);
bool palettize_resident(int64_t n, int npal);  // the clustering above would take the resident launch (then several processes each run it whole)
// tm_dedup.hip, Reduce over several processes (see there): a 16-byte key per distinct tile (rows[idx[r]], use[r]) and, on the gathered keys of
// all processes, the tiles that can be among the first `target` of the merged order (in_s: uint32 flags)
int reduce_make_keys(const void *rows, const void *idx, const void *use, int64_t n, int row_bytes, void *keys_out /* n x 16 bytes */, hipStream_t stream);
int reduce_select_candidates(const void *keys, int64_t n, int64_t target, void *in_s, hipStream_t stream);
// tm_dl3.hip: dl3quant on device pointers (blocking)
int run_dl3quant(const void *dev_rgb, int64_t npixels, int quant_to, int lookup_bpc, void *dev_pal, int *out_colors, hipStream_t stream);
// what the calling thread's last tile -> palette clustering and last colour quantisation ran through (tm_get_kmeans_iters)
struct KmeansRunStats { int tile_iters = 0; int64_t tile_points = 0; int pixel_iters = 0; int64_t pixel_colours = 0, pixels = 0, pixel_colour_iters = 0; };
KmeansRunStats &kmeans_run_stats();

// tm_kmodes.hip: A17, TKModes.ComputeKModes (kmodes.pas:923-1094); host pointers
int run_kmodes(const uint8_t *rows, int64_t n, int k, int num_init, int nmod, int max_iter, int32_t *labels_out, uint8_t *cent_out, uint64_t *cost_out,
               int *iters_out, hipStream_t stream);

// tm_optpal.hip (host only)
int optimize_palettes_host(std::vector<int32_t> &pals, int pal_count, int pal_size, int *sweeps_out);


// tm_gtm.hip (host only): SaveStream restatement, tilingencoder.pas:5177-5482
struct GtmInput {
  int tm_w = 0, tm_h = 0, nframes = 0;
  double fps = 0;
  std::vector<int32_t> kf_start;
  const uint8_t *pal_px = nullptr;   // [ntiles][64], final (Reindex) order
  std::vector<uint32_t> use;         // [ntiles]
  const int32_t *palettes = nullptr; // [pal_count][pal_size]
  int pal_count = 0, pal_size = 0;
  const tm_tilemap_item *tilemap = nullptr;  // [nframes][tm_h*tm_w]
  std::string settings;
};
int write_gtm(const char *path, const GtmInput &in);
void lz_compress(const std::vector<uint8_t> &raw, std::vector<uint8_t> &dst);
int lz_decompress(const uint8_t *src, size_t n, std::vector<uint8_t> &dst, size_t *consumed);
// LoadStream (tilingencoder.pas:4880-5175)
struct GtmLoaded {
  int header_w = 0, header_h = 0, header_frames = -1;  // from the GTMv header (-1: headerless stream)
  int tm_w = 0, tm_h = 0, nframes = 0, pal_size = 0, pal_count = 0;
  double fps = 0;
  std::vector<int32_t> kf_start;
  std::vector<uint8_t> pal_px;      // [tiles][64]
  std::vector<uint32_t> use;        // UseCount as SetTMI counts it (4968-4969)
  std::vector<int32_t> palettes;    // [pal_count][pal_size], alpha stripped (4951)
  std::vector<tm_tilemap_item> tilemap;  // [nframes][tm_h*tm_w]
  std::string settings;
};
int read_gtm(const char *path, GtmLoaded *out);

}  // namespace tmx
