// tm_api.hip -- extern "C" surface of libtilemotion.so: stage seam + KNN index + misc (include/tilemotion.h).
#include <cstdlib>

#include "tm_common.h"
#include "tm_internal.h"

using namespace tmx;

extern "C" {

const char *tm_last_error(void) { return tmx::get_error(); }

int tm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// the trailing token names the build of the KNN scan kernel: PMC passes under profiles/ are keyed by it (bench.py reads `traffic` from them)
const char *tm_version(void) { return "tilemotion-mi355x 0.3 (gfx950) knn-scan3-r05b"; }

int tm_stage_load(const void *frames, int nframes, int img_w, int img_h, int tm_w, int tm_h, void *tiles, void *flags,
                  void *lab_means, void *stream) {
  knobs_reload();
  return launch_load(frames, nframes, img_w, img_h, tm_w, tm_h, tiles, flags, lab_means, (hipStream_t)stream);
}

int tm_stage_rgb_to_lab(const void *rgb, int64_t n, void *out_lab, void *stream) {
  knobs_reload();
  TM_TRY(require_device());
  return launch_rgb_to_lab(rgb, n, out_lab, (hipStream_t)stream);
}

int tm_stage_features_rgb(const void *tiles, int64_t n, const void *mirror_flags, int mode, int use_lab, void *out_i16,
                          void *stream) {
  knobs_reload();
  return launch_features_rgb(tiles, n, mirror_flags, mode, use_lab, out_i16, (hipStream_t)stream);
}

int tm_stage_features_pal(const void *pal_px, const void *pal_idx, int64_t n, const void *palettes, int pal_size, int mode,
                          void *out_i16, void *stream) {
  knobs_reload();
  return launch_features_pal(pal_px, pal_idx, n, palettes, pal_size, mode, out_i16, (hipStream_t)stream);
}

int tm_stage_features_cluster(const void *tiles, int64_t n, int mode, void *out_i32, void *stream) {
  knobs_reload();
  return launch_features_cluster(tiles, n, mode, out_i32, (hipStream_t)stream);
}

int tm_stage_window_dcts(const void *frame_buffer, int width, int height, void *out_i16, void *stream) {
  knobs_reload();
  TM_TRY(require_device());
  return launch_window_dcts(frame_buffer, width, height, out_i16, (hipStream_t)stream);
}

int tm_stage_motion_search(const void *cur_i16, int tm_w, int tm_h, const void *window_dcts, int radius, void *out_err, void *out_px,
                           void *out_py, void *stream) {
  knobs_reload();
  TM_TRY(require_device());
  return launch_motion_search(cur_i16, tm_w, tm_h, window_dcts, radius, out_err, out_px, out_py, (hipStream_t)stream);
}

int tm_stage_knn_topk(const void *queries_i16, int64_t nq, const void *db_i16, int64_t nt, int k, void *out_idx, void *out_err, void *stream) {
  knobs_reload();
  TM_TRY(require_device());
  if (knobs().topk_brute) return launch_knn_topk(queries_i16, nq, db_i16, nt, k, out_idx, out_err, (hipStream_t)stream);  // the VALU brute force
  tm_knn_index_impl *ix = nullptr;
  TM_TRY(knn_index_create(db_i16, nt, (hipStream_t)stream, &ix));
  const int rc = knn_index_search_topk(ix, queries_i16, nq, k, out_idx, out_err, (hipStream_t)stream);
  knn_index_destroy(ix);
  return rc;
}

int tm_stage_epu_rerank(const void *queries_i16, int64_t nq, const void *knn_idx, int k, const void *pal_px, const void *tile_pal_idx,
                        int64_t ntiles, const void *palettes, int npal, int pal_size, void *out_tile, void *out_pal, void *out_err,
                        void *stream) {
  knobs_reload();
  TM_TRY(require_device());
  const double table_gib = knobs().epu_table_gib;
  if ((double)ntiles * npal * 384.0 > table_gib * 1073741824.0)  // no room for every tile under every palette: only the rows the queries name
    return launch_epu_rerank_ondemand(queries_i16, nq, knn_idx, k, tile_pal_idx, ntiles, pal_px, palettes, npal, pal_size, out_tile, out_pal, out_err, (hipStream_t)stream);
  DevBuf table;
  TM_TRY(table.alloc((size_t)std::max<int64_t>(ntiles, 1) * npal * 384));
  TM_TRY(launch_features_table(pal_px, ntiles, palettes, npal, pal_size, table.p, (hipStream_t)stream));
  TM_TRY(launch_epu_rerank(queries_i16, nq, knn_idx, k, tile_pal_idx, ntiles, npal, table.p, out_tile, out_pal, out_err, (hipStream_t)stream));
  TM_HIP(hipStreamSynchronize((hipStream_t)stream));  // the table is freed on return
  return TM_OK;
}

tm_knn_index *tm_knn_index_create(const void *db_i16, int64_t nt, void *stream) {
  knobs_reload();
  tm_knn_index_impl *ix = nullptr;
  if (knn_index_create(db_i16, nt, (hipStream_t)stream, &ix) != TM_OK) return nullptr;
  return reinterpret_cast<tm_knn_index *>(ix);
}

void tm_knn_index_destroy(tm_knn_index *ix) { knn_index_destroy(reinterpret_cast<tm_knn_index_impl *>(ix)); }

int tm_knn_index_search(tm_knn_index *ix, const void *queries_i16, int64_t nq, void *out_idx, void *out_err, void *stream) {
  knobs_reload();
  return knn_index_search(reinterpret_cast<tm_knn_index_impl *>(ix), queries_i16, nq, out_idx, out_err, (hipStream_t)stream);
}

int tm_knn_index_last_stats(tm_knn_index *ix, double *kernel_ms, int *k_bytes, int64_t *pairs) {
  knobs_reload();
  TM_CHECK(ix != nullptr, TM_E_INVAL, "null index");
  knn_index_stats(reinterpret_cast<tm_knn_index_impl *>(ix), kernel_ms, k_bytes, pairs);
  return TM_OK;
}

int tm_knn_last_plan(int *ht, int *hq, int *topk, int64_t *arena_retries) {
  long long r = 0;
  knn_last_plan(ht, hq, topk, &r);
  if (arena_retries) *arena_retries = (int64_t)r;
  return TM_OK;
}

int tm_stage_knn(const void *queries_i16, int64_t nq, const void *db_i16, int64_t nt, void *out_idx, void *out_err, void *stream) {
  knobs_reload();
  tm_knn_index_impl *ix = nullptr;
  TM_TRY(knn_index_create(db_i16, nt, (hipStream_t)stream, &ix));
  int rc = knn_index_search(ix, queries_i16, nq, out_idx, out_err, (hipStream_t)stream);
  knn_index_destroy(ix);
  return rc;
}

int tm_stage_dither(const void *tiles, const void *flags, const void *pal_idx, int64_t n, const void *palettes, int npal,
                    int pal_size, int use_thomas_knoll, int y2_mixed_colors, void *out_pal_px, void *stream) {
  knobs_reload();
  return launch_dither(tiles, flags, pal_idx, n, palettes, npal, pal_size, use_thomas_knoll, y2_mixed_colors, out_pal_px,
                       (hipStream_t)stream);
}

int tm_stage_dedup(const void *rows, int64_t n, int row_bytes, const void *use_in, void *remap, void *order, void *use_out,
                   int64_t *host_n_unique, void *stream) {
  knobs_reload();
  return run_dedup(rows, n, row_bytes, use_in, remap, order, use_out, host_n_unique, (hipStream_t)stream);
}

int tm_stage_kmeans(const void *pts_i32, const void *weights, int64_t n, int d, int k, int max_iter, void *assign, void *centroids,
                    int *host_k, int *host_iters, void *stream) {
  knobs_reload();
  return run_kmeans(pts_i32, weights, n, d, k, max_iter, assign, centroids, host_k, host_iters, (hipStream_t)stream);
}

int tm_stage_kmeans_seeded(const void *pts_i32, const void *weights, int64_t n, int d, int k, const int64_t *host_init_idx, int max_iter, void *assign,
                           void *centroids, int *host_k, int *host_iters, void *stream) {
  knobs_reload();
  TM_CHECK(host_init_idx != nullptr, TM_E_INVAL, "kmeans: null initial centres");
  return run_kmeans_seeded(pts_i32, weights, n, d, k, host_init_idx, max_iter, assign, centroids, host_k, host_iters, (hipStream_t)stream);
}

int tm_stage_quantize_palettes(const void *tiles, const void *pal_idx, int64_t n, int npal, int pal_size, int max_iter,
                               void *out_palettes, void *stream) {
  knobs_reload();
  return run_quantize_palettes(tiles, pal_idx, n, npal, pal_size, max_iter, out_palettes, (hipStream_t)stream);
}

int tm_stage_palettize(const void *feat_i32, const void *use, int64_t n, int npal, int max_iter, void *out_pal_idx, void *stream) {
  knobs_reload();
  return run_palettize(feat_i32, use, n, npal, max_iter, out_pal_idx, (hipStream_t)stream);
}

}  // extern "C"
