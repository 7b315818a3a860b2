// tm_fine.hip -- fine seam: drop-in twins of the ANN_short.dll imports (extern.pas:182-185) with the same per-call
// semantics, host pointers in and out.  Per-call use is latency bound on a GPU (one H2D + kernels + one D2H per
// query); it exists so unmodified Pascal call sites (tilingencoder.pas:1547, 1563, 4600, 4618) keep working.  The
// batch twin is the useful one.  Calls on one tree are serialised by a mutex: the reference calls search concurrently
// from its thread pool (tilingencoder.pas:1673).
#include <algorithm>
#include <mutex>

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {

// ANN.dll (double coordinates): exact nearest row by squared Euclidean distance, summed over the dimensions in order with one IEEE
// multiplication and one addition each (what a plain loop compiles to without FMA contraction); ties -> lowest index (the build's
// rule: ANN's traversal order is not recoverable).  One workgroup per query, threads over the rows; the trees of the one call site
// (DoPalettization, tilingencoder.pas:4128, 4183-4187) hold at most PaletteCount x 8 BICO centroids.
__global__ __launch_bounds__(256) void k_ann_f64(const double *__restrict__ rows, int n, int dd, const double *__restrict__ queries, int32_t *__restrict__ out_idx,
                                                 double *__restrict__ out_err) {
  __shared__ double s_d[256];
  __shared__ int s_i[256];
  const double *q = queries + (int64_t)blockIdx.x * dd;
  double best = 0.0;
  int bi = -1;
  for (int r = threadIdx.x; r < n; r += 256) {
    const double *p = rows + (int64_t)r * dd;
    double s = 0.0;
    for (int j = 0; j < dd; j++) { const double t = __dsub_rn(q[j], p[j]); s = __dadd_rn(s, __dmul_rn(t, t)); }
    if (bi < 0 || s < best) { best = s; bi = r; }
  }
  s_d[threadIdx.x] = best; s_i[threadIdx.x] = bi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const int oi = s_i[threadIdx.x + o];
      const double od = s_d[threadIdx.x + o];
      const int mi = s_i[threadIdx.x];
      if (oi >= 0 && (mi < 0 || od < s_d[threadIdx.x] || (od == s_d[threadIdx.x] && oi < mi))) { s_d[threadIdx.x] = od; s_i[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out_idx[blockIdx.x] = s_i[0]; out_err[blockIdx.x] = s_i[0] >= 0 ? s_d[0] : 0.0; }
}

}  // namespace tmx

using namespace tmx;

struct tm_annd {
  DevBuf rows;
  int n = 0, dd = 0;
  std::mutex mu;
};

struct tm_ann {
  DevBuf db, q, idx, err, all;
  tm_knn_index_impl *ix = nullptr;
  int n = 0;
  std::mutex mu;
  ~tm_ann() { if (ix) knn_index_destroy(ix); }
};

extern "C" {

// ---- ANN.dll (extern.pas:178-180): double coordinates, any dimension
tm_annd *ann_kdtree_create(double **rows, int n, int dd, int bs, int split) {
  knobs_reload();
  (void)bs; (void)split;  // bucket size / split rule shape a kd-tree; the exact answer does not depend on them
  if (require_device() != TM_OK) return nullptr;
  if (dd < 1 || n < 0 || (n > 0 && !rows)) { set_error("ann_kdtree_create: bad arguments (n %d, dd %d)", n, dd); return nullptr; }
  tm_annd *a = new tm_annd();
  a->n = n; a->dd = dd;
  std::vector<double> flat((size_t)n * dd);
  for (int i = 0; i < n; i++) memcpy(&flat[(size_t)i * dd], rows[i], (size_t)dd * 8);  // copied: no borrowed host pointers
  if (a->rows.alloc(std::max<size_t>(flat.size(), 1) * 8) != TM_OK || (n && hipMemcpy(a->rows.p, flat.data(), flat.size() * 8, hipMemcpyHostToDevice) != hipSuccess)) {
    delete a;
    return nullptr;
  }
  return a;
}

void ann_kdtree_destroy(tm_annd *a) { delete a; }

int ann_kdtree_search_batch(tm_annd *a, const double *queries, int nq, int32_t *idxs, double *errs) {
  knobs_reload();
  TM_CHECK(a && (nq == 0 || (queries && idxs)), TM_E_INVAL, "null argument");
  if (nq <= 0) return TM_OK;
  std::lock_guard<std::mutex> lk(a->mu);
  DevBuf q, i, e;
  TM_TRY(q.alloc((size_t)nq * a->dd * 8)); TM_TRY(i.alloc((size_t)nq * 4)); TM_TRY(e.alloc((size_t)nq * 8));
  TM_HIP(hipMemcpy(q.p, queries, (size_t)nq * a->dd * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_ann_f64, dim3((unsigned)nq), dim3(256), 0, nullptr, a->rows.as<double>(), a->n, a->dd, q.as<double>(), i.as<int32_t>(), e.as<double>());
  TM_HIP(hipGetLastError());
  TM_HIP(hipMemcpy(idxs, i.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
  if (errs) TM_HIP(hipMemcpy(errs, e.p, (size_t)nq * 8, hipMemcpyDeviceToHost));
  return TM_OK;
}

int ann_kdtree_search(tm_annd *a, const double *q, double eps, double *err) {
  knobs_reload();
  (void)eps;  // the reference passes 0.0 (exact, tilingencoder.pas:4128)
  int32_t idx = -1;
  double e = 0.0;
  if (!a || !q || ann_kdtree_search_batch(a, q, 1, &idx, &e) != TM_OK) idx = -1;
  if (err) *err = e;
  return idx;
}

// ---- ANN_short.dll (extern.pas:182-185): the same export names in ANN_short.dll, `_short` on the Pascal side
tm_ann *ann_kdtree_short_create(int16_t **rows, int n, int dd, int bs, int split) {
  knobs_reload();
  (void)bs; (void)split;  // bucket size / split rule shape a kd-tree; the exact answer does not depend on them
  if (require_device() != TM_OK) return nullptr;
  if (dd != 192 || n < 0 || (n > 0 && !rows)) { set_error("ann_kdtree_create: only dd = 192 (cTileDCTSize) is supported"); return nullptr; }
  tm_ann *a = new tm_ann();
  a->n = n;
  std::vector<int16_t> flat((size_t)n * 192);
  for (int i = 0; i < n; i++) memcpy(&flat[(size_t)i * 192], rows[i], 384);  // rows are copied: no borrowed host pointers
  if (a->db.alloc((size_t)std::max(n, 1) * 384) != TM_OK ||
      (n && hipMemcpy(a->db.p, flat.data(), flat.size() * 2, hipMemcpyHostToDevice) != hipSuccess) ||
      knn_index_create(a->db.p, n, nullptr, &a->ix) != TM_OK || a->q.alloc(384) != TM_OK || a->idx.alloc(4) != TM_OK ||
      a->err.alloc(4) != TM_OK) {
    delete a;
    return nullptr;
  }
  return a;
}

void ann_kdtree_short_destroy(tm_ann *a) { delete a; }

int ann_kdtree_short_search_batch(tm_ann *a, const int16_t *queries, int nq, int32_t *idxs, uint32_t *errs) {
  knobs_reload();
  TM_CHECK(a && (nq == 0 || (queries && idxs)), TM_E_INVAL, "null argument");
  if (nq <= 0) return TM_OK;
  std::lock_guard<std::mutex> lk(a->mu);
  DevBuf q, i, e;
  TM_TRY(q.alloc((size_t)nq * 384)); TM_TRY(i.alloc((size_t)nq * 4)); TM_TRY(e.alloc((size_t)nq * 4));
  TM_HIP(hipMemcpy(q.p, queries, (size_t)nq * 384, hipMemcpyHostToDevice));
  TM_TRY(knn_index_search(a->ix, q.p, nq, i.p, e.p, nullptr));
  TM_HIP(hipMemcpy(idxs, i.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
  if (errs) TM_HIP(hipMemcpy(errs, e.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
  return TM_OK;
}

int ann_kdtree_short_search(tm_ann *a, const int16_t *q, uint32_t eps, uint32_t *err) {
  knobs_reload();
  (void)eps;  // the reference always passes 0 (exact); a positive eps would only allow a worse answer
  int32_t idx = -1;
  uint32_t e = 0xffffffffu;
  if (!a || !q || ann_kdtree_short_search_batch(a, q, 1, &idx, &e) != TM_OK) idx = -1;
  if (err) *err = e;
  return idx;
}

void ann_kdtree_short_search_multi(tm_ann *a, int32_t *idxs, uint32_t *errs, int cnt, const int16_t *q, uint32_t eps) {
  knobs_reload();
  (void)eps;
  for (int i = 0; i < cnt; i++) { if (idxs) idxs[i] = -1; if (errs) errs[i] = 0xffffffffu; }
  if (!a || !q || !idxs || cnt <= 0 || a->n == 0) return;
  std::lock_guard<std::mutex> lk(a->mu);
  // the k nearest rows in the build's order (distance, index): the collection scan of tm_knn.hip for k <= 64, beyond that the
  // exact brute-force scan of tm_epu.hip in slices is not needed by any caller (the reference asks for 64, tilingencoder.pas:1433)
  const int k = std::min(cnt, 64);
  DevBuf dq, di, de;
  if (dq.alloc(384) != TM_OK || di.alloc((size_t)k * 4) != TM_OK || de.alloc((size_t)k * 4) != TM_OK) return;
  if (hipMemcpy(dq.p, q, 384, hipMemcpyHostToDevice) != hipSuccess) return;
  if (knn_index_search_topk(a->ix, dq.p, 1, k, di.p, de.p, nullptr) != TM_OK) return;
  if (hipMemcpy(idxs, di.p, (size_t)k * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
  if (errs && hipMemcpy(errs, de.p, (size_t)k * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// yakmo.dll / BICO.dll twins (extern.pas:198-203, 218-223).  Same call shapes, host pointers; the clustering itself
// is the build's deterministic k-means (tm_kmeans.hip) -- yakmo's k-means++ RNG and BICO's
// random projections are not recoverable, so results are the build's, not the DLLs'.  Inputs are rounded to int32
// (exact for the reference's pixel datasets, tilingencoder.pas:4477-4480); only 3- and 192-column data are built.
struct tm_yakmo {
  int k = 0, max_iter = 300, rows = 0, cols = 0, live = 0;
  std::vector<int32_t> pts;
  std::vector<double> cent;
};

struct tm_bico {
  int dim = 0;
  int64_t k = 0, coreset = 0;
  std::vector<int32_t> pts;
  std::vector<uint32_t> w;
};

static int kmeans_host(const std::vector<int32_t> &pts, const uint32_t *w, int n, int d, int k, int max_iter, int32_t *assign,
                       std::vector<double> &cent, int *live) {
  TM_TRY(require_device());
  TM_CHECK(d == 3 || d == 192, TM_E_UNSUPPORTED, "k-means twins: only 3 (pixels) or 192 (tile features) columns are built, got %d", d);
  DevBuf dp, dw, da, dc;
  TM_TRY(dp.alloc(pts.size() * 4)); TM_TRY(da.alloc((size_t)n * 4)); TM_TRY(dc.alloc((size_t)k * d * 8));
  TM_HIP(hipMemcpy(dp.p, pts.data(), pts.size() * 4, hipMemcpyHostToDevice));
  if (w) { TM_TRY(dw.alloc((size_t)n * 4)); TM_HIP(hipMemcpy(dw.p, w, (size_t)n * 4, hipMemcpyHostToDevice)); }
  int iters = 0;
  TM_TRY(run_kmeans(dp.p, w ? dw.p : nullptr, n, d, k, max_iter, da.p, dc.p, live, &iters, nullptr));
  cent.resize((size_t)k * d);
  TM_HIP(hipMemcpy(cent.data(), dc.p, cent.size() * 8, hipMemcpyDeviceToHost));
  if (assign) TM_HIP(hipMemcpy(assign, da.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return TM_OK;
}

extern "C" {

tm_yakmo *yakmo_create(uint32_t k, uint32_t restart_count, int max_iter, int init_type, int init_seed, int do_normalize, int is_verbose) {
  knobs_reload();
  (void)restart_count; (void)init_type; (void)init_seed; (void)do_normalize; (void)is_verbose;
  if (k == 0 || k > 65536) { set_error("yakmo_create: k out of range"); return nullptr; }
  tm_yakmo *y = new tm_yakmo();
  y->k = (int)k;
  y->max_iter = max_iter > 0 ? max_iter : 300;
  return y;
}
void yakmo_destroy(tm_yakmo *y) { delete y; }
void yakmo_set_num_threads(int) {}

void yakmo_load_train_data(tm_yakmo *y, uint32_t row_count, uint32_t col_count, double **dataset) {  // copies, like the DLL (4203, 4495)
  if (!y || !dataset) return;
  y->rows = (int)row_count;
  y->cols = (int)col_count;
  y->pts.resize((size_t)row_count * col_count);
  for (uint32_t r = 0; r < row_count; r++)
    for (uint32_t c = 0; c < col_count; c++) y->pts[(size_t)r * col_count + c] = (int32_t)llrint(dataset[r][c]);
}

void yakmo_train_on_data(tm_yakmo *y, int32_t *point_to_cluster) {
  knobs_reload();
  if (!y || y->rows <= 0) return;
  if (kmeans_host(y->pts, nullptr, y->rows, y->cols, std::min(y->k, y->rows), y->max_iter, point_to_cluster, y->cent, &y->live) != TM_OK)
    y->live = 0;
}

void yakmo_get_centroids(tm_yakmo *y, double **centroids) {
  knobs_reload();
  if (!y || !centroids) return;
  for (int c = 0; c < y->live; c++) memcpy(centroids[c], &y->cent[(size_t)c * y->cols], sizeof(double) * (size_t)y->cols);
}

tm_bico *bico_create(int64_t dimension, int64_t npoints, int64_t k, int64_t nrandproj, int64_t coresetsize, int random_seed) {
  knobs_reload();
  (void)nrandproj; (void)random_seed;
  if (dimension <= 0 || coresetsize <= 0) { set_error("bico_create: bad arguments"); return nullptr; }
  tm_bico *b = new tm_bico();
  b->dim = (int)dimension;
  b->k = k;
  b->coreset = coresetsize;
  b->pts.reserve((size_t)std::max<int64_t>(npoints, 0) * (size_t)dimension);
  return b;
}
void bico_destroy(tm_bico *b) { delete b; }
void bico_set_num_threads(int) {}
void bico_set_rebuild_properties(tm_bico *, uint32_t, double, double) {}

void bico_insert_line(tm_bico *b, const double *line, double weight) {
  knobs_reload();
  if (!b || !line) return;
  for (int c = 0; c < b->dim; c++) b->pts.push_back((int32_t)llrint(line[c]));
  b->w.push_back((uint32_t)std::max<long long>(1, llrint(weight)));
}

int64_t bico_get_results(tm_bico *b, double *centroids, double *weights) {
  knobs_reload();
  // the coreset of the inserted points = the build's k-means with coresetsize centres; weight = summed point weights
  if (!b || !centroids || !weights || b->w.empty()) return 0;
  const int n = (int)b->w.size(), k = (int)std::min<int64_t>(b->coreset, n);
  std::vector<int32_t> assign(n);
  std::vector<double> cent;
  int live = 0;
  if (kmeans_host(b->pts, b->w.data(), n, b->dim, k, 300, assign.data(), cent, &live) != TM_OK) return 0;
  for (int c = 0; c < live; c++) weights[c] = 0;
  for (int i = 0; i < n; i++) weights[assign[i]] += b->w[i];
  memcpy(centroids, cent.data(), sizeof(double) * (size_t)live * b->dim);
  return live;
}

}  // extern "C"
