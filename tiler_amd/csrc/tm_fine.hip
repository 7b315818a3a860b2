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

// all SSDs of one query against the database (CompareEuclideanDCTPtr, utils.pas:541-557): one lane per row
__global__ __launch_bounds__(256) void k_ssd_one_query(const int16_t *__restrict__ q, const int16_t *__restrict__ db, int64_t nt,
                                                       uint32_t *__restrict__ out) {
  __shared__ int s_q[192];
  for (int i = threadIdx.x; i < 192; i += 256) s_q[i] = q[i];
  __syncthreads();
  for (int64_t r = blockIdx.x * 256 + threadIdx.x; r < nt; r += (int64_t)gridDim.x * 256) {
    const int4 *p = reinterpret_cast<const int4 *>(db + r * 192);
    uint32_t ssd = 0;
    for (int v = 0; v < 24; v++) {
      const int4 x = p[v];
      const int w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int d0 = (int)(int16_t)(w[i] & 0xffff) - s_q[v * 8 + 2 * i];
        const int d1 = (w[i] >> 16) - s_q[v * 8 + 2 * i + 1];
        ssd += (uint32_t)(d0 * d0) + (uint32_t)(d1 * d1);
      }
    }
    out[r] = ssd;
  }
}

}  // namespace tmx

using namespace tmx;

struct tm_ann {
  DevBuf db, q, idx, err, all;
  tm_knn_index_impl *ix = nullptr;
  int n = 0;
  std::mutex mu;
  ~tm_ann() { if (ix) knn_index_destroy(ix); }
};

extern "C" {

tm_ann *ann_kdtree_short_create(int16_t **rows, int n, int dd, int bs, int split) {
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
  (void)eps;  // the reference always passes 0 (exact); a positive eps would only allow a worse answer
  int32_t idx = -1;
  uint32_t e = 0xffffffffu;
  if (!a || !q || ann_kdtree_short_search_batch(a, q, 1, &idx, &e) != TM_OK) idx = -1;
  if (err) *err = e;
  return idx;
}

void ann_kdtree_short_search_multi(tm_ann *a, int32_t *idxs, uint32_t *errs, int cnt, const int16_t *q, uint32_t eps) {
  (void)eps;
  for (int i = 0; i < cnt; i++) { if (idxs) idxs[i] = -1; if (errs) errs[i] = 0xffffffffu; }
  if (!a || !q || !idxs || cnt <= 0 || a->n == 0) return;
  std::lock_guard<std::mutex> lk(a->mu);
  if (a->all.alloc((size_t)a->n * 4) != TM_OK) return;
  if (hipMemcpy(a->q.p, q, 384, hipMemcpyHostToDevice) != hipSuccess) return;
  hipLaunchKernelGGL(k_ssd_one_query, dim3((unsigned)std::min<int64_t>(((int64_t)a->n + 255) / 256, 2048)), dim3(256), 0, nullptr,
                     a->q.as<int16_t>(), a->db.as<int16_t>(), (int64_t)a->n, a->all.as<uint32_t>());
  std::vector<uint32_t> d(a->n);
  if (hipMemcpy(d.data(), a->all.p, (size_t)a->n * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
  std::vector<int32_t> order(a->n);
  for (int i = 0; i < a->n; i++) order[i] = i;
  const int k = std::min(cnt, a->n);
  std::partial_sort(order.begin(), order.begin() + k, order.end(),
                    [&](int32_t x, int32_t y) { return d[x] != d[y] ? d[x] < d[y] : x < y; });  // build's order: (err, idx) ascending
  for (int i = 0; i < k; i++) { idxs[i] = order[i]; if (errs) errs[i] = d[order[i]]; }
}

}  // extern "C"
