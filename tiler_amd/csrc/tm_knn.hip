// tm_knn.hip -- exact nearest-neighbour search of int16[192] tile features as an int8 MFMA distance GEMM.
//
// Replaces ann_kdtree_short_{create,search} (extern.pas:182-184) as used by PrepareReconstruct (tilingencoder.pas:
// 4566-4613) and TFrame.Reconstruct.DoXY (1534-1557): for every query row find the database row minimising
// CompareEuclideanDCTPtr (utils.pas:541-557).  Brute force, bit-exact, ties -> lowest database index.
//
// Scheme (DESIGN.md "KNN"):
//   SSD(q,t) = |q-c|^2 + |t-c|^2 - 2 (q-c).(t-c) for any per-column centre c.  On each SIDE (database, queries) a
//   column whose centred values stay within +-127 fits one int8 digit; the others ("big": data dependent, DC/low
//   frequencies for source tiles, many more for dithered tiles) get a balanced base-256 split v = 256 h + l.  Columns
//   are permuted so each side's big columns are a prefix (the smaller set nested in the larger), HT / HQ chunks of 32:
//      X = 65536 * (T_H . Q_H)[:min] + 256 * (T_L[:HQ] . Q_H + T_H . Q_L[:HT]) + T_L . Q_L
//   = three int32 MFMA accumulators fed by v_mfma_i32_32x32x32_i8, K = 192 + 32 (HT + HQ + min(HT,HQ)) <= 768 bytes.
//   The query digits are stored NEGATED, so one lane computes, with nq2 = 2 * (|q-c|^2 >> 1),
//      d'' = |t-c|^2 + 2 * (acc2<<16 + acc1<<8 + acc0) + nq2  ==  SSD - (|q-c|^2 & 1)   (exact mod 2^32, SSD < 2^31)
//   with three v_lshl_add_u32 + one add per element, and keeps a running (min d'', first tile) per lane.  Database rows ride
//   the MFMA A operand (accumulator rows), queries the B operand (accumulator columns = lanes), so the argmin of a
//   query never leaves its lane until the final 2-lane merge.  A second tiny kernel rescans the winning 32-row
//   tile with the plain int16 SSD to produce (index, error) under the lowest-index rule.
#include <algorithm>
#include <climits>
#include <cstdlib>

#include "tm_common.h"
#include "tm_internal.h"
#include "tm_knn_kernel.h"

namespace tmx {

struct KnnPlan {
  int ht = 6, hq = 6;     // 32-column chunks that carry a high digit on the database / query side (0..6)
  int16_t centre[192];    // per source column
  int16_t perm[192];      // packed position -> source column (columns with high digits first, nested sets)
  int nbig_t = 192, nbig_q = 192;
};

__host__ __device__ inline int knn_tile_bytes(int hch) { return (6 + hch) * 1024 + 128; }

// ---------------------------------------------------------------------------------------------------------------
// per-column min/max over n rows.  192 threads: thread = (row slot 0..7, 16-byte vector 0..23).
__global__ __launch_bounds__(192) void k_col_minmax(const int16_t *__restrict__ feat, int64_t n, int *__restrict__ mn,
                                                    int *__restrict__ mx) {
  __shared__ int s_mn[8][192], s_mx[8][192];
  const int vec = threadIdx.x % 24, slot = threadIdx.x / 24;
  int lmn[8], lmx[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { lmn[i] = INT_MAX; lmx[i] = INT_MIN; }
  for (int64_t row = (int64_t)blockIdx.x * 8 + slot; row < n; row += (int64_t)gridDim.x * 8) {
    const v4i v = *reinterpret_cast<const v4i *>(feat + row * 192 + vec * 8);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int lo = (int)(int16_t)(v[i] & 0xffff), hi = v[i] >> 16;
      lmn[2 * i] = min(lmn[2 * i], lo); lmx[2 * i] = max(lmx[2 * i], lo);
      lmn[2 * i + 1] = min(lmn[2 * i + 1], hi); lmx[2 * i + 1] = max(lmx[2 * i + 1], hi);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) { s_mn[slot][vec * 8 + i] = lmn[i]; s_mx[slot][vec * 8 + i] = lmx[i]; }
  __syncthreads();
  const int c = threadIdx.x;
  int a = INT_MAX, b = INT_MIN;
#pragma unroll
  for (int s = 0; s < 8; s++) { a = min(a, s_mn[s][c]); b = max(b, s_mx[s][c]); }
  if (a != INT_MAX) { atomicMin(&mn[c], a); atomicMax(&mx[c], b); }
}

// ---------------------------------------------------------------------------------------------------------------
// Pack n rows into MFMA fragment order: per 32-row tile [kc][64 lanes][16 B] (lane = half*32 + row) followed by
// 32 u32 norms.  negate=1 (query side): digits of (c - v) and norm >> 1; negate=0 (database): digits of (v - c).
// Rows >= n replicate row n-1 (ties resolve to the lower, real index).  err_flag is set if a digit overflows int8.
__global__ __launch_bounds__(256) void k_knn_pack(const int16_t *__restrict__ feat, int64_t n, int64_t ntiles, int hch, int negate,
                                                  const int16_t *__restrict__ centre, const int16_t *__restrict__ perm,
                                                  uint8_t *__restrict__ out, int *__restrict__ err_flag) {
  __shared__ int16_t s_c[192], s_p[192];
  __shared__ int s_v[32][193];
  for (int i = threadIdx.x; i < 192; i += 256) { s_p[i] = perm[i]; s_c[i] = centre[perm[i]]; }
  const int kch = 6 + hch, tile_bytes = knn_tile_bytes(hch);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    // centred, permuted values of the 32 rows
    for (int i = threadIdx.x; i < 32 * 192; i += 256) {
      const int r = i / 192, p = i - r * 192;
      int64_t row = tile * 32 + r;
      if (row >= n) row = n - 1;
      const int v = (int)feat[row * 192 + s_p[p]] - (int)s_c[p];
      s_v[r][p] = negate ? -v : v;
    }
    __syncthreads();
    uint8_t *obase = out + tile * (int64_t)tile_bytes;
    bool bad = false;
    for (int piece = threadIdx.x; piece < kch * 64; piece += 256) {
      const int kc = piece >> 6, ln = piece & 63, half = ln >> 5, r = ln & 31;
      uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
      for (int b = 0; b < 16; b++) {
        const int kpos = kc * 32 + half * 16 + b;  // byte position along K
        int digit;
        if (kpos < 192) {
          const int v = s_v[r][kpos];
          digit = ((v + 128) & 255) - 128;  // low digit in [-128,127]
          if (kpos >= hch * 32) {           // column without a high digit: must fit
            if (v != digit) bad = true;
          }
        } else {
          const int v = s_v[r][kpos - 192];
          const int lo = ((v + 128) & 255) - 128;
          digit = (v - lo) >> 8;
          if (digit < -128 || digit > 127) bad = true;
        }
        w[b >> 2] |= (uint32_t)(digit & 255) << ((b & 3) * 8);
      }
      *reinterpret_cast<uint4 *>(obase + piece * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (threadIdx.x < 32) {
      uint32_t s = 0;
      for (int p = 0; p < 192; p++) { const int v = s_v[threadIdx.x][p]; s += (uint32_t)(v * v); }
      reinterpret_cast<uint32_t *>(obase + kch * 1024)[threadIdx.x] = negate ? (s >> 1) : s;
    }
    if (bad) atomicOr(err_flag, 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Rescan the winning 32-row tile of each query with the plain SSD (CompareEuclideanDCTPtr, utils.pas:541-557).
// One wave per query, lanes 0..31 = rows of the tile; min by (ssd, row).
__global__ __launch_bounds__(256) void k_knn_refine(const int16_t *__restrict__ queries, int64_t nq, const int16_t *__restrict__ db,
                                                    int64_t nt, const int *__restrict__ best_tile, int *__restrict__ out_idx,
                                                    uint32_t *__restrict__ out_err) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < nq; q += (int64_t)gridDim.x * 4) {
    const int64_t row = (int64_t)best_tile[q] * 32 + (lane & 31);
    unsigned long long key = ~0ull;
    if (lane < 32 && row < nt) {
      const v4i *qp = reinterpret_cast<const v4i *>(queries + q * 192);
      const v4i *tp = reinterpret_cast<const v4i *>(db + row * 192);
      uint32_t ssd = 0;
#pragma unroll 4
      for (int v = 0; v < 24; v++) {
        const v4i a = qp[v], b = tp[v];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int d0 = (int)(int16_t)(a[i] & 0xffff) - (int)(int16_t)(b[i] & 0xffff);
          const int d1 = (a[i] >> 16) - (b[i] >> 16);
          ssd += (uint32_t)(d0 * d0) + (uint32_t)(d1 * d1);
        }
      }
      key = ((unsigned long long)ssd << 32) | (unsigned long long)(uint32_t)row;
    }
    for (int o = 16; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(key, o);
      key = other < key ? other : key;
    }
    if (lane == 0) {
      out_idx[q] = (int)(uint32_t)(key & 0xffffffffull);
      out_err[q] = (uint32_t)(key >> 32);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side

struct ColStats { int mn[192], mx[192]; };

static int col_stats(const void *feat, int64_t n, ColStats *out, DevBuf &scratch, hipStream_t stream) {
  TM_TRY(scratch.alloc(384 * sizeof(int)));
  int init[384];
  for (int i = 0; i < 192; i++) { init[i] = INT_MAX; init[192 + i] = INT_MIN; }
  TM_HIP(hipMemcpyAsync(scratch.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
  if (n > 0) {
    int grid = (int)std::min<int64_t>((n + 7) / 8, 2048);
    hipLaunchKernelGGL(k_col_minmax, dim3(grid), dim3(192), 0, stream, (const int16_t *)feat, n, scratch.as<int>(),
                       scratch.as<int>() + 192);
    TM_HIP(hipGetLastError());
  }
  int res[384];
  TM_HIP(hipMemcpyAsync(res, scratch.p, sizeof(res), hipMemcpyDeviceToHost, stream));
  TM_HIP(hipStreamSynchronize(stream));
  memcpy(out->mn, res, sizeof(int) * 192);
  memcpy(out->mx, res + 192, sizeof(int) * 192);
  return TM_OK;
}

static void merge_stats(ColStats &a, const ColStats &b) {
  for (int i = 0; i < 192; i++) { a.mn[i] = std::min(a.mn[i], b.mn[i]); a.mx[i] = std::max(a.mx[i], b.mx[i]); }
}

// Per-side digit plan.  For every column pick the centre (midpoint of the union, of the database or of the query range)
// that needs the fewest int8 products, then nest the smaller big-set into the larger one so both are prefixes.
static int make_plan(const ColStats &ts, const ColStats &qs, KnnPlan *plan) {
  bool tb[192], qb[192];
  for (int c = 0; c < 192; c++) {
    int tlo = ts.mn[c], thi = ts.mx[c], qlo = qs.mn[c], qhi = qs.mx[c];
    if (tlo > thi) { tlo = qlo; thi = qhi; }
    if (qlo > qhi) { qlo = tlo; qhi = thi; }
    if (tlo > thi) { tlo = thi = qlo = qhi = 0; }
    const int ulo = std::min(tlo, qlo), uhi = std::max(thi, qhi);
    const int cand[3] = {ulo + (uhi - ulo) / 2, tlo + (thi - tlo) / 2, qlo + (qhi - qlo) / 2};
    int best_cost = 99, best_c = cand[0];
    bool bt = true, bq = true;
    for (int k = 0; k < 3; k++) {
      const int cc = cand[k];
      const bool t2 = (thi - cc > 127) || (cc - tlo > 127), q2 = (qhi - cc > 127) || (cc - qlo > 127);
      const int cost = 1 + (t2 ? 1 : 0) + (q2 ? 1 : 0) + (t2 && q2 ? 1 : 0);
      if (cost < best_cost) { best_cost = cost; best_c = cc; bt = t2; bq = q2; }
    }
    plan->centre[c] = (int16_t)best_c;
    tb[c] = bt;
    qb[c] = bq;
  }
  int nt = 0, nq = 0, nu = 0;
  for (int c = 0; c < 192; c++) { nt += tb[c]; nq += qb[c]; nu += (tb[c] || qb[c]); }
  auto chunks = [](int n) { return (n + 31) / 32; };
  // option A: queries' set inside the database's (database digits widened to the union); option B the other way round
  const int costA = chunks(nu) + 2 * chunks(nq), costB = chunks(nu) + 2 * chunks(nt);
  const bool a = costA <= costB;
  const bool *inner = a ? qb : tb;
  int p = 0;
  for (int c = 0; c < 192; c++) if (inner[c]) plan->perm[p++] = (int16_t)c;
  for (int c = 0; c < 192; c++) if (!inner[c] && (tb[c] || qb[c])) plan->perm[p++] = (int16_t)c;
  for (int c = 0; c < 192; c++) if (!tb[c] && !qb[c]) plan->perm[p++] = (int16_t)c;
  plan->ht = a ? chunks(nu) : chunks(nt);
  plan->hq = a ? chunks(nq) : chunks(nu);
  plan->nbig_t = nt;
  plan->nbig_q = nq;
  return TM_OK;
}

// does `plan` represent every value of one side's statistics exactly?  (both signs are checked: queries are negated)
static bool plan_covers(const KnnPlan &plan, const ColStats &st, int hch) {
  for (int p = 0; p < 192; p++) {
    const int c = plan.perm[p];
    if (st.mn[c] > st.mx[c]) continue;
    const int lo = st.mn[c] - plan.centre[c], hi = st.mx[c] - plan.centre[c];
    if (p >= hch * 32) {
      if (lo < -127 || hi > 127) return false;
    } else {
      if (lo < -32000 || hi > 32000) return false;
    }
  }
  return true;
}

struct tm_knn_index_impl {
  const int16_t *db = nullptr;  // borrowed, like ann_kdtree_create borrows its rows (tilingencoder.pas:4600, 4615-4624)
  int64_t nt = 0;
  ColStats tstats;
  KnnPlan plan;
  bool packed = false;
  DevBuf tpack, qpack, plan_dev, scratch, best_key, best_tile, err_flag;
  double last_ms = 0;
  int last_kbytes = 0;
  int64_t last_pairs = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  ~tm_knn_index_impl() {
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
  }
};

static int upload_plan(tm_knn_index_impl *ix, hipStream_t stream) {
  TM_TRY(ix->plan_dev.alloc(384 * sizeof(int16_t)));
  int16_t host[384];
  memcpy(host, ix->plan.centre, sizeof(int16_t) * 192);
  memcpy(host + 192, ix->plan.perm, sizeof(int16_t) * 192);
  TM_HIP(hipMemcpyAsync(ix->plan_dev.p, host, sizeof(host), hipMemcpyHostToDevice, stream));
  TM_HIP(hipStreamSynchronize(stream));  // host[] is on the stack
  return TM_OK;
}

static int run_pack(tm_knn_index_impl *ix, const void *feat, int64_t n, int negate, int hch, DevBuf &out, hipStream_t stream) {
  const int64_t ntiles = (n + 31) / 32;
  TM_TRY(out.alloc((size_t)ntiles * knn_tile_bytes(hch)));
  TM_TRY(ix->err_flag.alloc(sizeof(int)));
  int grid = (int)std::min<int64_t>(ntiles, 4096);
  hipLaunchKernelGGL(k_knn_pack, dim3(grid), dim3(256), 0, stream, (const int16_t *)feat, n, ntiles, hch, negate,
                     ix->plan_dev.as<int16_t>(), ix->plan_dev.as<int16_t>() + 192, out.as<uint8_t>(), ix->err_flag.as<int>());
  TM_HIP(hipGetLastError());
  return TM_OK;
}

static void launch_mfma(int ht, int hq, const KnnLaunch &a) {
  switch (ht) {
    case 0: knn_launch_ht<0>(hq, a); break;
    case 1: knn_launch_ht<1>(hq, a); break;
    case 2: knn_launch_ht<2>(hq, a); break;
    case 3: knn_launch_ht<3>(hq, a); break;
    case 4: knn_launch_ht<4>(hq, a); break;
    case 5: knn_launch_ht<5>(hq, a); break;
    default: knn_launch_ht<6>(hq, a); break;
  }
}

int knn_index_create(const void *db, int64_t nt, hipStream_t stream, tm_knn_index_impl **out) {
  TM_TRY(require_device());
  TM_CHECK(nt >= 0, TM_E_INVAL, "knn: negative row count");
  auto *ix = new tm_knn_index_impl();
  ix->db = (const int16_t *)db;
  ix->nt = nt;
  int rc = col_stats(db, nt, &ix->tstats, ix->scratch, stream);
  if (rc == TM_OK && (hipEventCreate(&ix->ev0) != hipSuccess || hipEventCreate(&ix->ev1) != hipSuccess)) {
    set_error("hipEventCreate failed");
    rc = TM_E_HIP;
  }
  if (rc != TM_OK) { delete ix; return rc; }
  *out = ix;
  return TM_OK;
}

void knn_index_destroy(tm_knn_index_impl *ix) { delete ix; }

int knn_index_search(tm_knn_index_impl *ix, const void *queries, int64_t nq, void *out_idx, void *out_err, hipStream_t stream) {
  TM_CHECK(ix != nullptr, TM_E_INVAL, "knn: null index");
  TM_CHECK(nq >= 0, TM_E_INVAL, "knn: negative query count");
  if (nq == 0) return TM_OK;
  if (ix->nt == 0) {  // ANN on an empty tree: the caller treats idx outside [0,T) as "none" (tilingencoder.pas:1549-1557)
    TM_HIP(hipMemsetAsync(out_idx, 0xff, (size_t)nq * 4, stream));
    TM_HIP(hipMemsetAsync(out_err, 0xff, (size_t)nq * 4, stream));
    return TM_OK;
  }
  ColStats qs;
  TM_TRY(col_stats(queries, nq, &qs, ix->scratch, stream));
  TM_TRY(ix->err_flag.alloc(sizeof(int)));
  TM_HIP(hipMemsetAsync(ix->err_flag.p, 0, sizeof(int), stream));  // both pack passes below report into it
  if (!ix->packed || !plan_covers(ix->plan, qs, ix->plan.hq)) {
    TM_TRY(make_plan(ix->tstats, qs, &ix->plan));
    TM_CHECK(plan_covers(ix->plan, ix->tstats, ix->plan.ht) && plan_covers(ix->plan, qs, ix->plan.hq), TM_E_UNSUPPORTED,
             "knn: feature range exceeds the exact two-digit int8 split");
    if (getenv("TM_KNN_DEBUG"))
      fprintf(stderr, "[tm_knn] nq=%lld nt=%lld big columns: database %d, queries %d -> HT=%d HQ=%d K=%d bytes\n", (long long)nq,
              (long long)ix->nt, ix->plan.nbig_t, ix->plan.nbig_q, ix->plan.ht, ix->plan.hq,
              192 + 32 * (ix->plan.ht + ix->plan.hq + std::min(ix->plan.ht, ix->plan.hq)));
    TM_TRY(upload_plan(ix, stream));
    TM_TRY(run_pack(ix, ix->db, ix->nt, 0, ix->plan.ht, ix->tpack, stream));
    ix->packed = true;
  }
  TM_TRY(run_pack(ix, queries, nq, 1, ix->plan.hq, ix->qpack, stream));
  const int64_t nqt = (nq + 31) / 32, ntt = (ix->nt + 31) / 32;
  TM_TRY(ix->best_key.alloc((size_t)nqt * 32 * 4));
  TM_TRY(ix->best_tile.alloc((size_t)nqt * 32 * 4));
  int dev = 0, ncu = 256;
  TM_HIP(hipGetDevice(&dev));
  TM_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  TM_HIP(hipEventRecord(ix->ev0, stream));
  int *bt = ix->best_tile.as<int>();
  launch_mfma(ix->plan.ht, ix->plan.hq,
              KnnLaunch{ix->tpack.as<uint8_t>(), 0, ntt, ix->qpack.as<uint8_t>(), nqt, ix->best_key.as<int>(), bt, 0, ncu, stream});
  TM_HIP(hipGetLastError());
  TM_HIP(hipEventRecord(ix->ev1, stream));
  {
    int grid = (int)std::min<int64_t>((nq + 3) / 4, 8192);
    hipLaunchKernelGGL(k_knn_refine, dim3(grid), dim3(256), 0, stream, (const int16_t *)queries, nq, ix->db, ix->nt, bt,
                       (int *)out_idx, (uint32_t *)out_err);
    TM_HIP(hipGetLastError());
  }
  int flag = 0;
  TM_HIP(hipMemcpyAsync(&flag, ix->err_flag.p, sizeof(int), hipMemcpyDeviceToHost, stream));
  TM_HIP(hipStreamSynchronize(stream));
  TM_CHECK(flag == 0, TM_E_UNSUPPORTED, "knn: feature range exceeds the exact two-digit int8 split (|v-c| >= 32640)");
  float ms = 0;
  TM_HIP(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
  ix->last_ms = ms;
  ix->last_kbytes = 192 + 32 * (ix->plan.ht + ix->plan.hq + std::min(ix->plan.ht, ix->plan.hq));
  ix->last_pairs = nq * ix->nt;
  return TM_OK;
}

void knn_index_stats(tm_knn_index_impl *ix, double *ms, int *kbytes, int64_t *pairs) {
  if (ms) *ms = ix->last_ms;
  if (kbytes) *kbytes = ix->last_kbytes;
  if (pairs) *pairs = ix->last_pairs;
}

}  // namespace tmx
