// tm_knn.hip -- exact nearest-neighbour search of int16[192] tile features as an int8 MFMA distance GEMM.
//
// Replaces ann_kdtree_short_{create,search} (extern.pas:182-184) as used by PrepareReconstruct (tilingencoder.pas:
// 4566-4613) and TFrame.Reconstruct.DoXY (1534-1557): for every query row find the database row minimising
// CompareEuclideanDCTPtr (utils.pas:541-557).  Brute force, bit-exact, ties -> lowest database index.
//
// Scheme (DESIGN.md "KNN"):
//   SSD(q,t) = |q-c|^2 + |t-c|^2 - 2 (q-c).(t-c) for any per-column centre c.  Columns whose value range over
//   Q u T is <= 254 fit one int8 digit after centring; the others ("big", data dependent, mostly DC/low
//   frequencies) get a balanced base-256 split v = 256 h + l.  With big columns permuted first:
//      X = 65536 * (T_H . Q_H) + 256 * (T_L[:H] . Q_H + T_H . Q_L[:H]) + T_L . Q_L
//   = three int32 MFMA accumulators fed by v_mfma_i32_32x32x32_i8, K = 192 + 3H bytes instead of 4*192.
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

namespace tmx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct KnnPlan {
  int hch = 6;            // H / 32, in {0,1,2,3,4,6}
  int16_t centre[192];    // per source column
  int16_t perm[192];      // packed position -> source column (big columns first)
  int nbig = 192;
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
// The distance GEMM.  NW waves per workgroup, each holding NQ query sub-tiles (32 queries) as MFMA B fragments in
// registers for the whole database sweep; database tiles (32 rows) stream global -> registers -> LDS (double
// buffered) and are read back as A fragments with lane-linear ds_read_b128.
template <int HCH, int NQ, int NW>
__global__ __launch_bounds__(NW * 64) void k_knn_mfma(const uint8_t *__restrict__ tpack, int64_t tile_begin, int64_t tile_end,
                                                      const uint8_t *__restrict__ qpack, int64_t n_qtiles,
                                                      int *__restrict__ best_key, int *__restrict__ best_tile, int accumulate) {
  constexpr int KCH = 6 + HCH;
  constexpr int TILE_BYTES = KCH * 1024 + 128;
  constexpr int TILE_VEC = TILE_BYTES / 16;
  constexpr int NT = NW * 64;
  constexpr int NST = (TILE_VEC + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) uint8_t lds[2][TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  constexpr int QT_PER_WG = NW * NQ;
  const int64_t n_wg_tiles = (n_qtiles + QT_PER_WG - 1) / QT_PER_WG;

  for (int64_t wgt = blockIdx.x; wgt < n_wg_tiles; wgt += gridDim.x) {
    v4i bq[NQ][KCH];
    int nq2[NQ], best[NQ], bestt[NQ];
    int64_t qtile[NQ];
#pragma unroll
    for (int s = 0; s < NQ; s++) {
      qtile[s] = wgt * QT_PER_WG + wave * NQ + s;
      const int64_t qt = qtile[s] < n_qtiles ? qtile[s] : n_qtiles - 1;
      const uint8_t *qb = qpack + qt * (int64_t)TILE_BYTES;
#pragma unroll
      for (int kc = 0; kc < KCH; kc++) bq[s][kc] = *reinterpret_cast<const v4i *>(qb + (kc * 64 + lane) * 16);
      nq2[s] = reinterpret_cast<const int *>(qb + KCH * 1024)[lane & 31] << 1;  // 2*(|q-c|^2 >> 1)
      best[s] = INT_MAX;
      bestt[s] = INT_MAX;
    }

    v4i st[NST];
    {  // prologue: first database tile -> LDS buffer 0
      const uint8_t *src = tpack + tile_begin * (int64_t)TILE_BYTES;
#pragma unroll
      for (int i = 0; i < NST; i++)
        if (tid + i * NT < TILE_VEC) st[i] = *reinterpret_cast<const v4i *>(src + (tid + i * NT) * 16);
#pragma unroll
      for (int i = 0; i < NST; i++)
        if (tid + i * NT < TILE_VEC) *reinterpret_cast<v4i *>(&lds[0][(tid + i * NT) * 16]) = st[i];
    }
    __syncthreads();

    for (int64_t t = tile_begin; t < tile_end; t++) {
      const int cur = (int)((t - tile_begin) & 1);
      const bool more = t + 1 < tile_end;
      if (more) {
        const uint8_t *src = tpack + (t + 1) * (int64_t)TILE_BYTES;
#pragma unroll
        for (int i = 0; i < NST; i++)
          if (tid + i * NT < TILE_VEC) st[i] = *reinterpret_cast<const v4i *>(src + (tid + i * NT) * 16);
      }
      const uint8_t *L = lds[cur];
      // accumulator row of register r: (r&3) + 8*(r>>2) + 4*half  -> norms as four 16-byte reads
      int nt[16];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const v4i x = *reinterpret_cast<const v4i *>(L + KCH * 1024 + (g * 8 + half * 4) * 4);
        nt[g * 4] = x[0]; nt[g * 4 + 1] = x[1]; nt[g * 4 + 2] = x[2]; nt[g * 4 + 3] = x[3];
      }
      // 2-deep software pipeline over the query sub-tiles: the MFMAs of sub-tile s run beside the VALU epilogue of
      // sub-tile s-1; the sched_barrier keeps hipcc from interleaving more sub-tiles (register budget).
      v16i acc0[2], acc1[2], acc2[2];
#pragma unroll
      for (int s = 0; s <= NQ; s++) {
        if (s < NQ) {
          const int b = s & 1;
#pragma unroll
          for (int r = 0; r < 16; r++) { acc0[b][r] = 0; acc1[b][r] = 0; acc2[b][r] = 0; }
#pragma unroll
          for (int kc = 0; kc < 6; kc++) {
            const v4i a = *reinterpret_cast<const v4i *>(L + (kc * 64 + lane) * 16);  // T_L chunk
            acc0[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][kc], acc0[b], 0, 0, 0);                     // T_L . Q_L
            if (kc < HCH) acc1[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][6 + kc], acc1[b], 0, 0, 0);  // T_L . Q_H
          }
#pragma unroll
          for (int kc = 0; kc < HCH; kc++) {
            const v4i a = *reinterpret_cast<const v4i *>(L + ((6 + kc) * 64 + lane) * 16);  // T_H chunk
            acc1[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][kc], acc1[b], 0, 0, 0);      // T_H . Q_L
            acc2[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][6 + kc], acc2[b], 0, 0, 0);  // T_H . Q_H
          }
        }
        if (s > 0) {
          const int b = (s - 1) & 1;
          int m = INT_MAX;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            int x = acc0[b][r];
            if (HCH > 0) x = (int)(((unsigned)(((unsigned)acc2[b][r] << 8) + (unsigned)acc1[b][r]) << 8) + (unsigned)acc0[b][r]);
            const int d = (int)(((unsigned)x << 1) + (unsigned)nt[r] + (unsigned)nq2[s - 1]);
            m = min(m, d);
          }
          if (m < best[s - 1]) { best[s - 1] = m; bestt[s - 1] = (int)t; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) {
#pragma unroll
        for (int i = 0; i < NST; i++)
          if (tid + i * NT < TILE_VEC) *reinterpret_cast<v4i *>(&lds[cur ^ 1][(tid + i * NT) * 16]) = st[i];
      }
      __syncthreads();
    }

#pragma unroll
    for (int s = 0; s < NQ; s++) {
      const int ob = __shfl_xor(best[s], 32), ot = __shfl_xor(bestt[s], 32);
      if (ob < best[s] || (ob == best[s] && ot < bestt[s])) { best[s] = ob; bestt[s] = ot; }
      if (lane < 32 && qtile[s] < n_qtiles) {
        const int64_t q = qtile[s] * 32 + lane;
        if (accumulate) {
          const int pk = best_key[q], pt = best_tile[q];
          if (pk < best[s] || (pk == best[s] && pt < bestt[s])) { best[s] = pk; bestt[s] = pt; }
        }
        best_key[q] = best[s];
        best_tile[q] = bestt[s];
      }
    }
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

static const int kHchVariants[] = {0, 1, 2, 3, 4, 6};

static int make_plan(const ColStats &st, KnnPlan *plan) {
  int big[192], nbig = 0, small_[192], nsmall = 0;
  for (int c = 0; c < 192; c++) {
    int lo = st.mn[c], hi = st.mx[c];
    if (lo > hi) { lo = 0; hi = 0; }  // no rows
    const int range = hi - lo;
    plan->centre[c] = (int16_t)(lo + range / 2);
    if (range > 254) big[nbig++] = c; else small_[nsmall++] = c;
  }
  int hch = -1;
  for (int v : kHchVariants)
    if (nbig <= v * 32) { hch = v; break; }
  TM_CHECK(hch >= 0, TM_E_INVAL, "knn plan: %d big columns", nbig);
  int p = 0;
  for (int i = 0; i < nbig; i++) plan->perm[p++] = (int16_t)big[i];
  for (int i = 0; i < nsmall; i++) plan->perm[p++] = (int16_t)small_[i];
  plan->hch = hch;
  plan->nbig = nbig;
  return TM_OK;
}

// does `plan` represent every value of `st` exactly?
static bool plan_covers(const KnnPlan &plan, const ColStats &st) {
  for (int p = 0; p < 192; p++) {
    const int c = plan.perm[p];
    if (st.mn[c] > st.mx[c]) continue;
    const int lo = st.mn[c] - plan.centre[c], hi = st.mx[c] - plan.centre[c];
    if (p >= plan.hch * 32) {
      if (lo < -127 || hi > 127) return false;  // both signs are packed (database v-c, queries c-v)
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

static int run_pack(tm_knn_index_impl *ix, const void *feat, int64_t n, int negate, DevBuf &out, hipStream_t stream) {
  const int64_t ntiles = (n + 31) / 32;
  TM_TRY(out.alloc((size_t)ntiles * knn_tile_bytes(ix->plan.hch)));
  TM_TRY(ix->err_flag.alloc(sizeof(int)));
  int grid = (int)std::min<int64_t>(ntiles, 4096);
  hipLaunchKernelGGL(k_knn_pack, dim3(grid), dim3(256), 0, stream, (const int16_t *)feat, n, ntiles, ix->plan.hch, negate,
                     ix->plan_dev.as<int16_t>(), ix->plan_dev.as<int16_t>() + 192, out.as<uint8_t>(), ix->err_flag.as<int>());
  TM_HIP(hipGetLastError());
  return TM_OK;
}

// Workgroup shapes: variant 0 = 8 waves x 2 query sub-tiles (2 waves/SIMD, <=256 VGPRs), variant 1 = 4 waves x 4
// (1 wave/SIMD, 512 VGPRs), variant 2 = 4 waves x 3.  TM_KNN_VARIANT overrides the default for A/B runs.
static int knn_variant() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("TM_KNN_VARIANT");
    v = e ? atoi(e) : 0;
    if (v < 0 || v > 2) v = 0;
  }
  return v;
}

template <int HCH, int NQ, int NW>
static void launch_mfma_v(const uint8_t *tpack, int64_t tb, int64_t te, const uint8_t *qpack, int64_t nqt, int *bk, int *bt, int acc,
                          int ncu, hipStream_t stream) {
  const int64_t wg_tiles = (nqt + NQ * NW - 1) / (NQ * NW);
  const int grid = (int)std::min<int64_t>(wg_tiles, ncu);
  hipLaunchKernelGGL((k_knn_mfma<HCH, NQ, NW>), dim3(grid), dim3(NW * 64), 0, stream, tpack, tb, te, qpack, nqt, bk, bt, acc);
}

template <int HCH> static void launch_mfma(const uint8_t *tpack, int64_t tb, int64_t te, const uint8_t *qpack, int64_t nqt, int *bk,
                                           int *bt, int acc, int ncu, hipStream_t stream) {
  switch (knn_variant()) {
    case 0: launch_mfma_v<HCH, 2, 8>(tpack, tb, te, qpack, nqt, bk, bt, acc, ncu, stream); break;
    case 2: launch_mfma_v<HCH, 3, 4>(tpack, tb, te, qpack, nqt, bk, bt, acc, ncu, stream); break;
    default: launch_mfma_v<HCH, 4, 4>(tpack, tb, te, qpack, nqt, bk, bt, acc, ncu, stream); break;
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
  if (!ix->packed || !plan_covers(ix->plan, qs)) {
    ColStats u = ix->tstats;
    merge_stats(u, qs);
    TM_TRY(make_plan(u, &ix->plan));
    TM_TRY(upload_plan(ix, stream));
    TM_TRY(run_pack(ix, ix->db, ix->nt, 0, ix->tpack, stream));
    ix->packed = true;
  }
  TM_TRY(run_pack(ix, queries, nq, 1, ix->qpack, stream));
  const int64_t nqt = (nq + 31) / 32, ntt = (ix->nt + 31) / 32;
  TM_TRY(ix->best_key.alloc((size_t)nqt * 32 * 4));
  TM_TRY(ix->best_tile.alloc((size_t)nqt * 32 * 4));
  int dev = 0, ncu = 256;
  TM_HIP(hipGetDevice(&dev));
  TM_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  TM_HIP(hipEventRecord(ix->ev0, stream));
  const uint8_t *tp = ix->tpack.as<uint8_t>(), *qp = ix->qpack.as<uint8_t>();
  int *bk = ix->best_key.as<int>(), *bt = ix->best_tile.as<int>();
  switch (ix->plan.hch) {
    case 0: launch_mfma<0>(tp, 0, ntt, qp, nqt, bk, bt, 0, ncu, stream); break;
    case 1: launch_mfma<1>(tp, 0, ntt, qp, nqt, bk, bt, 0, ncu, stream); break;
    case 2: launch_mfma<2>(tp, 0, ntt, qp, nqt, bk, bt, 0, ncu, stream); break;
    case 3: launch_mfma<3>(tp, 0, ntt, qp, nqt, bk, bt, 0, ncu, stream); break;
    case 4: launch_mfma<4>(tp, 0, ntt, qp, nqt, bk, bt, 0, ncu, stream); break;
    default: launch_mfma<6>(tp, 0, ntt, qp, nqt, bk, bt, 0, ncu, stream); break;
  }
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
  ix->last_kbytes = 192 + 3 * 32 * ix->plan.hch;
  ix->last_pairs = nq * ix->nt;
  return TM_OK;
}

void knn_index_stats(tm_knn_index_impl *ix, double *ms, int *kbytes, int64_t *pairs) {
  if (ms) *ms = ix->last_ms;
  if (kbytes) *kbytes = ix->last_kbytes;
  if (pairs) *pairs = ix->last_pairs;
}

}  // namespace tmx
