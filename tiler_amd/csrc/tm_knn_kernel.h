// tm_knn_kernel.h -- the int8 MFMA distance GEMM of the KNN stage (see tm_knn.hip for the scheme).  Included by the
// tm_knn_k<HT>.hip translation units, each instantiating one database high-chunk count HT with every query
// high-chunk count HQ (the per-side digit plan makes both data dependent).
#pragma once
#include <climits>

#include "tm_common.h"

namespace tmx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int KNN_NQ = 2;  // query sub-tiles (32 queries each) per wave
constexpr int KNN_NW = 4;  // waves per workgroup (256 queries; two workgroups share a CU)

// Database tiles (32 rows): [6 low chunks | HT high chunks] x [64 lanes] x 16 B, then 32 u32 norms |t-c|^2.
// Query tiles (32 queries): [6 low chunks | HQ high chunks] of the NEGATED centred values, then 32 u32 (|q-c|^2 >> 1).
// Columns are permuted so that the columns carrying a high digit are a prefix on each side (nested sets).
//   acc0 = T_L . Q_L (6 chunks)      acc1 = T_L[:HQ] . Q_H + T_H . Q_L[:HT]      acc2 = T_H[:m] . Q_H[:m], m = min(HT,HQ)
//   d''  = |t-c|^2 + 2*(acc2<<16 + acc1<<8 + acc0) + 2*(|q-c|^2>>1) = SSD - (|q-c|^2 & 1), exact mod 2^32.
//
// Pruned scan.  Both sides arrive sorted along a Morton curve of the three widest feature columns (the Y/U/V DC
// terms for tile features), so 32 consecutive rows form a compact box.  For ND columns the database keeps per-tile
// bounding boxes; the squared box-to-box distance over those columns is a lower bound of every SSD between a
// workgroup's (or wave's) queries and a tile's rows.  A workgroup first visits the K0 tiles nearest on the curve to
// get good running bests, then sweeps the tile list chunk by chunk: all threads test boxes in parallel and compact the
// survivors into LDS, survivors are re-tested against the (shrinking) largest best before being staged, and a wave
// skips the MFMAs of a staged tile its own queries rule out.  Exactness: the minimum VALUE is exact (a tile is only
// skipped when bound > best + 1); when a second tile reaches the same value the lane raises a tie flag and the refine
// stage settles the lowest-index rule.
constexpr int KNN_ND = 6;        // bounding-box columns
constexpr int KNN_K0 = 8;        // tiles visited first, around the workgroup's position on the curve
constexpr int KNN_CHUNK = 1024;  // tiles tested per compaction round (bests are re-read for each round)

struct KnnBoxes {
  const int *lo, *hi;   // [KNN_ND][n_ttiles] bounding boxes of the database tiles
  const uint32_t *tkey; // [n_ttiles] curve key of each tile's first row (ascending)
  int col[KNN_ND];      // source feature column of each box dimension
};

// Compacts into s_list the tiles of [chunk_base, chunk_base + KNN_CHUNK) that at least one query sub-tile of the
// workgroup can still use: squared box-to-box distance over the KNN_ND box columns <= that sub-tile's largest running
// best + 1 (d'' = SSD - parity, so SSD <= d'' + 1).  Kept out of line so its registers do not count against the MFMA loop.
__device__ __attribute__((noinline)) int knn_build_list(const int *__restrict__ box_lo, const int *__restrict__ box_hi, int64_t n_ttiles,
                                                        int chunk_base, int r0a, int r0b, int prune, const int *s_box_lo,
                                                        const int *s_box_hi, const int *s_smax, uint16_t *s_list, int *s_cnt) {
  constexpr int NS = KNN_NW * KNN_NQ, ND = KNN_ND, NT = KNN_NW * 64;
  const int tid = threadIdx.x;
  __syncthreads();
  if (tid == 0) *s_cnt = 0;
  __syncthreads();
  for (int k = tid; k < KNN_CHUNK && chunk_base + k < n_ttiles; k += NT) {
    const int t = chunk_base + k;
    if (t >= r0a && t < r0b) continue;  // done in round 0
    bool keep = !prune;
    if (prune) {
      int tlo[ND], thi[ND];
#pragma unroll
      for (int d = 0; d < ND; d++) { tlo[d] = box_lo[(int64_t)d * n_ttiles + t]; thi[d] = box_hi[(int64_t)d * n_ttiles + t]; }
      for (int q = 0; q < NS && !keep; q++) {
        long long lb = 0;
#pragma unroll
        for (int d = 0; d < ND; d++) {
          const long long g = max(0, max(tlo[d] - s_box_hi[q * ND + d], s_box_lo[q * ND + d] - thi[d]));
          lb += g * g;
        }
        keep = lb <= (long long)s_smax[q] + 1;
      }
    }
    if (keep) s_list[atomicAdd(s_cnt, 1)] = (uint16_t)k;
  }
  __syncthreads();
  return *s_cnt;
}

template <int HT, int HQ>
__global__ __launch_bounds__(KNN_NW * 64, 2) void k_knn_mfma(const uint8_t *__restrict__ tpack, int64_t n_ttiles, KnnBoxes bx,
                                                          const uint8_t *__restrict__ qpack, int64_t n_qtiles,
                                                          const int16_t *__restrict__ queries, const uint32_t *__restrict__ qperm,
                                                          const uint32_t *__restrict__ qkey, int64_t nq, int prune,
                                                          int *__restrict__ best_key, int *__restrict__ best_tile,
                                                          unsigned long long *__restrict__ visited) {
  constexpr int NQ = KNN_NQ, NW = KNN_NW, ND = KNN_ND;
  constexpr int KT = 6 + HT, KQ = 6 + HQ, HM = HT < HQ ? HT : HQ;
  constexpr int T_BYTES = KT * 1024 + 128 + 64, Q_BYTES = KQ * 1024 + 128;  // database tiles carry their box (2*ND ints) too
  constexpr int TILE_VEC = T_BYTES / 16;
  constexpr int NT = NW * 64;
  constexpr int NST = (TILE_VEC + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) uint8_t lds[2][T_BYTES];
  __shared__ int s_smax[2][NW][NQ];  // largest running best of each query sub-tile, double buffered
  __shared__ int s_box[2][NW][NQ][ND];  // [lo|hi][wave][sub-tile][dim] query boxes
  __shared__ int s_ctl[4];
  __shared__ uint16_t s_list[KNN_CHUNK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  constexpr int QT_PER_WG = NW * NQ;
  const int64_t wgt = blockIdx.x;

  v4i bq[NQ][KQ];
  int nq2[NQ], best[NQ], bestt[NQ], tie[NQ];  // bestt = tile << 5 | row of the first minimum
  int64_t qtile[NQ];
#pragma unroll
  for (int s = 0; s < NQ; s++) {
    qtile[s] = wgt * QT_PER_WG + wave * NQ + s;
    const int64_t qt = qtile[s] < n_qtiles ? qtile[s] : n_qtiles - 1;
    const uint8_t *qb = qpack + qt * (int64_t)Q_BYTES;
#pragma unroll
    for (int kc = 0; kc < KQ; kc++) bq[s][kc] = *reinterpret_cast<const v4i *>(qb + (kc * 64 + lane) * 16);
    nq2[s] = reinterpret_cast<const int *>(qb + KQ * 1024)[lane & 31] & ~1;  // 2*(|q-c|^2 >> 1); the parity bit returns in the refine stage
    best[s] = INT_MAX;
    bestt[s] = INT_MAX;
    tie[s] = 0;
  }
  // bounding boxes: per query sub-tile (lanes 0..31 / 32..63 hold sub-tile 0 / 1) and for the whole workgroup
  {
    static_assert(NQ == 2, "lane <-> query mapping below assumes two sub-tiles per wave");
    const int64_t p = min(wgt * QT_PER_WG * 32 + (int64_t)wave * NQ * 32 + lane, nq - 1);
    const int16_t *row = queries + (int64_t)qperm[p] * 192;
#pragma unroll
    for (int d = 0; d < ND; d++) {
      int lo = row[bx.col[d]], hi = lo;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
      if ((lane & 31) == 0) { s_box[0][wave][half][d] = lo; s_box[1][wave][half][d] = hi; }
    }
  }
  if (tid == 0) {  // position of the workgroup's first query on the curve: last tile whose first key <= it
    const uint32_t k0 = qkey[wgt * QT_PER_WG * 32];
    int64_t lo = 0, hi = n_ttiles;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (bx.tkey[mid] <= k0) lo = mid + 1; else hi = mid; }
    int start = (int)max((int64_t)0, lo - 1 - KNN_K0 / 2);
    start = (int)min((int64_t)start, max((int64_t)0, n_ttiles - KNN_K0));
    s_ctl[0] = start;
  }
  if (tid < NW * NQ) { (&s_smax[0][0][0])[tid] = INT_MAX; (&s_smax[1][0][0])[tid] = INT_MAX; }
  __syncthreads();
  const int r0a = s_ctl[0], r0b = (int)min((int64_t)r0a + KNN_K0, n_ttiles);  // round-0 tiles [r0a, r0b)
  long long nvisit = 0, nstaged = 0;

  // candidate iterator: round 0 = [r0a, r0b); then chunks of the tile list, compacted into s_list by all threads
  int phase = 0, r0next = r0a, chunk_base = 0, list_n = 0, list_i = 0;
  auto next_tile = [&](int sbuf) -> int {  // called by every thread at the same point (contains barriers)
    while (true) {
      if (phase == 0) {
        if (r0next < r0b) return r0next++;
        phase = 1;
        chunk_base = -KNN_CHUNK;
        list_n = list_i = 0;
      }
      if (list_i < list_n) return chunk_base + s_list[list_i++];
      chunk_base += KNN_CHUNK;
      if (chunk_base >= n_ttiles) return -1;
      list_n = knn_build_list(bx.lo, bx.hi, n_ttiles, chunk_base, r0a, r0b, prune, &s_box[0][0][0][0], &s_box[1][0][0][0],
                              &s_smax[sbuf][0][0], s_list, &s_ctl[1]);
      list_i = 0;
    }
  };

  v4i st[NST];
  int cur_tile = next_tile(0);
  if (cur_tile >= 0) {  // prologue: first database tile -> LDS buffer 0
    const uint8_t *src = tpack + cur_tile * (int64_t)T_BYTES;
#pragma unroll
    for (int i = 0; i < NST; i++)
      if (tid + i * NT < TILE_VEC) st[i] = *reinterpret_cast<const v4i *>(src + (tid + i * NT) * 16);
#pragma unroll
    for (int i = 0; i < NST; i++)
      if (tid + i * NT < TILE_VEC) *reinterpret_cast<v4i *>(&lds[0][(tid + i * NT) * 16]) = st[i];
  }
  __syncthreads();

  int cur = 0;
  while (cur_tile >= 0) {
    nstaged++;
    // list rounds read the bests published one iteration ago (stale = larger = prunes less: still exact)
    const int nxt_tile = next_tile(cur);
    if (nxt_tile >= 0) {
      const uint8_t *src = tpack + nxt_tile * (int64_t)T_BYTES;
#pragma unroll
      for (int i = 0; i < NST; i++)
        if (tid + i * NT < TILE_VEC) st[i] = *reinterpret_cast<const v4i *>(src + (tid + i * NT) * 16);
    }
    const uint8_t *L = lds[cur];
    // sub-tile level skip: can this tile still matter for any of the 32 queries of sub-tile s?  (tile box rides in LDS)
    bool do_sub[NQ];
    bool do_tile = false;
#pragma unroll
    for (int s = 0; s < NQ; s++) {
      int smax = best[s];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) smax = max(smax, __shfl_xor(smax, o));
      long long lb = 0;
      if (prune) {
        const int *tb = reinterpret_cast<const int *>(L + KT * 1024 + 128);
#pragma unroll
        for (int d = 0; d < ND; d++) {
          const long long g = max(0, max(tb[d] - s_box[1][wave][s][d], s_box[0][wave][s][d] - tb[ND + d]));
          lb += g * g;
        }
      }
      do_sub[s] = lb <= (long long)smax + 1;
      do_tile |= do_sub[s];
    }
    if (do_tile) {
      // accumulator row of register r: (r&3) + 8*(r>>2) + 4*half  -> norms as four 16-byte reads
      int nt[16];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const v4i x = *reinterpret_cast<const v4i *>(L + KT * 1024 + (g * 8 + half * 4) * 4);
        nt[g * 4] = x[0]; nt[g * 4 + 1] = x[1]; nt[g * 4 + 2] = x[2]; nt[g * 4 + 3] = x[3];
      }
      // 2-deep software pipeline over the query sub-tiles: the MFMAs of sub-tile s run beside the VALU epilogue of
      // sub-tile s-1 (and beside the other wave of this SIMD).
      v16i acc0[2], acc1[2], acc2[2];
#pragma unroll
      for (int s = 0; s <= NQ; s++) {
        if (s < NQ && do_sub[s]) {
          nvisit++;
          const int b = s & 1;
#pragma unroll
          for (int r = 0; r < 16; r++) { acc0[b][r] = 0; acc1[b][r] = 0; acc2[b][r] = 0; }
#pragma unroll
          for (int kc = 0; kc < 6; kc++) {
            const v4i a = *reinterpret_cast<const v4i *>(L + (kc * 64 + lane) * 16);  // T_L chunk
            acc0[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][kc], acc0[b], 0, 0, 0);                    // T_L . Q_L
            if (kc < HQ) acc1[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][6 + kc], acc1[b], 0, 0, 0);  // T_L . Q_H
          }
#pragma unroll
          for (int kc = 0; kc < HT; kc++) {
            const v4i a = *reinterpret_cast<const v4i *>(L + ((6 + kc) * 64 + lane) * 16);  // T_H chunk
            acc1[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][kc], acc1[b], 0, 0, 0);                    // T_H . Q_L
            if (kc < HM) acc2[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][6 + kc], acc2[b], 0, 0, 0);  // T_H . Q_H
          }
        }
        if (s > 0 && do_sub[s > 0 ? s - 1 : 0]) {
          const int b = (s - 1) & 1;
          int m = INT_MAX;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            unsigned x = (unsigned)acc0[b][r];
            if (HT + HQ > 0) {
              unsigned hi = (unsigned)acc1[b][r];
              if (HM > 0) hi += (unsigned)acc2[b][r] << 8;
              x += hi << 8;
            }
            const int d = (int)((x << 1) + (unsigned)nt[r] + (unsigned)nq2[s - 1]);
            m = min(m, d);
          }
          if (m == best[s - 1]) tie[s - 1] = 1;  // another tile reaches the same value
          if (__builtin_amdgcn_ballot_w64(m < best[s - 1])) {  // some lane improves (rare once the bests have settled)
            int row = 0, cnt = 0;
#pragma unroll
            for (int r = 15; r >= 0; r--) {
              unsigned x = (unsigned)acc0[b][r];
              if (HT + HQ > 0) {
                unsigned hi = (unsigned)acc1[b][r];
                if (HM > 0) hi += (unsigned)acc2[b][r] << 8;
                x += hi << 8;
              }
              const int d = (int)((x << 1) + (unsigned)nt[r] + (unsigned)nq2[s - 1]);
              if (d == m) { row = (r & 3) + 8 * (r >> 2) + 4 * half; cnt++; }
            }
            if (m < best[s - 1]) { best[s - 1] = m; bestt[s - 1] = (cur_tile << 5) | row; tie[s - 1] = cnt > 1; }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int s = 0; s < NQ; s++) {
      int smax = best[s];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) smax = max(smax, __shfl_xor(smax, o));
      if (lane == 0) s_smax[cur ^ 1][wave][s] = smax;
    }
    if (nxt_tile >= 0) {
#pragma unroll
      for (int i = 0; i < NST; i++)
        if (tid + i * NT < TILE_VEC) *reinterpret_cast<v4i *>(&lds[cur ^ 1][(tid + i * NT) * 16]) = st[i];
    }
    __syncthreads();
    cur ^= 1;
    cur_tile = nxt_tile;
  }

#pragma unroll
  for (int s = 0; s < NQ; s++) {
    const int ob = __shfl_xor(best[s], 32), ot = __shfl_xor(bestt[s], 32), oti = __shfl_xor(tie[s], 32);
    if (ob == best[s]) {
      tie[s] = 1;  // two rows (of one tile or two) reach the minimum: settled by original index later
      if (ot < bestt[s]) bestt[s] = ot;
    } else if (ob < best[s]) {
      best[s] = ob; bestt[s] = ot; tie[s] = oti;
    }
    if (lane < 32 && qtile[s] < n_qtiles) {
      const int64_t q = qtile[s] * 32 + lane;
      best_key[q] = best[s];
      best_tile[q] = (bestt[s] & 0x3fffffff) | (tie[s] ? (1 << 30) : 0);  // sorted row of the first minimum; bit 30: tie flag
    }
  }
  if (visited && lane == 0) atomicAdd(visited, (unsigned long long)nvisit);
  if (visited && tid == 0) atomicAdd(visited + 1, (unsigned long long)nstaged);
}

struct KnnLaunch {
  const uint8_t *tpack; int64_t n_ttiles; KnnBoxes bx;
  const uint8_t *qpack; int64_t n_qtiles; const int16_t *queries; const uint32_t *qperm, *qkey; int64_t nq; int prune;
  int *best_key, *best_tile; unsigned long long *visited; hipStream_t stream;
};

// one per HT, defined in tm_knn_k<HT>.hip
template <int HT> void knn_launch_ht(int hq, const KnnLaunch &a);


#define TM_KNN_CASE(HT, HQ)                                                                                              \
  case HQ:                                                                                                               \
    hipLaunchKernelGGL((k_knn_mfma<HT, HQ>), grid, block, 0, a.stream, a.tpack, a.n_ttiles, a.bx, a.qpack, a.n_qtiles,   \
                       a.queries, a.qperm, a.qkey, a.nq, a.prune, a.best_key, a.best_tile, a.visited);                   \
    break;

#define TM_KNN_DEFINE_HT(HT)                                                              \
  template <> void knn_launch_ht<HT>(int hq, const KnnLaunch &a) {                        \
    const int64_t wg_tiles = (a.n_qtiles + KNN_NQ * KNN_NW - 1) / (KNN_NQ * KNN_NW);      \
    const dim3 grid((unsigned)wg_tiles), block(KNN_NW * 64);                              \
    switch (hq) {                                                                         \
      TM_KNN_CASE(HT, 0) TM_KNN_CASE(HT, 1) TM_KNN_CASE(HT, 2) TM_KNN_CASE(HT, 3)          \
      TM_KNN_CASE(HT, 4) TM_KNN_CASE(HT, 5)                                               \
      default:                                                                            \
        hipLaunchKernelGGL((k_knn_mfma<HT, 6>), grid, block, 0, a.stream, a.tpack, a.n_ttiles, a.bx, a.qpack, a.n_qtiles, \
                           a.queries, a.qperm, a.qkey, a.nq, a.prune, a.best_key, a.best_tile, a.visited);                 \
    }                                                                                     \
  }

}  // namespace tmx
