// tm_knn_kernel.h -- the int8 MFMA distance GEMM of the KNN stage (see tm_knn.hip for the scheme).  Included by the
// tm_knn_k<HT>.hip translation units, each instantiating one database high-chunk count HT with every query
// high-chunk count HQ (the per-side digit plan makes both data dependent).
#pragma once
#include <climits>
#include <type_traits>

#include "tm_common.h"

namespace tmx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// Shape of a workgroup (build-time; measured on the 720p x 300 clip): NQ=2, NW=2 at two waves per SIMD is the
// fastest; NQ=1, NW=4 (same 128 queries per staged tile, 168 registers, three waves per SIMD) is 11 % slower pruned and 6 % dense.
#ifndef TM_KNN_NQ
#define TM_KNN_NQ 2
#endif
#ifndef TM_KNN_NW
#define TM_KNN_NW 2
#endif
#ifndef TM_KNN_OCC
#define TM_KNN_OCC 2
#endif
#ifndef TM_KNN_NBUF
#define TM_KNN_NBUF 3
#endif
#ifndef TM_KNN_STAMPS
#define TM_KNN_STAMPS 0  // diagnostic build: s_memtime stamps around the phases of the scan loop, summed into visited[2..9]
#endif
#if TM_KNN_STAMPS
#define TM_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define TM_STAMP(i) do { } while (0)
#endif
#ifndef TM_KNN_DIRECT
#define TM_KNN_DIRECT 0  // experiment: 1 = tiles are read from global memory (L2) as MFMA operands, no LDS ring (use with NW=1)
#endif
constexpr int KNN_NQ = TM_KNN_NQ;  // query sub-tiles (32 queries each) per wave
constexpr int KNN_NW = TM_KNN_NW;  // waves per workgroup (NQ * NW * 32 = 128 queries share every staged tile)

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
constexpr int KNN_NC = 6;        // bounding-box columns
// One more box dimension, radial: R = |v - c| over all the OTHER columns.  |R(q) - R(t)|^2 <= the squared distance over those
// columns (reverse triangle inequality), so its gap adds to the lower bound like a column's; it tells a noisy tile from a smooth
// one of the same mean colour, which the widest (low-frequency) columns cannot.  Stored as integers rounded outwards.
constexpr int KNN_ND = KNN_NC + 1;
constexpr int KNN_GROUP = 128;   // tiles per second-level box = threads of a workgroup: one pass of the list build's loop
#ifndef TM_KNN_K0
#define TM_KNN_K0 8
#endif
constexpr int KNN_K0 = TM_KNN_K0;        // tiles visited first, around the workgroup's position on the curve
#ifndef TM_KNN_CHUNK
#define TM_KNN_CHUNK 1024
#endif
constexpr int KNN_CHUNK = TM_KNN_CHUNK;  // tiles tested per compaction round (bests are re-read for each round)

struct KnnBoxes {
  const int *lo, *hi;   // [KNN_ND][n_ttiles] bounding boxes of the database tiles
  const int *glo, *ghi; // [KNN_ND][ceil(n_ttiles / KNN_GROUP)] boxes of runs of KNN_GROUP tiles: a run no sub-tile can use is skipped whole
  const uint32_t *tkey; // [n_ttiles] curve key of each tile's first row (ascending)
  int col[KNN_NC];      // source feature column of each box dimension
  int cen[KNN_NC];      // the digit plan's centre of that column (the radial dimension is measured from the centres)
};

// Conservative int32 form of the box bound: sum over the box columns of (gap >> 1)^2 <= (best + 1) >> 2 is implied by
// sum gap^2 <= best + 1, so failing it proves the tile cannot matter (gaps < 2^15, six terms: no overflow).
__device__ __forceinline__ bool knn_box_may_matter(const int *tlo, const int *thi, const int *qlo, const int *qhi, int smax) {
  int lb = 0;
#pragma unroll
  for (int d = 0; d < KNN_ND; d++) {
    const int g = max(0, max(tlo[d] - qhi[d], qlo[d] - thi[d])) >> 1;
    lb += g * g;
  }
  return lb <= (int)(((unsigned)smax + 1u) >> 2);
}

// max over the 64 lanes of a wave: DPP inside rows of 16, then one lane of each row through SGPRs
__device__ __forceinline__ int knn_wave_max(int x) {
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false));  // row_half_mirror
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false));  // row_mirror
  return max(max(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)),
             max(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
}

// s_waitcnt vmcnt(behind * NST) for a run-time `behind` in [0, MAXB]: this wave's pieces of the oldest tile in flight have landed
template <int NST, int MAXB>
__device__ __forceinline__ void knn_wait_vm(int behind) {
  if constexpr (MAXB == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if (behind >= MAXB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXB * NST) : "memory");
    else knn_wait_vm<NST, MAXB - 1>(behind);
  }
}

// Compacts into s_list / s_mask the tiles of [chunk_base, chunk_base + KNN_CHUNK) that at least one query sub-tile of the
// workgroup can still use (bit = wave * NQ + sub-tile), judged with that sub-tile's largest running best (d'' = SSD -
// parity, so SSD <= d'' + 1).
__device__ __forceinline__ int knn_build_list(const int *__restrict__ box_lo, const int *__restrict__ box_hi, const int *__restrict__ grp_lo,
                                              const int *__restrict__ grp_hi, int64_t n_ttiles, int chunk_base, int r0a, int r0b, int split, int split_idx,
                                              int prune, const int *s_box_lo, const int *s_box_hi, const int *s_smax, uint16_t *s_list, uint8_t *s_mask,
                                              int *s_cnt /* [0] list length, [1] group mask */) {
  constexpr int NS = KNN_NW * KNN_NQ, ND = KNN_ND, NT = KNN_NW * 64;
  constexpr bool GROUPS = NT == KNN_GROUP && KNN_CHUNK % KNN_GROUP == 0;  // pass i of the loop below = group chunk_base / KNN_GROUP + i
  const int tid = threadIdx.x;
  __syncthreads();
  if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
  __syncthreads();
  unsigned gmask = ~0u;
  if (GROUPS && prune) {  // second level: one thread per group of the chunk tests the group's box against the sub-tiles
    const int64_t n_groups = (n_ttiles + KNN_GROUP - 1) / KNN_GROUP;
    const int64_t g = chunk_base / KNN_GROUP + tid;
    if (tid < KNN_CHUNK / KNN_GROUP && g < n_groups) {
      int tlo[ND], thi[ND];
#pragma unroll
      for (int d = 0; d < ND; d++) { tlo[d] = grp_lo[(int64_t)d * n_groups + g]; thi[d] = grp_hi[(int64_t)d * n_groups + g]; }
      bool any = false;
      for (int q = 0; q < NS; q++) any |= knn_box_may_matter(tlo, thi, s_box_lo + q * ND, s_box_hi + q * ND, s_smax[q]);
      if (any) atomicOr(&s_cnt[1], 1 << tid);
    }
    __syncthreads();
    gmask = (unsigned)s_cnt[1];
  }
  int pass = 0;
  for (int k = tid; k < KNN_CHUNK && chunk_base + k < n_ttiles; k += NT, pass++) {
    if (GROUPS && !((gmask >> pass) & 1u)) continue;
    const int t = chunk_base + k;
    if (t >= r0a && t < r0b) continue;  // done in round 0
    if (split > 1 && t % split != split_idx) continue;  // another workgroup of this query group scans that tile
    unsigned mask = prune ? 0u : 0xffu;
    if (prune) {
      int tlo[ND], thi[ND];
#pragma unroll
      for (int d = 0; d < ND; d++) { tlo[d] = box_lo[(int64_t)d * n_ttiles + t]; thi[d] = box_hi[(int64_t)d * n_ttiles + t]; }
      for (int q = 0; q < NS; q++)
        if (knn_box_may_matter(tlo, thi, s_box_lo + q * ND, s_box_hi + q * ND, s_smax[q])) mask |= 1u << q;
    }
    if (mask) {
      const int slot = atomicAdd(s_cnt, 1);
      s_list[slot] = (uint16_t)k;
      s_mask[slot] = (uint8_t)mask;
    }
  }
  __syncthreads();
  return *s_cnt;
}

// Out-of-line copy for the nearest-neighbour kernel, whose loop is faster that way (20.4 -> 18.3 ms on the bench clip) although the
// call costs it two spilled operand quads; the collection kernel inlines the build: with its ladder counters live, the calling
// convention (values across a call must sit in the callee-saved half of the file) spilled 100-190 bytes, and every reload's
// s_waitcnt vmcnt(0) also waited for the tiles in flight (first collection pass of the reference-defaults run: 230 -> 160 ms).
__device__ __attribute__((noinline)) int knn_build_list_call(const int *__restrict__ box_lo, const int *__restrict__ box_hi, const int *__restrict__ grp_lo,
                                                             const int *__restrict__ grp_hi, int64_t n_ttiles, int chunk_base, int r0a, int r0b, int prune,
                                                             const int *s_box_lo, const int *s_box_hi, const int *s_smax, uint16_t *s_list, uint8_t *s_mask,
                                                             int *s_cnt) {
  return knn_build_list(box_lo, box_hi, grp_lo, grp_hi, n_ttiles, chunk_base, r0a, r0b, 1, 0, prune, s_box_lo, s_box_hi, s_smax, s_list, s_mask, s_cnt);
}

template <int HT, int HQ, bool TOPK>
__global__ __launch_bounds__(KNN_NW * 64, TM_KNN_OCC) void k_knn_mfma(const uint8_t *__restrict__ tpack, int64_t n_ttiles, KnnBoxes bx,
                                                          const uint8_t *__restrict__ qpack, int64_t n_qtiles,
                                                          const int16_t *__restrict__ queries, const uint32_t *__restrict__ qperm,
                                                          const uint32_t *__restrict__ qkey, int64_t nq, int prune,
                                                          int *__restrict__ best_key, int *__restrict__ best_tile,
                                                          unsigned long long *__restrict__ visited, int *__restrict__ tau,
                                                          uint2 *__restrict__ cand, int *__restrict__ cand_cnt, int cand_cap, int cand_k,
                                                          int tshift /* 1; 0 when the database digits are those of 2 (t - c) */) {
  // TOPK: collection mode for the k-nearest search (ann_kdtree_short_search_multi, tilingencoder.pas:1563): every
  // query has a fixed threshold (an upper bound of its k-th smallest SSD); pruning uses it instead of a running best, and
  // every row with d'' <= tau is appended to the query's candidate list (d'', sorted row).  The threshold also walks down a
  // ladder tau0 * (8 - j) / 8 while the scan runs: once cand_k rows with d'' <= a rung have been seen, the k-th smallest SSD
  // is at most that rung + 1 (parity), which becomes the threshold for the rest of the scan.
  constexpr int NQ = KNN_NQ, NW = KNN_NW, ND = KNN_ND;
  constexpr int KT = 6 + HT, KQ = 6 + HQ, HM = HT < HQ ? HT : HQ;
  constexpr int T_BYTES = KT * 1024 + 128 + 64, Q_BYTES = KQ * 1024 + 128;  // database tiles carry their box (2*ND ints) too
  constexpr int TILE_VEC = T_BYTES / 16;
  constexpr int NT = NW * 64;
  constexpr int NST = (TILE_VEC + NT - 1) / NT;       // LDS-DMA pieces (64 lanes x 16 B) each wave issues per tile
  constexpr int BUF_BYTES = NST * NW * 1024;          // >= T_BYTES: the tail is padding that clamped lanes land in
  constexpr int NBUF = TM_KNN_NBUF;                   // ring: tile i is read while tiles i+1 .. i+NBUF-1 are in flight
  __shared__ __attribute__((aligned(16))) uint8_t lds[TM_KNN_DIRECT ? 1 : NBUF][TM_KNN_DIRECT ? 16 : BUF_BYTES];
  __shared__ int s_smax[NW][NQ];     // largest running best of each query sub-tile
  __shared__ int s_box[2][NW][NQ][ND];  // [lo|hi][wave][sub-tile][dim] query boxes
  __shared__ int s_ctl[4];
  __shared__ uint16_t s_list[KNN_CHUNK];
  __shared__ uint8_t s_mask[KNN_CHUNK];  // which sub-tiles wanted the listed tile (bit = wave * NQ + sub-tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
#if TM_KNN_STAMPS
  const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = st_begin, st_build = 0, st_listed = 0, st_idle = 0;
#endif
  constexpr int QT_PER_WG = NW * NQ;
  // Workgroup -> query tile: identity.  XCD-contiguous ranges were measured (profiles/README.md): whole ranges per XCD
  // lose 25 % to load imbalance (window sizes vary along the curve), runs of 32 per XCD tie with identity.
  const int64_t wgt = blockIdx.x;
  v4i bq[NQ][KQ];
  int nq2[NQ], best[NQ], bestt[NQ], tie[NQ];  // bestt = tile << 5 | row of the first minimum
  // query tile of sub-tile s: a function of uniform values, recomputed where it is needed instead of living in four registers
  auto qtile_of = [&](int s) -> int64_t { return wgt * QT_PER_WG + s * NW + __builtin_amdgcn_readfirstlane(wave); };
  // The 20 operand quads of the queries.  The nearest-neighbour kernel reads them AGAIN after every out-of-line list build (six
  // times per workgroup, from L2): a value that is rewritten after a call is dead across it, which takes 80 registers out of what
  // must survive the call in the callee-saved half of the file -- without that the allocator sits on the edge of spilling two quads,
  // and a reload's s_waitcnt vmcnt(0) in the MFMA chain also waits for the tiles in flight (18.3 ms with no spill, 19.9 with two).
  auto load_bq = [&]() {
#pragma unroll
    for (int s = 0; s < NQ; s++) {
      const int64_t qt = qtile_of(s) < n_qtiles ? qtile_of(s) : n_qtiles - 1;
      const uint8_t *qb = qpack + qt * (int64_t)Q_BYTES;
      asm volatile("" : "+v"(qb));  // a pointer the compiler cannot match with the earlier loads
#pragma unroll
      for (int kc = 0; kc < KQ; kc++) bq[s][kc] = *reinterpret_cast<const v4i *>(qb + (kc * 64 + lane) * 16);
    }
  };
  load_bq();
#pragma unroll
  for (int s = 0; s < NQ; s++) {
    // interleaved: curve neighbours (which want the same tiles) sit in different waves
    const int64_t qt = qtile_of(s) < n_qtiles ? qtile_of(s) : n_qtiles - 1;
    const uint8_t *qb = qpack + qt * (int64_t)Q_BYTES;
    nq2[s] = reinterpret_cast<const int *>(qb + KQ * 1024)[lane & 31] & ~1;  // 2*(|q-c|^2 >> 1); the parity bit returns in the refine stage
    best[s] = TOPK ? tau[qt * 32 + (lane & 31)] : INT_MAX;
    bestt[s] = INT_MAX;
    tie[s] = 0;
  }
  // bounding boxes: per query sub-tile (lanes 0..31 / 32..63 hold sub-tile 0 / 1) and for the whole workgroup
  {
    static_assert(NQ == 1 || NQ == 2, "lane <-> query mapping: one sub-tile (both half-waves hold it) or two per wave");
    const int64_t p = min((wgt * QT_PER_WG + (NQ == 2 ? half : 0) * NW + wave) * 32 + (lane & 31), nq - 1);
    const int16_t *row = queries + (int64_t)qperm[p] * 192;
    long long boxsq = 0;  // squared distance from the centre over the box columns
#pragma unroll
    for (int d = 0; d < ND; d++) {
      int lo, hi;
      if (d < KNN_NC) {
        lo = hi = row[bx.col[d]];
        const long long c = lo - bx.cen[d];
        boxsq += c * c;
      } else {
        // radial: |q-c|^2 over the other columns = the packed norm (known up to its dropped parity bit) minus the box columns' part
        const long long n2 = (long long)(unsigned)(NQ == 2 && half ? nq2[NQ - 1] : nq2[0]);
        lo = max(0, (int)floor(sqrt((double)max(0ll, n2 - boxsq))) - 1);
        hi = (int)ceil(sqrt((double)max(0ll, n2 + 1 - boxsq))) + 1;
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
      if ((lane & 31) == 0 && (NQ == 2 || half == 0)) { s_box[0][wave][NQ == 2 ? half : 0][d] = lo; s_box[1][wave][NQ == 2 ? half : 0][d] = hi; }
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
  if (tid < NW * NQ) (&s_smax[0][0])[tid] = INT_MAX;
  __syncthreads();
  if (TOPK) {  // fixed sub-tile maxima
#pragma unroll
    for (int s = 0; s < NQ; s++) {
      const int smax = knn_wave_max(best[s]);
      if (lane == 0) s_smax[wave][s] = smax;
    }
    __syncthreads();
  }
  // box re-test, lane-parallel: lane 8 * s + d holds dimension d of sub-tile s of this wave (other lanes hold an empty box: gap 0)
  const int bx_s = lane >> 3, bx_d = lane & 7;
  const bool bx_on = bx_s < NQ && bx_d < ND;
  const int bx_qlo = bx_on ? s_box[0][wave][bx_on ? bx_s : 0][bx_on ? bx_d : 0] : INT_MIN / 2;
  const int bx_qhi = bx_on ? s_box[1][wave][bx_on ? bx_s : 0][bx_on ? bx_d : 0] : INT_MAX / 2;
  // (readfirstlane: the loop state below is uniform; saying so moves its arithmetic and branches to the scalar unit)
  // gridDim.y > 1 (collection passes over few queries): the workgroups of one query group share the tile list, tile t goes to
  // workgroup t mod gridDim.y; the seed round is dropped (its tiles come through the lists like all others)
  const int split = TOPK ? (int)gridDim.y : 1, split_idx = TOPK ? (int)blockIdx.y : 0;  // constants in the nearest-neighbour kernel
  const int r0a = __builtin_amdgcn_readfirstlane(s_ctl[0]), r0b = split > 1 ? r0a : (int)min((int64_t)r0a + KNN_K0, n_ttiles);  // round-0 tiles [r0a, r0b)
  long long nvisit = 0, nstaged = 0;

  // candidate iterator: round 0 = [r0a, r0b); then chunks of the tile list, compacted into s_list by all threads
  // Chunks are visited outwards from the one the workgroup sits in (home, home + 1, home - 1, ...): near tiles tighten the bests first.
  const int n_chunks = (int)((n_ttiles + KNN_CHUNK - 1) / KNN_CHUNK), home_chunk = r0a / KNN_CHUNK;
  int phase = 0, r0next = r0a, chunk_base = 0, chunk_j = 0, chunks_done = 0, list_n = 0, list_i = 0;
  uint16_t pre_k = 0;  // entry list_i, read one call early and only combined when it is used
  uint8_t pre_m = 0;
  // refill(): called by every thread at the same point (contains barriers) -- makes sure the next pop has an entry unless the stream
  // has ended; pop(): no calls, no barriers (it runs beside the MFMAs)
  auto refill = [&]() {
    if (phase == 0) {
      if (r0next < r0b) return;
      phase = 1;
      list_n = list_i = 0;
    }
    while (list_i >= list_n && chunks_done < n_chunks) {
      int c;
      do {  // j = 0, 1, 2, 3, ... -> home, home + 1, home - 1, home + 2, ...; out-of-range ones are skipped
        c = home_chunk + ((chunk_j & 1) ? (chunk_j + 1) >> 1 : -(chunk_j >> 1));
        chunk_j++;
      } while (c < 0 || c >= n_chunks);
      chunks_done++;
      chunk_base = c * KNN_CHUNK;
#if TM_KNN_STAMPS
      const unsigned long long tb_ = __builtin_amdgcn_s_memtime();
#endif
      if constexpr (TOPK)
        list_n = __builtin_amdgcn_readfirstlane(knn_build_list(bx.lo, bx.hi, bx.glo, bx.ghi, n_ttiles, chunk_base, r0a, r0b, split, split_idx, prune, &s_box[0][0][0][0],
                                                               &s_box[1][0][0][0], &s_smax[0][0], s_list, s_mask, &s_ctl[1]));
      else
      {
        list_n = __builtin_amdgcn_readfirstlane(knn_build_list_call(bx.lo, bx.hi, bx.glo, bx.ghi, n_ttiles, chunk_base, r0a, r0b, prune, &s_box[0][0][0][0],
                                                                    &s_box[1][0][0][0], &s_smax[0][0], s_list, s_mask, &s_ctl[1]));
        load_bq();
      }
#if TM_KNN_STAMPS
      st_build += __builtin_amdgcn_s_memtime() - tb_;
#endif
      list_i = 0;
      pre_k = s_list[0];
      pre_m = s_mask[0];
    }
  };
  auto pop = [&]() -> int {
    if (phase == 0) return (r0next++) | (0xff << 23);
    if (list_i >= list_n) return -1;
    const int r = chunk_base + __builtin_amdgcn_readfirstlane((int)pre_k | ((int)pre_m << 23));
    list_i++;
    const int k = min(list_i, KNN_CHUNK - 1);  // the entry of the next call: its LDS latency hides behind this tile's work
    pre_k = s_list[k];
    pre_m = s_mask[k];
    return r;
  };

  // async staging: every wave copies NST pieces of a tile straight into LDS (global_load_lds, no registers); lanes
  // past the end of the tile re-read its last vector into the buffer padding so all waves issue the same count
  auto issue = [&](int tile, int buf) {
    const uint8_t *src = tpack + (tile & 0x7fffff) * (int64_t)T_BYTES;
#pragma unroll
    for (int i = 0; i < (TM_KNN_DIRECT ? 0 : NST); i++) {
      const int piece = wave + i * NW;
      const int v = min(piece * 64 + lane, TILE_VEC - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + v * 16),
                                       (__attribute__((address_space(3))) void *)(&lds[buf][piece * 1024]), 16, 0, 0);
    }
  };
  // ring of NBUF LDS buffers: tile q[0] is read while tiles q[1..NBUF-2] are in flight
  int q[NBUF - 1];
  bool more = true;  // the candidate stream has not ended
#pragma unroll
  for (int i = 0; i < NBUF - 1; i++) {
    if (more) refill();
    q[i] = more ? pop() : -1;
    if (q[i] >= 0) issue(q[i], i); else more = false;
  }
  int cur_tile = q[0];

  int cur = 0;
  int smax_reg[NQ];
#pragma unroll
  for (int s = 0; s < NQ; s++) smax_reg[s] = TOPK ? s_smax[wave][s] : INT_MAX;
  int lad_step[NQ], lad_cnt[NQ][7];  // collection mode: rung spacing (tau0 / 8) and rows seen at or below rung j + 1
#pragma unroll
  for (int s = 0; s < NQ; s++) {
    lad_step[s] = TOPK ? best[s] >> 3 : 0;
#pragma unroll
    for (int j = 0; j < 7; j++) lad_cnt[s][j] = 0;
  }
  bool improved = false;
  TM_STAMP(0);
  while (cur_tile >= 0) {
    nstaged++;
    // tile `cur_tile` landed?  Only this wave's own pieces are counted; the barrier publishes everyone's.  It also
    // fences the previous iteration's LDS reads (buffer reuse) and s_smax writes.
    if (!TM_KNN_DIRECT) {
      int behind = 0;  // tiles in flight behind the current one
#pragma unroll
      for (int i = 1; i < NBUF - 1; i++) behind += q[i] >= 0 ? 1 : 0;
      knn_wait_vm<NST, NBUF - 2>(behind);
    }
    TM_STAMP(1);
    if (!TM_KNN_DIRECT || NW > 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    TM_STAMP(2);
    if (more) refill();
    int nn_tile = -1;  // NBUF - 1 ahead, chosen with the bests as they are now
    // Picking and staging the tile NBUF - 1 ahead is pure bookkeeping (list pop, address arithmetic, LDS-DMA issue).  It runs after this
    // wave's first MFMA chain has been issued, so that it executes while the matrix pipe works; buffer (cur + NBUF - 1) % NBUF was read
    // last in the previous iteration.
    auto stage_ahead = [&]() {
      if (more) { nn_tile = pop(); more = nn_tile >= 0; }
      TM_STAMP(7);
      if (nn_tile >= 0) issue(nn_tile, cur >= 1 ? cur - 1 : NBUF - 1);
      TM_STAMP(3);
    };
    const uint8_t *L = TM_KNN_DIRECT ? tpack + (cur_tile & 0x7fffff) * (int64_t)T_BYTES : lds[cur];
    // sub-tile level skip: the list round already judged every (tile, sub-tile) pair with the bests of that time;
    // pairs it kept are re-judged against the current best with the tile's box (it rides in LDS behind the norms)
    bool do_sub[NQ];
    [[maybe_unused]] bool do_tile = false;  // read by the TM_KNN_STAMPS build only
    {
      unsigned long long ok = ~0ull;
      if (prune) {
        const int *tb = reinterpret_cast<const int *>(L + KT * 1024 + 128);
        const int tlo = tb[bx_on ? bx_d : 0], thi = tb[ND + (bx_on ? bx_d : 0)];
        const int g = max(0, max(tlo - bx_qhi, bx_qlo - thi)) >> 1;
        int lb = g * g;
        lb += __builtin_amdgcn_update_dpp(0, lb, 0xB1, 0xf, 0xf, false);   // the 8 lanes of a sub-tile: quad, quad, half row
        lb += __builtin_amdgcn_update_dpp(0, lb, 0x4E, 0xf, 0xf, false);
        lb += __builtin_amdgcn_update_dpp(0, lb, 0x141, 0xf, 0xf, false);
        int thr = smax_reg[0];
#pragma unroll
        for (int s = 1; s < NQ; s++) thr = bx_s == s ? smax_reg[s] : thr;
        ok = __builtin_amdgcn_ballot_w64(lb <= (int)(((unsigned)thr + 1u) >> 2));
      }
#pragma unroll
      for (int s = 0; s < NQ; s++) {
        do_sub[s] = ((cur_tile >> (23 + wave * NQ + s)) & 1) != 0 && ((ok >> (8 * s)) & 1) != 0;
        do_tile |= do_sub[s];
      }
    }
    TM_STAMP(4);
#if TM_KNN_STAMPS
    {
      int listed = 0;
      for (int s = 0; s < NQ; s++) listed += (cur_tile >> (23 + wave * NQ + s)) & 1;
      st_listed += listed;
      st_idle += do_tile ? 0 : 1;
    }
#endif
    // one query sub-tile at a time (the other wave of the SIMD overlaps its MFMAs with this wave's VALU epilogue)
    v16i acc;
    auto run_mfma = [&](auto S) {
      constexpr int s = decltype(S)::value;
      nvisit++;
      // One accumulator, three phases: the products of the high digits come first and are shifted up by one digit before the mixed
      // products are added onto them, and again before the low ones (acc = ((T_H.Q_H << 8) + T_L.Q_H + T_H.Q_L) << 8) + T_L.Q_L, exact
      // mod 2^32).  Chunks are read from LDS once per phase that uses them; 16 accumulator registers instead of 48.
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = 0;
      if (HM > 0) {
#pragma unroll
        for (int kc = 0; kc < HM; kc++) {
          const v4i a = *reinterpret_cast<const v4i *>(L + ((6 + kc) * 64 + lane) * 16);  // T_H chunk
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][6 + kc], acc, 0, 0, 0);      // T_H . Q_H
        }
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = (int)((unsigned)acc[r] << 8);
      }
      if (HT + HQ > 0) {
#pragma unroll
        for (int kc = 0; kc < HQ; kc++) {
          const v4i a = *reinterpret_cast<const v4i *>(L + (kc * 64 + lane) * 16);          // T_L chunk
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][6 + kc], acc, 0, 0, 0);      // T_L . Q_H
        }
#pragma unroll
        for (int kc = 0; kc < HT; kc++) {
          const v4i a = *reinterpret_cast<const v4i *>(L + ((6 + kc) * 64 + lane) * 16);  // T_H chunk
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][kc], acc, 0, 0, 0);          // T_H . Q_L
        }
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = (int)((unsigned)acc[r] << 8);
      }
#pragma unroll
      for (int kc = 0; kc < 6; kc++) {
        const v4i a = *reinterpret_cast<const v4i *>(L + (kc * 64 + lane) * 16);            // T_L chunk
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][kc], acc, 0, 0, 0);            // T_L . Q_L
      }
    };
    auto run_epilogue = [&](auto S) {
      constexpr int s = decltype(S)::value;
      // accumulator row of register r: (r&3) + 8*(r>>2) + 4*half  -> norms as four 16-byte reads (here, not before the MFMAs: the
      // registers are free while the chain and the staging run)
      int nt[16];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const v4i x = *reinterpret_cast<const v4i *>(L + KT * 1024 + (g * 8 + half * 4) * 4);
        nt[g * 4] = x[0]; nt[g * 4 + 1] = x[1]; nt[g * 4 + 2] = x[2]; nt[g * 4 + 3] = x[3];
      }
      // d'' = 2 acc + |t-c|^2 + 2 (|q-c|^2 >> 1): the query's own term is the same for the 16 rows of a lane, so the minimum is taken
      // without it (every d'' is below 2^31 and so is the query term: no wrap between the two orders), and the rows' values are only
      // formed on the rare paths that need them
      int t[16];
      int tm = INT_MAX;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        t[r] = (int)(((unsigned)acc[r] << tshift) + (unsigned)nt[r]);
        tm = min(tm, t[r]);
      }
      const int m = (int)((unsigned)tm + (unsigned)nq2[s]);
#define TM_KNN_D(r) ((int)((unsigned)t[r] + (unsigned)nq2[s]))
      if (TOPK) {  // collection mode: every row within the query's threshold is a candidate
        const int64_t q = qtile_of(s) * 32 + (lane & 31);
        const bool hit = m <= best[s] && qtile_of(s) < n_qtiles && q < nq;
        if (hit) {
          // rung j (1..7) = (8 - j) * step <= tau0 - j * tau0 / 8; the last database tile pads with copies of its last row,
          // which must not be counted
          const bool countable = (cur_tile & 0x7fffff) != (int)n_ttiles - 1;
#pragma unroll
          for (int r = 0; r < 16; r++)
            if (TM_KNN_D(r) <= best[s]) {
              const int slot = atomicAdd(&cand_cnt[q], 1);
              if (slot < cand_cap)
                cand[q * cand_cap + slot] = make_uint2((unsigned)TM_KNN_D(r), (unsigned)(((cur_tile & 0x7fffff) << 5) | ((r & 3) + 8 * (r >> 2) + 4 * half)));
#pragma unroll
              for (int j = 1; j <= 7; j++) lad_cnt[s][j - 1] += (countable && TM_KNN_D(r) <= (8 - j) * lad_step[s]) ? 1 : 0;
            }
        }
        if (__builtin_amdgcn_ballot_w64(hit)) {  // both lanes of a query take part: the counts of its two row halves add up
          int rung = 0;
#pragma unroll
          for (int j = 1; j <= 7; j++) {
            const int c = lad_cnt[s][j - 1] + __shfl_xor(lad_cnt[s][j - 1], 32);
            if (c >= cand_k) rung = j;
          }
          // cand_k rows have d'' <= rung, i.e. SSD <= rung + 1: no row beyond that can be among the k nearest
          const int t = (8 - rung) * lad_step[s] + 1;
          if (rung > 0 && lad_step[s] > 0 && t < best[s]) { best[s] = t; improved = true; }
        }
        return;
      }
      if (m == best[s]) tie[s] = 1;  // another tile reaches the same value
      if (m < best[s]) {             // this lane improves: which row, and is it alone?
        int row = 0, cnt = 0;
#pragma unroll
        for (int r = 15; r >= 0; r--)
          if (t[r] == tm) { row = (r & 3) + 8 * (r >> 2) + 4 * half; cnt++; }
        best[s] = m;
        bestt[s] = ((cur_tile & 0x7fffff) << 5) | row;
        improved = true;
        tie[s] = cnt > 1;
      }
#undef TM_KNN_D
    };
    if (do_sub[0]) run_mfma(std::integral_constant<int, 0>{});
    stage_ahead();
    if (do_sub[0]) run_epilogue(std::integral_constant<int, 0>{});
    if constexpr (NQ == 2) {
      if (do_sub[1]) { run_mfma(std::integral_constant<int, 1>{}); run_epilogue(std::integral_constant<int, 1>{}); }
    }
    TM_STAMP(5);
    if (__builtin_amdgcn_ballot_w64(improved)) {  // some lane has a new best: refresh the sub-tile maxima
      improved = false;
#pragma unroll
      for (int s = 0; s < NQ; s++) {
        const int smax = knn_wave_max(best[s]);
        smax_reg[s] = smax;
        if (lane == 0) s_smax[wave][s] = smax;
      }
    }
    TM_STAMP(6);
    cur = cur == NBUF - 1 ? 0 : cur + 1;
#pragma unroll
    for (int i = 0; i < NBUF - 2; i++) q[i] = q[i + 1];
    q[NBUF - 2] = nn_tile;
    cur_tile = q[0];
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
    if (TOPK && lane < 32 && qtile_of(s) < n_qtiles) atomicMin(&tau[qtile_of(s) * 32 + lane], best[s]);  // the final threshold: k_topk_select drops what lies above it
    if (!TOPK && lane < 32 && qtile_of(s) < n_qtiles) {
      const int64_t q = qtile_of(s) * 32 + lane;
      best_key[q] = best[s];
      best_tile[q] = (bestt[s] & 0x3fffffff) | (tie[s] ? (1 << 30) : 0);  // sorted row of the first minimum; bit 30: tie flag
    }
  }
  if (visited && lane == 0) atomicAdd(visited, (unsigned long long)nvisit);
  if (visited && tid == 0) atomicAdd(visited + 1, (unsigned long long)nstaged);
#if TM_KNN_STAMPS
  if (visited && lane == 0) {
    for (int i = 0; i < 7; i++) atomicAdd(visited + 2 + i, st_acc[i]);
    atomicAdd(visited + 9, __builtin_amdgcn_s_memtime() - st_begin);
    atomicAdd(visited + 10, st_acc[7]);
    atomicAdd(visited + 11, st_build);
    atomicAdd(visited + 12, st_listed);
    atomicAdd(visited + 13, st_idle);
  }
#endif
}

struct KnnLaunch {
  const uint8_t *tpack; int64_t n_ttiles; KnnBoxes bx;
  const uint8_t *qpack; int64_t n_qtiles; const int16_t *queries; const uint32_t *qperm, *qkey; int64_t nq; int prune;
  int *best_key, *best_tile; unsigned long long *visited; hipStream_t stream;
  int *tau = nullptr; uint2 *cand = nullptr; int *cand_cnt = nullptr; int cand_cap = 0, cand_k = 0;  // collection mode (k nearest)
  int split = 1;  // collection mode: workgroups per query group (they share its tile list)
  int tshift = 1; // 0: the database pack holds the digits of 2 (t - c) (KnnPlan::tscale = 2)
};

// one per HT, defined in tm_knn_k<HT>.hip
template <int HT> void knn_launch_ht(int hq, const KnnLaunch &a);


#define TM_KNN_LAUNCH(HT, HQ, TOPK)                                                                                        \
  hipLaunchKernelGGL((k_knn_mfma<HT, HQ, TOPK>), grid, block, 0, a.stream, a.tpack, a.n_ttiles, a.bx, a.qpack, a.n_qtiles,     \
                     a.queries, a.qperm, a.qkey, a.nq, a.prune, a.best_key, a.best_tile, a.visited, a.tau, a.cand, a.cand_cnt, \
                     a.cand_cap, a.cand_k, a.tshift)
#define TM_KNN_CASE(HT, HQ)                                                   \
  case HQ:                                                                    \
    if (a.tau) TM_KNN_LAUNCH(HT, HQ, true); else TM_KNN_LAUNCH(HT, HQ, false); \
    break;

#define TM_KNN_DEFINE_HT(HT)                                                              \
  template <> void knn_launch_ht<HT>(int hq, const KnnLaunch &a) {                        \
    const int64_t wg_tiles = (a.n_qtiles + KNN_NQ * KNN_NW - 1) / (KNN_NQ * KNN_NW);      \
    const dim3 grid((unsigned)wg_tiles, (unsigned)(a.tau ? a.split : 1)), block(KNN_NW * 64);  \
    switch (hq) {                                                                         \
      TM_KNN_CASE(HT, 0) TM_KNN_CASE(HT, 1) TM_KNN_CASE(HT, 2) TM_KNN_CASE(HT, 3)          \
      TM_KNN_CASE(HT, 4) TM_KNN_CASE(HT, 5)                                               \
      default:                                                                            \
        if (a.tau) TM_KNN_LAUNCH(HT, 6, true); else TM_KNN_LAUNCH(HT, 6, false);          \
    }                                                                                     \
  }

}  // namespace tmx
