// tm_knn3_kernel.h -- the nearest-neighbour scan of the KNN stage, third shape ("KNN"): three kernels instead of one.
//
// The second shape (tm_knn2_kernel.h, round 2-3) did everything in one persistent kernel: a workgroup loaded its 384 queries, scored the
// tiles around its position on the curve, built tile lists with all 1024 threads, consumed them, built the next ... and its phase stamps
// showed 21 % of the wave time in phases where the matrix pipe has nothing to do (list building 10.8 %, the wait for the slowest wave at
// every list's end 6.8 %, prologue 3.7 %).  Here each phase is a kernel of its own, shaped for what it does:
//   1. k_knn_seed    every query sub-tile against the K3_SEEDS database tiles around its group's position on the curve: bests (one 64-bit
//                    word per query), tie values, and the sub-tile's bound sqrt(largest best) -- MFMA work with no list in sight, several
//                    workgroups per CU.
//   2. k_knn_lists   the tile list of every query group, judged against the seeds' bounds: runs of KNN_GROUP tiles first, then the tiles of
//                    surviving runs; per entry the box lower bound of each (tile, sub-tile) pair as a 16-bit square root.  No MFMA, no query
//                    operands: small workgroups, many per CU, whose global loads hide behind each other.  Lists go to an arena in HBM in
//                    segments of <= K3_LCAP entries.
//   3. k_knn_consume persistent workgroups (one per CU, 16 waves): a group's query operands and its seeds' bests into LDS, then one list
//                    segment after the other: copied into LDS, consumed by the waves on their own exactly as in the second shape (pop an
//                    entry, re-judge it against the sub-tiles' bounds AS THEY ARE NOW, tile into registers, one MFMA chain per sub-tile
//                    that wants it).  An entry judged with the seeds' bound instead of a later, tighter one costs one pop, not a block.
// Arithmetic, operands, boxes, curve and the exactness argument are those of the earlier shapes: a pair is skipped only when its lower
// bound exceeds the sub-tile's largest best + 1 (both sides of the 16-bit compare rounded the safe way), the minimum VALUE is exact, a
// second row reaching it raises the tie flag that k_knn_ties settles by original index.
#pragma once
#include "tm_knn_kernel.h"

namespace tmx {

#ifndef TM_KNN3_WAVES
#define TM_KNN3_WAVES 16  // consume: 16 = one workgroup per CU; 8 = two (half the LDS each, fewer sub-tiles per group)
#endif
constexpr int K3_NW = TM_KNN3_WAVES;
constexpr int K3_NT = K3_NW * 64;
constexpr int K3_WGS = 16 / K3_NW;       // consume workgroups per CU (they split its LDS)
constexpr int K3_LDS = 163840 / K3_WGS;
constexpr int K3_LCAP = 1024;            // entries of a list segment (what the consume kernel holds in LDS at a time)
#ifndef TM_KNN3_SEEDS
#define TM_KNN3_SEEDS 8
#endif
constexpr int K3_SEEDS = TM_KNN3_SEEDS;  // tiles around the group's position on the curve, scored first by every sub-tile
#ifndef TM_KNN3_XCD_CONTIG
#define TM_KNN3_XCD_CONTIG 0             // 1: every XCD walks one contiguous eighth of the groups (measured: no change in L2 hits)
#endif
constexpr int K3_XCD_RUN = 32 * K3_WGS;  // groups dealt to one XCD at a time (neighbours on the curve)
#ifndef TM_KNN3_REFRESH_EVERY
#define TM_KNN3_REFRESH_EVERY 64         // consume: list entries between two full refreshes of the sub-tiles' bounds (a power of two; 0 = never)
#endif
#ifndef TM_KNN3_STAMPS
#define TM_KNN3_STAMPS 0                 // diagnostic build: s_memtime spans of the consume kernel's phases, summed over all waves
#endif
#if TM_KNN3_STAMPS
#define K3_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define K3_STAMP(i) do { } while (0)
#endif
constexpr int K3_LIST_NT = 256;          // threads of a list-building workgroup

constexpr int k3_lds_bytes(int ns, int kq) { return ns * (kq * 1024 + 576) + 64 + 64 + 64 + K3_LCAP * (4 + 2 * ((ns + 1) & ~1)); }
constexpr int k3_ns(int kq) {
  int ns = 16;
  while (ns > 1 && k3_lds_bytes(ns, kq) > K3_LDS) ns--;
  return ns;
}

enum { K3_MODE_LISTS = 0, K3_MODE_DENSE = 1 };

struct Knn3Args {
  const uint8_t *tpack; int64_t n_ttiles, nt_rows;
  const int *box_lo, *box_hi, *grp_lo, *grp_hi;  // database tile boxes [KNN_ND][n_ttiles], boxes of runs of KNN_GROUP tiles
  const uint8_t *qpack; int64_t n_qtiles, nq;
  const int *qmeta;             // [n_qtiles][16]: box lo[7], home tile, box hi[7], mask of the sub-tile's non-zero high-digit chunks
  const uint8_t *thmask;        // [n_ttiles] the same mask for every database tile
  int ns;                       // sub-tiles per group (k3_ns of the queries' digit plan)
  int mode;                     // consume: K3_MODE_LISTS / K3_MODE_DENSE (every tile, every sub-tile; no seeds, no lists)
  int tdouble;                  // the database pack holds the digits of 2 (t - c)
  // between the kernels (HBM)
  unsigned long long *gbest;    // [n_qtiles * 32] (d'' + 1) << 32 | sorted row
  unsigned *gtie;               // [n_qtiles * 32] smallest d'' + 1 seen twice
  unsigned *gsmax;              // [n_qtiles] upper bound of sqrt(largest best + 1) of the sub-tile, 0xFFFE = some query has none
  uint2 *segs;                  // [n_groups][max_segs] (first entry, entries) of each list segment
  int *nsegs;                   // [n_groups]
  int max_segs;
  unsigned *ltile;              // arena [arena_cap] tile << 8 | mask of the tile's non-zero high-digit chunks, per entry
  uint16_t *llb;                // arena [arena_cap][nsp] lower bounds per sub-tile slot
  unsigned long long arena_cap; // entries
  unsigned long long *arena_cursor;  // entries handed out (keeps counting past the capacity: the host sizes the next arena from it)
  int *best_key, *best_tile;
  // collection mode (the k nearest rows, ann_kdtree_short_search_multi, tilingencoder.pas:1563): see k_knn_consume<.., TOPK = true>
  int *tau;                     // [n_qtiles * 32] every sorted query's threshold (d'' <= tau); the scan leaves its final one
  const int *step;              // [n_qtiles * 32] the spacing of its ladder's rungs below the threshold (null: tau / 8)
  uint2 *cand;                  // [nq][cand_cap] (d'', sorted row) of every row within the threshold
  int *cand_cnt;                // [nq] rows appended (keeps counting past cand_cap)
  int cand_cap, cand_k;
  int split;                    // workgroups that share one group's tile list (entry j goes to part j mod split)
  int no_seeds;                 // lists: no seed kernel ran (collection mode), so no tile is left out of the lists
  unsigned long long *stats;    // consume: [0] blocks evaluated, [1] tiles read, [2] exact (query, row) pairs, [3] list entries consumed
  unsigned long long *seed_stats;  // [64][4] striped by workgroup: blocks, tiles read, pairs of the seed kernel
  int64_t n_groups;
  int grid_blocks;              // consume: persistent workgroups
  unsigned *tickets;            // [8] zeroed before the consume launch: next run-slot of each XCD's share of the groups
};

__device__ __forceinline__ unsigned k3_wave_umax(unsigned x) {  // max over lanes 0..31 (every lane of a row of 16 ends with its row's)
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true));
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true));
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, true));
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, true));
  return max((unsigned)__builtin_amdgcn_readlane((int)x, 0), (unsigned)__builtin_amdgcn_readlane((int)x, 16));
}

// a fresh look at an LDS word other waves update (relaxed workgroup-scope load: a plain ds_read_b32 the compiler may not cache)
__device__ __forceinline__ unsigned k3_peek(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// largest r with r * r <= x (x < 2^31): the hardware's approximate root (within one unit of the last place, i.e. well within one of
// the integer root here) set right by one step either way
__device__ __forceinline__ unsigned k3_isqrt(unsigned x) {
  unsigned r = (unsigned)__builtin_amdgcn_sqrtf((float)x);
  r += ((r + 1u) * (r + 1u) <= x) ? 1u : 0u;
  r -= (r * r > x) ? 1u : 0u;
  return r;
}

// the bound a sub-tile's tiles are judged against, from its largest best mx = d'' + 1 (an SSD bound): an upper bound of its square root
__device__ __forceinline__ unsigned k3_bound_of(unsigned mx) { return mx == ~0u ? 0xFFFEu : min(0xFFFEu, (unsigned)__builtin_amdgcn_sqrtf((float)mx) + 3u); }

// One block's chain: X over the digit products on one accumulator shifted between the phases; the rows' own term `cin` rides in on the
// second shift.  With TD (database digits of 2 (t - c)) cin = |t-c|^2 and the chain ends in 2 X + |t-c|^2 = d'' - qn; without, cin =
// |t-c|^2 >> 1 and it ends in Y = X + (|t-c|^2 >> 1): d'' - qn = 2 Y + (|t-c|^2 & 1), which the epilogue makes only for the blocks that
// can matter (the doubling of sixteen registers was a fifth of a block's vector instructions).  `q` = the sub-tile's B operands in LDS at
// this lane's 16 bytes (chunk stride 1024).
// `tm` / `qm` (wave-uniform): bit kc set = high-digit chunk kc of the tile / of the sub-tile holds a non-zero digit.  A product with an
// all-zero chunk adds nothing and is skipped, its LDS read with it: the columns are packed widest first (make_plan_scaled), so a tile of
// smooth content has its high digits in the first chunk or two only.  (Measured against straight-line chains for the common mask shapes,
// picked by one or two uniform tests: a predicate per product is as fast or faster on both bench clips -- the other waves of the SIMD
// fill the matrix pipe while a product waits for its LDS read -- skips more, and needs no spilled register.)
template <int HT, int HQ, bool TD>
__device__ __forceinline__ v16i k3_chain(const v4i (&T)[6 + HT], const v16i &cin, const uint8_t *q, unsigned tm, unsigned qm) {
  constexpr int HM = HT < HQ ? HT : HQ;
  const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  v16i acc = cin;  // a single phase: the rows' term is the chain's starting value
  if constexpr (HT + HQ > 0) {
    // No register is zeroed for a block: whatever the masks say, the chain's FIRST product is one that always runs, with the constant 0 as
    // its C operand.  That is T_H0 . Q_H0 where any T_H . Q_H product runs, and the first product of the middle phase otherwise (T_H0 . Q_L0;
    // T_L0 . Q_H0 for a database without high digits) -- a chunk its mask calls empty is all zeros (k3_load_tile_masked fills a skipped
    // chunk with them), so the product then adds nothing: a matrix instruction instead of sixteen moves.
    constexpr int FA = HT > 0 ? 6 : 0;  // the middle phase's first product: chunk FA of the tile, chunk FQ of the sub-tile
    constexpr int FQ = HT > 0 ? 0 : 6;
    bool hh = false;
    if constexpr (HM > 0) hh = (tm & qm) != 0;
    if (hh) {
      if constexpr (HM > 0) {
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6], *reinterpret_cast<const v4i *>(q + 6 * 1024), zero, 0, 0, 0);                // T_H . Q_H
#pragma unroll
        for (int kc = 1; kc < HM; kc++)
          if (((tm & qm) >> kc) & 1u)
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6 + kc], *reinterpret_cast<const v4i *>(q + (6 + kc) * 1024), acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = (int)((unsigned)acc[r] << 8);
      }
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[FA], *reinterpret_cast<const v4i *>(q + FQ * 1024), acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[FA], *reinterpret_cast<const v4i *>(q + FQ * 1024), zero, 0, 0, 0);
    }
#pragma unroll
    for (int kc = (HT > 0 ? 0 : 1); kc < HQ; kc++)
      if ((qm >> kc) & 1u)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[kc], *reinterpret_cast<const v4i *>(q + (6 + kc) * 1024), acc, 0, 0, 0);      // T_L . Q_H
#pragma unroll
    for (int kc = 1; kc < HT; kc++)
      if ((tm >> kc) & 1u)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6 + kc], *reinterpret_cast<const v4i *>(q + kc * 1024), acc, 0, 0, 0);        // T_H . Q_L
  }
  // the first two operands of the last phase are asked for before the shift's sixteen instructions, which hide part of their latency
  // (asked for at the very top of the chain instead: four spilled registers, 12.35 against 12.30 ms)
  v4i qa = *reinterpret_cast<const v4i *>(q), qb = *reinterpret_cast<const v4i *>(q + 1024);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (HT + HQ > 0) {
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = (int)(((unsigned)acc[r] << 8) + (unsigned)cin[r]);
  }
  // T_L . Q_L, always six products: the LDS read of product i + 1 is issued BEFORE the matrix instruction of product i (two operand
  // buffers in turn; the scheduling barriers keep the order -- left alone the compiler reads each operand into the same four registers
  // right after the instruction before it and waits out the whole LDS latency in front of every one of the six)
  {
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[0], qa, acc, 0, 0, 0);
    qa = *reinterpret_cast<const v4i *>(q + 2 * 1024);
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[1], qb, acc, 0, 0, 0);
    qb = *reinterpret_cast<const v4i *>(q + 3 * 1024);
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[2], qa, acc, 0, 0, 0);
    qa = *reinterpret_cast<const v4i *>(q + 4 * 1024);
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[3], qb, acc, 0, 0, 0);
    qb = *reinterpret_cast<const v4i *>(q + 5 * 1024);
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[4], qa, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[5], qb, acc, 0, 0, 0);
  }
  return acc;
}

// matrix instructions k3_chain issues for a block with these masks
template <int HT, int HQ>
__device__ __forceinline__ int k3_chain_products(unsigned tm, unsigned qm) {
  constexpr int HM = HT < HQ ? HT : HQ;
  if (HT + HQ == 0) return 6;
  int n = 7;  // the last phase's six and the middle phase's first
  if (HM > 0 && (tm & qm)) n += 1 + __builtin_popcount((tm & qm) >> 1);
  n += HT > 0 ? __builtin_popcount(qm) + __builtin_popcount(tm >> 1) : __builtin_popcount(qm >> 1);
  return n;
}

// Two blocks at once: the tile against TWO sub-tiles, two independent accumulators fed in turn (the tile's A operands serve both).  A wave's
// lone chain leaves the matrix pipe idle between its dependent products and while each operand comes out of LDS -- with four waves per SIMD
// (the register file allows no more) the pipe sat busy 43 % of the time for three rounds; two chains interleaved in one instruction stream
// halve what a wave exposes of both latencies.  The arithmetic of each chain is k3_chain's, product for product.
#ifndef TM_KNN3_DUAL
#define TM_KNN3_DUAL 0  // measured (round 5): 13.9 against 12.5 ms at 16 waves (twelve spilled registers), 13.7 against ~14.1 at 12 waves: the chain's own latency is not what the pipe waits for
#endif
template <int HT, int HQ, bool TD>
__device__ __forceinline__ void k3_chain2(const v4i (&T)[6 + HT], const v16i &cin, const uint8_t *q0, const uint8_t *q1, unsigned tm, unsigned qm0, unsigned qm1,
                                          v16i &acc0, v16i &acc1) {
  constexpr int HM = HT < HQ ? HT : HQ;
#pragma unroll
  for (int r = 0; r < 16; r++) { acc0[r] = 0; acc1[r] = 0; }
  if (HT + HQ == 0) { acc0 = cin; acc1 = cin; }
  if (HM > 0) {
    const unsigned h0 = tm & qm0, h1 = tm & qm1;
    if (h0 | h1) {
#pragma unroll
      for (int kc = 0; kc < HM; kc++) {
        if ((h0 >> kc) & 1u) acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6 + kc], *reinterpret_cast<const v4i *>(q0 + (6 + kc) * 1024), acc0, 0, 0, 0);  // T_H . Q_H
        if ((h1 >> kc) & 1u) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6 + kc], *reinterpret_cast<const v4i *>(q1 + (6 + kc) * 1024), acc1, 0, 0, 0);
      }
      if (h0) {
#pragma unroll
        for (int r = 0; r < 16; r++) acc0[r] = (int)((unsigned)acc0[r] << 8);
      }
      if (h1) {
#pragma unroll
        for (int r = 0; r < 16; r++) acc1[r] = (int)((unsigned)acc1[r] << 8);
      }
    }
  }
  if (HT + HQ > 0) {
#pragma unroll
    for (int kc = 0; kc < HQ; kc++) {
      if ((qm0 >> kc) & 1u) acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[kc], *reinterpret_cast<const v4i *>(q0 + (6 + kc) * 1024), acc0, 0, 0, 0);      // T_L . Q_H
      if ((qm1 >> kc) & 1u) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[kc], *reinterpret_cast<const v4i *>(q1 + (6 + kc) * 1024), acc1, 0, 0, 0);
    }
#pragma unroll
    for (int kc = 0; kc < HT; kc++)
      if ((tm >> kc) & 1u) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6 + kc], *reinterpret_cast<const v4i *>(q0 + kc * 1024), acc0, 0, 0, 0);                            // T_H . Q_L
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6 + kc], *reinterpret_cast<const v4i *>(q1 + kc * 1024), acc1, 0, 0, 0);
      }
  }
  v4i qa = *reinterpret_cast<const v4i *>(q0), qb = *reinterpret_cast<const v4i *>(q1);
  __builtin_amdgcn_sched_barrier(0);
  if (HT + HQ > 0) {
#pragma unroll
    for (int r = 0; r < 16; r++) acc0[r] = (int)(((unsigned)acc0[r] << 8) + (unsigned)cin[r]);
#pragma unroll
    for (int r = 0; r < 16; r++) acc1[r] = (int)(((unsigned)acc1[r] << 8) + (unsigned)cin[r]);
  }
  // T_L . Q_L, six products a chain, the chains in turn: an operand buffer is asked for again right behind the matrix instruction that read it,
  // two matrix instructions before its next use
#pragma unroll
  for (int kc = 0; kc < 6; kc++) {
    __builtin_amdgcn_sched_barrier(0);
    acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[kc], qa, acc0, 0, 0, 0);
    if (kc < 5) qa = *reinterpret_cast<const v4i *>(q0 + (kc + 1) * 1024);
    __builtin_amdgcn_sched_barrier(0);
    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[kc], qb, acc1, 0, 0, 0);
    if (kc < 5) qb = *reinterpret_cast<const v4i *>(q1 + (kc + 1) * 1024);
  }
}

// a tile's MFMA A operands and the rows' terms, straight into registers (global_load_dwordx4, 1 KB contiguous per instruction).  The
// pack (k_knn_pack, database side) holds per row the chain's starting value -- |t-c|^2 where the digits are those of 2 (t - c), |t-c|^2 >> 1
// where not -- and in the 15th word of the tile's box the 32 parities |t-c|^2 & 1 (bit = row); `pwh` = that word shifted so that bit
// k3_prow(r) is accumulator register r's row in this half-wave.
__device__ __forceinline__ constexpr int k3_prow(int r) { return (r & 3) + 8 * (r >> 2); }
template <int KT>
__device__ __forceinline__ void k3_load_rows(const uint8_t *tb, int half, v16i &cin, unsigned &pwh) {
#pragma unroll
  for (int q4 = 0; q4 < 4; q4++) {  // accumulator register r holds row (r&3) + 8*(r>>2) + 4*half
    const v4i x = *reinterpret_cast<const v4i *>(tb + KT * 1024 + (q4 * 8 + half * 4) * 4);
    cin[q4 * 4] = x[0]; cin[q4 * 4 + 1] = x[1]; cin[q4 * 4 + 2] = x[2]; cin[q4 * 4 + 3] = x[3];
  }
  pwh = *reinterpret_cast<const unsigned *>(tb + KT * 1024 + 128 + 14 * 4) >> (4 * half);
}
template <int KT>
__device__ __forceinline__ void k3_load_tile(const uint8_t *tb, int lane, int half, v4i (&T)[KT], v16i &cin, unsigned &pwh) {
#pragma unroll
  for (int kc = 0; kc < KT; kc++) T[kc] = *reinterpret_cast<const v4i *>(tb + (kc * 64 + lane) * 16);
  k3_load_rows<KT>(tb, half, cin, pwh);
}

// ... leaving out the high-digit chunks the tile's mask says are all zero (the chain skips their products: bit kc clear = T[6 + kc] never
// read).  On the literal bench clip the consume kernel pulled 43.7 GB per launch through the L2s' far side -- 3.4 TB/s, the matrix pipe
// waiting on it -- and two in five of those bytes were zeros.
template <int KT>
__device__ __forceinline__ void k3_load_tile_masked(const uint8_t *tb, int lane, int half, unsigned tm, v4i (&T)[KT], v16i &cin, unsigned &pwh) {
#pragma unroll
  for (int kc = 0; kc < 6; kc++) T[kc] = *reinterpret_cast<const v4i *>(tb + (kc * 64 + lane) * 16);
#pragma unroll
  for (int kc = 6; kc < KT; kc++) {
    T[kc] = v4i{0, 0, 0, 0};
    if ((tm >> (kc - 6)) & 1u) T[kc] = *reinterpret_cast<const v4i *>(tb + (kc * 64 + lane) * 16);
  }
  k3_load_rows<KT>(tb, half, cin, pwh);
}

// the minimum of the chain's sixteen values of a lane
__device__ __forceinline__ int k3_min16(const int (&t)[16]) {
  return min(min(min(min(t[0], t[1]), min(t[2], t[3])), min(min(t[4], t[5]), min(t[6], t[7]))),
             min(min(min(t[8], t[9]), min(t[10], t[11])), min(min(t[12], t[13]), min(t[14], t[15]))));
}
// what a block's sixteen values say about a lane's query before anything is made of them: a LOWER bound of the smallest d'' - qn.  With
// TD the chain's values are d'' - qn themselves; without, d'' - qn = 2 Y + parity and 2 min(Y) is at most one below the smallest.
template <bool TD>
__device__ __forceinline__ int k3_first_look(const v16i &acc) {
  int y[16];
#pragma unroll
  for (int r = 0; r < 16; r++) y[r] = acc[r];
  const int ym = k3_min16(y);
  return TD ? ym : (int)((unsigned)ym << 1);
}
// Can the block matter to a query whose bound is `bound` (its best d'' + 1, or its threshold + 1)?  With TD the test the epilogue makes
// itself; without, the first look is a lower bound that can lie one below the truth -- at -1 where a query meets its own row and both norms
// are odd (d'' + 1 = 0) -- so the compare is a signed one against the bound cut to 2^31 - 1: a value that wraps goes the safe way (in).
template <bool TD>
__device__ __forceinline__ bool k3_may_matter(const v16i &acc, unsigned qn, unsigned bound) {
  const unsigned look = (unsigned)k3_first_look<TD>(acc) + qn + 1u;
  return TD ? look <= bound : (int)look <= (int)min(bound, 0x7FFFFFFFu);
}
// ... and the values themselves (d'' - qn of accumulator register r)
template <bool TD>
__device__ __forceinline__ void k3_values(const v16i &acc, unsigned pwh, int (&t)[16]) {
  if (!TD) asm volatile("" : "+v"(pwh));  // (the sixteen parity bits are taken out HERE, on the rare path: seen through, the compiler takes them out once per tile and keeps sixteen registers for them)
#pragma unroll
  for (int r = 0; r < 16; r++) t[r] = TD ? acc[r] : (int)(((unsigned)acc[r] << 1) | ((pwh >> k3_prow(r)) & 1u));
}

// A block's epilogue: the row minimum of each query (lane & 31; the two half-waves hold 16 rows each) against its running best in LDS.
// d'' = 2 X + |t-c|^2 + 2 (|q-c|^2 >> 1) = SSD - parity.  Returns true in lanes whose improvement may have lowered the sub-tile's largest
// best (the caller refreshes the bound when any lane says so); `sm_now` = the sub-tile's published bound, or 0 to ask for a refresh on
// every improvement.
template <bool TD>
__device__ __forceinline__ bool k3_epilogue(const v16i &acc, unsigned pwh, int tile, int half, unsigned qn, unsigned long long *best, unsigned *tie,
                                            unsigned sm_now, unsigned cur_hi /* the query's best as read BEFORE the chain: stale only on the safe side (a best only falls) */) {
  bool refresh = false;
  if (k3_may_matter<TD>(acc, qn, cur_hi)) {  // (by the counters nearly half of the listed blocks improve or tie some query's best)
    int t[16];
    k3_values<TD>(acc, pwh, t);
    const int tm = k3_min16(t);
    const unsigned key_hi = (unsigned)tm + qn + 1u;  // d'' + 1 >= 0
    if (key_hi <= cur_hi) {
      // which row (the first one reaching the minimum), and is it alone: a compare, a select and an add-with-carry per register
      int ridx = 0;
      unsigned cnt = 0;
#pragma unroll
      for (int r = 15; r >= 0; r--) {
        const bool e = t[r] == tm;
        ridx = e ? r : ridx;
        cnt += e ? 1u : 0u;
      }
      const int row = (ridx & 3) + ((ridx & 12) << 1) + 4 * half;
      const unsigned long long key = ((unsigned long long)key_hi << 32) | (unsigned)((tile << 5) | row);
      const unsigned long long pre = atomicMin(best, key);
      const unsigned pre_hi = (unsigned)(pre >> 32);
      if (pre_hi == key_hi || cnt > 1) atomicMin(tie, key_hi);  // the value was reached a second time
      // The sub-tile's largest best can only have moved if this query held it: its old best is then no smaller than what the published
      // bound was made from ((bound - 3)^2; a bound of 0xFFFE stands for "some query has no best yet").  A refresh skipped by a race only
      // leaves the bound loose (and is made good at the end of the segment).
      const unsigned thr = sm_now > 3u ? (sm_now - 3u) * (sm_now - 3u) : 0u;
      refresh = key_hi < pre_hi && pre_hi >= thr;
    }
  }
  return refresh;
}

// Collection mode's epilogue.  The query's state in LDS: `best` = (tau + 1) << 32 | step (tau: its threshold, d'' <= tau), `base` = the
// threshold it came with + 1, `lad` = 7 x 16-bit counters: rows seen with d'' + 1 <= base - j step for j = 1..7 -- the ladder's rungs hang
// below the pass's first threshold at the spacing the host chose (tau / 8 in a first pass; an eighth of the bracket the pass before left --
// its last threshold down to the rung below it that did NOT fill -- afterwards: on data whose distances bunch, a rung spacing of tau / 8
// moved the threshold by a third per pass).  Every row within the threshold is appended to the query's candidate list UNTIL THE LIST IS
// FULL (bit 31 of lad[3]: from then on the query is only counted -- it will be scanned again anyway, and its appends, a global atomic and
// eight bytes each, were most of the collection scans' time on such data); once cand_k rows lie at or below a rung, the k-th nearest is at
// most that rung + 1 and the threshold drops to it (the threshold only ever falls, every value it takes is a valid bound: the waves of a
// workgroup, and the parts of a split group, lower it on their own evidence).  Returns true in lanes that lowered a threshold the
// sub-tile's bound may hang on.
template <bool TD>
__device__ __forceinline__ bool k3_epilogue_topk(const v16i &acc, unsigned pwh, int tile, bool countable, int half, unsigned qn, unsigned long long *best,
                                                 unsigned *lad, const unsigned *base_p, bool qvalid, int64_t q, const Knn3Args &a, unsigned sm_now) {
  const unsigned tau1 = k3_peek(reinterpret_cast<unsigned *>(best) + 1);  // tau + 1
  bool refresh = false;
  // (the first look is a lower bound: a lane it lets in with no row within the threshold appends and counts nothing -- every row is tested
  // by its own value below -- and at most lowers its threshold on the rungs' counts as they stand, which is valid at any time)
  if (qvalid && k3_may_matter<TD>(acc, qn, tau1)) {
    int t[16];
    k3_values<TD>(acc, pwh, t);
    const unsigned step = k3_peek(reinterpret_cast<unsigned *>(best)), base = *base_p;
    bool full = (k3_peek(&lad[3]) >> 31) != 0, filled = false;
    unsigned c[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const unsigned d1 = (unsigned)t[r] + qn + 1u;  // d'' + 1
      if (d1 <= tau1) {
        if (!full) {
          const int slot = atomicAdd(&a.cand_cnt[q], 1);  // (ends above the capacity for a query whose list filled: how the select stage knows)
          if (slot < a.cand_cap) a.cand[q * a.cand_cap + slot] = make_uint2(d1 - 1u, (unsigned)((tile << 5) | ((r & 3) + 8 * (r >> 2) + 4 * half)));
          else { full = true; filled = true; }
        }
#pragma unroll
        for (int j = 1; j <= 7; j++) c[j - 1] += (countable && d1 + (unsigned)j * step <= base) ? 1u : 0u;
      }
    }
    if (filled) atomicOr(&lad[3], 0x80000000u);
    // the counters saturate where they stop mattering: a rung that has its cand_k rows takes no more (so no 16-bit field overflows
    // into its neighbour: at most cand_k + 16 rows x 2 half-waves x 16 waves land in one)
    const unsigned k = (unsigned)a.cand_k;
    unsigned cur[4];
#pragma unroll
    for (int w = 0; w < 4; w++) cur[w] = k3_peek(&lad[w]);
    unsigned add[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const unsigned have = (cur[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
      if (have < k) add[j >> 1] += c[j] << (16 * (j & 1));
    }
    unsigned now[4];
#pragma unroll
    for (int w = 0; w < 4; w++) now[w] = add[w] ? atomicAdd(&lad[w], add[w]) + add[w] : cur[w];
    now[3] &= 0x7FFFFFFFu;
    int rung = 0;
#pragma unroll
    for (int j = 1; j <= 7; j++)
      if (((now[(j - 1) >> 1] >> (16 * ((j - 1) & 1))) & 0xFFFFu) >= k) rung = j;
    // cand_k rows have d'' + 1 <= base - rung step, i.e. SSD <= that: no row beyond it can be among the k nearest
    const unsigned tnew1 = base - (unsigned)rung * step + 1u;  // (new tau = the rung's bound) + 1
    if (rung > 0 && step > 0 && tnew1 < tau1) {
      const unsigned pre = atomicMin(reinterpret_cast<unsigned *>(best) + 1, tnew1);
      const unsigned thr = sm_now > 3u ? (sm_now - 3u) * (sm_now - 3u) : 0u;
      refresh = tnew1 < pre && pre >= thr;
    }
  }
  return refresh;
}

// ------------------------------------------------------------------------------------------------------------------ 1. seeds
// One workgroup (8 waves) per query group: wave w holds seed tile w in registers; the group's sub-tiles pass through LDS three at a time
// (double buffered, LDS-DMA), every wave scoring each against its tile.  Bests meet in LDS (64-bit atomic minimum per query) and leave as
// gbest / gtie / gsmax.  ~62 KB of LDS: two workgroups per CU, 100 registers per wave.
constexpr int K3_SEED_SLICE = 3;
template <int HT, int HQ, bool TD, bool UNUSED>
__global__ __launch_bounds__(K3_SEEDS * 64, 4) void k_knn_seed(const Knn3Args a) {
  constexpr int KT = 6 + HT, KQ = 6 + HQ;
  constexpr int T_BYTES = KT * 1024 + 128 + 64, Q_BYTES = KQ * 1024 + 128;
  constexpr int SL = (2 * K3_SEED_SLICE * KQ * 1024 + 10 * 1024 > 80 * 1024) ? K3_SEED_SLICE - 1 : K3_SEED_SLICE, NW = K3_SEEDS;  // two workgroups per CU
  __shared__ __attribute__((aligned(16))) uint8_t s_q[2][SL * KQ * 1024];
  __shared__ unsigned long long s_best[16 * 32];
  __shared__ unsigned s_tie[16 * 32];
  __shared__ int s_qn[16 * 32];
  __shared__ unsigned s_qm[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
  const int NS = a.ns;
  const int64_t g = blockIdx.x, st0 = g * NS;
  const int nvalid = (int)min((int64_t)NS, a.n_qtiles - st0);
  const int home = a.qmeta[st0 * 16 + 7];
  const int r0a = (int)max((int64_t)0, min((int64_t)home - (K3_SEEDS / 2 - 1), a.n_ttiles - K3_SEEDS));
  const int n_seed = (int)min((int64_t)K3_SEEDS, a.n_ttiles - r0a);
  auto stage = [&](int slice, int buf) {  // the slice's query operands, SL * KQ pieces of 1 KB over the waves
    for (int piece = wave; piece < SL * KQ; piece += NW) {
      const int s = piece / KQ, kc = piece - s * KQ;
      const int64_t st = min(st0 + slice * SL + s, a.n_qtiles - 1);
      const uint8_t *src = a.qpack + st * (int64_t)Q_BYTES + kc * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(&s_q[buf][piece * 1024]), 16, 0, 0);
    }
  };
#if TM_KNN3_STAMPS
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
  const unsigned long long st_begin = st_last;
#endif
  stage(0, 0);
  for (int i = tid; i < NS * 32; i += NW * 64) {
    const int64_t st = min(st0 + (i >> 5), a.n_qtiles - 1);
    s_qn[i] = reinterpret_cast<const int *>(a.qpack + st * (int64_t)Q_BYTES + KQ * 1024)[i & 31] & ~1;
    s_best[i] = ~0ull;
    s_tie[i] = ~0u;
  }
  if (tid < 16) s_qm[tid] = tid < nvalid ? (unsigned)a.qmeta[(st0 + tid) * 16 + 15] : 0u;
  const bool active = wave < n_seed;
  const int tile = r0a + (active ? wave : 0);
  const unsigned tm = (unsigned)__builtin_amdgcn_readfirstlane((int)a.thmask[tile]);
  v4i T[KT];
  v16i cin;
  unsigned pwh;
  k3_load_tile<KT>(a.tpack + (int64_t)tile * T_BYTES, lane, half, T, cin, pwh);
  const int n_slices = (nvalid + SL - 1) / SL;
  long long nblocks = 0, npairs = 0;
  const int vt = (int)min((int64_t)32, a.nt_rows - (int64_t)tile * 32);
  K3_STAMP(0);  // set-up: addresses, the tile's loads issued
  for (int sl = 0; sl < n_slices; sl++) {
    __syncthreads();  // slice sl has landed (the barrier waits for the LDS-DMA pieces); buffer (sl + 1) & 1 is no longer read
    K3_STAMP(sl == 0 ? 1 : 2);  // waiting for the first slice (and the tile) / for a later slice and the other waves
    if (sl + 1 < n_slices) stage(sl + 1, (sl + 1) & 1);
    if (active) {
      for (int j = 0; j < SL && sl * SL + j < nvalid; j++) {
        const int s = sl * SL + j;
        const int qi = s * 32 + (lane & 31);
        const unsigned cur_hi = k3_peek(reinterpret_cast<unsigned *>(&s_best[qi]) + 1), qn = (unsigned)s_qn[qi];  // (asked for before the chain: they arrive under it)
        const v16i acc = k3_chain<HT, HQ, TD>(T, cin, &s_q[sl & 1][j * KQ * 1024] + lane * 16, tm, (unsigned)__builtin_amdgcn_readfirstlane((int)s_qm[s]));
        k3_epilogue<TD>(acc, pwh, tile, half, qn, &s_best[qi], &s_tie[qi], 0u, cur_hi);
        nblocks++;
        npairs += (long long)vt * (int)min((int64_t)32, a.nq - (st0 + s) * 32);
      }
    }
    K3_STAMP(3);  // the slice's blocks (behind the wait for the NEXT slice's pieces the compiler puts before the first LDS read)
  }
  __syncthreads();
  K3_STAMP(4);
  for (int i = tid; i < nvalid * 32; i += NW * 64) {
    a.gbest[st0 * 32 + i] = s_best[i];
    a.gtie[st0 * 32 + i] = s_tie[i];
  }
  for (int s = wave; s < nvalid; s += NW) {
    const unsigned mx = k3_wave_umax((unsigned)(s_best[s * 32 + (lane & 31)] >> 32));
    if (lane == 0) a.gsmax[st0 + s] = k3_bound_of(mx);
  }
#if TM_KNN3_STAMPS
  K3_STAMP(5);  // results
  if (a.stats && lane == 0) {
    for (int i = 0; i < 6; i++) atomicAdd(a.stats + 20 + i, st_acc[i]);
    atomicAdd(a.stats + 26, __builtin_amdgcn_s_memtime() - st_begin);
  }
#endif
  // The seeds' own counters (blocks, tiles read, pairs), one set of atomics per WORKGROUP into one of 64 striped slots: 90 000 waves adding to
  // three words of one cache line took their turns at the L2 and held every other request of the kernel up behind them (2.5 of its 3.3 ms).
  if (a.seed_stats) {
    __shared__ unsigned long long s_cnt[2];
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
    if (lane == 0 && active) { atomicAdd(&s_cnt[0], (unsigned long long)nblocks); atomicAdd(&s_cnt[1], (unsigned long long)npairs); }
    __syncthreads();
    if (tid == 0) {
      unsigned long long *slot = a.seed_stats + (blockIdx.x & 63) * 4;
      atomicAdd(slot, s_cnt[0]);
      atomicAdd(slot + 1, (unsigned long long)n_seed);
      atomicAdd(slot + 2, s_cnt[1]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ 2. lists
// One workgroup (256 threads) per query group.  Runs of KNN_GROUP tiles are judged first, 16 of them at a time in outward order from the
// home run (one thread per (run, sub-tile) pair); then only the tiles of surviving runs are tested, 128 threads per run, against the
// sub-tiles that survived the run's box.  Entries collect in LDS in that order; a segment that is full (or the list's end) takes its place
// in the arena with one atomic add and is copied out.  Past the arena's capacity nothing is written and the segment's count is 0: the
// cursor keeps counting, the host sees the overflow with the scan's other counters and repeats the search with a larger arena.
#ifdef TM_KNN3_WITH_LISTS  // (not a template: defined in one translation unit, tm_knn.hip)
__global__ __launch_bounds__(K3_LIST_NT) void k_knn_lists(const Knn3Args a) {
  constexpr int ND = KNN_ND, NT = K3_LIST_NT, LCAP = K3_LCAP;
  constexpr int RB = NT / 16, RS = NT / KNN_GROUP;  // runs per batch, runs per tile-test step
  static_assert(KNN_GROUP == 128 && RB <= 64 && RS >= 1, "run batches are compacted by one wave");
  __shared__ int s_qbox[16 * 16];
  __shared__ unsigned s_smax[16];
  __shared__ unsigned s_rmask[RB], s_runs[RB];
  __shared__ int s_ctl[4];  // [0] entries in the buffer, [1] surviving runs of the batch, [2] arena offset of the segment being flushed
  __shared__ unsigned s_ltile[LCAP];
  __shared__ unsigned s_llbw[LCAP * 8];  // [LCAP][nsp / 2] words
  const int tid = threadIdx.x, lane = tid & 63;
  const int NS = a.ns, NSP = (NS + 1) & ~1, NSW = NSP / 2;
  const int64_t g = blockIdx.x, st0 = g * NS, n_ttiles = a.n_ttiles;
  const int nvalid = (int)min((int64_t)NS, a.n_qtiles - st0);
  for (int i = tid; i < 16 * 16; i += NT) s_qbox[i] = (i >> 4) < nvalid ? a.qmeta[(st0 + (i >> 4)) * 16 + (i & 15)] : 0;
  if (tid < 16) s_smax[tid] = tid < nvalid ? a.gsmax[st0 + tid] : 0u;
  if (tid == 0) s_ctl[0] = 0;
  __syncthreads();
  const int home = s_qbox[7];
  const int r0a = a.no_seeds ? 0 : (int)max((int64_t)0, min((int64_t)home - (K3_SEEDS / 2 - 1), n_ttiles - K3_SEEDS));
  const int r0b = a.no_seeds ? 0 : (int)min((int64_t)r0a + K3_SEEDS, n_ttiles);
  const int n_grp = (int)((n_ttiles + KNN_GROUP - 1) / KNN_GROUP), home_run = home / KNN_GROUP;
  const int total_run_slots = 2 * max(home_run, n_grp - 1 - home_run) + 1, n_run_batches = (total_run_slots + RB - 1) / RB;
  int nseg = 0;
  auto flush = [&]() {  // the buffer's entries become a segment (every thread calls it; n is uniform)
    const int n = s_ctl[0];
    if (n > 0) {
      if (tid == 0) {
        const unsigned long long off = atomicAdd(a.arena_cursor, (unsigned long long)n);
        const bool fits = off + (unsigned long long)n <= a.arena_cap;
        s_ctl[2] = fits ? 1 : 0;
        s_ctl[3] = (int)(unsigned)off;  // (arena_cap < 2^32 entries: checked on the host)
        if (nseg < a.max_segs) a.segs[g * a.max_segs + nseg] = make_uint2((unsigned)off, fits ? (unsigned)n : 0u);
      }
      __syncthreads();
      if (s_ctl[2]) {
        const unsigned off = (unsigned)s_ctl[3];
        for (int i = tid; i < n; i += NT) a.ltile[off + i] = s_ltile[i];
        unsigned *dst = reinterpret_cast<unsigned *>(a.llb) + (size_t)off * NSW;
        for (int i = tid; i < n * NSW; i += NT) dst[i] = s_llbw[(i / NSW) * 8 + (i % NSW)];
      }
      nseg++;
      __syncthreads();
      if (tid == 0) s_ctl[0] = 0;
      __syncthreads();
    }
  };
  for (int run_batch = 0; run_batch < n_run_batches; run_batch++) {
    if (tid < RB) s_rmask[tid] = 0;
    __syncthreads();
    {
      const int jr = run_batch * RB + (tid >> 4), s = tid & 15;
      const int off = (jr + 1) >> 1, run = (jr & 1) ? home_run + off : home_run - off;
      if (jr < total_run_slots && run >= 0 && run < n_grp && s < nvalid) {
        unsigned lbq = 0;
#pragma unroll
        for (int d = 0; d < ND; d++) {
          const int tlo = a.grp_lo[d * n_grp + run], thi = a.grp_hi[d * n_grp + run];
          const int gap = max(0, max(tlo - s_qbox[s * 16 + 8 + d], s_qbox[s * 16 + d] - thi)) >> 1;
          lbq += (unsigned)(gap * gap);
        }
        if (2u * k3_isqrt(lbq) <= s_smax[s]) atomicOr(&s_rmask[tid >> 4], 1u << s);
      }
    }
    __syncthreads();
    if (tid < 64) {  // surviving runs of the batch, in outward order
      const int jr = run_batch * RB + lane;
      const int off = (jr + 1) >> 1, run = (jr & 1) ? home_run + off : home_run - off;
      const unsigned m = lane < RB ? s_rmask[lane] : 0u;
      const unsigned long long alive = __builtin_amdgcn_ballot_w64(m != 0);
      if (m) s_runs[__popcll(alive & ((1ull << lane) - 1ull))] = ((unsigned)run << 16) | m;
      if (lane == 0) s_ctl[1] = __popcll(alive);
    }
    __syncthreads();
    const int run_alive = s_ctl[1];
    for (int run_k = 0; run_k < run_alive; run_k += RS) {
      if (s_ctl[0] + NT > LCAP) flush();  // (uniform: every thread reads the same count behind the last barrier)
      // tile tests: RS surviving runs per step, thread = (run, tile of the run); the run's sub-tile mask is uniform in a wave
      const int k = run_k + (tid >> 7);
      const unsigned rm = k < run_alive ? s_runs[k] : 0u;
      const unsigned rmask = (unsigned)__builtin_amdgcn_readfirstlane((int)(rm & 0xFFFFu));
      const int64_t tile = (int64_t)(rm >> 16) * KNN_GROUP + (tid & 127);
      const bool valid = rmask != 0 && tile < n_ttiles && !(tile >= r0a && tile < r0b);
      // 16-bit lower bounds of this lane's tile against the 16 sub-tile slots: a 256-bit shift register, one value pushed per slot (so
      // the loop stays rolled: no run-time register index), slot s ends in bits 16 * (s & 1) of lbw[s >> 1]
      unsigned lbw[8];
#pragma unroll
      for (int p = 0; p < 8; p++) lbw[p] = 0xFFFFFFFFu;
      bool any = false;
      if (rmask) {
        int tlo[ND], thi[ND];
        const int64_t tc = valid ? tile : 0;
#pragma unroll
        for (int d = 0; d < ND; d++) { tlo[d] = a.box_lo[(int64_t)d * n_ttiles + tc]; thi[d] = a.box_hi[(int64_t)d * n_ttiles + tc]; }
#pragma unroll 1
        for (int s = 0; s < 16; s++) {
          unsigned v16 = 0xFFFFu;
          if ((rmask >> s) & 1u) {  // uniform
            const v4i q0 = *reinterpret_cast<const v4i *>(&s_qbox[s * 16]), q1 = *reinterpret_cast<const v4i *>(&s_qbox[s * 16 + 4]),
                      q2 = *reinterpret_cast<const v4i *>(&s_qbox[s * 16 + 8]), q3 = *reinterpret_cast<const v4i *>(&s_qbox[s * 16 + 12]);
            const int qlo[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
            const int qhi[8] = {q2[0], q2[1], q2[2], q2[3], q3[0], q3[1], q3[2], q3[3]};
            unsigned lbq = 0;
#pragma unroll
            for (int d = 0; d < ND; d++) {
              const int gap = max(0, max(tlo[d] - qhi[d], qlo[d] - thi[d])) >> 1;
              lbq += (unsigned)(gap * gap);
            }
            const unsigned lb16 = min(0xFFFEu, 2u * k3_isqrt(lbq));
            if (valid && lb16 <= s_smax[s]) { v16 = lb16; any = true; }
          }
#pragma unroll
          for (int p = 0; p < 7; p++) lbw[p] = __builtin_amdgcn_alignbit(lbw[p + 1], lbw[p], 16);
          lbw[7] = (lbw[7] >> 16) | (v16 << 16);
        }
      }
      {  // ordered append: the waves take their places one after the other (wave w's tiles come before wave w + 1's)
        const unsigned long long pb = __builtin_amdgcn_ballot_w64(any);
        __shared__ int s_wcnt[NT / 64];
        if (lane == 0) s_wcnt[tid >> 6] = __popcll(pb);
        __syncthreads();
        int base = s_ctl[0];
        for (int w = 0; w < (tid >> 6); w++) base += s_wcnt[w];
        const int idx = base + __popcll(pb & ((1ull << lane) - 1ull));
        if (any) {
          s_ltile[idx] = ((unsigned)tile << 8) | (unsigned)a.thmask[tile];  // the entry carries the tile's high-chunk mask: the consumer has it before the tile
#pragma unroll
          for (int p = 0; p < 8; p++) s_llbw[idx * 8 + p] = lbw[p];
        }
        __syncthreads();
        if (tid == 0) { int t = s_ctl[0]; for (int w = 0; w < NT / 64; w++) t += s_wcnt[w]; s_ctl[0] = t; }
        __syncthreads();
      }
    }
  }
  flush();
  if (tid == 0) a.nsegs[g] = min(nseg, a.max_segs);
}

// collection mode: the bound every sub-tile's tiles are judged against, from its queries' thresholds (what the seed kernel derives from
// the bests in the nearest-neighbour search)
__global__ __launch_bounds__(256) void k_knn_tau_bounds(const int *__restrict__ tau, int64_t nq, int64_t n_qtiles, unsigned *__restrict__ gsmax) {
  const int lane = threadIdx.x & 63;
  for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w * 2 < n_qtiles; w += (int64_t)gridDim.x * 4) {
    const int64_t st = w * 2 + (lane >> 5), q = st * 32 + (lane & 31);
    unsigned v = (st < n_qtiles && q < nq) ? (unsigned)(tau[q] + 1) : 0u;  // tau + 1 = the SSD bound
    for (int o = 16; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    if ((lane & 31) == 0 && st < n_qtiles) gsmax[st] = k3_bound_of(v);
  }
}
#endif  // TM_KNN3_WITH_LISTS

// ------------------------------------------------------------------------------------------------------------------ 3. consume
// TOPK: collection mode for the k nearest rows (ann_kdtree_short_search_multi, tilingencoder.pas:1563).  Every query comes with a threshold
// (an upper bound of its k-th smallest SSD, a.tau); nothing is seeded, the lists are judged against the thresholds' bounds, and instead of
// keeping a minimum the epilogue appends every row within the threshold to the query's candidate list and walks the threshold down its
// ladder (k3_epilogue_topk).  One sub-tile fewer per group pays for the ladder's counters in LDS.  a.split > 1: the group's list is
// shared by that many workgroups (passes over few queries would otherwise leave most of the chip idle).
constexpr int k3_ns_topk(int kq) { return k3_ns(kq) > 1 ? k3_ns(kq) - 1 : 1; }
template <int HT, int HQ, bool TD, bool TOPK>
__global__ __launch_bounds__(K3_NT) void k_knn_consume(const Knn3Args a) {
  constexpr int KT = 6 + HT, KQ = 6 + HQ;
  constexpr int T_BYTES = KT * 1024 + 128 + 64, Q_BYTES = KQ * 1024 + 128;
  constexpr int NS = TOPK ? k3_ns_topk(KQ) : k3_ns(KQ), NSP = (NS + 1) & ~1, NW = K3_NW, NT = K3_NT, LCAP = K3_LCAP;
  // one LDS object, carved by hand (16-byte aligned pieces)
  constexpr int OFF_QN = NS * KQ * 1024, OFF_BEST = OFF_QN + NS * 128, OFF_TIE = OFF_BEST + NS * 256, OFF_QBOX = OFF_TIE + NS * 128,
                OFF_SMAX = OFF_QBOX + NS * 64, OFF_QMASK = OFF_SMAX + 64, OFF_CTL = OFF_QMASK + 64, OFF_LTILE = OFF_CTL + 64, OFF_LLB = OFF_LTILE + LCAP * 4,
                OFF_LAD = OFF_LLB + LCAP * NSP * 2, LDS_TOTAL = OFF_LAD + (TOPK ? NS * 32 * 16 : 0);
  static_assert(OFF_LAD == k3_lds_bytes(NS, KQ) && LDS_TOTAL <= K3_LDS, "LDS carve");
  __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_TOTAL];
  int *const s_qn = reinterpret_cast<int *>(lds + OFF_QN);                                  // [NS][32] 2 * (|q-c|^2 >> 1)
  unsigned long long *const s_best = reinterpret_cast<unsigned long long *>(lds + OFF_BEST);  // [NS][32] (d'' + 1) << 32 | sorted row
  unsigned *const s_tie = reinterpret_cast<unsigned *>(lds + OFF_TIE);                      // [NS][32] smallest d'' + 1 seen twice
  [[maybe_unused]] int *const s_qbox = reinterpret_cast<int *>(lds + OFF_QBOX);             // [NS][16] (kept in the carve: the dense mode's home)
  unsigned *const s_smax = reinterpret_cast<unsigned *>(lds + OFF_SMAX);                    // [16] upper bound of sqrt(largest best + 1)
  unsigned *const s_qmask = reinterpret_cast<unsigned *>(lds + OFF_QMASK);                  // [16] non-zero high-digit chunks of each sub-tile
  int *const s_ctl = reinterpret_cast<int *>(lds + OFF_CTL);                                // [0] list length, [1] cursor, [3] group
  unsigned *const s_ltile = reinterpret_cast<unsigned *>(lds + OFF_LTILE);                  // [LCAP]
  uint16_t *const s_llb = reinterpret_cast<uint16_t *>(lds + OFF_LLB);                      // [LCAP][NSP]
  [[maybe_unused]] unsigned *const s_lad = reinterpret_cast<unsigned *>(lds + OFF_LAD);     // TOPK: [NS][32][4] the ladder's 7 x 16-bit counters

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
#if TM_KNN3_STAMPS
  unsigned long long st_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
  const unsigned long long st_begin = st_last;
#endif
  // Persistent workgroups (one per CU): a workgroup draws query groups until none is left.  Groups are dealt in runs of K3_XCD_RUN
  // consecutive groups per XCD (the id is read from the hardware: placement is a matter of speed only); an XCD whose share is exhausted
  // helps the next one.
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
  long long nblocks = 0, nloads = 0, npairs = 0, nlisted = 0, nmfma = 0;
  const bool dense = a.mode == K3_MODE_DENSE;
  const int split = TOPK ? max(1, a.split) : 1;
  const int64_t n_units = a.n_groups * split;  // what the tickets deal: (group, part of its list)
  for (;;) {
  __syncthreads();  // the previous group's LDS is no longer read
  if (tid == 0) {
    int64_t gsel = -1;
    for (int k = 0; k < 8 && gsel < 0; k++) {
      const unsigned x = (xcc + k) & 7u;
      const unsigned t = atomicAdd(&a.tickets[x], 1u);
#if TM_KNN3_XCD_CONTIG
      const int64_t per = (n_units + 7) / 8;
      const int64_t gg = (int64_t)x * per + t;
      if ((int64_t)t < per && gg < n_units) gsel = gg;
#else
      const int64_t gg = ((int64_t)(t / K3_XCD_RUN) * 8 + x) * K3_XCD_RUN + (t % K3_XCD_RUN);
      if (gg < n_units) gsel = gg;
#endif
    }
    s_ctl[3] = (int)gsel;
  }
  __syncthreads();
  const int64_t unit = __builtin_amdgcn_readfirstlane(s_ctl[3]);
  if (unit < 0) break;
  const int64_t g = unit / split;
  const int part = (int)(unit - g * split);
  const int64_t st0 = g * NS;
  const int nvalid = (int)min((int64_t)NS, a.n_qtiles - st0);
  const int64_t n_ttiles = a.n_ttiles;

  // ---- prologue: the group's query operands, norms, and what the seeds left
  // (each phase outside the consume loop takes its thread index afresh, through an empty asm the compiler cannot see through: otherwise
  // the addresses it derives from it are computed once, before the persistent loop, and live -- spilled -- across the consume loop)
  int tp = threadIdx.x;
  asm volatile("" : "+v"(tp));
  for (int piece = wave; piece < NS * KQ; piece += NW) {
    const int s = piece / KQ, kc = piece - s * KQ;
    const int64_t st = min(st0 + s, a.n_qtiles - 1);
    const uint8_t *src = a.qpack + st * (int64_t)Q_BYTES + kc * 1024 + lane * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)(lds + piece * 1024), 16, 0, 0);
  }
  for (int i = tp; i < NS * 32; i += NT) {
    const int64_t st = min(st0 + (i >> 5), a.n_qtiles - 1);
    const bool real = (i >> 5) < nvalid && !dense;
    s_qn[i] = reinterpret_cast<const int *>(a.qpack + st * (int64_t)Q_BYTES + KQ * 1024)[i & 31] & ~1;
    if constexpr (TOPK) {  // (tau + 1) << 32 | rung spacing; queries that are padding get tau = -1: nothing is within it
      const bool qreal = real && st0 * 32 + i < a.nq;
      const int tau = qreal ? a.tau[st0 * 32 + i] : -1;
      int stp = tau > 0 ? tau >> 3 : 0;  // (seven rungs fit below the threshold whatever the host asks for)
      if (a.step && qreal && stp > 0) stp = min(stp, max(1, a.step[st0 * 32 + i]));
      s_best[i] = ((unsigned long long)(unsigned)(tau + 1) << 32) | (unsigned)stp;
      s_tie[i] = (unsigned)(tau + 1);  // the ladder's base (the nearest-neighbour scan's tie words are not used in this mode)
      for (int w = 0; w < 4; w++) s_lad[i * 4 + w] = 0;
    } else {
      s_best[i] = real ? a.gbest[st0 * 32 + i] : ~0ull;
      s_tie[i] = real ? a.gtie[st0 * 32 + i] : ~0u;
    }
  }
  if (tp < 16) {
    s_smax[tp] = (tp < nvalid && !dense) ? a.gsmax[st0 + tp] : 0xFFFEu;
    s_qmask[tp] = tp < nvalid ? (unsigned)a.qmeta[(st0 + tp) * 16 + 15] : 0u;
  }
  const int nseg = dense ? (int)((n_ttiles + LCAP - 1) / LCAP) : a.nsegs[g];
  __syncthreads();  // (waits for the LDS-DMA pieces too)
  const unsigned qmask_v = s_qmask[lane & 15];  // lane s: sub-tile s's mask of non-zero high-digit chunks
  K3_STAMP(0);  // prologue

  for (int seg = 0; seg < nseg; seg++) {
    // ---------------------------------------------------------------- the next segment of the group's tile list, into LDS
    int list_n;
    int tl = threadIdx.x;
    asm volatile("" : "+v"(tl));
    if (dense) {  // every tile, every sub-tile
      list_n = (int)min((int64_t)LCAP, n_ttiles - (int64_t)seg * LCAP);
      for (int i = tl; i < list_n; i += NT) {
        s_ltile[i] = ((unsigned)(seg * LCAP + i) << 8) | (unsigned)a.thmask[seg * LCAP + i];
        for (int p = 0; p < NSP; p++) s_llb[i * NSP + p] = p < nvalid ? 0 : 0xFFFF;
      }
    } else {
      const uint2 sg = a.segs[g * a.max_segs + seg];
      list_n = (int)sg.y;
      if (list_n != 0 && ((unsigned long long)sg.x + sg.y > a.arena_cap || list_n > LCAP)) {  // never by construction (a segment that did not fit the arena has 0 entries): a guard against a corrupted segment table
        if (tl == 0) atomicMax(a.stats + 27, 0x100000000ull | sg.y);
        list_n = 0;
      }
      for (int i = tl; i < list_n; i += NT) s_ltile[i] = a.ltile[sg.x + i];
      const unsigned *src = reinterpret_cast<const unsigned *>(a.llb) + (size_t)sg.x * (NSP / 2);
#pragma unroll 1
      for (int i = tl; i < list_n * (NSP / 2); i += NT) reinterpret_cast<unsigned *>(s_llb)[i] = src[i];
    }
    if (tl == 0) s_ctl[1] = 0;
    __syncthreads();
    K3_STAMP(1);  // segment load
    // ---------------------------------------------------------------- consume: every wave on its own
    {
      nlisted += (wave == 0) ? list_n : 0;
      auto next_entry = [&](int &tile_o, unsigned &tm_o, int &lb_o, unsigned &mask_o, int &sm_o) -> bool {
        for (;;) {
          int j = 0;
          if (lane == 0) j = atomicAdd(&s_ctl[1], 1);
          j = __builtin_amdgcn_readfirstlane(j);
          if (TOPK) j = j * split + part;  // a split group: this workgroup takes every split-th entry
          if (j >= list_n) return false;
#if TM_KNN3_REFRESH_EVERY
          if ((j & (TM_KNN3_REFRESH_EVERY - 1)) == TM_KNN3_REFRESH_EVERY - 1)
            for (int s = 0; s < nvalid; s++) {  // every so many entries the popping wave makes every sub-tile's bound anew from its bests: what the
                                                // races of the refresh rule left loose ends here (the second shape did this at every list's end)
              const unsigned mx = k3_wave_umax(k3_peek(reinterpret_cast<unsigned *>(s_best) + (s * 32 + (lane & 31)) * 2 + 1));
              if (mx != ~0u && lane == 0) atomicMin(&s_smax[s], k3_bound_of(mx));
            }
#endif
          const int t = (int)s_ltile[j];
          const int lb = lane < NSP ? (int)s_llb[j * NSP + lane] : 0xFFFF;
          const int sm = lane < NS ? (int)k3_peek(&s_smax[lane]) : -1;
          const unsigned m = (unsigned)__builtin_amdgcn_ballot_w64(lb <= sm);
          // Entry word = tile << 8 | mask: the tile comes out by a SHIFT.  (An earlier layout, tile | mask << 24 with `tile = word & 0xFFFFFF`,
          // met a compiler combine that treats the 64-bit multiply by the tile size as a 24-bit multiply, drops the AND as redundant for one,
          // and then selects a full 32-bit v_mad_u64_u32: the loads went to word * 12 KB -- a memory aperture violation on the GPU box.)
          if (m) { tile_o = __builtin_amdgcn_readfirstlane((int)((unsigned)t >> 8)); tm_o = (unsigned)__builtin_amdgcn_readfirstlane(t) & 0xFFu; lb_o = lb; mask_o = m; sm_o = sm; return true; }
        }
      };
      int tile = 0, lbv = 0, smv = 0;
      unsigned mask = 0, tmw = 0;
      bool have = next_entry(tile, tmw, lbv, mask, smv);
      while (have) {
        const unsigned tm = tmw;
        if (tile >= n_ttiles) {  // never by construction: a list entry outside the database (a guard: the loads below must not follow it)
          if (lane == 0) atomicMax(a.stats + 27, 0x200000000ull | (unsigned)tile);
          break;
        }
        v4i T[KT];
        v16i cin;
        unsigned pwh;
        k3_load_tile_masked<KT>(a.tpack + (int64_t)tile * T_BYTES, lane, half, tm, T, cin, pwh);
        nloads++;
        // the entry after this one is chosen while the loads fly
        int ntile = 0, nlb = 0, nsm = 0;
        unsigned nmask = 0, ntm = 0;
        const bool nhave = next_entry(ntile, ntm, nlb, nmask, nsm);
#if TM_KNN3_STAMPS
        K3_STAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        K3_STAMP(4);  // waiting for the tile's loads (what is left of their latency behind the choice of the next entry)
#endif
        const int vt = (int)min((int64_t)32, a.nt_rows - (int64_t)tile * 32);
        // a block's epilogue and the refresh of its sub-tile's bound
        auto finish = [&](const v16i &acc, int s, unsigned sm_now, unsigned cur_hi, unsigned qn) {
          const int qi = s * 32 + (lane & 31);
          bool refresh;
          if constexpr (TOPK) {
            const int64_t q = (st0 + s) * 32 + (lane & 31);
            // (the last database tile pads with copies of its last row: they are candidates like any row -- the select stage drops them --
            // but must not count towards a rung)
            refresh = k3_epilogue_topk<TD>(acc, pwh, tile, tile != (int)n_ttiles - 1, half, qn, &s_best[qi], &s_lad[qi * 4], &s_tie[qi], q < a.nq, q, a, sm_now);
          } else {
            refresh = k3_epilogue<TD>(acc, pwh, tile, half, qn, &s_best[qi], &s_tie[qi], sm_now, cur_hi);
          }
          if (__builtin_amdgcn_ballot_w64(refresh)) {  // refresh the sub-tile's largest best (bests only go down: a late writer is only loose)
            const unsigned mx = k3_wave_umax(k3_peek(reinterpret_cast<unsigned *>(s_best) + qi * 2 + 1));  // = largest d'' + 1
            if (mx != ~0u && lane == 0) atomicMin(&s_smax[s], k3_bound_of(mx));
          }
          nblocks++;
          npairs += (long long)vt * (int)min((int64_t)32, a.nq - (st0 + s) * 32);
        };
        // the next sub-tile that still wants the tile (its best may have tightened since the entry was popped); -1: none left
        auto pick = [&](unsigned &sm_o) -> int {
          while (mask) {
            const int s = __builtin_ctz(mask);
            mask &= mask - 1;
            const int lbs = __builtin_amdgcn_readlane(lbv, s);
#ifndef TM_KNN3_REPEEK
#define TM_KNN3_REPEEK 1
#endif
            const unsigned sm_now = TM_KNN3_REPEEK ? (unsigned)__builtin_amdgcn_readfirstlane((int)k3_peek(&s_smax[s])) : (unsigned)__builtin_amdgcn_readlane(smv, s);
            if (lbs > (int)sm_now) continue;
            sm_o = sm_now;
            return s;
          }
          return -1;
        };
        for (;;) {
          unsigned sm0 = 0, sm1 = 0;
          const int s0 = pick(sm0);
          K3_STAMP(5);  // choosing the block's sub-tile (the bound's fresh look)
          if (s0 < 0) break;
          // the sub-tile's mask comes out of a register (lane s of qmask_v), the query's best and norm are asked for before the chain: every
          // LDS round trip a block can do without, or start early, is one the wave does not sit out between its matrix instructions
          const unsigned qm0 = (unsigned)__builtin_amdgcn_readlane((int)qmask_v, s0);
          const int qi0 = s0 * 32 + (lane & 31);
#ifndef TM_KNN3_PRE_QN
#define TM_KNN3_PRE_QN 1
#endif
          const unsigned cur0 = TOPK ? 0u : k3_peek(reinterpret_cast<unsigned *>(&s_best[qi0]) + 1);
          unsigned qn0 = 0;
          if (TM_KNN3_PRE_QN) qn0 = (unsigned)s_qn[qi0];
          const int s1 = (TM_KNN3_DUAL && !TOPK) ? pick(sm1) : -1;
          if (s1 >= 0) {  // two blocks at once
            const unsigned qm1 = (unsigned)__builtin_amdgcn_readlane((int)qmask_v, s1);
            const int qi1 = s1 * 32 + (lane & 31);
            const unsigned cur1 = k3_peek(reinterpret_cast<unsigned *>(&s_best[qi1]) + 1), qn1 = (unsigned)s_qn[qi1];
            v16i acc0, acc1;
            k3_chain2<HT, HQ, TD>(T, cin, lds + s0 * (KQ * 1024) + lane * 16, lds + s1 * (KQ * 1024) + lane * 16, tm, qm0, qm1, acc0, acc1);
            nmfma += 12 + __builtin_popcount(tm & qm0) + __builtin_popcount(qm0) + __builtin_popcount(tm & qm1) + __builtin_popcount(qm1) + 2 * __builtin_popcount(tm);
            finish(acc0, s0, sm0, cur0, qn0);
            finish(acc1, s1, sm1, cur1, qn1);
          } else {
            // (s_setprio 2 around the chain -- a wave inside its chain ahead of the waves between chains -- measured: no change)
            // (measured in round 5 and dropped: every product's operand read one product ahead -- the reads issued whatever the predicates say, with
            // no lane enabled where the product will not run, so that the waits can name the read they need: 13.5 against 11.8 ms; the thirteen
            // reads that read nothing still pass through the CU's one LDS queue)
            const v16i acc = k3_chain<HT, HQ, TD>(T, cin, lds + s0 * (KQ * 1024) + lane * 16, tm, qm0);
            nmfma += k3_chain_products<HT, HQ>(tm, qm0);
            K3_STAMP(6);  // the chain, up to the issue of its last matrix instruction
            if (!TM_KNN3_PRE_QN) qn0 = (unsigned)s_qn[qi0];
            finish(acc, s0, sm0, cur0, qn0);
            K3_STAMP(7);  // the epilogue (behind the wait for the chain's result)
          }
        }
        tile = ntile; tmw = ntm; lbv = nlb; mask = nmask; have = nhave; smv = nsm;
      }
    }
    K3_STAMP(2);  // consuming
    __syncthreads();
    if (seg + 1 < nseg)
      for (int s = wave; s < nvalid; s += NW) {  // every sub-tile's bound made anew from its bests: what the races of the refresh rule left loose ends here
        const unsigned mx = k3_wave_umax(k3_peek(reinterpret_cast<unsigned *>(s_best) + (s * 32 + (lane & 31)) * 2 + 1));
        if (mx != ~0u && lane == 0) atomicMin(&s_smax[s], k3_bound_of(mx));
      }
    K3_STAMP(3);  // waiting for the other waves at the end of a segment
  }

  // ---- results, in the format k_knn_refine / k_knn_ties read
  int tr = threadIdx.x;
  asm volatile("" : "+v"(tr));
  for (int i = tr; i < NS * 32; i += NT) {
    const int64_t st = st0 + (i >> 5);
    if (st >= a.n_qtiles) continue;
    const unsigned long long k = s_best[i];
    const unsigned hi = (unsigned)(k >> 32);
    const int64_t q = st * 32 + (i & 31);
    if constexpr (TOPK) {  // the final threshold: k_topk_select drops the stored candidates above it
      if (q < a.nq && hi > 0) atomicMin(&a.tau[q], (int)(hi - 1u));
    } else {
      a.best_key[q] = (int)(hi - 1u);
      a.best_tile[q] = (int)(((unsigned)k & 0x3fffffffu) | (s_tie[i] == hi ? (1u << 30) : 0u));
    }
  }
  K3_STAMP(0);  // results (counted with the prologue)
  }  // next query group
#if TM_KNN3_STAMPS
  if (a.stats && lane == 0) {
    for (int i = 0; i < 5; i++) atomicAdd(a.stats + 4 + i, st_acc[i]);
    atomicAdd(a.stats + 9, __builtin_amdgcn_s_memtime() - st_begin);
    for (int i = 5; i < 8; i++) atomicAdd(a.stats + 5 + i, st_acc[i]);  // [10..12]: inside "consume" (the host reads them before it reuses the slots)
  }
#endif
  if (a.stats && lane == 0) {
    atomicAdd(a.stats, (unsigned long long)nblocks);
    atomicAdd(a.stats + 1, (unsigned long long)nloads);
    atomicAdd(a.stats + 2, (unsigned long long)npairs);
    if (wave == 0) atomicAdd(a.stats + 3, (unsigned long long)nlisted);
    atomicAdd(a.stats + 19, (unsigned long long)nmfma);  // matrix instructions issued (a full chain has 6 + HT + HQ + min(HT, HQ))
  }
}

// one per HT, defined in tm_knn3_k<HT>.hip
template <int HT> void knn3_launch_seed_ht(int hq, const Knn3Args &a, hipStream_t stream);
template <int HT> void knn3_launch_consume_ht(int hq, const Knn3Args &a, hipStream_t stream);
template <int HT> void knn3_launch_collect_ht(int hq, const Knn3Args &a, hipStream_t stream);  // k_knn_consume<.., TOPK = true>
int knn3_sub_tiles(int hq);       // NS of the queries' digit plan
int knn3_sub_tiles_topk(int hq);  // ... in collection mode

// FLAG: the kernels' fourth template argument (k_knn_consume's TOPK; the seed kernel ignores its own)
#define TM_KNN3_LAUNCH(KERNEL, HT, HQ, FLAG)                                                            \
  do {                                                                                                 \
    if (a.tdouble) hipLaunchKernelGGL((KERNEL<HT, HQ, true, FLAG>), grid, block, 0, stream, a);        \
    else hipLaunchKernelGGL((KERNEL<HT, HQ, false, FLAG>), grid, block, 0, stream, a);                 \
  } while (0)
#define TM_KNN3_CASE(KERNEL, HT, HQ, FLAG) \
  case HQ: TM_KNN3_LAUNCH(KERNEL, HT, HQ, FLAG); break;
#define TM_KNN3_SWITCH(KERNEL, HT, FLAG)                                                                                      \
  switch (hq) {                                                                                                               \
    TM_KNN3_CASE(KERNEL, HT, 0, FLAG) TM_KNN3_CASE(KERNEL, HT, 1, FLAG) TM_KNN3_CASE(KERNEL, HT, 2, FLAG)                     \
    TM_KNN3_CASE(KERNEL, HT, 3, FLAG) TM_KNN3_CASE(KERNEL, HT, 4, FLAG) TM_KNN3_CASE(KERNEL, HT, 5, FLAG)                     \
    default: TM_KNN3_LAUNCH(KERNEL, HT, 6, FLAG);                                                                             \
  }

#define TM_KNN3_DEFINE_HT(HT)                                                                         \
  template <> void knn3_launch_seed_ht<HT>(int hq, const Knn3Args &a, hipStream_t stream) {          \
    const dim3 grid((unsigned)a.n_groups), block(K3_SEEDS * 64);                                      \
    TM_KNN3_SWITCH(k_knn_seed, HT, false)                                                             \
  }                                                                                                   \
  template <> void knn3_launch_consume_ht<HT>(int hq, const Knn3Args &a, hipStream_t stream) {       \
    const dim3 grid((unsigned)a.grid_blocks), block(K3_NT);                                           \
    TM_KNN3_SWITCH(k_knn_consume, HT, false)                                                          \
  }                                                                                                   \
  template <> void knn3_launch_collect_ht<HT>(int hq, const Knn3Args &a, hipStream_t stream) {       \
    const dim3 grid((unsigned)a.grid_blocks), block(K3_NT);                                           \
    TM_KNN3_SWITCH(k_knn_consume, HT, true)                                                           \
  }

}  // namespace tmx
