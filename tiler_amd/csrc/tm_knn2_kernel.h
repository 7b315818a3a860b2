// tm_knn2_kernel.h -- the nearest-neighbour scan of the KNN stage, second shape ("KNN").
//
// Same arithmetic as tm_knn_kernel.h (exact SSD through int8 digit products on v_mfma_i32_32x32x32_i8, one accumulator shifted
// between three phases) and the same packed operands, boxes and curve; what changed is who holds what:
//   * A workgroup is a whole CU's worth of waves (16, four per SIMD, <= 128 registers each) and owns NS consecutive query
//     sub-tiles (32 queries each, NS = 10..16 by the queries' digit plan).  Their MFMA B operands sit in LDS for the whole
//     launch of the group (up to 144 KB), their running minima too (one 64-bit word per query: d''+1 | sorted row, updated with
//     ds_min_rtn_u64).
//   * Database tiles never touch LDS.  A wave takes the next entry of the group's tile list (one LDS atomic), reads the tile's
//     11-12 KB straight into registers as MFMA A operands (global_load_dwordx4, 1 KB contiguous per instruction) and runs one
//     19-MFMA chain per sub-tile that still wants the tile.  Waves never wait for each other inside a list: no per-tile barrier,
//     no staging ring, and the matrix pipe of a SIMD is fed by whichever of its four waves has a tile in registers.
//   * The list carries, per entry and sub-tile, the box lower bound of that pair as a 16-bit square root (rounded down), made
//     once by all threads.  Re-judging a pair against the sub-tile's current largest best is then one compare of two 16-bit
//     values instead of a 7-dimensional box test.
//   * Workgroup -> query group mapping keeps the 32 groups that run together on one XCD consecutive on the curve, so the tiles
//     they stream are the same ones and come from that XCD's L2.
// Exactness is unchanged: a pair is skipped only when its lower bound exceeds the sub-tile's largest best + 1 (both sides of the
// 16-bit compare are rounded the safe way), the minimum VALUE is exact, and a second row reaching the same value raises the tie
// flag that k_knn_ties settles by original index.
#pragma once
#include "tm_knn_kernel.h"

namespace tmx {

#ifndef TM_KNN2_WAVES
#define TM_KNN2_WAVES 16  // 16: one workgroup per CU; 8: two (half the LDS each, fewer sub-tiles per group)
#endif
#ifndef TM_KNN2_WGS
#define TM_KNN2_WGS (16 / TM_KNN2_WAVES)  // workgroups per CU (they split its LDS)
#endif
constexpr int K2_NW = TM_KNN2_WAVES;
constexpr int K2_NT = K2_NW * 64;
constexpr int K2_WGS = TM_KNN2_WGS;
constexpr int K2_LDS = 163840 / K2_WGS;
constexpr int K2_LCAP = K2_NT;        // list entries: one chunk of tile slots always fits
#ifndef TM_KNN2_BREAK
#define TM_KNN2_BREAK (K2_LCAP / 2)
#endif
#ifndef TM_KNN2_SEEDS
#define TM_KNN2_SEEDS 8
#endif
constexpr int K2_SEEDS = TM_KNN2_SEEDS;  // tiles around the group's position on the curve, visited first by every sub-tile
constexpr int K2_XCD_RUN = 32 * K2_WGS;  // workgroups that run together on one XCD
#ifndef TM_KNN2_FAST_ISQRT
#define TM_KNN2_FAST_ISQRT 1
#endif
#ifndef TM_KNN2_QNT
#define TM_KNN2_QNT 0  // 2: the group's query operands are fetched with the non-temporal hint (read once per group: they need not displace tiles in L2)
#endif
#ifndef TM_KNN2_PAIR
#define TM_KNN2_PAIR 0  // 1: two blocks (sub-tiles) of a tile in flight per wave
#endif
#ifndef TM_KNN2_REFRESH
#define TM_KNN2_REFRESH 1  // when a sub-tile's bound is made anew: 0 on every improvement, 1 when the improved query may have held the maximum, 2 = 1 + every eighth block
#endif
#ifndef TM_KNN2_XCD_CONTIG
#define TM_KNN2_XCD_CONTIG 0
#endif
#ifndef TM_KNN2_STAMPS
#define TM_KNN2_STAMPS 0  // diagnostic build: s_memtime spans of the phases, summed over all waves into stats[4..]
#endif
#if TM_KNN2_STAMPS
#define K2_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#define K2_FS(i, expr) do { fs[i] += (expr); } while (0)
#define K2_NOW() __builtin_amdgcn_s_memtime()
// the time after which a vector result is there: an instruction that reads it, issued in order before the clock is read
#define K2_AFTER(v) do { const int d_ = __builtin_amdgcn_readfirstlane(v); asm volatile("" :: "s"(d_)); } while (0)
#else
#define K2_FS(i, expr) do { } while (0)
#define K2_NOW() 0ull
#define K2_AFTER(v) do { } while (0)
#define K2_STAMP(i) do { } while (0)
#endif

constexpr int k2_lds_bytes(int ns, int kq) { return ns * (kq * 1024 + 576) + 64 + 576 + K2_LCAP * (4 + 2 * ((ns + 1) & ~1)); }
constexpr int k2_ns(int kq) {
  int ns = 16;
  while (ns > 1 && k2_lds_bytes(ns, kq) > K2_LDS) ns--;
  return ns;
}

struct Knn2Args {
  const uint8_t *tpack; int64_t n_ttiles, nt_rows;
  const int *box_lo, *box_hi, *grp_lo, *grp_hi;  // database tile boxes [KNN_ND][n_ttiles], boxes of runs of KNN_GROUP tiles
  const uint8_t *qpack; int64_t n_qtiles, nq;
  const int *qmeta;   // [n_qtiles][16]: box lo[7], home tile, box hi[7], pad
  int prune;
  int tdouble;        // the database pack holds the digits of 2 (t - c)
  int *best_key, *best_tile;
  unsigned long long *stats;  // [0] (tile, sub-tile) blocks evaluated, [1] tiles read, [2] exact (query, row) pairs, [3] list entries
  int64_t n_groups;
  int grid_blocks;    // persistent workgroups: one per CU (two with 8-wave workgroups)
  unsigned *tickets;  // [8] zeroed before the launch: next run-slot of each XCD's share of the groups
};

__device__ __forceinline__ unsigned k2_wave_umax(unsigned x) {  // max over lanes 0..31 (every lane of a row of 16 ends with its row's)
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true));
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true));
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, true));
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, true));
  return max((unsigned)__builtin_amdgcn_readlane((int)x, 0), (unsigned)__builtin_amdgcn_readlane((int)x, 16));
}

// a fresh look at an LDS word other waves update (relaxed workgroup-scope load: a plain ds_read_b32 the compiler may not cache)
__device__ __forceinline__ unsigned k2_peek(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// largest r with r * r <= x (x < 2^31): the hardware's approximate root (within one unit of the last place, i.e. well within one of
// the integer root here) set right by one step either way -- the correctly rounded sqrtf costs a dozen instructions more per test
__device__ __forceinline__ unsigned k2_isqrt(unsigned x) {
#if TM_KNN2_FAST_ISQRT
  unsigned r = (unsigned)__builtin_amdgcn_sqrtf((float)x);
  r += ((r + 1u) * (r + 1u) <= x) ? 1u : 0u;
  r -= (r * r > x) ? 1u : 0u;
  return r;
#else
  unsigned r = (unsigned)sqrtf((float)x);
  r -= (r * r > x) ? 1u : 0u;
  r -= (r * r > x) ? 1u : 0u;
  return r;
#endif
}

// One block's chain: X over the digit products on one accumulator shifted between the phases (tm_knn_kernel.h); with TD the rows' own
// term rides in on the second shift.  `q` = the sub-tile's B operands in LDS at this lane's 16 bytes.
template <int HT, int HQ, bool TD>
__device__ __forceinline__ v16i k2_chain(const v4i (&T)[6 + HT], const v16i &ntr, const uint8_t *q) {
  constexpr int HM = HT < HQ ? HT : HQ;
  v16i acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0;
  if (TD && HT + HQ == 0) acc = ntr;  // a single phase: the rows' term is the chain's starting value
  if (HM > 0) {
#pragma unroll
    for (int kc = 0; kc < HM; kc++)
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6 + kc], *reinterpret_cast<const v4i *>(q + (6 + kc) * 1024), acc, 0, 0, 0);  // T_H . Q_H
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = (int)((unsigned)acc[r] << 8);
  }
  if (HT + HQ > 0) {
#pragma unroll
    for (int kc = 0; kc < HQ; kc++)
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[kc], *reinterpret_cast<const v4i *>(q + (6 + kc) * 1024), acc, 0, 0, 0);      // T_L . Q_H
#pragma unroll
    for (int kc = 0; kc < HT; kc++)
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6 + kc], *reinterpret_cast<const v4i *>(q + kc * 1024), acc, 0, 0, 0);        // T_H . Q_L
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = (int)(((unsigned)acc[r] << 8) + (TD ? (unsigned)ntr[r] : 0u));
  }
#pragma unroll
  for (int kc = 0; kc < 6; kc++)
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[kc], *reinterpret_cast<const v4i *>(q + kc * 1024), acc, 0, 0, 0);              // T_L . Q_L
  return acc;
}

// Two blocks of one tile at once: the phases of the two chains alternate, so that one chain's shifts (and the wait for the results they
// read) run under the other chain's products -- a wave on its own keeps the matrix pipe fed through both chains instead of leaving it to
// the other waves of its SIMD between the phases.
template <int HT, int HQ, bool TD>
__device__ __forceinline__ void k2_chain2(const v4i (&T)[6 + HT], const v16i &ntr, const uint8_t *qa, const uint8_t *qb, v16i &A, v16i &B) {
  constexpr int HM = HT < HQ ? HT : HQ;
#pragma unroll
  for (int r = 0; r < 16; r++) { A[r] = 0; B[r] = 0; }
  if (TD && HT + HQ == 0) { A = ntr; B = ntr; }
  if (HM > 0) {
#pragma unroll
    for (int kc = 0; kc < HM; kc++) A = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6 + kc], *reinterpret_cast<const v4i *>(qa + (6 + kc) * 1024), A, 0, 0, 0);
#pragma unroll
    for (int kc = 0; kc < HM; kc++) B = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6 + kc], *reinterpret_cast<const v4i *>(qb + (6 + kc) * 1024), B, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) A[r] = (int)((unsigned)A[r] << 8);
  }
  if (HT + HQ > 0) {
#pragma unroll
    for (int kc = 0; kc < HQ; kc++) A = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[kc], *reinterpret_cast<const v4i *>(qa + (6 + kc) * 1024), A, 0, 0, 0);
#pragma unroll
    for (int kc = 0; kc < HT; kc++) A = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6 + kc], *reinterpret_cast<const v4i *>(qa + kc * 1024), A, 0, 0, 0);
    if (HM > 0) {
#pragma unroll
      for (int r = 0; r < 16; r++) B[r] = (int)((unsigned)B[r] << 8);
    }
#pragma unroll
    for (int kc = 0; kc < HQ; kc++) B = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[kc], *reinterpret_cast<const v4i *>(qb + (6 + kc) * 1024), B, 0, 0, 0);
#pragma unroll
    for (int kc = 0; kc < HT; kc++) B = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[6 + kc], *reinterpret_cast<const v4i *>(qb + kc * 1024), B, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) A[r] = (int)(((unsigned)A[r] << 8) + (TD ? (unsigned)ntr[r] : 0u));
  }
#pragma unroll
  for (int kc = 0; kc < 6; kc++) A = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[kc], *reinterpret_cast<const v4i *>(qa + kc * 1024), A, 0, 0, 0);
  if (HT + HQ > 0) {
#pragma unroll
    for (int r = 0; r < 16; r++) B[r] = (int)(((unsigned)B[r] << 8) + (TD ? (unsigned)ntr[r] : 0u));
  }
#pragma unroll
  for (int kc = 0; kc < 6; kc++) B = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[kc], *reinterpret_cast<const v4i *>(qb + kc * 1024), B, 0, 0, 0);
}

// TD: the database digits are those of 2 (t - c) (KnnPlan::tscale = 2), so the chain yields 2 X at once and the rows' own term
// |t-c|^2 rides in on the second digit shift (one v_lshl_add_u32 per register instead of a shift there and a shift-add at the end)
template <int HT, int HQ, bool TD>
__global__ __launch_bounds__(K2_NT) void k_knn_scan2(const Knn2Args a) {
  constexpr int KT = 6 + HT, KQ = 6 + HQ, ND = KNN_ND;
  constexpr int T_BYTES = KT * 1024 + 128 + 64, Q_BYTES = KQ * 1024 + 128;
  constexpr int NS = k2_ns(KQ), NSP = (NS + 1) & ~1, NW = K2_NW, NT = K2_NT, LCAP = K2_LCAP;
  // one LDS object, carved by hand (16-byte aligned pieces)
  constexpr int OFF_QN = NS * KQ * 1024, OFF_BEST = OFF_QN + NS * 128, OFF_TIE = OFF_BEST + NS * 256, OFF_QBOX = OFF_TIE + NS * 128,
                OFF_SMAX = OFF_QBOX + NS * 64, OFF_CTL = OFF_SMAX + 64, OFF_LTILE = OFF_CTL + 576, OFF_LLB = OFF_LTILE + LCAP * 4,
                LDS_TOTAL = OFF_LLB + LCAP * NSP * 2;
  static_assert(LDS_TOTAL == k2_lds_bytes(NS, KQ) && LDS_TOTAL <= K2_LDS, "LDS carve");
  __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_TOTAL];
  int *const s_qn = reinterpret_cast<int *>(lds + OFF_QN);                                  // [NS][32] 2 * (|q-c|^2 >> 1)
  unsigned long long *const s_best = reinterpret_cast<unsigned long long *>(lds + OFF_BEST);  // [NS][32] (d'' + 1) << 32 | sorted row
  unsigned *const s_tie = reinterpret_cast<unsigned *>(lds + OFF_TIE);                      // [NS][32] smallest d'' + 1 seen twice
  int *const s_qbox = reinterpret_cast<int *>(lds + OFF_QBOX);                              // [NS][16] lo[8] | hi[8]
  unsigned *const s_smax = reinterpret_cast<unsigned *>(lds + OFF_SMAX);                    // [16] upper bound of sqrt(largest best + 1)
  int *const s_ctl = reinterpret_cast<int *>(lds + OFF_CTL);                                // [0] list length, [1] cursor, [2] surviving runs
  unsigned *const s_rmask = reinterpret_cast<unsigned *>(lds + OFF_CTL + 64);               // [64] sub-tiles that want a run of the batch
  unsigned *const s_runs = reinterpret_cast<unsigned *>(lds + OFF_CTL + 320);               // [64] surviving runs: run << 16 | sub-tile mask
  unsigned *const s_ltile = reinterpret_cast<unsigned *>(lds + OFF_LTILE);                  // [LCAP]
  uint16_t *const s_llb = reinterpret_cast<uint16_t *>(lds + OFF_LLB);                      // [LCAP][NSP]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
#if TM_KNN2_STAMPS
  unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
  unsigned long long fs[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // inside the consume loop: [0] block top, [1] chain(s), [2] epilogue(s), [3] blocks that enter the update branch, [4] refreshes, [5] single blocks, [6] pairs, [7] tile load issue
  const unsigned long long st_begin = st_last;
#endif
  // Persistent workgroups (one per CU): a workgroup draws query groups until none is left, so that no CU waits for a 16-wave
  // workgroup with 155 KB of LDS to be launched 44 times over.  Groups are dealt in runs of K2_XCD_RUN consecutive groups per XCD (the
  // id is read from the hardware: placement is a matter of speed only), so that the groups running together on one XCD are
  // neighbours on the curve and stream the same tiles through that XCD's L2; an XCD whose share is exhausted helps the next one.
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
  long long nblocks = 0, nloads = 0, npairs = 0, nlisted = 0;
  for (;;) {
  __syncthreads();  // the previous group's LDS is no longer read
  if (tid == 0) {
    int64_t gsel = -1;
    for (int k = 0; k < 8 && gsel < 0; k++) {
      const unsigned x = (xcc + k) & 7u;
      const unsigned t = atomicAdd(&a.tickets[x], 1u);
#if TM_KNN2_XCD_CONTIG
      const int64_t per = (a.n_groups + 7) / 8;  // every XCD walks one contiguous eighth of the curve, group after group
      const int64_t gg = (int64_t)x * per + t;
      if ((int64_t)t < per && gg < a.n_groups) gsel = gg;
#else
      const int64_t gg = ((int64_t)(t / K2_XCD_RUN) * 8 + x) * K2_XCD_RUN + (t % K2_XCD_RUN);
      if (gg < a.n_groups) gsel = gg;
#endif
    }
    s_ctl[3] = (int)gsel;
  }
  __syncthreads();
  const int64_t g = __builtin_amdgcn_readfirstlane(s_ctl[3]);
  if (g < 0) break;
  const int64_t st0 = g * NS;
  const int nvalid = (int)min((int64_t)NS, a.n_qtiles - st0);
  const int64_t n_ttiles = a.n_ttiles;

  // ---- prologue: the group's query operands, norms, boxes
  for (int piece = wave; piece < NS * KQ; piece += NW) {
    const int s = piece / KQ, kc = piece - s * KQ;
    const int64_t st = min(st0 + s, a.n_qtiles - 1);
    const uint8_t *src = a.qpack + st * (int64_t)Q_BYTES + kc * 1024 + lane * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)(lds + piece * 1024), 16, 0, TM_KNN2_QNT);
  }
  for (int i = tid; i < NS * 32; i += NT) {
    const int64_t st = min(st0 + (i >> 5), a.n_qtiles - 1);
    s_qn[i] = reinterpret_cast<const int *>(a.qpack + st * (int64_t)Q_BYTES + KQ * 1024)[i & 31] & ~1;
    s_best[i] = ~0ull;
    s_tie[i] = ~0u;
  }
  for (int i = tid; i < NS * 16; i += NT) s_qbox[i] = a.qmeta[min(st0 + (i >> 4), a.n_qtiles - 1) * 16 + (i & 15)];
  if (tid < 16) s_smax[tid] = 0xFFFEu;
  if (tid == 0) { s_ctl[0] = 0; s_ctl[1] = 0; }
  __syncthreads();  // (waits for the LDS-DMA pieces too)

  K2_STAMP(0);  // prologue
  const int prune = a.prune;
  const int home = prune ? s_qbox[7] : 0;
  const int r0a = prune ? (int)max((int64_t)0, min((int64_t)home - (K2_SEEDS / 2 - 1), n_ttiles - K2_SEEDS)) : 0;
  const int r0b = prune ? (int)min((int64_t)r0a + K2_SEEDS, n_ttiles) : 0;
  const int64_t total_slots = prune ? 2 * max((int64_t)home, n_ttiles - 1 - home) + 1 : n_ttiles;
  const int n_chunks = (int)((total_slots + NT - 1) / NT);

  int chunk = 0;
  // pruned lists: runs of KNN_GROUP tiles in outward order from the home tile's run, NT / 16 runs per batch
  const int n_grp = (int)((n_ttiles + KNN_GROUP - 1) / KNN_GROUP), home_run = home / KNN_GROUP;
  const int total_run_slots = 2 * max(home_run, n_grp - 1 - home_run) + 1, n_run_batches = (total_run_slots + NT / 16 - 1) / (NT / 16);
  int run_batch = 0, run_k = 0, run_alive = 0;
  int round = prune ? -1 : 0;
  for (;;) {
    // ---------------------------------------------------------------- the group's tile list
    if (round < 0) {  // seeds: every sub-tile visits the tiles around the group's position, lower bound 0; two entries per tile (each
                      // with half of the sub-tiles), so that all the waves of the workgroup get one
      if (tid < 2 * (r0b - r0a)) {
        s_ltile[tid] = (unsigned)(r0a + (tid >> 1));
        const int h0 = (tid & 1) ? (nvalid + 1) / 2 : 0, h1 = (tid & 1) ? nvalid : (nvalid + 1) / 2;
        for (int p = 0; p < NSP; p++) s_llb[tid * NSP + p] = (p >= h0 && p < h1) ? 0 : 0xFFFF;
      }
      if (tid == 0) { s_ctl[0] = 2 * (r0b - r0a); s_ctl[1] = 0; }
      __syncthreads();
    } else if (!prune) {  // dense: every tile, every sub-tile, one chunk of NT tiles per list
      const int64_t tile = (int64_t)chunk * NT + tid;
      if (tile < n_ttiles) {
        s_ltile[tid] = (unsigned)tile;
        for (int p = 0; p < NSP; p++) s_llb[tid * NSP + p] = p < nvalid ? 0 : 0xFFFF;
      }
      if (tid == 0) { s_ctl[0] = (int)min((int64_t)NT, n_ttiles - (int64_t)chunk * NT); s_ctl[1] = 0; }
      chunk++;
      __syncthreads();
    } else {
      // Pruned.  Runs of KNN_GROUP tiles are judged first, RB of them at a time in outward order from the home run (one thread per
      // (run, sub-tile) pair); then only the tiles of surviving runs are tested, 128 threads per run, against the sub-tiles that
      // survived the run's box.  A step that does not fit behind the earlier entries is dropped and repeated after the consume.
      constexpr int RB = NT / 16, RS = NT / KNN_GROUP;  // runs per batch, runs per tile-test step
      static_assert(KNN_GROUP == 128 && RB <= 64, "run batches are compacted by one wave");
      if (tid == 0) { s_ctl[0] = 0; s_ctl[1] = 0; }
      __syncthreads();
      for (;;) {
        if (run_k >= run_alive) {  // next batch of runs
          if (run_batch >= n_run_batches) break;
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
              if (2u * k2_isqrt(lbq) <= k2_peek(&s_smax[s])) atomicOr(&s_rmask[tid >> 4], 1u << s);
            }
          }
          __syncthreads();
          if (wave == 0) {  // surviving runs of the batch, in outward order
            const int jr = run_batch * RB + lane;
            const int off = (jr + 1) >> 1, run = (jr & 1) ? home_run + off : home_run - off;
            const unsigned m = lane < RB ? s_rmask[lane] : 0u;
            const unsigned long long alive = __builtin_amdgcn_ballot_w64(m != 0);
            if (m) s_runs[__popcll(alive & ((1ull << lane) - 1ull))] = ((unsigned)run << 16) | m;
            if (lane == 0) s_ctl[2] = __popcll(alive);
          }
          __syncthreads();
          run_alive = __builtin_amdgcn_readfirstlane(s_ctl[2]);
          run_k = 0;
          run_batch++;
          continue;
        }
        const int n0 = __builtin_amdgcn_readfirstlane((int)k2_peek(reinterpret_cast<unsigned *>(&s_ctl[0])));
        if (n0 >= TM_KNN2_BREAK) break;  // enough for now: what comes later is judged with tighter bests
        // tile tests: RS surviving runs per step, thread = (run, tile of the run); the run's sub-tile mask is uniform in a wave
        const int k = run_k + (tid >> 7);
        const unsigned rm = k < run_alive ? s_runs[k] : 0u;
        const unsigned rmask = (unsigned)__builtin_amdgcn_readfirstlane((int)(rm & 0xFFFFu));
        const int64_t tile = (int64_t)(rm >> 16) * KNN_GROUP + (tid & 127);
        const bool valid = rmask != 0 && tile < n_ttiles && !(tile >= r0a && tile < r0b);
        // 16-bit lower bounds of this lane's tile against the 16 sub-tile slots: a 256-bit shift register, one value pushed per
        // slot (so the loop stays rolled: no run-time register index), slot s ends in bits 16 * (s & 1) of lbw[s >> 1]
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
              const unsigned lb16 = min(0xFFFEu, 2u * k2_isqrt(lbq));
              if (valid && lb16 <= k2_peek(&s_smax[s])) { v16 = lb16; any = true; }
            }
#pragma unroll
            for (int p = 0; p < 7; p++) lbw[p] = __builtin_amdgcn_alignbit(lbw[p + 1], lbw[p], 16);
            lbw[7] = (lbw[7] >> 16) | (v16 << 16);
          }
        }
        {  // ordered append inside the wave, one LDS atomic per wave
          const unsigned long long pb = __builtin_amdgcn_ballot_w64(any);
          int base = 0;
          if (pb) {
            if (lane == 0) base = atomicAdd(&s_ctl[0], __popcll(pb));
            base = __builtin_amdgcn_readfirstlane(base);
          }
          const int idx = base + __popcll(pb & ((1ull << lane) - 1ull));
          if (any && idx < LCAP) {
            s_ltile[idx] = (unsigned)tile;
#pragma unroll
            for (int p = 0; p < NSP / 2; p++) reinterpret_cast<unsigned *>(s_llb)[idx * (NSP / 2) + p] = lbw[p];
          }
        }
        __syncthreads();
        const int n1 = __builtin_amdgcn_readfirstlane((int)k2_peek(reinterpret_cast<unsigned *>(&s_ctl[0])));
        if (n1 > LCAP) {  // did not fit behind the earlier steps: drop it, consume, do it again
          __syncthreads();
          if (tid == 0) s_ctl[0] = n0;
          __syncthreads();
          break;
        }
        run_k += RS;
        __syncthreads();  // everyone has read n1 before the next step's atomics move it
      }
    }
    K2_STAMP(1);  // list building (with its barriers)
    // ---------------------------------------------------------------- consume: every wave on its own
    {
      const int list_n = __builtin_amdgcn_readfirstlane((int)k2_peek(reinterpret_cast<unsigned *>(&s_ctl[0])));
      nlisted += (wave == 0) ? list_n : 0;
      auto next_entry = [&](int &tile_o, int &lb_o, unsigned &mask_o) -> bool {
        for (;;) {
          int j = 0;
          if (lane == 0) j = atomicAdd(&s_ctl[1], 1);
          j = __builtin_amdgcn_readfirstlane(j);
          if (j >= list_n) return false;
          const int t = (int)s_ltile[j];
          const int lb = lane < NSP ? (int)s_llb[j * NSP + lane] : 0xFFFF;
          const int sm = lane < NS ? (int)k2_peek(&s_smax[lane]) : -1;
          const unsigned m = (unsigned)__builtin_amdgcn_ballot_w64(lb <= sm);
          if (m) { tile_o = __builtin_amdgcn_readfirstlane(t); lb_o = lb; mask_o = m; return true; }
        }
      };
      int tile = 0, lbv = 0;
      unsigned mask = 0;
      bool have = next_entry(tile, lbv, mask);
      while (have) {
        // the tile's MFMA A operands and norms, straight into registers
        [[maybe_unused]] const unsigned long long fl0 = K2_NOW();
        const uint8_t *tb = a.tpack + (int64_t)tile * T_BYTES;
        v4i T[KT];
#pragma unroll
        for (int kc = 0; kc < KT; kc++) T[kc] = *reinterpret_cast<const v4i *>(tb + (kc * 64 + lane) * 16);
        v16i ntr;  // |t-c|^2 of accumulator row r: (r&3) + 8*(r>>2) + 4*half
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
          const v4i x = *reinterpret_cast<const v4i *>(tb + KT * 1024 + (q4 * 8 + half * 4) * 4);
          ntr[q4 * 4] = x[0]; ntr[q4 * 4 + 1] = x[1]; ntr[q4 * 4 + 2] = x[2]; ntr[q4 * 4 + 3] = x[3];
        }
        nloads++;
        K2_FS(7, K2_NOW() - fl0);
        // the entry after this one is chosen while the loads fly
        int ntile = 0, nlb = 0;
        unsigned nmask = 0;
#if TM_KNN2_STAMPS
        const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
#endif
        const bool nhave = next_entry(ntile, nlb, nmask);
#if TM_KNN2_STAMPS
        st_acc[6] += __builtin_amdgcn_s_memtime() - tp0;  // popping the next entry
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_acc[7] += __builtin_amdgcn_s_memtime() - tp0;  // ... until the tile has landed
#endif
        const int vt = (int)min((int64_t)32, a.nt_rows - (int64_t)tile * 32);
        // a block's epilogue: the row minimum of each query against its running best
        auto epilogue = [&](const int s, const v16i &acc) {
          // d'' = 2 X + |t-c|^2 + 2 (|q-c|^2 >> 1) = SSD - parity (X: the chain over undoubled digits; with TD the chain holds 2 X + |t-c|^2
          // already); the row minimum is taken without the query's own term
          int t[16];
#pragma unroll
          for (int r = 0; r < 16; r++) t[r] = TD ? acc[r] : (int)(((unsigned)acc[r] << 1) + (unsigned)ntr[r]);
          const int tm = min(min(min(min(t[0], t[1]), min(t[2], t[3])), min(min(t[4], t[5]), min(t[6], t[7]))),
                             min(min(min(t[8], t[9]), min(t[10], t[11])), min(min(t[12], t[13]), min(t[14], t[15]))));
          const int qi = s * 32 + (lane & 31);
          const unsigned key_hi = (unsigned)tm + (unsigned)s_qn[qi] + 1u;  // d'' + 1 >= 0
          const unsigned cur_hi = k2_peek(reinterpret_cast<unsigned *>(s_best) + qi * 2 + 1);
          bool refresh = false;
          K2_FS(3, __builtin_amdgcn_ballot_w64(key_hi <= cur_hi) ? 1 : 0);
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
            const unsigned sm_now = k2_peek(&s_smax[s]);  // (its latency rides with the atomic's)
            const unsigned long long pre = atomicMin(&s_best[qi], key);
            const unsigned pre_hi = (unsigned)(pre >> 32);
            if (pre_hi == key_hi || cnt > 1) atomicMin(&s_tie[qi], key_hi);  // the value was reached a second time
            // The sub-tile's largest best can only have moved if this query held it: its old best is then no smaller than what the
            // published bound was made from ((bound - 3)^2; a bound of 0xFFFE stands for "some query has no best yet").  A refresh
            // skipped by a race only leaves the bound loose (and is made good at the end of the list).
            const unsigned thr = sm_now > 3u ? (sm_now - 3u) * (sm_now - 3u) : 0u;
#if TM_KNN2_REFRESH == 0
            refresh = key_hi < pre_hi;
#elif TM_KNN2_REFRESH == 1
            refresh = key_hi < pre_hi && pre_hi >= thr;
#else
            refresh = key_hi < pre_hi && (pre_hi >= thr || (nblocks & 7) == 0);
#endif
          }
          if (__builtin_amdgcn_ballot_w64(refresh)) {  // refresh the sub-tile's largest best (bests only go down: a late writer is only loose)
            K2_FS(4, 1);
            const unsigned h = k2_peek(reinterpret_cast<unsigned *>(s_best) + (s * 32 + (lane & 31)) * 2 + 1);
            const unsigned mx = k2_wave_umax(h);  // = largest d'' + 1 = the SSD bound the box test compares with
            if (mx != ~0u && lane == 0) atomicMin(&s_smax[s], min(0xFFFEu, (unsigned)__builtin_amdgcn_sqrtf((float)mx) + 3u));
          }
          nblocks++;
          npairs += (long long)vt * (int)min((int64_t)32, a.nq - (st0 + s) * 32);
        };
        while (mask) {
          [[maybe_unused]] const unsigned long long f0 = K2_NOW();
#if TM_KNN2_PAIR
          {  // the sub-tiles' bests may have tightened since the entry was popped: one fresh look at all of them
            const int sm = lane < NS ? (int)k2_peek(&s_smax[lane]) : -1;
            mask &= (unsigned)__builtin_amdgcn_ballot_w64(lbv <= sm);
            if (!mask) break;
          }
          const int sa = __builtin_ctz(mask);
          mask &= mask - 1;
          if (mask) {  // two blocks of the tile at once
            const int sb = __builtin_ctz(mask);
            mask &= mask - 1;
            v16i accA, accB;
            [[maybe_unused]] const unsigned long long f1 = K2_NOW();
            k2_chain2<HT, HQ, TD>(T, ntr, lds + sa * (KQ * 1024) + lane * 16, lds + sb * (KQ * 1024) + lane * 16, accA, accB);
            K2_AFTER(accB[0]);
            [[maybe_unused]] const unsigned long long f2 = K2_NOW();
            epilogue(sa, accA);
            epilogue(sb, accB);
            K2_FS(0, f1 - f0); K2_FS(1, f2 - f1); K2_FS(2, K2_NOW() - f2); K2_FS(6, 1);
          } else {
            [[maybe_unused]] const unsigned long long f1 = K2_NOW();
            const v16i acc = k2_chain<HT, HQ, TD>(T, ntr, lds + sa * (KQ * 1024) + lane * 16);
            K2_AFTER(acc[0]);
            [[maybe_unused]] const unsigned long long f2 = K2_NOW();
            epilogue(sa, acc);
            K2_FS(0, f1 - f0); K2_FS(1, f2 - f1); K2_FS(2, K2_NOW() - f2); K2_FS(5, 1);
          }
#else
          const int s = __builtin_ctz(mask);
          mask &= mask - 1;
          // the sub-tile's best may have tightened since the entry was popped
          const int lbs = __builtin_amdgcn_readlane(lbv, s);
          const int sms = __builtin_amdgcn_readfirstlane((int)k2_peek(&s_smax[s]));
          if (lbs > sms) { K2_FS(0, K2_NOW() - f0); continue; }
          [[maybe_unused]] const unsigned long long f1 = K2_NOW();
          const v16i acc = k2_chain<HT, HQ, TD>(T, ntr, lds + s * (KQ * 1024) + lane * 16);
          K2_AFTER(acc[0]);
          [[maybe_unused]] const unsigned long long f2 = K2_NOW();
          epilogue(s, acc);
          K2_FS(0, f1 - f0); K2_FS(1, f2 - f1); K2_FS(2, K2_NOW() - f2); K2_FS(5, 1);
#endif
        }
        tile = ntile; lbv = nlb; mask = nmask; have = nhave;
      }
    }
    K2_STAMP(round < 0 ? 2 : 3);  // consuming: seeds / lists
    __syncthreads();
#if TM_KNN2_REFRESH != 0
    for (int s = wave; s < nvalid; s += NW) {  // every sub-tile's bound made anew from its bests: what the races of the refresh rule above left loose ends here
      const unsigned mx = k2_wave_umax(k2_peek(reinterpret_cast<unsigned *>(s_best) + (s * 32 + (lane & 31)) * 2 + 1));
      if (mx != ~0u && lane == 0) atomicMin(&s_smax[s], min(0xFFFEu, (unsigned)__builtin_amdgcn_sqrtf((float)mx) + 3u));
    }
#endif
    K2_STAMP(round < 0 ? 4 : 5);  // waiting for the other waves at the end of a list
    if (round >= 0 && (prune ? (run_batch >= n_run_batches && run_k >= run_alive) : chunk >= n_chunks)) break;
    round++;
  }

  // ---- results, in the first scan shape's format (k_knn_refine / k_knn_ties read them)
  for (int i = tid; i < NS * 32; i += NT) {
    const int64_t st = st0 + (i >> 5);
    if (st >= a.n_qtiles) continue;
    const unsigned long long k = s_best[i];
    const unsigned hi = (unsigned)(k >> 32);
    const int64_t q = st * 32 + (i & 31);
    a.best_key[q] = (int)(hi - 1u);
    a.best_tile[q] = (int)(((unsigned)k & 0x3fffffffu) | (s_tie[i] == hi ? (1u << 30) : 0u));
  }
  }  // next query group
#if TM_KNN2_STAMPS
  K2_STAMP(8);  // results
  if (a.stats && lane == 0) {
    for (int i = 0; i < 9; i++) atomicAdd(a.stats + 4 + i, st_acc[i]);
    for (int i = 0; i < 8; i++) atomicAdd(a.stats + 18 + i, fs[i]);  // (bytes 160.. of the counters: behind the group tickets)
    atomicAdd(a.stats + 13, __builtin_amdgcn_s_memtime() - st_begin);
  }
#endif
  if (a.stats && lane == 0) {
    atomicAdd(a.stats, (unsigned long long)nblocks);
    atomicAdd(a.stats + 1, (unsigned long long)nloads);
    atomicAdd(a.stats + 2, (unsigned long long)npairs);
    if (wave == 0) atomicAdd(a.stats + 3, (unsigned long long)nlisted);
  }
}

// one per HT, defined in tm_knn2_k<HT>.hip
template <int HT> void knn2_launch_ht(int hq, const Knn2Args &a, hipStream_t stream);
int knn2_sub_tiles(int hq);  // NS of the queries' digit plan

#define TM_KNN2_LAUNCH(HT, HQ)                                                                          \
  do {                                                                                                 \
    if (a.tdouble) hipLaunchKernelGGL((k_knn_scan2<HT, HQ, true>), grid, block, 0, stream, a);         \
    else hipLaunchKernelGGL((k_knn_scan2<HT, HQ, false>), grid, block, 0, stream, a);                  \
  } while (0)
#define TM_KNN2_CASE(HT, HQ) \
  case HQ: TM_KNN2_LAUNCH(HT, HQ); break;

#define TM_KNN2_DEFINE_HT(HT)                                                                         \
  template <> void knn2_launch_ht<HT>(int hq, const Knn2Args &a, hipStream_t stream) {               \
    const dim3 grid((unsigned)a.grid_blocks), block(K2_NT);                                           \
    switch (hq) {                                                                                     \
      TM_KNN2_CASE(HT, 0) TM_KNN2_CASE(HT, 1) TM_KNN2_CASE(HT, 2) TM_KNN2_CASE(HT, 3)                 \
      TM_KNN2_CASE(HT, 4) TM_KNN2_CASE(HT, 5)                                                         \
      default: TM_KNN2_LAUNCH(HT, 6);                                                                 \
    }                                                                                                 \
  }

}  // namespace tmx
