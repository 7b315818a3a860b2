// tm_kmodes.hip -- A17: TKModes.ComputeKModes (kmodes.pas:923-1094) for rows of cKModesFeatureCount = 80 bytes.
//
// Unreachable in the reference snapshot (nothing calls it) but named by the north star, so it is built as an operator of its
// own: tm_stage_kmodes (host pointers, like the Pascal arrays) and tm_stage_kmodes_dev (device pointers).  Everything the clustering
// keeps -- memberships, member counts, the k x 80 x num_modalities histograms of MovePointCat, the modes, the LCG's seed -- lives in HBM:
//   * MatchingDissim (kmodes.pas:248-259) = sum |a - b| + 2048 x #(a != b) over the 80 bytes of a row: twenty v_sad_u8
//     (four bytes each) plus a count of the non-zero bytes of a XOR b, the GPU form of the reference's psadbw / pcmpeqb / popcnt
//     (kmodes.pas:314-450);
//   * GetMinMatchingDissim for a range of points against the k modes (the LAST minimum wins, `dis <= best`);
//   * the farthest-first initialisation's min-distance update and its pick (the LAST largest among unused points wins, 757-762);
//   * KModesIter (851-921) bin by bin, as the reference walks it (a bin of 960 points is scored against the modes as they stand at its
//     start, then its points move one after the other): k_kmodes_argmin over the bin, k_kmodes_walk (the moves in order, the
//     empty-cluster repair with RandInt, 88-92), k_kmodes_apply (Huang's online mode update MovePointCat, 774-803, one wave per
//     (cluster, attribute) pair: the pairs are independent, the moves inside a pair are applied in order).
// What bounds it is that bin-serial rule, not bytes: three dependent launches per 960 points.  The host keeps the stopping rule with its
// three graces (1040-1049) and reads sixteen bytes per iteration.
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <vector>

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {
namespace {

typedef unsigned long long u64;
constexpr int KM_ATTRS = 80, KM_WORDS = 20;

__device__ __forceinline__ unsigned km_dissim(const uint32_t *__restrict__ a, const uint32_t (&b)[KM_WORDS]) {
  unsigned sad = 0, zero_bytes = 0;
#pragma unroll
  for (int w = 0; w < KM_WORDS; w++) {
    const uint32_t x = a[w], y = b[w];
    sad = __builtin_amdgcn_sad_u8(x, y, sad);
    const uint32_t d = x ^ y;
    // 0x80 in every byte of d that is zero
    zero_bytes += __builtin_popcount(~(((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d | 0x7f7f7f7fu));
  }
  return sad + ((unsigned)(KM_ATTRS - zero_bytes) << 11);  // at most 80 * (255 + 2048): 32 bits are plenty
}

// clust[i], dis[i] for points [first, n): the LAST mode reaching the minimum wins (kmodes.pas:272, 414)
__global__ __launch_bounds__(256) void k_kmodes_argmin(const uint32_t *__restrict__ rows, int64_t first, int64_t n, const uint32_t *__restrict__ modes, int k,
                                                       int32_t *__restrict__ clust, unsigned *__restrict__ dis) {
  extern __shared__ uint32_t s_modes[];  // [k][20]
  for (int e = threadIdx.x; e < k * KM_WORDS; e += 256) s_modes[e] = modes[e];
  __syncthreads();
  for (int64_t i = first + blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    uint32_t p[KM_WORDS];
#pragma unroll
    for (int w = 0; w < KM_WORDS; w += 4) {
      const uint4 v = *reinterpret_cast<const uint4 *>(rows + i * KM_WORDS + w);
      p[w] = v.x; p[w + 1] = v.y; p[w + 2] = v.z; p[w + 3] = v.w;
    }
    unsigned best = 0xffffffffu;
    int res = -1;
    for (int c = 0; c < k; c++) {
      const unsigned d = km_dissim(s_modes + c * KM_WORDS, p);
      if (d <= best) { best = d; res = c; }
    }
    clust[i] = res;
    dis[i] = best;
  }
}

// farthest-first: mind[i] = min(mind[i], dissim(centre, row i)) for unused points, then this block's candidate for the next pick:
// the LAST index among the unused points with the largest mind (kmodes.pas:757-762) -> key = mind << 32 | index, maximum
__global__ __launch_bounds__(256) void k_kmodes_ff(const uint32_t *__restrict__ rows, int64_t n, int64_t centre, const uint8_t *__restrict__ used,
                                                   unsigned *__restrict__ mind, u64 *__restrict__ partial) {
  __shared__ u64 s_best[4];
  uint32_t c[KM_WORDS];
#pragma unroll
  for (int w = 0; w < KM_WORDS; w++) c[w] = rows[centre * KM_WORDS + w];
  u64 best = 0;
  bool any = false;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (used[i] || i == centre) continue;
    const unsigned d = km_dissim(rows + i * KM_WORDS, c);
    unsigned m = mind[i];
    if (d < m) { m = d; mind[i] = m; }
    const u64 key = ((u64)m << 32) | (u64)(uint32_t)i;
    if (!any || key >= best) { best = key; any = true; }
  }
  u64 enc = any ? best + 1 : 0;  // 0 = no candidate; keys shifted by one so that (mind 0, index 0) is still a candidate
  for (int o = 32; o > 0; o >>= 1) { const u64 other = __shfl_xor(enc, o); enc = other > enc ? other : enc; }
  if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = enc;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) enc = s_best[w] > enc ? s_best[w] : enc;
    partial[blockIdx.x] = enc;
  }
}

__host__ __device__ inline uint32_t rand_int(uint32_t range, uint32_t &seed) {  // kmodes.pas:88-92
  seed = (uint32_t)((int32_t)(seed * 0x08088405u) + 1);
  return (uint32_t)(((uint64_t)seed * (uint64_t)range) >> 32);
}

// ---- the clustering's state lives in HBM: memberships, member counts, the per-(cluster, attribute) histograms of
// MovePointCat (k x 80 x num_modalities counters), the modes, the LCG's seed, the iteration's cost and move counters
struct KmState {
  int32_t *memb, *clust;      // [n]
  unsigned *dis;              // [n]
  int32_t *members;           // [k]
  int32_t *freq;              // [k][80][nmod]
  uint8_t *cent;              // [k][80]
  uint32_t *seed;             // [1]
  unsigned long long *cost;   // [1] of the running iteration
  unsigned *moves;            // [1]
  int3 *mlist;                // the bin's moves in order: (point, to, from), at most 2 x 960
  unsigned *nmoves;           // [1]
  int *bad;                   // [1] set when something that cannot happen did (no donor for a repair)
};

__global__ void k_kmodes_validate(const uint8_t *__restrict__ rows, int64_t nbytes, int nmod, int *__restrict__ bad) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nbytes; i += (int64_t)gridDim.x * blockDim.x)
    if (rows[i] >= nmod) atomicExch(bad, 1);
}

__global__ void k_kmodes_take_rows(const uint8_t *__restrict__ rows, const int64_t *__restrict__ which, int k, uint8_t *__restrict__ cent) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < k * KM_ATTRS; e += gridDim.x * blockDim.x) cent[e] = rows[which[e / KM_ATTRS] * KM_ATTRS + e % KM_ATTRS];
}

// initial membership = the first scores; member counts and histograms by integer atomics (order-free) -- kmodes.pas:978-996
__global__ void k_kmodes_init_tables(const uint8_t *__restrict__ rows, int64_t n, int nmod, KmState st) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * KM_ATTRS; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / KM_ATTRS;
    const int a = (int)(e - i * KM_ATTRS), c = st.clust[i];
    if (a == 0) { st.memb[i] = c; atomicAdd(&st.members[c], 1); }
    atomicAdd(&st.freq[((int64_t)c * KM_ATTRS + a) * nmod + rows[e]], 1);
  }
}

// the first largest counter of a histogram row held four entries per lane (entry = lane + 64 slot): GetMaxValueIndex, kmodes.pas:155-167
__device__ __forceinline__ int km_first_largest(const int (&t)[4], int nmod, int lane) {
  int bv = INT_MIN, bi = 0x7fffffff;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int idx = lane + 64 * s;
    if (idx < nmod && t[s] > bv) { bv = t[s]; bi = idx; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int ov = __shfl_xor(bv, o), oi = __shfl_xor(bi, o);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  return bi;
}

// the modes after the initial assignment (997-1011): an empty cluster takes, attribute by attribute, the value of a row drawn with
// RandInt -- clusters in order, attributes in order, one draw each: a single thread walks the LCG; the others' modes come from their histograms
__global__ __launch_bounds__(64) void k_kmodes_init_modes(const uint8_t *__restrict__ rows, int64_t n, int k, int nmod, KmState st) {
  const int lane = threadIdx.x;
  if (blockIdx.x == 0) {
    if (lane == 0) {
      uint32_t seed = *st.seed;
      for (int c = 0; c < k; c++)
        if (st.members[c] == 0)
          for (int a = 0; a < KM_ATTRS; a++) st.cent[c * KM_ATTRS + a] = rows[(int64_t)rand_int((uint32_t)n, seed) * KM_ATTRS + a];
      *st.seed = seed;
    }
    return;
  }
  for (int pr = blockIdx.x - 1; pr < k * KM_ATTRS; pr += gridDim.x - 1) {
    const int c = pr / KM_ATTRS;
    if (st.members[c] == 0) continue;
    int t[4];
#pragma unroll
    for (int sl = 0; sl < 4; sl++) t[sl] = lane + 64 * sl < nmod ? st.freq[(int64_t)pr * nmod + lane + 64 * sl] : 0;
    const int m = km_first_largest(t, nmod, lane);
    if (lane == 0) st.cent[pr] = (uint8_t)m;
  }
}

// ---- KModesIter (851-921), two launches per bin.
// The reference walks bins of 960 points: a bin is scored against the modes as they stand at its start, then its points move one after
// the other, every move updating two clusters' histograms and modes (MovePointCat, 774-803).  Rounds 2-3 ran that as three dependent
// launches per bin (score, walk, apply: 56 ms per iteration at config 5's 1.6 M rows, all of it launch turnaround; 760 ms for the first
// iteration, in whose bins nearly every point moves).  Now (25.7 ms per iteration over iterations 2-5, 115 ms for init + the first):
//   * k_kmodes_owner, G workgroups: owner w holds the clusters c = w (mod G) -- nobody else touches their histograms and modes.  It
//     applies the PREVIOUS bin's moves that enter or leave its clusters (in order: Huang's online update), then scores THIS bin's points
//     against its own modes only and leaves, per point, its own best (distance, cluster) as one key;
//   * k_kmodes_walker, one workgroup: the minimum over the owners' keys of every point of the bin (the LAST minimum wins, kmodes.pas:272) --
//     the reference's score of the bin at its start, since every owner applied every earlier move before it scored --, then the bin's
//     points in order (a point whose score names another cluster moves; a move that empties a cluster is followed by the repair of
//     879-897), the bin's move list for the next owner launch.
// (Built and measured first as ONE resident launch per iteration, owners and walker meeting through flags in memory: the same work, 70 ms --
// on eight XCDs with an L2 each every hand-over between workgroups is an agent-scope release / acquire and several microseconds, and a
// bin has four of them in a row; two kernel boundaries cost less.  The iteration's 2 x bins + 1 launches are a graph, replayed.)
// MovePointCat's histogram side for one (cluster, attribute) pair: the wave holds the pair's row of counters (four per lane) and its mode and
// applies, in order, the moves flagged in `m` (lane = move: `cur_l` its point's value of the attribute, `in` = it enters the cluster) -- the
// counter of the value up and the mode following it when it is overtaken; the counter down and, when the mode itself lost a member, the
// first largest counter as the new mode (kmodes.pas:774-803).
__device__ __forceinline__ void km_apply_moves(int (&t)[4], int &mode, unsigned long long m, int cur_l, bool in, int nmod, int lane) {
  while (m) {
    const int bit = __builtin_ctzll(m);
    m &= m - 1;
    const int cur = __builtin_amdgcn_readlane(cur_l, bit);
    const bool enters = __builtin_amdgcn_readlane((int)in, bit) != 0;
    const int own = cur & 63, sl = cur >> 6;
    const int d = enters ? 1 : -1;
    if (lane == own) { t[0] += sl == 0 ? d : 0; t[1] += sl == 1 ? d : 0; t[2] += sl == 2 ? d : 0; t[3] += sl == 3 ? d : 0; }
    if (enters) {
      const int msl = mode >> 6;
      const int vm = __builtin_amdgcn_readlane(msl == 0 ? t[0] : msl == 1 ? t[1] : msl == 2 ? t[2] : t[3], mode & 63);
      const int vc = __builtin_amdgcn_readlane(sl == 0 ? t[0] : sl == 1 ? t[1] : sl == 2 ? t[2] : t[3], own);
      if (vm < vc) mode = cur;
    } else if (mode == cur) {
      mode = km_first_largest(t, nmod, lane);
    }
  }
}

#ifndef TM_KM_OWNERS
#define TM_KM_OWNERS 64
#endif
constexpr int KM_BIN = 960, KM_WT = 1024, KM_MAX_OWNERS = TM_KM_OWNERS, KM_STAGE = 768;

// bin < 0: only the moves of the last bin are applied (the iteration's closing launch); first = 1: there are no moves to apply yet
// (the body of k_kmodes_owner; `w` = the owner's number, `s_dyn` = its dynamic LDS.  Returns, in every thread, whether a word of an owned mode
// changed -- what k_kmodes_fast stops on.)
__device__ __forceinline__ bool km_owner_body(const uint8_t *__restrict__ rows, int64_t n, int k, int nmod, int G, int w, int64_t bin, int first, KmState st,
                                              unsigned *__restrict__ partial /* [G][KM_BIN] keys: distance << 12 | 4095 - cluster */, int *s_dyn) {
  // s_dyn: the owned clusters' modes as words [owned][20], their touched flags, the previous bin's relevant moves, those moves' rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bool changed = false;
  const int nown = (k - w + G - 1) / G;  // clusters w, w + G, w + 2 G, ...
  const int max_own = (k + G - 1) / G;
  uint32_t *const s_mode = reinterpret_cast<uint32_t *>(s_dyn);                 // [nown][20]
  int *const s_touch = s_dyn + max_own * KM_WORDS;                              // [nown]: a move of the previous bin enters or leaves the cluster
  int3 *const s_rel = reinterpret_cast<int3 *>(s_touch + max_own);              // those moves, in order
  uint4 *const s_rows = reinterpret_cast<uint4 *>(s_dyn + ((max_own * (KM_WORDS + 1) + 3 * (2 * KM_BIN + 8) + 3) & ~3));  // [KM_STAGE][5]: their points
  __shared__ unsigned s_nrel;
  __shared__ int s_hist[KM_WT / 64][256];
  // (what the launch needs from memory and can name at once -- the count of pending moves, the bin's own points -- is asked for here,
  // with the modes: a round trip each that the launch, which is all latency, does not wait for later)
  const unsigned nmv = first ? 0u : *st.nmoves;
  const int64_t b0 = max((int64_t)0, bin) * KM_BIN, b1 = min(b0 + (int64_t)KM_BIN, n);
  const int nb = bin < 0 ? 0 : (int)(b1 - b0);
  uint32_t p[KM_WORDS];
  if (tid < nb) {
#pragma unroll
    for (int q = 0; q < KM_WORDS; q += 4) {
      const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint32_t *>(rows) + (b0 + tid) * KM_WORDS + q);
      p[q] = v.x; p[q + 1] = v.y; p[q + 2] = v.z; p[q + 3] = v.w;
    }
  }
  for (int e = tid; e < KM_WT / 64 * 256; e += KM_WT) (&s_hist[0][0])[e] = 0;
  for (int e = tid; e < nown * KM_WORDS; e += KM_WT) s_mode[e] = reinterpret_cast<const uint32_t *>(st.cent)[(size_t)(w + (e / KM_WORDS) * G) * KM_WORDS + e % KM_WORDS];
  for (int e = tid; e < nown; e += KM_WT) s_touch[e] = 0;
  if (tid == 0) s_nrel = 0;
  __syncthreads();
  if (nmv > 0) {
    // ---- MovePointCat's histogram side (774-803) for the owned clusters, a wave per (cluster, attribute) pair
    if (wave == 0) {  // ordered compaction of the relevant moves
      unsigned nrel = 0;
      for (unsigned e0 = 0; e0 < nmv; e0 += 64) {
        const unsigned e = e0 + lane;
        int3 mv = make_int3(0, -1, -1);
        if (e < nmv) mv = st.mlist[e];
        const bool rin = mv.y >= 0 && mv.y % G == w, rout = mv.z >= 0 && mv.z % G == w;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(rin || rout);
        if (rin || rout) s_rel[nrel + __popcll(m & ((1ull << lane) - 1ull))] = mv;
        if (rin) s_touch[mv.y / G] = 1;
        if (rout) s_touch[mv.z / G] = 1;
        nrel += __popcll(m);
      }
      if (lane == 0) s_nrel = nrel;
    }
    __syncthreads();
    const unsigned nrel = s_nrel;
    // The moved points' rows come into LDS first, KM_STAGE moves at a time (in a bin of the first iteration nearly every point moves: a
    // (pair, 64 moves) step that fetched its values from memory itself waited a round trip each, 240 microseconds per bin)
    for (unsigned r0 = 0; r0 < nrel; r0 += KM_STAGE) {
      const unsigned nr = min((unsigned)KM_STAGE, nrel - r0);
      if (r0) __syncthreads();
      for (unsigned e = tid; e < nr * 5; e += KM_WT) s_rows[e] = reinterpret_cast<const uint4 *>(rows)[(int64_t)s_rel[r0 + e / 5].x * 5 + e % 5];
      __syncthreads();
      // a wave's pairs in batches of KM_PB: the batch's histogram rows are all asked for before any is used
      constexpr int KM_PB = 5;
      for (int pr0 = wave; pr0 < nown * KM_ATTRS; pr0 += KM_PB * (KM_WT / 64)) {
        int t[KM_PB][4];
        bool act[KM_PB];
#pragma unroll
        for (int i = 0; i < KM_PB; i++) {
          const int pr = pr0 + i * (KM_WT / 64);
          act[i] = pr < nown * KM_ATTRS && s_touch[pr / KM_ATTRS] != 0;
          if (act[i]) {
            const int64_t gp = (int64_t)(w + (pr / KM_ATTRS) * G) * KM_ATTRS + pr % KM_ATTRS;
#pragma unroll
            for (int sl = 0; sl < 4; sl++) t[i][sl] = lane + 64 * sl < nmod ? st.freq[gp * nmod + lane + 64 * sl] : 0;
          }
        }
#pragma unroll
        for (int i = 0; i < KM_PB; i++) {
          if (!act[i]) continue;
          const int pr = pr0 + i * (KM_WT / 64);
          const int oc = pr / KM_ATTRS, a = pr - oc * KM_ATTRS, c = w + oc * G;
          const int64_t gp = (int64_t)c * KM_ATTRS + a;
          int mode = (int)reinterpret_cast<const uint8_t *>(s_mode + oc * KM_WORDS)[a];
          // The counters first, all the pair's moves at once (LDS atomics on the wave's own row of 256).  Huang's mode is at every moment A
          // largest counter -- an arrival takes the mode over only by strictly passing it, a departure of the mode's value re-reads the
          // first largest -- so when the counters after the last move have ONE largest, that is the mode whatever the order was.  Only a
          // tie at the end needs the moves replayed in order (a hot cluster takes nearly all 960 moves of a first-iteration bin: replayed
          // one by one for each of its 80 attributes that was 240 microseconds per bin).
          int *const h = &s_hist[wave][0];
          bool any = false;
          for (unsigned e0 = 0; e0 < nr; e0 += 64) {
            const unsigned e = e0 + lane;
            int3 mv = make_int3(0, -1, -1);
            if (e < nr) mv = s_rel[r0 + e];
            const bool in = mv.y == c, out = mv.z == c;
            if (in || out) atomicAdd(&h[reinterpret_cast<const uint8_t *>(s_rows)[e * KM_ATTRS + a]], in ? 1 : -1);
            any = any || __builtin_amdgcn_ballot_w64(in || out) != 0;
          }
          if (!any) continue;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          int nt[4];
#pragma unroll
          for (int sl = 0; sl < 4; sl++) { nt[sl] = t[i][sl] + h[lane + 64 * sl]; h[lane + 64 * sl] = 0; }
          int mx = -1, at = 0;
#pragma unroll
          for (int sl = 0; sl < 4; sl++) if (lane + 64 * sl < nmod && nt[sl] > mx) { mx = nt[sl]; at = lane + 64 * sl; }
          int wmx = mx;
          for (int o = 32; o > 0; o >>= 1) wmx = max(wmx, __shfl_xor(wmx, o));
          int ties = 0;
#pragma unroll
          for (int sl = 0; sl < 4; sl++) ties += (lane + 64 * sl < nmod && nt[sl] == wmx) ? 1 : 0;
          const unsigned long long holders = __builtin_amdgcn_ballot_w64(ties > 0);
          const bool one = __popcll(holders) == 1 && __builtin_amdgcn_readlane(ties, __builtin_ctzll(holders)) == 1;
          if (one) {
            mode = __builtin_amdgcn_readlane(at, __builtin_ctzll(holders));
#pragma unroll
            for (int sl = 0; sl < 4; sl++) t[i][sl] = nt[sl];
          } else {
            for (unsigned e0 = 0; e0 < nr; e0 += 64) {
              const unsigned e = e0 + lane;
              int3 mv = make_int3(0, -1, -1);
              if (e < nr) mv = s_rel[r0 + e];
              const bool in = mv.y == c, out = mv.z == c;
              const unsigned long long m = __builtin_amdgcn_ballot_w64(in || out);
              if (!m) continue;
              const int cur_l = (in || out) ? (int)reinterpret_cast<const uint8_t *>(s_rows)[e * KM_ATTRS + a] : 0;
              km_apply_moves(t[i], mode, m, cur_l, in, nmod, lane);
            }
          }
#pragma unroll
          for (int sl = 0; sl < 4; sl++) if (lane + 64 * sl < nmod) st.freq[gp * nmod + lane + 64 * sl] = t[i][sl];
          if (lane == 0) reinterpret_cast<uint8_t *>(s_mode + oc * KM_WORDS)[a] = (uint8_t)mode;
        }
      }
    }
    if (nrel > 0) {
      __syncthreads();
      // the owned modes go back where the next launches (and the host) read them
      for (int e = tid; e < nown * KM_WORDS; e += KM_WT) {
        uint32_t *dst = reinterpret_cast<uint32_t *>(st.cent) + (size_t)(w + (e / KM_WORDS) * G) * KM_WORDS + e % KM_WORDS;
        if (*dst != s_mode[e]) {
          changed = true;
#ifdef TM_KM_COUNT_MODE_CHANGES
          atomicAdd(st.seed + 1, 1u);  // (diagnostic build: words of modes that changed, in the scalars' padding)
#endif
        }
        *dst = s_mode[e];
      }
      changed = __syncthreads_or(changed ? 1 : 0) != 0;
    }
  }
  if (bin < 0) return changed;
  // ---- the bin's points against the owned modes as they stand now
  if (tid < nb) {
    // the owner's own arg-min: a distance has 18 bits (80 * (255 + 2048)), a cluster 12; among equal distances the LAST cluster wins
    unsigned key = 0xffffffffu;
    for (int oc = 0; oc < nown; oc++) key = min(key, (km_dissim(s_mode + oc * KM_WORDS, p) << 12) | (unsigned)(4095 - (w + oc * G)));
    partial[(size_t)w * KM_BIN + tid] = key;
  }
  return changed;
}
// (`start`: the bin this iteration's bin-by-bin work begins at -- 0, or where the fast leg handed over: the graph's launches for the bins before
// it return at once, and the first one behind it has no moves to apply)
__global__ __launch_bounds__(KM_WT) void k_kmodes_owner(const uint8_t *__restrict__ rows, int64_t n, int k, int nmod, int G, int64_t bin, int first, KmState st,
                                                        unsigned *__restrict__ partial, const long long *__restrict__ start) {
  extern __shared__ int s_dyn[];
  const long long s0 = *start;
  if (bin >= 0 && bin < s0) return;
  (void)km_owner_body(rows, n, k, nmod, G, (int)blockIdx.x, bin, (first || bin == s0) ? 1 : 0, st, partial, s_dyn);
}

// what a launch that walks many bins keeps on chip between them (k_kmodes_fast): the member counts stay in s_members, these in LDS
struct KmFastAcc { unsigned long long cost; unsigned moves; uint32_t seed; };
// (the body of k_kmodes_walker; `s_members` = k words of LDS.  acc != nullptr (k_kmodes_fast): the bin's scores are st.clust / st.dis as a
// scoring launch over many bins left them, the member counts are in s_members already and cost, move count and seed are acc's: a bin then
// costs one round trip to memory, for its points.  Returns the number of moves the bin made, in every thread.)
__device__ __forceinline__ unsigned km_walker_body(int64_t n, int k, int G, int64_t bin, KmState st, const unsigned *__restrict__ partial, int *s_members, KmFastAcc *acc) {
  __shared__ int s_mb[KM_BIN], s_cl[KM_BIN];
  __shared__ unsigned long long s_cost[KM_WT / 64];
  __shared__ int s_cnt[KM_WT], s_wtot[KM_WT / 64];
  __shared__ int s_state, s_from, s_to;   // 0 done, 1 a repair is wanted: cluster s_to is empty, donor s_from
  __shared__ unsigned s_pick;
  __shared__ long long s_found;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b0 = bin * KM_BIN, b1 = min(b0 + (int64_t)KM_BIN, n);
  const int nb = (int)(b1 - b0);
  if (!acc) for (int c = tid; c < k; c += KM_WT) s_members[c] = st.members[c];
  // the bin's scores: the minimum over the owners' keys (smallest distance, then the LAST cluster: kmodes.pas:272, 414).  One workgroup
  // fetches all of them and a CU has only so many misses in flight: with a word per (point, cluster) -- 245 KB at 64 clusters -- this
  // fetch alone was 12 of the walker's 16 microseconds, so the owners are few (KM_MAX_OWNERS) and reduce their clusters themselves
  // (all of the point's keys, its membership and the generator's seed in flight together: the launch is a chain of round trips to memory,
  // and four batches of sixteen keys were four of them)
  unsigned long long cost = 0;
  const uint32_t seed0 = acc ? acc->seed : *st.seed;
  if (tid < nb && acc) {
    s_cl[tid] = st.clust[b0 + tid];
    s_mb[tid] = st.memb[b0 + tid];
    cost = st.dis[b0 + tid];
  } else if (tid < nb) {
    static_assert(KM_MAX_OWNERS <= 64, "the walker holds one key per owner");
    const int mb = st.memb[b0 + tid];
    unsigned key = 0xffffffffu;
#ifndef TM_KM_WALKER_BATCH
#define TM_KM_WALKER_BATCH 64
#endif
    for (int g0 = 0; g0 < G; g0 += TM_KM_WALKER_BATCH) {
      unsigned d[TM_KM_WALKER_BATCH];
#pragma unroll
      for (int u = 0; u < TM_KM_WALKER_BATCH; u++) d[u] = g0 + u < G ? partial[(size_t)(g0 + u) * KM_BIN + tid] : 0xffffffffu;
#pragma unroll
      for (int u = 0; u < TM_KM_WALKER_BATCH; u++) key = min(key, d[u]);
    }
    const unsigned best = key >> 12;
    const int res = 4095 - (int)(key & 4095u);
    s_cl[tid] = res;
    s_mb[tid] = mb;
    cost = best;
  }
  for (int o = 32; o > 0; o >>= 1) cost += __shfl_xor(cost, o);
  if (lane == 0) s_cost[wave] = cost;
  __syncthreads();
  unsigned nmv = 0, moves = 0;
  uint32_t seed = seed0;
  int chunk = 0;                 // wave 0's place in the bin: 64 points at a time
  unsigned long long todo = 0;   // the chunk's points still to look at
  bool fresh = true;
  const int64_t per = (n + KM_WT - 1) / KM_WT;
  for (;;) {
    if (wave == 0) {  // (every lane of the wave runs the walk with the same values: the list is written by lane 0)
      int state = 0;
      while (chunk * 64 < nb) {
        const int j0 = chunk * 64;
        if (fresh) {
          todo = j0 + 64 <= nb ? ~0ull : ((1ull << (nb - j0)) - 1ull);
          fresh = false;
          // The chunk's moves at once when none of them can empty a cluster: every leaver takes its member away first -- if no count
          // reaches zero with the departures alone, none does in the points' order with the arrivals in between either, there is no repair and
          // the outcome (memberships, counts, the list in the points' order) is the walk's.  Otherwise the departures are put back
          // and the chunk is walked a point at a time.
          const bool mv1 = j0 + lane < nb && s_mb[j0 + lane] != s_cl[j0 + lane];
          const unsigned long long all = __builtin_amdgcn_ballot_w64(mv1);
          if (!all) { chunk++; fresh = true; continue; }
          const int old1 = mv1 ? s_mb[j0 + lane] : 0, to1 = mv1 ? s_cl[j0 + lane] : 0;
          const int pre = mv1 ? atomicSub(&s_members[old1], 1) : 2;
          if (!__builtin_amdgcn_ballot_w64(mv1 && pre <= 1)) {
            if (mv1) {
              atomicAdd(&s_members[to1], 1);
              st.memb[b0 + j0 + lane] = to1;
              s_mb[j0 + lane] = to1;
              st.mlist[nmv + __popcll(all & ((1ull << lane) - 1ull))] = make_int3((int)(b0 + j0 + lane), to1, old1);
            }
            nmv += __popcll(all);
            moves += __popcll(all);
            chunk++;
            fresh = true;
            continue;
          }
          if (mv1) atomicAdd(&s_members[old1], 1);
        }
        const bool mine = j0 + lane < nb && ((todo >> lane) & 1ull) && s_mb[j0 + lane] != s_cl[j0 + lane];
        const unsigned long long mv = __builtin_amdgcn_ballot_w64(mine);
        if (!mv) { chunk++; fresh = true; continue; }
        const int bit = __builtin_ctzll(mv), j = j0 + bit;
        todo &= bit == 63 ? 0ull : (~0ull << (bit + 1));  // everything up to and including j has been looked at
        const int to = s_cl[j], old = s_mb[j];
        if (lane == 0) {
          st.memb[b0 + j] = to;
          s_mb[j] = to;
          st.mlist[nmv] = make_int3((int)(b0 + j), to, old);
          s_members[to]++;
          s_members[old]--;
        }
        nmv++;
        moves++;
        const int left = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_members[old], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (left == 0) {  // CountClusterMembers(old_clust) = 0: the donor is the LAST largest cluster (667), its RandInt(size)-th member
          int from = 0, mc = 0;
          for (int c0 = 0; c0 < k; c0 += 64) {
            const int c = c0 + lane;
            const int v = c < k ? __hip_atomic_load(&s_members[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : -1;
            int bv = v, bc = c;
            for (int o = 32; o > 0; o >>= 1) {
              const int ov = __shfl_xor(bv, o), oc = __shfl_xor(bc, o);
              if (ov > bv || (ov == bv && oc > bc)) { bv = ov; bc = oc; }
            }
            if (bv >= mc) { mc = bv; from = bc; }
          }
          const uint32_t pick = rand_int((uint32_t)mc, seed);
          if (lane == 0) { s_from = from; s_to = old; s_pick = pick; }
          state = 1;
          break;
        }
      }
      if (lane == 0) s_state = state;
      if (state) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // a repair: the memberships written so far are what the search below reads (a write-back of the L2 -- microseconds: only then)
    }
    __syncthreads();
    if (s_state == 0) break;
    // ---- all threads: the s_pick-th point (in index order) whose membership is s_from
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int from = s_from;
    const int64_t lo = min((int64_t)tid * per, n), hi = min(lo + per, n);
    int cnt = 0;
    for (int64_t i = lo; i < hi; i++) cnt += __hip_atomic_load(&st.memb[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == from ? 1 : 0;
    int inc = cnt;  // inclusive prefix inside the wave
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    s_cnt[tid] = inc - cnt;
    if (lane == 63) s_wtot[wave] = inc;
    if (tid == 0) s_found = -1;
    __syncthreads();
    int before = s_cnt[tid];
    for (int w2 = 0; w2 < wave; w2++) before += s_wtot[w2];
    const int pick = (int)s_pick;
    if (pick >= before && pick < before + cnt) {
      int seen = before;
      for (int64_t i = lo; i < hi; i++)
        if (__hip_atomic_load(&st.memb[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == from) {
          if (seen == pick) { s_found = i; break; }
          seen++;
        }
    }
    __syncthreads();
    const long long r = s_found;
    if (r < 0) { if (tid == 0) *st.bad = 2; break; }
    if (wave == 0) {  // the repair's own move: r leaves s_from for the emptied cluster
      const int to = s_to;
      if (lane == 0) {
        st.memb[r] = to;
        st.mlist[nmv] = make_int3((int)r, to, from);
        s_members[to]++;
        s_members[from]--;
        if (r >= b0 && r < b1) s_mb[r - b0] = to;  // a point of this very bin: looked at again with its new membership when its turn comes
      }
      nmv++;
    }
    __syncthreads();
  }
  __syncthreads();
  if (!acc) for (int c = tid; c < k; c += KM_WT) st.members[c] = s_members[c];
  if (tid == 0) {
    unsigned long long c = 0;
    for (int w2 = 0; w2 < KM_WT / 64; w2++) c += s_cost[w2];
    if (acc) {
      acc->cost += c; acc->moves += moves; acc->seed = seed;
    } else {
      *st.cost += c;
      *st.moves += moves;
      *st.nmoves = nmv;
      *st.seed = seed;
    }
  }
  __shared__ unsigned s_nmv_out;
  if (tid == 0) s_nmv_out = nmv;
  __syncthreads();
  return s_nmv_out;
}
__global__ __launch_bounds__(KM_WT) void k_kmodes_walker(int64_t n, int k, int G, int64_t bin, KmState st, const unsigned *__restrict__ partial,
                                                         const long long *__restrict__ start) {
  extern __shared__ int s_members_dyn[];  // [k]
  if (bin < *start) return;
  (void)km_walker_body(n, k, G, bin, st, partial, s_members_dyn, nullptr);
}

// ---- the fast leg of an iteration (round 5): as long as no MODE changes, the scores of all remaining points are known in advance.
// A bin's scores depend on the modes alone, and late in a clustering the modes hardly ever move (at config 5's shape: 23 116 moves and at most
// 128 changed mode words in the second iteration, 1 825 moves and none in the third, none at all in the fourth) -- yet every bin paid two
// dependent launches, 13 microseconds, for a scoring whose outcome was known.  Here ONE scoring launch (k_kmodes_argmin over all remaining
// points, every CU) leaves clust / dis, and ONE workgroup then walks bin after bin in a single launch: the walker's body on those scores, then the
// owners' body as the one owner of every cluster (MovePointCat's histograms and modes for the bin's moves, at once rather than at the next
// bin's start: nothing reads them in between).  The first bin whose moves change a word of a mode ends the launch -- the scores of the bins
// behind it are no longer the reference's -- and leaves its number; the host scores again from there, or, after KM_FAST_STOPS such stops in
// an iteration, finishes it bin by bin with the two launches.  The bins walked this way are exactly the reference's: same scores (same modes),
// same walk, same updates in the same order.
constexpr int KM_FAST_STOPS = 4;
// MovePointCat (774-803) for `nmv` moves (point, to, from) of one bin, one after the other as `list` has them: a thread per (side, attribute)
// of a move -- its point leaves `from` and enters `to`, two different clusters, 80 independent (cluster, attribute) pairs each.  An arrival
// takes the pair's mode over only by strictly passing it; a departure of the mode's own value re-reads the first largest counter.  A bin's few
// moves nearly always touch different clusters: then they are independent of each other too, and up to KM_FAST_PAR of them go at once (a
// thread per (move, side, attribute): one chain of round trips to memory for the bin instead of one per move).  Sets *changed when a mode
// changes.  (The owners' batched form of this, made for bins in which hundreds of points move, is a dozen barriers and round trips whatever
// the count: 70 us a bin where a late bin has one or two moves.)
constexpr unsigned KM_FAST_PAR = KM_WT / (2 * KM_ATTRS);
__device__ __forceinline__ void km_light_apply(const uint8_t *__restrict__ rows, int nmod, KmState st, const int3 *list, unsigned nmv, int *changed) {
  const int tid = threadIdx.x;
  bool par = nmv <= KM_FAST_PAR;
  if (par && nmv > 1) {
    int3 mvs[KM_FAST_PAR];
#pragma unroll
    for (unsigned m = 0; m < KM_FAST_PAR; m++) mvs[m] = m < nmv ? list[m] : make_int3(0, -1 - 2 * (int)m, -2 - 2 * (int)m);
#pragma unroll
    for (unsigned x = 0; x < KM_FAST_PAR; x++)
#pragma unroll
      for (unsigned y = x + 1; y < KM_FAST_PAR; y++)
        if (mvs[x].y == mvs[y].y || mvs[x].y == mvs[y].z || mvs[x].z == mvs[y].y || mvs[x].z == mvs[y].z) par = false;
  }
  const unsigned steps = par ? 1u : nmv;
  for (unsigned step = 0; step < steps; step++) {
    const unsigned m = par ? (unsigned)tid / (2 * KM_ATTRS) : step;
    const int lt = par ? tid % (2 * KM_ATTRS) : tid;
    if (m < nmv && (par || tid < 2 * KM_ATTRS)) {
      const int3 mv = list[m];
      const bool enters = lt >= KM_ATTRS;
      const int a = enters ? lt - KM_ATTRS : lt, c = enters ? mv.y : mv.z;
      if (c >= 0) {
        // two round trips to memory: the point's value and the pair's mode; then the counters (an arrival's two, all of a departure's row --
        // four in five values of a leaving point are their pair's mode, whose departure asks for the first largest)
        int *const t = st.freq + ((int64_t)c * KM_ATTRS + a) * nmod;
        uint8_t *const mp = st.cent + (int64_t)c * KM_ATTRS + a;
        const int v = rows[(int64_t)mv.x * KM_ATTRS + a];
        const int mode = *mp;
        int nm = mode;
        if (enters) {
          const int tv0 = t[v], tm0 = t[mode];
          const int tv = tv0 + 1;
          t[v] = tv;
          if ((mode == v ? tv : tm0) < tv) nm = v;
        } else {
          int bv = INT_MIN, bi = mode;  // GetMaxValueIndex (155-167) over the row as it is AFTER the departure: the first largest
          for (int i = 0; i < nmod; i++) { const int x = t[i] - (i == v ? 1 : 0); if (x > bv) { bv = x; bi = i; } }
          t[v] -= 1;  // (the row's line is in the cache)
          if (mode == v) nm = bi;
        }
        if (nm != mode) { *mp = (uint8_t)nm; *changed = 1; }
      }
    }
    __syncthreads();  // (the next move of the list may meet the same pair)
  }
}

// The walk looks KM_SB bins ahead: one pass over their points (scores, memberships, distances: one round trip for 15 360 points) gives every
// bin's cost and the points that want to move, in order.  A bin without movers is done with its cost.  A bin whose movers cannot empty a
// cluster (every cluster keeps a member whatever the order: the departures alone leave it one) needs no walk either: memberships and counts
// change by what the movers say, MovePointCat goes over them in their order.  Only a bin in which a cluster might run empty -- the repair
// of 879-897 draws from the generator and moves a point from anywhere -- takes the walker's body; the look-ahead starts again behind it.
constexpr int KM_SB = 16, KM_SB_MOVERS = 128;
__global__ __launch_bounds__(KM_WT) void k_kmodes_fast(const uint8_t *__restrict__ rows, int64_t n, int k, int nmod, int64_t bin_begin, int64_t nbins, KmState st,
                                                       long long *__restrict__ stop /* the first bin not walked */) {
  extern __shared__ int s_members[];  // [k]
  __shared__ KmFastAcc s_acc;
  __shared__ int s_changed, s_unsafe;
  __shared__ unsigned long long s_sbcost[KM_SB];
  __shared__ unsigned s_nmover;
  __shared__ int3 s_mover[KM_SB_MOVERS], s_sorted[KM_SB_MOVERS];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int c = tid; c < k; c += KM_WT) s_members[c] = st.members[c];
  if (tid == 0) { s_acc.cost = 0; s_acc.moves = 0; s_acc.seed = *st.seed; s_changed = 0; }
  __syncthreads();
  int64_t bin = bin_begin;  // the first bin not walked yet
  while (bin < nbins && !s_changed) {
    const int nsb = (int)min((int64_t)KM_SB, nbins - bin);
    const int64_t p0 = bin * KM_BIN, p1 = min(n, (bin + nsb) * KM_BIN);
    if (tid < KM_SB) s_sbcost[tid] = 0;
    if (tid == 0) { s_nmover = 0; s_unsafe = 0; }
    __syncthreads();
    for (int64_t base = p0; base < p1; base += KM_WT) {
      const int64_t p = base + tid;
      const bool in = p < p1;
      const int cl = in ? st.clust[p] : 0, mb = in ? st.memb[p] : 0;
      const unsigned long long d = in ? st.dis[p] : 0;
      const int b = in ? (int)((p - p0) / KM_BIN) : -1;
      // a wave's 64 consecutive points lie in one bin or two
      const int bA = __builtin_amdgcn_readfirstlane(b);
      int bB = -1;
      {
        const unsigned long long other = __builtin_amdgcn_ballot_w64(b >= 0 && b != bA);
        if (other) bB = __builtin_amdgcn_readlane(b, __builtin_ctzll(other));
      }
      unsigned long long sa = b == bA ? d : 0, sb = (bB >= 0 && b == bB) ? d : 0;
      for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
      if (lane == 0) {
        if (bA >= 0 && sa) atomicAdd(&s_sbcost[bA], sa);
        if (bB >= 0 && sb) atomicAdd(&s_sbcost[bB], sb);
      }
      if (in && cl != mb) {
        const unsigned slot = atomicAdd(&s_nmover, 1u);
        if (slot < KM_SB_MOVERS) s_mover[slot] = make_int3((int)p, cl, mb);
      }
    }
    __syncthreads();
    const unsigned nm = s_nmover;
    if (nm <= KM_SB_MOVERS) {  // into the points' order
      if ((unsigned)tid < nm) {
        const int3 me = s_mover[tid];
        unsigned rank = 0;
        for (unsigned j = 0; j < nm; j++) rank += s_mover[j].x < me.x ? 1u : 0u;
        s_sorted[rank] = me;
      }
      __syncthreads();
    }
    unsigned at = 0;  // the movers of the bins before `bin` are done
    bool rescan = false;
    for (int b = 0; b < nsb && !rescan; b++, bin++) {
      unsigned mb_ = 0;
      bool heavy = nm > KM_SB_MOVERS;
      if (!heavy) {
        const int64_t pend = (bin + 1) * KM_BIN;
        while (at + mb_ < nm && s_sorted[at + mb_].x < pend) mb_++;
        if (mb_ == 0) {
          if (tid == 0) s_acc.cost += s_sbcost[b];
          continue;
        }
        // can a cluster run empty?  not if every cluster the bin's movers leave keeps a member with all of them gone
        if ((unsigned)tid < mb_) {
          const int f = s_sorted[at + tid].z;
          int leavers = 0;
          for (unsigned j = 0; j < mb_; j++) leavers += s_sorted[at + j].z == f ? 1 : 0;
          if (s_members[f] - leavers < 1) s_unsafe = 1;
        }
        __syncthreads();
        heavy = s_unsafe != 0;
      }
      if (heavy) {  // the walk proper, then its moves; what the look-ahead saw of later bins may be stale (a repair moves a point from anywhere)
        const unsigned nmv = km_walker_body(n, k, 1, bin, st, nullptr, s_members, &s_acc);
        if (nmv) km_light_apply(rows, nmod, st, st.mlist, nmv, &s_changed);
        __syncthreads();
        rescan = nm <= KM_SB_MOVERS;  // (more movers than the look-ahead holds: every bin of it takes the walk, nothing of it is used)
      } else {
        if ((unsigned)tid < mb_) {
          const int3 mv = s_sorted[at + tid];
          st.memb[mv.x] = mv.y;
          atomicSub(&s_members[mv.z], 1);
          atomicAdd(&s_members[mv.y], 1);
        }
        if (tid == 0) { s_acc.cost += s_sbcost[b]; s_acc.moves += mb_; }
        __syncthreads();
        km_light_apply(rows, nmod, st, s_sorted + at, mb_, &s_changed);
        at += mb_;
      }
      if (s_changed) { bin++; break; }
    }
    __syncthreads();
  }
  __syncthreads();
  for (int c = tid; c < k; c += KM_WT) st.members[c] = s_members[c];
  if (tid == 0) {
    *st.cost += s_acc.cost;
    *st.moves += s_acc.moves;
    *st.seed = s_acc.seed;
    *st.nmoves = 0;  // (every move is applied)
    *stop = bin;
  }
}

}  // namespace

// TKModes.ComputeKModes on DEVICE pointers: rows [n][80], labels [n], centroids [k][80]; the host keeps the stopping rule
// (1040-1049) and the run bookkeeping, one small read-back per iteration.
int run_kmodes_dev(const uint8_t *rows, int64_t n, int k, int num_init, int nmod, int max_iter, int32_t *labels_out, uint8_t *cent_out, uint64_t *cost_out,
                   int *iters_out, int64_t *point_iters_out, hipStream_t stream) {
  TM_TRY(require_device());
  TM_CHECK(rows && labels_out && cent_out, TM_E_INVAL, "kmodes: null argument");
  TM_CHECK(n >= 1 && n < (1ll << 31), TM_E_INVAL, "kmodes: %lld points", (long long)n);
  TM_CHECK(k >= 1 && k <= 4096, TM_E_INVAL, "kmodes: %d clusters outside 1..4096", k);
  TM_CHECK(nmod >= 1 && nmod <= 256, TM_E_INVAL, "kmodes: %d modalities outside 1..256", nmod);
  if (max_iter < 0) max_iter = INT_MAX;
  const int nruns = num_init <= 0 ? 1 : num_init;
  std::vector<int64_t> starts((size_t)nruns);
  if (num_init <= 0) {
    starts[0] = -(int64_t)num_init;
  } else {  // 952-964: golden-ratio-like spread of the starting points, Single arithmetic
    const float ratio = (float)std::pow((double)n, 1.0 / (double)num_init);
    float acc = 1.0f;
    for (int i = 0; i < nruns; i++) {
      starts[(size_t)i] = (int64_t)std::nearbyint((double)acc) - 1;  // Round: half to even
      if (i > 0 && starts[(size_t)i] <= starts[(size_t)i - 1]) starts[(size_t)i] = std::min(n - 1, starts[(size_t)i - 1] + 1);
      acc = acc * ratio;
    }
  }
  for (int64_t sp : starts) TM_CHECK(sp >= 0 && sp < n, TM_E_INVAL, "kmodes: starting point %lld outside the %lld points", (long long)sp, (long long)n);
  DevBuf dclust, ddis, dmemb, dmembers, dfreq, dcent, dused, dmind, dpartial, dwhich, dscal, dmlist, dbestm, dbestc;
  const int nblk = (int)std::min<int64_t>((n + 255) / 256, 2048);
  TM_TRY(dclust.alloc((size_t)n * 4)); TM_TRY(ddis.alloc((size_t)n * 4)); TM_TRY(dmemb.alloc((size_t)n * 4)); TM_TRY(dmembers.alloc((size_t)k * 4));
  TM_TRY(dfreq.alloc((size_t)k * KM_ATTRS * nmod * 4)); TM_TRY(dcent.alloc((size_t)k * KM_ATTRS)); TM_TRY(dused.alloc((size_t)n)); TM_TRY(dmind.alloc((size_t)n * 4));
  TM_TRY(dpartial.alloc((size_t)nblk * 8)); TM_TRY(dwhich.alloc((size_t)k * 8)); TM_TRY(dscal.alloc(64)); TM_TRY(dmlist.alloc((size_t)(2 * KM_BIN + 8) * sizeof(int3)));
  TM_TRY(dbestm.alloc((size_t)n * 4)); TM_TRY(dbestc.alloc((size_t)k * KM_ATTRS));
  // scalars: [0] seed (u32), [8] cost (u64), [16] moves (u32), [20] nmoves (u32), [24] bad (int)
  KmState st;
  st.memb = dmemb.as<int32_t>(); st.clust = dclust.as<int32_t>(); st.dis = ddis.as<unsigned>(); st.members = dmembers.as<int32_t>(); st.freq = dfreq.as<int32_t>();
  st.cent = dcent.as<uint8_t>(); st.seed = dscal.as<uint32_t>(); st.cost = reinterpret_cast<unsigned long long *>(dscal.as<uint8_t>() + 8);
  st.moves = reinterpret_cast<unsigned *>(dscal.as<uint8_t>() + 16); st.nmoves = reinterpret_cast<unsigned *>(dscal.as<uint8_t>() + 20);
  st.bad = reinterpret_cast<int *>(dscal.as<uint8_t>() + 24); st.mlist = dmlist.as<int3>();
  {
    uint8_t init[64] = {0};
    const uint32_t seed0 = 0x42381337u;  // 933
    memcpy(init, &seed0, 4);
    TM_HIP(hipMemcpyAsync(dscal.p, init, sizeof(init), hipMemcpyHostToDevice, stream));
    TM_HIP(hipStreamSynchronize(stream));  // init[] is on the stack
  }
  hipLaunchKernelGGL(k_kmodes_validate, dim3((unsigned)std::min<int64_t>((n * KM_ATTRS + 255) / 256, 4096)), dim3(256), 0, stream, rows, n * KM_ATTRS, nmod, st.bad);
  // KModesIter's two kernels: G owners (clusters dealt round-robin) and the walker
  const int G = std::min(k, KM_MAX_OWNERS);
  const int max_own = (k + G - 1) / G;
  const size_t owner_lds = (size_t)((max_own * (KM_WORDS + 1) + 3 * (2 * KM_BIN + 8) + 3) & ~3) * 4 + (size_t)KM_STAGE * KM_ATTRS;
  TM_CHECK(owner_lds <= 120 * 1024 && (size_t)k * 4 <= 48 * 1024, TM_E_INVAL, "kmodes: %d clusters need %zu bytes of LDS per workgroup", k, owner_lds);
  if (owner_lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_kmodes_owner), hipFuncAttributeMaxDynamicSharedMemorySize, (int)owner_lds);
  DevBuf dpart;
  TM_TRY(dpart.alloc((size_t)KM_BIN * G * 4));
  // One iteration is 2 * bins + 1 dependent launches with the same arguments every time: a graph, built once per call and replayed
  // (the host's share of a launch is then paid once, not 3 400 times per iteration)
  DevBuf dstop;  // [0] where the fast leg stopped, [1] the bin the iteration's graph starts at (its launches for earlier bins return at once)
  TM_TRY(dstop.alloc(16));
  TM_HIP(hipMemsetAsync(dstop.p, 0, 16, stream));
  struct IterGraph {
    hipGraph_t g = nullptr;
    hipGraphExec_t exec = nullptr;
    ~IterGraph() { if (exec) (void)hipGraphExecDestroy(exec); if (g) (void)hipGraphDestroy(g); }
  } iter_graph;
  {
    TM_HIP(hipGraphCreate(&iter_graph.g, 0));
    const int64_t nbins = (n + KM_BIN - 1) / KM_BIN;
    unsigned *part_p = dpart.as<unsigned>();
    const long long *start_p = dstop.as<long long>() + 1;
    int k_ = k, nmod_ = nmod, G_ = G;
    int64_t n_ = n;
    const uint8_t *rows_ = rows;
    hipGraphNode_t prev = nullptr;
    auto add = [&](const void *fn, unsigned grid, size_t shm, void **params) -> int {
      hipKernelNodeParams np;
      memset(&np, 0, sizeof np);
      np.func = const_cast<void *>(fn);
      np.gridDim = dim3(grid); np.blockDim = dim3(KM_WT); np.sharedMemBytes = (unsigned)shm; np.kernelParams = params;
      hipGraphNode_t node;
      TM_HIP(hipGraphAddKernelNode(&node, iter_graph.g, prev ? &prev : nullptr, prev ? 1 : 0, &np));
      prev = node;
      return TM_OK;
    };
    for (int64_t bin = 0; bin <= nbins; bin++) {
      int64_t b = bin < nbins ? bin : -1;  // the closing launch applies the last bin's moves
      int first = bin == 0 ? 1 : 0;
      void *po[] = {&rows_, &n_, &k_, &nmod_, &G_, &b, &first, &st, &part_p, &start_p};
      TM_TRY(add(reinterpret_cast<const void *>(&k_kmodes_owner), (unsigned)G, owner_lds, po));
      if (bin == nbins) break;
      void *pw[] = {&n_, &k_, &G_, &b, &st, &part_p, &start_p};
      TM_TRY(add(reinterpret_cast<const void *>(&k_kmodes_walker), 1u, (size_t)k * 4, pw));
    }
    TM_HIP(hipGraphInstantiate(&iter_graph.exec, iter_graph.g, nullptr, nullptr, 0));
  }
  // the fast leg's workgroup owns every cluster: its LDS holds all k modes and the member counts beside the walker's and the owners' fixed arrays
  const int64_t nbins = (n + KM_BIN - 1) / KM_BIN;
  const size_t fast_lds = (size_t)k * 4;
  const bool fast_ok = !knobs().kmodes_binwise;
  const size_t lds = (size_t)k * KM_ATTRS;
  TM_CHECK(lds <= 160 * 1024 - 1024, TM_E_INVAL, "kmodes: the modes of %d clusters do not fit LDS", k);
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_kmodes_argmin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  auto score = [&](int64_t first, int64_t last) -> int {  // clust / dis of points [first, last) against the modes as they stand
    hipLaunchKernelGGL(k_kmodes_argmin, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((last - first + 255) / 256, 2048))), dim3(256), lds, stream,
                       reinterpret_cast<const uint32_t *>(rows), first, last, dcent.as<uint32_t>(), k, dclust.as<int32_t>(), ddis.as<unsigned>());
    TM_HIP(hipGetLastError());
    return TM_OK;
  };
  struct Scal { uint32_t seed, pad; unsigned long long cost; unsigned moves, nmoves; int bad; };
  auto read_scal = [&](Scal *h) -> int {
    TM_HIP(hipMemcpyAsync(h, dscal.p, sizeof(Scal), hipMemcpyDeviceToHost, stream));
    TM_HIP(hipStreamSynchronize(stream));
    TM_CHECK(h->bad != 1, TM_E_INVAL, "kmodes: a value is not below the %d modalities", nmod);
    TM_CHECK(h->bad == 0, TM_E_INVAL, "kmodes: empty-cluster repair found no donor");
    return TM_OK;
  };
  {  // nothing below may index a histogram with a value that is not a modality
    Scal h;
    TM_TRY(read_scal(&h));
  }
  std::vector<u64> partial((size_t)nblk);
  std::vector<int64_t> which((size_t)k);
  uint64_t all_best = ~0ull;
  int all_iters = 0;
  int64_t point_iters = 0;
  for (int run = 0; run < nruns; run++) {
    // ---- InitFarthestFirst (694-772): the picks settle on the host from every block's candidate
    TM_HIP(hipMemsetAsync(dmind.p, 0xff, (size_t)n * 4, stream));
    TM_HIP(hipMemsetAsync(dused.p, 0, (size_t)n, stream));
    int64_t far = starts[(size_t)run];
    for (int c = 0; c < k; c++) {
      which[(size_t)c] = far;
      TM_HIP(hipMemsetAsync(dused.as<uint8_t>() + far, 1, 1, stream));
      if (c == k - 1) break;
      hipLaunchKernelGGL(k_kmodes_ff, dim3(nblk), dim3(256), 0, stream, reinterpret_cast<const uint32_t *>(rows), n, far, dused.as<uint8_t>(), dmind.as<unsigned>(), dpartial.as<u64>());
      TM_HIP(hipGetLastError());
      TM_HIP(hipMemcpyAsync(partial.data(), dpartial.p, (size_t)nblk * 8, hipMemcpyDeviceToHost, stream));
      TM_HIP(hipStreamSynchronize(stream));
      u64 best = 0;
      for (u64 v : partial) best = std::max(best, v);
      far = best ? (int64_t)(uint32_t)(best - 1) : starts[(size_t)run];  // no unused point left: ifarthest stays InitPoint (756)
    }
    TM_HIP(hipMemcpyAsync(dwhich.p, which.data(), (size_t)k * 8, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_kmodes_take_rows, dim3((unsigned)((k * KM_ATTRS + 255) / 256)), dim3(256), 0, stream, rows, dwhich.as<int64_t>(), k, st.cent);
    // ---- initial assignment and modes (978-1011)
    TM_HIP(hipMemsetAsync(dfreq.p, 0, (size_t)k * KM_ATTRS * nmod * 4, stream));
    TM_HIP(hipMemsetAsync(dmembers.p, 0, (size_t)k * 4, stream));
    TM_TRY(score(0, n));
    hipLaunchKernelGGL(k_kmodes_init_tables, dim3((unsigned)std::min<int64_t>((n * KM_ATTRS + 255) / 256, 8192)), dim3(256), 0, stream, rows, n, nmod, st);
    hipLaunchKernelGGL(k_kmodes_init_modes, dim3((unsigned)std::min(k * KM_ATTRS, 2048) + 1), dim3(64), 0, stream, rows, n, k, nmod, st);
    TM_HIP(hipGetLastError());
    TM_HIP(hipStreamSynchronize(stream));  // `which` is host memory the copy above reads
    int itr = 0, worse = 0, bestitr = 0;
    long long prev_moves = LLONG_MAX;
    bool converged = false;
    uint64_t prevcost = ~0ull, bestcost = ~0ull;
    while (itr < max_iter && !converged) {
      itr++;
      point_iters += n;
      // ---- KModesIter (851-921): bins of 960 points, each scored against the modes as they stand when its turn comes
      TM_HIP(hipMemsetAsync(dscal.as<uint8_t>() + 8, 0, 12, stream));  // cost, moves
      const auto t_it0 = std::chrono::steady_clock::now();
      // (not where the iteration before moved more than 16 points a bin: with that many movers nearly every bin changes a mode or has more
      // moves than the one workgroup's move-by-move MovePointCat is good for)
      if (fast_ok && itr >= 2 && (prev_moves <= 16 * nbins || knobs().kmodes_fast_always)) {
        // the fast leg (k_kmodes_fast): all remaining points scored at once, bins walked by one launch until a mode changes
        int64_t b = 0;
        int stops = 0;
        while (b < nbins && stops < KM_FAST_STOPS) {
          TM_TRY(score(b * KM_BIN, n));
          hipLaunchKernelGGL(k_kmodes_fast, dim3(1), dim3(KM_WT), fast_lds, stream, rows, n, k, nmod, b, nbins, st, dstop.as<long long>());
          TM_HIP(hipGetLastError());
          long long stop = 0;
          TM_HIP(hipMemcpyAsync(&stop, dstop.p, 8, hipMemcpyDeviceToHost, stream));
          TM_HIP(hipStreamSynchronize(stream));
          TM_CHECK(stop > b && stop <= nbins, TM_E_HIP, "kmodes: the fast leg stopped at bin %lld of [%lld, %lld]", stop, (long long)b, (long long)nbins);
          if (stop < nbins) stops++;
          b = stop;
        }
        if (knobs().pp_debug) fprintf(stderr, "[tm_kmodes] iteration %d: the fast leg walked %lld of %lld bins (%d stops)\n", itr, (long long)b, (long long)nbins, stops);
        if (b < nbins) {  // the modes keep moving: the rest of the iteration bin by bin, by the graph from bin b on (the fast leg left no move unapplied)
          const long long bb = b, zero = 0;
          TM_HIP(hipMemcpyAsync(dstop.as<long long>() + 1, &bb, 8, hipMemcpyHostToDevice, stream));
          TM_HIP(hipGraphLaunch(iter_graph.exec, stream));
          TM_HIP(hipMemcpyAsync(dstop.as<long long>() + 1, &zero, 8, hipMemcpyHostToDevice, stream));
          TM_HIP(hipStreamSynchronize(stream));  // (bb and zero are this frame's)
        }
      } else {
        TM_HIP(hipGraphLaunch(iter_graph.exec, stream));
      }
      Scal h;
      TM_TRY(read_scal(&h));
      const uint64_t cost = h.cost;
      const int moves = (int)h.moves;
      if (knobs().pp_debug)
        fprintf(stderr, "[tm_kmodes] run %d iteration %d: cost %llu, %d moves, %.2f ms (mode words changed so far, counted by a diagnostic build: %u)\n", run, itr,
                (unsigned long long)cost, moves, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_it0).count(), h.pad);
      converged = cost >= prevcost;
      if (converged) {  // SameValue(cost, prevcost, prevcost div 1000), 1041
        const double a = (double)cost, b = (double)prevcost;
        double eps = (double)(prevcost / 1000);
        if (eps == 0) eps = std::max(std::min(std::fabs(a), std::fabs(b)) * 1e-12, 1e-12);
        const bool same = a > b ? (a - b) <= eps : (b - a) <= eps;
        if (same) { worse++; if (worse < 3) converged = false; }
      }
      converged = converged || moves == 0;
      if (cost < bestcost) {
        bestitr = itr; bestcost = cost;
        TM_HIP(hipMemcpyAsync(dbestm.p, dmemb.p, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
        TM_HIP(hipMemcpyAsync(dbestc.p, dcent.p, (size_t)k * KM_ATTRS, hipMemcpyDeviceToDevice, stream));
      }
      prevcost = cost;
      prev_moves = moves;
    }
    if (bestcost < all_best) {  // 1078-1085: the first run with the strictly smallest cost
      all_best = bestcost;
      all_iters = bestitr;
      TM_HIP(hipMemcpyAsync(labels_out, dbestm.p, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
      TM_HIP(hipMemcpyAsync(cent_out, dbestc.p, (size_t)k * KM_ATTRS, hipMemcpyDeviceToDevice, stream));
    }
  }
  TM_HIP(hipStreamSynchronize(stream));
  if (cost_out) *cost_out = all_best;
  if (iters_out) *iters_out = all_iters;
  if (point_iters_out) *point_iters_out = point_iters;
  return TM_OK;
}

// the Pascal arrays' form: HOST pointers, an upload and a read-back round the device run
int run_kmodes(const uint8_t *rows, int64_t n, int k, int num_init, int nmod, int max_iter, int32_t *labels_out, uint8_t *cent_out, uint64_t *cost_out,
               int *iters_out, hipStream_t stream) {
  TM_TRY(require_device());
  TM_CHECK(rows && labels_out && cent_out, TM_E_INVAL, "kmodes: null argument");
  TM_CHECK(n >= 1 && k >= 1 && k <= 4096, TM_E_INVAL, "kmodes: %lld points, %d clusters", (long long)n, k);
  DevBuf drows, dlab, dcen;
  TM_TRY(drows.alloc((size_t)n * KM_ATTRS)); TM_TRY(dlab.alloc((size_t)n * 4)); TM_TRY(dcen.alloc((size_t)k * KM_ATTRS));
  TM_HIP(hipMemcpyAsync(drows.p, rows, (size_t)n * KM_ATTRS, hipMemcpyHostToDevice, stream));
  TM_TRY(run_kmodes_dev(drows.as<uint8_t>(), n, k, num_init, nmod, max_iter, dlab.as<int32_t>(), dcen.as<uint8_t>(), cost_out, iters_out, nullptr, stream));
  TM_HIP(hipMemcpyAsync(labels_out, dlab.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
  TM_HIP(hipMemcpyAsync(cent_out, dcen.p, (size_t)k * KM_ATTRS, hipMemcpyDeviceToHost, stream));
  TM_HIP(hipStreamSynchronize(stream));
  return TM_OK;
}

}  // namespace tmx

extern "C" int tm_stage_kmodes_dev(const uint8_t *dev_rows, int64_t n, int num_clusters, int num_init, int num_modalities, int max_iter, int32_t *dev_labels,
                                   uint8_t *dev_centroids, uint64_t *host_cost, int *host_iters, int64_t *host_point_iters, void *stream) {
  return tmx::run_kmodes_dev(dev_rows, n, num_clusters, num_init, num_modalities, max_iter, dev_labels, dev_centroids, host_cost, host_iters, host_point_iters, (hipStream_t)stream);
}

extern "C" int tm_stage_kmodes(const uint8_t *host_rows, int64_t n, int num_clusters, int num_init, int num_modalities, int max_iter, int32_t *host_labels,
                               uint8_t *host_centroids, uint64_t *host_cost, int *host_iters, void *stream) {
  return tmx::run_kmodes(host_rows, n, num_clusters, num_init, num_modalities, max_iter, host_labels, host_centroids, host_cost, host_iters, (hipStream_t)stream);
}
