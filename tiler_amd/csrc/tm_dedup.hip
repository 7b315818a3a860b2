// tm_dedup.hip -- exact tile deduplication + reindexing, A8/A16.
//
// MakeTilesUnique (tilingencoder.pas:4720-4781) sorts tile pointers by pixel content and merges equal runs
// (MergeTiles 4783-4813: sum UseCount into the run's first tile); ReindexTiles (4626-4700) then packs the tiles that
// are Active with UseCount > 0 and sorts them by UseCount descending, content ascending (CompareTileUseCountRev,
// 584-599).  Content order is CompareDWord on the 64 RGB dwords (unsigned) or CompareByte on the 64 palette
// indices (940-948).
//
// GPU form: rows are first GROUPED by a 64-bit content hash (stable radix sort of (hash, index): equal rows become
// neighbours, lowest index first); every non-head row is compared in full with its predecessor, so a hash collision
// cannot merge different rows -- it only sends the call down the plain path, a stable merge sort of all row indices
// with a comparator that reads the rows.  Only the distinct rows are then merge-sorted by content (the order ReindexTiles
// needs), run bookkeeping uses integer atomics for the merged use counts (order free), and a stable radix sort on
// ~UseCount finishes.  rocPRIM supplies the sort and scan primitives; the hash, comparator, run detection, merge
// bookkeeping and ranking kernels are ours.  Representative of a run = its lowest original index (the reference's
// choice among byte-identical tiles is implementation defined).
#include <cstring>

#include <rocprim/device/device_merge_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cmath>

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {

struct RowLess {
  const uint32_t *rows;
  int dwords;
  int bytewise;  // CompareByte order: compare dwords big-endian
  __device__ int cmp(uint32_t a, uint32_t b) const {
    const uint4 *pa = reinterpret_cast<const uint4 *>(rows + (int64_t)a * dwords);
    const uint4 *pb = reinterpret_cast<const uint4 *>(rows + (int64_t)b * dwords);
    for (int i = 0; i < dwords / 4; i++) {
      const uint4 x = pa[i], y = pb[i];
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (xs[k] != ys[k]) {
          const uint32_t u = bytewise ? __builtin_bswap32(xs[k]) : xs[k];
          const uint32_t v = bytewise ? __builtin_bswap32(ys[k]) : ys[k];
          return u < v ? -1 : 1;
        }
      }
    }
    return 0;
  }
  __device__ bool operator()(const uint32_t &a, const uint32_t &b) const { return cmp(a, b) < 0; }
};

// 64-bit content hash: 16 lanes per row, one uint4 each per 256 bytes; position enters every term, the terms add up
// (`lead`: the row's first dword as lane `sub` 0 read it)
__device__ __forceinline__ unsigned long long row_hash16(const uint32_t *__restrict__ rows, int64_t row, int64_t n, int dwords, int sub, uint32_t &lead) {
  unsigned long long h = 0;
  if (row < n)
    for (int v = sub; v < dwords / 4; v += 16) {
      const uint4 x = *reinterpret_cast<const uint4 *>(rows + row * dwords + v * 4);
      if (v == 0) lead = x.x;
      unsigned long long a = ((unsigned long long)x.y << 32 | x.x) + 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * v + 1);
      unsigned long long b = ((unsigned long long)x.w << 32 | x.z) + 0xC2B2AE3D27D4EB4Full * (unsigned long long)(2 * v + 2);
      a ^= a >> 32; a *= 0xD6E8FEB86659FD93ull; a ^= a >> 32;
      b ^= b >> 29; b *= 0xBF58476D1CE4E5B9ull; b ^= b >> 32;
      h += a * 0x94D049BB133111EBull + b;
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) h += __shfl_xor(h, o);
  return h;
}
__global__ __launch_bounds__(256) void k_row_hash(const uint32_t *__restrict__ rows, int64_t n, int dwords, int degrade, int shift,
                                                  unsigned long long *__restrict__ hash) {
  const int sub = threadIdx.x & 15;
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t r0 = blockIdx.x * (int64_t)16; r0 < n; r0 += stride) {  // a block covers 16 rows per pass
    const int64_t row = r0 + (threadIdx.x >> 4);
    uint32_t lead = 0;
    const unsigned long long h = row_hash16(rows, row, n, dwords, sub, lead);
    if (row < n && sub == 0) hash[row] = degrade ? (h & 3) : (h >> shift);  // degrade: test hook that forces collisions; shift: only the top bits are kept (run_dedup_ex)
  }
}

// ---- grouping by a hash TABLE (the default front end since round 5; the hash SORT above stays behind TM_DEDUP_SORT) ------------------------
// The sort made equal rows neighbours -- seven radix passes over (hash, index) of the clip's 4.32 M frame tiles, head marks, two scans and the
// run bookkeeping: 0.96 of Reduce's 2.2 ms -- where all that is asked is each row's group: its lowest-index member and the group's use count.
// An open-addressing table of 64-bit hashes (linear probing, at most two thirds full) gives that with one compare-and-swap per probe:
//   k_dd_insert   hashes a row (16 lanes) and claims or finds its hash's slot; the claimant leaves its index as the slot's OWNER, a row that
//                 finds the hash present lowers the slot's MINDUP (atomic minimum).  Both words of a slot lie side by side.
//   k_dd_resolve  representative = min(owner, mindup); a row that is not its own representative adds its use to the representative's count
//   k_dd_verify   ... and is compared with it IN FULL (a hash shared by different rows puts them into one slot, where at least one of them
//                 differs from the slot's lowest row: the flag sends the call down the plain path exactly as the sort's full compare did).
//   k_dd_compact  the representatives in index order (an exclusive scan of the head marks), their own use added, and what the partial order
//                 below wants of them -- use count and leading dword -- as arrays in that order (its three passes gathered both per row).
// Integer atomics only: the groups, their representatives and counts do not depend on who came first.
struct DdSlot { uint32_t owner, mindup; };
// (a workgroup hashes 256 rows sixteen at a time -- 16 lanes per row: coalesced 256-byte reads -- and then every thread takes ONE row to the
// table: with the row's first lane doing its probing the kernel had four atomics in flight per wave and took 607 us, 380 of them waiting)
__global__ __launch_bounds__(256) void k_dd_insert(const uint32_t *__restrict__ rows, int64_t n, int dwords, int degrade, int bytewise,
                                                   unsigned long long *__restrict__ tkey, DdSlot *__restrict__ tslot, uint32_t mask,
                                                   uint32_t *__restrict__ slot_of, uint32_t *__restrict__ lead_of) {
  __shared__ unsigned long long s_h[256];
  __shared__ uint32_t s_lead[256];
  const int sub = threadIdx.x & 15;
  for (int64_t c0 = blockIdx.x * (int64_t)256; c0 < n; c0 += (int64_t)gridDim.x * 256) {
    __syncthreads();  // (the previous chunk's hashes have been taken)
#pragma unroll 4
    for (int p = 0; p < 16; p++) {
      const int lr = p * 16 + (threadIdx.x >> 4);
      uint32_t lead = 0;
      const unsigned long long h = row_hash16(rows, c0 + lr, n, dwords, sub, lead);
      if (sub == 0) { s_h[lr] = h; s_lead[lr] = lead; }
    }
    __syncthreads();
    const int64_t row = c0 + threadIdx.x;
    if (row < n) {
      unsigned long long h = s_h[threadIdx.x];
      if (degrade) h &= 3;
      const unsigned long long key = h ? h : 1ull;  // (0 = an empty slot)
      uint32_t s = (uint32_t)(h ^ (h >> 32)) & mask;
      for (;;) {
        const unsigned long long old = atomicCAS(&tkey[s], 0ull, key);
        if (old == 0ull) { tslot[s].owner = (uint32_t)row; break; }  // (read by the next kernel)
        if (old == key) { atomicMin(&tslot[s].mindup, (uint32_t)row); break; }
        s = (s + 1) & mask;
      }
      slot_of[row] = s;
      const uint32_t lead = s_lead[threadIdx.x];
      lead_of[row] = bytewise ? __builtin_bswap32(lead) : lead;
    }
  }
}
// a thread per row: its representative, head mark, and a duplicate's use added to the representative's count
__global__ __launch_bounds__(256) void k_dd_resolve(int64_t n, const DdSlot *__restrict__ tslot, const uint32_t *__restrict__ slot_of, const uint32_t *__restrict__ use_in,
                                                    uint32_t *__restrict__ rep, uint32_t *__restrict__ head, uint32_t *__restrict__ use_rep) {
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < n; row += (int64_t)gridDim.x * blockDim.x) {
    const DdSlot sl = tslot[slot_of[row]];
    const uint32_t r = min(sl.owner, sl.mindup);
    rep[row] = r;
    head[row] = r == (uint32_t)row ? 1u : 0u;
    if (r != (uint32_t)row) {
      const uint32_t u = use_in ? use_in[row] : 1u;
      if (u) atomicAdd(&use_rep[r], u);
    }
  }
}
// every row that is not its own representative against it in full: a wave looks at 64 rows at once (one coalesced read of their
// representatives) and compares the duplicates among them four at a time, 16 lanes per row (a walk over all rows 16 lanes each was a chain of
// dependent loads: 77 us for a clip without a single duplicate)
__global__ __launch_bounds__(256) void k_dd_verify(const uint32_t *__restrict__ rows, int64_t n, int dwords, const uint32_t *__restrict__ rep, int *__restrict__ collision) {
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4, vecs = dwords / 4;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  bool diff = false;
  for (int64_t base = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64; base < n; base += nw * 64) {
    const int64_t row = base + lane;
    const uint32_t r = row < n ? rep[row] : (uint32_t)row;
    unsigned long long m = __builtin_amdgcn_ballot_w64(r != (uint32_t)row);
    while (m) {
      int mine = -1;  // this 16-lane group's duplicate of the round: the grp-th of the next four
#pragma unroll
      for (int t = 0; t < 4; t++) {
        if (m) {
          const int b = __builtin_ctzll(m);
          m &= m - 1;
          if (t == grp) mine = b;
        }
      }
      const uint32_t rr = (uint32_t)__shfl((int)r, mine >= 0 ? mine : 0);  // (every lane takes part: the source lane may belong to a group that sits this round out)
      if (mine >= 0) {
        const uint4 *pa = reinterpret_cast<const uint4 *>(rows + (int64_t)rr * dwords);
        const uint4 *pb = reinterpret_cast<const uint4 *>(rows + (base + mine) * dwords);
        for (int v = sub; v < vecs; v += 16) {
          const uint4 x = pa[v], y = pb[v];
          diff |= (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
        }
      }
    }
  }
  if (__builtin_amdgcn_ballot_w64(diff) && lane == 0) *collision = 1;
}
__global__ void k_dd_compact(int64_t n, const uint32_t *__restrict__ head, const uint32_t *__restrict__ head_excl, const uint32_t *__restrict__ use_in,
                             const uint32_t *__restrict__ lead_of, uint32_t *__restrict__ use_rep, uint32_t *__restrict__ uniq, uint32_t *__restrict__ cuse,
                             uint32_t *__restrict__ clead) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (head[i]) {
      const uint32_t j = head_excl[i], u = use_rep[i] + (use_in ? use_in[i] : 1u);
      use_rep[i] = u;
      uniq[j] = (uint32_t)i;
      cuse[j] = u;
      clead[j] = lead_of[i];
    }
}

// runs of equal hash: head flags as k_mark_heads writes them; a non-head row that differs from its predecessor is a collision.  16 lanes per row: a uint4 each, so a 256-byte row is one coalesced read per side (a thread
// walking both rows on its own took 0.43 ms for the bench clip's 1.08 M non-head rows)
__global__ __launch_bounds__(256) void k_mark_heads_hash(const uint32_t *__restrict__ sorted, const unsigned long long *__restrict__ hsorted, int64_t n,
                                                         RowLess less, uint32_t *__restrict__ head, uint32_t *__restrict__ headpos, int *__restrict__ collision) {
  const int sub = threadIdx.x & 15, vecs = less.dwords / 4;
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t i0 = blockIdx.x * (int64_t)16; i0 < n; i0 += stride) {
    const int64_t i = i0 + (threadIdx.x >> 4);
    bool h = true, diff = false;
    if (i < n) {
      h = (i == 0) || hsorted[i] != hsorted[i - 1];
      if (!h) {
        const uint4 *pa = reinterpret_cast<const uint4 *>(less.rows + (int64_t)sorted[i - 1] * less.dwords);
        const uint4 *pb = reinterpret_cast<const uint4 *>(less.rows + (int64_t)sorted[i] * less.dwords);
        for (int v = sub; v < vecs; v += 16) {
          const uint4 x = pa[v], y = pb[v];
          diff |= (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
        }
      }
      if (sub == 0) {
        head[i] = h ? 1u : 0u;
        headpos[i] = h ? (uint32_t)i : 0u;
      }
    }
    if (__builtin_amdgcn_ballot_w64(diff) && (threadIdx.x & 63) == 0) *collision = 1;
  }
}

// sort key of a distinct row: its first two dwords in comparison order ride along, so that most comparisons never touch the rows
struct PrefixKey {
  unsigned long long prefix;
  uint32_t row, pad;
};
struct PrefixLess {
  RowLess less;
  __device__ bool operator()(const PrefixKey &a, const PrefixKey &b) const {
    if (a.prefix != b.prefix) return a.prefix < b.prefix;
    return less.cmp(a.row, b.row) < 0;
  }
};
__global__ void k_prefix_keys(const uint32_t *__restrict__ uniq, int64_t nu, RowLess less, PrefixKey *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nu; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t r = uniq[i];
    const uint2 d = *reinterpret_cast<const uint2 *>(less.rows + (int64_t)r * less.dwords);
    const uint32_t d0 = less.bytewise ? __builtin_bswap32(d.x) : d.x, d1 = less.bytewise ? __builtin_bswap32(d.y) : d.y;
    out[i] = PrefixKey{((unsigned long long)d0 << 32) | d1, r, 0u};
  }
}
__global__ void k_prefix_rows(const PrefixKey *__restrict__ keys, int64_t nu, uint32_t *__restrict__ uniq) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nu; i += (int64_t)gridDim.x * blockDim.x) uniq[i] = keys[i].row;
}

// ---- the content sort of MANY distinct rows: by the 8-byte prefix with a radix sort, whole rows only where prefixes tie -----------------
// The comparator merge sort of (prefix, row) keys takes 2 ms for the clip's 4.32 M distinct frame tiles (a dedup without a tile budget: the
// whole order is asked for).  Most keys differ in their prefix: a radix sort of (prefix, row) pairs orders those for good, and what is left
// are short runs of equal prefixes, each put into content order by the thread at its head (insertion sort with the full comparator; a stable
// sort leaves equal prefixes in the order they came in, any order would do).  A run longer than PK_MAX_RUN rows -- flat content: thousands of
// rows may share their first pixels -- sets a flag and the call takes the comparator merge sort as before: the same result by construction
// (one total order).  Only from TM_DEDUP_RADIX_MIN keys on (default 2^20): below that the merge sort's fixed cost is small and a wasted
// radix sort is not (Reindex's 321 k palette-index rows tie in long runs on the bench clip: measured, 0.14 ms lost).
__global__ void k_pk_split(const PrefixKey *__restrict__ pk, int64_t m, unsigned long long *__restrict__ key, uint32_t *__restrict__ row) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) { key[i] = pk[i].prefix; row[i] = pk[i].row; }
}
constexpr int PK_MAX_RUN = 8;
__global__ void k_pk_ties(const unsigned long long *__restrict__ key, uint32_t *__restrict__ row, int64_t m, RowLess less, int *__restrict__ too_long) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long k = key[i];
    if ((i > 0 && key[i - 1] == k) || i + 1 >= m || key[i + 1] != k) continue;  // not the head of a run of two or more
    int len = 2;
    while (len <= PK_MAX_RUN && i + len < m && key[i + len] == k) len++;
    if (len > PK_MAX_RUN) { *too_long = 1; continue; }
    uint32_t r[PK_MAX_RUN];
#pragma unroll
    for (int j = 0; j < PK_MAX_RUN; j++) r[j] = j < len ? row[i + j] : 0u;
#pragma unroll
    for (int a = 1; a < PK_MAX_RUN; a++) {  // insertion sort (a register array: compile-time indices, the moves by selects)
      if (a < len) {
        const uint32_t x = r[a];
        int pos = a;
#pragma unroll
        for (int b = a - 1; b >= 0; b--)
          if (pos == b + 1 && less.cmp(x, r[b]) < 0) { r[b + 1] = r[b]; pos = b; }
#pragma unroll
        for (int b = 0; b < PK_MAX_RUN; b++) if (b == pos) r[b] = x;
      }
    }
#pragma unroll
    for (int j = 0; j < PK_MAX_RUN; j++) if (j < len) row[i + j] = r[j];
  }
}

// ---- only the first `exact_first` positions of the final order matter (Reduce keeps that many tiles): which distinct rows can be there --
// use counts of the distinct rows, clamped to 1023: a histogram per workgroup in LDS (most rows are used once: one global counter would take
// every row's atomic in turn), flushed with one atomic per occupied bin
// (the flush goes to the histogram copy of the XCD the workgroup runs on -- hist8: [8][bins], folded by k_po_fold --: every workgroup
// holds single-use rows, and 4 096 workgroups of eight XCDs adding to hist[1] passed that one word round for most of the kernel's 70 us)
__device__ __forceinline__ unsigned po_xcc() {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
  return xcc & 7u;
}
__global__ void k_po_fold(const uint32_t *__restrict__ hist8, int bins, uint32_t *__restrict__ hist) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < bins; e += gridDim.x * blockDim.x) {
    uint32_t s = 0;
#pragma unroll
    for (int x = 0; x < 8; x++) s += hist8[x * bins + e];
    hist[e] = s;
  }
}
// (cuse / clead: the distinct rows' use counts and leading dwords in uniq's order where the table front end made them; null: gathered per row)
__global__ __launch_bounds__(256) void k_po_use_hist(const uint32_t *__restrict__ uniq, int64_t nu, const uint32_t *__restrict__ use_rep, const uint32_t *__restrict__ cuse,
                                                     uint32_t *__restrict__ hist8) {
  __shared__ uint32_t s_h[1024];
  for (int e = threadIdx.x; e < 1024; e += 256) s_h[e] = 0;
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nu; i += (int64_t)gridDim.x * blockDim.x) atomicAdd(&s_h[min(cuse ? cuse[i] : use_rep[uniq[i]], 1023u)], 1u);
  __syncthreads();
  uint32_t *hist = hist8 + po_xcc() * 1024;
  for (int e = threadIdx.x; e < 1024; e += 256) if (s_h[e]) atomicAdd(&hist[e], s_h[e]);
}
__device__ __forceinline__ uint32_t po_lead(const RowLess &less, uint32_t row) {  // the row's leading dword in comparison order
  const uint32_t d = less.rows[(int64_t)row * less.dwords];
  return less.bytewise ? __builtin_bswap32(d) : d;
}
// among the rows used exactly `use_star` times: a histogram of the leading dword's top 12 bits
__global__ __launch_bounds__(256) void k_po_lead_hist(const uint32_t *__restrict__ uniq, int64_t nu, const uint32_t *__restrict__ use_rep, uint32_t use_star, RowLess less,
                                                      const uint32_t *__restrict__ cuse, const uint32_t *__restrict__ clead, uint32_t *__restrict__ hist8) {
  __shared__ uint32_t s_h[4096];
  for (int e = threadIdx.x; e < 4096; e += 256) s_h[e] = 0;
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nu; i += (int64_t)gridDim.x * blockDim.x) {
    if (cuse) {
      if (cuse[i] == use_star) atomicAdd(&s_h[clead[i] >> 20], 1u);
    } else {
      const uint32_t r = uniq[i];
      if (use_rep[r] == use_star) atomicAdd(&s_h[po_lead(less, r) >> 20], 1u);
    }
  }
  __syncthreads();
  uint32_t *hist = hist8 + po_xcc() * 4096;
  for (int e = threadIdx.x; e < 4096; e += 256) if (s_h[e]) atomicAdd(&hist[e], s_h[e]);
}
// candidate = used more often than use_star, or exactly that often with a leading dword in the first buckets
__global__ void k_po_flags(const uint32_t *__restrict__ uniq, int64_t nu, const uint32_t *__restrict__ use_rep, uint32_t use_star, uint32_t last_bucket, RowLess less,
                           const uint32_t *__restrict__ cuse, const uint32_t *__restrict__ clead, uint32_t *__restrict__ flag) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nu; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t u = cuse ? cuse[i] : use_rep[uniq[i]];
    flag[i] = (u > use_star || (u == use_star && ((cuse ? clead[i] : po_lead(less, uniq[i])) >> 20) <= last_bucket)) ? 1u : 0u;
  }
}
// candidates to the front (as sort keys: ~use and the leading dword ride along), the others behind them in the order they come
__global__ void k_po_split(const uint32_t *__restrict__ uniq, int64_t nu, const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos, int64_t ncand,
                           const uint32_t *__restrict__ use_rep, RowLess less, const uint32_t *__restrict__ cuse, const uint32_t *__restrict__ clead,
                           PrefixKey *__restrict__ cand, uint32_t *__restrict__ order_out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nu; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t r = uniq[i];
    if (flag[i]) cand[pos[i]] = PrefixKey{((unsigned long long)(~(cuse ? cuse[i] : use_rep[r])) << 32) | (cuse ? clead[i] : po_lead(less, r)), r, 0u};
    else order_out[ncand + (i - pos[i])] = r;
  }
}

__global__ void k_iota(uint32_t *p, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

// head[i] = 1 when sorted[i] starts a run of equal rows; headpos[i] = i at heads else 0 (for a max-scan)
__global__ void k_mark_heads(const uint32_t *__restrict__ sorted, int64_t n, RowLess less, uint32_t *__restrict__ head,
                             uint32_t *__restrict__ headpos) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool h = (i == 0) || less.cmp(sorted[i - 1], sorted[i]) != 0;
    head[i] = h ? 1u : 0u;
    headpos[i] = h ? (uint32_t)i : 0u;
  }
}

// rep[row] = first row of its run; use_rep[rep] += use_in[row]; uniq[rank of run] = rep (content order)
__global__ void k_merge_runs(const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ headpos_scanned,
                             const uint32_t *__restrict__ head_excl, const uint32_t *__restrict__ head, int64_t n,
                             const uint32_t *__restrict__ use_in, uint32_t *__restrict__ rep, uint32_t *__restrict__ use_rep,
                             uint32_t *__restrict__ uniq) {
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x; i0 < n; i0 += stride) {  // (workgroup-uniform bound: every lane reaches the ballot)
    const int64_t i = i0 + threadIdx.x;
    const bool in = i < n;
    const uint32_t hp = in ? headpos_scanned[i] : 0xffffffffu;
    const uint32_t row = in ? sorted[i] : 0u, r = in ? sorted[hp] : 0u;
    if (in) {
      rep[row] = r;
      if (head[i]) uniq[head_excl[i]] = r;
    }
    if (use_in) {
      // the same with counts to add up: a segmented sum over the lanes of a run (the hp of a run's rows are equal and the runs are
      // neighbours), one atomic per run and wave
      uint32_t v = in ? use_in[row] : 0u;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t ov = (uint32_t)__shfl_down((int)v, o);
        const uint32_t ohp = (uint32_t)__shfl_down((int)hp, o);
        if (lane + o < 64 && ohp == hp) v += ov;
      }
      const uint32_t prev = (uint32_t)__shfl_up((int)hp, 1);
      if (in && (lane == 0 || hp != prev) && v) atomicAdd(&use_rep[r], v);
    } else {
      // rows of a run are neighbours here, and a run's last row knows where the run began: its length is a plain store (3.2 M atomics on
      // words scattered over 17 MB, passed between the XCDs' L2s, were most of this kernel)
      if (in && (i == n - 1 || head[i + 1])) use_rep[r] = (uint32_t)(i - hp + 1);
    }
  }
}

__global__ void k_rank_keys(const uint32_t *__restrict__ uniq, int64_t nu, const uint32_t *__restrict__ use_rep, int by_index,
                            uint32_t *__restrict__ key, unsigned long long *__restrict__ live_count) {
  unsigned long long local = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nu; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t u = use_rep[uniq[i]];
    key[i] = by_index ? uniq[i] : ~u;  // ascending ~use == descending use; zero-use rows (~0) sink to the end
    local += (u > 0 || by_index) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(live_count, local);
}

__global__ void k_scatter_pos(const uint32_t *__restrict__ order, int64_t nlive, const uint32_t *__restrict__ use_rep,
                              int32_t *__restrict__ pos, uint32_t *__restrict__ use_out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nlive; i += (int64_t)gridDim.x * blockDim.x) {
    pos[order[i]] = (int32_t)i;
    use_out[i] = use_rep[order[i]];
  }
}

__global__ void k_remap(const uint32_t *__restrict__ rep, const int32_t *__restrict__ pos, int64_t n, int32_t *__restrict__ remap) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) remap[i] = pos[rep[i]];
}

static inline int gridn(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 16)); }

// m keys (prefix, row) into content order; the rows of that order to out_rows
static int sort_prefix_keys(DevBuf &pk, int64_t m, const RowLess &less, uint32_t *out_rows, DevBuf &tmp, hipStream_t stream) {
  if (m <= 0) return TM_OK;
  if (m >= knobs().dedup_radix_min) {
    DevBuf k1, k2, r1, r2, flag;
    TM_TRY(k1.alloc((size_t)m * 8)); TM_TRY(k2.alloc((size_t)m * 8)); TM_TRY(r1.alloc((size_t)m * 4)); TM_TRY(r2.alloc((size_t)m * 4)); TM_TRY(flag.alloc(4));
    TM_HIP(hipMemsetAsync(flag.p, 0, 4, stream));
    hipLaunchKernelGGL(k_pk_split, dim3(gridn(m)), dim3(256), 0, stream, pk.as<PrefixKey>(), m, k1.as<unsigned long long>(), r1.as<uint32_t>());
    size_t tb = 0;
    TM_HIP(rocprim::radix_sort_pairs(nullptr, tb, k1.as<unsigned long long>(), k2.as<unsigned long long>(), r1.as<uint32_t>(), r2.as<uint32_t>(), (size_t)m, 0, 64, stream));
    TM_TRY(tmp.alloc(tb));
    TM_HIP(rocprim::radix_sort_pairs(tmp.p, tb, k1.as<unsigned long long>(), k2.as<unsigned long long>(), r1.as<uint32_t>(), r2.as<uint32_t>(), (size_t)m, 0, 64, stream));
    hipLaunchKernelGGL(k_pk_ties, dim3(gridn(m)), dim3(256), 0, stream, k2.as<unsigned long long>(), r2.as<uint32_t>(), m, less, flag.as<int>());
    TM_HIP(hipMemcpyAsync(out_rows, r2.p, (size_t)m * 4, hipMemcpyDeviceToDevice, stream));
    int too_long = 0;
    {
      HostRead hr_(stream);
      TM_TRY(hr_.get(&too_long, flag.p, 4));
      TM_TRY(hr_.wait());
    }
    if (!too_long) return TM_OK;
  }
  DevBuf pk2;
  TM_TRY(pk2.alloc((size_t)m * sizeof(PrefixKey)));
  const PrefixLess pless{less};
  size_t tbu = 0;
  TM_HIP(rocprim::merge_sort(nullptr, tbu, pk.as<PrefixKey>(), pk2.as<PrefixKey>(), (size_t)m, pless, stream));
  TM_TRY(tmp.alloc(tbu));
  TM_HIP(rocprim::merge_sort(tmp.p, tbu, pk.as<PrefixKey>(), pk2.as<PrefixKey>(), (size_t)m, pless, stream));
  hipLaunchKernelGGL(k_prefix_rows, dim3(gridn(m)), dim3(256), 0, stream, pk2.as<PrefixKey>(), m, out_rows);
  return TM_OK;
}

// by_index = 0: the reference's ReindexTiles order (use desc, content asc), zero-use rows dropped.
// by_index = 1: representatives in ascending original index (used to search only distinct database rows; any
//               row_bytes multiple of 16, content compared as dwords).
// exact_first > 0 (only with by_index = 0 and no use_in): the caller keeps the first exact_first rows of the order and nothing of the rest but
//               their number -- those positions are exact, the ones behind them hold the remaining rows in no particular order.
int run_dedup_ex(const void *rows, int64_t n, int row_bytes, const void *use_in, void *remap, void *order, void *use_out,
                 int64_t *host_n_unique, int by_index, hipStream_t stream, int64_t exact_first) {
  TM_TRY(require_device());
  TM_CHECK(by_index ? (row_bytes > 0 && row_bytes % 16 == 0) : (row_bytes == 256 || row_bytes == 64), TM_E_INVAL,
           "dedup: row_bytes must be 256 (RGB) or 64 (palette indices)");
  TM_CHECK(n >= 0 && n < (int64_t)1 << 31, TM_E_INVAL, "dedup: row count out of range");
  if (host_n_unique) *host_n_unique = 0;
  if (n == 0) return TM_OK;
  RowLess less{(const uint32_t *)rows, row_bytes / 4, (!by_index && row_bytes == 64) ? 1 : 0};
  DevBuf idx, sorted, head, headpos, hps, head_excl, rep, use_rep, uniq, key, key2, ord2, pos, tmp, cnt;
  TM_TRY(idx.alloc(n * 4)); TM_TRY(sorted.alloc(n * 4)); TM_TRY(head.alloc(n * 4)); TM_TRY(headpos.alloc(n * 4));
  TM_TRY(head_excl.alloc(n * 4)); TM_TRY(hps.alloc(n * 4)); TM_TRY(rep.alloc(n * 4)); TM_TRY(use_rep.alloc(n * 4)); TM_TRY(uniq.alloc(n * 4));
  TM_TRY(key.alloc(n * 4)); TM_TRY(key2.alloc(n * 4)); TM_TRY(ord2.alloc(n * 4)); TM_TRY(pos.alloc(n * 4)); TM_TRY(cnt.alloc(16));
  hipLaunchKernelGGL(k_iota, dim3(gridn(n)), dim3(256), 0, stream, idx.as<uint32_t>(), n);
  bool grouped = false;  // true: `sorted` is in hash order (runs of equal rows, lowest index first), not yet in content order
  DevBuf hflag;  // set by the full compares of the hash groups: two different rows shared a hash
  TM_TRY(hflag.alloc(4));
  TM_HIP(hipMemsetAsync(hflag.p, 0, 4, stream));
  auto plain_heads = [&]() -> int {  // the plain path: a stable merge sort of all row indices with a comparator that reads the rows
    size_t tb = 0;
    TM_HIP(rocprim::merge_sort(nullptr, tb, idx.as<uint32_t>(), sorted.as<uint32_t>(), (size_t)n, less, stream));
    TM_TRY(tmp.alloc(tb));
    TM_HIP(rocprim::merge_sort(tmp.p, tb, idx.as<uint32_t>(), sorted.as<uint32_t>(), (size_t)n, less, stream));
    hipLaunchKernelGGL(k_mark_heads, dim3(gridn(n)), dim3(256), 0, stream, sorted.as<uint32_t>(), n, less, head.as<uint32_t>(),
                       headpos.as<uint32_t>());
    return TM_OK;
  };
  // the table front end (see k_dd_insert): rep, use_rep, uniq (in index order) and the distinct count without a sort
  DevBuf cuse, clead;
  bool table_ok = false;
  uint32_t last_excl = 0, last_head = 0;
  if (!knobs().dedup_plain && !knobs().dedup_sort) {
    int64_t slots = 1024;
    while (slots * 2 < n * 3) slots *= 2;
    DevBuf tkey, tslot, slot_of, lead_of;
    TM_TRY(tkey.alloc((size_t)slots * 8)); TM_TRY(tslot.alloc((size_t)slots * sizeof(DdSlot))); TM_TRY(slot_of.alloc(n * 4)); TM_TRY(lead_of.alloc(n * 4));
    TM_TRY(cuse.alloc(n * 4)); TM_TRY(clead.alloc(n * 4));
    TM_HIP(hipMemsetAsync(tkey.p, 0, (size_t)slots * 8, stream));
    TM_HIP(hipMemsetAsync(tslot.p, 0xff, (size_t)slots * sizeof(DdSlot), stream));
    TM_HIP(hipMemsetAsync(use_rep.p, 0, n * 4, stream));
    hipLaunchKernelGGL(k_dd_insert, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 256 * 8)), dim3(256), 0, stream, (const uint32_t *)rows, n, row_bytes / 4,
                       knobs().dedup_degrade_hash ? 1 : 0, less.bytewise, tkey.as<unsigned long long>(), tslot.as<DdSlot>(), (uint32_t)(slots - 1), slot_of.as<uint32_t>(),
                       lead_of.as<uint32_t>());
    hipLaunchKernelGGL(k_dd_resolve, dim3(gridn(n)), dim3(256), 0, stream, n, tslot.as<DdSlot>(), slot_of.as<uint32_t>(), (const uint32_t *)use_in, rep.as<uint32_t>(),
                       head.as<uint32_t>(), use_rep.as<uint32_t>());
    hipLaunchKernelGGL(k_dd_verify, dim3(gridn(n)), dim3(256), 0, stream, (const uint32_t *)rows, n, row_bytes / 4, rep.as<uint32_t>(), hflag.as<int>());
    size_t tb3 = 0;
    TM_HIP(rocprim::exclusive_scan(nullptr, tb3, head.as<uint32_t>(), head_excl.as<uint32_t>(), 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    TM_TRY(tmp.alloc(tb3));
    TM_HIP(rocprim::exclusive_scan(tmp.p, tb3, head.as<uint32_t>(), head_excl.as<uint32_t>(), 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(k_dd_compact, dim3(gridn(n)), dim3(256), 0, stream, n, head.as<uint32_t>(), head_excl.as<uint32_t>(), (const uint32_t *)use_in,
                       lead_of.as<uint32_t>(), use_rep.as<uint32_t>(), uniq.as<uint32_t>(), cuse.as<uint32_t>(), clead.as<uint32_t>());
    int collision = 0;
    {
      HostRead hr_(stream);
      TM_TRY(hr_.get(&last_excl, head_excl.as<uint32_t>() + (n - 1), 4));
      TM_TRY(hr_.get(&last_head, head.as<uint32_t>() + (n - 1), 4));
      TM_TRY(hr_.get(&collision, hflag.p, 4));
      TM_TRY(hr_.wait());
    }
    if (!collision) grouped = table_ok = true;  // (two different rows under one hash, about once in 2^20 calls of the clip's size: the plain path below, exact and slow)
  }
  if (!knobs().dedup_plain && knobs().dedup_sort) {
    DevBuf hkey, hkey2;
    TM_TRY(hkey.alloc(n * 8)); TM_TRY(hkey2.alloc(n * 8));
    // The sort only has to bring equal rows together, so the hash keeps only as many of its top bits (whole 8-bit passes of the sort) as hold
    // the chance of two different rows among n sharing them below 2^-12 -- 48 for Reindex's 321 k rows, 56 for the bench clip's 4.32 M frame tiles
    // (seven passes instead of eight).  Rows that share them and differ are caught by the full compare like any collision (the plain path then: exact, slow).
    // The kept bits are moved DOWN and sorted as bits [0, hbits): a range that ends at bit 64 without starting at 0 sends rocPRIM's
    // merge-sort path (up to ~1 M keys) through a mask built with a shift by 64 -- it then orders by the wrong bits and its merge reads out
    // of bounds (found the hard way: a memory access fault on the GPU box).
    int hbits = 64;
    const int degrade = knobs().dedup_degrade_hash ? 1 : 0;
    if (!degrade) {
      hbits = std::min(64, ((int)std::ceil(2.0 * std::log2((double)std::max<int64_t>(n, 2)) + 11.0) + 7) / 8 * 8);  // n^2 / 2^(bits + 1) <= 2^-12
    }
    hipLaunchKernelGGL(k_row_hash, dim3((unsigned)std::min<int64_t>((n + 15) / 16, 256 * 32)), dim3(256), 0, stream, (const uint32_t *)rows, n,
                       row_bytes / 4, degrade, 64 - hbits, hkey.as<unsigned long long>());
    size_t tbh = 0;
    TM_HIP(rocprim::radix_sort_pairs(nullptr, tbh, hkey.as<unsigned long long>(), hkey2.as<unsigned long long>(), idx.as<uint32_t>(),
                                     sorted.as<uint32_t>(), (size_t)n, 0, hbits, stream));
    TM_TRY(tmp.alloc(tbh));
    TM_HIP(rocprim::radix_sort_pairs(tmp.p, tbh, hkey.as<unsigned long long>(), hkey2.as<unsigned long long>(), idx.as<uint32_t>(),
                                     sorted.as<uint32_t>(), (size_t)n, 0, hbits, stream));
    hipLaunchKernelGGL(k_mark_heads_hash, dim3((unsigned)std::min<int64_t>((n + 15) / 16, 256 * 32)), dim3(256), 0, stream, sorted.as<uint32_t>(),
                       hkey2.as<unsigned long long>(), n, less, head.as<uint32_t>(), headpos.as<uint32_t>(), hflag.as<int>());
    grouped = true;  // until the flag says otherwise: it is read with the distinct count below, one round trip for both
  }
  if (!grouped) TM_TRY(plain_heads());
  while (!table_ok) {
    size_t tb2 = 0;
    TM_HIP(rocprim::inclusive_scan(nullptr, tb2, headpos.as<uint32_t>(), hps.as<uint32_t>(), (size_t)n, rocprim::maximum<uint32_t>(), stream));
    size_t tb3 = 0;
    TM_HIP(rocprim::exclusive_scan(nullptr, tb3, head.as<uint32_t>(), head_excl.as<uint32_t>(), 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    TM_TRY(tmp.alloc(std::max(tb2, tb3)));
    TM_HIP(rocprim::inclusive_scan(tmp.p, tb2, headpos.as<uint32_t>(), hps.as<uint32_t>(), (size_t)n, rocprim::maximum<uint32_t>(), stream));
    TM_HIP(rocprim::exclusive_scan(tmp.p, tb3, head.as<uint32_t>(), head_excl.as<uint32_t>(), 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    TM_HIP(hipMemsetAsync(use_rep.p, 0, n * 4, stream));
    hipLaunchKernelGGL(k_merge_runs, dim3(gridn(n)), dim3(256), 0, stream, sorted.as<uint32_t>(), hps.as<uint32_t>(),
                       head_excl.as<uint32_t>(), head.as<uint32_t>(), n, (const uint32_t *)use_in, rep.as<uint32_t>(),
                       use_rep.as<uint32_t>(), uniq.as<uint32_t>());
    // number of runs = head_excl[n-1] + head[n-1]
    int collision = 0;
    {
      HostRead hr_(stream);
      TM_TRY(hr_.get(&last_excl, head_excl.as<uint32_t>() + (n - 1), 4));
      TM_TRY(hr_.get(&last_head, head.as<uint32_t>() + (n - 1), 4));
      TM_TRY(hr_.get(&collision, hflag.p, 4));
      TM_TRY(hr_.wait());
    }
    if (!(grouped && collision)) break;
    grouped = false;  // a collision among the hashes (about once in 2^12 calls by the choice of bits above): the same again over the plain order
    TM_TRY(plain_heads());
  }
  const int64_t nu = (int64_t)last_excl + last_head;
  bool ranked = false;  // ord2 already holds the final order
  unsigned long long live = 0;
  if (grouped && !by_index && !use_in && exact_first > 0 && exact_first < nu && !knobs().dedup_full_order) {
    // Only the first exact_first positions have to be right.  The order is use count descending, content ascending: a histogram of the use
    // counts finds the count u* of position exact_first, a histogram of the leading dword's top bits among the rows used u* times finds
    // the bucket that position falls into; rows used more often, or u* times with a leading dword up to that bucket, are the only ones
    // that can come before it.  They alone are sorted (the comparator reads the rows where the keys tie); the others follow as they come.
    // 3.24 M distinct rows for a budget of 321 k on the bench clip: a merge sort of a tenth of them instead of all.
    DevBuf h1, h2, h8, flag, fpos;
    TM_TRY(h1.alloc(1024 * 4)); TM_TRY(h2.alloc(4096 * 4)); TM_TRY(h8.alloc(8 * 4096 * 4));
    const int po_grid = std::min(gridn(nu), 1024);  // (every workgroup flushes its bins: fewer workgroups, fewer atomics on the popular ones)
    TM_HIP(hipMemsetAsync(h8.p, 0, 8 * 1024 * 4, stream));
    const uint32_t *cu = table_ok ? cuse.as<uint32_t>() : nullptr, *cl = table_ok ? clead.as<uint32_t>() : nullptr;
    hipLaunchKernelGGL(k_po_use_hist, dim3(po_grid), dim3(256), 0, stream, uniq.as<uint32_t>(), nu, use_rep.as<uint32_t>(), cu, h8.as<uint32_t>());
    hipLaunchKernelGGL(k_po_fold, dim3(4), dim3(256), 0, stream, h8.as<uint32_t>(), 1024, h1.as<uint32_t>());
    std::vector<uint32_t> hh1(1024), hh2(4096);
    {
      HostRead hr_(stream);
      TM_TRY(hr_.get(hh1.data(), h1.p, 1024 * 4));
      TM_TRY(hr_.wait());
    }
    int64_t above = 0;
    int ustar = -1;
    for (int u = 1023; u >= 1; u--) {
      if (above + hh1[(size_t)u] >= exact_first) { ustar = u; break; }
      above += hh1[(size_t)u];
    }
    if (ustar >= 1 && ustar < 1023) {  // (a cut inside the clamped bin -- 1023 uses and more -- takes the full sort below)
      TM_HIP(hipMemsetAsync(h8.p, 0, 8 * 4096 * 4, stream));
      hipLaunchKernelGGL(k_po_lead_hist, dim3(po_grid), dim3(256), 0, stream, uniq.as<uint32_t>(), nu, use_rep.as<uint32_t>(), (uint32_t)ustar, less, cu, cl, h8.as<uint32_t>());
      hipLaunchKernelGGL(k_po_fold, dim3(16), dim3(256), 0, stream, h8.as<uint32_t>(), 4096, h2.as<uint32_t>());
      {
        HostRead hr_(stream);
        TM_TRY(hr_.get(hh2.data(), h2.p, 4096 * 4));
        TM_TRY(hr_.wait());
      }
      int64_t ncand = above;
      int bstar = 4095;
      for (int b = 0; b < 4096; b++) {
        ncand += hh2[(size_t)b];
        if (ncand >= exact_first) { bstar = b; break; }
      }
      TM_TRY(flag.alloc((size_t)nu * 4)); TM_TRY(fpos.alloc((size_t)nu * 4));
      hipLaunchKernelGGL(k_po_flags, dim3(gridn(nu)), dim3(256), 0, stream, uniq.as<uint32_t>(), nu, use_rep.as<uint32_t>(), (uint32_t)ustar, (uint32_t)bstar, less, cu, cl, flag.as<uint32_t>());
      size_t tbs = 0;
      TM_HIP(rocprim::exclusive_scan(nullptr, tbs, flag.as<uint32_t>(), fpos.as<uint32_t>(), 0u, (size_t)nu, rocprim::plus<uint32_t>(), stream));
      TM_TRY(tmp.alloc(tbs));
      TM_HIP(rocprim::exclusive_scan(tmp.p, tbs, flag.as<uint32_t>(), fpos.as<uint32_t>(), 0u, (size_t)nu, rocprim::plus<uint32_t>(), stream));
      DevBuf pk;
      TM_TRY(pk.alloc((size_t)ncand * sizeof(PrefixKey)));
      hipLaunchKernelGGL(k_po_split, dim3(gridn(nu)), dim3(256), 0, stream, uniq.as<uint32_t>(), nu, flag.as<uint32_t>(), fpos.as<uint32_t>(), ncand, use_rep.as<uint32_t>(), less, cu, cl,
                         pk.as<PrefixKey>(), ord2.as<uint32_t>());
      TM_TRY(sort_prefix_keys(pk, ncand, less, ord2.as<uint32_t>(), tmp, stream));
      TM_HIP(hipGetLastError());
      ranked = true;
      live = (unsigned long long)nu;
    }
  }
  if (!ranked && grouped && !by_index) {  // the distinct rows, now in hash order -> content order (what the stable ranking sort below relies on; the
                               // by-index form ranks by row number alone: any order of the distinct rows will do)
    DevBuf pk;
    TM_TRY(pk.alloc((size_t)nu * sizeof(PrefixKey)));
    hipLaunchKernelGGL(k_prefix_keys, dim3(gridn(nu)), dim3(256), 0, stream, uniq.as<uint32_t>(), nu, less, pk.as<PrefixKey>());
    TM_TRY(sort_prefix_keys(pk, nu, less, uniq.as<uint32_t>(), tmp, stream));
    // pk / pk2 go back to the pool here; later users are ordered behind these kernels on the same stream (as with `tmp`)
  }
  if (!ranked) {
  TM_HIP(hipMemsetAsync(cnt.p, 0, 16, stream));
  hipLaunchKernelGGL(k_rank_keys, dim3(gridn(nu)), dim3(256), 0, stream, uniq.as<uint32_t>(), nu, use_rep.as<uint32_t>(), by_index,
                     key.as<uint32_t>(), cnt.as<unsigned long long>());
  size_t tb4 = 0;
  TM_HIP(rocprim::radix_sort_pairs(nullptr, tb4, key.as<uint32_t>(), key2.as<uint32_t>(), uniq.as<uint32_t>(), ord2.as<uint32_t>(),
                                   (size_t)nu, 0, 32, stream));
  TM_TRY(tmp.alloc(tb4));
  TM_HIP(rocprim::radix_sort_pairs(tmp.p, tb4, key.as<uint32_t>(), key2.as<uint32_t>(), uniq.as<uint32_t>(), ord2.as<uint32_t>(),
                                   (size_t)nu, 0, 32, stream));
  {
    HostRead hr_(stream);
    TM_TRY(hr_.get(&live, cnt.p, 8));
    TM_TRY(hr_.wait());
  }
  }
  TM_HIP(hipMemsetAsync(pos.p, 0xff, n * 4, stream));
  hipLaunchKernelGGL(k_scatter_pos, dim3(gridn((int64_t)live)), dim3(256), 0, stream, ord2.as<uint32_t>(), (int64_t)live,
                     use_rep.as<uint32_t>(), pos.as<int32_t>(), (uint32_t *)use_out);
  hipLaunchKernelGGL(k_remap, dim3(gridn(n)), dim3(256), 0, stream, rep.as<uint32_t>(), pos.as<int32_t>(), n, (int32_t *)remap);
  TM_HIP(hipMemcpyAsync(order, ord2.p, (size_t)live * 4, hipMemcpyDeviceToDevice, stream));
  TM_HIP(hipGetLastError());
  TM_HIP(hipStreamSynchronize(stream));  // the scratch DevBufs die with this frame
  if (host_n_unique) *host_n_unique = (int64_t)live;
  return TM_OK;
}

namespace {
__global__ void k_scatter_kept(const int32_t *__restrict__ keep, const uint32_t *__restrict__ pos, int64_t n, int32_t *__restrict__ out_idx) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (keep[i]) out_idx[pos[i]] = (int32_t)i;
}
}  // namespace

// indices of the flagged items in ascending order (TransferTiles' gather, tilingencoder.pas:4048-4103, made deterministic);
// pos[i] = rank of item i among the kept ones (valid where keep[i] != 0)
int compact_kept(const void *keep, int64_t n, void *out_idx, void *pos, int64_t *host_count, hipStream_t stream) {
  TM_CHECK(n >= 0 && n < (int64_t)1 << 31, TM_E_INVAL, "compact: count out of range");
  *host_count = 0;
  if (n == 0) return TM_OK;
  size_t tb = 0;
  DevBuf tmp;
  TM_HIP(rocprim::exclusive_scan(nullptr, tb, (const uint32_t *)keep, (uint32_t *)pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
  TM_TRY(tmp.alloc(tb));
  TM_HIP(rocprim::exclusive_scan(tmp.p, tb, (const uint32_t *)keep, (uint32_t *)pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
  hipLaunchKernelGGL(k_scatter_kept, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, stream, (const int32_t *)keep,
                     (const uint32_t *)pos, n, (int32_t *)out_idx);
  TM_HIP(hipGetLastError());
  uint32_t last_pos = 0;
  int32_t last_keep = 0;
  {
    HostRead hr_(stream);
    TM_TRY(hr_.get(&last_pos, (const uint32_t *)pos + (n - 1), 4));
    TM_TRY(hr_.get(&last_keep, (const int32_t *)keep + (n - 1), 4));
    TM_TRY(hr_.wait());
  }
  *host_count = (int64_t)last_pos + (last_keep ? 1 : 0);
  return TM_OK;
}

namespace {
__global__ void k_iota_u32(uint32_t *__restrict__ v, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}
}  // namespace

// member lists of a dedup: off[g] .. off[g+1] index `members`, which holds the rows of group g in ascending row order
// (remap = row -> group, counts = rows per group, as run_dedup_ex(by_index = 1) returns them)
int build_groups(const void *remap, int64_t n, const void *counts, int64_t ngroups, void *off, void *members, hipStream_t stream) {
  TM_CHECK(n >= 0 && n < (int64_t)1 << 31 && ngroups >= 0, TM_E_INVAL, "groups: count out of range");
  if (n == 0) return TM_OK;
  DevBuf tmp, keys_out, iota;
  size_t tb = 0;
  TM_HIP(rocprim::exclusive_scan(nullptr, tb, (const uint32_t *)counts, (uint32_t *)off, 0u, (size_t)ngroups + 1, rocprim::plus<uint32_t>(), stream));
  TM_TRY(tmp.alloc(tb));
  // counts has ngroups entries; the scan's extra input element is read but its value never reaches an output we use
  TM_HIP(rocprim::exclusive_scan(tmp.p, tb, (const uint32_t *)counts, (uint32_t *)off, 0u, (size_t)ngroups, rocprim::plus<uint32_t>(), stream));
  const uint32_t total = (uint32_t)n;
  TM_HIP(hipMemcpyAsync((uint32_t *)off + ngroups, &total, 4, hipMemcpyHostToDevice, stream));
  TM_TRY(keys_out.alloc((size_t)n * 4)); TM_TRY(iota.alloc((size_t)n * 4));
  hipLaunchKernelGGL(k_iota_u32, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, stream, iota.as<uint32_t>(), n);
  size_t tb2 = 0;
  TM_HIP(rocprim::radix_sort_pairs(nullptr, tb2, (const uint32_t *)remap, keys_out.as<uint32_t>(), iota.as<uint32_t>(), (uint32_t *)members, (size_t)n, 0,
                                   32, stream));
  TM_TRY(tmp.alloc(tb2));
  TM_HIP(rocprim::radix_sort_pairs(tmp.p, tb2, (const uint32_t *)remap, keys_out.as<uint32_t>(), iota.as<uint32_t>(), (uint32_t *)members, (size_t)n, 0, 32,
                                   stream));
  TM_HIP(hipGetLastError());
  TM_HIP(hipStreamSynchronize(stream));  // `total` is on the stack; temporaries are released on return
  return TM_OK;
}

// ---- Reduce over several processes: which distinct tiles have to travel ---------------------------------------------------------------
// Every process has deduplicated its own frame tiles; the merged order (use count descending, content ascending) keeps the first `target`
// tiles.  Instead of all-gathering every process's distinct tiles (857 MB on the 720p bench clip for 321 k survivors) the processes
// exchange a 16-byte KEY per distinct tile -- a 64-bit content hash, the first dword of the content (its leading bytes in comparison
// order) and the local use count -- and every process runs this selection on the gathered keys (identical input, identical result):
//   * keys whose hash no other key shares are SINGLES: no other process holds that content, the local use count is the true one;
//   * keys that share their hash form a group: duplicates of one tile across processes (or a hash collision -- the groups are never
//     trusted to be equal content, they only decide what travels); the group's summed use bounds every member's true use from above;
//   * the singles ordered by (use descending, leading dword ascending): the key at position `target` is the cut-off -- every single
//     beyond it (strictly) has `target` tiles before it in the true order whatever the groups turn out to be, and so has every member
//     of a group whose SUM stays below the cut-off's use count.  Everything else is a candidate.
// The candidates' tiles (a superset of the true first `target`, whole groups always) are then all-gathered and deduplicated exactly,
// full compares and all, as the union was before.
struct ReduceKey { unsigned long long hash; uint32_t prefix, use; };

namespace {
__global__ __launch_bounds__(256) void k_reduce_keys(const uint32_t *__restrict__ rows, const int32_t *__restrict__ idx, const uint32_t *__restrict__ use, int64_t n,
                                                     int dwords, int degrade /* test hook: force hash collisions */, ReduceKey *__restrict__ out) {
  const int sub = threadIdx.x & 15;
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t r0 = blockIdx.x * (int64_t)16; r0 < n; r0 += stride) {  // k_row_hash's terms, through an index
    const int64_t r = r0 + (threadIdx.x >> 4);
    const int64_t row = r < n ? (int64_t)idx[r] : 0;
    unsigned long long h = 0;
    if (r < n)
      for (int v = sub; v < dwords / 4; v += 16) {
        const uint4 x = *reinterpret_cast<const uint4 *>(rows + row * dwords + v * 4);
        unsigned long long a = ((unsigned long long)x.y << 32 | x.x) + 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * v + 1);
        unsigned long long b = ((unsigned long long)x.w << 32 | x.z) + 0xC2B2AE3D27D4EB4Full * (unsigned long long)(2 * v + 2);
        a ^= a >> 32; a *= 0xD6E8FEB86659FD93ull; a ^= a >> 32;
        b ^= b >> 29; b *= 0xBF58476D1CE4E5B9ull; b ^= b >> 32;
        h += a * 0x94D049BB133111EBull + b;
      }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) h += __shfl_xor(h, o);
    if (r < n && sub == 0) out[r] = ReduceKey{degrade ? (h & 3) : h, rows[row * dwords], use[r]};
  }
}
}  // namespace

namespace {
__global__ void k_rk_split(const ReduceKey *__restrict__ keys, int64_t n, unsigned long long *__restrict__ hash, uint32_t *__restrict__ idx) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { hash[i] = keys[i].hash; idx[i] = (uint32_t)i; }
}
__global__ void k_rk_heads(const unsigned long long *__restrict__ hs, int64_t n, uint32_t *__restrict__ head) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) head[j] = (j == 0 || hs[j] != hs[j - 1]) ? 1u : 0u;
}
// gid = inclusive scan of the heads - 1; every key adds itself to its group
__global__ void k_rk_groups(const uint32_t *__restrict__ sorted_idx, const uint32_t *__restrict__ head_incl, const ReduceKey *__restrict__ keys, int64_t n,
                            uint32_t *__restrict__ gid_of, unsigned long long *__restrict__ gsum, uint32_t *__restrict__ gsize) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t g = head_incl[j] - 1u, i = sorted_idx[j];
    gid_of[i] = g;
    atomicAdd(&gsum[g], (unsigned long long)keys[i].use);
    atomicAdd(&gsize[g], 1u);
  }
}
__global__ void k_rk_single_keys(const ReduceKey *__restrict__ keys, const uint32_t *__restrict__ gid_of, const uint32_t *__restrict__ gsize, int64_t n,
                                 unsigned long long *__restrict__ skey, unsigned long long *__restrict__ nsingles) {
  unsigned long long local = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool single = gsize[gid_of[i]] == 1;
    skey[i] = single ? ((unsigned long long)(~keys[i].use) << 32) | keys[i].prefix : ~0ull;
    local += single ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(nsingles, local);
}
__global__ void k_rk_select(const unsigned long long *__restrict__ skey, const uint32_t *__restrict__ gid_of, const unsigned long long *__restrict__ gsum, int64_t n,
                            unsigned long long cutoff, unsigned long long min_use, uint32_t *__restrict__ in_s) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    in_s[i] = skey[i] != ~0ull ? (skey[i] <= cutoff ? 1u : 0u) : (gsum[gid_of[i]] >= min_use ? 1u : 0u);
}
}  // namespace

int reduce_make_keys(const void *rows, const void *idx, const void *use, int64_t n, int row_bytes, void *keys_out, hipStream_t stream) {
  if (n <= 0) return TM_OK;
  hipLaunchKernelGGL(k_reduce_keys, dim3((unsigned)std::min<int64_t>((n + 15) / 16, 256 * 32)), dim3(256), 0, stream, (const uint32_t *)rows, (const int32_t *)idx,
                     (const uint32_t *)use, n, row_bytes / 4, knobs().dedup_degrade_hash ? 1 : 0, (ReduceKey *)keys_out);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

int reduce_select_candidates(const void *keys_v, int64_t n, int64_t target, void *in_s, hipStream_t stream) {
  TM_CHECK(n > 0 && n < (int64_t)1 << 31, TM_E_INVAL, "reduce: key count out of range");
  const ReduceKey *keys = (const ReduceKey *)keys_v;
  DevBuf hash, hash2, idx, idx2, head, head_incl, gid_of, gsum, gsize, skey, skey2, tmp, cnt;
  TM_TRY(hash.alloc(n * 8)); TM_TRY(hash2.alloc(n * 8)); TM_TRY(idx.alloc(n * 4)); TM_TRY(idx2.alloc(n * 4)); TM_TRY(head.alloc(n * 4)); TM_TRY(head_incl.alloc(n * 4));
  TM_TRY(gid_of.alloc(n * 4)); TM_TRY(gsum.alloc(n * 8)); TM_TRY(gsize.alloc(n * 4)); TM_TRY(skey.alloc(n * 8)); TM_TRY(skey2.alloc(n * 8)); TM_TRY(cnt.alloc(8));
  hipLaunchKernelGGL(k_rk_split, dim3(gridn(n)), dim3(256), 0, stream, keys, n, hash.as<unsigned long long>(), idx.as<uint32_t>());
  size_t tb = 0;
  TM_HIP(rocprim::radix_sort_pairs(nullptr, tb, hash.as<unsigned long long>(), hash2.as<unsigned long long>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), (size_t)n, 0, 64, stream));
  TM_TRY(tmp.alloc(tb));
  TM_HIP(rocprim::radix_sort_pairs(tmp.p, tb, hash.as<unsigned long long>(), hash2.as<unsigned long long>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), (size_t)n, 0, 64, stream));
  hipLaunchKernelGGL(k_rk_heads, dim3(gridn(n)), dim3(256), 0, stream, hash2.as<unsigned long long>(), n, head.as<uint32_t>());
  size_t tb2 = 0;
  TM_HIP(rocprim::inclusive_scan(nullptr, tb2, head.as<uint32_t>(), head_incl.as<uint32_t>(), (size_t)n, rocprim::plus<uint32_t>(), stream));
  TM_TRY(tmp.alloc(tb2));
  TM_HIP(rocprim::inclusive_scan(tmp.p, tb2, head.as<uint32_t>(), head_incl.as<uint32_t>(), (size_t)n, rocprim::plus<uint32_t>(), stream));
  TM_HIP(hipMemsetAsync(gsum.p, 0, n * 8, stream));
  TM_HIP(hipMemsetAsync(gsize.p, 0, n * 4, stream));
  TM_HIP(hipMemsetAsync(cnt.p, 0, 8, stream));
  hipLaunchKernelGGL(k_rk_groups, dim3(gridn(n)), dim3(256), 0, stream, idx2.as<uint32_t>(), head_incl.as<uint32_t>(), keys, n, gid_of.as<uint32_t>(),
                     gsum.as<unsigned long long>(), gsize.as<uint32_t>());
  hipLaunchKernelGGL(k_rk_single_keys, dim3(gridn(n)), dim3(256), 0, stream, keys, gid_of.as<uint32_t>(), gsize.as<uint32_t>(), n, skey.as<unsigned long long>(),
                     cnt.as<unsigned long long>());
  unsigned long long nsingles = 0;
  {
    HostRead hr_(stream);
    TM_TRY(hr_.get(&nsingles, cnt.p, 8));
    TM_TRY(hr_.wait());
  }
  unsigned long long cutoff = ~0ull - 1ull, min_use = 0;  // fewer singles than the budget: everything travels
  if (target > 0 && (unsigned long long)target <= nsingles) {
    size_t tb3 = 0;
    TM_HIP(rocprim::radix_sort_keys(nullptr, tb3, skey.as<unsigned long long>(), skey2.as<unsigned long long>(), (size_t)n, 0, 64, stream));
    TM_TRY(tmp.alloc(tb3));
    TM_HIP(rocprim::radix_sort_keys(tmp.p, tb3, skey.as<unsigned long long>(), skey2.as<unsigned long long>(), (size_t)n, 0, 64, stream));
    {
      HostRead hr_(stream);
      TM_TRY(hr_.get(&cutoff, skey2.as<unsigned long long>() + (target - 1), 8));
      TM_TRY(hr_.wait());
    }
    min_use = (unsigned long long)(uint32_t)(~(uint32_t)(cutoff >> 32));
  }
  hipLaunchKernelGGL(k_rk_select, dim3(gridn(n)), dim3(256), 0, stream, skey.as<unsigned long long>(), gid_of.as<uint32_t>(), gsum.as<unsigned long long>(), n, cutoff, min_use,
                     (uint32_t *)in_s);
  TM_HIP(hipGetLastError());
  TM_HIP(hipStreamSynchronize(stream));  // the scratch DevBufs die with this frame
  return TM_OK;
}

int run_dedup(const void *rows, int64_t n, int row_bytes, const void *use_in, void *remap, void *order, void *use_out,
              int64_t *host_n_unique, hipStream_t stream, int64_t exact_first) {
  return run_dedup_ex(rows, n, row_bytes, use_in, remap, order, use_out, host_n_unique, 0, stream, exact_first);
}

}  // namespace tmx
