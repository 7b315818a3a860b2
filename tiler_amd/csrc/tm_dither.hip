// tm_dither.hip -- positional (ordered) dithering of global tiles, A12.
//
// Restates Dither (tilingencoder.pas:1873-1907) = PreparePlan (2268-2301) + DitherTile (2688-2724) +
// DeviseBestMixingPlanThomasKnoll (2565-2612) + ColorCompare (2323-2337) + the byte QuickSort of extern.pas:370-418.
// One wave (= one workgroup) per tile, one lane per pixel; every lane runs its own 64-step error-feedback plan and
// then sorts its own 64 picks by luma with a literal emulation of the reference's unstable quicksort (explicit
// stack, pivot value held constant because the reference tracks the pivot element through swaps), so equal-luma
// picks land exactly where the reference puts them.  Integer only.  All values fit int32: |e| <= 64*255,
// |t| <= 255 + 16320*9/100, penalty < 2^29 (the reference uses Int64).
#include <algorithm>
#include <climits>
#include <cstdlib>

#include <vector>

#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {

__device__ __forceinline__ int div_trunc_1000(int v) { return v / 1000; }  // Pascal div: toward zero, like C
// full-rate 24-bit multiplies (32-bit v_mul_lo_u32 is quarter rate); callers guarantee |operands| < 2^23.  Inline assembly because the
// compiler re-associates __mul24(a, b) + c chains into separate multiplies and three-operand adds (and 32-bit multiplies for squares)
__device__ __forceinline__ int mul24(int a, int b) { int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int mad24(int a, int b, int c) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int mad24s(int a, int b, int c) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c)); return r; }  // b wave-uniform

// QuickSort(List[0], 0, last, 1, PlanCompareLuma) of extern.pas:370-418 on one lane's list, iterative form: explicit
// stack for the "recurse left, loop right" shape; the pivot VALUE is constant during a partition pass because the
// reference tracks the pivot element through swaps.  Entries are rank << 8 | plan index; only the rank is compared.
__device__ __forceinline__ void lane_quicksort(uint16_t (*s_list)[64], uint16_t (*s_stack)[64], int lane, int last_in) {
  if (last_in <= 0) return;
  int first = 0, last = last_in, sp = 0;
  while (true) {
    int i = first, j = last;
    const int pv = s_list[(first + last) >> 1][lane] >> 8;
    do {
      while ((s_list[i][lane] >> 8) < pv) i++;
      while ((s_list[j][lane] >> 8) > pv) j--;
      if (i <= j) {
        const uint16_t a = s_list[i][lane], b = s_list[j][lane];
        s_list[i][lane] = b;
        s_list[j][lane] = a;
        i++;
        j--;
      }
    } while (i <= j);
    if (first < j) {  // recurse left, remember (i, last) for the loop tail
      s_stack[sp++][lane] = (uint16_t)((i << 8) | last);
      last = j;
      continue;
    }
    bool done = false;  // first := i; until i >= last
    while (i >= last) {
      if (sp == 0) { done = true; break; }
      const uint16_t fr = s_stack[--sp][lane];
      i = fr >> 8;
      last = fr & 0xff;
    }
    if (done) break;
    first = i;
  }
}

__global__ __launch_bounds__(64) void k_dither_tk(const uint32_t *__restrict__ tiles, const uint8_t *__restrict__ flags,
                                                  const int32_t *__restrict__ pal_idx, int64_t n, const int32_t *__restrict__ palettes,
                                                  int npal, int pal_size, const uint8_t *__restrict__ cls /* null, or 1 = k_dither_tk_fast's */,
                                                  const uint8_t *__restrict__ dither_map, uint8_t *__restrict__ out) {
  __shared__ int4 s_plan[64];          // r, g, b, luma of live entries (Y2Palette / LumaPal)
  __shared__ uint8_t s_rank[64];       // #entries with strictly smaller luma: order-isomorphic to LumaPal incl. ties
  __shared__ uint8_t s_remap[64];      // Plan.Remap
  __shared__ uint16_t s_list[64][64];  // [position][lane] = rank<<8 | plan index
  __shared__ uint16_t s_stack[64][64]; // [depth][lane] = i<<8 | last
  const int lane = threadIdx.x;
  const int map_value = dither_map[lane];  // cDitheringMap[((y and 7) shl 3) or (x and 7)], natural orientation
  int cached_pal = -1, cnt = 0;
  // 64 tiles looked at per pass (a lane each), the ones that are this kernel's taken one by one: with the counting kernel's class being the
  // rule, a wave that reads one palette index per pass spends the launch waiting for 300 000 dependent loads
  for (int64_t base = (int64_t)blockIdx.x * 64; base < n; base += (int64_t)gridDim.x * 64) {
   const int64_t tl = base + lane;
   const int pl = tl < n ? pal_idx[tl] : 0;
   unsigned long long todo = __ballot(tl < n && !(cls && pl >= 0 && pl < npal && cls[pl]));
   while (todo) {
    const int64_t t = base + (__ffsll((long long)todo) - 1);
    todo &= todo - 1;
    const int pi = pal_idx[t];
    if (pi != cached_pal) {  // PreparePlan: drop cDitheringNullColor entries, keep order
      __syncthreads();
      int col = TM_NULL_COLOR;
      if (lane < pal_size && pi >= 0 && pi < npal) col = palettes[(int64_t)pi * pal_size + lane];
      const bool live = col != TM_NULL_COLOR;
      const unsigned long long m = __ballot(live);
      cnt = __popcll(m);
      const int pos = __popcll(m & ((1ull << lane) - 1ull));
      const int r = col & 0xff, g = (col >> 8) & 0xff, b = (col >> 16) & 0xff;
      const int luma = r * 299 + g * 587 + b * 114;
      if (live) {
        s_plan[pos] = make_int4(r, g, b, luma);
        s_remap[pos] = (uint8_t)lane;
      }
      __syncthreads();
      if (lane < cnt) {
        const int my = s_plan[lane].w;
        int rk = 0;
        for (int i = 0; i < cnt; i++) rk += (s_plan[i].w < my) ? 1 : 0;
        s_rank[lane] = (uint8_t)rk;
      }
      __syncthreads();
      cached_pal = pi;
    }
    if (cnt == 0) {  // a palette with no colours cannot own tiles in the reference (would divide by zero at 2589)
      out[t * 64 + lane] = 0;
      continue;
    }
    const int f = flags ? flags[t] : 0;
    const int y = lane >> 3, x = lane & 7;
    const int src = (((f & 2) ? 7 - y : y) << 3) | ((f & 1) ? 7 - x : x);  // un-mirror (2696-2697)
    const uint32_t c = tiles[t * 64 + src];
    const int s0 = c & 0xff, s1 = (c >> 8) & 0xff, s2 = (c >> 16) & 0xff;
    int e0 = 0, e1 = 0, e2 = 0;
    for (int k = 0; k < 64; k++) {
      const int t0 = s0 + (e0 * 9) / 100, t1 = s1 + (e1 * 9) / 100, t2 = s2 + (e2 * 9) / 100;
      const int lt = t0 * 299 + t1 * 587 + t2 * 114;
      int least = INT_MAX, chosen = 0;
      for (int i = 0; i < cnt; i++) {
        const int4 p = s_plan[i];
        const int dr = t0 - p.x, dg = t1 - p.y, db = t2 - p.z;
        const int ld = div_trunc_1000(lt - p.w);
        const int pen = (dr * dr + dg * dg + db * db) * 13 + ((ld * ld) << 5);
        if (pen < least) { least = pen; chosen = i; }
      }
      s_list[k][lane] = (uint16_t)((s_rank[chosen] << 8) | chosen);
      const int4 p = s_plan[chosen];
      e0 += s0 - p.x; e1 += s1 - p.y; e2 += s2 - p.z;
    }
    lane_quicksort(s_list, s_stack, lane, 63);
    const int pick = s_list[map_value][lane] & 0xff;
    out[t * 64 + src] = s_remap[pick];  // re-mirror (2721-2722): natural (y,x) lives at canonical position src
   }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Fast Thomas-Knoll path for the common palette shape: 1..16 live colours whose lumas are pairwise distinct.
//  * The plan is wave-uniform, so it lives in SGPRs (16 x r,g,b,luma) and the 16-way search is fully unrolled; the
//    argmin is one v_min_i32 per entry on (16 penalty + index) - 208 |t|^2: the lowest index wins equal penalties exactly like
//    the reference's strict `<` scan in plan order (2597-2605), and the term common to all entries never has to be computed:
//    13 |t - p|^2 - 13 |t|^2 = 13 |p|^2 - 26 t.p is three multiply-adds with per-entry constants.  Range: the error entering
//    step k is at most 63*255 in magnitude, so t is in [-1445, 1700] per channel, 208 |t|^2 <= 1.81e9, and
//    208 (|p|^2 - 2 t.p) + 512 ld^2 + 15 <= 5.0e8 + 1.48e9: every value compared fits int32.
//  * With distinct lumas the unstable QuickSort of the 64 picks (2611) has only one possible outcome -- picks ordered
//    by luma, equal picks being the same byte -- so the lane does not sort: it counts its picks per luma rank and
//    reads position cDitheringMap[..] of the sorted list off the running totals.
// Palettes with a luma tie between different colours, or more than 16 live colours, take k_dither_tk (literal sort).
// position map_value of a pixel's 64 picks in luma order, from the counts per luma rank (8 bits each)
__device__ __forceinline__ int rank_at(const uint4 bins, int map_value) {
  int acc = 0, pick_rank = 0;
  bool found = false;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const uint32_t w = (r >> 2) == 0 ? bins.x : (r >> 2) == 1 ? bins.y : (r >> 2) == 2 ? bins.z : bins.w;
    acc += (int)((w >> ((r & 3) * 8)) & 0xff);
    if (!found && acc > map_value) { pick_rank = r; found = true; }
  }
  return pick_rank;
}

__global__ void k_palette_class(const int32_t *__restrict__ palettes, int npal, int pal_size, uint8_t *__restrict__ cls) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npal) return;
  int luma[64], cnt = 0;
  for (int i = 0; i < pal_size; i++) {
    const int col = palettes[(int64_t)p * pal_size + i];
    if (col == TM_NULL_COLOR) continue;
    luma[cnt++] = (col & 0xff) * 299 + ((col >> 8) & 0xff) * 587 + ((col >> 16) & 0xff) * 114;
  }
  bool fast = cnt >= 1 && cnt <= 16;
  for (int i = 0; fast && i < cnt; i++)
    for (int j = i + 1; j < cnt; j++)
      if (luma[i] == luma[j]) fast = false;
  cls[p] = fast ? 1 : 0;
}

// UNIQ: the same plan for a list of distinct (palette, colour) pairs instead of tiles -- a pixel's 64 picks depend on its colour and its
// palette only, the position in the tile just selects one of them -- in chunks of 64 colours of one palette; the counts per luma
// rank (16 bytes) go to dd.bins, the palette's rank -> slot table to dd.rs, and k_dd_lookup finishes every pixel (see launch_dither).
struct DdArgs {
  const uint32_t *ucol;         // [nu] distinct colours, grouped by palette
  const int32_t *chunk_pal;     // [nchunks]
  const uint32_t *chunk_begin;  // [nchunks] first colour of the chunk
  const uint32_t *chunk_end;    // [nchunks] end of the chunk's palette segment
  uint4 *bins;                  // [nu]
  uint8_t *rs;                  // [npal][16]
};
template <bool UNIQ>
__global__ __launch_bounds__(64) void k_dither_tk_fast(const uint32_t *__restrict__ tiles, const uint8_t *__restrict__ flags,
                                                       const int32_t *__restrict__ pal_idx, int64_t n, const int32_t *__restrict__ palettes,
                                                       int npal, int pal_size, const uint8_t *__restrict__ cls,
                                                       const uint8_t *__restrict__ dither_map, uint8_t *__restrict__ out, DdArgs dd) {
  __shared__ int4 s_plan[16];      // r, g, b, luma rank
  __shared__ int s_luma[16];
  __shared__ uint8_t s_remap[16];  // plan index -> palette slot
  __shared__ uint8_t s_by_rank[16];  // luma rank -> plan index
  __shared__ uint4 s_inc[16];        // plan index -> 1 in byte (luma rank) of 128 bits
  const int lane = threadIdx.x;
  const int map_value = dither_map[lane];
  int pr[16], pg[16], pb[16], pk[16];  // -416 r, -416 g, -416 b (wave-uniform: scalar registers), 208 (r^2 + g^2 + b^2) + i
  float pl[16];                         // luma
  int cached_pal = -1;
  for (int64_t t = blockIdx.x; t < n; t += gridDim.x) {  // n: tiles, or chunks of distinct colours
    const int pi = UNIQ ? dd.chunk_pal[t] : pal_idx[t];
    if (pi < 0 || pi >= npal || !cls[pi]) continue;  // wave-uniform
    if (pi != cached_pal) {  // PreparePlan (2268-2301): drop cDitheringNullColor entries, keep order
      __syncthreads();
      int col = TM_NULL_COLOR;
      if (lane < pal_size) col = palettes[(int64_t)pi * pal_size + lane];
      const bool live = col != TM_NULL_COLOR;
      const unsigned long long m = __ballot(live);
      const int cnt = __popcll(m);
      const int pos = __popcll(m & ((1ull << lane) - 1ull));
      const int r = col & 0xff, g = (col >> 8) & 0xff, b = (col >> 16) & 0xff;
      if (live) {
        s_plan[pos] = make_int4(r, g, b, 0);
        s_luma[pos] = r * 299 + g * 587 + b * 114;
        s_remap[pos] = (uint8_t)lane;
      }
      __syncthreads();
      if (lane < cnt) {
        const int my = s_luma[lane];
        int rk = 0;
        for (int i = 0; i < cnt; i++) rk += (s_luma[i] < my) ? 1 : 0;
        s_plan[lane].w = rk;
        s_by_rank[rk] = (uint8_t)lane;
        const uint32_t one = 1u << ((rk & 3) * 8);
        s_inc[lane] = make_uint4((rk >> 2) == 0 ? one : 0u, (rk >> 2) == 1 ? one : 0u, (rk >> 2) == 2 ? one : 0u, (rk >> 2) == 3 ? one : 0u);
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; i++) {  // entries past the live count repeat entry 0: same penalty, higher index, never chosen
        const int j = i < cnt ? i : 0;
        const int4 p = s_plan[j];
        pr[i] = __builtin_amdgcn_readfirstlane(-416 * p.x);
        pg[i] = __builtin_amdgcn_readfirstlane(-416 * p.y);
        pb[i] = __builtin_amdgcn_readfirstlane(-416 * p.z);
        pk[i] = 208 * (p.x * p.x + p.y * p.y + p.z * p.z) + i;
        pl[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)s_luma[j])));
      }
      if (UNIQ && lane < 16) dd.rs[pi * 16 + lane] = lane < cnt ? s_remap[s_by_rank[lane]] : 0;  // (every chunk of the palette writes the same bytes)
      cached_pal = pi;
    }
    int src = 0;
    uint32_t c = 0;
    uint32_t u = 0;
    bool valid = true;
    if (UNIQ) {
      u = dd.chunk_begin[t] + lane;
      valid = u < dd.chunk_end[t];
      c = valid ? dd.ucol[u] : 0u;
    } else {
      const int f = flags ? flags[t] : 0;
      const int y = lane >> 3, x = lane & 7;
      src = (((f & 2) ? 7 - y : y) << 3) | ((f & 1) ? 7 - x : x);  // un-mirror (2696-2697)
      c = tiles[t * 64 + src];
    }
    const int s0 = c & 0xff, s1 = (c >> 8) & 0xff, s2 = (c >> 16) & 0xff;
    int e0 = 0, e1 = 0, e2 = 0;
    uint32_t bins0 = 0, bins1 = 0, bins2 = 0, bins3 = 0;  // 8 bits per luma rank: how many of the 64 picks have that rank (at most 64: no carry)
    for (int k = 0; k < 64; k++) {
      // (e * 9) div 100 = trunc(float(e) * 0.09f) for |e| <= 16 400 (every value checked, tests/test_host_logic.py): three full-rate
      // instructions instead of a quarter-rate v_mul_hi and its fix-ups
      const int t0 = s0 + (int)((float)e0 * 0.09f), t1 = s1 + (int)((float)e1 * 0.09f), t2 = s2 + (int)((float)e2 * 0.09f);
      const int lt = mad24(t2, 114, mad24(t1, 587, mul24(t0, 299)));
      const float ltf = (float)lt;  // |lt| < 2^21: exact, and so is its difference with a luma
      int best = INT_MAX;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        // 16 * 13 (|p|^2 - 2 t.p) + i: all factors fit 24 bits (|t| <= 1700, 416 p <= 106 080): full-rate v_mad_i32_i24 instead of
        // quarter-rate 32-bit multiplies
        const int x = mad24s(t2, pb[i], mad24s(t1, pg[i], mad24s(t0, pr[i], pk[i])));
        // |(lt - luma) div 1000| = floor(|lt - luma| / 1000): exact as trunc(fma(a, 0.001f, 0.0005f)) for a < 2^22 (both ends
        // of every thousand checked in exact arithmetic, monotone in between); only its square is used (|.| is a source modifier)
        const int ld = (int)__builtin_fmaf(__builtin_fabsf(ltf - pl[i]), 0.001f, 0.0005f);
        best = min(best, mad24(mul24(ld, ld), 512, x));  // + 16 * 32 ld^2 (ld^2 < 2^22)
      }
      const int4 p = s_plan[best & 15u];
      e0 += s0 - p.x; e1 += s1 - p.y; e2 += s2 - p.z;
      const uint4 inc = s_inc[best & 15];
      bins0 += inc.x; bins1 += inc.y; bins2 += inc.z; bins3 += inc.w;
    }
    if (UNIQ) {
      if (valid) dd.bins[u] = make_uint4(bins0, bins1, bins2, bins3);
      continue;
    }
    out[t * 64 + src] = s_remap[s_by_rank[rank_at(make_uint4(bins0, bins1, bins2, bins3), map_value)]];  // re-mirror (2721-2722)
  }
}

// ---- duplicate pixels ---------------------------------------------------------------------------------------------------------
// Video tiles repeat colours: on the bench clip 20.5 M pixels of the global tiles hold 2.3 M distinct (palette, colour) pairs.  The
// distinct pairs are found without sorting: a bitmap over (palette, G, R) x B, one bit per pair seen (k_dd_mark), a popcount per
// (palette, G, R) entry and an exclusive scan give every pair its index = scan[entry] + (set bits below B) -- pairs grouped by palette,
// (G, R, B) ascending inside --, k_dd_expand lists the pairs, k_dither_tk_fast<true> plans each once, k_dd_lookup gives every pixel
// its pair's counts and reads the position cDitheringMap[..] names off them.  Only palettes of the counting kernel's class take part.
constexpr int DD_MAX_PAL = 4096;  // 2.25 MB of table per palette, and no more table entries than four times the pixels (the table is cleared, counted and scanned per call)
__global__ void k_dd_mark(const uint32_t *__restrict__ tiles, const int32_t *__restrict__ pal_idx, int64_t n, int npal, const uint8_t *__restrict__ cls,
                          uint32_t *__restrict__ bits) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * 64; i += (int64_t)gridDim.x * blockDim.x) {
    const int pi = pal_idx[i >> 6];
    if (pi < 0 || pi >= npal || !cls[pi]) continue;
    const uint32_t c = tiles[i];
    const uint32_t e = ((uint32_t)pi << 16) | (c & 0xff00u) | (c & 0xffu), b = (c >> 16) & 0xff;  // entry = palette, G, R
    const uint32_t m = 1u << (b & 31);
    uint32_t *w = bits + (size_t)e * 8 + (b >> 5);
    if (!(*w & m)) atomicOr(w, m);  // (a stale read only repeats the atomic)
  }
}
// the same table from a list of distinct pixel keys (palette << 24 | G << 16 | R << 8 | B: QuantizeUsingYakmo's, tm_kmeans.hip)
__global__ void k_dd_mark_keys(const unsigned long long *__restrict__ keys, int64_t nk, int npal, const uint8_t *__restrict__ cls, uint32_t *__restrict__ bits) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nk; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long k = keys[i];
    const long long pi = (long long)(k >> 24);
    if (pi >= npal || !cls[pi]) continue;
    const uint32_t e = ((uint32_t)pi << 16) | (uint32_t)((k >> 8) & 0xffff), b = (uint32_t)(k & 0xff);
    atomicOr(bits + (size_t)e * 8 + (b >> 5), 1u << (b & 31));
  }
}
__global__ void k_dd_count(const uint32_t *__restrict__ bits, int64_t nent, uint32_t *__restrict__ cnt) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e <= nent; e += (int64_t)gridDim.x * blockDim.x) {
    uint32_t c = 0;
    if (e < nent) {
      const uint4 a = reinterpret_cast<const uint4 *>(bits)[e * 2], b = reinterpret_cast<const uint4 *>(bits)[e * 2 + 1];
      c = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
    }
    cnt[e] = c;  // one element more than entries: its scan value is the number of pairs
  }
}
__global__ void k_dd_seg(const uint32_t *__restrict__ off, int npal, uint32_t *__restrict__ seg) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p <= npal) seg[p] = off[(size_t)p << 16];
}
__global__ void k_dd_expand(const uint32_t *__restrict__ bits, const uint32_t *__restrict__ off, int64_t nent, uint32_t *__restrict__ ucol) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < nent; e += (int64_t)gridDim.x * blockDim.x) {
    uint32_t o = off[e];
    if (off[e + 1] == o) continue;
    const uint32_t gr = (uint32_t)(e & 0xffff);  // G << 8 | R: a pixel's low 16 bits
    for (int j = 0; j < 8; j++) {
      uint32_t w = bits[e * 8 + j];
      while (w) {
        const int b = __ffs(w) - 1;
        w &= w - 1;
        ucol[o++] = gr | ((uint32_t)(j * 32 + b) << 16);
      }
    }
  }
}
__global__ __launch_bounds__(256) void k_dd_lookup(const uint32_t *__restrict__ tiles, const uint8_t *__restrict__ flags, const int32_t *__restrict__ pal_idx,
                                                   int64_t n, int npal, const uint8_t *__restrict__ cls, const uint32_t *__restrict__ bits,
                                                   const uint32_t *__restrict__ off, const uint4 *__restrict__ bins, const uint8_t *__restrict__ rs,
                                                   const uint8_t *__restrict__ dither_map, uint8_t *__restrict__ out, int *__restrict__ missing) {
  const int lane = threadIdx.x & 63;
  const int map_value = dither_map[lane];
  for (int64_t t = blockIdx.x * 4 + (threadIdx.x >> 6); t < n; t += (int64_t)gridDim.x * 4) {
    const int pi = pal_idx[t];
    if (pi < 0 || pi >= npal || !cls[pi]) continue;
    const int f = flags ? flags[t] : 0;
    const int y = lane >> 3, x = lane & 7;
    const int src = (((f & 2) ? 7 - y : y) << 3) | ((f & 1) ? 7 - x : x);  // un-mirror (2696-2697)
    const uint32_t c = tiles[t * 64 + src];
    const size_t e = ((size_t)pi << 16) | (c & 0xffffu);
    const int b = (c >> 16) & 0xff, bw = b >> 5;
    const uint4 w0 = reinterpret_cast<const uint4 *>(bits)[e * 2], w1 = reinterpret_cast<const uint4 *>(bits)[e * 2 + 1];
    const uint32_t ws[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    uint32_t u = off[e];
#pragma unroll
    for (int j = 0; j < 8; j++) u += __popc(j < bw ? ws[j] : j == bw ? ws[j] & ((1u << (b & 31)) - 1u) : 0u);
    bool present = false;
#pragma unroll
    for (int j = 0; j < 8; j++) if (j == bw) present = (ws[j] >> (b & 31)) & 1u;
    if (!present) { *missing = 1; continue; }  // a caller's key list that does not cover these tiles: the call fails
    out[t * 64 + src] = rs[pi * 16 + rank_at(bins[u], map_value)];  // re-mirror (2721-2722)
  }
}

// DeviseBestMixingPlanYliluoma, the live SSE4.1 path (ASM_DBMP, tilingencoder.pas:2339-2563): greedy mix of up to
// Y2MixedColors palette entries; candidate averages use the reciprocal table FVecInv[4t..4t+3] = 65536 div t
// (1698-1699) on four 32-bit lanes (r, g, b, luma div 1000), penalty = 13 (dR^2+dG^2+dB^2) + 32 dL^2 as a 32-bit
// wrapping sum compared unsigned, first strict minimum wins; the list is sorted by luma and read at
// (map * count) >> 6 (2716).
__global__ __launch_bounds__(64) void k_dither_yliluoma(const uint32_t *__restrict__ tiles, const uint8_t *__restrict__ flags,
                                                        const int32_t *__restrict__ pal_idx, int64_t n, const int32_t *__restrict__ palettes,
                                                        int npal, int pal_size, int y2_mixed, const uint8_t *__restrict__ dither_map,
                                                        uint8_t *__restrict__ out) {
  __shared__ int4 s_plan[64];  // r, g, b, luma div 1000 (Y2Palette)
  __shared__ int s_luma[64];   // LumaPal
  __shared__ uint8_t s_rank[64], s_remap[64];
  __shared__ uint16_t s_list[64][64], s_stack[64][64];
  const int lane = threadIdx.x;
  int cached_pal = -1, cnt = 0;
  for (int64_t t = blockIdx.x; t < n; t += gridDim.x) {
    const int pi = pal_idx[t];
    if (pi != cached_pal) {
      __syncthreads();
      int col = TM_NULL_COLOR;
      if (lane < pal_size && pi >= 0 && pi < npal) col = palettes[(int64_t)pi * pal_size + lane];
      const bool live = col != TM_NULL_COLOR;
      const unsigned long long m = __ballot(live);
      cnt = __popcll(m);
      const int pos = __popcll(m & ((1ull << lane) - 1ull));
      const int r = col & 0xff, g = (col >> 8) & 0xff, b = (col >> 16) & 0xff;
      const int luma = r * 299 + g * 587 + b * 114;
      if (live) { s_plan[pos] = make_int4(r, g, b, luma / 1000); s_luma[pos] = luma; s_remap[pos] = (uint8_t)lane; }
      __syncthreads();
      if (lane < cnt) {
        const int my = s_luma[lane];
        int rk = 0;
        for (int i = 0; i < cnt; i++) rk += (s_luma[i] < my) ? 1 : 0;
        s_rank[lane] = (uint8_t)rk;
      }
      __syncthreads();
      cached_pal = pi;
    }
    if (cnt == 0) { out[t * 64 + lane] = 0; continue; }
    const int f = flags ? flags[t] : 0;
    const int y = lane >> 3, x = lane & 7;
    const int src = (((f & 2) ? 7 - y : y) << 3) | ((f & 1) ? 7 - x : x);
    const uint32_t c = tiles[t * 64 + src];
    uint32_t tgt[4] = {c & 0xff, (c >> 8) & 0xff, (c >> 16) & 0xff, 0};
    tgt[3] = (tgt[0] * 299u + tgt[1] * 587u + tgt[2] * 114u) / 1000u;
    const uint32_t wgt[4] = {13, 13, 13, 32};
    int plan_count = 0;
    uint32_t so_far[4] = {0, 0, 0, 0};
    while (plan_count < y2_mixed) {
      const int max_test = plan_count == 0 ? 1 : plan_count;
      unsigned long long best = (1ull << 63) - 1ull;
      int chosen = 0, chosen_amount = 1;
      for (int idx = 0; idx < cnt; idx++) {
        const int4 p = s_plan[idx];
        uint32_t sum[4] = {so_far[0], so_far[1], so_far[2], so_far[3]};
        uint32_t add[4] = {(uint32_t)p.x, (uint32_t)p.y, (uint32_t)p.z, (uint32_t)p.w};
        for (int tt = plan_count + 1; tt <= plan_count + max_test; tt++) {
          const uint32_t inv = 65536u / (uint32_t)tt;
          uint32_t pen = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            sum[k] += add[k];
            add[k] += 1;
            const uint32_t d = ((sum[k] * inv) >> 16) - tgt[k];
            pen += d * d * wgt[k];
          }
          if ((unsigned long long)pen < best) { best = pen; chosen = idx; chosen_amount = tt - plan_count; }
        }
      }
      if (chosen_amount > 64 - plan_count) chosen_amount = 64 - plan_count;  // list room (<= 30 entries for Y2MixedColors <= 16)
      const uint16_t e = (uint16_t)((s_rank[chosen] << 8) | chosen);
      for (int k = 0; k < chosen_amount; k++) s_list[plan_count + k][lane] = e;
      plan_count += chosen_amount;
      const int4 p = s_plan[chosen];
      so_far[0] += (uint32_t)p.x * chosen_amount; so_far[1] += (uint32_t)p.y * chosen_amount;
      so_far[2] += (uint32_t)p.z * chosen_amount; so_far[3] += (uint32_t)p.w * chosen_amount;
    }
    lane_quicksort(s_list, s_stack, lane, plan_count - 1);
    const int map_value = (dither_map[lane] * plan_count) >> 6;
    out[t * 64 + src] = s_remap[s_list[map_value][lane] & 0xff];
  }
}

int launch_dither(const void *tiles, const void *flags, const void *pal_idx, int64_t n, const void *palettes, int npal, int pal_size,
                  int use_tk, int y2_mixed, void *out_pal_px, hipStream_t stream, int64_t *pairs_planned, const void *pair_keys, int64_t n_pair_keys) {
  if (pairs_planned) *pairs_planned = 0;
  const DeviceTables *tab;
  TM_TRY(get_tables(&tab));
  TM_CHECK(pal_size >= 2 && pal_size <= 64, TM_E_INVAL, "PaletteSize %d outside 2..64 (tilingencoder.pas:2965)", pal_size);
  TM_CHECK(use_tk || (y2_mixed >= 1 && y2_mixed <= 16), TM_E_INVAL, "DitheringYliluoma2MixedColors %d outside 1..16 (tilingencoder.pas:2922)", y2_mixed);
  if (n <= 0) return TM_OK;
  int grid = (int)std::min<int64_t>(n, 256 * 40);
  if (use_tk) {
    const bool literal_only = knobs().dither_literal;  // debugging aid: every tile through the literal sort
    DevBuf cls;
    if (!literal_only) {
      TM_TRY(cls.alloc((size_t)npal));
      hipLaunchKernelGGL(k_palette_class, dim3((npal + 63) / 64), dim3(64), 0, stream, (const int32_t *)palettes, npal, pal_size, cls.as<uint8_t>());
      // distinct (palette, colour) pairs first: worth it when at most half of the pixels are distinct
      const bool no_dedup = knobs().dither_no_dedup;
      const int64_t nent = (int64_t)npal << 16;
      bool dedup = !no_dedup && npal <= DD_MAX_PAL && n >= 1024 && nent <= n * 256;
      DevBuf bits, cnt, off, seg, scan_tmp, missing;
      std::vector<uint32_t> hseg((size_t)npal + 1, 0);
      if (dedup) {
        TM_TRY(bits.alloc((size_t)nent * 32)); TM_TRY(cnt.alloc((size_t)(nent + 1) * 4)); TM_TRY(off.alloc((size_t)(nent + 1) * 4)); TM_TRY(seg.alloc((size_t)(npal + 1) * 4));
        TM_HIP(hipMemsetAsync(bits.p, 0, (size_t)nent * 32, stream));
        if (pair_keys && n_pair_keys > 0)
          hipLaunchKernelGGL(k_dd_mark_keys, dim3((unsigned)std::min<int64_t>((n_pair_keys + 255) / 256, 256 * 16)), dim3(256), 0, stream, (const unsigned long long *)pair_keys, n_pair_keys, npal,
                             cls.as<uint8_t>(), bits.as<uint32_t>());
        else
          hipLaunchKernelGGL(k_dd_mark, dim3((unsigned)std::min<int64_t>((n * 64 + 255) / 256, 256 * 32)), dim3(256), 0, stream, (const uint32_t *)tiles, (const int32_t *)pal_idx, n, npal,
                             cls.as<uint8_t>(), bits.as<uint32_t>());
        hipLaunchKernelGGL(k_dd_count, dim3((unsigned)std::min<int64_t>((nent + 256) / 256, 256 * 16)), dim3(256), 0, stream, bits.as<uint32_t>(), nent, cnt.as<uint32_t>());
        size_t tb = 0;
        TM_HIP(rocprim::exclusive_scan(nullptr, tb, cnt.as<uint32_t>(), off.as<uint32_t>(), 0u, (size_t)(nent + 1), rocprim::plus<uint32_t>(), stream));
        TM_TRY(scan_tmp.alloc(tb));
        TM_HIP(rocprim::exclusive_scan(scan_tmp.p, tb, cnt.as<uint32_t>(), off.as<uint32_t>(), 0u, (size_t)(nent + 1), rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(k_dd_seg, dim3((npal + 1 + 63) / 64), dim3(64), 0, stream, off.as<uint32_t>(), npal, seg.as<uint32_t>());
        {
          HostRead hr_(stream);
          TM_TRY(hr_.get(hseg.data(), seg.p, hseg.size() * 4));
          TM_TRY(hr_.wait());
        }
        dedup = (int64_t)hseg[npal] * 2 <= n * 64;

      }
      if (dedup && hseg[npal] > 0) {
        const uint32_t nu = hseg[npal];
        if (pairs_planned) *pairs_planned = nu;
        std::vector<int32_t> cpal;
        std::vector<uint32_t> cbeg, cend;
        for (int p = 0; p < npal; p++)
          for (uint32_t a = hseg[p]; a < hseg[p + 1]; a += 64) { cpal.push_back(p); cbeg.push_back(a); cend.push_back(hseg[p + 1]); }
        const int64_t nch = (int64_t)cpal.size();
        DevBuf ucol, ubins, rs, dpal, dbeg, dend;
        TM_TRY(ucol.alloc((size_t)nu * 4)); TM_TRY(ubins.alloc((size_t)nu * 16)); TM_TRY(rs.alloc((size_t)npal * 16)); TM_TRY(missing.alloc(4));
        TM_HIP(hipMemsetAsync(missing.p, 0, 4, stream));
        TM_TRY(dpal.alloc((size_t)nch * 4)); TM_TRY(dbeg.alloc((size_t)nch * 4)); TM_TRY(dend.alloc((size_t)nch * 4));
        TM_HIP(hipMemcpyAsync(dpal.p, cpal.data(), (size_t)nch * 4, hipMemcpyHostToDevice, stream));
        TM_HIP(hipMemcpyAsync(dbeg.p, cbeg.data(), (size_t)nch * 4, hipMemcpyHostToDevice, stream));
        TM_HIP(hipMemcpyAsync(dend.p, cend.data(), (size_t)nch * 4, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_dd_expand, dim3((unsigned)std::min<int64_t>((nent + 255) / 256, 256 * 16)), dim3(256), 0, stream, bits.as<uint32_t>(), off.as<uint32_t>(), nent, ucol.as<uint32_t>());
        const DdArgs dd{ucol.as<uint32_t>(), dpal.as<int32_t>(), dbeg.as<uint32_t>(), dend.as<uint32_t>(), ubins.as<uint4>(), rs.as<uint8_t>()};
        hipLaunchKernelGGL(k_dither_tk_fast<true>, dim3((unsigned)std::min<int64_t>(nch, 256 * 40)), dim3(64), 0, stream, (const uint32_t *)nullptr, (const uint8_t *)nullptr,
                           (const int32_t *)nullptr, nch, (const int32_t *)palettes, npal, pal_size, cls.as<uint8_t>(), tab->dither_map, (uint8_t *)nullptr, dd);
        hipLaunchKernelGGL(k_dd_lookup, dim3((unsigned)std::min<int64_t>((n + 3) / 4, 256 * 32)), dim3(256), 0, stream, (const uint32_t *)tiles, (const uint8_t *)flags, (const int32_t *)pal_idx, n,
                           npal, cls.as<uint8_t>(), bits.as<uint32_t>(), off.as<uint32_t>(), ubins.as<uint4>(), rs.as<uint8_t>(), tab->dither_map, (uint8_t *)out_pal_px,
                           missing.as<int>());
        TM_HIP(hipGetLastError());
        int hmiss = 0;
        {
          HostRead hr_(stream);
          TM_TRY(hr_.get(&hmiss, missing.p, 4));
          TM_TRY(hr_.wait());  // (the chunk tables and the scratch die with this scope)
        }
        TM_CHECK(!hmiss, TM_E_INVAL, "dither: the list of distinct pixel keys does not cover these tiles");
      } else if (!dedup)
        hipLaunchKernelGGL(k_dither_tk_fast<false>, dim3(grid), dim3(64), 0, stream, (const uint32_t *)tiles, (const uint8_t *)flags,
                           (const int32_t *)pal_idx, n, (const int32_t *)palettes, npal, pal_size, cls.as<uint8_t>(), tab->dither_map,
                           (uint8_t *)out_pal_px, DdArgs{});
    }
    hipLaunchKernelGGL(k_dither_tk, dim3(grid), dim3(64), 0, stream, (const uint32_t *)tiles, (const uint8_t *)flags,
                       (const int32_t *)pal_idx, n, (const int32_t *)palettes, npal, pal_size, literal_only ? nullptr : cls.as<uint8_t>(),
                       tab->dither_map, (uint8_t *)out_pal_px);
    TM_HIP(hipGetLastError());
    TM_HIP(hipStreamSynchronize(stream));  // cls is freed on return
  } else
    hipLaunchKernelGGL(k_dither_yliluoma, dim3(grid), dim3(64), 0, stream, (const uint32_t *)tiles, (const uint8_t *)flags,
                       (const int32_t *)pal_idx, n, (const int32_t *)palettes, npal, pal_size, y2_mixed, tab->dither_map,
                       (uint8_t *)out_pal_px);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

}  // namespace tmx
