// tm_epu.hip -- (f)#3 FrameTilingExtendedPaletteUsage: the k = 64 branch of TFrame.Reconstruct.DoXY
// (tilingencoder.pas:1559-1610).
//   k_knn_topk    ann_kdtree_short_search_multi(k, eps 0) for a batch of queries: the k database rows with the smallest true
//                 L2 distance, ordered by (distance, index) -- the build's tie rule; ANN's own order is not recoverable.
//   k_epu_rerank  every unique tile index of that list x every unique palette of the list, scored with
//                 CompareEuclideanDCTPtr_asm as written (utils.pas:559-725, quirks as in tm_motion.hip), first strict
//                 minimum in ascending (tile, palette) order.  The candidate vectors come from a table of the features of
//                 EVERY global tile under EVERY palette (T x P rows, built once per Reconstruct with k_features_i16<3>)
//                 instead of being recomputed per query as 1590-1591 do: same values, 3-4 orders of magnitude fewer DCTs.
// The shipped top-k runs on the pruned MFMA scan (tm_knn.hip: knn_index_search_topk); k_knn_topk below is the exact VALU
// brute force (v_dot2c_i32_i16, database rows through the scalar cache) kept as its last-resort fallback and as the
// independent implementation the tests compare it with (TM_TOPK_BRUTE=1).
#include <algorithm>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {
namespace {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}

// One wave = 64 queries (lane = query, its 192 coefficients in 96 registers); database rows are wave-uniform and arrive
// through scalar loads.  SSD = |q|^2 + |t|^2 - 2 q.t, every term mod 2^32 (exact: SSD < 2^32).  Per lane a list of the k best
// (distance << 32 | index) keys in LDS [slot][lane]; a new key enters only if it beats the list's maximum.
__global__ __launch_bounds__(64) void k_knn_topk(const uint32_t *__restrict__ queries, int64_t nq, const uint32_t *__restrict__ db,
                                                 const uint32_t *__restrict__ db_norm, int nt, int k, int32_t *__restrict__ out_idx,
                                                 uint32_t *__restrict__ out_err) {
  extern __shared__ u64 s_keys[];  // [k][64]
  const int lane = threadIdx.x;
  const int64_t qi = (int64_t)blockIdx.x * 64 + lane;
  const bool valid = qi < nq;
  uint32_t q[96];
  uint32_t qn = 0;
#pragma unroll
  for (int j = 0; j < 96; j += 4) {
    const uint4 v = valid ? *reinterpret_cast<const uint4 *>(queries + qi * 96 + j) : make_uint4(0, 0, 0, 0);
    q[j] = v.x; q[j + 1] = v.y; q[j + 2] = v.z; q[j + 3] = v.w;
  }
#pragma unroll
  for (int j = 0; j < 96; j++) qn = (uint32_t)dot2(q[j], q[j], (int)qn);
  int cnt = 0, mslot = 0;
  u64 mx = 0;
  for (int t = 0; t < nt; t++) {
    const uint32_t *row = db + (int64_t)t * 96;  // uniform address: scalar loads
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 96; j++) acc = dot2(q[j], row[j], acc);
    const uint32_t d = qn + db_norm[t] - 2u * (uint32_t)acc;
    const u64 key = ((u64)d << 32) | (uint32_t)t;
    if (cnt < k) {  // rows come in index order and k <= nt is the common case: the first k rows fill the list (wave-uniform branch)
      s_keys[cnt * 64 + lane] = key;
      if (key > mx || cnt == 0) { mx = key; mslot = cnt; }
      cnt++;
    } else if (key < mx) {
      s_keys[mslot * 64 + lane] = key;
      mx = 0;
      for (int s = 0; s < k; s++) {
        const u64 v = s_keys[s * 64 + lane];
        if (v > mx) { mx = v; mslot = s; }
      }
    }
  }
  if (!valid) return;
  // selection sort of the lane's list into (distance, index) order; k is small
  for (int o = 0; o < k; o++) {
    if (o >= cnt) { out_idx[qi * k + o] = -1; out_err[qi * k + o] = 0xffffffffu; continue; }
    u64 best = ~0ull;
    int bs = 0;
    for (int s = 0; s < cnt; s++) {
      const u64 v = s_keys[s * 64 + lane];
      if (v < best) { best = v; bs = s; }
    }
    s_keys[bs * 64 + lane] = ~0ull;
    out_idx[qi * k + o] = (int32_t)(best & 0xffffffffu);
    out_err[qi * k + o] = (uint32_t)(best >> 32);
  }
}

__global__ void k_row_norms(const int16_t *__restrict__ rows, int64_t n, uint32_t *__restrict__ norm) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t s = 0;
    for (int j = 0; j < 192; j++) { const int v = rows[i * 192 + j]; s += (uint32_t)(v * v); }
    norm[i] = s;
  }
}

__device__ __forceinline__ uint32_t sat_sub2(uint32_t a, uint32_t b) {
  const s16x2 r = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b));
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t sq2(uint32_t d) { return (uint32_t)dot2(d, d, 0); }
__device__ __forceinline__ uint32_t block_term(const uint4 a, const uint4 b) {
  return sq2(sat_sub2(a.x, b.x)) + sq2(sat_sub2(a.y, b.y)) + sq2(sat_sub2(a.z, b.z)) + sq2(sat_sub2(a.w, b.w));
}

// The two lists of a query (1565-1577): the tile indices of its k nearest rows and the palettes of those tiles, each sorted and
// made unique across the wave (lane s holds list slot s).  -1 entries (the pads of a database with fewer than k rows, 1572-1573)
// sort first.  Every kernel that walks a query's (tile, palette) pairs builds the lists with this one function, so they agree on
// the order o = tile slot * nup + palette slot.
__device__ __forceinline__ void epu_lists(const int32_t *__restrict__ knn_idx, int64_t qi, int k, const int32_t *__restrict__ tile_pal, int64_t ntiles,
                                          int32_t *s_t, int32_t *s_p, int32_t *s_ut, int32_t *s_up, int *s_n /* [2]: nut, nup */) {
  const int lane = threadIdx.x;
  int32_t t = -1, p = -1;
  if (lane < k) {
    t = knn_idx[qi * k + lane];
    if (t >= 0 && t < ntiles) p = tile_pal[t]; else t = -1;  // 1565-1574
  }
  s_t[lane] = lane < k ? t : 0x7fffffff;  // padding sorts last and is dropped
  s_p[lane] = lane < k ? p : 0x7fffffff;
  if (lane == 0) { s_n[0] = 0; s_n[1] = 0; }
  __syncthreads();
  // rank sort (64 values): position = number of smaller values, ties by slot; then unique
  const int32_t mt = s_t[lane], mp = s_p[lane];
  int rt = 0, rp = 0;
  for (int i = 0; i < 64; i++) {
    const int32_t ot = s_t[i], op = s_p[i];
    rt += (ot < mt || (ot == mt && i < lane)) ? 1 : 0;
    rp += (op < mp || (op == mp && i < lane)) ? 1 : 0;
  }
  __syncthreads();
  s_t[rt] = mt;
  s_p[rp] = mp;
  __syncthreads();
  const int32_t vt = s_t[lane], vp = s_p[lane];
  const bool ft = vt != 0x7fffffff && (lane == 0 || s_t[lane - 1] != vt);
  const bool fp = vp != 0x7fffffff && (lane == 0 || s_p[lane - 1] != vp);
  const unsigned long long bt = __ballot(ft), bp = __ballot(fp);
  if (ft) s_ut[__popcll(bt & ((1ull << lane) - 1ull))] = vt;
  if (fp) s_up[__popcll(bp & ((1ull << lane) - 1ull))] = vp;
  if (lane == 0) { s_n[0] = __popcll(bt); s_n[1] = __popcll(bp); }
  __syncthreads();
}

// One wave per query; 8 lanes share a (tile, palette) pair exactly as k_motion_search's lanes share a candidate (same quirk
// handling).  The pair's feature vector comes either from the table of every tile under every palette (ntiles x npal rows), or --
// when that table would not fit (PaletteCount = 1024, the reference's default: T x P x 384 bytes is tens of terabytes) -- from the
// rows made for just the pairs this batch of queries names: row_of[pair_off[q] + o].
template <bool ONDEMAND>
__global__ __launch_bounds__(64) void k_epu_rerank(const int16_t *__restrict__ queries, int64_t nq, const int32_t *__restrict__ knn_idx, int k,
                                                   const int32_t *__restrict__ tile_pal, int64_t ntiles, int npal,
                                                   const int16_t *__restrict__ table /* [ntiles][npal][192], or the on-demand rows */,
                                                   const unsigned long long *__restrict__ pair_off, const uint32_t *__restrict__ row_of,
                                                   int32_t *__restrict__ out_tile, int32_t *__restrict__ out_pal, uint32_t *__restrict__ out_err) {
  __shared__ int32_t s_t[64], s_p[64], s_ut[64], s_up[64];
  __shared__ int s_n[2];
  const int64_t qi = blockIdx.x;
  if (qi >= nq) return;
  const int lane = threadIdx.x, j8 = lane & 7, grp = lane >> 3, role = j8 & 3;
  epu_lists(knn_idx, qi, k, tile_pal, ntiles, s_t, s_p, s_ut, s_up, s_n);
  const int nut = s_n[0], nup = s_n[1], npairs = nut * nup;
  const uint4 *pq = reinterpret_cast<const uint4 *>(queries + qi * 192) + j8 * 3;
  const uint4 a0 = pq[0], a1 = pq[1], a2 = pq[2];
  uint32_t best = 0xffffffffu;
  int bo = 0x7fffffff;
  for (int o = grp; o < npairs; o += 8) {
    const int ti = o / nup, pi = o - ti * nup;
    const int32_t tile = s_ut[ti], pal = s_up[pi];
    uint32_t acc = 0;
    // -1 entries sort first and are skipped by the reference's `<> prevTileIdx` / `<> prevPalIdx` tests, which start at -1 (1582-1588)
    const bool ok = tile >= 0 && pal >= 0;
    if (ok) {
      const int64_t row = ONDEMAND ? (int64_t)row_of[pair_off[qi] + (unsigned long long)o] : (int64_t)tile * npal + pal;
      const uint4 *pb = reinterpret_cast<const uint4 *>(table + row * 192) + j8 * 3;
      const uint4 b0 = pb[0], b1 = pb[1], b2 = pb[2];
      uint4 b5 = make_uint4(0, 0, 0, 0);
      if (role == 2) b5 = pb[-1];
      uint4 d;
      d.x = sat_sub2(a0.x, b5.x); d.y = sat_sub2(a0.y, b5.y); d.z = sat_sub2(a0.z, b5.z); d.w = sat_sub2(a0.w, b5.w);
      acc += block_term(d, b0);
      const uint32_t p0 = sq2(sat_sub2(a1.x, b1.x)), p1 = sq2(sat_sub2(a1.y, b1.y)), p2 = sq2(sat_sub2(a1.z, b1.z)), p3 = sq2(sat_sub2(a1.w, b1.w));
      acc += p0 + p1 + p2 + p3;
      if (j8 == 2) acc += sq2(p0) + sq2(p1) + sq2(p2) + sq2(p3);
      if (role != 1) acc += block_term(a2, b2);
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    if (ok && acc < best) { best = acc; bo = o; }
  }
  // across the 8 groups: smallest error, then smallest order index
  for (int sh = 8; sh < 64; sh <<= 1) {
    const uint32_t oe = __shfl_xor(best, sh);
    const int oo = __shfl_xor(bo, sh);
    if (oe < best || (oe == best && oo < bo)) { best = oe; bo = oo; }
  }
  if (lane == 0) {
    if (bo == 0x7fffffff) { out_tile[qi] = -1; out_pal[qi] = -1; out_err[qi] = 0xffffffffu; }
    else { out_tile[qi] = s_ut[bo / nup]; out_pal[qi] = s_up[bo % nup]; out_err[qi] = best; }
  }
}

// ---- the on-demand rows: which (tile, palette) pairs does a batch of queries name?
__global__ __launch_bounds__(64) void k_epu_count(int64_t nq, const int32_t *__restrict__ knn_idx, int k, const int32_t *__restrict__ tile_pal, int64_t ntiles,
                                                  unsigned long long *__restrict__ count) {
  __shared__ int32_t s_t[64], s_p[64], s_ut[64], s_up[64];
  __shared__ int s_n[2];
  const int64_t qi = blockIdx.x;
  if (qi >= nq) return;
  epu_lists(knn_idx, qi, k, tile_pal, ntiles, s_t, s_p, s_ut, s_up, s_n);
  if (threadIdx.x == 0) count[qi] = (unsigned long long)(s_n[0] * s_n[1]);
}
__global__ __launch_bounds__(64) void k_epu_emit(int64_t nq, const int32_t *__restrict__ knn_idx, int k, const int32_t *__restrict__ tile_pal, int64_t ntiles,
                                                 const unsigned long long *__restrict__ pair_off, unsigned long long *__restrict__ keys, uint32_t *__restrict__ pos) {
  __shared__ int32_t s_t[64], s_p[64], s_ut[64], s_up[64];
  __shared__ int s_n[2];
  const int64_t qi = blockIdx.x;
  if (qi >= nq) return;
  epu_lists(knn_idx, qi, k, tile_pal, ntiles, s_t, s_p, s_ut, s_up, s_n);
  const int nup = s_n[1], npairs = s_n[0] * nup;
  const unsigned long long off = pair_off[qi];
  for (int o = threadIdx.x; o < npairs; o += 64) {
    const int ti = o / nup, pi = o - ti * nup;
    const int32_t tile = s_ut[ti], pal = s_up[pi];
    keys[off + o] = (tile >= 0 && pal >= 0) ? ((unsigned long long)(uint32_t)tile << 32) | (uint32_t)pal : ~0ull;  // a pad pair: no row
    pos[off + o] = (uint32_t)(off + o);
  }
}
__global__ void k_epu_heads(const unsigned long long *__restrict__ keys, int64_t n, uint32_t *__restrict__ head) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ void k_epu_rows(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ pos, const uint32_t *__restrict__ rank_incl, int64_t n,
                           uint32_t *__restrict__ row_of, unsigned long long *__restrict__ ukeys) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t r = rank_incl[i] - 1u;
    row_of[pos[i]] = r;
    if (i == 0 || keys[i] != keys[i - 1]) ukeys[r] = keys[i];
  }
}

}  // namespace

int launch_knn_topk(const void *queries, int64_t nq, const void *db, int64_t nt, int k, void *out_idx, void *out_err, hipStream_t stream) {
  TM_CHECK(k >= 1 && k <= 64, TM_E_INVAL, "top-k: k %d outside 1..64", k);
  TM_CHECK(nt >= 0 && nt < (int64_t)1 << 31, TM_E_INVAL, "top-k: database size out of range");
  if (nq <= 0) return TM_OK;
  DevBuf norm;
  TM_TRY(norm.alloc((size_t)std::max<int64_t>(nt, 1) * 4));
  if (nt > 0)
    hipLaunchKernelGGL(k_row_norms, dim3((unsigned)std::min<int64_t>((nt + 255) / 256, 4096)), dim3(256), 0, stream, (const int16_t *)db, nt,
                       norm.as<uint32_t>());
  hipLaunchKernelGGL(k_knn_topk, dim3((unsigned)((nq + 63) / 64)), dim3(64), (size_t)k * 64 * 8, stream, (const uint32_t *)queries, nq,
                     (const uint32_t *)db, norm.as<uint32_t>(), (int)nt, k, (int32_t *)out_idx, (uint32_t *)out_err);
  TM_HIP(hipGetLastError());
  TM_HIP(hipStreamSynchronize(stream));  // norm is freed on return
  return TM_OK;
}

int launch_epu_rerank(const void *queries, int64_t nq, const void *knn_idx, int k, const void *tile_pal, int64_t ntiles, int npal,
                      const void *table, void *out_tile, void *out_pal, void *out_err, hipStream_t stream) {
  TM_CHECK(k >= 1 && k <= 64 && npal >= 1, TM_E_INVAL, "epu: bad arguments");
  if (nq <= 0) return TM_OK;
  hipLaunchKernelGGL(k_epu_rerank<false>, dim3((unsigned)nq), dim3(64), 0, stream, (const int16_t *)queries, nq, (const int32_t *)knn_idx, k,
                     (const int32_t *)tile_pal, ntiles, npal, (const int16_t *)table, nullptr, nullptr, (int32_t *)out_tile, (int32_t *)out_pal, (uint32_t *)out_err);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

// The re-rank without the T x P table: the (tile, palette) pairs the queries of the batch name are collected, sorted and made unique,
// the feature vectors of exactly those pairs are built (k_features_i16<4>), and the re-rank reads them through a per-pair row index.
// Same vectors, same order, same result as with the table.  The batch is cut so that the pair arrays stay within `budget` bytes.
int launch_epu_rerank_ondemand(const void *queries, int64_t nq, const void *knn_idx, int k, const void *tile_pal, int64_t ntiles, const void *pal_px,
                               const void *palettes, int npal, int pal_size, void *out_tile, void *out_pal, void *out_err, hipStream_t stream) {
  TM_CHECK(k >= 1 && k <= 64 && npal >= 1, TM_E_INVAL, "epu: bad arguments");
  const int64_t step = std::max<int64_t>(1, std::min<int64_t>(nq, ((int64_t)1 << 26) / ((int64_t)k * k)));  // at most 64 M pairs per batch (k x k per query)
  for (int64_t q0 = 0; q0 < nq; q0 += step) {
    const int64_t n = std::min(step, nq - q0);
    const int32_t *idx = (const int32_t *)knn_idx + q0 * k;
    DevBuf cnt, off, tmp, keys, keys2, pos, pos2, head, rank, row_of, ukeys, feat;
    TM_TRY(cnt.alloc((size_t)(n + 1) * 8)); TM_TRY(off.alloc((size_t)(n + 1) * 8));
    TM_HIP(hipMemsetAsync(cnt.p, 0, (size_t)(n + 1) * 8, stream));
    hipLaunchKernelGGL(k_epu_count, dim3((unsigned)n), dim3(64), 0, stream, n, idx, k, (const int32_t *)tile_pal, ntiles, cnt.as<unsigned long long>());
    size_t tb = 0;
    TM_HIP(rocprim::exclusive_scan(nullptr, tb, cnt.as<unsigned long long>(), off.as<unsigned long long>(), 0ull, (size_t)(n + 1), rocprim::plus<unsigned long long>(), stream));
    TM_TRY(tmp.alloc(tb));
    TM_HIP(rocprim::exclusive_scan(tmp.p, tb, cnt.as<unsigned long long>(), off.as<unsigned long long>(), 0ull, (size_t)(n + 1), rocprim::plus<unsigned long long>(), stream));
    unsigned long long m = 0;
    {
      HostRead hr_(stream);
      TM_TRY(hr_.get(&m, off.as<unsigned long long>() + n, 8));
      TM_TRY(hr_.wait());
    }
    if (m > 0) {
      TM_CHECK(m < (1ull << 32), TM_E_NOMEM, "epu: %llu pairs in one batch", m);
      TM_TRY(keys.alloc((size_t)m * 8)); TM_TRY(keys2.alloc((size_t)m * 8)); TM_TRY(pos.alloc((size_t)m * 4)); TM_TRY(pos2.alloc((size_t)m * 4));
      TM_TRY(head.alloc((size_t)m * 4)); TM_TRY(rank.alloc((size_t)m * 4)); TM_TRY(row_of.alloc((size_t)m * 4));
      hipLaunchKernelGGL(k_epu_emit, dim3((unsigned)n), dim3(64), 0, stream, n, idx, k, (const int32_t *)tile_pal, ntiles, off.as<unsigned long long>(),
                         keys.as<unsigned long long>(), pos.as<uint32_t>());
      TM_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys.as<unsigned long long>(), keys2.as<unsigned long long>(), pos.as<uint32_t>(), pos2.as<uint32_t>(), (size_t)m, 0, 64, stream));
      TM_TRY(tmp.alloc(tb));
      TM_HIP(rocprim::radix_sort_pairs(tmp.p, tb, keys.as<unsigned long long>(), keys2.as<unsigned long long>(), pos.as<uint32_t>(), pos2.as<uint32_t>(), (size_t)m, 0, 64, stream));
      const int g = (int)std::min<unsigned long long>((m + 255) / 256, 8192);
      hipLaunchKernelGGL(k_epu_heads, dim3(g), dim3(256), 0, stream, keys2.as<unsigned long long>(), (int64_t)m, head.as<uint32_t>());
      TM_HIP(rocprim::inclusive_scan(nullptr, tb, head.as<uint32_t>(), rank.as<uint32_t>(), (size_t)m, rocprim::plus<uint32_t>(), stream));
      TM_TRY(tmp.alloc(tb));
      TM_HIP(rocprim::inclusive_scan(tmp.p, tb, head.as<uint32_t>(), rank.as<uint32_t>(), (size_t)m, rocprim::plus<uint32_t>(), stream));
      uint32_t nu = 0;
      {
        HostRead hr_(stream);
        TM_TRY(hr_.get(&nu, rank.as<uint32_t>() + (m - 1), 4));
        TM_TRY(hr_.wait());
      }
      TM_TRY(ukeys.alloc((size_t)nu * 8)); TM_TRY(feat.alloc((size_t)nu * 384));
      hipLaunchKernelGGL(k_epu_rows, dim3(g), dim3(256), 0, stream, keys2.as<unsigned long long>(), pos2.as<uint32_t>(), rank.as<uint32_t>(), (int64_t)m, row_of.as<uint32_t>(),
                         ukeys.as<unsigned long long>());
      TM_HIP(hipGetLastError());
      TM_TRY(launch_features_pairs(pal_px, ukeys.p, nu, palettes, pal_size, feat.p, stream));
    }
    hipLaunchKernelGGL(k_epu_rerank<true>, dim3((unsigned)n), dim3(64), 0, stream, (const int16_t *)queries + q0 * 192, n, idx, k, (const int32_t *)tile_pal, ntiles, npal,
                       feat.as<int16_t>(), off.as<unsigned long long>(), row_of.as<uint32_t>(), (int32_t *)out_tile + q0, (int32_t *)out_pal + q0, (uint32_t *)out_err + q0);
    TM_HIP(hipGetLastError());
    TM_HIP(hipStreamSynchronize(stream));  // the batch's buffers go back to the pool
  }
  return TM_OK;
}

}  // namespace tmx
