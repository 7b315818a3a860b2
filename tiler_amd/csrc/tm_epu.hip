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

// One wave per query.  Lane s holds list slot s; the two lists are sorted and made unique across the wave, then 8 lanes
// share a (tile, palette) pair exactly as k_motion_search's lanes share a candidate (same quirk handling).
__global__ __launch_bounds__(64) void k_epu_rerank(const int16_t *__restrict__ queries, int64_t nq, const int32_t *__restrict__ knn_idx, int k,
                                                   const int32_t *__restrict__ tile_pal, int64_t ntiles, int npal,
                                                   const int16_t *__restrict__ table /* [ntiles][npal][192] */, int32_t *__restrict__ out_tile,
                                                   int32_t *__restrict__ out_pal, uint32_t *__restrict__ out_err) {
  __shared__ int32_t s_t[64], s_p[64], s_ut[64], s_up[64];
  __shared__ int s_nut, s_nup;
  const int64_t qi = blockIdx.x;
  if (qi >= nq) return;
  const int lane = threadIdx.x, j8 = lane & 7, grp = lane >> 3, role = j8 & 3;
  int32_t t = -1, p = -1;
  if (lane < k) {
    t = knn_idx[qi * k + lane];
    if (t >= 0 && t < ntiles) p = tile_pal[t]; else t = -1;  // 1565-1574
  }
  s_t[lane] = lane < k ? t : 0x7fffffff;  // padding sorts last and is dropped
  s_p[lane] = lane < k ? p : 0x7fffffff;
  if (lane == 0) { s_nut = 0; s_nup = 0; }
  __syncthreads();
  // rank sort (64 values): position = number of smaller values, ties by slot; then unique
  {
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
    if (lane == 0) { s_nut = __popcll(bt); s_nup = __popcll(bp); }
    __syncthreads();
  }
  const int nut = s_nut, nup = s_nup, npairs = nut * nup;
  const uint4 *pq = reinterpret_cast<const uint4 *>(queries + qi * 192) + j8 * 3;
  const uint4 a0 = pq[0], a1 = pq[1], a2 = pq[2];
  uint32_t best = 0xffffffffu;
  int bo = 0x7fffffff;
  for (int o = grp; o < npairs; o += 8) {
    const int ti = o / nup, pi = o - ti * nup;
    const int32_t tile = s_ut[ti], pal = s_up[pi];
    uint32_t acc = 0;
    // -1 entries (1572-1573: the pads of a database with fewer than k rows) sort first and are skipped by the reference's
    // `<> prevTileIdx` / `<> prevPalIdx` tests, which start at -1 (1582-1588)
    const bool ok = tile >= 0 && pal >= 0;
    if (ok) {
      const uint4 *pb = reinterpret_cast<const uint4 *>(table + ((int64_t)tile * npal + pal) * 192) + j8 * 3;
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
  hipLaunchKernelGGL(k_epu_rerank, dim3((unsigned)nq), dim3(64), 0, stream, (const int16_t *)queries, nq, (const int32_t *)knn_idx, k,
                     (const int32_t *)tile_pal, ntiles, npal, (const int16_t *)table, (int32_t *)out_tile, (int32_t *)out_pal, (uint32_t *)out_err);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

}  // namespace tmx
