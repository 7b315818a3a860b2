// tm_knn.hip -- exact nearest-neighbour search of int16[192] tile features as an int8 MFMA distance GEMM.
//
// Replaces ann_kdtree_short_{create,search} (extern.pas:182-184) as used by PrepareReconstruct (tilingencoder.pas:
// 4566-4613) and TFrame.Reconstruct.DoXY (1534-1557): for every query row find the database row minimising
// CompareEuclideanDCTPtr (utils.pas:541-557).  Brute force, bit-exact, ties -> lowest database index.
//
// Scheme (DESIGN.md "KNN"):
//   SSD(q,t) = |q-c|^2 + |t-c|^2 - 2 (q-c).(t-c) for any per-column centre c.  On each SIDE (database, queries) a
//   column whose centred values stay within +-127 fits one int8 digit; the others ("big": data dependent, DC/low
//   frequencies for source tiles, many more for dithered tiles) get a balanced base-256 split v = 256 h + l.  Columns
//   are permuted so each side's big columns are a prefix (the smaller set nested in the larger), HT / HQ chunks of 32:
//      X = 65536 * (T_H . Q_H)[:min] + 256 * (T_L[:HQ] . Q_H + T_H . Q_L[:HT]) + T_L . Q_L
//   = three int32 MFMA accumulators fed by v_mfma_i32_32x32x32_i8, K = 192 + 32 (HT + HQ + min(HT,HQ)) <= 768 bytes.
//   The query digits are stored NEGATED, so one lane computes, with nq2 = 2 * (|q-c|^2 >> 1),
//      d'' = |t-c|^2 + 2 * (acc2<<16 + acc1<<8 + acc0) + nq2  ==  SSD - (|q-c|^2 & 1)   (exact mod 2^32, SSD < 2^31)
//   with three v_lshl_add_u32 + one add per element, and keeps a running (min d'', first tile) per lane.  Database rows ride
//   the MFMA A operand (accumulator rows), queries the B operand (accumulator columns = lanes), so the argmin of a
//   query never leaves its lane until the final 2-lane merge.  A second tiny kernel rescans the winning 32-row
//   tile with the plain int16 SSD to produce (index, error) under the lowest-index rule.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdlib>

#include "tm_common.h"
#include "tm_internal.h"
#include "tm_knn_kernel.h"
#define TM_KNN3_WITH_LISTS
#include "tm_knn3_kernel.h"

namespace tmx {

struct KnnPlan {
  int ht = 6, hq = 6;     // 32-column chunks that carry a high digit on the database / query side (0..6)
  int16_t centre[192];    // per source column
  int16_t perm[192];      // packed position -> source column (columns with high digits first, nested sets)
  int nbig_t = 192, nbig_q = 192;
  int tscale = 1;         // database digits are those of tscale * (t - c): 2 lets the scan's chain deliver 2 X without a final doubling
};

__host__ __device__ inline int knn_tile_bytes(int hch, int with_box) { return (6 + hch) * 1024 + 128 + (with_box ? 64 : 0); }

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

struct CurveSpec {
  int col[KNN_NC];
  int lo[3], range[3];      // the three curve columns: union range
  float scale[4], off[4];   // quantised coordinate of dimension d (three columns, radial) = (value - off) * scale, clamped to its bits
  int bits[4];
  int rlog;                 // radial coordinate taken as log2(R + 1) instead of R
};

// ---------------------------------------------------------------------------------------------------------------
// Pack n rows into MFMA fragment order: per 32-row tile [kc][64 lanes][16 B] (lane = half*32 + row) followed by
// 32 u32 norms.  negate=1 (query side): digits of (c - v) and norm >> 1; negate=0 (database): digits of (v - c).
// Rows >= n replicate row n-1 (ties resolve to the lower, real index).  err_flag is set if a digit overflows int8.
// scale (database side, KnnPlan::tscale): the digits are those of scale * (v - c); the norms stay those of v - c.
__global__ __launch_bounds__(256) void k_knn_pack(const int16_t *__restrict__ feat, int64_t n, int64_t ntiles, int hch, int negate, int scale,
                                                  const int16_t *__restrict__ centre, const int16_t *__restrict__ perm,
                                                  const uint32_t *__restrict__ rowperm, int with_box, CurveSpec cs,
                                                  int *__restrict__ box_lo, int *__restrict__ box_hi, uint8_t *__restrict__ out,
                                                  int *__restrict__ err_flag, int *__restrict__ qmeta /* query side: [ntiles][16] box, home tile, high-chunk mask */,
                                                  uint8_t *__restrict__ hmask /* database side: [ntiles] which high-digit chunks of the tile hold a non-zero digit */) {
  __shared__ int16_t s_c[192], s_p[192];
  __shared__ __attribute__((aligned(16))) int s_v[32][196];  // (pitch 196: a row's 16-value groups are 16-byte aligned, and sixteen rows' groups cover the 64 banks once)
  __shared__ uint32_t s_norm[32];
  __shared__ unsigned s_hm;  // bit kc: high-digit chunk kc of this tile is not all zero
  __shared__ long long s_bsq[32];  // query side: squared distance of each row from the centres over the box columns
  __shared__ __attribute__((aligned(16))) int16_t s_raw[32][200];  // the tile's rows as they lie in memory (pitch 400 B)
  for (int i = threadIdx.x; i < 192; i += 256) { s_p[i] = perm[i]; s_c[i] = centre[perm[i]]; }
  const int kch = 6 + hch, tile_bytes = knn_tile_bytes(hch, with_box);
  // the 32 rows of a tile come in as 16-byte vectors (three per thread) and are permuted out of LDS (the column permutation would otherwise
  // turn the read into 6 144 two-byte loads per tile); the NEXT tile's vectors are fetched while this one is worked on, and the row
  // numbers (curve order) of the one after: a workgroup walks its tiles one after the other, and two dependent round trips to memory per
  // tile were most of the kernel
  int pr[3], pv[3];
#pragma unroll
  for (int u = 0; u < 3; u++) { const int i = threadIdx.x + u * 256; pr[u] = i / 24; pv[u] = i - pr[u] * 24; }
  auto row_of = [&](int64_t tile, int r) -> int64_t {
    int64_t row = std::min<int64_t>(tile * 32 + r, n - 1);
    return rowperm ? (int64_t)rowperm[row] : row;  // rows are packed in curve order
  };
  int64_t nrow[3];   // rows of the tile after next
  uint4 nvec[3];     // vectors of the next tile
  {
    const int64_t t0 = blockIdx.x, t1 = (int64_t)blockIdx.x + gridDim.x;
#pragma unroll
    for (int u = 0; u < 3; u++) {
      nvec[u] = t0 < ntiles ? *reinterpret_cast<const uint4 *>(feat + row_of(t0, pr[u]) * 192 + pv[u] * 8) : make_uint4(0, 0, 0, 0);
      nrow[u] = t1 < ntiles ? row_of(t1, pr[u]) : 0;
    }
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    if (threadIdx.x == 0) s_hm = 0;
#pragma unroll
    for (int u = 0; u < 3; u++) *reinterpret_cast<uint4 *>(&s_raw[pr[u]][pv[u] * 8]) = nvec[u];
    {
      const int64_t t1 = tile + gridDim.x, t2 = tile + 2 * (int64_t)gridDim.x;
#pragma unroll
      for (int u = 0; u < 3; u++) {
        if (t1 < ntiles) nvec[u] = *reinterpret_cast<const uint4 *>(feat + nrow[u] * 192 + pv[u] * 8);
        if (t2 < ntiles) nrow[u] = row_of(t2, pr[u]);
      }
    }
    __syncthreads();
    // centred, permuted values of the 32 rows
    if (qmeta && threadIdx.x >= 192 && threadIdx.x < 224) {
      // the sub-tile's bounding box over the box columns, as the first scan shape computed it in its prologue: the rows are in LDS here (a
      // kernel of its own gathered six scattered columns of every row again, 0.33 ms for 3.2 M rows), and these lanes have nothing else to do
      const int r = threadIdx.x - 192;
      int lo[KNN_NC], hi[KNN_NC];
      long long boxsq = 0;
#pragma unroll
      for (int d = 0; d < KNN_NC; d++) {
        const int v = s_raw[r][cs.col[d]];
        lo[d] = hi[d] = v;
        const long long c = v - (int)centre[cs.col[d]];
        boxsq += c * c;
      }
      s_bsq[r] = boxsq;
      for (int o = 16; o > 0; o >>= 1)  // the six dimensions' exchanges of a step are independent: they overlap
#pragma unroll
        for (int d = 0; d < KNN_NC; d++) { lo[d] = min(lo[d], __shfl_xor(lo[d], o)); hi[d] = max(hi[d], __shfl_xor(hi[d], o)); }
      if (r == 0)
#pragma unroll
        for (int d = 0; d < KNN_NC; d++) { qmeta[tile * 16 + d] = lo[d]; qmeta[tile * 16 + 8 + d] = hi[d]; }
    }
    if (threadIdx.x < 192) {  // a thread per (permuted) column: no index arithmetic in the loop; the fourth wave's lanes beyond 192 sit it out
      const int p = threadIdx.x, sp = s_p[p], c = s_c[p];
#pragma unroll 8
      for (int r = 0; r < 32; r++) {
        const int v = (int)s_raw[r][sp] - c;
        s_v[r][p] = negate ? -v : v;
      }
    }
    __syncthreads();
    uint8_t *obase = out + tile * (int64_t)tile_bytes;
    bool bad = false;
    for (int piece = threadIdx.x; piece < kch * 64; piece += 256) {
      const int kc = piece >> 6, ln = piece & 63, half = ln >> 5, r = ln & 31;
      // the piece's sixteen values as four 16-byte LDS reads (sixteen 4-byte ones were most of this loop's instructions)
      const int kpos0 = kc * 32 + half * 16;  // byte position along K of the piece's first value
      const bool high = kpos0 >= 192;         // (uniform in the piece: 192 is a multiple of 16)
      const int4 *src = reinterpret_cast<const int4 *>(&s_v[r][high ? kpos0 - 192 : kpos0]);
      const int4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
      const int vals[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
      const bool must_fit = !high && kpos0 >= hch * 32;  // columns without a high digit (hch * 32 is a multiple of 16 too)
      uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
      for (int b = 0; b < 16; b++) {
        const int v = vals[b] * scale;
        const int lo = ((v + 128) & 255) - 128;  // low digit in [-128,127]
        int digit;
        if (!high) {
          digit = lo;
          if (must_fit && v != lo) bad = true;
        } else {
          digit = (v - lo) >> 8;
          if (digit < -128 || digit > 127) bad = true;
        }
        w[b >> 2] |= (uint32_t)(digit & 255) << ((b & 3) * 8);
      }
      *reinterpret_cast<uint4 *>(obase + piece * 16) = make_uint4(w[0], w[1], w[2], w[3]);
      if (high && (w[0] | w[1] | w[2] | w[3])) atomicOr(&s_hm, 1u << (kc - 6));
    }
    {  // |v-c|^2 of every row (the kernel drops the query side's parity bit): eight lanes per row, integer sums (32 threads walking 192
       // values each were the longest leg of a tile)
      const int r = threadIdx.x >> 3, part = threadIdx.x & 7;
      uint32_t sq = 0;
#pragma unroll 8
      for (int p = part; p < 192; p += 8) { const int v = s_v[r][p]; sq += (uint32_t)(v * v); }
      sq += __shfl_xor(sq, 1); sq += __shfl_xor(sq, 2); sq += __shfl_xor(sq, 4);
      // what the pack keeps per row is what the scan's chain starts from (k3_chain's `cin`): the query side's |q-c|^2 (the kernel drops its
      // parity), the database side's |t-c|^2 where its digits are those of 2 (t - c), and |t-c|^2 >> 1 where not -- the parities then go
      // into the tile's box (word 14)
      if (part == 0) { s_norm[r] = sq; reinterpret_cast<uint32_t *>(obase + kch * 1024)[r] = (with_box && scale == 1) ? sq >> 1 : sq; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // (rows >= n replicate row n - 1: they add no digit the real rows do not have)
      if (hmask) hmask[tile] = (uint8_t)s_hm;
      if (qmeta) qmeta[tile * 16 + 15] = (int)s_hm;
    }
    if (qmeta && threadIdx.x < 32) {  // the radial dimension of the sub-tile's box (the columns' part was done beside the centring phase)
      const int r = threadIdx.x;
      const long long n2 = (long long)(s_norm[r] & ~1u), boxsq = s_bsq[r];
      int lo = max(0, (int)floor(sqrt((double)max(0ll, n2 - boxsq))) - 1);
      int hi = (int)ceil(sqrt((double)max(0ll, n2 + 1 - boxsq))) + 1;
      for (int o = 16; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
      if (r == 0) { qmeta[tile * 16 + KNN_NC] = lo; qmeta[tile * 16 + 8 + KNN_NC] = hi; }
    }
    if (threadIdx.x < 32) {
      const uint32_t s = with_box ? s_norm[threadIdx.x] : 0u;
      if (with_box) {  // radial box dimension: |v-c| over the columns that are not box columns, rounded outwards, min/max over the rows
        int64_t row = std::min<int64_t>(tile * 32 + threadIdx.x, n - 1);
        if (rowperm) row = rowperm[row];
        long long boxsq = 0;
        for (int d = 0; d < KNN_NC; d++) { const long long c = (long long)feat[row * 192 + cs.col[d]] - centre[cs.col[d]]; boxsq += c * c; }
        const long long rest = std::max(0ll, (long long)s - boxsq);
        int lo = max(0, (int)floor(sqrt((double)rest)) - 1), hi = (int)ceil(sqrt((double)rest)) + 1;
        for (int o = 16; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
        const unsigned par = (unsigned)__builtin_amdgcn_ballot_w64((s & 1u) != 0);  // (lanes 0..31: one per row)
        if (threadIdx.x == 0) {
          int *tb = reinterpret_cast<int *>(obase + kch * 1024 + 128);
          tb[14] = (int)par;
          tb[15] = 0;
          tb[KNN_NC] = lo;
          tb[KNN_ND + KNN_NC] = hi;
          box_lo[(int64_t)KNN_NC * ntiles + tile] = lo;
          box_hi[(int64_t)KNN_NC * ntiles + tile] = hi;
        }
      }
    }
    if (bad) atomicOr(err_flag, 1);
    if (with_box && threadIdx.x >= 64 && threadIdx.x < 64 + KNN_NC) {  // bounding box of the tile over the box columns (raw values)
      const int d = threadIdx.x - 64;
      int a = INT_MAX, b = INT_MIN;
      for (int r = 0; r < 32; r++) {
        int64_t row = tile * 32 + r;
        if (row >= n) break;
        if (rowperm) row = rowperm[row];
        const int v = feat[row * 192 + cs.col[d]];
        a = min(a, v);
        b = max(b, v);
      }
      int *tb = reinterpret_cast<int *>(obase + kch * 1024 + 128);
      tb[d] = a;
      tb[KNN_ND + d] = b;
      box_lo[(int64_t)d * ntiles + tile] = a;
      box_hi[(int64_t)d * ntiles + tile] = b;
    }
  }
}

// second-level boxes: min / max of the tile boxes over runs of KNN_GROUP tiles, per box dimension
__global__ void k_group_boxes(const int *__restrict__ box_lo, const int *__restrict__ box_hi, int64_t ntiles, int64_t ngroups, int *__restrict__ grp_lo,
                              int *__restrict__ grp_hi) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < ngroups * KNN_ND; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t d = i / ngroups, g = i - d * ngroups;
    int a = INT_MAX, b = INT_MIN;
    for (int64_t t = g * KNN_GROUP; t < std::min<int64_t>((g + 1) * KNN_GROUP, ntiles); t++) { a = min(a, box_lo[d * ntiles + t]); b = max(b, box_hi[d * ntiles + t]); }
    grp_lo[i] = a;
    grp_hi[i] = b;
  }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
  v &= 0x3ff;
  v = (v | (v << 16)) & 0x030000ff;
  v = (v | (v << 8)) & 0x0300f00f;
  v = (v | (v << 4)) & 0x030c30c3;
  v = (v | (v << 2)) & 0x09249249;
  return v;
}

__device__ __forceinline__ uint32_t spread8(uint32_t v) {  // 8 bits -> every fourth bit
  v &= 0xff;
  v = (v | (v << 12)) & 0x000f000f;
  v = (v | (v << 6)) & 0x03030303;
  v = (v | (v << 3)) & 0x11111111;
  return v;
}

// R of every row, R = |v - c| over the columns that are not box columns (the radial box dimension of tm_knn_kernel.h),
// and its range over the rows (floats >= 0: their bit patterns order like the values).  8 lanes per row, 48 bytes each.
__global__ __launch_bounds__(256) void k_row_radial(const int16_t *__restrict__ feat, int64_t n, CurveSpec cs, const int16_t *__restrict__ centre,
                                                    float *__restrict__ out, unsigned int *__restrict__ range /* [0] min, [1] max */,
                                                    uint2 *__restrict__ ccol /* [n]: the row's three curve columns, for k_curve_keys */) {
  // this R only places the row on the curve (the box dimension gets its exact, outward-rounded values in k_knn_pack and in the scan's
  // prologue), so single precision is enough: the lane's 24 centres live in registers and every element is one subtract and one fma
  const int j8 = threadIdx.x & 7;
  float cen[24];
#pragma unroll
  for (int e = 0; e < 24; e++) cen[e] = (float)centre[j8 * 24 + e];
  float keep[24];  // 0 for the box columns, which do not count: a factor instead of a second, dependent round of loads
#pragma unroll
  for (int e = 0; e < 24; e++) {
    keep[e] = 1.0f;
#pragma unroll
    for (int d = 0; d < KNN_NC; d++) if (cs.col[d] == j8 * 24 + e) keep[e] = 0.0f;
  }
  unsigned int lmin = 0x7f800000u, lmax = 0u;
  constexpr int RG = 4;  // row groups of 32 per workgroup pass: 12 loads of 16 bytes in flight per lane
  for (int64_t base = (int64_t)blockIdx.x * (32 * RG); base < n; base += (int64_t)gridDim.x * (32 * RG)) {
    v4i x[RG][3];
    int16_t cc[RG][3];  // lane 0 of a row: its three curve columns (the lines are the ones the row's own loads fetch)
#pragma unroll
    for (int g = 0; g < RG; g++) {
      const int64_t i = min(base + g * 32 + (threadIdx.x >> 3), n - 1);
      const v4i *rp = reinterpret_cast<const v4i *>(feat + i * 192) + j8 * 3;
#pragma unroll
      for (int v = 0; v < 3; v++) x[g][v] = rp[v];
      if (j8 == 0)
#pragma unroll
        for (int d = 0; d < 3; d++) cc[g][d] = feat[i * 192 + cs.col[d]];
    }
#pragma unroll
    for (int g = 0; g < RG; g++) {
      const int64_t i = base + g * 32 + (threadIdx.x >> 3);
      float sq = 0.0f;
#pragma unroll
      for (int v = 0; v < 3; v++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float c0 = (float)(int16_t)(x[g][v][j] & 0xffff) - cen[v * 8 + 2 * j], c1 = (float)(x[g][v][j] >> 16) - cen[v * 8 + 2 * j + 1];
          sq = fmaf(c0 * keep[v * 8 + 2 * j], c0, fmaf(c1 * keep[v * 8 + 2 * j + 1], c1, sq));
        }
      sq += __shfl_xor(sq, 1); sq += __shfl_xor(sq, 2); sq += __shfl_xor(sq, 4);
      if (i < n && j8 == 0) {
        const float lr = sqrtf(fmaxf(sq, 0.0f));
        out[i] = lr;
        ccol[i] = make_uint2((uint32_t)(uint16_t)cc[g][0] | ((uint32_t)(uint16_t)cc[g][1] << 16), (uint32_t)(uint16_t)cc[g][2]);
        lmin = min(lmin, __float_as_uint(lr));
        lmax = max(lmax, __float_as_uint(lr));
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) { lmin = min(lmin, (unsigned)__shfl_xor((int)lmin, o)); lmax = max(lmax, (unsigned)__shfl_xor((int)lmax, o)); }
  // one pair of atomics per workgroup: the two words are the same for the whole launch, and their atomics queue up one behind the other
  __shared__ unsigned int s_rng[2][4];
  if ((threadIdx.x & 63) == 0) { s_rng[0][threadIdx.x >> 6] = lmin; s_rng[1][threadIdx.x >> 6] = lmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMin(&range[0], min(min(s_rng[0][0], s_rng[0][1]), min(s_rng[0][2], s_rng[0][3])));
    atomicMax(&range[1], max(max(s_rng[1][0], s_rng[1][1]), max(s_rng[1][2], s_rng[1][3])));
  }
}

// Morton key, value = row index: the three widest columns at 8 bits each over the union range, plus 8 bits of the radial coordinate
// over ITS range, so that the rows of a tile are alike in texture energy as well as in mean colour -- which is what the radial
// box dimension needs in order to prune (30 % fewer evaluated pairs on the bench clip than a 3 x 10-bit curve of the columns alone).
__global__ void k_curve_keys(const uint2 *__restrict__ ccol /* k_row_radial's copy of the three curve columns */, int64_t n, CurveSpec cs, const float *__restrict__ radial,
                             uint32_t *__restrict__ key, uint32_t *__restrict__ idx) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    // per-dimension bit counts, interleaved from the top: a dimension with more bits splits first
    uint32_t q[4];
    const uint2 c3 = ccol[i];
    const int cv[3] = {(int)(int16_t)(c3.x & 0xffff), (int)(int16_t)(c3.x >> 16), (int)(int16_t)(c3.y & 0xffff)};
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const float v = d < 3 ? (float)cv[d] : (cs.rlog ? log2f(radial[i] + 1.0f) : radial[i]);
      q[d] = (uint32_t)min((float)((1u << cs.bits[d]) - 1u), max(0.0f, (v - cs.off[d]) * cs.scale[d]));
    }
    uint32_t k = 0;
    for (int b = 15; b >= 0; b--)
#pragma unroll
      for (int d = 0; d < 4; d++)
        if (cs.bits[d] > b) k = (k << 1) | ((q[d] >> b) & 1u);
    key[i] = k;
    idx[i] = (uint32_t)i;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Rescan the winning 32-row tile of each query with the plain SSD (CompareEuclideanDCTPtr, utils.pas:541-557) and
// apply the lowest-ORIGINAL-index rule inside it.  One wave per (sorted) query, lanes 0..31 = rows of the tile.
// Queries whose minimum was reached by a second tile (tie flag) are queued for k_knn_ties.
__device__ __forceinline__ uint32_t ssd_rows(const int16_t *__restrict__ a, const int16_t *__restrict__ b) {
  const v4i *qp = reinterpret_cast<const v4i *>(a);
  const v4i *tp = reinterpret_cast<const v4i *>(b);
  uint32_t ssd = 0;
#pragma unroll 4
  for (int v = 0; v < 24; v++) {
    const v4i x = qp[v], y = tp[v];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int d0 = (int)(int16_t)(x[i] & 0xffff) - (int)(int16_t)(y[i] & 0xffff);
      const int d1 = (x[i] >> 16) - (y[i] >> 16);
      ssd += (uint32_t)(d0 * d0) + (uint32_t)(d1 * d1);
    }
  }
  return ssd;
}

__global__ __launch_bounds__(256) void k_knn_refine(int64_t nq, const uint32_t *__restrict__ qperm, int64_t nt,
                                                    const uint32_t *__restrict__ tperm, const uint8_t *__restrict__ qpack, int q_bytes,
                                                    const int *__restrict__ best_key, const int *__restrict__ best_row,
                                                    int *__restrict__ out_idx, uint32_t *__restrict__ out_err,
                                                    uint32_t *__restrict__ tie_list, unsigned int *__restrict__ tie_count) {
  // The kernel already knows the sorted row of the (first) minimum and d'' = SSD - (|q-c|^2 & 1): undo both mappings.
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nq; p += (int64_t)gridDim.x * blockDim.x) {
    const int br = best_row[p];
    const int64_t srow = min((int64_t)(br & 0x3fffffff), nt - 1);  // padded rows replicate row nt-1
    const uint32_t nqv = reinterpret_cast<const uint32_t *>(qpack + (p >> 5) * (int64_t)q_bytes + q_bytes - 128)[p & 31];
    const int64_t q = qperm[p];
    out_idx[q] = (int)tperm[srow];
    out_err[q] = (uint32_t)best_key[p] + (nqv & 1u);
    if ((br & (1 << 30)) || (int64_t)(br & 0x3fffffff) >= nt) tie_list[atomicAdd(tie_count, 1u)] = (uint32_t)p;
  }
}

// Tie settlement: another tile reached the same minimum.  Every row with SSD == best lives in a tile whose box is
// within sqrt(best) of the query, so a box test over all tiles finds the candidates; keep the lowest original index.
// One workgroup per tie, thread = tile.
__global__ __launch_bounds__(256) void k_knn_ties(const int16_t *__restrict__ queries, const uint32_t *__restrict__ qperm,
                                                  const int16_t *__restrict__ db, int64_t nt, int64_t n_ttiles,
                                                  const uint32_t *__restrict__ tperm, KnnBoxes bx,
                                                  const uint32_t *__restrict__ tie_list, const unsigned int *__restrict__ tie_count,
                                                  int *__restrict__ out_idx, const uint32_t *__restrict__ out_err) {
  __shared__ unsigned int s_min;
  __shared__ int s_nt, s_tlist[2048], s_ng, s_glist[256];
  for (unsigned int k = blockIdx.x; k < *tie_count; k += gridDim.x) {
    const uint32_t p = tie_list[k];
    const int64_t q = qperm[p];
    const int16_t *qrow = queries + q * 192;
    const uint32_t best = out_err[q];
    int qv[KNN_NC];  // the tie rescan prunes with the column boxes only
#pragma unroll
    for (int d = 0; d < KNN_NC; d++) qv[d] = qrow[bx.col[d]];
    if (threadIdx.x == 0) { s_min = 0xffffffffu; s_nt = 0; s_ng = 0; }
    __syncthreads();
    // (the scan's winner reaches the minimum: only rows of a LOWER original index can replace it, the others are not even read)
    unsigned int mine = (unsigned int)out_idx[q];
    auto rows = [&](int64_t t, int r0, int r1) {
      for (int r = r0; r < r1; r++) {
        const int64_t sr = t * 32 + r;
        if (sr >= nt) break;
        const uint32_t orow = tperm[sr];
        if (orow < mine && ssd_rows(qrow, db + (int64_t)orow * 192) == best) mine = orow;
      }
    };
    // The runs of KNN_GROUP tiles whose box admits the minimum first (a few of them: the minimum is the NEAREST row's distance), then the
    // tiles of those runs; the surviving tiles go on a list and their rows are spread over the threads.  (One box test per tile of the
    // whole database was 2 ms of the literal bench clip's step: 92 000 ties x 5 300 tiles.)
    auto tile_test = [&](int64_t t) {
      long long lb = 0;
#pragma unroll
      for (int d = 0; d < KNN_NC; d++) {
        const long long g = max(0, max(bx.lo[(int64_t)d * n_ttiles + t] - qv[d], qv[d] - bx.hi[(int64_t)d * n_ttiles + t]));
        lb += g * g;
      }
      if (lb > (long long)best) return;
      const int slot = atomicAdd(&s_nt, 1);
      if (slot < 2048) s_tlist[slot] = (int)t;
      else rows(t, 0, 32);  // list full: this thread takes the tile's rows itself
    };
    const int64_t n_runs = (n_ttiles + KNN_GROUP - 1) / KNN_GROUP;
    for (int64_t g = threadIdx.x; g < n_runs; g += 256) {
      long long lb = 0;
#pragma unroll
      for (int d = 0; d < KNN_NC; d++) {
        const long long e = max(0, max(bx.glo[(int64_t)d * n_runs + g] - qv[d], qv[d] - bx.ghi[(int64_t)d * n_runs + g]));
        lb += e * e;
      }
      if (lb > (long long)best) continue;
      const int slot = atomicAdd(&s_ng, 1);
      if (slot < 256) s_glist[slot] = (int)g;
      else for (int64_t t = g * KNN_GROUP; t < min(n_ttiles, (g + 1) * KNN_GROUP); t++) tile_test(t);  // list full: the thread walks the run itself
    }
    __syncthreads();
    {
      const int runs = min(s_ng, 256);
      for (int e = threadIdx.x; e < runs * KNN_GROUP; e += 256) {
        const int64_t t = (int64_t)s_glist[e / KNN_GROUP] * KNN_GROUP + e % KNN_GROUP;
        if (t < n_ttiles) tile_test(t);
      }
    }
    __syncthreads();
    const int total = min(s_nt, 2048) * 32;
    for (int e = threadIdx.x; e < total; e += 256) rows(s_tlist[e >> 5], e & 31, (e & 31) + 1);
    if (mine != (unsigned int)out_idx[q]) atomicMin(&s_min, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_min != 0xffffffffu) out_idx[q] = (int)s_min;
    __syncthreads();
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
  {
    HostRead hr_(stream);
    TM_TRY(hr_.get(res, scratch.p, sizeof(res)));
    TM_TRY(hr_.wait());
  }
  memcpy(out->mn, res, sizeof(int) * 192);
  memcpy(out->mx, res + 192, sizeof(int) * 192);
  return TM_OK;
}

static void merge_stats(ColStats &a, const ColStats &b) {
  for (int i = 0; i < 192; i++) { a.mn[i] = std::min(a.mn[i], b.mn[i]); a.mx[i] = std::max(a.mx[i], b.mx[i]); }
}

// Per-side digit plan.  For every column pick the centre (midpoint of the query range, of the union or of the database)
// that needs the fewest int8 products, then nest the smaller big-set into the larger one so both are prefixes.
static int make_plan_scaled(const ColStats &ts, const ColStats &qs, KnnPlan *plan, int tscale) {
  bool tb[192], qb[192];
  plan->tscale = tscale;
  for (int c = 0; c < 192; c++) {
    int tlo = ts.mn[c], thi = ts.mx[c], qlo = qs.mn[c], qhi = qs.mx[c];
    if (tlo > thi) { tlo = qlo; thi = qhi; }
    if (qlo > qhi) { qlo = tlo; qhi = thi; }
    if (tlo > thi) { tlo = thi = qlo = qhi = 0; }
    const int ulo = std::min(tlo, qlo), uhi = std::max(thi, qhi);
    // the queries' midpoint first: among centres of equal digit cost it is the one about which the radial box dimension prunes best
    // (measured on the bench clip: 272 instead of 326 tiles read per query group, 1.93 % instead of 2.04 % of the pairs evaluated)
    const int cand[3] = {qlo + (qhi - qlo) / 2, ulo + (uhi - ulo) / 2, tlo + (thi - tlo) / 2};
    int best_cost = 99, best_c = cand[0];
    bool bt = true, bq = true;
    for (int k = 0; k < 3; k++) {
      const int cc = cand[k];
      const bool t2 = (tscale * (thi - cc) > 127) || (tscale * (cc - tlo) > 127), q2 = (qhi - cc > 127) || (cc - qlo > 127);
      const int cost = 1 + (t2 ? 1 : 0) + (q2 ? 1 : 0) + (t2 && q2 ? 1 : 0);
      if (cost < best_cost) { best_cost = cost; best_c = cc; bt = t2; bq = q2; }
    }
    plan->centre[c] = (int16_t)best_c;
    tb[c] = bt;
    qb[c] = bq;
  }
  int nt = 0, nq = 0, nu = 0;
  for (int c = 0; c < 192; c++) { nt += tb[c]; nq += qb[c]; nu += (tb[c] || qb[c]); }
  auto chunks = [](int n) { return (n + 31) / 32; };
  // option A: queries' set inside the database's (database digits widened to the union); option B the other way round
  const int costA = chunks(nu) + 2 * chunks(nq), costB = chunks(nu) + 2 * chunks(nt);
  const bool a = costA <= costB;
  const bool *inner = a ? qb : tb;
  // Inside each class the widest columns come first: a 32-row tile whose values all stay within one digit on a chunk of 32 columns has
  // an all-zero high-digit chunk there, and the scan skips the products with it (tm_knn3_kernel.h) -- with the wide columns (the DC terms,
  // the lowest frequencies) packed into the first chunks, the later chunks are empty for most tiles.
  int order[192];
  for (int c = 0; c < 192; c++) order[c] = c;
  auto halfrange = [&](int c) {
    const int lo = std::min(ts.mn[c] <= ts.mx[c] ? ts.mn[c] : INT_MAX, qs.mn[c] <= qs.mx[c] ? qs.mn[c] : INT_MAX);
    const int hi = std::max(ts.mn[c] <= ts.mx[c] ? ts.mx[c] : INT_MIN, qs.mn[c] <= qs.mx[c] ? qs.mx[c] : INT_MIN);
    return hi >= lo ? std::max(hi - (int)plan->centre[c], (int)plan->centre[c] - lo) : 0;
  };
  std::stable_sort(order, order + 192, [&](int x, int y) { return halfrange(x) > halfrange(y); });
  int p = 0;
  for (int i = 0; i < 192; i++) { const int c = order[i]; if (inner[c]) plan->perm[p++] = (int16_t)c; }
  for (int i = 0; i < 192; i++) { const int c = order[i]; if (!inner[c] && (tb[c] || qb[c])) plan->perm[p++] = (int16_t)c; }
  for (int i = 0; i < 192; i++) { const int c = order[i]; if (!tb[c] && !qb[c]) plan->perm[p++] = (int16_t)c; }
  plan->ht = a ? chunks(nu) : chunks(nt);
  plan->hq = a ? chunks(nq) : chunks(nu);
  plan->nbig_t = nt;
  plan->nbig_q = nq;
  return TM_OK;
}

// does `plan` represent every value of one side's statistics exactly?  (both signs are checked: queries are negated)
static bool plan_covers(const KnnPlan &plan, const ColStats &st, int hch, int scale = 1) {
  for (int p = 0; p < 192; p++) {
    const int c = plan.perm[p];
    if (st.mn[c] > st.mx[c]) continue;
    const int lo = scale * (st.mn[c] - plan.centre[c]), hi = scale * (st.mx[c] - plan.centre[c]);
    if (p >= hch * 32) {
      if (lo < -127 || hi > 127) return false;
    } else {
      if (lo < -32000 || hi > 32000) return false;
    }
  }
  return true;
}

// The database digits doubled whenever the doubled values still fit two digits and cost no more products than the plain plan: the
// scan's block epilogue is 16 vector instructions shorter with them.
static int make_plan(const ColStats &ts, const ColStats &qs, KnnPlan *plan) {
  KnnPlan p2, p1;
  make_plan_scaled(ts, qs, &p2, 2);
  make_plan_scaled(ts, qs, &p1, 1);
  auto cost = [](const KnnPlan &p) { return p.ht + p.hq + std::min(p.ht, p.hq); };
  if (plan_covers(p2, ts, p2.ht, 2) && plan_covers(p2, qs, p2.hq) && cost(p2) <= cost(p1)) { *plan = p2; return TM_OK; }
  *plan = p1;
  return TM_OK;
}

struct tm_knn_index_impl {
  const int16_t *db = nullptr;  // borrowed, like ann_kdtree_create borrows its rows (tilingencoder.pas:4600, 4615-4624)
  int64_t nt = 0;
  ColStats tstats;
  KnnPlan plan;
  bool packed = false;
  DevBuf tpack, qpack, plan_dev, scratch, best_key, best_tile, err_flag;
  DevBuf tperm, tkey, box_lo, box_hi, grp_lo, grp_hi;  // database sorted along the curve, per-tile boxes, boxes of runs of KNN_GROUP tiles
  DevBuf qperm, qkey, skey, skey2, sidx, sort_tmp;  // queries sorted along the curve
  DevBuf rrange, tradial, qradial;                  // radial coordinate of the rows (curve key) and its range
  CurveSpec curve;
  DevBuf tie_list, counters;                        // counters: [0] tie count (u32), [2..3] visited (u64)
  DevBuf tccol, qccol;                              // the rows' three curve columns (k_row_radial -> k_curve_keys)
  DevBuf qmeta;                                     // per query sub-tile: box, home tile, high-chunk mask
  // third scan shape: what the seed kernel leaves for the other two (bests, tie values, bounds) and the groups' tile lists
  DevBuf gbest, gtie, gsmax, segs, nsegs, arena_tile, arena_lb;
  DevBuf thmask;                                    // per database tile: which of its high-digit chunks are not all zero
  uint64_t arena_cap = 0, arena_want = 0;           // list entries the arena holds / the largest cursor a search has reported
  hipEvent_t ev_seed = nullptr, ev_lists = nullptr;
  double last_seed_ms = 0, last_lists_ms = 0, last_consume_ms = 0;
  int64_t last_blocks = 0, last_loads = 0, last_listed = 0;
  int64_t last_visited = 0, last_ties = 0;
  double last_ms = 0;
  int last_kbytes = 0;
  int64_t last_pairs = 0, last_seed_pairs = 0, last_mfma = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  ~tm_knn_index_impl() {
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev_seed) (void)hipEventDestroy(ev_seed);
    if (ev_lists) (void)hipEventDestroy(ev_lists);
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

// rows sorted along the Morton curve: perm (row order) and the sorted keys
// log2(R + 1) per row into `radial` and the running range into ix->rrange (two uint32, reset by the caller)
static int row_radial(tm_knn_index_impl *ix, const void *feat, int64_t n, DevBuf &radial, DevBuf &ccol, hipStream_t stream) {
  TM_TRY(radial.alloc((size_t)std::max<int64_t>(n, 1) * 4));
  TM_TRY(ccol.alloc((size_t)std::max<int64_t>(n, 1) * 8));
  if (n <= 0) return TM_OK;
  hipLaunchKernelGGL(k_row_radial, dim3((unsigned)std::min<int64_t>((n + 127) / 128, 2048)), dim3(256), 0, stream, (const int16_t *)feat, n, ix->curve,
                     ix->plan_dev.as<int16_t>(), radial.as<float>(), ix->rrange.as<unsigned int>(), ccol.as<uint2>());
  TM_HIP(hipGetLastError());
  return TM_OK;
}

static int sort_by_curve(tm_knn_index_impl *ix, const DevBuf &ccol, int64_t n, const DevBuf &radial, DevBuf &perm, DevBuf &keys_sorted, hipStream_t stream) {
  TM_TRY(ix->skey.alloc((size_t)n * 4)); TM_TRY(ix->sidx.alloc((size_t)n * 4));
  TM_TRY(perm.alloc((size_t)n * 4)); TM_TRY(keys_sorted.alloc((size_t)n * 4));
  const int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_curve_keys, dim3(grid), dim3(256), 0, stream, ccol.as<uint2>(), n, ix->curve, radial.as<float>(),
                     ix->skey.as<uint32_t>(), ix->sidx.as<uint32_t>());
  size_t tb = 0;
  TM_HIP(rocprim::radix_sort_pairs(nullptr, tb, ix->skey.as<uint32_t>(), keys_sorted.as<uint32_t>(), ix->sidx.as<uint32_t>(),
                                   perm.as<uint32_t>(), (size_t)n, 0, 32, stream));
  TM_TRY(ix->sort_tmp.alloc(tb));
  TM_HIP(rocprim::radix_sort_pairs(ix->sort_tmp.p, tb, ix->skey.as<uint32_t>(), keys_sorted.as<uint32_t>(), ix->sidx.as<uint32_t>(),
                                   perm.as<uint32_t>(), (size_t)n, 0, 32, stream));
  TM_HIP(hipGetLastError());
  return TM_OK;
}

static int run_pack(tm_knn_index_impl *ix, const void *feat, int64_t n, int negate, int hch, const DevBuf &perm, int with_box,
                    DevBuf &out, hipStream_t stream) {
  const int scale = negate ? 1 : ix->plan.tscale;
  const int64_t ntiles = (n + 31) / 32;
  TM_TRY(out.alloc((size_t)ntiles * knn_tile_bytes(hch, with_box)));
  TM_TRY(ix->err_flag.alloc(sizeof(int)));
  if (negate) TM_TRY(ix->qmeta.alloc((size_t)std::max<int64_t>(ntiles, 1) * 16 * 4));
  else TM_TRY(ix->thmask.alloc((size_t)std::max<int64_t>(ntiles, 1)));
  int grid = (int)std::min<int64_t>(ntiles, 4096);
  hipLaunchKernelGGL(k_knn_pack, dim3(grid), dim3(256), 0, stream, (const int16_t *)feat, n, ntiles, hch, negate, scale,
                     ix->plan_dev.as<int16_t>(), ix->plan_dev.as<int16_t>() + 192, perm.as<uint32_t>(), with_box, ix->curve,
                     ix->box_lo.as<int>(), ix->box_hi.as<int>(), out.as<uint8_t>(), ix->err_flag.as<int>(), negate ? ix->qmeta.as<int>() : nullptr,
                     negate ? nullptr : ix->thmask.as<uint8_t>());
  TM_HIP(hipGetLastError());
  return TM_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// k nearest rows (ann_kdtree_short_search_multi, tilingencoder.pas:1563) on the pruned MFMA scan.
//  1. k_topk_tau: every query's k-th smallest exact SSD among the TOPK_WINDOW database tiles around its position on the
//     curve = an upper bound tau of its true k-th smallest SSD (any k rows give one).
//  2. k_knn_mfma<.., TOPK = true>: the scan with those fixed thresholds; every row with d'' <= tau lands in the query's
//     candidate list (at most `cap` entries, the count keeps running).
//  3. k_topk_select: exact SSD (d'' + the query norm's parity bit), original row index, rank by (SSD, index), first k out.
//     A query whose list overflowed lowers its tau to the k-th smallest of what it did store (still a valid bound) and is
//     scanned again with the other overflowed queries.
#ifndef TM_TOPK_EST_STRIDE
#define TM_TOPK_EST_STRIDE 16  // the sample a large search's first thresholds come from: every 16th row ...
#endif
#ifndef TM_TOPK_EST_K
#define TM_TOPK_EST_K 12       // ... and the distance of its 12th nearest: about 192 rows of the whole database lie within it, give or take 55
#endif
#ifndef TM_TOPK_STEP_SHIFT
#define TM_TOPK_STEP_SHIFT 3  // a first pass's rungs (and a restart's) hang at tau >> this below the threshold
#endif
#ifndef TM_TOPK_WINDOW
#define TM_TOPK_WINDOW 32
#endif
constexpr int TOPK_WINDOW_DEFAULT = TM_TOPK_WINDOW;  // tiles (of 32 rows) sampled for the first threshold (8: 1.45 s, 32: 0.99 s, 128: 1.00 s on the bench clip)

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int topk_dot2(uint32_t a, uint32_t b, int c) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}

// lane = sorted query; the window's rows are wave-uniform (scalar loads); per lane the k smallest distances in LDS [slot][lane]
__global__ __launch_bounds__(64) void k_topk_tau(const uint32_t *__restrict__ queries, const uint32_t *__restrict__ qperm, const uint32_t *__restrict__ qkey,
                                                 int64_t nq, const uint32_t *__restrict__ db, const uint32_t *__restrict__ tperm,
                                                 const uint32_t *__restrict__ tnorm /* |row|^2 in sorted order */,
                                                 const uint32_t *__restrict__ tkey, int64_t nt, int64_t ntt, int k, int window, int *__restrict__ tau) {
  extern __shared__ uint32_t s_d[];  // [k][64]
  const int lane = threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.x * 64, p = p0 + lane;
  const int64_t pq = min(p, nq - 1);
  const uint32_t *qrow = queries + (int64_t)qperm[pq] * 96;
  uint32_t q[96];
#pragma unroll
  for (int j = 0; j < 96; j += 4) {
    const uint4 v = *reinterpret_cast<const uint4 *>(qrow + j);
    q[j] = v.x; q[j + 1] = v.y; q[j + 2] = v.z; q[j + 3] = v.w;
  }
  uint32_t qn = 0;
#pragma unroll
  for (int j = 0; j < 96; j++) qn = (uint32_t)topk_dot2(q[j], q[j], (int)qn);
  // window: the tiles around the curve position of the wave's first query (as round 0 of the scan does for a workgroup)
  const uint32_t k0 = qkey[min(p0, nq - 1)];
  int64_t lo = 0, hi = ntt;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (tkey[mid] <= k0) lo = mid + 1; else hi = mid; }
  int64_t start = max((int64_t)0, lo - 1 - window / 2);
  start = min(start, max((int64_t)0, ntt - window));
  const int64_t r0 = start * 32, r1 = min(nt, (start + window) * 32);
  // The k smallest so far sit in LDS [slot][lane]; what decides whether a row enters is their largest.  The slots are taken in groups
  // of eight with each group's largest (and where it sits) in registers: replacing the largest re-reads ITS group only -- with 64 lanes
  // some lane replaces at almost every row, and a re-scan of all k slots per row was three quarters of this kernel.
  int cnt = 0, mslot = 0;
  uint32_t mx = 0;
  uint32_t gm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int gs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int ngr = (k + 7) >> 3;
  auto regroup = [&](int g) {  // group g's largest and its slot, then the overall ones
    uint32_t m = 0;
    int at = g * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int sl = g * 8 + i;
      const uint32_t v = sl < k ? s_d[sl * 64 + lane] : 0u;
      if (sl < k && (v > m || i == 0)) { m = v; at = sl; }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) if (j == g) { gm[j] = m; gs[j] = at; }
    mx = gm[0]; mslot = gs[0];
#pragma unroll
    for (int j = 1; j < 8; j++) if (j < ngr && gm[j] > mx) { mx = gm[j]; mslot = gs[j]; }
  };
  for (int64_t r = r0; r < r1; r++) {
    const uint32_t *row = db + (int64_t)tperm[r] * 96;
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 96; j++) acc = topk_dot2(q[j], row[j], acc);
    const uint32_t d = qn + tnorm[r] - 2u * (uint32_t)acc;
    if (cnt < k) {
      s_d[cnt * 64 + lane] = d;
      cnt++;
      if (cnt == k)
        for (int g = 0; g < ngr; g++) regroup(g);
    } else if (d < mx) {
      s_d[mslot * 64 + lane] = d;
      regroup(mslot >> 3);
    }
  }
  tau[p] = (cnt >= k && mx < 0x7fffffffu) ? (int)mx : 0x7ffffffe;  // fewer than k rows in the window: everything is a candidate
}

__global__ void k_sorted_row_norms(const int16_t *__restrict__ rows, const uint32_t *__restrict__ perm, int64_t n, uint32_t *__restrict__ norm) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int16_t *r = rows + (int64_t)perm[i] * 192;
    uint32_t s = 0;
    for (int j = 0; j < 192; j++) { const int v = r[j]; s += (uint32_t)(v * v); }
    norm[i] = s;
  }
}

// one wave per (sorted) query: rank its candidates by (SSD, original index); the first k go out in that order.  With member
// lists (grp_off != null) a candidate is a DISTINCT row standing for all its duplicates: every member has the candidate's SSD
// and its own index, so the output positions of a candidate's members start after all members of strictly nearer candidates and
// interleave by index with the members of other candidates at exactly the same SSD (rare).
__global__ __launch_bounds__(64) void k_topk_select(int64_t nq, const uint32_t *__restrict__ qperm, const uint8_t *__restrict__ qpack, int q_bytes,
                                                    const uint32_t *__restrict__ tperm, int64_t nt, const uint2 *__restrict__ cand,
                                                    const int *__restrict__ cand_cnt, int cap, int k, int *__restrict__ tau, const int *__restrict__ tau_in /* the thresholds the scan started from */,
                                                    int *__restrict__ step /* in: the pass's rung spacing; out, overflowed queries: the next pass's */,
                                                    const uint32_t *__restrict__ out_map /* null: qperm */, int32_t *__restrict__ out_idx,
                                                    uint32_t *__restrict__ out_err, uint32_t *__restrict__ ovf_list, unsigned int *__restrict__ ovf_count,
                                                    const uint32_t *__restrict__ grp_off, const uint32_t *__restrict__ grp_members, int nofilter,
                                                    uint32_t *__restrict__ unf_list /* non-null: the thresholds were ESTIMATES (topk_estimate) -- a query with fewer than k rows within its
                                                    threshold goes on this list (count: ovf_count[1]) and is searched again from a bound that holds */) {
  extern __shared__ unsigned long long s_key[];  // [cap rounded up to a power of two]
  __shared__ uint32_t s_mult[64];
  const int64_t p = blockIdx.x;
  if (p >= nq) return;
  const int lane = threadIdx.x;
  const int total = cand_cnt[p], stored = min(total, cap);
  // The scan left its final threshold in tau (it walks down the ladder while rows come in): stored candidates above it cannot be among
  // the k nearest, and dropping them before the sort shrinks it (a full list of 512 typically keeps about a hundred).
  const int th = nofilter ? INT_MAX : tau[p];
  if (total > cap) {
    // The list filled up: the query is scanned again.  Its threshold is the one the scan's ladder ended on; the next pass's ladder hangs eight
    // rungs over the bracket this pass left -- from that threshold down to the rung below it, which did not fill.  (Where the rungs hang is a
    // matter of speed only: every threshold a filled rung gives is a valid bound.  The k-th smallest of the rows that WERE stored, a bound
    // too, is no longer worked out: the first `cap` rows met say little where thousands lie within the threshold, and sorting them for it
    // was most of this kernel's time on such data.)
    if (lane == 0) {
      const int tn = min(th, 0x7ffffffe), t_in = tau_in[p], st = max(1, min(step[p], t_in >> 3));  // (the spacing as the scan clamped it)
      tau[p] = tn;
      // ... unless the threshold ended on the ladder's LOWEST rung: then nothing says how far below it the k-th nearest lies, and a ladder
      // an eighth as wide would only crawl down by its own width per pass: eighths of the threshold again
      const bool lowest = (long long)tn <= (long long)t_in - 7ll * st;
      step[p] = lowest ? max(1, tn >> TM_TOPK_STEP_SHIFT) : max(1, st >> 3);
      ovf_list[atomicAdd(ovf_count, 1u)] = (uint32_t)p;
    }
    return;
  }
  const uint32_t parity = reinterpret_cast<const uint32_t *>(qpack + (p >> 5) * (int64_t)q_bytes + q_bytes - 128)[p & 31] & 1u;
  int n = 0;
  // (the stored candidates are asked for eight chunks of 64 at a time: a chunk per round trip to memory was most of this kernel -- the sort
  // below is 1.5 ms of the bench clip's 23)
  for (int base0 = 0; base0 < stored; base0 += 512) {
    uint2 cbuf[8];
    uint32_t orow[8];
    bool ok[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const int i = base0 + u * 64 + lane; cbuf[u] = i < stored ? cand[p * cap + i] : make_uint2(0x7fffffffu, 0xffffffffu); }
#pragma unroll
    for (int u = 0; u < 8; u++) {  // (the original indices of the rows that pass: gathered together as well)
      const int i = base0 + u * 64 + lane;
      // signed, like the scan's own test: d'' = SSD - parity is -1 for an exact match of a query with an odd norm
      ok[u] = i < stored && (int)cbuf[u].x <= th && (int64_t)cbuf[u].y < nt;  // padded rows of the last tile replicate row nt-1: not rows
      orow[u] = ok[u] ? tperm[cbuf[u].y] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (base0 + u * 64 >= stored) break;  // (uniform)
      const bool valid = ok[u];
      const unsigned long long key = ((unsigned long long)(cbuf[u].x + parity) << 32) | orow[u];
      const unsigned long long m = __ballot(valid);
      if (valid) s_key[n + __popcll(m & ((1ull << lane) - 1ull))] = key;
      n += __popcll(m);
    }
  }
  if (unf_list && n < k) {  // (uniform in the wave.  n counts DISTINCT rows: with member lists k rows may need fewer, the second search only costs time)
    if (lane == 0) unf_list[atomicAdd(ovf_count + 1, 1u)] = (uint32_t)p;
    return;
  }
  if (n > 1024) {
    // the same from LDS, for the long lists the last passes give their few queries (up to 8 192 rows at, or tied with, the k-th distance)
    __syncthreads();
    uint32_t lo = 0xffffffffu, hi = 0;
    for (int i = lane; i < n; i += 64) { const uint32_t v = (uint32_t)(s_key[i] >> 32); lo = min(lo, v); hi = max(hi, v); }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, o)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o)); }
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      int c = 0;
      for (int i0 = 0; i0 < n; i0 += 64) c += __popcll(__ballot(i0 + lane < n && (uint32_t)(s_key[min(i0 + lane, n - 1)] >> 32) <= mid));
      if (c >= k) hi = mid; else lo = mid + 1;
    }
    int m2 = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {  // in place: a chunk's survivors land at or before the chunk
      const unsigned long long key = s_key[min(i0 + lane, n - 1)];
      const bool keep = i0 + lane < n && (uint32_t)(key >> 32) <= lo;
      const unsigned long long m = __ballot(keep);
      if (keep) s_key[m2 + __popcll(m & ((1ull << lane) - 1ull))] = key;
      m2 += __popcll(m);
    }
    n = m2;
  } else if (n > 2 * k) {
    // Only the k smallest matter: the smallest SSD V with k candidates at or below it, by bisection over the values (sixteen keys a lane in
    // registers, a ballot a chunk and step), then only the candidates up to V go through the sort -- a full bitonic sort of several hundred
    // keys in LDS was this kernel's time (~1.2 microseconds of a CU's LDS bandwidth per query).
    __syncthreads();
    uint32_t ssd[16], idx[16];
    uint32_t lo = 0xffffffffu, hi = 0;
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int i = u * 64 + lane;
      const unsigned long long key = i < n ? s_key[i] : ~0ull;
      ssd[u] = (uint32_t)(key >> 32); idx[u] = (uint32_t)key;
      if (i < n) { lo = min(lo, ssd[u]); hi = max(hi, ssd[u]); }
    }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, o)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o)); }
    const int nch = (n + 63) >> 6;
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      int c = 0;
#pragma unroll
      for (int u = 0; u < 16; u++)
        if (u < nch) c += __popcll(__ballot(u * 64 + lane < n && ssd[u] <= mid));
      if (c >= k) hi = mid; else lo = mid + 1;
    }
    __syncthreads();  // every key is in registers
    int m2 = 0;
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (u >= nch) break;  // (uniform)
      const bool keep = u * 64 + lane < n && ssd[u] <= lo;
      const unsigned long long m = __ballot(keep);
      if (keep) s_key[m2 + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)ssd[u] << 32) | idx[u];
      m2 += __popcll(m);
    }
    n = m2;
  }
  int n2 = 64;
  while (n2 < n) n2 <<= 1;
  for (int i = n + lane; i < n2; i += 64) s_key[i] = ~0ull;
  __syncthreads();
  // bitonic sort of the keys (one wave): (SSD, index of the row / of the distinct row's first occurrence) ascending
  for (int ks = 2; ks <= n2; ks <<= 1)
    for (int j = ks >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < (n2 >> 1); t += 64) {  // a lane per compare-exchange: the lower element of pair t (every lane works, not every other one)
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), o = i | j;
        const unsigned long long a = s_key[i], b = s_key[o];
        const bool up = (i & ks) == 0;
        if ((a > b) == up) { s_key[i] = b; s_key[o] = a; }
      }
      __syncthreads();
    }
  const int64_t q = out_map ? out_map[p] : qperm[p];
  // The k nearest ROWS come from the first k candidates: a member of a later candidate has at least k members before it.
  const int m = min(n, k);
  unsigned long long me = ~0ull;
  if (lane < m) me = s_key[lane];
  const bool real = me != ~0ull;
  const uint32_t ssd = (uint32_t)(me >> 32), id = (uint32_t)me;
  if (lane < 64) s_mult[lane] = real ? (grp_off ? grp_off[id + 1] - grp_off[id] : 1u) : 0u;
  __syncthreads();
  if (!real) return;
  uint32_t before = 0;  // members of strictly nearer candidates (without lists: the candidate's own rank)
  bool shared = false;  // another candidate at exactly this SSD
  for (int j = 0; j < m; j++) {
    const unsigned long long o = s_key[j];
    if (o == ~0ull) continue;
    const uint32_t os = (uint32_t)(o >> 32);
    if (grp_off) {
      if (os < ssd) before += s_mult[j];
      else if (os == ssd && j != lane) shared = true;
    } else {
      before += j < lane ? 1u : 0u;
    }
  }
  if (before >= (uint32_t)k) return;
  if (!grp_off) { out_idx[q * k + before] = (int32_t)id; out_err[q * k + before] = ssd; return; }
  const uint32_t o0 = grp_off[id], mult = s_mult[lane];
  for (uint32_t a = 0; a < mult; a++) {
    const uint32_t idx = grp_members[o0 + a];
    uint32_t pos = before + a;
    if (shared) {  // members of the other candidates at this SSD with a smaller index come first
      for (int j = 0; j < m; j++) {
        const unsigned long long o = s_key[j];
        if (j == lane || o == ~0ull || (uint32_t)(o >> 32) != ssd) continue;
        const uint32_t oo = grp_off[(uint32_t)o], om = s_mult[j];
        uint32_t lo = 0, hi = om;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (grp_members[oo + mid] < idx) lo = mid + 1; else hi = mid; }
        pos += lo;
      }
    }
    if (pos >= (uint32_t)k) { if (!shared) break; else continue; }
    out_idx[q * k + pos] = (int32_t)idx;
    out_err[q * k + pos] = ssd;
  }
}
__global__ void k_topk_fill(int32_t *__restrict__ out_idx, uint32_t *__restrict__ out_err, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { out_idx[i] = -1; out_err[i] = 0xffffffffu; }
}

// per query sub-tile: the database tile its first query falls into on the curve (the sub-tile's box is written by k_knn_pack)
__global__ __launch_bounds__(256) void k_knn_qmeta(const uint32_t *__restrict__ qkey, int64_t n_qtiles, KnnBoxes bx, int64_t n_ttiles, int *__restrict__ qmeta) {
  for (int64_t st = (int64_t)blockIdx.x * 256 + threadIdx.x; st < n_qtiles; st += (int64_t)gridDim.x * 256) {
    const uint32_t k0 = qkey[st * 32];  // last tile whose first key <= the sub-tile's first key
    int64_t lo = 0, hi = n_ttiles;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (bx.tkey[mid] <= k0) lo = mid + 1; else hi = mid; }
    qmeta[st * 16 + 7] = (int)max((int64_t)0, lo - 1);
  }
}

#define TM_KNN3_BY_HT(FN)                                          \
  switch (ht) {                                                      \
    case 0: FN<0>(hq, a, stream); break;                             \
    case 1: FN<1>(hq, a, stream); break;                             \
    case 2: FN<2>(hq, a, stream); break;                             \
    case 3: FN<3>(hq, a, stream); break;                             \
    case 4: FN<4>(hq, a, stream); break;                             \
    case 5: FN<5>(hq, a, stream); break;                             \
    default: FN<6>(hq, a, stream); break;                            \
  }
static std::atomic<double> g_list_entries_per_group{640.0};
// diagnostics for the tests (tm_knn_last_plan): the digit plan and mode of the calling thread's last scan (which instantiation of the
// kernels ran), and how many times a scan of this process was repeated because its tile lists outgrew the arena
struct KnnLastPlan { int ht = -1, hq = -1, topk = 0; };
static thread_local KnnLastPlan t_last_plan;
static std::atomic<long long> g_arena_retries{0};
void knn_last_plan(int *ht, int *hq, int *topk, long long *arena_retries) {
  if (ht) *ht = t_last_plan.ht;
  if (hq) *hq = t_last_plan.hq;
  if (topk) *topk = t_last_plan.topk;
  if (arena_retries) *arena_retries = g_arena_retries.load();
}
// the arena's first size: TM_KNN_ARENA_ENTRIES (tests: a tiny arena, so that the repeat-with-the-counted-size path runs) or the experience
static uint64_t arena_first_size(int64_t n_groups, double factor) {
  if (knobs().knn_arena_entries > 0) return (uint64_t)knobs().knn_arena_entries;
  return std::max<uint64_t>(1u << 16, (uint64_t)((double)n_groups * factor * g_list_entries_per_group.load()));
}
static void launch_seed3(int ht, int hq, const Knn3Args &a, hipStream_t stream) { TM_KNN3_BY_HT(knn3_launch_seed_ht) }
static void launch_consume3(int ht, int hq, const Knn3Args &a, hipStream_t stream) { TM_KNN3_BY_HT(knn3_launch_consume_ht) }
static void launch_collect3(int ht, int hq, const Knn3Args &a, hipStream_t stream) { TM_KNN3_BY_HT(knn3_launch_collect_ht) }

static int device_cus() {  // compute units of the current device (persistent kernels launch one workgroup per CU)
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    ncu = std::max(1, prop.multiProcessorCount);
  }
  return ncu;
}

// The third scan shape (tm_knn3_kernel.h): seeds -> lists -> consume, all queued on `stream`; the host looks at nothing in between.  The
// list arena is sized from experience (640 entries per group to begin with; the bench clip needs ~400); a search whose lists did not fit
// is told so by the cursor it reads back with its other counters (knn_index_search) and runs again with the arena the cursor asks for.
static int launch_scan3(tm_knn_index_impl *ix, int64_t nq, int64_t nqt, int64_t ntt, int prune, const KnnBoxes &bx, unsigned long long *stats, hipStream_t stream) {
  const int ns = knn3_sub_tiles(ix->plan.hq), nsp = (ns + 1) & ~1;
  Knn3Args a;
  memset(&a, 0, sizeof(a));  // (collection-mode fields stay off: no_seeds = 0, split = 0)
  a.tpack = ix->tpack.as<uint8_t>(); a.n_ttiles = ntt; a.nt_rows = ix->nt;
  a.box_lo = bx.lo; a.box_hi = bx.hi; a.grp_lo = bx.glo; a.grp_hi = bx.ghi;
  a.qpack = ix->qpack.as<uint8_t>(); a.n_qtiles = nqt; a.nq = nq; a.qmeta = ix->qmeta.as<int>();
  a.thmask = ix->thmask.as<uint8_t>();
  TM_CHECK(ntt < (1 << 24), TM_E_UNSUPPORTED, "knn: %lld database tiles exceed the list entries' 24-bit tile index", (long long)ntt);
  a.ns = ns; a.mode = prune ? K3_MODE_LISTS : K3_MODE_DENSE; a.tdouble = ix->plan.tscale == 2;
  a.n_groups = (nqt + ns - 1) / ns;
  a.max_segs = (int)(ntt / (K3_LCAP - K3_LIST_NT) + 2);  // every segment but a list's last holds more than K3_LCAP - K3_LIST_NT entries
  if (prune) {
    TM_TRY(ix->gbest.alloc((size_t)nqt * 32 * 8)); TM_TRY(ix->gtie.alloc((size_t)nqt * 32 * 4)); TM_TRY(ix->gsmax.alloc((size_t)nqt * 4));
    TM_TRY(ix->segs.alloc((size_t)a.n_groups * a.max_segs * 8)); TM_TRY(ix->nsegs.alloc((size_t)a.n_groups * 4));
    // entries per group: what the process's searches have needed so far (+ 30 %), 640 to begin with -- an index lives for one Reconstruct,
    // the experience is kept beside it
    const uint64_t want = std::max<uint64_t>(ix->arena_want, arena_first_size(a.n_groups, 1.0));
    TM_CHECK(want < (1ull << 32), TM_E_UNSUPPORTED, "knn: %llu list entries exceed the arena's 32-bit offsets", (unsigned long long)want);
    TM_TRY(ix->arena_tile.alloc((size_t)want * 4)); TM_TRY(ix->arena_lb.alloc((size_t)want * nsp * 2));  // (no-ops while they are large enough)
    ix->arena_cap = want;
  }
  a.gbest = ix->gbest.as<unsigned long long>(); a.gtie = ix->gtie.as<unsigned>(); a.gsmax = ix->gsmax.as<unsigned>();
  a.segs = ix->segs.as<uint2>(); a.nsegs = ix->nsegs.as<int>();
  a.ltile = ix->arena_tile.as<unsigned>(); a.llb = ix->arena_lb.as<uint16_t>(); a.arena_cap = ix->arena_cap;
  a.arena_cursor = stats + 18;  // (bytes 160.. of the counters: behind the group tickets)
  a.best_key = ix->best_key.as<int>(); a.best_tile = ix->best_tile.as<int>(); a.stats = stats;
  a.seed_stats = reinterpret_cast<unsigned long long *>(ix->counters.as<uint8_t>() + 256);
  a.grid_blocks = (int)std::min<int64_t>(a.n_groups, (int64_t)device_cus() * K3_WGS);
  a.tickets = reinterpret_cast<unsigned *>(ix->counters.as<uint8_t>() + 128);
  if (prune) {
    launch_seed3(ix->plan.ht, ix->plan.hq, a, stream);
    TM_HIP(hipEventRecord(ix->ev_seed, stream));
    hipLaunchKernelGGL(k_knn_lists, dim3((unsigned)a.n_groups), dim3(K3_LIST_NT), 0, stream, a);
    TM_HIP(hipEventRecord(ix->ev_lists, stream));
  } else {
    TM_HIP(hipEventRecord(ix->ev_seed, stream));
    TM_HIP(hipEventRecord(ix->ev_lists, stream));
  }
  t_last_plan.ht = ix->plan.ht; t_last_plan.hq = ix->plan.hq; t_last_plan.topk = 0;
  launch_consume3(ix->plan.ht, ix->plan.hq, a, stream);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

int knn_index_create(const void *db, int64_t nt, hipStream_t stream, tm_knn_index_impl **out) {
  TM_TRY(require_device());
  TM_CHECK(nt >= 0, TM_E_INVAL, "knn: negative row count");
  auto *ix = new tm_knn_index_impl();
  ix->db = (const int16_t *)db;
  ix->nt = nt;
  int rc = col_stats(db, nt, &ix->tstats, ix->scratch, stream);
  if (rc == TM_OK && (hipEventCreate(&ix->ev0) != hipSuccess || hipEventCreate(&ix->ev1) != hipSuccess || hipEventCreate(&ix->ev_seed) != hipSuccess ||
                      hipEventCreate(&ix->ev_lists) != hipSuccess)) {
    set_error("hipEventCreate failed");
    rc = TM_E_HIP;
  }
  if (rc != TM_OK) { delete ix; return rc; }
  *out = ix;
  return TM_OK;
}

void knn_index_destroy(tm_knn_index_impl *ix) { delete ix; }

// everything a search needs before the scan: digit plan (database repacked if the batch widens it), both sides sorted along
// the curve and packed in MFMA fragment order
static int prepare_search(tm_knn_index_impl *ix, const void *queries, int64_t nq, hipStream_t stream, const void *query_colmm = nullptr) {
  ColStats qs;
  bool fresh_radial = false;  // the queries' radial coordinates were computed while the index was being built
  if (query_colmm) {  // the producer of the queries kept their column ranges
    int res[384];
    {
      HostRead hr_(stream);
      TM_TRY(hr_.get(res, query_colmm, sizeof(res)));
      TM_TRY(hr_.wait());
    }
    memcpy(qs.mn, res, sizeof(int) * 192);
    memcpy(qs.mx, res + 192, sizeof(int) * 192);
  } else
  TM_TRY(col_stats(queries, nq, &qs, ix->scratch, stream));
  TM_TRY(ix->err_flag.alloc(sizeof(int)));
  TM_HIP(hipMemsetAsync(ix->err_flag.p, 0, sizeof(int), stream));  // both pack passes below report into it
  const int64_t ntt = (ix->nt + 31) / 32;
  {  // exactness domain: all arithmetic is mod 2^32 and compared as signed, which needs every SSD < 2^31.  Tile features
     // satisfy it by construction (<= 1.35e9); arbitrary int16 data may not.
    long long bound = 0;
    for (int c = 0; c < 192; c++) {
      const long long lo = std::min(ix->tstats.mn[c], qs.mn[c]), hi = std::max(ix->tstats.mx[c], qs.mx[c]);
      if (hi > lo) bound += (hi - lo) * (hi - lo);
    }
    TM_CHECK(bound < (1ll << 31) - 2, TM_E_UNSUPPORTED,
             "knn: column ranges allow an SSD of %lld >= 2^31, outside the exact domain of the int8/int32 kernel", bound);
  }
  if (!ix->packed || !plan_covers(ix->plan, qs, ix->plan.hq)) {
    TM_TRY(make_plan(ix->tstats, qs, &ix->plan));
    TM_CHECK(plan_covers(ix->plan, ix->tstats, ix->plan.ht, ix->plan.tscale) && plan_covers(ix->plan, qs, ix->plan.hq), TM_E_UNSUPPORTED,
             "knn: feature range exceeds the exact two-digit int8 split");
    if (knobs().knn_debug)
      fprintf(stderr, "[tm_knn] nq=%lld nt=%lld big columns: database %d (digits x%d), queries %d -> HT=%d HQ=%d K=%d bytes\n", (long long)nq,
              (long long)ix->nt, ix->plan.nbig_t, ix->plan.tscale, ix->plan.nbig_q, ix->plan.ht, ix->plan.hq,
              192 + 32 * (ix->plan.ht + ix->plan.hq + std::min(ix->plan.ht, ix->plan.hq)));
    TM_TRY(upload_plan(ix, stream));
    {  // curve + box columns: the KNN_ND widest columns of the union; the first three drive the Morton order
      int order[192];
      for (int c = 0; c < 192; c++) order[c] = c;
      auto urange = [&](int c) {
        const int lo = std::min(ix->tstats.mn[c], qs.mn[c]), hi = std::max(ix->tstats.mx[c], qs.mx[c]);
        return hi >= lo ? hi - lo : 0;
      };
      std::stable_sort(order, order + 192, [&](int a, int b) { return urange(a) > urange(b); });
      for (int d = 0; d < KNN_NC; d++) ix->curve.col[d] = order[d];
      for (int d = 0; d < 3; d++) {
        const int c = order[d];
        ix->curve.lo[d] = std::min(ix->tstats.mn[c], qs.mn[c]);
        ix->curve.range[d] = std::max(1, urange(c));
      }
    }
    {  // radial coordinate of every database row and of this batch of queries; its range (fixed with the index) scales the key's 8 bits
      TM_TRY(ix->rrange.alloc(8));
      const unsigned int init[2] = {0x7f800000u, 0u};
      TM_HIP(hipMemcpyAsync(ix->rrange.p, init, 8, hipMemcpyHostToDevice, stream));
      TM_TRY(row_radial(ix, ix->db, ix->nt, ix->tradial, ix->tccol, stream));
      TM_TRY(row_radial(ix, queries, nq, ix->qradial, ix->qccol, stream));
      unsigned int rr[2];
      {
        HostRead hr_(stream);
        TM_TRY(hr_.get(rr, ix->rrange.p, 8));
        TM_TRY(hr_.wait());
      }
      float rlo, rhi;
      memcpy(&rlo, &rr[0], 4); memcpy(&rhi, &rr[1], 4);
      if (!(rhi > rlo)) { rlo = 0.0f; rhi = 1.0f; }
      CurveSpec &cs = ix->curve;
      // Measured on the bench clip (column ranges 20262 / 13399 / 13118, R in 2566..5284): every dimension over its own range with
      // 8, 7, 7, 8 bits and log2 R -- R cells of 0.3 % -- evaluates 15.3 G pairs (scan 18.4 ms); 8, 8, 8, 8: 14.7 G but 20.0 ms;
      // isotropic cells (9, 8, 8, 6 bits, linear R): 18.6 G, 21.4 ms; columns only (10, 10, 10): 28.9 G, 30.2 ms.
      // The k-nearest scans use the same curve (measured after their kernel stopped spilling: first collection pass of the
      // extended-palette run 112 ms on this curve, 146 ms on 10, 10, 10 bits of the columns alone).
      {
        const int nb[4] = {8, 7, 7, 8};
        cs.rlog = 1;
        for (int d = 0; d < 3; d++) { cs.bits[d] = nb[d]; cs.off[d] = (float)cs.lo[d]; cs.scale[d] = (float)((1 << nb[d]) - 1) / (float)cs.range[d]; }
        cs.bits[3] = nb[3]; cs.off[3] = log2f(rlo + 1.0f);
        cs.scale[3] = ((float)(1 << nb[3]) - 0.001f) / std::max(1e-6f, log2f(rhi + 1.0f) - log2f(rlo + 1.0f));
      }
      if (knobs().knn_debug)
        fprintf(stderr, "[tm_knn] curve: column ranges %d %d %d, radial %.1f..%.1f -> bits %d %d %d %d (%s)\n", cs.range[0], cs.range[1], cs.range[2], rlo, rhi,
                cs.bits[0], cs.bits[1], cs.bits[2], cs.bits[3], "own ranges, log radial");
      fresh_radial = true;
    }
    TM_TRY(sort_by_curve(ix, ix->tccol, ix->nt, ix->tradial, ix->tperm, ix->skey2, stream));
    ix->tccol.release();
    ix->tradial.release();
    TM_TRY(ix->tkey.alloc((size_t)ntt * 4));
    TM_HIP(hipMemcpy2DAsync(ix->tkey.p, 4, ix->skey2.p, 128, 4, (size_t)ntt, hipMemcpyDeviceToDevice, stream));  // key of each tile's first row
    TM_TRY(ix->box_lo.alloc((size_t)ntt * KNN_ND * 4));
    TM_TRY(ix->box_hi.alloc((size_t)ntt * KNN_ND * 4));
    TM_TRY(run_pack(ix, ix->db, ix->nt, 0, ix->plan.ht, ix->tperm, 1, ix->tpack, stream));
    {
      const int64_t ng = (ntt + KNN_GROUP - 1) / KNN_GROUP;
      TM_TRY(ix->grp_lo.alloc((size_t)ng * KNN_ND * 4)); TM_TRY(ix->grp_hi.alloc((size_t)ng * KNN_ND * 4));
      hipLaunchKernelGGL(k_group_boxes, dim3((unsigned)std::min<int64_t>((ng * KNN_ND + 255) / 256, 1024)), dim3(256), 0, stream, ix->box_lo.as<int>(),
                         ix->box_hi.as<int>(), ntt, ng, ix->grp_lo.as<int>(), ix->grp_hi.as<int>());
      TM_HIP(hipGetLastError());
    }
    ix->packed = true;
  }
  if (!fresh_radial) TM_TRY(row_radial(ix, queries, nq, ix->qradial, ix->qccol, stream));  // a later batch on a built index (its range result is not used)
  TM_TRY(sort_by_curve(ix, ix->qccol, nq, ix->qradial, ix->qperm, ix->qkey, stream));
  TM_TRY(run_pack(ix, queries, nq, 1, ix->plan.hq, ix->qperm, 0, ix->qpack, stream));
  return TM_OK;
}

int knn_index_search(tm_knn_index_impl *ix, const void *queries, int64_t nq, void *out_idx, void *out_err, hipStream_t stream, const void *query_colmm) {
  TM_CHECK(ix != nullptr, TM_E_INVAL, "knn: null index");
  TM_CHECK(nq >= 0, TM_E_INVAL, "knn: negative query count");
  if (nq == 0) return TM_OK;
  if (ix->nt == 0) {  // ANN on an empty tree: the caller treats idx outside [0,T) as "none" (tilingencoder.pas:1549-1557)
    TM_HIP(hipMemsetAsync(out_idx, 0xff, (size_t)nq * 4, stream));
    TM_HIP(hipMemsetAsync(out_err, 0xff, (size_t)nq * 4, stream));
    return TM_OK;
  }
  TM_TRY(prepare_search(ix, queries, nq, stream, query_colmm));
  const int64_t nqt = (nq + 31) / 32, ntt = (ix->nt + 31) / 32;
  TM_TRY(ix->best_key.alloc((size_t)nqt * 32 * 4));
  TM_TRY(ix->best_tile.alloc((size_t)nqt * 32 * 4));
  TM_TRY(ix->tie_list.alloc((size_t)nq * 4));
  TM_TRY(ix->counters.alloc(256 + 2048));  // [4..15]: phase stamps of a diagnostic build; bytes 128..159: the group tickets; bytes 256..: the seed kernel's striped counters
  const int prune = knobs().knn_noprune ? 0 : 1;  // diagnostic: full scan with the same kernel (bench.py roofline_dense)
  int *bt = ix->best_tile.as<int>();
  KnnBoxes bx;
  bx.lo = ix->box_lo.as<int>();
  bx.hi = ix->box_hi.as<int>();
  bx.glo = ix->grp_lo.as<int>();
  bx.ghi = ix->grp_hi.as<int>();
  bx.tkey = ix->tkey.as<uint32_t>();
  for (int d = 0; d < KNN_NC; d++) { bx.col[d] = ix->curve.col[d]; bx.cen[d] = ix->plan.centre[ix->curve.col[d]]; }
  unsigned long long *stats = reinterpret_cast<unsigned long long *>(ix->counters.as<uint8_t>() + 16);
  hipLaunchKernelGGL(k_knn_qmeta, dim3((unsigned)std::min<int64_t>((nqt + 255) / 256, 4096)), dim3(256), 0, stream, ix->qkey.as<uint32_t>(), nqt, bx, ntt,
                     ix->qmeta.as<int>());
  TM_HIP(hipGetLastError());
  int flag = 0;
  unsigned long long cnt[32 + 256] = {0};
  [[maybe_unused]] unsigned long long stamps_in_consume[3] = {0, 0, 0};
  for (int attempt = 0;; attempt++) {
  TM_HIP(hipMemsetAsync(ix->counters.p, 0, 256 + 2048, stream));
  TM_HIP(hipEventRecord(ix->ev0, stream));
  TM_TRY(launch_scan3(ix, nq, nqt, ntt, prune, bx, stats, stream));
  TM_HIP(hipGetLastError());
  TM_HIP(hipEventRecord(ix->ev1, stream));
  {
    int grid = (int)std::min<int64_t>((nq + 255) / 256, 4096);
    hipLaunchKernelGGL(k_knn_refine, dim3(grid), dim3(256), 0, stream, nq, ix->qperm.as<uint32_t>(), ix->nt, ix->tperm.as<uint32_t>(),
                       ix->qpack.as<uint8_t>(), knn_tile_bytes(ix->plan.hq, 0), ix->best_key.as<int>(), bt, (int *)out_idx,
                       (uint32_t *)out_err, ix->tie_list.as<uint32_t>(), ix->counters.as<unsigned int>());
    hipLaunchKernelGGL(k_knn_ties, dim3(1024), dim3(256), 0, stream, (const int16_t *)queries, ix->qperm.as<uint32_t>(), ix->db, ix->nt, ntt,
                       ix->tperm.as<uint32_t>(), bx, ix->tie_list.as<uint32_t>(), ix->counters.as<unsigned int>(), (int *)out_idx,
                       (const uint32_t *)out_err);
    TM_HIP(hipGetLastError());
  }
  {
    HostRead hr_(stream);
    TM_TRY(hr_.get(&flag, ix->err_flag.p, sizeof(int)));
    TM_TRY(hr_.get(cnt, ix->counters.p, 256 + 2048));
    TM_TRY(hr_.wait());
  }
  for (int i = 0; i < 3; i++) stamps_in_consume[i] = cnt[12 + i];  // (a diagnostic build's: stats[10..12])
  cnt[12] = cnt[13] = cnt[14] = 0;  // the seed kernel's blocks, tiles read, pairs: summed over its 64 striped slots
  for (int i = 0; i < 64; i++) { cnt[12] += cnt[32 + i * 4]; cnt[13] += cnt[32 + i * 4 + 1]; cnt[14] += cnt[32 + i * 4 + 2]; }
  if (prune) {  // remember what the lists needed (never below the starting guess: a small search says little about the next)
    const int ns_ = knn3_sub_tiles(ix->plan.hq);
    const double per = 1.3 * (double)cnt[20] / (double)std::max<int64_t>(1, (nqt + ns_ - 1) / ns_);
    double cur = g_list_entries_per_group.load();
    while (per > cur && !g_list_entries_per_group.compare_exchange_weak(cur, per)) {}
  }
  TM_CHECK(cnt[29] == 0, TM_E_HIP, "knn: the scan met a corrupted tile list (guard word %llx)", cnt[29]);
  if (prune && cnt[20] > ix->arena_cap) {  // the tile lists did not fit the arena: the cursor says what they need
    TM_CHECK(attempt < 2, TM_E_HIP, "knn: the list arena overflowed again after growing to %llu entries", (unsigned long long)ix->arena_cap);
    if (knobs().knn_debug) fprintf(stderr, "[tm_knn] list arena: %llu entries needed, %llu held -- searching again\n", cnt[20], (unsigned long long)ix->arena_cap);
    ix->arena_want = cnt[20] + cnt[20] / 4;
    g_arena_retries.fetch_add(1);
    continue;
  }
  break;
  }
  TM_CHECK(flag == 0, TM_E_UNSUPPORTED, "knn: feature range exceeds the exact two-digit int8 split (|v-c| >= 32640)");
  float ms = 0;
  TM_HIP(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
  ix->last_ms = ms;
  {  // the three kernels on their own (ev0 | seeds | ev_seed | lists | ev_lists | consume | ev1)
    float a_ = 0, b_ = 0, c_ = 0;
    TM_HIP(hipEventElapsedTime(&a_, ix->ev0, ix->ev_seed));
    TM_HIP(hipEventElapsedTime(&b_, ix->ev_seed, ix->ev_lists));
    TM_HIP(hipEventElapsedTime(&c_, ix->ev_lists, ix->ev1));
    ix->last_seed_ms = a_; ix->last_lists_ms = b_; ix->last_consume_ms = c_;
  }
  ix->last_kbytes = 192 + 32 * (ix->plan.ht + ix->plan.hq + std::min(ix->plan.ht, ix->plan.hq));
  ix->last_visited = (int64_t)cnt[2];
  ix->last_ties = (int64_t)(cnt[0] & 0xffffffffull);
  // pairs actually evaluated: exact (real query, real row) pairs; cnt[12..14]: the seed kernel's blocks, tiles, pairs
  ix->last_pairs = (int64_t)(cnt[4] + cnt[14]);
  ix->last_seed_pairs = (int64_t)cnt[14];
  ix->last_mfma = (int64_t)cnt[21];
  ix->last_blocks = (int64_t)(cnt[2] + cnt[12]); ix->last_loads = (int64_t)(cnt[3] + cnt[13]); ix->last_listed = (int64_t)cnt[5];
  if (knobs().knn_debug) {
    const int nsg = knn3_sub_tiles(ix->plan.hq);
    const int64_t groups = (nqt + nsg - 1) / nsg;
    fprintf(stderr, "[tm_knn] seeds %.3f ms, lists %.3f ms (%.1f entries per group, arena %.0f %% full), consume %.3f ms, %.2f of %d matrix instructions per block\n", ix->last_seed_ms, ix->last_lists_ms,
            (double)cnt[20] / (double)groups, 100.0 * (double)cnt[20] / (double)std::max<uint64_t>(1, ix->arena_cap), ix->last_consume_ms,
            (double)cnt[21] / (double)std::max<unsigned long long>(1, cnt[2]), 6 + ix->plan.ht + ix->plan.hq + std::min(ix->plan.ht, ix->plan.hq));
    fprintf(stderr, "[tm_knn] scan %.3f ms, evaluated %.3f%% of %lld x %lld pairs (%lld blocks; workgroups read %.3f%% of tiles, %.1f per group; %.1f list entries per group), %lld tie settlements\n",
            ms, 100.0 * (double)ix->last_pairs / ((double)nq * (double)ix->nt), (long long)nq, (long long)ix->nt, (long long)ix->last_blocks,
            100.0 * (double)ix->last_loads / ((double)groups * (double)ntt), (double)ix->last_loads / (double)groups, (double)ix->last_listed / (double)groups, (long long)ix->last_ties);
  }
#if TM_KNN3_STAMPS
  {
    static const char *names3[6] = {"prologue + results", "segment load", "consume", "end-of-segment wait", "waiting for the tile", "total"};
    for (int i = 0; i < 6; i++) fprintf(stderr, "[tm_knn3 stamps] %-24s %6.2f %% of the consume kernel's wave time\n", names3[i], 100.0 * (double)cnt[6 + i] / (double)cnt[11]);
    static const char *names3b[3] = {"  of consume: pick", "  of consume: chain", "  of consume: epilogue"};
    for (int i = 0; i < 3; i++) fprintf(stderr, "[tm_knn3 stamps] %-24s %6.2f %% (%.0f ticks per block)\n", names3b[i], 100.0 * (double)stamps_in_consume[i] / (double)cnt[11], (double)stamps_in_consume[i] / (double)std::max<unsigned long long>(1, cnt[2]));
    static const char *names_s[7] = {"set-up", "wait: first slice + tile", "wait: later slices", "blocks", "end barrier", "results", "total"};
    for (int i = 0; i < 7; i++) fprintf(stderr, "[tm_knn3 stamps] seeds: %-24s %6.2f %% of wave time (%.0f clock ticks per wave)\n", names_s[i], 100.0 * (double)cnt[22 + i] / (double)cnt[28],
                                        (double)cnt[22 + i] / (8.0 * (double)((nqt + knn3_sub_tiles(ix->plan.hq) - 1) / knn3_sub_tiles(ix->plan.hq))));
  }
#endif
  return TM_OK;
}


__global__ void k_topk_sorted_aux(const uint32_t *__restrict__ qperm, int64_t n, int64_t n_pad, const int *__restrict__ tau_by_row,
                                  const int *__restrict__ step_by_row, const uint32_t *__restrict__ rowmap, int *__restrict__ tau_sorted,
                                  int *__restrict__ tau_in_sorted, int *__restrict__ step_sorted, uint32_t *__restrict__ map_sorted) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n_pad; p += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t row = qperm[min(p, n - 1)];
    if (tau_by_row) tau_sorted[p] = tau_by_row[row];
    const int tau = tau_sorted[p];
    tau_in_sorted[p] = tau;  // (the scan overwrites tau_sorted with the thresholds it ends on)
    step_sorted[p] = step_by_row ? step_by_row[row] : (tau > 0 ? max(1, tau >> TM_TOPK_STEP_SHIFT) : 0);  // a first pass: rungs at this fraction of the threshold
    if (p < n) map_sorted[p] = rowmap ? rowmap[row] : row;
  }
}
__global__ void k_topk_gather_sub(const int16_t *__restrict__ feats, const uint32_t *__restrict__ qperm, const uint32_t *__restrict__ list, int64_t n,
                                  const int *__restrict__ tau_sorted, const int *__restrict__ step_sorted, const uint32_t *__restrict__ map_sorted,
                                  int16_t *__restrict__ sub, int *__restrict__ sub_tau, int *__restrict__ sub_step, uint32_t *__restrict__ sub_map) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * 24; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = e / 24;
    const int v = (int)(e - j * 24);
    const uint32_t p = list[j];
    reinterpret_cast<uint4 *>(sub)[e] = reinterpret_cast<const uint4 *>(feats + (int64_t)qperm[p] * 192)[v];
    if (v == 0) { sub_tau[j] = tau_sorted[p]; sub_step[j] = step_sorted[p]; sub_map[j] = map_sorted[p]; }
  }
}
__global__ void k_topk_scatter(const int32_t *__restrict__ idx, const uint32_t *__restrict__ err, const uint32_t *__restrict__ map, int64_t n, int k,
                               int32_t *__restrict__ out_idx, uint32_t *__restrict__ out_err) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n * k; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = e / k;
    out_idx[(int64_t)map[j] * k + (e - j * k)] = idx[e];
    out_err[(int64_t)map[j] * k + (e - j * k)] = err[e];
  }
}

// every `stride`-th row of the database: the sample a large search takes its first thresholds from (knn_index_search_topk)
__global__ void k_topk_sample_rows(const int16_t *__restrict__ db, int64_t nt, int stride, int64_t ns, int16_t *__restrict__ out) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < ns * 24; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = e / 24;
    const int v = (int)(e - j * 24);
    reinterpret_cast<uint4 *>(out)[e] = reinterpret_cast<const uint4 *>(db + min(j * stride, nt - 1) * 192)[v];
  }
}
// the sample search's ke-th distance as the full search's first threshold (0xFFFFFFFF: the sample had fewer than ke rows for this query)
__global__ void k_topk_tau_from_sample(const uint32_t *__restrict__ err, int64_t nq, int ke, int *__restrict__ tau) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t e = err[i * ke + ke - 1];
    tau[i] = e < 0x7ffffffeu ? (int)e : 0x7ffffffe;
  }
}

static int topk_pow2(int v) { int r = 64; while (r < v) r <<= 1; return r; }
static int gridn_k(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096)); }

// one scan of `n` query rows (feats) with thresholds (tau_by_row, or the curve-window estimate when null); results go to row
// rowmap[i] (or i) of out_idx / out_err; overflowed queries recurse with their tightened thresholds
struct TopkExpand { const uint32_t *grp_off = nullptr, *grp_members = nullptr; const void *full_db = nullptr; int64_t full_nt = 0; };

static int topk_pass(tm_knn_index_impl *ix, const int16_t *feats, int64_t n, const int *tau_by_row, const int *step_by_row, const uint32_t *rowmap, int k,
                     int32_t *out_idx, uint32_t *out_err, int depth, hipStream_t stream, const TopkExpand &ex, bool estimated = false) {
  const auto t_start = std::chrono::steady_clock::now();
  TM_TRY(prepare_search(ix, feats, n, stream));
  const int64_t nqt = (n + 31) / 32, ntt = (ix->nt + 31) / 32, n_pad = ((nqt + 1) / 2) * 64;
  // candidates a query may store: 512 in the first pass; the passes over the overflowed queries have far fewer queries and take what 24 GB
  // hold, up to 1024 -- the threshold an overflowed query leaves is the k-th smallest of what it STORED, and on data whose distances
  // bunch (the literal bench clip: four in five queries overflow the first pass) 512 stored rows moved it by a third per pass; 4096 made the
  // select kernel's sort the cost instead (a pass of 826 000 queries: 522 ms against 25)
#ifndef TM_TOPK_CAP_LATER
#define TM_TOPK_CAP_LATER 1024
#endif
#ifndef TM_TOPK_CAP_FIRST
#define TM_TOPK_CAP_FIRST 512
#endif
#ifndef TM_TOPK_BUDGET_GIB
#define TM_TOPK_BUDGET_GIB 24  // candidate lists of a pass (also capped at a third of the free device memory)
#endif
  // (from the fourth pass on -- a few thousand queries at most -- up to 8 192: what is left by then are queries with hundreds of rows AT their
  // k-th distance, which no threshold separates; the select stage picks the k smallest of a long list by bisection)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)32 << 30; }
  const int64_t budget = std::max<int64_t>((int64_t)4 << 30, std::min<int64_t>((int64_t)TM_TOPK_BUDGET_GIB << 30, (int64_t)(free_b / 3)));
#ifndef TM_TOPK_CAP_FIRST_SMALLK
#define TM_TOPK_CAP_FIRST_SMALLK 512  // ... of a search for fewer than 32 rows (the sample search of knn_index_search_topk)
#endif
  const int cap = (int)std::max<int64_t>(2 * k, std::min<int64_t>(depth == 0 ? (k < 32 ? TM_TOPK_CAP_FIRST_SMALLK : TM_TOPK_CAP_FIRST) : depth < 3 ? TM_TOPK_CAP_LATER : 8192, budget / (n * 8)));
  DevBuf tau, tau_in, step, map_sorted, cand, cand_cnt, ovf, counter, unf;
  if (estimated) TM_TRY(unf.alloc((size_t)n * 4));
  TM_TRY(tau.alloc((size_t)n_pad * 4)); TM_TRY(tau_in.alloc((size_t)n_pad * 4)); TM_TRY(step.alloc((size_t)n_pad * 4)); TM_TRY(map_sorted.alloc((size_t)n * 4));
  TM_TRY(cand.alloc((size_t)n * cap * 8)); TM_TRY(cand_cnt.alloc((size_t)n * 4));
  TM_TRY(ovf.alloc((size_t)n * 4)); TM_TRY(counter.alloc(16));
  TM_HIP(hipMemsetAsync(cand_cnt.p, 0, (size_t)n * 4, stream));
  TM_HIP(hipMemsetAsync(counter.p, 0, 16, stream));
  if (!tau_by_row) {
    DevBuf tnorm;  // the plan (hence the sort order) can change between passes, so the norms are made per pass: 171 K rows, microseconds
    TM_TRY(tnorm.alloc((size_t)ix->nt * 4));
    hipLaunchKernelGGL(k_sorted_row_norms, dim3(gridn_k(ix->nt)), dim3(256), 0, stream, ix->db, ix->tperm.as<uint32_t>(), ix->nt, tnorm.as<uint32_t>());
    hipLaunchKernelGGL(k_topk_tau, dim3((unsigned)(n_pad / 64)), dim3(64), (size_t)k * 64 * 4, stream, (const uint32_t *)feats, ix->qperm.as<uint32_t>(),
                       ix->qkey.as<uint32_t>(), n, (const uint32_t *)ix->db, ix->tperm.as<uint32_t>(), tnorm.as<uint32_t>(), ix->tkey.as<uint32_t>(), ix->nt, ntt, k,
                       TOPK_WINDOW_DEFAULT, tau.as<int>());
  }
  hipLaunchKernelGGL(k_topk_sorted_aux, dim3(gridn_k(n_pad)), dim3(256), 0, stream, ix->qperm.as<uint32_t>(), n, n_pad, tau_by_row, step_by_row, rowmap, tau.as<int>(),
                     tau_in.as<int>(), step.as<int>(), map_sorted.as<uint32_t>());
  KnnBoxes bx;
  bx.lo = ix->box_lo.as<int>();
  bx.hi = ix->box_hi.as<int>();
  bx.glo = ix->grp_lo.as<int>();
  bx.ghi = ix->grp_hi.as<int>();
  bx.tkey = ix->tkey.as<uint32_t>();
  for (int d = 0; d < KNN_NC; d++) { bx.col[d] = ix->curve.col[d]; bx.cen[d] = ix->plan.centre[ix->curve.col[d]]; }
  {
    // The third scan shape in collection mode (tm_knn3_kernel.h): bounds from the thresholds, tile lists judged against them (no seeds:
    // every tile goes through the lists), then the consume kernel appending every row within its query's threshold.
    const int ns = knn3_sub_tiles_topk(ix->plan.hq), nsp = (ns + 1) & ~1;
    hipLaunchKernelGGL(k_knn_qmeta, dim3((unsigned)std::min<int64_t>((nqt + 255) / 256, 4096)), dim3(256), 0, stream, ix->qkey.as<uint32_t>(), nqt, bx, ntt,
                       ix->qmeta.as<int>());
    TM_TRY(ix->counters.alloc(256 + 2048));
    Knn3Args a;
    memset(&a, 0, sizeof(a));
    a.tpack = ix->tpack.as<uint8_t>(); a.n_ttiles = ntt; a.nt_rows = ix->nt;
    a.box_lo = bx.lo; a.box_hi = bx.hi; a.grp_lo = bx.glo; a.grp_hi = bx.ghi;
    a.qpack = ix->qpack.as<uint8_t>(); a.n_qtiles = nqt; a.nq = n; a.qmeta = ix->qmeta.as<int>();
    a.thmask = ix->thmask.as<uint8_t>();
    TM_CHECK(ntt < (1 << 24), TM_E_UNSUPPORTED, "knn: %lld database tiles exceed the list entries' 24-bit tile index", (long long)ntt);
    a.ns = ns; a.mode = K3_MODE_LISTS; a.tdouble = ix->plan.tscale == 2;
    a.n_groups = (nqt + ns - 1) / ns;
    a.max_segs = (int)(ntt / (K3_LCAP - K3_LIST_NT) + 2);
    a.no_seeds = 1;
    a.tau = tau.as<int>(); a.step = step.as<int>(); a.cand = cand.as<uint2>(); a.cand_cnt = cand_cnt.as<int>(); a.cand_cap = cap; a.cand_k = k;
    // few queries left: their few workgroups would each walk most of the database one after the other -- share the tile lists
    a.split = a.n_groups >= 512 ? 1 : (int)std::max<int64_t>(1, std::min<int64_t>(64, 1024 / std::max<int64_t>(a.n_groups, 1)));
    TM_TRY(ix->gsmax.alloc((size_t)nqt * 4));
    TM_TRY(ix->segs.alloc((size_t)a.n_groups * a.max_segs * 8)); TM_TRY(ix->nsegs.alloc((size_t)a.n_groups * 4));
    a.gsmax = ix->gsmax.as<unsigned>(); a.segs = ix->segs.as<uint2>(); a.nsegs = ix->nsegs.as<int>();
    unsigned long long *stats = reinterpret_cast<unsigned long long *>(ix->counters.as<uint8_t>() + 16);
    a.stats = stats; a.seed_stats = nullptr; a.arena_cursor = stats + 18;
    a.tickets = reinterpret_cast<unsigned *>(ix->counters.as<uint8_t>() + 128);
    a.grid_blocks = (int)std::min<int64_t>(a.n_groups * a.split, (int64_t)device_cus() * K3_WGS);
    hipLaunchKernelGGL(k_knn_tau_bounds, dim3((unsigned)std::min<int64_t>((nqt + 7) / 8, 2048)), dim3(256), 0, stream, tau.as<int>(), n, nqt, a.gsmax);
    for (int attempt = 0;; attempt++) {
      // collection lists are long (a threshold from 32 tiles of the curve is loose): twice the nearest-neighbour search's experience to begin with
      const uint64_t want = std::max<uint64_t>(ix->arena_want, arena_first_size(a.n_groups, 2.0));
      TM_CHECK(want < (1ull << 32), TM_E_UNSUPPORTED, "knn: %llu list entries exceed the arena's 32-bit offsets", (unsigned long long)want);
      TM_TRY(ix->arena_tile.alloc((size_t)want * 4)); TM_TRY(ix->arena_lb.alloc((size_t)want * nsp * 2));
      ix->arena_cap = want;
      a.ltile = ix->arena_tile.as<unsigned>(); a.llb = ix->arena_lb.as<uint16_t>(); a.arena_cap = ix->arena_cap;
      TM_HIP(hipMemsetAsync(ix->counters.p, 0, 256, stream));
      hipLaunchKernelGGL(k_knn_lists, dim3((unsigned)a.n_groups), dim3(K3_LIST_NT), 0, stream, a);
      unsigned long long cursor = 0;
      {  // the lists must fit before anything is collected through them (a second collection pass would double the candidates)
        HostRead hr_(stream);
        TM_TRY(hr_.get(&cursor, stats + 18, 8));
        TM_TRY(hr_.wait());
      }
      if (cursor <= ix->arena_cap) break;
      TM_CHECK(attempt < 2, TM_E_HIP, "knn: the list arena overflowed again after growing to %llu entries", (unsigned long long)ix->arena_cap);
      ix->arena_want = cursor + cursor / 4;
      g_arena_retries.fetch_add(1);
    }
    t_last_plan.ht = ix->plan.ht; t_last_plan.hq = ix->plan.hq; t_last_plan.topk = 1;
    launch_collect3(ix->plan.ht, ix->plan.hq, a, stream);
    TM_HIP(hipGetLastError());
  }
  if ((size_t)topk_pow2(cap) * 8 > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_topk_select), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)topk_pow2(cap) * 8));
  hipLaunchKernelGGL(k_topk_select, dim3((unsigned)n), dim3(64), (size_t)topk_pow2(cap) * 8, stream, n, ix->qperm.as<uint32_t>(), ix->qpack.as<uint8_t>(),
                     knn_tile_bytes(ix->plan.hq, 0), ix->tperm.as<uint32_t>(), ix->nt, cand.as<uint2>(), cand_cnt.as<int>(), cap, k, tau.as<int>(), tau_in.as<int>(), step.as<int>(),
                     map_sorted.as<uint32_t>(), out_idx, out_err, ovf.as<uint32_t>(), counter.as<unsigned int>(), ex.grp_off, ex.grp_members,
                     0, estimated ? unf.as<uint32_t>() : (uint32_t *)nullptr);
  TM_HIP(hipGetLastError());
  unsigned int novf = 0, nunf = 0;
  int flag = 0;
  unsigned long long guard = 0;
  {
    HostRead hr_(stream);
    TM_TRY(hr_.get(&novf, counter.p, 4));
    TM_TRY(hr_.get(&nunf, counter.as<uint8_t>() + 4, 4));
    TM_TRY(hr_.get(&flag, ix->err_flag.p, sizeof(int)));
    TM_TRY(hr_.get(&guard, ix->counters.as<uint8_t>() + 16 + 27 * 8, 8));
    TM_TRY(hr_.wait());
  }
  TM_CHECK(guard == 0, TM_E_HIP, "knn: the collection scan met a corrupted tile list (guard word %llx)", guard);
  TM_CHECK(flag == 0, TM_E_UNSUPPORTED, "knn: feature range exceeds the exact two-digit int8 split (|v-c| >= 32640)");
  if (knobs().knn_debug)
    fprintf(stderr, "[tm_knn] top-%d pass %d: %lld queries of %lld rows, cap %d, %u overflowed, %u fell short of their estimate, %.1f ms\n", k, depth, (long long)n,
            (long long)ix->nt, cap, novf, nunf, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  if (novf == 0 && nunf == 0) return TM_OK;
  // (both subsets are gathered before either is searched: a search re-sorts the index's query side)
  DevBuf usub, usub_tau, usub_step, usub_map;
  if (nunf > 0) {
    TM_TRY(usub.alloc((size_t)nunf * 384)); TM_TRY(usub_tau.alloc((size_t)nunf * 4)); TM_TRY(usub_step.alloc((size_t)nunf * 4)); TM_TRY(usub_map.alloc((size_t)nunf * 4));
    hipLaunchKernelGGL(k_topk_gather_sub, dim3(gridn_k((int64_t)nunf * 24)), dim3(256), 0, stream, feats, ix->qperm.as<uint32_t>(), unf.as<uint32_t>(),
                       (int64_t)nunf, tau.as<int>(), step.as<int>(), map_sorted.as<uint32_t>(), usub.as<int16_t>(), usub_tau.as<int>(), usub_step.as<int>(), usub_map.as<uint32_t>());
    TM_HIP(hipGetLastError());
  }
  DevBuf sub, sub_tau, sub_step, sub_map;
  if (novf > 0) {
    TM_TRY(sub.alloc((size_t)novf * 384)); TM_TRY(sub_tau.alloc((size_t)novf * 4)); TM_TRY(sub_step.alloc((size_t)novf * 4)); TM_TRY(sub_map.alloc((size_t)novf * 4));
    hipLaunchKernelGGL(k_topk_gather_sub, dim3(gridn_k((int64_t)novf * 24)), dim3(256), 0, stream, feats, ix->qperm.as<uint32_t>(), ovf.as<uint32_t>(),
                       (int64_t)novf, tau.as<int>(), step.as<int>(), map_sorted.as<uint32_t>(), sub.as<int16_t>(), sub_tau.as<int>(), sub_step.as<int>(), sub_map.as<uint32_t>());
    TM_HIP(hipGetLastError());
  }
  cand.release();  // the recursions allocate their own
  if (nunf > 0)  // from the curve window's bound (any k rows give one), as a search without estimates starts
    TM_TRY(topk_pass(ix, usub.as<int16_t>(), nunf, nullptr, nullptr, usub_map.as<uint32_t>(), k, out_idx, out_err, depth + 1, stream, ex));
  if (novf == 0) return TM_OK;
  // Every pass cuts the bracket its ladder spans to an eighth (or, from the lowest rung, the threshold itself): a dozen passes take any threshold
  // down to single units.  What still overflows then has more rows at exactly the k-th distance than a list holds: exact brute force for those.
  if (depth >= 12 || (depth >= 6 && (int64_t)novf * 10 > n * 9)) {
    DevBuf bi, be;
    TM_TRY(bi.alloc((size_t)novf * k * 4)); TM_TRY(be.alloc((size_t)novf * k * 4));
    TM_TRY(launch_knn_topk(sub.p, novf, ex.full_db ? ex.full_db : (const void *)ix->db, ex.full_db ? ex.full_nt : ix->nt, k, bi.p, be.p, stream));
    hipLaunchKernelGGL(k_topk_scatter, dim3(gridn_k((int64_t)novf * k)), dim3(256), 0, stream, bi.as<int32_t>(), be.as<uint32_t>(), sub_map.as<uint32_t>(),
                       (int64_t)novf, k, out_idx, out_err);
    TM_HIP(hipGetLastError());
    TM_HIP(hipStreamSynchronize(stream));
    return TM_OK;
  }
  return topk_pass(ix, sub.as<int16_t>(), novf, sub_tau.as<int>(), sub_step.as<int>(), sub_map.as<uint32_t>(), k, out_idx, out_err, depth + 1, stream, ex);
}

int knn_index_search_topk(tm_knn_index_impl *ix, const void *queries, int64_t nq, int k, void *out_idx, void *out_err, hipStream_t stream,
                          const void *grp_off, const void *grp_members, const void *full_db, int64_t full_nt) {
  TM_CHECK(ix != nullptr, TM_E_INVAL, "knn: null index");
  TM_CHECK(k >= 1 && k <= 64, TM_E_INVAL, "top-k: k %d outside 1..64", k);
  if (nq <= 0) return TM_OK;
  hipLaunchKernelGGL(k_topk_fill, dim3(gridn_k(nq * k)), dim3(256), 0, stream, (int32_t *)out_idx, (uint32_t *)out_err, nq * k);
  TM_HIP(hipGetLastError());
  if (ix->nt == 0) return TM_OK;
  TopkExpand ex;
  ex.grp_off = (const uint32_t *)grp_off; ex.grp_members = (const uint32_t *)grp_members; ex.full_db = full_db; ex.full_nt = full_nt;
  // Many queries against a database of some size: the first thresholds come from a SAMPLE of the database.  The curve window's bound (the k-th
  // smallest of 1 024 rows near the query on the curve) holds but is loose -- on the literal bench clip the 512-th nearest row is 5 % farther
  // than the 64-th, a bound that is off by a factor two lets thousands of rows in, four queries in five overflowed their lists and took three
  // more passes to bracket their k-th distance.  The ke-th nearest row among every S-th row of the database is an ESTIMATE of the (ke S)-th
  // nearest row's distance whatever the distances' law is (the rows within it number ke S give or take S sqrt(ke)): no bound, so a query that
  // finds fewer than k rows within it is searched again the old way (k_topk_select's list of those), but nearly all find between k and the
  // list's capacity at once.  The sample's own search is this same function on a sixteenth of the rows (where the curve window is a third of
  // the database and its bound is good).
  const int est = knobs().topk_estimate;  // -1: by size, 0: never, 1: whenever the sample has ke rows
  constexpr int S = TM_TOPK_EST_STRIDE, KE = TM_TOPK_EST_K;
  const int64_t ns = (ix->nt + S - 1) / S;
  if (est != 0 && k >= 32 && ns >= 4 * KE && (est == 1 || (ix->nt >= 16384 && nq >= 4 * ix->nt))) {
    DevBuf srows, eidx, eerr, tau_est;
    TM_TRY(srows.alloc((size_t)ns * 384)); TM_TRY(eidx.alloc((size_t)nq * KE * 4)); TM_TRY(eerr.alloc((size_t)nq * KE * 4)); TM_TRY(tau_est.alloc((size_t)nq * 4));
    hipLaunchKernelGGL(k_topk_sample_rows, dim3(gridn_k(ns * 24)), dim3(256), 0, stream, ix->db, ix->nt, S, ns, srows.as<int16_t>());
    TM_HIP(hipGetLastError());
    tm_knn_index_impl *six = nullptr;
    TM_TRY(knn_index_create(srows.p, ns, stream, &six));
    const int rc = knn_index_search_topk(six, queries, nq, KE, eidx.p, eerr.p, stream, nullptr, nullptr, nullptr, 0);
    if (rc == TM_OK) TM_HIP(hipStreamSynchronize(stream));  // (the sample index owns scratch the stream may still read)
    knn_index_destroy(six);
    if (rc != TM_OK) return rc;
    hipLaunchKernelGGL(k_topk_tau_from_sample, dim3(gridn_k(nq)), dim3(256), 0, stream, eerr.as<uint32_t>(), nq, KE, tau_est.as<int>());
    TM_HIP(hipGetLastError());
    eidx.release(); srows.release();
    return topk_pass(ix, (const int16_t *)queries, nq, tau_est.as<int>(), nullptr, nullptr, k, (int32_t *)out_idx, (uint32_t *)out_err, 0, stream, ex, true);
  }
  return topk_pass(ix, (const int16_t *)queries, nq, nullptr, nullptr, nullptr, k, (int32_t *)out_idx, (uint32_t *)out_err, 0, stream, ex);
}

void knn_index_kernel_split(tm_knn_index_impl *ix, double ms[3], int64_t pairs[3]) {
  ms[0] = ix->last_seed_ms; ms[1] = ix->last_lists_ms; ms[2] = ix->last_consume_ms;
  pairs[0] = ix->last_seed_pairs; pairs[1] = ix->last_pairs - ix->last_seed_pairs; pairs[2] = ix->last_mfma;
}

void knn_index_stats(tm_knn_index_impl *ix, double *ms, int *kbytes, int64_t *pairs) {
  if (ms) *ms = ix->last_ms;
  if (kbytes) *kbytes = ix->last_kbytes;
  if (pairs) *pairs = ix->last_pairs;
}

int knn3_sub_tiles(int hq) { return k3_ns(6 + std::min(std::max(hq, 0), 6)); }
int knn3_sub_tiles_topk(int hq) { return k3_ns_topk(6 + std::min(std::max(hq, 0), 6)); }

}  // namespace tmx
