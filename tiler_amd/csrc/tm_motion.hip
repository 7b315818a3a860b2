// tm_motion.hip -- (f)#1 motion prediction: the search of PredictMotion.DoXY (tilingencoder.pas:1184-1264) and of the
// redo inside Reconstruct.DoXY (1496-1532), the KNN-vs-motion decision and frame-buffer drawing (1534-1654), frame
// buffers, and the device side of Reduce's tile-count search (STCGREval 4014-4041).
//
// The distance is CompareEuclideanDCTPtr_asm AS WRITTEN (utils.pas:559-725), not the true L2: per 96-coefficient half,
// block 5 (8 coefficients) never enters, block 6's difference is a6 - b5 - b6 (two saturating subtractions), and the
// never-loaded xmm7 contributes pmaddwd(xmm7): 0 in the first half (entry value taken as 0, the only choice that does
// not depend on the caller's registers), and in the second half the first half's block-7 pair sums re-squared as int16
// pairs.  All subtractions saturate (psubsw), sums wrap mod 2^32 (paddd).  oracle/tm_oracle.c:tmo_ssd_i16_sse_quirk
// states the same thing in scalar C.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {
namespace {

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t sat_sub2(uint32_t a, uint32_t b) {  // psubsw on one dword (two int16)
  const s16x2 r = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b));
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t sq2(uint32_t d) {  // pmaddwd of a dword with itself, mod 2^32 (the three-operand form: no accumulator to zero first)
  uint32_t r;
  asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(r) : "v"(d));
  return r;
}
__device__ __forceinline__ uint32_t sq2acc(uint32_t d, uint32_t acc) {  // acc + pmaddwd(d, d): v_dot2c accumulates in place
  return (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, d), __builtin_bit_cast(s16x2, d), (int)acc, false);
}
__device__ __forceinline__ uint32_t block_term(const uint4 a, const uint4 b, uint32_t acc = 0) {  // acc + the block's eight squared saturated differences
  acc = sq2acc(sat_sub2(a.x, b.x), acc);
  acc = sq2acc(sat_sub2(a.y, b.y), acc);
  acc = sq2acc(sat_sub2(a.z, b.z), acc);
  return sq2acc(sat_sub2(a.w, b.w), acc);
}

constexpr int MS_TB = 2;        // MS_TB x MS_TB tiles per workgroup share one sweep over the union of their windows
constexpr int MS_GROUPS = 32;   // 8 lanes per candidate, 256 threads

// One workgroup = a 2x2 block of tiles.  8 lanes share a candidate position: lane j holds 3 of its 24 eight-coefficient
// blocks (48 contiguous bytes, so a wave reads 8 consecutive candidates = 3 KB contiguous per load); the quirks are all
// local to a lane: lanes 1 and 5 skip their third block (block 5 of each half), lanes 2 and 6 fetch that block of b as
// well for the double subtraction, lane 2 adds the re-squared pair sums of block 7.
__global__ __launch_bounds__(256) void k_motion_search(const int16_t *__restrict__ cur, int tm_w, int tm_h,
                                                       const int16_t *__restrict__ win, int r, uint32_t *__restrict__ best_err,
                                                       int8_t *__restrict__ out_px, int8_t *__restrict__ out_py, const int *__restrict__ only_if = nullptr) {
  if (only_if && !*only_if) return;  // the matrix-core search (k_mo_search_mfma) took this frame
  __shared__ uint32_t s_err[MS_TB * MS_TB][MS_GROUPS];
  __shared__ int s_pos[MS_TB * MS_TB][MS_GROUPS];
  const int tid = threadIdx.x, j8 = tid & 7, grp = tid >> 3;
  const int bw = (tm_w + MS_TB - 1) / MS_TB;
  const int by = blockIdx.x / bw, bx = blockIdx.x - by * bw;
  const int sw = tm_w * 8, sh = tm_h * 8, ww = sw - 7;
  const int role = j8 & 3;  // position of this lane's three blocks inside its half: blocks 3*role .. 3*role+2

  uint4 a[MS_TB * MS_TB][3];
  int dx[MS_TB * MS_TB], dy[MS_TB * MS_TB];
  bool valid[MS_TB * MS_TB];
  uint32_t best[MS_TB * MS_TB];
  int bpos[MS_TB * MS_TB];
#pragma unroll
  for (int t = 0; t < MS_TB * MS_TB; t++) {
    const int sy = by * MS_TB + t / MS_TB, sx = bx * MS_TB + t % MS_TB;
    valid[t] = sy < tm_h && sx < tm_w;
    dx[t] = sx * 8; dy[t] = sy * 8;
    best[t] = 0xffffffffu; bpos[t] = 0x7fffffff;
    if (valid[t]) {
      const uint4 *pa = reinterpret_cast<const uint4 *>(cur + ((int64_t)sy * tm_w + sx) * 192) + j8 * 3;
      a[t][0] = pa[0]; a[t][1] = pa[1]; a[t][2] = pa[2];
    } else {
      a[t][0] = a[t][1] = a[t][2] = make_uint4(0, 0, 0, 0);
    }
  }
  // each tile's window (1218-1221: oymn = max(0, dy - r - 1), oymx = min(sh - 8, dy + r); same in x) as start + unsigned
  // extent: one subtract-and-compare per axis in the loop; an absent tile gets an empty window
  unsigned wy0[MS_TB * MS_TB], wyr[MS_TB * MS_TB], wx0[MS_TB * MS_TB], wxr[MS_TB * MS_TB];
#pragma unroll
  for (int t = 0; t < MS_TB * MS_TB; t++) {
    const int y0 = max(0, dy[t] - r - 1), y1 = min(sh - 8, dy[t] + r), x0 = max(0, dx[t] - r - 1), x1 = min(sw - 8, dx[t] + r);
    wy0[t] = (unsigned)y0; wyr[t] = valid[t] ? (unsigned)(y1 - y0) : 0u;
    wx0[t] = valid[t] ? (unsigned)x0 : 0x40000000u; wxr[t] = (unsigned)(x1 - x0);
  }
  // union of the windows
  const int ly = std::min(by * MS_TB + MS_TB - 1, tm_h - 1) * 8, lx = std::min(bx * MS_TB + MS_TB - 1, tm_w - 1) * 8;
  const int uy0 = max(0, by * MS_TB * 8 - r - 1), uy1 = min(sh - 8, ly + r);
  const int ux0 = max(0, bx * MS_TB * 8 - r - 1), ux1 = min(sw - 8, lx + r);
  const int ucols = ux1 - ux0 + 1, ucount = (uy1 - uy0 + 1) * ucols;

  // branch-free inner loop: the per-lane quirk roles become selects, the 8-lane sums go through DPP (quad swaps, then the
  // half-row mirror pairs lane i with 7 - i; after the quad steps every lane of a quad holds the quad's sum)
  const bool r2 = role == 2, r1 = role == 1, l2 = j8 == 2;
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  for (int c = grp; c < ucount; c += MS_GROUPS) {
    const int row = c / ucols, oy = uy0 + row, ox = ux0 + (c - row * ucols);
    const uint4 *pb = reinterpret_cast<const uint4 *>(win + ((int64_t)oy * ww + ox) * 192) + j8 * 3;
    const uint4 b0 = pb[0], b1 = pb[1], b2 = pb[2];
    const uint4 b5raw = pb[r2 ? -1 : 0];
    const uint4 b5 = r2 ? b5raw : zero4;
#pragma unroll
    for (int t = 0; t < MS_TB * MS_TB; t++) {
      const bool in = (unsigned)oy - wy0[t] <= wyr[t] && (unsigned)ox - wx0[t] <= wxr[t];
      // first block of the lane: for role 2 it is block 6 -> (a6 -sat b5) -sat b6; b5 = 0 for the other roles: a -sat 0 = a
      uint4 d;
      d.x = sat_sub2(a[t][0].x, b5.x); d.y = sat_sub2(a[t][0].y, b5.y); d.z = sat_sub2(a[t][0].z, b5.z); d.w = sat_sub2(a[t][0].w, b5.w);
      // second block: for lane 2 (first half, role 2) it is block 7, whose pair sums come back re-squared in the second half
      const uint32_t p0 = sq2(sat_sub2(a[t][1].x, b1.x)), p1 = sq2(sat_sub2(a[t][1].y, b1.y)), p2 = sq2(sat_sub2(a[t][1].z, b1.z)),
                     p3 = sq2(sat_sub2(a[t][1].w, b1.w));
      const uint32_t resq = sq2acc(p3, sq2acc(p2, sq2acc(p1, sq2(p0))));
      // third block: block 5 of each half (role 1) never enters
      const uint32_t third = block_term(a[t][2], b2);
      uint32_t acc = block_term(d, b0, (p0 + p1) + (p2 + p3));
      acc += (l2 ? resq : 0u) + (r1 ? 0u : third);
      acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
      acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
      acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x141, 0xf, 0xf, false);  // row_half_mirror
      uint32_t err;  // + manhattan penalty (1236): two v_sad_u32 (|ox - dx| + |oy - dy| + acc; all four coordinates are >= 0)
      asm("v_sad_u32 %0, %1, %2, %3" : "=v"(err) : "v"(ox), "v"(dx[t]), "v"(acc));
      asm("v_sad_u32 %0, %1, %2, %3" : "=v"(err) : "v"(oy), "v"(dy[t]), "v"(err));
      const bool take = in && err < best[t];  // candidates come in raster order: first minimum stays
      best[t] = take ? err : best[t];
      bpos[t] = take ? c : bpos[t];
    }
  }
  if (j8 == 0) {
#pragma unroll
    for (int t = 0; t < MS_TB * MS_TB; t++) { s_err[t][grp] = best[t]; s_pos[t][grp] = bpos[t]; }
  }
  __syncthreads();
  if (tid < MS_TB * MS_TB) {
    const int t = tid;
    const int sy = by * MS_TB + t / MS_TB, sx = bx * MS_TB + t % MS_TB;
    if (sy < tm_h && sx < tm_w) {
      uint32_t be = 0xffffffffu;
      int bp = 0x7fffffff;
      for (int g = 0; g < MS_GROUPS; g++) {
        const uint32_t e = s_err[t][g];
        const int p = s_pos[t][g];
        if (e < be || (e == be && p < bp)) { be = e; bp = p; }
      }
      const int64_t i = (int64_t)sy * tm_w + sx;
      best_err[i] = be;
      const int row = bp / ucols, oy = uy0 + row, ox = ux0 + (bp - row * ucols);
      out_px[i] = (int8_t)(ox - sx * 8);
      out_py[i] = (int8_t)(oy - sy * 8);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The search on the matrix cores.  Of the 24 blocks of CompareEuclideanDCTPtr_asm 20 are plain squared differences (blocks 0-4 and 7-11
// of each half; block 7 of the first half ALSO feeds the re-squared pair sums); whenever no psubsw can saturate there -- every
// coefficient of those blocks within +-16383, checked on the data of every frame -- their sum is |a|^2 + |b|^2 - 2 a.b with everything
// mod 2^32, exactly what paddd's wrap-around gives.  a.b over the 160 plain coefficients goes through v_mfma_i32_32x32x32_i8 with both
// sides split into two int8 digits (one accumulator shifted between the HH, mixed and LL products, as in the KNN scan).  Block 6 of
// both halves, (a6 -sat b5) -sat b6, joins them as a6 . (b5 + b6) whenever neither subtraction can saturate (those three blocks
// within +-10922 on both sides; checked as well).  What stays on the VALU, psubsw arithmetic untouched, is the re-squaring of block
// 7's pair sums: 8 coefficient operations per (tile, window) pair instead of 192.
//  * k_mo_pack_win: the window features of a frame in MFMA fragment order: per block of 32 consecutive window positions of a row
//    [12 chunks][64 lanes][16 B] digits (low digits of the 160 plain coefficients + the 16 of b5 + b6, then high), 32 norms, block 7 raw.
//  * k_mo_search_mfma: a workgroup = 4 x 8 tiles (a 32 x 64 pixel region) as the B operand (tile = lane: a tile's running minimum never
//    leaves its lane, its quirk coefficients sit in registers); its four waves share the (window row, block of 32) items of the union of
//    the tiles' windows; per item 20 MFMAs, then per window row of the accumulator the quirk terms of the lanes whose tile can use it.
// Candidates are compared by (error, raster position), which is "first strict minimum in raster order" whatever order they are seen in.
// A frame whose data could saturate raises a flag on the device: this kernel then leaves at once and k_motion_search runs instead.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// (MM_CH, MM_DIG, MM_NORM, MM_QUIRK, MM_BLK_BYTES, MM_LIMIT, MM_LIMIT6: tm_internal.h -- k_window_dcts<true> writes the same layout)
constexpr int MM_TR = 4, MM_TC = 8;                                  // tiles of a workgroup: 4 rows x 8 columns

__device__ __forceinline__ int mm_plain_col(int pb) {  // first coefficient of plain block pb (0..19): blocks 0-4, 7-11 of each half
  const int hf = pb / 10, bi = pb - hf * 10;
  return hf * 96 + (bi < 5 ? bi : bi + 2) * 8;
}
// matrix block mb (8 coefficients; 0..19 plain, 20 / 21 = block 6 of the first / second half, 22 / 23 padding) of a TILE's row: its
// first coefficient, or -1 for padding
__device__ __forceinline__ int mm_tile_col(int mb) { return mb < 20 ? mm_plain_col(mb) : mb == 20 ? 48 : mb == 21 ? 96 + 48 : -1; }

__global__ __launch_bounds__(256) void k_mo_pack_win(const int16_t *__restrict__ win, int ww, int wh, int nbx, const int16_t *__restrict__ cur, int ntiles,
                                                     uint8_t *__restrict__ out, int *__restrict__ flag) {
  __shared__ __attribute__((aligned(16))) int16_t s_raw[32][200];  // the 32 windows' rows as they lie in memory (pitch 400 B)
  bool bad = false;
  // the tile side's range check (its digits are made inside the search kernel)
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < (int64_t)ntiles * 24; i += (int64_t)gridDim.x * 256) {
    const int blk = (int)(i % 24) % 12;
    if (blk == 5) continue;  // never enters on the tile side
    const int lim = blk == 6 ? MM_LIMIT6 : MM_LIMIT;
    const uint4 v = reinterpret_cast<const uint4 *>(cur)[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int lo = (int16_t)(w[k] & 0xffff), hi = (int16_t)(w[k] >> 16);
      bad |= lo > lim || lo < -lim || hi > lim || hi < -lim;
    }
  }
  const int64_t nblocks = (int64_t)wh * nbx;
  for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int wy = (int)(blk / nbx), bx = (int)(blk - (int64_t)wy * nbx);
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 24; i += 256) {
      const int r = i / 24, v = i - r * 24, wx = bx * 32 + r;
      uint4 x = make_uint4(0, 0, 0, 0);
      if (wx < ww) x = *reinterpret_cast<const uint4 *>(win + ((int64_t)wy * ww + wx) * 192 + v * 8);
      *reinterpret_cast<uint4 *>(&s_raw[r][v * 8]) = x;
    }
    __syncthreads();
    uint8_t *obase = out + blk * (int64_t)MM_BLK_BYTES;
    for (int piece = threadIdx.x; piece < 2 * MM_CH * 64; piece += 256) {
      const int kc = piece >> 6, ln = piece & 63, half = ln >> 5, r = ln & 31, c = kc % MM_CH;
      const bool high = kc >= MM_CH;
      uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
      for (int h8 = 0; h8 < 2; h8++) {  // two blocks of eight coefficients, 16 bytes each in LDS
        const int mb = 4 * c + 2 * half + h8;
        uint32_t x[4] = {0, 0, 0, 0};
        int lim = MM_LIMIT;
        if (mb < 20) {
          const uint4 t4 = *reinterpret_cast<const uint4 *>(&s_raw[r][mm_plain_col(mb)]);
          x[0] = t4.x; x[1] = t4.y; x[2] = t4.z; x[3] = t4.w;
        } else if (mb < 22) {  // b5 + b6 of a half: the tile's a6 meets their sum
          const uint4 t5 = *reinterpret_cast<const uint4 *>(&s_raw[r][(mb - 20) * 96 + 40]), t6 = *reinterpret_cast<const uint4 *>(&s_raw[r][(mb - 20) * 96 + 48]);
          const uint32_t a5[4] = {t5.x, t5.y, t5.z, t5.w}, a6[4] = {t6.x, t6.y, t6.z, t6.w};
          lim = 2 * MM_LIMIT6;
#pragma unroll
          for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int e = 0; e < 2; e++) {
              const int v5 = (int16_t)(a5[k] >> (16 * e)), v6 = (int16_t)(a6[k] >> (16 * e));
              bad |= v5 > MM_LIMIT6 || v5 < -MM_LIMIT6 || v6 > MM_LIMIT6 || v6 < -MM_LIMIT6;
              x[k] |= (uint32_t)((v5 + v6) & 0xffff) << (16 * e);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
          for (int e = 0; e < 2; e++) {
            const int v = (int16_t)(x[k] >> (16 * e));
            bad |= v > lim || v < -lim;
            const int lo = ((v + 128) & 255) - 128;
            const int digit = high ? (v - lo) >> 8 : lo;
            const int b = h8 * 8 + k * 2 + e;
            w[b >> 2] |= (uint32_t)(digit & 255) << ((b & 3) * 8);
          }
        }
      }
      *reinterpret_cast<uint4 *>(obase + piece * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    {  // |b|^2 over the 176 matrix coefficients: 8 threads per window, one coefficient of every block each
      const int r = threadIdx.x >> 3, j = threadIdx.x & 7;
      uint32_t sq = 0;
#pragma unroll
      for (int pb = 0; pb < 20; pb++) { const int v = s_raw[r][mm_plain_col(pb) + j]; sq += (uint32_t)(v * v); }
#pragma unroll
      for (int hf = 0; hf < 2; hf++) { const int v = s_raw[r][hf * 96 + 40 + j] + s_raw[r][hf * 96 + 48 + j]; sq += (uint32_t)(v * v); }
      sq += (uint32_t)__shfl_xor((int)sq, 1); sq += (uint32_t)__shfl_xor((int)sq, 2); sq += (uint32_t)__shfl_xor((int)sq, 4);
      if (j == 0) reinterpret_cast<uint32_t *>(obase + MM_NORM)[r] = sq;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 32) {  // block 7 of the first half, raw: its pair sums are re-squared on the VALU
      const int r = threadIdx.x - 64;
      *reinterpret_cast<uint4 *>(obase + MM_QUIRK + r * 16) = *reinterpret_cast<const uint4 *>(&s_raw[r][56]);
    }
  }
  if (bad) atomicOr(flag, 1);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_mo_search_mfma(const int16_t *__restrict__ cur, int tm_w, int tm_h, int ngroups, const uint8_t *__restrict__ packed, int nbx,
                                                        int r, const int *__restrict__ flag, uint32_t *__restrict__ best_err,
                                                        int8_t *__restrict__ out_px, int8_t *__restrict__ out_py) {
  if (*flag) return;  // a coefficient beyond +-16383 somewhere in this frame: k_motion_search takes it
  __shared__ __attribute__((aligned(16))) uint32_t s_q[4][32][4];  // per wave: block 7 (first half) of the item's 32 windows
  __shared__ uint32_t s_err[8][32];
  __shared__ int s_pos[8][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, t = lane & 31;  // (wave: in a scalar register, and with it the item loop's arithmetic)
  const int gw = (tm_w + MM_TC - 1) / MM_TC;
  // workgroups are dealt to the XCDs round robin: workgroup b works on group (b % 8) * ceil(n / 8) + b / 8, so that the groups running
  // on one XCD are neighbours in the picture and find each other's window blocks in that XCD's L2 (placement is a matter of speed only)
  const int per_xcd = gridDim.x >> 3;  // (the grid is 8 x ceil(groups / 8) workgroups; the ones past the last group have nothing to do)
  const int gsel = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (gsel >= ngroups) return;
  const int gy = gsel / gw, gx = gsel - gy * gw;
  const int sw = tm_w * 8, sh = tm_h * 8, ww = sw - 7;
  const int sy = gy * MM_TR + (t >> 3), sx = gx * MM_TC + (t & 7);
  const bool tvalid = sy < tm_h && sx < tm_w;
  const int dy = sy * 8, dx = sx * 8;
  const int16_t *arow = cur + ((int64_t)(tvalid ? sy : 0) * tm_w + (tvalid ? sx : 0)) * 192;
  // B operand: the tile's digits in fragment order (lane = half * 32 + tile holds positions half * 16 .. + 15 of every chunk)
  v4i Bf[2 * MM_CH];
  uint32_t na = 0;
#pragma unroll
  for (int c = 0; c < MM_CH; c++) {
    uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int h8 = 0; h8 < 2; h8++) {
      const int tcol = mm_tile_col(4 * c + 2 * half + h8);
      const uint4 x = tcol >= 0 ? *reinterpret_cast<const uint4 *>(arow + tcol) : make_uint4(0, 0, 0, 0);
      const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int v = tvalid ? (int)(int16_t)(w[k] >> (16 * e)) : 0;
          const int l8 = ((v + 128) & 255) - 128, h8v = (v - l8) >> 8;
          const int b = h8 * 8 + k * 2 + e;
          lo[b >> 2] |= (uint32_t)(l8 & 255) << ((b & 3) * 8);
          hi[b >> 2] |= (uint32_t)(h8v & 255) << ((b & 3) * 8);
          na += (uint32_t)(v * v);
        }
      }
    }
    Bf[c] = v4i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3]};
    Bf[MM_CH + c] = v4i{(int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  }
  na += (uint32_t)__shfl_xor((int)na, 32);  // both halves of the tile's 176 coefficients
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  const uint4 a7h1 = tvalid ? *reinterpret_cast<const uint4 *>(arow + 56) : zero4;
  // the tile's window (1218-1221) as start + unsigned extent; an absent tile gets an empty one
  const int y0 = max(0, dy - r - 1), y1 = min(sh - 8, dy + r), x0 = max(0, dx - r - 1), x1 = min(sw - 8, dx + r);
  const unsigned wy0 = (unsigned)y0, wyr = tvalid ? (unsigned)(y1 - y0) : 0u, wx0 = tvalid ? (unsigned)x0 : 0x40000000u, wxr = (unsigned)(x1 - x0);
  // union of the group's windows, in (row, block of 32) items
  const int ly = min(gy * MM_TR + MM_TR - 1, tm_h - 1) * 8, lx = min(gx * MM_TC + MM_TC - 1, tm_w - 1) * 8;
  const int uy0 = max(0, gy * MM_TR * 8 - r - 1), uy1 = min(sh - 8, ly + r);
  const int ux0 = max(0, gx * MM_TC * 8 - r - 1), ux1 = min(sw - 8, lx + r);
  const int bx0 = ux0 >> 5, nbu = (ux1 >> 5) - bx0 + 1, nitems = (uy1 - uy0 + 1) * nbu;
  uint32_t best = 0xffffffffu;
  int bposa = 0x7ffffff0;  // the best candidate's raster position, less 4 * half
  const unsigned wx0h = wx0 - 4u * (unsigned)half;  // (window of row q: wxa + 4 * half)
  const int dxh = dx + 4 - 4 * half;
  const uint32_t *sq_lane = &s_q[wave][4 * half][0];
  for (int it = wave; it < nitems; it += 4) {
    const int row = it / nbu, wy = uy0 + row, bx = bx0 + (it - row * nbu);
    const uint8_t *base = packed + ((int64_t)wy * nbx + bx) * MM_BLK_BYTES;
    const uint4 quirk = *reinterpret_cast<const uint4 *>(base + MM_QUIRK + (lane & 31) * 16);  // (first: its way through LDS then does not wait for the operands)
    v4i Af[2 * MM_CH];
#pragma unroll
    for (int kc = 0; kc < 2 * MM_CH; kc++) Af[kc] = *reinterpret_cast<const v4i *>(base + (kc * 64 + lane) * 16);
    uint32_t nbr[16];  // |b|^2 of accumulator row q: window (q & 3) + 8 * (q >> 2) + 4 * half of the block
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) {
      const v4i x = *reinterpret_cast<const v4i *>(base + MM_NORM + (q4 * 8 + half * 4) * 4);
      nbr[q4 * 4] = (uint32_t)x[0]; nbr[q4 * 4 + 1] = (uint32_t)x[1]; nbr[q4 * 4 + 2] = (uint32_t)x[2]; nbr[q4 * 4 + 3] = (uint32_t)x[3];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of the previous item's block 7 are over
    if (lane < 32) reinterpret_cast<uint4 *>(&s_q[wave][0][0])[lane] = quirk;
    v16i acc;
#pragma unroll
    for (int q = 0; q < 16; q++) acc[q] = 0;
#pragma unroll
    for (int c = 0; c < MM_CH; c++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Af[MM_CH + c], Bf[MM_CH + c], acc, 0, 0, 0);  // b_H . a_H
#pragma unroll
    for (int q = 0; q < 16; q++) acc[q] = (int)((unsigned)acc[q] << 8);
#pragma unroll
    for (int c = 0; c < MM_CH; c++) {
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Af[c], Bf[MM_CH + c], acc, 0, 0, 0);  // b_L . a_H
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Af[MM_CH + c], Bf[c], acc, 0, 0, 0);  // b_H . a_L
    }
#pragma unroll
    for (int q = 0; q < 16; q++) acc[q] = (int)((unsigned)acc[q] << 8);
#pragma unroll
    for (int c = 0; c < MM_CH; c++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Af[c], Bf[c], acc, 0, 0, 0);  // b_L . a_L
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // block 7 of the item's windows is in LDS (written by this wave's own lanes)
    // The sixteen accumulator rows: row q is window m0 = (q & 3) + 8 (q >> 2) of the block for the lanes of the first half, m0 + 4 for the second.
    // Everything about the window that does not depend on the lane is scalar (wave, item, q are); a lane sees its candidates in raster order
    // (items go up the rows and along them, m0 goes up with q), so a tie never replaces what it holds: err < best alone decides.
    const bool rowin = (unsigned)wy - wy0 <= wyr;
    const int posrow = wy * ww + bx * 32;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int m0 = (q & 3) + 8 * (q >> 2), wxa = bx * 32 + m0;
      if (wxa + 4 < ux0 || wxa > ux1) continue;  // neither window of the row is in reach of a tile of the group
      const bool in = rowin && (unsigned)wxa - wx0h <= wxr;
      const uint4 b7h1 = *reinterpret_cast<const uint4 *>(&sq_lane[m0 * 4]);
      uint32_t e = na + nbr[q] - 2u * (uint32_t)acc[q];  // the 20 plain blocks and both block 6 terms
      const uint32_t p0 = sq2(sat_sub2(a7h1.x, b7h1.x)), p1 = sq2(sat_sub2(a7h1.y, b7h1.y)), p2 = sq2(sat_sub2(a7h1.z, b7h1.z)),
                     p3 = sq2(sat_sub2(a7h1.w, b7h1.w));
      e = sq2acc(p3, sq2acc(p2, sq2acc(p1, sq2acc(p0, e))));  // the pair sums re-squared as int16 pairs (their plain sum is in the matrix part)
      uint32_t err;  // + manhattan penalty (1236)
      asm("v_sad_u32 %0, %1, %2, %3" : "=v"(err) : "s"(wxa + 4), "v"(dxh), "v"(e));  // |wx - dx|, both sides + 4 - 4 * half (unsigned operands)
      asm("v_sad_u32 %0, %1, %2, %3" : "=v"(err) : "s"(wy), "v"(dy), "v"(err));
      const bool take = in && err < best;
      best = take ? err : best;
      bposa = take ? posrow + m0 : bposa;
    }
  }
  const int bpos = bposa + 4 * half;
  s_err[wave * 2 + half][t] = best;
  s_pos[wave * 2 + half][t] = bpos;
  __syncthreads();
  if (tid < 32) {
    const int ty = gy * MM_TR + (tid >> 3), tx = gx * MM_TC + (tid & 7);
    if (ty < tm_h && tx < tm_w) {
      uint32_t be = 0xffffffffu;
      int bp = 0x7fffffff;
      for (int g = 0; g < 8; g++) {
        const uint32_t e = s_err[g][tid];
        const int p = s_pos[g][tid];
        if (e < be || (e == be && p < bp)) { be = e; bp = p; }
      }
      const int64_t i = (int64_t)ty * tm_w + tx;
      best_err[i] = be;
      const int oy = bp / ww, ox = bp - oy * ww;
      out_px[i] = (int8_t)(ox - tx * 8);
      out_py[i] = (int8_t)(oy - ty * 8);
    }
  }
}
// front buffer of PredictMotion (1255-1260): the frame's tiles, un-mirrored, laid out as a tm_w*8 x tm_h*8 image
__global__ void k_tiles_to_screen(const uint32_t *__restrict__ tiles, const uint8_t *__restrict__ flags, int tm_w, int tm_h,
                                  uint32_t *__restrict__ screen) {
  const int64_t n = (int64_t)tm_w * tm_h * 64;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = e >> 6;
    const int p = (int)(e & 63), y = p >> 3, x = p & 7, f = flags[t];
    const int src = (((f & 2) ? 7 - y : y) << 3) | ((f & 1) ? 7 - x : x);
    const int sy = (int)(t / tm_w), sx = (int)(t - (int64_t)sy * tm_w);
    screen[((int64_t)sy * 8 + y) * (tm_w * 8) + sx * 8 + x] = tiles[t * 64 + src];
  }
}

// Reconstruct.DoXY after both searches (1534-1654), one wave per tile-map item, lane = pixel
__global__ __launch_bounds__(64) void k_recon_decide(int tm_w, int per, int pal_from_map /* EPU: the item's own PalIdx, not the tile's */,
                                                     const uint32_t *__restrict__ mp_err /* null: key frame start */,
                                                     const uint8_t *__restrict__ fflags, const int32_t *__restrict__ gpal_idx,
                                                     const uint8_t *__restrict__ gpal_px, const int32_t *__restrict__ palettes, int pal_size,
                                                     const uint32_t *__restrict__ back, uint32_t *__restrict__ front,
                                                     int32_t *__restrict__ tm_tile, int32_t *__restrict__ tm_pal, uint32_t *__restrict__ tm_err,
                                                     const int8_t *__restrict__ px, const int8_t *__restrict__ py, uint8_t *__restrict__ pred) {
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= per) return;
  const uint32_t mp = mp_err ? mp_err[i] : 0xffffffffu;
  const bool perfect = mp <= 192u;  // IsZero(mpErr, cTileDCTSize): motion prediction has priority when perfect
  int32_t tile = tm_tile[i];
  const uint32_t knn = (perfect || tile < 0) ? 0xffffffffu : tm_err[i];
  if (perfect) tile = -1;
  const int32_t pal = tile >= 0 ? (pal_from_map ? tm_pal[i] : gpal_idx[tile]) : -1;
  const int64_t diff = (int64_t)knn - (int64_t)mp;
  const bool knn_best = (diff > 192 || diff < -192) && knn < mp;  // CompareValue(knnErr, mpErr, cTileDCTSize) = LessThanValue
  const int sy = i / tm_w, sx = i - sy * tm_w, sw = tm_w * 8;
  const int ty = lane >> 3, tx = lane & 7;
  uint32_t col;
  if (knn_best) {
    const int f = fflags[i];
    const int tym = (f & 2) ? 7 - ty : ty, txm = (f & 1) ? 7 - tx : tx;  // TMI^.VMirror / HMirror, 1627-1633
    col = (uint32_t)palettes[(int64_t)pal * pal_size + gpal_px[(int64_t)tile * 64 + tym * 8 + txm]];
  } else {
    col = back[((int64_t)sy * 8 + py[i] + ty) * sw + sx * 8 + px[i] + tx];  // 1650
  }
  front[((int64_t)sy * 8 + ty) * sw + sx * 8 + tx] = col;
  if (lane == 0) {
    tm_tile[i] = tile;
    tm_pal[i] = pal;
    tm_err[i] = knn_best ? knn : mp;
    pred[i] = knn_best ? 0 : 1;
  }
}

// ---- Reduce with motion prediction: per group (distinct tile content) the largest prediction error among its members,
// separately for members on a key frame's first frame (their PSNR is divided by 10, 4028-4029) and the others
__global__ void k_group_max_err(const int32_t *__restrict__ group, const uint32_t *__restrict__ pm_err, const uint8_t *__restrict__ frame_is_kf,
                                int per, int64_t q, uint32_t *__restrict__ gmax /* [2][ngroups] as (err + 1), 0 = no member */, int64_t ng) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < q; i += (int64_t)gridDim.x * blockDim.x) {
    const int kf = frame_is_kf[i / per];
    const uint32_t e = pm_err[i];
    atomicMax(&gmax[(int64_t)kf * ng + group[i]], e == 0xffffffffu ? e : e + 1);
  }
}
// STCGREval's count for one threshold: groups with a member that is not predicted (err > the largest error still predicted)
__global__ void k_count_groups(const uint32_t *__restrict__ gmax, int64_t ng, uint32_t emax_plain, uint32_t emax_kf, int has_plain,
                               int has_kf, unsigned long long *__restrict__ count) {
  unsigned long long c = 0;
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < ng; g += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t a = gmax[g], b = gmax[ng + g];  // stored +1
    const bool un_plain = a != 0 && (!has_plain || a - 1 > emax_plain);
    const bool un_kf = b != 0 && (!has_kf || b - 1 > emax_kf);
    c += (un_plain || un_kf) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}
__global__ void k_mark_predicted(const uint32_t *__restrict__ pm_err, const uint8_t *__restrict__ frame_is_kf, int per, int64_t q,
                                 uint32_t emax_plain, uint32_t emax_kf, int has_plain, int has_kf, uint8_t *__restrict__ pred,
                                 int32_t *__restrict__ keep /* 1 where NOT predicted */) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < q; i += (int64_t)gridDim.x * blockDim.x) {
    const bool kf = frame_is_kf[i / per] != 0;
    const uint32_t e = pm_err[i];
    const bool p = kf ? (has_kf && e <= emax_kf) : (has_plain && e <= emax_plain);
    pred[i] = p ? 1 : 0;
    keep[i] = p ? 0 : 1;
  }
}

static inline int gridn(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 16)); }

}  // namespace

int launch_motion_search(const void *cur, int tm_w, int tm_h, const void *win, int radius, void *best_err, void *px, void *py,
                         hipStream_t stream) {
  TM_CHECK(tm_w > 0 && tm_h > 0 && radius >= 1 && radius <= 128, TM_E_INVAL, "motion search: bad arguments");
  const int bw = (tm_w + MS_TB - 1) / MS_TB, bh = (tm_h + MS_TB - 1) / MS_TB;
  if (knobs().motion_valu) {  // the VALU kernel alone (A/B runs)
    hipLaunchKernelGGL(k_motion_search, dim3(bw * bh), dim3(256), 0, stream, (const int16_t *)cur, tm_w, tm_h, (const int16_t *)win, radius - 1,
                       (uint32_t *)best_err, (int8_t *)px, (int8_t *)py, (const int *)nullptr);
    TM_HIP(hipGetLastError());
    return TM_OK;
  }
  // matrix-core path: pack the windows, search; a frame whose coefficients could saturate a plain difference falls to the VALU kernel
  // (decided on the device: both kernels are queued, one of them leaves at once)
  const int ww = tm_w * 8 - 7, wh = tm_h * 8 - 7, nbx = (ww + 31) / 32;
  DevBuf packed, flag;  // (released at return: the pool hands memory back out in stream order, and everything here is on `stream`)
  TM_TRY(packed.alloc((size_t)wh * nbx * MM_BLK_BYTES));
  TM_TRY(flag.alloc(sizeof(int)));
  TM_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), stream));
  hipLaunchKernelGGL(k_mo_pack_win, dim3(std::min(wh * nbx, 8192)), dim3(256), 0, stream, (const int16_t *)win, ww, wh, nbx, (const int16_t *)cur, tm_w * tm_h,
                     packed.as<uint8_t>(), flag.as<int>());
  const int gw = (tm_w + MM_TC - 1) / MM_TC, gh = (tm_h + MM_TR - 1) / MM_TR;
  hipLaunchKernelGGL(k_mo_search_mfma, dim3(8 * ((gw * gh + 7) / 8)), dim3(256), 0, stream, (const int16_t *)cur, tm_w, tm_h, gw * gh, packed.as<uint8_t>(), nbx, radius - 1, flag.as<int>(),
                     (uint32_t *)best_err, (int8_t *)px, (int8_t *)py);
  hipLaunchKernelGGL(k_motion_search, dim3(bw * bh), dim3(256), 0, stream, (const int16_t *)cur, tm_w, tm_h, (const int16_t *)win, radius - 1,
                     (uint32_t *)best_err, (int8_t *)px, (int8_t *)py, flag.as<int>());
  TM_HIP(hipGetLastError());
  return TM_OK;
}

int launch_motion_search_fb(const void *cur, int tm_w, int tm_h, const void *fb, void *win, int radius, void *best_err, void *px, void *py,
                            hipStream_t stream) {
  TM_CHECK(tm_w > 0 && tm_h > 0 && radius >= 1 && radius <= 128, TM_E_INVAL, "motion search: bad arguments");
  const int sw = tm_w * 8, sh = tm_h * 8;
  if (knobs().motion_valu || knobs().motion_pack_separate || knobs().window_dcts_by_tile) {
    TM_TRY(launch_window_dcts(fb, sw, sh, win, stream));
    return launch_motion_search(cur, tm_w, tm_h, win, radius, best_err, px, py, stream);
  }
  const int bw = (tm_w + MS_TB - 1) / MS_TB, bh = (tm_h + MS_TB - 1) / MS_TB;
  const int ww = sw - 7, wh = sh - 7, nbx = (ww + 31) / 32;
  DevBuf packed, flag;  // (released at return: the pool hands memory back out in stream order, and everything here is on `stream`)
  TM_TRY(packed.alloc((size_t)wh * nbx * MM_BLK_BYTES));
  TM_TRY(flag.alloc(sizeof(int)));
  if (knobs().motion_force_flag) { const int one = 1; TM_HIP(hipMemcpyAsync(flag.p, &one, sizeof(int), hipMemcpyHostToDevice, stream)); TM_HIP(hipStreamSynchronize(stream)); }
  else TM_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), stream));
  TM_TRY(launch_window_dcts_packed(fb, sw, sh, cur, tm_w * tm_h, packed.p, flag.as<int>(), stream));
  const int gw = (tm_w + MM_TC - 1) / MM_TC, gh = (tm_h + MM_TR - 1) / MM_TR;
  hipLaunchKernelGGL(k_mo_search_mfma, dim3(8 * ((gw * gh + 7) / 8)), dim3(256), 0, stream, (const int16_t *)cur, tm_w, tm_h, gw * gh, packed.as<uint8_t>(), nbx, radius - 1, flag.as<int>(),
                     (uint32_t *)best_err, (int8_t *)px, (int8_t *)py);
  // a frame beyond the matrix form's range (decided on the device): the int16 rows after all, and the VALU search -- both leave at once otherwise
  TM_TRY(launch_window_dcts(fb, sw, sh, win, stream, flag.as<int>()));
  hipLaunchKernelGGL(k_motion_search, dim3(bw * bh), dim3(256), 0, stream, (const int16_t *)cur, tm_w, tm_h, (const int16_t *)win, radius - 1,
                     (uint32_t *)best_err, (int8_t *)px, (int8_t *)py, flag.as<int>());
  TM_HIP(hipGetLastError());
  return TM_OK;
}

int launch_tiles_to_screen(const void *tiles, const void *flags, int tm_w, int tm_h, void *screen, hipStream_t stream) {
  hipLaunchKernelGGL(k_tiles_to_screen, dim3(gridn((int64_t)tm_w * tm_h * 64)), dim3(256), 0, stream, (const uint32_t *)tiles,
                     (const uint8_t *)flags, tm_w, tm_h, (uint32_t *)screen);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

int launch_recon_decide(int tm_w, int per, int pal_from_map, const void *mp_err, const void *fflags, const void *gpal_idx, const void *gpal_px,
                        const void *palettes, int pal_size, const void *back, void *front, void *tm_tile, void *tm_pal, void *tm_err,
                        const void *px, const void *py, void *pred, hipStream_t stream) {
  hipLaunchKernelGGL(k_recon_decide, dim3(per), dim3(64), 0, stream, tm_w, per, pal_from_map, (const uint32_t *)mp_err, (const uint8_t *)fflags,
                     (const int32_t *)gpal_idx, (const uint8_t *)gpal_px, (const int32_t *)palettes, pal_size, (const uint32_t *)back,
                     (uint32_t *)front, (int32_t *)tm_tile, (int32_t *)tm_pal, (uint32_t *)tm_err, (const int8_t *)px, (const int8_t *)py,
                     (uint8_t *)pred);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

// EuclideanToPSNR, utils.pas:1074-1078 (host: log10 must be the host's)
float euclidean_to_psnr(uint32_t e) {
  const float r = (float)((double)e * (1.0 / 192));
  const float m = r > 0.5f ? r : 0.5f;
  return (float)(10 * std::log10(255 * 255 / (double)m));
}

// largest error whose PSNR (divided by `div`) still exceeds x; false if none does.  PSNR is non-increasing in the error.
static bool largest_predicted_err(double x, double div, uint32_t *emax) {
  auto pred = [&](uint32_t e) { return (double)euclidean_to_psnr(e) / div > x; };
  if (!pred(0)) return false;
  uint32_t lo = 0, hi = 0xffffffffu;  // pred(lo) true
  if (pred(hi)) { *emax = hi; return true; }
  while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (pred(mid)) lo = mid; else hi = mid; }
  *emax = lo;
  return true;
}

int solve_tile_count(const void *group, int64_t ngroups, const void *pm_err, const void *frame_is_kf, int per, int64_t q, double target,
                     void *pred, void *keep, double *x_out, int *probes_out, hipStream_t stream) {
  // SolveTileCount (4043-4046) = GoldenRatioSearch (utils.pas:1044-1072) over STCGREval (4014-4041)
  DevBuf gmax, count;
  TM_TRY(gmax.alloc((size_t)ngroups * 8));
  TM_TRY(count.alloc(8));
  TM_HIP(hipMemsetAsync(gmax.p, 0, (size_t)ngroups * 8, stream));
  hipLaunchKernelGGL(k_group_max_err, dim3(gridn(q)), dim3(256), 0, stream, (const int32_t *)group, (const uint32_t *)pm_err,
                     (const uint8_t *)frame_is_kf, per, q, gmax.as<uint32_t>(), ngroups);
  TM_HIP(hipGetLastError());
  const double phi = (1.0 + std::sqrt(5.0)) / 2.0, inv_phi = 1.0 / phi;
  double mn = 0.0, mx = 10.0 * std::log(255.0 * 255.0 / 0.5) / std::log(10.0), last = 0.0;
  uint32_t ep = 0, ek = 0;
  bool hp = false, hk = false;
  int n = 0;
  for (;;) {
    if (std::fabs(mn - mx) <= 1e-6) break;
    const double x = mn < mx ? mn + (mx - mn) * (1.0 - inv_phi) : mn + (mx - mn) * inv_phi;
    hp = largest_predicted_err(x, 1.0, &ep);
    hk = largest_predicted_err(x, 10.0, &ek);
    TM_HIP(hipMemsetAsync(count.p, 0, 8, stream));
    hipLaunchKernelGGL(k_count_groups, dim3(gridn(ngroups)), dim3(256), 0, stream, gmax.as<uint32_t>(), ngroups, ep, ek, hp ? 1 : 0, hk ? 1 : 0,
                       count.as<unsigned long long>());
    unsigned long long c = 0;
    {
      HostRead hr_(stream);
      TM_TRY(hr_.get(&c, count.p, 8));
      TM_TRY(hr_.wait());
    }
    const double y = (double)c;
    last = x; n++;
    if (std::fabs(y - target) <= 0.5) break;
    if (y < target) mn = x; else mx = x;
  }
  TM_CHECK(n > 0, TM_E_INVAL, "tile-count search made no probe");
  hipLaunchKernelGGL(k_mark_predicted, dim3(gridn(q)), dim3(256), 0, stream, (const uint32_t *)pm_err, (const uint8_t *)frame_is_kf, per, q, ep, ek,
                     hp ? 1 : 0, hk ? 1 : 0, (uint8_t *)pred, (int32_t *)keep);
  TM_HIP(hipGetLastError());
  if (x_out) *x_out = last;
  if (probes_out) *probes_out = n;
  return TM_OK;
}

}  // namespace tmx
