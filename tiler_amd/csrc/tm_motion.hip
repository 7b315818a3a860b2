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

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {
namespace {

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t sat_sub2(uint32_t a, uint32_t b) {  // psubsw on one dword (two int16)
  const s16x2 r = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b));
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t sq2(uint32_t d) {  // pmaddwd of a dword with itself, mod 2^32: v_dot2c_i32_i16
  return (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, d), __builtin_bit_cast(s16x2, d), 0, false);
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
                                                       int8_t *__restrict__ out_px, int8_t *__restrict__ out_py) {
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
  hipLaunchKernelGGL(k_motion_search, dim3(bw * bh), dim3(256), 0, stream, (const int16_t *)cur, tm_w, tm_h, (const int16_t *)win, radius - 1,
                     (uint32_t *)best_err, (int8_t *)px, (int8_t *)py);
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
    TM_HIP(hipMemcpyAsync(&c, count.p, 8, hipMemcpyDeviceToHost, stream));
    TM_HIP(hipStreamSynchronize(stream));
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
