// tm_features.hip -- load-side (A1-A3) and feature-extraction (A4-A6) kernels for gfx950.
//
// Bit-exactness rules (DESIGN.md "Arithmetic"): every floating-point step is a single IEEE +,-,*,/ issued through
// the __f*_rn / __d*_rn intrinsics (never contracted into FMA), in the order the reference's code performs it
// (DCTInner_asm, utils.pas:874-1035; RGBToYUV 478-490; RGBToLAB 374-410).  Transcendentals are replaced by host
// built tables (sRGB gamma, cosines) or a +,-,*,/-only Newton cube root, so CPU and GPU agree to the bit.
#include "tm_common.h"
#include "tm_device.h"
#include "tm_internal.h"

namespace tmx {

// ---------------------------------------------------------------------------------------------------------------
// A1+A2+A3: one wave per tile, one lane per pixel; a wave works through LT_BATCH tiles, then 3 * LT_BATCH of its lanes add up the Lab
// planes -- each sum is 64 Single additions in raster order (1349-1362), a dependent chain that would hold a whole wave per tile.
// LoadFromImage (tilingencoder.pas:1293-1320) -> PrepareInterFrameData (1329-1367) -> mirror heuristics
// (4865-4878) + H/V flip (1393-1411).
#ifndef TM_LT_BATCH
#define TM_LT_BATCH 8  // tiles a wave converts before 3 x that many of its lanes add up the Lab planes (LDS: 780 bytes per tile and wave; measured: 4-6 1.80-1.85 ms, 8 1.78, 12 1.93, 16 2.19, 20 2.65 -- the waves a CU holds matter more than the lanes the sums use)
#endif
constexpr int LT_BATCH = TM_LT_BATCH, LT_TILE = 195, LT_PLANE = 65;  // strides in floats: lane (tile, plane) of the summing phase reads bank 3 * tile + plane
__global__ __launch_bounds__(256) void k_load_tiles(const uint32_t *__restrict__ frames, int nframes, int img_w, int img_h,
                                                    int tm_w, int tm_h, const float *__restrict__ srgb_lut,
                                                    uint32_t *__restrict__ tiles, uint8_t *__restrict__ flags,
                                                    float *__restrict__ lab_means) {
  __shared__ float s_lab[4][LT_BATCH * LT_TILE];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform: the tile arithmetic below stays on the scalar unit)
  const int64_t tiles_per_frame = (int64_t)tm_w * tm_h;
  const int64_t total = tiles_per_frame * nframes;
  const int64_t nbatch = (total + 4 * LT_BATCH - 1) / (4 * LT_BATCH);
  float *lab = s_lab[wave];  // a wave only touches its own part: no workgroup barrier
  const int py = lane >> 3, pxl = lane & 7;
  const unsigned lane_off = (unsigned)(py * img_w + pxl);
  for (int64_t it = blockIdx.x; it < nbatch; it += gridDim.x) {
    const int64_t base = (it * 4 + wave) * LT_BATCH;
    // frame, tile row and tile column of the batch's first tile by division, of the others by stepping
    int64_t f = base / tiles_per_frame;
    int sy, sx;
    {
      const int ti = (int)(base - f * tiles_per_frame);
      sy = ti / tm_w;
      sx = ti - sy * tm_w;
    }
    for (int j = 0; j < LT_BATCH; j++, sx++) {
      const int64_t t = base + j;
      if (t >= total) break;
      if (sx == tm_w) { sx = 0; if (++sy == tm_h) { sy = 0; f++; } }
      // the tile's first pixel on the scalar unit, the lane's pixel as a 32-bit offset from it (a 64-bit index per lane was a dozen vector instructions)
      const uint32_t *fp = frames + ((f * img_h + (int64_t)sy * 8) * (int64_t)img_w + (int64_t)sx * 8);
      uint32_t px = 0;
      if (sy * 8 + py < img_h && sx * 8 + pxl < img_w) px = swap_rb(fp[lane_off]);
      float l, a, b;
      rgb_to_lab_det(px & 0xff, (px >> 8) & 0xff, (px >> 16) & 0xff, srgb_lut, l, a, b);
      float *lt = lab + j * LT_TILE + lane;
      lt[0] = l;
      lt[LT_PLANE] = a;
      lt[2 * LT_PLANE] = b;
      // quadrant luma sums (GetTileZoneSum, 4842-4863): integer, order free.  Sums of 4 pixels by quad permutes; lanes 4g..4g+3 then hold
      // group g = (pixel row g / 2, half g & 1).  A row of 16 lanes holds groups 4r..4r+3: row_shr:8 leaves (4r) + (4r+2) in its lanes 8..11
      // and (4r+1) + (4r+3) in lanes 12..15 -- the left and right halves of pixel rows 2r and 2r+1 --, and the scalar unit adds two rows each
      int luma = (int)(px & 0xff) * 299 + (int)((px >> 8) & 0xff) * 587 + (int)((px >> 16) & 0xff) * 114;
      luma += __builtin_amdgcn_update_dpp(0, luma, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
      luma += __builtin_amdgcn_update_dpp(0, luma, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
      luma += __builtin_amdgcn_update_dpp(0, luma, 0x118, 0xf, 0xf, true);   // row_shr:8 (lanes 0..7 of a row add nothing)
      const int q00 = __builtin_amdgcn_readlane(luma, 8) + __builtin_amdgcn_readlane(luma, 24);    // [bottom][right]
      const int q01 = __builtin_amdgcn_readlane(luma, 12) + __builtin_amdgcn_readlane(luma, 28);
      const int q10 = __builtin_amdgcn_readlane(luma, 40) + __builtin_amdgcn_readlane(luma, 56);
      const int q11 = __builtin_amdgcn_readlane(luma, 44) + __builtin_amdgcn_readlane(luma, 60);
      const bool hm = q00 + q10 < q01 + q11;
      const bool vm = q00 + q01 < q10 + q11;
      const int src = ((vm ? 7 - py : py) << 3) | (hm ? 7 - pxl : pxl);
      const uint32_t canon = __shfl(px, src);
      (tiles + t * 64)[lane] = canon;
      if (lane == 0) flags[t] = (uint8_t)((hm ? 1 : 0) | (vm ? 2 : 0));
    }
    // Result[di+c] += lab, 64 Singles in raster order, then *= 1/64 (1349-1362): one lane per (tile, plane)
    if (lane < 3 * LT_BATCH) {
      const int j = lane / 3, c = lane - j * 3;
      if (base + j < total) {
        const float *p = lab + j * LT_TILE + c * LT_PLANE;
        float s = 0.0f;
        for (int k = 0; k < 64; k++) s = __fadd_rn(s, p[k]);
        lab_means[(base + j) * 3 + c] = __fmul_rn(s, 1.0f / 64);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// A4+A5: int16[192] features.  One wave per tile; lane = DCT position (v,u) and holds its 64-entry LUT row in
// registers; the 3x64 component planes live in LDS and are read as wave-wide broadcasts.
// SRC: 0 = RGB tiles, 1 = palette-index tiles through their palette, 2 = every 8x8 window of a frame buffer
// (PredictMotion.DoDCTs / Reconstruct.DoDCTs, tilingencoder.pas:1157-1182, 1437-1462: `tiles` is the buffer, pal_size its
// width in pixels, tile t = window at (t mod (width-7), t div (width-7))), 3 = every palette-index tile under every palette
// (row t = tile t div P under palette t mod P, P passed in use_lab: the vectors 1590-1591 recompute per query).
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The first look's verdict on one coefficient, in Single arithmetic (double-precision instructions issue at a quarter of the Singles' rate
// here, and the verdict was a third of a coefficient's): t = z w from the separable transform (double), sa >= sum |pixel x LUT entry|, wabs = |w|.
// The reference's z_ref w lies within slack = sa (2^-23 + 2^-24)(1 + 2^-17)(1 + 2^-20) |w| + 1e-9 (|t| + 1) of t (k_features_i16 has the
// derivation), and tf = Single(t) within |t| 2^-24 of t: when tf is farther than the two together from every half-integer, Round(z_ref w) =
// rint(tf).  Returns false (in doubt: the caller sums the coefficient in the reference's order) otherwise, and for values no int16 path needs.
// (sw = sa x 1.82e-7 |w|: the caller folds what is constant per coefficient.  slack < 0.25 also bounds |t| by 4 x 10^6: rintf is exact there.)
__device__ __forceinline__ bool first_look_rounds(double t, float sw, int &o) {
  const float tf = (float)t;
  const float slack = fmaf(fabsf(tf), 6.21e-8f, sw + 1.1e-9f);
  const float fr = __builtin_amdgcn_fractf(tf);  // tf - floor(tf), exact
  o = (int)rintf(tf);
  return fabsf(fr - 0.5f) > slack && slack < 0.25f;
}

template <int SRC>
__global__ __launch_bounds__(256) void k_features_i16(const uint32_t *__restrict__ tiles, const uint8_t *__restrict__ pal_px,
                                                      const int32_t *__restrict__ pal_idx, const int32_t *__restrict__ palettes,
                                                      int pal_size, const uint8_t *__restrict__ mirror_flags, int64_t n,
                                                      int weighted, int use_lab, const float *__restrict__ lut,
                                                      const double *__restrict__ weights, const uint8_t *__restrict__ snake,
                                                      const float *__restrict__ srgb_lut, int16_t *__restrict__ out,
                                                      int *__restrict__ colmm /* null, or [384]: running min / max of the 192 output columns (atomics) */,
                                                      const double *__restrict__ cosd /* [8][8] the LUT's cosine factor, cos((x + 0.5) u pi / div) */, int plain) {
  __shared__ __attribute__((aligned(16))) double s_cpn[2][4][192];  // the tile's planes, widened once where they are made
  __shared__ __attribute__((aligned(16))) float s_lut[4096];    // FDCTLut as the reference holds it (Singles): the in-order sums read it
  __shared__ __attribute__((aligned(16))) double s_row[4][64];  // a wave's row transforms, [u][y]
  int mmn[3] = {INT_MAX, INT_MAX, INT_MAX}, mmx[3] = {INT_MIN, INT_MIN, INT_MIN};  // of this lane's three output columns
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < 1024; e += 256) reinterpret_cast<float4 *>(s_lut)[e] = reinterpret_cast<const float4 *>(lut)[e];
  // lane = v * 8 + u is coefficient (u, v); in the row pass the same lane stands for (u, row y = lane >> 3)
  double au[8], av[8];
#pragma unroll
  for (int x = 0; x < 8; x++) { au[x] = cosd[(lane & 7) * 8 + x]; av[x] = cosd[(lane >> 3) * 8 + x]; }
  // cDCTUVRatio (utils.pas:100-109): 0.5, Single(sqrt(0.5)), 1
  const double ruv = (lane == 0) ? 0.5 : (((lane & 7) == 0 || (lane >> 3) == 0) ? 0.707106769084930419921875 : 1.0);
  double w[3];
#pragma unroll
  for (int c = 0; c < 3; c++) w[c] = weights[c * 64 + lane];
  const int zz = snake[lane];
  const int64_t niter = (n + 3) / 4;
  int buf = 0;
  // a tile's pixel of this lane: the colour the reference's ConvertToCpnPixels would read
  auto fetch = [&](int64_t t) -> uint32_t {
    const int f = mirror_flags ? mirror_flags[t] : 0;
    const int y = lane >> 3, x = lane & 7;
    const int src = (((f & 2) ? 7 - y : y) << 3) | ((f & 1) ? 7 - x : x);  // ConvertToCpnPixels 3080-3085 / 3093-3098
    if (SRC == 1) return (uint32_t)palettes[(int64_t)pal_idx[t] * pal_size + pal_px[t * 64 + src]];
    if (SRC == 3) {
      const int64_t tile = t / use_lab;
      return (uint32_t)palettes[(t - tile * use_lab) * pal_size + pal_px[tile * 64 + src]];
    }
    if (SRC == 4) {  // row t = the (tile, palette) pair pairs[t] = tile << 32 | palette (all ones: a pad, black)
      const unsigned long long pr = reinterpret_cast<const unsigned long long *>(tiles)[t];
      return pr == ~0ull ? 0u : (uint32_t)palettes[(int64_t)(uint32_t)pr * pal_size + pal_px[(int64_t)(pr >> 32) * 64 + src]];
    }
    if (SRC == 2) {
      const int ww = pal_size - 7;
      const int64_t wy = t / ww, wx = t - wy * ww;
      return tiles[(wy + y) * pal_size + wx + x];  // CopyRGBPixels(ABackBuffer, x, AIndex), 879-887
    }
    return tiles[(pal_idx ? (int64_t)pal_idx[t] : t) * 64 + src];  // (SRC 0 with a row list: row t of the output is tile pal_idx[t])
  };
  // A wave works on its own tile and its own LDS rows: nothing here is shared between waves, so no barrier -- and the NEXT tile's
  // pixel (two or three dependent loads deep) is asked for before this tile's 192 dot products start, not after them
  uint32_t col = 0;
  if (blockIdx.x < niter && blockIdx.x * 4ll + wave < n) col = fetch(blockIdx.x * 4ll + wave);
  __syncthreads();  // s_lut is whole
  float lnorm;  // the Euclidean norm of this lane's LUT row, rounded up (4 for the plain DCT's rows)
  {
    double q2 = 0.0;
    for (int k = 0; k < 64; k++) { const double v = (double)s_lut[lane * 64 + ((k + lane) & 63)]; q2 = fma(v, v, q2); }  // (rotated: no bank conflict)
    lnorm = (float)(sqrt(q2) * (1.0 + 1e-6));
  }
  for (int64_t it = blockIdx.x; it < niter; it += gridDim.x, buf ^= 1) {
    const int64_t t_out = it * 4 + wave;
    const bool valid = t_out < n;
    float mine[3] = {0.0f, 0.0f, 0.0f};  // this lane's pixel, per plane
    if (valid) {
      float yy, uu, vv;
      if (use_lab && SRC != 3 && SRC != 4)
        rgb_to_lab_det(col & 0xff, (col >> 8) & 0xff, (col >> 16) & 0xff, srgb_lut, yy, uu, vv);
      else
        rgb_to_yuv(col & 0xff, (col >> 8) & 0xff, (col >> 16) & 0xff, yy, uu, vv);
      s_cpn[buf][wave][lane] = (double)yy;
      s_cpn[buf][wave][64 + lane] = (double)uu;
      s_cpn[buf][wave][128 + lane] = (double)vv;
      mine[0] = yy; mine[1] = uu; mine[2] = vv;
    }
    {
      const int64_t tn = (it + gridDim.x) * 4 + wave;
      if (it + gridDim.x < niter && tn < n) col = fetch(tn);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the wave's LDS operations are in order: its reads below see its writes above)
    if (valid) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double *cp = &s_cpn[buf][wave][c * 64];
        // The reference's sum (exact_sum below) costs 130 vector instructions per coefficient and plane: 64 Single products, Single sums
        // of pairs, 32 conversions, 34 double additions, in DCTInner_asm's order.  What is stored is only Round(z w).  So z is first
        // computed the cheap way -- the LUT is cos x cos x ratio: a row transform shared by the eight coefficients of a column of the
        // LUT, then a column transform, 16 fused multiply-adds in double -- together with a bound on how far the reference's z can lie
        // from it:  |z_ref - z_fast| <= sum|p_k| (2^-23 + 2^-24)(1 + 2^-20),  p_k = pixel x LUT entry (every product and every pair
        // sum of the reference rounds to Single once: 2 x 2^-24 relative; the LUT's Singles are within 2^-24 of the doubles they were
        // rounded from, which are this path's cos x cos x ratio to 2^-52; the double-precision steps of both paths stay below 2^-47),
        // and sum|p_k| <= |pixels| |LUT row| (Euclidean norms).  When z_fast w is farther than that (times |w|) from every half-integer, both
        // round to the same integer.  Otherwise -- a few coefficients per thousand -- the coefficient is summed the reference's way, by
        // the whole wave (lane k = product k; the order's pairings are lane exchanges: xor 4 in Single, xor 8 and xor 2 in double, then the
        // four 16-element steps in sequence).
        auto exact_sum = [&](int coef) -> double {  // z of coefficient `coef` in DCTInner_asm's order (utils.pas:892-921), uniform over the wave
          const float e = __fmul_rn(mine[c], s_lut[coef * 64 + lane]);
          const float s4 = __fadd_rn(e, __shfl_xor(e, 4));                 // (p0+p4, p1+p5, p2+p6, p3+p7 | p8+p12, ...): Single
          double d = (double)s4;
          d = __dadd_rn(d, __shfl_xor(d, 8));                              // (double)s + (double)t
          d = __dadd_rn(d, __shfl_xor(d, 2));                              // ... + the second pair: a0 in lane 16 j, a1 in lane 16 j + 1
          double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
          for (int j = 0; j < 4; j++) { acc0 = __dadd_rn(acc0, __shfl(d, 16 * j)); acc1 = __dadd_rn(acc1, __shfl(d, 16 * j + 1)); }
          return __dadd_rn(acc0, acc1);
        };
        double t;
        bool doubtful;
        int o_fast = 0;
        if (plain) {
          t = 0.0;
          doubtful = true;
        } else {
          // sum of the pixels' squares over the wave (row operations and four lane reads, no LDS; 64 non-negative Singles: within 2^-17 of exact)
          float sq = mine[c] * mine[c];
          sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0xB1, 0xf, 0xf, true));
          sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x4E, 0xf, 0xf, true));
          sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x141, 0xf, 0xf, true));
          sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x140, 0xf, 0xf, true));
          const float sq_all = (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq), 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq), 16))) +
                               (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq), 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sq), 48)));
          const float sa = lnorm * (__builtin_amdgcn_sqrtf(sq_all) * 1.00001f);  // >= sum |pixel x LUT entry| (Cauchy-Schwarz; lnorm = the LUT row's norm, rounded up; the hardware's root is good to 1 ulp)
          const double2 *fp = reinterpret_cast<const double2 *>(cp + (lane >> 3) * 8);
          const double2 f0 = fp[0], f1 = fp[1], f2 = fp[2], f3 = fp[3];
          double r = au[0] * f0.x;
          r = fma(au[1], f0.y, r); r = fma(au[2], f1.x, r); r = fma(au[3], f1.y, r);
          r = fma(au[4], f2.x, r); r = fma(au[5], f2.y, r); r = fma(au[6], f3.x, r); r = fma(au[7], f3.y, r);
          s_row[wave][(lane & 7) * 8 + (lane >> 3)] = r;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          const double2 *rp = reinterpret_cast<const double2 *>(&s_row[wave][(lane & 7) * 8]);
          const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next plane's row transforms land after these reads)
          double z = av[0] * r0.x;
          z = fma(av[1], r0.y, z); z = fma(av[2], r1.x, z); z = fma(av[3], r1.y, z);
          z = fma(av[4], r2.x, z); z = fma(av[5], r2.y, z); z = fma(av[6], r3.x, z); z = fma(av[7], r3.y, z);
          z *= ruv;
          t = weighted ? z * w[c] : z;
          doubtful = !first_look_rounds(t, sa * (1.82e-7f * (weighted ? fabsf((float)w[c]) * 1.000001f : 1.0f)), o_fast);  // 1.82e-7 > (2^-23 + 2^-24) (1 + 2^-17)(1 + 2^-20)
        }
        for (unsigned long long m = __builtin_amdgcn_ballot_w64(doubtful); m; m &= m - 1) {
          const int coef = __builtin_ctzll(m);
          const double z = exact_sum(coef);
          if (lane == coef) t = weighted ? __dmul_rn(z, w[c]) : z;
        }
        // Round(): half to even (3126), then the store into a SmallInt (the low 16 bits)
        const int16_t o = !doubtful ? (int16_t)o_fast : fabs(t) < 2.0e9 ? (int16_t)__double2int_rn(t) : (int16_t)__double2ll_rn(t);
        out[t_out * 192 + c * 64 + zz] = o;
        mmn[c] = min(mmn[c], (int)o);
        mmx[c] = max(mmx[c], (int)o);
      }
    }
  }
  if (colmm) {  // the search's digit plan wants the columns' ranges: kept here, a pass over all rows saved there (the four waves of a workgroup
                // hold the same columns: one atomic pair per column and workgroup)
    __shared__ int s_mm[4][384];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; c++) { s_mm[wave][c * 64 + zz] = mmn[c]; s_mm[wave][192 + c * 64 + zz] = mmx[c]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 192; i += 256) {
      const int a = min(min(s_mm[0][i], s_mm[1][i]), min(s_mm[2][i], s_mm[3][i]));
      const int b = max(max(s_mm[0][192 + i], s_mm[1][192 + i]), max(s_mm[2][192 + i], s_mm[3][192 + i]));
      if (a != INT_MAX) { atomicMin(&colmm[i], a); atomicMax(&colmm[192 + i], b); }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// A6 as DoPalettization uses it (tilingencoder.pas:4126, 4160): UseLAB planes, double LUT, sequential double sum
// (DCTInner<PDouble>, utils.pas:782-872), weight, then the build's Round() to int32 (DESIGN.md "k-means").
__global__ __launch_bounds__(256) void k_features_cluster_i32(const uint32_t *__restrict__ tiles, int64_t n, int weighted,
                                                              const double *__restrict__ lut, const double *__restrict__ weights,
                                                              const uint8_t *__restrict__ snake, const float *__restrict__ srgb_lut,
                                                              int32_t *__restrict__ out, const double *__restrict__ cosd /* [8][8] the LUT's cosine factor */, int plain) {
  // the planes as DOUBLES, widened once by the lane that made them
  __shared__ __attribute__((aligned(16))) double s_cpn[2][4][192];
  __shared__ __attribute__((aligned(16))) double s_row[4][64];  // a wave's row transforms, [u][y]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // The reference sums a coefficient's 64 double products one after the other (DCTInner<PDouble>, utils.pas:782-872) and stores Round(z w).
  // As in k_features_i16: z first by the LUT's two cosine factors (a row transform shared through LDS, then a column transform: 16 fused
  // multiply-adds), which differs from the sequential sum by less than 128 x 2^-53 x sum|pixel x LUT| < 4e-10 here (|pixel| <= 128, 64
  // of them, |w| < 2.7); a coefficient whose z w lies within 1e-6 of a half-integer -- two in a million -- is summed the reference's way.
  double au[8], av[8];
#pragma unroll
  for (int x = 0; x < 8; x++) { au[x] = cosd[(lane & 7) * 8 + x]; av[x] = cosd[(lane >> 3) * 8 + x]; }
  const double ruv = (lane == 0) ? 0.5 : (((lane & 7) == 0 || (lane >> 3) == 0) ? 0.707106769084930419921875 : 1.0);
  double w[3];
#pragma unroll
  for (int c = 0; c < 3; c++) w[c] = weights[c * 64 + lane];
  const int zz = snake[lane];
  const int64_t niter = (n + 3) / 4;
  int buf = 0;
  for (int64_t it = blockIdx.x; it < niter; it += gridDim.x, buf ^= 1) {
    const int64_t t = it * 4 + wave;
    const bool valid = t < n;
    if (valid) {
      const uint32_t col = tiles[t * 64 + lane];
      float yy, uu, vv;
      rgb_to_lab_det(col & 0xff, (col >> 8) & 0xff, (col >> 16) & 0xff, srgb_lut, yy, uu, vv);
      s_cpn[buf][wave][lane] = (double)yy;
      s_cpn[buf][wave][64 + lane] = (double)uu;
      s_cpn[buf][wave][128 + lane] = (double)vv;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (a wave reads only what it wrote: its LDS operations are in order)
    if (valid) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double *cp = &s_cpn[buf][wave][c * 64];
        auto in_order = [&]() -> double {  // the reference's sum
          double r = 0.0;
          for (int k = 0; k < 64; k += 2) {
            const double2 cv = *reinterpret_cast<const double2 *>(cp + k);
            const double2 lv = *reinterpret_cast<const double2 *>(lut + lane * 64 + k);
            r = __dadd_rn(r, __dmul_rn(cv.x, lv.x));
            r = __dadd_rn(r, __dmul_rn(cv.y, lv.y));
          }
          return weighted ? __dmul_rn(r, w[c]) : r;
        };
        double tv = 0.0;
        bool doubtful = true;
        if (!plain) {
          const double2 *fp = reinterpret_cast<const double2 *>(cp + (lane >> 3) * 8);
          const double2 f0 = fp[0], f1 = fp[1], f2 = fp[2], f3 = fp[3];
          double r = au[0] * f0.x;
          r = fma(au[1], f0.y, r); r = fma(au[2], f1.x, r); r = fma(au[3], f1.y, r);
          r = fma(au[4], f2.x, r); r = fma(au[5], f2.y, r); r = fma(au[6], f3.x, r); r = fma(au[7], f3.y, r);
          s_row[wave][(lane & 7) * 8 + (lane >> 3)] = r;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          const double2 *rp = reinterpret_cast<const double2 *>(&s_row[wave][(lane & 7) * 8]);
          const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          double z = av[0] * r0.x;
          z = fma(av[1], r0.y, z); z = fma(av[2], r1.x, z); z = fma(av[3], r1.y, z);
          z = fma(av[4], r2.x, z); z = fma(av[5], r2.y, z); z = fma(av[6], r3.x, z); z = fma(av[7], r3.y, z);
          z *= ruv;
          tv = weighted ? z * w[c] : z;
          doubtful = !(fabs(tv - floor(tv) - 0.5) > 1e-6) || !(fabs(tv) < 1.0e9);
        }
        if (__builtin_amdgcn_ballot_w64(doubtful)) {  // (uniform)
          const double ex = in_order();
          if (doubtful) tv = ex;
        }
        out[t * 192 + c * 64 + zz] = (int32_t)__double2ll_rn(tv);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// PearsonCorrelation (tilingencoder.pas:2201-2228) of consecutive frames' Lab tile means: one thread per frame, the
// reference's exact sequence (Math.mean sums in double; everything else sequential Single), so the sums are bit
// identical to a CPU restatement and 300 frames run side by side.  Output: num, sum dx^2, sum dy^2 per frame.
constexpr int PC = 2048;  // values per staged chunk
__global__ __launch_bounds__(64) void k_pearson_frames(const float *__restrict__ lab, int nframes, int per, float *__restrict__ correl) {
  // one wave per frame.  Only the ADDITIONS of the reference's sums are a chain; everything that feeds them is element-wise, so all
  // 64 lanes prepare a chunk in LDS (the doubles of the mean's terms, then the three products, each rounded exactly as the sequential
  // code rounds it) and lanes 0..2 only add, reading 16 bytes at a time
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[2 * PC * 8];
  __shared__ double s_sum[2];
  double (*s_d)[PC] = reinterpret_cast<double (*)[PC]>(s_raw);  // [2][PC], first pass
  float (*s_p)[PC] = reinterpret_cast<float (*)[PC]>(s_raw);    // [3][PC], second pass
  const int f = blockIdx.x, lane = threadIdx.x;
  if (f == 0) { if (lane < 3) correl[lane] = 0.0f; return; }
  const float *x = lab + (int64_t)(f - 1) * per, *y = lab + (int64_t)f * per;
  double acc = 0.0;
  for (int c0 = 0; c0 < per; c0 += PC) {
    const int n = min(PC, per - c0);
    __syncthreads();
    {  // all of a chunk's loads in flight at once: a lone wave has nothing else to cover their latency with
      float xv[PC / 64], yv[PC / 64];
#pragma unroll
      for (int u = 0; u < PC / 64; u++) { const int i = u * 64 + lane; xv[u] = i < n ? x[c0 + i] : 0.0f; yv[u] = i < n ? y[c0 + i] : 0.0f; }
#pragma unroll
      for (int u = 0; u < PC / 64; u++) { const int i = u * 64 + lane; if (i < n) { s_d[0][i] = (double)xv[u]; s_d[1][i] = (double)yv[u]; } }
    }
    __syncthreads();
    if (lane < 2) {
      const double *src = s_d[lane];
      int i = 0;
#pragma unroll 1
      for (; i + 32 <= n; i += 32) {  // sixteen reads issued before the first addition: their latency is paid once per 32 terms
        double2 v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = *reinterpret_cast<const double2 *>(src + i + 2 * u);
#pragma unroll
        for (int u = 0; u < 16; u++) { acc = __dadd_rn(acc, v[u].x); acc = __dadd_rn(acc, v[u].y); }
      }
      for (; i < n; i++) acc = __dadd_rn(acc, src[i]);
    }
  }
  if (lane < 2) s_sum[lane] = acc;
  __syncthreads();
  const float mx = (float)__ddiv_rn(s_sum[0], (double)per), my = (float)__ddiv_rn(s_sum[1], (double)per);
  float chain = 0.0f;  // lane 0: num = sum dx*dy, lane 1: denx = sum dx*dx, lane 2: deny = sum dy*dy
  for (int c0 = 0; c0 < per; c0 += PC) {
    const int n = min(PC, per - c0);
    __syncthreads();
    {
      float xv[PC / 64], yv[PC / 64];
#pragma unroll
      for (int u = 0; u < PC / 64; u++) { const int i = u * 64 + lane; xv[u] = i < n ? x[c0 + i] : 0.0f; yv[u] = i < n ? y[c0 + i] : 0.0f; }
#pragma unroll
      for (int u = 0; u < PC / 64; u++) {
        const int i = u * 64 + lane;
        if (i < n) {
          const float dx = __fsub_rn(xv[u], mx), dy = __fsub_rn(yv[u], my);
          s_p[0][i] = __fmul_rn(dx, dy);
          s_p[1][i] = __fmul_rn(dx, dx);
          s_p[2][i] = __fmul_rn(dy, dy);
        }
      }
    }
    __syncthreads();
    if (lane < 3) {
      const float *src = s_p[lane];
      int i = 0;
#pragma unroll 1
      for (; i + 64 <= n; i += 64) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = *reinterpret_cast<const float4 *>(src + i + 4 * u);
#pragma unroll
        for (int u = 0; u < 16; u++) {
          chain = __fadd_rn(chain, v[u].x); chain = __fadd_rn(chain, v[u].y); chain = __fadd_rn(chain, v[u].z); chain = __fadd_rn(chain, v[u].w);
        }
      }
      for (; i < n; i++) chain = __fadd_rn(chain, src[i]);
    }
  }
  // the three raw sums go back to the host, which finishes with IEEE sqrt/divide (device __fsqrt_rn is a bare
  // v_sqrt_f32: 1 ulp, not correctly rounded)
  if (lane < 3) correl[f * 3 + lane] = chain;
}

int launch_pearson(const void *lab, int nframes, int per, void *correl, hipStream_t stream) {
  TM_TRY(require_device());
  if (nframes <= 0) return TM_OK;
  hipLaunchKernelGGL(k_pearson_frames, dim3(nframes), dim3(64), 0, stream, (const float *)lab, nframes, per, (float *)correl);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

static int grid_for(int64_t work_items, int per_block) {
  int64_t g = (work_items + per_block - 1) / per_block;
  const int64_t cap = 256 * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// RGBToLAB (utils.pas:374-410) of n colours 0x00RRGGBB -> [n][3] Singles: the colour math of the load and feature kernels on its own
__global__ void k_rgb_to_lab(const uint32_t *__restrict__ rgb, int64_t n, const float *__restrict__ srgb_lut, float *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t c = rgb[i];
    float l, a, b;
    rgb_to_lab_det((c >> 16) & 0xff, (c >> 8) & 0xff, c & 0xff, srgb_lut, l, a, b);
    out[i * 3] = l; out[i * 3 + 1] = a; out[i * 3 + 2] = b;
  }
}
int launch_rgb_to_lab(const void *rgb, int64_t n, void *out, hipStream_t stream) {
  const DeviceTables *tab;
  TM_TRY(get_tables(&tab));
  TM_CHECK(n >= 0 && (n == 0 || (rgb && out)), TM_E_INVAL, "tm_stage_rgb_to_lab: bad arguments");
  if (n == 0) return TM_OK;
  hipLaunchKernelGGL(k_rgb_to_lab, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 256 * 16)), dim3(256), 0, stream, (const uint32_t *)rgb, n, tab->srgb_lut, (float *)out);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

int launch_load(const void *frames, int nframes, int img_w, int img_h, int tm_w, int tm_h, void *tiles, void *flags,
                void *lab_means, hipStream_t stream) {
  const DeviceTables *tab;
  TM_TRY(get_tables(&tab));
  TM_CHECK(nframes >= 0 && img_w > 0 && img_h > 0 && tm_w > 0 && tm_h > 0, TM_E_INVAL, "tm_stage_load: bad dimensions");
  const int64_t total = (int64_t)nframes * tm_w * tm_h;
  if (total == 0) return TM_OK;
  hipLaunchKernelGGL(k_load_tiles, dim3(grid_for(total, 4 * LT_BATCH)), dim3(256), 0, stream, (const uint32_t *)frames, nframes, img_w,
                     img_h, tm_w, tm_h, tab->srgb_lut, (uint32_t *)tiles, (uint8_t *)flags, (float *)lab_means);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// A4 + A5 for RGB tiles, plain (non-"special") DCT modes: k_features_i16<0>'s work with a lane per (tile, row) / (tile, u) instead of a lane
// per coefficient -- a wave takes EIGHT tiles.  A lane converts its row's eight pixels, takes them through the 8-point fast DCT (35 double-precision
// operations for the row's eight outputs), the outputs change hands through LDS (transposed), and the lane of (tile, u) takes the eight
// row transforms R[u][y] through the same DCT down the column: 70 operations a lane for 8 x 8 coefficients where a lane per coefficient spends
// sixteen multiply-adds on one.  Weight, the first look's verdict, the in-doubt coefficients in the reference's order by the whole wave,
// exactly as in k_features_i16 and k_window_dcts; a plane's coefficients leave through LDS as 128-byte lines.  (The first look's z may be
// summed in any order: its bound has nine decimal orders of room.)
constexpr int F8_TP = 72;  // pitch (doubles) of a tile's transposed row transforms: eight tiles' lanes then cover the banks evenly
__global__ __launch_bounds__(256) void k_features_tiles8(const uint32_t *__restrict__ tiles, const int32_t *__restrict__ rows, const uint8_t *__restrict__ mirror_flags, int64_t n,
                                                         int weighted, int use_lab, const float *__restrict__ lut, const double *__restrict__ weights,
                                                         const uint8_t *__restrict__ snake, const float *__restrict__ srgb_lut, int16_t *__restrict__ out,
                                                         int *__restrict__ colmm /* null, or [384]: running min / max of the 192 output columns (atomics) */, int plain) {
  __shared__ double s_cw[3][64];
  __shared__ float s_kw[3][64];
  __shared__ int s_zz[64];
  __shared__ __attribute__((aligned(16))) float s_px[4][3][8][64];   // the planes as Singles, for the in-order sums; at the very end: the columns' ranges
  __shared__ __attribute__((aligned(16))) double s_t[4][8 * F8_TP];   // a wave's eight tiles, one plane: row transforms [tile][u][y]
  __shared__ __attribute__((aligned(16))) int16_t s_out[4][8 * 64];   // ... and its coefficients in output order
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane >> 3, y = lane & 7, u = y;
  if (tid < 64) {
    const double ruv = (tid == 0) ? 0.5 : (((tid & 7) == 0 || (tid >> 3) == 0) ? 0.707106769084930419921875 : 1.0);  // cDCTUVRatio (utils.pas:100-109)
    double q2 = 0.0;
    for (int k = 0; k < 64; k++) { const double v = (double)lut[tid * 64 + k]; q2 = fma(v, v, q2); }
    const float ln = (float)(sqrt(q2) * (1.0 + 1e-6));  // the LUT row's Euclidean norm, rounded up
    for (int c = 0; c < 3; c++) {
      const double wv = weighted ? weights[c * 64 + tid] : 1.0;
      s_cw[c][tid] = ruv * wv;
      s_kw[c][tid] = 1.82e-7f * (fabsf((float)wv) * 1.000001f) * ln * 1.000001f;  // 1.82e-7 > (2^-23 + 2^-24) (1 + 2^-17)(1 + 2^-20)
    }
    s_zz[tid] = snake[tid];
  }
  __syncthreads();
  constexpr double C1 = 0.98078528040323044913, C2 = 0.92387953251128675613, C3 = 0.83146961230254523708, C4 = 0.70710678118654752440,
                   C5 = 0.55557023301960222474, C6 = 0.38268343236508977173, C7 = 0.19509032201612826785;  // cos(k pi / 16)
  auto dct8 = [&](const double (&p)[8], double (&r)[8]) {
    const double s0 = p[0] + p[7], s1 = p[1] + p[6], s2 = p[2] + p[5], s3 = p[3] + p[4], d0 = p[0] - p[7], d1 = p[1] - p[6], d2 = p[2] - p[5], d3 = p[3] - p[4];
    const double e0 = s0 + s3, e1 = s1 + s2, e2 = s0 - s3, e3 = s1 - s2;
    r[0] = e0 + e1;
    r[4] = C4 * (e0 - e1);
    r[2] = fma(C2, e2, C6 * e3);
    r[6] = fma(C6, e2, -(C2 * e3));
    r[1] = fma(C1, d0, fma(C3, d1, fma(C5, d2, C7 * d3)));
    r[3] = fma(C3, d0, fma(-C7, d1, fma(-C1, d2, -(C5 * d3))));
    r[5] = fma(C5, d0, fma(-C1, d1, fma(C7, d2, C3 * d3)));
    r[7] = fma(C7, d0, fma(-C5, d1, fma(C3, d2, -(C1 * d3))));
  };
  int zzr[8];
#pragma unroll
  for (int v = 0; v < 8; v++) zzr[v] = j * 64 + s_zz[v * 8 + u];  // the place of coefficient (u, v) in the wave's run
  int mmn[3] = {INT_MAX, INT_MAX, INT_MAX}, mmx[3] = {INT_MIN, INT_MIN, INT_MIN};  // of output column `lane` of each plane
  int16_t *const so = s_out[wave];
  double *const st = s_t[wave];
  const int64_t ngroups = (n + 7) >> 3;
  for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < ngroups; g += (int64_t)gridDim.x * 4) {
    const int64_t t_out = g * 8 + j;
    const bool valid = t_out < n;
    float pl[3][8];
    {  // this lane's row of its tile, as ConvertToCpnPixels reads it (mirrors 3080-3085 / 3093-3098)
      const int64_t tile = valid ? (rows ? (int64_t)rows[t_out] : t_out) : (rows ? (int64_t)rows[n - 1] : n - 1);
      const int f = (valid && mirror_flags) ? mirror_flags[t_out] : 0;
      const uint4 *src = reinterpret_cast<const uint4 *>(tiles + tile * 64 + ((f & 2) ? 7 - y : y) * 8);
      const uint4 a = src[0], b = src[1];
      uint32_t px[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      if (f & 1) {
#pragma unroll
        for (int x = 0; x < 4; x++) { const uint32_t tmp = px[x]; px[x] = px[7 - x]; px[7 - x] = tmp; }
      }
#pragma unroll
      for (int x = 0; x < 8; x++) {
        if (use_lab) rgb_to_lab_det(px[x] & 0xff, (px[x] >> 8) & 0xff, (px[x] >> 16) & 0xff, srgb_lut, pl[0][x], pl[1][x], pl[2][x]);
        else rgb_to_yuv(px[x] & 0xff, (px[x] >> 8) & 0xff, (px[x] >> 16) & 0xff, pl[0][x], pl[1][x], pl[2][x]);
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        float4 *dst = reinterpret_cast<float4 *>(&s_px[wave][c][j][y * 8]);
        dst[0] = make_float4(pl[c][0], pl[c][1], pl[c][2], pl[c][3]);
        dst[1] = make_float4(pl[c][4], pl[c][5], pl[c][6], pl[c][7]);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {  // (unrolled: the planes' registers and ranges are then addressed statically -- no scratch)
      unsigned dmask = 0;  // bit v: coefficient (u, v) of tile j is in doubt
      if (!plain) {
        float sq = 0.0f;
        double p[8], r[8];
#pragma unroll
        for (int x = 0; x < 8; x++) { sq = fmaf(pl[c][x], pl[c][x], sq); p[x] = (double)pl[c][x]; }
        // the tile's sum of squares over its eight lanes (row moves: lane ^ 1, lane ^ 2, the mirror image inside eight)
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0xB1, 0xf, 0xf, true));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x4E, 0xf, 0xf, true));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sq), 0x141, 0xf, 0xf, true));
        const float root = __builtin_amdgcn_sqrtf(sq) * 1.00001f;  // x the LUT row's norm >= sum |pixel x LUT entry| (Cauchy-Schwarz)
        dct8(p, r);
#pragma unroll
        for (int uu = 0; uu < 8; uu++) st[j * F8_TP + uu * 8 + y] = r[uu];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (a wave's LDS operations are in order)
        {
          const double2 *rp = reinterpret_cast<const double2 *>(st + j * F8_TP + u * 8);
          const double2 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
          p[0] = r0.x; p[1] = r0.y; p[2] = r1.x; p[3] = r1.y; p[4] = r2.x; p[5] = r2.y; p[6] = r3.x; p[7] = r3.y;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next plane's row transforms land after these reads)
        double z[8];
        dct8(p, z);
#pragma unroll
        for (int v = 0; v < 8; v++) {
          const double t = z[v] * s_cw[c][v * 8 + u];  // cDCTUVRatio and the weight in one factor
          int o;
          const bool ok = first_look_rounds(t, s_kw[c][v * 8 + u] * root, o);
          so[zzr[v]] = (int16_t)o;
          dmask |= ok ? 0u : (1u << v);
        }
        if (!valid) dmask = 0;
      } else dmask = valid ? 0xffu : 0u;
      // the coefficients in doubt, one after the other by the whole wave: lane k = product k of DCTInner_asm's sum (utils.pas:892-921)
      for (unsigned long long m = __builtin_amdgcn_ballot_w64(dmask != 0); m; m &= m - 1) {
        const int src = __builtin_ctzll(m);
        const int sj = src >> 3, su = src & 7;
        const float mine = s_px[wave][c][sj][lane];
        for (unsigned vm = (unsigned)__builtin_amdgcn_readlane((int)dmask, src); vm; vm &= vm - 1) {
          const int coef = __builtin_ctz(vm) * 8 + su;
          const float e = __fmul_rn(mine, lut[coef * 64 + lane]);
          const float s4 = __fadd_rn(e, __shfl_xor(e, 4));
          double d = (double)s4;
          d = __dadd_rn(d, __shfl_xor(d, 8));
          d = __dadd_rn(d, __shfl_xor(d, 2));
          double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
          for (int q = 0; q < 4; q++) { acc0 = __dadd_rn(acc0, __shfl(d, 16 * q)); acc1 = __dadd_rn(acc1, __shfl(d, 16 * q + 1)); }
          const double zex = __dadd_rn(acc0, acc1);
          const double t = weighted ? __dmul_rn(zex, weights[c * 64 + coef]) : zex;
          // Round(): half to even (3126), then the store into a SmallInt (the low 16 bits)
          if (lane == src) so[sj * 64 + s_zz[coef]] = fabs(t) < 2.0e9 ? (int16_t)__double2int_rn(t) : (int16_t)__double2ll_rn(t);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (colmm) {  // output column `lane` of this plane over the run's tiles
#pragma unroll
        for (int jj = 0; jj < 8; jj++)
          if (g * 8 + jj < n) { const int v = so[jj * 64 + lane]; mmn[c] = min(mmn[c], v); mmx[c] = max(mmx[c], v); }
      }
      if (valid) *reinterpret_cast<uint4 *>(out + t_out * 192 + c * 64 + u * 8) = *reinterpret_cast<const uint4 *>(so + j * 64 + u * 8);  // lane = (tile, 16-byte piece)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next plane's coefficients land after these reads)
    }
  }
  if (colmm) {  // the search's digit plan wants the columns' ranges: one atomic pair per column and workgroup
    int *s_mm = reinterpret_cast<int *>(&s_px[0][0][0][0]);  // [4][384]
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; c++) { s_mm[wave * 384 + c * 64 + lane] = mmn[c]; s_mm[wave * 384 + 192 + c * 64 + lane] = mmx[c]; }
    __syncthreads();
    for (int i = tid; i < 192; i += 256) {
      const int a = min(min(s_mm[i], s_mm[384 + i]), min(s_mm[768 + i], s_mm[1152 + i]));
      const int b = max(max(s_mm[192 + i], s_mm[384 + 192 + i]), max(s_mm[768 + 192 + i], s_mm[1152 + 192 + i]));
      if (a != INT_MAX) { atomicMin(&colmm[i], a); atomicMax(&colmm[192 + i], b); }
    }
  }
}

// k_features_tiles8 where it applies (RGB tiles, the plain DCT's cosines); TM_FEATURES_BY_TILE=1: always the tile-at-a-time kernel (tests, A/B)
static bool features_tiles8(int mode, int64_t n, const void *tiles, const void *rows, const void *mirror_flags, int use_lab, void *out, void *colmm, const DeviceTables *tab,
                            hipStream_t stream) {
  if (mode_special(mode) || knobs().features_by_tile) return false;
  const int64_t groups = (n + 7) / 8;
  hipLaunchKernelGGL(k_features_tiles8, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((groups + 3) / 4, 256 * 6))), dim3(256), 0, stream, (const uint32_t *)tiles,
                     (const int32_t *)rows, (const uint8_t *)mirror_flags, n, mode_weighted(mode) ? 1 : 0, use_lab, tab->dct_lut_f32[0], tab->weights, tab->snake, tab->srgb_lut,
                     (int16_t *)out, (int *)colmm, knobs().features_plain ? 1 : 0);
  return true;
}

int launch_features_rgb(const void *tiles, int64_t n, const void *mirror_flags, int mode, int use_lab, void *out,
                        hipStream_t stream) {
  const DeviceTables *tab;
  TM_TRY(get_tables(&tab));
  TM_CHECK(mode != TM_PVS_WAVELETS, TM_E_UNSUPPORTED, "wavelet features on int16 vectors are unimplemented in the reference too (tilingencoder.pas:3111)");
  TM_CHECK(mode >= 0 && mode <= 4, TM_E_INVAL, "bad TPsyVisMode %d", mode);
  if (n <= 0) return TM_OK;
  if (features_tiles8(mode, n, tiles, nullptr, mirror_flags, use_lab, out, nullptr, tab, stream)) { TM_HIP(hipGetLastError()); return TM_OK; }
  hipLaunchKernelGGL(k_features_i16<0>, dim3(grid_for(n, 4)), dim3(256), 0, stream, (const uint32_t *)tiles, nullptr,
                     nullptr, nullptr, 0, (const uint8_t *)mirror_flags, n, mode_weighted(mode) ? 1 : 0, use_lab,
                     tab->dct_lut_f32[mode_special(mode)], tab->weights, tab->snake, tab->srgb_lut, (int16_t *)out, (int *)nullptr, tab->dct_cos_f64[mode_special(mode)], knobs().features_plain ? 1 : 0);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

// the same for a list of rows: output row i = features of tile rows[i] (Reconstruct's queries are the DISTINCT frame tiles)
int launch_features_rgb_rows(const void *tiles, const void *rows, int64_t n, int mode, int use_lab, void *out, hipStream_t stream, void *colmm) {
  const DeviceTables *tab;
  TM_TRY(get_tables(&tab));
  TM_CHECK(mode != TM_PVS_WAVELETS && mode >= 0 && mode <= 4, TM_E_INVAL, "bad TPsyVisMode %d", mode);
  if (n <= 0) return TM_OK;
  if (features_tiles8(mode, n, tiles, rows, nullptr, use_lab, out, colmm, tab, stream)) { TM_HIP(hipGetLastError()); return TM_OK; }
  const int grid = grid_for(n, 4);
  hipLaunchKernelGGL(k_features_i16<0>, dim3(grid), dim3(256), 0, stream, (const uint32_t *)tiles, nullptr,
                     (const int32_t *)rows, nullptr, 0, (const uint8_t *)nullptr, n, mode_weighted(mode) ? 1 : 0, use_lab,
                     tab->dct_lut_f32[mode_special(mode)], tab->weights, tab->snake, tab->srgb_lut, (int16_t *)out, (int *)colmm, tab->dct_cos_f64[mode_special(mode)], knobs().features_plain ? 1 : 0);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

int launch_features_pal(const void *pal_px, const void *pal_idx, int64_t n, const void *palettes, int pal_size, int mode,
                        void *out, hipStream_t stream) {
  const DeviceTables *tab;
  TM_TRY(get_tables(&tab));
  TM_CHECK(mode != TM_PVS_WAVELETS && mode >= 0 && mode <= 4, TM_E_INVAL, "bad TPsyVisMode %d", mode);
  TM_CHECK(pal_size >= 1 && pal_size <= 256, TM_E_INVAL, "bad palette size %d", pal_size);
  if (n <= 0) return TM_OK;
  hipLaunchKernelGGL(k_features_i16<1>, dim3(grid_for(n, 4)), dim3(256), 0, stream, nullptr, (const uint8_t *)pal_px,
                     (const int32_t *)pal_idx, (const int32_t *)palettes, pal_size, nullptr, n, mode_weighted(mode) ? 1 : 0, 0,
                     tab->dct_lut_f32[mode_special(mode)], tab->weights, tab->snake, tab->srgb_lut, (int16_t *)out, (int *)nullptr, tab->dct_cos_f64[mode_special(mode)], knobs().features_plain ? 1 : 0);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

int launch_features_pairs(const void *pal_px, const void *pairs, int64_t n, const void *palettes, int pal_size, void *out, hipStream_t stream) {
  const DeviceTables *tab;
  TM_TRY(get_tables(&tab));
  if (n <= 0) return TM_OK;
  hipLaunchKernelGGL(k_features_i16<4>, dim3(grid_for(n, 4)), dim3(256), 0, stream, (const uint32_t *)pairs, (const uint8_t *)pal_px, nullptr,
                     (const int32_t *)palettes, pal_size, nullptr, n, 1, 0, tab->dct_lut_f32[0], tab->weights, tab->snake, tab->srgb_lut, (int16_t *)out, (int *)nullptr, tab->dct_cos_f64[0], knobs().features_plain ? 1 : 0);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

int launch_features_table(const void *pal_px, int64_t ntiles, const void *palettes, int npal, int pal_size, void *out, hipStream_t stream) {
  const DeviceTables *tab;
  TM_TRY(get_tables(&tab));
  TM_CHECK(pal_size >= 1 && pal_size <= 256 && npal >= 1, TM_E_INVAL, "bad palette shape %d x %d", npal, pal_size);
  const int64_t n = ntiles * npal;
  if (n <= 0) return TM_OK;
  hipLaunchKernelGGL(k_features_i16<3>, dim3(grid_for(n, 4)), dim3(256), 0, stream, nullptr, (const uint8_t *)pal_px, nullptr,
                     (const int32_t *)palettes, pal_size, nullptr, n, 1, npal, tab->dct_lut_f32[mode_special(TM_PVS_WEIGHTED_DCT)], tab->weights,
                     tab->snake, tab->srgb_lut, (int16_t *)out, (int *)nullptr, tab->dct_cos_f64[mode_special(TM_PVS_WEIGHTED_DCT)], knobs().features_plain ? 1 : 0);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Every 8 x 8 window of a frame buffer (PredictMotion.DoDCTs / Reconstruct.DoDCTs, tilingencoder.pas:1157-1182, 1437-1462), weighted DCT on
// YUV: what k_features_i16<2> computes a window at a time -- 64 colour conversions, eight row transforms and a column transform per window
// and plane -- shares nearly all of it between windows: a pixel is converted once per strip (not 64 times), a row's transform at a
// horizontal position serves the eight windows that contain it (not one), and only the column transform and the rounding are a window's
// own.  The reference's order of summation shares nothing (DCTInner_asm, fixed order in Single and double), but only Round(z w) is stored:
// z the cheap way first, the reference's order when the rounding is in doubt, exactly as in k_features_i16 (same bound, same fallback).
// A workgroup takes strips of WD_RY x WD_CX windows: (1) the strip's pixels -> three planes of Singles in LDS; per plane: (2) the row
// transform of every (row, horizontal position) by an 8-point fast DCT in double (35 operations for eight outputs: any order of summation
// serves, the bound on |z_ref - z_fast| has nine decimal orders of room for it) and the row's sum of squares, (3) the windows' sums of
// squares, (4) a wave per window, lane = coefficient (u, v): eight multiply-adds down the column, weight, slack, Round.
constexpr int WD_RY = 8, WD_CX = 32, WD_PW = WD_CX + 8;  // window rows and columns of a strip; pitch of a plane row (WD_CX + 7 pixels)
// PACK: the coefficients leave in the motion search's matrix layout (tm_internal.h: launch_window_dcts_packed) instead of as int16 rows.  A strip's
// 32 columns are one block of that layout; lane (window j, piece u) holds block b = 8 c + u of the reference's 24 blocks of eight coefficients
// (CompareEuclideanDCTPtr_asm, utils.pas:559-725: two halves of twelve) and knows where its digits go: plain blocks (0-4, 7-11 of a half) to matrix
// block 10 half + (0..9), block 6 joined by block 5 (its lane neighbour) to matrix block 20 + half as b5 + b6, block 5's lane fills the padding
// (matrix blocks 22 / 23) with zeros; block 7 of the first half is stored raw as well; the squares of what went into the matrix are summed per window
// over the three planes.  Low digit = the int16's low byte, signed; high digit = its high byte + the low byte's top bit.
typedef short wd_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short wd_u16x2 __attribute__((ext_vector_type(2)));
template <bool PACK>
__global__ __launch_bounds__(256) void k_window_dcts(const uint32_t *__restrict__ fb, int w, int h, const float *__restrict__ lut, const double *__restrict__ weights,
                                                     const uint8_t *__restrict__ snake, const double *__restrict__ cosd, int16_t *__restrict__ out, int plain,
                                                     const int *__restrict__ only_if, uint8_t *__restrict__ packed, const int16_t *__restrict__ cur, int ntiles,
                                                     int *__restrict__ flag) {
  if (only_if && !*only_if) return;  // (the fallback's launch: the matrix search took this frame)
  __shared__ uint32_t s_nrm[PACK ? WD_RY : 1][WD_CX];  // PACK: the windows' sums of squares of what the matrix holds, over the planes
  bool bad = false;
  if constexpr (PACK) {  // the tile side's range check (its digits are made inside the search kernel)
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < (int64_t)ntiles * 24; i += (int64_t)gridDim.x * 256) {
      const int blk = (int)(i % 24) % 12;
      if (blk == 5) continue;  // never enters on the tile side
      const int lim = blk == 6 ? MM_LIMIT6 : MM_LIMIT;
      const uint4 v = reinterpret_cast<const uint4 *>(cur)[i];
      const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int lo = (int16_t)(wv[k] & 0xffff), hi = (int16_t)(wv[k] >> 16);
        bad |= lo > lim || lo < -lim || hi > lim || hi < -lim;
      }
    }
  }
  __shared__ float s_pl[3][WD_RY + 7][WD_PW];                                     // the strip's planes (Singles, as ConvertToCpnPixels leaves them)
  __shared__ __attribute__((aligned(16))) double s_r[WD_RY + 7][WD_CX][8];         // one plane's row transforms [row][horizontal position][u]
  __shared__ float s_rs[WD_RY + 7][WD_CX];                                          // ... and the rows' sums of squares
  __shared__ float s_ws[WD_RY][WD_CX];                                              // the windows' sums of squares
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ww = w - 7, wh = h - 7;
  const int nsx = (ww + WD_CX - 1) / WD_CX, nsy = (wh + WD_RY - 1) / WD_RY;
  // per coefficient cf = v * 8 + u: cDCTUVRatio (utils.pas:100-109) x weight, the slack's factor, the zig-zag place
  // (read from LDS per coefficient: eight doubles, sixteen Singles and eight indices per lane in registers instead cost a wave per SIMD: 0.35 against 0.32 ms)
  __shared__ double s_cw[3][64];
  __shared__ float s_kw[3][64];  // 1.82e-7 x |weight| x the LUT row's norm: what of the slack is the coefficient's own
  __shared__ int s_zz[64];
  __shared__ __attribute__((aligned(16))) int16_t s_out[4][8 * 64];  // a wave's run of eight windows, one plane
  if (tid < 64) {
    const double ruv = (tid == 0) ? 0.5 : (((tid & 7) == 0 || (tid >> 3) == 0) ? 0.707106769084930419921875 : 1.0);
    double q2 = 0.0;
    for (int k = 0; k < 64; k++) { const double v = (double)lut[tid * 64 + k]; q2 = fma(v, v, q2); }
    const float ln = (float)(sqrt(q2) * (1.0 + 1e-6));
    for (int c = 0; c < 3; c++) { const double wv = weights[c * 64 + tid]; s_cw[c][tid] = ruv * wv; s_kw[c][tid] = 1.82e-7f * (fabsf((float)wv) * 1.000001f) * ln * 1.000001f; }
    s_zz[tid] = snake[tid];
  }
  (void)cosd;
  // cos(k pi / 16)
  constexpr double C1 = 0.98078528040323044913, C2 = 0.92387953251128675613, C3 = 0.83146961230254523708, C4 = 0.70710678118654752440,
                   C5 = 0.55557023301960222474, C6 = 0.38268343236508977173, C7 = 0.19509032201612826785;
  for (int strip = blockIdx.x; strip < nsx * nsy; strip += gridDim.x) {
    const int sy = strip / nsx, sx = strip - sy * nsx;
    const int y0 = sy * WD_RY, x0 = sx * WD_CX;
    __syncthreads();  // the strip before is done with the planes
    for (int e = tid; e < (WD_RY + 7) * (WD_CX + 7); e += 256) {  // pixels beyond the frame repeat its edge: they only feed windows that are not stored
      const int py = e / (WD_CX + 7), px = e - py * (WD_CX + 7);
      const uint32_t col = fb[(int64_t)min(y0 + py, h - 1) * w + min(x0 + px, w - 1)];  // CopyRGBPixels(ABackBuffer, x, AIndex), 879-887
      float yy, uu, vv;
      rgb_to_yuv(col & 0xff, (col >> 8) & 0xff, (col >> 16) & 0xff, yy, uu, vv);
      s_pl[0][py][px] = yy; s_pl[1][py][px] = uu; s_pl[2][py][px] = vv;
    }
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
      __syncthreads();  // the planes are whole / the plane before is done with s_r
      if (!plain)
      for (int e = tid; e < (WD_RY + 7) * WD_CX; e += 256) {
        const int ry = e / WD_CX, rx = e - ry * WD_CX;
        const float *pp = &s_pl[c][ry][rx];
        const float f0 = pp[0], f1 = pp[1], f2 = pp[2], f3 = pp[3], f4 = pp[4], f5 = pp[5], f6 = pp[6], f7 = pp[7];
        s_rs[ry][rx] = ((f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3)) + ((f4 * f4 + f5 * f5) + (f6 * f6 + f7 * f7));
        const double p0 = f0, p1 = f1, p2 = f2, p3 = f3, p4 = f4, p5 = f5, p6 = f6, p7 = f7;
        const double s0 = p0 + p7, s1 = p1 + p6, s2 = p2 + p5, s3 = p3 + p4, d0 = p0 - p7, d1 = p1 - p6, d2 = p2 - p5, d3 = p3 - p4;
        const double e0 = s0 + s3, e1 = s1 + s2, e2 = s0 - s3, e3 = s1 - s2;
        double *r = s_r[ry][rx];
        r[0] = e0 + e1;
        r[4] = C4 * (e0 - e1);
        r[2] = fma(C2, e2, C6 * e3);
        r[6] = fma(C6, e2, -(C2 * e3));
        r[1] = fma(C1, d0, fma(C3, d1, fma(C5, d2, C7 * d3)));
        r[3] = fma(C3, d0, fma(-C7, d1, fma(-C1, d2, -(C5 * d3))));
        r[5] = fma(C5, d0, fma(-C1, d1, fma(C7, d2, C3 * d3)));
        r[7] = fma(C7, d0, fma(-C5, d1, fma(C3, d2, -(C1 * d3))));
      }
      __syncthreads();
      if (!plain) {
        const int wy = tid / WD_CX, wx = tid - wy * WD_CX;  // 256 threads = WD_RY x WD_CX windows
        float sq = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) sq += s_rs[wy + k][wx];
        s_ws[wy][wx] = sq;
      }
      __syncthreads();
      // (4) a wave per run of EIGHT consecutive windows of a strip row, lane = (window j, u): the lane's eight row transforms R[wy .. wy + 7][wx][u] go
      // through the same 8-point fast DCT down the column -- 35 double-precision operations for its eight coefficients v = 0..7 where a lane per
      // coefficient spent eight multiply-adds on ONE -- then weight, verdict, and the run's 8 x 64 coefficients of this plane leave through LDS as
      // eight 128-byte lines (a lane per 16 bytes; a lane per coefficient stored two bytes at a time).
      // (the lane's eight factors of this plane, and its zig-zag places, once per plane instead of once per coefficient)
      double cwr[8];
      float kwr[8];
      int zzr[8];
#pragma unroll
      for (int v = 0; v < 8; v++) { cwr[v] = s_cw[c][v * 8 + (lane & 7)]; kwr[v] = s_kw[c][v * 8 + (lane & 7)]; zzr[v] = (lane >> 3) * 64 + s_zz[v * 8 + (lane & 7)]; }  // (zzr: the place in the wave's run)
      // PACK: this lane's block of eight coefficients in this plane, where it goes, its range (see the kernel's head)
      const int pk_b = c * 8 + (lane & 7), pk_hf = pk_b >= 12 ? 1 : 0, pk_bi = pk_b - pk_hf * 12;
      const int pk_mb = pk_bi < 5 ? pk_hf * 10 + pk_bi : pk_bi == 5 ? 22 + pk_hf : pk_bi == 6 ? 20 + pk_hf : pk_hf * 10 + pk_bi - 2;
      const uint32_t pk_lim = ((pk_bi == 5 || pk_bi == 6) ? MM_LIMIT6 : MM_LIMIT) * 0x00010001u;
      const uint32_t keep_own = pk_bi == 5 ? 0u : 0xffffffffu, keep_left = pk_bi == 6 ? 0xffffffffu : 0u;
      const unsigned piece0 = (unsigned)(((pk_mb >> 2) * 64 + ((pk_mb >> 1) & 1) * 32 + (lane >> 3)) * 16 + (pk_mb & 1) * 8);
      for (int gi = wave; gi < WD_RY * (WD_CX / 8); gi += 4) {
        const int wy = gi / (WD_CX / 8), wx0 = (gi - wy * (WD_CX / 8)) * 8;
        if (y0 + wy >= wh || x0 + wx0 >= ww) continue;  // (uniform in the wave)
        const int j = lane >> 3, u = lane & 7, wx = wx0 + j;
        unsigned dmask = 0;  // bit v: coefficient (u, v) of window j is in doubt
        int16_t *so = s_out[wave];
        if (!plain) {
          const double *rp = &s_r[wy][wx][u];
          const double p0 = rp[0], p1 = rp[WD_CX * 8], p2 = rp[2 * WD_CX * 8], p3 = rp[3 * WD_CX * 8], p4 = rp[4 * WD_CX * 8], p5 = rp[5 * WD_CX * 8], p6 = rp[6 * WD_CX * 8],
                       p7 = rp[7 * WD_CX * 8];
          const double s0 = p0 + p7, s1 = p1 + p6, s2 = p2 + p5, s3 = p3 + p4, d0 = p0 - p7, d1 = p1 - p6, d2 = p2 - p5, d3 = p3 - p4;
          const double e0 = s0 + s3, e1 = s1 + s2, e2 = s0 - s3, e3 = s1 - s2;
          double z[8];
          z[0] = e0 + e1;
          z[4] = C4 * (e0 - e1);
          z[2] = fma(C2, e2, C6 * e3);
          z[6] = fma(C6, e2, -(C2 * e3));
          z[1] = fma(C1, d0, fma(C3, d1, fma(C5, d2, C7 * d3)));
          z[3] = fma(C3, d0, fma(-C7, d1, fma(-C1, d2, -(C5 * d3))));
          z[5] = fma(C5, d0, fma(-C1, d1, fma(C7, d2, C3 * d3)));
          z[7] = fma(C7, d0, fma(-C5, d1, fma(C3, d2, -(C1 * d3))));
          const float root = __builtin_amdgcn_sqrtf(s_ws[wy][wx]) * 1.00001f;
#pragma unroll
          for (int v = 0; v < 8; v++) {
            const double t = z[v] * cwr[v];  // cDCTUVRatio and the weight in one factor
            int o;
            const bool ok = first_look_rounds(t, kwr[v] * root, o);  // root x the LUT row's norm >= sum |pixel x LUT entry| (Cauchy-Schwarz)
            so[zzr[v]] = (int16_t)o;
            dmask |= ok ? 0u : (1u << v);
          }
        } else dmask = 0xffu;
        // the coefficients in doubt, one after the other by the whole wave: lane k = product k of DCTInner_asm's sum (utils.pas:892-921)
        for (unsigned long long m = __builtin_amdgcn_ballot_w64(dmask != 0); m; m &= m - 1) {
          const int src = __builtin_ctzll(m);
          const int sj = src >> 3, su = src & 7;
          const float mine = s_pl[c][wy + (lane >> 3)][wx0 + sj + (lane & 7)];  // this lane's pixel of window sj
          for (unsigned vm = (unsigned)__builtin_amdgcn_readlane((int)dmask, src); vm; vm &= vm - 1) {
            const int coef = __builtin_ctz(vm) * 8 + su;
            const float e = __fmul_rn(mine, lut[coef * 64 + lane]);
            const float s4 = __fadd_rn(e, __shfl_xor(e, 4));
            double d = (double)s4;
            d = __dadd_rn(d, __shfl_xor(d, 8));
            d = __dadd_rn(d, __shfl_xor(d, 2));
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++) { acc0 = __dadd_rn(acc0, __shfl(d, 16 * q)); acc1 = __dadd_rn(acc1, __shfl(d, 16 * q + 1)); }
            const double t = __dmul_rn(__dadd_rn(acc0, acc1), weights[c * 64 + coef]);
            // Round(): half to even (3126), then the store into a SmallInt (the low 16 bits)
            if (lane == src) so[sj * 64 + s_zz[coef]] = fabs(t) < 2.0e9 ? (int16_t)__double2int_rn(t) : (int16_t)__double2ll_rn(t);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (a wave's LDS operations are in order: its reads below see its writes above)
        if constexpr (!PACK) {
          if (x0 + wx < ww) {  // lane = (window j, 16-byte piece u) of the plane's 128 bytes
            const int64_t t_out = (int64_t)(y0 + wy) * ww + x0 + wx;
            *reinterpret_cast<uint4 *>(out + t_out * 192 + c * 64 + u * 8) = *reinterpret_cast<const uint4 *>(so + j * 64 + u * 8);
          }
        } else {
          const bool live = x0 + wx < ww;  // (a window position past the row's end is all zeros, as k_mo_pack_win leaves it)
          const uint4 raw = live ? *reinterpret_cast<const uint4 *>(so + j * 64 + u * 8) : make_uint4(0, 0, 0, 0);
          uint32_t x[4] = {raw.x, raw.y, raw.z, raw.w};
          // range: |v| <= lim  <=>  (uint16)(v + lim) <= 2 lim
          wd_u16x2 rg = __builtin_bit_cast(wd_u16x2, x[0]) + __builtin_bit_cast(wd_u16x2, pk_lim);
#pragma unroll
          for (int k = 1; k < 4; k++) rg = __builtin_elementwise_max(rg, __builtin_bit_cast(wd_u16x2, x[k]) + __builtin_bit_cast(wd_u16x2, pk_lim));
          bad |= __builtin_bit_cast(uint32_t, __builtin_elementwise_max(rg, __builtin_bit_cast(wd_u16x2, 2u * pk_lim))) != 2u * pk_lim;
          uint32_t sq = 0, hd[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x[k], 0x111, 0xf, 0xf, true);  // the lane before: block 5 beside block 6
            const uint32_t v = __builtin_bit_cast(uint32_t, __builtin_bit_cast(wd_u16x2, x[k] & keep_own) + __builtin_bit_cast(wd_u16x2, left & keep_left));
            sq = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(wd_s16x2, v), __builtin_bit_cast(wd_s16x2, v), (int)sq, false);
            x[k] = v;
            hd[k] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(wd_u16x2, v) + __builtin_bit_cast(wd_u16x2, 0x00800080u));  // high digit = high byte of v + 128
          }
          uint8_t *obase = packed + ((int64_t)(y0 + wy) * nsx + sx) * MM_BLK_BYTES;  // (uniform in the wave)
          const unsigned piece = piece0 + (unsigned)wx0 * 16u;
          *reinterpret_cast<uint2 *>(obase + piece) = make_uint2(__builtin_amdgcn_perm(x[1], x[0], 0x06040200u), __builtin_amdgcn_perm(x[3], x[2], 0x06040200u));
          *reinterpret_cast<uint2 *>(obase + MM_CH * 1024 + piece) = make_uint2(__builtin_amdgcn_perm(hd[1], hd[0], 0x07050301u), __builtin_amdgcn_perm(hd[3], hd[2], 0x07050301u));
          if (pk_b == 7) *reinterpret_cast<uint4 *>(obase + MM_QUIRK + wx * 16) = raw;  // block 7 of the first half, raw: its pair sums are re-squared on the VALU
          sq += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sq, 0xB1, 0xf, 0xf, true);   // over the window's eight lanes
          sq += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sq, 0x4E, 0xf, 0xf, true);
          sq += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)sq, 0x141, 0xf, 0xf, true);
          if (u == 0) {  // (the same wave has this run in every plane)
            const uint32_t tot = (c == 0 ? 0u : s_nrm[wy][wx]) + sq;
            if (c < 2) s_nrm[wy][wx] = tot;
            else reinterpret_cast<uint32_t *>(obase + MM_NORM)[wx] = tot;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next run's coefficients land after these reads)
      }
    }
  }
  if constexpr (PACK) { if (bad) atomicOr(flag, 1); }
}

int launch_window_dcts_packed(const void *fb, int w, int h, const void *cur, int ntiles, void *packed, int *flag, hipStream_t stream) {
  const DeviceTables *tab;
  TM_TRY(get_tables(&tab));
  TM_CHECK(w >= 8 && h >= 8, TM_E_INVAL, "frame buffer %dx%d smaller than a tile", w, h);
  const int strips = ((w - 7 + WD_CX - 1) / WD_CX) * ((h - 7 + WD_RY - 1) / WD_RY);
  hipLaunchKernelGGL(k_window_dcts<true>, dim3(std::min(strips, 256 * 4)), dim3(256), 0, stream, (const uint32_t *)fb, w, h, tab->dct_lut_f32[mode_special(TM_PVS_WEIGHTED_DCT)],
                     tab->weights, tab->snake, tab->dct_cos_f64[mode_special(TM_PVS_WEIGHTED_DCT)], (int16_t *)nullptr, knobs().features_plain ? 1 : 0, (const int *)nullptr,
                     (uint8_t *)packed, (const int16_t *)cur, ntiles, flag);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

int launch_window_dcts(const void *fb, int w, int h, void *out, hipStream_t stream, const int *only_if) {
  const DeviceTables *tab;
  TM_TRY(get_tables(&tab));
  TM_CHECK(w >= 8 && h >= 8, TM_E_INVAL, "frame buffer %dx%d smaller than a tile", w, h);
  const int64_t n = (int64_t)(w - 7) * (h - 7);
  if (!knobs().window_dcts_by_tile) {
    const int strips = ((w - 7 + WD_CX - 1) / WD_CX) * ((h - 7 + WD_RY - 1) / WD_RY);
    hipLaunchKernelGGL(k_window_dcts<false>, dim3(std::min(strips, 256 * 4)), dim3(256), 0, stream, (const uint32_t *)fb, w, h, tab->dct_lut_f32[mode_special(TM_PVS_WEIGHTED_DCT)],
                       tab->weights, tab->snake, tab->dct_cos_f64[mode_special(TM_PVS_WEIGHTED_DCT)], (int16_t *)out, knobs().features_plain ? 1 : 0, only_if, (uint8_t *)nullptr,
                       (const int16_t *)nullptr, 0, (int *)nullptr);
    TM_HIP(hipGetLastError());
    return TM_OK;
  }
  TM_CHECK(only_if == nullptr, TM_E_INVAL, "the window-at-a-time kernel has no conditional form");  // (launch_motion_search_fb takes the two-pass form under TM_WINDOW_DCTS_BY_TILE)
  hipLaunchKernelGGL(k_features_i16<2>, dim3(grid_for(n, 4)), dim3(256), 0, stream, (const uint32_t *)fb, nullptr, nullptr, nullptr, w,
                     nullptr, n, 1, 0, tab->dct_lut_f32[mode_special(TM_PVS_WEIGHTED_DCT)], tab->weights, tab->snake, tab->srgb_lut,
                     (int16_t *)out, (int *)nullptr, tab->dct_cos_f64[mode_special(TM_PVS_WEIGHTED_DCT)], knobs().features_plain ? 1 : 0);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

int launch_features_cluster(const void *tiles, int64_t n, int mode, void *out, hipStream_t stream) {
  const DeviceTables *tab;
  TM_TRY(get_tables(&tab));
  TM_CHECK(mode != TM_PVS_WAVELETS && mode >= 0 && mode <= 4, TM_E_INVAL, "bad TPsyVisMode %d", mode);
  if (n <= 0) return TM_OK;
  hipLaunchKernelGGL(k_features_cluster_i32, dim3(grid_for(n, 4)), dim3(256), 0, stream, (const uint32_t *)tiles, n,
                     mode_weighted(mode) ? 1 : 0, tab->dct_lut_f64[mode_special(mode)], tab->weights, tab->snake, tab->srgb_lut,
                     (int32_t *)out, tab->dct_cos_f64[mode_special(mode)], knobs().features_plain ? 1 : 0);
  TM_HIP(hipGetLastError());
  return TM_OK;
}

}  // namespace tmx
