/*
 * tm_oracle.c -- CPU restatement of the TileMotion per-frame tile pipeline.  TEST INFRASTRUCTURE ONLY
 * (see tm_oracle.h).  Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * Citations are file:line in the gligli/tiler reference tree.
 */
#define _GNU_SOURCE
#include "tm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tables (utils.pas:47-109) */

const uint8_t tmo_dithering_map[64] = { /* cDitheringMap, utils.pas:47-56 */
    0, 48, 12, 60, 3, 51, 15, 63, 32, 16, 44, 28, 35, 19, 47, 31, 8,  56, 4,  52, 11, 59, 7,  55, 40, 24, 36, 20, 43, 27, 39, 23,
    2, 50, 14, 62, 1, 49, 13, 61, 34, 18, 46, 30, 33, 17, 45, 29, 10, 58, 6,  54, 9,  57, 5,  53, 42, 26, 38, 22, 41, 25, 37, 21};

const uint8_t tmo_dct_snake[64] = { /* cDCTSnake, utils.pas:59-68 */
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

const double tmo_dct_weights[3][8][8] = { /* cDCTWeights, utils.pas:72-97 */
    {{1.6193873005, 2.2901594831, 2.08509755623, 1.48366094411, 1.00227514334, 0.678296995242, 0.466224900598, 0.3265091542},
     {2.2901594831, 1.94321815382, 2.04793073064, 1.68731108984, 1.2305666963, 0.868920337363, 0.61280991668, 0.436405793551},
     {2.08509755623, 2.04793073064, 1.34329019223, 1.09205635862, 0.875748795257, 0.670882927016, 0.501731932449, 0.372504254596},
     {1.48366094411, 1.68731108984, 1.09205635862, 0.772819797575, 0.605636379554, 0.48309405692, 0.380429446972, 0.295774038565},
     {1.00227514334, 1.2305666963, 0.875748795257, 0.605636379554, 0.448996256676, 0.352889268808, 0.283006984131, 0.226951348204},
     {0.678296995242, 0.868920337363, 0.670882927016, 0.48309405692, 0.352889268808, 0.27032073436, 0.215017739696, 0.17408067321},
     {0.466224900598, 0.61280991668, 0.501731932449, 0.380429446972, 0.283006984131, 0.215017739696, 0.168869545842, 0.136153931001},
     {0.3265091542, 0.436405793551, 0.372504254596, 0.295774038565, 0.226951348204, 0.17408067321, 0.136153931001, 0.109083846276}},
    {{1.91113096927, 2.46074210438, 1.18284184739, 1.14982565193, 1.05017074788, 0.898018824055, 0.74725392039, 0.615105596242},
     {2.46074210438, 1.58529308355, 1.21363250036, 1.38190029285, 1.33100189972, 1.17428548929, 0.996404342439, 0.830890433625},
     {1.18284184739, 1.21363250036, 0.978712413627, 1.02624506078, 1.03145147362, 0.960060382087, 0.849823426169, 0.731221236837},
     {1.14982565193, 1.38190029285, 1.02624506078, 0.861317501629, 0.801821139099, 0.751437590932, 0.685398513368, 0.608694761374},
     {1.05017074788, 1.33100189972, 1.03145147362, 0.801821139099, 0.676555426187, 0.605503172737, 0.55002013668, 0.495804539034},
     {0.898018824055, 1.17428548929, 0.960060382087, 0.751437590932, 0.605503172737, 0.514674450957, 0.454353482512, 0.407050308965},
     {0.74725392039, 0.996404342439, 0.849823426169, 0.685398513368, 0.55002013668, 0.454353482512, 0.389234902883, 0.342353999733},
     {0.615105596242, 0.830890433625, 0.731221236837, 0.608694761374, 0.495804539034, 0.407050308965, 0.342353999733, 0.295530605237}},
    {{2.03871978502, 2.62502345193, 1.26180942886, 1.11019789803, 1.01397751469, 0.867069376285, 0.721500455585, 0.593906509971},
     {2.62502345193, 1.69112867013, 1.17180569821, 1.3342742857, 1.28513006198, 1.13381474809, 0.962064122248, 0.802254508198},
     {1.26180942886, 1.17180569821, 0.944981930573, 0.990876405848, 0.995903384143, 0.926972725286, 0.820534991409, 0.706020324706},
     {1.11019789803, 1.3342742857, 0.990876405848, 0.831632933426, 0.77418706195, 0.725539939514, 0.661776842059, 0.587716619023},
     {1.01397751469, 1.28513006198, 0.995903384143, 0.77418706195, 0.653238524286, 0.584635025748, 0.531064164893, 0.478717061273},
     {0.867069376285, 1.13381474809, 0.926972725286, 0.725539939514, 0.584635025748, 0.496936637883, 0.438694579826, 0.393021669543},
     {0.721500455585, 0.962064122248, 0.820534991409, 0.661776842059, 0.531064164893, 0.438694579826, 0.375820256136, 0.330555063063},
     {0.593906509971, 0.802254508198, 0.706020324706, 0.587716619023, 0.478717061273, 0.393021669543, 0.330555063063, 0.285345396658}}};

static float g_lut_f32[2][4096];
static double g_lut_f64[2][4096];
static double g_inv_lut_f64[4096];
static float g_srgb_lut[256];
static int g_luts_ready = 0;

static float uv_ratio(int v, int u) { /* cDCTUVRatio, utils.pas:100-109 (TFloat = Single) */
  if (v == 0 && u == 0) return 0.5f;
  if (v == 0 || u == 0) return (float)sqrt(0.5);
  return 1.0f;
}

static void init_luts(void) { /* InitLuts, tilingencoder.pas:1703-1726 */
  if (g_luts_ready) return;
  int i = 0;
  for (int v = 0; v < 8; v++)
    for (int u = 0; u < 8; u++)
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
          double r = (double)uv_ratio(v, u);
          double a = cos((x + 0.5) * u * M_PI / 8) * cos((y + 0.5) * v * M_PI / 8) * r;
          double b = cos((x + 0.5) * u * M_PI / 16) * cos((y + 0.5) * v * M_PI / 16) * r;
          g_lut_f64[0][i] = a;
          g_lut_f64[1][i] = b;
          g_lut_f32[0][i] = (float)a;
          g_lut_f32[1][i] = (float)b;
          i++;
        }
  i = 0;
  for (int v = 0; v < 8; v++)
    for (int u = 0; u < 8; u++)
      for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
          g_inv_lut_f64[i] =
              cos((u + 0.5) * x * M_PI / 8) * cos((v + 0.5) * y * M_PI / 8) * (double)uv_ratio(y, x) * 2 / 8 * 2 / 8;
          i++;
        }
  for (int c = 0; c < 256; c++) { /* utils.pas:378-384 */
    float r = (float)(c / 255.0);
    if ((double)r > 0.04045)
      r = (float)pow(((double)r + 0.055) / 1.055, 2.4);
    else
      r = (float)((double)r / 12.92);
    g_srgb_lut[c] = r;
  }
  g_luts_ready = 1;
}

const float *tmo_dct_lut_f32(int special) { init_luts(); return g_lut_f32[special ? 1 : 0]; }
const double *tmo_dct_lut_f64(int special) { init_luts(); return g_lut_f64[special ? 1 : 0]; }
const double *tmo_inv_dct_lut_f64(void) { init_luts(); return g_inv_lut_f64; }
const float *tmo_srgb_lut_f32(void) { init_luts(); return g_srgb_lut; }

/* Pascal Round(): half-to-even to Int64 (default FP rounding mode) */
static inline int64_t pas_round(double x) { return (int64_t)llrint(x); }
static inline int clampi(int64_t v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); }

/* ------------------------------------------------------------------ colour */

uint32_t tmo_swap_rb(uint32_t c) { /* utils.pas:238-241 */
  return ((c & 0xff) << 16) | ((c >> 16) & 0xff) | (c & 0xff00);
}

void tmo_rgb_to_yuv(int r, int g, int b, float *y, float *u, float *v) { /* utils.pas:478-490 */
  float yy = (float)(r * (299.0 / 1000.0) + g * (587.0 / 1000.0) + b * (114.0 / 1000.0));
  float uu = (float)(((double)b - (double)yy) * 0.492);
  float vv = (float)(((double)r - (double)yy) * 0.877);
  *y = yy; *u = uu; *v = vv;
}

int32_t tmo_yuv_to_rgb(float y, float u, float v) { /* utils.pas:492-509 */
  float r = (float)((double)y + (double)v * 1.13983);
  float g = (float)((double)y - (double)u * 0.39465 - (double)v * 0.58060);
  float b = (float)((double)y + (double)u * 2.03211);
  int rr = clampi(pas_round(r), 0, 255), gg = clampi(pas_round(g), 0, 255), bb = clampi(pas_round(b), 0, 255);
  return (bb << 16) | (gg << 8) | rr;
}

/* Deterministic cube root for x in (0, 4): integer-seeded Newton in IEEE double, +,-,*,/ only, so the
 * HIP kernels reproduce it bit for bit.  Build rule standing in for power(x, 1/3) at utils.pas:403-405. */
double tmo_cbrt_det(double x) {
  union { double d; uint64_t u; } c;
  c.d = x;
  c.u = c.u / 3 + 0x2A9F7893782DA1CEull; /* classic exponent/3 seed, ~5% */
  double y = c.d;
  for (int i = 0; i < 6; i++) {
    double y2 = y * y;
    y = y - (y2 * y - x) / (3.0 * y2);
  }
  return y;
}

static void lab_core(int ir, int ig, int ib, int det, float *ol, float *oa, float *ob) { /* utils.pas:374-410 */
  init_luts();
  float r = g_srgb_lut[ir], g = g_srgb_lut[ig], b = g_srgb_lut[ib];
  float x = (float)(((double)r * 0.49000 + (double)g * 0.31000 + (double)b * 0.20000) / 0.17697);
  float y = (float)(((double)r * 0.17697 + (double)g * 0.81240 + (double)b * 0.01063) / 0.17697);
  float z = (float)(((double)r * 0.00000 + (double)g * 0.01000 + (double)b * 0.99000) / 0.17697);
  x = (float)((double)x * (1 / (96.6797 / 100)));
  y = (float)((double)y * (1 / (100.000 / 100)));
  z = (float)((double)z * (1 / (82.5188 / 100)));
#define LABF(t)                                                                  \
  if ((double)(t) > 0.008856)                                                    \
    (t) = (float)(det ? tmo_cbrt_det((double)(t)) : pow((double)(t), 1.0 / 3));  \
  else                                                                           \
    (t) = (float)((7.787 * (double)(t)) + 16.0 / 116)
  LABF(x);
  LABF(y);
  LABF(z);
#undef LABF
  *ol = (float)((116 * (double)y) - 16);
  *oa = (float)(500 * (double)(float)(x - y));
  *ob = (float)(200 * (double)(float)(y - z));
}

/* The form the HIP kernels execute (tm_device.h): no division.  x^(-1/3) by four Newton steps in multiplications and fma, the root as
 * x a a with one multiplicative correction; the / 0.17697 of utils.pas:391-393 as a reciprocal product refined by two fma.  It is NOT
 * a definition of its own: tmo_lab_domain_check proves it equal to tmo_rgb_to_lab_det on every one of the 2^24 colours. */
static double cbrt_mul(double x) {
  union { double d; uint64_t u; } c;
  c.d = x;
  c.u = 0x553EF0FF289DD796ull - c.u / 3; /* exponent / -3 seed */
  double a = c.d;
  for (int i = 0; i < 4; i++) {
    double e = fma(-x, a * a * a, 1.0);
    a = fma(a * e, 1.0 / 3, a);
  }
  double y = x * a * a;
  return fma(fma(-y * y, y, x) * (a * a), 1.0 / 3, y);
}

static double div_k(double n) {
  const double k = 1.0 / 0.17697;
  double q = n * k;
  return fma(fma(-q, 0.17697, n), k, q);
}

void tmo_rgb_to_lab_fast(int ir, int ig, int ib, float *ol, float *oa, float *ob) {
  init_luts();
  float r = g_srgb_lut[ir], g = g_srgb_lut[ig], b = g_srgb_lut[ib];
  float v[3];
  v[0] = (float)div_k((double)r * 0.49000 + (double)g * 0.31000 + (double)b * 0.20000);
  v[1] = (float)div_k((double)r * 0.17697 + (double)g * 0.81240 + (double)b * 0.01063);
  v[2] = (float)div_k((double)r * 0.00000 + (double)g * 0.01000 + (double)b * 0.99000);
  v[0] = (float)((double)v[0] * (1 / (96.6797 / 100)));
  v[1] = (float)((double)v[1] * (1 / (100.000 / 100)));
  v[2] = (float)((double)v[2] * (1 / (82.5188 / 100)));
  for (int i = 0; i < 3; i++)
    v[i] = (double)v[i] > 0.008856 ? (float)cbrt_mul((double)v[i]) : (float)((7.787 * (double)v[i]) + 16.0 / 116);
  *ol = (float)((116 * (double)v[1]) - 16);
  *oa = (float)(500 * (double)(float)(v[0] - v[1]));
  *ob = (float)(200 * (double)(float)(v[1] - v[2]));
}

/* every 24-bit colour: out[0] = colours where the deterministic form differs from libm pow (the reference's power(), utils.pas:403),
 * out[1] = colours where the kernels' division-free form differs from the deterministic form */
void tmo_lab_domain_check(int64_t *out) {
  out[0] = out[1] = 0;
  for (int c = 0; c < (1 << 24); c++) {
    float p[3], d[3], f[3];
    lab_core(c & 255, (c >> 8) & 255, c >> 16, 0, &p[0], &p[1], &p[2]);
    lab_core(c & 255, (c >> 8) & 255, c >> 16, 1, &d[0], &d[1], &d[2]);
    tmo_rgb_to_lab_fast(c & 255, (c >> 8) & 255, c >> 16, &f[0], &f[1], &f[2]);
    if (memcmp(p, d, 12)) out[0]++;
    if (memcmp(f, d, 12)) out[1]++;
  }
}

/* RGBToLAB of n colours 0x00RRGGBB -> out[n][3], det = 1: through the deterministic root (for whole-domain checks of the device's form) */
void tmo_rgb_to_lab_array(const uint32_t *rgb, int64_t n, int det, float *out) {
  for (int64_t i = 0; i < n; i++) lab_core((int)((rgb[i] >> 16) & 255), (int)((rgb[i] >> 8) & 255), (int)(rgb[i] & 255), det, &out[i * 3], &out[i * 3 + 1], &out[i * 3 + 2]);
}

void tmo_rgb_to_lab(int r, int g, int b, float *ol, float *oa, float *ob) { lab_core(r, g, b, 0, ol, oa, ob); }
void tmo_rgb_to_lab_det(int r, int g, int b, float *ol, float *oa, float *ob) { lab_core(r, g, b, 1, ol, oa, ob); }

int32_t tmo_lab_to_rgb(float ll, float aa, float bb) { /* utils.pas:422-466 */
  float y = (float)(((double)ll + 16) / 116);
  float x = (float)((double)aa / 500 + (double)y);
  float z = (float)((double)y - (double)bb / 200);
#define INVF(t)                                                        \
  {                                                                    \
    double t3 = (double)(t) * (double)(t) * (double)(t);               \
    if (t3 > 0.008856)                                                 \
      (t) = (float)t3;                                                 \
    else                                                               \
      (t) = (float)(((double)(t) - 16.0 / 116) / 7.787);               \
  }
  INVF(y);
  INVF(x);
  INVF(z);
#undef INVF
  x = (float)(96.6797 / 100 * (double)x);
  y = (float)(100.000 / 100 * (double)y);
  z = (float)(82.5188 / 100 * (double)z);
  float r = (float)((double)x * 0.41847 + (double)y * (-0.15866) + (double)z * (-0.082835));
  float g = (float)((double)x * (-0.091169) + (double)y * 0.25243 + (double)z * 0.015708);
  float b = (float)((double)x * 0.00092090 + (double)y * (-0.0025498) + (double)z * 0.17860);
#define GAM(t)                                                              \
  if ((double)(t) > 0.0031308)                                              \
    (t) = (float)(1.055 * pow((double)(t), 1 / 2.4) - 0.055);               \
  else                                                                      \
    (t) = (float)(12.92 * (double)(t))
  GAM(r);
  GAM(g);
  GAM(b);
#undef GAM
  int rr = clampi(pas_round((double)r * 255.0), 0, 255);
  int gg = clampi(pas_round((double)g * 255.0), 0, 255);
  int bq = clampi(pas_round((double)b * 255.0), 0, 255);
  return (bq << 16) | (gg << 8) | rr;
}

/* Windows MulDiv: (a*b)/c rounded half away from zero */
static int muldiv(int a, int b, int c) {
  int64_t p = (int64_t)a * b;
  int64_t q = (p >= 0 ? p : -p), cc = c >= 0 ? c : -c;
  int64_t res = (q + cc / 2) / cc;
  return (int)(((p < 0) != (c < 0)) ? -res : res);
}

void tmo_rgb_to_hsv(uint32_t col, uint8_t *h, uint8_t *s, uint8_t *v) { /* utils.pas:278-325 */
  int rr = col & 0xff, gg = (col >> 8) & 0xff, bb = (col >> 16) & 0xff;
  int mx = rr, mn = rr;
  if (mx < gg) mx = gg;
  if (mx < bb) mx = bb;
  if (mn > gg) mn = gg;
  if (mn > bb) mn = bb;
  int hh = 0, ss = 0, ll = mx;
  if (ll != mn) {
    int delta = ll - mn;
    ss = muldiv(delta, 255, ll);
    if (rr == ll)
      hh = muldiv(42, gg - bb, delta);
    else if (gg == ll)
      hh = muldiv(42, bb - rr, delta) + 84;
    else if (bb == ll)
      hh = muldiv(42, rr - gg, delta) + 168;
    hh = hh % 252; /* Pascal mod: sign follows dividend, like C */
  }
  *h = (uint8_t)(hh & 0xff);
  *s = (uint8_t)(ss & 0xff);
  *v = (uint8_t)(ll & 0xff);
}

/* ------------------------------------------------------------------ A1-A3 load side */

void tmo_load_from_image(const uint32_t *img, int img_w, int img_h, int tm_w, int tm_h, uint32_t *tiles) {
  /* TFrame.LoadFromImage, tilingencoder.pas:1293-1320.  Tiles not covered by the image stay zero. */
  int sw = tm_w * 8, sh = tm_h * 8;
  memset(tiles, 0, (size_t)tm_w * tm_h * 64 * sizeof(uint32_t));
  for (int j = 0; j < img_h; j++)
    for (int i = 0; i < img_w; i++) {
      uint32_t col = img[(size_t)j * img_w + i];
      if (j < sh && i < sw) {
        int ti = tm_w * (j >> 3) + (i >> 3);
        tiles[(size_t)ti * 64 + (j & 7) * 8 + (i & 7)] = tmo_swap_rb(col);
      }
    }
}

void tmo_inter_frame_data(const uint32_t *tiles, int ntiles, float *out3) { /* tilingencoder.pas:1329-1367 */
  for (int t = 0; t < ntiles; t++) {
    float sl = 0, sa = 0, sb = 0;
    for (int p = 0; p < 64; p++) {
      uint32_t c = tiles[(size_t)t * 64 + p];
      float l, a, b;
      tmo_rgb_to_lab_det(c & 0xff, (c >> 8) & 0xff, (c >> 16) & 0xff, &l, &a, &b);
      sl += l;
      sa += a;
      sb += b;
    }
    const float inv = 1.0f / 64;
    out3[t * 3 + 0] = sl * inv;
    out3[t * 3 + 1] = sa * inv;
    out3[t * 3 + 2] = sb * inv;
  }
}

float tmo_pearson(const float *x, const float *y, int n) { /* tilingencoder.pas:2201-2228; Math.mean sums in double */
  double sx = 0, sy = 0;
  for (int i = 0; i < n; i++) { sx += x[i]; sy += y[i]; }
  float mx = (float)(sx / n), my = (float)(sy / n);
  float num = 0, denx = 0, deny = 0;
  for (int i = 0; i < n; i++) {
    float dx = x[i] - mx, dy = y[i] - my;
    num += dx * dy;
    denx += dx * dx;
    deny += dy * dy;
  }
  denx = sqrtf(denx);
  deny = sqrtf(deny);
  float den = denx * deny;
  return den != 0.0f ? num / den : 1.0f;
}

static int zone_sum(const uint32_t *t, int x, int y) { /* GetTileZoneSum, tilingencoder.pas:4842-4863 */
  int s = 0;
  for (int j = y; j < y + 4; j++)
    for (int i = x; i < x + 4; i++) {
      uint32_t c = t[j * 8 + i];
      s += (int)(c & 0xff) * 299 + (int)((c >> 8) & 0xff) * 587 + (int)((c >> 16) & 0xff) * 114;
    }
  return s;
}

void tmo_mirror_heuristics(const uint32_t *tile, int *hm, int *vm) { /* tilingencoder.pas:4865-4878 */
  int q00 = zone_sum(tile, 0, 0), q01 = zone_sum(tile, 4, 0), q10 = zone_sum(tile, 0, 4), q11 = zone_sum(tile, 4, 4);
  *hm = (q00 + q10) < (q01 + q11);
  *vm = (q00 + q01) < (q10 + q11);
}

#define MIRROR_IMPL(NAME, T, H)                                    \
  void NAME(T *t) {                                                \
    for (int j = 0; j < (H ? 8 : 4); j++)                          \
      for (int i = 0; i < (H ? 4 : 8); i++) {                      \
        int a = j * 8 + i, b = H ? j * 8 + (7 - i) : (7 - j) * 8 + i; \
        T v = t[a];                                                \
        t[a] = t[b];                                               \
        t[b] = v;                                                  \
      }                                                            \
  }
MIRROR_IMPL(tmo_hmirror_u32, uint32_t, 1) /* HMirrorTile, tilingencoder.pas:3285-3311 */
MIRROR_IMPL(tmo_vmirror_u32, uint32_t, 0) /* VMirrorTile, tilingencoder.pas:3257-3283 */
MIRROR_IMPL(tmo_hmirror_u8, uint8_t, 1)
MIRROR_IMPL(tmo_vmirror_u8, uint8_t, 0)

void tmo_canonicalise_tiles(uint32_t *tiles, int ntiles, uint8_t *flags) { /* tilingencoder.pas:1393-1411 */
  for (int t = 0; t < ntiles; t++) {
    int h, v;
    uint32_t *p = tiles + (size_t)t * 64;
    tmo_mirror_heuristics(p, &h, &v);
    if (h) tmo_hmirror_u32(p);
    if (v) tmo_vmirror_u32(p);
    flags[t] = (uint8_t)(h | (v << 1));
  }
}

int tmo_find_keyframes(const float *correl, int nframes, double fps, double max_s, double min_s, double lo, uint8_t *is_kf) {
  /* FindKeyFrames automatic mode, tilingencoder.pas:3373-3411 */
  int64_t last = INT32_MIN;
  int n = 0;
  for (int f = 0; f < nframes; f++) {
    int kf = 0;
    if (f == 0) kf = 1;
    if (!kf && (double)correl[f] < lo) kf = 1;
    if (!kf && (double)(f - last) >= max_s * fps) kf = 1;
    if ((double)(f - last) < min_s * fps) kf = 0;
    is_kf[f] = (uint8_t)kf;
    if (kf) { last = f; n++; }
  }
  return n;
}

/* ------------------------------------------------------------------ A4-A6 features */

static void to_cpn(uint32_t col, int use_lab, float *cpn, int pos) { /* ToCpn, tilingencoder.pas:3051-3070 */
  int r = col & 0xff, g = (col >> 8) & 0xff, b = (col >> 16) & 0xff;
  float yy, uu, vv;
  if (use_lab)
    tmo_rgb_to_lab_det(r, g, b, &yy, &uu, &vv);
  else
    tmo_rgb_to_yuv(r, g, b, &yy, &uu, &vv);
  cpn[pos] = yy;
  cpn[64 + pos] = uu;
  cpn[128 + pos] = vv;
}

void tmo_cpn_from_rgb(const uint32_t *rgb, int use_lab, int hm, int vm, float cpn[192]) { /* tilingencoder.pas:3090-3099 */
  for (int y = 0; y < 8; y++)
    for (int x = 0; x < 8; x++) {
      int xx = hm ? 7 - x : x, yy = vm ? 7 - y : y;
      to_cpn(rgb[yy * 8 + xx], use_lab, cpn, y * 8 + x);
    }
}

void tmo_cpn_from_pal(const uint8_t *pp, const int32_t *pal, int use_lab, int hm, int vm, float cpn[192]) { /* 3077-3086 */
  for (int y = 0; y < 8; y++)
    for (int x = 0; x < 8; x++) {
      int xx = hm ? 7 - x : x, yy = vm ? 7 - y : y;
      to_cpn((uint32_t)pal[pp[yy * 8 + xx]], use_lab, cpn, y * 8 + x);
    }
}

static double dct_inner_asm(const float *c, const float *l) { /* DCTInner_asm, utils.pas:874-1035 */
  double acc0 = 0, acc1 = 0;
  for (int k = 0; k < 64; k += 16) {
    float p[16];
    for (int j = 0; j < 16; j++) p[j] = c[k + j] * l[k + j];   /* mulps */
    float s[4], t[4];
    for (int j = 0; j < 4; j++) { s[j] = p[j] + p[j + 4]; t[j] = p[j + 8] + p[j + 12]; } /* addps */
    double a0 = (double)s[0] + (double)t[0], a1 = (double)s[1] + (double)t[1];           /* addpd xmm2,xmm4 */
    double b0 = (double)s[2] + (double)t[2], b1 = (double)s[3] + (double)t[3];           /* addpd xmm6,xmm8 */
    a0 = a0 + b0;                                                                       /* addpd xmm2,xmm6 */
    a1 = a1 + b1;
    acc0 = acc0 + a0;                                                                   /* addpd xmm0,xmm2 */
    acc1 = acc1 + a1;
  }
  return acc0 + acc1; /* haddpd */
}

static int mode_special(int mode) { return mode == TMO_PVS_SPE_DCT || mode == TMO_PVS_WEIGHTED_SPE_DCT; }
static int mode_weighted(int mode) { return mode == TMO_PVS_WEIGHTED_DCT || mode == TMO_PVS_WEIGHTED_SPE_DCT; }

void tmo_features_i16(const float cpn[192], int mode, int16_t out[192]) { /* tilingencoder.pas:3103-3131 */
  const float *lut = tmo_dct_lut_f32(mode_special(mode));
  for (int c = 0; c < 3; c++)
    for (int v = 0; v < 8; v++)
      for (int u = 0; u < 8; u++) {
        double z = dct_inner_asm(cpn + c * 64, lut + (v * 8 + u) * 64);
        if (mode_weighted(mode)) z *= tmo_dct_weights[c][v][u];
        out[c * 64 + tmo_dct_snake[v * 8 + u]] = (int16_t)pas_round(z);
      }
}

static double dct_inner_f64(const double *c, const double *l) { /* DCTInner<PDouble>, utils.pas:782-872 */
  double r = 0;
  for (int k = 0; k < 64; k++) r += c[k] * l[k];
  return r;
}

void tmo_features_f64(const float cpn[192], int mode, double out[192]) { /* tilingencoder.pas:3133-3182 (DCT modes) */
  const double *lut = tmo_dct_lut_f64(mode_special(mode));
  double cd[192], loc[192];
  for (int i = 0; i < 192; i++) cd[i] = cpn[i];
  for (int c = 0; c < 3; c++)
    for (int v = 0; v < 8; v++)
      for (int u = 0; u < 8; u++) {
        double z = dct_inner_f64(cd + c * 64, lut + (v * 8 + u) * 64);
        if (mode_weighted(mode)) z *= tmo_dct_weights[c][v][u];
        loc[c * 64 + v * 8 + u] = z;
      }
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < 64; i++) out[tmo_dct_snake[i] + c * 64] = loc[i + c * 64];
}

void tmo_inv_features_f64(const double dct[192], int mode, int use_lab, uint32_t rgb_out[64]) { /* 3184-3255 */
  const double *ilut = tmo_inv_dct_lut_f64();
  double loc[192], cpn[192];
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < 64; i++) {
      double d = dct[tmo_dct_snake[i] + c * 64];
      loc[c * 64 + i] = mode_weighted(mode) ? d / tmo_dct_weights[c][i >> 3][i & 7] : d;
    }
  for (int c = 0; c < 3; c++)
    for (int p = 0; p < 64; p++) cpn[c * 64 + p] = dct_inner_f64(loc + c * 64, ilut + p * 64);
  for (int p = 0; p < 64; p++) {
    float yy = (float)cpn[p], uu = (float)cpn[64 + p], vv = (float)cpn[128 + p];
    rgb_out[p] = (uint32_t)(use_lab ? tmo_lab_to_rgb(yy, uu, vv) : tmo_yuv_to_rgb(yy, uu, vv));
  }
}

void tmo_tiles_features_i16(const uint32_t *tiles, int n, const uint8_t *mf, int mode, int use_lab, int16_t *out) {
  for (int t = 0; t < n; t++) {
    float cpn[192];
    int f = mf ? mf[t] : 0;
    tmo_cpn_from_rgb(tiles + (size_t)t * 64, use_lab, f & 1, (f >> 1) & 1, cpn);
    tmo_features_i16(cpn, mode, out + (size_t)t * 192);
  }
}

void tmo_paltiles_features_i16(const uint8_t *pp, const int32_t *pal_idx, int n, const int32_t *palettes, int pal_size,
                               int mode, int16_t *out) { /* PrepareReconstruct.DoPsyV, tilingencoder.pas:4570-4583 */
  for (int t = 0; t < n; t++) {
    float cpn[192];
    tmo_cpn_from_pal(pp + (size_t)t * 64, palettes + (size_t)pal_idx[t] * pal_size, 0, 0, 0, cpn);
    tmo_features_i16(cpn, mode, out + (size_t)t * 192);
  }
}

void tmo_tiles_features_cluster_i32(const uint32_t *tiles, int n, int mode, int32_t *out) {
  /* DoPalettization feature (tilingencoder.pas:4126,4160): A6 with UseLAB, then the build's Round() to int32 */
  for (int t = 0; t < n; t++) {
    float cpn[192];
    double f[192];
    tmo_cpn_from_rgb(tiles + (size_t)t * 64, 1, 0, 0, cpn);
    tmo_features_f64(cpn, mode, f);
    for (int i = 0; i < 192; i++) out[(size_t)t * 192 + i] = (int32_t)pas_round(f[i]);
  }
}

/* ------------------------------------------------------------------ A15 distances */

uint32_t tmo_ssd_i16(const int16_t *a, const int16_t *b) { /* CompareEuclideanDCTPtr, utils.pas:541-557 */
  uint32_t r = 0;
  for (int i = 0; i < 192; i++) {
    int32_t d = (int32_t)a[i] - (int32_t)b[i];
    r += (uint32_t)(d * d);
  }
  return r;
}

static inline int16_t sat16(int32_t v) { return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }

uint32_t tmo_ssd_i16_sse_quirk(const int16_t *a, const int16_t *b) {
  /* CompareEuclideanDCTPtr_asm as written (utils.pas:559-725): per 96-element half, 12 blocks of 8 int16 in
   * xmm1..xmm12.  Block 5 is loaded into xmm6 then overwritten by block 6 (591-592); xmm6 gets psubsw of b block 5
   * AND b block 6 (604-605); xmm7 is never loaded yet is squared (618) and summed (628).  After half 1, xmm7 holds
   * pmaddwd(xmm7_in) + block-7 squares, which half 2 re-squares as int16 pairs (678).  Entry xmm7 taken as 0
   * (Win64 ABI leaves it undefined). */
  uint32_t total = 0;
  uint32_t x7[4] = {0, 0, 0, 0};
  for (int half = 0; half < 2; half++) {
    const int16_t *pa = a + half * 96, *pb = b + half * 96;
    for (int k = 0; k < 4; k++) { /* pmaddwd xmm7,xmm7 */
      int32_t lo = (int16_t)(x7[k] & 0xffff), hi = (int16_t)(x7[k] >> 16);
      x7[k] = (uint32_t)(lo * lo) + (uint32_t)(hi * hi);
    }
    for (int blk = 0; blk < 12; blk++) {
      if (blk == 5) continue; /* dropped */
      for (int j = 0; j < 8; j++) {
        int16_t d;
        if (blk == 6) {
          d = sat16((int32_t)pa[48 + j] - pb[40 + j]); /* psubsw xmm6,[rdx+$50] */
          d = sat16((int32_t)d - pb[48 + j]);          /* psubsw xmm6,[rdx+$60] */
        } else {
          d = sat16((int32_t)pa[blk * 8 + j] - pb[blk * 8 + j]);
        }
        uint32_t sq = (uint32_t)((int32_t)d * (int32_t)d);
        if (blk == 7) x7[j >> 1] += sq; /* paddd xmm7,xmm8 */
        else total += sq;
      }
    }
    for (int k = 0; k < 4; k++) total += x7[k]; /* paddd xmm5,xmm7 ... phaddd */
  }
  return total;
}

float tmo_euclidean_to_psnr(uint32_t e) { /* utils.pas:1074-1078 */
  float r = (float)((double)e * (1.0 / 192));
  float m = r > 0.5f ? r : 0.5f;
  return (float)(10 * log10(255 * 255 / (double)m));
}

/* ------------------------------------------------------------------ A13/A14 KNN */

void tmo_knn1(const int16_t *q, int64_t nq, const int16_t *db, int64_t nt, int32_t *idx, uint32_t *err) {
  /* exact nearest neighbour = what ann_kdtree_search(eps=0) returns (tilingencoder.pas:1547); build's tie rule: lowest index */
  for (int64_t i = 0; i < nq; i++) {
    uint32_t best = UINT32_MAX;
    int32_t bi = -1;
    for (int64_t t = 0; t < nt; t++) {
      uint32_t d = tmo_ssd_i16(q + i * 192, db + t * 192);
      if (d < best || bi < 0) { best = d; bi = (int32_t)t; }
    }
    idx[i] = bi;
    err[i] = best;
  }
}

void tmo_knnk(const int16_t *q, int64_t nq, const int16_t *db, int64_t nt, int k, int32_t *idx, uint32_t *err) {
  /* ann_kdtree_search_multi(k=64, eps=0) (tilingencoder.pas:1563); build's order: (err asc, idx asc); short lists pad -1 */
  for (int64_t i = 0; i < nq; i++) {
    int32_t *oi = idx + i * k;
    uint32_t *oe = err + i * k;
    int cnt = 0;
    for (int64_t t = 0; t < nt; t++) {
      uint32_t d = tmo_ssd_i16(q + i * 192, db + t * 192);
      if (cnt == k && d >= oe[k - 1]) continue;
      int p = cnt < k ? cnt : k - 1;
      while (p > 0 && oe[p - 1] > d) { oe[p] = oe[p - 1]; oi[p] = oi[p - 1]; p--; }
      oe[p] = d;
      oi[p] = (int32_t)t;
      if (cnt < k) cnt++;
    }
    for (int p = cnt; p < k; p++) { oi[p] = -1; oe[p] = UINT32_MAX; }
  }
}

/* ------------------------------------------------------------------ kd-tree (what the reference searches with)
 * ann_kdtree_short_create(rows, T, 192, bucket 32, ANN_KD_STD) + ann_kdtree_short_search(eps 0), tilingencoder.pas:4600, 1547.
 * ANN.dll's source is not in the tree, so this is the published algorithm of Mount & Arya's ANN
 * ("standard" kd split rule: cut the dimension of maximum spread at the median; standard search with incremental
 * box-distance updates), not a restatement of a reference file.  With eps = 0 the answer is the exact nearest neighbour, so
 * it must equal tmo_knn1 (build's tie rule: lowest index).  Used as the CPU baseline the reference actually runs (bench.py). */
typedef struct { int32_t dim; int32_t cut; int32_t left, right; int64_t lo, hi; } tmo_kdnode;
struct tmo_kdtree { const int16_t *db; int64_t n; int32_t *perm; tmo_kdnode *nodes; int64_t nnodes, cap; int bucket; };

static int64_t kd_build(tmo_kdtree *t, int64_t lo, int64_t hi) {
  if (t->nnodes == t->cap) { t->cap = t->cap ? t->cap * 2 : 1024; t->nodes = (tmo_kdnode *)realloc(t->nodes, (size_t)t->cap * sizeof(tmo_kdnode)); }
  const int64_t me = t->nnodes++;
  tmo_kdnode nd = {-1, 0, -1, -1, lo, hi};
  if (hi - lo > t->bucket) {
    int best_dim = 0, best_spread = -1;
    for (int d = 0; d < 192; d++) {
      int mn = 32767, mx = -32768;
      for (int64_t i = lo; i < hi; i++) { const int v = t->db[(int64_t)t->perm[i] * 192 + d]; if (v < mn) mn = v; if (v > mx) mx = v; }
      if (mx - mn > best_spread) { best_spread = mx - mn; best_dim = d; }
    }
    if (best_spread > 0) {
      /* median by quickselect on the chosen coordinate (ties between equal coordinates: by index, so the tree is deterministic) */
      const int64_t mid = lo + (hi - lo) / 2;
      int64_t a = lo, b = hi - 1;
      while (a < b) {
        const int32_t pv = t->perm[a + (b - a) / 2];
        const int pk = t->db[(int64_t)pv * 192 + best_dim];
        int64_t i = a, j = b;
        while (i <= j) {
          for (;;) { const int32_t x = t->perm[i]; const int xk = t->db[(int64_t)x * 192 + best_dim]; if (xk < pk || (xk == pk && x < pv)) i++; else break; }
          for (;;) { const int32_t x = t->perm[j]; const int xk = t->db[(int64_t)x * 192 + best_dim]; if (xk > pk || (xk == pk && x > pv)) j--; else break; }
          if (i <= j) { const int32_t tmp = t->perm[i]; t->perm[i] = t->perm[j]; t->perm[j] = tmp; i++; j--; }
        }
        if (mid <= j) b = j; else if (mid >= i) a = i; else break;
      }
      nd.dim = best_dim;
      nd.cut = t->db[(int64_t)t->perm[mid] * 192 + best_dim];
      t->nodes[me] = nd;
      const int64_t l = kd_build(t, lo, mid), r = kd_build(t, mid, hi);
      t->nodes[me].left = (int32_t)l;
      t->nodes[me].right = (int32_t)r;
      return me;
    }
  }
  t->nodes[me] = nd;
  return me;
}

tmo_kdtree *tmo_kdtree_build(const int16_t *db, int64_t n, int bucket) {
  tmo_kdtree *t = (tmo_kdtree *)calloc(1, sizeof(tmo_kdtree));
  t->db = db; t->n = n; t->bucket = bucket > 0 ? bucket : 32;
  t->perm = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  for (int64_t i = 0; i < n; i++) t->perm[i] = (int32_t)i;
  if (n > 0) kd_build(t, 0, n);
  return t;
}
void tmo_kdtree_free(tmo_kdtree *t) { if (t) { free(t->perm); free(t->nodes); free(t); } }

typedef struct { const tmo_kdtree *t; const int16_t *q; int64_t off[192]; uint32_t best; int32_t bi; int64_t visited; } kd_search;

static void kd_visit(kd_search *s, int64_t node, int64_t rd /* squared distance from q to the node's box */) {
  const tmo_kdnode *nd = &s->t->nodes[node];
  if (nd->dim < 0) {
    for (int64_t i = nd->lo; i < nd->hi; i++) {
      const int32_t r = s->t->perm[i];
      const uint32_t d = tmo_ssd_i16(s->q, s->t->db + (int64_t)r * 192);
      s->visited++;
      if (s->bi < 0 || d < s->best || (d == s->best && r < s->bi)) { s->best = d; s->bi = r; }
    }
    return;
  }
  const int64_t diff = (int64_t)s->q[nd->dim] - nd->cut;  /* left holds coordinates <= cut (by index among equals), right >= cut */
  const int64_t near = diff < 0 ? nd->left : nd->right, far = diff < 0 ? nd->right : nd->left;
  kd_visit(s, near, rd);
  const int64_t old = s->off[nd->dim];
  const int64_t nrd = rd - old * old + diff * diff;  /* incremental box distance (Arya & Mount) */
  if (s->bi < 0 || nrd <= (int64_t)s->best) {        /* <=: an equally near row with a lower index may sit on the far side */
    s->off[nd->dim] = diff;
    kd_visit(s, far, nrd);
    s->off[nd->dim] = old;
  }
}

int64_t tmo_kdtree_search1(const tmo_kdtree *t, const int16_t *q, int64_t nq, int32_t *idx, uint32_t *err) {
  int64_t visited = 0;
  for (int64_t i = 0; i < nq; i++) {
    kd_search s;
    memset(&s, 0, sizeof(s));
    s.t = t; s.q = q + i * 192; s.bi = -1; s.best = UINT32_MAX;
    if (t->n > 0) kd_visit(&s, 0, 0);
    idx[i] = s.bi;
    err[i] = s.best;
    visited += s.visited;
  }
  return visited;  /* rows whose distance was computed: how far the tree is from a brute force */
}

/* ------------------------------------------------------------------ QuickSort (extern.pas:370-418) */

void tmo_quicksort(void *data, int64_t first, int64_t last, int isz, tmo_cmp_fn cmp, void *user) {
  if (last <= first) return;
  uint8_t *pd = (uint8_t *)data;
  uint8_t tmp[4096];
  int64_t i, j, p;
  do {
    i = first;
    j = last;
    p = (first + last) >> 1;
    do {
      while (cmp(pd + i * isz, pd + p * isz, user) < 0) i++;
      while (cmp(pd + j * isz, pd + p * isz, user) > 0) j--;
      if (i <= j) {
        memcpy(tmp, pd + j * isz, (size_t)isz);
        memcpy(pd + j * isz, pd + i * isz, (size_t)isz);
        memcpy(pd + i * isz, tmp, (size_t)isz);
        if (p == i)
          p = j;
        else if (p == j)
          p = i;
        i++;
        j--;
      }
    } while (i <= j);
    if (first < j) tmo_quicksort(data, first, j, isz, cmp, user);
    first = i;
  } while (i < last);
}

/* ------------------------------------------------------------------ A12 dithering */

void tmo_prepare_plan(tmo_plan *plan, const int32_t *pal, int pal_size, int y2_mixed) { /* tilingencoder.pas:2268-2301 */
  memset(plan, 0, sizeof(*plan));
  plan->y2_mixed_colors = y2_mixed;
  int cnt = 0;
  for (int i = 0; i < pal_size; i++) {
    if (pal[i] == TMO_NULL_COLOR) continue;
    int r = pal[i] & 0xff, g = (pal[i] >> 8) & 0xff, b = (pal[i] >> 16) & 0xff;
    plan->luma[cnt] = r * 299 + g * 587 + b * 114;
    plan->y2[cnt][0] = r;
    plan->y2[cnt][1] = g;
    plan->y2[cnt][2] = b;
    plan->y2[cnt][3] = plan->luma[cnt] / 1000;
    plan->remap[cnt] = (uint8_t)i;
    cnt++;
  }
  plan->count = cnt;
}

int64_t tmo_color_compare(int64_t r1, int64_t g1, int64_t b1, int64_t r2, int64_t g2, int64_t b2) { /* 2323-2337 */
  int64_t luma1 = r1 * 299 + g1 * 587 + b1 * 114;
  int64_t luma2 = r2 * 299 + g2 * 587 + b2 * 114;
  int64_t ld = (luma1 - luma2) / 1000; /* div: toward zero */
  int64_t dr = r1 - r2, dg = g1 - g2, db = b1 - b2;
  return dr * dr * 13 + dg * dg * 13 + db * db * 13 + ((ld * ld) << 5);
}

static int cmp_luma(const void *a, const void *b, void *user) { /* PlanCompareLuma, tilingencoder.pas:2310-2321 */
  const int32_t *l = (const int32_t *)user;
  int32_t x = l[*(const uint8_t *)a], y = l[*(const uint8_t *)b];
  return (x > y) - (x < y);
}

void tmo_mixing_plan_tk(const tmo_plan *plan, uint32_t col, uint8_t list[64]) { /* tilingencoder.pas:2565-2612 */
  int64_t s[3] = {col & 0xff, (col >> 8) & 0xff, (col >> 16) & 0xff}, e[3] = {0, 0, 0}, t[3];
  for (int c = 0; c < 64; c++) {
    for (int k = 0; k < 3; k++) t[k] = s[k] + (e[k] * 9) / 100;
    int64_t least = INT64_MAX;
    int chosen = c % plan->count;
    for (int i = 0; i < plan->count; i++) {
      int64_t pen = tmo_color_compare(t[0], t[1], t[2], plan->y2[i][0], plan->y2[i][1], plan->y2[i][2]);
      if (pen < least) { least = pen; chosen = i; }
    }
    list[c] = (uint8_t)chosen;
    for (int k = 0; k < 3; k++) e[k] += s[k] - plan->y2[chosen][k];
  }
  tmo_quicksort(list, 0, 63, 1, cmp_luma, (void *)plan->luma);
}

int tmo_mixing_plan_yliluoma(const tmo_plan *plan, uint32_t col, uint8_t list[256]) {
  /* DeviseBestMixingPlanYliluoma, live SSE4.1 path (ASM_DBMP), tilingencoder.pas:2339-2563:
   * averages via the reciprocal table FVecInv[4t..4t+3] = 65536 div t (1698-1699), 32-bit lanes incl. luma,
   * penalty = 13(dR^2+dG^2+dB^2)+32 dL^2 as a 32-bit sum compared unsigned, first strict minimum wins. */
  int32_t tgt[4] = {(int32_t)(col & 0xff), (int32_t)((col >> 8) & 0xff), (int32_t)((col >> 16) & 0xff), 0};
  tgt[3] = (int32_t)((uint32_t)(tgt[0] * 299 + tgt[1] * 587 + tgt[2] * 114) / 1000u);
  const uint32_t w[4] = {13, 13, 13, 32};
  int plan_count = 0;
  int32_t so_far[4] = {0, 0, 0, 0};
  while (plan_count < plan->y2_mixed_colors) {
    int max_test = plan_count == 0 ? 1 : plan_count;
    uint64_t best = ((uint64_t)1 << 63) - 1;
    int chosen = 0, chosen_amount = 1;
    for (int idx = 0; idx < plan->count; idx++) {
      uint32_t sum[4], add[4];
      for (int k = 0; k < 4; k++) { sum[k] = (uint32_t)so_far[k]; add[k] = (uint32_t)plan->y2[idx][k]; }
      for (int t = plan_count + 1; t <= plan_count + max_test; t++) {
        uint32_t inv = 65536u / (uint32_t)t, pen = 0;
        for (int k = 0; k < 4; k++) {
          sum[k] += add[k];
          add[k] += 1;
          uint32_t avg = (sum[k] * inv) >> 16;
          uint32_t d = avg - (uint32_t)tgt[k];
          pen += d * d * w[k];
        }
        if ((uint64_t)pen < best) { best = pen; chosen = idx; chosen_amount = t - plan_count; }
      }
    }
    if (chosen_amount > 256 - plan_count) chosen_amount = 256 - plan_count;
    memset(list + plan_count, chosen, (size_t)chosen_amount);
    plan_count += chosen_amount;
    for (int k = 0; k < 4; k++) so_far[k] += plan->y2[chosen][k] * chosen_amount;
  }
  tmo_quicksort(list, 0, plan_count - 1, 1, cmp_luma, (void *)plan->luma);
  return plan_count;
}

void tmo_dither_tile(const uint32_t *rgb_canon, int hm, int vm, const tmo_plan *plan, int use_tk, uint8_t pal_out[64]) {
  /* DitherTile, tilingencoder.pas:2688-2724 */
  uint32_t nat[64];
  memcpy(nat, rgb_canon, sizeof(nat));
  if (hm) tmo_hmirror_u32(nat);
  if (vm) tmo_vmirror_u32(nat);
  for (int y = 0; y < 8; y++)
    for (int x = 0; x < 8; x++) {
      int map_value = tmo_dithering_map[((y & 7) << 3) | (x & 7)];
      if (use_tk) {
        uint8_t l[64];
        tmo_mixing_plan_tk(plan, nat[y * 8 + x], l);
        pal_out[y * 8 + x] = plan->remap[l[map_value]];
      } else {
        uint8_t l[256];
        int count = tmo_mixing_plan_yliluoma(plan, nat[y * 8 + x], l);
        map_value = (map_value * count) >> 6;
        pal_out[y * 8 + x] = plan->remap[l[map_value]];
      }
    }
  if (hm) tmo_hmirror_u8(pal_out);
  if (vm) tmo_vmirror_u8(pal_out);
}

void tmo_dither_tiles(const uint32_t *tiles, const uint8_t *flags, const int32_t *pal_idx, int64_t n, const int32_t *palettes,
                      int pal_size, int use_tk, int y2_mixed, uint8_t *pal_out) { /* Dither, tilingencoder.pas:1873-1907 */
  int maxp = 0;
  for (int64_t t = 0; t < n; t++)
    if (pal_idx[t] > maxp) maxp = pal_idx[t];
  tmo_plan *plans = (tmo_plan *)malloc(sizeof(tmo_plan) * (size_t)(maxp + 1));
  for (int p = 0; p <= maxp; p++) tmo_prepare_plan(&plans[p], palettes + (size_t)p * pal_size, pal_size, y2_mixed);
  for (int64_t t = 0; t < n; t++) {
    int f = flags ? flags[t] : 0;
    tmo_dither_tile(tiles + t * 64, f & 1, (f >> 1) & 1, &plans[pal_idx[t]], use_tk, pal_out + t * 64);
  }
  free(plans);
}

/* ------------------------------------------------------------------ A8/A16 exact dedup + reindex */

typedef struct { const void *keys; int width; int is_u8; const uint32_t *use; } sort_ctx;
static sort_ctx g_sc; /* qsort has no user pointer; the oracle is single-threaded */

static int cmp_content(int64_t a, int64_t b) { /* CompareDWord / CompareByte, tilingencoder.pas:940-948 */
  if (g_sc.is_u8) {
    int r = memcmp((const uint8_t *)g_sc.keys + a * g_sc.width, (const uint8_t *)g_sc.keys + b * g_sc.width, (size_t)g_sc.width);
    return (r > 0) - (r < 0);
  }
  const uint32_t *pa = (const uint32_t *)g_sc.keys + a * g_sc.width, *pb = (const uint32_t *)g_sc.keys + b * g_sc.width;
  for (int i = 0; i < g_sc.width; i++)
    if (pa[i] != pb[i]) return pa[i] < pb[i] ? -1 : 1;
  return 0;
}
static int cmp_rows(const void *x, const void *y) {
  int64_t a = *(const int64_t *)x, b = *(const int64_t *)y;
  int r = cmp_content(a, b);
  return r ? r : (a > b) - (a < b);
}
static int cmp_use_desc(const void *x, const void *y) { /* CompareTileUseCountRev, tilingencoder.pas:584-599 */
  int64_t a = *(const int64_t *)x, b = *(const int64_t *)y;
  if (g_sc.use[a] != g_sc.use[b]) return g_sc.use[a] > g_sc.use[b] ? -1 : 1;
  return cmp_content(a, b);
}

static int64_t dedup_common(const void *keys, int64_t n, int width, int is_u8, const uint32_t *use_in, int64_t *rep,
                            int64_t *order, uint32_t *use_out, int64_t *remap) {
  /* MakeTilesUnique (tilingencoder.pas:4720-4781) + MergeTiles (4783-4813) + ReindexTiles (4626-4700) */
  if (n <= 0) return 0;
  int64_t *srt = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  uint32_t *use = (uint32_t *)calloc((size_t)n, sizeof(uint32_t));
  for (int64_t i = 0; i < n; i++) srt[i] = i;
  g_sc.keys = keys; g_sc.width = width; g_sc.is_u8 = is_u8; g_sc.use = use;
  qsort(srt, (size_t)n, sizeof(int64_t), cmp_rows);
  int64_t nu = 0, first = 0;
  for (int64_t i = 0; i <= n; i++) {
    if (i == n || (i > 0 && cmp_content(srt[i - 1], srt[i]) != 0)) {
      if (i > first || i == n) {
        int64_t r = srt[first]; /* lowest original index of the run: build's representative rule */
        uint32_t u = 0;
        for (int64_t j = first; j < i; j++) { rep[srt[j]] = r; u += use_in ? use_in[srt[j]] : 1u; }
        use[r] = u;
        order[nu++] = r;
      }
      first = i;
    }
  }
  /* drop zero-use (ReindexTiles packs Active and UseCount>0 only), then sort by use desc, content asc */
  int64_t m = 0;
  for (int64_t i = 0; i < nu; i++)
    if (use[order[i]] > 0) order[m++] = order[i];
  qsort(order, (size_t)m, sizeof(int64_t), cmp_use_desc);
  int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  for (int64_t i = 0; i < n; i++) pos[i] = -1;
  for (int64_t i = 0; i < m; i++) { pos[order[i]] = i; if (use_out) use_out[i] = use[order[i]]; }
  if (remap)
    for (int64_t i = 0; i < n; i++) remap[i] = pos[rep[i]];
  free(pos); free(use); free(srt);
  return m;
}

int64_t tmo_dedup_u32(const uint32_t *keys, int64_t n, int kd, const uint32_t *use_in, int64_t *rep, int64_t *order,
                      uint32_t *use_out, int64_t *remap) {
  return dedup_common(keys, n, kd, 0, use_in, rep, order, use_out, remap);
}
int64_t tmo_dedup_u8(const uint8_t *keys, int64_t n, int kb, const uint32_t *use_in, int64_t *rep, int64_t *order,
                     uint32_t *use_out, int64_t *remap) {
  return dedup_common(keys, n, kb, 1, use_in, rep, order, use_out, remap);
}

int tmo_equal_quality_tile_count(double tc) { /* utils.pas:1038-1041; TFloat argument */
  float f = (float)tc;
  return (int)pas_round(sqrt((double)f) * log2(1 + (double)f));
}

/* ------------------------------------------------------------------ A9/A10 k-means of the build */

static inline int64_t sqdist_i(const int32_t *a, const int32_t *b, int d) {
  int64_t s = 0;
  for (int i = 0; i < d; i++) { int64_t t = (int64_t)a[i] - b[i]; s += t * t; }
  return s;
}

/* Lloyd from kk given centroids: double distances accumulated in dimension order with a fused multiply-add, ties -> lowest centroid;
 * exact integer weighted sums; stops when no assignment changes */
static int kmeans_lloyd(const int32_t *pts, const uint32_t *w, int64_t n, int d, int kk, int max_iter, int32_t *assign, double *cent, int *iters_out) {
  int64_t *sum = (int64_t *)malloc(sizeof(int64_t) * (size_t)kk * d);
  int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (size_t)kk);
  for (int64_t i = 0; i < n; i++) assign[i] = -1;
  int it = 0;
  for (; it < max_iter; it++) {
    int64_t changed = 0;
    memset(sum, 0, sizeof(int64_t) * (size_t)kk * d);
    memset(cnt, 0, sizeof(int64_t) * (size_t)kk);
    for (int64_t i = 0; i < n; i++) {
      double bd = 0;
      int bc = -1;
      for (int c = 0; c < kk; c++) {
        double s = 0;
        for (int j = 0; j < d; j++) {
          double t = (double)pts[i * d + j] - cent[(size_t)c * d + j];
          s = fma(t, t, s); /* one fused multiply-add per dimension (the build's rule) */
        }
        if (bc < 0 || s < bd) { bd = s; bc = c; }
      }
      if (assign[i] != bc) { assign[i] = bc; changed++; }
      int64_t wi = w ? w[i] : 1;
      cnt[bc] += wi;
      for (int j = 0; j < d; j++) sum[(size_t)bc * d + j] += wi * pts[i * d + j];
    }
    if (!changed) break;
    for (int c = 0; c < kk; c++)
      if (cnt[c] > 0)
        for (int j = 0; j < d; j++) cent[(size_t)c * d + j] = (double)sum[(size_t)c * d + j] / (double)cnt[c];
  }
  if (iters_out) *iters_out = it;
  free(sum); free(cnt);
  return kk;
}

int tmo_kmeans_i32(const int32_t *pts, const uint32_t *w, int64_t n, int d, int k, int max_iter, int32_t *assign,
                   double *cent, int *iters_out) {
  if (iters_out) *iters_out = 0;
  if (n <= 0 || k <= 0) return 0;
  /* farthest-first ("maximin") init from point 0, ties -> lowest index (cf. InitFarthestFirst, kmodes.pas:694) */
  int64_t *mind = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  for (int64_t i = 0; i < n; i++) mind[i] = INT64_MAX;
  int kk = 0;
  int64_t cur = 0;
  while (kk < k) {
    for (int j = 0; j < d; j++) cent[(size_t)kk * d + j] = (double)pts[cur * d + j];
    kk++;
    int64_t best = 0, bi = -1;
    for (int64_t i = 0; i < n; i++) {
      int64_t dd = sqdist_i(pts + i * d, pts + cur * d, d);
      if (dd < mind[i]) mind[i] = dd;
      if (mind[i] > best) { best = mind[i]; bi = i; }
    }
    if (bi < 0) break; /* no distinct point left */
    cur = bi;
  }
  free(mind);
  return kmeans_lloyd(pts, w, n, d, kk, max_iter, assign, cent, iters_out);
}

/* The build's seeding for the tile -> palette clustering: k-means++-style D^2 sampling made deterministic.
 * The generator is a 64-bit LCG (Knuth's MMIX constants) started at TMO_PP_SEED; pick t draws r = floor(x_t * total / 2^64) over
 * the exact integer masses q_i = weight_i * (squared distance of point i to its nearest centre so far) (q_i = weight_i for the first
 * pick), 128-bit sums, and takes the first point whose running sum exceeds r.  A pick with total mass 0 (no point apart from the
 * centres) ends the seeding with fewer centres.  seeds_out (may be NULL) receives the picked point indices. */
uint64_t tmo_pp_next(uint64_t *state) { *state = *state * 6364136223846793005ull + 1442695040888963407ull; return *state; }
int tmo_kmeans_pp_seeds(const int32_t *pts, const uint32_t *w, int64_t n, int d, int k, int64_t *seeds_out) {
  if (n <= 0 || k <= 0) return 0;
  typedef unsigned __int128 u128;
  int64_t *mind = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  uint64_t rng = TMO_PP_SEED;
  int kk = 0;
  while (kk < k) {
    u128 total = 0;
    for (int64_t i = 0; i < n; i++) total += (u128)(w ? w[i] : 1) * (u128)(kk == 0 ? 1 : (uint64_t)mind[i]);
    if (total == 0) break;
    const uint64_t x = tmo_pp_next(&rng);
    const u128 r = (u128)x * (uint64_t)(total >> 64) + (((u128)x * (uint64_t)total) >> 64); /* floor(x * total / 2^64) < total */
    u128 run = 0;
    int64_t pick = -1;
    for (int64_t i = 0; i < n; i++) {
      run += (u128)(w ? w[i] : 1) * (u128)(kk == 0 ? 1 : (uint64_t)mind[i]);
      if (run > r) { pick = i; break; }
    }
    if (seeds_out) seeds_out[kk] = pick;
    for (int64_t i = 0; i < n; i++) {
      int64_t dd = sqdist_i(pts + i * d, pts + pick * d, d);
      if (kk == 0 || dd < mind[i]) mind[i] = dd;
    }
    kk++;
  }
  free(mind);
  return kk;
}

int tmo_kmeans_pp_i32(const int32_t *pts, const uint32_t *w, int64_t n, int d, int k, int max_iter, int32_t *assign, double *cent, int *iters_out) {
  if (iters_out) *iters_out = 0;
  if (n <= 0 || k <= 0) return 0;
  int64_t *seeds = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
  const int kk = tmo_kmeans_pp_seeds(pts, w, n, d, k, seeds);
  for (int c = 0; c < kk; c++)
    for (int j = 0; j < d; j++) cent[(size_t)c * d + j] = (double)pts[seeds[c] * d + j];
  free(seeds);
  return kmeans_lloyd(pts, w, n, d, kk, max_iter, assign, cent, iters_out);
}

static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return (x > y) - (x < y); }

typedef struct { uint8_t v, s, h, r, g, b; int idx; } cm_item;
static int cmp_cm(const void *a, const void *b) { /* CompareCountIndexVSH, utils.pas:741-748; then r,g,b,idx (build rule) */
  const cm_item *x = (const cm_item *)a, *y = (const cm_item *)b;
  if (x->v != y->v) return x->v < y->v ? -1 : 1;
  if (x->s != y->s) return x->s < y->s ? -1 : 1;
  if (x->h != y->h) return x->h < y->h ? -1 : 1;
  if (x->r != y->r) return x->r < y->r ? -1 : 1;
  if (x->g != y->g) return x->g < y->g ? -1 : 1;
  if (x->b != y->b) return x->b < y->b ? -1 : 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

void tmo_quantize_palette(const uint32_t *pixels, int64_t npx, int pal_size, int max_iter, int32_t *pal_out) {
  /* QuantizeUsingYakmo + DoQuantization, tilingencoder.pas:4434-4564, on the (G,R,B)-sorted colour histogram */
  for (int i = 0; i < pal_size; i++) pal_out[i] = TMO_NULL_COLOR;
  if (npx <= 0) return;
  uint32_t *key = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)npx);
  for (int64_t i = 0; i < npx; i++) { /* CompareDSPixel: G, R, B (tilingencoder.pas:1046-1056) */
    uint32_t c = pixels[i];
    key[i] = (((c >> 8) & 0xff) << 16) | ((c & 0xff) << 8) | ((c >> 16) & 0xff);
  }
  qsort(key, (size_t)npx, sizeof(uint32_t), cmp_u32);
  int64_t nu = 0;
  for (int64_t i = 0; i < npx; i++)
    if (i == 0 || key[i] != key[i - 1]) nu++;
  int32_t *pts = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)nu);
  uint32_t *w = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nu);
  int64_t u = -1;
  for (int64_t i = 0; i < npx; i++) {
    if (i == 0 || key[i] != key[i - 1]) {
      u++;
      pts[u * 3 + 0] = (int32_t)((key[i] >> 8) & 0xff);  /* R */
      pts[u * 3 + 1] = (int32_t)((key[i] >> 16) & 0xff); /* G */
      pts[u * 3 + 2] = (int32_t)(key[i] & 0xff);         /* B */
      w[u] = 0;
    }
    w[u]++;
  }
  int32_t *assign = (int32_t *)malloc(sizeof(int32_t) * (size_t)nu);
  double cent[64 * 3];
  int kk = tmo_kmeans_i32(pts, w, nu, 3, pal_size, max_iter, assign, cent, NULL);
  cm_item items[64];
  for (int i = 0; i < kk; i++) {
    items[i].r = (uint8_t)clampi(pas_round(cent[i * 3 + 0]), 0, 255); /* Posterize(v,255) = v, utils.pas:526-534 */
    items[i].g = (uint8_t)clampi(pas_round(cent[i * 3 + 1]), 0, 255);
    items[i].b = (uint8_t)clampi(pas_round(cent[i * 3 + 2]), 0, 255);
    items[i].idx = i;
    tmo_rgb_to_hsv((uint32_t)items[i].r | ((uint32_t)items[i].g << 8) | ((uint32_t)items[i].b << 16), &items[i].h, &items[i].s,
                   &items[i].v);
  }
  qsort(items, (size_t)kk, sizeof(cm_item), cmp_cm);
  for (int i = 0; i < kk; i++) pal_out[i] = (int32_t)((uint32_t)items[i].r | ((uint32_t)items[i].g << 8) | ((uint32_t)items[i].b << 16));
  free(assign); free(w); free(pts); free(key);
}

void tmo_palettize_tiles(const int32_t *feat, const uint32_t *use, int64_t n, int pal_count, int max_iter, int32_t *pal_idx_out) {
  /* DoPalettization, tilingencoder.pas:4105-4245, with BICO+ANN+yakmo replaced by the build's k-means on all tiles */
  double *cent = (double *)malloc(sizeof(double) * (size_t)pal_count * 192);
  int32_t *assign = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  tmo_kmeans_pp_i32(feat, use, n, 192, pal_count, max_iter, assign, cent, NULL);
  int64_t *cnt = (int64_t *)calloc((size_t)pal_count, sizeof(int64_t));
  for (int64_t i = 0; i < n; i++) cnt[assign[i]]++; /* Inc(FPalettes[..].UseCount) per tile, 4229-4230 */
  int *ord = (int *)malloc(sizeof(int) * (size_t)pal_count), *lut = (int *)malloc(sizeof(int) * (size_t)pal_count);
  for (int i = 0; i < pal_count; i++) ord[i] = i;
  for (int i = 1; i < pal_count; i++) { /* use count desc (ComparePaletteUseCount, utils.pas:750-753), then initial index */
    int v = ord[i], j = i;
    while (j > 0 && cnt[ord[j - 1]] < cnt[v]) { ord[j] = ord[j - 1]; j--; }
    ord[j] = v;
  }
  for (int i = 0; i < pal_count; i++) lut[ord[i]] = i;
  for (int64_t i = 0; i < n; i++) pal_idx_out[i] = lut[assign[i]];
  free(lut); free(ord); free(cnt); free(assign); free(cent);
}

/* ------------------------------------------------------------------ A11 OptimizePalettes + Powell (powell.pas) */

typedef double (*fn1_t)(double t, void *ctx);

static void sw(double *a, double *b) { double t = *a; *a = *b; *b = t; }
static double sgn(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); }

static void bracket(fn1_t f, void *ctx, double xa, double xb, double out[3]) { /* Bracket, powell.pas:56-147 */
  const double Gold = (1 + sqrt(5.0)) / 2, Small = 1e-21, GrowLimit = 110;
  double fa = f(xa, ctx), fb = f(xb, ctx), fc, xc, w, fw, tmp1, tmp2, val, denom, wlim;
  int iter = 0;
  if (fa < fb) { sw(&xa, &xb); sw(&fa, &fb); }
  xc = xb + Gold * (xb - xa);
  fc = f(xc, ctx);
  while (fc < fb) {
    tmp1 = (xb - xa) * (fb - fc);
    tmp2 = (xb - xc) * (fb - fa);
    val = tmp2 - tmp1;
    denom = fabs(val) < Small ? 2 * Small : 2 * val;
    w = xb - ((xb - xc) * tmp2 - (xb - xa) * tmp1) / denom;
    wlim = xb + GrowLimit * (xc - xb);
    if (iter > 1000) break; /* the reference raises; unreachable for bounded objectives */
    iter++;
    fw = 0;
    if ((w - xc) * (xb - w) > 0) {
      fw = f(w, ctx);
      if (fw < fc) { xa = xb; xb = w; fa = fb; fb = fw; break; }
      else if (fw > fb) { xc = w; fc = fw; break; }
      w = xc + Gold * (xc - xb);
      fw = f(w, ctx);
    } else if ((w - wlim) * (wlim - xc) >= 0) {
      w = wlim;
      fw = f(w, ctx);
    } else if ((w - wlim) * (xc - w) > 0) {
      fw = f(w, ctx);
      if (fw < fc) { xb = xc; xc = w; w = xc + Gold * (xc - xb); fb = fc; fc = fw; fw = f(w, ctx); }
    } else {
      w = xc + Gold * (xc - xb);
      fw = f(w, ctx);
    }
    xa = xb; xb = xc; xc = w;
    fa = fb; fb = fc; fc = fw;
  }
  if (xa > xc) { sw(&xa, &xc); sw(&fa, &fc); }
  out[0] = xa; out[1] = xb; out[2] = xc;
}

static void brent(fn1_t f, void *ctx, double xtol, int maxiter, double *xmin, double *fmin) { /* Brent + BrentHelper, 149-266 */
  const double CG = (3 - sqrt(5.0)) / 2;
  double br[3];
  bracket(f, ctx, 0, 1, br);
  double a = br[0], x = br[1], b = br[2], fx = f(x, ctx);
  if (a > b) sw(&a, &b);
  double w = x, v = x, fw = fx, fv = fx, deltax = 0, rat = 0, xmid, tmp1, tmp2, p, dx_temp, u, fu;
  int iter = 0;
  while (iter < maxiter) {
    xmid = 0.5 * (a + b);
    if (fabs(x - xmid) <= 2 * xtol - 0.5 * (b - a)) break;
    if (fabs(deltax) <= xtol) {
      deltax = x >= xmid ? a - x : b - x;
      rat = CG * deltax;
    } else {
      tmp1 = (x - w) * (fx - fv);
      tmp2 = (x - v) * (fx - fw);
      p = (x - v) * tmp2 - (x - w) * tmp1;
      tmp2 = 2 * (tmp2 - tmp1);
      if (tmp2 > 0) p = -p;
      tmp2 = fabs(tmp2);
      dx_temp = deltax;
      deltax = rat;
      if (p > tmp2 * (a - x) && p < tmp2 * (b - x) && fabs(p) < fabs(0.5 * tmp2 * dx_temp)) {
        rat = p / tmp2;
        u = x + rat;
        if (u - a < xtol || b - u < xtol) rat = sgn(xmid - x) * xtol;
      } else {
        deltax = x >= xmid ? a - x : b - x;
        rat = CG * deltax;
      }
    }
    u = fabs(rat) > xtol ? x + rat : x + sgn(rat) * xtol;
    fu = f(u, ctx);
    if (fu > fx) {
      if (u < x) a = u; else b = u;
      if (fu <= fw || w == x) { v = w; w = u; fv = fw; fw = fu; }
      else if (fu <= fv || v == x || v == w) { v = u; fv = fu; }
    } else {
      if (u >= x) a = x; else b = x;
      v = w; w = x; x = u;
      fv = fw; fw = fx; fx = fu;
    }
    iter++;
  }
  *xmin = x;
  *fmin = fx;
}

typedef double (*fnn_t)(const double *x, void *data);
typedef struct { fnn_t f; void *data; const double *p, *xi; int n; double *tmp; } ray_ctx;
static double along_ray(double t, void *c) { /* AlongRay1, powell.pas:273-282 */
  ray_ctx *r = (ray_ctx *)c;
  for (int i = 0; i < r->n; i++) r->tmp[i] = r->p[i] + t * r->xi[i];
  return r->f(r->tmp, r->data);
}

static double linesearch(fnn_t f, void *data, double *p, double *xi, int n, double xtol, double *scratch) { /* 285-314 */
  ray_ctx rc = {f, data, p, xi, n, scratch};
  double sos = 0;
  for (int i = 0; i < n; i++) sos += xi[i] * xi[i];
  double sqsos = sqrt(sos), atol = 1.0;
  if (sqsos != 0) atol = 5 * xtol / sqsos;
  if (atol > 0.1) atol = 0.1;
  double alpha, fret;
  brent(along_ray, &rc, atol, 100, &alpha, &fret);
  for (int i = 0; i < n; i++) { xi[i] = xi[i] * alpha; p[i] = p[i] + xi[i]; }
  return fret;
}

static double powell_minimize(fnn_t f, void *data, double *x, int n, double scale, double xtol, double ftol, int maxiter) {
  /* PowellMinimize, powell.pas:316-384.  FPC dynamic arrays are references: "direc[bigind] := direc[n-1];
   * direc[n-1] := direc1" aliases the rows, and direc1 keeps being overwritten -- modelled with row pointers. */
  double *store = (double *)calloc((size_t)(n + 1) * n + 3 * (size_t)n, sizeof(double));
  double **direc = (double **)malloc(sizeof(double *) * (size_t)n);
  for (int i = 0; i < n; i++) { direc[i] = store + (size_t)i * n; direc[i][i] = scale; }
  double *direc1 = store + (size_t)n * n, *tmp = direc1 + n, *x1 = tmp + n, *scratch = x1 + n;
  double fval = f(x, data);
  memcpy(x1, x, sizeof(double) * (size_t)n);
  int iter = 0;
  for (;;) {
    double fx = fval, delta = 0;
    int bigind = 0;
    for (int i = 0; i < n; i++) {
      double fx2 = fval;
      fval = linesearch(f, data, x, direc[i], n, xtol, scratch);
      if (fx2 - fval > delta) { delta = fx2 - fval; bigind = i; }
    }
    iter++;
    if (fx - fval <= ftol || iter >= maxiter) break;
    for (int i = 0; i < n; i++) { direc1[i] = x[i] - x1[i]; tmp[i] = x[i] + direc1[i]; x1[i] = x[i]; }
    double fx2 = f(tmp, data);
    if (fx > fx2) {
      double t = 2 * (fx + fx2 - 2 * fval), temp = fx - fval - delta;
      t = t * temp * temp;
      temp = fx - fx2;
      t = t - delta * temp * temp;
      if (t < 0) {
        fval = linesearch(f, data, x, direc1, n, xtol, scratch);
        direc[bigind] = direc[n - 1];
        direc[n - 1] = direc1;
      }
    }
  }
  free(direc);
  free(store);
  return fval;
}

typedef struct { int pal_size, cur; const int32_t *pals; int32_t *newpal; uint64_t mean[3]; uint64_t acc[3][64]; } op_ctx;
typedef struct { int count, index; } perm_item;
static int cmp_perm(const void *a, const void *b) { /* ComparePerms, tilingencoder.pas:4258-4263 */
  const perm_item *x = (const perm_item *)a, *y = (const perm_item *)b;
  if (x->count != y->count) return x->count < y->count ? -1 : 1;
  return (x->index > y->index) - (x->index < y->index);
}

static double powell_op(const double *x, void *data) { /* PowellOP, tilingencoder.pas:4265-4307 */
  op_ctx *c = (op_ctx *)data;
  perm_item perm[64];
  perm[0].index = 0; perm[0].count = 0;
  for (int i = 1; i < c->pal_size; i++) { perm[i].index = i; perm[i].count = (int)pas_round(x[i - 1] * 1000); }
  qsort(perm, (size_t)c->pal_size, sizeof(perm_item), cmp_perm);
  uint64_t sd[3] = {0, 0, 0};
  for (int i = 0; i < c->pal_size; i++) {
    const uint32_t col = (uint32_t)c->pals[(size_t)c->cur * c->pal_size + perm[i].index];
    c->newpal[(size_t)c->cur * c->pal_size + i] = (int32_t)col;
    const uint64_t ch[3] = {col & 0xff, (col >> 8) & 0xff, (col >> 16) & 0xff};
    for (int k = 0; k < 3; k++) { const uint64_t d = c->acc[k][i] + ch[k] - c->mean[k]; sd[k] += d * d; } /* UInt64 wrap = signed square */
  }
  const double r = (299 * sqrt((double)sd[0] / c->pal_size) + 587 * sqrt((double)sd[1] / c->pal_size) +
                    114 * sqrt((double)sd[2] / c->pal_size)) / 1000;
  return -r;
}

int tmo_optimize_palettes(int32_t *pals, int pal_count, int pal_size) { /* OptimizePalettes, tilingencoder.pas:4309-4432 */
  int32_t *newpal = (int32_t *)malloc(sizeof(int32_t) * (size_t)pal_count * pal_size);
  double *f = (double *)malloc(sizeof(double) * (size_t)pal_count);
  uint64_t mean[3] = {0, 0, 0};
  for (int p = 0; p < pal_count; p++)
    for (int i = 0; i < pal_size; i++) {
      const uint32_t col = (uint32_t)pals[(size_t)p * pal_size + i];
      mean[0] += col & 0xff; mean[1] += (col >> 8) & 0xff; mean[2] += (col >> 16) & 0xff;
    }
  for (int k = 0; k < 3; k++) mean[k] /= (uint64_t)pal_size;
  int iteration = 0;
  double fsum = 0, prev = 0;
  do {
    prev = fsum > prev ? fsum : prev;
    iteration++;
    for (int a = 0; a < pal_count; a++) { /* DoPal; reads the palettes as they stood before this sweep */
      op_ctx c;
      memset(&c, 0, sizeof(c));
      c.pal_size = pal_size; c.cur = a; c.pals = pals; c.newpal = newpal;
      memcpy(c.mean, mean, sizeof(mean));
      for (int p = 0; p < pal_count; p++)
        if (p != a)
          for (int i = 0; i < pal_size; i++) {
            const uint32_t col = (uint32_t)pals[(size_t)p * pal_size + i];
            c.acc[0][i] += col & 0xff; c.acc[1][i] += (col >> 8) & 0xff; c.acc[2][i] += (col >> 16) & 0xff;
          }
      double x[64];
      for (int i = 1; i < pal_size; i++) x[i - 1] = i;
      powell_minimize(powell_op, &c, x, pal_size - 1, 1.0, 1.0, 1.0, 2147483647);
      f[a] = -powell_op(x, &c);
    }
    fsum = 0;
    for (int p = 0; p < pal_count; p++) fsum += f[p];
    memcpy(pals, newpal, sizeof(int32_t) * (size_t)pal_count * pal_size);
    fsum /= pal_count;
  } while (!(fsum <= prev));
  free(f);
  free(newpal);
  return iteration;
}

/* Test hooks for A11's scalar pieces (tests/test_oracle_pins.py): Bracket and Brent of powell.pas on a handful of fixed scalar functions
 * (+ - * / and fabs only, so that Python restates them bit for bit), with the function evaluations counted -- the fixture they are held to
 * was recorded from scipy.optimize (tests/golden/make_powell_fixtures.py), the code powell.pas says it was taken from. */
double tmo_test_scalar_fn(int id, double x) {
  switch (id) {
    case 0: return (x - 2.0) * (x - 2.0) + 1.0;
    case 1: return x * x * x * x - 3.0 * x * x * x + 2.0;
    case 2: return (x + 1.5) * (x + 1.5) * (x - 0.3) * (x - 0.3) + 0.1 * x;
    case 3: return fabs(x - 0.7) + 0.01 * x * x;
    case 4: return x * x / (1.0 + x * x) - 0.2 * x;
    case 5: return -1.0 / (1.0 + (x - 3.0) * (x - 3.0));
    case 6: return (x - 0.25) * (x - 0.25) * (x - 0.25) * (x - 0.25) + 0.5 * (x - 0.25) * (x - 0.25);
    default: return 1e3 * (x + 40.0) * (x + 40.0) - 7.0;
  }
}
typedef struct { int id; int calls; } test_fn_ctx;
static double test_fn_counted(double x, void *c) { test_fn_ctx *t = (test_fn_ctx *)c; t->calls++; return tmo_test_scalar_fn(t->id, x); }
int tmo_test_bracket(int fn, double xa, double xb, double out[4]) { /* xa, xb, xc as Bracket returns them (sorted ends), function evaluations */
  test_fn_ctx c = {fn, 0};
  bracket(test_fn_counted, &c, xa, xb, out);
  out[3] = c.calls;
  return 0;
}
int tmo_test_brent(int fn, double xtol, int maxiter, double out[3]) { /* xmin, f(xmin), function evaluations (Bracket's included) */
  test_fn_ctx c = {fn, 0};
  brent(test_fn_counted, &c, xtol, maxiter, &out[0], &out[1]);
  out[2] = c.calls;
  return 0;
}

/* =====================================================================================================================
 * (f)#2 checker: LZMA-alone decoder, restating decoders/htmljs/lzma.js (decodeHeader :395-450, decodeBody :477-576,
 * RangeDecoder :128-190, LenDecoder :252-264, Decoder2 :270-300).  Used by tests/ to read back what the product's
 * .gtm writer emits (the reference compresses with LZCompress, extern.pas:420-439: lc 8, lp 0, pb 2, end marker).
 * ===================================================================================================================== */
typedef struct { const uint8_t *p, *end; uint32_t range, code; int overrun; } lz_rc;
static inline uint8_t lz_next(lz_rc *rc) { if (rc->p < rc->end) return *rc->p++; rc->overrun = 1; return 0; }
static inline int lz_bit(lz_rc *rc, uint16_t *prob) {
  const uint32_t bound = (rc->range >> 11) * *prob;
  int b;
  if (rc->code < bound) { rc->range = bound; *prob = (uint16_t)(*prob + ((2048 - *prob) >> 5)); b = 0; }
  else { rc->range -= bound; rc->code -= bound; *prob = (uint16_t)(*prob - (*prob >> 5)); b = 1; }
  if (rc->range < (1u << 24)) { rc->range <<= 8; rc->code = (rc->code << 8) | lz_next(rc); }
  return b;
}
static inline uint32_t lz_direct(lz_rc *rc, int nbits) {
  uint32_t r = 0;
  for (int i = 0; i < nbits; i++) {
    rc->range >>= 1;
    const uint32_t t = (rc->code - rc->range) >> 31;  /* 1 if code < range */
    rc->code -= rc->range & (t - 1);
    r = (r << 1) | (1 - t);
    if (rc->range < (1u << 24)) { rc->range <<= 8; rc->code = (rc->code << 8) | lz_next(rc); }
  }
  return r;
}
static inline uint32_t lz_tree(lz_rc *rc, uint16_t *probs, int nbits) {
  uint32_t m = 1;
  for (int i = 0; i < nbits; i++) m = (m << 1) | (uint32_t)lz_bit(rc, &probs[m]);
  return m - (1u << nbits);
}
static inline uint32_t lz_rtree(lz_rc *rc, uint16_t *probs, int nbits) {
  uint32_t m = 1, sym = 0;
  for (int i = 0; i < nbits; i++) { const int b = lz_bit(rc, &probs[m]); m = (m << 1) | (uint32_t)b; sym |= (uint32_t)b << i; }
  return sym;
}
typedef struct { uint16_t choice[2], low[16][8], mid[16][8], high[256]; } lz_len;
static uint32_t lz_len_decode(lz_rc *rc, lz_len *l, uint32_t pos_state) {
  if (!lz_bit(rc, &l->choice[0])) return lz_tree(rc, l->low[pos_state], 3);
  if (!lz_bit(rc, &l->choice[1])) return 8 + lz_tree(rc, l->mid[pos_state], 3);
  return 16 + lz_tree(rc, l->high, 8);
}

int64_t tmo_lzma_decode(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *consumed, int *props_out) {
  if (n < 13 + 5) return -1;
  int props = src[0];
  if (props_out) { props_out[0] = props; props_out[1] = (int)(src[1] | (src[2] << 8) | (src[3] << 16) | ((uint32_t)src[4] << 24)); }
  const int lc = props % 9; props /= 9;
  const int lp = props % 5, pb = props / 5;
  uint64_t usize = 0;
  for (int i = 0; i < 8; i++) usize |= (uint64_t)src[5 + i] << (8 * i);
  if (props_out) props_out[2] = usize == UINT64_MAX ? -1 : (int)usize;
  const size_t nlit = (size_t)0x300 << (lc + lp);
  uint16_t *lit = (uint16_t *)malloc(nlit * sizeof(uint16_t));
  static _Thread_local uint16_t is_match[12 << 4], is_rep0_long[12 << 4], is_rep[12], is_g0[12], is_g1[12], is_g2[12], pos_slot[4][64],
      pos_dec[128], pos_align[16];
  static _Thread_local lz_len len_dec, rep_len_dec;
#define LZ_INIT(a) do { uint16_t *q_ = (uint16_t *)(a); for (size_t i_ = 0; i_ < sizeof(a) / 2; i_++) q_[i_] = 1024; } while (0)
  LZ_INIT(is_match); LZ_INIT(is_rep0_long); LZ_INIT(is_rep); LZ_INIT(is_g0); LZ_INIT(is_g1); LZ_INIT(is_g2); LZ_INIT(pos_slot);
  LZ_INIT(pos_dec); LZ_INIT(pos_align);
  { uint16_t *q = (uint16_t *)&len_dec; for (size_t i = 0; i < sizeof(len_dec) / 2; i++) q[i] = 1024; }
  { uint16_t *q = (uint16_t *)&rep_len_dec; for (size_t i = 0; i < sizeof(rep_len_dec) / 2; i++) q[i] = 1024; }
  for (size_t i = 0; i < nlit; i++) lit[i] = 1024;
  lz_rc rc = { src + 13, src + n, 0xFFFFFFFFu, 0, 0 };
  for (int i = 0; i < 5; i++) rc.code = (rc.code << 8) | lz_next(&rc);
  uint32_t state = 0, rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0;
  uint64_t pos = 0;
  uint8_t prev = 0;
  int64_t result = -1;
  for (;;) {
    if (usize != UINT64_MAX && pos >= usize) { result = (int64_t)pos; break; }
    if (rc.overrun) break;
    const uint32_t pos_state = (uint32_t)pos & ((1u << pb) - 1);
    if (!lz_bit(&rc, &is_match[(state << 4) + pos_state])) {
      uint16_t *probs = lit + (size_t)0x300 * ((((uint32_t)pos & ((1u << lp) - 1)) << lc) + (prev >> (8 - lc)));
      uint32_t sym = 1;
      if (state >= 7) {
        uint32_t mb = dst[pos - rep0 - 1];
        do {
          const uint32_t mbit = (mb >> 7) & 1;
          mb <<= 1;
          const int b = lz_bit(&rc, &probs[((1 + mbit) << 8) + sym]);
          sym = (sym << 1) | (uint32_t)b;
          if (mbit != (uint32_t)b) { while (sym < 0x100) sym = (sym << 1) | (uint32_t)lz_bit(&rc, &probs[sym]); break; }
        } while (sym < 0x100);
      } else {
        do sym = (sym << 1) | (uint32_t)lz_bit(&rc, &probs[sym]); while (sym < 0x100);
      }
      if (pos >= cap) break;
      prev = (uint8_t)sym;
      dst[pos++] = prev;
      state = state < 4 ? 0 : state - (state < 10 ? 3 : 6);
      continue;
    }
    uint32_t len;
    if (lz_bit(&rc, &is_rep[state])) {
      len = 0;
      if (!lz_bit(&rc, &is_g0[state])) {
        if (!lz_bit(&rc, &is_rep0_long[(state << 4) + pos_state])) { state = state < 7 ? 9 : 11; len = 1; }
      } else {
        uint32_t dist;
        if (!lz_bit(&rc, &is_g1[state])) dist = rep1;
        else {
          if (!lz_bit(&rc, &is_g2[state])) dist = rep2;
          else { dist = rep3; rep3 = rep2; }
          rep2 = rep1;
        }
        rep1 = rep0;
        rep0 = dist;
      }
      if (len == 0) { len = 2 + lz_len_decode(&rc, &rep_len_dec, pos_state); state = state < 7 ? 8 : 11; }
    } else {
      rep3 = rep2; rep2 = rep1; rep1 = rep0;
      len = 2 + lz_len_decode(&rc, &len_dec, pos_state);
      state = state < 7 ? 7 : 10;
      const uint32_t slot = lz_tree(&rc, pos_slot[len <= 5 ? len - 2 : 3], 6);
      if (slot >= 4) {
        const int nd = (int)(slot >> 1) - 1;
        rep0 = (2u | (slot & 1)) << nd;
        if (slot < 14) rep0 += lz_rtree(&rc, pos_dec + rep0 - slot - 1, nd);
        else {
          rep0 += lz_direct(&rc, nd - 4) << 4;
          rep0 += lz_rtree(&rc, pos_align, 4);
          if (rep0 == 0xFFFFFFFFu) { result = (int64_t)pos; break; }  /* end marker */
        }
      } else rep0 = slot;
    }
    if ((uint64_t)rep0 >= pos || pos + len > cap) break;  /* lzma.js:561: corrupt */
    for (uint32_t i = 0; i < len; i++, pos++) dst[pos] = dst[pos - rep0 - 1];
    prev = dst[pos - 1];
  }
#undef LZ_INIT
  free(lit);
  if (rc.overrun) result = -1;
  if (consumed) *consumed = (size_t)(rc.p - src);
  return result;
}

/* =====================================================================================================================
 * (f)#1 motion prediction: PredictMotion (tilingencoder.pas:1154-1282), the redo inside Reconstruct (1496-1532), the
 * tile-count search of Reduce (4014-4046, utils.pas:1044-1072).
 * ===================================================================================================================== */

void tmo_window_dcts(const uint32_t *fb, int w, int h, int16_t *out) {
  /* DoDCTs (1157-1182 / 1437-1462): for every top-left (x, y) of an 8x8 window inside the w x h frame buffer, the
   * pvsWeightedDCT features of that window (ConvertToCpnPixels YUV, no mirrors); row-major, (w-7) per row */
  const int ww = w - 7;
  for (int y = 0; y + 8 <= h; y++)
    for (int x = 0; x + 8 <= w; x++) {
      uint32_t tile[64];
      float cpn[192];
      for (int j = 0; j < 8; j++) memcpy(tile + j * 8, fb + (size_t)(y + j) * w + x, 32); /* CopyRGBPixels, 879-887 */
      tmo_cpn_from_rgb(tile, 0, 0, 0, cpn);
      tmo_features_i16(cpn, TMO_PVS_WEIGHTED_DCT, out + ((size_t)y * ww + x) * 192);
    }
}

void tmo_motion_search(const int16_t *cur, int tm_w, int tm_h, const int16_t *win, int radius, uint32_t *best_err, int8_t *px,
                       int8_t *py) {
  /* DoXY (1209-1253) == the redo (1496-1532): cur = features of the frame's tiles in their ORIGINAL orientation, win =
   * tmo_window_dcts of the back buffer (screen = tm_w*8 x tm_h*8), radius = the MotionPredictRadius setting (the
   * reference decrements it first, 1271/1666).  err = CompareEuclideanDCTPtr_asm(cur, prev) + manhattan distance to
   * the tile's own position; first strict minimum in raster order wins.  The QuickTest early-out (1230, 1513) cannot
   * change the outcome: its value is one of the non-negative terms of err (true while no err wraps past 2^32; tile features
   * of 8-bit images stay far below, for arbitrary int16 data this restatement is "minimum of the wrapped values"). */
  const int sw = tm_w * 8, sh = tm_h * 8, ww = sw - 7, r = radius - 1;
  for (int sy = 0; sy < tm_h; sy++)
    for (int sx = 0; sx < tm_w; sx++) {
      const int dx = sx * 8, dy = sy * 8, i = sy * tm_w + sx;
      const int oymn = dy - r - 1 > 0 ? dy - r - 1 : 0, oymx = dy + r < sh - 8 ? dy + r : sh - 8;
      const int oxmn = dx - r - 1 > 0 ? dx - r - 1 : 0, oxmx = dx + r < sw - 8 ? dx + r : sw - 8;
      uint32_t best = UINT32_MAX;
      int bx = 0, by = 0;
      for (int oy = oymn; oy <= oymx; oy++)
        for (int ox = oxmn; ox <= oxmx; ox++) {
          uint32_t err = tmo_ssd_i16_sse_quirk(cur + (size_t)i * 192, win + ((size_t)oy * ww + ox) * 192);
          err += (uint32_t)(abs(ox - dx) + abs(oy - dy));
          if (err < best) { best = err; bx = ox; by = oy; }
        }
      best_err[i] = best;
      px[i] = (int8_t)(bx - dx);
      py[i] = (int8_t)(by - dy);
    }
}

/* GoldenRatioSearch(STCGREval) of SolveTileCount (4043-4046; utils.pas:1044-1072) on the sorted per-group minima of
 * the effective PSNR (keyframe-start frames: PSNR / 10, 4028-4029): f(x) = number of groups (distinct tile contents)
 * with at least one member that is NOT predicted at threshold x, i.e. with min effective PSNR <= x.
 * Returns the LAST x the search evaluated (the encoder's state is whatever that probe left, 4036-4040);
 * *probes = evaluations made (0: the interval was already closed, no state). */
double tmo_solve_tile_count(const double *sorted_min_psnr, int64_t ngroups, double target, int *probes) {
  const double phi = (1.0 + sqrt(5.0)) / 2.0, inv_phi = 1.0 / phi; /* cPhi, cInvPhi utils.pas:42-43 */
  double mn = 0.0, mx = 10.0 * log(255.0 * 255.0 / 0.5) / log(10.0), last = 0.0; /* cPsnrMaxValue, utils.pas:111 */
  int n = 0;
  for (;;) {
    if (fabs(mn - mx) <= 1e-6) break; /* SameValue(MinX, MaxX, cPsyVEpsilon) */
    const double x = mn < mx ? mn + (mx - mn) * (1.0 - inv_phi) : mn + (mx - mn) * inv_phi;
    int64_t lo = 0, hi = ngroups; /* count of minima <= x */
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted_min_psnr[mid] <= x) lo = mid + 1; else hi = mid; }
    const double y = (double)lo;
    last = x; n++;
    if (fabs(y - target) <= 0.5) break;       /* CompareValue(y, ObjectiveY, 0.5) = Equals */
    if (y < target) mn = x; else mx = x;
  }
  if (probes) *probes = n;
  return last;
}

/* =====================================================================================================================
 * (f)#3 FrameTilingExtendedPaletteUsage: the k = 64 branch of TFrame.Reconstruct.DoXY (tilingencoder.pas:1559-1610).
 * knn_idx = the 64 nearest database rows of the query (ann_kdtree_short_search_multi, 1563; -1 pads a short list),
 * tile_pal_idx = PalIdx_Initial per global tile.  Every unique tile index is tried with every unique palette of the
 * list, in ascending (tile, palette) order, by the asm distance (CompareEuclideanDCTPtr_asm, quirks included); the first
 * strict minimum wins.  QuickTestEuclideanDCTPtr_asm (1593) cannot change the outcome (its value is one of the
 * non-negative terms of the distance).
 * ===================================================================================================================== */
static int cmp_i32(const void *a, const void *b) { int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return (x > y) - (x < y); }

void tmo_epu_rerank(const int16_t *q, const int32_t *knn_idx, int k, const uint8_t *pal_px, const int32_t *tile_pal_idx, int64_t ntiles,
                    const int32_t *palettes, int pal_size, int32_t *out_tile, int32_t *out_pal, uint32_t *out_err) {
  int32_t tiles[64], pals[64];
  if (k > 64) k = 64;
  for (int i = 0; i < k; i++) {
    if (knn_idx[i] >= 0 && knn_idx[i] < ntiles) { tiles[i] = knn_idx[i]; pals[i] = tile_pal_idx[knn_idx[i]]; } /* 1565-1574 */
    else { tiles[i] = -1; pals[i] = -1; }
  }
  qsort(tiles, (size_t)k, sizeof(int32_t), cmp_i32); /* QuickSort + CompareIntegers, 1576-1577: any correct sort of integers */
  qsort(pals, (size_t)k, sizeof(int32_t), cmp_i32);
  uint32_t best = UINT32_MAX;
  int32_t bt = -1, bp = -1, prev_t = -1;
  for (int ti = 0; ti < k; ti++) {
    if (tiles[ti] == prev_t) continue; /* also skips the -1 pads of a short list: prevTileIdx starts at -1 (1582) */
    int32_t prev_p = -1;
    for (int pi = 0; pi < k; pi++) {
      if (pals[pi] == prev_p) continue;
      float cpn[192];
      int16_t cur[192];
      tmo_cpn_from_pal(pal_px + (size_t)tiles[ti] * 64, palettes + (size_t)pals[pi] * pal_size, 0, 0, 0, cpn); /* 1590 */
      tmo_features_i16(cpn, TMO_PVS_WEIGHTED_DCT, cur);
      const uint32_t err = tmo_ssd_i16_sse_quirk(q, cur); /* 1595 */
      if (err < best) { best = err; bt = tiles[ti]; bp = pals[pi]; }
      prev_p = pals[pi];
    }
    prev_t = tiles[ti];
  }
  *out_tile = bt; *out_pal = bp; *out_err = best;
}

void tmo_epu_rerank_batch(const int16_t *q, int64_t nq, const int32_t *knn_idx, int k, const uint8_t *pal_px, const int32_t *tile_pal_idx,
                          int64_t ntiles, const int32_t *palettes, int pal_size, int32_t *out_tile, int32_t *out_pal, uint32_t *out_err) {
  for (int64_t i = 0; i < nq; i++)
    tmo_epu_rerank(q + i * 192, knn_idx + i * k, k, pal_px, tile_pal_idx, ntiles, palettes, pal_size, out_tile + i, out_pal + i, out_err + i);
}

/* ------------------------------------------------------------------ A17 k-modes (kmodes.pas; unreachable in the reference, named by north_star)
 * TKModes.ComputeKModes (kmodes.pas:923-1094) on rows of cKModesFeatureCount = 80 bytes (the asm paths hard-code 80: 338-342).
 * Restated step by step: MatchingDissim (248-259: sum |a-b| + 2048 per differing byte), GetMinMatchingDissim (the LAST minimum
 * wins: `dis <= best`, 272 / asm `ja worst` 414), InitFarthestFirst (694-772: the LAST largest min-distance among unused points wins,
 * `>=` at 759), the initial assignment and modes (978-1011: first largest count wins, GetMaxValueIndex 155-167; an empty
 * cluster draws one RandInt per attribute), KModesIter (851-921: bins of 960 points scored against the centroids as they stand at
 * the start of the bin, then moved one by one with Huang's online mode update MovePointCat 774-803 and the empty-cluster repair
 * with the LCG RandInt 88-92), the stopping rule (1040-1049: cost not below the previous one, with three graces while within
 * prevcost div 1000) and the best-cost bookkeeping (1051-1057).  Parity unpinned: the reference holds no vector for it. */
static uint32_t kmodes_randint(uint32_t range, uint32_t *seed) { /* kmodes.pas:88-92 */
  *seed = (uint32_t)((int32_t)(*seed * 0x08088405u) + 1);
  return (uint32_t)(((uint64_t)*seed * (uint64_t)range) >> 32);
}
static uint64_t kmodes_dissim(const uint8_t *a, const uint8_t *b) { /* kmodes.pas:248-259 */
  uint64_t r = 0;
  for (int i = 0; i < 80; i++) {
    if (a[i] != b[i]) r += (uint64_t)1 << 11;
    r += (uint64_t)llabs((long long)a[i] - (long long)b[i]);
  }
  return r;
}
static int kmodes_argmin(const uint8_t *cent, int k, const uint8_t *row, uint64_t *best_out) { /* kmodes.pas:263-281 */
  int res = -1;
  uint64_t best = UINT64_MAX;
  for (int i = 0; i < k; i++) {
    const uint64_t d = kmodes_dissim(cent + (size_t)i * 80, row);
    if (d <= best) { best = d; res = i; }
  }
  *best_out = best;
  return res;
}
typedef struct {
  const uint8_t *x; int64_t n; int k, nmod;
  int32_t *memb; int64_t *members; uint8_t *cent; int32_t *freq; /* [k][80][nmod] */
} kmodes_t;
static int kmodes_maxidx(const int32_t *arr, int n) { /* GetMaxValueIndex, kmodes.pas:155-167 */
  int res = -1, best = INT32_MIN;
  for (int i = 0; i < n; i++) if (arr[i] > best) { best = arr[i]; res = i; }
  return res;
}
static void kmodes_move(kmodes_t *s, int64_t ipoint, int to, int from) { /* MovePointCat, kmodes.pas:774-803 */
  const uint8_t *p = s->x + ipoint * 80;
  s->memb[ipoint] = to;
  s->members[to]++;
  s->members[from]--;
  for (int a = 0; a < 80; a++) {
    const int cur = p[a];
    int32_t *tc = s->freq + ((size_t)to * 80 + a) * s->nmod, *fc = s->freq + ((size_t)from * 80 + a) * s->nmod;
    tc[cur]++;
    if (tc[s->cent[(size_t)to * 80 + a]] < tc[cur]) s->cent[(size_t)to * 80 + a] = (uint8_t)cur;
    fc[cur]--;
    if (s->cent[(size_t)from * 80 + a] == cur) s->cent[(size_t)from * 80 + a] = (uint8_t)kmodes_maxidx(fc, s->nmod);
  }
}
int tmo_kmodes(const uint8_t *x, int64_t n, int k, int num_init, int nmod, int max_iter, int32_t *labels_out, uint8_t *cent_out, uint64_t *cost_out,
               int *iters_out) {
  uint32_t seed = 0x42381337u;
  if (n <= 0 || k <= 0) return 0;
  if (max_iter < 0) max_iter = INT32_MAX;
  const int nruns = num_init <= 0 ? 1 : num_init;
  int64_t *starts = (int64_t *)malloc(sizeof(int64_t) * (size_t)nruns);
  if (num_init <= 0) starts[0] = -num_init;
  else { /* 952-964: Single arithmetic */
    const float ratio = (float)pow((double)n, 1.0 / (double)num_init); /* power(NumPoints, 1 / ANumInit) narrowed to Single */
    float acc = 1.0f;
    for (int i = 0; i < nruns; i++) {
      starts[i] = (int64_t)pas_round((double)acc) - 1;
      if (i > 0 && starts[i] <= starts[i - 1]) starts[i] = starts[i - 1] + 1 < n - 1 ? starts[i - 1] + 1 : n - 1;
      acc = acc * ratio;
    }
  }
  kmodes_t s;
  s.x = x; s.n = n; s.k = k; s.nmod = nmod;
  s.memb = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  s.members = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k + 1)) + 1; /* members[-1]: points not assigned yet */
  s.cent = (uint8_t *)malloc((size_t)k * 80);
  s.freq = (int32_t *)malloc(sizeof(int32_t) * (size_t)k * 80 * nmod);
  int32_t *clust = (int32_t *)malloc(sizeof(int32_t) * (size_t)n), *bestm = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  uint64_t *dis = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n), *mind = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
  uint8_t *used = (uint8_t *)malloc((size_t)n), *bestc = (uint8_t *)malloc((size_t)k * 80);
  uint64_t all_best = UINT64_MAX;
  int all_iters = 0;
  for (int run = 0; run < nruns; run++) {
    /* InitFarthestFirst, 694-772 */
    memset(s.cent, 0xff, (size_t)k * 80);
    memset(used, 0, (size_t)n);
    for (int64_t i = 0; i < n; i++) mind[i] = UINT64_MAX;
    int64_t far = starts[run];
    for (int c = 0; c < k; c++) {
      if (c > 0) {
        uint64_t mx = 0;
        far = starts[run];
        for (int64_t i = 0; i < n; i++) if (mind[i] >= mx && !used[i]) { mx = mind[i]; far = i; }
      }
      memcpy(s.cent + (size_t)c * 80, x + far * 80, 80);
      used[far] = 1;
      for (int64_t i = 0; i < n; i++) if (!used[i]) { const uint64_t d = kmodes_dissim(x + far * 80, x + i * 80); if (d < mind[i]) mind[i] = d; }
    }
    /* initial assignment and modes, 978-1011 */
    memset(s.freq, 0, sizeof(int32_t) * (size_t)k * 80 * nmod);
    for (int c = -1; c < k; c++) s.members[c] = 0;
    for (int64_t i = 0; i < n; i++) {
      uint64_t d;
      s.memb[i] = kmodes_argmin(s.cent, k, x + i * 80, &d);
      s.members[s.memb[i]]++;
      for (int a = 0; a < 80; a++) s.freq[((size_t)s.memb[i] * 80 + a) * nmod + x[i * 80 + a]]++;
    }
    for (int c = 0; c < k; c++) {
      if (s.members[c] == 0) { for (int a = 0; a < 80; a++) s.cent[(size_t)c * 80 + a] = x[(int64_t)kmodes_randint((uint32_t)n, &seed) * 80 + a]; }
      else for (int a = 0; a < 80; a++) s.cent[(size_t)c * 80 + a] = (uint8_t)kmodes_maxidx(s.freq + ((size_t)c * 80 + a) * nmod, nmod);
    }
    int itr = 0, converged = 0, worse = 0, bestitr = 0;
    uint64_t prevcost = UINT64_MAX, bestcost = UINT64_MAX;
    while (itr < max_iter && !converged) {
      itr++;
      /* KModesIter, 851-921 */
      int moves = 0;
      uint64_t cost = 0;
      for (int64_t b0 = 0; b0 < n; b0 += 960) {
        const int64_t b1 = b0 + 960 < n ? b0 + 960 : n;
        for (int64_t i = b0; i < b1; i++) clust[i] = kmodes_argmin(s.cent, k, x + i * 80, &dis[i]);
        for (int64_t i = b0; i < b1; i++) {
          cost += dis[i];
          if (s.memb[i] != clust[i]) {
            moves++;
            const int old = s.memb[i];
            kmodes_move(&s, i, clust[i], old);
            if (s.members[old] == 0) { /* CountClusterMembers(old_clust) = 0 */
              int from = 0;
              int64_t mc = 0;
              for (int c = 0; c < k; c++) if (s.members[c] >= mc) { mc = s.members[c]; from = c; } /* GetMaxClusterMembers: last largest */
              const uint32_t pick = kmodes_randint((uint32_t)mc, &seed);
              int64_t r = -1, cnt = 0;
              for (int64_t j = 0; j < n; j++) if (s.memb[j] == from) { if (cnt == (int64_t)pick) { r = j; break; } cnt++; }
              kmodes_move(&s, r, old, from);
            }
          }
        }
      }
      converged = cost >= prevcost;
      if (converged) { /* SameValue(cost, prevcost, prevcost div 1000) in floating point, 1041 */
        const double a = (double)cost, b = (double)prevcost;
        double eps = (double)(prevcost / 1000);
        if (eps == 0) { const double m = fabs(a) < fabs(b) ? fabs(a) : fabs(b); eps = m * 1e-12 > 1e-12 ? m * 1e-12 : 1e-12; }
        const int same = a > b ? (a - b) <= eps : (b - a) <= eps;
        if (same) { worse++; if (worse < 3) converged = 0; }
      }
      converged = converged || moves == 0;
      if (cost < bestcost) { bestitr = itr; bestcost = cost; memcpy(bestm, s.memb, sizeof(int32_t) * (size_t)n); memcpy(bestc, s.cent, (size_t)k * 80); }
      prevcost = cost;
    }
    if (bestcost < all_best) { /* 1078-1085: the first run with the strictly smallest cost */
      all_best = bestcost;
      all_iters = bestitr;
      memcpy(labels_out, bestm, sizeof(int32_t) * (size_t)n);
      memcpy(cent_out, bestc, (size_t)k * 80);
    }
  }
  if (cost_out) *cost_out = all_best;
  if (iters_out) *iters_out = all_iters;
  free(starts); free(s.memb); free(s.members - 1); free(s.cent); free(s.freq); free(clust); free(bestm); free(dis); free(mind); free(used); free(bestc);
  return k;
}

/* ---------------------------------------------------------------------------------------------------------------
 * A17, the other half: DL3 quantisation (Dennis Lee), dlquant/quantizer.c.  Restated with the types of the Win64
 * (LLP64) build the reference ships for: ulong = 32-bit unsigned (sums wrap at 2^32 as there), slong = 32-bit signed,
 * err = float, sqrtf and the float products and sums in IEEE single, CScale = 1 (:23).  The progress callbacks
 * (progress_init / _update / _end, dllmain.c) only report; the stop button never fires here.
 */
typedef struct {
  uint32_t r, g, b, pixel_count; /* CUBE3, :61-67 */
  float err;
  int32_t cc;
  uint8_t rr, gg, bb;
} tmo_cube3;

static void dl3_setrgb(tmo_cube3 *rec) { /* setrgb, :478-484 */
  const int v = (int)rec->pixel_count, v2 = v >> 1;
  rec->rr = (uint8_t)((rec->r + (uint32_t)v2) / (uint32_t)v);
  rec->gg = (uint8_t)((rec->g + (uint32_t)v2) / (uint32_t)v);
  rec->bb = (uint8_t)((rec->b + (uint32_t)v2) / (uint32_t)v);
}

static float dl3_calc_err(const tmo_cube3 *t, int c1, int c2) { /* calc_err, :520-541 */
  const uint32_t P1 = t[c1].pixel_count, P2 = t[c2].pixel_count, P3 = P1 + P2;
  const int R3 = (int)((t[c1].r + t[c2].r + (P3 >> 1)) / P3), G3 = (int)((t[c1].g + t[c2].g + (P3 >> 1)) / P3),
            B3 = (int)((t[c1].b + t[c2].b + (P3 >> 1)) / P3);
  const int R1 = t[c1].rr, G1 = t[c1].gg, B1 = t[c1].bb, R2 = t[c2].rr, G2 = t[c2].gg, B2 = t[c2].bb;
  /* squares3[] holds floats of i*i (:472): the sums below are exact in single (< 2^24) */
  float dist1 = (float)((R3 - R1) * (R3 - R1)) + (float)((G3 - G1) * (G3 - G1)) + (float)((B3 - B1) * (B3 - B1));
  dist1 = sqrtf(dist1) * (float)P1;
  float dist2 = (float)((R2 - R3) * (R2 - R3)) + (float)((G2 - G3) * (G2 - G3)) + (float)((B2 - B3) * (B2 - B3));
  dist2 = sqrtf(dist2) * (float)P2;
  return dist1 + dist2;
}

static void dl3_recount_next(tmo_cube3 *t, int tot, int i) { /* recount_next, :543-559: the first j > i of least error */
  int c2 = 0;
  float err = HUGE_VALF;
  for (int j = i + 1; j < tot; j++) {
    const float cur = dl3_calc_err(t, i, j);
    if (cur < err) { err = cur; c2 = j; }
  }
  t[i].err = err;
  t[i].cc = c2;
}

static void dl3_recount_dist(tmo_cube3 *t, int tot, int c1) { /* recount_dist, :561-581 */
  dl3_recount_next(t, tot, c1);
  for (int i = 0; i < c1; i++) {
    if (t[i].cc == c1) dl3_recount_next(t, tot, i);
    else {
      const float cur = dl3_calc_err(t, i, c1);
      if (cur < t[i].err) { t[i].err = cur; t[i].cc = c1; }
    }
  }
}

int tmo_dl3quant(const uint8_t *rgb, int64_t npixels, int quant_to, int lookup_bpc, uint8_t *pal_out) {
  if (!rgb || !pal_out || npixels <= 0 || quant_to < 1 || lookup_bpc < 1 || lookup_bpc > 8) return -1;
  const int64_t lookup_size = (int64_t)1 << (lookup_bpc * 3); /* :441 */
  tmo_cube3 *t = (tmo_cube3 *)calloc((size_t)lookup_size, sizeof(tmo_cube3)); /* init_table, :457-470 */
  if (!t) return -1;
  /* build_table3, :486-518 */
  const int mbpc = (1 << lookup_bpc) - 1;
  for (int64_t i = 0; i < npixels; i++) {
    const uint8_t *px = rgb + i * 3;
    const int r = px[0] * mbpc / 255, g = px[1] * mbpc / 255, b = px[2] * mbpc / 255;
    const int64_t index = (int64_t)b | ((int64_t)g << lookup_bpc) | ((int64_t)r << (lookup_bpc << 1));
    t[index].r += px[0]; /* * CScale = 1 */
    t[index].g += px[1];
    t[index].b += px[2];
    t[index].pixel_count++;
  }
  int tot = 0;
  for (int64_t i = 0; i < lookup_size; i++)
    if (t[i].pixel_count) {
      dl3_setrgb(t + i);
      t[tot++] = t[i];
    }
  /* reduce_table3, :583-648 */
  int i;
  for (i = 0; i < tot - 1; i++) dl3_recount_next(t, tot, i);
  t[i].err = HUGE_VALF;
  t[i].cc = tot;
  while (tot > quant_to) {
    int c1 = 0;
    float err = HUGE_VALF;
    for (i = 0; i < tot; i++)
      if (t[i].err < err) { err = t[i].err; c1 = i; }
    const int c2 = t[c1].cc;
    t[c2].r += t[c1].r;
    t[c2].g += t[c1].g;
    t[c2].b += t[c1].b;
    t[c2].pixel_count += t[c1].pixel_count;
    dl3_setrgb(t + c2);
    tot--;
    t[c1] = t[tot];
    t[tot - 1].err = HUGE_VALF;
    t[tot - 1].cc = tot;
    for (i = 0; i < c1; i++)
      if (t[i].cc == tot) t[i].cc = c1;
    for (i = c1 + 1; i < tot; i++)
      if (t[i].cc == tot) dl3_recount_next(t, tot, i);
    dl3_recount_dist(t, tot, c1);
    if (c2 != tot) dl3_recount_dist(t, tot, c2);
  }
  /* set_palette3 (:650-664) + copy_pal: planar R, G, B */
  for (i = 0; i < tot; i++) {
    pal_out[i] = t[i].rr;
    pal_out[quant_to + i] = t[i].gg;
    pal_out[2 * quant_to + i] = t[i].bb;
  }
  free(t);
  return tot;
}
