/*
 * tm_oracle.h -- CPU restatement ("oracle") of the TileMotion encoder's per-frame tile pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped path (tiler_amd/, libtilemotion.so) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * Every function cites the reference file:line it restates (paths relative to the gligli/tiler tree).
 * Pinning status (see DESIGN.md "Oracle"):
 *   - colour conversions + DCT LUT/weights/zig-zag: pinned by the properties of TTilingEncoder.Test
 *     (tilingencoder.pas:3847-3902), checked in tests/test_oracle_pins.py;
 *   - everything downstream of the binary-only DLLs (ANN tie order, yakmo, BICO): PARITY UNPINNED --
 *     the reference holds no vectors for them; the build's own deterministic rules are stated here.
 *
 * FreePascal semantics preserved: Round = half-to-even; div truncates toward zero; TFloat = Single;
 * mixed int/single or single/double expressions evaluate in double and narrow once on assignment.
 */
#ifndef TM_ORACLE_H
#define TM_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { TMO_TILE_W = 8, TMO_TILE_PX = 64, TMO_CPNS = 3, TMO_DCT = 192 };

/* TPsyVisMode, tilingencoder.pas:21 */
enum { TMO_PVS_DCT = 0, TMO_PVS_WEIGHTED_DCT = 1, TMO_PVS_WAVELETS = 2, TMO_PVS_SPE_DCT = 3, TMO_PVS_WEIGHTED_SPE_DCT = 4 };

#define TMO_NULL_COLOR ((int32_t)0xffff00ff) /* cDitheringNullColor, utils.pas:45 */

/* ---- tables (utils.pas:47-109) ---- */
extern const uint8_t tmo_dithering_map[64];
extern const uint8_t tmo_dct_snake[64];
extern const double tmo_dct_weights[3][8][8];
const float *tmo_dct_lut_f32(int special);   /* FDCTLut, tilingencoder.pas:1703-1714 */
const double *tmo_dct_lut_f64(int special);  /* FDCTLutDouble */
const double *tmo_inv_dct_lut_f64(void);     /* FInvDCTLutDouble, tilingencoder.pas:1718-1726 */
const float *tmo_srgb_lut_f32(void);         /* inverse-sRGB of c/255 as the Single the reference stores, utils.pas:378-384 */

/* ---- colour (utils.pas:238-509) ---- */
uint32_t tmo_swap_rb(uint32_t c);
void tmo_rgb_to_yuv(int r, int g, int b, float *y, float *u, float *v);
int32_t tmo_yuv_to_rgb(float y, float u, float v);
void tmo_rgb_to_lab(int r, int g, int b, float *ol, float *oa, float *ob);      /* libm pow: for the Test pin */
void tmo_rgb_to_lab_det(int r, int g, int b, float *ol, float *oa, float *ob);  /* deterministic cbrt (build rule) */
int32_t tmo_lab_to_rgb(float l, float a, float b);
void tmo_rgb_to_hsv(uint32_t col, uint8_t *h, uint8_t *s, uint8_t *v);
double tmo_cbrt_det(double x);
void tmo_rgb_to_lab_array(const uint32_t *rgb, int64_t n, int det, float *out); /* n colours 0x00RRGGBB -> [n][3] */
void tmo_rgb_to_lab_fast(int r, int g, int b, float *ol, float *oa, float *ob);  /* the kernels' division-free form of _det */
void tmo_lab_domain_check(int64_t *out);  /* all 2^24 colours: [0] det != libm pow, [1] fast != det */

/* ---- A1-A3 load side ---- */
void tmo_load_from_image(const uint32_t *img, int img_w, int img_h, int tm_w, int tm_h, uint32_t *tiles);
void tmo_inter_frame_data(const uint32_t *tiles, int ntiles, float *out3);
float tmo_pearson(const float *x, const float *y, int n);
void tmo_mirror_heuristics(const uint32_t *tile, int *hmirror, int *vmirror);
void tmo_hmirror_u32(uint32_t *tile);
void tmo_vmirror_u32(uint32_t *tile);
void tmo_hmirror_u8(uint8_t *tile);
void tmo_vmirror_u8(uint8_t *tile);
/* whole AsyncLoadFromImage mirror pass: canonicalises tiles in place, writes flags bit0=H bit1=V */
void tmo_canonicalise_tiles(uint32_t *tiles, int ntiles, uint8_t *flags);
/* FindKeyFrames, tilingencoder.pas:3361-3433 (automatic mode). is_kf[nframes] out; returns keyframe count */
int tmo_find_keyframes(const float *correl, int nframes, double fps, double max_s, double min_s, double lo_thres, uint8_t *is_kf);

/* ---- A4-A6 features ---- */
void tmo_cpn_from_rgb(const uint32_t *rgb, int use_lab, int hmirror, int vmirror, float cpn[192]);
void tmo_cpn_from_pal(const uint8_t *pal_px, const int32_t *palette, int use_lab, int hmirror, int vmirror, float cpn[192]);
void tmo_features_i16(const float cpn[192], int mode, int16_t out[192]);
void tmo_features_f64(const float cpn[192], int mode, double out[192]);
void tmo_inv_features_f64(const double dct[192], int mode, int use_lab, uint32_t rgb_out[64]);
/* batched helpers (what the HIP kernels are compared against) */
void tmo_tiles_features_i16(const uint32_t *tiles, int n, const uint8_t *mirror_flags, int mode, int use_lab, int16_t *out);
void tmo_paltiles_features_i16(const uint8_t *pal_px, const int32_t *pal_idx, int n, const int32_t *palettes, int pal_size,
                               int mode, int16_t *out);
/* clustering features of the build: A6 double DCT (UseLAB, deterministic Lab), Round()ed to int32 */
void tmo_tiles_features_cluster_i32(const uint32_t *tiles, int n, int mode, int32_t *out);

/* ---- A15 distances ---- */
uint32_t tmo_ssd_i16(const int16_t *a, const int16_t *b);               /* utils.pas:541-557 */
uint32_t tmo_ssd_i16_sse_quirk(const int16_t *a, const int16_t *b);      /* utils.pas:559-725 with xmm7_in = 0 */
float tmo_euclidean_to_psnr(uint32_t e);                                  /* utils.pas:1074-1078 */

/* ---- A13/A14 KNN (exact brute force, lowest index wins ties) ---- */
void tmo_knn1(const int16_t *queries, int64_t nq, const int16_t *db, int64_t nt, int32_t *idx, uint32_t *err);
/* k smallest by (err asc, idx asc) */
/* exact kd-tree (ANN's published standard split + standard search, eps 0): what the reference's CPU path searches with */
typedef struct tmo_kdtree tmo_kdtree;
tmo_kdtree *tmo_kdtree_build(const int16_t *db, int64_t n, int bucket);
void tmo_kdtree_free(tmo_kdtree *t);
int64_t tmo_kdtree_search1(const tmo_kdtree *t, const int16_t *queries, int64_t nq, int32_t *idx, uint32_t *err);
void tmo_knnk(const int16_t *queries, int64_t nq, const int16_t *db, int64_t nt, int k, int32_t *idx, uint32_t *err);

/* ---- generic QuickSort, extern.pas:370-418 ---- */
typedef int (*tmo_cmp_fn)(const void *a, const void *b, void *user);
void tmo_quicksort(void *data, int64_t first, int64_t last, int item_size, tmo_cmp_fn cmp, void *user);

/* ---- A12 dithering ---- */
typedef struct {
  int count;            /* live entries */
  int32_t luma[64];     /* LumaPal */
  int32_t y2[64][4];    /* Y2Palette r,g,b,luma div 1000 */
  uint8_t remap[64];
  int y2_mixed_colors;
} tmo_plan;
void tmo_prepare_plan(tmo_plan *plan, const int32_t *pal, int pal_size, int y2_mixed_colors);
int64_t tmo_color_compare(int64_t r1, int64_t g1, int64_t b1, int64_t r2, int64_t g2, int64_t b2);
void tmo_mixing_plan_tk(const tmo_plan *plan, uint32_t col, uint8_t list[64]);
int tmo_mixing_plan_yliluoma(const tmo_plan *plan, uint32_t col, uint8_t list[256]);
/* DitherTile: tile given in canonical (mirrored) orientation with its initial mirror flags */
void tmo_dither_tile(const uint32_t *rgb_canon, int hmirror, int vmirror, const tmo_plan *plan, int use_tk, uint8_t pal_out[64]);
void tmo_dither_tiles(const uint32_t *tiles, const uint8_t *flags, const int32_t *pal_idx, int64_t n, const int32_t *palettes,
                      int pal_size, int use_tk, int y2_mixed, uint8_t *pal_out);

/* ---- A8/A16 exact dedup + reindex ---- */
/* keys: n rows of key_dwords uint32 (RGB: 64) or key_bytes (pal: 64 bytes passed as 16 dwords is NOT equivalent:
 * CompareByte order differs) -> two entry points. use_in may be NULL (all 1).
 * Outputs: rep[n] = index of the representative (lowest original index of the equal run),
 *          order[nu] = representatives sorted by (use desc, content asc), use_out[nu], remap[n] = final index.
 * Returns nu = number of unique rows. */
int64_t tmo_dedup_u32(const uint32_t *keys, int64_t n, int key_dwords, const uint32_t *use_in,
                      int64_t *rep, int64_t *order, uint32_t *use_out, int64_t *remap);
int64_t tmo_dedup_u8(const uint8_t *keys, int64_t n, int key_bytes, const uint32_t *use_in,
                     int64_t *rep, int64_t *order, uint32_t *use_out, int64_t *remap);
int tmo_equal_quality_tile_count(double tile_count); /* utils.pas:1038-1041 */

/* ---- A9/A10 k-means of the build (deterministic; replaces BICO+ANN+yakmo, documented deviation) ---- */
/* points int32 [n][d], weights u32[n] (NULL = 1). Farthest-first init from point 0 (cf. kmodes.pas:694),
 * Lloyd with exact integer sums, double centroids. Returns number of live centroids (<= k). */
int tmo_kmeans_i32(const int32_t *pts, const uint32_t *w, int64_t n, int d, int k, int max_iter,
                   int32_t *assign, double *centroids, int *iters_out);
/* QuantizeUsingYakmo restated on the (G,R,B)-sorted unique-colour histogram of the given pixels;
 * writes pal_size colours (unused = TMO_NULL_COLOR) ordered by (Val,Sat,Hue). tilingencoder.pas:4434-4564 */
void tmo_quantize_palette(const uint32_t *pixels, int64_t npx, int pal_size, int max_iter, int32_t *palette_out);
/* tile->palette assignment + ranking by use count (tilingencoder.pas:4221-4244) */
/* the build's deterministic D^2 seeding (tile -> palette clustering) and the k-means run from it */
#define TMO_PP_SEED 0x42381337ull
uint64_t tmo_pp_next(uint64_t *state);
int tmo_kmeans_pp_seeds(const int32_t *pts, const uint32_t *w, int64_t n, int d, int k, int64_t *seeds_out);
int tmo_kmeans_pp_i32(const int32_t *pts, const uint32_t *w, int64_t n, int d, int k, int max_iter, int32_t *assign, double *cent, int *iters_out);
void tmo_palettize_tiles(const int32_t *feat, const uint32_t *use, int64_t n, int pal_count, int max_iter, int32_t *pal_idx_out);

/* ---- A11 OptimizePalettes (tilingencoder.pas:4309-4432) with Powell/Brent (powell.pas); in place, returns sweeps ---- */
int tmo_optimize_palettes(int32_t *palettes, int pal_count, int pal_size);
/* test hooks: Bracket (powell.pas:56-147) and Brent (:149-266) on fixed scalar functions, evaluations counted; held to a fixture recorded
 * from scipy.optimize (tests/golden/make_powell_fixtures.py) */
double tmo_test_scalar_fn(int id, double x);
int tmo_test_bracket(int fn, double xa, double xb, double out[4]);
int tmo_test_brent(int fn, double xtol, int maxiter, double out[3]);

/* ---- (f)#1 motion prediction (tilingencoder.pas:1154-1282, 1496-1532) and the Reduce threshold search (4014-4046) ---- */
void tmo_window_dcts(const uint32_t *fb, int w, int h, int16_t *out /* [(h-7)*(w-7)][192] */);
void tmo_motion_search(const int16_t *cur /* [tm_h*tm_w][192] */, int tm_w, int tm_h, const int16_t *win, int radius,
                       uint32_t *best_err, int8_t *px, int8_t *py);
double tmo_solve_tile_count(const double *sorted_min_psnr, int64_t ngroups, double target, int *probes);

/* ---- (f)#3 FrameTilingExtendedPaletteUsage re-rank (tilingencoder.pas:1559-1610); knn_idx [nq][k] from tmo_knnk ---- */
void tmo_epu_rerank(const int16_t *q, const int32_t *knn_idx, int k, const uint8_t *pal_px, const int32_t *tile_pal_idx, int64_t ntiles,
                    const int32_t *palettes, int pal_size, int32_t *out_tile, int32_t *out_pal, uint32_t *out_err);
void tmo_epu_rerank_batch(const int16_t *q, int64_t nq, const int32_t *knn_idx, int k, const uint8_t *pal_px, const int32_t *tile_pal_idx,
                          int64_t ntiles, const int32_t *palettes, int pal_size, int32_t *out_tile, int32_t *out_pal, uint32_t *out_err);

/* ---- (f)#2 checker: LZMA-alone decoder (decoders/htmljs/lzma.js:395-576).  Returns the decoded size or -1;
 * props_out (may be NULL) = {props byte, dictionary size, header size field or -1}. ---- */
int64_t tmo_lzma_decode(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *consumed, int *props_out);

/* A17: TKModes.ComputeKModes (kmodes.pas:923-1094) on rows of 80 bytes; labels 0-based as the code returns them; returns the number of centroids */
int tmo_kmodes(const uint8_t *rows, int64_t n, int num_clusters, int num_init, int num_modalities, int max_iter, int32_t *labels_out,
               uint8_t *centroids_out, uint64_t *cost_out, int *iters_out);

/* A17, the other half: dl3quant (dlquant/quantizer.c:437-455) = build_table3 (:486-518: histogram at lookup_bpc bits per channel, entries
 * compacted in index order) + reduce_table3 (:583-648: greedy merging of the pair with the least calc_err, :520-541) + set_palette3
 * (:650-664).  `rgb` = npixels x (R, G, B) bytes; pal_out = [3][quant_to] planar like userpal; returns the number of colours left
 * (<= quant_to), or -1 on bad arguments.  LLP64 types as the Win64 build: ulong = uint32, slong = int32.  PARITY UNPINNED: the
 * reference holds no output of it and its build cannot be made here (Windows.h progress callbacks). */
int tmo_dl3quant(const uint8_t *rgb, int64_t npixels, int quant_to, int lookup_bpc, uint8_t *pal_out);

#ifdef __cplusplus
}
#endif

#endif
