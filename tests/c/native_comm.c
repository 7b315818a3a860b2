/* native_comm.c -- a plain C host (what a FreePascal host would do) driving libtilemotion through its
 * NATIVE multi-process path: tm_comm_unique_id / tm_comm_init put RCCL, linked into the library, behind Run(step); no callback,
 * no Python.  Test infrastructure: built and started as a fresh child process by tests/test_gpu_native_comm.py.
 *
 *   native_comm single <out>            one encoder, no communicator                     -> result dump in <out>
 *   native_comm rank <r> <world> <idfile> <out>
 *                                        rank r of `world` processes; rank 0 writes the 128-byte id to <idfile>, the others wait
 *                                        for it; with world = 1 the sharded paths still run (TM_COMM_FORCE_DIST=1 in the environment)
 * The dump holds every global tile (header, palette indices, RGB), the palettes and all tile maps: the caller compares dumps byte
 * for byte ("every process ends each step with the same global tiles, palettes and merged tile maps as a single-process run").
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "tilemotion.h"

#define W 96
#define H 72
#define F 24

#define CHECK(call)                                                                   \
  do {                                                                                \
    const int rc_ = (call);                                                           \
    if (rc_ != TM_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, tm_last_error()); return 1; } \
  } while (0)

/* a small clip with exact duplicates, near duplicates and a scene cut: gradients that drift, noise from a 64-bit LCG on a quarter
 * of the tiles, every third tile column static */
static void make_clip(uint32_t *px) {
  uint64_t lcg = 0x42381337ull;
  for (int f = 0; f < F; f++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const int tx = x >> 3, ty = y >> 3;
        const int drift = (tx % 3 == 0) ? 0 : f;
        int r = (x * 255 / W + 2 * drift) & 255, g = (y * 255 / H + drift) & 255, b = ((x + y) * 255 / (W + H) + 3 * drift) & 255;
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        if (((tx * 7 + ty * 13 + f * 5) & 3) == 0) {
          r += (int)((lcg >> 33) % 17) - 8; g += (int)((lcg >> 41) % 17) - 8; b += (int)((lcg >> 49) % 17) - 8;
          r = r < 0 ? 0 : r > 255 ? 255 : r; g = g < 0 ? 0 : g > 255 ? 255 : g; b = b < 0 ? 0 : b > 255 ? 255 : b;
        }
        if (f >= F / 2) { const int t = r; r = g; g = b; b = t; }
        px[((size_t)f * H + y) * W + x] = 0xFF000000u | ((uint32_t)r << 16) | ((uint32_t)g << 8) | (uint32_t)b;
      }
}

static int dump(tm_encoder *e, const char *path) {
  int64_t tiles = 0;
  int frames = 0, palettes = 0, tw = 0, th = 0, kfs = 0;
  CHECK(tm_get_counts(e, &tiles, &frames, &palettes, &tw, &th, &kfs));
  FILE *o = fopen(path, "wb");
  if (!o) { perror(path); return 1; }
  fwrite(&tiles, 8, 1, o); fwrite(&frames, 4, 1, o); fwrite(&palettes, 4, 1, o); fwrite(&kfs, 4, 1, o);
  tm_tile_hdr *hdr = malloc((size_t)tiles * sizeof(tm_tile_hdr));
  uint8_t *pal = malloc((size_t)tiles * 64);
  uint32_t *rgb = malloc((size_t)tiles * 256);
  CHECK(tm_get_tiles(e, 0, tiles, hdr, pal, rgb));
  fwrite(hdr, sizeof(tm_tile_hdr), (size_t)tiles, o); fwrite(pal, 64, (size_t)tiles, o); fwrite(rgb, 256, (size_t)tiles, o);
  int64_t psz = 0;
  CHECK(tm_get_int(e, "PaletteSize", &psz));
  int32_t *pc = malloc((size_t)psz * 4);
  for (int p = 0; p < palettes; p++) { CHECK(tm_get_palette(e, p, pc)); fwrite(pc, 4, (size_t)psz, o); }
  tm_tilemap_item *tm = malloc((size_t)tw * th * sizeof(tm_tilemap_item));
  for (int f = 0; f < frames; f++) { CHECK(tm_get_tilemap(e, f, tm)); fwrite(tm, sizeof(tm_tilemap_item), (size_t)tw * th, o); }
  fclose(o);
  printf("%lld tiles, %d palettes, %d key frames\n", (long long)tiles, palettes, kfs);
  free(hdr); free(pal); free(rgb); free(pc); free(tm);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: native_comm single <out> | rank <r> <world> <idfile> <out>\n"); return 2; }
  const int single = strcmp(argv[1], "single") == 0;
  if (!single && argc < 6) return 2;
  const int rank = single ? 0 : atoi(argv[2]), world = single ? 1 : atoi(argv[3]);
  const char *out = single ? argv[2] : argv[5];
  if (tm_device_count() <= 0) { fprintf(stderr, "no device: %s\n", tm_last_error()); return 3; }
  tm_encoder *e = tm_create();
  if (!e) { fprintf(stderr, "tm_create: %s\n", tm_last_error()); return 1; }
  CHECK(tm_set_device(e, rank % tm_device_count()));
  CHECK(tm_load_default_settings(e));
  CHECK(tm_set_int(e, "PaletteCount", 3));
  CHECK(tm_set_int(e, "MotionPredictRadius", 0));
  CHECK(tm_set_bool(e, "FrameTilingExtendedPaletteUsage", 0));
  CHECK(tm_set_float(e, "ShotTransMinSecondsPerKF", 0.1));
  CHECK(tm_set_video(e, W, H, 24.0, F));
  if (!single) {
    uint8_t id[TM_COMM_ID_BYTES];
    if (rank == 0) {
      CHECK(tm_comm_unique_id(id));
      char tmp[1024];
      snprintf(tmp, sizeof tmp, "%s.tmp", argv[4]);
      FILE *o = fopen(tmp, "wb");
      if (!o || fwrite(id, 1, sizeof id, o) != sizeof id) { perror(tmp); return 1; }
      fclose(o);
      if (rename(tmp, argv[4]) != 0) { perror("rename"); return 1; }
    } else {
      FILE *i = NULL;
      for (int t = 0; t < 600 && !(i = fopen(argv[4], "rb")); t++) usleep(100000);  /* at most a minute */
      if (!i || fread(id, 1, sizeof id, i) != sizeof id) { fprintf(stderr, "no communicator id in %s\n", argv[4]); return 1; }
      fclose(i);
    }
    CHECK(tm_comm_init(e, id, rank, world));
    /* Reconstruct matches this process's frames (the query shard of tm_set_query_shard; the other steps shard by themselves) */
    const int f0 = F * rank / world, f1 = F * (rank + 1) / world;
    CHECK(tm_set_query_shard(e, f0, f1 - f0));
  }
  uint32_t *px = malloc((size_t)F * H * W * 4);
  make_clip(px);
  for (int f = 0; f < F; f++) CHECK(tm_push_frame_rgb32(e, f, px + (size_t)f * H * W, W));
  CHECK(tm_run(e, TM_STEP_ALL));
  if (dump(e, out)) return 1;
  if (!single) CHECK(tm_comm_destroy(e));
  tm_destroy(e);
  free(px);
  return 0;
}
