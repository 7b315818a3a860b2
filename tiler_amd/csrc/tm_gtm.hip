// tm_gtm.hip -- (f)#2: the .gtm bitstream writer, host code only.
//
// Restates TTilingEncoder.SaveStream (tilingencoder.pas:5177-5482): 40-byte GTMv header + 28-byte GTMk per keyframe
// (30-51), then per keyframe one LZMA-alone-like stream (LZCompress, extern.pas:420-439: props byte 0x62 = lc 8 / lp 0
// / pb 2, 4 MiB dictionary, eight 0xFF size bytes, end marker) of 16-bit commands (data << 4 | cmd; 53-86, 5200-5206):
// ExtendedCommand(settings) + SetDimensions + TileSet(tiles with UseCount > 1) + LoadPalette x P in the first
// keyframe, then per frame tile-map items / SkipBlocks (5392-5437) and FrameEnd.
// The LZMA encoder is our own (hash-chain greedy parse: literals, matches, rep0 matches); any valid LZMA stream with
// these properties decodes in the reference's decoders/htmljs/lzma.js -- byte equality with the Pascal LZMA SDK port's
// optimal parser is not a goal, only the properties and the decoded command stream are.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {
namespace {

// ---------------------------------------------------------------------------------------------------------------
class RangeEncoder {
 public:
  explicit RangeEncoder(std::vector<uint8_t> &out) : out_(out) {}
  void bit(uint16_t &prob, int b) {
    const uint32_t bound = (range_ >> 11) * prob;
    if (!b) { range_ = bound; prob = (uint16_t)(prob + ((2048 - prob) >> 5)); }
    else { low_ += bound; range_ -= bound; prob = (uint16_t)(prob - (prob >> 5)); }
    while (range_ < (1u << 24)) { range_ <<= 8; shift_low(); }
  }
  void direct(uint32_t v, int nbits) {
    for (int i = nbits - 1; i >= 0; i--) {
      range_ >>= 1;
      if ((v >> i) & 1) low_ += range_;
      while (range_ < (1u << 24)) { range_ <<= 8; shift_low(); }
    }
  }
  void flush() { for (int i = 0; i < 5; i++) shift_low(); }

 private:
  void shift_low() {
    if ((uint32_t)low_ < 0xFF000000u || (low_ >> 32) != 0) {
      uint8_t temp = cache_;
      do { out_.push_back((uint8_t)(temp + (uint8_t)(low_ >> 32))); temp = 0xFF; } while (--cache_size_ != 0);
      cache_ = (uint8_t)((uint32_t)low_ >> 24);
    }
    cache_size_++;
    low_ = (low_ & 0x00FFFFFFu) << 8;
  }
  std::vector<uint8_t> &out_;
  uint64_t low_ = 0, cache_size_ = 1;
  uint32_t range_ = 0xFFFFFFFFu;
  uint8_t cache_ = 0;
};

struct LenCoder {
  uint16_t choice = 1024, choice2 = 1024, low[16][8], mid[16][8], high[256];
  LenCoder() {
    for (auto &r : low) for (auto &p : r) p = 1024;
    for (auto &r : mid) for (auto &p : r) p = 1024;
    for (auto &p : high) p = 1024;
  }
};

class LzmaEncoder {  // lc/lp/pb as LZCompress sets them: SetLcLpPb(8,0,2), end marker on (extern.pas:429-430)
 public:
  static constexpr int kLc = 8, kLp = 0, kPb = 2;
  static constexpr uint32_t kDict = 1u << 22;

  LzmaEncoder(std::vector<uint8_t> &out) : rc_(out), lit_((size_t)0x300 << (kLc + kLp), 1024) {
    for (auto &r : is_match_) for (auto &p : r) p = 1024;
    for (auto &r : is_rep0_long_) for (auto &p : r) p = 1024;
    for (auto &p : is_rep_) p = 1024;
    for (auto &p : is_rep_g0_) p = 1024;
    for (auto &p : is_rep_g1_) p = 1024;
    for (auto &p : is_rep_g2_) p = 1024;
    for (auto &r : pos_slot_) for (auto &p : r) p = 1024;
    for (auto &p : pos_special_) p = 1024;
    for (auto &p : pos_align_) p = 1024;
  }

  void encode(const uint8_t *src, size_t n) {
    std::vector<int32_t> head(1 << 16, -1), prev(n ? n : 1, -1);
    auto hash3 = [&](size_t i) { return (uint32_t)((src[i] | (src[i + 1] << 8) | (src[i + 2] << 16)) * 2654435761u) >> 16; };
    size_t pos = 0;
    while (pos < n) {
      const uint32_t pos_state = (uint32_t)pos & ((1u << kPb) - 1);
      // candidates: rep0, then the hash chain
      size_t best_len = 0, best_dist = 0, rep_len = 0;
      const size_t max_len = std::min<size_t>(273, n - pos);
      if (pos > rep_[0]) {
        const uint8_t *a = src + pos, *b = src + pos - rep_[0] - 1;
        while (rep_len < max_len && a[rep_len] == b[rep_len]) rep_len++;
      }
      if (pos + 3 <= n) {
        const uint32_t h = hash3(pos);
        int32_t cand = head[h];
        for (int depth = 0; cand >= 0 && depth < 32; depth++, cand = prev[cand]) {
          const size_t dist = pos - (size_t)cand;  // >= 1
          if (dist > kDict) break;
          const uint8_t *a = src + pos, *b = src + cand;
          size_t l = 0;
          while (l < max_len && a[l] == b[l]) l++;
          if (l > best_len) { best_len = l; best_dist = dist - 1; if (l == max_len) break; }
        }
      }
      size_t step;
      if (rep_len >= 2 && rep_len + 1 >= best_len) {  // rep0 long match: cheapest way to say "same distance again"
        rc_.bit(is_match_[state_][pos_state], 1);
        rc_.bit(is_rep_[state_], 1);
        rc_.bit(is_rep_g0_[state_], 0);
        rc_.bit(is_rep0_long_[state_][pos_state], 1);
        encode_len(rep_len_, (uint32_t)rep_len, pos_state);
        state_ = state_ < 7 ? 8 : 11;
        step = rep_len;
      } else if (best_len >= 3 || (best_len == 2 && best_dist < 128)) {
        encode_match((uint32_t)best_dist, (uint32_t)best_len, pos_state);
        step = best_len;
      } else {
        encode_literal(src, pos, pos_state);
        step = 1;
      }
      for (size_t k = 0; k < step; k++) {  // index every position we pass
        const size_t i = pos + k;
        if (i + 3 <= n) { const uint32_t h = hash3(i); prev[i] = head[h]; head[h] = (int32_t)i; }
      }
      pos += step;
    }
    // end marker: a match with distance 0xFFFFFFFF and the minimum length
    encode_match(0xFFFFFFFFu, 2, (uint32_t)pos & ((1u << kPb) - 1));
    rc_.flush();
  }

 private:
  void tree(uint16_t *probs, int nbits, uint32_t sym) {
    uint32_t m = 1;
    for (int i = nbits - 1; i >= 0; i--) { const int b = (sym >> i) & 1; rc_.bit(probs[m], b); m = (m << 1) | b; }
  }
  void rtree(uint16_t *probs, int nbits, uint32_t sym) {
    uint32_t m = 1;
    for (int i = 0; i < nbits; i++) { const int b = sym & 1; rc_.bit(probs[m], b); m = (m << 1) | b; sym >>= 1; }
  }
  void encode_len(LenCoder &lc, uint32_t len, uint32_t pos_state) {
    len -= 2;
    if (len < 8) { rc_.bit(lc.choice, 0); tree(lc.low[pos_state], 3, len); }
    else if (len < 16) { rc_.bit(lc.choice, 1); rc_.bit(lc.choice2, 0); tree(lc.mid[pos_state], 3, len - 8); }
    else { rc_.bit(lc.choice, 1); rc_.bit(lc.choice2, 1); tree(lc.high, 8, len - 16); }
  }
  void encode_literal(const uint8_t *src, size_t pos, uint32_t pos_state) {
    rc_.bit(is_match_[state_][pos_state], 0);
    const uint8_t prev_byte = pos ? src[pos - 1] : 0;
    uint16_t *probs = &lit_[(size_t)0x300 * ((((uint32_t)pos & ((1u << kLp) - 1)) << kLc) + (prev_byte >> (8 - kLc)))];
    const uint32_t cur = src[pos];
    if (state_ < 7) {
      tree(probs, 8, cur);
    } else {  // after a match the literal is coded against the byte the last distance points at
      uint32_t match_byte = src[pos - rep_[0] - 1], offs = 0x100, symbol = cur | 0x100;
      do {
        match_byte <<= 1;
        rc_.bit(probs[offs + (match_byte & offs) + (symbol >> 8)], (symbol >> 7) & 1);
        symbol <<= 1;
        offs &= ~(match_byte ^ symbol);
      } while (symbol < 0x10000);
    }
    state_ = state_ < 4 ? 0 : (state_ < 10 ? state_ - 3 : state_ - 6);
  }
  static uint32_t pos_slot_of(uint32_t dist) {
    if (dist < 4) return dist;
    const int n = 31 - __builtin_clz(dist);
    return (uint32_t)(2 * n) + ((dist >> (n - 1)) & 1);
  }
  void encode_match(uint32_t dist, uint32_t len, uint32_t pos_state) {
    rc_.bit(is_match_[state_][pos_state], 1);
    rc_.bit(is_rep_[state_], 0);
    encode_len(len_, len, pos_state);
    state_ = state_ < 7 ? 7 : 10;
    const uint32_t slot = pos_slot_of(dist);
    tree(pos_slot_[std::min<uint32_t>(len - 2, 3)], 6, slot);
    if (slot >= 4) {
      const int footer = (int)(slot >> 1) - 1;
      const uint32_t base = (2u | (slot & 1)) << footer, reduced = dist - base;
      if (slot < 14) rtree(pos_special_ + ((int)base - (int)slot - 1), footer, reduced);  // index 0 is reached with m = 1
      else { rc_.direct(reduced >> 4, footer - 4); rtree(pos_align_, 4, reduced & 15); }
    }
    rep_[3] = rep_[2]; rep_[2] = rep_[1]; rep_[1] = rep_[0]; rep_[0] = dist;
  }

  RangeEncoder rc_;
  std::vector<uint16_t> lit_;
  uint16_t is_match_[12][16], is_rep0_long_[12][16], is_rep_[12], is_rep_g0_[12], is_rep_g1_[12], is_rep_g2_[12];
  uint16_t pos_slot_[4][64], pos_special_[128], pos_align_[16];
  LenCoder len_, rep_len_;
  uint32_t state_ = 0;
  size_t rep_[4] = {0, 0, 0, 0};
};

void lz_compress_impl(const std::vector<uint8_t> &raw, std::vector<uint8_t> &dst) {  // LZCompress, extern.pas:420-439
  dst.push_back((uint8_t)((LzmaEncoder::kPb * 5 + LzmaEncoder::kLp) * 9 + LzmaEncoder::kLc));  // 0x62
  for (int i = 0; i < 4; i++) dst.push_back((uint8_t)(LzmaEncoder::kDict >> (8 * i)));
  for (int i = 0; i < 8; i++) dst.push_back(0xFF);
  LzmaEncoder enc(dst);
  enc.encode(raw.data(), raw.size());
}

struct Stream {
  std::vector<uint8_t> b;
  void u8(uint32_t v) { b.push_back((uint8_t)v); }
  void u16(uint32_t v) { u8(v); u8(v >> 8); }
  void u32(uint32_t v) { u16(v); u16(v >> 16); }
  void cmd(int c, uint32_t data) { u16((data << 4) | (uint32_t)c); }  // DoCmd, 5200-5206
};
enum { gtPredShort = 0, gtPredLong = 1, gtShortShort = 2, gtLongShort = 3, gtLongLong = 4, gtIntra = 5, gtSkip = 6, gtFrameEnd = 11,
       gtLoadPalette = 12, gtTileSet = 13, gtSetDimensions = 14, gtExtended = 15 };  // TGTMCommand, 72-86

void put_le32(std::vector<uint8_t> &f, size_t at, uint32_t v) { for (int i = 0; i < 4; i++) f[at + i] = (uint8_t)(v >> (8 * i)); }

}  // namespace

void lz_compress(const std::vector<uint8_t> &raw, std::vector<uint8_t> &dst) { lz_compress_impl(raw, dst); }

int write_gtm(const char *path, const GtmInput &in) {
  TM_CHECK(path && in.tm_w > 0 && in.tm_h > 0 && in.nframes > 0 && in.fps > 0 && !in.kf_start.empty(), TM_E_INVAL, "write_gtm: bad input");
  const int tm_size = in.tm_w * in.tm_h, nkf = (int)in.kf_start.size();
  const int64_t ntiles = (int64_t)in.use.size();
  std::vector<uint8_t> file(40 + 28 * (size_t)nkf, 0);
  memcpy(&file[0], "GTMv", 4);
  put_le32(file, 4, 32);                         // RIFFSize
  put_le32(file, 8, (uint32_t)file.size());      // WholeHeaderSize
  put_le32(file, 12, 4);                         // EncoderVersion (5351)
  put_le32(file, 16, (uint32_t)in.tm_w * 8);
  put_le32(file, 20, (uint32_t)in.tm_h * 8);
  put_le32(file, 24, (uint32_t)nkf);
  put_le32(file, 28, (uint32_t)in.nframes);
  for (int k = 0; k < nkf; k++) {
    const size_t o = 40 + 28 * (size_t)k;
    memcpy(&file[o], "GTMk", 4);
    put_le32(file, o + 4, 20);
    put_le32(file, o + 8, (uint32_t)k);
    put_le32(file, o + 12, (uint32_t)in.kf_start[k]);
    put_le32(file, o + 24, (uint32_t)llrint(1000.0 * in.kf_start[k] / in.fps));
  }
  Stream z;
  // WriteSettings (5331-5335): ExtendedCommand 0 + WriteAnsiString(settings)
  z.cmd(gtExtended, 0);
  z.u32((uint32_t)in.settings.size());
  z.b.insert(z.b.end(), in.settings.begin(), in.settings.end());
  // WriteDimensions (5318-5329)
  z.cmd(gtSetDimensions, 0);
  z.u16((uint32_t)in.tm_w);
  z.u16((uint32_t)in.tm_h);
  z.u32((uint32_t)llrint(1000.0 * 1000 * 1000 / in.fps));
  z.u32((uint32_t)ntiles);
  // WriteTiles (5292-5316): tiles before the first UseCount = 1 (sorted by use, most used first) go into one TileSet
  int64_t reused = 0;
  for (int64_t t = 0; t < ntiles; t++)
    if (in.use[t] == 1) { reused = t; break; }
  if (reused > 0) {
    z.cmd(gtTileSet, (uint32_t)in.pal_size);
    z.u32(0);
    z.u32((uint32_t)(reused - 1));
    z.b.insert(z.b.end(), in.pal_px, in.pal_px + reused * 64);
  }
  // WritePalettes (5270-5290)
  for (int p = 0; p < in.pal_count; p++) {
    z.cmd(gtLoadPalette, 0);
    z.u16((uint32_t)p);
    for (int c = 0; c < in.pal_size; c++) {
      uint32_t col = (uint32_t)in.palettes[(size_t)p * in.pal_size + c];
      if ((int32_t)col == TM_NULL_COLOR) col = 0xffffff;
      z.u32(col | 0xff000000u);
    }
  }
  double avg_bytes = 0;
  uint32_t kf_max = 0;
  int last_kf = 0;
  for (int k = 0; k < nkf; k++) {
    const int f0 = in.kf_start[k], f1 = (k + 1 < nkf ? in.kf_start[k + 1] : in.nframes) - 1;
    for (int f = f0; f <= f1; f++) {
      const tm_tilemap_item *tm = in.tilemap + (size_t)f * tm_size;
      int cs = 0, skip = 0;
      for (int yx = 0; yx < tm_size; yx++) {
        if (skip > 0) { skip--; continue; }
        int run = 0;  // smoothed = predicted with a zero offset (GetIsSmoothed, 621-624)
        for (int s = yx; s < tm_size; s++) {
          const tm_tilemap_item &it = tm[s];
          if (!((it.Flags & 4) && it.PredictedX == 0 && it.PredictedY == 0)) break;
          run++;
        }
        run = std::min(4096, run);
        if (run >= 4) {  // CMinBlkSkipCount
          z.cmd(gtSkip, (uint32_t)(run - 1));
          cs += run;
          skip = run - 1;
          continue;
        }
        const tm_tilemap_item &it = tm[yx];  // DoTMI, 5208-5268
        if (it.Flags & 4) {
          if (it.PredictedX < -32 || it.PredictedX > 31 || it.PredictedY < -32 || it.PredictedY > 31) {
            z.cmd(gtPredLong, 0);
            z.u8((uint8_t)it.PredictedX);
            z.u8((uint8_t)it.PredictedY);
          } else {
            z.cmd(gtPredShort, ((uint8_t)it.PredictedX & 63u) | (((uint8_t)it.PredictedY & 63u) << 6));
          }
        } else {
          const uint32_t tile = (uint32_t)std::max(0, it.TileIdx), pal = (uint32_t)std::max(0, it.PalIdx) & 0xffff;
          const bool intra = (int64_t)tile < ntiles && in.use[tile] <= 1;
          const uint32_t attrs = ((it.Flags & 2) ? 2u : 0u) | ((it.Flags & 1) ? 1u : 0u);
          if (intra) {
            z.cmd(gtIntra, attrs);
            z.u16(pal);
            z.b.insert(z.b.end(), in.pal_px + (size_t)tile * 64, in.pal_px + (size_t)tile * 64 + 64);
          } else if (tile <= 0xffff && pal < 1024) {
            z.cmd(gtShortShort, attrs | (pal << 2));
            z.u16(tile);
          } else if (pal < 1024) {
            z.cmd(gtLongShort, attrs | (pal << 2));
            z.u32(tile);
          } else {
            z.cmd(gtLongLong, attrs);
            z.u16(pal);
            z.u32(tile);
          }
        }
        cs++;
      }
      TM_CHECK(cs == tm_size && skip == 0, TM_E_INVAL, "write_gtm: incomplete tile map (frame %d)", f);
      const bool is_kf_end = f == f1;
      z.cmd(gtFrameEnd, is_kf_end ? 1 : 0);
      if (is_kf_end) {
        const int kf_count = f1 - last_kf + 1;
        last_kf = f1 + 1;
        const size_t before = file.size();
        lz_compress_impl(z.b, file);
        const uint32_t kf_size = (uint32_t)(file.size() - before);
        put_le32(file, 40 + 28 * (size_t)k + 16, (uint32_t)z.b.size());  // RawSize
        put_le32(file, 40 + 28 * (size_t)k + 20, kf_size);               // CompressedSize
        if (k > 0 || nkf == 1) kf_max = std::max(kf_max, (uint32_t)llrint(kf_size * in.fps / kf_count));
        avg_bytes += kf_size;
        z.b.clear();
      }
    }
  }
  put_le32(file, 32, (uint32_t)llrint(avg_bytes * in.fps / in.nframes));
  put_le32(file, 36, kf_max);
  FILE *fp = fopen(path, "wb");
  TM_CHECK(fp != nullptr, TM_E_IO, "cannot create %s", path);
  const size_t w = fwrite(file.data(), 1, file.size(), fp);
  fclose(fp);
  TM_CHECK(w == file.size(), TM_E_IO, "short write to %s", path);
  return TM_OK;
}

}  // namespace tmx

extern "C" int tm_write_gtm_host(const char *path, int tm_w, int tm_h, int nframes, double fps, const int32_t *kf_start, int nkf,
                                 const uint8_t *pal_px, const uint32_t *use, int64_t ntiles, const int32_t *palettes, int pal_count,
                                 int pal_size, const tm_tilemap_item *tilemap, const char *settings) {
  if (!path || !kf_start || (!pal_px && ntiles) || !palettes || !tilemap) { tmx::set_error("tm_write_gtm_host: null argument"); return TM_E_INVAL; }
  tmx::GtmInput in;
  in.tm_w = tm_w; in.tm_h = tm_h; in.nframes = nframes; in.fps = fps;
  in.kf_start.assign(kf_start, kf_start + nkf);
  in.pal_px = pal_px;
  in.use.assign(use, use + ntiles);
  in.palettes = palettes; in.pal_count = pal_count; in.pal_size = pal_size;
  in.tilemap = tilemap;
  in.settings = settings ? settings : "";
  return tmx::write_gtm(path, in);
}

// LZCompress (extern.pas:420-439) on host buffers; *out_n = bytes needed/written; TM_E_INVAL if cap is too small.
extern "C" int tm_lz_compress_host(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_n) {
  if ((!src && n) || !out_n) { tmx::set_error("tm_lz_compress_host: null argument"); return TM_E_INVAL; }
  std::vector<uint8_t> raw(src, src + n), out;
  tmx::lz_compress(raw, out);
  *out_n = out.size();
  if (!dst || cap < out.size()) { tmx::set_error("tm_lz_compress_host: %zu bytes needed, %zu given", out.size(), cap); return TM_E_INVAL; }
  memcpy(dst, out.data(), out.size());
  return TM_OK;
}
