// tm_gtm.hip -- (f)#2: the .gtm bitstream writer, host code only.
//
// Restates TTilingEncoder.SaveStream (tilingencoder.pas:5177-5482): 40-byte GTMv header + 28-byte GTMk per keyframe
// (30-51), then per keyframe one LZMA-alone-like stream (LZCompress, extern.pas:420-439: props byte 0x62 = lc 8 / lp 0
// / pb 2, 4 MiB dictionary, eight 0xFF size bytes, end marker) of 16-bit commands (data << 4 | cmd; 53-86, 5200-5206):
// ExtendedCommand(settings) + SetDimensions + TileSet(tiles with UseCount > 1) + LoadPalette x P in the first
// keyframe, then per frame tile-map items / SkipBlocks (5392-5437) and FrameEnd.
// The LZMA encoder is our own (LzmaOptEncoder below: priced optimal parsing; the greedy parser of round 1 is kept beside it for A/B
// runs); any valid LZMA stream with these properties decodes in the reference's decoders/htmljs/lzma.js -- byte equality with the
// Pascal LZMA SDK port's output is not a goal, the properties, the decoded command stream and the stream SIZE are (1.2 % below the
// reference's on its own demo stream, tests/test_gtm.py).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
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

// ---------------------------------------------------------------------------------------------------------------
// The coder: LZMA-alone with lc/lp/pb as LZCompress sets them (SetLcLpPb(8,0,2), end marker on: extern.pas:429-430), priced optimal
// parsing.  LZMA/ULZMAEncoder.pas (the Pascal port of the LZMA SDK the reference links) chooses its literals / matches / repeated matches by
// dynamic programming over bit prices; round 1's greedy parser left 12 % on the table on the reference's own command streams.  This one
// restates that scheme in its own structure:
//  * match finder: exact 2-byte table, 3-byte and 4-byte hashes with a chain on the latter; per position the list of (length, nearest
//    distance) pairs of strictly increasing length;
//  * prices in 1/16 bit from the live probabilities (bit prices tabulated per 16 probability steps); length and distance price tables
//    refreshed every 64 lengths / 128 matches, literal and choice prices read directly;
//  * a window of up to 4096 positions: every node keeps the cheapest way to reach it with the state and the four repeat distances that
//    way implies; from a node: literal, one-byte repeat, the four repeats at every length, every match length at its nearest distance,
//    and literal-then-repeat0 (the pattern of a command stream: one field changes, the rest repeats).  A match or repeat of at least
//    kNice bytes is taken at once.
// Decoder-side semantics are the format's: lzma.js and the oracle's decoder read these streams.  (Round 1's greedy parser, 12.5 % larger on the
// reference's own command stream, is gone.)
class LzmaOptEncoder {
 public:
  static constexpr int kLc = 8, kLp = 0, kPb = 2, kNice = 128, kOpts = 4096, kDepth = 96;
  static constexpr uint32_t kDict = 1u << 22, kInf = 1u << 30;

  explicit LzmaOptEncoder(std::vector<uint8_t> &out) : rc_(out), lit_((size_t)0x300 << (kLc + kLp), 1024), opt_(kOpts) {
    for (auto &r : is_match_) for (auto &p : r) p = 1024;
    for (auto &r : is_rep0_long_) for (auto &p : r) p = 1024;
    for (auto &p : is_rep_) p = 1024;
    for (auto &p : is_rep_g0_) p = 1024;
    for (auto &p : is_rep_g1_) p = 1024;
    for (auto &p : is_rep_g2_) p = 1024;
    for (auto &r : pos_slot_) for (auto &p : r) p = 1024;
    for (auto &p : pos_special_) p = 1024;
    for (auto &p : pos_align_) p = 1024;
    for (int i = 0; i < 128; i++) {  // price of a bit whose probability is (i * 16 + 8) / 2048, in 1/16 bit
      const double pr = (i * 16 + 8) / 2048.0;
      bit_price_[i] = (uint32_t)std::lround(-std::log2(pr) * 16.0);
    }
  }

  void encode(const uint8_t *src, size_t n) {
    src_ = src; n_ = n;
    head2_.assign(1 << 16, -1); head3_.assign(1 << 16, -1); head4_.assign(1 << 20, -1); chain_.assign(n ? n : 1, -1);
    fill_len_prices(len_, len_prices_); fill_len_prices(rep_len_, rep_len_prices_);
    fill_dist_prices(); fill_align_prices();
    len_counter_ = rep_len_counter_ = 64; match_counter_ = 0; align_counter_ = 0;
    mf_pos_ = 0;
    size_t pos = 0;
    while (pos < n) {
      const int nsteps = optimum(pos);
      for (int i = 0; i < nsteps; i++) {
        const Step &st = steps_[i];
        const uint32_t pos_state = (uint32_t)pos & ((1u << kPb) - 1);
        if (st.back == kLit) emit_literal(pos, pos_state);
        else if (st.back < 4) emit_rep(st.back, st.len, pos_state);
        else emit_match(st.back - 4, st.len, pos_state);
        pos += st.len;
      }
    }
    emit_match(0xFFFFFFFFu, 2, (uint32_t)pos & ((1u << kPb) - 1));  // end marker
    rc_.flush();
  }

 private:
  static constexpr uint32_t kLit = 0xFFFFFFFFu;
  struct Step { uint32_t back, len; };  // back: kLit, 0..3 = repeat (len 1: the one-byte repeat0), 4 + distance = match
  struct Opt {
    uint32_t price, pos_prev, back_prev;
    bool lit_first;  // the way here is literal-then-repeat0: a literal at pos_prev, then repeat0 of len - 1
    uint32_t state, reps[4];
  };

  // ---- prices
  uint32_t price(uint16_t prob, int bit) const { return bit_price_[(bit ? 2048 - prob : prob) >> 4]; }
  uint32_t tree_price(const uint16_t *probs, int nbits, uint32_t sym) const {
    uint32_t m = 1, pr = 0;
    for (int i = nbits - 1; i >= 0; i--) { const int b = (sym >> i) & 1; pr += price(probs[m], b); m = (m << 1) | b; }
    return pr;
  }
  uint32_t rtree_price(const uint16_t *probs, int nbits, uint32_t sym) const {
    uint32_t m = 1, pr = 0;
    for (int i = 0; i < nbits; i++) { const int b = sym & 1; pr += price(probs[m], b); m = (m << 1) | b; sym >>= 1; }
    return pr;
  }
  uint32_t literal_price(size_t pos, uint32_t state, uint32_t rep0) const {
    const uint8_t prev_byte = pos ? src_[pos - 1] : 0;
    const uint16_t *probs = &lit_[(size_t)0x300 * prev_byte];
    const uint32_t cur = src_[pos];
    if (state < 7) return tree_price(probs, 8, cur);
    uint32_t match_byte = src_[pos - rep0 - 1], offs = 0x100, symbol = cur | 0x100, pr = 0;
    do {
      match_byte <<= 1;
      pr += price(probs[offs + (match_byte & offs) + (symbol >> 8)], (symbol >> 7) & 1);
      symbol <<= 1;
      offs &= ~(match_byte ^ symbol);
    } while (symbol < 0x10000);
    return pr;
  }
  void fill_len_prices(const LenCoder &lc, uint32_t (*tab)[272]) {
    for (int ps = 0; ps < (1 << kPb); ps++) {
      const uint32_t a0 = price(lc.choice, 0), a1 = price(lc.choice, 1), b0 = a1 + price(lc.choice2, 0), b1 = a1 + price(lc.choice2, 1);
      for (int l = 0; l < 8; l++) tab[ps][l] = a0 + tree_price(lc.low[ps], 3, l);
      for (int l = 8; l < 16; l++) tab[ps][l] = b0 + tree_price(lc.mid[ps], 3, l - 8);
      for (int l = 16; l < 272; l++) tab[ps][l] = b1 + tree_price(lc.high, 8, l - 16);
    }
  }
  static uint32_t pos_slot_of(uint32_t dist) {
    if (dist < 4) return dist;
    const int n = 31 - __builtin_clz(dist);
    return (uint32_t)(2 * n) + ((dist >> (n - 1)) & 1);
  }
  void fill_dist_prices() {
    for (int ls = 0; ls < 4; ls++) {
      for (uint32_t slot = 0; slot < 64; slot++) {
        uint32_t pr = tree_price(pos_slot_[ls], 6, slot);
        if (slot >= 14) pr += (uint32_t)(((slot >> 1) - 1) - 4) * 16;  // direct bits
        slot_prices_[ls][slot] = pr;
      }
      for (uint32_t d = 0; d < 128; d++) {
        const uint32_t slot = pos_slot_of(d);
        uint32_t pr = slot_prices_[ls][slot];
        if (slot >= 4) {
          const int footer = (int)(slot >> 1) - 1;
          const uint32_t base = (2u | (slot & 1)) << footer;
          pr += rtree_price(pos_special_ + ((int)base - (int)slot - 1), footer, d - base);
        }
        dist_prices_[ls][d] = pr;
      }
    }
    match_counter_ = 0;
  }
  void fill_align_prices() {
    for (uint32_t i = 0; i < 16; i++) align_prices_[i] = rtree_price(pos_align_, 4, i);
    align_counter_ = 0;
  }
  uint32_t dist_len_price(uint32_t dist, uint32_t len, uint32_t pos_state) const {
    const uint32_t ls = std::min<uint32_t>(len - 2, 3);
    const uint32_t pr = dist < 128 ? dist_prices_[ls][dist] : slot_prices_[ls][pos_slot_of(dist)] + align_prices_[dist & 15];
    return pr + len_prices_[pos_state][len - 2];
  }
  uint32_t pure_rep_price(uint32_t k, uint32_t state, uint32_t pos_state) const {
    if (k == 0) return price(is_rep_g0_[state], 0) + price(is_rep0_long_[state][pos_state], 1);
    uint32_t pr = price(is_rep_g0_[state], 1);
    if (k == 1) return pr + price(is_rep_g1_[state], 0);
    return pr + price(is_rep_g1_[state], 1) + price(is_rep_g2_[state], (int)k - 2);
  }
  static uint32_t st_lit(uint32_t s) { return s < 4 ? 0 : (s < 10 ? s - 3 : s - 6); }
  static uint32_t st_match(uint32_t s) { return s < 7 ? 7 : 10; }
  static uint32_t st_rep(uint32_t s) { return s < 7 ? 8 : 11; }
  static uint32_t st_shortrep(uint32_t s) { return s < 7 ? 9 : 11; }

  // ---- match finder
  static uint32_t h3(const uint8_t *p) { return (uint32_t)((p[0] | (p[1] << 8) | (p[2] << 16)) * 2654435761u) >> 16; }
  static uint32_t h4(const uint8_t *p) { return (uint32_t)((p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)) * 2654435761u) >> 12; }
  void mf_insert(size_t i) {
    if (i + 2 <= n_) head2_[src_[i] | (src_[i + 1] << 8)] = (int32_t)i;
    if (i + 3 <= n_) head3_[h3(src_ + i)] = (int32_t)i;
    if (i + 4 <= n_) { const uint32_t h = h4(src_ + i); chain_[i] = head4_[h]; head4_[h] = (int32_t)i; }
  }
  size_t match_len(size_t a, size_t b, size_t from, size_t max_len) const {  // a > b
    size_t l = from;
    while (l < max_len && src_[a + l] == src_[b + l]) l++;
    return l;
  }
  // (length, distance - 1) pairs at position i, lengths strictly increasing; inserts i.  Positions must come in order.
  int mf_find(size_t i, uint32_t *lens, uint32_t *dists) {
    if (mf_pos_ == i + 1 && last_i_ == i) {  // asked again for the position a parse ended on: the lists it got then
      for (int q = 0; q < last_np_; q++) { lens[q] = last_lens_[q]; dists[q] = last_dists_[q]; }
      return last_np_;
    }
    while (mf_pos_ < i) mf_insert(mf_pos_++);
    const size_t max_len = std::min<size_t>(273, n_ - i);
    int np = 0;
    size_t best = 1;
    if (max_len >= 2) {
      const int32_t c2 = head2_[src_[i] | (src_[i + 1] << 8)];
      if (c2 >= 0 && i - (size_t)c2 <= kDict) {
        best = match_len(i, (size_t)c2, 2, max_len);
        lens[np] = (uint32_t)best; dists[np++] = (uint32_t)(i - (size_t)c2 - 1);
      }
      if (max_len >= 3 && best < max_len) {
        const int32_t c3 = head3_[h3(src_ + i)];
        if (c3 >= 0 && c3 != c2 && i - (size_t)c3 <= kDict && src_[c3] == src_[i] && src_[c3 + 1] == src_[i + 1] && src_[c3 + 2] == src_[i + 2]) {
          const size_t l = match_len(i, (size_t)c3, 3, max_len);
          if (l > best) { best = l; lens[np] = (uint32_t)l; dists[np++] = (uint32_t)(i - (size_t)c3 - 1); }
        }
      }
      if (max_len >= 4 && best < max_len) {
        int32_t c = head4_[h4(src_ + i)];
        for (int depth = 0; c >= 0 && depth < kDepth; depth++, c = chain_[c]) {
          const size_t dist = i - (size_t)c;
          if (dist > kDict) break;
          if (src_[c + best] != src_[i + best] || src_[c] != src_[i]) continue;
          const size_t l = match_len(i, (size_t)c, 0, max_len);
          if (l > best) {
            best = l; lens[np] = (uint32_t)l; dists[np++] = (uint32_t)(dist - 1);
            if (l == max_len) break;
          }
        }
      }
    }
    mf_insert(i);
    mf_pos_ = i + 1;
    last_i_ = i; last_np_ = np;
    for (int q = 0; q < np; q++) { last_lens_[q] = lens[q]; last_dists_[q] = dists[q]; }
    return np;
  }

  // ---- the parse of one stretch starting at pos: fills steps_, returns their number
  int optimum(size_t pos) {
    const size_t avail0 = std::min<size_t>(273, n_ - pos);
    uint32_t mlens[280], mdists[280];
    int np = mf_find(pos, mlens, mdists);
    auto single = [&](uint32_t back, uint32_t len) { steps_[0] = Step{back, len}; return 1; };
    if (avail0 < 2) return single(kLit, 1);
    uint32_t rep_lens[4], best_rep = 0;
    for (int k = 0; k < 4; k++) {
      rep_lens[k] = pos > reps_[k] ? (uint32_t)match_len(pos, pos - reps_[k] - 1, 0, avail0) : 0;
      if (rep_lens[k] > rep_lens[best_rep]) best_rep = (uint32_t)k;
    }
    if (rep_lens[best_rep] >= (uint32_t)kNice) return single(best_rep, rep_lens[best_rep]);
    const uint32_t main_len = np ? mlens[np - 1] : 0;
    if (main_len >= (uint32_t)kNice) return single(mdists[np - 1] + 4, main_len);
    const uint8_t cur_byte = src_[pos];
    const bool has_rep0 = pos > reps_[0];
    const uint8_t match_byte = has_rep0 ? src_[pos - reps_[0] - 1] : (uint8_t)~cur_byte;
    if (main_len < 2 && cur_byte != match_byte && rep_lens[best_rep] < 2) return single(kLit, 1);

    Opt *o = opt_.data();
    o[0].state = state_; for (int k = 0; k < 4; k++) o[0].reps[k] = (uint32_t)reps_[k];
    o[0].price = 0;
    uint32_t pos_state = (uint32_t)pos & 3;
    o[1].price = price(is_match_[state_][pos_state], 0) + literal_price(pos, state_, (uint32_t)reps_[0]);
    o[1].pos_prev = 0; o[1].back_prev = kLit; o[1].lit_first = false;
    const uint32_t match_price = price(is_match_[state_][pos_state], 1), rep_match_price = match_price + price(is_rep_[state_], 1);
    if (match_byte == cur_byte) {
      const uint32_t sp = rep_match_price + price(is_rep_g0_[state_], 0) + price(is_rep0_long_[state_][pos_state], 0);
      if (sp < o[1].price) { o[1].price = sp; o[1].back_prev = 0; }
    }
    uint32_t len_end = std::max(main_len, rep_lens[best_rep]);
    if (len_end < 2) return single(o[1].back_prev, 1);
    for (uint32_t l = 2; l <= len_end; l++) o[l].price = kInf;
    for (uint32_t k = 0; k < 4; k++) {
      if (rep_lens[k] < 2) continue;
      const uint32_t base = rep_match_price + pure_rep_price(k, state_, pos_state);
      for (uint32_t l = 2; l <= rep_lens[k]; l++) {
        const uint32_t pr = base + rep_len_prices_[pos_state][l - 2];
        if (pr < o[l].price) { o[l].price = pr; o[l].pos_prev = 0; o[l].back_prev = k; o[l].lit_first = false; }
      }
    }
    {
      const uint32_t normal = match_price + price(is_rep_[state_], 0);
      int pi = 0;
      for (uint32_t l = rep_lens[0] >= 2 ? rep_lens[0] + 1 : 2; l <= main_len; l++) {
        while (mlens[pi] < l) pi++;
        const uint32_t pr = normal + dist_len_price(mdists[pi], l, pos_state);
        if (pr < o[l].price) { o[l].price = pr; o[l].pos_prev = 0; o[l].back_prev = mdists[pi] + 4; o[l].lit_first = false; }
      }
    }
    uint32_t cur = 0;
    for (;;) {
      cur++;
      if (cur == len_end) return backward(cur);
      np = mf_find(pos + cur, mlens, mdists);
      const uint32_t new_len0 = np ? mlens[np - 1] : 0;
      if (new_len0 >= (uint32_t)kNice) return backward(cur);  // the long match is taken by the next call
      // the state and the repeat distances the cheapest way to this node implies
      Opt &nd = o[cur];
      {
        uint32_t pp = nd.pos_prev;
        uint32_t st = o[pp].state;
        uint32_t rp[4] = {o[pp].reps[0], o[pp].reps[1], o[pp].reps[2], o[pp].reps[3]};
        if (nd.lit_first) {  // literal at pp, then repeat0 up to here
          st = st_rep(st_lit(st));
        } else if (nd.back_prev == kLit) {
          st = st_lit(st);
        } else if (nd.back_prev < 4) {
          if (cur - pp == 1) st = st_shortrep(st);  // only repeat0 of one byte gets here with length 1
          else {
            st = st_rep(st);
            const uint32_t k = nd.back_prev, d = rp[k];
            for (uint32_t j = k; j > 0; j--) rp[j] = rp[j - 1];
            rp[0] = d;
          }
        } else {
          st = st_match(st);
          rp[3] = rp[2]; rp[2] = rp[1]; rp[1] = rp[0]; rp[0] = nd.back_prev - 4;
        }
        nd.state = st;
        for (int k = 0; k < 4; k++) nd.reps[k] = rp[k];
      }
      const size_t p = pos + cur;
      pos_state = (uint32_t)p & 3;
      const uint32_t st = nd.state, cp = nd.price;
      const uint8_t cb = src_[p];
      const bool hr0 = p > nd.reps[0];
      const uint8_t mb = hr0 ? src_[p - nd.reps[0] - 1] : (uint8_t)~cb;
      const uint32_t lit_pr = cp + price(is_match_[st][pos_state], 0) + literal_price(p, st, nd.reps[0]);
      bool next_is_lit = false;
      if (lit_pr < o[cur + 1].price) {
        o[cur + 1].price = lit_pr; o[cur + 1].pos_prev = cur; o[cur + 1].back_prev = kLit; o[cur + 1].lit_first = false;
        next_is_lit = true;
      }
      const uint32_t mp = cp + price(is_match_[st][pos_state], 1), rmp = mp + price(is_rep_[st], 1);
      if (mb == cb && !(o[cur + 1].pos_prev < cur && o[cur + 1].back_prev == 0 && !o[cur + 1].lit_first)) {
        const uint32_t sp = rmp + price(is_rep_g0_[st], 0) + price(is_rep0_long_[st][pos_state], 0);
        if (sp <= o[cur + 1].price) {
          o[cur + 1].price = sp; o[cur + 1].pos_prev = cur; o[cur + 1].back_prev = 0; o[cur + 1].lit_first = false;
          next_is_lit = false;
        }
      }
      const size_t avail_full = std::min<size_t>(273, n_ - p);
      size_t avail = std::min<size_t>(avail_full, (size_t)kOpts - 1 - cur);
      if (avail < 2) continue;
      if (avail > (size_t)kNice) avail = kNice;
      auto reach = [&](uint32_t to) { while (len_end < to) o[++len_end].price = kInf; };
      // literal, then repeat0 (one changed byte inside a repeating record)
      if (next_is_lit && hr0 && mb != cb && avail_full >= 3) {
        const size_t lim = std::min<size_t>(avail_full - 1, kNice);
        const size_t l2 = match_len(p + 1, p - nd.reps[0], 0, lim);  // (p + 1) against (p + 1 - rep0 - 1)
        if (l2 >= 2 && cur + 1 + l2 < (size_t)kOpts) {
          const uint32_t st2 = st_lit(st), ps2 = (uint32_t)(p + 1) & 3;
          const uint32_t pr = lit_pr + price(is_match_[st2][ps2], 1) + price(is_rep_[st2], 1) + pure_rep_price(0, st2, ps2) + rep_len_prices_[ps2][l2 - 2];
          const uint32_t to = cur + 1 + (uint32_t)l2;
          reach(to);
          if (pr < o[to].price) { o[to].price = pr; o[to].pos_prev = cur; o[to].back_prev = 0; o[to].lit_first = true; }
        }
      }
      uint32_t start_len = 2;
      for (uint32_t k = 0; k < 4; k++) {
        if (p <= nd.reps[k]) continue;
        const size_t lk = match_len(p, p - nd.reps[k] - 1, 0, avail);
        if (lk < 2) continue;
        reach(cur + (uint32_t)lk);
        const uint32_t base = rmp + pure_rep_price(k, st, pos_state);
        for (uint32_t l = (uint32_t)lk; l >= 2; l--) {
          const uint32_t pr = base + rep_len_prices_[pos_state][l - 2];
          Opt &t = o[cur + l];
          if (pr < t.price) { t.price = pr; t.pos_prev = cur; t.back_prev = k; t.lit_first = false; }
        }
        if (k == 0) start_len = (uint32_t)lk + 1;
      }
      uint32_t new_len = new_len0;
      if (new_len > avail) { new_len = (uint32_t)avail; }
      if (new_len >= start_len) {
        const uint32_t normal = mp + price(is_rep_[st], 0);
        reach(cur + new_len);
        int pi = 0;
        for (uint32_t l = start_len; l <= new_len; l++) {
          while (pi < np - 1 && mlens[pi] < l) pi++;
          const uint32_t pr = normal + dist_len_price(mdists[pi], l, pos_state);
          Opt &t = o[cur + l];
          if (pr < t.price) { t.price = pr; t.pos_prev = cur; t.back_prev = mdists[pi] + 4; t.lit_first = false; }
        }
      }
    }
  }
  int backward(uint32_t cur) {
    int nrev = 0;
    const Opt *o = opt_.data();
    while (cur > 0) {
      const Opt &nd = o[cur];
      if (nd.lit_first) {
        rev_[nrev++] = Step{0, cur - nd.pos_prev - 1};  // repeat0
        rev_[nrev++] = Step{kLit, 1};
      } else {
        rev_[nrev++] = Step{nd.back_prev, cur - nd.pos_prev};
      }
      cur = nd.pos_prev;
    }
    for (int i = 0; i < nrev; i++) steps_[i] = rev_[nrev - 1 - i];
    return nrev;
  }

  // ---- emission (the model moves here, and only here)
  void tree(uint16_t *probs, int nbits, uint32_t sym) {
    uint32_t m = 1;
    for (int i = nbits - 1; i >= 0; i--) { const int b = (sym >> i) & 1; rc_.bit(probs[m], b); m = (m << 1) | b; }
  }
  void rtree(uint16_t *probs, int nbits, uint32_t sym) {
    uint32_t m = 1;
    for (int i = 0; i < nbits; i++) { const int b = sym & 1; rc_.bit(probs[m], b); m = (m << 1) | b; sym >>= 1; }
  }
  void emit_len(LenCoder &lc, uint32_t len, uint32_t pos_state, uint32_t (*tab)[272], int &counter) {
    len -= 2;
    if (len < 8) { rc_.bit(lc.choice, 0); tree(lc.low[pos_state], 3, len); }
    else if (len < 16) { rc_.bit(lc.choice, 1); rc_.bit(lc.choice2, 0); tree(lc.mid[pos_state], 3, len - 8); }
    else { rc_.bit(lc.choice, 1); rc_.bit(lc.choice2, 1); tree(lc.high, 8, len - 16); }
    if (--counter <= 0) { fill_len_prices(lc, tab); counter = 64; }
  }
  void emit_literal(size_t pos, uint32_t pos_state) {
    rc_.bit(is_match_[state_][pos_state], 0);
    const uint8_t prev_byte = pos ? src_[pos - 1] : 0;
    uint16_t *probs = &lit_[(size_t)0x300 * prev_byte];
    const uint32_t cur = src_[pos];
    if (state_ < 7) {
      tree(probs, 8, cur);
    } else {
      uint32_t match_byte = src_[pos - reps_[0] - 1], offs = 0x100, symbol = cur | 0x100;
      do {
        match_byte <<= 1;
        rc_.bit(probs[offs + (match_byte & offs) + (symbol >> 8)], (symbol >> 7) & 1);
        symbol <<= 1;
        offs &= ~(match_byte ^ symbol);
      } while (symbol < 0x10000);
    }
    state_ = st_lit(state_);
  }
  void emit_rep(uint32_t k, uint32_t len, uint32_t pos_state) {
    rc_.bit(is_match_[state_][pos_state], 1);
    rc_.bit(is_rep_[state_], 1);
    if (k == 0) {
      rc_.bit(is_rep_g0_[state_], 0);
      rc_.bit(is_rep0_long_[state_][pos_state], len == 1 ? 0 : 1);
      if (len == 1) { state_ = st_shortrep(state_); return; }
    } else {
      rc_.bit(is_rep_g0_[state_], 1);
      if (k == 1) rc_.bit(is_rep_g1_[state_], 0);
      else { rc_.bit(is_rep_g1_[state_], 1); rc_.bit(is_rep_g2_[state_], (int)k - 2); }
      const size_t d = reps_[k];
      for (uint32_t j = k; j > 0; j--) reps_[j] = reps_[j - 1];
      reps_[0] = d;
    }
    emit_len(rep_len_, len, pos_state, rep_len_prices_, rep_len_counter_);
    state_ = st_rep(state_);
  }
  void emit_match(uint32_t dist, uint32_t len, uint32_t pos_state) {
    rc_.bit(is_match_[state_][pos_state], 1);
    rc_.bit(is_rep_[state_], 0);
    emit_len(len_, len, pos_state, len_prices_, len_counter_);
    state_ = st_match(state_);
    const uint32_t slot = pos_slot_of(dist);
    tree(pos_slot_[std::min<uint32_t>(len - 2, 3)], 6, slot);
    if (slot >= 4) {
      const int footer = (int)(slot >> 1) - 1;
      const uint32_t base = (2u | (slot & 1)) << footer, reduced = dist - base;
      if (slot < 14) rtree(pos_special_ + ((int)base - (int)slot - 1), footer, reduced);
      else {
        rc_.direct(reduced >> 4, footer - 4);
        rtree(pos_align_, 4, reduced & 15);
        if (++align_counter_ >= 16) fill_align_prices();
      }
    }
    reps_[3] = reps_[2]; reps_[2] = reps_[1]; reps_[1] = reps_[0]; reps_[0] = dist;
    if (++match_counter_ >= 128) fill_dist_prices();
  }

  RangeEncoder rc_;
  std::vector<uint16_t> lit_;
  uint16_t is_match_[12][16], is_rep0_long_[12][16], is_rep_[12], is_rep_g0_[12], is_rep_g1_[12], is_rep_g2_[12];
  uint16_t pos_slot_[4][64], pos_special_[128], pos_align_[16];
  LenCoder len_, rep_len_;
  uint32_t state_ = 0;
  size_t reps_[4] = {0, 0, 0, 0};
  const uint8_t *src_ = nullptr;
  size_t n_ = 0, mf_pos_ = 0;
  std::vector<int32_t> head2_, head3_, head4_, chain_;
  std::vector<Opt> opt_;
  Step steps_[kOpts + 8], rev_[kOpts + 8];
  uint32_t bit_price_[128];
  uint32_t len_prices_[4][272], rep_len_prices_[4][272], slot_prices_[4][64], dist_prices_[4][128], align_prices_[16];
  int len_counter_ = 64, rep_len_counter_ = 64, match_counter_ = 0, align_counter_ = 0;
  size_t last_i_ = (size_t)-1;
  int last_np_ = 0;
  uint32_t last_lens_[280], last_dists_[280];
};

void lz_compress_impl(const std::vector<uint8_t> &raw, std::vector<uint8_t> &dst) {  // LZCompress, extern.pas:420-439
  dst.push_back((uint8_t)((LzmaOptEncoder::kPb * 5 + LzmaOptEncoder::kLp) * 9 + LzmaOptEncoder::kLc));  // 0x62
  for (int i = 0; i < 4; i++) dst.push_back((uint8_t)(LzmaOptEncoder::kDict >> (8 * i)));
  for (int i = 0; i < 8; i++) dst.push_back(0xFF);
  auto enc = std::make_unique<LzmaOptEncoder>(dst);  // (its parse window lives in the object: not on the stack)
  enc->encode(raw.data(), raw.size());
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


// ---------------------------------------------------------------------------------------------------------------
// LZDecompress (extern.pas:441-458): one LZMA-alone stream (5 property bytes, 8 size bytes, end marker); returns the bytes
// consumed so the next key frame's stream can follow.
namespace tmx {
namespace {
struct RangeDecoder {
  const uint8_t *p, *end;
  uint32_t range = 0xFFFFFFFFu, code = 0;
  bool overrun = false;
  uint8_t next() { if (p < end) return *p++; overrun = true; return 0; }
  void init() { for (int i = 0; i < 5; i++) code = (code << 8) | next(); }
  int bit(uint16_t &prob) {
    const uint32_t bound = (range >> 11) * prob;
    int b;
    if (code < bound) { range = bound; prob = (uint16_t)(prob + ((2048 - prob) >> 5)); b = 0; }
    else { range -= bound; code -= bound; prob = (uint16_t)(prob - (prob >> 5)); b = 1; }
    if (range < (1u << 24)) { range <<= 8; code = (code << 8) | next(); }
    return b;
  }
  uint32_t direct(int nbits) {
    uint32_t r = 0;
    for (int i = 0; i < nbits; i++) {
      range >>= 1;
      const uint32_t t = (code - range) >> 31;
      code -= range & (t - 1);
      r = (r << 1) | (1 - t);
      if (range < (1u << 24)) { range <<= 8; code = (code << 8) | next(); }
    }
    return r;
  }
  uint32_t tree(uint16_t *probs, int nbits) {
    uint32_t m = 1;
    for (int i = 0; i < nbits; i++) m = (m << 1) | (uint32_t)bit(probs[m]);
    return m - (1u << nbits);
  }
  uint32_t rtree(uint16_t *probs, int nbits) {
    uint32_t m = 1, sym = 0;
    for (int i = 0; i < nbits; i++) { const int b = bit(probs[m]); m = (m << 1) | (uint32_t)b; sym |= (uint32_t)b << i; }
    return sym;
  }
};
struct LenDecoder {
  uint16_t choice[2], low[16][8], mid[16][8], high[256];
  LenDecoder() { uint16_t *q = &choice[0]; for (size_t i = 0; i < sizeof(*this) / 2; i++) q[i] = 1024; }
  uint32_t decode(RangeDecoder &rc, uint32_t ps) {
    if (!rc.bit(choice[0])) return rc.tree(low[ps], 3);
    if (!rc.bit(choice[1])) return 8 + rc.tree(mid[ps], 3);
    return 16 + rc.tree(high, 8);
  }
};
}  // namespace

int lz_decompress(const uint8_t *src, size_t n, std::vector<uint8_t> &dst, size_t *consumed) {
  TM_CHECK(n >= 18, TM_E_IO, "LZMA stream too short");
  int props = src[0];
  const int lc = props % 9; props /= 9;
  const int lp = props % 5, pb = props / 5;
  TM_CHECK(pb <= 4, TM_E_IO, "bad LZMA properties");
  std::vector<uint16_t> lit((size_t)0x300 << (lc + lp), 1024);
  uint16_t is_match[12 << 4], is_rep0_long[12 << 4], is_rep[12], is_g0[12], is_g1[12], is_g2[12], pos_slot[4][64], pos_dec[128], pos_align[16];
  auto fill = [](uint16_t *q, size_t cnt) { for (size_t i = 0; i < cnt; i++) q[i] = 1024; };
  fill(is_match, 12 << 4); fill(is_rep0_long, 12 << 4); fill(is_rep, 12); fill(is_g0, 12); fill(is_g1, 12); fill(is_g2, 12);
  fill(&pos_slot[0][0], 256); fill(pos_dec, 128); fill(pos_align, 16);
  LenDecoder len_dec, rep_len_dec;
  RangeDecoder rc{src + 13, src + n};
  rc.init();
  uint32_t state = 0, rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0;
  uint8_t prev = 0;
  dst.clear();
  bool ok = false;
  while (!rc.overrun) {
    const size_t pos = dst.size();
    const uint32_t ps = (uint32_t)pos & ((1u << pb) - 1);
    if (!rc.bit(is_match[(state << 4) + ps])) {
      uint16_t *probs = &lit[(size_t)0x300 * ((((uint32_t)pos & ((1u << lp) - 1)) << lc) + (prev >> (8 - lc)))];
      uint32_t sym = 1;
      if (state >= 7) {
        uint32_t mb = dst[pos - rep0 - 1];
        do {
          const uint32_t mbit = (mb >> 7) & 1;
          mb <<= 1;
          const int b = rc.bit(probs[((1 + mbit) << 8) + sym]);
          sym = (sym << 1) | (uint32_t)b;
          if (mbit != (uint32_t)b) { while (sym < 0x100) sym = (sym << 1) | (uint32_t)rc.bit(probs[sym]); break; }
        } while (sym < 0x100);
      } else {
        do sym = (sym << 1) | (uint32_t)rc.bit(probs[sym]); while (sym < 0x100);
      }
      prev = (uint8_t)sym;
      dst.push_back(prev);
      state = state < 4 ? 0 : state - (state < 10 ? 3 : 6);
      continue;
    }
    uint32_t len;
    if (rc.bit(is_rep[state])) {
      len = 0;
      if (!rc.bit(is_g0[state])) {
        if (!rc.bit(is_rep0_long[(state << 4) + ps])) { state = state < 7 ? 9 : 11; len = 1; }
      } else {
        uint32_t dist;
        if (!rc.bit(is_g1[state])) dist = rep1;
        else {
          if (!rc.bit(is_g2[state])) dist = rep2;
          else { dist = rep3; rep3 = rep2; }
          rep2 = rep1;
        }
        rep1 = rep0;
        rep0 = dist;
      }
      if (len == 0) { len = 2 + rep_len_dec.decode(rc, ps); state = state < 7 ? 8 : 11; }
    } else {
      rep3 = rep2; rep2 = rep1; rep1 = rep0;
      len = 2 + len_dec.decode(rc, ps);
      state = state < 7 ? 7 : 10;
      const uint32_t slot = rc.tree(pos_slot[len <= 5 ? len - 2 : 3], 6);
      if (slot >= 4) {
        const int nd = (int)(slot >> 1) - 1;
        rep0 = (2u | (slot & 1)) << nd;
        if (slot < 14) rep0 += rc.rtree(pos_dec + ((int)rep0 - (int)slot - 1), nd);
        else {
          rep0 += rc.direct(nd - 4) << 4;
          rep0 += rc.rtree(pos_align, 4);
          if (rep0 == 0xFFFFFFFFu) { ok = true; break; }  // end marker
        }
      } else rep0 = slot;
    }
    if ((size_t)rep0 >= dst.size()) break;  // corrupt
    for (uint32_t i = 0; i < len; i++) dst.push_back(dst[dst.size() - rep0 - 1]);
    prev = dst.back();
  }
  TM_CHECK(ok && !rc.overrun, TM_E_IO, "corrupt LZMA stream");
  if (consumed) *consumed = (size_t)(rc.p - src);
  return TM_OK;
}

// LoadStream (tilingencoder.pas:4880-5175): the command streams of all key frames back into tables
int read_gtm(const char *path, GtmLoaded *out) {
  FILE *fp = fopen(path, "rb");
  TM_CHECK(fp != nullptr, TM_E_IO, "cannot open %s", path);
  std::vector<uint8_t> file;
  {
    uint8_t buf[1 << 16];
    size_t r;
    while ((r = fread(buf, 1, sizeof(buf), fp)) > 0) file.insert(file.end(), buf, buf + r);
    fclose(fp);
  }
  auto le32 = [&](size_t at) { return (uint32_t)file[at] | ((uint32_t)file[at + 1] << 8) | ((uint32_t)file[at + 2] << 16) | ((uint32_t)file[at + 3] << 24); };
  size_t pos = 0;
  out->header_frames = -1;
  if (file.size() >= 40 && memcmp(file.data(), "GTMv", 4) == 0) {  // 5015-5033
    out->header_w = (int)le32(16); out->header_h = (int)le32(20); out->header_frames = (int)le32(28);
    pos = le32(8);
  }
  int frm = -1, tm_pos = 0, last_tile = -1, loaded = 0;
  bool frame_open = false;
  std::vector<uint8_t> kf;
  while (pos < file.size()) {
    size_t used = 0;
    TM_TRY(lz_decompress(file.data() + pos, file.size() - pos, kf, &used));
    pos += used;
    out->kf_start.push_back(loaded);
    size_t p = 0;
    auto need = [&](size_t k) { return p + k <= kf.size(); };
    auto u8 = [&]() { return (uint32_t)kf[p++]; };
    auto u16 = [&]() { const uint32_t v = kf[p] | (kf[p + 1] << 8); p += 2; return v; };
    auto u32 = [&]() { const uint32_t v = (uint32_t)kf[p] | ((uint32_t)kf[p + 1] << 8) | ((uint32_t)kf[p + 2] << 16) | ((uint32_t)kf[p + 3] << 24); p += 4; return v; };
    auto item = [&]() -> tm_tilemap_item * {  // "next frame if needed" (5091-5093 ...)
      if (!frame_open) { frm++; out->tilemap.resize((size_t)(frm + 1) * out->tm_w * out->tm_h); frame_open = true; }
      tm_tilemap_item *it = &out->tilemap[(size_t)frm * out->tm_w * out->tm_h + tm_pos];
      memset(it, 0, sizeof(*it));
      it->TileIdx = -1; it->PalIdx = -1;
      return it;
    };
    bool kf_end = false;
    while (!kf_end) {
      TM_CHECK(need(2), TM_E_IO, "%s: truncated command stream", path);
      const uint32_t w = u16(), cmd = w & 15, data = w >> 4;
      switch (cmd) {
        case gtExtended: { TM_CHECK(need(4), TM_E_IO, "truncated"); const uint32_t k = u32(); TM_CHECK(need(k), TM_E_IO, "truncated"); if (data == 0) out->settings.assign((const char *)&kf[p], k); p += k; break; }
        case gtSetDimensions: {
          TM_CHECK(need(12), TM_E_IO, "truncated");
          out->tm_w = (int)u16(); out->tm_h = (int)u16();
          const uint32_t ns = u32();
          out->fps = 1000.0 * 1000 * 1000 / ns;
          const uint32_t tc = u32();
          TM_CHECK(out->tm_w > 0 && out->tm_h > 0 && ns > 0, TM_E_IO, "bad dimensions");
          out->pal_px.assign((size_t)tc * 64, 0);
          out->use.assign(tc, 0);
          break;
        }
        case gtTileSet: {
          TM_CHECK(need(8), TM_E_IO, "truncated");
          const uint32_t a = u32(), b = u32();
          TM_CHECK(b >= a && (size_t)b < out->use.size() && need((size_t)(b - a + 1) * 64), TM_E_IO, "bad tile set");
          memcpy(&out->pal_px[(size_t)a * 64], &kf[p], (size_t)(b - a + 1) * 64);
          p += (size_t)(b - a + 1) * 64;
          last_tile = std::max(last_tile, (int)b);
          out->pal_size = (int)data;
          break;
        }
        case gtLoadPalette: {
          TM_CHECK(need(2 + (size_t)out->pal_size * 4), TM_E_IO, "truncated");
          const uint32_t pi = u16();
          if (out->palettes.size() < (size_t)(pi + 1) * out->pal_size) out->palettes.resize((size_t)(pi + 1) * out->pal_size, 0);
          for (int c = 0; c < out->pal_size; c++) out->palettes[(size_t)pi * out->pal_size + c] = (int32_t)(u32() & 0xffffff);
          break;
        }
        case gtFrameEnd:
          TM_CHECK(tm_pos == out->tm_w * out->tm_h, TM_E_IO, "incomplete tile map");
          tm_pos = 0;
          frame_open = false;
          loaded++;
          kf_end = (data & 1) != 0;
          break;
        case gtSkip:
          for (uint32_t k = 0; k <= data; k++) {
            TM_CHECK(out->tm_w > 0 && tm_pos < out->tm_w * out->tm_h, TM_E_IO, "skip past the tile map");
            tm_tilemap_item *it = item();
            it->Flags = 4;
            tm_pos++;
          }
          break;
        case gtShortShort: case gtLongShort: case gtLongLong: {
          TM_CHECK(out->tm_w > 0 && tm_pos < out->tm_w * out->tm_h && need(cmd == gtShortShort ? 2 : (cmd == gtLongShort ? 4 : 6)), TM_E_IO, "bad tile-map item");
          uint32_t pal = cmd == gtLongLong ? u16() : (data >> 2) & 1023;
          const uint32_t tile = cmd == gtShortShort ? u16() : u32();
          tm_tilemap_item *it = item();
          it->TileIdx = (int32_t)tile; it->PalIdx = (int32_t)pal; it->Flags = data & 3;
          if ((size_t)tile < out->use.size()) out->use[tile]++;
          tm_pos++;
          break;
        }
        case gtPredShort: {
          TM_CHECK(out->tm_w > 0 && tm_pos < out->tm_w * out->tm_h, TM_E_IO, "bad tile-map item");
          tm_tilemap_item *it = item();
          it->PredictedX = (int8_t)((int)(data & 31) - (int)(data & 32));
          it->PredictedY = (int8_t)((int)((data >> 6) & 31) - (int)((data >> 6) & 32));
          it->Flags = 4;
          tm_pos++;
          break;
        }
        case gtPredLong: {
          TM_CHECK(out->tm_w > 0 && tm_pos < out->tm_w * out->tm_h && need(2), TM_E_IO, "bad tile-map item");
          tm_tilemap_item *it = item();
          it->PredictedX = (int8_t)u8(); it->PredictedY = (int8_t)u8();
          it->Flags = 4;
          tm_pos++;
          break;
        }
        case gtIntra: {
          TM_CHECK(out->tm_w > 0 && tm_pos < out->tm_w * out->tm_h && need(66), TM_E_IO, "bad intra tile");
          const uint32_t pal = u16();
          last_tile++;
          TM_CHECK((size_t)last_tile < out->use.size(), TM_E_IO, "more intra tiles than the tile count allows");
          memcpy(&out->pal_px[(size_t)last_tile * 64], &kf[p], 64);
          p += 64;
          tm_tilemap_item *it = item();
          it->TileIdx = last_tile; it->PalIdx = (int32_t)pal; it->Flags = data & 3;
          out->use[(size_t)last_tile]++;
          tm_pos++;
          break;
        }
        default: set_error("%s: unknown command %u", path, cmd); return TM_E_IO;
      }
    }
  }
  out->nframes = loaded;
  out->pal_count = out->pal_size > 0 ? (int)(out->palettes.size() / out->pal_size) : 0;
  TM_CHECK(out->tilemap.size() == (size_t)loaded * out->tm_w * out->tm_h, TM_E_IO, "%s: frame count mismatch", path);
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

// LZDecompress (extern.pas:441-458) on host buffers; *out_n = decoded size (also when cap is too small), *consumed = stream bytes
extern "C" int tm_lz_decompress_host(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_n, size_t *consumed) {
  if (!src || !out_n) { tmx::set_error("tm_lz_decompress_host: null argument"); return TM_E_INVAL; }
  std::vector<uint8_t> out;
  TM_TRY(tmx::lz_decompress(src, n, out, consumed));
  *out_n = out.size();
  if (!dst || cap < out.size()) { tmx::set_error("tm_lz_decompress_host: %zu bytes needed, %zu given", out.size(), cap); return TM_E_INVAL; }
  memcpy(dst, out.data(), out.size());
  return TM_OK;
}
