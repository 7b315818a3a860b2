// tm_dl3.hip -- A17, the other half: dl3quant (dlquant/quantizer.c:437-455), Dennis Lee's DL3 colour quantiser.
//
// Unreachable in the reference snapshot (extern.pas:196 imports it, nothing calls it, the DLL is not shipped) but named by the
// north star, so it is built as an operator of its own: tm_stage_dl3quant, plus the fine-seam twin `dl3quant` with the import's
// signature.  PARITY UNPINNED: the reference holds no output of it; parity is against the oracle's restatement (tmo_dl3quant).
//
//   build_table3 (:486-518)   histogram of the pixels at lookup_bpc bits per channel: sums of R, G, B and a count per cell (32-bit,
//                             wrapping like the Win64 build's `ulong`), the occupied cells compacted in index order
//                             -> k_dl3_hist (integer atomics: order-free), an exclusive scan, k_dl3_compact
//   reduce_table3 (:583-648)  pass 1: every entry's cheapest partner among the LATER entries (recount_next, :543-559; calc_err,
//                             :520-541) -> k_dl3_pass1, one wave per entry, the whole chip;
//                             pass 2: merge the entry of least error into its partner, move the last entry into the hole, repair
//                             the partners that pointed at either -- one merge after the other, each depending on the one before:
//                             k_dl3_reduce, ONE persistent workgroup of 1024 threads that walks the table (kept in L2) in parallel
//                             inside a merge and meets at barriers between its phases.  The phases are the reference's statements in
//                             the reference's order; what runs side by side inside a phase does not depend on its order (a
//                             recount reads colour sums only, which a phase does not change).
//   set_palette3 (:650-664)   the entries' rounded means, planar.
// Arithmetic as the reference's: integer divisions in 32 bits, `float` errors with a correctly rounded square root (a table of
// sqrtf over the 195 076 possible sums of three squares, made on the host), products and the sum in IEEE single, strict `<` so the
// first index wins every tie.
#include <rocprim/device/device_scan.hpp>

#include <cmath>
#include <vector>

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {
namespace {

constexpr int DL3_SQ_MAX = 3 * 255 * 255;  // largest sum of three squared byte differences
constexpr int DL3_NT = 1024;

struct Dl3Table {  // structure of arrays, `cap` entries each
  uint32_t *r, *g, *b, *cnt;
  float *err;
  int32_t *cc;
  uint32_t *q;  // rr | gg << 8 | bb << 16
};

__device__ __forceinline__ uint32_t dl3_setrgb(uint32_t r, uint32_t g, uint32_t b, uint32_t cnt) {  // setrgb, :478-484
  const uint32_t v2 = (uint32_t)((int)cnt >> 1);
  return (((r + v2) / cnt) & 255u) | ((((g + v2) / cnt) & 255u) << 8) | ((((b + v2) / cnt) & 255u) << 16);
}

struct Dl3Entry { uint32_t r, g, b, cnt, q; };
__device__ __forceinline__ Dl3Entry dl3_load(const Dl3Table &t, int i) { return Dl3Entry{t.r[i], t.g[i], t.b[i], t.cnt[i], t.q[i]}; }

__device__ __forceinline__ float dl3_calc_err(const Dl3Entry &a, const Dl3Entry &c, const float *__restrict__ sqrt_tab) {  // calc_err, :520-541
  const uint32_t P3 = a.cnt + c.cnt, h = P3 >> 1;
  const int R3 = (int)((a.r + c.r + h) / P3), G3 = (int)((a.g + c.g + h) / P3), B3 = (int)((a.b + c.b + h) / P3);
  const int R1 = a.q & 255, G1 = (a.q >> 8) & 255, B1 = (a.q >> 16) & 255, R2 = c.q & 255, G2 = (c.q >> 8) & 255, B2 = (c.q >> 16) & 255;
  const int s1 = (R3 - R1) * (R3 - R1) + (G3 - G1) * (G3 - G1) + (B3 - B1) * (B3 - B1);
  const int s2 = (R2 - R3) * (R2 - R3) + (G2 - G3) * (G2 - G3) + (B2 - B3) * (B2 - B3);
  // R3 is a byte-range mean of byte-range means only while the 32-bit sums have not wrapped; clamp the index, not the arithmetic
  const float d1 = __fmul_rn(sqrt_tab[min(s1, DL3_SQ_MAX)], (float)a.cnt), d2 = __fmul_rn(sqrt_tab[min(s2, DL3_SQ_MAX)], (float)c.cnt);
  return __fadd_rn(d1, d2);
}

// (error, index) pairs ordered as the reference's scans order them: the smaller error, and among equal errors the lower index
__device__ __forceinline__ bool dl3_less(float e1, int i1, float e2, int i2) { return e1 < e2 || (e1 == e2 && i1 < i2); }
__device__ __forceinline__ void dl3_wave_min(float &e, int &i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float e2 = __shfl_xor(e, o);
    const int i2 = __shfl_xor(i, o);
    if (dl3_less(e2, i2, e, i)) { e = e2; i = i2; }
  }
}

// recount_next (:543-559) by one wave: the first j > i of least calc_err(i, j); an empty range leaves (HUGE_VALF, 0) as there
__device__ __forceinline__ void dl3_recount_next_wave(const Dl3Table &t, int tot, int i, const float *__restrict__ sqrt_tab, int lane) {
  const Dl3Entry a = dl3_load(t, i);
  float e = HUGE_VALF;
  int c = 0x7fffffff;
  for (int j = i + 1 + lane; j < tot; j += 64) {
    const float cur = dl3_calc_err(a, dl3_load(t, j), sqrt_tab);
    if (cur < e) { e = cur; c = j; }
  }
  dl3_wave_min(e, c);
  if (lane == 0) { t.err[i] = e; t.cc[i] = e < HUGE_VALF ? c : 0; }
}

__global__ __launch_bounds__(256) void k_dl3_hist(const uint8_t *__restrict__ rgb, int64_t npixels, int bpc, uint32_t *__restrict__ cell /* [4][lookup_size] */,
                                                  int64_t lookup_size) {
  const int mbpc = (1 << bpc) - 1;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npixels; p += (int64_t)gridDim.x * blockDim.x) {
    const int R = rgb[p * 3], G = rgb[p * 3 + 1], B = rgb[p * 3 + 2];
    const int64_t idx = (int64_t)(B * mbpc / 255) | ((int64_t)(G * mbpc / 255) << bpc) | ((int64_t)(R * mbpc / 255) << (bpc << 1));
    atomicAdd(&cell[idx], (uint32_t)R);
    atomicAdd(&cell[lookup_size + idx], (uint32_t)G);
    atomicAdd(&cell[2 * lookup_size + idx], (uint32_t)B);
    atomicAdd(&cell[3 * lookup_size + idx], 1u);
  }
}

__global__ void k_dl3_flags(const uint32_t *__restrict__ cell, int64_t lookup_size, uint32_t *__restrict__ flag) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < lookup_size; i += (int64_t)gridDim.x * blockDim.x) flag[i] = cell[3 * lookup_size + i] != 0;
}

__global__ void k_dl3_compact(const uint32_t *__restrict__ cell, int64_t lookup_size, const uint32_t *__restrict__ pos, Dl3Table t) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < lookup_size; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t n = cell[3 * lookup_size + i];
    if (!n) continue;
    const uint32_t o = pos[i], r = cell[i], g = cell[lookup_size + i], b = cell[2 * lookup_size + i];
    t.r[o] = r; t.g[o] = g; t.b[o] = b; t.cnt[o] = n;
    t.q[o] = dl3_setrgb(r, g, b, n);
    t.err[o] = HUGE_VALF;
    t.cc[o] = 0;
  }
}

// pass 1 of reduce_table3 (:588-598): recount_next for every entry but the last, which gets (HUGE_VALF, tot)
__global__ __launch_bounds__(256) void k_dl3_pass1(Dl3Table t, int tot, const float *__restrict__ sqrt_tab) {
  const int lane = threadIdx.x & 63, wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * (int64_t)blockDim.x) >> 6);
  // entry i scans tot - 1 - i partners: entries are dealt from both ends so that every wave gets about the same work
  for (int k = wave; k < tot - 1; k += nwaves) {
    const int i = (k & 1) ? tot - 2 - (k >> 1) : (k >> 1);
    dl3_recount_next_wave(t, tot, i, sqrt_tab, lane);
  }
  if (wave == 0 && lane == 0) { t.err[tot - 1] = HUGE_VALF; t.cc[tot - 1] = tot; }
}

// pass 2 of reduce_table3 (:600-643): one workgroup, one merge per trip round the loop
__global__ __launch_bounds__(DL3_NT) void k_dl3_reduce(Dl3Table t, int tot0, int num_colors, const float *__restrict__ sqrt_tab, int *__restrict__ tot_out) {
  __shared__ float s_e[DL3_NT / 64];
  __shared__ int s_i[DL3_NT / 64];
  __shared__ int s_c1, s_c2, s_tot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tot = tot0;
  // the workgroup's recount of ONE entry (its partners spread over all threads): used for c1 and c2 themselves
  auto recount_next_group = [&](int i) {
    const Dl3Entry a = dl3_load(t, i);
    float e = HUGE_VALF;
    int c = 0x7fffffff;
    for (int j = i + 1 + tid; j < tot; j += DL3_NT) {
      const float cur = dl3_calc_err(a, dl3_load(t, j), sqrt_tab);
      if (cur < e) { e = cur; c = j; }
    }
    dl3_wave_min(e, c);
    if (lane == 0) { s_e[wave] = e; s_i[wave] = c; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < DL3_NT / 64; w++)
        if (dl3_less(s_e[w], s_i[w], e, c)) { e = s_e[w]; c = s_i[w]; }
      t.err[i] = e;
      t.cc[i] = e < HUGE_VALF ? c : 0;
    }
    __syncthreads();
  };
  // recount_dist (:561-581): the entry itself, then every earlier entry -- re-scanned if it pointed at `c`, else offered `c` as a partner
  auto recount_dist = [&](int c) {
    recount_next_group(c);
    const Dl3Entry ec = dl3_load(t, c);
    for (int base = wave * 64; base < c; base += DL3_NT) {
      const int i = base + lane;
      bool rescan = false;
      if (i < c) {
        if (t.cc[i] == c) rescan = true;
        else {
          const float cur = dl3_calc_err(dl3_load(t, i), ec, sqrt_tab);
          if (cur < t.err[i]) { t.err[i] = cur; t.cc[i] = c; }
        }
      }
      unsigned long long m = __builtin_amdgcn_ballot_w64(rescan);
      while (m) {  // the wave re-scans them one after the other, all lanes on each
        const int bit = __builtin_ctzll(m);
        m &= m - 1;
        dl3_recount_next_wave(t, tot, base + bit, sqrt_tab, lane);
      }
    }
    __syncthreads();
  };
  while (tot > num_colors) {
    {  // the entry of least error, the first of them (:609-617)
      float e = HUGE_VALF;
      int c = 0x7fffffff;
      for (int i = tid; i < tot; i += DL3_NT) {
        const float v = t.err[i];
        if (v < e) { e = v; c = i; }
      }
      dl3_wave_min(e, c);
      if (lane == 0) { s_e[wave] = e; s_i[wave] = c; }
      __syncthreads();
      if (tid == 0) {
        for (int w = 1; w < DL3_NT / 64; w++)
          if (dl3_less(s_e[w], s_i[w], e, c)) { e = s_e[w]; c = s_i[w]; }
        const int c1 = c == 0x7fffffff ? 0 : c, c2 = t.cc[c1];
        // merge c1 into its partner, move the last entry into c1's place (:618-629)
        const uint32_t r = t.r[c2] + t.r[c1], g = t.g[c2] + t.g[c1], b = t.b[c2] + t.b[c1], n = t.cnt[c2] + t.cnt[c1];
        t.r[c2] = r; t.g[c2] = g; t.b[c2] = b; t.cnt[c2] = n;
        t.q[c2] = dl3_setrgb(r, g, b, n);
        const int nt = tot - 1;
        t.r[c1] = t.r[nt]; t.g[c1] = t.g[nt]; t.b[c1] = t.b[nt]; t.cnt[c1] = t.cnt[nt]; t.q[c1] = t.q[nt]; t.err[c1] = t.err[nt]; t.cc[c1] = t.cc[nt];
        t.err[nt - 1] = HUGE_VALF;
        t.cc[nt - 1] = nt;
        s_c1 = c1; s_c2 = c2; s_tot = nt;
      }
      __syncthreads();
    }
    const int c1 = s_c1, c2 = s_c2;
    tot = s_tot;
    // partners that pointed at the moved entry: earlier entries follow it to c1, later ones look again (:631-639)
    for (int base = wave * 64; base < tot; base += DL3_NT) {
      const int i = base + lane;
      bool rescan = false;
      if (i < tot && t.cc[i] == tot) {
        if (i < c1) t.cc[i] = c1;
        else if (i > c1) rescan = true;
      }
      unsigned long long m = __builtin_amdgcn_ballot_w64(rescan);
      while (m) {
        const int bit = __builtin_ctzll(m);
        m &= m - 1;
        dl3_recount_next_wave(t, tot, base + bit, sqrt_tab, lane);
      }
    }
    __syncthreads();
    recount_dist(c1);                 // (:641)
    if (c2 != tot) recount_dist(c2);  // (:642)
  }
  if (tid == 0) *tot_out = tot;
}

__global__ void k_dl3_palette(Dl3Table t, int tot, int quant_to, uint8_t *__restrict__ pal) {  // set_palette3 (:650-664) + copy_pal
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += gridDim.x * blockDim.x) {
    const uint32_t q = t.q[i];
    pal[i] = (uint8_t)(q & 255);
    pal[quant_to + i] = (uint8_t)((q >> 8) & 255);
    pal[2 * quant_to + i] = (uint8_t)((q >> 16) & 255);
  }
}

}  // namespace

int run_dl3quant(const void *dev_rgb, int64_t npixels, int quant_to, int lookup_bpc, void *dev_pal, int *out_colors, hipStream_t stream) {
  TM_TRY(require_device());
  TM_CHECK(dev_rgb && dev_pal && out_colors, TM_E_INVAL, "dl3quant: null argument");
  TM_CHECK(npixels > 0 && quant_to >= 1 && quant_to <= 65536 && lookup_bpc >= 1 && lookup_bpc <= 8, TM_E_INVAL,
           "dl3quant: %lld pixels to %d colours at %d bits per channel", (long long)npixels, quant_to, lookup_bpc);
  const int64_t lookup_size = (int64_t)1 << (lookup_bpc * 3);
  DevBuf cell, flag, pos, tmp, tab, sq, dtot;
  TM_TRY(cell.alloc((size_t)lookup_size * 16));
  TM_TRY(flag.alloc((size_t)lookup_size * 4));
  TM_TRY(pos.alloc((size_t)lookup_size * 4));
  TM_HIP(hipMemsetAsync(cell.p, 0, (size_t)lookup_size * 16, stream));
  TM_HIP(hipMemsetAsync(dev_pal, 0, (size_t)quant_to * 3, stream));
  hipLaunchKernelGGL(k_dl3_hist, dim3((unsigned)std::min<int64_t>((npixels + 255) / 256, 4096)), dim3(256), 0, stream, (const uint8_t *)dev_rgb, npixels, lookup_bpc,
                     cell.as<uint32_t>(), lookup_size);
  const unsigned g = (unsigned)std::min<int64_t>((lookup_size + 255) / 256, 4096);
  hipLaunchKernelGGL(k_dl3_flags, dim3(g), dim3(256), 0, stream, cell.as<uint32_t>(), lookup_size, flag.as<uint32_t>());
  size_t tb = 0;
  TM_HIP(rocprim::exclusive_scan(nullptr, tb, flag.as<uint32_t>(), pos.as<uint32_t>(), 0u, (size_t)lookup_size, rocprim::plus<uint32_t>(), stream));
  TM_TRY(tmp.alloc(tb));
  TM_HIP(rocprim::exclusive_scan(tmp.p, tb, flag.as<uint32_t>(), pos.as<uint32_t>(), 0u, (size_t)lookup_size, rocprim::plus<uint32_t>(), stream));
  uint32_t last[2] = {0, 0};
  TM_HIP(hipMemcpyAsync(&last[0], pos.as<uint32_t>() + lookup_size - 1, 4, hipMemcpyDeviceToHost, stream));
  TM_HIP(hipMemcpyAsync(&last[1], flag.as<uint32_t>() + lookup_size - 1, 4, hipMemcpyDeviceToHost, stream));
  TM_HIP(hipStreamSynchronize(stream));
  const int tot0 = (int)(last[0] + last[1]);  // build_table3's tot_colors
  TM_CHECK(tot0 >= 1, TM_E_INVAL, "dl3quant: empty histogram");
  TM_TRY(tab.alloc((size_t)tot0 * 28));
  Dl3Table t;
  t.r = tab.as<uint32_t>(); t.g = t.r + tot0; t.b = t.g + tot0; t.cnt = t.b + tot0;
  t.err = reinterpret_cast<float *>(t.cnt + tot0);
  t.cc = reinterpret_cast<int32_t *>(t.err + tot0);
  t.q = reinterpret_cast<uint32_t *>(t.cc + tot0);
  hipLaunchKernelGGL(k_dl3_compact, dim3(g), dim3(256), 0, stream, cell.as<uint32_t>(), lookup_size, pos.as<uint32_t>(), t);
  {  // sqrtf of every possible sum of three squares: the host's is correctly rounded (the device's v_sqrt_f32 is not)
    static std::vector<float> host_sq;
    if (host_sq.empty()) {
      host_sq.resize(DL3_SQ_MAX + 1);
      for (int i = 0; i <= DL3_SQ_MAX; i++) host_sq[i] = sqrtf((float)i);
    }
    TM_TRY(sq.alloc(host_sq.size() * 4));
    TM_HIP(hipMemcpyAsync(sq.p, host_sq.data(), host_sq.size() * 4, hipMemcpyHostToDevice, stream));
  }
  TM_TRY(dtot.alloc(4));
  int tot = tot0;
  if (tot0 > quant_to) {
    hipLaunchKernelGGL(k_dl3_pass1, dim3((unsigned)std::min<int64_t>(((int64_t)tot0 + 3) / 4, 2048)), dim3(256), 0, stream, t, tot0, sq.as<float>());
    hipLaunchKernelGGL(k_dl3_reduce, dim3(1), dim3(DL3_NT), 0, stream, t, tot0, quant_to, sq.as<float>(), dtot.as<int>());
    TM_HIP(hipGetLastError());
    TM_HIP(hipMemcpyAsync(&tot, dtot.p, 4, hipMemcpyDeviceToHost, stream));
    TM_HIP(hipStreamSynchronize(stream));
  }
  hipLaunchKernelGGL(k_dl3_palette, dim3((unsigned)std::max(1, (tot + 255) / 256)), dim3(256), 0, stream, t, tot, quant_to, (uint8_t *)dev_pal);
  TM_HIP(hipGetLastError());
  TM_HIP(hipStreamSynchronize(stream));
  *out_colors = tot;
  return TM_OK;
}

}  // namespace tmx

using namespace tmx;

int tm_stage_dl3quant(const uint8_t *dev_rgb, int64_t npixels, int quant_to, int lookup_bpc, uint8_t *dev_palette, int *out_colors, void *stream) {
  return run_dl3quant(dev_rgb, npixels, quant_to, lookup_bpc, dev_palette, out_colors, (hipStream_t)stream);
}

// the import of extern.pas:196 / quantizer.h:20-21: HOST pointers, userpal[3][PALETTE_MAX] planar; 0 = success
int dl3quant(unsigned char *inbuf, int width, int height, int quant_to, int lookup_bpc, unsigned char userpal[3][65536]) {
  if (!inbuf || !userpal || width <= 0 || height <= 0) { set_error("dl3quant: bad arguments"); return 1; }
  const int64_t n = (int64_t)width * height;
  DevBuf rgb, pal;
  if (rgb.alloc((size_t)n * 3) != TM_OK || pal.alloc((size_t)std::max(quant_to, 1) * 3) != TM_OK) return 1;
  if (hipMemcpy(rgb.p, inbuf, (size_t)n * 3, hipMemcpyHostToDevice) != hipSuccess) { set_error("dl3quant: upload failed"); return 1; }
  int tot = 0;
  if (run_dl3quant(rgb.p, n, quant_to, lookup_bpc, pal.p, &tot, nullptr) != TM_OK) return 1;
  std::vector<uint8_t> h((size_t)quant_to * 3);
  if (hipMemcpy(h.data(), pal.p, h.size(), hipMemcpyDeviceToHost) != hipSuccess) { set_error("dl3quant: read-back failed"); return 1; }
  for (int c = 0; c < 3; c++) memcpy(userpal[c], h.data() + (size_t)c * quant_to, (size_t)tot);
  return 0;
}
