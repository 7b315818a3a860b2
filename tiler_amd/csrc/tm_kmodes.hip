// tm_kmodes.hip -- A17: TKModes.ComputeKModes (kmodes.pas:923-1094) for rows of cKModesFeatureCount = 80 bytes.
//
// Unreachable in the reference snapshot (nothing calls it) but named by the north star, so it is built as an operator of its
// own: tm_stage_kmodes.  What is parallel in the algorithm runs on the GPU --
//   * MatchingDissim (kmodes.pas:248-259) = sum |a - b| + 2048 x #(a != b) over the 80 bytes of a row: twenty v_sad_u8
//     (four bytes each) plus a count of the non-zero bytes of a XOR b, the GPU form of the reference's psadbw / pcmpeqb / popcnt
//     (kmodes.pas:314-450);
//   * GetMinMatchingDissim for a range of points against the k modes (the LAST minimum wins, `dis <= best`);
//   * the farthest-first initialisation's min-distance update and its pick (the LAST largest among unused points wins, 757-762)
// -- and what is inherently serial stays on the host exactly as written: Huang's online mode update MovePointCat (774-803) one
// moved point after the other, the empty-cluster repair with the LCG RandInt (88-92), the stopping rule with its three graces.
// KModesIter (851-921) scores a bin of 960 points against the modes as they stand at the start of the bin; here all remaining
// points are scored in one launch and the launch is repeated from the next bin on only after a bin that changed a mode, which is
// the same thing (a score only depends on the modes).
#include <algorithm>
#include <climits>
#include <cmath>
#include <vector>

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {
namespace {

typedef unsigned long long u64;
constexpr int KM_ATTRS = 80, KM_WORDS = 20;

__device__ __forceinline__ unsigned km_dissim(const uint32_t *__restrict__ a, const uint32_t (&b)[KM_WORDS]) {
  unsigned sad = 0, zero_bytes = 0;
#pragma unroll
  for (int w = 0; w < KM_WORDS; w++) {
    const uint32_t x = a[w], y = b[w];
    sad = __builtin_amdgcn_sad_u8(x, y, sad);
    const uint32_t d = x ^ y;
    // 0x80 in every byte of d that is zero
    zero_bytes += __builtin_popcount(~(((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d | 0x7f7f7f7fu));
  }
  return sad + ((unsigned)(KM_ATTRS - zero_bytes) << 11);  // at most 80 * (255 + 2048): 32 bits are plenty
}

// clust[i], dis[i] for points [first, n): the LAST mode reaching the minimum wins (kmodes.pas:272, 414)
__global__ __launch_bounds__(256) void k_kmodes_argmin(const uint32_t *__restrict__ rows, int64_t first, int64_t n, const uint32_t *__restrict__ modes, int k,
                                                       int32_t *__restrict__ clust, unsigned *__restrict__ dis) {
  extern __shared__ uint32_t s_modes[];  // [k][20]
  for (int e = threadIdx.x; e < k * KM_WORDS; e += 256) s_modes[e] = modes[e];
  __syncthreads();
  for (int64_t i = first + blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    uint32_t p[KM_WORDS];
#pragma unroll
    for (int w = 0; w < KM_WORDS; w += 4) {
      const uint4 v = *reinterpret_cast<const uint4 *>(rows + i * KM_WORDS + w);
      p[w] = v.x; p[w + 1] = v.y; p[w + 2] = v.z; p[w + 3] = v.w;
    }
    unsigned best = 0xffffffffu;
    int res = -1;
    for (int c = 0; c < k; c++) {
      const unsigned d = km_dissim(s_modes + c * KM_WORDS, p);
      if (d <= best) { best = d; res = c; }
    }
    clust[i] = res;
    dis[i] = best;
  }
}

// farthest-first: mind[i] = min(mind[i], dissim(centre, row i)) for unused points, then this block's candidate for the next pick:
// the LAST index among the unused points with the largest mind (kmodes.pas:757-762) -> key = mind << 32 | index, maximum
__global__ __launch_bounds__(256) void k_kmodes_ff(const uint32_t *__restrict__ rows, int64_t n, int64_t centre, const uint8_t *__restrict__ used,
                                                   unsigned *__restrict__ mind, u64 *__restrict__ partial) {
  __shared__ u64 s_best[4];
  uint32_t c[KM_WORDS];
#pragma unroll
  for (int w = 0; w < KM_WORDS; w++) c[w] = rows[centre * KM_WORDS + w];
  u64 best = 0;
  bool any = false;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (used[i] || i == centre) continue;
    const unsigned d = km_dissim(rows + i * KM_WORDS, c);
    unsigned m = mind[i];
    if (d < m) { m = d; mind[i] = m; }
    const u64 key = ((u64)m << 32) | (u64)(uint32_t)i;
    if (!any || key >= best) { best = key; any = true; }
  }
  u64 enc = any ? best + 1 : 0;  // 0 = no candidate; keys shifted by one so that (mind 0, index 0) is still a candidate
  for (int o = 32; o > 0; o >>= 1) { const u64 other = __shfl_xor(enc, o); enc = other > enc ? other : enc; }
  if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = enc;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) enc = s_best[w] > enc ? s_best[w] : enc;
    partial[blockIdx.x] = enc;
  }
}

uint32_t rand_int(uint32_t range, uint32_t &seed) {  // kmodes.pas:88-92
  seed = (uint32_t)((int32_t)(seed * 0x08088405u) + 1);
  return (uint32_t)(((uint64_t)seed * (uint64_t)range) >> 32);
}

struct Modes {
  const uint8_t *x; int64_t n; int k, nmod;
  std::vector<int32_t> memb, freq;   // freq [k][80][nmod]
  std::vector<int64_t> members;
  std::vector<uint8_t> cent;
  bool cent_changed = false;
  static int max_index(const int32_t *arr, int n) {  // GetMaxValueIndex, kmodes.pas:155-167: first largest
    int res = -1, best = INT_MIN;
    for (int i = 0; i < n; i++) if (arr[i] > best) { best = arr[i]; res = i; }
    return res;
  }
  void move(int64_t ipoint, int to, int from) {  // MovePointCat, kmodes.pas:774-803
    const uint8_t *p = x + ipoint * KM_ATTRS;
    memb[(size_t)ipoint] = to;
    members[(size_t)to]++;
    members[(size_t)from]--;
    for (int a = 0; a < KM_ATTRS; a++) {
      const int cur = p[a];
      int32_t *tc = &freq[((size_t)to * KM_ATTRS + a) * nmod], *fc = &freq[((size_t)from * KM_ATTRS + a) * nmod];
      tc[cur]++;
      uint8_t &tcent = cent[(size_t)to * KM_ATTRS + a];
      if (tc[tcent] < tc[cur]) { tcent = (uint8_t)cur; cent_changed = true; }
      fc[cur]--;
      uint8_t &fcent = cent[(size_t)from * KM_ATTRS + a];
      if (fcent == cur) {
        const uint8_t nv = (uint8_t)max_index(fc, nmod);
        if (nv != fcent) { fcent = nv; cent_changed = true; }
      }
    }
  }
};

}  // namespace

int run_kmodes(const uint8_t *rows, int64_t n, int k, int num_init, int nmod, int max_iter, int32_t *labels_out, uint8_t *cent_out, uint64_t *cost_out,
               int *iters_out, hipStream_t stream) {
  TM_TRY(require_device());
  TM_CHECK(rows && labels_out && cent_out, TM_E_INVAL, "kmodes: null argument");
  TM_CHECK(n >= 1 && n < (1ll << 31), TM_E_INVAL, "kmodes: %lld points", (long long)n);
  TM_CHECK(k >= 1 && k <= 4096, TM_E_INVAL, "kmodes: %d clusters outside 1..4096", k);
  TM_CHECK(nmod >= 1 && nmod <= 256, TM_E_INVAL, "kmodes: %d modalities outside 1..256", nmod);
  for (int64_t i = 0; i < n * KM_ATTRS; i++) TM_CHECK(rows[i] < nmod, TM_E_INVAL, "kmodes: value %d at byte %lld is not below the %d modalities", rows[i], (long long)i, nmod);
  if (max_iter < 0) max_iter = INT_MAX;
  const int nruns = num_init <= 0 ? 1 : num_init;
  std::vector<int64_t> starts((size_t)nruns);
  if (num_init <= 0) {
    starts[0] = -(int64_t)num_init;
  } else {  // 952-964: golden-ratio-like spread of the starting points, Single arithmetic
    const float ratio = (float)std::pow((double)n, 1.0 / (double)num_init);
    float acc = 1.0f;
    for (int i = 0; i < nruns; i++) {
      starts[(size_t)i] = (int64_t)std::nearbyint((double)acc) - 1;  // Round: half to even
      if (i > 0 && starts[(size_t)i] <= starts[(size_t)i - 1]) starts[(size_t)i] = std::min(n - 1, starts[(size_t)i - 1] + 1);
      acc = acc * ratio;
    }
  }
  for (int64_t sp : starts) TM_CHECK(sp >= 0 && sp < n, TM_E_INVAL, "kmodes: starting point %lld outside the %lld points", (long long)sp, (long long)n);
  DevBuf drows, dmodes, dclust, ddis, dused, dmind, dpartial;
  const int nblk = (int)std::min<int64_t>((n + 255) / 256, 2048);
  TM_TRY(drows.alloc((size_t)n * KM_ATTRS)); TM_TRY(dmodes.alloc((size_t)k * KM_ATTRS)); TM_TRY(dclust.alloc((size_t)n * 4)); TM_TRY(ddis.alloc((size_t)n * 4));
  TM_TRY(dused.alloc((size_t)n)); TM_TRY(dmind.alloc((size_t)n * 4)); TM_TRY(dpartial.alloc((size_t)nblk * 8));
  TM_HIP(hipMemcpyAsync(drows.p, rows, (size_t)n * KM_ATTRS, hipMemcpyHostToDevice, stream));
  Modes s;
  s.x = rows; s.n = n; s.k = k; s.nmod = nmod;
  s.memb.assign((size_t)n, -1);
  s.members.assign((size_t)k, 0);
  s.cent.assign((size_t)k * KM_ATTRS, 0xff);
  s.freq.assign((size_t)k * KM_ATTRS * nmod, 0);
  std::vector<int32_t> clust((size_t)n), bestm((size_t)n);
  std::vector<unsigned> dis((size_t)n);
  std::vector<uint8_t> used((size_t)n), bestc((size_t)k * KM_ATTRS);
  std::vector<u64> partial((size_t)nblk);
  const size_t lds = (size_t)k * KM_ATTRS;
  TM_CHECK(lds <= 160 * 1024 - 1024, TM_E_INVAL, "kmodes: the modes of %d clusters do not fit LDS", k);
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_kmodes_argmin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  auto score_from = [&](int64_t first) -> int {  // clust / dis of points [first, n) against the modes as they stand
    TM_HIP(hipMemcpyAsync(dmodes.p, s.cent.data(), (size_t)k * KM_ATTRS, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_kmodes_argmin, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n - first + 255) / 256, 2048))), dim3(256), lds, stream,
                       drows.as<uint32_t>(), first, n, dmodes.as<uint32_t>(), k, dclust.as<int32_t>(), ddis.as<unsigned>());
    TM_HIP(hipGetLastError());
    TM_HIP(hipMemcpyAsync(clust.data() + first, dclust.as<int32_t>() + first, (size_t)(n - first) * 4, hipMemcpyDeviceToHost, stream));
    TM_HIP(hipMemcpyAsync(dis.data() + first, ddis.as<unsigned>() + first, (size_t)(n - first) * 4, hipMemcpyDeviceToHost, stream));
    TM_HIP(hipStreamSynchronize(stream));
    s.cent_changed = false;
    return TM_OK;
  };
  uint32_t seed = 0x42381337u;  // 933
  uint64_t all_best = ~0ull;
  int all_iters = 0;
  for (int run = 0; run < nruns; run++) {
    // ---- InitFarthestFirst (694-772)
    std::fill(used.begin(), used.end(), 0);
    std::fill(s.cent.begin(), s.cent.end(), 0xff);
    TM_HIP(hipMemsetAsync(dmind.p, 0xff, (size_t)n * 4, stream));
    TM_HIP(hipMemsetAsync(dused.p, 0, (size_t)n, stream));
    int64_t far = starts[(size_t)run];
    for (int c = 0; c < k; c++) {
      memcpy(&s.cent[(size_t)c * KM_ATTRS], rows + far * KM_ATTRS, KM_ATTRS);
      used[(size_t)far] = 1;
      const uint8_t one = 1;
      TM_HIP(hipMemcpyAsync(dused.as<uint8_t>() + far, &one, 1, hipMemcpyHostToDevice, stream));
      if (c == k - 1) break;
      hipLaunchKernelGGL(k_kmodes_ff, dim3(nblk), dim3(256), 0, stream, drows.as<uint32_t>(), n, far, dused.as<uint8_t>(), dmind.as<unsigned>(), dpartial.as<u64>());
      TM_HIP(hipGetLastError());
      TM_HIP(hipMemcpyAsync(partial.data(), dpartial.p, (size_t)nblk * 8, hipMemcpyDeviceToHost, stream));
      TM_HIP(hipStreamSynchronize(stream));
      u64 best = 0;
      for (u64 v : partial) best = std::max(best, v);
      far = best ? (int64_t)(uint32_t)(best - 1) : starts[(size_t)run];  // no unused point left: ifarthest stays InitPoint (756)
    }
    // ---- initial assignment and modes (978-1011)
    std::fill(s.freq.begin(), s.freq.end(), 0);
    std::fill(s.members.begin(), s.members.end(), 0);
    TM_TRY(score_from(0));
    for (int64_t i = 0; i < n; i++) {
      const int c = clust[(size_t)i];
      s.memb[(size_t)i] = c;
      s.members[(size_t)c]++;
      for (int a = 0; a < KM_ATTRS; a++) s.freq[((size_t)c * KM_ATTRS + a) * nmod + rows[i * KM_ATTRS + a]]++;
    }
    for (int c = 0; c < k; c++) {
      if (s.members[(size_t)c] == 0) {
        for (int a = 0; a < KM_ATTRS; a++) s.cent[(size_t)c * KM_ATTRS + a] = rows[(int64_t)rand_int((uint32_t)n, seed) * KM_ATTRS + a];
      } else {
        for (int a = 0; a < KM_ATTRS; a++) s.cent[(size_t)c * KM_ATTRS + a] = (uint8_t)Modes::max_index(&s.freq[((size_t)c * KM_ATTRS + a) * nmod], nmod);
      }
    }
    int itr = 0, worse = 0, bestitr = 0;
    bool converged = false;
    uint64_t prevcost = ~0ull, bestcost = ~0ull;
    while (itr < max_iter && !converged) {
      itr++;
      // ---- KModesIter (851-921)
      int moves = 0;
      uint64_t cost = 0;
      TM_TRY(score_from(0));
      for (int64_t b0 = 0; b0 < n; b0 += 960) {
        const int64_t b1 = std::min<int64_t>(b0 + 960, n);
        if (s.cent_changed) TM_TRY(score_from(b0));  // a mode moved since the scores were taken: the bins from here on see the new modes
        for (int64_t i = b0; i < b1; i++) {
          cost += dis[(size_t)i];
          if (s.memb[(size_t)i] != clust[(size_t)i]) {
            moves++;
            const int old = s.memb[(size_t)i];
            s.move(i, clust[(size_t)i], old);
            if (s.members[(size_t)old] == 0) {  // CountClusterMembers(old_clust) = 0: refill it from the largest cluster (the LAST largest, 667)
              int from = 0;
              int64_t mc = 0;
              for (int c = 0; c < k; c++) if (s.members[(size_t)c] >= mc) { mc = s.members[(size_t)c]; from = c; }
              const uint32_t pick = rand_int((uint32_t)mc, seed);
              int64_t r = -1, cnt = 0;
              for (int64_t j = 0; j < n; j++) if (s.memb[(size_t)j] == from) { if (cnt == (int64_t)pick) { r = j; break; } cnt++; }
              TM_CHECK(r >= 0, TM_E_INVAL, "kmodes: empty-cluster repair found no donor");
              s.move(r, old, from);
            }
          }
        }
      }
      converged = cost >= prevcost;
      if (converged) {  // SameValue(cost, prevcost, prevcost div 1000), 1041
        const double a = (double)cost, b = (double)prevcost;
        double eps = (double)(prevcost / 1000);
        if (eps == 0) eps = std::max(std::min(std::fabs(a), std::fabs(b)) * 1e-12, 1e-12);
        const bool same = a > b ? (a - b) <= eps : (b - a) <= eps;
        if (same) { worse++; if (worse < 3) converged = false; }
      }
      converged = converged || moves == 0;
      if (cost < bestcost) { bestitr = itr; bestcost = cost; bestm = s.memb; bestc = s.cent; }
      prevcost = cost;
    }
    if (bestcost < all_best) {  // 1078-1085: the first run with the strictly smallest cost
      all_best = bestcost;
      all_iters = bestitr;
      memcpy(labels_out, bestm.data(), (size_t)n * 4);
      memcpy(cent_out, bestc.data(), (size_t)k * KM_ATTRS);
    }
  }
  if (cost_out) *cost_out = all_best;
  if (iters_out) *iters_out = all_iters;
  return TM_OK;
}

}  // namespace tmx

extern "C" int tm_stage_kmodes(const uint8_t *host_rows, int64_t n, int num_clusters, int num_init, int num_modalities, int max_iter, int32_t *host_labels,
                               uint8_t *host_centroids, uint64_t *host_cost, int *host_iters, void *stream) {
  return tmx::run_kmodes(host_rows, n, num_clusters, num_init, num_modalities, max_iter, host_labels, host_centroids, host_cost, host_iters, (hipStream_t)stream);
}
