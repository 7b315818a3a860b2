// tm_kmeans.hip -- the build's deterministic k-means, A9/A10 (replaces BICO.dll + ANN.dll + yakmo.dll; the DLLs'
// RNG/tie behaviour is not recoverable, SURVEY.md section 8c, so the algorithm below IS the specification and the
// oracle (oracle/tm_oracle.c: tmo_kmeans_i32) states the same thing for the CPU):
//   * points are int32 vectors with uint32 weights, grouped in contiguous segments (one independent problem each);
//   * init = farthest-first from the segment's first point, exact int64 distances, ties -> lowest index
//     (cf. TKModes.InitFarthestFirst, kmodes.pas:694); stops early when no distinct point is left;
//   * Lloyd: distance = sum over dimensions in order of (p - c)^2 in IEEE double (no FMA), ties -> lowest centroid;
//     centroid = exact integer weighted sum / weight (one IEEE division); empty clusters keep their centroid;
//     stop when no assignment changes or after max_iter (cYakmoMaxIterations = 300, utils.pas:17).
// Integer sums are order independent, so the parallel reduction is bit-reproducible.
//
// Callers: tm_stage_kmeans (one segment), run_palettize (DoPalettization, tilingencoder.pas:4105-4245, D = 192) and
// run_quantize_palettes (QuantizeUsingYakmo + DoQuantization, 4434-4564, D = 3, one segment per palette, run on the
// (G,R,B)-sorted colour histogram of each palette's pixels).
#include <type_traits>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <cstdlib>
#include <climits>
#include <vector>

#include "tm_common.h"
#include "tm_internal.h"

namespace tmx {

typedef unsigned long long u64;

// One resident launch at a time per process: the resident kernels (k_h_resident, k_kmeans3_persistent) need all their workgroups on the chip
// together, and two of them started from two host threads could each be dealt part of the CUs and wait for the rest for ever (until their
// barriers give up).  Held from the launch to the read-back that ends it.  (Two PROCESSES on one device are not covered: there the barrier's
// time limit and the launches-per-iteration path are the answer -- a development set-up, one process per device is the deployment.)
static std::mutex g_resident_launch;

struct Seg {      // per segment state, device resident
  int64_t begin;  // first point
  int64_t count;  // number of points
  int kk;         // live centroids so far
  int init_done;
  int64_t cur;    // point index chosen as the newest centroid
  int changed;
  int nseg;       // element 0 only: number of segments
  int blk_first;  // 1-D grids: first workgroup of this segment and how many it owns (proportional to its size)
  int blk_count;
};

// 1-D grid -> (segment, workgroup index inside it, workgroups it owns)
__device__ __forceinline__ int find_seg(const Seg *__restrict__ segs, int &bx, int &nbx) {
  int lo = 0, hi = segs[0].nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].blk_first <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  bx = (int)blockIdx.x - segs[lo].blk_first;
  nbx = segs[lo].blk_count;
  return lo;
}

// ---- farthest-first ------------------------------------------------------------------------------------------
// best key: larger mindist wins, then lower index.  mindist can reach 2^38 (D=192), indices 2^31: two words.
struct BestKey { long long dist; long long negidx; };
struct FfCandOut { long long *dist, *gidx; int32_t *row; };  // where a process writes its candidate (FfCand's fields)
__device__ __forceinline__ bool better(const BestKey &a, const BestKey &b) {
  return a.dist > b.dist || (a.dist == b.dist && a.negidx > b.negidx);
}

template <int D>
__global__ __launch_bounds__(256) void k_ff_update(const int32_t *__restrict__ pts, Seg *__restrict__ segs, int k,
                                                   long long *__restrict__ mind, BestKey *__restrict__ partial) {
  __shared__ BestKey s_best[4];
  int bx, nbx;
  const int seg = find_seg(segs, bx, nbx);
  if (nbx <= 0) return;  // only when every segment is empty
  const Seg sg = segs[seg];
  BestKey mine{0, LLONG_MIN};
  if (!sg.init_done && sg.kk <= k) {
    int32_t c[D];
#pragma unroll
    for (int j = 0; j < D; j++) c[j] = pts[sg.cur * D + j];  // wave-uniform: scalar loads
    for (int64_t i = bx * 256 + threadIdx.x; i < sg.count; i += (int64_t)nbx * 256) {
      const int32_t *p = pts + (sg.begin + i) * D;
      long long d = 0;
#pragma unroll
      for (int j = 0; j < D; j++) { const long long t = (long long)p[j] - c[j]; d += t * t; }
      long long m = mind[sg.begin + i];
      if (d < m) { m = d; mind[sg.begin + i] = m; }
      const BestKey cand{m, -(long long)i};
      if (better(cand, mine)) mine = cand;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    BestKey other{__shfl_xor(mine.dist, o), __shfl_xor(mine.negidx, o)};
    if (better(other, mine)) mine = other;
  }
  if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++)
      if (better(s_best[w], mine)) mine = s_best[w];
    partial[blockIdx.x] = mine;
  }
}

// D=192 variant: the centre row does not fit registers as scalars cheaply; read it through LDS.
__global__ __launch_bounds__(256) void k_ff_update_wide(const int32_t *__restrict__ pts, int d, Seg *__restrict__ segs, int k,
                                                        long long *__restrict__ mind, BestKey *__restrict__ partial) {
  __shared__ BestKey s_best[4];
  __shared__ int32_t s_c[256];
  int bx, nbx;
  const int seg = find_seg(segs, bx, nbx);
  if (nbx <= 0) return;  // only when every segment is empty
  const Seg sg = segs[seg];
  BestKey mine{0, LLONG_MIN};
  const bool active = !sg.init_done && sg.kk <= k;
  if (active)
    for (int j = threadIdx.x; j < d; j += 256) s_c[j] = pts[sg.cur * d + j];
  __syncthreads();
  if (active) {
    for (int64_t i = bx * 256 + threadIdx.x; i < sg.count; i += (int64_t)nbx * 256) {
      const int4 *p = reinterpret_cast<const int4 *>(pts + (sg.begin + i) * d);
      long long dd = 0;
      for (int j = 0; j < d / 4; j++) {
        const int4 v = p[j];
        const long long t0 = (long long)v.x - s_c[4 * j], t1 = (long long)v.y - s_c[4 * j + 1];
        const long long t2 = (long long)v.z - s_c[4 * j + 2], t3 = (long long)v.w - s_c[4 * j + 3];
        dd += t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3;
      }
      long long m = mind[sg.begin + i];
      if (dd < m) { m = dd; mind[sg.begin + i] = m; }
      const BestKey cand{m, -(long long)i};
      if (better(cand, mine)) mine = cand;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    BestKey other{__shfl_xor(mine.dist, o), __shfl_xor(mine.negidx, o)};
    if (better(other, mine)) mine = other;
  }
  if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++)
      if (better(s_best[w], mine)) mine = s_best[w];
    partial[blockIdx.x] = mine;
  }
}

// one thread per segment: fold block partials, append the next centre (or finish)
// one workgroup per segment: the blocks' partial bests are reduced by 256 threads (`better` is a total order, so the reduction tree
// gives the same winner as a serial scan), then the picked point becomes the next centre
__global__ __launch_bounds__(256) void k_ff_pick(Seg *__restrict__ segs, int nseg, int k, const BestKey *__restrict__ partial, int nblk,
                                                 const int32_t *__restrict__ pts, int d, double *__restrict__ cent) {
  __shared__ BestKey s_best[256];
  __shared__ int64_t s_cur;
  const int seg = blockIdx.x, tid = threadIdx.x;
  if (seg >= nseg) return;
  Seg sg = segs[seg];
  if (sg.init_done) return;  // uniform over the workgroup
  BestKey best{0, LLONG_MIN};
  for (int b = sg.blk_first + tid; b < sg.blk_first + sg.blk_count; b += 256) {
    const BestKey c = partial[b];
    if (better(c, best)) best = c;
  }
  s_best[tid] = best;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o && better(s_best[tid + o], s_best[tid])) s_best[tid] = s_best[tid + o];
    __syncthreads();
  }
  best = s_best[0];
  const bool done = sg.kk >= k || best.dist <= 0;  // enough centres, or no distinct point left
  if (tid == 0) s_cur = done ? -1 : sg.begin + (-best.negidx);
  __syncthreads();
  if (!done) {
    const int64_t cur = s_cur;
    for (int j = tid; j < d; j += 256) cent[((int64_t)seg * k + sg.kk) * d + j] = (double)pts[cur * d + j];
  }
  if (tid == 0) {
    if (done) sg.init_done = 1;
    else { sg.cur = s_cur; sg.kk++; }
    segs[seg] = sg;
  }
}

__global__ void k_ff_first(Seg *__restrict__ segs, int nseg, int k, const int32_t *__restrict__ pts, int d, double *__restrict__ cent) {
  const int seg = blockIdx.x * blockDim.x + threadIdx.x;
  if (seg >= nseg) return;
  Seg sg = segs[seg];
  sg.kk = 0;
  sg.init_done = 0;
  sg.changed = 0;
  if (sg.count <= 0 || k <= 0) {
    sg.init_done = 1;
  } else {
    sg.cur = sg.begin;
    for (int j = 0; j < d; j++) cent[((int64_t)seg * k) * d + j] = (double)pts[sg.cur * d + j];
    sg.kk = 1;
  }
  segs[seg] = sg;
}

// ---- Lloyd ---------------------------------------------------------------------------------------------------
constexpr int KCH = 16;  // centroids scored per pass (register accumulators)

// Assignment step for D = 3 (pixel colours; D = 192 has its own kernel, k_assign192): points are read directly and, with
// FUSE_ACC, the exact integer sums of the new assignment are accumulated in LDS in the same pass.
template <int D, bool FUSE_ACC>
__global__ __launch_bounds__(256) void k_assign(const int32_t *__restrict__ pts, const uint32_t *__restrict__ w, Seg *__restrict__ segs,
                                                int k, const double *__restrict__ cent, int32_t *__restrict__ assign,
                                                u64 *__restrict__ sums, u64 *__restrict__ cnts, const int *__restrict__ quiet) {
  if (*quiet >= 0) return;  // converged earlier in this batch of launches (the host polls every few iterations)
  extern __shared__ double s_dyn[];
  // [KCH][3] centroid chunk (double) | [NCOPY][kk][4] u64 partial sums (FUSE_ACC)
  static_assert(D == 3, "k_assign is the pixel kernel");
  double *s_cent = s_dyn;
  u64 *s_acc = reinterpret_cast<u64 *>(s_dyn + KCH * D);
  int bx, nbx;
  const int seg = find_seg(segs, bx, nbx);
  if (nbx <= 0) return;  // only when every segment is empty
  const Seg sg = segs[seg];
  const int kk = sg.kk;
  int changed = 0;
  constexpr int NCOPY = 16;  // private copies of the LDS sums (lane & 15): same-cluster lanes no longer serialise on one word
  if (FUSE_ACC) {
    for (int e = threadIdx.x; e < NCOPY * kk * (D + 1); e += 256) s_acc[e] = 0;
  }
  const int64_t iters = (sg.count + (int64_t)nbx * 256 - 1) / ((int64_t)nbx * 256);
  const bool cent_resident = D == 3 && kk <= KCH;  // the usual palette size: the centroids are staged once, not once per 256 points
  if (cent_resident) {
    for (int e = threadIdx.x; e < kk * D; e += 256) s_cent[e] = cent[((int64_t)seg * k) * D + e];
    __syncthreads();
  }
  // the next point's loads are issued before the current one is scored (each thread walks ~9 points; without this every step
  // waits out a full memory round trip)
  int32_t n3[3] = {0, 0, 0};
  long long nwi = 1;
  int nold = -1;
  bool nvalid = false;
  auto fetch = [&](int64_t it) {
    const int64_t i = (it * nbx + bx) * 256 + threadIdx.x;
    nvalid = it < iters && i < sg.count;
    if (nvalid) {
      n3[0] = pts[(sg.begin + i) * 3]; n3[1] = pts[(sg.begin + i) * 3 + 1]; n3[2] = pts[(sg.begin + i) * 3 + 2];
      nwi = w ? (long long)w[sg.begin + i] : 1;
      nold = assign[sg.begin + i];
    }
  };
  fetch(0);
  for (int64_t it = 0; it < iters; it++) {
    const int64_t base = (it * nbx + bx) * 256;
    const int64_t i = base + threadIdx.x;
    const bool valid = nvalid;
    const int32_t p3[3] = {n3[0], n3[1], n3[2]};
    const long long cur_w = nwi;
    const int cur_old = nold;
    fetch(it + 1);
    double bd = 0.0;
    int bc = -1;
    for (int c0 = 0; c0 < kk; c0 += KCH) {
      const int nc = min(KCH, kk - c0);
      double s[KCH];
#pragma unroll
      for (int c = 0; c < KCH; c++) s[c] = 0.0;
      if (D == 3) {
        if (!cent_resident) {
          __syncthreads();
          for (int e = threadIdx.x; e < nc * D; e += 256) s_cent[e] = cent[((int64_t)seg * k + c0) * D + e];
          __syncthreads();
        }
        if (valid) {
          if (nc == KCH) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
              const double pj = (double)p3[j];
#pragma unroll
              for (int c = 0; c < KCH; c++) { const double t = __dsub_rn(pj, s_cent[c * 3 + j]); s[c] = __fma_rn(t, t, s[c]); }
            }
          } else {
            for (int j = 0; j < 3; j++) {
              const double pj = (double)p3[j];
              for (int c = 0; c < nc; c++) { const double t = __dsub_rn(pj, s_cent[c * 3 + j]); s[c] = __fma_rn(t, t, s[c]); }
            }
          }
        }
      }
      if (valid) {
        if (nc == KCH) {
#pragma unroll
          for (int c = 0; c < KCH; c++)
            if (bc < 0 || s[c] < bd) { bd = s[c]; bc = c0 + c; }
        } else {
          for (int c = 0; c < nc; c++)
            if (bc < 0 || s[c] < bd) { bd = s[c]; bc = c0 + c; }
        }
      }
    }
    if (valid) {
      if (cur_old != bc) { assign[sg.begin + i] = bc; changed++; }
      if (FUSE_ACC && cur_old != bc) {
        // The exact integer sums are carried from iteration to iteration, as in k_assign192: only a point that changes cluster touches
        // them -- its weighted row is added to the new cluster and subtracted from the old one (u64 arithmetic: exact, order-free).
        // After the first few iterations almost no point moves, and the LDS atomics, which bounded this kernel, are gone.
        const long long wi = cur_w;
        u64 *acc = s_acc + ((threadIdx.x & (NCOPY - 1)) * kk + bc) * (D + 1);
        atomicAdd(&acc[D], (u64)wi);
#pragma unroll
        for (int j = 0; j < 3; j++) atomicAdd(&acc[j], (u64)(wi * p3[j]));
        if (cur_old >= 0) {
          u64 *old = s_acc + ((threadIdx.x & (NCOPY - 1)) * kk + cur_old) * (D + 1);
          atomicAdd(&old[D], (u64)0 - (u64)wi);
#pragma unroll
          for (int j = 0; j < 3; j++) atomicAdd(&old[j], (u64)0 - (u64)(wi * p3[j]));
        }
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) changed += __shfl_xor(changed, o);
  if ((threadIdx.x & 63) == 0 && changed) atomicAdd(&segs[seg].changed, changed);
  if (FUSE_ACC) {
    __syncthreads();
    for (int e = threadIdx.x; e < kk * (D + 1); e += 256) {
      u64 v = 0;
#pragma unroll
      for (int cp = 0; cp < NCOPY; cp++) v += s_acc[cp * kk * (D + 1) + e];
      if (v == 0) continue;
      const int c = e / (D + 1), j = e - c * (D + 1);
      if (j == D) atomicAdd(&cnts[(int64_t)seg * k + c], v);
      else atomicAdd(&sums[((int64_t)seg * k + c) * D + j], v);
    }
  }
}

// Assignment step for D = 192, register-tiled: every thread scores PPT points against 16 centroids at a time, so each
// centroid value fetched from LDS (a wave-wide broadcast) feeds PPT x 3 double-precision operations instead of 3 -- the
// untiled form is bound by LDS return bandwidth, not by the FP64 pipe.  The arithmetic per (point, centroid) is unchanged:
// sum over dimensions in order of (p - c)^2, one IEEE subtraction and one fused multiply-add each.  One workgroup per CU-sized slice of the points
// (rows_per_block <= 256 * PPT, chosen by the host so that the slices fill the chip evenly); the next 8-dimension chunk is
// fetched into registers while the current one is being scored.
// The exact integer sums are carried from iteration to iteration: a point that changes cluster adds its row to the new
// cluster and subtracts it from the old one (u64 arithmetic: exact and order-free), all threads of the workgroup
// cooperating on one moved row at a time (coalesced read, one dimension per thread), accumulated in LDS and flushed once.
constexpr int A_DCH = 8;   // dimensions staged per pass
template <int PPT>
__global__ __launch_bounds__(256) void k_assign192(const int32_t *__restrict__ pts, const int32_t *__restrict__ pts_chunked, int64_t n_total,
                                                   const uint32_t *__restrict__ w, Seg *__restrict__ segs,
                                                   int k, const double *__restrict__ cent, int32_t *__restrict__ assign,
                                                   u64 *__restrict__ sums, u64 *__restrict__ cnts, int rows_per_block, int lds_delta,
                                                   const int *__restrict__ quiet, double *__restrict__ ub = nullptr, double *__restrict__ lb = nullptr) {
  if (*quiet >= 0) return;  // converged earlier in this batch of launches
  constexpr int D = 192, ROWS = 256 * PPT, PITCH = A_DCH + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  double(*s_cent)[KCH] = reinterpret_cast<double(*)[KCH]>(s_raw);                       // [A_DCH][KCH]
  int32_t *s_pts = reinterpret_cast<int32_t *>(s_raw + A_DCH * KCH * 8);               // [ROWS][PITCH]
  int32_t *s_moved = s_pts + ROWS * PITCH;                                             // [ROWS][3]: row, old, new
  u64 *s_delta = reinterpret_cast<u64 *>(s_moved + ROWS * 3 + (ROWS & 1));             // [kk][D+1] when lds_delta
  __shared__ int s_nmoved;
  const int seg = blockIdx.y;
  const Seg sg = segs[seg];
  const int kk = sg.kk, tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  const int nrows = (int)max((int64_t)0, min((int64_t)rows_per_block, sg.count - row0));
  if (nrows <= 0) return;
  if (tid == 0) s_nmoved = 0;
  if (lds_delta)
    for (int e = tid; e < kk * (D + 1); e += 256) s_delta[e] = 0;
  double bd[PPT], bd2[PPT];  // smallest and second smallest distance (the second only feeds the bounds of the later iterations)
  int bc[PPT];
#pragma unroll
  for (int m = 0; m < PPT; m++) { bd[m] = 0.0; bd2[m] = 1.0e300; bc[m] = -1; }
  const int4 zero4 = make_int4(0, 0, 0, 0);
#pragma unroll 1
  for (int c0 = 0; c0 < kk; c0 += KCH) {
    double s[PPT][KCH];
#pragma unroll
    for (int m = 0; m < PPT; m++)
#pragma unroll
      for (int c = 0; c < KCH; c++) s[m][c] = 0.0;
    int4 pre[2 * PPT];
    double pre_c = 0.0;
    auto fetch = [&](int j0) {  // global -> registers: 2 threads x 16 B per row, 128 rows per slot; one centroid value per thread < 128
#pragma unroll
      for (int m = 0; m < 2 * PPT; m++) {
        const int r = (tid >> 1) + 128 * m;
        // chunk-major copy [j0 / 8][point][8]: the workgroup's rows of one chunk are one contiguous block (row-major pts would
        // give 32 useful bytes per 128-byte line and re-fetch every line four times over the 24 chunks)
        pre[m] = r < nrows ? *reinterpret_cast<const int4 *>(pts_chunked + ((int64_t)(j0 / A_DCH) * n_total + sg.begin + row0 + r) * A_DCH + (tid & 1) * 4) : zero4;
      }
      if (tid < A_DCH * KCH) {
        const int j = tid / KCH, c = tid - j * KCH;
        pre_c = c0 + c < kk ? cent[((int64_t)seg * k + c0 + c) * D + j0 + j] : 0.0;
      }
    };
    auto stage = [&]() {  // registers -> LDS
#pragma unroll
      for (int m = 0; m < 2 * PPT; m++) {
        int32_t *dst = s_pts + ((tid >> 1) + 128 * m) * PITCH + (tid & 1) * 4;
        dst[0] = pre[m].x; dst[1] = pre[m].y; dst[2] = pre[m].z; dst[3] = pre[m].w;
      }
      if (tid < A_DCH * KCH) s_cent[tid / KCH][tid % KCH] = pre_c;
    };
    fetch(0);
    __syncthreads();  // previous pass (or the zeroing above) done with the buffers
    stage();
    __syncthreads();
#pragma unroll 1
    for (int j0 = 0; j0 < D; j0 += A_DCH) {
      if (j0 + A_DCH < D) fetch(j0 + A_DCH);
#pragma unroll 2
      for (int j = 0; j < A_DCH; j++) {
        double pj[PPT];
#pragma unroll
        for (int m = 0; m < PPT; m++) pj[m] = (double)s_pts[(tid + 256 * m) * PITCH + j];
#pragma unroll
        for (int c = 0; c < KCH; c += 2) {
          const double2 cv = *reinterpret_cast<const double2 *>(&s_cent[j][c]);
#pragma unroll
          for (int m = 0; m < PPT; m++) {
            const double t0 = __dsub_rn(pj[m], cv.x), t1 = __dsub_rn(pj[m], cv.y);
            s[m][c] = __fma_rn(t0, t0, s[m][c]);
            s[m][c + 1] = __fma_rn(t1, t1, s[m][c + 1]);
          }
        }
      }
      __syncthreads();
      if (j0 + A_DCH < D) stage();
      __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < PPT; m++)
#pragma unroll
      for (int c = 0; c < KCH; c++)
        if (c0 + c < kk) {
          if (bc[m] < 0 || s[m][c] < bd[m]) { if (bc[m] >= 0) bd2[m] = bd[m]; bd[m] = s[m][c]; bc[m] = c0 + c; }
          else if (s[m][c] < bd2[m]) bd2[m] = s[m][c];
        }
  }
  if (ub) {  // Euclidean bounds for the skipping iterations, rounded the safe way: ub >= the distance to the own centroid, lb <= to any other
#pragma unroll
    for (int m = 0; m < PPT; m++) {
      const int r = tid + 256 * m;
      if (r >= nrows) continue;
      const int64_t gi = sg.begin + row0 + r;
      ub[gi] = sqrt(bd[m]) * (1.0 + 1e-12);
      lb[gi] = sqrt(bd2[m]) * (1.0 - 1e-12);
    }
  }
  // moved points -> list
#pragma unroll
  for (int m = 0; m < PPT; m++) {
    const int r = tid + 256 * m;
    if (r >= nrows) continue;
    const int64_t gi = sg.begin + row0 + r;
    const int old = assign[gi];
    if (old == bc[m]) continue;
    assign[gi] = bc[m];
    const int slot = atomicAdd(&s_nmoved, 1);
    s_moved[slot * 3] = r; s_moved[slot * 3 + 1] = old; s_moved[slot * 3 + 2] = bc[m];
  }
  __syncthreads();
  const int nmoved = s_nmoved;
  if (nmoved == 0) return;
  if (tid == 0) atomicAdd(&segs[seg].changed, nmoved);
#pragma unroll 2
  for (int e = tid >> 6; e < nmoved; e += 4) {  // a wave per moved row (four rows in flight, eight with the unrolling): lane -> dimensions lane, +64, +128; 192 = the weight
    const int r = s_moved[e * 3], old = s_moved[e * 3 + 1], nw = s_moved[e * 3 + 2];
    const int64_t gi = sg.begin + row0 + r;
    const long long wi = w ? (long long)w[gi] : 1;
#pragma unroll
    for (int j = tid & 63; j <= D; j += 64) {
      const u64 v = j < D ? (u64)(wi * pts[gi * D + j]) : (u64)wi;
      if (lds_delta) {
        atomicAdd(&s_delta[nw * (D + 1) + j], v);
        if (old >= 0) atomicAdd(&s_delta[old * (D + 1) + j], (u64)0 - v);
      } else {
        u64 *base = j < D ? sums + (int64_t)seg * k * D : cnts + (int64_t)seg * k;
        const int64_t stride = j < D ? D : 1, off = j < D ? j : 0;
        atomicAdd(&base[nw * stride + off], v);
        if (old >= 0) atomicAdd(&base[old * stride + off], (u64)0 - v);
      }
    }
  }
  if (lds_delta) {
    __syncthreads();
    for (int e = tid; e < kk * (D + 1); e += 256) {
      const u64 v = s_delta[e];
      if (v == 0) continue;
      const int c = e / (D + 1), j = e - c * (D + 1);
      if (j == D) atomicAdd(&cnts[(int64_t)seg * k + c], v);
      else atomicAdd(&sums[((int64_t)seg * k + c) * D + j], v);
    }
  }
}

__global__ void k_chunk_major(const int32_t *__restrict__ pts, int64_t n, int32_t *__restrict__ out) {  // [n][192] -> [24][n][8]
  const int64_t total = n * 48;  // int4 elements
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / 48;
    const int v = (int)(e - i * 48), ch = v >> 1, half = v & 1;
    reinterpret_cast<int4 *>(out)[((int64_t)ch * n + i) * 2 + half] = reinterpret_cast<const int4 *>(pts)[e];
  }
}

// exact integer weighted sums: LDS partials per workgroup for up to KCH_ACC clusters x D, flushed with global atomics
template <int D>
__global__ __launch_bounds__(256) void k_accumulate(const int32_t *__restrict__ pts, const uint32_t *__restrict__ w,
                                                    const Seg *__restrict__ segs, int k, const int32_t *__restrict__ assign,
                                                    u64 *__restrict__ sums, u64 *__restrict__ cnts, const int *__restrict__ quiet) {
  if (*quiet >= 0) return;
  extern __shared__ u64 s_acc[];  // [kk][D+1] when it fits, else straight to global
  int bx, nbx;
  const int seg = find_seg(segs, bx, nbx);
  if (nbx <= 0) return;  // only when every segment is empty
  const Seg sg = segs[seg];
  const int kk = sg.kk;
  const bool use_lds = (size_t)kk * (D + 1) * 8 <= 64 * 1024;
  if (use_lds) {
    for (int e = threadIdx.x; e < kk * (D + 1); e += 256) s_acc[e] = 0;
    __syncthreads();
  }
  for (int64_t i = bx * 256 + threadIdx.x; i < sg.count; i += (int64_t)nbx * 256) {
    const int c = assign[sg.begin + i];
    const long long wi = w ? (long long)w[sg.begin + i] : 1;
    const int32_t *p = pts + (sg.begin + i) * D;
    if (use_lds) {
      atomicAdd(&s_acc[c * (D + 1) + D], (u64)wi);
      for (int j = 0; j < D; j++) atomicAdd(&s_acc[c * (D + 1) + j], (u64)(wi * p[j]));
    } else {
      atomicAdd(&cnts[(int64_t)seg * k + c], (u64)wi);
      for (int j = 0; j < D; j++) atomicAdd(&sums[((int64_t)seg * k + c) * D + j], (u64)(wi * p[j]));
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int e = threadIdx.x; e < kk * (D + 1); e += 256) {
      const u64 v = s_acc[e];
      if (v == 0) continue;
      const int c = e / (D + 1), j = e - c * (D + 1);
      if (j == D) atomicAdd(&cnts[(int64_t)seg * k + c], v);
      else atomicAdd(&sums[((int64_t)seg * k + c) * D + j], v);
    }
  }
}

// One block: centroid = sum / weight where weight > 0 (segments that changed), reset sums/counts/changed, and latch the
// first iteration in which nothing changed anywhere (so the host can poll rarely).
__global__ __launch_bounds__(1024) void k_update_all(Seg *__restrict__ segs, int nseg, int k, int d, int carry, u64 *__restrict__ sums,
                                                     u64 *__restrict__ cnts, double *__restrict__ cent, int it,
                                                     int *__restrict__ quiet_iter, int *host_quiet = nullptr /* page-locked host word that gets the flag too */) {
  if (*quiet_iter >= 0) return;
  __shared__ int s_any;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  // carry: the assignment kernels (k_assign192, fused k_assign<3>) keep the sums current with +/- deltas; the unfused D = 3 path rebuilds them
  const int64_t total = (int64_t)nseg * k * d;
  for (int64_t e = threadIdx.x; e < total; e += 1024) {
    const int64_t sc = e / d;
    const int seg = (int)(sc / k);
    const u64 cn = cnts[sc];
    if (segs[seg].changed && cn > 0) cent[e] = __ddiv_rn((double)(long long)sums[e], (double)(long long)cn);
    if (!carry) sums[e] = 0;
  }
  __syncthreads();
  for (int seg = threadIdx.x; seg < nseg; seg += 1024) {
    if (segs[seg].changed) s_any = 1;
    segs[seg].changed = 0;
  }
  if (!carry)
    for (int64_t sc = threadIdx.x; sc < (int64_t)nseg * k; sc += 1024) cnts[sc] = 0;
  __syncthreads();
  if (threadIdx.x == 0 && s_any == 0 && *quiet_iter < 0) {
    *quiet_iter = it;
    if (host_quiet) __hip_atomic_store(host_quiet, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <int PPT>
static void launch_assign192_t(dim3 grid, size_t lds, hipStream_t stream, const int32_t *pts, const int32_t *ptsc, int64_t ntot, const uint32_t *w, Seg *ds,
                               int k, const double *cent, int32_t *assign, u64 *sums, u64 *cnts, int rows, int lds_delta, const int *quiet, double *ub, double *lb) {
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_assign192<PPT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); attr_set = true; }
  hipLaunchKernelGGL(k_assign192<PPT>, grid, dim3(256), lds, stream, pts, ptsc, ntot, w, ds, k, cent, assign, sums, cnts, rows, lds_delta, quiet, ub, lb);
}
static void launch_assign192(int ppt, dim3 grid, size_t lds, hipStream_t stream, const int32_t *pts, const int32_t *ptsc, int64_t ntot, const uint32_t *w,
                             Seg *ds, int k, const double *cent, int32_t *assign, u64 *sums, u64 *cnts, int rows, int lds_delta, const int *quiet,
                             double *ub = nullptr, double *lb = nullptr) {
  switch (ppt) {
    case 1: launch_assign192_t<1>(grid, lds, stream, pts, ptsc, ntot, w, ds, k, cent, assign, sums, cnts, rows, lds_delta, quiet, ub, lb); break;
    case 2: launch_assign192_t<2>(grid, lds, stream, pts, ptsc, ntot, w, ds, k, cent, assign, sums, cnts, rows, lds_delta, quiet, ub, lb); break;
    case 3: launch_assign192_t<3>(grid, lds, stream, pts, ptsc, ntot, w, ds, k, cent, assign, sums, cnts, rows, lds_delta, quiet, ub, lb); break;
    case 4: launch_assign192_t<4>(grid, lds, stream, pts, ptsc, ntot, w, ds, k, cent, assign, sums, cnts, rows, lds_delta, quiet, ub, lb); break;
    default: launch_assign192_t<5>(grid, lds, stream, pts, ptsc, ntot, w, ds, k, cent, assign, sums, cnts, rows, lds_delta, quiet, ub, lb); break;
  }
}

// ---- D = 192: iterations that skip what cannot change (Hamerly's bounds, made exact) ------------------------------------------
// After a few full iterations most tiles sit firmly in their cluster and the centroids barely move, yet the assignment step costs
// the same 16 x 192 double-precision distance terms per tile every time.  Per point two Euclidean bounds are kept: ub >= its distance
// to its own centroid, lb <= its distance to every other one; a centroid update moves them by the centroids' displacements.  While
//      ub < max(lb, half the distance from the own centroid to the nearest other one)
// holds WITH a relative margin of 1e-9 on both sides, the own centroid is strictly the nearest by a margin six orders of magnitude above
// the rounding of the distance arithmetic (192 fused multiply-adds: relative error below 1e-13) and of the bound bookkeeping (every
// step rounds the safe way, with margins of 1e-12), so the assignment the full computation would make -- computed distances, ties to
// the lowest index -- is the one the point already has: it is skipped.  Otherwise the distance to the own centroid is computed
// (tightening ub), and if the test still fails the point is listed and goes through k_assign192 itself, which reads its rows through the list.
// The result is therefore bit for bit that of the plain iterations (and of the oracle); only the work differs.
constexpr int H_MAXK = 64;       // centroids kept in LDS by the skipping kernels
constexpr double H_ETA = 1e-9;   // margin of the skip test
constexpr int H_SLICE = 1024;    // points per workgroup of k_h_bounds
__global__ __launch_bounds__(256) void k_h_bounds(const int32_t *__restrict__ pts, int64_t n, const Seg *__restrict__ segs,
                                                  const double *__restrict__ cent /* [k][192] */, const int32_t *__restrict__ assign, double *__restrict__ ub, double *__restrict__ lb,
                                                  const double *__restrict__ cmove /* [k] displacement of each centroid, then the largest, the second largest, whose */,
                                                  const double *__restrict__ shalf /* [k] half the distance to the nearest other centroid */, int k,
                                                  int32_t *__restrict__ need, unsigned *__restrict__ need_cnt, const int *__restrict__ quiet) {
  __shared__ int s_list[H_SLICE], s_need[H_SLICE];
  __shared__ int s_nlist, s_nneed;
  __shared__ unsigned s_base;
  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * H_SLICE;
  // (the slice's assignments and bounds are asked for together with the flag and the displacements: one round trip, not two)
  constexpr int RB = H_SLICE / 256;
  int b_a[RB];
  double b_u[RB], b_l[RB];
#pragma unroll
  for (int r = 0; r < RB; r++) {
    const int64_t i = i0 + r * 256 + tid;
    const int64_t ii = i < n ? i : i0;  // (the slice's first point exists)
    b_a[r] = assign[ii]; b_u[r] = ub[ii]; b_l[r] = lb[ii];
  }
  const double dmax = cmove[k], dmax2 = cmove[k + 1];
  const int amax = (int)cmove[k + 2];
  if (*quiet >= 0) return;
  if (tid == 0) { s_nlist = 0; s_nneed = 0; }
  __syncthreads();
  // pass 1, every point of the slice: move the bounds with the centroids; the points whose loosened bounds no longer prove them -> LDS list.
  // The four points of a thread go through it side by side -- their loads first, then the table look-ups that depend on them, then the
  // arithmetic, one list append per wave: with a loop that could leave early and an LDS atomic per listed point the compiler kept the
  // points apart, and every point paid its two dependent round trips to memory on its own (this launch is ~20 % of an iteration)
  {
    constexpr int R = H_SLICE / 256;
    int a[R];
    double u[R], l[R], mv[R], sh[R];
    bool valid[R], listed[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      valid[r] = i0 + r * 256 + tid < n;
      a[r] = b_a[r]; u[r] = b_u[r]; l[r] = b_l[r];
    }
#pragma unroll
    for (int r = 0; r < R; r++) { mv[r] = cmove[a[r]]; sh[r] = shalf[a[r]]; }
    int cnt = 0;
    unsigned long long bal[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const int64_t i = i0 + r * 256 + tid;
      const double un = (u[r] + mv[r]) * (1.0 + 1e-15);
      double ln = l[r] - (a[r] == amax ? dmax2 : dmax);  // lb bounds the OTHER centroids: the own one's displacement does not loosen it
      ln -= fabs(ln) * 1e-15;
      if (valid[r]) { ub[i] = un; lb[i] = ln; }
      listed[r] = valid[r] && !(un * (1.0 + H_ETA) < fmax(sh[r], ln) * (1.0 - H_ETA));
      bal[r] = __builtin_amdgcn_ballot_w64(listed[r]);
      cnt += __popcll(bal[r]);
    }
    if (cnt) {  // (uniform in the wave)
      const int lane = tid & 63;
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_nlist, cnt);
      base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
      for (int r = 0; r < R; r++) {
        if (listed[r]) s_list[base + __popcll(bal[r] & ((1ull << lane) - 1ull))] = r * 256 + tid;
        base += __popcll(bal[r]);
      }
    }
  }
  __syncthreads();
  // pass 2, the listed ones: the distance to the own centroid tightens ub; 16 lanes per point (12 dimensions each, the row and the centroid's
  // row read as they lie in memory -- a slice without listed points, most of them late in a clustering, reads no centroid at all).  The
  // partial sums add in another order than the scoring's chain does: both stay within 2.2e-14 (relative) of the exact sum, far inside the
  // factor 1 + 1e-12 that makes the root an upper bound of the distance AS SCORED.  Still unproven -> the global list
  const int nlist = s_nlist;
  if (nlist == 0) return;
  const int l16 = tid & 15;
  for (int t0 = 0; t0 < nlist; t0 += 16) {
    const int t = t0 + (tid >> 4);
    const bool act = t < nlist;
    const int64_t i = i0 + s_list[act ? t : 0];
    const int a = assign[i];
    const int4 *p = reinterpret_cast<const int4 *>(pts + i * 192 + l16 * 12);
    const double2 *c = reinterpret_cast<const double2 *>(cent + (int64_t)a * 192 + l16 * 12);
    const int4 v0 = p[0], v1 = p[1], v2 = p[2];
    const double2 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
    const int pv[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
    const double cv[12] = {c0.x, c0.y, c1.x, c1.y, c2.x, c2.y, c3.x, c3.y, c4.x, c4.y, c5.x, c5.y};
    double sd = 0.0;
#pragma unroll
    for (int j = 0; j < 12; j++) { const double d0 = __dsub_rn((double)pv[j], cv[j]); sd = __fma_rn(d0, d0, sd); }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sd += __shfl_xor(sd, o);
    if (act && l16 == 0) {
      const double u = sqrt(sd) * (1.0 + 1e-12);
      ub[i] = u;
      if (!(u * (1.0 + H_ETA) < fmax(shalf[a], lb[i]) * (1.0 - H_ETA))) s_need[atomicAdd(&s_nneed, 1)] = s_list[t];
    }
  }
  // the workgroup's share of the global list with ONE atomic on its counter (a counter every listed point of the launch adds to
  // serialises them: ~6 ns each, and the early iterations list tens of thousands)
  __syncthreads();
  const int nneed = s_nneed;
  if (nneed == 0) return;
  if (tid == 0) s_base = atomicAdd(need_cnt, (unsigned)nneed);
  __syncthreads();
  for (int t = tid; t < nneed; t += 256) need[s_base + t] = (int32_t)(i0 + s_need[t]);
}

// The listed points through the full computation: k_assign192's arithmetic (sum over dimensions in order of (p - c)^2, one IEEE subtraction
// and one fused multiply-add each, ties -> lowest centroid) and its carried sums, shaped for FEW points: one point per lane read straight
// from the chunk-major copy (32 bytes per chunk, the next chunk in flight), the centroids broadcast from LDS (staged from the transposed
// copy k_h_update leaves; reading them as scalar operands through the scalar cache instead measured 20 % slower: 24 KB of centroids do
// not stay in it).  (A thread-per-point form of it was the first list kernel; the four-lane form below replaced it.)
// The same, a point spread over 4 lanes (each lane scores 4 of the 16 centroids of a pass): the thread-per-point shape leaves a lone wave per
// SIMD with 6 144 dependent-ish double-precision operations and its workgroup's four waves queueing for 24 KB of LDS reads per point; here
// the chain is a quarter as long and the list covers four times as many compute units.  Every accumulator still sums its 192 terms in
// order, so the distances are the same doubles; the lanes' (best, second best) merge by (distance, centroid index), which is what the
// in-order scan with its strict `<` computes.
template <int N>
__device__ __forceinline__ void pin_accumulators(double (&s)[N]) {  // an empty statement the optimiser must have the values ready for
#pragma unroll
  for (int c = 0; c < N; c++) asm volatile("" : "+v"(s[c]));
}

template <int Q>
__device__ __forceinline__ int quad_bcast(int v) {  // lane Q of every group of four lanes, to its whole group
  return __builtin_amdgcn_update_dpp(0, v, Q | (Q << 2) | (Q << 4) | (Q << 6), 0xf, 0xf, true);
}

__device__ __forceinline__ void assign192_list4_body(const int32_t *__restrict__ pts, const int32_t *__restrict__ pts_chunked, int64_t n_total,
                                                         const uint32_t *__restrict__ w, Seg *__restrict__ segs, int k, const double *__restrict__ cent_t /* [192][kt] */,
                                                         int kt, int32_t *__restrict__ assign, u64 *__restrict__ sums, u64 *__restrict__ cnts,
                                                         double *__restrict__ ub, double *__restrict__ lb, const int32_t *__restrict__ need,
                                                         const unsigned cnt /* the list's length; no list: every point */) {
  constexpr int D = 192, NP = 64, CPL = KCH / 4;  // points per pass of a workgroup, centroids per lane and pass
  if (blockIdx.x * (unsigned)NP >= cnt) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  __shared__ int s_nmoved;
  const int kk = segs[0].kk, tid = threadIdx.x, slot = tid >> 2, sub = tid & 3;
  double *s_c = reinterpret_cast<double *>(s_raw);                    // [D][KCH]
  u64 *s_delta = reinterpret_cast<u64 *>(s_raw + D * KCH * 8);       // [kk][D + 1]
  int32_t *s_moved = reinterpret_cast<int32_t *>(s_delta + kk * (D + 1));  // [NP][3]: slot, old, new
  // a fixed grid walks the list (a workgroup per 64 listed points was 5 000 workgroups launched to find that 4 950 have nothing to do,
  // each staging the centroids first); with at most KCH centroids they are staged once per workgroup
  const bool single = kk <= KCH;
  // The four lanes of a point each fetch a quarter of its row (48 dimensions, 12 loads of 16 bytes, all of them in flight together: ONE
  // round trip to memory per point -- the chunk-major copy the plain iterations stream cost a listed point twelve round trips, two chunks
  // at a time, and a short list is all latency) and hand the values round inside their group of four with DPP broadcasts, in the order
  // of the dimensions.  The first pass's rows are asked for before anything else.
  auto fetch = [&](unsigned row0, int4 (&x)[12], int64_t &gi, bool &active) {
    active = row0 + slot < cnt;
    gi = need ? need[active ? row0 + slot : row0] : (int64_t)(active ? row0 + slot : row0);
    const int4 *src = reinterpret_cast<const int4 *>(pts + gi * D + sub * 48);
#pragma unroll
    for (int u = 0; u < 12; u++) x[u] = src[u];
  };
  int4 x[12];
  int64_t gi;
  bool active;
  fetch(blockIdx.x * (unsigned)NP, x, gi, active);
  for (int e = tid; e < kk * (D + 1); e += 256) s_delta[e] = 0;
  if (tid == 0) s_nmoved = 0;
  if (single)
    for (int e = tid; e < D * KCH; e += 256) s_c[e] = cent_t[(int64_t)(e / KCH) * kt + (e % KCH)];
  __syncthreads();
  int total_moved = 0;
#pragma unroll 1
  for (unsigned row0 = blockIdx.x * (unsigned)NP; row0 < cnt; row0 += gridDim.x * (unsigned)NP) {
    double bd = 1.0e300, bd2 = 1.0e300;
    int bc = 0x7fffffff;
#pragma unroll 1
    for (int c0 = 0; c0 < kk; c0 += KCH) {
      if (!single) {
        __syncthreads();
        for (int e = tid; e < D * KCH; e += 256) s_c[e] = cent_t[(int64_t)(e / KCH) * kt + c0 + (e % KCH)];
        __syncthreads();
      }
      double s[CPL];
#pragma unroll
      for (int c = 0; c < CPL; c++) s[c] = 0.0;
      auto term = [&](int v, int j) {  // dimension j of the point against this lane's CPL centroids
        const double pj = (double)v;
        const double *cj = s_c + j * KCH + sub * CPL;
#pragma unroll
        for (int c = 0; c < CPL; c += 2) {
          const double2 cv = *reinterpret_cast<const double2 *>(cj + c);
          const double t0 = __dsub_rn(pj, cv.x), t1 = __dsub_rn(pj, cv.y);
          s[c] = __fma_rn(t0, t0, s[c]);
          s[c + 1] = __fma_rn(t1, t1, s[c + 1]);
        }
      };
      auto quarter = [&](auto qtag) {  // the 48 dimensions lane Q of the group holds
        constexpr int Q = decltype(qtag)::value;
#pragma unroll
        for (int u = 0; u < 12; u++) {
          term(quad_bcast<Q>(x[u].x), Q * 48 + u * 4);
          term(quad_bcast<Q>(x[u].y), Q * 48 + u * 4 + 1);
          term(quad_bcast<Q>(x[u].z), Q * 48 + u * 4 + 2);
          term(quad_bcast<Q>(x[u].w), Q * 48 + u * 4 + 3);
          // the accumulators pinned here: without it the optimiser sinks the whole unrolled chain of multiply-adds below its 384 centroid
          // reads, which then all have to stay live (1 500 spilled registers, the kernel eight times slower)
          pin_accumulators(s);
        }
      };
      quarter(std::integral_constant<int, 0>{});
      quarter(std::integral_constant<int, 1>{});
      quarter(std::integral_constant<int, 2>{});
      quarter(std::integral_constant<int, 3>{});
#pragma unroll
      for (int c = 0; c < CPL; c++) {
        const int ci = c0 + sub * CPL + c;
        if (ci < kk) {  // this lane's centroids come in ascending order: strict `<` keeps the lowest index among equals
          if (s[c] < bd) { bd2 = bd; bd = s[c]; bc = ci; }
          else if (s[c] < bd2) bd2 = s[c];
        }
      }
    }
    const int64_t gi_cur = gi;
    const bool active_cur = active;
    {  // the next pass's rows, while this one's results are merged and written
      const unsigned nrow0 = row0 + gridDim.x * (unsigned)NP;
      if (nrow0 < cnt) fetch(nrow0, x, gi, active);
    }
    // the four lanes of a point: the best by (distance, index); the second best distance = the smallest of the rest
#pragma unroll
    for (int o = 1; o < 4; o <<= 1) {
      const double od = __shfl_xor(bd, o), od2 = __shfl_xor(bd2, o);
      const int oc = __shfl_xor(bc, o);
      const bool take = od < bd || (od == bd && oc < bc);
      const double loser = take ? bd : od;
      bd2 = fmin(fmin(bd2, od2), loser);
      if (take) { bd = od; bc = oc; }
    }
    if (active_cur && sub == 0) {
      ub[gi_cur] = sqrt(bd) * (1.0 + 1e-12);
      lb[gi_cur] = sqrt(bd2) * (1.0 - 1e-12);
      const int old = assign[gi_cur];
      if (old != bc) {
        assign[gi_cur] = bc;
        const int m = atomicAdd(&s_nmoved, 1);
        s_moved[m * 3] = slot; s_moved[m * 3 + 1] = old; s_moved[m * 3 + 2] = bc;
      }
    }
    __syncthreads();
    const int nmoved = s_nmoved;
    total_moved += nmoved;
#pragma unroll 2
    for (int e = tid >> 6; e < nmoved; e += 4) {  // a wave per moved row between the carried sums (coalesced read, three dimensions per lane)
      const int old = s_moved[e * 3 + 1], nw = s_moved[e * 3 + 2];
      const int64_t mi = need ? need[row0 + s_moved[e * 3]] : (int64_t)(row0 + s_moved[e * 3]);
      const long long wi = w ? (long long)w[mi] : 1;
#pragma unroll
      for (int j = tid & 63; j <= D; j += 64) {
        const u64 v = j < D ? (u64)(wi * pts[mi * D + j]) : (u64)wi;
        atomicAdd(&s_delta[nw * (D + 1) + j], v);
        if (old >= 0) atomicAdd(&s_delta[old * (D + 1) + j], (u64)0 - v);
      }
    }
    __syncthreads();  // the moved list has been read
    if (tid == 0) s_nmoved = 0;
    __syncthreads();
  }
  if (total_moved == 0) return;
  if (tid == 0) atomicAdd(&segs[0].changed, total_moved);
  for (int e = tid; e < kk * (D + 1); e += 256) {
    const u64 v = s_delta[e];
    if (v == 0) continue;
    const int c = e / (D + 1), j = e - c * (D + 1);
    if (j == D) atomicAdd(&cnts[c], v);
    else atomicAdd(&sums[(int64_t)c * D + j], v);
  }
}

// The same for at most KCH centroids, a lane per (point, centroid) pair: 16 points per pass of a workgroup, their rows staged through
// LDS (the 16 lanes of a point read one address), one chain of 192 terms per lane instead of four of them.  Late in a clustering the list
// holds a few thousand points: with 64 points per workgroup that was 16-80 busy workgroups each working through four-chain passes; here
// it is four times as many workgroups with passes a third as long.  Every accumulator still sums its 192 terms in order; the 16 lanes'
// (best, second best) merge by (distance, centroid index), which is what the in-order scan with its strict `<` computes.
constexpr int L16_P = 16;
__device__ __forceinline__ void assign192_list16_body(const int32_t *__restrict__ pts, int64_t n_total, const uint32_t *__restrict__ w, Seg *__restrict__ segs,
                                                          const double *__restrict__ cent_t /* [192][kt] */, int kt, int32_t *__restrict__ assign, u64 *__restrict__ sums,
                                                          u64 *__restrict__ cnts, double *__restrict__ ub, double *__restrict__ lb,
                                                          const int32_t *__restrict__ need, const unsigned cnt) {
  constexpr int D = 192;
  if (blockIdx.x * (unsigned)L16_P >= cnt) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  double *const s_c = reinterpret_cast<double *>(s_raw);                 // [D][KCH] (zero beyond kk: cent_t is)
  u64 *const s_delta = reinterpret_cast<u64 *>(s_raw + D * KCH * 8);    // [kk][D + 1]
  __shared__ __attribute__((aligned(16))) int s_rows[L16_P * D];
  __shared__ int s_moved[L16_P * 3], s_gi[L16_P], s_nmoved;
  const int kk = segs[0].kk, tid = threadIdx.x, pslot = tid >> 4, cl = tid & 15, wave = tid >> 6, lane = tid & 63;
  auto stage = [&](unsigned row0, int4 (&x)[3]) {  // the pass's rows: 16 x 768 bytes, three 16-byte pieces per thread
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int piece = u * 256 + tid, pr = piece / 48, off = piece - pr * 48;
      const int64_t gi = need[min(row0 + (unsigned)pr, cnt - 1)];
      x[u] = reinterpret_cast<const int4 *>(pts + gi * D)[off];
    }
  };
  int4 x[3];
  stage(blockIdx.x * (unsigned)L16_P, x);
  for (int e = tid; e < kk * (D + 1); e += 256) s_delta[e] = 0;
  if (tid == 0) s_nmoved = 0;
  for (int e = tid; e < D * KCH; e += 256) s_c[e] = cent_t[(int64_t)(e / KCH) * kt + (e % KCH)];
  int total_moved = 0;
#pragma unroll 1
  for (unsigned row0 = blockIdx.x * (unsigned)L16_P; row0 < cnt; row0 += gridDim.x * (unsigned)L16_P) {
    __syncthreads();  // the rows of the pass before are no longer read (first pass: s_c, s_delta are whole)
#pragma unroll
    for (int u = 0; u < 3; u++) reinterpret_cast<int4 *>(s_rows)[u * 256 + tid] = x[u];
    if (tid < L16_P) s_gi[tid] = need[min(row0 + (unsigned)tid, cnt - 1)];
    __syncthreads();
    {
      const unsigned nrow0 = row0 + gridDim.x * (unsigned)L16_P;
      if (nrow0 < cnt) stage(nrow0, x);  // the next pass's rows, while this one is scored
    }
    const bool active = row0 + pslot < cnt;
    const int64_t gi = s_gi[pslot];
    double sacc = 0.0;
    {
      const int *rp = s_rows + pslot * D;
      const double *cp = s_c + cl;
#pragma unroll 16
      for (int j = 0; j < D; j++) {
        const double t0 = __dsub_rn((double)rp[j], cp[j * KCH]);
        sacc = __fma_rn(t0, t0, sacc);
      }
    }
    double bd = cl < kk ? sacc : 1.0e300, bd2 = 1.0e300;
    int bc = cl < kk ? cl : 0x7fffffff;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {  // the 16 lanes of a point: the best by (distance, index); the second best distance = the smallest of the rest
      const double od = __shfl_xor(bd, o), od2 = __shfl_xor(bd2, o);
      const int oc = __shfl_xor(bc, o);
      const bool take = od < bd || (od == bd && oc < bc);
      const double loser = take ? bd : od;
      bd2 = fmin(fmin(bd2, od2), loser);
      if (take) { bd = od; bc = oc; }
    }
    if (active && cl == 0) {
      ub[gi] = sqrt(bd) * (1.0 + 1e-12);
      lb[gi] = sqrt(bd2) * (1.0 - 1e-12);
      const int old = assign[gi];
      if (old != bc) {
        assign[gi] = bc;
        const int m = atomicAdd(&s_nmoved, 1);
        s_moved[m * 3] = pslot; s_moved[m * 3 + 1] = old; s_moved[m * 3 + 2] = bc;
      }
    }
    __syncthreads();
    const int nmoved = s_nmoved;
    total_moved += nmoved;
    for (int e = wave; e < nmoved; e += 4) {  // a wave per moved row between the carried sums (its row is still in LDS)
      const int old = s_moved[e * 3 + 1], nw = s_moved[e * 3 + 2], ps = s_moved[e * 3];
      const int64_t mi = s_gi[ps];
      const long long wi = w ? (long long)w[mi] : 1;
#pragma unroll
      for (int j = lane; j <= D; j += 64) {
        const u64 v = j < D ? (u64)(wi * s_rows[ps * D + j]) : (u64)wi;
        atomicAdd(&s_delta[nw * (D + 1) + j], v);
        if (old >= 0) atomicAdd(&s_delta[old * (D + 1) + j], (u64)0 - v);
      }
    }
    __syncthreads();  // the moved list has been read
    if (tid == 0) s_nmoved = 0;
  }
  if (total_moved == 0) return;
  if (tid == 0) atomicAdd(&segs[0].changed, total_moved);
  for (int e = tid; e < kk * (D + 1); e += 256) {
    const u64 v = s_delta[e];
    if (v == 0) continue;
    const int c = e / (D + 1), j = e - c * (D + 1);
    if (j == D) atomicAdd(&cnts[c], v);
    else atomicAdd(&sums[(int64_t)c * D + j], v);
  }
}

// The list kernel: the four-lanes-per-point passes for long lists (the early iterations: tens of thousands of unproven points, where 64
// points per pass keep the chip's double-precision pipes full), the lane-per-pair passes for short ones (measured on the two bench
// clips: 31.9 against 21.7 microseconds per launch over the frozen clip's 87 iterations, 17.3 against 19.7 over the literal clip's 295).
#ifndef TM_LIST16_BELOW
#define TM_LIST16_BELOW 8192
#endif
constexpr unsigned LIST16_BELOW = TM_LIST16_BELOW;
__global__ __launch_bounds__(256) void k_assign192_list4(const int32_t *__restrict__ pts, const int32_t *__restrict__ pts_chunked, int64_t n_total,
                                                         const uint32_t *__restrict__ w, Seg *__restrict__ segs, int k, const double *__restrict__ cent_t /* [192][kt] */,
                                                         int kt, int32_t *__restrict__ assign, u64 *__restrict__ sums, u64 *__restrict__ cnts, const int *__restrict__ quiet,
                                                         double *__restrict__ ub, double *__restrict__ lb, const int32_t *__restrict__ need,
                                                         const unsigned *__restrict__ need_cnt) {
  const int q0 = *quiet;                                   // (the two control words in one round trip)
  const unsigned cnt0 = need ? *need_cnt : (unsigned)n_total;
  if (q0 >= 0) return;
  if (need && k <= KCH && cnt0 < LIST16_BELOW) assign192_list16_body(pts, n_total, w, segs, cent_t, kt, assign, sums, cnts, ub, lb, need, cnt0);
  else assign192_list4_body(pts, pts_chunked, n_total, w, segs, k, cent_t, kt, assign, sums, cnts, ub, lb, need, cnt0);
}

// the seeds' centroids into the transposed copy the list kernels read (k_h_update keeps it current afterwards)
__global__ void k_cent_transpose(const Seg *__restrict__ segs, const double *__restrict__ cent, double *__restrict__ cent_t, int kt) {
  const int kk = segs[0].kk;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < kk * 192; e += gridDim.x * blockDim.x) { const int c = e / 192, j = e - c * 192; cent_t[(int64_t)j * kt + c] = cent[e]; }
}

// k_update_all for one segment, plus what the bounds need: how far every centroid moved (rounded up) and half its distance to the
// nearest other centroid (rounded down)
__global__ __launch_bounds__(1024) void k_h_update(Seg *__restrict__ segs, int k, u64 *__restrict__ sums, u64 *__restrict__ cnts, double *__restrict__ cent,
                                                   double *__restrict__ cent_t /* [192][kt], zero beyond kk */, int kt, double *__restrict__ cmove,
                                                   double *__restrict__ shalf, unsigned *__restrict__ need_cnt, int it, int *__restrict__ quiet_iter,
                                                   int *host_quiet = nullptr /* page-locked host word that gets the flag too */) {
  extern __shared__ double s_new[];  // [kk][193] (odd pitch: the pair loop reads two rows at once)
  __shared__ unsigned long long s_min[H_MAXK];
  __shared__ double s_move[H_MAXK];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // (the first centroid of every wave is asked for together with the two words that say how many there are and whether anything moved:
  // one round trip to memory instead of two at the head of a launch that is all latency)
  u64 cn0 = 0, sm0[3] = {0, 0, 0};
  double old0[3] = {0.0, 0.0, 0.0};
  if (wave < k) {
    cn0 = cnts[wave];
#pragma unroll
    for (int u = 0; u < 3; u++) { old0[u] = cent[wave * 192 + lane + 64 * u]; sm0[u] = sums[wave * 192 + lane + 64 * u]; }
  }
  const int kk = segs[0].kk;
  const bool changed = segs[0].changed != 0;
  if (*quiet_iter >= 0) return;
  auto wave_sum = [&](double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; };
  if (tid < H_MAXK) s_min[tid] = 0x7ff0000000000000ull;  // +inf
  if (tid == 0) *need_cnt = 0;
  // a wave per centroid, 3 dimensions per lane: new position, displacement (the sums only feed the bounds, margins of 1e-9: their order is free)
  for (int c = wave; c < kk; c += 16) {
    const u64 cn = c == wave ? cn0 : cnts[c];
    double sd = 0.0;
#pragma unroll
    for (int u = 0; u < 3; u++) {
      const int j = lane + 64 * u;
      const double old = c == wave ? old0[u] : cent[c * 192 + j];
      double nw = old;
      if (changed && cn > 0) { nw = __ddiv_rn((double)(long long)(c == wave ? sm0[u] : sums[c * 192 + j]), (double)(long long)cn); cent[c * 192 + j] = nw; }
      s_new[c * 193 + j] = nw;
      cent_t[(int64_t)j * kt + c] = nw;
      const double t = nw - old;
      sd += t * t;
    }
    sd = wave_sum(sd);
    if (lane == 0) { const double mv = sqrt(sd) * (1.0 + 1e-9); cmove[c] = mv; s_move[c] = mv; }
  }
  __syncthreads();
  for (int pr = tid >> 4; pr < kk * kk; pr += 64) {  // pairwise distances, 16 lanes per pair: the smallest per centroid (non-negative doubles order like their bit patterns)
    const int a = pr / kk, b = pr - a * kk;
    if (a >= b) continue;  // (uniform in a group of 16 lanes, and the exchanges below stay inside one)
    double sd = 0.0;
#pragma unroll
    for (int u = 0; u < 12; u++) { const int j = (tid & 15) + 16 * u; const double t = s_new[a * 193 + j] - s_new[b * 193 + j]; sd += t * t; }
    for (int o = 8; o > 0; o >>= 1) sd += __shfl_xor(sd, o);
    if ((tid & 15) == 0) {
      atomicMin(&s_min[a], (unsigned long long)__double_as_longlong(sd));
      atomicMin(&s_min[b], (unsigned long long)__double_as_longlong(sd));
    }
  }
  __syncthreads();
  if (tid < kk) shalf[tid] = kk > 1 ? 0.5 * sqrt(__longlong_as_double((long long)s_min[tid])) * (1.0 - 1e-9) : 1.0e300;
  if (wave == 0) {  // the largest displacement, the largest among the others, and whose the largest is (the first of several): over the lanes of a wave
    static_assert(H_MAXK <= 64, "one lane per centroid");
    const double v = lane < kk ? s_move[lane] : 0.0;
    double mx = v;
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    const int amx = __builtin_ctzll(__builtin_amdgcn_ballot_w64(v == mx && (lane < kk || mx == 0.0)));
    double mx2 = lane == amx ? 0.0 : v;
    for (int o = 32; o > 0; o >>= 1) mx2 = fmax(mx2, __shfl_xor(mx2, o));
    if (lane == 0) {
      cmove[k] = mx; cmove[k + 1] = mx2; cmove[k + 2] = (double)amx;
      if (!changed && *quiet_iter < 0) {
        *quiet_iter = it;
        if (host_quiet) __hip_atomic_store(host_quiet, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      segs[0].changed = 0;
    }
  }
}

// ---- D = 192, at most KCH centroids: ALL skipping iterations in ONE resident launch (round 5) -----------------------------------------
// The three launches of a skipping iteration (k_h_bounds, k_assign192_list4, k_h_update) are each a chain of dependent round trips to
// memory -- 11 + 14.5 + 8.3 microseconds for a few thousand unproven points out of 320 705, 295 times on the literal bench clip.  Here one
// workgroup of 1024 threads per CU stays resident for the whole clustering:
//   * a workgroup OWNS up to 2 x 1024 points (interleaved over the grid, so that unproven points spread evenly); their assignment and
//     their two bounds live in LDS (bounds as Singles rounded the safe way: a bound only has to be a bound), so the pass over all points
//     that moves the bounds with the centroids touches no memory at all;
//   * every workgroup keeps its own copy of the centroids and of the carried integer sums, and applies every iteration's update itself
//     (16 x 192 quotients, displacements, pairwise half-distances: identical arithmetic in every workgroup, so no exchange);
//   * what crosses workgroups per iteration is ONE thing: the integer deltas of the sums caused by the points that moved (u64 atomic adds
//     into one of three rotating buffers, exact and order-free) plus their count, behind ONE barrier of the grid.  Every cross-workgroup
//     datum is an agent-scope atomic on both sides (adds, relaxed 8-byte loads, relaxed stores to clear), every storing wave drains its
//     vmcnt before its workgroup arrives, every load of the data comes after a workgroup barrier behind the poll: the hand-off form of
//     MI355X_MICROARCH.md "Valid forms" that needs no L2 write-back and no L1 invalidate -- the two fences were 23 of the 103 microseconds
//     of round 2's resident attempt, its thread-per-point passes most of the rest;
//   * unproven points: the distance to the own centroid first (16 lanes per point, k_h_bounds' arithmetic), then the full scoring with a
//     lane per (point, centroid) pair, 64 points per pass, rows in LDS -- k_assign192_list4's lane-per-pair arithmetic: every accumulator
//     sums its 192 terms in order, ties to the lowest centroid.
// The skip rule is sound (every bound rounded the safe way, the margins of k_h_bounds), so the assignments, hence the sums, centroids and
// iteration count, are bit for bit those of the plain iterations, of the three-launch path (TM_KM_LAUNCHES=1) and of the oracle.
#ifndef TM_KMR_STAMPS
#define TM_KMR_STAMPS 0
#endif
constexpr int HR_NT = 1024, HR_P = 64, HR_PITCH = KCH + 1, HR_MAXR = 2, HR_E = KCH * 193;
constexpr int HR_NSTAMP = 12;
struct HrState {                    // zeroed before the launch
  u64 delta[3][HR_E];              // the sums' deltas of one iteration ([c][193], the count last); three in rotation
  unsigned changed[3];             // points that moved in that iteration
  unsigned timeout;                // a barrier gave up (a workgroup was not resident)
  // the barrier: arrivals are counted in eight shards (workgroup g on shard g % 8: atomics on one word take their turns, ~12 ns each); the
  // last arrival of a shard adds one to each of the eight replicas of `top` (one instruction, eight lanes); a waiting workgroup polls its
  // shard's replica until all shards are in (loads on one word queue up like atomics do: 32 pollers a line).  A round trip to the memory
  // side is about a microsecond here, so the count of dependent ones is the barrier's price: arrival, replica add, poll.
  // Every word on a 128-byte line of its own.
  struct alignas(128) Line { unsigned v; unsigned pad[31]; };
  Line bar[8], top[8];
  u64 stamps[HR_NSTAMP + 4];       // diagnostic build: s_memtime spans of workgroup 0's phases; listed / rechecked-and-failed points of all workgroups
#if TM_KMR_STAMPS
  unsigned log[300][10];           // per iteration: workgroup 0's spans of phases 1-7, its listed and scored points, the points scored by all
#endif
};
#if TM_KMR_STAMPS
#define HR_STAMP(i) do { if (g == 0 && tid == 0) { const u64 t_ = __builtin_amdgcn_s_memtime(); s_stamp[i] += t_ - st_last; st_last = t_; } } while (0)
#else
#define HR_STAMP(i) do { } while (0)
#endif

// exchanges inside a row of 16 lanes without the LDS crossbar a __shfl_xor goes through (a data-parallel-primitive move is one vector
// instruction): lane ^ 1, lane ^ 2 (quad permutations), then the mirror image inside 8 and inside 16 lanes -- after the four steps every lane
// of a row has combined all sixteen
template <int CTRL>
__device__ __forceinline__ int hr_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ double hr_dpp(double v) { return __hiloint2double(hr_dpp<CTRL>(__double2hiint(v)), hr_dpp<CTRL>(__double2loint(v))); }
constexpr int HR_X1 = 0xB1, HR_X2 = 0x4E, HR_M8 = 0x141, HR_M16 = 0x140;  // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ double hr_sum16(double v) {
  v += hr_dpp<HR_X1>(v); v += hr_dpp<HR_X2>(v); v += hr_dpp<HR_M8>(v); v += hr_dpp<HR_M16>(v);
  return v;
}

__device__ __forceinline__ float hr_up(double x) { return (float)(x * (1.0 + 1.2e-7)); }                       // a Single >= x (x >= 0, far below FLT_MAX)
__device__ __forceinline__ float hr_down(double x) { x = fmin(x, 1.0e37); return (float)(x - fabs(x) * 1.2e-7); }  // a Single <= x

// every thread of the workgroup calls it; false: the spin gave up, the caller leaves.  No fence: see the header comment.
__device__ __forceinline__ bool hr_barrier(HrState *st, unsigned &epoch, unsigned nblk, int *s_ok) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's atomics and stores have been performed
  __syncthreads();
  if (threadIdx.x < 64) {
    epoch++;
    int ok = 1;
    if (nblk > 1) {
      const unsigned sh = blockIdx.x & 7u, nsh = nblk < 8u ? nblk : 8u;
      const unsigned mine = (nblk - sh + 7u) >> 3;  // workgroups on this shard
      bool last = false;
      if (threadIdx.x == 0) last = __hip_atomic_fetch_add(&st->bar[sh].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == epoch * mine;
      last = __builtin_amdgcn_readfirstlane((int)last) != 0;
      if (last && threadIdx.x < 8) __hip_atomic_fetch_add(&st->top[threadIdx.x].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // one instruction, eight lines
      if (threadIdx.x == 0)
      for (unsigned spins = 1; __hip_atomic_load(&st->top[sh].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * nsh; spins++) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255u) == 0 && (spins > (1u << 21) || __hip_atomic_load(&st->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {  // (~2 s: an iteration is microseconds)
          __hip_atomic_store(&st->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    if (threadIdx.x == 0) *s_ok = ok;
  }
  __syncthreads();
  return *s_ok != 0;
}

template <int ROUNDS /* 1024-point rounds a workgroup owns: a constant, so that the LDS arrays' places are */>
__global__ __launch_bounds__(HR_NT) void k_h_resident(const int32_t *__restrict__ pts, const uint32_t *__restrict__ w, int64_t n, Seg *__restrict__ segs, int k,
                                                      double *__restrict__ cent /* [k][192], in and out */, const u64 *__restrict__ sums, const u64 *__restrict__ cnts,
                                                      const double *__restrict__ cmove, const double *__restrict__ shalf, int32_t *__restrict__ assign,
                                                      const double *__restrict__ ub, const double *__restrict__ lb, HrState *__restrict__ st, int it0, int max_iter,
                                                      int *__restrict__ quiet_iter) {
  constexpr int D = 192, rounds = ROUNDS;
  static_assert(ROUNDS >= 1 && ROUNDS <= HR_MAXR, "rounds");
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  double *const s_c = reinterpret_cast<double *>(s_raw);                                      // [D][HR_PITCH]: centroid c, dimension j at j * HR_PITCH + c
  u64 *const s_sum = reinterpret_cast<u64 *>(s_raw + D * HR_PITCH * 8);                      // [KCH][193] carried sums, the count last
  u64 *const s_delta = s_sum + HR_E;                                                          // [KCH][193] this workgroup's deltas of the iteration
  int *const s_rows = reinterpret_cast<int *>(s_delta + HR_E);                                // [HR_P][D] rows of the points being scored
  float *const s_ub = reinterpret_cast<float *>(s_rows + HR_P * D);                           // [rounds * 1024]
  float *const s_lb = s_ub + rounds * HR_NT;
  uint16_t *const s_list = reinterpret_cast<uint16_t *>(s_lb + rounds * HR_NT);              // slots whose loosened bounds prove nothing
  uint16_t *const s_need = s_list + rounds * HR_NT;                                           // slots to score
  unsigned *const s_w = reinterpret_cast<unsigned *>(s_need + rounds * HR_NT);                // the points' weights (a moved point's comes from here, not from memory behind the chain)
  uint8_t *const s_a = reinterpret_cast<uint8_t *>(s_w + rounds * HR_NT);                     // assignment (0xff: no point in the slot)
  __shared__ double s_move[KCH + 2], s_half[KCH];
  __shared__ unsigned long long s_min[KCH];
  __shared__ int s_amax, s_nlist, s_nneed, s_nmoved, s_ok;
  __shared__ int s_moved[HR_P * 3];
  __shared__ unsigned s_wt[HR_P];
  __shared__ uint16_t s_pair[KCH * (KCH - 1) / 2];  // the centroid pairs a < b, a | b << 8
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = tid >> 4, l16 = tid & 15;
  const int g = blockIdx.x;
  const unsigned G = gridDim.x;
  if (*quiet_iter >= 0) return;  // converged in the plain iterations (every workgroup reads the same word)
  const int kk = segs[0].kk;
  auto gidx = [&](int slot) { return (int64_t)slot * G + g; };  // point i belongs to workgroup i % G: every workgroup the same share of every stretch of the tiles
#if TM_KMR_STAMPS
  __shared__ u64 s_stamp[HR_NSTAMP];  // (accumulated in LDS, written out at the end: a global read-modify-write per stamp costs more than most phases)
  __shared__ u64 s_prev[8];
  if (tid < HR_NSTAMP) s_stamp[tid] = 0;
  if (tid < 8) s_prev[tid] = 0;
  __syncthreads();
  u64 st_last = __builtin_amdgcn_s_memtime();
#endif
  // ---- state in: the centroids, the carried sums, the bounds the last plain iteration left, what the last update says about the centroids
  for (int e = tid; e < D * HR_PITCH; e += HR_NT) { const int j = e / HR_PITCH, c = e - j * HR_PITCH; s_c[e] = c < kk ? cent[c * D + j] : 0.0; }
  for (int e = tid; e < HR_E; e += HR_NT) {
    const int c = e / 193, j = e - c * 193;
    s_sum[e] = c < kk ? (j < D ? sums[c * D + j] : cnts[c]) : 0;
    s_delta[e] = 0;
  }
  for (int r = 0; r < rounds; r++) {
    const int slot = r * HR_NT + tid;
    const int64_t i = gidx(slot);
    const bool valid = i < n;
    s_a[slot] = valid ? (uint8_t)assign[i] : (uint8_t)0xff;
    s_ub[slot] = valid ? hr_up(ub[i]) : 0.0f;
    s_lb[slot] = valid ? hr_down(lb[i]) : 0.0f;
    s_w[slot] = valid && w ? w[i] : 1u;
  }
  if (tid < KCH) { s_move[tid] = tid < kk ? cmove[tid] : 0.0; s_half[tid] = tid < kk ? shalf[tid] : 0.0; }
  if (tid < kk * kk) {
    const int a = tid / kk, b2 = tid - a * kk;
    if (a < b2) s_pair[a * kk - a * (a + 1) / 2 + (b2 - a - 1)] = (uint16_t)(a | (b2 << 8));
  }
  if (tid == 0) { s_move[KCH] = cmove[k]; s_move[KCH + 1] = cmove[k + 1]; s_amax = (int)cmove[k + 2]; s_nlist = 0; s_nneed = 0; s_nmoved = 0; }
  unsigned epoch = 0;
  int it = it0, quiet_at = -1;
  __syncthreads();
  HR_STAMP(0);
  for (; it < max_iter; it++) {
    const int b = it % 3;
    // ---- pass over all owned points: the bounds move with the centroids; what they no longer prove goes on the list
    {
      const double dmax = s_move[KCH], dmax2 = s_move[KCH + 1];
      const int amax = s_amax;
      int cnt = 0;
      unsigned long long bal[HR_MAXR];
      bool listed[HR_MAXR];
#pragma unroll
      for (int r = 0; r < HR_MAXR; r++) {
        listed[r] = false;
        if (r < rounds) {
          const int slot = r * HR_NT + tid;
          const int a = s_a[slot];
          const bool valid = a != 0xff;
          const int ac = valid ? a : 0;
          const double un = (double)s_ub[slot] + s_move[ac];
          const double ln = (double)s_lb[slot] - (a == amax ? dmax2 : dmax);  // lb bounds the OTHER centroids: the own one's displacement does not loosen it
          const float unf = hr_up(un), lnf = hr_down(ln);
          if (valid) { s_ub[slot] = unf; s_lb[slot] = lnf; }
          listed[r] = valid && !((double)unf * (1.0 + H_ETA) < fmax(s_half[ac], (double)lnf) * (1.0 - H_ETA));
        }
        bal[r] = __builtin_amdgcn_ballot_w64(listed[r]);
        cnt += __popcll(bal[r]);
      }
      if (cnt) {  // (uniform in the wave)
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_nlist, cnt);
        base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
        for (int r = 0; r < HR_MAXR; r++) {
          if (listed[r]) s_list[base + __popcll(bal[r] & ((1ull << lane) - 1ull))] = (uint16_t)(r * HR_NT + tid);
          base += __popcll(bal[r]);
        }
      }
    }
    __syncthreads();
    HR_STAMP(1);
    // ---- the listed points: the distance to the own centroid tightens ub (16 lanes per point, 12 dimensions each: k_h_bounds' arithmetic -- the
    // partial sums add in another order than the scoring's chain; both stay within 2.2e-14 of the exact sum, far inside the factor 1 + 1e-12).
    // Still unproven -> the need list; the first HR_P of them leave their rows in LDS for the scoring.
    const int nlist = s_nlist;
    {
      auto rfetch = [&](int t0, int4 (&x)[3], int &slot) {
        const int t = t0 + grp;
        slot = s_list[t < nlist ? t : 0];
        const int4 *p = reinterpret_cast<const int4 *>(pts + gidx(slot) * D + l16 * 12);
        x[0] = p[0]; x[1] = p[1]; x[2] = p[2];
      };
      int4 x[3] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};
      int slot = 0;
      if (nlist > 0) rfetch(0, x, slot);
#pragma unroll 1
      for (int t0 = 0; t0 < nlist; t0 += HR_P) {
        const bool act = t0 + grp < nlist;
        const int4 v0 = x[0], v1 = x[1], v2 = x[2];
        const int cslot = slot;
        if (t0 + HR_P < nlist) rfetch(t0 + HR_P, x, slot);  // the next pass's rows, while this one's are summed
        const int a = s_a[cslot];
        const int pv[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
        double sd = 0.0;
#pragma unroll
        for (int j = 0; j < 12; j++) { const double d0 = __dsub_rn((double)pv[j], s_c[(l16 * 12 + j) * HR_PITCH + a]); sd = __fma_rn(d0, d0, sd); }
        sd = hr_sum16(sd);
        int np = -1;
        if (act && l16 == 0) {
          const float uf = hr_up(sqrt(sd) * (1.0 + 1e-12));
          s_ub[cslot] = uf;
          if (!((double)uf * (1.0 + H_ETA) < fmax(s_half[a], (double)s_lb[cslot]) * (1.0 - H_ETA))) { np = atomicAdd(&s_nneed, 1); s_need[np] = (uint16_t)cslot; }
        }
        np = __shfl(np, lane & 48);
        if (np >= 0 && np < HR_P) {
          int4 *dst = reinterpret_cast<int4 *>(s_rows + np * D + l16 * 12);
          dst[0] = v0; dst[1] = v1; dst[2] = v2;
        }
      }
    }
    __syncthreads();
    HR_STAMP(2);
    // ---- full scoring of the need list, HR_P points per pass, a lane per (point, centroid) pair
    const int nneed = s_nneed;
    int total_moved = 0;
#if TM_KMR_STAMPS
    if (tid == 0 && (nlist | nneed)) { atomicAdd(&st->stamps[HR_NSTAMP], (u64)nlist); atomicAdd(&st->stamps[HR_NSTAMP + 1], (u64)nneed); }
#endif
#pragma unroll 1
    for (int base = 0; base < nneed; base += HR_P) {
      if (base > 0) {  // (the first pass's rows came from the recheck)
#pragma unroll
        for (int u = 0; u < 3; u++) {
          const int piece = u * HR_NT + tid, pr = piece / 48, off = piece - pr * 48;
          const int sl = s_need[min(base + pr, nneed - 1)];
          reinterpret_cast<int4 *>(s_rows)[piece] = reinterpret_cast<const int4 *>(pts + gidx(sl) * D)[off];
        }
        __syncthreads();
      }
      const bool active = base + grp < nneed;
      const int slot = s_need[active ? base + grp : base];
      double bd = 1.0e300, bd2 = 1.0e300;
      int bc = 0x7fffffff;
      if (base + (wave << 2) < nneed) {  // (uniform in the wave: the waves without a point skip the chain)
        // 192 terms in order, four dimensions per step: the NEXT step's LDS reads (a 16-byte read of the row, four centroid values) are
        // issued before this step's arithmetic -- the scheduling barriers keep them there: left alone the compiler reads each operand right
        // before its use and waits out an LDS round trip every second term (24 000 cycles per pass with a lone wave per SIMD; the stamps)
        double sacc = 0.0;
        const int4 *rp = reinterpret_cast<const int4 *>(s_rows + grp * D);
        const double *cp = s_c + l16;
        auto ld = [&](int jb, int4 &r, double (&c)[4]) {
          r = rp[jb];
#pragma unroll
          for (int u = 0; u < 4; u++) c[u] = cp[(jb * 4 + u) * HR_PITCH];
        };
        auto acc = [&](const int4 &r, const double (&c)[4]) {
          const int v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
          for (int u = 0; u < 4; u++) { const double t = __dsub_rn((double)v[u], c[u]); sacc = __fma_rn(t, t, sacc); }
        };
        int4 ra, rb;
        double ca[4], cb[4];
        ld(0, ra, ca);
#pragma unroll
        for (int jb = 0; jb < D / 4; jb += 2) {
          ld(jb + 1, rb, cb);
          __builtin_amdgcn_sched_barrier(0);
          acc(ra, ca);
          __builtin_amdgcn_sched_barrier(0);
          if (jb + 2 < D / 4) ld(jb + 2, ra, ca);
          __builtin_amdgcn_sched_barrier(0);
          acc(rb, cb);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (l16 < kk) { bd = sacc; bc = l16; }
        // the 16 lanes of a point: the best by (distance, index); the second best distance = the smallest of the rest (whatever the pairing)
        auto merge = [&](auto ctrl) {
          constexpr int C = decltype(ctrl)::value;
          const double od = hr_dpp<C>(bd), od2 = hr_dpp<C>(bd2);
          const int oc = hr_dpp<C>(bc);
          const bool take = od < bd || (od == bd && oc < bc);
          const double loser = take ? bd : od;
          bd2 = fmin(fmin(bd2, od2), loser);
          if (take) { bd = od; bc = oc; }
        };
        merge(std::integral_constant<int, HR_X1>{}); merge(std::integral_constant<int, HR_X2>{});
        merge(std::integral_constant<int, HR_M8>{}); merge(std::integral_constant<int, HR_M16>{});
        if (active && l16 == 0) {
          s_ub[slot] = hr_up(sqrt(bd) * (1.0 + 1e-12));
          s_lb[slot] = hr_down(sqrt(bd2) * (1.0 - 1e-12));
          const int old = s_a[slot];
          if (old != bc) {
            s_a[slot] = (uint8_t)bc;
            const int m = atomicAdd(&s_nmoved, 1);
            s_moved[m * 3] = grp; s_moved[m * 3 + 1] = old; s_moved[m * 3 + 2] = bc;
            s_wt[grp] = s_w[slot];
          }
        }
      }
      __syncthreads();
      const int nmoved = s_nmoved;
      total_moved += nmoved;
      for (int e = wave; e < nmoved; e += HR_NT / 64) {  // a wave per moved row between the carried sums (its row is in LDS)
        const int ps = s_moved[e * 3], old = s_moved[e * 3 + 1], nw = s_moved[e * 3 + 2];
        const long long wi = (long long)s_wt[ps];
#pragma unroll
        for (int j = lane; j <= D; j += 64) {
          const u64 v = j < D ? (u64)(wi * s_rows[ps * D + j]) : (u64)wi;
          atomicAdd(&s_delta[nw * 193 + j], v);
          atomicAdd(&s_delta[old * 193 + j], (u64)0 - v);
        }
      }
      __syncthreads();  // the moved list and the rows have been read
      if (tid == 0) s_nmoved = 0;
    }
    HR_STAMP(3);
    // ---- this workgroup's deltas -> the iteration's buffer
    if (total_moved) {
      for (int e = tid; e < HR_E; e += HR_NT) {
        const u64 v = s_delta[e];
        if (v == 0) continue;
        s_delta[e] = 0;
        __hip_atomic_fetch_add(&st->delta[b][e], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (tid == 0) __hip_atomic_fetch_add(&st->changed[b], (unsigned)total_moved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    HR_STAMP(4);
    if (!hr_barrier(st, epoch, G, &s_ok)) return;
    HR_STAMP(5);
    if (tid == 0) { s_nlist = 0; s_nneed = 0; }  // (every thread has read them: the barrier; the next pass over the points comes behind further ones)
    // ---- every workgroup: the iteration's deltas into its own sums; new centroids; what the bounds need
    const unsigned tot = __hip_atomic_load(&st->changed[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
      u64 dv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int e = u * HR_NT + tid; dv[u] = e < HR_E ? __hip_atomic_load(&st->delta[b][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0; }
      // the buffer of the iteration after next: read for the last time before the barrier just passed, added to again only behind the next one
      const int b2 = (it + 2) % 3;
      const int per = (HR_E + (int)G - 1) / (int)G;
      if (tid < per && g * per + tid < HR_E) __hip_atomic_store(&st->delta[b2][g * per + tid], (u64)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g == 0 && tid == 0) __hip_atomic_store(&st->changed[b2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 4; u++) { const int e = u * HR_NT + tid; if (e < HR_E && dv[u]) s_sum[e] += dv[u]; }
    }
    if (tot == 0) { quiet_at = it; break; }
    if (tid < KCH) s_min[tid] = 0x7ff0000000000000ull;  // +inf
    __syncthreads();
    HR_STAMP(6);
    // a wave per centroid, 3 dimensions per lane: new position = exact integer sum / weight (one IEEE division; an empty cluster keeps its
    // centroid), displacement (rounded up; the sums here only feed the bounds, margins of 1e-9: their order is free)
    if (wave < kk) {
      const int c = wave;
      const u64 cn = s_sum[c * 193 + D];
      double sd = 0.0;
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int j = lane + 64 * u;
        const double old = s_c[j * HR_PITCH + c];
        double nw = old;
        if (cn > 0) { nw = __ddiv_rn((double)(long long)s_sum[c * 193 + j], (double)(long long)cn); s_c[j * HR_PITCH + c] = nw; }
        const double t = nw - old;
        sd += t * t;
      }
      sd = hr_sum16(sd);
      sd += __shfl_xor(sd, 16);
      sd += __shfl_xor(sd, 32);
      if (lane == 0) s_move[c] = sqrt(sd) * (1.0 + 1e-9);
    }
    __syncthreads();
    for (int pr = grp; pr < kk * (kk - 1) / 2; pr += HR_NT / 16) {  // pairwise distances, 16 lanes per pair: the smallest per centroid (non-negative doubles order like their bit patterns)
      const int ca = s_pair[pr] & 0xff, cb = s_pair[pr] >> 8;
      double sd = 0.0;
#pragma unroll
      for (int u = 0; u < 12; u++) { const int j = l16 + 16 * u; const double t = s_c[j * HR_PITCH + ca] - s_c[j * HR_PITCH + cb]; sd += t * t; }
      sd = hr_sum16(sd);
      if (l16 == 0) {
        atomicMin(&s_min[ca], (unsigned long long)__double_as_longlong(sd));
        atomicMin(&s_min[cb], (unsigned long long)__double_as_longlong(sd));
      }
    }
    if (tid == HR_NT - 1) {  // the largest displacement, the largest among the others, and whose the largest is (the first of several): sixteen values, one lane
      double mv[KCH];
#pragma unroll
      for (int c = 0; c < KCH; c++) mv[c] = s_move[c];
      double mx = 0.0, mx2 = 0.0;
      int amx = 0;
#pragma unroll
      for (int c = 0; c < KCH; c++) {
        const double v = c < kk ? mv[c] : 0.0;
        if (v > mx) { mx2 = mx; mx = v; amx = c; } else mx2 = fmax(mx2, v);
      }
      s_move[KCH] = mx; s_move[KCH + 1] = mx2; s_amax = amx;
    }
    __syncthreads();
    if (tid < kk) s_half[tid] = kk > 1 ? 0.5 * sqrt(__longlong_as_double((long long)s_min[tid])) * (1.0 - 1e-9) : 1.0e300;
    __syncthreads();
    HR_STAMP(7);
#if TM_KMR_STAMPS
    if (g == 0 && tid < 7 && it < 300) { st->log[it][tid] = (unsigned)(s_stamp[tid + 1] - s_prev[tid]); s_prev[tid] = s_stamp[tid + 1]; }
    if (g == 0 && tid == 0 && it < 300) { st->log[it][7] = (unsigned)nlist; st->log[it][8] = (unsigned)nneed; st->log[it][9] = tot; }
#endif
  }
  // ---- state out
  __syncthreads();
#if TM_KMR_STAMPS
  if (g == 0 && tid < HR_NSTAMP) st->stamps[tid] = s_stamp[tid];
#endif
  for (int r = 0; r < rounds; r++) {
    const int slot = r * HR_NT + tid;
    const int64_t i = gidx(slot);
    if (i < n) assign[i] = (int32_t)s_a[slot];
  }
  if (g == 0) {
    for (int e = tid; e < kk * D; e += HR_NT) { const int c = e / D, j = e - c * D; cent[e] = s_c[j * HR_PITCH + c]; }
    if (tid == 0) { segs[0].changed = 0; if (quiet_at >= 0) *quiet_iter = quiet_at; }
  }
}

// ---- D = 3, one launch for the whole clustering -----------------------------------------------------------------
// The pixel k-means of QuantizeUsingYakmo (tilingencoder.pas:4434-4532) runs ~180 Lloyd iterations over a few hundred thousand
// distinct colours per palette: a few microseconds of arithmetic per iteration, so as separate launches (two per iteration, two per
// farthest-first pick) it was bound by launch latency alone.  Here every workgroup keeps its 4096 points in REGISTERS for the whole
// clustering (packed colour, weight, assignment), the workgroups of one segment (= one palette) meet at a barrier of their own once
// per iteration (a counter in global memory: agent-scope release / acquire around a relaxed poll), and the segments run their own
// number of iterations side by side.  Everything that crosses workgroups is an integer atomic -- the carried sums and counts
// (exact, order-free), the farthest-first pick (64-bit max of distance << 32 | ~index: largest distance, then lowest index), the
// changed-points counter -- so the result is the one the multi-launch path and the oracle give, bit for bit.
#ifndef TM_KM3_STAMPS
#define TM_KM3_STAMPS 0
#endif
#if TM_KM3_STAMPS
#define P3_STAMP(i) do { if (bx == 0 && tid == 0) { const u64 t_ = __builtin_amdgcn_s_memtime(); st->stamps[i] += t_ - st_last; st_last = t_; } } while (0)
#else
#define P3_STAMP(i) do { } while (0)
#endif
#ifndef TM_KM3_NT
#define TM_KM3_NT 256
#endif
#ifndef TM_KM3_PPT
#define TM_KM3_PPT 16
#endif
#ifndef TM_KM3_PU
#define TM_KM3_PU 4  // points of a thread whose bounds are tested together (the dependent LDS round trips of a batch overlap)
#endif
constexpr int P3_PPT = TM_KM3_PPT, P3_NT = TM_KM3_NT, P3_ROWS = P3_PPT * P3_NT, P3_MAXK = 64, P3_NCOPY = 4;
// workgroups a CU is asked to hold (LDS: ten bytes a point + 12 KB).  Measured (round 3): 1024 x 12 and 512 x 20 / 24 points per workgroup, one
// per CU and a third as many participants at a palette's barrier, take 13.8-13.9 / 14.3 / 14.9 ms for PreparePalettes against 13.7 with 256 x 16
constexpr int P3_WGS = (P3_ROWS * 10 + 12288) * 3 <= 160 * 1024 && P3_NT * 3 <= 1024 ? 3 : 1;
static_assert(TM_KM3_NT != 256 || TM_KM3_PPT != 16 || P3_WGS == 3, "the shipped shape holds three workgroups per CU");
static_assert(P3_PPT % 4 == 0 && P3_ROWS <= 65535 && P3_NT >= 192, "pixel k-means shape");
constexpr int P3_UNIT = 128;  // the per-point distance bounds are 16-bit fixed point, 1/128 of a colour step (distances stay below 442)

struct Seg3 {
  int64_t begin, count;
  int blk_first, blk_count;
  int kk, iters;        // out
  int nseg;             // element 0 only
  int pad;
};
struct Seg3State {      // zeroed before the launch
  u64 sums[P3_MAXK][3];
  u64 cnts[P3_MAXK];
  u64 pick[P3_MAXK];
  unsigned changed[3], timeout, pad[4];
  // the barrier of the segment's workgroups, as k_h_resident's (HrState): arrivals counted in up to eight shards, a shard's last arrival adds to
  // eight replicas of `top`, the waiting workgroups poll their shard's replica; every word on a 128-byte line of its own
  struct alignas(128) Line { unsigned v; unsigned pad[31]; };
  Line bar[8], top[8];
#if TM_KM3_STAMPS
  u64 stamps[8];  // diagnostic build: s_memtime spans of workgroup 0's phases, summed over the iterations
#endif
};

__device__ __forceinline__ bool p3_barrier(Seg3State *st, unsigned &epoch, unsigned nblk, unsigned bx) {
  // every thread of the workgroup calls it; false: the spin gave up (a workgroup of the segment is not resident), the caller leaves.
  // No fence: everything that crosses workgroups here is an agent-scope atomic on both sides (adds and maxima, relaxed loads), every wave
  // drains its vmcnt before its workgroup arrives, every load of the data comes behind a workgroup barrier behind the poll -- the hand-off form
  // of MI355X_MICROARCH.md "Valid forms" that needs no L2 write-back and no L1 invalidate (1.7 us each, twice per iteration, before).
  __shared__ int s_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {
    epoch++;
    int ok = 1;
    if (nblk > 1) {
      const unsigned sh = bx & 7u, nsh = nblk < 8u ? nblk : 8u;
      const unsigned mine = (nblk - sh + 7u) >> 3;  // workgroups on this shard
      bool last = false;
      if (threadIdx.x == 0) last = __hip_atomic_fetch_add(&st->bar[sh].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == epoch * mine;
      last = __builtin_amdgcn_readfirstlane((int)last) != 0;
      if (last && threadIdx.x < 8) __hip_atomic_fetch_add(&st->top[threadIdx.x].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // one instruction, eight lines
      if (threadIdx.x == 0)
      for (unsigned spins = 1; __hip_atomic_load(&st->top[sh].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * nsh; spins++) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255u) == 0 && (spins > (1u << 24) || __hip_atomic_load(&st->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {  // (a second round trip: rarely)
          __hip_atomic_store(&st->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    if (threadIdx.x == 0) s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

__global__ __launch_bounds__(P3_NT, P3_WGS) void k_kmeans3_persistent(const int32_t *__restrict__ pts, const uint32_t *__restrict__ w, Seg3 *__restrict__ segs,
                                                             Seg3State *__restrict__ state, int k, int max_iter, int32_t *__restrict__ assign,
                                                             double *__restrict__ cent) {
  // the workgroup's points stay on chip for the whole clustering: packed colour and assignment in LDS (slot m * NT + tid: no bank
  // conflicts), so the loops over a thread's points stay rolled and the register file holds only the P3_G points in flight
  __shared__ double s_cent[P3_MAXK][3];
  __shared__ double s_dsq[P3_MAXK][3];  // squared displacement of each centroid coordinate in the last update
  __shared__ int s_half[P3_MAXK];      // half the distance to the nearest other centroid, rounded down (P3_UNIT)
  __shared__ int2 s_mh[P3_MAXK];       // (s_move, s_half) of a centroid side by side: the pass over all points fetches both with one read
  __shared__ int s_move[P3_MAXK + 3];  // displacement of each centroid in the last update, rounded up; then the largest, the second largest, whose
  __shared__ uint32_t s_col[P3_ROWS];   // colour | assignment << 24 (0xff: none yet, 0xfe: slot past the end of the segment)
  // farthest-first distances, then the points' bounds: ub >= the distance to the own centroid, lb <= the distance to every other one
  __shared__ union { int md[P3_ROWS]; uint32_t bnd[P3_ROWS]; } s_u;  // bnd: ub | lb << 16
  __shared__ u64 s_acc[P3_NCOPY][P3_MAXK][4];                      // the sums' deltas of one iteration
  __shared__ uint16_t s_list[P3_NT / 64][P3_ROWS / (P3_NT / 64)];  // every wave's list of the points it has to score
  __shared__ u64 s_red[P3_NT / 64];
  __shared__ int s_chg;
  const int tid = threadIdx.x;
  int seg;
  {
    int lo = 0, hi = segs[0].nseg - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (segs[mid].blk_first <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    seg = lo;
  }
  const Seg3 sg = segs[seg];
  const int bx = (int)blockIdx.x - sg.blk_first;
  const unsigned nbx = (unsigned)sg.blk_count;
  Seg3State *st = state + seg;
  unsigned epoch = 0;
  const int64_t base = (int64_t)bx * P3_ROWS;
  for (int r = tid; r < P3_ROWS; r += P3_NT) {
    uint32_t cc = 0xfe000000u;
    if (base + r < sg.count) {
      const int32_t *p = pts + (sg.begin + base + r) * 3;
      cc = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | 0xff000000u;
    }
    s_col[r] = cc;
    s_u.md[r] = INT_MAX;
  }
  // ---- farthest-first from the segment's first point
  int kk = 1;
  int cr, cg, cb;
  {
    const int32_t *p0 = pts + sg.begin * 3;
    cr = p0[0]; cg = p0[1]; cb = p0[2];
  }
  if (tid < 3) s_cent[0][tid] = (double)(tid == 0 ? cr : tid == 1 ? cg : cb);
  for (int c = 1; c < k; c++) {
    u64 best = 0;
    for (int r = tid; r < P3_ROWS; r += P3_NT) {  // (each thread only ever touches its own slots: no barrier needed for s_col / md)
      const uint32_t cc = s_col[r];
      if ((cc >> 24) == 0xfeu) continue;
      const int dr = (int)(cc & 0xff) - cr, dg = (int)((cc >> 8) & 0xff) - cg, db = (int)((cc >> 16) & 0xff) - cb;
      const int m = min(s_u.md[r], dr * dr + dg * dg + db * db);
      s_u.md[r] = m;
      const u64 key = ((u64)(uint32_t)m << 32) | (u64)(0xffffffffu - (uint32_t)(base + r));
      best = key > best ? key : best;
    }
    for (int o = 32; o > 0; o >>= 1) { const u64 other = __shfl_xor(best, o); best = other > best ? other : best; }
    if ((tid & 63) == 0) s_red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
      for (int wv = 1; wv < P3_NT / 64; wv++) best = s_red[wv] > best ? s_red[wv] : best;
      if (best >> 32) atomicMax(&st->pick[c], best);
    }
    if (!p3_barrier(st, epoch, nbx, (unsigned)bx)) return;
    const u64 win = __hip_atomic_load(&st->pick[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((win >> 32) == 0) break;  // no distinct point left
    const int32_t *pc = pts + (sg.begin + (int64_t)(0xffffffffu - (uint32_t)win)) * 3;
    cr = pc[0]; cg = pc[1]; cb = pc[2];
    if (tid < 3) s_cent[kk][tid] = (double)(tid == 0 ? cr : tid == 1 ? cg : cb);
    kk++;
  }
  // ---- Lloyd
  // Per point two bounds are kept (Hamerly): ub >= its distance to its own centroid, lb <= its distance to every other one; a centroid
  // update loosens them by the centroids' displacements.  While ub <= max(lb, half the distance from the own centroid to the nearest
  // other one) the own centroid is strictly the nearest -- every bound is rounded the safe way to 1/128 and carries a margin of a
  // whole unit, six orders of magnitude above the rounding of the distance arithmetic, so the order of the COMPUTED distances is the
  // same and strict -- and the point keeps its assignment without being scored.  A point that fails first gets its ub tightened
  // (distance to its own centroid, in the arithmetic of the scoring); if it still fails it goes on its wave's list and is scored
  // against every centroid as before.  Each wave compacts and scores its own 1024 points: no barrier between the two.
  int it = 0;
#if TM_KM3_STAMPS
  u64 st_last = __builtin_amdgcn_s_memtime();
#endif
  const int wave = tid >> 6, lane = tid & 63;
  if (tid < P3_MAXK) { s_move[tid] = 0; s_half[tid] = 0; s_mh[tid] = make_int2(0, 0); }
  if (tid < 3) s_move[P3_MAXK + tid] = 0;
  for (;;) {
    __syncthreads();  // s_cent, s_move, s_half of this iteration are in place; the farthest-first pass is done with the union; the previous flush with s_acc
    for (int e = tid; e < P3_NCOPY * P3_MAXK * 4; e += P3_NT) (&s_acc[0][0][0])[e] = 0;
    if (tid == 0) s_chg = 0;
    __syncthreads();
    P3_STAMP(0);  // zeroing
    const int mv1 = s_move[P3_MAXK], mv2 = s_move[P3_MAXK + 1], amax = s_move[P3_MAXK + 2];
    int nlist = 0;  // uniform in the wave
    uint16_t *const mylist = s_list[wave];
    // Pass A, every point: the bounds loosened by the centroids' displacements; the points whose bounds no longer prove their assignment (and
    // the ones not assigned yet) go on the wave's list.  Pass B, the listed points only, 64 at a time: the real distance to the own centroid
    // as the new ub; what still fails stays on the list (compacted in place, order kept) and is scored.  One in seven points is listed, one
    // in fifteen scored: with the recheck inside pass A every wave ran its double-precision arithmetic for all of its points, because some
    // lane of every batch needed it.
    int ntight = 0;
    constexpr int PU = TM_KM3_PU;  // points of a thread in flight: their LDS round trips overlap
#pragma unroll 1
    for (int m0 = 0; m0 < P3_PPT; m0 += PU) {
      uint32_t cc[PU], bn[PU];
#pragma unroll
      for (int i = 0; i < PU; i++) {
        const int r = (m0 + i) * P3_NT + tid;
        cc[i] = s_col[r];
        bn[i] = s_u.bnd[r];
      }
      // (no branch around a point's table look-up: the compiler then waits for every look-up on its own, one LDS round trip after the
      // other; fetched for all the points of the batch at once -- entry 62 / 63 for the slots without a centroid, never used -- they overlap)
      int2 mh[PU];
#pragma unroll
      for (int i = 0; i < PU; i++) mh[i] = s_mh[(cc[i] >> 24) & (P3_MAXK - 1)];
#pragma unroll
      for (int i = 0; i < PU; i++) {
        const int r = (m0 + i) * P3_NT + tid;
        const int a = (int)(cc[i] >> 24);
        const bool has = a < 0xfe;
        const int u = min(65535, (int)(bn[i] & 0xffffu) + mh[i].x);
        const int l = max(0, (int)(bn[i] >> 16) - (a == amax ? mv2 : mv1));  // lb bounds the OTHER centroids: the own one's displacement does not loosen it
        s_u.bnd[r] = has ? ((uint32_t)u | ((uint32_t)l << 16)) : bn[i];
        const bool listed = has ? u > max(l, mh[i].y) : a == 0xff;  // (not assigned yet: scored)
        const unsigned long long tb = __builtin_amdgcn_ballot_w64(listed);
        if (listed) mylist[ntight + __popcll(tb & ((1ull << lane) - 1ull))] = (uint16_t)r;
        ntight += __popcll(tb);
#if TM_KM3_STAMPS
        if (bx == 0 && lane == 0) atomicAdd(&st->stamps[7], (u64)__popcll(tb) << 32);
#endif
      }
    }
#pragma unroll 1
    for (int e0 = 0; e0 < ntight; e0 += 64) {
      const bool valid = e0 + lane < ntight;
      const int r = valid ? (int)mylist[e0 + lane] : tid;
      const uint32_t cc = s_col[r], bn = s_u.bnd[r];
      const int a = (int)(cc >> 24);
      bool full = valid;
      if (valid && a < 0xfe) {
        // the distance to the own centroid, in the scoring's arithmetic, as the new ub (single-precision root, as in the scoring)
        const double t0 = __dsub_rn((double)(int)(cc & 0xff), s_cent[a][0]), t1 = __dsub_rn((double)(int)((cc >> 8) & 0xff), s_cent[a][1]),
                     t2 = __dsub_rn((double)(int)((cc >> 16) & 0xff), s_cent[a][2]);
        const double sd = __fma_rn(t2, t2, __fma_rn(t1, t1, __fma_rn(t0, t0, 0.0)));
        const int u = min(65535, (int)(__fsqrt_rn((float)sd) * (float)P3_UNIT) + 2), l = (int)(bn >> 16);
        s_u.bnd[r] = (uint32_t)u | ((uint32_t)l << 16);
        full = u > max(l, s_half[a]);
      }
      const unsigned long long fb = __builtin_amdgcn_ballot_w64(full);
      if (full) mylist[nlist + __popcll(fb & ((1ull << lane) - 1ull))] = (uint16_t)r;  // (nlist <= e0: never over an entry still to be read)
      nlist += __popcll(fb);
#if TM_KM3_STAMPS
      if (bx == 0 && lane == 0) atomicAdd(&st->stamps[7], (u64)__popcll(fb));
#endif
    }
    P3_STAMP(1);  // bounds of the 16 passes
    int changed = 0;
    // G listed points per lane at a time against one centroid after the other: a centroid read from LDS (a broadcast) serves G points.
    // Per (point, centroid): sum over dimensions in order of (p - c)^2, one IEEE subtraction and one fused multiply-add each;
    // ties -> lowest centroid.
    auto score = [&](auto gtag, const int e0) {
      constexpr int G = decltype(gtag)::value;
      double px[G][3], bd[G], bd2[G];
      int bc[G], rr[G];
      uint32_t cc[G], wv[G];
#pragma unroll
      for (int m = 0; m < G; m++) {
        const int e = e0 + m * 64 + lane;
        rr[m] = e < nlist ? (int)mylist[e] : -1;
        // the weight is only needed if the point moves, but it comes from memory: asked for now, it arrives under the distance arithmetic
        // instead of behind it (a round trip to L2 or HBM at the end of every scoring step of every iteration)
        wv[m] = (w && rr[m] >= 0) ? w[sg.begin + base + rr[m]] : 1u;
        cc[m] = s_col[rr[m] < 0 ? tid : rr[m]];
        px[m][0] = (double)(int)(cc[m] & 0xff); px[m][1] = (double)(int)((cc[m] >> 8) & 0xff); px[m][2] = (double)(int)((cc[m] >> 16) & 0xff);
        bd[m] = 0.0; bd2[m] = 1.0e300;
        bc[m] = -1;
      }
#pragma unroll 1
      for (int c = 0; c < kk; c++) {
        const double c0 = s_cent[c][0], c1 = s_cent[c][1], c2 = s_cent[c][2];
#pragma unroll
        for (int m = 0; m < G; m++) {
          const double t0 = __dsub_rn(px[m][0], c0), t1 = __dsub_rn(px[m][1], c1), t2 = __dsub_rn(px[m][2], c2);
          const double sd = __fma_rn(t2, t2, __fma_rn(t1, t1, __fma_rn(t0, t0, 0.0)));
          if (bc[m] < 0 || sd < bd[m]) { bd2[m] = bc[m] < 0 ? bd2[m] : bd[m]; bd[m] = sd; bc[m] = c; }
          else if (sd < bd2[m]) bd2[m] = sd;
        }
      }
#pragma unroll
      for (int m = 0; m < G; m++) {
        const int r = rr[m];
        if (r < 0) continue;
        // single-precision roots: their error (2e-7 relative, 0.011 units at most) is far inside the margins of a whole unit
        s_u.bnd[r] = (uint32_t)min(65535, (int)(__fsqrt_rn((float)bd[m]) * (float)P3_UNIT) + 2) |
                     ((uint32_t)(bd2[m] > 1.0e12 ? 65535 : max(0, (int)(__fsqrt_rn((float)bd2[m]) * (float)P3_UNIT) - 1)) << 16);
        const int old = (int)(cc[m] >> 24);
        if (old == bc[m]) continue;
        // only a point that changes cluster touches the carried sums
        const long long wi = (long long)wv[m];
        const int pi[3] = {(int)(cc[m] & 0xff), (int)((cc[m] >> 8) & 0xff), (int)((cc[m] >> 16) & 0xff)};
        u64 *acc = &s_acc[tid & (P3_NCOPY - 1)][bc[m]][0];
        atomicAdd(&acc[3], (u64)wi);
#pragma unroll
        for (int j = 0; j < 3; j++) atomicAdd(&acc[j], (u64)(wi * pi[j]));
        if (old != 0xff) {
          u64 *oacc = &s_acc[tid & (P3_NCOPY - 1)][old][0];
          atomicAdd(&oacc[3], (u64)0 - (u64)wi);
#pragma unroll
          for (int j = 0; j < 3; j++) atomicAdd(&oacc[j], (u64)0 - (u64)(wi * pi[j]));
        }
        s_col[r] = (cc[m] & 0xffffffu) | ((uint32_t)bc[m] << 24);
        changed++;
      }
    };
    {  // 256 listed points per step while there are many, then 128, then 64: a short list costs one short step
      int e0 = 0;
      for (; nlist - e0 > 128; e0 += 256) score(std::integral_constant<int, 4>{}, e0);
      if (nlist - e0 > 64) { score(std::integral_constant<int, 2>{}, e0); e0 += 128; }
      if (nlist - e0 > 0) score(std::integral_constant<int, 1>{}, e0);
    }
    P3_STAMP(2);  // full scoring of the listed points
    for (int o = 32; o > 0; o >>= 1) changed += __shfl_xor(changed, o);
    if ((tid & 63) == 0 && changed) atomicAdd(&s_chg, changed);
    __syncthreads();
    P3_STAMP(3);  // waiting for the workgroup's other waves
    for (int e = tid; e < kk * 4; e += P3_NT) {
      u64 v = 0;
#pragma unroll
      for (int cp = 0; cp < P3_NCOPY; cp++) v += s_acc[cp][e >> 2][e & 3];
      if (v == 0) continue;
      if ((e & 3) == 3) atomicAdd(&st->cnts[e >> 2], v); else atomicAdd(&st->sums[e >> 2][e & 3], v);
    }
    if (tid == 0 && s_chg) atomicAdd(&st->changed[it % 3], (unsigned)s_chg);
    P3_STAMP(4);  // flush
    if (!p3_barrier(st, epoch, nbx, (unsigned)bx)) return;
    P3_STAMP(5);  // barrier of the segment's workgroups
    // (the three loads leave together: one round trip instead of three)
    const int uc = min(tid / 3, P3_MAXK - 1), uj = tid - (tid / 3) * 3;
    const unsigned tot = __hip_atomic_load(&st->changed[it % 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 cn = __hip_atomic_load(&st->cnts[uc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 sm = __hip_atomic_load(&st->sums[uc][uj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bx == 0 && tid == 0) __hip_atomic_store(&st->changed[(it + 2) % 3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // last read two barriers ago
    if (tot == 0) break;
    // new centroids: exact integer sum / weight, one IEEE division; an empty cluster keeps its centroid
    if (tid < kk * 3) {
      const int c = uc, j = uj;
      const double old = s_cent[c][j];
      double nw = old;
      if (cn > 0) nw = __ddiv_rn((double)(long long)sm, (double)(long long)cn);
      s_cent[c][j] = nw;
      const double dd = nw - old;
      s_dsq[c][j] = dd * dd;
    }
    if (tid < P3_MAXK) s_half[tid] = 0x7fffffff;
    __syncthreads();
    // what the bounds need (identical in every workgroup of the segment): how far each centroid moved, rounded up (a centroid that did
    // not move costs nothing), and half its distance to the nearest other one, rounded down
    if (tid < kk) {
      const double m2 = s_dsq[tid][0] + s_dsq[tid][1] + s_dsq[tid][2];
      s_move[tid] = m2 == 0.0 ? 0 : (int)(__fsqrt_rn((float)m2) * (float)P3_UNIT) + 2;
    }
    for (int pr = tid; pr < kk * kk; pr += P3_NT) {
      const int ca = pr / kk, cb2 = pr - ca * kk;
      if (ca >= cb2) continue;
      const double t0 = s_cent[ca][0] - s_cent[cb2][0], t1 = s_cent[ca][1] - s_cent[cb2][1], t2 = s_cent[ca][2] - s_cent[cb2][2];
      // (single-precision root, as everywhere the bounds are made: its error, 0.006 units at most, is far inside the whole unit of margin;
      // the double-precision one is a few dozen instructions on every iteration's critical path)
      const int h = max(0, (int)(0.5f * __fsqrt_rn((float)(t0 * t0 + t1 * t1 + t2 * t2)) * (float)P3_UNIT) - 1);
      atomicMin(&s_half[ca], h);
      atomicMin(&s_half[cb2], h);
    }
    __syncthreads();
    if (tid < kk) s_mh[tid] = make_int2(s_move[tid], s_half[tid]);  // (both final: the barrier above)
    if (wave == 0) {  // the largest displacement, whose it is (the first, if several), and the largest among the others: over the lanes of one
                      // wave (a thread walking the centroids read them one after the other: sixteen dependent LDS round trips per iteration)
      static_assert(P3_MAXK <= 64, "one lane per centroid");
      const int v = lane < kk ? s_move[lane] : 0;
      int m1 = v;
      for (int o = 32; o > 0; o >>= 1) m1 = max(m1, __shfl_xor(m1, o));
      const int am = __builtin_ctzll(__builtin_amdgcn_ballot_w64(v == m1 && (lane < kk || m1 == 0)));
      int m2 = lane == am ? 0 : v;
      for (int o = 32; o > 0; o >>= 1) m2 = max(m2, __shfl_xor(m2, o));
      if (lane == 0) { s_move[P3_MAXK] = m1; s_move[P3_MAXK + 1] = m2; s_move[P3_MAXK + 2] = am; }
    }
    P3_STAMP(6);  // counts + sums read back, new centroids
    it++;
    if (it >= max_iter) break;
  }
  __syncthreads();
  for (int r = tid; r < P3_ROWS; r += P3_NT)
    if (base + r < sg.count) assign[sg.begin + base + r] = (int32_t)(s_col[r] >> 24);
  if (bx == 0) {
    for (int e = tid; e < kk * 3; e += P3_NT) cent[((int64_t)seg * k + e / 3) * 3 + e % 3] = s_cent[e / 3][e % 3];
    if (tid == 0) { segs[seg].kk = kk; segs[seg].iters = it; }
  }
}

// host side of the above; *used = 0 when the shape does not fit one resident launch (the caller then takes the multi-launch path)
static int kmeans3_persistent(const int32_t *pts, const uint32_t *w, const std::vector<int64_t> &seg_begin, const std::vector<int64_t> &seg_count, int k,
                              int max_iter, int32_t *assign, double *cent, std::vector<int> *host_kk, int *host_iters, hipStream_t stream, int *used) {
  *used = 0;
  const int nseg = (int)seg_begin.size();
  if (k > P3_MAXK) return TM_OK;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  std::vector<Seg3> hs;  // empty segments take no workgroup and are answered on the host
  std::vector<int> which;
  int nblk = 0;
  for (int s = 0; s < nseg; s++) {
    if (seg_count[s] <= 0) continue;
    Seg3 g;
    memset(&g, 0, sizeof(g));
    g.begin = seg_begin[s]; g.count = seg_count[s];
    g.blk_first = nblk;
    g.blk_count = (int)((seg_count[s] + P3_ROWS - 1) / P3_ROWS);
    nblk += g.blk_count;
    hs.push_back(g);
    which.push_back(s);
  }
  // All workgroups of a launch must be resident together: what the runtime says fits a CU (registers, LDS, the launch bound of three), not a
  // guess.  More colours than the chip holds at once (3.1 M: the motion-prediction configurations of the bench clip) go as several launches,
  // each a run of whole segments (palettes are independent), one after the other on the stream.
  std::vector<std::pair<size_t, size_t>> batches;  // [first, last) of hs
  {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_kmeans3_persistent, P3_NT, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    per_cu = std::min(per_cu, P3_WGS);
    const int capacity = per_cu * cus;
    size_t b0 = 0;
    int acc = 0;
    for (size_t i = 0; i < hs.size(); i++) {
      if (hs[i].blk_count > capacity) return TM_OK;  // one palette alone does not fit: the launches-per-iteration path
      if (acc + hs[i].blk_count > capacity) { batches.push_back({b0, i}); b0 = i; acc = 0; }
      acc += hs[i].blk_count;
    }
    if (b0 < hs.size()) batches.push_back({b0, hs.size()});
  }
  if (host_kk) host_kk->assign(nseg, 0);
  if (host_iters) *host_iters = 0;
  *used = 1;
  if (hs.empty()) return TM_OK;
  for (const auto &b : batches) {  // a launch sees its own segments only: workgroup indices from 0, the count in its first element
    const int base = hs[b.first].blk_first;
    for (size_t i = b.first; i < b.second; i++) hs[i].blk_first -= base;
    hs[b.first].nseg = (int)(b.second - b.first);
  }
  DevBuf dsegs, dstate, dcent;
  TM_TRY(dsegs.alloc(sizeof(Seg3) * hs.size()));
  TM_TRY(dstate.alloc(sizeof(Seg3State) * hs.size()));
  TM_TRY(dcent.alloc(sizeof(double) * hs.size() * k * 3));
  TM_HIP(hipMemcpyAsync(dsegs.p, hs.data(), sizeof(Seg3) * hs.size(), hipMemcpyHostToDevice, stream));
  TM_HIP(hipMemsetAsync(dstate.p, 0, sizeof(Seg3State) * hs.size(), stream));
  TM_HIP(hipMemsetAsync(dcent.p, 0, sizeof(double) * hs.size() * k * 3, stream));
  TM_CHECK(nblk >= 1 && k >= 1 && k <= P3_MAXK, TM_E_INVAL, "k-means: resident launch of %d workgroups for %d centres (at most %d)", nblk, k, P3_MAXK);
  std::unique_lock<std::mutex> resident_lock(g_resident_launch);
  for (const auto &b : batches) {
    int grid = 0;
    for (size_t i = b.first; i < b.second; i++) grid += hs[i].blk_count;
    hipLaunchKernelGGL(k_kmeans3_persistent, dim3(grid), dim3(P3_NT), 0, stream, pts, w, dsegs.as<Seg3>() + b.first, dstate.as<Seg3State>() + b.first, k, max_iter, assign,
                       dcent.as<double>() + b.first * (size_t)k * 3);
  }
  TM_HIP(hipGetLastError());
  std::vector<Seg3State> hstate(hs.size());
  std::vector<double> hcent(hs.size() * (size_t)k * 3);
  {
    HostRead hr_(stream);
    TM_TRY(hr_.get(hs.data(), dsegs.p, sizeof(Seg3) * hs.size()));
    TM_TRY(hr_.get(hstate.data(), dstate.p, sizeof(Seg3State) * hs.size()));
    TM_TRY(hr_.get(hcent.data(), dcent.p, hcent.size() * 8));
    TM_TRY(hr_.wait());
  }
  resident_lock.unlock();
  int iters = 0;
  for (size_t i = 0; i < hs.size(); i++)
    if (hstate[i].timeout != 0) {  // a workgroup of a segment never became resident (the barrier gave up): the launches-per-iteration path instead
      fprintf(stderr, "[tm_kmeans] the resident pixel k-means gave up at its barrier (segment %zu); falling back to one launch per iteration\n", i);
      *used = 0;
      return TM_OK;
    }
  kmeans_run_stats().pixel_colour_iters = 0;
  for (size_t i = 0; i < hs.size(); i++) {
    if (host_kk) (*host_kk)[which[i]] = hs[i].kk;
    iters = std::max(iters, hs[i].iters);
    kmeans_run_stats().pixel_colour_iters += hs[i].count * (int64_t)hs[i].iters;
#if TM_KM3_STAMPS
    fprintf(stderr, "[tm_km3 stamps] segment %zu: %d workgroups, %d iterations; per iteration (s_memtime ticks): zero+thresholds %.0f, own test %.0f, scoring %.0f, "
            "workgroup sync %.0f, flush %.0f, barrier %.0f, read-back %.0f\n", i, hs[i].blk_count, hs[i].iters, (double)hstate[i].stamps[0] / std::max(1, hs[i].iters),
            (double)hstate[i].stamps[1] / std::max(1, hs[i].iters), (double)hstate[i].stamps[2] / std::max(1, hs[i].iters), (double)hstate[i].stamps[3] / std::max(1, hs[i].iters),
            (double)hstate[i].stamps[4] / std::max(1, hs[i].iters), (double)hstate[i].stamps[5] / std::max(1, hs[i].iters), (double)hstate[i].stamps[6] / std::max(1, hs[i].iters));
    fprintf(stderr, "[tm_km3 stamps]   workgroup 0: %.1f points per iteration fail the own-centroid test, %.1f are scored (whole passes)\n",
            (double)(hstate[i].stamps[7] >> 32) / std::max(1, hs[i].iters), (double)(hstate[i].stamps[7] & 0xffffffffu) / std::max(1, hs[i].iters));
#endif
  }
  if (host_iters) *host_iters = iters;
  // centroids back in the caller's [nseg][k][3] layout (device)
  std::vector<double> full((size_t)nseg * k * 3, 0.0);
  for (size_t i = 0; i < hs.size(); i++) memcpy(&full[(size_t)which[i] * k * 3], &hcent[i * (size_t)k * 3], sizeof(double) * k * 3);
  TM_HIP(hipMemcpyAsync(cent, full.data(), full.size() * 8, hipMemcpyHostToDevice, stream));
  TM_HIP(hipStreamSynchronize(stream));
  return TM_OK;
}

// ---- driver ----------------------------------------------------------------------------------------------------
// Batched k-means over nseg contiguous segments.  seg_begin/seg_count are host arrays.  Outputs assign (global point
// order), cent [nseg][k][d], host_kk[nseg] live centroid counts.
// the caller's own initial centres instead of the farthest-first picks: point indices (relative to the segment), -1 = none
__global__ void k_seed_centres(Seg *__restrict__ segs, int nseg, int k, const int32_t *__restrict__ pts, int d, const long long *__restrict__ idx, double *__restrict__ cent) {
  const int seg = blockIdx.x;
  if (seg >= nseg) return;
  Seg sg = segs[seg];
  int kk = 0;
  for (int c = 0; c < k; c++) {
    const long long i = idx[(int64_t)seg * k + c];
    if (i < 0 || i >= sg.count) continue;  // uniform over the workgroup
    for (int j = threadIdx.x; j < d; j += blockDim.x) cent[((int64_t)seg * k + kk) * d + j] = (double)pts[(sg.begin + i) * d + j];
    kk++;
  }
  __syncthreads();
  if (threadIdx.x == 0) { sg.kk = kk; sg.init_done = 1; sg.changed = 0; segs[seg] = sg; }
}

static int kmeans_batched(const int32_t *pts, const uint32_t *w, int d, const std::vector<int64_t> &seg_begin,
                          const std::vector<int64_t> &seg_count, int k, int max_iter, int32_t *assign, double *cent,
                          std::vector<int> *host_kk, int *host_iters, hipStream_t stream, const int64_t *init_idx = nullptr,
                          const long long *dev_init_idx = nullptr /* the same on the device (D2 seeding leaves them there) */, bool allow_resident = true) {
  TM_CHECK(d == 3 || d == 192, TM_E_INVAL, "kmeans: only d = 3 (pixels) or 192 (tile features) are built");
  TM_CHECK(k >= 1 && k <= 65536, TM_E_INVAL, "kmeans: k out of range");
  const int nseg = (int)seg_begin.size();
  if (host_iters) *host_iters = 0;
  if (nseg == 0) return TM_OK;
  const bool seeded = init_idx != nullptr || dev_init_idx != nullptr;
  if (d == 3 && !seeded) {  // the whole clustering in one launch when its workgroups fit the chip together
    int used = 0;
    TM_TRY(kmeans3_persistent(pts, w, seg_begin, seg_count, k, max_iter, assign, cent, host_kk, host_iters, stream, &used));
    if (used) return TM_OK;
  }
  int64_t n = 0, maxcount = 0;
  for (int s = 0; s < nseg; s++) { n = std::max(n, seg_begin[s] + seg_count[s]); maxcount = std::max(maxcount, seg_count[s]); }
  std::vector<Seg> hs(nseg);
  for (int s = 0; s < nseg; s++) { memset(&hs[s], 0, sizeof(Seg)); hs[s].begin = seg_begin[s]; hs[s].count = seg_count[s]; }
  // workgroups are shared out in proportion to segment size (about 4 per CU in total), so a large palette does not
  // leave most of the chip idle while its few workgroups loop
  int64_t total_pts = 0;
  for (int s = 0; s < nseg; s++) total_pts += seg_count[s];
  constexpr int blk_target = 768;
  const int64_t rows_per_blk = std::max<int64_t>(256, (total_pts / blk_target + 255) / 256 * 256);
  int nblk = 0;
  for (int s = 0; s < nseg; s++) {
    hs[s].blk_first = nblk;
    hs[s].blk_count = (int)((seg_count[s] + rows_per_blk - 1) / rows_per_blk);
    nblk += hs[s].blk_count;
  }
  hs[0].nseg = nseg;
  nblk = std::max(nblk, 1);
  DevBuf dsegs, mind, partial, sums, cnts, flag;
  TM_TRY(dsegs.alloc(sizeof(Seg) * nseg));
  TM_TRY(mind.alloc(std::max<int64_t>(n, 1) * 8));
  TM_TRY(partial.alloc(sizeof(BestKey) * (size_t)nblk));
  TM_TRY(sums.alloc((size_t)nseg * k * d * 8));
  TM_TRY(cnts.alloc((size_t)nseg * k * 8));
  TM_TRY(flag.alloc(4));
  TM_HIP(hipMemcpyAsync(dsegs.p, hs.data(), sizeof(Seg) * nseg, hipMemcpyHostToDevice, stream));
  if (!seeded) TM_HIP(hipMemsetAsync(mind.p, 0x7f, std::max<int64_t>(n, 1) * 8, stream));  // 0x7f7f... ~ 9.2e18 > any distance (the farthest-first picks' running minima)
  TM_HIP(hipMemsetAsync(sums.p, 0, (size_t)nseg * k * d * 8, stream));
  TM_HIP(hipMemsetAsync(cnts.p, 0, (size_t)nseg * k * 8, stream));
  TM_HIP(hipMemsetAsync(assign, 0xff, std::max<int64_t>(n, 1) * 4, stream));
  TM_HIP(hipMemsetAsync(cent, 0, (size_t)nseg * k * d * 8, stream));
  Seg *ds = dsegs.as<Seg>();
  const int sg_grid = (nseg + 63) / 64;
  DevBuf didx;
  if (dev_init_idx) {
    hipLaunchKernelGGL(k_seed_centres, dim3(nseg), dim3(64), 0, stream, ds, nseg, k, pts, d, dev_init_idx, cent);
  } else if (init_idx) {
    TM_TRY(didx.alloc((size_t)nseg * k * 8));
    TM_HIP(hipMemcpyAsync(didx.p, init_idx, (size_t)nseg * k * 8, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_seed_centres, dim3(nseg), dim3(64), 0, stream, ds, nseg, k, pts, d, didx.as<long long>(), cent);
  } else
  hipLaunchKernelGGL(k_ff_first, dim3(sg_grid), dim3(64), 0, stream, ds, nseg, k, pts, d, cent);
  for (int c = 1; c < k && !seeded; c++) {  // k-1 further picks (segments with no distinct point left latch init_done)
    if (d == 3)
      hipLaunchKernelGGL(k_ff_update<3>, dim3(nblk), dim3(256), 0, stream, pts, ds, k, mind.as<long long>(), partial.as<BestKey>());
    else
      hipLaunchKernelGGL(k_ff_update_wide, dim3(nblk), dim3(256), 0, stream, pts, d, ds, k, mind.as<long long>(), partial.as<BestKey>());
    hipLaunchKernelGGL(k_ff_pick, dim3(nseg), dim3(256), 0, stream, ds, nseg, k, partial.as<BestKey>(), nblk, pts, d, cent);
  }
  TM_HIP(hipGetLastError());
  const size_t lds_assign = (size_t)KCH * 3 * 8 + (size_t)16 * k * 4 * 8;
  const size_t lds_acc = std::min<size_t>((size_t)k * (d + 1) * 8, 64 * 1024);
  const bool fuse3 = d == 3 && (size_t)16 * k * 4 * 8 <= 48 * 1024;
  // D = 192: slices of the largest segment sized so that one round of workgroups fills the chip evenly
  int ppt192 = 1, nblk192 = 1, rows192 = 256, lds_delta192 = 0;
  size_t lds192 = 0;
  DevBuf ptsc;
  if (d == 192) {
    TM_TRY(ptsc.alloc((size_t)std::max<int64_t>(n, 1) * 192 * 4));
    hipLaunchKernelGGL(k_chunk_major, dim3((unsigned)std::min<int64_t>((n * 48 + 255) / 256, 8192)), dim3(256), 0, stream, pts, n, ptsc.as<int32_t>());
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    constexpr int occ = 2;  // workgroups per CU the slices are sized for
    const int64_t slots = std::max<int64_t>(1, (int64_t)cus * occ / std::max(1, std::min(nseg, cus * occ)));  // workgroups per segment in one round
    const int64_t per_slot = (maxcount + slots - 1) / slots;
    const int64_t rounds = (per_slot + 256 * 5 - 1) / (256 * 5);
    rows192 = (int)std::max<int64_t>(1, (per_slot + rounds - 1) / rounds);
    ppt192 = (rows192 + 255) / 256;
    nblk192 = (int)((maxcount + rows192 - 1) / rows192);
    const size_t fixed = (size_t)A_DCH * KCH * 8 + (size_t)256 * ppt192 * (A_DCH + 1) * 4 + (size_t)(256 * ppt192 * 3 + 1) * 4;
    lds_delta192 = fixed + (size_t)k * 193 * 8 <= 150 * 1024 ? 1 : 0;
    lds192 = fixed + (lds_delta192 ? (size_t)k * 193 * 8 : 0) + 16;
  }
  DevBuf quiet;
  TM_TRY(quiet.alloc(4));
  TM_HIP(hipMemsetAsync(quiet.p, 0xff, 4, stream));
  // D = 192, one segment, centroids that fit LDS: after H_WARM plain iterations the assignment step only touches the points whose
  // bounds do not prove their assignment (k_h_bounds, then k_assign192 over the list); the arithmetic, hence the result, is unchanged
  constexpr int h_warm = 5;
  const int h_kt = (k + KCH - 1) / KCH * KCH;  // row pitch of the transposed centroids (hcent_t)
  const bool skipping = d == 192 && nseg == 1 && k <= H_MAXK;
  DevBuf hub, hlb, hcent_t, hmove, hhalf, hneed, hcnt;
  const size_t l_lds = (size_t)192 * KCH * 8 + (size_t)k * 193 * 8 + (size_t)256 * 3 * 4 + 16;  // k_assign192_list
  if (skipping) {
    // a dynamic-LDS request above the CU's 160 KB comes back from the launch as a bare "invalid argument" (round 2's scratch records hold one,
    // from a k = 64 build of these kernels that kept more in LDS): refuse it here, by name
    TM_CHECK(l_lds <= 160 * 1024 && (size_t)k * 193 * 8 <= 160 * 1024, TM_E_INVAL, "k-means: %d centroids need %zu bytes of LDS in the list kernel (the CU has 163840)", k, l_lds);
    TM_TRY(hub.alloc((size_t)std::max<int64_t>(n, 1) * 8)); TM_TRY(hlb.alloc((size_t)std::max<int64_t>(n, 1) * 8)); TM_TRY(hcent_t.alloc((size_t)h_kt * 192 * 8));
    TM_TRY(hmove.alloc((size_t)(k + 3) * 8)); TM_TRY(hhalf.alloc((size_t)k * 8)); TM_TRY(hneed.alloc((size_t)std::max<int64_t>(n, 1) * 4)); TM_TRY(hcnt.alloc(8));
    TM_HIP(hipMemsetAsync(hcnt.p, 0, 8, stream));
    TM_HIP(hipMemsetAsync(hcent_t.p, 0, (size_t)h_kt * 192 * 8, stream));
    hipLaunchKernelGGL(k_cent_transpose, dim3(12), dim3(256), 0, stream, ds, cent, hcent_t.as<double>(), h_kt);
    if ((size_t)k * 193 * 8 > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_h_update), hipFuncAttributeMaxDynamicSharedMemorySize, k * 193 * 8);
    if (l_lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_assign192_list4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l_lds);
  }
  int it = 0, issued = 0;
  // D = 192, at most KCH centroids, points that fit the chip's LDS two rounds deep: the plain iterations, then ALL skipping iterations as one
  // resident launch (k_h_resident), a workgroup per CU.  Should a workgroup not become resident (another process holding CUs with a resident
  // launch of its own) the barrier gives up and the clustering is repeated from its seeds through the launches-per-iteration path below.
  bool resident_done = false;
  if (skipping && allow_resident && !knobs().km_launches && k <= KCH && max_iter > h_warm) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = (int)std::min<int64_t>(cus, (n + HR_NT - 1) / HR_NT);
    const int rounds = (int)(((n + grid - 1) / grid + HR_NT - 1) / HR_NT);  // a workgroup owns the points i with i % grid == its index: slots of 1024
    const size_t r_lds = (size_t)192 * HR_PITCH * 8 + (size_t)HR_E * 16 + (size_t)HR_P * 192 * 4 + (size_t)rounds * HR_NT * (4 + 4 + 2 + 2 + 4 + 1) + 16;
    if (rounds <= HR_MAXR && r_lds <= 160 * 1024 - 2048) {
      DevBuf hstate;
      TM_TRY(hstate.alloc(sizeof(HrState)));
      TM_HIP(hipMemsetAsync(hstate.p, 0, sizeof(HrState), stream));
      std::unique_lock<std::mutex> resident_lock(g_resident_launch);
      auto kres = rounds == 1 ? &k_h_resident<1> : &k_h_resident<2>;
      static_assert(HR_MAXR == 2, "one instantiation per number of rounds");
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kres), hipFuncAttributeMaxDynamicSharedMemorySize, (int)r_lds);
      for (; issued < h_warm; issued++) {
        const bool last_plain = issued == h_warm - 1;
        launch_assign192(ppt192, dim3(nblk192, nseg), lds192, stream, pts, ptsc.as<int32_t>(), n, w, ds, k, cent, assign, sums.as<u64>(), cnts.as<u64>(), rows192, lds_delta192,
                         quiet.as<int>(), last_plain ? hub.as<double>() : nullptr, last_plain ? hlb.as<double>() : nullptr);
        hipLaunchKernelGGL(k_h_update, dim3(1), dim3(1024), (size_t)k * 193 * 8, stream, ds, k, sums.as<u64>(), cnts.as<u64>(), cent, hcent_t.as<double>(), h_kt, hmove.as<double>(),
                           hhalf.as<double>(), hcnt.as<unsigned>(), issued, quiet.as<int>(), (int *)nullptr);
      }
      hipLaunchKernelGGL(kres, dim3(grid), dim3(HR_NT), r_lds, stream, pts, w, n, ds, k, cent, sums.as<u64>(), cnts.as<u64>(), hmove.as<double>(), hhalf.as<double>(), assign,
                         hub.as<double>(), hlb.as<double>(), hstate.as<HrState>(), h_warm, max_iter, quiet.as<int>());
      TM_HIP(hipGetLastError());
      int q = -1;
      HrState *hs_dev = hstate.as<HrState>();
      unsigned timed_out = 0;
#if TM_KMR_STAMPS
      std::vector<u64> stamps(HR_NSTAMP + 4);
      std::vector<unsigned> hlog(300 * 10);
#endif
      {
        HostRead hr_(stream);
        TM_TRY(hr_.get(&q, quiet.p, 4));
        TM_TRY(hr_.get(&timed_out, &hs_dev->timeout, 4));
#if TM_KMR_STAMPS
        TM_TRY(hr_.get(stamps.data(), hs_dev->stamps, stamps.size() * 8));
        TM_TRY(hr_.get(hlog.data(), hs_dev->log, hlog.size() * 4));
#endif
        TM_TRY(hr_.wait());
      }
      resident_lock.unlock();
      if (knobs().km_resident_fail) timed_out = 1;  // (tests: the path a barrier that gave up takes)
      if (timed_out) {
        fprintf(stderr, "[tm_kmeans] the resident tile k-means gave up at its barrier; repeating the clustering with one launch per step\n");
        return kmeans_batched(pts, w, d, seg_begin, seg_count, k, max_iter, assign, cent, host_kk, host_iters, stream, init_idx, dev_init_idx, false);
      }
      it = q >= 0 ? q : max_iter;
#if TM_KMR_STAMPS
      {
        const double ni = std::max(1, it - h_warm + (q >= 0 ? 1 : 0));
        static const char *names[8] = {"state in", "bounds pass", "own-centroid recheck", "scoring", "flush", "barrier", "deltas in", "update"};
        fprintf(stderr, "[tm_kmr stamps] %d workgroups x %d rounds, %d resident iterations; workgroup 0, s_memtime ticks per iteration:", grid, rounds, (int)ni);
        for (int i2 = 1; i2 < 8; i2++) fprintf(stderr, " %s %.0f,", names[i2], (double)stamps[i2] / ni);
        fprintf(stderr, " state in %.0f (once); per iteration %.1f points listed, %.1f scored (all workgroups)\n", (double)stamps[0], (double)stamps[HR_NSTAMP] / ni,
                (double)stamps[HR_NSTAMP + 1] / ni);
        for (int i2 = h_warm; i2 < std::min(it, 300); i2 += (i2 < 16 ? 1 : i2 < 64 ? 8 : 32))
          fprintf(stderr, "[tm_kmr log] iteration %3d: bounds %5u recheck %5u scoring %6u flush %5u barrier %6u deltas %5u update %5u | wg0 listed %4u scored %4u | moved (all) %u\n", i2, hlog[i2 * 10],
                  hlog[i2 * 10 + 1], hlog[i2 * 10 + 2], hlog[i2 * 10 + 3], hlog[i2 * 10 + 4], hlog[i2 * 10 + 5], hlog[i2 * 10 + 6], hlog[i2 * 10 + 7], hlog[i2 * 10 + 8], hlog[i2 * 10 + 9]);
      }
#endif
      resident_done = true;
    }
  }
  // Convergence is a flag on the device; the launches after it return at once.  The host queues the iterations in batches and reads the
  // flag of a batch while the NEXT batch runs (a copy into page-locked memory and an event behind every batch): the device never waits
  // for the host to look, and at most two short batches of launches are wasted at the end.  (Batches of 16 with the stream drained at
  // every poll left the device idle ~30 us six times per clustering and, on the bench clip, 42 no-op launches -- 0.45 ms -- behind the
  // 87th iteration.)
  // (the flag reaches the host in a page-locked word the update kernel itself writes: a 4-byte copy behind every batch was a launch of its own
  // -- 75 of them per clustering on the literal bench clip)
  int *pin = pinned_words();
  int *pin_dev = nullptr;
  if (pin && hipHostGetDevicePointer(reinterpret_cast<void **>(&pin_dev), pin, 0) != hipSuccess) { (void)hipGetLastError(); pin = nullptr; pin_dev = nullptr; }
  if (pin) __atomic_store_n(&pin[0], -1, __ATOMIC_RELEASE);
  const int poll_every = pin ? 4 : 16;
  hipEvent_t pev[2] = {nullptr, nullptr};
  struct EvGuard { hipEvent_t *e; ~EvGuard() { for (int i = 0; i < 2; i++) if (e[i]) (void)hipEventDestroy(e[i]); } } ev_guard{pev};
  if (pin) {
    TM_HIP(hipEventCreateWithFlags(&pev[0], hipEventDisableTiming));
    TM_HIP(hipEventCreateWithFlags(&pev[1], hipEventDisableTiming));
  }
  int nbatch = 0, qflag = -1;
  while (issued < max_iter && !resident_done) {
    const int batch = std::min(poll_every, max_iter - issued);
    for (int b = 0; b < batch; b++, issued++) {
      if (skipping) {
        const int gb = (int)((n + H_SLICE - 1) / H_SLICE);
        if (issued < h_warm) {
          const bool last_plain = issued == h_warm - 1;
          launch_assign192(ppt192, dim3(nblk192, nseg), lds192, stream, pts, ptsc.as<int32_t>(), n, w, ds, k, cent, assign, sums.as<u64>(), cnts.as<u64>(), rows192, lds_delta192,
                           quiet.as<int>(), last_plain ? hub.as<double>() : nullptr, last_plain ? hlb.as<double>() : nullptr);
        } else {
          hipLaunchKernelGGL(k_h_bounds, dim3(gb), dim3(256), 0, stream, pts, n, ds, (const double *)cent, assign, hub.as<double>(), hlb.as<double>(), hmove.as<double>(),
                             hhalf.as<double>(), k, hneed.as<int32_t>(), hcnt.as<unsigned>(), quiet.as<int>());
          hipLaunchKernelGGL(k_assign192_list4, dim3((unsigned)std::min<int64_t>((n + 63) / 64, k <= KCH ? 1024 : 768)), dim3(256), l_lds, stream, pts, ptsc.as<int32_t>(), n, w, ds, k, hcent_t.as<double>(), h_kt, assign, sums.as<u64>(),
                             cnts.as<u64>(), quiet.as<int>(), hub.as<double>(), hlb.as<double>(), hneed.as<int32_t>(), hcnt.as<unsigned>());
        }
        hipLaunchKernelGGL(k_h_update, dim3(1), dim3(1024), (size_t)k * 193 * 8, stream, ds, k, sums.as<u64>(), cnts.as<u64>(), cent, hcent_t.as<double>(), h_kt, hmove.as<double>(),
                           hhalf.as<double>(), hcnt.as<unsigned>(), issued, quiet.as<int>(), pin_dev);
        continue;
      }
      if (d == 3) {
        if (fuse3) {
          hipLaunchKernelGGL((k_assign<3, true>), dim3(nblk), dim3(256), lds_assign, stream, pts, w, ds, k, cent, assign, sums.as<u64>(), cnts.as<u64>(), quiet.as<int>());
        } else {
          hipLaunchKernelGGL((k_assign<3, false>), dim3(nblk), dim3(256), lds_assign, stream, pts, w, ds, k, cent, assign, sums.as<u64>(), cnts.as<u64>(), quiet.as<int>());
          hipLaunchKernelGGL(k_accumulate<3>, dim3(nblk), dim3(256), lds_acc, stream, pts, w, ds, k, assign, sums.as<u64>(), cnts.as<u64>(), quiet.as<int>());
        }
      } else {
        launch_assign192(ppt192, dim3(nblk192, nseg), lds192, stream, pts, ptsc.as<int32_t>(), n, w, ds, k, cent, assign, sums.as<u64>(), cnts.as<u64>(), rows192, lds_delta192, quiet.as<int>());
      }
      hipLaunchKernelGGL(k_update_all, dim3(1), dim3(1024), 0, stream, ds, nseg, k, d, (d == 192 || fuse3) ? 1 : 0, sums.as<u64>(), cnts.as<u64>(), cent, issued, quiet.as<int>(), pin_dev);
    }
    if (!pin) {
      int q = -1;
      {
        HostRead hr_(stream);
        TM_TRY(hr_.get(&q, quiet.p, 4));
        TM_TRY(hr_.wait());
      }
      if (q >= 0) { it = q; qflag = q; break; }
      it = issued;
      continue;
    }
    TM_HIP(hipEventRecord(pev[nbatch & 1], stream));
    nbatch++;
    it = issued;
    if (nbatch >= 2) {  // the batch before the one just queued
      TM_HIP(hipEventSynchronize(pev[nbatch & 1]));
      qflag = __atomic_load_n(&pin[0], __ATOMIC_ACQUIRE);  // (written once: -1 until an update finds the clustering quiet)
      if (qflag >= 0) { it = qflag; break; }
    }
  }
  if (pin && qflag < 0 && nbatch >= 1 && !resident_done) {  // the last batch queued
    TM_HIP(hipEventSynchronize(pev[(nbatch - 1) & 1]));
    qflag = __atomic_load_n(&pin[0], __ATOMIC_ACQUIRE);
    if (qflag >= 0) it = qflag;
  }
  TM_HIP(hipGetLastError());
  if (host_iters) *host_iters = it;
  {
    HostRead hr_(stream);
    TM_TRY(hr_.get(hs.data(), dsegs.p, sizeof(Seg) * nseg));
    TM_TRY(hr_.wait());
  }
  if (host_kk) {
    host_kk->resize(nseg);
    for (int s = 0; s < nseg; s++) (*host_kk)[s] = hs[s].kk;
  }
  return TM_OK;
}

int run_kmeans(const void *pts, const void *weights, int64_t n, int d, int k, int max_iter, void *assign, void *centroids, int *host_k,
               int *host_iters, hipStream_t stream) {
  return run_kmeans_seeded(pts, weights, n, d, k, nullptr, max_iter, assign, centroids, host_k, host_iters, stream);
}

int run_kmeans_seeded(const void *pts, const void *weights, int64_t n, int d, int k, const int64_t *host_init_idx, int max_iter, void *assign, void *centroids,
                      int *host_k, int *host_iters, hipStream_t stream) {
  TM_TRY(require_device());
  TM_CHECK(n >= 0, TM_E_INVAL, "kmeans: negative point count");
  if (host_k) *host_k = 0;
  if (host_iters) *host_iters = 0;
  if (n == 0) return TM_OK;
  std::vector<int64_t> b{0}, c{n};
  std::vector<int> kk;
  TM_TRY(kmeans_batched((const int32_t *)pts, (const uint32_t *)weights, d, b, c, k, max_iter, (int32_t *)assign, (double *)centroids,
                        &kk, host_iters, stream, host_init_idx));
  if (host_k) *host_k = kk[0];
  return TM_OK;
}

// ---- the build's seeding of the tile -> palette clustering: D^2 sampling, deterministic ---------------------------------------
// k-means++-style seeding measured 0.1-0.7 dB (mean 0.5) above farthest-first on the bench clip with as many or fewer final tiles
// (profiles/r02_seeding_experiment*.json); the build makes it reproducible: a 64-bit LCG (Knuth's MMIX constants) from PP_SEED, pick t draws
// r = floor(x_t * total / 2^64) over the exact integer masses q_i = weight_i * (squared distance of point i to its nearest centre so
// far) (q_i = weight_i for the first pick) in 128-bit sums, and takes the first point whose running sum exceeds r; a total of 0 (no
// point apart from the centres) ends the seeding.  The oracle states the same rule (tmo_kmeans_pp_seeds).
// (Measured in round 3 and dropped: skipping the rows the triangle inequality rules out -- D(own centre, new centre)^2 >= 4 md, exact in
// integers -- took 0.07 ms off the sixteen passes: the pass is not bound by the rows' bytes.)
// Per pick: k_pp_mass (distances to the newest centre folded into the running minimum, masses, one 128-bit sum per 512 points) and
// k_pp_pick (the block holding r, then the point inside it).
typedef unsigned __int128 u128;
constexpr u64 PP_SEED = 0x42381337ull, PP_MUL = 6364136223846793005ull, PP_INC = 1442695040888963407ull;
constexpr int PP_BLOCK = 512;   // points per workgroup of k_pp_mass = per partial sum
struct PpState { u64 rng; int kk, done; long long pick; u64 tot_lo, tot_hi; };
struct PpSum { u64 lo, hi; };
__device__ __forceinline__ u128 pp_mass(const uint32_t *w, const long long *mind, int64_t i, int first) {
  return (u128)(w ? w[i] : 1u) * (u128)(first ? 1ull : (u64)mind[i]);
}
__device__ __forceinline__ u128 pp_draw(u64 x, u128 total) {  // floor(x * total / 2^64) < total
  return (u128)x * (u64)(total >> 64) + (((u128)x * (u64)total) >> 64);
}
// A point over 16 lanes (three 16-byte pieces each: a wave's load covers four whole rows); the squared distance is a sum of integers
// mod 2^64, so the lanes' partial sums add up to the value the in-order loop gives.  A thread per point, each striding through its own
// 768-byte row, measured 74 us per launch at 64 k points (192 us with smaller workgroups: every line fetched eight times over).
constexpr int PP_NT = PP_BLOCK;  // (1024 / 1024 and 256 / 256 measured 0.3 and 0.1 ms per clip slower)
__global__ __launch_bounds__(PP_NT) void k_pp_mass(const int32_t *__restrict__ pts, const uint32_t *__restrict__ w, int64_t n, const int32_t *__restrict__ cur_row,
                                                   const PpState *__restrict__ st, int first, long long *__restrict__ mind, PpSum *__restrict__ bsum) {
  __shared__ PpSum s_part[PP_NT / 64];
  const int tid = threadIdx.x, l = tid & 15, grp = tid >> 4;
  const bool update = !first && !st->done;
  u128 mine = 0;
  if (update) {
    int4 c[3];
#pragma unroll
    for (int q = 0; q < 3; q++) c[q] = reinterpret_cast<const int4 *>(cur_row)[l + 16 * q];
#pragma unroll 4
    for (int m = 0; m < PP_BLOCK / (PP_NT / 16); m++) {
      const int64_t i = (int64_t)blockIdx.x * PP_BLOCK + m * (PP_NT / 16) + grp;
      const bool valid = i < n;
      const int4 *p = reinterpret_cast<const int4 *>(pts + (valid ? i : 0) * 192);
      u64 dd = 0;
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int4 v = p[l + 16 * q];
        const long long t0 = (long long)v.x - c[q].x, t1 = (long long)v.y - c[q].y, t2 = (long long)v.z - c[q].z, t3 = (long long)v.w - c[q].w;
        dd += (u64)(t0 * t0) + (u64)(t1 * t1) + (u64)(t2 * t2) + (u64)(t3 * t3);
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) dd += __shfl_xor(dd, o);
      if (valid && l == 0) {
        long long md = mind[i];
        if ((long long)dd < md) { md = (long long)dd; mind[i] = md; }
        mine += (u128)(w ? w[i] : 1u) * (u128)(u64)md;
      }
    }
  } else {
    for (int m = 0; m < PP_BLOCK / PP_NT; m++) {
      const int64_t i = (int64_t)blockIdx.x * PP_BLOCK + m * PP_NT + tid;
      if (i < n) mine += pp_mass(w, mind, i, first);
    }
  }
  u64 lo = (u64)mine, hi = (u64)(mine >> 64);
  for (int o = 32; o > 0; o >>= 1) {
    const u128 other = ((u128)__shfl_xor(hi, o) << 64) | __shfl_xor(lo, o);
    const u128 sum = (((u128)hi << 64) | lo) + other;
    lo = (u64)sum; hi = (u64)(sum >> 64);
  }
  if ((tid & 63) == 0) s_part[tid >> 6] = PpSum{lo, hi};
  __syncthreads();
  if (tid == 0) {
    u128 t = 0;
    for (int wv = 0; wv < PP_NT / 64; wv++) t += ((u128)s_part[wv].hi << 64) | s_part[wv].lo;
    bsum[blockIdx.x] = PpSum{(u64)t, (u64)(t >> 64)};
  }
}
// inclusive prefix (128-bit) of one value per thread over a workgroup of 256, in thread order: shuffles inside the waves, the four wave
// totals through LDS; *total = the sum of all.  Every thread of the workgroup calls it.
__device__ __forceinline__ u128 pp_scan256(u128 v, PpSum *s_w, u128 *total) {
  u64 lo = (u64)v, hi = (u64)(v >> 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u64 olo = __shfl_up(lo, o), ohi = __shfl_up(hi, o);
    if (lane >= o) {
      const u128 s = (((u128)hi << 64) | lo) + (((u128)ohi << 64) | olo);
      lo = (u64)s; hi = (u64)(s >> 64);
    }
  }
  if (lane == 63) s_w[wave] = PpSum{lo, hi};
  __syncthreads();
  u128 before = 0, tot = 0;
#pragma unroll
  for (int wv = 0; wv < 4; wv++) {
    const u128 t = ((u128)s_w[wv].hi << 64) | s_w[wv].lo;
    if (wv < wave) before += t;
    tot += t;
  }
  *total = tot;
  __syncthreads();  // (s_w may be written again)
  return before + (((u128)hi << 64) | lo);
}
// the first thread (in thread order) whose flag is set, 256 if none: a ballot per wave, the four answers through LDS
__device__ __forceinline__ int pp_first256(bool flag, int *s_f) {
  const unsigned long long b = __builtin_amdgcn_ballot_w64(flag);
  if ((threadIdx.x & 63) == 0) s_f[threadIdx.x >> 6] = b ? (int)(threadIdx.x & ~63u) + __builtin_ctzll(b) : 256;
  __syncthreads();
  const int f = min(min(s_f[0], s_f[1]), min(s_f[2], s_f[3]));
  __syncthreads();
  return f;
}

// One workgroup.  mode 0: the whole pick (single process): total, draw, block, point -> st->pick, cur_row, cent, seeds.
// mode 1 (several processes): only this process's total -> st->tot_*.  mode 2: the draw against the totals of all processes (rank
// order = global point order); the owner of r finds the point, everybody else reports no candidate.
// "The first point whose running sum exceeds r" is found with prefix sums over the workgroup instead of one thread walking 2 x 256
// partial sums (24 -> 9 us per pick: sixteen picks wait for it one after the other).
__global__ __launch_bounds__(256) void k_pp_pick(const int32_t *__restrict__ pts, const uint32_t *__restrict__ w, int64_t n, const long long *__restrict__ mind,
                                                 const PpSum *__restrict__ bsum, int nb, int first, int k, PpState *__restrict__ st, int mode,
                                                 const PpSum *__restrict__ totals, int rank, int world, long long global_begin,
                                                 int32_t *__restrict__ cur_row, double *__restrict__ cent, long long *__restrict__ seeds, FfCandOut cand) {
  __shared__ PpSum s_w[4];
  __shared__ int s_f[4];
  __shared__ long long s_pick;
  __shared__ u64 s_r[2];
  const int tid = threadIdx.x;
  if (st->done || st->kk >= k) {
    if (mode == 2 && cand.dist && tid == 0) { *cand.dist = -1; *cand.gidx = 0x7fffffffffffffffll; }
    return;
  }
  // the blocks' sums: thread t adds its share (consecutive blocks); their prefix over the workgroup
  const int per = (nb + 255) / 256;
  u128 part = 0;
  for (int b = tid * per; b < min(nb, (tid + 1) * per); b++) part += ((u128)bsum[b].hi << 64) | bsum[b].lo;
  u128 local = 0;
  const u128 incl = pp_scan256(part, s_w, &local);
  if (tid == 0) {
    s_pick = -1;  // -1: the point is another process's, -2: nothing left to pick, -3: ours
    if (mode == 1) { st->tot_lo = (u64)local; st->tot_hi = (u64)(local >> 64); }
    else {
      u128 total = local, before = 0;
      if (mode == 2) {
        total = 0;
        for (int r = 0; r < world; r++) {
          const u128 t = ((u128)totals[r].hi << 64) | totals[r].lo;
          if (r < rank) before += t;
          total += t;
        }
      }
      if (total == 0) { st->done = 1; s_pick = -2; }
      else {
        const u64 x = st->rng * PP_MUL + PP_INC;
        st->rng = x;
        const u128 r = pp_draw(x, total);
        if (r >= before && r < before + local) {  // the point is one of ours
          const u128 rl = r - before;
          s_r[0] = (u64)rl; s_r[1] = (u64)(rl >> 64);
          s_pick = -3;
        }
      }
    }
  }
  __syncthreads();
  if (mode == 1) return;
  long long blk = s_pick;
  if (blk == -2) {
    if (mode == 2 && cand.dist && tid == 0) { *cand.dist = -1; *cand.gidx = 0x7fffffffffffffffll; }
    return;
  }
  if (blk == -3) {  // (uniform) which thread's share of blocks, which block of it, which point of the block
    const u128 rl = ((u128)s_r[1] << 64) | s_r[0];
    const int ts = pp_first256(incl > rl, s_f);  // rl < local: there is one
    if (tid == ts) {
      u128 run = incl - part;
      int b = tid * per;
      for (; b < min(nb, (tid + 1) * per) - 1; b++) {
        const u128 v = ((u128)bsum[b].hi << 64) | bsum[b].lo;
        if (run + v > rl) break;
        run += v;
      }
      s_pick = b;
      const u128 rest = rl - run;
      s_r[0] = (u64)rest; s_r[1] = (u64)(rest >> 64);
    }
    __syncthreads();
    blk = s_pick;
    const u128 rest = ((u128)s_r[1] << 64) | s_r[0];
    // masses of the block's points in index order: thread t holds PP_BLOCK / 256 consecutive ones
    constexpr int E = PP_BLOCK / 256;
    u128 q[E], qs = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int64_t i = blk * PP_BLOCK + tid * E + e;
      q[e] = i < n ? pp_mass(w, mind, i, first) : (u128)0;
      qs += q[e];
    }
    u128 unused;
    const u128 qincl = pp_scan256(qs, s_w, &unused);
    const int tq = pp_first256(qincl > rest, s_f);
    if (tid == tq) {
      u128 run = qincl - qs;
      long long pick = -1;
#pragma unroll
      for (int e = 0; e < E; e++) {
        run += q[e];
        if (pick < 0 && run > rest) pick = blk * PP_BLOCK + tid * E + e;
      }
      s_pick = pick;
    }
    if (tq >= 256 && tid == 0) s_pick = -1;  // (cannot happen: rest < the block's sum)
    __syncthreads();
  }
  const long long pick = blk >= 0 ? s_pick : -1;  // (blk = -1: another process's)
  if (mode == 0) {
    if (pick >= 0) {
      const int kk = st->kk;
      for (int j = tid; j < 192; j += 256) { const int32_t v = pts[pick * 192 + j]; cur_row[j] = v; cent[(int64_t)kk * 192 + j] = (double)v; }
      __syncthreads();
      if (tid == 0) { seeds[kk] = pick; st->pick = pick; st->kk = kk + 1; }
    }
  } else {  // this process's candidate for the all-gather: the row of the picked point, or none
    if (tid == 0) { *cand.dist = pick >= 0 ? 1 : -1; *cand.gidx = pick >= 0 ? global_begin + pick : 0x7fffffffffffffffll; }
    if (tid < 192) cand.row[tid] = pick >= 0 ? pts[pick * 192 + tid] : 0;
  }
}

// ---- DoPalettization -------------------------------------------------------------------------------------------
__global__ void k_count_assign(const int32_t *__restrict__ assign, int64_t n, int k, u64 *__restrict__ cnt) {
  extern __shared__ unsigned int s_cnt[];  // per-workgroup histogram when k fits (a handful of hot global counters would serialise)
  const bool use_lds = k <= 8192;
  if (use_lds) {
    for (int e = threadIdx.x; e < k; e += blockDim.x) s_cnt[e] = 0;
    __syncthreads();
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (use_lds) atomicAdd(&s_cnt[assign[i]], 1u);
    else atomicAdd(&cnt[assign[i]], 1ull);
  }
  if (use_lds) {
    __syncthreads();
    for (int e = threadIdx.x; e < k; e += blockDim.x)
      if (s_cnt[e]) atomicAdd(&cnt[e], (u64)s_cnt[e]);
  }
}
__global__ void k_apply_lut(const int32_t *__restrict__ assign, int64_t n, const int32_t *__restrict__ lut, int32_t *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = lut[assign[i]];
}

// the k seed points of one process's whole point set (indices, -1 beyond the centres found)
// out (host) and / or dev_out (the device buffer itself: k indices, -1 beyond the centres found); nothing is read back unless `out` is asked for
static int pp_seeds(const int32_t *pts, const uint32_t *w, int64_t n, int k, std::vector<int64_t> *out, hipStream_t stream, DevBuf *dev_out = nullptr) {
  DevBuf mind, bsum, state, cur_row, cent, seeds;
  const int nb = (int)((n + PP_BLOCK - 1) / PP_BLOCK);
  TM_TRY(mind.alloc((size_t)n * 8)); TM_TRY(bsum.alloc(sizeof(PpSum) * (size_t)nb)); TM_TRY(state.alloc(sizeof(PpState)));
  TM_TRY(cur_row.alloc(192 * 4)); TM_TRY(cent.alloc((size_t)k * 192 * 8)); TM_TRY(seeds.alloc((size_t)k * 8));
  TM_HIP(hipMemsetAsync(mind.p, 0x7f, (size_t)n * 8, stream));
  TM_HIP(hipMemsetAsync(seeds.p, 0xff, (size_t)k * 8, stream));
  PpState h0;
  memset(&h0, 0, sizeof(h0));
  h0.rng = PP_SEED;
  TM_HIP(hipMemcpyAsync(state.p, &h0, sizeof(h0), hipMemcpyHostToDevice, stream));
  for (int c = 0; c < k; c++) {
    hipLaunchKernelGGL(k_pp_mass, dim3(nb), dim3(PP_NT), 0, stream, pts, w, n, cur_row.as<int32_t>(), state.as<PpState>(), c == 0 ? 1 : 0, mind.as<long long>(), bsum.as<PpSum>());
    hipLaunchKernelGGL(k_pp_pick, dim3(1), dim3(256), 0, stream, pts, w, n, mind.as<long long>(), bsum.as<PpSum>(), nb, c == 0 ? 1 : 0, k, state.as<PpState>(), 0,
                       (const PpSum *)nullptr, 0, 1, 0ll, cur_row.as<int32_t>(), cent.as<double>(), seeds.as<long long>(), FfCandOut{nullptr, nullptr, nullptr});
  }
  TM_HIP(hipGetLastError());
  if (out) {
    out->assign((size_t)k, -1);
    HostRead hr_(stream);
    TM_TRY(hr_.get(out->data(), seeds.p, (size_t)k * 8));
    TM_TRY(hr_.wait());
  }
  if (dev_out) *dev_out = std::move(seeds);  // (the other buffers go back to the pool: what is queued on this stream after them is ordered behind their last use)
  return TM_OK;
}

KmeansRunStats &kmeans_run_stats() {
  static thread_local KmeansRunStats st;
  return st;
}

// would run_palettize put the clustering through the resident launch?  (one process per GPU: then every process clusters ALL global tiles
// itself -- 6 ms, no collective -- instead of a share of them with an all-reduce per Lloyd iteration)
bool palettize_resident(int64_t n, int npal) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return !knobs().km_launches && npal <= KCH && n > 0 && n <= (int64_t)HR_MAXR * HR_NT * cus;
}

int run_palettize(const void *feat, const void *use, int64_t n, int npal, int max_iter, void *out_pal_idx, hipStream_t stream) {
  TM_TRY(require_device());
  TM_CHECK(npal >= 1 && npal <= 65536, TM_E_INVAL, "PaletteCount %d outside 1..65536 (tilingencoder.pas:2959)", npal);
  if (n <= 0) return TM_OK;
  DevBuf assign, cent, cnt, lut;
  TM_TRY(assign.alloc(n * 4));
  TM_TRY(cent.alloc((size_t)npal * 192 * 8));
  TM_TRY(cnt.alloc((size_t)npal * 8));
  TM_TRY(lut.alloc((size_t)npal * 4));
  int iters = 0;
  {
    DevBuf dseeds;  // the seeds never leave the device: no read-back, no drain of the stream, no upload between the seeding and the iterations
    TM_TRY(pp_seeds((const int32_t *)feat, (const uint32_t *)use, n, npal, nullptr, stream, &dseeds));
    std::vector<int64_t> b{0}, c{n};
    std::vector<int> kks;
    TM_TRY(kmeans_batched((const int32_t *)feat, (const uint32_t *)use, 192, b, c, npal, max_iter, assign.as<int32_t>(), cent.as<double>(), &kks, &iters, stream, nullptr,
                          dseeds.as<long long>()));
  }
  kmeans_run_stats().tile_iters = iters;
  kmeans_run_stats().tile_points = n;
  // palettes ranked by number of tiles, descending (tilingencoder.pas:4229-4234); ties keep the initial order
  TM_HIP(hipMemsetAsync(cnt.p, 0, (size_t)npal * 8, stream));
  hipLaunchKernelGGL(k_count_assign, dim3((int)std::min<int64_t>((n + 255) / 256, 512)), dim3(256), npal <= 8192 ? (size_t)npal * 4 : 0, stream,
                     assign.as<int32_t>(), n, npal, cnt.as<u64>());
  std::vector<u64> hc(npal);
  {
    HostRead hr_(stream);
    TM_TRY(hr_.get(hc.data(), cnt.p, (size_t)npal * 8));
    TM_TRY(hr_.wait());
  }
  std::vector<int> ord(npal), hl(npal);
  for (int i = 0; i < npal; i++) ord[i] = i;
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return hc[a] > hc[b]; });
  for (int i = 0; i < npal; i++) hl[ord[i]] = i;
  TM_HIP(hipMemcpyAsync(lut.p, hl.data(), (size_t)npal * 4, hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(k_apply_lut, dim3((int)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, stream, assign.as<int32_t>(), n,
                     lut.as<int32_t>(), (int32_t *)out_pal_idx);
  TM_HIP(hipGetLastError());
  TM_HIP(hipStreamSynchronize(stream));
  return TM_OK;
}

// ---- DoPalettization over several processes ---------------------------------------------------------------------
struct FfCand { long long dist, gidx; int32_t row[192]; };  // one farthest-first candidate per process: largest min-distance, then lowest global index
struct FfState { int kk, done; };

// every process makes the same choice among the gathered candidates
__global__ __launch_bounds__(256) void k_ffd_pick(const FfCand *__restrict__ cands, int world, int k, FfState *__restrict__ st, int32_t *__restrict__ cur_row,
                                                  double *__restrict__ cent) {
  __shared__ int s_win;
  if (threadIdx.x == 0) {
    int win = -1;
    if (!st->done && st->kk < k)
      for (int r = 0; r < world; r++)
        if (cands[r].dist > 0 && (win < 0 || cands[r].dist > cands[win].dist || (cands[r].dist == cands[win].dist && cands[r].gidx < cands[win].gidx))) win = r;
    s_win = win;
    if (win < 0) st->done = 1;
  }
  __syncthreads();
  const int win = s_win;
  if (win < 0) return;
  const int kk = st->kk;
  if (threadIdx.x < 192) {
    cur_row[threadIdx.x] = cands[win].row[threadIdx.x];
    cent[(int64_t)kk * 192 + threadIdx.x] = (double)cands[win].row[threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x == 0) st->kk = kk + 1;
}
__global__ void k_kmd_pack(const u64 *__restrict__ sums, const u64 *__restrict__ cnts, const Seg *__restrict__ segs, int k, u64 *__restrict__ red) {
  const int total = k * 192 + k + 1;  // sums | counts | number of points that changed cluster
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x)
    red[e] = e < k * 192 ? sums[e] : e < k * 192 + k ? cnts[e - k * 192] : (u64)(long long)segs[0].changed;
}
__global__ __launch_bounds__(1024) void k_kmd_update(const u64 *__restrict__ red, Seg *__restrict__ segs, int k, double *__restrict__ cent) {
  const bool changed = red[k * 192 + k] != 0;
  if (changed)
    for (int e = threadIdx.x; e < k * 192; e += 1024) {
      const u64 cn = red[k * 192 + e / 192];
      if (cn > 0) cent[e] = __ddiv_rn((double)(long long)red[e], (double)(long long)cn);
    }
  if (threadIdx.x == 0) segs[0].changed = 0;
}

int run_palettize_dist(const void *feat_local, const void *use_local, int64_t n, int64_t global_begin, int npal, int max_iter, void *out_pal_idx_local,
                       const Collectives &co, hipStream_t stream) {
  TM_TRY(require_device());
  TM_CHECK(npal >= 1 && npal <= 65536, TM_E_INVAL, "PaletteCount %d outside 1..65536 (tilingencoder.pas:2959)", npal);
  TM_CHECK(co.world >= 1 && co.allgather && co.allreduce_sum_i64, TM_E_INVAL, "palettize: collectives missing");
  const int k = npal, d = 192;
  const int32_t *pts = (const int32_t *)feat_local;
  const uint32_t *w = (const uint32_t *)use_local;
  const int64_t n1 = std::max<int64_t>(n, 1);
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  DevBuf dsegs, mind, partial, sums, cnts, cent, assign, cur_row, cand, cands, state, red, ptsc, quiet;
  Seg hs;
  memset(&hs, 0, sizeof(hs));
  hs.begin = 0; hs.count = n; hs.nseg = 1; hs.blk_first = 0; hs.blk_count = 1;
  const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1024));
  TM_TRY(dsegs.alloc(sizeof(Seg))); TM_TRY(mind.alloc((size_t)n1 * 8)); TM_TRY(partial.alloc(sizeof(BestKey) * (size_t)nblk));
  TM_TRY(sums.alloc((size_t)k * d * 8)); TM_TRY(cnts.alloc((size_t)k * 8)); TM_TRY(cent.alloc((size_t)k * d * 8)); TM_TRY(assign.alloc((size_t)n1 * 4));
  TM_TRY(cur_row.alloc(192 * 4)); TM_TRY(cand.alloc(sizeof(FfCand))); TM_TRY(cands.alloc(sizeof(FfCand) * (size_t)co.world)); TM_TRY(state.alloc(sizeof(FfState)));
  TM_TRY(red.alloc((size_t)(k * d + k + 1) * 8)); TM_TRY(quiet.alloc(4));
  TM_HIP(hipMemsetAsync(mind.p, 0x7f, (size_t)n1 * 8, stream));
  TM_HIP(hipMemsetAsync(sums.p, 0, (size_t)k * d * 8, stream));
  TM_HIP(hipMemsetAsync(cnts.p, 0, (size_t)k * 8, stream));
  TM_HIP(hipMemsetAsync(cent.p, 0, (size_t)k * d * 8, stream));
  TM_HIP(hipMemsetAsync(assign.p, 0xff, (size_t)n1 * 4, stream));
  TM_HIP(hipMemsetAsync(state.p, 0, sizeof(FfState), stream));
  TM_HIP(hipMemsetAsync(quiet.p, 0xff, 4, stream));
  // D^2 seeding over all processes (k_pp_mass / k_pp_pick): every process keeps the same generator state; per pick the processes'
  // total masses are all-gathered (rank order = global point order), the owner of the draw finds the point, and the candidates
  // (one real, the others empty) are all-gathered like the farthest-first ones were
  FfState hst{0, 0};
  DevBuf ppstate, bsum, totals, mytot, ppseeds;
  const int nb = (int)std::max<int64_t>(1, (n + PP_BLOCK - 1) / PP_BLOCK);
  TM_TRY(ppstate.alloc(sizeof(PpState))); TM_TRY(bsum.alloc(sizeof(PpSum) * (size_t)nb)); TM_TRY(totals.alloc(sizeof(PpSum) * (size_t)co.world));
  TM_TRY(mytot.alloc(sizeof(PpSum))); TM_TRY(ppseeds.alloc((size_t)k * 8));
  {
    PpState h0;
    memset(&h0, 0, sizeof(h0));
    h0.rng = PP_SEED;
    TM_HIP(hipMemcpyAsync(ppstate.p, &h0, sizeof(h0), hipMemcpyHostToDevice, stream));
    TM_HIP(hipMemsetAsync(bsum.p, 0, sizeof(PpSum) * (size_t)nb, stream));
    TM_HIP(hipStreamSynchronize(stream));  // h0 is on the stack
  }
  FfCand *cd = cand.as<FfCand>();
  for (int c = 0; c < k; c++) {
    {
      const int first = c == 0 ? 1 : 0;
      if (n > 0)
        hipLaunchKernelGGL(k_pp_mass, dim3(nb), dim3(PP_NT), 0, stream, pts, w, n, cur_row.as<int32_t>(), ppstate.as<PpState>(), first, mind.as<long long>(), bsum.as<PpSum>());
      hipLaunchKernelGGL(k_pp_pick, dim3(1), dim3(256), 0, stream, pts, w, n, mind.as<long long>(), bsum.as<PpSum>(), n > 0 ? nb : 0, first, k, ppstate.as<PpState>(), 1,
                         (const PpSum *)nullptr, co.rank, co.world, (long long)global_begin, cur_row.as<int32_t>(), cent.as<double>(), ppseeds.as<long long>(),
                         FfCandOut{nullptr, nullptr, nullptr});
      TM_HIP(hipGetLastError());
      // tot_lo, tot_hi sit side by side in PpState: 16 bytes per process
      TM_TRY(co.allgather(reinterpret_cast<uint8_t *>(ppstate.p) + offsetof(PpState, tot_lo), totals.p, (int64_t)sizeof(PpSum)));
      hipLaunchKernelGGL(k_pp_pick, dim3(1), dim3(256), 0, stream, pts, w, n, mind.as<long long>(), bsum.as<PpSum>(), n > 0 ? nb : 0, first, k, ppstate.as<PpState>(), 2,
                         totals.as<PpSum>(), co.rank, co.world, (long long)global_begin, cur_row.as<int32_t>(), cent.as<double>(), ppseeds.as<long long>(),
                         FfCandOut{&cd->dist, &cd->gidx, cd->row});
    }
    TM_HIP(hipGetLastError());
    TM_TRY(co.allgather(cand.p, cands.p, (int64_t)sizeof(FfCand)));
    hipLaunchKernelGGL(k_ffd_pick, dim3(1), dim3(256), 0, stream, cands.as<FfCand>(), co.world, k, state.as<FfState>(), cur_row.as<int32_t>(), cent.as<double>());
    TM_HIP(hipGetLastError());
    if ((c & 3) == 3 || c == k - 1) {  // "no distinct point left" ends the picks early; looked at every few picks
      {
        HostRead hr_(stream);
        TM_TRY(hr_.get(&hst, state.p, sizeof(FfState)));
        TM_TRY(hr_.wait());
      }
      if (hst.done) break;
    }
  }
  {
    HostRead hr_(stream);
    TM_TRY(hr_.get(&hst, state.p, sizeof(FfState)));
    TM_TRY(hr_.wait());
  }
  hs.kk = hst.kk;
  hs.init_done = 1;
  TM_CHECK(hs.kk >= 1, TM_E_INVAL, "palettize: no point anywhere");
  TM_HIP(hipMemcpyAsync(dsegs.p, &hs, sizeof(Seg), hipMemcpyHostToDevice, stream));
  // Lloyd: local assignment (the sums of this process's points are carried with +/- deltas), all-reduce, identical update everywhere
  TM_TRY(ptsc.alloc((size_t)n1 * 192 * 4));
  if (n > 0) hipLaunchKernelGGL(k_chunk_major, dim3((unsigned)std::min<int64_t>((n * 48 + 255) / 256, 8192)), dim3(256), 0, stream, pts, n, ptsc.as<int32_t>());
  int ppt192 = 1, nblk192 = 1, rows192 = 256, lds_delta192 = 0;
  size_t lds192 = 0;
  {
    const int64_t slots = std::max<int64_t>(1, (int64_t)cus * 2);
    const int64_t per_slot = (n1 + slots - 1) / slots;
    const int64_t rounds = (per_slot + 256 * 5 - 1) / (256 * 5);
    rows192 = (int)std::max<int64_t>(1, (per_slot + rounds - 1) / rounds);
    ppt192 = (rows192 + 255) / 256;
    nblk192 = (int)((n1 + rows192 - 1) / rows192);
    const size_t fixed = (size_t)A_DCH * KCH * 8 + (size_t)256 * ppt192 * (A_DCH + 1) * 4 + (size_t)(256 * ppt192 * 3 + 1) * 4;
    lds_delta192 = fixed + (size_t)k * 193 * 8 <= 150 * 1024 ? 1 : 0;
    lds192 = fixed + (lds_delta192 ? (size_t)k * 193 * 8 : 0) + 16;
  }
  for (int it = 0; it < max_iter; it++) {
    if (n > 0)
      launch_assign192(ppt192, dim3(nblk192, 1), lds192, stream, pts, ptsc.as<int32_t>(), n, w, dsegs.as<Seg>(), k, cent.as<double>(), assign.as<int32_t>(),
                       sums.as<u64>(), cnts.as<u64>(), rows192, lds_delta192, quiet.as<int>());
    hipLaunchKernelGGL(k_kmd_pack, dim3(32), dim3(256), 0, stream, sums.as<u64>(), cnts.as<u64>(), dsegs.as<Seg>(), k, red.as<u64>());
    TM_HIP(hipGetLastError());
    TM_TRY(co.allreduce_sum_i64(red.p, (int64_t)k * d + k + 1));
    hipLaunchKernelGGL(k_kmd_update, dim3(1), dim3(1024), 0, stream, red.as<u64>(), dsegs.as<Seg>(), k, cent.as<double>());
    u64 changed = 0;
    {
      HostRead hr_(stream);
      TM_TRY(hr_.get(&changed, red.as<u64>() + (size_t)k * d + k, 8));
      TM_TRY(hr_.wait());
    }
    if (changed == 0) break;
  }
  // palettes ranked by number of tiles over all processes, descending (tilingencoder.pas:4229-4234); ties keep the initial order
  DevBuf cnt, lut;
  TM_TRY(cnt.alloc((size_t)npal * 8)); TM_TRY(lut.alloc((size_t)npal * 4));
  TM_HIP(hipMemsetAsync(cnt.p, 0, (size_t)npal * 8, stream));
  if (n > 0)
    hipLaunchKernelGGL(k_count_assign, dim3((int)std::min<int64_t>((n + 255) / 256, 512)), dim3(256), npal <= 8192 ? (size_t)npal * 4 : 0, stream,
                       assign.as<int32_t>(), n, npal, cnt.as<u64>());
  TM_HIP(hipGetLastError());
  TM_TRY(co.allreduce_sum_i64(cnt.p, npal));
  std::vector<u64> hc(npal);
  {
    HostRead hr_(stream);
    TM_TRY(hr_.get(hc.data(), cnt.p, (size_t)npal * 8));
    TM_TRY(hr_.wait());
  }
  std::vector<int> ord(npal), hl(npal);
  for (int i = 0; i < npal; i++) ord[i] = i;
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return hc[a] > hc[b]; });
  for (int i = 0; i < npal; i++) hl[ord[i]] = i;
  TM_HIP(hipMemcpyAsync(lut.p, hl.data(), (size_t)npal * 4, hipMemcpyHostToDevice, stream));
  if (n > 0)
    hipLaunchKernelGGL(k_apply_lut, dim3((int)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, stream, assign.as<int32_t>(), n, lut.as<int32_t>(),
                       (int32_t *)out_pal_idx_local);
  TM_HIP(hipGetLastError());
  TM_HIP(hipStreamSynchronize(stream));
  return TM_OK;
}

// ---- QuantizeUsingYakmo + DoQuantization -----------------------------------------------------------------------
// pixel key = palette << 24 | G << 16 | R << 8 | B  (CompareDSPixel: G, then R, then B; tilingencoder.pas:1046-1056)
__global__ void k_pixel_keys(const uint32_t *__restrict__ tiles, const int32_t *__restrict__ pal_idx, int64_t n, u64 *__restrict__ keys) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * 64; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t c = tiles[i];
    const u64 p = (u64)(uint32_t)pal_idx[i >> 6];
    keys[i] = (p << 24) | ((u64)((c >> 8) & 0xff) << 16) | ((u64)(c & 0xff) << 8) | (u64)((c >> 16) & 0xff);
  }
}
__global__ void k_palette_bounds(const u64 *__restrict__ ukeys, int64_t nu, int npal, long long *__restrict__ lb) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;  // lb[p] = first unique key whose palette field is >= p; lb[npal] = first >= npal
  if (p > npal) return;
  int64_t lo = 0, hi = nu;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((long long)(ukeys[mid] >> 24) < (long long)p) lo = mid + 1; else hi = mid;
  }
  lb[p] = lo;
}
__global__ void k_unpack_colours(const u64 *__restrict__ ukeys, int64_t nu, int32_t *__restrict__ pts) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nu; i += (int64_t)gridDim.x * blockDim.x) {
    const u64 kx = ukeys[i];
    pts[i * 3 + 0] = (int32_t)((kx >> 8) & 0xff);   // R
    pts[i * 3 + 1] = (int32_t)((kx >> 16) & 0xff);  // G
    pts[i * 3 + 2] = (int32_t)(kx & 0xff);          // B
  }
}

static int muldiv_win(int a, int b, int c) {  // Windows MulDiv: round half away from zero
  long long p = (long long)a * b, q = p >= 0 ? p : -p, cc = c >= 0 ? c : -c;
  long long r = (q + cc / 2) / cc;
  return (int)(((p < 0) != (c < 0)) ? -r : r);
}
static void rgb_to_hsv_bytes(int rr, int gg, int bb, int &h, int &s, int &v) {  // RGBToHSV, utils.pas:278-325
  int mx = std::max(rr, std::max(gg, bb)), mn = std::min(rr, std::min(gg, bb));
  int hh = 0, ss = 0, ll = mx;
  if (ll != mn) {
    const int delta = ll - mn;
    ss = muldiv_win(delta, 255, ll);
    if (rr == ll) hh = muldiv_win(42, gg - bb, delta);
    else if (gg == ll) hh = muldiv_win(42, bb - rr, delta) + 84;
    else if (bb == ll) hh = muldiv_win(42, rr - gg, delta) + 168;
    hh = hh % 252;
  }
  h = hh & 0xff; s = ss & 0xff; v = ll & 0xff;
}

int run_quantize_palettes(const void *tiles, const void *pal_idx, int64_t n, int npal, int pal_size, int max_iter, void *out_palettes,
                          hipStream_t stream, DevBuf *keep_keys, int64_t *keep_n) {
  return run_quantize_palettes_part(tiles, pal_idx, n, npal, pal_size, max_iter, out_palettes, 0, 1, stream, keep_keys, keep_n);
}

int run_quantize_palettes_part(const void *tiles, const void *pal_idx, int64_t n, int npal, int pal_size, int max_iter, void *out_palettes,
                               int pal_rank, int pal_world, hipStream_t stream, DevBuf *keep_keys, int64_t *keep_n) {
  TM_TRY(require_device());
  if (keep_n) *keep_n = 0;
  TM_CHECK(pal_world >= 1 && pal_rank >= 0 && pal_rank < pal_world, TM_E_INVAL, "quantize: bad palette share %d of %d", pal_rank, pal_world);
  TM_CHECK(npal >= 1 && npal <= 65536, TM_E_INVAL, "PaletteCount %d outside 1..65536", npal);
  TM_CHECK(pal_size >= 2 && pal_size <= 64, TM_E_INVAL, "PaletteSize %d outside 2..64 (tilingencoder.pas:2965)", pal_size);
  std::vector<int32_t> hpal((size_t)npal * pal_size, TM_NULL_COLOR);  // unused slots: cDitheringNullColor (4557-4558)
  if (n > 0) {
    const int64_t npx = n * 64;
    DevBuf keys, keys2, ukeys, ucnt, nruns, tmp, pts, assign, cent;
    TM_TRY(keys.alloc(npx * 8)); TM_TRY(keys2.alloc(npx * 8)); TM_TRY(ukeys.alloc(npx * 8)); TM_TRY(ucnt.alloc(npx * 4));
    TM_TRY(nruns.alloc(8));
    hipLaunchKernelGGL(k_pixel_keys, dim3((int)std::min<int64_t>((npx + 255) / 256, 4096)), dim3(256), 0, stream, (const uint32_t *)tiles,
                       (const int32_t *)pal_idx, n, keys.as<u64>());
    size_t tb = 0;
    int key_bits = 25;  // 24 bits of colour + the palette number's: every 8 bits less is a pass over all pixels less
    while (key_bits < 41 && (1ll << (key_bits - 24)) < npal) key_bits++;
    TM_HIP(rocprim::radix_sort_keys(nullptr, tb, keys.as<u64>(), keys2.as<u64>(), (size_t)npx, 0, key_bits, stream));
    TM_TRY(tmp.alloc(tb));
    TM_HIP(rocprim::radix_sort_keys(tmp.p, tb, keys.as<u64>(), keys2.as<u64>(), (size_t)npx, 0, key_bits, stream));
    size_t tb2 = 0;
    TM_HIP(rocprim::run_length_encode(nullptr, tb2, keys2.as<u64>(), (unsigned int)npx, ukeys.as<u64>(), ucnt.as<uint32_t>(),
                                      nruns.as<unsigned int>(), stream));
    TM_TRY(tmp.alloc(tb2));
    TM_HIP(rocprim::run_length_encode(tmp.p, tb2, keys2.as<u64>(), (unsigned int)npx, ukeys.as<u64>(), ucnt.as<uint32_t>(),
                                      nruns.as<unsigned int>(), stream));
    unsigned int nu = 0;
    {
      HostRead hr_(stream);
      TM_TRY(hr_.get(&nu, nruns.p, 4));
      TM_TRY(hr_.wait());
    }
    // segment boundaries per palette: lower bound of each palette number in the sorted unique keys, found on the device
    DevBuf dlb;
    TM_TRY(dlb.alloc((size_t)(npal + 1) * 8));
    hipLaunchKernelGGL(k_palette_bounds, dim3((npal + 1 + 63) / 64), dim3(64), 0, stream, ukeys.as<u64>(), (int64_t)nu, npal, dlb.as<long long>());
    std::vector<long long> lb((size_t)npal + 1);
    {
      HostRead hr_(stream);
      TM_TRY(hr_.get(lb.data(), dlb.p, lb.size() * 8));
      TM_TRY(hr_.wait());
    }
    std::vector<int64_t> sb(npal, 0), sc(npal, 0);
    {
      for (int p = 0; p < npal; p++) { sb[p] = lb[p]; sc[p] = p % pal_world == pal_rank ? lb[p + 1] - lb[p] : 0; }  // other processes' palettes: empty segments
      TM_CHECK(lb[npal] == (long long)nu, TM_E_INVAL, "quantize: a tile names palette >= PaletteCount");
    }
    TM_TRY(pts.alloc((size_t)std::max<unsigned>(nu, 1) * 12));
    TM_TRY(assign.alloc((size_t)std::max<unsigned>(nu, 1) * 4));
    TM_TRY(cent.alloc((size_t)npal * pal_size * 3 * 8));
    hipLaunchKernelGGL(k_unpack_colours, dim3((int)std::min<int64_t>(((int64_t)nu + 255) / 256, 4096)), dim3(256), 0, stream,
                       ukeys.as<u64>(), (int64_t)nu, pts.as<int32_t>());
    std::vector<int> kk;
    int iters = 0;
    const bool dbg = knobs().pp_debug;
    const auto t_km = std::chrono::steady_clock::now();
    if (dbg) (void)hipStreamSynchronize(stream);
    const auto t_km0 = std::chrono::steady_clock::now();
    kmeans_run_stats().pixel_colour_iters = 0;
    TM_TRY(kmeans_batched(pts.as<int32_t>(), ucnt.as<uint32_t>(), 3, sb, sc, pal_size, max_iter, assign.as<int32_t>(), cent.as<double>(),
                          &kk, &iters, stream));
    kmeans_run_stats().pixel_iters = iters;  // (pixel_colour_iters: summed by the persistent launch, reset before it below)
    kmeans_run_stats().pixel_colours = (int64_t)nu;
    kmeans_run_stats().pixels = n * 64;
    if (dbg) fprintf(stderr, "[tm_pp]   colour keys + sort + runs %7.3f ms, k-means of %u colours %7.3f ms (%d iterations)\n",
                     std::chrono::duration<double, std::milli>(t_km0 - t_km).count() , nu,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_km0).count(), iters);
    std::vector<double> hc((size_t)npal * pal_size * 3);
    {
      HostRead hr_(stream);
      TM_TRY(hr_.get(hc.data(), cent.p, hc.size() * 8));
      TM_TRY(hr_.wait());
    }
    // host tail (P x PaletteSize colours): Round, clamp, Posterize(.,255) = identity, sort by (Val, Sat, Hue)
    // -- tilingencoder.pas:4513-4558, utils.pas:526-534, 741-748
    struct Item { int v, s, h, r, g, b, idx; };
    for (int p = 0; p < npal; p++) {
      std::vector<Item> items;
      for (int i = 0; i < kk[p]; i++) {
        const double *c = &hc[((size_t)p * pal_size + i) * 3];
        Item it;
        it.r = (int)std::min<long long>(255, std::max<long long>(0, llrint(c[0])));
        it.g = (int)std::min<long long>(255, std::max<long long>(0, llrint(c[1])));
        it.b = (int)std::min<long long>(255, std::max<long long>(0, llrint(c[2])));
        it.idx = i;
        rgb_to_hsv_bytes(it.r, it.g, it.b, it.h, it.s, it.v);
        items.push_back(it);
      }
      std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) {
        if (a.v != b.v) return a.v < b.v;
        if (a.s != b.s) return a.s < b.s;
        if (a.h != b.h) return a.h < b.h;
        if (a.r != b.r) return a.r < b.r;
        if (a.g != b.g) return a.g < b.g;
        if (a.b != b.b) return a.b < b.b;
        return a.idx < b.idx;
      });
      for (size_t i = 0; i < items.size(); i++) hpal[(size_t)p * pal_size + i] = (items[i].b << 16) | (items[i].g << 8) | items[i].r;
    }
    if (keep_keys && keep_n) { *keep_keys = std::move(ukeys); *keep_n = (int64_t)nu; }
  }
  if (pal_world > 1)
    for (int p = 0; p < npal; p++)
      if (p % pal_world != pal_rank)
        for (int i = 0; i < pal_size; i++) hpal[(size_t)p * pal_size + i] = 0;
  TM_HIP(hipMemcpyAsync(out_palettes, hpal.data(), hpal.size() * 4, hipMemcpyHostToDevice, stream));
  TM_HIP(hipStreamSynchronize(stream));
  return TM_OK;
}

}  // namespace tmx
