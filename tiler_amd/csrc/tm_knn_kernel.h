// tm_knn_kernel.h -- the int8 MFMA distance GEMM of the KNN stage (see tm_knn.hip for the scheme).  Included by the
// tm_knn_k<HT>.hip translation units, each instantiating one database high-chunk count HT with every query
// high-chunk count HQ (the per-side digit plan makes both data dependent).
#pragma once
#include <climits>

#include "tm_common.h"

namespace tmx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int KNN_NQ = 2;  // query sub-tiles (32 queries each) per wave
constexpr int KNN_NW = 8;  // waves per workgroup (2 per SIMD)

// Database tiles (32 rows): [6 low chunks | HT high chunks] x [64 lanes] x 16 B, then 32 u32 norms |t-c|^2.
// Query tiles (32 queries): [6 low chunks | HQ high chunks] of the NEGATED centred values, then 32 u32 (|q-c|^2 >> 1).
// Columns are permuted so that the columns carrying a high digit are a prefix on each side (nested sets).
//   acc0 = T_L . Q_L (6 chunks)      acc1 = T_L[:HQ] . Q_H + T_H . Q_L[:HT]      acc2 = T_H[:m] . Q_H[:m], m = min(HT,HQ)
//   d''  = |t-c|^2 + 2*(acc2<<16 + acc1<<8 + acc0) + 2*(|q-c|^2>>1) = SSD - (|q-c|^2 & 1), exact mod 2^32.
template <int HT, int HQ>
__global__ __launch_bounds__(KNN_NW * 64) void k_knn_mfma(const uint8_t *__restrict__ tpack, int64_t tile_begin, int64_t tile_end,
                                                          const uint8_t *__restrict__ qpack, int64_t n_qtiles,
                                                          int *__restrict__ best_key, int *__restrict__ best_tile, int accumulate) {
  constexpr int NQ = KNN_NQ, NW = KNN_NW;
  constexpr int KT = 6 + HT, KQ = 6 + HQ, HM = HT < HQ ? HT : HQ;
  constexpr int T_BYTES = KT * 1024 + 128, Q_BYTES = KQ * 1024 + 128;
  constexpr int TILE_VEC = T_BYTES / 16;
  constexpr int NT = NW * 64;
  constexpr int NST = (TILE_VEC + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) uint8_t lds[2][T_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  constexpr int QT_PER_WG = NW * NQ;
  const int64_t n_wg_tiles = (n_qtiles + QT_PER_WG - 1) / QT_PER_WG;

  for (int64_t wgt = blockIdx.x; wgt < n_wg_tiles; wgt += gridDim.x) {
    v4i bq[NQ][KQ];
    int nq2[NQ], best[NQ], bestt[NQ];
    int64_t qtile[NQ];
#pragma unroll
    for (int s = 0; s < NQ; s++) {
      qtile[s] = wgt * QT_PER_WG + wave * NQ + s;
      const int64_t qt = qtile[s] < n_qtiles ? qtile[s] : n_qtiles - 1;
      const uint8_t *qb = qpack + qt * (int64_t)Q_BYTES;
#pragma unroll
      for (int kc = 0; kc < KQ; kc++) bq[s][kc] = *reinterpret_cast<const v4i *>(qb + (kc * 64 + lane) * 16);
      nq2[s] = reinterpret_cast<const int *>(qb + KQ * 1024)[lane & 31] << 1;  // 2*(|q-c|^2 >> 1)
      best[s] = INT_MAX;
      bestt[s] = INT_MAX;
    }

    v4i st[NST];
    {  // prologue: first database tile -> LDS buffer 0
      const uint8_t *src = tpack + tile_begin * (int64_t)T_BYTES;
#pragma unroll
      for (int i = 0; i < NST; i++)
        if (tid + i * NT < TILE_VEC) st[i] = *reinterpret_cast<const v4i *>(src + (tid + i * NT) * 16);
#pragma unroll
      for (int i = 0; i < NST; i++)
        if (tid + i * NT < TILE_VEC) *reinterpret_cast<v4i *>(&lds[0][(tid + i * NT) * 16]) = st[i];
    }
    __syncthreads();

    for (int64_t t = tile_begin; t < tile_end; t++) {
      const int cur = (int)((t - tile_begin) & 1);
      const bool more = t + 1 < tile_end;
      if (more) {
        const uint8_t *src = tpack + (t + 1) * (int64_t)T_BYTES;
#pragma unroll
        for (int i = 0; i < NST; i++)
          if (tid + i * NT < TILE_VEC) st[i] = *reinterpret_cast<const v4i *>(src + (tid + i * NT) * 16);
      }
      const uint8_t *L = lds[cur];
      // accumulator row of register r: (r&3) + 8*(r>>2) + 4*half  -> norms as four 16-byte reads
      int nt[16];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const v4i x = *reinterpret_cast<const v4i *>(L + KT * 1024 + (g * 8 + half * 4) * 4);
        nt[g * 4] = x[0]; nt[g * 4 + 1] = x[1]; nt[g * 4 + 2] = x[2]; nt[g * 4 + 3] = x[3];
      }
      // 2-deep software pipeline over the query sub-tiles: the MFMAs of sub-tile s run beside the VALU epilogue of
      // sub-tile s-1 (and beside the other wave of this SIMD).
      v16i acc0[2], acc1[2], acc2[2];
#pragma unroll
      for (int s = 0; s <= NQ; s++) {
        if (s < NQ) {
          const int b = s & 1;
#pragma unroll
          for (int r = 0; r < 16; r++) { acc0[b][r] = 0; acc1[b][r] = 0; acc2[b][r] = 0; }
#pragma unroll
          for (int kc = 0; kc < 6; kc++) {
            const v4i a = *reinterpret_cast<const v4i *>(L + (kc * 64 + lane) * 16);  // T_L chunk
            acc0[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][kc], acc0[b], 0, 0, 0);                    // T_L . Q_L
            if (kc < HQ) acc1[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][6 + kc], acc1[b], 0, 0, 0);  // T_L . Q_H
          }
#pragma unroll
          for (int kc = 0; kc < HT; kc++) {
            const v4i a = *reinterpret_cast<const v4i *>(L + ((6 + kc) * 64 + lane) * 16);  // T_H chunk
            acc1[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][kc], acc1[b], 0, 0, 0);                    // T_H . Q_L
            if (kc < HM) acc2[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[s][6 + kc], acc2[b], 0, 0, 0);  // T_H . Q_H
          }
        }
        if (s > 0) {
          const int b = (s - 1) & 1;
          int m = INT_MAX;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            unsigned x = (unsigned)acc0[b][r];
            if (HT + HQ > 0) {
              unsigned hi = (unsigned)acc1[b][r];
              if (HM > 0) hi += (unsigned)acc2[b][r] << 8;
              x += hi << 8;
            }
            const int d = (int)((x << 1) + (unsigned)nt[r] + (unsigned)nq2[s - 1]);
            m = min(m, d);
          }
          if (m < best[s - 1]) { best[s - 1] = m; bestt[s - 1] = (int)t; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) {
#pragma unroll
        for (int i = 0; i < NST; i++)
          if (tid + i * NT < TILE_VEC) *reinterpret_cast<v4i *>(&lds[cur ^ 1][(tid + i * NT) * 16]) = st[i];
      }
      __syncthreads();
    }

#pragma unroll
    for (int s = 0; s < NQ; s++) {
      const int ob = __shfl_xor(best[s], 32), ot = __shfl_xor(bestt[s], 32);
      if (ob < best[s] || (ob == best[s] && ot < bestt[s])) { best[s] = ob; bestt[s] = ot; }
      if (lane < 32 && qtile[s] < n_qtiles) {
        const int64_t q = qtile[s] * 32 + lane;
        if (accumulate) {
          const int pk = best_key[q], pt = best_tile[q];
          if (pk < best[s] || (pk == best[s] && pt < bestt[s])) { best[s] = pk; bestt[s] = pt; }
        }
        best_key[q] = best[s];
        best_tile[q] = bestt[s];
      }
    }
  }
}

struct KnnLaunch {
  const uint8_t *tpack; int64_t tile_begin, tile_end;
  const uint8_t *qpack; int64_t n_qtiles;
  int *best_key, *best_tile; int accumulate; int ncu; hipStream_t stream;
};

// one per HT, defined in tm_knn_k<HT>.hip
template <int HT> void knn_launch_ht(int hq, const KnnLaunch &a);

#define TM_KNN_DEFINE_HT(HT)                                                                                                   \
  template <> void knn_launch_ht<HT>(int hq, const KnnLaunch &a) {                                                             \
    const int64_t wg_tiles = (a.n_qtiles + KNN_NQ * KNN_NW - 1) / (KNN_NQ * KNN_NW);                                           \
    const dim3 grid((unsigned)(wg_tiles < a.ncu ? wg_tiles : a.ncu)), block(KNN_NW * 64);                                      \
    switch (hq) {                                                                                                              \
      case 0: hipLaunchKernelGGL((k_knn_mfma<HT, 0>), grid, block, 0, a.stream, a.tpack, a.tile_begin, a.tile_end, a.qpack, a.n_qtiles, a.best_key, a.best_tile, a.accumulate); break; \
      case 1: hipLaunchKernelGGL((k_knn_mfma<HT, 1>), grid, block, 0, a.stream, a.tpack, a.tile_begin, a.tile_end, a.qpack, a.n_qtiles, a.best_key, a.best_tile, a.accumulate); break; \
      case 2: hipLaunchKernelGGL((k_knn_mfma<HT, 2>), grid, block, 0, a.stream, a.tpack, a.tile_begin, a.tile_end, a.qpack, a.n_qtiles, a.best_key, a.best_tile, a.accumulate); break; \
      case 3: hipLaunchKernelGGL((k_knn_mfma<HT, 3>), grid, block, 0, a.stream, a.tpack, a.tile_begin, a.tile_end, a.qpack, a.n_qtiles, a.best_key, a.best_tile, a.accumulate); break; \
      case 4: hipLaunchKernelGGL((k_knn_mfma<HT, 4>), grid, block, 0, a.stream, a.tpack, a.tile_begin, a.tile_end, a.qpack, a.n_qtiles, a.best_key, a.best_tile, a.accumulate); break; \
      case 5: hipLaunchKernelGGL((k_knn_mfma<HT, 5>), grid, block, 0, a.stream, a.tpack, a.tile_begin, a.tile_end, a.qpack, a.n_qtiles, a.best_key, a.best_tile, a.accumulate); break; \
      default: hipLaunchKernelGGL((k_knn_mfma<HT, 6>), grid, block, 0, a.stream, a.tpack, a.tile_begin, a.tile_end, a.qpack, a.n_qtiles, a.best_key, a.best_tile, a.accumulate); break; \
    }                                                                                                                          \
  }

}  // namespace tmx
