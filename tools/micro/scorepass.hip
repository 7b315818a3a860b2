// development aid: what the pieces of one scoring pass of k_h_resident (tm_kmeans.hip) cost, on a single workgroup: a kernel repeats the pass REPS
// times with pieces switched on by template flags (1 chain, 2 merge, 4 results + moved list, 8 the two barriers + moved rows), timed from the host.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scorepass.hip -o scorepass && ./scorepass
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
constexpr int D = 192, PITCH = 17, NT = 1024, REPS = 2000;
__device__ __forceinline__ float hr_up(double x) { return (float)(x * (1.0 + 1.2e-7)); }
__device__ __forceinline__ float hr_down(double x) { x = fmin(x, 1.0e37); return (float)(x - fabs(x) * 1.2e-7); }
template <int F, int BD>
__global__ __launch_bounds__(NT) void k(const int *rows, const double *cent, const unsigned *w, double *out, int nneed, int kk) {
  __shared__ __attribute__((aligned(16))) int s_rows[64 * D];
  __shared__ double s_c[D * PITCH];
  __shared__ u64 s_delta[16 * 193];
  __shared__ float s_ub[2048], s_lb[2048];
  __shared__ uint8_t s_a[2048];
  __shared__ uint16_t s_need[2048];
  __shared__ int s_moved[64 * 3], s_nmoved;
  __shared__ unsigned s_wt[64];
  const int tid = threadIdx.x, grp = tid >> 4, l16 = tid & 15, wave = tid >> 6, lane = tid & 63;
  for (int e = tid; e < 64 * D; e += NT) s_rows[e] = rows[e];
  for (int e = tid; e < D * PITCH; e += NT) s_c[e] = cent[e];
  for (int e = tid; e < 16 * 193; e += NT) s_delta[e] = 0;
  for (int e = tid; e < 2048; e += NT) { s_ub[e] = 0; s_lb[e] = 0; s_a[e] = 3; s_need[e] = (uint16_t)e; }
  if (tid == 0) s_nmoved = 0;
  __syncthreads();
  double keep = 0.0;
#pragma unroll 1
  for (int rep = 0; rep < REPS; rep++) {
    const int base = 0;
    const bool active = base + grp < nneed;
    const int slot = s_need[active ? base + grp : base];
    const int64_t gi = (int64_t)slot * 256 + 3 + rep;
    unsigned wv = 1u;
    if ((F & 4) && active && l16 == 0 && w) wv = w[gi];
    double bd = 1.0e300, bd2 = 1.0e300;
    int bc = 0x7fffffff;
    if (base + (wave << 2) < nneed) {
      double sacc = 0.0;
      if (F & 1) {
        const int4 *rp = reinterpret_cast<const int4 *>(s_rows + grp * D);
        const double *cp = s_c + l16;
        constexpr int NR = BD / 4;
        auto ld = [&](int jb, int4 (&r)[NR], double (&c)[BD]) {
#pragma unroll
          for (int u = 0; u < NR; u++) r[u] = rp[jb * NR + u];
#pragma unroll
          for (int u = 0; u < BD; u++) c[u] = cp[(jb * BD + u) * PITCH];
        };
        auto acc = [&](const int4 (&r)[NR], const double (&c)[BD]) {
#pragma unroll
          for (int u = 0; u < NR; u++) {
            const int v[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
#pragma unroll
            for (int q = 0; q < 4; q++) { const double t = __dsub_rn((double)v[q], c[u * 4 + q]); sacc = __fma_rn(t, t, sacc); }
          }
        };
        int4 ra[NR], rb[NR];
        double ca[BD], cb[BD];
        ld(0, ra, ca);
#pragma unroll
        for (int jb = 0; jb < D / BD; jb += 2) {
          ld(jb + 1, rb, cb);
          __builtin_amdgcn_sched_barrier(0);
          acc(ra, ca);
          __builtin_amdgcn_sched_barrier(0);
          if (jb + 2 < D / BD) ld(jb + 2, ra, ca);
          __builtin_amdgcn_sched_barrier(0);
          acc(rb, cb);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else sacc = (double)(l16 + rep);
      if (l16 < kk) { bd = sacc; bc = l16; }
      if (F & 2) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          const double od = __shfl_xor(bd, o), od2 = __shfl_xor(bd2, o);
          const int oc = __shfl_xor(bc, o);
          const bool take = od < bd || (od == bd && oc < bc);
          const double loser = take ? bd : od;
          bd2 = fmin(fmin(bd2, od2), loser);
          if (take) { bd = od; bc = oc; }
        }
      }
      if ((F & 4) && active && l16 == 0) {
        s_ub[slot] = hr_up(sqrt(bd) * (1.0 + 1e-12));
        s_lb[slot] = hr_down(sqrt(bd2) * (1.0 - 1e-12));
        const int old = s_a[slot];
        if (old != bc) {
          s_a[slot] = (uint8_t)(bc ^ (rep & 1));
          const int m = atomicAdd(&s_nmoved, 1);
          s_moved[m * 3] = grp; s_moved[m * 3 + 1] = old; s_moved[m * 3 + 2] = bc;
          s_wt[grp] = wv;
        }
      }
      keep += bd + bd2;
    }
    if (F & 8) {
      __syncthreads();
      const int nmoved = s_nmoved;
      for (int e = wave; e < nmoved; e += NT / 64) {
        const int ps = s_moved[e * 3], old = s_moved[e * 3 + 1], nw = s_moved[e * 3 + 2];
        const long long wi = (long long)s_wt[ps];
#pragma unroll
        for (int j = lane; j <= D; j += 64) {
          const u64 v = j < D ? (u64)(wi * s_rows[ps * D + j]) : (u64)wi;
          atomicAdd(&s_delta[nw * 193 + j], v);
          atomicAdd(&s_delta[old * 193 + j], (u64)0 - v);
        }
      }
      __syncthreads();
      if (tid == 0) s_nmoved = 0;
    }
  }
  out[tid] = keep + (double)s_delta[tid];
}
template <int F, int BD>
static void run(const char *what, const int *rows, const double *cent, const unsigned *w, double *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-46s", what);
  for (int nneed : {1, 5, 16, 64}) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL((k<F, BD>), dim3(1), dim3(NT), 0, 0, rows, cent, w, out, nneed, 16);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("  %2d points %6.2f us", nneed, best * 1000.0f / REPS);
  }
  printf("\n");
}
int main() {
  int *rows; double *cent, *out; unsigned *w;
  hipMalloc(&rows, 64 * D * 4); hipMalloc(&cent, D * PITCH * 8); hipMalloc(&out, 1024 * 8); hipMalloc(&w, 16 << 20);
  hipMemset(rows, 1, 64 * D * 4); hipMemset(cent, 0, D * PITCH * 8); hipMemset(w, 1, 16 << 20);
  run<1, 8>("chain (8 dimensions a step)", rows, cent, w, out);
  run<1, 4>("chain (4 dimensions a step)", rows, cent, w, out);
  run<3, 8>("chain + merge", rows, cent, w, out);
  run<7, 8>("chain + merge + results (w from memory)", rows, cent, w, out);
  run<15, 8>("the whole pass", rows, cent, w, out);
  run<15, 4>("the whole pass (4 dimensions a step)", rows, cent, w, out);
  run<14, 8>("the whole pass without the chain", rows, cent, w, out);
  run<8, 8>("barriers + moved rows only", rows, cent, w, out);
  return 0;
}
