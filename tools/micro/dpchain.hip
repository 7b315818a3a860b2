// development aid: what a lone wave's chain of 192 x (cvt, sub, fma) in double costs on gfx950 -- register operands, LDS operands read one step ahead,
// one / two / four waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off dpchain.hip -o dpchain && ./dpchain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
constexpr int D = 192, PITCH = 17;
template <int MODE>
__global__ __launch_bounds__(1024) void k(const int *rows, const double *cent, double *out, u64 *ticks, int nwaves_active) {
  __shared__ __attribute__((aligned(16))) int s_rows[64 * D];
  __shared__ double s_c[D * PITCH];
  const int tid = threadIdx.x, grp = tid >> 4, l16 = tid & 15, wave = tid >> 6;
  for (int e = tid; e < 64 * D; e += 1024) s_rows[e] = rows[e];
  for (int e = tid; e < D * PITCH; e += 1024) s_c[e] = cent[e];
  __syncthreads();
  const u64 t0 = __builtin_amdgcn_s_memtime();
  double sacc = 0.0;
  if (wave < nwaves_active) {
    const int4 *rp = reinterpret_cast<const int4 *>(s_rows + grp * D);
    const double *cp = s_c + l16;
    if (MODE == 0) {  // pipelined LDS operands (the kernel's form)
      auto ld = [&](int jb, int4 (&r)[2], double (&c)[8]) {
        r[0] = rp[jb * 2]; r[1] = rp[jb * 2 + 1];
#pragma unroll
        for (int u = 0; u < 8; u++) c[u] = cp[(jb * 8 + u) * PITCH];
      };
      auto acc = [&](const int4 (&r)[2], const double (&c)[8]) {
        const int v[8] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w};
#pragma unroll
        for (int u = 0; u < 8; u++) { const double t = __dsub_rn((double)v[u], c[u]); sacc = __fma_rn(t, t, sacc); }
      };
      int4 ra[2], rb[2];
      double ca[8], cb[8];
      ld(0, ra, ca);
#pragma unroll
      for (int jb = 0; jb < D / 8; jb += 2) {
        ld(jb + 1, rb, cb);
        __builtin_amdgcn_sched_barrier(0);
        acc(ra, ca);
        __builtin_amdgcn_sched_barrier(0);
        if (jb + 2 < D / 8) ld(jb + 2, ra, ca);
        __builtin_amdgcn_sched_barrier(0);
        acc(rb, cb);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 1) {  // register operands only: the arithmetic's own latency
      double c = cp[0];
      int v = rp[0].x;
#pragma unroll 16
      for (int j = 0; j < D; j++) { const double t = __dsub_rn((double)(v + j), c); sacc = __fma_rn(t, t, sacc); asm volatile("" : "+v"(sacc)); }
    } else if (MODE == 2) {  // only the fma chain (differences precomputed 8 at a time)
      double c = cp[0];
      int v = rp[0].x;
#pragma unroll 1
      for (int j0 = 0; j0 < D; j0 += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = __dsub_rn((double)(v + j0 + u), c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) sacc = __fma_rn(t[u], t[u], sacc);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  out[tid] = sacc;
  if (tid == 0) ticks[0] = t1 - t0;
}
int main() {
  int *rows; double *cent, *out; u64 *ticks;
  hipMalloc(&rows, 64 * D * 4); hipMalloc(&cent, D * PITCH * 8); hipMalloc(&out, 1024 * 8); hipMalloc(&ticks, 8);
  hipMemset(rows, 1, 64 * D * 4); hipMemset(cent, 0, D * PITCH * 8);
  for (int mode = 0; mode < 3; mode++)
    for (int nw : {1, 4, 8, 16}) {
      u64 best = ~0ull;
      for (int rep = 0; rep < 5; rep++) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(1024), 0, 0, rows, cent, out, ticks, nw);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(1024), 0, 0, rows, cent, out, ticks, nw);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(1024), 0, 0, rows, cent, out, ticks, nw);
        u64 h; hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        if (h < best) best = h;
      }
      printf("mode %d (%s), %2d waves of the workgroup active: %llu ticks for 192 terms (wave 0)\n", mode, mode == 0 ? "LDS operands, one step ahead" : mode == 1 ? "register operands" : "fma chain, differences ahead", nw, best);
    }
  return 0;
}
