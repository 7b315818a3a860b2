// tm_device.h -- device-side colour math shared by the kernels.  Every operation is one IEEE op via the
// *_rn intrinsics so hipcc can never fuse or reassociate it; see DESIGN.md "Arithmetic".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tmx {

__device__ __forceinline__ uint32_t swap_rb(uint32_t c) {  // SwapRB, utils.pas:238-241 (alpha dropped)
  return ((c & 0xff) << 16) | ((c >> 16) & 0xff) | (c & 0xff00);
}

// RGBToYUV, utils.pas:478-490.  Right-hand sides evaluate in double (constants are doubles), one narrowing
// to Single per assignment.
__device__ __forceinline__ void rgb_to_yuv(int r, int g, int b, float &y, float &u, float &v) {
  const double yd = __dadd_rn(__dadd_rn(__dmul_rn((double)r, 299.0 / 1000.0), __dmul_rn((double)g, 587.0 / 1000.0)),
                              __dmul_rn((double)b, 114.0 / 1000.0));
  const float yy = (float)yd;
  u = (float)__dmul_rn(__dsub_rn((double)b, (double)yy), 0.492);
  v = (float)__dmul_rn(__dsub_rn((double)r, (double)yy), 0.877);
  y = yy;
}

// The build's stand-in for power(x, 1/3) at utils.pas:403-405.  The oracle defines it as an integer-seeded Newton root in +,-,*,/
// (tmo_cbrt_det) and proves that root equal to libm pow -- the reference's power() -- after narrowing to Single on every one of the 2^24
// colours (tests/test_oracle_pins.py).  Only that Single leaves RGBToLAB, so any double within a few units in the last place of the true
// root serves, as long as it narrows the same way on the whole domain.  Here: x^(-1/3) seeded by the hardware's log2 / exp2 (v_log_f32,
// v_exp_f32: about 3e-7 off), one Newton step in multiplications and fma (2e-13), the root as x a a with one multiplicative correction
// (the rounding of the last operations).  No division, no integer seed arithmetic: 12 double operations where the four-step form from
// the exponent seed took 27 (they were 70 % of k_load_tiles' instructions).  Equality with the oracle on the whole domain is checked ON
// THE DEVICE: tests/test_gpu_parity.py::test_lab_of_every_colour runs all 2^24 colours through tm_stage_rgb_to_lab.
#ifndef TM_CBRT_NEWTON
#define TM_CBRT_NEWTON 1
#endif
__device__ __forceinline__ double cbrt_det(float xf) {
  const double x = (double)xf;
  double a = (double)__builtin_amdgcn_exp2f(__fmul_rn(__builtin_amdgcn_logf(xf), -1.0f / 3));
#pragma unroll
  for (int i = 0; i < TM_CBRT_NEWTON; i++) {
    const double e = __fma_rn(-x, __dmul_rn(__dmul_rn(a, a), a), 1.0);
    a = __fma_rn(__dmul_rn(a, e), 1.0 / 3, a);
  }
  const double y = __dmul_rn(__dmul_rn(x, a), a);
  return __fma_rn(__dmul_rn(__fma_rn(-__dmul_rn(y, y), y, x), __dmul_rn(a, a)), 1.0 / 3, y);
}

// n / 0.17697 (utils.pas:391-393) as a reciprocal product refined by two fma; same whole-domain proof as cbrt_det
__device__ __forceinline__ double div_xyz(double n) {
  const double k = 1.0 / 0.17697;
  const double q = __dmul_rn(n, k);
  return __fma_rn(__fma_rn(-q, 0.17697, n), k, q);
}

__device__ __forceinline__ float lab_f(float t) {
  if ((double)t > 0.008856) return (float)cbrt_det(t);
  return (float)__dadd_rn(__dmul_rn(7.787, (double)t), 16.0 / 116);
}

// RGBToLAB, utils.pas:374-410, with the gamma expansion taken from the host-built 256-entry Single table.
__device__ __forceinline__ void rgb_to_lab_det(int ir, int ig, int ib, const float *__restrict__ srgb_lut, float &ol, float &oa,
                                               float &ob) {
  const double r = (double)srgb_lut[ir], g = (double)srgb_lut[ig], b = (double)srgb_lut[ib];
  float x = (float)div_xyz(__dadd_rn(__dadd_rn(__dmul_rn(r, 0.49000), __dmul_rn(g, 0.31000)), __dmul_rn(b, 0.20000)));
  float y = (float)div_xyz(__dadd_rn(__dadd_rn(__dmul_rn(r, 0.17697), __dmul_rn(g, 0.81240)), __dmul_rn(b, 0.01063)));
  float z = (float)div_xyz(__dadd_rn(__dadd_rn(__dmul_rn(r, 0.00000), __dmul_rn(g, 0.01000)), __dmul_rn(b, 0.99000)));
  x = (float)__dmul_rn((double)x, 1 / (96.6797 / 100));
  y = (float)__dmul_rn((double)y, 1 / (100.000 / 100));
  z = (float)__dmul_rn((double)z, 1 / (82.5188 / 100));
  x = lab_f(x);
  y = lab_f(y);
  z = lab_f(z);
  ol = (float)__dsub_rn(__dmul_rn(116.0, (double)y), 16.0);
  oa = (float)__dmul_rn(500.0, (double)__fsub_rn(x, y));
  ob = (float)__dmul_rn(200.0, (double)__fsub_rn(y, z));
}

}  // namespace tmx
