// tdr_atan2f.h — glibc's atan2f on the device, bit for bit (shared by tdr_raster.hip and tdr_geo.hip).
#ifndef TDR_ATAN2F_H_
#define TDR_ATAN2F_H_
#include <hip/hip_runtime.h>

// atan2f exactly as glibc computes it (the reference calls the host libm, src/scan_renderer_polar.cpp:97).
// glibc's float atan2f / atanf are the fdlibm algorithms (argument reduction to four intervals + an 11-term odd/even
// polynomial, all in float); restated here operation for operation — compiled with -ffp-contract=off, IEEE divide —
// so the device result is bit-identical to glibc 2.35's (checked against glibc on 8e7 inputs on the CPU, and by
// tests/test_gpu_parity.py::test_atan2f_bit_exact on the GPU).  The device math library's atan2f differs in the last
// ulp on ~1e-5 of the inputs, which would move a point into the neighbouring theta bin.
__device__ __forceinline__ float tdr_atanf(float x) {
  const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                        9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                        4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
  const int hx = __float_as_int(x), ix = hx & 0x7fffffff;
  float hi = 0.f, lo = 0.f;
  int id;
  if (ix >= 0x4c000000) {  // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {   // |x| < 0.4375
    if (ix < 0x31000000) return x;
    id = -1;
  } else {
    x = fabsf(x);
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) { id = 0; hi = atanhi[0]; lo = atanlo[0]; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else { id = 1; hi = atanhi[1]; lo = atanlo[1]; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { id = 2; hi = atanhi[2]; lo = atanlo[2]; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else { id = 3; hi = atanhi[3]; lo = atanlo[3]; x = -1.0f / x; }
    }
  }
  const float z = x * x, w = z * z;
  const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  const float r = hi - ((x * (s1 + s2) - lo) - x);
  return hx < 0 ? -r : r;
}
__device__ __forceinline__ float tdr_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
              pi_lo = -8.7422776573e-08f;
  const int hx = __float_as_int(x), ix = hx & 0x7fffffff, hy = __float_as_int(y), iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return tdr_atanf(y);
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
    return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = tdr_atanf(fabsf(y / x));
  switch (m) {
    case 0: return z;
    case 1: return __int_as_float(__float_as_int(z) ^ (int)0x80000000);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}
#endif  // TDR_ATAN2F_H_
