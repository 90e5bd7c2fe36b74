// tdr_logf.h — the host libm's logf, bit for bit, on the device.
//
// libstdc++'s std::normal_distribution<float> (bits/random.tcc: Marsaglia's polar method) evaluates
// std::sqrt(-2 * std::log(r2) / r2) in float, i.e. glibc's logf: the reference's propagate noise
// (src/state_particle.cpp:64-73) depends on its roundings.  glibc >= 2.27 (the image ships 2.35) computes logf in double —
// sysdeps/ieee754/flt-32/e_logf.c, the ARM optimized-routines algorithm: x = 2^k z with z in [0.7, 1.4), a 16-entry table
// of {1/c, log c}, a cubic in r = z/c - 1 — and rounds once to float.  Restated here operation for operation (a third-party
// libm algorithm, like tdr_sincosf.h; the table is the library's own data).  As for sinf / cosf, x86-64 glibc picks at load
// time between the plain build and one compiled with -mfma, in which the five `a * b + c` of the evaluation are fused:
// FMA = true / false, chosen by tdr_libm_variant().  tools/logf_sweep.cpp sweeps all 2^32 arguments against the host's
// logf; tests/test_libm.py checks strided sweeps on the CPU and the device against the host on the GPU.
#ifndef TDR_LOGF_H_
#define TDR_LOGF_H_
#include <stdint.h>

#include "tdr_sincosf.h"   // mad<FMA>, f2u

namespace tdr_libm {

__host__ __device__ inline float u2f(uint32_t u) {
  union { uint32_t u; float f; } v;
  v.u = u;
  return v.f;
}

template <bool FMA>
__host__ __device__ inline float logf_t(float x) {
  // __logf_data: {invc, logc} for the 16 subintervals of [0x1.66p-1, 0x1.66p0)
  const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
                           0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
                           0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
                           0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
  const double logc[16] = {-0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
                           -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
                           -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
                           0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
  const double Ln2 = 0x1.62e42fefa39efp-1;
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  uint32_t ix = f2u(x);
  if (ix == 0x3f800000u) return 0.f;                       // log(1) = +0 exactly
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {     // x < 0x1p-126, inf or nan
    if (ix * 2 == 0) return -__builtin_inff();             // log(+-0) = -inf (divide-by-zero)
    if (ix == 0x7f800000u) return x;                       // log(inf) = inf
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return __builtin_nanf("");   // x < 0, nan (the sign / payload of
    ix = f2u(x * 0x1p23f);                                 // glibc's NaN is not reproduced: no caller looks at it)
    ix -= 23u << 23;                                       // subnormal: normalise
  }
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) & 15u);
  const int k = (int32_t)tmp >> 23;                        // arithmetic shift
  const uint32_t iz = ix - (tmp & (0x1ffu << 23));
  const double z = (double)u2f(iz);
  const double r = mad<FMA>(-1.0, z, invc[i]);             // z * invc - 1
  const double y0 = mad<FMA>(logc[i], (double)k, Ln2);     // logc + k * Ln2
  const double r2 = r * r;
  double y = mad<FMA>(A2, A1, r);                          // A[1] * r + A[2]
  y = mad<FMA>(y, A0, r2);                                 // A[0] * r2 + y
  y = mad<FMA>(y0 + r, y, r2);                             // y * r2 + (y0 + r)
  return (float)y;
}
__host__ __device__ inline float logf_v(float x, int fma) { return fma ? logf_t<true>(x) : logf_t<false>(x); }

}  // namespace tdr_libm
#endif  // TDR_LOGF_H_
