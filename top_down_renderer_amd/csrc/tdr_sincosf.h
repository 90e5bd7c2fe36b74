// tdr_sincosf.h — the host libm's sinf / cosf, bit for bit, on the device.
//
// The reference calls std::sin / std::cos on floats (Eigen::Rotation2D<float> in src/state_particle.cpp:58, cos/sin in
// src/top_down_map.cpp:381-385), i.e. glibc's sinf / cosf.  Cell indices are ROUNDED products of those values, so a
// 1-ulp difference flips a sample now and then: the device must reproduce the library's own roundings.  glibc >= 2.28
// (the image ships 2.35) evaluates both in double — sysdeps/ieee754/flt-32/{s_sinf.c,s_cosf.c,sincosf.h}, the ARM
// optimized-routines algorithm: |x| < pi/4 a polynomial, |x| < 120 a one-step reduction by pi/2, larger arguments a
// 192-bit 4/pi table — and rounds once to float.  It is restated here operation for operation (a third-party libm
// algorithm, like the atan2f in tdr_raster.hip; not reference code).
//
// One degree of freedom: on x86-64 glibc selects at load time (ifunc) between the plain build of that C code and one
// compiled with -mfma -mavx2, in which every `a + b * c` is a fused multiply-add.  The two differ in the last double
// bit, which reaches the float result for a handful of arguments.  FMA = true reproduces the fused build (every x86
// CPU since Haswell / Zen), FMA = false the plain one; tdr_libm_variant() (tdr_filter.hip) probes the host's sinf on
// arguments where the two differ and the kernels follow it.  tests/test_libm.py checks the restatement against the
// host's sinf / cosf on all 2^32 arguments (CPU) and the device against it (GPU).
#ifndef TDR_SINCOSF_H_
#define TDR_SINCOSF_H_
#include <stdint.h>

#ifndef __HIPCC__
#define __host__
#define __device__
#endif

namespace tdr_libm {

struct SinCosTab {
  double sign[4];
  double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3;
};

template <bool FMA>
__host__ __device__ inline double mad(double a, double b, double c) {   // a + b * c as the respective build rounds it
  if constexpr (FMA) return __builtin_fma(b, c, a);
  else return a + b * c;
}

__host__ __device__ inline uint32_t f2u(float f) {
  union { float f; uint32_t u; } v;
  v.f = f;
  return v.u;
}
__host__ __device__ inline uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ff; }

// sinf_poly (sincosf.h): the sine polynomial for even n, the cosine polynomial for odd n; NEG selects the table whose
// cosine coefficients are negated (quadrants 2 and 3)
template <bool FMA>
__host__ __device__ inline float sin_poly(double x, double x2, bool neg, int n) {
  const double c0 = neg ? -0x1p0 : 0x1p0;
  const double c1 = neg ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
  const double c2 = neg ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5;
  const double c3 = neg ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
  const double c4 = neg ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  if ((n & 1) == 0) {
    const double x3 = x * x2;
    const double t1 = mad<FMA>(s2, x2, s3);
    const double x7 = x3 * x2;
    const double s = mad<FMA>(x, x3, s1);
    return (float)mad<FMA>(s, x7, t1);
  } else {
    const double x4 = x2 * x2;
    const double t2 = mad<FMA>(c3, x2, c4);
    const double t1 = mad<FMA>(c0, x2, c1);
    const double x6 = x4 * x2;
    const double c = mad<FMA>(t1, x4, c2);
    return (float)mad<FMA>(c, x6, t2);
  }
}

// reduce_fast (sincosf.h, TOINT_INTRINSICS == 0): x - n * pi/2, n = round(x * 2/pi) through a 2^24-scaled conversion
template <bool FMA>
__host__ __device__ inline double reduce_fast(double x, int* np) {
  const double r = x * 0x1.45F306DC9C883p+23;
  const int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return mad<FMA>(x, -(double)n, 0x1.921FB54442D18p0);   // x - n * hpi
}

// reduce_large (sincosf.h): |x| >= 120 against 4/pi to 192 bits (__inv_pio4), integer arithmetic
__host__ __device__ inline double reduce_large(uint32_t xi, int* np) {
  const uint32_t inv_pio4[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
                                 0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
                                 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
                                 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
  const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
  const int shift = (xi >> 23) & 7;
  uint64_t n, res0, res1, res2;
  xi = (xi & 0xffffff) | 0x800000;
  xi <<= shift;
  res0 = xi * arr[0];
  res1 = (uint64_t)xi * arr[4];
  res2 = (uint64_t)xi * arr[8];
  res0 = (res2 >> 32) | (res0 << 32);
  res0 += res1;
  n = (res0 + (1ULL << 61)) >> 62;
  res0 -= n << 62;
  const double x = (double)(int64_t)res0;
  *np = (int)n;
  return x * 0x1.921FB54442D18p-62;
}

__host__ __device__ inline double quadrant_sign(int n) { return ((n + 1) & 2) ? -1.0 : 1.0; }   // {1, -1, -1, 1}[n & 3]

// COS = false: sinf (s_sinf.c); COS = true: cosf (s_cosf.c)
template <bool FMA, bool COS>
__host__ __device__ inline float sincos_one(float y) {
  double x = y;
  int n;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    const double x2 = x * x;
    if (abstop12(y) < abstop12(0x1p-12f)) return COS ? 1.0f : y;
    return sin_poly<FMA>(x, x2, false, COS ? 1 : 0);
  } else if (abstop12(y) < abstop12(120.0f)) {
    x = reduce_fast<FMA>(x, &n);
    const double s = quadrant_sign(n);
    return sin_poly<FMA>(x * s, x * x, (n & 2) != 0, COS ? n ^ 1 : n);
  } else if (abstop12(y) < 0x7f8) {
    const uint32_t xi = f2u(y);
    const int sign = (int)(xi >> 31);
    x = reduce_large(xi, &n);
    const double s = quadrant_sign(n + sign);
    return sin_poly<FMA>(x * s, x * x, ((n + sign) & 2) != 0, COS ? n ^ 1 : n);
  }
  return y - y;   // inf / NaN -> NaN (__math_invalidf)
}

template <bool FMA>
__host__ __device__ inline float sinf_(float y) { return sincos_one<FMA, false>(y); }
template <bool FMA>
__host__ __device__ inline float cosf_(float y) { return sincos_one<FMA, true>(y); }

// run-time choice of the variant (wave-uniform on the device)
__host__ __device__ inline float sinf_v(float y, int fma) { return fma ? sinf_<true>(y) : sinf_<false>(y); }
__host__ __device__ inline float cosf_v(float y, int fma) { return fma ? cosf_<true>(y) : cosf_<false>(y); }

}  // namespace tdr_libm
#endif  // TDR_SINCOSF_H_
