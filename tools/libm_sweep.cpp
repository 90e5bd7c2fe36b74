// libm_sweep.cpp — sweeps ALL 2^32 float arguments: csrc/tdr_sincosf.h (both builds of glibc's sinf / cosf) against the
// sinf / cosf of the libm this program links.  On the image's glibc 2.35 (Xeon, FMA): fused build 0 mismatches for
// both functions, plain build 12 (sinf) / 22 (cosf).
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -fopenmp -I top_down_renderer_amd/csrc tools/libm_sweep.cpp -o /tmp/sweep && /tmp/sweep
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <omp.h>
#include "tdr_sincosf.h"
int main() {
  long long bad[4] = {0,0,0,0}; // sin plain, sin fma, cos plain, cos fma
  uint32_t ex[4] = {0,0,0,0};
#pragma omp parallel for schedule(dynamic, 1<<20) reduction(+:bad[:4])
  for (long long i = 0; i < (1LL<<32); i++) {
    uint32_t u = (uint32_t)i; float y; memcpy(&y, &u, 4);
    float s = sinf(y), c = cosf(y);
    float r[4] = {tdr_libm::sinf_<false>(y), tdr_libm::sinf_<true>(y), tdr_libm::cosf_<false>(y), tdr_libm::cosf_<true>(y)};
    float ref[4] = {s, s, c, c};
    for (int k = 0; k < 4; k++) {
      uint32_t a, b; memcpy(&a, &r[k], 4); memcpy(&b, &ref[k], 4);
      bool same = a == b || (r[k] != r[k] && ref[k] != ref[k]);
      if (!same) { bad[k]++; ex[k] = u; }
    }
  }
  printf("mismatches: sin plain %lld (ex %08x), sin fma %lld (ex %08x), cos plain %lld (ex %08x), cos fma %lld (ex %08x)\n",
         bad[0], ex[0], bad[1], ex[1], bad[2], ex[2], bad[3], ex[3]);
}
