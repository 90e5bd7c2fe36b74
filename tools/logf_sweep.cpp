// logf_sweep.cpp — sweeps ALL 2^32 float arguments: csrc/tdr_logf.h (both builds of glibc's logf) against the host libm.
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -fopenmp -I top_down_renderer_amd/csrc tools/logf_sweep.cpp -o /tmp/logf_sweep && /tmp/logf_sweep
// This image (glibc 2.35): "mismatches vs host: fused 0, plain 0; fused != plain on 0 arguments" (15 s on 8 cores).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "tdr_logf.h"

int main() {
  long bad_f = 0, bad_p = 0, differ = 0;
#pragma omp parallel for reduction(+ : bad_f, bad_p, differ) schedule(static, 1 << 20)
  for (long long u = 0; u < (1ll << 32); u++) {
    const uint32_t b = (uint32_t)u;
    float x;
    memcpy(&x, &b, 4);
    const float h = logf(x), a = tdr_libm::logf_t<true>(x), p = tdr_libm::logf_t<false>(x);
    uint32_t hb, ab, pb;
    memcpy(&hb, &h, 4);
    memcpy(&ab, &a, 4);
    memcpy(&pb, &p, 4);
    const bool hn = h != h, an = a != a, pn = p != p;   // NaN results: any NaN matches any NaN
    if (!(hn && an) && hb != ab) {
      bad_f++;
      if (bad_f < 5) printf("fused mismatch x=%a host=%a mine=%a\n", x, h, a);
    }
    if (!(hn && pn) && hb != pb) bad_p++;
    if (!(an && pn) && ab != pb) differ++;
  }
  printf("mismatches vs host: fused %ld, plain %ld; fused != plain on %ld arguments\n", bad_f, bad_p, differ);
  return bad_f || bad_p ? 1 : 0;
}
