// tools/ta_cost.hip — what one 64-lane gather costs the L1 address path (TA/TCP) on the device at hand, by record size
// and by how many 128-byte lines the lanes touch.  All data stays L1/L2-resident: this prices the address path, not HBM.
//   hipcc --offload-arch=gfx950 -O2 tools/ta_cost.hip -o tools/_bin/ta_cost && tools/_bin/ta_cost
// Every kernel: 8 waves per SIMD on every SIMD, each wave issues ITER x 8 independent gathers; the figure printed is
// CU cycles (wall time x 2.4 GHz nominal) per wave-gather per CU, i.e. the reciprocal throughput of one CU's address path.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 256

template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const T* __restrict__ buf, const unsigned* __restrict__ lane_off,
                                                     unsigned step, unsigned mask, unsigned* __restrict__ out) {
  // lane_off[l]: element offset of lane l inside a block of lines; each iteration moves every lane by `step` elements
  unsigned idx = lane_off[threadIdx.x & 63] + (blockIdx.x & 7) * 7 * step;
  unsigned acc = 0;
  for (int it = 0; it < ITER; it++) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = buf[(idx + u * step) & mask];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const unsigned* w = reinterpret_cast<const unsigned*>(&v[u]);
      for (unsigned k = 0; k < (sizeof(T) + 3) / 4; k++) acc ^= w[k];
    }
    idx += 8 * step;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void gather_u8_kernel(const unsigned char* __restrict__ buf, const unsigned* __restrict__ lane_off,
                                                        unsigned step, unsigned mask, unsigned* __restrict__ out) {
  unsigned idx = lane_off[threadIdx.x & 63] + (blockIdx.x & 7) * 7 * step;
  unsigned acc = 0;
  for (int it = 0; it < ITER; it++) {
    unsigned v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = buf[(idx + u * step) & mask];
#pragma unroll
    for (int u = 0; u < 8; u++) acc ^= v[u];
    idx += 8 * step;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}


int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
  const int cus = prop.multiProcessorCount, blocks = cus * 8;
  const size_t bytes = 1 << 20;   // 1 MiB working set: L2-resident, mostly L1 hits for the compact patterns
  void* buf;
  unsigned *lane_off, *out;
  (void)hipMalloc(&buf, bytes);
  (void)hipMemset(buf, 1, bytes);
  (void)hipMalloc(&lane_off, 64 * sizeof(unsigned));
  (void)hipMalloc(&out, sizeof(unsigned) * blocks * 256);
  printf("# %s, %d CUs, 8 waves per SIMD; cycles at 2.4 GHz nominal per wave-gather per CU\n", prop.name, cus);
  printf("# record  pattern                         cyc/gather   (ms)\n");
  const int sizes[4] = {1, 4, 8, 16};
  for (int si = 0; si < 4; si++) {
    const int sz = sizes[si];
    const unsigned per_line = 128 / sz;
    struct Pat { const char* name; int lines; int mode; };   // mode 0: all lanes one element; 1: consecutive; 2: L lines
    const Pat pats[] = {{"one element", 1, 0}, {"consecutive elements", 0, 1}, {"1 line, spread", 1, 2},
                        {"2 lines", 2, 2}, {"4 lines", 4, 2}, {"8 lines", 8, 2}, {"16 lines", 16, 2},
                        {"32 lines", 32, 2}, {"64 lines", 64, 2}};
    for (const Pat& p : pats) {
      std::vector<unsigned> lo(64);
      for (int l = 0; l < 64; l++) {
        if (p.mode == 0) lo[l] = 0;
        else if (p.mode == 1) lo[l] = l;
        else {
          const int line = l % p.lines, within = (l / p.lines) % per_line;
          lo[l] = (unsigned)(line * 37 % 509) * per_line + (unsigned)((within * 5) % per_line);   // lines far apart
        }
      }
      (void)hipMemcpy(lane_off, lo.data(), 64 * sizeof(unsigned), hipMemcpyHostToDevice);
      const unsigned step = per_line * 3;                       // every gather moves to other lines
      const unsigned mask = (unsigned)(bytes / sz) - 1;
      hipEvent_t a, b;
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      for (int rep = 0; rep < 2; rep++) {
        if (rep == 1) (void)hipEventRecord(a, 0);
        if (sz == 1) hipLaunchKernelGGL(gather_u8_kernel, dim3(blocks), dim3(256), 0, 0, (const unsigned char*)buf, lane_off, step, mask, out);
        else if (sz == 4) hipLaunchKernelGGL(gather_kernel<unsigned>, dim3(blocks), dim3(256), 0, 0, (const unsigned*)buf, lane_off, step, mask, out);
        else if (sz == 8) hipLaunchKernelGGL(gather_kernel<uint2>, dim3(blocks), dim3(256), 0, 0, (const uint2*)buf, lane_off, step, mask, out);
        else hipLaunchKernelGGL(gather_kernel<uint4>, dim3(blocks), dim3(256), 0, 0, (const uint4*)buf, lane_off, step, mask, out);
      }
      (void)hipEventRecord(b, 0);
      (void)hipEventSynchronize(b);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, a, b);
      // per CU: 32 waves x ITER x 8 gathers
      const double gathers_per_cu = 32.0 * ITER * 8;
      printf("%3d B     %-30s %8.1f   (%.3f)\n", sz, p.name, ms * 1e-3 * 2.4e9 / gathers_per_cu, ms);
    }
  }
  return 0;
}
