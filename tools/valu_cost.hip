// tools/valu_cost.hip — issue cost of the vector instructions the scoring loops are made of, on the device at hand.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
// Every kernel runs 8 independent chains of ONE instruction, WAVES waves per SIMD on every SIMD; the figure printed is
// SIMD cycles (s_memtime) per wave-instruction at that occupancy, i.e. the reciprocal throughput the scheduler delivers.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 2000

#define KERNEL(NAME, TXT)                                                                                       \
  __global__ __launch_bounds__(256) void NAME(unsigned* __restrict__ out, unsigned long long* __restrict__ cyc, \
                                              float sf) {                                                       \
    unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6,    \
             r7 = r0 + 7;                                                                                       \
    unsigned long long q0 = r0, q1 = r1, q2 = r2, q3 = r3;                                                      \
    unsigned a = blockIdx.x + 3, b = threadIdx.x * 7 + 1;                                                       \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                 \
    for (int it = 0; it < ITER; it++) {                                                                         \
      asm volatile(TXT TXT TXT TXT                                                                              \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(q0),   \
                     "+v"(q1), "+v"(q2), "+v"(q3)                                                               \
                   : "v"(a), "v"(b), "s"(sf));                                                                  \
    }                                                                                                           \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                 \
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (unsigned)(q0 ^ q1 ^ q2 ^ q3);  \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                            \
  }

// 8 instructions per TXT (x4 per iteration = 32), operands: %0-%7 32-bit chains, %8-%11 64-bit chains, %12 %13 v, %14 s
#define I8(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" \
                     OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"
#define I8P(OP, TAIL) OP " %8, %8" TAIL "\n" OP " %9, %9" TAIL "\n" OP " %10, %10" TAIL "\n" OP " %11, %11" TAIL "\n" \
                      OP " %8, %8" TAIL "\n" OP " %9, %9" TAIL "\n" OP " %10, %10" TAIL "\n" OP " %11, %11" TAIL "\n"

KERNEL(k_add_f32, I8("v_add_f32", ", %12"))
KERNEL(k_add_f32_s, I8("v_add_f32", ", %14"))
KERNEL(k_fma_f32, I8("v_fma_f32", ", %12, %13"))
KERNEL(k_fma_f32_s, I8("v_fma_f32", ", %14, %13"))
KERNEL(k_pk_add_f32, I8P("v_pk_add_f32", ", %9"))
KERNEL(k_pk_fma_f32, I8P("v_pk_fma_f32", ", %10, %11"))
KERNEL(k_pk_mul_f32, I8P("v_pk_mul_f32", ", %9"))
KERNEL(k_med3_f32, I8("v_med3_f32", ", %12, %13"))
KERNEL(k_cvt_flr, "v_cvt_flr_i32_f32 %0, %0\nv_cvt_flr_i32_f32 %1, %1\nv_cvt_flr_i32_f32 %2, %2\nv_cvt_flr_i32_f32 %3, %3\n"
                  "v_cvt_flr_i32_f32 %4, %4\nv_cvt_flr_i32_f32 %5, %5\nv_cvt_flr_i32_f32 %6, %6\nv_cvt_flr_i32_f32 %7, %7\n")
KERNEL(k_cvt_rpi, "v_cvt_rpi_i32_f32 %0, %0\nv_cvt_rpi_i32_f32 %1, %1\nv_cvt_rpi_i32_f32 %2, %2\nv_cvt_rpi_i32_f32 %3, %3\n"
                  "v_cvt_rpi_i32_f32 %4, %4\nv_cvt_rpi_i32_f32 %5, %5\nv_cvt_rpi_i32_f32 %6, %6\nv_cvt_rpi_i32_f32 %7, %7\n")
KERNEL(k_sub_u32, I8("v_sub_u32", ", %12"))
KERNEL(k_bfe_i32_v, I8("v_bfe_i32", ", %12, 1"))
KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0\nv_cvt_f32_u32 %1, %1\nv_cvt_f32_u32 %2, %2\nv_cvt_f32_u32 %3, %3\n"
                      "v_cvt_f32_u32 %4, %4\nv_cvt_f32_u32 %5, %5\nv_cvt_f32_u32 %6, %6\nv_cvt_f32_u32 %7, %7\n")
KERNEL(k_cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0\nv_cvt_f32_ubyte0 %1, %1\nv_cvt_f32_ubyte0 %2, %2\nv_cvt_f32_ubyte0 %3, %3\n"
                         "v_cvt_f32_ubyte0 %4, %4\nv_cvt_f32_ubyte0 %5, %5\nv_cvt_f32_ubyte0 %6, %6\nv_cvt_f32_ubyte0 %7, %7\n")
KERNEL(k_mad_i32_i24, I8("v_mad_i32_i24", ", %12, %13"))
KERNEL(k_mad_u32_u24, I8("v_mad_u32_u24", ", %12, %13"))
KERNEL(k_mul_u32_u24, I8("v_mul_u32_u24", ", %12"))
KERNEL(k_mul_lo_u32, I8("v_mul_lo_u32", ", %12"))
KERNEL(k_lshl_add_u32, I8("v_lshl_add_u32", ", 5, %13"))
KERNEL(k_add_lshl_u32, I8("v_add_lshl_u32", ", %12, 3"))
KERNEL(k_lshl_or_b32, I8("v_lshl_or_b32", ", 5, %13"))
KERNEL(k_and_or_b32, I8("v_and_or_b32", ", %12, %13"))
KERNEL(k_and_b32, I8("v_and_b32", ", %12"))
KERNEL(k_and_b32_lit, "v_and_b32 %0, 0xffc, %0\nv_and_b32 %1, 0xffc, %1\nv_and_b32 %2, 0xffc, %2\nv_and_b32 %3, 0xffc, %3\n"
                      "v_and_b32 %4, 0xffc, %4\nv_and_b32 %5, 0xffc, %5\nv_and_b32 %6, 0xffc, %6\nv_and_b32 %7, 0xffc, %7\n")
KERNEL(k_ashrrev_i32, "v_ashrrev_i32 %0, 2, %0\nv_ashrrev_i32 %1, 2, %1\nv_ashrrev_i32 %2, 2, %2\nv_ashrrev_i32 %3, 2, %3\n"
                      "v_ashrrev_i32 %4, 2, %4\nv_ashrrev_i32 %5, 2, %5\nv_ashrrev_i32 %6, 2, %6\nv_ashrrev_i32 %7, 2, %7\n")
KERNEL(k_bfe_u32, I8("v_bfe_u32", ", 12, 10"))
KERNEL(k_bfi_b32, I8("v_bfi_b32", ", %12, %13"))
KERNEL(k_add_u32, I8("v_add_u32", ", %12"))
KERNEL(k_add3_u32, I8("v_add3_u32", ", %12, %13"))
KERNEL(k_perm_b32, I8("v_perm_b32", ", %12, %13"))
KERNEL(k_alignbit_b32, I8("v_alignbit_b32", ", %12, 7"))
KERNEL(k_dot4_u32_u8, I8("v_dot4_u32_u8", ", %12, %13"))
KERNEL(k_bcnt, I8("v_bcnt_u32_b32", ", %12"))
KERNEL(k_cndmask, I8("v_cndmask_b32", ", %12, vcc"))
KERNEL(k_max_f32, I8("v_max_f32", ", %12"))
KERNEL(k_floor_f32, "v_floor_f32 %0, %0\nv_floor_f32 %1, %1\nv_floor_f32 %2, %2\nv_floor_f32 %3, %3\n"
                    "v_floor_f32 %4, %4\nv_floor_f32 %5, %5\nv_floor_f32 %6, %6\nv_floor_f32 %7, %7\n")
KERNEL(k_lshrrev_b64, "v_lshrrev_b64 %8, 3, %8\nv_lshrrev_b64 %9, 3, %9\nv_lshrrev_b64 %10, 3, %10\nv_lshrrev_b64 %11, 3, %11\n"
                      "v_lshrrev_b64 %8, 3, %8\nv_lshrrev_b64 %9, 3, %9\nv_lshrrev_b64 %10, 3, %10\nv_lshrrev_b64 %11, 3, %11\n")
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %8, vcc, %12, %13, %8\nv_mad_u64_u32 %9, vcc, %12, %13, %9\nv_mad_u64_u32 %10, vcc, %12, %13, %10\n"
                      "v_mad_u64_u32 %11, vcc, %12, %13, %11\nv_mad_u64_u32 %8, vcc, %12, %13, %8\nv_mad_u64_u32 %9, vcc, %12, %13, %9\n"
                      "v_mad_u64_u32 %10, vcc, %12, %13, %10\nv_mad_u64_u32 %11, vcc, %12, %13, %11\n")
KERNEL(k_mov_b32, "v_mov_b32 %0, %12\nv_mov_b32 %1, %12\nv_mov_b32 %2, %12\nv_mov_b32 %3, %12\n"
                  "v_mov_b32 %4, %12\nv_mov_b32 %5, %12\nv_mov_b32 %6, %12\nv_mov_b32 %7, %12\n")
KERNEL(k_snop, "s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\n")
KERNEL(k_pk_add_then_med3, "v_pk_add_f32 %8, %8, %9\nv_med3_f32 %0, %0, %12, %13\nv_pk_add_f32 %9, %9, %9\nv_med3_f32 %1, %1, %12, %13\n"
                           "v_pk_add_f32 %10, %10, %9\nv_med3_f32 %2, %2, %12, %13\nv_pk_add_f32 %11, %11, %9\nv_med3_f32 %3, %3, %12, %13\n")

typedef void (*kern_t)(unsigned*, unsigned long long*, float);
struct Entry { const char* name; kern_t k; };
#define E(n) {#n, n}
static Entry entries[] = {E(k_add_f32), E(k_add_f32_s), E(k_fma_f32), E(k_fma_f32_s), E(k_pk_add_f32), E(k_pk_fma_f32),
                          E(k_pk_mul_f32), E(k_med3_f32), E(k_cvt_flr), E(k_cvt_rpi), E(k_sub_u32), E(k_bfe_i32_v), E(k_cvt_f32_u32), E(k_cvt_f32_ubyte0),
                          E(k_mad_i32_i24), E(k_mad_u32_u24), E(k_mul_u32_u24), E(k_mul_lo_u32), E(k_lshl_add_u32),
                          E(k_add_lshl_u32), E(k_lshl_or_b32), E(k_and_or_b32), E(k_and_b32), E(k_and_b32_lit),
                          E(k_ashrrev_i32), E(k_bfe_u32), E(k_bfi_b32), E(k_add_u32), E(k_add3_u32), E(k_perm_b32),
                          E(k_alignbit_b32), E(k_dot4_u32_u8), E(k_bcnt), E(k_cndmask), E(k_max_f32), E(k_floor_f32),
                          E(k_lshrrev_b64), E(k_mad_u64_u32), E(k_mov_b32), E(k_snop), E(k_pk_add_then_med3)};

int main(int argc, char** argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = one per SIMD of a CU
  unsigned* out;
  unsigned long long* cyc;
  hipMalloc(&out, sizeof(unsigned) * blocks * 256);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  std::vector<unsigned long long> h(blocks);
  printf("# %s, %d CUs, %d waves per SIMD, %d x 32 instructions per wave\n", prop.name, cus, waves_per_simd, ITER);
  printf("# instruction                cycles per wave-instruction per SIMD   (ms per launch)\n");
  for (const Entry& e : entries) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);   // warm-up
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double tot = 0;
    for (int i = 0; i < blocks; i++) tot += (double)h[i];
    // a wave's elapsed cycles cover ITER*32 of its own instructions while waves_per_simd waves share the SIMD
    const double per = tot / blocks / ((double)ITER * 32) / waves_per_simd;
    printf("%-28s %8.2f   (%.3f ms)\n", e.name + 2, per, ms);
  }
  return 0;
}
