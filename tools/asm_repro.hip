// tools/asm_repro.hip — minimal reproduction harness for the inline-assembly address chain of score_polar_su_kernel:
// every lane recomputes the chain in C++ and counts mismatches.
//   hipcc --offload-arch=gfx950 -O3 tools/asm_repro.hip -o tools/_bin/asm_repro && tools/_bin/asm_repro [variant]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int VARIANT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ tab, const float* __restrict__ offs, int nsteps,
                                         float rmaxf, float cmaxf, int krow4, int kconst, unsigned* __restrict__ bad,
                                         unsigned* __restrict__ first) {
  const int lane = threadIdx.x;
  const int gid = blockIdx.x * 256 + lane;
  const v2f offv = {offs[2 * gid], offs[2 * gid + 1]};
  typedef const float __attribute__((address_space(4))) * cf;
  const cf tb = (cf)tab;
  unsigned nbad = 0;
  const uint64_t half2 = 0x3EFFFFFF3EFFFFFFull;
  for (int rep = 0; rep < 4; rep++) {
    unsigned toff = 0;
    unsigned acc_asm = 0;
    unsigned nleft = nsteps - 1;
    // asm: for every step, mask offset of 4 samples, XOR-accumulated
    asm volatile(
        "v_mov_b32 v41, %[kconst]\n"
        "v_readfirstlane_b32 s65, %[toff]\n"
        "v_readfirstlane_b32 s67, %[nleft]\n"
        ".Lstep%=:\n"
        "s_load_dwordx8 s[40:47], %[tb], s65\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_pk_add_f32 v[8:9], %[offv], s[40:41]\n"
        "v_pk_add_f32 v[10:11], %[offv], s[42:43]\n"
        "v_pk_add_f32 v[12:13], %[offv], s[44:45]\n"
        "v_pk_add_f32 v[14:15], %[offv], s[46:47]\n"
        "s_nop 1\n"
        "v_med3_f32 v8, v8, %[rmax], -1.0\n"
        "v_med3_f32 v9, v9, %[cmax], -1.0\n"
        "v_med3_f32 v10, v10, %[rmax], -1.0\n"
        "v_med3_f32 v11, v11, %[cmax], -1.0\n"
        "v_med3_f32 v12, v12, %[rmax], -1.0\n"
        "v_med3_f32 v13, v13, %[cmax], -1.0\n"
        "v_med3_f32 v14, v14, %[rmax], -1.0\n"
        "v_med3_f32 v15, v15, %[cmax], -1.0\n"
        "s_nop 1\n"
        "v_pk_add_f32 v[8:9], v[8:9], %[half]\n"
        "v_pk_add_f32 v[10:11], v[10:11], %[half]\n"
        "v_pk_add_f32 v[12:13], v[12:13], %[half]\n"
        "v_pk_add_f32 v[14:15], v[14:15], %[half]\n"
        "s_nop 1\n"
        "v_cvt_flr_i32_f32 v8, v8\n"
        "v_cvt_flr_i32_f32 v9, v9\n"
        "v_cvt_flr_i32_f32 v10, v10\n"
        "v_cvt_flr_i32_f32 v11, v11\n"
        "v_cvt_flr_i32_f32 v12, v12\n"
        "v_cvt_flr_i32_f32 v13, v13\n"
        "v_cvt_flr_i32_f32 v14, v14\n"
        "v_cvt_flr_i32_f32 v15, v15\n"
        "s_nop 1\n"
        "v_mad_i32_i24 v16, v8, %[krow4], v41\n"
        "v_mad_i32_i24 v17, v10, %[krow4], v41\n"
        "v_mad_i32_i24 v18, v12, %[krow4], v41\n"
        "v_mad_i32_i24 v19, v14, %[krow4], v41\n"
        "v_ashrrev_i32 v20, 5, v9\n"
        "v_ashrrev_i32 v21, 5, v11\n"
        "v_ashrrev_i32 v22, 5, v13\n"
        "v_ashrrev_i32 v23, 5, v15\n"
        "s_nop 1\n"
        "v_lshl_add_u32 v16, v20, 2, v16\n"
        "v_lshl_add_u32 v17, v21, 2, v17\n"
        "v_lshl_add_u32 v18, v22, 2, v18\n"
        "v_lshl_add_u32 v19, v23, 2, v19\n"
        "s_nop 1\n"
        "v_xor_b32 %[acc], %[acc], v16\n"
        "v_xor_b32 %[acc], %[acc], v17\n"
        "v_xor_b32 %[acc], %[acc], v18\n"
        "v_xor_b32 %[acc], %[acc], v19\n"
        "s_add_u32 s65, s65, 32\n"
        "s_sub_u32 s67, s67, 1\n"
        "s_cbranch_scc0 .Lstep%=\n"
        : [acc] "+v"(acc_asm)
        : [offv] "v"(offv), [krow4] "v"(krow4), [tb] "s"(tb), [rmax] "v"(rmaxf), [cmax] "v"(cmaxf), [half] "s"(half2),
          [kconst] "s"(kconst), [toff] "v"(toff), [nleft] "v"(nleft)
        : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",
          "v41", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s65", "s67", "vcc", "memory");
    // the same in C++
    unsigned acc_c = 0;
    for (int s = 0; s < nsteps; s++) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        v2f pv = {tb[8 * s + 2 * u], tb[8 * s + 2 * u + 1]};
        pv = pv + offv;
        v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
        qv = qv + 0.49999997f;
        const int ri = (int)__builtin_floorf(qv.x), ci = (int)__builtin_floorf(qv.y);
        acc_c ^= (unsigned)(ri * krow4 + kconst + (ci >> 5) * 4);
      }
    }
    if (acc_c != acc_asm) {
      nbad++;
      atomicMin(first, (unsigned)gid);
    }
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 4096, nsteps = 64;
  std::vector<float> tab(8 * nsteps), offs((size_t)blocks * 256 * 2);
  srand(1);
  for (auto& t : tab) t = (rand() % 51200) / 100.f - 256.f;
  for (auto& o : offs) o = (rand() % 440000) / 100.f - 200.f;
  float *dtab, *doffs;
  unsigned *dbad, *dfirst;
  (void)hipMalloc(&dtab, tab.size() * 4);
  (void)hipMalloc(&doffs, offs.size() * 4);
  (void)hipMalloc(&dbad, 4);
  (void)hipMalloc(&dfirst, 4);
  (void)hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(doffs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice);
  for (int it = 0; it < 8; it++) {
    unsigned zero = 0, big = 0xFFFFFFFFu, bad = 0, first = 0;
    (void)hipMemcpy(dbad, &zero, 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dfirst, &big, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, dtab, doffs, nsteps, 4000.f, 4000.f, 508, 512, dbad, dfirst);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&first, dfirst, 4, hipMemcpyDeviceToHost);
    printf("launch %d: %u mismatching (lane, repetition) pairs of %d, first lane %u\n", it, bad, blocks * 256 * 4, first);
  }
  return 0;
}
