#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// does VGPR relative indexing (s_set_gpr_idx_on) work on gfx950 for the 64-bit destination / addend of v_mad_u64_u32?
__global__ void k(const uint32_t* code, const uint32_t* val, uint64_t* out) {
  // six 64-bit accumulators in FIXED registers v[42:53]
  asm volatile(
      "v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"
      "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n" ::: "v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53");
  for (int e = 0; e < 8; e++) {
    uint32_t c = __builtin_amdgcn_readfirstlane(code[e]);   // class 0..5
    uint32_t v = __builtin_amdgcn_readfirstlane(val[e]);
    uint32_t m = threadIdx.x + 1;
    uint32_t idx = 2 * c;
    asm volatile(
        "s_set_gpr_idx_on %0, 0xC\n"          // bits: SRC0 1, SRC1 2, SRC2 4, DST 8 -> SRC2 | DST
        "v_mad_u64_u32 v[42:43], vcc, %1, %2, v[42:43]\n"
        "s_set_gpr_idx_off\n"
        :: "s"(idx), "s"(v), "v"(m) : "vcc", "m0", "v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53");
  }
  uint32_t lo[6], hi[6];
  asm volatile("v_mov_b32 %0, v42\n v_mov_b32 %1, v43\n v_mov_b32 %2, v44\n v_mov_b32 %3, v45\n v_mov_b32 %4, v46\n v_mov_b32 %5, v47\n"
               : "=v"(lo[0]), "=v"(hi[0]), "=v"(lo[1]), "=v"(hi[1]), "=v"(lo[2]), "=v"(hi[2]));
  asm volatile("v_mov_b32 %0, v48\n v_mov_b32 %1, v49\n v_mov_b32 %2, v50\n v_mov_b32 %3, v51\n v_mov_b32 %4, v52\n v_mov_b32 %5, v53\n"
               : "=v"(lo[3]), "=v"(hi[3]), "=v"(lo[4]), "=v"(hi[4]), "=v"(lo[5]), "=v"(hi[5]));
  for (int c = 0; c < 6; c++) out[threadIdx.x * 6 + c] = ((uint64_t)hi[c] << 32) | lo[c];
}
int main() {
  uint32_t code[8] = {0, 5, 3, 3, 1, 0, 4, 2}, val[8] = {3, 7, 0x80000000u, 0x80000000u, 9, 100, 11, 13};
  uint32_t *dc, *dv; uint64_t* dout;
  hipMalloc(&dc, 32); hipMalloc(&dv, 32); hipMalloc(&dout, 64 * 6 * 8);
  hipMemcpy(dc, code, 32, hipMemcpyHostToDevice); hipMemcpy(dv, val, 32, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dc, dv, dout);
  uint64_t out[64 * 6];
  if (hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost) != hipSuccess) { puts("hip error"); return 1; }
  int bad = 0;
  for (int t = 0; t < 64; t++) {
    uint64_t want[6] = {0, 0, 0, 0, 0, 0};
    for (int e = 0; e < 8; e++) want[code[e]] += (uint64_t)val[e] * (uint64_t)(t + 1);
    for (int c = 0; c < 6; c++) bad += out[t * 6 + c] != want[c];
  }
  printf("gpr-idx accumulate: %d mismatches; lane 63 class 3 = %llu\n", bad, (unsigned long long)out[63 * 6 + 3]);
  return bad != 0;
}
