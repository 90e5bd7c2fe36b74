// tdr_score_dev.h — device helpers shared by the scoring kernels (tdr_score.hip, tdr_score_su.hip).
#ifndef TDR_SCORE_DEV_H_
#define TDR_SCORE_DEV_H_
#include "tdr_common.h"

__device__ __forceinline__ int rot_shift_dev(float rot, int nb) {
  // state_particle.cpp:124-128
  int s = (int)round((double)(rot * (float)nb / 2) / M_PI);
  s %= nb;
  if (s < 0) s += nb;
  return s;
}

// roundf (half away from zero) of a coordinate already clamped to [-1, limit], as an int, in two VALU ops:
//     roundf(x) == floor(fl(x + (0.5 - 2^-25)))   for every float x in [-1, 2^23]
// (the float addition's own rounding lands exact .5 ties on the next integer and everything below them under it;
// checked exhaustively on the CPU over [-1, 8] and on the GPU by tests/test_gpu_parity.py).  The generic expansion
// of roundf costs seven.
__device__ __forceinline__ int round_half_away_clamped(float x) {
  const float y = x + 0.49999997f;
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(y));
  return r;
}

// ---- compact records (tdr_cmap.hip): geometry, load, decode — shared by the polar and the Cartesian kernel ------------
template <int RF, bool KSLOT>
struct CmapShape {
  static constexpr int ND = KSLOT ? RF - 2 : RF - 1;                               // distance slots of a record
  static constexpr int CW = (ND + 2) / 3 <= 1 ? 1 : ((ND + 2) / 3 == 2 ? 2 : 4);   // dwords of a compact record
  static constexpr int LC = CW == 1 ? 3 : (CW == 2 ? 2 : 1);                       // tile = (1 << LC) rows x 4 columns
};
// Byte offset of cell (ri, ci), ri in [-1, rows], ci in [-1, cols] (the clamped sample coordinate): it lives in tile
// ((ri >> LC) + 1, (ci >> 2) + 1) — tiles ordered column by column, tiles_r per column — at record
// (ri & (2^LC - 1)) * 4 + (ci & 3) (tdr_cmap.hip).  Cells outside the map are guard records (distance 0, unknown).
// With RB = 4 CW record bytes, CS = 128 tiles_r bytes per tile column and 2^LC * 4 RB = 128 the offset is separable and
// its row part linear:
//     128 * tile + RB * ((ri & ..) * 4 + (ci & 3)) = (ci >> 2) * (CS - 4 RB) + ci * RB + ri * 4 RB + (CS + 128)
// ckcol = CS - 4 RB, ckconst = CS + 128: four integer ops.
template <int CW, int LC>
__device__ __forceinline__ unsigned cmap_offset(int ri, int ci, int ckcol, int ckconst) {
  int t1, t2, off;
  const int cq = ci >> 2;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(cq), "s"(ckcol), "v"(ckconst));   // one SGPR operand at most
  asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(t2) : "v"(ci), "n"(CW == 1 ? 2 : (CW == 2 ? 3 : 4)), "v"(t1));
  asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(off) : "v"(ri), "n"(CW == 1 ? 4 : (CW == 2 ? 5 : 6)), "v"(t2));
  return (unsigned)off;
}
template <int CW>
__device__ __forceinline__ void cmap_load(const char* __restrict__ crecb, unsigned off, uint32_t (&w)[CW]) {
  if constexpr (CW == 1) {
    w[0] = *reinterpret_cast<const uint32_t*>(crecb + off);
  } else if constexpr (CW == 2) {
    const uint2 v = *reinterpret_cast<const uint2*>(crecb + off);
    w[0] = v.x; w[1] = v.y;
  } else {
    const uint4 v = *reinterpret_cast<const uint4*>(crecb + off);
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
  }
}
// ---- the known mask behind the compact tiles (tdr_cmap.hip): 1 bit per cell in 32 x 32-cell tiles of 32 words — 128 bytes,
// one cache line: lanes whose cells are a few cells apart, in whatever direction, read the same line.  With r' = r + 32
// and c' = c + 32 (a guard band of unknown cells around the map) cell (r, c), r in [-1, rows], c in [-1, cols], is bit c & 31
// of word r' & 31 of tile (r' >> 5, c' >> 5); the tiles are stored COLUMN by column, kmask_trows(rows) per tile column, so
// the words of one tile column are simply its rows in order: word index = (c' >> 5) * 32 kmask_trows + r'.
__host__ __device__ inline int kmask_tcols(int cols) { return (cols >> 5) + 2; }
__host__ __device__ inline int kmask_trows(int rows) { return (rows >> 5) + 2; }
// Byte offset of the cell's word: col_bytes = 128 * kmask_trows (a tile column), kconst = (byte offset of the mask) +
// col_bytes + 128 — the two guard bands folded in: (c + 32) >> 5 = (c >> 5) + 1 and 4 (r + 32) = 4 r + 128.  Three integer
// ops (the row-by-row order of the tiles this replaced took seven).
__device__ __forceinline__ unsigned kmask_offset(int ri, int ci, int col_bytes, int kconst) {
  int off;
  const int cq = ci >> 5;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(off) : "v"(cq), "s"(col_bytes), "v"(kconst));
  asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(off) : "v"(ri), "v"(off));
  return (unsigned)off;
}

// ---- the class planes behind the mask (tdr_cmap.hip): per class a 16-bit value per cell (dictionary index << 2 | known << 15)
// in tiles of 8 x 8 cells (128 bytes), tiles column by column, plane_trows per tile column, a guard band of 8 cells.  With
// r' = r + 8, c' = c + 8 the cell's byte offset inside its plane is (c' >> 3) * CS + r' * 16 + (c' & 7) * 2, CS = 128
// plane_trows; as the row part is linear:  (c >> 3) * (CS - 16) + 2 c + 16 r + (CS + 128).
__host__ __device__ inline int plane_trows(int rows) { return (rows >> 3) + 2; }
__host__ __device__ inline int plane_tcols(int cols) { return (cols >> 3) + 2; }
__device__ __forceinline__ unsigned plane_offset(int ri, int ci, int pkcol, int pkconst) {   // pkcol = CS - 16, pkconst = base + CS + 128
  int t1, off;
  const int cq = ci >> 3;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(cq), "s"(pkcol), "v"(pkconst));
  asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(t1) : "v"(ci), "v"(t1));
  asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(off) : "v"(ri), "v"(t1));
  return (unsigned)off;
}

// The two device words that decide between the integer and the float form of a launch (ray_prep_kernel writes them):
// flags[0] != 0: a scan count or the dictionary has no integer form; flags[1]: an upper bound of (the scan's total count) /
// 256 — below 2^24 the total stays below 2^32, the normalisation sums fit 32 bits and the class sums 64.
__device__ __forceinline__ bool int_form_off(const int32_t* flags) {
  return flags[0] != 0 || (uint32_t)flags[1] >= (1u << 24);
}

// one compact record -> the RF operands the dense record would have delivered, bit for bit (ldict: the dictionary in LDS)
template <int RF, bool KSLOT>
__device__ __forceinline__ void cmap_decode(const uint32_t (&w)[CmapShape<RF, KSLOT>::CW], const float* ldict, float (&m)[RF]) {
  constexpr int ND = CmapShape<RF, KSLOT>::ND, CW = CmapShape<RF, KSLOT>::CW;
#pragma unroll
  for (int k = 0; k < ND; k++) {
    const uint32_t ww = w[k / 3];
    const int sh = 10 * (k % 3);                        // field at bits [2 + sh, 12 + sh): (ww >> sh) & 0xFFC = index * 4
    const uint32_t boff = sh ? ((ww >> sh) & 0xFFCu) : (ww & 0xFFCu);
    m[k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ldict) + boff);
  }
  const float kf = (float)(w[CW - 1] & 1u);
  if (KSLOT) m[RF - 2] = kf;
  m[RF - 1] = kf;
}

// the WIDE compact record (tdr_cmap.hip): 16-bit fields, two per dword, a dictionary of up to 4096 values
template <int RF, bool KSLOT>
__device__ __forceinline__ void cmap_decode_wide(const uint32_t (&w)[4], const float* ldict, float (&m)[RF]) {
  constexpr int ND = CmapShape<RF, KSLOT>::ND;
  static_assert(ND <= 7, "wide records hold up to seven distances");
#pragma unroll
  for (int k = 0; k < ND; k++) {
    const uint32_t boff = (k & 1) ? (w[k / 2] >> 16) : (w[k / 2] & 0xFFFCu);   // index * 4
    m[k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ldict) + boff);
  }
  const float kf = (float)(w[3] & 1u);
  if (KSLOT) m[RF - 2] = kf;
  m[RF - 1] = kf;
}
#endif  // TDR_SCORE_DEV_H_
