// tdr_score_cart.hip — the Cartesian scoring kernel that skips what an empty scan bin does not need.
//
// score_cart_kernel (tdr_score.hip) gives every window sample the full treatment — an 8/16-byte record gather, the decode of
// every class, one FMA per record slot — although the scan operand of sample (i, j) is the same for every lane (the
// rotation lives in the sampling: src/top_down_map.cpp:367-389, 429-459) and a Cartesian render of a LiDAR scan is mostly
// empty.  Here the scan side of a sample is a four-dword DESCRIPTOR read through the scalar cache (cart_prep_kernel, the
// same encoding as the polar shift-uniform kernel's, tdr_score_su.hip), and what a sample costs is a wave-uniform choice:
//   empty bin        coordinates + ONE 4-byte gather from the map's known mask (1 bit per cell in 32 x 32-cell tiles: 1024
//                    cells per cache line instead of 16) — no record, no decode, no FMA
//   one class        coordinates + ONE dword of the compact record (it carries the known bit too), one decode, 2 FMAs
//   several classes  the whole record against the packed scan record, classes with a zero count skipped
// Skipping an FMA whose scan operand is zero leaves the accumulator unchanged bit for bit (finite operands; a non-finite
// dictionary or scan value turns the skipping off for the bin: SU_CODE_FULL_ALL), and the samples are visited in the order
// of score_cart_kernel — blocks of 4 window rows, the chunk's columns inside — so both kernels produce IDENTICAL partial sums
// (tests/test_gpu_parity.py::test_cart_skip_kernel_equals_general_kernel).
//
// Compiled with -mllvm -structurizecfg-skip-uniform-regions like tdr_score_su.hip (wave-uniform branch trees).
#include <type_traits>

#include "tdr_score_cart.h"
#include "tdr_score_dev.h"
#include "tdr_score_su.h"   // the ordering passes (tdr_su_order) and the ray order's helpers
#include "tdr_sincosf.h"

#define CART_CODE_FULL 0xFFu       // several classes present
#define CART_CODE_FULL_ALL 0xFEu   // a non-finite value in play: every class multiplied
#define CART_U 4                   // window rows per step (TDR_SCORE_U of score_cart_kernel: the order of the sums)

// Descriptor of bin (j, i) (window column j, row i; scan_pk is [cols][rows][rf]), four dwords:
//   [0] code: 0 = every class zero; c + 1 = class c alone; CART_CODE_FULL / CART_CODE_FULL_ALL
//   [1] the bin's sum over the classes (slot rf - 1 of the packed record) — for a single class: its value
//   [2] cmap_offset's constant advanced to the dword the class lives in: ckconst + 4 * (c / 3)
//   [3] bit offset of the class's field in that dword minus 2 (10 * (c % 3))
// as_int: descriptor word [1] holds the count as an integer (the integer form's kernels) instead of float bits
// pbase != 0 (the integer form's dense kernel): word [2] of a bin with ONE class is plane_offset's constant for the class's
// plane instead (pbase + c * plane_bytes, a byte offset from crec) — 8 x 8-cell tiles of 2-byte cells: a wave whose
// particles lie a few cells apart touches a third of the lines the 4 x 4-cell record tiles cost it (tdr_score_su.hip)
// full_list != NULL (the descriptors score_cart_su_kernel reads): a bin holding SEVERAL classes gets code 0 — an empty bin to
// the sample loop, which still counts its known bit — and goes on the list of its column chunk (chunks of cpc_su columns,
// list_cap entries each) as row << 16 | column: the kernel adds those bins' products afterwards, in any order (exact sums).
__global__ __launch_bounds__(256) void cart_prep_kernel(const float* __restrict__ scan_pk, int rows, int cols, int rf, int ncls,
                                                        int ckconst, const float* __restrict__ dict, int dict_n,
                                                        uint32_t* __restrict__ desc, int as_int, unsigned pbase,
                                                        unsigned plane_bytes, uint32_t* __restrict__ full_list = nullptr,
                                                        int32_t* __restrict__ full_cnt = nullptr, int cpc_su = 1,
                                                        int64_t list_cap = 0, uint32_t* __restrict__ bdesc = nullptr) {
  bool bad = false;   // the dictionary is small: every workgroup checks it for itself
  for (int k = threadIdx.x; k < dict_n; k += blockDim.x) bad |= !(fabsf(dict[k]) <= 3.402823466e+38f);
  const bool dict_bad = __syncthreads_or(bad);
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)rows * cols) return;
  auto classify = [&](int64_t bin, float& val, uint32_t& ckc, uint32_t& sh) -> uint32_t {
    const float* r = scan_pk + bin * rf;
    int nz = 0, first = 0;
    bool finite = true;
    for (int c = 0; c < ncls; c++) {
      finite &= fabsf(r[c]) <= 3.402823466e+38f;
      if (r[c] != 0.f) {
        if (!nz) first = c;
        nz++;
      }
    }
    val = 0.f; ckc = (uint32_t)ckconst; sh = 0;
    if (dict_bad || !finite) { val = r[rf - 1]; return CART_CODE_FULL_ALL; }
    if (nz == 1) {
      val = r[first];
      if (pbase) ckc = pbase + (uint32_t)first * plane_bytes;
      else { ckc += 4u * (uint32_t)(first / 3); sh = 10u * (uint32_t)(first % 3); }
      return (uint32_t)first + 1u;
    }
    if (nz > 1) { val = r[rf - 1]; return CART_CODE_FULL; }
    return 0u;
  };
  float val;
  uint32_t ckc, sh;
  uint32_t code = classify(t, val, ckc, sh);
  if (full_list && code >= CART_CODE_FULL_ALL) {
    const int j = (int)(t / rows), i = (int)(t - (int64_t)j * rows);   // scan_pk is [cols][rows][rf]
    const int chunk = j / cpc_su;
    full_list[(int64_t)chunk * list_cap + atomicAdd(&full_cnt[chunk], 1)] = ((uint32_t)i << 16) | (uint32_t)j;
    code = 0;
    val = 0.f;
  }
  if (bdesc) {
    // The same bins once more in the BLOCK layout the generated loop streams (tools/gen_cart_asm.py): one dword per bin —
    // count in bits 0-23, class code in bits 24-26 — at [(column group * row blocks + row block) * 32 + (column & 7) * 4 +
    // (row & 3)], i.e. half a block (4 rows x 4 columns) is one 64-byte scalar load.  A step's first dword also says which
    // of its four bins hold a class (bits 28-31), a block's first whether any of its 32 does (bit 27).  The array is zeroed
    // before this kernel and only OR-ed into: the flags come from other threads.  (rows is a multiple of 4 here.)
    const int j = (int)(t / rows), i = (int)(t - (int64_t)j * rows);
    const int64_t blk = ((int64_t)(j >> 3) * (rows >> 2) + (i >> 2)) * 32;
    const int jc = j & 7, u = i & 3;
    if (code != 0) {
      atomicOr(&bdesc[blk + jc * 4 + u], (code << 24) | ((uint32_t)val & 0xFFFFFFu));
      atomicOr(&bdesc[blk + jc * 4], 1u << (28 + u));
      atomicOr(&bdesc[blk], 1u << 27);
    }
  }
  desc[4 * t] = code;
  desc[4 * t + 1] = as_int ? (uint32_t)val : __float_as_uint(val);
  desc[4 * t + 2] = ckc;
  desc[4 * t + 3] = sh;
}

__device__ __forceinline__ float cart_linspaced(int i, int size1, float low, float high, float step) {
  // Eigen LinSpaced<float>, |high| == |low| here, so never the flipped branch of linspaced_op_impl
  return (i == size1) ? high : (low + (float)i * step);
}

// lane = particle; grid.y = chunk of a.cpc window columns; the sample order and the partition into partial sums are
// score_cart_kernel's.
// INT: the integer form — scan counts times the dictionary's integers (tdr_cmap.hip), accumulated in 64 bits: exact, so the
// sums equal score_cart_ray_kernel's whatever the order (see tdr_score_su.hip); the slot list may hold padding (-1).
template <int NV4, bool KSLOT, bool INT = false>
__global__ __launch_bounds__(256) void score_cart_skip_kernel(CartArgs a) {
  constexpr int RF = 4 * NV4;
  constexpr int ND = CmapShape<RF, KSLOT>::ND, CW = CmapShape<RF, KSLOT>::CW, LC = CmapShape<RF, KSLOT>::LC;
  __shared__ uint32_t ldict[TDR_CMAP_MAX_DICT];   // the dictionary: float bits, or (INT) the integers
  if (a.flags && int_form_off(a.flags) == (a.run_if_int != 0)) return;   // (uniform) the other form does this launch
  for (int t = threadIdx.x; t < a.dict_n; t += 256) ldict[t] = INT ? a.dict_int[t] : __float_as_uint(a.dict[t]);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t slot = ((int64_t)blockIdx.x * 4 + wave) * 64 + lane;
  const int64_t nact = a.count ? (int64_t)*a.count : a.n;
  if ((int64_t)blockIdx.x * 256 >= nact) return;
  const bool valid = slot < nact;
  int64_t p = a.order ? (int64_t)a.order[valid ? slot : 0] : (valid ? slot : 0);
  const bool real = valid && p >= 0;
  if (p < 0) p = a.order[((int64_t)blockIdx.x * 4 + wave) * 64];   // padding of the slot list: a wave's first slot never is
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];
  const float theta = a.st[TDR_ST_THETA * a.cap + p];
  const float off0 = cy / a.resolution;  // samplePts(center/resolution, ...): x_vals += center[1] (top_down_map.cpp:387)
  const float off1 = cx / a.resolution;  // y_vals += center[0] (:388)
  const float resq = (a.res * scale) / a.resolution;  // res/params_.resolution (:434)
  // cos(rot), sin(rot) of top_down_map.cpp:381-385 = the host libm's cosf / sinf, bit for bit (tdr_sincosf.h)
  const float c = tdr_libm::cosf_v(theta, a.libm_fma), s = tdr_libm::sinf_v(theta, a.libm_fma);
  const float ns = -s;
  typedef float tdr_v2f __attribute__((ext_vector_type(2)));
  const tdr_v2f cs = {c, s}, offv = {off0, off1};
  const float lo_r = (float)((double)(-resq * (float)(a.rows - 1)) / 2.), hi_r = (float)((double)(resq * (float)(a.rows - 1)) / 2.);
  const float lo_c = (float)((double)(-resq * (float)(a.cols - 1)) / 2.), hi_c = (float)((double)(resq * (float)(a.cols - 1)) / 2.);
  const float step_r = a.rows == 1 ? 0.f : (hi_r - lo_r) / (float)(a.rows - 1);
  const float step_c = a.cols == 1 ? 0.f : (hi_c - lo_c) / (float)(a.cols - 1);
  const int r1 = a.rows == 1 ? 1 : a.rows - 1, c1 = a.cols == 1 ? 1 : a.cols - 1;

  const int j0 = blockIdx.y * a.cpc, j1 = min(a.cols, j0 + a.cpc);
  const float rmaxf = (float)a.map_rows, cmaxf = (float)a.map_cols;
  const uint32_t* __restrict__ crec = a.crec;
  const int ckcol = a.ctiles_r * 128 - 16 * CW, ckconst = a.ctiles_r * 128 + 128;   // cmap_offset
  const int mrow = a.kmask_row, mconst = (int)a.kmask_off + a.kmask_row + 128;       // kmask_offset
  typedef const float __attribute__((address_space(4))) * tdr_const_f;
  typedef const uint32_t __attribute__((address_space(4))) * tdr_const_u;
  const tdr_const_f scanc = (tdr_const_f)a.scan_pk;
  const tdr_const_u descc = (tdr_const_u)a.desc;

  auto field = [&](const uint32_t (&w)[CW], int k) -> uint32_t {   // distance k of a compact record (cmap_decode, one field)
    const uint32_t ww = w[k / 3];
    const int sh = 10 * (k % 3);
    const uint32_t boff = sh ? ((ww >> sh) & 0xFFCu) : (ww & 0xFFCu);
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ldict) + boff);
  };
  auto field1 = [&](uint32_t ww, int k) -> uint32_t {   // ... when the sample loaded only the dword class k lives in
    const int sh = 10 * (k % 3);
    const uint32_t boff = sh ? ((ww >> sh) & 0xFFCu) : (ww & 0xFFCu);
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ldict) + boff);
  };
  typedef typename std::conditional<INT, unsigned long long, float>::type acc_t;
  acc_t acc[ND];
#pragma unroll
  for (int k = 0; k < ND; k++) acc[k] = 0;
  float norm = 0.f;       // float form
  uint32_t inorm = 0;     // integer form
  uint32_t known = 0;
  // v: the bin's value as the descriptor carries it — float bits, or (INT) the count as an integer
  const bool planes = INT && a.pkcol != 0;   // (uniform) a bin with one class reads the class's plane
  const int pkcol = a.pkcol;
  auto single_class = [&](uint32_t cd, uint32_t v, uint32_t ww) {   // a switch over a wave-uniform value
    switch (cd) {
#define CART_CASE(K)                                                                                \
  case K + 1:                                                                                       \
    if constexpr (K < ND) {                                                                         \
      /* (a plane's cell: the dictionary index * 4 in bits 2..11, whatever the class) */            \
      const uint32_t m = field1(ww, planes ? 0 : (K < ND ? K : 0));                                 \
      if constexpr (INT) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[K < ND ? K : 0]) : "s"(v), "v"(m) : "vcc"); \
      else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[K < ND ? K : 0]) : "s"(v), "v"(m));      \
    }                                                                                               \
    break;
      CART_CASE(0) CART_CASE(1) CART_CASE(2) CART_CASE(3) CART_CASE(4) CART_CASE(5)
      CART_CASE(6) CART_CASE(7) CART_CASE(8) CART_CASE(9) CART_CASE(10)
#undef CART_CASE
      default: break;
    }
  };
  // rotm * pts (:383-385) for window sample (row value yi, column terms AB = {-s * xj, c * xj}), centre added (:387-388),
  // rounded (:437) — the float operations of score_cart_kernel
  auto cell = [&](tdr_v2f cyi, tdr_v2f AB, int& ri, int& ci) {
    tdr_v2f pv = cyi + AB;         // p0 = c * yi + (-s * xj), p1 = s * yi + c * xj
    pv = pv + offv;
    tdr_v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
    qv = qv + 0.49999997f;         // see round_half_away_clamped
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ri) : "v"(qv.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ci) : "v"(qv.y));
  };
  auto column_terms = [&](int j) -> tdr_v2f {
    const float xj = cart_linspaced(j, c1, lo_c, hi_c, step_c);
    return (tdr_v2f){ns * xj, c * xj};
  };
  // NS samples of one window column (rows i .. i + NS - 1, descriptors D): one gather each — the mask word of an empty
  // bin, the record dword of a single class, dword 0 of a bin with several — all requested before the first is used
  auto samples = [&](auto ns_c, const tdr_v2f* cyi, tdr_v2f AB, tdr_const_u D, int64_t bin0) {
    constexpr int NS = decltype(ns_c)::value;
    uint32_t w[NS];
    int cis[NS];
    unsigned offs[NS];
#pragma unroll
    for (int u = 0; u < NS; u++) {
      int ri, ci;
      cell(cyi[u], AB, ri, ci);
      cis[u] = ci;
      if (D[4 * u] == 0) {   // wave-uniform
        offs[u] = kmask_offset(ri, ci, mrow, mconst);
      } else if (planes && D[4 * u] < CART_CODE_FULL_ALL) {   // one class: the 2-byte cell of its plane (plane_offset), read
        int t1, t2;                                           // as the low half of a dword at a 2-byte aligned address
        const int cq = ci >> 3;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(cq), "v"(pkcol), "s"(D[4 * u + 2]));
        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(t2) : "v"(ci), "v"(t1));
        asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(offs[u]) : "v"(ri), "v"(t2));
      } else {
        int t1, t2;
        const int cq = ci >> 2;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(cq), "v"(ckcol), "s"(D[4 * u + 2]));
        asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(t2) : "v"(ci), "n"(CW == 1 ? 2 : (CW == 2 ? 3 : 4)), "v"(t1));
        asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(offs[u]) : "v"(ri), "n"(CW == 1 ? 4 : (CW == 2 ? 5 : 6)), "v"(t2));
      }
      // a plain load: the compiler tracks the destination and places the waits (all NS requests are issued before the first
      // value is used below).  (Rounds 3-4 issued these through inline assembly with hand-counted s_waitcnt; a register
      // copy between a load and its wait gave wrong weights twice — DESIGN.md 5.1.)
      w[u] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(crec) + offs[u]);
    }
#pragma unroll
    for (int u = 0; u < NS; u++) {
      const uint32_t cd = D[4 * u];
      if (cd == 0) {   // wave-uniform
        int kmsk;      // 0 / -1: the cell's known bit
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(kmsk) : "v"(w[u]), "v"(cis[u]));
        known -= (uint32_t)kmsk;
      } else {
        // bit 0 of every dword of a compact record, bit 15 of a plane's cell (tdr_cmap.hip)
        const uint32_t kb = (planes && cd < CART_CODE_FULL_ALL) ? (w[u] >> 15) & 1u : w[u] & 1u;
        known += kb;
        const uint32_t vb = D[4 * u + 1];   // float bits (INT: the count as an integer — cart_prep_kernel writes both forms)
        if (cd < CART_CODE_FULL_ALL) {
          // the bin's sum x known (state_particle.cpp:141-142)
          if constexpr (INT) inorm += (0u - kb) & vb;
          else norm = norm + __uint_as_float((0u - kb) & vb);   // fma(val, 1 or 0, norm) for a finite val
          single_class(cd, vb, w[u]);
        } else {   // several classes (or a non-finite value in play): the whole record, the packed scan record
          uint32_t wr[CW];
          wr[0] = w[u];
#pragma unroll
          for (int d = 1; d < CW; d++)
            wr[d] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(crec) + 4 * d + offs[u]);
          const tdr_const_f S = scanc + (bin0 + u) * RF;
          if constexpr (INT) inorm += (0u - kb) & vb;
          else norm = __builtin_fmaf(__uint_as_float(vb), (float)kb, norm);
#pragma unroll
          for (int k = 0; k < ND; k++) {
            const float sk = S[k];
            if constexpr (INT) {
              if (sk != 0.f) acc[k] += (unsigned long long)(uint32_t)sk * field(wr, k);
            } else {
              if (cd == CART_CODE_FULL_ALL || sk != 0.f) acc[k] = __builtin_fmaf(sk, __uint_as_float(field(wr, k)), acc[k]);
            }
          }
        }
      }
    }
  };
  // Order (score_cart_kernel's): blocks of CART_U window rows, and within a block the chunk's columns one after the other
  int i = 0;
  for (; i + CART_U <= a.rows - 1; i += CART_U) {   // LinSpaced without the select of its last element; the last row: below
    tdr_v2f cyi[CART_U];
#pragma unroll
    for (int u = 0; u < CART_U; u++) cyi[u] = cs * (lo_r + (float)(i + u) * step_r);
    for (int j = j0; j < j1; j++) {
      const int64_t bin0 = (int64_t)j * a.rows + i;   // wave-uniform
      samples(std::integral_constant<int, CART_U>{}, cyi, column_terms(j), descc + bin0 * 4, bin0);
    }
  }
  for (; i < a.rows; i++) {   // the remaining rows (the last one among them), column by column
    const tdr_v2f cyi = cs * cart_linspaced(i, r1, lo_r, hi_r, step_r);
    for (int j = j0; j < j1; j++) {
      const int64_t bin0 = (int64_t)j * a.rows + i;
      samples(std::integral_constant<int, 1>{}, &cyi, column_terms(j), descc + bin0 * 4, bin0);
    }
  }
  if constexpr (INT) {
    if (real) {   // [chunk][2 ncls + 2][npad] words, like score_polar_su_kernel
      const int ncls = a.ncls;
      uint32_t* o = reinterpret_cast<uint32_t*>(a.part) + (int64_t)blockIdx.y * (2 * ncls + 2) * a.npad + slot;
#pragma unroll
      for (int k = 0; k < ND; k++)
        if (k < ncls) {
          o[(int64_t)(2 * k) * a.npad] = (uint32_t)acc[k];
          o[(int64_t)(2 * k + 1) * a.npad] = (uint32_t)((unsigned long long)acc[k] >> 32);
        }
      o[(int64_t)(2 * ncls) * a.npad] = inorm;
      o[(int64_t)(2 * ncls + 1) * a.npad] = known;
    }
  } else if (slot < a.npad) {
    float* o = a.part + (int64_t)blockIdx.y * (RF + 1) * a.npad + slot;
#pragma unroll
    for (int k = 0; k < ND; k++) o[(int64_t)k * a.npad] = (float)acc[k];
    o[(int64_t)(RF - 1) * a.npad] = norm;
    o[(int64_t)RF * a.npad] = (float)known;
  }
}


// =========================================================================================================================
// score_cart_su_kernel — the dense share of the Cartesian integer form for maps with two-dword compact records (4-6
// classes): score_cart_skip_kernel<.., INT>'s sums, with the sample loop in GENERATED assembly (tdr_score_cart_asm.h,
// tools/gen_cart_asm.py — loads and their waits inside one text, a planned register file, statically checked) and the
// known mask STAGED IN LDS per wave:
//   * lane = particle, grid.y = chunk of a.cpc window columns (a multiple of 8), walked in groups of 8 columns; a group in
//     SEGMENTS of x.seg_rows window rows;
//   * per (group, segment) a wave works out the box of map cells its 64 windows can reach (rounding is monotone and the
//     window coordinates are monotone along both window axes: the four corners bound it), stages the known mask of that box
//     in its own quarter of the LDS area — no workgroup barrier: a wave's LDS operations execute in order — and runs the
//     loop variant the box allows: every cell known (an EMPTY scan bin, 84 % of config 4's, then costs no instruction at
//     all) / every cell inside the map (no clamp) / general.  A box that does not fit the area (a large scale, particles
//     far apart) takes the plain C++ steps with the mask gathered from global memory — compiler-tracked loads;
//   * bins holding several classes are empty to the loop (cart_prep_kernel) and added from the chunk's list at the end.
// Integer sums: exact, so the sums equal score_cart_skip_kernel<.., INT>'s and score_cart_ray_kernel's whatever the order
// (tests/test_ray.py::test_cartesian_integer_form, tests/test_cart_su.py).
#include "tdr_score_cart_asm.h"
#define CART_SU_WBOX 1024   // LDS words of one wave's staged known mask (4 KB; a 32-row segment of config 4 needs ~200)

struct CartSuArgs {
  const uint32_t* bdesc;       // the descriptors in the block layout the generated loop streams (cart_prep_kernel)
  unsigned pbase0, pbytes;     // plane_offset's constant for the plane of class code c: pbase0 + c * pbytes (a byte offset from crec)
  const uint32_t* full_list;   // [chunks][list_cap]: bins with several classes, row << 16 | column (cart_prep_kernel)
  const int32_t* full_cnt;     // [chunks]
  int64_t list_cap;
  int seg_rows;                // window rows per segment (a multiple of 4)
  uint32_t* stats;             // NULL, or (profiling) counters of the variants the wave-segments ran: tdr_profile_variants
};
struct CartSuLds {             // ONE object so that the dictionary sits at LDS address 0 (the assembly reads it there)
  uint32_t dict[TDR_CMAP_MAX_DICT];
  uint32_t bits[4][CART_SU_WBOX];
};

// (five waves per SIMD: the kernel sits at 96-97 registers, and 4 waves instead of 5 cost it 12 % — 16.4 against 18.3 ms at config 4)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void score_cart_su_kernel(CartArgs a, CartSuArgs x) {
  constexpr int RF = 8, ND = 6, CW = 2;
  __shared__ CartSuLds lds;
  if (int_form_off(a.flags)) return;   // (uniform) the float form does this launch
  for (int t = threadIdx.x; t < a.dict_n; t += 256) lds.dict[t] = a.dict_int[t];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nact = (int64_t)*a.count;
  if ((int64_t)blockIdx.x * 256 >= nact) return;
  const int64_t wbase = ((int64_t)blockIdx.x * 4 + wave) * 64;
  if (wbase >= nact) return;           // an idle wave (no barrier below this line)
  const int64_t slot = wbase + lane;
  int64_t p = a.order[slot];
  const bool real = p >= 0;
  if (p < 0) p = a.order[wbase];       // padding of the slot list: a wave's first slot never is
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];
  const float theta = a.st[TDR_ST_THETA * a.cap + p];
  // the float operations of score_cart_kernel / score_cart_skip_kernel (src/top_down_map.cpp:367-389, 429-437)
  const float off0 = cy / a.resolution, off1 = cx / a.resolution;
  const float resq = (a.res * scale) / a.resolution;
  const float c = tdr_libm::cosf_v(theta, a.libm_fma), sn = tdr_libm::sinf_v(theta, a.libm_fma);
  const float ns = -sn;
  typedef float tdr_v2f __attribute__((ext_vector_type(2)));
  const tdr_v2f cs = {c, sn}, offv = {off0, off1};
  const float lo_r = (float)((double)(-resq * (float)(a.rows - 1)) / 2.), hi_r = (float)((double)(resq * (float)(a.rows - 1)) / 2.);
  const float lo_c = (float)((double)(-resq * (float)(a.cols - 1)) / 2.), hi_c = (float)((double)(resq * (float)(a.cols - 1)) / 2.);
  const float step_r = a.rows == 1 ? 0.f : (hi_r - lo_r) / (float)(a.rows - 1);
  const float step_c = a.cols == 1 ? 0.f : (hi_c - lo_c) / (float)(a.cols - 1);
  const int r1 = a.rows == 1 ? 1 : a.rows - 1, c1 = a.cols == 1 ? 1 : a.cols - 1;
  const float rmaxf = (float)a.map_rows, cmaxf = (float)a.map_cols;
  const bool weird = !(fabsf(off0) <= 1e9f) || !(fabsf(off1) <= 1e9f) || !(fabsf(resq) <= 1e6f);

  const int j0 = blockIdx.y * a.cpc, j1 = min(a.cols, j0 + a.cpc);
  const uint32_t* __restrict__ crec = a.crec;
  const char* __restrict__ crecb = reinterpret_cast<const char*>(a.crec);
  const int ckcol = a.ctiles_r * 128 - 16 * CW, ckconst = a.ctiles_r * 128 + 128;   // cmap_offset
  const int mrow = a.kmask_row, mconst = (int)a.kmask_off + a.kmask_row + 128;       // kmask_offset
  const int pkcol = a.pkcol;
  const uint32_t* __restrict__ kmask = reinterpret_cast<const uint32_t*>(crecb + a.kmask_off);
  const int kcolw = a.kmask_row >> 2;   // words of one tile column of the mask
  typedef const float __attribute__((address_space(4))) * tdr_const_f;
  typedef const uint32_t __attribute__((address_space(4))) * tdr_const_u;
  const tdr_const_f scanc = (tdr_const_f)a.scan_pk;
  const tdr_const_u descc = (tdr_const_u)a.desc;
  const tdr_const_u bdescc = (tdr_const_u)x.bdesc;
  const unsigned lds_base = (unsigned)(uintptr_t)&lds;
  const unsigned my_bits_lds = (unsigned)(uintptr_t)&lds.bits[wave][0];

  unsigned long long acc[ND];
#pragma unroll
  for (int k = 0; k < ND; k++) acc[k] = 0;
  uint32_t inorm = 0, known = 0;

  auto cell = [&](tdr_v2f cyi, tdr_v2f AB, int& ri, int& ci) {
    tdr_v2f pv = cyi + AB;         // p0 = c * yi + (-s * xj), p1 = s * yi + c * xj
    pv = pv + offv;
    tdr_v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
    qv = qv + 0.49999997f;         // see round_half_away_clamped
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ri) : "v"(qv.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ci) : "v"(qv.y));
  };
  auto column_terms = [&](int j) -> tdr_v2f {
    const float xj = cart_linspaced(j, c1, lo_c, hi_c, step_c);
    return (tdr_v2f){ns * xj, c * xj};
  };
  auto add_class = [&](uint32_t cd, uint32_t v, uint32_t m) {   // acc[cd - 1] += v * m: a switch over a wave-uniform value
    switch (cd) {
#define CART_SU_CASE(K) \
  case K + 1: asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[K]) : "s"(v), "v"(m) : "vcc"); break;
      CART_SU_CASE(0) CART_SU_CASE(1) CART_SU_CASE(2) CART_SU_CASE(3) CART_SU_CASE(4) CART_SU_CASE(5)
#undef CART_SU_CASE
      default: break;
    }
  };
  // NS samples of one window column in plain C++ (rows of cyi[], descriptors D): one compiler-tracked gather each — the
  // mask word of an empty bin, the plane cell of a bin with one class — all requested before the first is used
  auto samples = [&](auto ns_c, const tdr_v2f* cyi, tdr_v2f AB, tdr_const_u D) {
    constexpr int NS = decltype(ns_c)::value;
    uint32_t w[NS];
    int cis[NS];
#pragma unroll
    for (int u = 0; u < NS; u++) {
      int ri, ci;
      cell(cyi[u], AB, ri, ci);
      cis[u] = ci;
      if (D[4 * u] == 0) {   // wave-uniform
        w[u] = *reinterpret_cast<const uint32_t*>(crecb + kmask_offset(ri, ci, mrow, mconst));
      } else {
        int t1, t2;
        unsigned off;
        const int cq = ci >> 3;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(cq), "v"(pkcol), "s"(D[4 * u + 2]));
        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(t2) : "v"(ci), "v"(t1));
        asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(off) : "v"(ri), "v"(t2));
        w[u] = *reinterpret_cast<const uint16_t*>(crecb + off);
      }
    }
#pragma unroll
    for (int u = 0; u < NS; u++) {
      const uint32_t cd = D[4 * u];
      if (cd == 0) {   // wave-uniform
        known += (w[u] >> (cis[u] & 31)) & 1u;
      } else {
        const uint32_t kb = w[u] >> 15;          // bit 15 of a plane's cell (tdr_cmap.hip)
        known += kb;
        const uint32_t v = D[4 * u + 1];
        inorm += (0u - kb) & v;                  // the bin's count x known (state_particle.cpp:141-142)
        add_class(cd, v, *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds.dict) + (w[u] & 0xFFCu)));
      }
    }
  };
  // rows [ia, ib) x columns [ja, jb) in plain C++: blocks of 4 rows while they lie wholly below the last row, then row by row
  // (LinSpaced selects `high` for its last element: cart_linspaced)
  auto cpp_rows = [&](int ia, int ib, int ja, int jb) {
    int i = ia;
    for (; i + 4 <= ib && i + 4 <= a.rows - 1; i += 4) {
      tdr_v2f cyi[4];
#pragma unroll
      for (int u = 0; u < 4; u++) cyi[u] = cs * (lo_r + (float)(i + u) * step_r);
      for (int j = ja; j < jb; j++) samples(std::integral_constant<int, 4>{}, cyi, column_terms(j), descc + ((int64_t)j * a.rows + i) * 4);
    }
    for (; i < ib; i++) {
      const tdr_v2f cyi = cs * cart_linspaced(i, r1, lo_r, hi_r, step_r);
      for (int j = ja; j < jb; j++) samples(std::integral_constant<int, 1>{}, &cyi, column_terms(j), descc + ((int64_t)j * a.rows + i) * 4);
    }
  };

  // rows the assembly loop may take: whole blocks of 4 below the last row
  const int iend = ((a.rows - 1) / 4) * 4;
  const bool asm_ok = lds_base == 0 && (a.rows & 3) == 0 && x.seg_rows >= 4 && (x.seg_rows & 3) == 0;
  const uint64_t half2 = 0x3EFFFFFF3EFFFFFFull;         // {0.49999997f, 0.49999997f}
  const float rmax_s = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rmaxf)));
  const float cmax_s = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(cmaxf)));
  const uint32_t pbase0_s = __builtin_amdgcn_readfirstlane(x.pbase0), pbytes_s = __builtin_amdgcn_readfirstlane(x.pbytes);
  for (int jg = j0; jg < j1; jg += CART_ASM_NCOL) {
    if (!asm_ok || jg + CART_ASM_NCOL > j1) {   // (uniform) a partial group, or a shape the loop does not take
      cpp_rows(0, a.rows, jg, min(j1, jg + CART_ASM_NCOL));
      continue;
    }
    tdr_v2f ab[CART_ASM_NCOL];
#pragma unroll
    for (int jc = 0; jc < CART_ASM_NCOL; jc++) ab[jc] = column_terms(jg + jc);
    for (int ia = 0; ia < iend; ia += x.seg_rows) {
      const int ib = min(iend, ia + x.seg_rows);
      // cells this lane's samples of the segment can fall on: the corners bound them (see the kernel's comment)
      const tdr_v2f ya = cs * (lo_r + (float)ia * step_r), yb = cs * (lo_r + (float)(ib - 1) * step_r);
      const tdr_v2f p00 = (ya + ab[0]) + offv, p01 = (ya + ab[CART_ASM_NCOL - 1]) + offv;
      const tdr_v2f p10 = (yb + ab[0]) + offv, p11 = (yb + ab[CART_ASM_NCOL - 1]) + offv;
      const float a0 = fminf(fminf(p00.x, p01.x), fminf(p10.x, p11.x)), b0 = fmaxf(fmaxf(p00.x, p01.x), fmaxf(p10.x, p11.x));
      const float a1 = fminf(fminf(p00.y, p01.y), fminf(p10.y, p11.y)), b1 = fmaxf(fmaxf(p00.y, p01.y), fmaxf(p10.y, p11.y));
      int rl = (int)fminf(fmaxf(floorf(a0) - 1.f, -1.f), rmaxf), rh = (int)fminf(fmaxf(ceilf(b0) + 1.f, -1.f), rmaxf);
      int cl = (int)fminf(fmaxf(floorf(a1) - 1.f, -1.f), cmaxf), ch = (int)fminf(fmaxf(ceilf(b1) + 1.f, -1.f), cmaxf);
      if (weird || !(a0 == a0) || !(b0 == b0) || !(a1 == a1) || !(b1 == b1)) { rl = -1; rh = a.map_rows; cl = -1; ch = a.map_cols; }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        rl = min(rl, __shfl_xor(rl, d)); rh = max(rh, __shfl_xor(rh, d));
        cl = min(cl, __shfl_xor(cl, d)); ch = max(ch, __shfl_xor(ch, d));
      }
      const int wrl = __builtin_amdgcn_readfirstlane(rl), wrh = __builtin_amdgcn_readfirstlane(rh);
      const int wcl = __builtin_amdgcn_readfirstlane(cl), wch = __builtin_amdgcn_readfirstlane(ch);
      const int wl = (wcl >> 5) + 1, wh = (wch >> 5) + 1;   // mask words (a guard band of one word: kmask_offset)
      const int Hw = wrh - wrl + 1, Wbw = wh - wl + 1;
      if ((int64_t)Hw * Wbw > CART_SU_WBOX) {   // (wave-uniform) the box does not fit: the plain steps, the mask gathered
        if (x.stats && lane == 0) atomicAdd(&x.stats[11], 1u);
        cpp_rows(ia, ib, jg, jg + CART_ASM_NCOL);
        continue;
      }
      // (the all-known test looks at the box's own columns only: the bits of its first and last word that lie outside the
      // box count as known — a word is 32 cells wide, a 32-row segment's box about as many)
      const uint32_t edge_lo = (1u << (wcl & 31)) - 1u, edge_hi = ~((2u << (wch & 31)) - 1u);
      uint32_t ev = 0xFFFFFFFFu;
      for (int wc = 0; wc < Wbw; wc++) {     // (row fastest: consecutive lanes read consecutive words of one tile column)
        const uint32_t outside = (wc == 0 ? edge_lo : 0u) | (wc == Wbw - 1 ? edge_hi : 0u);
        for (int row = lane; row < Hw; row += 64) {
          const uint32_t wv = kmask[(int64_t)(wl + wc) * kcolw + (wrl + row + 32)];
          lds.bits[wave][row * Wbw + wc] = wv;
          ev &= wv | outside;
        }
      }
      // the wave reads what its lanes just wrote, through the assembly's ds_read: order the stores in front
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int allknown = __builtin_amdgcn_readfirstlane(__all(ev == 0xFFFFFFFFu) ? 1 : 0);   // (cells outside the map are unknown)
      const int inside = __builtin_amdgcn_readfirstlane((wrl >= 0 && wrh < a.map_rows && wcl >= 0 && wch < a.map_cols) ? 1 : 0);
      const int krow4 = Wbw * 4;
      const int kconst_s = __builtin_amdgcn_readfirstlane((int)my_bits_lds + (1 - wl - wrl * Wbw) * 4);
      const uint32_t i0_s = __builtin_amdgcn_readfirstlane((uint32_t)ia);
      const uint32_t nblk_s = __builtin_amdgcn_readfirstlane((uint32_t)((ib - ia) >> 2));
      // byte offset of block (column group jg / 8, row block ia / 4) in the block layout: 128 bytes a block
      const uint32_t doff_s = __builtin_amdgcn_readfirstlane((((uint32_t)jg >> 3) * ((uint32_t)a.rows >> 2) + ((uint32_t)ia >> 2)) * 128u);
#define CART_ASM_OPERANDS                                                                                                  \
      /* the accumulators are TIED to v[42:53]: the loop reaches them by VGPR-relative indexing (tools/gen_cart_asm.py) */       \
      : [a0] "+{v[42:43]}"(acc[0]), [a1] "+{v[44:45]}"(acc[1]), [a2] "+{v[46:47]}"(acc[2]), [a3] "+{v[48:49]}"(acc[3]),      \
        [a4] "+{v[50:51]}"(acc[4]), [a5] "+{v[52:53]}"(acc[5]), [norm] "+v"(inorm), [known] "+v"(known)                       \
      : [ab0] "v"(ab[0]), [ab1] "v"(ab[1]), [ab2] "v"(ab[2]), [ab3] "v"(ab[3]), [ab4] "v"(ab[4]), [ab5] "v"(ab[5]),         \
        [ab6] "v"(ab[6]), [ab7] "v"(ab[7]), [offv] "v"(offv), [cs] "v"(cs), [lor] "v"(lo_r), [stepr] "v"(step_r),           \
        [krow4] "v"(krow4), [pkcol] "v"(pkcol), [db] "s"(bdescc), [crec] "s"(crec), [rmax] "s"(rmax_s), [cmax] "s"(cmax_s), \
        [half] "s"(half2), [kconst] "s"(kconst_s), [i0] "s"(i0_s), [nblk] "s"(nblk_s), [doff] "s"(doff_s),                  \
        [pbase0] "s"(pbase0_s), [pbytes] "s"(pbytes_s)                                                                      \
      : CART_ASM_CLOBBERS
      if (x.stats && lane == 0) atomicAdd(&x.stats[allknown ? 8 : (inside ? 9 : 10)], 1u);   // [8..10] all known / inside / general, [11] plain steps
      if (allknown) asm volatile(CART_ASM_ALLKNOWN CART_ASM_OPERANDS);
      else if (inside) asm volatile(CART_ASM_NOCLAMP CART_ASM_OPERANDS);
      else asm volatile(CART_ASM CART_ASM_OPERANDS);
#undef CART_ASM_OPERANDS
    }
    cpp_rows(iend, a.rows, jg, jg + CART_ASM_NCOL);   // the last rows (LinSpaced's last element among them)
  }
  // bins holding several classes: the chunk's list (any order: the sums are exact); their known bit was counted above
  {
    const uint32_t* __restrict__ lst = x.full_list + (int64_t)blockIdx.y * x.list_cap;
    const int nfull = __builtin_amdgcn_readfirstlane(x.full_cnt[blockIdx.y]);
    for (int e = 0; e < nfull; e++) {
      const uint32_t w = __builtin_amdgcn_readfirstlane(lst[e]);
      const int i = (int)(w >> 16), j = (int)(w & 0xFFFFu);
      int ri, ci;
      cell(cs * cart_linspaced(i, r1, lo_r, hi_r, step_r), column_terms(j), ri, ci);
      const unsigned off = cmap_offset<CW, 2>(ri, ci, ckcol, ckconst);
      const uint2 rec = *reinterpret_cast<const uint2*>(crecb + off);
      const uint32_t wr[2] = {rec.x, rec.y};
      const uint32_t kb = rec.x & 1u;   // bit 0 of every dword of a narrow record is the known bit (tdr_cmap.hip)
      const tdr_const_f S = scanc + ((int64_t)j * a.rows + i) * RF;
      inorm += (0u - kb) & (uint32_t)S[RF - 1];
#pragma unroll
      for (int k = 0; k < ND; k++) {
        const float sk = S[k];
        if (sk != 0.f) {   // wave-uniform
          const uint32_t ww = wr[k / 3];
          const int sh = 10 * (k % 3);
          const uint32_t boff = sh ? ((ww >> sh) & 0xFFCu) : (ww & 0xFFCu);
          acc[k] += (unsigned long long)(uint32_t)sk *
                    *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds.dict) + boff);
        }
      }
    }
  }
  if (real) {   // [chunk][2 ncls + 2][npad] words, like score_polar_su_kernel
    const int ncls = a.ncls;
    uint32_t* o = reinterpret_cast<uint32_t*>(a.part) + (int64_t)blockIdx.y * (2 * ncls + 2) * a.npad + slot;
#pragma unroll
    for (int k = 0; k < ND; k++)
      if (k < ncls) {
        o[(int64_t)(2 * k) * a.npad] = (uint32_t)acc[k];
        o[(int64_t)(2 * k + 1) * a.npad] = (uint32_t)(acc[k] >> 32);
      }
    o[(int64_t)(2 * ncls) * a.npad] = inorm;
    o[(int64_t)(2 * ncls + 1) * a.npad] = known;
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static int g_cart_skip = 1;
extern "C" int tdr_config_cart_skip(int on) {   // < 0: query only
  if (on >= 0) g_cart_skip = on ? 1 : 0;
  return g_cart_skip;
}
extern "C" int tdr_cmap_words(int ncls);
extern "C" size_t tdr_cmap_tile_words(int ncls, int rows, int cols);

bool tdr_cart_skip_ok(const tdr_map_desc* map, int rf) {
  return g_cart_skip && rf <= 12 && map->cwords > 0 && map->cwords == tdr_cmap_words(map->ncls) && map->crec && map->dict &&
         map->dict_n > 0 && map->dict_n <= TDR_CMAP_MAX_DICT;
}

int tdr_cart_skip_launch(CartArgs a, const tdr_map_desc* map, int rf, uint32_t* desc_ws, hipStream_t s) {
  const int lc = map->cwords == 1 ? 3 : (map->cwords == 2 ? 2 : 1);
  const int ckconst = ((map->rows >> lc) + 2) * 128 + 128;   // cmap_offset (tdr_score_dev.h)
  const int64_t nbins = (int64_t)a.rows * a.cols;
  hipLaunchKernelGGL(cart_prep_kernel, dim3((unsigned)cdiv(nbins, 256)), dim3(256), 0, s, a.scan_pk, a.rows, a.cols, rf,
                     map->ncls, ckconst, map->dict, map->dict_n, desc_ws, 0, 0u, 0u);
  LAUNCH_CHECK("cart_prep");
  a.desc = desc_ws;
  a.kmask_off = (unsigned)(tdr_cmap_tile_words(map->ncls, map->rows, map->cols) * 4);   // the mask lies behind the tiles
  a.kmask_row = kmask_trows(map->rows) * 128;
  const dim3 grid((unsigned)cdiv(a.n, 256), (unsigned)a.nchunks), block(256);
  const bool ks = tdr_has_kslot(map->ncls, rf);
#define TDR_LAUNCH_CART_SKIP(NV4)                                                           \
  if (ks) hipLaunchKernelGGL((score_cart_skip_kernel<NV4, true>), grid, block, 0, s, a);    \
  else hipLaunchKernelGGL((score_cart_skip_kernel<NV4, false>), grid, block, 0, s, a);
  switch (rf / 4) {
    case 1: TDR_LAUNCH_CART_SKIP(1) break;
    case 2: TDR_LAUNCH_CART_SKIP(2) break;
    case 3: TDR_LAUNCH_CART_SKIP(3) break;
    default: return fail(TDR_ERR_ARG, "score_cart: no skipping kernel for record size %d", rf);
  }
#undef TDR_LAUNCH_CART_SKIP
  LAUNCH_CHECK("score_cart_skip");
  return TDR_OK;
}

// =========================================================================================================================
// The integer form of a Cartesian launch: score_cart_skip_kernel<.., INT> for the dense particles, score_cart_ray_kernel for
// the scattered ones — one WAVE per particle, lanes = 64 consecutive window columns of one window row, i.e. 64 cells along a
// line of the map (the rotation lives in the sampling, src/top_down_map.cpp:367-389): the mapping, the data (class planes,
// coarse mask plane, integer dictionary) and the loop of score_polar_ray_kernel (tdr_score_ray.hip).  The sample offsets are
// not a table here but three float operations per coordinate from the particle's rotation, computed per lane exactly like
// the lane = particle kernels compute them; the scan descriptors are not rotated (shift 0).
static inline int cart_ray_gq(int cols) { return cols <= 64 ? 1 : (cols <= 128 ? 2 : 4); }
static inline int cart_ray_blocks(int cols) { return (int)cdiv(cols, 64 * cart_ray_gq(cols)); }

// One thread per (window row i, padded column j): the 16-bit descriptor code << 12 | count of bin (i, j) in ray order
// [((i * blocks + b) * 64 + lane) * GQ + g], the list of bins with several classes or a count >= 4096 (as i << 16 | j), and
// the two words of int_form_off (see ray_prep_kernel, tdr_score_ray.hip).
__global__ __launch_bounds__(256) void cart_ray_prep_kernel(const float* __restrict__ scan_pk, int rows, int cols, int rf, int ncls,
                                                            int gq, int blocks, const uint32_t* __restrict__ dict_tail,
                                                            uint16_t* __restrict__ desc_ray, uint32_t* __restrict__ list,
                                                            int32_t* __restrict__ n_list, int32_t* __restrict__ flags) {
  const int cpad = blocks * gq * 64;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0 && dict_tail[1] != 1u) atomicOr(flags, 1);
  const bool live = t < (int64_t)rows * cpad;
  const int i = live ? (int)(t / cpad) : 0, j = live ? (int)(t - (int64_t)i * cpad) : 0;
  const bool real = live && j < cols;
  const int64_t bin = (int64_t)j * rows + i;   // scan_pk is [cols][rows][rf]
  uint32_t mass = 0;
  if (real) {
    const float sum = scan_pk[bin * rf + rf - 1];
    if (sum >= 1.f && sum < 16777216.f) mass = ((uint32_t)sum >> 8) + 1u;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mass += __shfl_xor(mass, d, 64);
  if ((threadIdx.x & 63) == 0 && mass) atomicAdd(reinterpret_cast<unsigned*>(flags) + 1, mass);
  if (!live) return;
  uint32_t d = 0;
  if (real) {
    const float* r = scan_pk + bin * rf;
    int nz = 0, first = 0;
    bool ok = true;
    for (int c = 0; c < ncls; c++) {
      const float v = r[c];
      ok &= v >= 0.f && v < 16777216.f && v == floorf(v);
      if (v != 0.f) {
        if (!nz) first = c;
        nz++;
      }
    }
    const float sum = r[rf - 1];
    ok &= sum >= 0.f && sum < 16777216.f && sum == floorf(sum);
    if (!ok) atomicOr(flags, 1);
    else if (nz == 1 && r[first] < 4096.f) d = (uint32_t)r[first] | ((uint32_t)(first + 1) << 12);
    else if (nz >= 1) list[atomicAdd(n_list, 1)] = ((uint32_t)i << 16) | (uint32_t)j;
  }
  const int g = j >> 6, l = j & 63, b = g / gq;
  desc_ray[(((int64_t)i * blocks + b) * 64 + l) * gq + (g - b * gq)] = (uint16_t)d;
}

struct CartRayArgs {
  const uint32_t* crec;
  unsigned planes_off, plane_bytes, cmask_off;
  int pkcol;
  const uint32_t* dict_int;
  int dict_n;
  int map_rows, map_cols;
  float resolution;
  const uint16_t* desc_ray;
  const uint32_t* list;
  const int32_t* n_list;
  const float* scan_pk;
  int rf, ncls, rows, cols, blocks;
  float res;
  const float* st;
  int64_t cap;
  const int32_t* slots;
  const int32_t* counts;
  const int32_t* flags;
  int nsplit, libm_fma;
  int64_t npad;
  uint32_t* part;
};

template <int GQ>
__global__ __launch_bounds__(256) void score_cart_ray_kernel(CartRayArgs a) {
  extern __shared__ unsigned long long lacc[];   // [4 waves][ncls + 1][64 lanes]
  __shared__ uint32_t ldict[TDR_CMAP_MAX_DICT];
  __shared__ uint4 lut[16];
  if (int_form_off(a.flags)) return;
  const int nsparse = a.counts[1];
  if ((int64_t)blockIdx.x * 4 >= (int64_t)nsparse * a.nsplit) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int t = threadIdx.x; t < a.dict_n; t += 256) ldict[t] = a.dict_int[t];
  const unsigned pconst = (unsigned)(a.pkcol + 16) + 128u;
  if (threadIdx.x < 16) {
    const int code = threadIdx.x;
    uint4 e;
    if (code >= 1 && code <= a.ncls) {
      e.x = a.planes_off + (unsigned)(code - 1) * a.plane_bytes + pconst;
      e.y = 0; e.z = 15; e.w = (unsigned)code * 512u;
    } else {
      e.x = a.cmask_off + pconst;
      e.y = 2; e.z = 0; e.w = 0;
    }
    lut[code] = e;
  }
  unsigned long long* const my = lacc + (size_t)wave * (a.ncls + 1) * 64 + lane;
  for (int c = 0; c <= a.ncls; c++) my[c * 64] = 0;
  __syncthreads();
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
  const int64_t q = gw / a.nsplit;
  const int part_id = (int)(gw - q * a.nsplit);
  if (q >= nsparse) return;
  const int64_t slot = (int64_t)a.counts[0] + q;
  const int64_t p = a.slots[slot];
  // the window of getLocalMap(center, rot = theta, res * scale) (src/top_down_map.cpp:429-459 via samplePts :367-389): the
  // float operations of score_cart_skip_kernel
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];
  const float theta = a.st[TDR_ST_THETA * a.cap + p];
  const float off0 = cy / a.resolution, off1 = cx / a.resolution;
  const float resq = (a.res * scale) / a.resolution;
  const float c = tdr_libm::cosf_v(theta, a.libm_fma), s = tdr_libm::sinf_v(theta, a.libm_fma);
  const float ns = -s;
  typedef float tdr_v2f __attribute__((ext_vector_type(2)));
  const tdr_v2f cs = {c, s}, offv = {off0, off1};
  const float lo_r = (float)((double)(-resq * (float)(a.rows - 1)) / 2.), hi_r = (float)((double)(resq * (float)(a.rows - 1)) / 2.);
  const float lo_c = (float)((double)(-resq * (float)(a.cols - 1)) / 2.), hi_c = (float)((double)(resq * (float)(a.cols - 1)) / 2.);
  const float step_r = a.rows == 1 ? 0.f : (hi_r - lo_r) / (float)(a.rows - 1);
  const float step_c = a.cols == 1 ? 0.f : (hi_c - lo_c) / (float)(a.cols - 1);
  const int r1 = a.rows == 1 ? 1 : a.rows - 1, c1 = a.cols == 1 ? 1 : a.cols - 1;
  const float rmaxf = (float)a.map_rows, cmaxf = (float)a.map_cols;
  const char* __restrict__ crecb = reinterpret_cast<const char*>(a.crec);
  auto cell = [&](tdr_v2f cyi, int j, int& ri, int& ci) {
    const float xj = cart_linspaced(j, c1, lo_c, hi_c, step_c);
    const tdr_v2f AB = {ns * xj, c * xj};
    tdr_v2f pv = cyi + AB;
    pv = pv + offv;
    tdr_v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
    qv = qv + 0.49999997f;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ri) : "v"(qv.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ci) : "v"(qv.y));
  };
  auto row_term = [&](int i) -> tdr_v2f { return cs * cart_linspaced(i, r1, lo_r, hi_r, step_r); };
  // The column terms {-sin * x_j, cos * x_j} of the lane's columns: a lane meets the same GQ * blocks columns in every
  // window row, so for windows of up to two blocks (512 columns) they are computed ONCE (2 GQ register pairs) instead of
  // in every step — seven of the step's forty vector instructions.  A column the window does not have gets a term that
  // throws its cell off the map: the guard cell is unknown and its descriptor is empty, so it counts nothing.
  // (GQ == 4 only — windows of 129 .. 512 columns, config 4's: the narrower instantiations would pay for the registers with
  // a wave per SIMD)
  constexpr int NT = GQ == 4 ? 2 * GQ : 1;
  const bool tab = GQ == 4 && a.blocks <= 2;   // (uniform)
  tdr_v2f ABt[NT];
#pragma unroll
  for (int k = 0; k < (GQ == 4 ? NT : 0); k++) {
    const int j = k * 64 + lane;
    const float xj = cart_linspaced(j, c1, lo_c, hi_c, step_c);
    ABt[k] = (tab && j < a.cols) ? (tdr_v2f){ns * xj, c * xj} : (tdr_v2f){1.0e30f, 1.0e30f};
  }
  auto cell_tab = [&](tdr_v2f cyi, tdr_v2f AB, int& ri, int& ci) {
    tdr_v2f pv = cyi + AB;
    pv = pv + offv;
    tdr_v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
    qv = qv + 0.49999997f;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ri) : "v"(qv.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ci) : "v"(qv.y));
  };

  // rows of steps: m = window row * blocks + block; this wave's share
  const int rows_all = a.rows * a.blocks, per = (rows_all + a.nsplit - 1) / a.nsplit;
  const int m0 = part_id * per, m1 = min(rows_all, m0 + per);
  uint32_t known = 0, norm = 0;
  constexpr int U = 8 / GQ;
  typedef uint16_t desc_t __attribute__((ext_vector_type(GQ)));
  const desc_t* __restrict__ descv = reinterpret_cast<const desc_t*>(a.desc_ray);
  auto rows_step = [&](auto cnt_c, auto tab_c, int m) {
    constexpr int N = decltype(cnt_c)::value;
    constexpr bool TAB = decltype(tab_c)::value;
    desc_t dd[N];
#pragma unroll
    for (int u = 0; u < N; u++) dd[u] = descv[(int64_t)(m + u) * 64 + lane];
    uint32_t v[N * GQ], shb[N * GQ], cnt[N * GQ], acc_at[N * GQ], ok[N * GQ];
#pragma unroll
    for (int u = 0; u < N; u++) {
      // (wave-uniform; with the table there are at most two blocks: no division)
      const int i = TAB ? (a.blocks == 2 ? (m + u) >> 1 : m + u) : (m + u) / a.blocks, b = (m + u) - i * a.blocks;
      const tdr_v2f cyi = row_term(i);
#pragma unroll
      for (int g = 0; g < GQ; g++) {
        const int sidx = u * GQ + g;
        const int j = (b * GQ + g) * 64 + lane;
        int ri, ci;
        if constexpr (TAB) {
          cell_tab(cyi, b ? ABt[(GQ + g) % NT] : ABt[g % NT], ri, ci);
          ok[sidx] = 1u;                     // (a column the window does not have fell on the guard cell: unknown)
        } else {
          cell(cyi, j, ri, ci);
          ok[sidx] = j < a.cols ? 1u : 0u;   // a column the window does not have counts nothing
        }
        const uint32_t d = dd[u][g];
        cnt[sidx] = d & 0xFFFu;
        const uint4 e = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lut) + ((d >> 8) & 0xF0u));
        const int rr = ri >> (int)e.y, cc = ci >> (int)e.y;
        const unsigned off = plane_offset(rr, cc, a.pkcol, (int)e.x);
        shb[sidx] = ((((uint32_t)ri & 3u) << 2) | ((uint32_t)ci & 3u)) | e.z;
        acc_at[sidx] = e.w;
        v[sidx] = *reinterpret_cast<const uint16_t*>(crecb + off);
      }
    }
#pragma unroll
    for (int sidx = 0; sidx < N * GQ; sidx++) {
      uint32_t kbit;
      asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(kbit) : "v"(v[sidx]), "v"(shb[sidx]));
      kbit &= ok[sidx];
      known += kbit;
      norm = __umul24(cnt[sidx], kbit) + norm;
      const uint32_t D = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ldict) + (v[sidx] & 0xFFCu));
      const unsigned long long prod = (unsigned long long)cnt[sidx] * D;
      unsigned long long* const acc = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(my) + acc_at[sidx]);
      __hip_atomic_fetch_add(acc, prod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  int m = m0;
  if constexpr (GQ == 4) {
    if (tab) {
      for (; m + U <= m1; m += U) rows_step(std::integral_constant<int, U>{}, std::true_type{}, m);
      for (; m < m1; m++) rows_step(std::integral_constant<int, 1>{}, std::true_type{}, m);
    }
  }
  {
    for (; m + U <= m1; m += U) rows_step(std::integral_constant<int, U>{}, std::false_type{}, m);
    for (; m < m1; m++) rows_step(std::integral_constant<int, 1>{}, std::false_type{}, m);
  }
  {   // the list: bins with several classes (or one large count)
    const int nm = *a.n_list, mper = (nm + a.nsplit - 1) / a.nsplit;
    const int e1 = min(nm, (part_id + 1) * mper);
    for (int e = part_id * mper + lane; e < e1; e += 64) {
      const uint32_t w = a.list[e];
      const int i = (int)(w >> 16), j = (int)(w & 0xFFFFu);
      const int64_t bin = (int64_t)j * a.rows + i;
      int ri, ci;
      cell(row_term(i), j, ri, ci);
      const unsigned cell_off = plane_offset(ri, ci, a.pkcol, (int)(a.planes_off + pconst));
      uint32_t kbit = 0;
      for (int cl = 0; cl < a.ncls; cl++) {
        const float sv = a.scan_pk[bin * a.rf + cl];
        if (sv != 0.f) {
          const uint32_t vv = *reinterpret_cast<const uint16_t*>(crecb + cell_off + (unsigned)cl * a.plane_bytes);
          kbit = vv >> 15;
          __hip_atomic_fetch_add(&my[(cl + 1) * 64], (unsigned long long)(uint32_t)sv * ldict[(vv & 0xFFCu) >> 2],
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      norm += (uint32_t)a.scan_pk[bin * a.rf + a.rf - 1] & (0u - kbit);
    }
  }
  uint32_t* o = a.part + (int64_t)part_id * (2 * a.ncls + 2) * a.npad + slot;
  for (int cl = 0; cl < a.ncls; cl++) {
    unsigned long long sum = __hip_atomic_load(&my[(cl + 1) * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) sum += __shfl_xor(sum, dlt, 64);
    if (lane == 0) {
      o[(int64_t)(2 * cl) * a.npad] = (uint32_t)sum;
      o[(int64_t)(2 * cl + 1) * a.npad] = (uint32_t)(sum >> 32);
    }
  }
#pragma unroll
  for (int dlt = 32; dlt > 0; dlt >>= 1) {
    known += __shfl_xor(known, dlt, 64);
    norm += __shfl_xor(norm, dlt, 64);
  }
  if (lane == 0) {
    o[(int64_t)(2 * a.ncls) * a.npad] = norm;
    o[(int64_t)(2 * a.ncls + 1) * a.npad] = known;
  }
}

// ---- host side of the integer form ---------------------------------------------------------------------------------------
extern "C" size_t tdr_cmap_plane_offset_words(int ncls, int rows, int cols);
extern "C" size_t tdr_cmap_plane_words(int ncls, int rows, int cols);
extern "C" int tdr_config_shift_uniform(int mode);

bool tdr_cart_int_ok(const tdr_map_desc* map, int rf, int rows, int cols, int64_t n_total) {
  // (mode 0 of tdr_config_shift_uniform switches the integer forms off altogether: A/B measurements, tests)
  return tdr_config_shift_uniform(-1) != 0 && tdr_cart_skip_ok(map, rf) && tdr_ray_map_ok(map) && rows < 65536 && cols < 65536 &&
         (tdr_config_shift_uniform(-1) == 2 || n_total >= 4096);
}
static inline int64_t cart_ray_desc_words(int rows, int cols) {
  return ((int64_t)rows * cart_ray_blocks(cols) * cart_ray_gq(cols) * 64 + 1) / 2 + 64;
}
static SuWs cart_order_ws(int64_t n) { return tdr_su_ws(1, 4, 4, std::max<int64_t>(n, 1)); }
// score_cart_su_kernel's tables: its descriptors (bins with several classes as empty ones), their per-chunk lists (room for
// every bin) and the lists' counters
static inline int64_t cart_bdesc_words(int rows, int cols) {   // block layout: whole groups of 8 columns + a block of pad
  return (int64_t)((cols + 7) / 8) * ((rows + 3) / 4) * 32 + 64;
}
static inline int64_t cart_su_words(int rows, int cols) {
  return tdr_cart_desc_words(rows, cols) + ((int64_t)rows * cols + 63) / 64 * 64 + ((int64_t)cols + 63) / 64 * 64 +
         cart_bdesc_words(rows, cols);
}
int64_t tdr_cart_int_words(int rows, int cols, int64_t n) {
  // [integer descriptors rows * cols * 4][ray descriptors][list rows * cols][ordering workspace][score_cart_su_kernel's tables]
  return tdr_cart_desc_words(rows, cols) + cart_ray_desc_words(rows, cols) + (int64_t)rows * cols + 64 + cart_order_ws(n).total +
         cart_su_words(rows, cols);
}
// rows of a segment of score_cart_su_kernel (a multiple of 4; tdr_config_tuning("cart_seg_rows", n)): the smaller the
// segment, the smaller the box of cells a wave stages and the likelier it holds no unknown cell; the larger, the fewer
// box computations.  Measured on config 4 (ms per step): see DESIGN.md 5.5
static int g_cart_seg_rows = 32;
extern "C" int tdr_config_cart_seg_rows(int n) {   // < 0: query only; 0: the assembly loop off (A/B: the plain kernel)
  if (n >= 0) g_cart_seg_rows = n - n % 4;
  return g_cart_seg_rows;
}

int tdr_cart_int_launch(CartArgs a, const tdr_map_desc* map, int rf, uint32_t* desc_ws, int32_t* ws, float span, hipStream_t s,
                        CartIntOut* out) {
  const int64_t n = a.n;
  // workspace
  uint32_t* desc_int = reinterpret_cast<uint32_t*>(ws);
  uint16_t* desc_ray = reinterpret_cast<uint16_t*>(ws + tdr_cart_desc_words(a.rows, a.cols));
  uint32_t* list = reinterpret_cast<uint32_t*>(ws + tdr_cart_desc_words(a.rows, a.cols) + cart_ray_desc_words(a.rows, a.cols));
  int32_t* ows = ws + tdr_cart_desc_words(a.rows, a.cols) + cart_ray_desc_words(a.rows, a.cols) + (int64_t)a.rows * a.cols + 64;
  const SuWs W = cart_order_ws(n);
  // ordering: dense particles (their 64 neighbours in the caller's pose order lie within the span) first, padded to whole
  // waves, then the scattered ones
  SuLaunch L{};
  L.map = map; L.st = a.st; L.cap = a.cap; L.n = n; L.perm = a.order; L.nb = 1; L.span = span; L.ws = ows;
  L.npad = su_npad(std::max<int64_t>(n, 1), 1);
  const int32_t* slots = nullptr;
  const int32_t* counts = nullptr;
  if (int rc = tdr_su_order(L, W, s, &slots, &counts)) return rc;
  int32_t* ints = const_cast<int32_t*>(counts);   // [counts 3][n_list][inexact][mass bound]
  // descriptors: the float form's (desc_ws, for the fallback), the integer form's, the ray order's + the flags
  const int lc = map->cwords == 1 ? 3 : (map->cwords == 2 ? 2 : 1);
  const int ckconst = ((map->rows >> lc) + 2) * 128 + 128;
  const int64_t nbins = (int64_t)a.rows * a.cols;
  const unsigned plane_bytes = (unsigned)(tdr_cmap_plane_words(map->ncls, map->rows, map->cols) * 4);
  const unsigned pbase = (unsigned)(tdr_cmap_plane_offset_words(map->ncls, map->rows, map->cols) * 4) +
                         (unsigned)(plane_trows(map->rows) * 128) + 128u;   // plane_offset's constant for class 0's plane
  hipLaunchKernelGGL(cart_prep_kernel, dim3((unsigned)cdiv(nbins, 256)), dim3(256), 0, s, a.scan_pk, a.rows, a.cols, rf,
                     map->ncls, ckconst, map->dict, map->dict_n, desc_ws, 0, 0u, 0u);
  hipLaunchKernelGGL(cart_prep_kernel, dim3((unsigned)cdiv(nbins, 256)), dim3(256), 0, s, a.scan_pk, a.rows, a.cols, rf,
                     map->ncls, ckconst, map->dict, map->dict_n, desc_int, 1, pbase, plane_bytes);
  LAUNCH_CHECK("cart_prep");
  const int gq = cart_ray_gq(a.cols), blocks = cart_ray_blocks(a.cols);
  const int64_t T = (int64_t)a.rows * blocks * gq * 64;
  hipLaunchKernelGGL(cart_ray_prep_kernel, dim3((unsigned)cdiv(T, 256)), dim3(256), 0, s, a.scan_pk, a.rows, a.cols, rf, map->ncls,
                     gq, blocks, reinterpret_cast<const uint32_t*>(map->dict) + 2 * TDR_CMAP_MAX_DICT, desc_ray, list, ints + 3,
                     ints + 4);
  LAUNCH_CHECK("cart_ray_prep");
  const int32_t* flags = ints + 4;
  // the dense share through score_cart_su_kernel (generated sample loop, staged mask) when the records are the two-dword
  // ones: its own descriptors, the lists of bins with several classes per column chunk
  const bool su_kernel = rf == 8 && tdr_has_kslot(map->ncls, rf) && map->cwords == 2 && g_cart_seg_rows >= 4 && (a.rows & 3) == 0 &&
                         a.rows >= 8 && (int64_t)a.rows * a.cols * 16 < (int64_t)1 << 31;
  const int cpc_su = (a.cpc + CART_ASM_NCOL - 1) / CART_ASM_NCOL * CART_ASM_NCOL;
  const int nchunks_su = (int)cdiv(a.cols, cpc_su);   // <= a.nchunks: the partial sums' rows suffice
  uint32_t* desc_su = reinterpret_cast<uint32_t*>(ows + W.total);
  uint32_t* full_list = desc_su + tdr_cart_desc_words(a.rows, a.cols);
  int32_t* full_cnt = reinterpret_cast<int32_t*>(full_list + ((int64_t)a.rows * a.cols + 63) / 64 * 64);
  uint32_t* bdesc = reinterpret_cast<uint32_t*>(full_cnt + ((int64_t)a.cols + 63) / 64 * 64);
  const int64_t list_cap = (int64_t)a.rows * cpc_su;
  if (su_kernel) {
    HIP_TRY(hipMemsetAsync(full_cnt, 0, sizeof(int32_t) * (size_t)nchunks_su, s));
    HIP_TRY(hipMemsetAsync(bdesc, 0, sizeof(uint32_t) * (size_t)cart_bdesc_words(a.rows, a.cols), s));
    hipLaunchKernelGGL(cart_prep_kernel, dim3((unsigned)cdiv(nbins, 256)), dim3(256), 0, s, a.scan_pk, a.rows, a.cols, rf,
                       map->ncls, ckconst, map->dict, map->dict_n, desc_su, 1, pbase, plane_bytes, full_list, full_cnt, cpc_su,
                       list_cap, bdesc);
    LAUNCH_CHECK("cart_prep(su)");
  }
  a.kmask_off = (unsigned)(tdr_cmap_tile_words(map->ncls, map->rows, map->cols) * 4);
  a.kmask_row = kmask_trows(map->rows) * 128;
  a.ncls = map->ncls;
  const bool ks = tdr_has_kslot(map->ncls, rf);
  const int64_t npad_int = L.npad;
  // scattered particles: one wave each
  {
    CartRayArgs r;
    r.crec = map->crec;
    const size_t pw = tdr_cmap_plane_words(map->ncls, map->rows, map->cols);
    r.planes_off = (unsigned)(tdr_cmap_plane_offset_words(map->ncls, map->rows, map->cols) * 4);
    r.plane_bytes = (unsigned)(pw * 4);
    r.cmask_off = r.planes_off + (unsigned)map->ncls * r.plane_bytes;
    r.pkcol = plane_trows(map->rows) * 128 - 16;
    r.dict_int = reinterpret_cast<const uint32_t*>(map->dict) + TDR_CMAP_MAX_DICT;
    r.dict_n = map->dict_n;
    r.map_rows = map->rows; r.map_cols = map->cols; r.resolution = map->resolution;
    r.desc_ray = desc_ray; r.list = list; r.n_list = ints + 3; r.scan_pk = a.scan_pk;
    r.rf = rf; r.ncls = map->ncls; r.rows = a.rows; r.cols = a.cols; r.blocks = blocks; r.res = a.res;
    r.st = a.st; r.cap = a.cap; r.slots = slots; r.counts = counts; r.flags = flags;
    r.nsplit = tdr_ray_splits(a.rows, a.cols, n);
    r.libm_fma = a.libm_fma;
    r.npad = npad_int; r.part = reinterpret_cast<uint32_t*>(a.part);
    const dim3 grid((unsigned)cdiv(n * r.nsplit, 4)), block(256);
    const size_t lds = (size_t)4 * (map->ncls + 1) * 64 * sizeof(unsigned long long);
    if (gq == 1) hipLaunchKernelGGL((score_cart_ray_kernel<1>), grid, block, lds, s, r);
    else if (gq == 2) hipLaunchKernelGGL((score_cart_ray_kernel<2>), grid, block, lds, s, r);
    else hipLaunchKernelGGL((score_cart_ray_kernel<4>), grid, block, lds, s, r);
    LAUNCH_CHECK("score_cart_ray");
    out->ray_split = r.nsplit;
  }
  // dense particles: the skipping kernel with integer accumulators over the dense slots
  {
    CartArgs d = a;
    d.desc = desc_int;
    d.dict_int = reinterpret_cast<const uint32_t*>(map->dict) + TDR_CMAP_MAX_DICT;
    d.flags = flags; d.run_if_int = 1;
    d.order = slots; d.count = counts; d.npad = npad_int;
    d.pkcol = plane_trows(map->rows) * 128 - 16;
    out->nchunks_dense = a.nchunks;
    if (su_kernel) {
      CartSuArgs x;
      x.full_list = full_list; x.full_cnt = full_cnt; x.list_cap = list_cap; x.seg_rows = g_cart_seg_rows;
      x.bdesc = bdesc; x.pbytes = plane_bytes; x.pbase0 = pbase - plane_bytes;
      x.stats = tdr_profile_stats_ptr();
      d.desc = desc_su;
      d.cpc = cpc_su;
      d.nchunks = nchunks_su;
      out->nchunks_dense = nchunks_su;
      hipLaunchKernelGGL(score_cart_su_kernel, dim3((unsigned)cdiv(npad_int, 256), (unsigned)nchunks_su), dim3(256), 0, s, d, x);
      LAUNCH_CHECK("score_cart_su");
    } else {
    const dim3 grid((unsigned)cdiv(npad_int, 256), (unsigned)a.nchunks), block(256);
#define TDR_LAUNCH_CART_INT(NV4)                                                                 \
  if (ks) hipLaunchKernelGGL((score_cart_skip_kernel<NV4, true, true>), grid, block, 0, s, d);    \
  else hipLaunchKernelGGL((score_cart_skip_kernel<NV4, false, true>), grid, block, 0, s, d);
    switch (rf / 4) {
      case 1: TDR_LAUNCH_CART_INT(1) break;
      case 2: TDR_LAUNCH_CART_INT(2) break;
      case 3: TDR_LAUNCH_CART_INT(3) break;
      default: return fail(TDR_ERR_ARG, "score_cart: no skipping kernel for record size %d", rf);
    }
#undef TDR_LAUNCH_CART_INT
    LAUNCH_CHECK("score_cart_skip(int)");
    }
  }
  // the float form, for a scan / map without an integer form (returns at once otherwise)
  {
    CartArgs f = a;
    f.desc = desc_ws;
    f.flags = flags; f.run_if_int = 0;
    const dim3 grid((unsigned)cdiv(a.n, 256), (unsigned)a.nchunks), block(256);
#define TDR_LAUNCH_CART_FLT(NV4)                                                           \
  if (ks) hipLaunchKernelGGL((score_cart_skip_kernel<NV4, true>), grid, block, 0, s, f);    \
  else hipLaunchKernelGGL((score_cart_skip_kernel<NV4, false>), grid, block, 0, s, f);
    switch (rf / 4) {
      case 1: TDR_LAUNCH_CART_FLT(1) break;
      case 2: TDR_LAUNCH_CART_FLT(2) break;
      default: TDR_LAUNCH_CART_FLT(3) break;
    }
#undef TDR_LAUNCH_CART_FLT
    LAUNCH_CHECK("score_cart_skip(float form)");
  }
  out->slots = slots; out->counts = counts; out->flags = flags; out->npad = npad_int;
  return TDR_OK;
}
