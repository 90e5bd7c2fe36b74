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
#include "tdr_sincosf.h"

#define CART_CODE_FULL 0xFFu       // several classes present
#define CART_CODE_FULL_ALL 0xFEu   // a non-finite value in play: every class multiplied
#define CART_U 4                   // window rows per step (TDR_SCORE_U of score_cart_kernel: the order of the sums)

// Descriptor of bin (j, i) (window column j, row i; scan_pk is [cols][rows][rf]), four dwords:
//   [0] code: 0 = every class zero; c + 1 = class c alone; CART_CODE_FULL / CART_CODE_FULL_ALL
//   [1] the bin's sum over the classes (slot rf - 1 of the packed record) — for a single class: its value
//   [2] cmap_offset's constant advanced to the dword the class lives in: ckconst + 4 * (c / 3)
//   [3] bit offset of the class's field in that dword minus 2 (10 * (c % 3))
__global__ __launch_bounds__(256) void cart_prep_kernel(const float* __restrict__ scan_pk, int rows, int cols, int rf, int ncls,
                                                        int ckconst, const float* __restrict__ dict, int dict_n,
                                                        uint32_t* __restrict__ desc) {
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
    if (nz == 1) { val = r[first]; ckc += 4u * (uint32_t)(first / 3); sh = 10u * (uint32_t)(first % 3); return (uint32_t)first + 1u; }
    if (nz > 1) { val = r[rf - 1]; return CART_CODE_FULL; }
    return 0u;
  };
  float val;
  uint32_t ckc, sh;
  const uint32_t code = classify(t, val, ckc, sh);
  desc[4 * t] = code;
  desc[4 * t + 1] = __float_as_uint(val);
  desc[4 * t + 2] = ckc;
  desc[4 * t + 3] = sh;
}

__device__ __forceinline__ float cart_linspaced(int i, int size1, float low, float high, float step) {
  // Eigen LinSpaced<float>, |high| == |low| here, so never the flipped branch of linspaced_op_impl
  return (i == size1) ? high : (low + (float)i * step);
}

// lane = particle; grid.y = chunk of a.cpc window columns; the sample order and the partition into partial sums are
// score_cart_kernel's.
template <int NV4, bool KSLOT>
__global__ __launch_bounds__(256) void score_cart_skip_kernel(CartArgs a) {
  constexpr int RF = 4 * NV4;
  constexpr int ND = CmapShape<RF, KSLOT>::ND, CW = CmapShape<RF, KSLOT>::CW, LC = CmapShape<RF, KSLOT>::LC;
  __shared__ float ldict[TDR_CMAP_MAX_DICT];
  for (int t = threadIdx.x; t < a.dict_n; t += 256) ldict[t] = a.dict[t];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t slot = ((int64_t)blockIdx.x * 4 + wave) * 64 + lane;
  if ((int64_t)blockIdx.x * 256 >= a.n) return;
  const bool valid = slot < a.n;
  const int64_t p = a.order ? (int64_t)a.order[valid ? slot : 0] : (valid ? slot : 0);
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

  auto field = [&](const uint32_t (&w)[CW], int k) -> float {   // distance k of a compact record (cmap_decode, one field)
    const uint32_t ww = w[k / 3];
    const int sh = 10 * (k % 3);
    const uint32_t boff = sh ? ((ww >> sh) & 0xFFCu) : (ww & 0xFFCu);
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ldict) + boff);
  };
  auto field1 = [&](uint32_t ww, int k) -> float {   // ... when the sample loaded only the dword class k lives in
    const int sh = 10 * (k % 3);
    const uint32_t boff = sh ? ((ww >> sh) & 0xFFCu) : (ww & 0xFFCu);
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ldict) + boff);
  };
  float acc[ND];
#pragma unroll
  for (int k = 0; k < ND; k++) acc[k] = 0.f;
  float norm = 0.f;
  uint32_t known = 0;
  auto single_class = [&](uint32_t cd, float v, uint32_t ww) {   // a switch over a wave-uniform value
    switch (cd) {
#define CART_CASE(K)                                                                                \
  case K + 1:                                                                                       \
    if constexpr (K < ND) {                                                                         \
      const float m = field1(ww, K < ND ? K : 0);                                                   \
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[K < ND ? K : 0]) : "s"(v), "v"(m));           \
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
      } else {
        int t1, t2;
        const int cq = ci >> 2;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(cq), "v"(ckcol), "s"(D[4 * u + 2]));
        asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(t2) : "v"(ci), "n"(CW == 1 ? 2 : (CW == 2 ? 3 : 4)), "v"(t1));
        asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(offs[u]) : "v"(ri), "n"(CW == 1 ? 4 : (CW == 2 ? 5 : 6)), "v"(t2));
      }
      asm volatile("global_load_dword %0, %1, %2" : "=v"(w[u]) : "v"(offs[u]), "s"(crec));
    }
#pragma unroll
    for (int u = 0; u < NS; u++) {
      // (the requests return in order: sample u is there once all but the NS - 1 - u behind it are)
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[u]) : "n"(NS - 1 - u));
      const uint32_t cd = D[4 * u];
      if (cd == 0) {   // wave-uniform
        int kmsk;      // 0 / -1: the cell's known bit
        asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(kmsk) : "v"(w[u]), "v"(cis[u]));
        known -= (uint32_t)kmsk;
      } else {
        const uint32_t kb = w[u] & 1u;   // bit 0 of every dword of a compact record (tdr_cmap.hip)
        known += kb;
        const float v = __uint_as_float(D[4 * u + 1]);
        if (cd < CART_CODE_FULL_ALL) {
          // the bin's sum x known (state_particle.cpp:141-142): fma(val, 1 or 0, norm) for a finite val
          norm = norm + __uint_as_float((0u - kb) & __float_as_uint(v));
          single_class(cd, v, w[u]);
        } else {   // several classes (or a non-finite value in play): the whole record, the packed scan record
          uint32_t wr[CW];
          wr[0] = w[u];
#pragma unroll
          for (int d = 1; d < CW; d++)
            wr[d] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(crec) + 4 * d + offs[u]);
          const tdr_const_f S = scanc + (bin0 + u) * RF;
          norm = __builtin_fmaf(v, (float)kb, norm);
#pragma unroll
          for (int k = 0; k < ND; k++) {
            const float sk = S[k];
            if (cd == CART_CODE_FULL_ALL || sk != 0.f) acc[k] = __builtin_fmaf(sk, field(wr, k), acc[k]);
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
  if (slot < a.npad) {
    float* o = a.part + (int64_t)blockIdx.y * (RF + 1) * a.npad + slot;
#pragma unroll
    for (int k = 0; k < ND; k++) o[(int64_t)k * a.npad] = acc[k];
    o[(int64_t)(RF - 1) * a.npad] = norm;
    o[(int64_t)RF * a.npad] = (float)known;
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static int g_cart_skip = [] {
  const char* e = getenv("TDR_CART_SKIP");
  return e ? atoi(e) : 1;
}();
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
                     map->ncls, ckconst, map->dict, map->dict_n, desc_ws);
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
