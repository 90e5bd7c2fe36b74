// tdr_score.hip — per-particle window gather + class-wise score: polar, Cartesian, finalize, the 40-rotation init search.
#include <type_traits>

#include "tdr_common.h"
#include "tdr_sincosf.h"

// ------------------------------------------------------------------------------------------------------------------
// K2: scoring.  lane = particle, 64 particles per wave, 4 waves per workgroup; grid.y = chunk of range rings.
// All lanes of a wave visit the same window sample (i,j) at the same time, so their map reads fall on neighbouring
// cells when the particles are neighbours (tdr_k_locality_order) and coalesce in L1/L2 instead of being 64
// unrelated gathers; the rotation enters only as a per-lane row offset into the ring of scan records held in LDS.
struct ScoreArgs {
  const float* rec;     // map cell records
  int rows, cols;       // map
  float resolution;
  const float* tab;     // [P][2]
  const float* utab;    // [P][2] (tab*scale)*res when all particles share one scale, else NULL
  const float* scan_pk; // [nr][nb][rf]
  int nb, nr;
  float res;
  const float* st;      // [7][cap]
  int64_t cap, n;
  const int32_t* order; // slot -> particle (NULL = identity)
  const int32_t* count; // optional device count limiting the active slots (init search; the sparse share of a mixed launch)
  const int32_t* slot_base;  // optional device word: this launch's slot 0 is slot *slot_base of `order` and of `part`
  int use_theta_override;
  float theta_override;
  int only_uninit;      // score only workgroups that hold a particle without a heading (the geometric init search)
  int group, nchunks;   // rings per workgroup (score_group_rings), number of groups
  int64_t npad;         // slots padded to a multiple of 64
  float* part;          // [nchunks][rf+1][npad]
  // compact form of the records (tdr_cmap.hip), read by the COMPACT instantiations
  const uint32_t* crec;
  const float* dict;
  int dict_n;           // dictionary entries in use
  int ctiles_r;         // tiles per tile column
  // the map's known mask (tdr_cmap.hip, layout: kmask_offset), read by the SKIP instantiations: its byte offset from
  // crec, bytes per row of its tiles
  unsigned kmask_off;
  int kmask_row;         // bytes of one tile column of it
  const int32_t* run_if = nullptr; // optional device words (int_form_off): the launch runs only then (the float form of a launch whose
                         // integer form — tdr_score_su.hip, tdr_score_ray.hip — applies)
};

#include "tdr_score_dev.h"   // rot_shift_dev, the coordinate rounding, compact-record geometry / load / decode
#include "tdr_score_su.h"    // the shift-uniform kernel's host interface (tdr_score_su.hip)
#include "tdr_score_cart.h"  // CartArgs, the Cartesian kernel that skips empty scan bins (tdr_score_cart.hip)


#ifdef TDR_SCORE_TIMELINE   // diagnostic build: start / end time stamp (100 MHz) of every workgroup
#define TDR_TL_MAX (1 << 17)
__device__ unsigned long long g_timeline[2 * TDR_TL_MAX];
extern "C" int tdr_debug_read_timeline(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * 2 * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif
// COMPACT: this instantiation reads the compact records (10-bit dictionary indices, decoded through LDS; bit-identical
// operands) instead of the dense ones — a quarter of the bytes through L1 / L2 / HBM for six classes.  It is the one
// that runs whenever the map has a compact form (A/B on MI355X, config 2, 100 k particles, ms per launch dense ->
// compact: bench mix 15.5 -> 9.6, 100 % Gaussian 30 px 7.3 -> 5.9, Gaussian 5 px 6.2 -> 5.8, 100 % uniform 97 -> 67,
// 8 clusters 20.3 -> 6.6); the dense instantiation remains for maps without one (> 1024 distinct values, > 11 classes).
//
// Work decomposition: grid.y = GROUP of a.group consecutive range rings (a constant of the image shape, see
// score_group_rings), so the summation tree of a particle's score is a pure function of the configuration — never of the
// number of particles in the launch or of the rank count.  Inside a group the samples are visited RAY-major: for each
// window row i (a direction) the group's rings in turn, i.e. up to a.group consecutive cells along one ray.  Compact
// records are tiled 4 x 4 cells per 128-byte line, so a ray stays in a tile for ~3 steps: a lane whose neighbours are far
// away (scattered particles) fetches ~0.5 lines per sample instead of one.  The scan rows of the whole group sit in LDS
// ([ring][plane][row], not doubled: the row (i + shift) mod nb is computed once per direction).
// WIDE: the compact records are the wide form (maps of more than 1024 distinct values, tdr_cmap.hip)
// SKIP (the scattered particles of a mixed launch, tdr_k_score_polar): a sample whose scan bin is empty in every class
// reads the cell's bit of the known mask instead of its record.  Scattered particles share no cache lines and are bound
// by the lines they pull through the fabric; the empty bins of a LiDAR scan are mostly its outer rings, where a window
// touches the most lines: half of a scattered particle's lines are never requested.  The FMAs of such a
// sample are still executed, with a zero scan operand against the dictionary's entry 0: the sums are bit-identical.
template <int NV4, int U, bool KSLOT, bool USCALE, bool COMPACT, bool WIDE = false, bool SKIP = false>
__global__ __launch_bounds__(256, COMPACT ? (WIDE ? 3 : 5) : 1) void score_polar_kernel(ScoreArgs a) {
  constexpr int RF = 4 * NV4;
  static_assert(!WIDE || (COMPACT && NV4 == 2), "wide compact records: 8-float dense records only");
  static_assert(!SKIP || (COMPACT && !WIDE), "SKIP: narrow compact records only");
  constexpr int CW = WIDE ? 4 : CmapShape<RF, KSLOT>::CW, LC = WIDE ? 1 : CmapShape<RF, KSLOT>::LC;
  constexpr int NDICT = WIDE ? TDR_CMAP_WIDE_MAX_DICT : TDR_CMAP_MAX_DICT;
#ifdef TDR_SCORE_TIMELINE
  const unsigned tl_id = blockIdx.y * gridDim.x + blockIdx.x;
  if (threadIdx.x == 0 && tl_id < TDR_TL_MAX) g_timeline[2 * tl_id] = wall_clock64();
#endif
  extern __shared__ float4 ring[];  // [nb rows][rs]: a row's (ring, plane) records side by side, rs = group * NV4 | 1
  __shared__ float ldict[COMPACT ? NDICT : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#if TDR_XCD_SWIZZLE
  // Workgroups are dealt round-robin over the 8 XCDs; remap so that each XCD (its own L2) gets a contiguous run of
  // particle batches (Morton neighbours) instead of every 8th one.  Speed only: any mapping gives the same results.
  const unsigned nbx = gridDim.x, per = (nbx + 7) / 8;
  unsigned bx = (blockIdx.x % 8) * per + blockIdx.x / 8;
  if (nbx % 8 != 0) bx = blockIdx.x;  // keep it a bijection
#else
  const unsigned bx = blockIdx.x;
#endif
  const int64_t slot = ((int64_t)bx * 4 + wave) * 64 + lane;
  if (a.run_if && !int_form_off(a.run_if)) return;  // (uniform)
  const int64_t nact = a.count ? (int64_t)*a.count : a.n;
  if ((int64_t)bx * 256 >= nact) return;  // whole workgroup idle (uniform)
  const bool valid = slot < nact;
  const int64_t sbase = a.slot_base ? (int64_t)*a.slot_base : 0;
  const int64_t p = a.order ? (int64_t)a.order[sbase + (valid ? slot : 0)] : (valid ? slot : 0);
  if (a.only_uninit && !__syncthreads_or(valid && a.st[TDR_ST_HAVE_INIT * a.cap + p] == 0.f)) return;
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];  // state_particle.cpp:161
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];  // :162
  const float off0 = cy / a.resolution;  // top_down_map_polar.cpp:29
  const float off1 = cx / a.resolution;  // :30
  const float theta = a.use_theta_override ? a.theta_override : a.st[TDR_ST_THETA * a.cap + p];
  const int shift = rot_shift_dev(theta, a.nb);  // scan row paired with window row i is (i + shift) mod nb

  const int j0 = blockIdx.y * a.group, gn = min(a.nr - j0, a.group);   // this workgroup's rings: [j0, j0 + gn)
  const int rowstride = (a.cols + 2) * (RF * 4);            // bytes per guarded map row
  const int kbase = (a.cols + 3) * (RF * 4);                // byte offset of cell (0,0)
  const float rmaxf = (float)a.rows, cmaxf = (float)a.cols;
  const char* __restrict__ recb = reinterpret_cast<const char*>(a.rec);
  const char* __restrict__ crecb = reinterpret_cast<const char*>(a.crec);
  const int ckcol = a.ctiles_r * 128 - 16 * CW, ckconst = a.ctiles_r * 128 + 128;   // cmap_offset
  // The sample table is read-only for the whole launch and every lane of a wave reads the same entry: it is addressed
  // through the CONSTANT address space so that these are scalar loads whatever else the kernel contains.  (Left to its
  // own no-clobber analysis the compiler gives up in the compact kernel — the dictionary staging is one store too many —
  // and emits vector loads with a full wait in front of every address computation.)
  typedef const float __attribute__((address_space(4))) * tdr_const_f;
  const tdr_const_f tabc = (tdr_const_f)(USCALE ? a.utab : a.tab);
  auto tab_at = [&](int64_t k) { return make_float2(tabc[2 * k], tabc[2 * k + 1]); };
  const float4* __restrict__ scan4 = reinterpret_cast<const float4*>(a.scan_pk);
  const int nb = a.nb;

  // stage the group's scan rows and (compact) the dictionary
  bool dict_bad = false;   // a non-finite dictionary value: 0 x inf must stay NaN, nothing may be skipped
  if constexpr (COMPACT)
    for (int t = threadIdx.x; t < a.dict_n; t += 256) {
      const float v = a.dict[t];
      ldict[t] = v;
      if constexpr (SKIP) dict_bad |= !(fabsf(v) <= 3.402823466e+38f);
    }
  // One row of the LDS image = the scan records (ring, plane) of one direction, 16 bytes each, side by side: a step
  // reads them with ONE address per lane and immediate offsets.  The row stride is an ODD number of 16-byte slots, so
  // lanes on different rows (different headings) fall on different banks.
  const int rs = (a.group * NV4) | 1;
  for (int jj = 0; jj < gn; jj++)
    for (int t = threadIdx.x; t < nb * NV4; t += 256) {
      const float4 v = scan4[(int64_t)(j0 + jj) * nb * NV4 + t];
      const int row = t / NV4, pl = t - row * NV4;
      ring[row * rs + jj * NV4 + pl] = v;
    }
  bool skip_ok = false;
  if constexpr (SKIP) skip_ok = !__syncthreads_or(dict_bad);
  else __syncthreads();

  // USCALE: every particle has the same scale, so (tab*scale)*res was evaluated once per step into a.utab and is
  // wave-uniform here; otherwise it is evaluated per lane.  Identical float operations either way.
  // Returns the byte offset of the sample's record (dense: guarded row-major grid; compact: tiled).
  typedef float tdr_v2f __attribute__((ext_vector_type(2)));   // both coordinates in one v_pk_add_f32 / v_pk_mul_f32
  const tdr_v2f offv = {off0, off1};
  unsigned moff[SKIP ? U : 1];   // SKIP: byte offset (from crec) of the known-mask word of sample u's cell ...
  int mbit[SKIP ? U : 1];        // ... and the cell's column (its low 5 bits: the bit in that word)
  const int mconst = (int)a.kmask_off + a.kmask_row + 128;   // kmask_offset
  auto cell_offset = [&](float2 t, int u) -> unsigned {
    tdr_v2f pv = {t.x, t.y};
    if constexpr (!USCALE) pv = (pv * scale) * a.res;  // top_down_map_polar.cpp:28
    pv = pv + offv;                                     // :29-30
    // clamp into the guard ring, then round like `pts.round().cast<int>()` (:31): roundf(x) == floor(fl(x + (0.5 - 2^-25)))
    // on [-1, 2^23] (round_half_away_clamped), the addition done for both coordinates at once
    tdr_v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
    qv = qv + 0.49999997f;
    int ri, ci;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ri) : "v"(qv.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ci) : "v"(qv.y));
    if constexpr (SKIP) {
      moff[u] = kmask_offset(ri, ci, a.kmask_row, mconst);
      mbit[u] = ci;
    }
    if constexpr (COMPACT) {
      // cells of the guard ring are zero records in their own right (distance 0, unknown): no select needed
      return cmap_offset<CW, LC>(ri, ci, ckcol, ckconst);
    } else {
#if TDR_OOB_ALIAS
      // every out-of-bounds sample reads the SAME guard record (always cache-resident) instead of a distinct one
      const bool inb = (unsigned)ri < (unsigned)a.rows && (unsigned)ci < (unsigned)a.cols;
      return inb ? (unsigned)(__mul24(ri, rowstride) + (ci * (RF * 4) + kbase)) : 0u;
#else
      return (unsigned)(__mul24(ri, rowstride) + (ci * (RF * 4) + kbase));  // v_mad_i32_i24 + v_lshl_add
#endif
    }
  };
  // the record at `off` as the RF operands of the product sums (compact: decoded, bit-identical to the dense record)
  struct Raw { uint32_t w[COMPACT ? CW : 1]; float4 q[COMPACT ? 1 : NV4]; };
  auto load_raw = [&](unsigned off, Raw& r) {
    if constexpr (COMPACT) cmap_load<CW>(crecb, off, r.w);
    else {
#pragma unroll
      for (int v = 0; v < NV4; v++) r.q[v] = *reinterpret_cast<const float4*>(recb + off + 16 * v);
    }
  };
  auto operands = [&](const Raw& r, float (&m)[RF]) {
    if constexpr (WIDE) cmap_decode_wide<RF, KSLOT>(r.w, ldict, m);
    else if constexpr (COMPACT) cmap_decode<RF, KSLOT>(r.w, ldict, m);
    else {
#pragma unroll
      for (int v = 0; v < NV4; v++) { m[4 * v] = r.q[v].x; m[4 * v + 1] = r.q[v].y; m[4 * v + 2] = r.q[v].z; m[4 * v + 3] = r.q[v].w; }
    }
  };

  float acc[RF];
#pragma unroll
  for (int k = 0; k < RF; k++) acc[k] = 0.f;
  float known = 0.f;
  // U samples per step: all addresses first, then all loads (map records + LDS scan records) in flight together, then
  // the FMAs.  The order of the FMAs — direction i ascending, ring ascending within it — is the same in every
  // instantiation and independent of U and of the launch: results are a pure function of the inputs.
  auto step = [&](auto full_step, const unsigned (&boff)[U], const float4* const (&sp)[U], int cnt) {
    constexpr bool FULL = decltype(full_step)::value;   // all U samples are real: no per-sample predicate
    Raw raw[U];
    float4 s[U][NV4];
    if constexpr (SKIP) {
      // the scan records first: a lane whose bin is empty in every class asks for the mask word instead of the record
#pragma unroll
      for (int u = 0; u < U; u++)
        if (FULL || u < cnt) {
#pragma unroll
          for (int v = 0; v < NV4; v++) s[u][v] = sp[u][v];
        }
      bool empty[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        if (FULL || u < cnt) {
          constexpr int ND = CmapShape<RF, KSLOT>::ND;
          uint32_t any = 0;
#pragma unroll
          for (int k = 0; k < ND; k++) {
            const float4 q = s[u][k / 4];
            any |= __float_as_uint(k % 4 == 0 ? q.x : (k % 4 == 1 ? q.y : (k % 4 == 2 ? q.z : q.w)));
          }
          empty[u] = skip_ok && any == 0;
          load_raw(empty[u] ? moff[u] : boff[u], raw[u]);
        }
#pragma unroll
      for (int u = 0; u < U; u++)
        if (FULL || u < cnt) {
          // an empty bin's "record": dictionary entry 0 for every distance, the known bit from the mask
          const uint32_t kb = (raw[u].w[0] >> (mbit[u] & 31)) & 1u;
#pragma unroll
          for (int d = 0; d < CW; d++) raw[u].w[d] = empty[u] ? kb : raw[u].w[d];
        }
    } else {
#pragma unroll
      for (int u = 0; u < U; u++)
        if (FULL || u < cnt) load_raw(boff[u], raw[u]);
#pragma unroll
      for (int u = 0; u < U; u++)
        if (FULL || u < cnt) {
#pragma unroll
          for (int v = 0; v < NV4; v++) s[u][v] = sp[u][v];
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
      if (FULL || u < cnt) {
        float m[RF];
        operands(raw[u], m);
#pragma unroll
        for (int v = 0; v < NV4; v++) {
          acc[4 * v + 0] = __builtin_fmaf(s[u][v].x, m[4 * v + 0], acc[4 * v + 0]);
          acc[4 * v + 1] = __builtin_fmaf(s[u][v].y, m[4 * v + 1], acc[4 * v + 1]);
          acc[4 * v + 2] = __builtin_fmaf(s[u][v].z, m[4 * v + 2], acc[4 * v + 2]);
          acc[4 * v + 3] = __builtin_fmaf(s[u][v].w, m[4 * v + 3], acc[4 * v + 3]);
        }
        if (!KSLOT) known += m[RF - 1];
      }
  };
  if (gn >= U) {
    // ray-major: consecutive samples of a lane are consecutive cells along one ray
    const int gfull = gn - gn % U;
    for (int i = 0; i < nb; i++) {
      int row = i + shift;
      row -= row >= nb ? nb : 0;
      const float4* const rl = ring + __mul24(row, rs);
      for (int jj = 0; jj < gfull; jj += U) {
        unsigned boff[U];
        const float4* sp[U];
        const float4* const rj = rl + jj * NV4;
#pragma unroll
        for (int u = 0; u < U; u++) {
          boff[u] = cell_offset(tab_at((int64_t)(j0 + jj + u) * nb + i), u);
          sp[u] = rj + u * NV4;
        }
        step(std::true_type{}, boff, sp, U);
      }
    }
    if (gfull < gn) {   // the group's last rings when gn is not a multiple of U: a pass of their own over the directions
      for (int i = 0; i < nb; i++) {
        int row = i + shift;
        row -= row >= nb ? nb : 0;
        const float4* const rl = ring + __mul24(row, rs);
        unsigned boff[U];
        const float4* sp[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int jc = min(gfull + u, gn - 1);
          boff[u] = cell_offset(tab_at((int64_t)(j0 + jc) * nb + i), u);
          sp[u] = rl + jc * NV4;
        }
        step(std::false_type{}, boff, sp, gn - gfull);
      }
    }
  } else {
    // fewer rings than loads to keep in flight (very long rows): U consecutive directions of one ring at a time
    for (int jj = 0; jj < gn; jj++) {
      const float4* const rj = ring + jj * NV4;
      for (int i = 0; i < nb; i += U) {
        unsigned boff[U];
        const float4* sp[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int ic = min(i + u, nb - 1);
          int row = ic + shift;
          row -= row >= nb ? nb : 0;
          boff[u] = cell_offset(tab_at((int64_t)(j0 + jj) * nb + ic), u);
          sp[u] = rj + __mul24(row, rs);
        }
        step(std::false_type{}, boff, sp, min(U, nb - i));
      }
    }
  }
  if (sbase + slot < a.npad) {
    float* o = a.part + (int64_t)blockIdx.y * (RF + 1) * a.npad + sbase + slot;
#pragma unroll
    for (int k = 0; k < RF; k++) o[(int64_t)k * a.npad] = acc[k];
    o[(int64_t)RF * a.npad] = KSLOT ? acc[RF - 2] : known;
  }
#ifdef TDR_SCORE_TIMELINE
  __syncthreads();
  if (threadIdx.x == 0 && tl_id < TDR_TL_MAX) g_timeline[2 * tl_id + 1] = wall_clock64();
#endif
}

// K2c: Cartesian scoring (BASELINE config 4).  The reference's StateParticle never reaches the Cartesian
// TopDownMap::getLocalMap, so the Cartesian score is DEFINED as: window sampled by getLocalMap
// (src/top_down_map.cpp:429-459 via samplePts :367-389) at rot = theta, res = res*scale, scored by getCostForRot
// with shift 0 (src/state_particle.cpp:132-143) (definition recorded in include/tdr.h:tdr_k_score_cart and DESIGN.md).
// Same mapping as the polar kernel (lane = particle); the rotation now lives in the sampling, so the scan record of
// sample (i,j) is the same for every lane and comes through the scalar cache instead of LDS.
__device__ __forceinline__ float linspaced_dev(int i, int size1, float low, float high, float step) {
  // Eigen LinSpaced<float>, |high| == |low| here, so never the flipped branch of linspaced_op_impl
  return (i == size1) ? high : (low + (float)i * step);
}

template <int NV4, int U, bool KSLOT, bool COMPACT, bool WIDE = false>
__global__ __launch_bounds__(256) void score_cart_kernel(CartArgs a) {
  constexpr int RF = 4 * NV4;
  static_assert(!WIDE || (COMPACT && NV4 == 2), "wide compact records: 8-float dense records only");
  constexpr int CW = WIDE ? 4 : CmapShape<RF, KSLOT>::CW, LC = WIDE ? 1 : CmapShape<RF, KSLOT>::LC;
  __shared__ float ldict[COMPACT ? (WIDE ? TDR_CMAP_WIDE_MAX_DICT : TDR_CMAP_MAX_DICT) : 1];
  if constexpr (COMPACT) {
    for (int t = threadIdx.x; t < a.dict_n; t += 256) ldict[t] = a.dict[t];
    __syncthreads();
  }
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
  const int rowstride = (a.map_cols + 2) * (RF * 4);
  const int kbase = (a.map_cols + 3) * (RF * 4);
  const float rmaxf = (float)a.map_rows, cmaxf = (float)a.map_cols;
  const char* __restrict__ recb = reinterpret_cast<const char*>(a.rec);
  const char* __restrict__ crecb = reinterpret_cast<const char*>(a.crec);
  const int ckcol = a.ctiles_r * 128 - 16 * CW, ckconst = a.ctiles_r * 128 + 128;   // cmap_offset
  // the scan record of sample (i, j) is the same for every lane: read through the CONSTANT address space = scalar loads
  // whatever else the kernel contains (see score_polar_kernel)
  typedef const float __attribute__((address_space(4))) * tdr_const_f;
  const tdr_const_f scanc = (tdr_const_f)a.scan_pk;

  float acc[RF];
#pragma unroll
  for (int k = 0; k < RF; k++) acc[k] = 0.f;
  float known = 0.f;
  // the record of one sample as RF operands: dense records as they are, compact ones decoded (bit-identical)
  auto fetch = [&](unsigned off, float (&m)[RF]) {
    if constexpr (COMPACT) {
      uint32_t w[CW];
      cmap_load<CW>(crecb, off, w);
      if constexpr (WIDE) cmap_decode_wide<RF, KSLOT>(w, ldict, m);
      else cmap_decode<RF, KSLOT>(w, ldict, m);
    } else {
#pragma unroll
      for (int v = 0; v < NV4; v++) {
        const float4 q = *reinterpret_cast<const float4*>(recb + off + 16 * v);
        m[4 * v + 0] = q.x; m[4 * v + 1] = q.y; m[4 * v + 2] = q.z; m[4 * v + 3] = q.w;
      }
    }
  };
  // rotm * pts (:383-385) for window sample (row value yi, column terms AB = {-s * xj, c * xj}), centre added (:387-388),
  // rounded (:437): both coordinates in packed instructions — the same float operations, two at a time
  auto cell_offset = [&](tdr_v2f cyi, tdr_v2f AB) -> unsigned {
    tdr_v2f pv = cyi + AB;         // p0 = c * yi + (-s * xj), p1 = s * yi + c * xj
    pv = pv + offv;
    tdr_v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
    qv = qv + 0.49999997f;         // see round_half_away_clamped
    int ri, ci;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ri) : "v"(qv.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ci) : "v"(qv.y));
    const bool inb = (unsigned)ri < (unsigned)a.map_rows && (unsigned)ci < (unsigned)a.map_cols;
    if constexpr (COMPACT) return cmap_offset<CW, LC>(ri, ci, ckcol, ckconst);
    else return inb ? (unsigned)(__mul24(ri, rowstride) + (ci * (RF * 4) + kbase)) : 0u;
  };
  auto column_terms = [&](int j) -> tdr_v2f {
    const float xj = linspaced_dev(j, c1, lo_c, hi_c, step_c);
    return (tdr_v2f){ns * xj, c * xj};
  };
  // Order: blocks of U window rows, and within a block the chunk's columns one after the other — a patch of U x cpc
  // neighbouring cells (rotated), so that a lane whose neighbours are far away meets each 4 x 4-cell tile of the compact
  // records once per patch instead of once per column.  (Fixed by the configuration: the sums are a pure function of it.)
  int i = 0;
  for (; i + U <= a.rows - 1; i += U) {   // LinSpaced without the select of its last element; the last row: below
    tdr_v2f cyi[U];
#pragma unroll
    for (int u = 0; u < U; u++) cyi[u] = cs * (lo_r + (float)(i + u) * step_r);
    for (int j = j0; j < j1; j++) {
      const tdr_v2f AB = column_terms(j);
      unsigned boff[U];
#pragma unroll
      for (int u = 0; u < U; u++) boff[u] = cell_offset(cyi[u], AB);
      float m[U][RF];
#pragma unroll
      for (int u = 0; u < U; u++) fetch(boff[u], m[u]);
      const int64_t sbase = ((int64_t)j * a.rows + i) * RF;  // wave-uniform
#pragma unroll
      for (int u = 0; u < U; u++) {
#pragma unroll
        for (int k = 0; k < RF; k++) acc[k] = __builtin_fmaf(scanc[sbase + u * RF + k], m[u][k], acc[k]);
        if (!KSLOT) known += m[u][RF - 1];
      }
    }
  }
  for (; i < a.rows; i++) {   // the remaining rows (the last one among them), column by column
    const tdr_v2f cyi = cs * linspaced_dev(i, r1, lo_r, hi_r, step_r);
    for (int j = j0; j < j1; j++) {
      float m[RF];
      fetch(cell_offset(cyi, column_terms(j)), m);
      const int64_t sbase = ((int64_t)j * a.rows + i) * RF;
#pragma unroll
      for (int k = 0; k < RF; k++) acc[k] = __builtin_fmaf(scanc[sbase + k], m[k], acc[k]);
      if (!KSLOT) known += m[RF - 1];
    }
  }
  if (slot < a.npad) {
    float* o = a.part + (int64_t)blockIdx.y * (RF + 1) * a.npad + slot;
#pragma unroll
    for (int k = 0; k < RF; k++) o[(int64_t)k * a.npad] = acc[k];
    o[(int64_t)RF * a.npad] = KSLOT ? acc[RF - 2] : known;
  }
}

// getLocalMap as a function of its own (src/top_down_map_polar.cpp:21-53, src/top_down_map.cpp:429-459): the window
// of ONE pose written out as the reference's arrays — dists [ncls][rows*cols] column-major images, mask [rows*cols]
// (1 = unknown / out of bounds).  The scoring kernels never materialise it; this is the class-surface method and the
// direct parity check of the window addressing (tests/test_gpu_parity.py::test_local_map_*).  One thread per sample,
// the same float operations as the scoring loops.
struct LocalMapArgs {
  const float* rec;
  int map_rows, map_cols, ncls, rf;
  float resolution;
  const float* tab;     // polar: [P][2]
  int rows, cols;       // window shape (polar: nb x nr)
  float cx, cy, scale_or_rot, res;
  float* dists;
  uint8_t* mask;
  int libm_fma;
};
template <bool POLAR>
__global__ void local_map_kernel(LocalMapArgs a) {
  const int64_t P = (int64_t)a.rows * a.cols;
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const float off0 = a.cy / a.resolution, off1 = a.cx / a.resolution;
  float p0, p1;
  if constexpr (POLAR) {
    p0 = (a.tab[2 * k] * a.scale_or_rot) * a.res + off0;      // top_down_map_polar.cpp:28-30
    p1 = (a.tab[2 * k + 1] * a.scale_or_rot) * a.res + off1;
  } else {
    const int i = (int)(k % a.rows), j = (int)(k / a.rows);
    const float resq = a.res / a.resolution;                    // top_down_map.cpp:434
    const float c = tdr_libm::cosf_v(a.scale_or_rot, a.libm_fma), s = tdr_libm::sinf_v(a.scale_or_rot, a.libm_fma);   // :381-385
    const float lo_r = (float)((double)(-resq * (float)(a.rows - 1)) / 2.), hi_r = (float)((double)(resq * (float)(a.rows - 1)) / 2.);
    const float lo_c = (float)((double)(-resq * (float)(a.cols - 1)) / 2.), hi_c = (float)((double)(resq * (float)(a.cols - 1)) / 2.);
    const float step_r = a.rows == 1 ? 0.f : (hi_r - lo_r) / (float)(a.rows - 1);
    const float step_c = a.cols == 1 ? 0.f : (hi_c - lo_c) / (float)(a.cols - 1);
    const float yi = linspaced_dev(i, a.rows == 1 ? 1 : a.rows - 1, lo_r, hi_r, step_r);
    const float xj = linspaced_dev(j, a.cols == 1 ? 1 : a.cols - 1, lo_c, hi_c, step_c);
    p0 = (c * yi + (-s) * xj) + off0;                           // samplePts :383-388
    p1 = (s * yi + c * xj) + off1;
  }
  p0 = __builtin_amdgcn_fmed3f(p0, -1.f, (float)a.map_rows);
  p1 = __builtin_amdgcn_fmed3f(p1, -1.f, (float)a.map_cols);
  const int ri = round_half_away_clamped(p0), ci = round_half_away_clamped(p1);
  const bool inb = (unsigned)ri < (unsigned)a.map_rows && (unsigned)ci < (unsigned)a.map_cols;
  const float* r = a.rec + ((int64_t)(ri + 1) * (a.map_cols + 2) + (ci + 1)) * a.rf;   // guard ring: always valid
  for (int c = 0; c < a.ncls; c++) a.dists[(int64_t)c * P + k] = inb ? r[c] : 0.f;
  a.mask[k] = inb ? (uint8_t)(r[a.rf - 1] == 0.f ? 1 : 0) : (uint8_t)1;
}
static int launch_local_map(bool polar, const tdr_map_desc* map, const float* tab, int rows, int cols, float cx, float cy,
                            float scale_or_rot, float res, float* dists_out, uint8_t* mask_out, void* stream) {
  if (!map || !map->rec || !dists_out || !mask_out || (polar && !tab))
    return fail(TDR_ERR_ARG, "local_map: null pointer");
  if (rows < 1 || cols < 1) return fail(TDR_ERR_ARG, "local_map: bad window shape");
  if (!(map->resolution > 0.f)) return fail(TDR_ERR_ARG, "local_map: map resolution must be > 0");
  LocalMapArgs a;
  a.rec = map->rec; a.map_rows = map->rows; a.map_cols = map->cols; a.ncls = map->ncls; a.rf = map->rec_floats;
  a.resolution = map->resolution; a.tab = tab; a.rows = rows; a.cols = cols; a.cx = cx; a.cy = cy;
  a.scale_or_rot = scale_or_rot; a.res = res; a.dists = dists_out; a.mask = mask_out;
  a.libm_fma = tdr_libm_fma();
  const dim3 grid((unsigned)cdiv((int64_t)rows * cols, 256)), block(256);
  if (polar) hipLaunchKernelGGL((local_map_kernel<true>), grid, block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((local_map_kernel<false>), grid, block, 0, (hipStream_t)stream, a);
  LAUNCH_CHECK("local_map");
  return TDR_OK;
}
extern "C" int tdr_k_local_map_polar(const tdr_map_desc* map, const float* tab, int nb, int nr, float cx, float cy,
                                     float scale, float res, float* dists_out, uint8_t* mask_out, void* stream) {
  return launch_local_map(true, map, tab, nb, nr, cx, cy, scale, res, dists_out, mask_out, stream);
}
extern "C" int tdr_k_local_map_cart(const tdr_map_desc* map, int rows, int cols, float cx, float cy, float rot,
                                    float res, float* dists_out, uint8_t* mask_out, void* stream) {
  return launch_local_map(false, map, nullptr, rows, cols, cx, cy, rot, res, dists_out, mask_out, stream);
}

// Gates of state_particle.cpp:163-176.  scale_lo / scale_hi = pow(10, scale_log_min/max) evaluated on the host
// (glibc pow, like the reference).
struct GateArgs {
  int force_on_map, scale_unknown;
  float width, height;  // map size * resolution (state_particle.cpp:11,46-47)
  double scale_lo, scale_hi;
};
static GateArgs make_gate(const tdr_filter_params* fp, const tdr_map_desc* map) {
  GateArgs g;
  g.force_on_map = fp->force_on_map;
  g.scale_unknown = fp->fixed_scale < 0;
  g.width = (float)map->cols * map->resolution;
  g.height = (float)map->rows * map->resolution;
  g.scale_lo = std::pow(10, fp->scale_log_min);
  g.scale_hi = std::pow(10, fp->scale_log_max);
  return g;
}
__device__ __forceinline__ bool particle_gated(const GateArgs& g, float cx, float cy, float scale) {
  if (g.force_on_map) {
    if (cx < 0 || cy < 0 || cx > g.width || cy > g.height) return true;  // :163-168
  }
  if (g.scale_unknown) {
    if ((double)scale < g.scale_lo || (double)scale > g.scale_hi) return true;  // :169-176
  }
  return false;
}

struct FinalizeArgs {
  const float* part;
  int rf, nchunks;
  int64_t npad, n, cap;
  const int32_t* order;
  const int32_t* count;
  float* st;
  tdr_filter_params fp;
  GateArgs gate;
  int64_t P;
  int ncls;
  int mode;             // 0: write raw weight; 1: init-search accumulate (best cost / theta)
  int first;            // mode 1: first rotation (initialise best)
  float theta_override;
  float* raw_w;
  float* best_cost;
  float* best_theta;
  // the geometric term of getCostForRot (state_particle.cpp:145-152, commented out in the reference; opt-in here):
  // partial sums of a second scoring launch over the 2-layer geometric map and the sums of the two geometric scan images
  const float* gpart;   // [gnchunks][5][npad], NULL = no geometric term
  int gnchunks;
  float gsum0, gsum1;
  int only_uninit;      // mode 1: slots whose particle already has a heading are left alone
  int tlog;             // 2^tlog neighbouring lanes share a slot's chunks (launch_finalize)
  const int32_t* run_if = nullptr;   // optional device words: run only when the integer form is off (see ScoreArgs)
  // the integer form (score_finalize_exact_kernel): `part` holds 64-bit integer sums
  const uint32_t* ipart = nullptr;   // [chunks][2 ncls + 2][npad]
  const uint32_t* dict_tail = nullptr;   // {q, ...}: a sum is a multiple of 2^-q
  const int32_t* counts = nullptr;   // {slots of the dense share: nchunks chunk rows each; ...; all slots}: the others have ray_split rows
  const int32_t* inexact = nullptr;
  int ray_split = 1;
};

__global__ __launch_bounds__(256) void score_finalize_kernel(FinalizeArgs a) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int T = 1 << a.tlog, t = (int)(gid & (T - 1));
  const int64_t slot = gid >> a.tlog;
  if (a.run_if && !int_form_off(a.run_if)) return;
  const int64_t nact = a.count ? (int64_t)*a.count : a.n;
  if (slot >= nact) return;
  const int64_t p = a.order ? (int64_t)a.order[slot] : slot;
  if (p < 0) return;   // a padding slot of the shift-uniform order (tdr_score_su.h)
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];
  if (a.mode == 0 && particle_gated(a.gate, cx, cy, scale)) {
    a.raw_w[p] = 0.f;
    return;
  }
  if (a.mode == 1 && a.only_uninit && a.st[TDR_ST_HAVE_INIT * a.cap + p] != 0.f) return;
  // Per-chunk partial sums -> double totals, chunk order ascending.  The loads of FIN_B chunks x all slots are issued
  // together (independent addresses, coalesced over the particles) before the dependent additions.  A small particle
  // set has many chunks and few slots — 128 x 1000 at the reference's test size — and one lane per slot would wait for
  // memory 32 times in a row: there 2^tlog neighbouring lanes take a contiguous share of a slot's chunks each and
  // their double totals are added up in a fixed butterfly.
  constexpr int FIN_B = 4, FIN_S = TDR_MAX_CLASSES + 2;   // slots: ncls class dots, normalisation, known count
  double tot[FIN_S];
#pragma unroll
  for (int k = 0; k < FIN_S; k++) tot[k] = 0;
  const int64_t cstride = (int64_t)(a.rf + 1) * a.npad;
  auto slot_row = [&](int k) { return k < a.ncls ? k : (k == a.ncls ? a.rf - 1 : a.rf); };
  const int share = (a.nchunks + T - 1) >> a.tlog, cend = min(a.nchunks, (t + 1) * share);
  int c0 = t * share;
  for (; c0 + FIN_B <= cend; c0 += FIN_B) {
    float v[FIN_B][FIN_S];
#pragma unroll
    for (int b = 0; b < FIN_B; b++)
#pragma unroll
      for (int k = 0; k < FIN_S; k++)
        if (k < a.ncls + 2) v[b][k] = a.part[(int64_t)(c0 + b) * cstride + (int64_t)slot_row(k) * a.npad + slot];
#pragma unroll
    for (int b = 0; b < FIN_B; b++)
#pragma unroll
      for (int k = 0; k < FIN_S; k++)
        if (k < a.ncls + 2) tot[k] += (double)v[b][k];
  }
  for (; c0 < cend; c0++) {
#pragma unroll
    for (int k = 0; k < FIN_S; k++)
      if (k < a.ncls + 2) tot[k] += (double)a.part[(int64_t)c0 * cstride + (int64_t)slot_row(k) * a.npad + slot];
  }
  if (T > 1) {   // (the lanes of a slot left or stayed together above)
    for (int sft = T >> 1; sft >= 1; sft >>= 1)
#pragma unroll
      for (int k = 0; k < FIN_S; k++)
        if (k < a.ncls + 2) tot[k] += __shfl_xor(tot[k], sft, 64);
    if (t != 0) return;
  }
  double known = 0, norm = 0;
#pragma unroll
  for (int k = 0; k < FIN_S; k++) {
    if (k == a.ncls) norm = tot[k];
    if (k == a.ncls + 1) known = tot[k];
  }
  // known fraction gate (state_particle.cpp:117-120); counts are exact integers in float
  float cost;
  if ((float)known / (float)a.P < 0.5) {
    cost = __builtin_nanf("");
  } else {
    cost = 0.f;
#pragma unroll
    for (int k = 0; k < TDR_MAX_CLASSES; k++)
      if (k < a.ncls) cost = (float)((double)cost + (double)(float)tot[k] * 0.01 * (double)a.fp.class_weights[k]);  // :136-139
    float normf = (float)norm;
    if (a.gpart) {   // :145-152: cost += (geo_i . geo_cls_i) * 0.01; normalization += geo_i.sum()
      double g[2] = {0, 0};
      const int64_t gstride = (int64_t)5 * a.npad;
      for (int c0g = 0; c0g < a.gnchunks; c0g++) {
        g[0] += (double)a.gpart[(int64_t)c0g * gstride + slot];
        g[1] += (double)a.gpart[(int64_t)c0g * gstride + a.npad + slot];
      }
      cost = (float)((double)cost + (double)(float)g[0] * 0.01);
      normf = normf + a.gsum0;
      cost = (float)((double)cost + (double)(float)g[1] * 0.01);
      normf = normf + a.gsum1;
    }
    cost = cost / normf;  // :154
  }
  if (a.mode == 0) {
    a.raw_w[p] = (float)(1. / (double)(cost + a.fp.regularization));  // :212
  } else {
    float best = a.first ? 3.402823466e+38f : a.best_cost[slot];
    float bt = a.first ? 0.f : a.best_theta[slot];
    if (cost < best) { best = cost; bt = a.theta_override; }  // :200-203 (NaN never wins)
    a.best_cost[slot] = best;
    a.best_theta[slot] = bt;
  }
}

// The integer form of a launch (tdr_score_su.hip, tdr_score_ray.hip): a slot's chunk rows hold, per class, the 64-bit
// integer sum of count x (distance 2^q) over the chunk's samples, and the normalisation and the known-cell count as
// integers.  Integer sums are exact: whatever kernel, split or order produced the rows, their total is the same number, and
// so is the weight.  dot_c = total 2^-q rounded to float once (the reference: a float sum of float products in Eigen's
// order, state_particle.cpp:136-138), then the reference's own arithmetic (:136-139, 154, 212).
// A workgroup = 64 slots x 4 shares of a slot's chunk rows (a large filter has 25 rows of 18 words per slot: one lane per
// slot walked them one after the other with a quarter of the waves); the shares meet in LDS — integers: any grouping.
__global__ __launch_bounds__(256) void score_finalize_exact_kernel(FinalizeArgs a) {
  __shared__ unsigned long long red[3][TDR_MAX_CLASSES + 2][64];
  if (int_form_off(a.inexact)) return;
  const int lane = threadIdx.x & 63, t = threadIdx.x >> 6;
  const int64_t nslots = (int64_t)a.counts[2];
  if ((int64_t)blockIdx.x * 64 >= nslots) return;   // (uniform)
  const int64_t slot = min((int64_t)blockIdx.x * 64 + lane, nslots - 1);
  const int rows = 2 * a.ncls + 2;
  const int nch = slot < (int64_t)a.counts[0] ? a.nchunks : a.ray_split;
  unsigned long long tot[TDR_MAX_CLASSES], norm = 0, known = 0;
#pragma unroll
  for (int k = 0; k < TDR_MAX_CLASSES; k++) tot[k] = 0;
  for (int c = t; c < nch; c += 4) {
    const uint32_t* o = a.ipart + (int64_t)c * rows * a.npad + slot;
#pragma unroll
    for (int k = 0; k < TDR_MAX_CLASSES; k++)
      if (k < a.ncls)
        tot[k] += (unsigned long long)o[(int64_t)(2 * k) * a.npad] | ((unsigned long long)o[(int64_t)(2 * k + 1) * a.npad] << 32);
    norm += o[(int64_t)(2 * a.ncls) * a.npad];
    known += o[(int64_t)(2 * a.ncls + 1) * a.npad];
  }
  if (t > 0) {
#pragma unroll
    for (int k = 0; k < TDR_MAX_CLASSES; k++)
      if (k < a.ncls) red[t - 1][k][lane] = tot[k];
    red[t - 1][TDR_MAX_CLASSES][lane] = norm;
    red[t - 1][TDR_MAX_CLASSES + 1][lane] = known;
  }
  __syncthreads();
  if (t > 0 || (int64_t)blockIdx.x * 64 + lane >= nslots) return;
  for (int u = 0; u < 3; u++) {
#pragma unroll
    for (int k = 0; k < TDR_MAX_CLASSES; k++)
      if (k < a.ncls) tot[k] += red[u][k][lane];
    norm += red[u][TDR_MAX_CLASSES][lane];
    known += red[u][TDR_MAX_CLASSES + 1][lane];
  }
  const int64_t p = a.order[slot];
  if (p < 0) return;   // a padding slot of the shift-uniform order
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];
  if (particle_gated(a.gate, cx, cy, scale)) {
    a.raw_w[p] = 0.f;
    return;
  }
  float cost;
  if ((float)known / (float)a.P < 0.5) {   // state_particle.cpp:117-120
    cost = __builtin_nanf("");
  } else {
    const int q = (int)a.dict_tail[0];
    cost = 0.f;
#pragma unroll
    for (int k = 0; k < TDR_MAX_CLASSES; k++)
      if (k < a.ncls) {
        const float dot = (float)ldexp((double)tot[k], -q);
        cost = (float)((double)cost + (double)dot * 0.01 * (double)a.fp.class_weights[k]);  // :136-139
      }
    cost = cost / (float)norm;  // :154
  }
  a.raw_w[p] = (float)(1. / (double)(cost + a.fp.regularization));  // :212
}

// nslots: slots the launch covers (a.n / a.count still bound the active ones)
static void launch_finalize(FinalizeArgs& f, int64_t nslots, hipStream_t s) {
  int tl = 0;
  while (tl < 4 && (f.nchunks >> (tl + 1)) >= 4 && (nslots << (tl + 1)) <= 131072) tl++;
  f.tlog = tl;
  hipLaunchKernelGGL(score_finalize_kernel, dim3((unsigned)cdiv(nslots << tl, 256)), dim3(256), 0, s, f);
}

// The 40-rotation initialisation search of state_particle.cpp:195-206 in ONE pass over the window: the candidate
// rotations are the same for every particle, so for rotation t the scan row paired with window row i — (i + s_t) mod nb
// — is the same for all lanes, and a map record gathered once is multiplied against all candidates' scan records
// (the reference also gathers once and scores 40 times).  One workgroup = one batch of 64 particles; its 4 waves
// split the candidates (INIT_TW each), gather the same records (the repeats hit L1), read the candidates' scan records
// through the scalar cache (they are wave-uniform) and keep INIT_TW x rf accumulators per lane.  Sums run in float over the whole window, which is only used to pick the best rotation:
// the weight itself is then produced by the regular scoring pass at that rotation.
#ifndef INIT_TW
#define INIT_TW 6          // candidate rotations per wave
#endif
#ifndef INIT_WAVES
#define INIT_WAVES 8       // waves per workgroup, all on the same 64 particles (A/B on MI355X, 250k particles:
#endif                     // 4x11 401 ms, 6x8 630 ms, 8x6 285 ms, 12x4 422 ms, 16x3 447 ms; scalar-cache scan reads 773 ms)
#define INIT_MAXROT (INIT_WAVES * INIT_TW)
struct InitArgs {
  const float* rec;
  int rows, cols;
  float resolution;
  const float* tab;
  const float* utab;
  const float* scan_pk;
  int nb, nr;
  float res;
  const float* st;     // read-only here: results go to res_theta / res_flag (keeps every other load scalarisable)
  int64_t cap, n;
  const int32_t* order;
  tdr_filter_params fp;
  GateArgs gate;
  int64_t P;
  int ncls;
  const int* nrot;
  const int* shift;    // [nrot] device arrays (filled by init_rot_kernel)
  const float* theta;
  const int* only_if;  // optional: the kernel runs only when this device word is non-zero (fallback after the MFMA pass)
  float* res_theta;  // [n] chosen rotation
  float* res_flag;   // [n] 0 = untouched, 1 = initialised, 2 = initialised but every rotation scored NaN
                     //     (weight 1/(FLT_MAX + reg), state_particle.cpp:193,212)
};

template <int NV4, bool KSLOT, bool USCALE>
__global__ __launch_bounds__(64 * INIT_WAVES) void score_init_kernel(InitArgs a) {
  constexpr int RF = 4 * NV4;
  constexpr int U = 1;
#if TDR_INIT_SCAN_LDS
  extern __shared__ float4 ring[];  // [NV4 planes][2*nb rows]
  const int nb2 = 2 * a.nb;
#endif
  __shared__ float x_cost[INIT_WAVES][64];
  __shared__ int x_rot[INIT_WAVES][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, and the compiler knows it
  if (a.only_if && *a.only_if == 0) return;               // (uniform) the MFMA pass already produced the results
  const int64_t slot = (int64_t)blockIdx.x * 64 + lane;   // all four waves work on the same 64 particles
  const bool valid = slot < a.n;
  const int64_t p = a.order ? (int64_t)a.order[valid ? slot : 0] : (valid ? slot : 0);
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];
  const bool want = valid && a.st[TDR_ST_HAVE_INIT * a.cap + p] == 0.f && !particle_gated(a.gate, cx, cy, scale);
  // every wave of the workgroup looks at the same 64 particles, so this per-wave vote is the same in all of them
  // (and, unlike __syncthreads_or, involves no LDS atomic that would stop the compiler from using scalar loads)
  if (__ballot(want) == 0) return;  // nothing to initialise in this batch
  const float off0 = cy / a.resolution, off1 = cx / a.resolution;
  const int rowstride = (a.cols + 2) * (RF * 4);
  const int kbase = (a.cols + 3) * (RF * 4);
  const float rmaxf = (float)a.rows, cmaxf = (float)a.cols;
  const char* __restrict__ recb = reinterpret_cast<const char*>(a.rec);
  const float2* __restrict__ tab2 = reinterpret_cast<const float2*>(USCALE ? a.utab : a.tab);
  const float4* __restrict__ scan4 = reinterpret_cast<const float4*>(a.scan_pk);
  const int nrot = *a.nrot;
  int sh[INIT_TW];
#pragma unroll
  for (int r = 0; r < INIT_TW; r++) {
    const int t = wave * INIT_TW + r;
    sh[r] = t < nrot ? a.shift[t] : 0;
  }
  float acc[INIT_TW][RF];
#pragma unroll
  for (int r = 0; r < INIT_TW; r++)
#pragma unroll
    for (int k = 0; k < RF; k++) acc[r][k] = 0.f;
  float known = 0.f;

  auto cell_offset = [&](float2 t) -> unsigned {
    float p0, p1;
    if constexpr (USCALE) { p0 = t.x; p1 = t.y; }
    else { p0 = (t.x * scale) * a.res; p1 = (t.y * scale) * a.res; }
    p0 = p0 + off0;
    p1 = p1 + off1;
    p0 = __builtin_amdgcn_fmed3f(p0, -1.f, rmaxf);
    p1 = __builtin_amdgcn_fmed3f(p1, -1.f, cmaxf);
    const int ri = round_half_away_clamped(p0), ci = round_half_away_clamped(p1);
    const bool inb = (unsigned)ri < (unsigned)a.rows && (unsigned)ci < (unsigned)a.cols;
    return inb ? (unsigned)(__mul24(ri, rowstride) + (ci * (RF * 4) + kbase)) : 0u;
  };

  for (int j = 0; j < a.nr; j++) {
    const float2* trow = tab2 + (int64_t)j * a.nb;
    const float4* srow = scan4 + (int64_t)j * a.nb * NV4;  // ring j of the packed scan
#if TDR_INIT_SCAN_LDS
    __syncthreads();
    for (int t = threadIdx.x; t < a.nb * NV4; t += 64 * INIT_WAVES) {
      const float4 v = srow[t];
      const int row = t / NV4, pl = t - row * NV4;
      ring[pl * nb2 + row] = v;
      ring[pl * nb2 + row + a.nb] = v;
    }
    __syncthreads();
#endif
    int i = 0;
    for (; i + U <= a.nb; i += U) {
      unsigned boff[U];
#pragma unroll
      for (int u = 0; u < U; u++) boff[u] = cell_offset(trow[i + u]);
      float4 m[U][NV4];
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int v = 0; v < NV4; v++) m[u][v] = *reinterpret_cast<const float4*>(recb + boff[u] + 16 * v);
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (!KSLOT) known += m[u][NV4 - 1].w;
#pragma unroll
        for (int r = 0; r < INIT_TW; r++) {
#pragma unroll
          for (int v = 0; v < NV4; v++) {
#if TDR_INIT_SCAN_LDS
            const float4 sv = ring[v * nb2 + sh[r] + i + u];  // same address in every lane: LDS broadcast
#else
            int row = sh[r] + i + u;                    // wave-uniform: the record comes through the scalar cache
            row -= row >= a.nb ? a.nb : 0;
            const float4 sv = srow[row * NV4 + v];
#endif
            acc[r][4 * v + 0] = __builtin_fmaf(sv.x, m[u][v].x, acc[r][4 * v + 0]);
            acc[r][4 * v + 1] = __builtin_fmaf(sv.y, m[u][v].y, acc[r][4 * v + 1]);
            acc[r][4 * v + 2] = __builtin_fmaf(sv.z, m[u][v].z, acc[r][4 * v + 2]);
            acc[r][4 * v + 3] = __builtin_fmaf(sv.w, m[u][v].w, acc[r][4 * v + 3]);
          }
        }
      }
    }
    for (; i < a.nb; i++) {
      const unsigned bo = cell_offset(trow[i]);
      float4 m[NV4];
#pragma unroll
      for (int v = 0; v < NV4; v++) m[v] = *reinterpret_cast<const float4*>(recb + bo + 16 * v);
      if (!KSLOT) known += m[NV4 - 1].w;
#pragma unroll
      for (int r = 0; r < INIT_TW; r++)
#pragma unroll
        for (int v = 0; v < NV4; v++) {
#if TDR_INIT_SCAN_LDS
          const float4 sv = ring[v * nb2 + sh[r] + i];
#else
          int row = sh[r] + i;
          row -= row >= a.nb ? a.nb : 0;
          const float4 sv = srow[row * NV4 + v];
#endif
          acc[r][4 * v + 0] = __builtin_fmaf(sv.x, m[v].x, acc[r][4 * v + 0]);
          acc[r][4 * v + 1] = __builtin_fmaf(sv.y, m[v].y, acc[r][4 * v + 1]);
          acc[r][4 * v + 2] = __builtin_fmaf(sv.z, m[v].z, acc[r][4 * v + 2]);
          acc[r][4 * v + 3] = __builtin_fmaf(sv.w, m[v].w, acc[r][4 * v + 3]);
        }
    }
  }
  // cost of each candidate (state_particle.cpp:117-120,136-139,154), best of this wave's candidates in order
  const float kn = KSLOT ? acc[0][RF - 2] : known;
  const bool unknown = (kn / (float)a.P) < 0.5;
  float cw[RF];
#pragma unroll
  for (int k = 0; k < RF; k++) cw[k] = k < 16 ? a.fp.class_weights[k < 16 ? k : 0] : 0.f;
  float best = 3.402823466e+38f;
  int best_t = -1;
#pragma unroll
  for (int r = 0; r < INIT_TW; r++) {
    const int t = wave * INIT_TW + r;
    float cost = 0.f;
#pragma unroll
    for (int k = 0; k < RF - 1; k++)   // constant indices only: a dynamic index would push the arguments to scratch
      if (k < a.ncls) cost = (float)((double)cost + (double)acc[r][k] * 0.01 * (double)cw[k]);
    cost = cost / acc[r][RF - 1];
    if (unknown) cost = __builtin_nanf("");
    if (t < nrot && cost < best) { best = cost; best_t = t; }  // :200-203 (NaN never wins)
  }
  x_cost[wave][lane] = best;
  x_rot[wave][lane] = best_t;
  __syncthreads();
  if (wave == 0 && want) {
    float b = 3.402823466e+38f;
    int bt = -1;
    for (int wv = 0; wv < INIT_WAVES; wv++)   // waves hold the candidates in loop order: strict '<' keeps the first minimum
      if (x_cost[wv][lane] < b) { b = x_cost[wv][lane]; bt = x_rot[wv][lane]; }
    a.res_theta[p] = bt >= 0 ? a.theta[bt] : 0.f;  // :205 (best_theta stays 0 if nothing won)
    a.res_flag[p] = bt < 0 ? 2.f : 1.f;
  }
}

// The same search on the matrix cores (records of 8 floats, i.e. 4-6 classes).  For one particle the 40 candidate
// costs are  cost[m] = sum_{i,j,c} scan_c[(i + s_m) mod nb, j] * (w_c d_c[cell(i,j)]):  a contraction over
// k = (sample, class) of a matrix A[m][k] that is the same for every particle (shifted scan records, from LDS) with
// the particle's gathered window B[k][n].  v_mfma_f32_16x16x32_f16: 16 rotations x 16 particles x (4 samples x 8
// record slots) per instruction.  Lane l holds, as its B fragment, the 8 slots of the record of particle l&15 at
// sample 4t + (l>>4) — exactly the record it gathered — and as its A fragment the packed scan record at row
// (4t + (l>>4) + s_m), m = l&15 (+16, +32 for the second and third tile of candidates), one ds_read_b128 each.
//   * scan counts are integers: exact in f16 up to 2048 (a larger count raises *inexact and score_init_kernel redoes
//     the search on the vector units);
//   * distances (times 0.01 w_c, in f32) are split hi + lo into two f16 (relative error <= 2^-20), two MFMAs;
//   * the normalisation  sum scanΣ * known  is a third MFMA with only slot 7 of B set; the known count is a plain add.
// Products are exact and accumulate in f32 like the vector version.  Only the choice of the rotation comes out of
// here; the weight itself is computed by the regular scoring pass at that rotation.
typedef _Float16 tdr_h8 __attribute__((ext_vector_type(8)));
typedef __fp16 tdr_h2 __attribute__((ext_vector_type(2)));   // what v_cvt_pkrtz_f16_f32 returns
typedef float tdr_f4 __attribute__((ext_vector_type(4)));
#define INITM_TILES 3   // 48 candidate rows >= the 40 (41) rotations of the search
static_assert(INITM_TILES * 16 >= INIT_MAXROT || INIT_MAXROT == 48, "rotation tiles");

// UNITW: all class weights are equal — a common factor does not move the minimum, so the distances go in unweighted.
// SEVEN: 7 classes — slot 6 of the record is a seventh distance (no spare slot), slot 7 still `known` / the scan's sum.
template <bool USCALE, bool UNITW, bool SEVEN>
__global__ __launch_bounds__(256) void score_init_mfma_kernel(InitArgs a, int* __restrict__ inexact) {
  constexpr int RF = 8;
  extern __shared__ uint4 ring16[];   // [2*nb] packed scan records as 8 x f16 (row r and r+nb hold scan row r) + 1 zero row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, q = lane >> 4;
  const int64_t slot = (int64_t)blockIdx.x * 64 + wave * 16 + col;
  const bool valid = slot < a.n;
  const int64_t p = a.order ? (int64_t)a.order[valid ? slot : 0] : (valid ? slot : 0);
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];
  const bool want = valid && a.st[TDR_ST_HAVE_INIT * a.cap + p] == 0.f && !particle_gated(a.gate, cx, cy, scale);
  if (!__syncthreads_or(want)) return;   // nothing to initialise in this batch of 64 particles
  const float off0 = cy / a.resolution, off1 = cx / a.resolution;
  const int rowstride = (a.cols + 2) * (RF * 4);
  const int kbase = (a.cols + 3) * (RF * 4);
  const float rmaxf = (float)a.rows, cmaxf = (float)a.cols;
  const char* __restrict__ recb = reinterpret_cast<const char*>(a.rec);
  const float2* __restrict__ tab2 = reinterpret_cast<const float2*>(USCALE ? a.utab : a.tab);
  const float4* __restrict__ scan4 = reinterpret_cast<const float4*>(a.scan_pk);
  const int nrot = *a.nrot;
  // byte offset of the lane's candidate row within the ring for every tile; candidates past nrot read the zero row
  const int zero_row = 2 * a.nb;
  int sh[INITM_TILES];
#pragma unroll
  for (int T = 0; T < INITM_TILES; T++) {
    const int m = 16 * T + col;
    sh[T] = m < nrot ? a.shift[m] : -1;
  }
  if (threadIdx.x == 0) ring16[zero_row] = make_uint4(0u, 0u, 0u, 0u);
  float wc[7];
#pragma unroll
  for (int c = 0; c < 7; c++) wc[c] = c < a.ncls ? (float)(0.01 * (double)a.fp.class_weights[c]) : 0.f;
  tdr_f4 accC[INITM_TILES], accN[INITM_TILES];
#pragma unroll
  for (int T = 0; T < INITM_TILES; T++) { accC[T] = (tdr_f4){0.f, 0.f, 0.f, 0.f}; accN[T] = (tdr_f4){0.f, 0.f, 0.f, 0.f}; }
  float known = 0.f;
  const int steps = (a.nb + 3) / 4;

  for (int j = 0; j < a.nr; j++) {
    const float2* trow = tab2 + (int64_t)j * a.nb;
    const float4* srow = scan4 + (int64_t)j * a.nb * 2;
    __syncthreads();
    bool big = false;
    for (int t = threadIdx.x; t < a.nb; t += 256) {
      const float4 v0 = srow[2 * t], v1 = srow[2 * t + 1];
      big |= v0.x > 2048.f || v0.y > 2048.f || v0.z > 2048.f || v0.w > 2048.f || v1.x > 2048.f || v1.y > 2048.f ||
             v1.z > 2048.f || v1.w > 2048.f;
      union { tdr_h2 h[4]; uint4 u; } pk;
      pk.h[0] = __builtin_amdgcn_cvt_pkrtz(v0.x, v0.y);
      pk.h[1] = __builtin_amdgcn_cvt_pkrtz(v0.z, v0.w);
      pk.h[2] = __builtin_amdgcn_cvt_pkrtz(v1.x, v1.y);
      pk.h[3] = __builtin_amdgcn_cvt_pkrtz(v1.z, v1.w);
      ring16[t] = pk.u;
      ring16[t + a.nb] = pk.u;
    }
    if (big) atomicOr(inexact, 1);
    __syncthreads();
    // software pipeline: the record of step t+1 (and the table entry of step t+2) are requested before the matrix
    // work of step t, so every wave keeps two gathers in flight
    auto tab_at = [&](int t) -> float2 { return trow[min(4 * t + q, a.nb - 1)]; };
    auto rec_addr = [&](float2 tv) -> const char* {
      float p0, p1;
      if constexpr (USCALE) { p0 = tv.x; p1 = tv.y; }
      else { p0 = (tv.x * scale) * a.res; p1 = (tv.y * scale) * a.res; }   // top_down_map_polar.cpp:28
      p0 = __builtin_amdgcn_fmed3f(p0 + off0, -1.f, rmaxf);
      p1 = __builtin_amdgcn_fmed3f(p1 + off1, -1.f, cmaxf);
      const int ri = round_half_away_clamped(p0), ci = round_half_away_clamped(p1);   // :31
      const bool inb = (unsigned)ri < (unsigned)a.rows && (unsigned)ci < (unsigned)a.cols;
      return recb + (inb ? (unsigned)(__mul24(ri, rowstride) + (ci * (RF * 4) + kbase)) : 0u);
    };
    float2 tv_next = tab_at(1);
    float4 n0, n1;
    {
      const char* r0 = rec_addr(tab_at(0));
      n0 = *reinterpret_cast<const float4*>(r0);
      n1 = *reinterpret_cast<const float4*>(r0 + 16);
    }
    for (int t = 0; t < steps; t++) {
      const int i = 4 * t + q;
      const bool in = i < a.nb;
      const int ic = in ? i : a.nb - 1;
      float4 m0 = n0, m1 = n1;
      {
        const char* r1 = rec_addr(tv_next);        // step t+1 (clamped to the ring: an in-range address)
        tv_next = tab_at(t + 2);
        n0 = *reinterpret_cast<const float4*>(r1);
        n1 = *reinterpret_cast<const float4*>(r1 + 16);
      }
      if (!in) { m0 = make_float4(0.f, 0.f, 0.f, 0.f); m1 = m0; }
      known += m1.w;
      float v[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, SEVEN ? m1.z : 0.f, 0.f};
      if constexpr (!UNITW) {
#pragma unroll
        for (int c = 0; c < 7; c++) v[c] *= wc[c];
      }
      union { tdr_h2 h[4]; tdr_h8 v8; } bh, bl, bn;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const tdr_h2 hi = __builtin_amdgcn_cvt_pkrtz(v[2 * c], v[2 * c + 1]);
        bh.h[c] = hi;
        bl.h[c] = __builtin_amdgcn_cvt_pkrtz(v[2 * c] - (float)hi[0], v[2 * c + 1] - (float)hi[1]);
      }
      const tdr_h2 zero2 = __builtin_amdgcn_cvt_pkrtz(0.f, 0.f);
      if constexpr (SEVEN) {
        const tdr_h2 hi = __builtin_amdgcn_cvt_pkrtz(v[6], 0.f);
        bh.h[3] = hi;
        bl.h[3] = __builtin_amdgcn_cvt_pkrtz(v[6] - (float)hi[0], 0.f);
      } else {
        bh.h[3] = zero2;
        bl.h[3] = zero2;
      }
      bn.h[0] = zero2; bn.h[1] = zero2; bn.h[2] = zero2;
      bn.h[3] = __builtin_amdgcn_cvt_pkrtz(0.f, m1.w);
      union { uint4 u; tdr_h8 v8; } av[INITM_TILES];
#pragma unroll
      for (int T = 0; T < INITM_TILES; T++) av[T].u = ring16[sh[T] < 0 ? zero_row : ic + sh[T]];
      // dependent MFMAs (same accumulator) are kept three instructions apart
#pragma unroll
      for (int T = 0; T < INITM_TILES; T++) accC[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[T].v8, bh.v8, accC[T], 0, 0, 0);
#pragma unroll
      for (int T = 0; T < INITM_TILES; T++) accN[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[T].v8, bn.v8, accN[T], 0, 0, 0);
#pragma unroll
      for (int T = 0; T < INITM_TILES; T++) accC[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[T].v8, bl.v8, accC[T], 0, 0, 0);
    }
  }
  // this lane holds rows 4q..4q+3 of every tile for particle `col`; the four lanes of a particle share the samples
  known += __shfl_xor(known, 16, 64);
  known += __shfl_xor(known, 32, 64);
  const bool unknown = (known / (float)a.P) < 0.5;   // state_particle.cpp:117-120
  float best = 3.402823466e+38f;
  int bm = -1;
#pragma unroll
  for (int T = 0; T < INITM_TILES; T++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int m = 16 * T + 4 * q + r;
      float cost = accC[T][r] / accN[T][r];             // :154
      if (unknown) cost = __builtin_nanf("");
      if (m < nrot && cost < best) { best = cost; bm = m; }   // :200-203 (NaN never wins)
    }
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {   // first minimum in rotation order over the particle's four lanes
    const float oc = __shfl_xor(best, o, 64);
    const int om = __shfl_xor(bm, o, 64);
    const bool take = om >= 0 && (bm < 0 || oc < best || (oc == best && om < bm));
    if (take) { best = oc; bm = om; }
  }
  if (q == 0 && want) {
    a.res_theta[p] = bm >= 0 ? a.theta[bm] : 0.f;  // :205 (best_theta stays 0 if nothing won)
    a.res_flag[p] = bm < 0 ? 2.f : 1.f;
  }
}

// The same for records of 12 and 16 floats (8-15 classes): a sample's record is two groups of 8 slots, each group its own
// pair of fragments — B from the gathered record, A from a second LDS image of the scan row — so a step of 4 samples is
// 2 x (hi + lo) products per tile instead of one, plus the normalisation product on the group that holds slot RF - 1.
// Class weights are always folded in (no unit-weight form); slots past the class count meet a zero weight.
template <int NV4, bool USCALE>
__global__ __launch_bounds__(256) void score_init_mfma_wide_kernel(InitArgs a, int* __restrict__ inexact) {
  constexpr int RF = 4 * NV4, NH = 2;
  static_assert(NV4 == 3 || NV4 == 4, "records of 12 or 16 floats");
  constexpr int HN = (RF - 1) / 8, KN = (RF - 1) % 8;   // group and slot of `known` / the scan's sum
  extern __shared__ uint4 ring16[];   // [2*nb + 1 rows][NH groups]: 8 x f16 each; rows r and r+nb hold scan row r, the last is zero
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, q = lane >> 4;
  const int64_t slot = (int64_t)blockIdx.x * 64 + wave * 16 + col;
  const bool valid = slot < a.n;
  const int64_t p = a.order ? (int64_t)a.order[valid ? slot : 0] : (valid ? slot : 0);
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];
  const bool want = valid && a.st[TDR_ST_HAVE_INIT * a.cap + p] == 0.f && !particle_gated(a.gate, cx, cy, scale);
  if (!__syncthreads_or(want)) return;   // nothing to initialise in this batch of 64 particles
  const float off0 = cy / a.resolution, off1 = cx / a.resolution;
  const int rowstride = (a.cols + 2) * (RF * 4);
  const int kbase = (a.cols + 3) * (RF * 4);
  const float rmaxf = (float)a.rows, cmaxf = (float)a.cols;
  const char* __restrict__ recb = reinterpret_cast<const char*>(a.rec);
  const float2* __restrict__ tab2 = reinterpret_cast<const float2*>(USCALE ? a.utab : a.tab);
  const float4* __restrict__ scan4 = reinterpret_cast<const float4*>(a.scan_pk);
  const int nrot = *a.nrot;
  const int zero_row = 2 * a.nb;
  int sh[INITM_TILES];
#pragma unroll
  for (int T = 0; T < INITM_TILES; T++) {
    const int m = 16 * T + col;
    sh[T] = m < nrot ? a.shift[m] : -1;
  }
  if (threadIdx.x < NH) ring16[zero_row * NH + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
  float wc[16];
#pragma unroll
  for (int c = 0; c < 16; c++) wc[c] = (c < a.ncls && c < TDR_MAX_CLASSES) ? (float)(0.01 * (double)a.fp.class_weights[c < TDR_MAX_CLASSES ? c : 0]) : 0.f;
  tdr_f4 accC[INITM_TILES], accN[INITM_TILES];
#pragma unroll
  for (int T = 0; T < INITM_TILES; T++) { accC[T] = (tdr_f4){0.f, 0.f, 0.f, 0.f}; accN[T] = (tdr_f4){0.f, 0.f, 0.f, 0.f}; }
  float known = 0.f;
  const int steps = (a.nb + 3) / 4;
  const tdr_h2 zero2 = __builtin_amdgcn_cvt_pkrtz(0.f, 0.f);

  for (int j = 0; j < a.nr; j++) {
    const float2* trow = tab2 + (int64_t)j * a.nb;
    const float4* srow = scan4 + (int64_t)j * a.nb * NV4;
    __syncthreads();
    bool big = false;
    for (int t = threadIdx.x; t < a.nb; t += 256) {
      float f[16];
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const float4 x = v < NV4 ? srow[NV4 * t + (v < NV4 ? v : 0)] : make_float4(0.f, 0.f, 0.f, 0.f);
        f[4 * v] = x.x; f[4 * v + 1] = x.y; f[4 * v + 2] = x.z; f[4 * v + 3] = x.w;
      }
#pragma unroll
      for (int k = 0; k < 16; k++) big |= f[k] > 2048.f;
#pragma unroll
      for (int h = 0; h < NH; h++) {
        union { tdr_h2 h2[4]; uint4 u; } pk;
#pragma unroll
        for (int c = 0; c < 4; c++) pk.h2[c] = __builtin_amdgcn_cvt_pkrtz(f[8 * h + 2 * c], f[8 * h + 2 * c + 1]);
        ring16[t * NH + h] = pk.u;
        ring16[(t + a.nb) * NH + h] = pk.u;
      }
    }
    if (big) atomicOr(inexact, 1);
    __syncthreads();
    auto tab_at = [&](int t) -> float2 { return trow[min(4 * t + q, a.nb - 1)]; };
    auto rec_addr = [&](float2 tv) -> const char* {
      float p0, p1;
      if constexpr (USCALE) { p0 = tv.x; p1 = tv.y; }
      else { p0 = (tv.x * scale) * a.res; p1 = (tv.y * scale) * a.res; }   // top_down_map_polar.cpp:28
      p0 = __builtin_amdgcn_fmed3f(p0 + off0, -1.f, rmaxf);
      p1 = __builtin_amdgcn_fmed3f(p1 + off1, -1.f, cmaxf);
      const int ri = round_half_away_clamped(p0), ci = round_half_away_clamped(p1);   // :31
      const bool inb = (unsigned)ri < (unsigned)a.rows && (unsigned)ci < (unsigned)a.cols;
      return recb + (inb ? (unsigned)(__mul24(ri, rowstride) + (ci * (RF * 4) + kbase)) : 0u);
    };
    float2 tv_next = tab_at(1);
    float4 nx[NV4];
    {
      const char* r0 = rec_addr(tab_at(0));
#pragma unroll
      for (int v = 0; v < NV4; v++) nx[v] = *reinterpret_cast<const float4*>(r0 + 16 * v);
    }
    for (int t = 0; t < steps; t++) {
      const int i = 4 * t + q;
      const bool in = i < a.nb;
      const int ic = in ? i : a.nb - 1;
      float v[16];
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = 0.f;
#pragma unroll
      for (int g = 0; g < NV4; g++) { v[4 * g] = nx[g].x; v[4 * g + 1] = nx[g].y; v[4 * g + 2] = nx[g].z; v[4 * g + 3] = nx[g].w; }
      {
        const char* r1 = rec_addr(tv_next);        // step t+1 (clamped to the ring: an in-range address)
        tv_next = tab_at(t + 2);
#pragma unroll
        for (int g = 0; g < NV4; g++) nx[g] = *reinterpret_cast<const float4*>(r1 + 16 * g);
      }
      if (!in) {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = 0.f;
      }
      const float kn = v[RF - 1];
      known += kn;
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] *= wc[k];   // (slots past the classes: weight 0)
      union { tdr_h2 h[4]; tdr_h8 v8; } bh[NH], bl[NH], bn;
#pragma unroll
      for (int h = 0; h < NH; h++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const tdr_h2 hi = __builtin_amdgcn_cvt_pkrtz(v[8 * h + 2 * c], v[8 * h + 2 * c + 1]);
          bh[h].h[c] = hi;
          bl[h].h[c] = __builtin_amdgcn_cvt_pkrtz(v[8 * h + 2 * c] - (float)hi[0], v[8 * h + 2 * c + 1] - (float)hi[1]);
        }
#pragma unroll
      for (int c = 0; c < 4; c++) bn.h[c] = zero2;
      bn.h[KN / 2] = (KN & 1) ? __builtin_amdgcn_cvt_pkrtz(0.f, kn) : __builtin_amdgcn_cvt_pkrtz(kn, 0.f);
      union { uint4 u; tdr_h8 v8; } av[INITM_TILES][NH];
#pragma unroll
      for (int T = 0; T < INITM_TILES; T++)
#pragma unroll
        for (int h = 0; h < NH; h++) av[T][h].u = ring16[(sh[T] < 0 ? zero_row : ic + sh[T]) * NH + h];
      // dependent MFMAs (same accumulator) are kept three instructions apart
#pragma unroll
      for (int h = 0; h < NH; h++) {
#pragma unroll
        for (int T = 0; T < INITM_TILES; T++) accC[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[T][h].v8, bh[h].v8, accC[T], 0, 0, 0);
        if (h == HN) {
#pragma unroll
          for (int T = 0; T < INITM_TILES; T++) accN[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[T][h].v8, bn.v8, accN[T], 0, 0, 0);
        }
#pragma unroll
        for (int T = 0; T < INITM_TILES; T++) accC[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[T][h].v8, bl[h].v8, accC[T], 0, 0, 0);
      }
    }
  }
  known += __shfl_xor(known, 16, 64);
  known += __shfl_xor(known, 32, 64);
  const bool unknown = (known / (float)a.P) < 0.5;   // state_particle.cpp:117-120
  float best = 3.402823466e+38f;
  int bm = -1;
#pragma unroll
  for (int T = 0; T < INITM_TILES; T++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int m = 16 * T + 4 * q + r;
      float cost = accC[T][r] / accN[T][r];             // :154
      if (unknown) cost = __builtin_nanf("");
      if (m < nrot && cost < best) { best = cost; bm = m; }   // :200-203 (NaN never wins)
    }
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {   // first minimum in rotation order over the particle's four lanes
    const float oc = __shfl_xor(best, o, 64);
    const int om = __shfl_xor(bm, o, 64);
    const bool take = om >= 0 && (bm < 0 || oc < best || (oc == best && om < bm));
    if (take) { best = oc; bm = om; }
  }
  if (q == 0 && want) {
    a.res_theta[p] = bm >= 0 ? a.theta[bm] : 0.f;  // :205 (best_theta stays 0 if nothing won)
    a.res_flag[p] = bm < 0 ? 2.f : 1.f;
  }
}

// ---- the matrix-core search on pre-split half records ------------------------------------------------------------------
// score_init_mfma_kernel spends most of its vector instructions turning a gathered f32 record into the f16 hi / lo
// operands (weights, two conversions and a subtraction per pair, the zeroing of ragged lanes), per sample per particle.
// That work depends on the cell alone.  half_records_kernel does it ONCE per cell into a 32-byte record
//     H = {hi_0 .. hi_5, hi_6 | 0, 0}   L = {lo_0 .. lo_5, lo_6 | 0, unknown}      (f16; hi + lo = w_c * 0.01 * d_c)
// laid out like the dense records (guarded row-major grid, guard cells = distance 0, unknown; one all-zero record behind
// the grid for lanes without a sample), so that a lane's two 16-byte loads ARE the B fragments: H as it is, L with its
// last half cleared.  The scan side is ONE LDS image per ring, {c_0 .. c_5, c_6 | 0, sum c}: its slot 7 meets a zero in H
// and in the cleared L.  The normalisation  sum_samples (sum c) * known  is taken as  S - sum_samples (sum c) * unknown
// with S the sum of the whole scan (the same for every candidate): the third product, {0 .. 0, unknown}, is issued only
// in steps where some lane of the wave met an unknown cell — none, for a window inside the mapped area.
// Same f16 operands as score_init_mfma_kernel, summed in another order (four rings of one direction per instruction).
// The records carry the class weights, so they are rebuilt at every search (one pass over the map, ~0.35 ms for 4000^2
// cells) into scratch memory the map's owner provides (tdr_map_desc.rec16).
// RF: floats of the dense record read (4: up to 3 classes, 8: 4 to 7) — the half record is the same 32 bytes for both
template <int RF>
__global__ __launch_bounds__(256) void half_records_kernel(const float4* __restrict__ rec, int64_t ncells, int unitw,
                                                           tdr_filter_params fp, int ncls, uint4* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  float m[RF];
#pragma unroll
  for (int q = 0; q < RF / 4; q++) {
    const float4 t = rec[(RF / 4) * c + q];
    m[4 * q] = t.x; m[4 * q + 1] = t.y; m[4 * q + 2] = t.z; m[4 * q + 3] = t.w;
  }
  const float known = m[RF - 1];
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = (k < RF - 1 && k < 7 && k < ncls) ? m[k < RF ? k : 0] : 0.f;   // the distances
  if (!unitw) {
#pragma unroll
    for (int k = 0; k < 7; k++) v[k] *= k < ncls ? (float)(0.01 * (double)fp.class_weights[k]) : 0.f;
  }
  union { tdr_h2 h[4]; uint4 u; } H, L;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const tdr_h2 hi = __builtin_amdgcn_cvt_pkrtz(v[2 * k], v[2 * k + 1]);   // v[7] == 0
    H.h[k] = hi;
    L.h[k] = __builtin_amdgcn_cvt_pkrtz(v[2 * k] - (float)hi[0], k == 3 ? 1.f - known : v[2 * k + 1] - (float)hi[1]);
  }
  out[2 * c] = H.u;
  out[2 * c + 1] = L.u;
  if (c == 0) {   // the record behind the grid: nothing at all (what lanes without a sample read)
    out[2 * ncells] = make_uint4(0u, 0u, 0u, 0u);
    out[2 * ncells + 1] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// Work of one MFMA (k = 4 samples x 8 slots): the SAME direction i on 4 consecutive range rings — four neighbouring cells
// along a ray for each of the wave's 16 (neighbouring) particles, so that one gather instruction touches few cache
// lines (the L1 looks up one line per clock: with four samples a quarter ring apart the counters showed 42 line
// accesses per instruction and the L1, not the matrix or the vector units, setting the pace).  Rings are staged four at
// a time (one LDS image per ring).
// AHEAD: record loads in flight — those of step t + AHEAD are issued before the matrix work of step t.  The step loop is
// unrolled AHEAD + 1 times so that the buffers rotate by name (no register copies); the step count is padded to a multiple
// of that, the padding steps read the zero guard record.
template <bool USCALE, int AHEAD>
__global__ __launch_bounds__(256) void score_init_half_kernel(InitArgs a, const uint4* __restrict__ rec16,
                                                              int* __restrict__ inexact, int img, int rfs) {
  constexpr int R = AHEAD + 1;
  // LDS: [4 rings][img] scan records {c0..c5, c6|0, sum c} as 8 x f16, row r and r + nb of an image hold scan row r; the
  // img - 2 nb >= R rows behind them stay zero (padding steps; the exact count is chosen on the host so that the four
  // images sit on the banks with the fewest conflicts, init_half_image_rows) — followed by the rings' sample-table rows,
  // [4][nb + 2 R] float2
  extern __shared__ uint4 ringh[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, q = lane >> 4;
  // (an XCD-contiguous order of the workgroups and 1, 2 or 3 record loads in flight all run within 1 %: A/B on MI355X)
  const int64_t slot = (int64_t)blockIdx.x * 64 + wave * 16 + col;
  const bool valid = slot < a.n;
  const int64_t p = a.order ? (int64_t)a.order[valid ? slot : 0] : (valid ? slot : 0);
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];
  const bool want = valid && a.st[TDR_ST_HAVE_INIT * a.cap + p] == 0.f && !particle_gated(a.gate, cx, cy, scale);
  if (!__syncthreads_or(want)) return;   // nothing to initialise in this batch of 64 particles
  typedef float tdr_v2f __attribute__((ext_vector_type(2)));
  const tdr_v2f offv = {cy / a.resolution, cx / a.resolution};
  const int rowstride = (a.cols + 2) * 32;
  const int kbase = (a.cols + 3) * 32;
  const float rmaxf = (float)a.rows, cmaxf = (float)a.cols;
  const char* __restrict__ recb = reinterpret_cast<const char*>(rec16);
  const float2* __restrict__ tab2 = reinterpret_cast<const float2*>(USCALE ? a.utab : a.tab);
  const int nrot = *a.nrot;
  const int nb = a.nb;
  const int ntab = nb + 2 * R;    // entries per table row in LDS
  float2* const ltab = reinterpret_cast<float2*>(ringh + 4 * img);
  // LDS byte address of this lane's candidate row at direction 0, per tile (the lane's ring image); candidates past nrot
  // read row 0 (their results are ignored)
  int arow[INITM_TILES];
#pragma unroll
  for (int T = 0; T < INITM_TILES; T++) {
    const int m = 16 * T + col;
    arow[T] = (q * img + (m < nrot ? a.shift[m] : 0)) * 16;
  }
  const int npad = img - 2 * nb;
  for (int t = threadIdx.x; t < 4 * npad; t += 256) ringh[(t / npad) * img + 2 * nb + t % npad] = make_uint4(0u, 0u, 0u, 0u);
  tdr_f4 accC[INITM_TILES], accN[INITM_TILES];   // accN: the normalisation's deficit, sum (sum c) * unknown
#pragma unroll
  for (int T = 0; T < INITM_TILES; T++) { accC[T] = (tdr_f4){0.f, 0.f, 0.f, 0.f}; accN[T] = (tdr_f4){0.f, 0.f, 0.f, 0.f}; }
  unsigned ucount = 0;   // 60 per unknown cell met
  float ssum = 0.f;      // this thread's share of S, the sum of the whole scan
  const int rounds = (nb + R - 1) / R;   // R directions each
  const char* const ringb = reinterpret_cast<const char*>(ringh);
  const float2* const ltq = ltab + q * ntab;
  const unsigned none_off = (unsigned)(a.rows + 2) * (unsigned)(a.cols + 2) * 32u;   // the all-zero record behind the grid

  // byte offset of the half record of table entry tv; a lane without a sample (ring >= nr, padding step) gets `none`
  auto rec_off = [&](float2 tv, bool in) -> unsigned {
    tdr_v2f pv = {tv.x, tv.y};
    if constexpr (!USCALE) pv = (pv * scale) * a.res;   // top_down_map_polar.cpp:28
    pv = pv + offv;                                      // :29-30
    tdr_v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
    qv = qv + 0.49999997f;                               // :31, see round_half_away_clamped
    int ri, ci;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ri) : "v"(qv.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ci) : "v"(qv.y));
    const unsigned off = (unsigned)(__mul24(ri, rowstride) + (ci * 32 + kbase));   // guard cells: distance 0, unknown
    return in ? off : none_off;
  };

  for (int j0 = 0; j0 < a.nr; j0 += 4) {
    __syncthreads();
    bool big = false;
    for (int t = threadIdx.x; t < 4 * nb; t += 256) {
      const int kb = t / nb, r = t - kb * nb;
      union { tdr_h2 h[4]; uint4 u; } pc;
      pc.u = make_uint4(0u, 0u, 0u, 0u);
      if (j0 + kb < a.nr) {
        // packed scan record of rfs floats: the class counts first, the sum of the counts last
        const float* srow = a.scan_pk + ((int64_t)(j0 + kb) * nb + r) * rfs;
        float cnt[8];
#pragma unroll
        for (int k = 0; k < 7; k++) cnt[k] = k < a.ncls ? srow[k] : 0.f;
        cnt[7] = srow[rfs - 1];
#pragma unroll
        for (int k = 0; k < 8; k++) big |= cnt[k] > 2048.f;
#pragma unroll
        for (int k = 0; k < 4; k++) pc.h[k] = __builtin_amdgcn_cvt_pkrtz(cnt[2 * k], cnt[2 * k + 1]);
        ssum += cnt[7];
      }
      ringh[kb * img + r] = pc.u;
      ringh[kb * img + r + nb] = pc.u;
    }
    for (int t = threadIdx.x; t < 4 * ntab; t += 256) {
      const int kb = t / ntab, r = t - kb * ntab;
      ltab[t] = tab2[(int64_t)min(j0 + kb, a.nr - 1) * nb + min(r, nb - 1)];
    }
    if (big) atomicOr(inexact, 1);
    __syncthreads();
    const bool ring_ok = j0 + q < a.nr;
    // Software pipeline: the two record loads of step t + AHEAD are issued before the matrix work of step t (the table
    // entry comes from LDS, so the address costs no trip to memory).  The scheduling barriers keep the compiler from
    // sinking the loads next to their use, which would expose a full memory latency in every step.
    uint4 h[R], l[R];
#pragma unroll
    for (int k = 0; k < AHEAD; k++) {
      const char* r = recb + rec_off(ltq[k], ring_ok && k < nb);
      h[k] = *reinterpret_cast<const uint4*>(r);
      l[k] = *reinterpret_cast<const uint4*>(r + 16);
    }
    int ar[INITM_TILES];
#pragma unroll
    for (int T = 0; T < INITM_TILES; T++) ar[T] = arow[T];
    int inext = AHEAD;   // direction of the loads issued next
    for (int rd = 0; rd < rounds; rd++) {
#pragma unroll
      for (int u = 0; u < R; u++) {
        {
          const char* r = recb + rec_off(ltq[inext], ring_ok && inext < nb);
          h[(u + AHEAD) % R] = *reinterpret_cast<const uint4*>(r);
          l[(u + AHEAD) % R] = *reinterpret_cast<const uint4*>(r + 16);
          inext++;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (rd * R + u < nb) {   // (uniform) not a padding step
          union { uint4 u4; tdr_h8 v8; } bh, bl, bn;
          bh.u4 = h[u];
          bl.u4 = l[u];
          bn.u4 = make_uint4(0u, 0u, 0u, bl.u4.w & 0xFFFF0000u);   // {0 .. 0, unknown}
          bl.u4.w &= 0x0000FFFFu;
          ucount += bn.u4.w >> 24;   // f16 1.0 = 0x3C00: its high byte, 60 per unknown cell
          union { uint4 u4; tdr_h8 v8; } ac[INITM_TILES];
#pragma unroll
          for (int T = 0; T < INITM_TILES; T++) ac[T].u4 = *reinterpret_cast<const uint4*>(ringb + ar[T]);
#pragma unroll
          for (int T = 0; T < INITM_TILES; T++) accC[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ac[T].v8, bh.v8, accC[T], 0, 0, 0);
          if (__builtin_amdgcn_ballot_w64(bn.u4.w != 0u) != 0) {   // (uniform) some lane met an unknown cell
#pragma unroll
            for (int T = 0; T < INITM_TILES; T++) accN[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ac[T].v8, bn.v8, accN[T], 0, 0, 0);
          }
#pragma unroll
          for (int T = 0; T < INITM_TILES; T++) accC[T] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ac[T].v8, bl.v8, accC[T], 0, 0, 0);
        }
#pragma unroll
        for (int T = 0; T < INITM_TILES; T++) ar[T] += 16;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // S: every thread staged its share of every ring
  __syncthreads();
  float* const red = reinterpret_cast<float*>(ringh);
  red[threadIdx.x] = ssum;
  __syncthreads();
  float stotal = 0.f;
  for (int t = 0; t < 256; t++) stotal += red[t];   // same order in every lane
  // samples this lane went through: nb directions on each of its rings j = q, q + 4, ...
  const int my_rings = (a.nr - q + 3) / 4;
  float known = (float)(my_rings * nb - (int)(ucount / 60u));
  known += __shfl_xor(known, 16, 64);
  known += __shfl_xor(known, 32, 64);
  const bool unknown = (known / (float)a.P) < 0.5;   // state_particle.cpp:117-120
  float best = 3.402823466e+38f;
  int bm = -1;
#pragma unroll
  for (int T = 0; T < INITM_TILES; T++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int m = 16 * T + 4 * q + r;
      float cost = accC[T][r] / (stotal - accN[T][r]);  // :154
      if (unknown) cost = __builtin_nanf("");
      if (m < nrot && cost < best) { best = cost; bm = m; }   // :200-203 (NaN never wins)
    }
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {   // first minimum in rotation order over the particle's four lanes
    const float oc = __shfl_xor(best, o, 64);
    const int om = __shfl_xor(bm, o, 64);
    const bool take = om >= 0 && (bm < 0 || oc < best || (oc == best && om < bm));
    if (take) { best = oc; bm = om; }
  }
  if (q == 0 && want) {
    a.res_theta[p] = bm >= 0 ? a.theta[bm] : 0.f;  // :205 (best_theta stays 0 if nothing won)
    a.res_flag[p] = bm < 0 ? 2.f : 1.f;
  }
}

// candidate rotations of the search, generated exactly like the reference's loop (state_particle.cpp:197: float t,
// double increment) together with their bin shifts (:124-128)
__global__ void init_rot_kernel(int nb, int* __restrict__ shift, float* __restrict__ theta, int* __restrict__ nrot) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  nrot[1] = 0;   // the "scan counts too large for f16" flag of score_init_mfma_kernel
  int k = 0;
  for (float t = 0; t < 2 * M_PI; t += 2 * M_PI / 40) {
    if (k >= INIT_MAXROT) break;
    theta[k] = t;
    shift[k] = rot_shift_dev(t, nb);
    k++;
  }
  *nrot = k;
}

// state_.theta = best_theta; state_.have_init = true (state_particle.cpp:205-206)
__global__ void init_apply_kernel(const float* __restrict__ res_theta, const float* __restrict__ res_flag, int64_t n,
                                  float* __restrict__ st, int64_t cap) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n && res_flag[p] != 0.f) {
    st[TDR_ST_THETA * cap + p] = res_theta[p];
    st[TDR_ST_HAVE_INIT * cap + p] = 1.f;
  }
}
// particles whose init search found no valid rotation keep best_cost = FLT_MAX (:193) -> weight 1/(FLT_MAX + reg)
__global__ void init_fixup_kernel(const float* __restrict__ res_flag, int64_t n, float regularization,
                                  float* __restrict__ raw_w) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n && res_flag[p] == 2.f) raw_w[p] = (float)(1. / (double)(3.402823466e+38f + regularization));
}

static int g_init_mfma = 1;   // 0 = vector-unit search only (A/B and debugging)
static bool init_use_mfma() { return g_init_mfma != 0; }
extern "C" int tdr_config_init_mfma(int on) {   // < 0: query only
  if (on >= 0) g_init_mfma = on ? 1 : 0;
  return g_init_mfma;
}
// the Cartesian kernel likes twice as many, shorter waves (A/B on MI355X, config 4: x1 183 ms, x2 179 ms, x4 177 ms)
#define TDR_CART_WAVE_MUL 2
// Tuning knobs of the scoring launches (tdr_config_tuning, include/tdr.h) — defaults here, no environment variables: the
// library reads nothing from the process environment.
// score_waves: many short waves beat few long ones (A/B on MI355X, config 2: 16k waves 21.8 ms, 128k 15.2 ms): workgroups
// of one ring chunk run together, so the concurrently touched part of the map is a thin annulus that L2 can hold, and
// the slow (scattered) batches no longer leave a long tail.
static int64_t g_score_waves = 131072;
static int g_score_group = 0;   // 0: from the shapes (score_group_rings)
static int g_su_group = 0;      // 0: from the shapes (tdr_score_workspace)
static int g_init_ahead = 1;    // record loads the init search keeps in flight per wave (1..3)
static int64_t score_wave_target() { return g_score_waves; }
extern "C" int tdr_config_prefix_head(int);        // tdr_prefix.hip
extern "C" int tdr_config_ray_block_major(int);    // tdr_score_ray.hip
extern "C" int tdr_config_ray_patch(int);
extern "C" int tdr_config_ray_borrow(int);
extern "C" int tdr_config_cart_seg_rows(int);      // tdr_score_cart.hip
extern "C" int tdr_config_mt_stretches(int);       // tdr_rng.hip
extern "C" int tdr_config_su_wave_span(int);       // tdr_score_su.hip
extern "C" int tdr_config_su_lds_pad(int);
extern "C" int64_t tdr_config_tuning(const char* name, int64_t value) {   // value < 0: query only
  if (!name) return -1;
  const std::string n(name);
  if (n == "score_waves") { if (value > 0) g_score_waves = value; return g_score_waves; }
  if (n == "score_group") { if (value >= 0) g_score_group = (int)value; return g_score_group; }
  if (n == "su_group") { if (value >= 0) g_su_group = (int)value; return g_su_group; }
  if (n == "init_ahead") { if (value >= 1) g_init_ahead = (int)std::min<int64_t>(value, 3); return g_init_ahead; }
  if (n == "prefix_head") return tdr_config_prefix_head((int)std::max<int64_t>(value, -1));
  if (n == "ray_borrow") return tdr_config_ray_borrow((int)std::max<int64_t>(value, -1));
  if (n == "ray_patch") return tdr_config_ray_patch((int)std::max<int64_t>(value, -1));
  if (n == "ray_block_major") return tdr_config_ray_block_major((int)std::max<int64_t>(value, -1));
  if (n == "cart_seg_rows") return tdr_config_cart_seg_rows((int)std::max<int64_t>(value, -1));
  if (n == "mt_stretches") return tdr_config_mt_stretches((int)std::max<int64_t>(value, -1));
  if (n == "su_lds_pad") return tdr_config_su_lds_pad((int)std::max<int64_t>(value, -1));
  if (n == "su_wave_span") return tdr_config_su_wave_span((int)std::max<int64_t>(value, -1));
  return -1;
}
static void choose_chunks(int64_t n, int nr, int& rpc, int& nchunks, int target_mul = 1) {
  int64_t nbatches = cdiv(std::max<int64_t>(n, 1), 64);
  int64_t want = std::max<int64_t>(1, cdiv(score_wave_target() * target_mul, nbatches));  // enough waves to fill the chip
  nchunks = (int)std::min<int64_t>(nr, want);
  rpc = (int)cdiv(nr, nchunks);
  nchunks = (int)cdiv(nr, rpc);
}

__global__ void utab_kernel(const float* __restrict__ tab, int64_t n2, float scale, float res,
                            float* __restrict__ utab) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n2) utab[k] = (tab[k] * scale) * res;  // `ang_sample_pts_*scale*res` (top_down_map_polar.cpp:28)
}

// Rings per workgroup of score_polar_kernel.  Larger groups give a ray more consecutive cells (tile reuse) and fewer,
// longer workgroups; a small filter needs many short ones to fill 256 CUs.  So the group is sized from the image shape
// (the group's scan rows must fit the LDS budget; 32 KB keeps four workgroups per CU; at most 8 rings; whole steps of
// TDR_SCORE_U) and from the TOTAL particle count of the filter — n_total, the same on every rank of a sharded filter,
// never the size of one launch or shard: the partition of a particle's score into partial sums is then the same in an
// N-rank run as in the 1-rank run.  Aim: >= 2048 workgroups.  tdr_config_tuning("score_group", g) overrides (tuning).
static int score_group_rings(int nb, int nr, int rf, int64_t n_total) {
  const int forced = g_score_group;
  const int64_t ring_bytes = std::max<int64_t>((int64_t)nb * rf * 4, 1);
  int g = (int)std::min<int64_t>(8, (32 * 1024) / ring_bytes);
  const int64_t chunks_wanted = cdiv(2048, cdiv(std::max<int64_t>(n_total, 1), 256));
  g = (int)std::min<int64_t>(g, std::max<int64_t>(1, nr / std::max<int64_t>(chunks_wanted, 1)));
  if (forced > 0) g = forced;
  g = std::max(1, std::min<int>(g, (int)((60 * 1024) / ring_bytes)));
  if (g >= TDR_SCORE_U) g -= g % TDR_SCORE_U;
  return std::max(g, 1);
}
// bytes of the scratch behind tdr_map_desc.rec16 (0: this record size has no matrix-core search)
extern "C" size_t tdr_map_rec16_bytes(int ncls, int rows, int cols) {
  if (ncls < 1 || ncls > 7 || rows < 1 || cols < 1) return 0;
  // the search addresses this grid with 32-bit byte offsets and a 24-bit row multiply: a map beyond that has no half
  // records (0: the caller passes none and the search splits the dense records on the fly)
  const uint64_t bytes = (uint64_t)(rows + 2) * (uint64_t)(cols + 2) * 32 + 32;   // + the all-zero record behind the grid
  if (bytes > 0xFFFFFFFFull || (uint64_t)(cols + 2) * 32 >= (1u << 24)) return 0;
  return (size_t)bytes;
}
// Rebuilding the half records is one pass over the whole map: it pays from a few thousand particles on (4000^2 cells:
// 0.35 ms, the price of searching ~2000 particles with 256 x 256 windows on the fly).  Filters below the threshold
// ignore the scratch — the filter's TOTAL particle count decides (n_total, the same on every rank), so that the ranks of
// a sharded filter take the kernel the one-rank filter takes and choose the same rotations where candidates tie.
// tdr_config_rec16_min_particles(INT64_MAX) turns the path off (A/B).
static int64_t g_rec16_min = 8192;
extern "C" int64_t tdr_config_rec16_min_particles(int64_t n) {   // < 0: query only
  if (n >= 0) g_rec16_min = n;
  return g_rec16_min;
}
// Rows per LDS scan image of score_init_half_kernel: 2 nb + R + c with the c in [0, 16) that gives the ds_read_b128 of
// the candidates' rows the fewest bank conflicts.  A lane (candidate m, ring q) reads row q * img + i + shift_m; the LDS
// serves the instruction in four groups of 16 lanes (MI355X_MICROARCH.md, LDS) and two lanes of a group collide when
// their rows differ by a multiple of 16.  The shifts are multiples of nb / 40, so only a few residues occur and the
// image stride decides how the rings' residues interleave (nb = 256: 3.7 LDS cycles per read at the worst stride, 2.0 at
// the best).
static int init_half_image_rows(int nb, int R) {
  int sh[48] = {0};
  int k = 0;
  for (float t = 0; t < 2 * M_PI && k < 48; t += 2 * M_PI / 40) {   // as init_rot_kernel / rot_shift_dev
    int s = (int)round((double)(t * (float)nb / 2) / M_PI);
    s %= nb;
    if (s < 0) s += nb;
    sh[k++] = s;
  }
  static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                 {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                 {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                 {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  int best_c = 0;
  long best = -1;
  for (int c = 0; c < 16; c++) {
    const int img = 2 * nb + R + c;
    long tot = 0;
    for (int T = 0; T < 3; T++)
      for (int i = 0; i < 16; i++)   // the pattern repeats with i mod 16
        for (int g = 0; g < 4; g++) {
          int rows[16], worst = 1;
          for (int l = 0; l < 16; l++) rows[l] = (grp[g][l] >> 4) * img + i + sh[16 * T + (grp[g][l] & 15)];
          for (int x = 0; x < 16; x++) {
            int distinct = 1;   // distinct rows on the bank quad of rows[x]
            for (int y = 0; y < x; y++)
              if ((rows[y] - rows[x]) % 16 == 0 && rows[y] != rows[x]) {
                bool seen = false;
                for (int z = 0; z < y; z++) seen |= rows[z] == rows[y];
                if (!seen) distinct++;
              }
            if (distinct > worst) worst = distinct;
          }
          tot += worst;
        }
    if (best < 0 || tot < best) { best = tot; best_c = c; }
  }
  return 2 * nb + R + best_c;
}
// The scoring workspace (floats): [partial sums nchunks*(rf+1)*npad_part][res_flag | best_cost npad][res_theta |
// best_theta npad][list npad + 64: rotation table of the init search][uniform-scale table 2*nb*nr][shift-uniform order,
// tdr_score_su.h].  npad_part = the slot count of the shift-uniform order where the shapes allow it (its partial sums are
// slot-indexed and the slots include the padding), else npad.
extern "C" int tdr_cmap_words(int ncls);   // tdr_cmap.hip
struct ScoreWs {
  int group, nchunks;
  int su_group, su_nchunks;   // the ring groups of the shift-uniform kernel (integer sums: any partition gives the same bits)
  int64_t npad, npad_part, off_aux, off_utab, off_su, total;
  bool su;
  SuWs suw;
};
static ScoreWs score_ws(int ncls, int nb, int nr, int64_t n, int64_t n_total) {
  ScoreWs w;
  const int rf = tdr_rec_floats(ncls);
  if (n_total <= 0) n_total = n;
  w.group = score_group_rings(nb, nr, rf, n_total);
  // the shift-uniform kernel steps through a ray four rings at a time: where it can take the launch the groups are whole
  // fours (a small filter's single-ring groups — the reference's 20 000 particles on 100 x 25 bins — become groups of 4; the
  // last group of an image whose ring count is no multiple of 4 is ragged)
  if (w.group % 4 != 0 && tdr_cmap_words(ncls) != 0 && tdr_su_shape_ok(nb, nr, 4, n_total))
    w.group = std::max(4, w.group - w.group % 4);
  w.nchunks = (int)cdiv(nr, w.group);
  w.npad = cdiv(std::max<int64_t>(n, 1), 64) * 64;
  w.su = tdr_su_shape_ok(nb, nr, w.group, n_total) && tdr_cmap_words(ncls) != 0;
  w.su_group = w.group;
  {
    const int forced = g_su_group;
    // eight rings per group where that divides the image and still leaves thousands of workgroups: a sector's mask is
    // staged half as often (config 2: 3.50 against 3.56 ms; 16 rings: 3.69, the staged boxes grow)
    if (w.su && w.group == 4 && nr % 8 == 0 && cdiv(n_total, 256) * (nr / 8) >= 4096) w.su_group = 8;
    if (w.su && forced >= 4 && forced % 4 == 0) w.su_group = forced;
  }
  w.su_nchunks = (int)cdiv(nr, w.su_group);
  w.npad_part = w.su ? su_npad(std::max<int64_t>(n, 1), nb) : w.npad;
  // partial sums: [chunks][rows][slots] — float form rf + 1 rows; integer form (tdr_score_su.hip) 2 ncls + 2 rows of
  // words and room for the chunk rows of a scattered particle's window (tdr_score_ray.hip)
  w.off_aux = w.su ? (int64_t)std::max(std::max(w.nchunks, w.su_nchunks), TDR_RAY_MAX_SPLIT) * std::max(rf + 1, 2 * ncls + 2) * w.npad_part
                   : (int64_t)w.nchunks * (rf + 1) * w.npad_part;
  w.off_utab = w.off_aux + 3 * w.npad + 64;
  w.off_su = (w.off_utab + 2 * (int64_t)nb * nr + 63) / 64 * 64;
  w.total = w.off_su;
  if (w.su) {
    w.suw = tdr_su_ws(nb, nr, w.su_group, std::max<int64_t>(n, 1));
    w.total += w.suw.total;
  }
  return w;
}
extern "C" size_t tdr_score_workspace_floats(int ncls, int nb, int nr, int64_t n, int64_t n_total) {
  if (ncls < 1 || ncls > TDR_MAX_CLASSES || nb < 1 || nr < 1 || n < 0) return 0;
  return (size_t)score_ws(ncls, nb, nr, n, n_total).total;
}
static int fill_utab(ScoreArgs& a, float* workspace, const ScoreWs& W, float uniform_scale, hipStream_t s) {
  a.utab = nullptr;
  if (!(uniform_scale > 0.f)) return TDR_OK;
  float* ut = workspace + W.off_utab;
  const int64_t n2 = 2 * (int64_t)a.nb * a.nr;
  hipLaunchKernelGGL(utab_kernel, dim3((unsigned)cdiv(n2, 256)), dim3(256), 0, s, a.tab, n2, uniform_scale, a.res, ut);
  LAUNCH_CHECK("utab");
  a.utab = ut;
  return TDR_OK;
}

// Optional in-library timing of the dominant kernel (bench.py's roofline figure): HIP events recorded on the launch
// stream right around score_polar_kernel, read back after the timed region.
static bool g_prof_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;
static size_t g_prof_used = 0;
struct ScoreProfScope {
  hipStream_t s;
  hipEvent_t stop = nullptr;
  explicit ScoreProfScope(hipStream_t s_, bool on = true) : s(s_) {
    if (!g_prof_on || !on) return;
    if (g_prof_used == g_prof_events.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      g_prof_events.emplace_back(a, b);
    }
    auto& ev = g_prof_events[g_prof_used++];
    (void)hipEventRecord(ev.first, s);
    stop = ev.second;
  }
  ~ScoreProfScope() {
    if (stop) (void)hipEventRecord(stop, s);
  }
};
// ... and of each of the two kernels of an integer-form launch on its own: the last launch's {dense, scattered} durations
// and the device words that say how many particles each share held (tdr_profile_shares).
static hipEvent_t g_share_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
static bool g_share_valid = false;
static const int32_t* g_share_counts = nullptr;
struct ShareProfScope {
  hipStream_t s;
  hipEvent_t stop = nullptr;
  ShareProfScope(int which, hipStream_t s_, bool on) : s(s_) {
    if (!g_prof_on || !on) return;
    for (int k = 0; k < 2; k++)
      if (!g_share_ev[which][k] && hipEventCreate(&g_share_ev[which][k]) != hipSuccess) return;
    (void)hipEventRecord(g_share_ev[which][0], s);
    stop = g_share_ev[which][1];
  }
  ~ShareProfScope() {
    if (stop) (void)hipEventRecord(stop, s);
  }
};
// ... and which loop VARIANT the dense kernels' waves ran, counted on the device while profiling is on (16 counters, see
// tdr_profile_variants in tdr.h): what fraction of the wave-sectors / wave-segments found every reachable cell known.
static uint32_t* g_variant_stats = nullptr;
static bool g_prof_variants = false;   // tdr_profile_enable(2): the counters cost the kernels an atomic per wave-sector
uint32_t* tdr_profile_stats_ptr() {   // NULL unless the variant counters are on (the kernels then count nothing)
  if (!g_prof_on || !g_prof_variants) return nullptr;
  if (!g_variant_stats) {
    if (hipMalloc((void**)&g_variant_stats, 16 * sizeof(uint32_t)) != hipSuccess) { g_variant_stats = nullptr; return nullptr; }
    (void)hipMemset(g_variant_stats, 0, 16 * sizeof(uint32_t));
  }
  return g_variant_stats;
}
extern "C" int tdr_profile_variants(int64_t out[16]) {   // reads and resets (synchronises)
  if (!out) return fail(TDR_ERR_ARG, "profile_variants: null pointer");
  for (int k = 0; k < 16; k++) out[k] = 0;
  if (!g_variant_stats) return TDR_OK;
  uint32_t h[16];
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h, g_variant_stats, sizeof(h), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(g_variant_stats, 0, sizeof(h)));
  for (int k = 0; k < 16; k++) out[k] = h[k];
  return TDR_OK;
}
extern "C" int tdr_profile_enable(int on) {
  g_prof_on = on != 0;
  g_prof_variants = on == 2;
  g_prof_used = 0;
  g_share_valid = false;
  return TDR_OK;
}
extern "C" int tdr_profile_shares(double* dense_ms, double* scattered_ms, int64_t* scattered_particles) {
  if (!dense_ms || !scattered_ms || !scattered_particles) return fail(TDR_ERR_ARG, "profile_shares: null pointer");
  if (!g_share_valid || !g_share_counts) return fail(TDR_ERR_ARG, "profile_shares: no integer-form launch was profiled");
  float ms[2] = {0.f, 0.f};
  for (int k = 0; k < 2; k++) {
    HIP_TRY(hipEventSynchronize(g_share_ev[k][1]));
    HIP_TRY(hipEventElapsedTime(&ms[k], g_share_ev[k][0], g_share_ev[k][1]));
  }
  int32_t counts[3] = {0, 0, 0};
  HIP_TRY(hipMemcpy(counts, g_share_counts, sizeof(counts), hipMemcpyDeviceToHost));
  *scattered_ms = ms[0];
  *dense_ms = ms[1];
  *scattered_particles = counts[1];
  return TDR_OK;
}
extern "C" int tdr_profile_score_ms(double* total_ms, int64_t* launches) {
  if (!total_ms || !launches) return fail(TDR_ERR_ARG, "profile_score_ms: null pointer");
  double tot = 0;
  for (size_t i = 0; i < g_prof_used; i++) {
    HIP_TRY(hipEventSynchronize(g_prof_events[i].second));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, g_prof_events[i].first, g_prof_events[i].second));
    tot += ms;
  }
  *total_ms = tot;
  *launches = (int64_t)g_prof_used;
  g_prof_used = 0;
  return TDR_OK;
}

// The scoring loops address map records with 32-bit byte offsets (v_mad_i32_i24 + 32-bit adds): the guarded record
// grid must stay below 4 GiB (11 583^2 cells of 32 bytes) and a guarded row below 2^24 bytes.
static int check_map_addressing(const tdr_map_desc* map, int rf, const char* who) {
  const uint64_t row_bytes = (uint64_t)(map->cols + 2) * rf * 4;
  const uint64_t total = row_bytes * (uint64_t)(map->rows + 2);
  if (row_bytes >= (1u << 24) || total > 0xFFFFFFFFull)
    return fail(TDR_ERR_ARG, "%s: map of %d x %d cells (%d-float records) exceeds the 4 GiB the kernels address", who,
                map->rows, map->cols, rf);
  return TDR_OK;
}

// The compact records are used whenever the map has them; tdr_config_compact(0) forces the dense ones (A/B, tests).
static int g_use_compact = 1;
extern "C" int tdr_config_compact(int on) {   // < 0: query only
  if (on >= 0) g_use_compact = on ? 1 : 0;
  return g_use_compact;
}
extern "C" int tdr_cmap_words(int ncls);
extern "C" size_t tdr_cmap_tile_words(int ncls, int rows, int cols);

template <int NV4, bool KS, bool US, bool CM, bool SK>
static void launch_polar_kernel(dim3 grid, dim3 block, size_t lds, hipStream_t s, const ScoreArgs& a) {
  auto kfn = score_polar_kernel<NV4, TDR_SCORE_U, KS, US, CM, false, SK>;
  if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, grid, block, lds, s, a);
}
template <bool CM>
static int launch_score_form(const ScoreArgs& a, int rf, int ncls, hipStream_t s) {
  dim3 grid((unsigned)cdiv(a.n, 256), (unsigned)a.nchunks), block(256);
  size_t lds = (size_t)a.nb * ((a.group * (rf / 4)) | 1) * 16;   // [row][group * planes | 1] float4 (+ 4 KB dictionary)
  const bool ks = tdr_has_kslot(ncls, rf);
  const bool us = a.utab != nullptr;
#define TDR_LAUNCH_SCORE_F(NV4, SK)                                                                 \
  if (ks && us) launch_polar_kernel<NV4, true, true, CM, SK>(grid, block, lds, s, a);                \
  else if (ks) launch_polar_kernel<NV4, true, false, CM, SK>(grid, block, lds, s, a);                \
  else if (us) launch_polar_kernel<NV4, false, true, CM, SK>(grid, block, lds, s, a);                \
  else launch_polar_kernel<NV4, false, false, CM, SK>(grid, block, lds, s, a);
#define TDR_LAUNCH_SCORE(NV4)                                  \
  if constexpr (CM) {                                          \
    if (a.kmask_row) { TDR_LAUNCH_SCORE_F(NV4, true) }         \
    else { TDR_LAUNCH_SCORE_F(NV4, false) }                    \
  } else { TDR_LAUNCH_SCORE_F(NV4, false) }
  switch (rf / 4) {
    case 1: TDR_LAUNCH_SCORE(1) break;
    case 2: TDR_LAUNCH_SCORE(2) break;
    case 3: TDR_LAUNCH_SCORE(3) break;
    case 4:
      if constexpr (!CM) { TDR_LAUNCH_SCORE(4) }   // 12-15 classes have no compact form
      break;
    default: return fail(TDR_ERR_ARG, "score: unsupported record size %d", rf);
  }
#undef TDR_LAUNCH_SCORE
#undef TDR_LAUNCH_SCORE_F
  LAUNCH_CHECK("score_polar");
  return TDR_OK;
}
static bool map_is_wide(const tdr_map_desc* map, int rf) {   // tdr_cmap.hip: 16-bit fields
  return rf == 8 && map->cwords == 4 && map->dict_n > TDR_CMAP_MAX_DICT;
}
static bool map_has_compact(const tdr_map_desc* map, int rf) {
  if (!(g_use_compact && map->cwords > 0 && map->crec && map->dict && map->dict_n > 0 && rf <= 12 &&
        (map->cwords == tdr_cmap_words(map->ncls) || map_is_wide(map, rf))))
    return false;
  if (map->dict_n > (map_is_wide(map, rf) ? TDR_CMAP_WIDE_MAX_DICT : TDR_CMAP_MAX_DICT)) return false;
  const int lc = map->cwords == 1 ? 3 : (map->cwords == 2 ? 2 : 1);
  return (int64_t)((map->rows >> lc) + 2) * 128 < (1 << 23) && map->cols < (1 << 24);   // cmap_offset: 24-bit operands
}
static int launch_score(ScoreArgs a, const tdr_map_desc* map, int rf, int ncls, hipStream_t s, bool profile = true) {
  a.crec = nullptr; a.dict = nullptr; a.dict_n = 0; a.ctiles_r = 0;
  ScoreProfScope prof(profile ? s : nullptr, profile);
  if (!map_has_compact(map, rf)) return launch_score_form<false>(a, rf, ncls, s);
  const int lc = map->cwords == 1 ? 3 : (map->cwords == 2 ? 2 : 1);
  a.crec = map->crec;
  a.dict = map->dict;
  a.dict_n = map->dict_n;
  a.ctiles_r = (map->rows >> lc) + 2;
  if (map_is_wide(map, rf)) {   // more than 1024 distinct values: 16-bit fields, a 16 KB dictionary in LDS
    dim3 grid((unsigned)cdiv(a.n, 256), (unsigned)a.nchunks), block(256);
    const size_t lds = (size_t)a.nb * ((a.group * 2) | 1) * 16;
    const bool ks = tdr_has_kslot(ncls, rf), us = a.utab != nullptr;
    if (ks && us) hipLaunchKernelGGL((score_polar_kernel<2, TDR_SCORE_U, true, true, true, true>), grid, block, lds, s, a);
    else if (ks) hipLaunchKernelGGL((score_polar_kernel<2, TDR_SCORE_U, true, false, true, true>), grid, block, lds, s, a);
    else if (us) hipLaunchKernelGGL((score_polar_kernel<2, TDR_SCORE_U, false, true, true, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((score_polar_kernel<2, TDR_SCORE_U, false, false, true, true>), grid, block, lds, s, a);
    LAUNCH_CHECK("score_polar(wide)");
    return TDR_OK;
  }
  return launch_score_form<true>(a, rf, ncls, s);
}

// tdr_score_ctx (tdr.h): what a scoring call keeps BETWEEN calls — the span tuner of tdr_score_su.h.  It belongs to one
// caller (a filter handle): nothing of it is shared between filters, threads or devices.  Without a context a call uses
// the configured span.  (Rounds 3 and 4 also gave a call a stream of its own to run its two kernels side by side: measured
// on every configuration, the two never overlapped usefully — config 2: 5.22 ms on one stream, 5.32 on two; config 5: 16.4
// against 17.0 — because the dense kernel waits for its gathers right after issuing them and becomes latency-bound as soon
// as another kernel fills the L1's queues.  The kernels of a call now run one after the other on the caller's stream.)
struct tdr_score_ctx {
  int device = 0;
  SpanTuner tuner;
  const float* fac = nullptr;   // the table's factors on the device (tdr_polar_factors_host), the caller's memory
  int fac_nb = 0, fac_nr = 0;
};
extern "C" int tdr_score_ctx_set_polar_factors(tdr_score_ctx* c, const float* fac_dev, int nb, int nr) {
  if (!c || (fac_dev && (nb < 1 || nr < 1))) return fail(TDR_ERR_ARG, "score_ctx_set_polar_factors: bad arguments");
  c->fac = fac_dev;
  c->fac_nb = fac_dev ? nb : 0;
  c->fac_nr = fac_dev ? nr : 0;
  return TDR_OK;
}
extern "C" int tdr_score_ctx_create(tdr_score_ctx** out) {
  if (!out) return fail(TDR_ERR_ARG, "score_ctx_create: null pointer");
  *out = nullptr;
  tdr_score_ctx* c = new (std::nothrow) tdr_score_ctx;
  if (!c) return fail(TDR_ERR_NOMEM, "score_ctx_create: out of memory");
  const hipError_t e = hipGetDevice(&c->device);
  if (e != hipSuccess) {
    delete c;
    return fail(TDR_ERR_HIP, "score_ctx_create: %s", hipGetErrorString(e));
  }
  *out = c;
  return TDR_OK;
}
extern "C" void tdr_score_ctx_destroy(tdr_score_ctx* c) {
  if (!c) return;
  if (c->tuner.e0) (void)hipEventDestroy(c->tuner.e0);
  if (c->tuner.e1) (void)hipEventDestroy(c->tuner.e1);
  delete c;
}
extern "C" float tdr_score_ctx_span(const tdr_score_ctx* c) {   // the span the context's tuner has settled on so far
  return c ? c->tuner.best : tdr_config_shift_uniform_span(-1.f);
}
extern "C" int64_t tdr_score_ctx_trial_calls(const tdr_score_ctx* c) {   // launches spent on trial spans so far
  return c ? c->tuner.trial_calls : 0;
}
// closes the tuner's measurement on EVERY way out of a call
struct TunerScope {
  tdr_score_ctx* c;
  hipStream_t s;
  TunerScope(tdr_score_ctx* c_, hipStream_t s_) : c(c_), s(s_) {}
  ~TunerScope() {
    if (c) tdr_su_span_end(&c->tuner, s);
  }
};

extern "C" int tdr_k_score_polar(const tdr_map_desc* map, const float* tab, const float* scan_pk, int nb, int nr,
                                 float res, const tdr_filter_params* fp, float* st, int64_t cap, int64_t n,
                                 int64_t n_total, const int32_t* perm, float uniform_scale, int init_search,
                                 float* raw_w, float* workspace, void* stream) {
  return tdr_k_score_polar_ctx(map, tab, scan_pk, nb, nr, res, fp, st, cap, n, n_total, perm, uniform_scale, init_search,
                               raw_w, workspace, nullptr, stream);
}
extern "C" int tdr_k_score_polar_ctx(const tdr_map_desc* map, const float* tab, const float* scan_pk, int nb, int nr,
                                     float res, const tdr_filter_params* fp, float* st, int64_t cap, int64_t n,
                                     int64_t n_total, const int32_t* perm, float uniform_scale, int init_search,
                                     float* raw_w, float* workspace, tdr_score_ctx* ctx, void* stream) {
  if (!map || !map->rec || !tab || !scan_pk || !fp || !st || !raw_w || !workspace)
    return fail(TDR_ERR_ARG, "score: null pointer");
  if (n_total <= 0) n_total = n;
  if (n < 0 || cap < n) return fail(TDR_ERR_ARG, "score: n=%lld exceeds capacity %lld", (long long)n, (long long)cap);
  if (n == 0) return TDR_OK;
  if (nb < 1 || nr < 1) return fail(TDR_ERR_ARG, "score: bad image shape");
  if (map->ncls < 1 || map->ncls > TDR_MAX_CLASSES || fp->num_classes != map->ncls)
    return fail(TDR_ERR_ARG, "score: class count mismatch (map %d, params %d)", map->ncls, fp->num_classes);
  const int rf = tdr_rec_floats(map->ncls);
  if (map->rec_floats != rf) return fail(TDR_ERR_ARG, "score: map record size %d != %d", map->rec_floats, rf);
  if ((size_t)2 * nb * rf * 4 > 64 * 1024) return fail(TDR_ERR_ARG, "score: nb too large for the LDS scan ring");
  if (!(map->resolution > 0.f)) return fail(TDR_ERR_ARG, "score: map resolution must be > 0");
  if (int rc0 = check_map_addressing(map, rf, "score")) return rc0;
  hipStream_t s = (hipStream_t)stream;

  ScoreArgs a;
  a.rec = map->rec; a.rows = map->rows; a.cols = map->cols; a.resolution = map->resolution;
  a.tab = tab; a.scan_pk = scan_pk; a.nb = nb; a.nr = nr; a.res = res;
  a.st = st; a.cap = cap; a.n = n; a.order = perm; a.count = nullptr; a.slot_base = nullptr; a.kmask_off = 0; a.kmask_row = 0;
  a.use_theta_override = 0; a.theta_override = 0.f; a.only_uninit = 0;
  const ScoreWs W = score_ws(map->ncls, nb, nr, n, n_total);
  a.group = W.group;
  a.nchunks = W.nchunks;
  a.npad = W.npad;
  a.part = workspace;
  int rc = fill_utab(a, workspace, W, uniform_scale, s);
  if (rc) return rc;
  float* res_flag = workspace + W.off_aux;                             // npad floats
  float* res_theta = res_flag + a.npad;                                // npad floats
  if (init_search) {
    // state_particle.cpp:195-206 first: it fixes theta / have_init of the un-initialised particles, the regular pass
    // below then scores every particle at its (possibly just chosen) rotation
    InitArgs ia;
    ia.rec = a.rec; ia.rows = a.rows; ia.cols = a.cols; ia.resolution = a.resolution;
    ia.tab = a.tab; ia.utab = a.utab; ia.scan_pk = a.scan_pk; ia.nb = nb; ia.nr = nr; ia.res = res;
    ia.st = st; ia.cap = cap; ia.n = n; ia.order = perm; ia.fp = *fp; ia.gate = make_gate(fp, map);
    ia.P = (int64_t)nb * nr; ia.ncls = map->ncls; ia.res_flag = res_flag; ia.res_theta = res_theta;
    // rotation table lives behind the result arrays: [shift INIT_MAXROT][theta INIT_MAXROT][nrot]
    int* d_shift = reinterpret_cast<int*>(res_theta + a.npad);
    float* d_theta = reinterpret_cast<float*>(d_shift + INIT_MAXROT);
    int* d_nrot = reinterpret_cast<int*>(d_theta + INIT_MAXROT);
    hipLaunchKernelGGL(init_rot_kernel, dim3(1), dim3(64), 0, s, nb, d_shift, d_theta, d_nrot);
    LAUNCH_CHECK("init_rot");
    ia.shift = d_shift; ia.theta = d_theta; ia.nrot = d_nrot;
    HIP_TRY(hipMemsetAsync(res_flag, 0, sizeof(float) * (size_t)n, s));
    dim3 grid((unsigned)cdiv(n, 64)), block(64 * INIT_WAVES);
    const size_t lds = TDR_INIT_SCAN_LDS ? (size_t)2 * nb * rf * 4 : 0;
    const bool ks = tdr_has_kslot(map->ncls, rf), us = a.utab != nullptr;
    ia.only_if = nullptr;
    bool unitw = true;
    for (int c = 1; c < map->ncls; c++) unitw &= fp->class_weights[c] == fp->class_weights[0];
    unitw &= fp->class_weights[0] > 0.f;
    int* d_inexact = d_nrot + 1;
    const bool half_path = (rf == 4 || rf == 8) && map->ncls <= 7 && init_use_mfma() && map->rec16 &&
                           tdr_map_rec16_bytes(map->ncls, map->rows, map->cols) != 0 && n_total >= g_rec16_min &&
                           (size_t)4 * (2 * nb + 20) * 16 + (size_t)4 * (nb + 8) * 8 <= 64 * 1024;
    if (half_path) {
      // matrix-core pass on pre-split half records (weights folded in), built into the map owner's scratch first; the
      // vector kernel below then runs only if a scan count did not fit f16
      const int64_t ncells = (int64_t)(map->rows + 2) * (map->cols + 2);
      const dim3 hgrid((unsigned)cdiv(ncells, 256)), hblock(256);
      const float4* rec4 = reinterpret_cast<const float4*>(map->rec);
      uint4* r16 = reinterpret_cast<uint4*>(map->rec16);
      if (rf == 4) hipLaunchKernelGGL((half_records_kernel<4>), hgrid, hblock, 0, s, rec4, ncells, unitw ? 1 : 0, *fp, map->ncls, r16);
      else hipLaunchKernelGGL((half_records_kernel<8>), hgrid, hblock, 0, s, rec4, ncells, unitw ? 1 : 0, *fp, map->ncls, r16);
      LAUNCH_CHECK("half_records");
      const int ahead = g_init_ahead;   // tuning: record loads kept in flight per wave (1..3)
      const int R = ahead + 1;
      const int img = init_half_image_rows(nb, R);
      const size_t ldsh = (size_t)4 * img * 16 + (size_t)4 * (nb + 2 * R) * 8;
      const uint4* r16c = r16;
#define TDR_LAUNCH_HALF(AH)                                                                                              \
  if (us) hipLaunchKernelGGL((score_init_half_kernel<true, AH>), grid, dim3(256), ldsh, s, ia, r16c, d_inexact, img, rf);   \
  else hipLaunchKernelGGL((score_init_half_kernel<false, AH>), grid, dim3(256), ldsh, s, ia, r16c, d_inexact, img, rf);
      if (ahead == 1) { TDR_LAUNCH_HALF(1) } else if (ahead == 2) { TDR_LAUNCH_HALF(2) } else { TDR_LAUNCH_HALF(3) }
#undef TDR_LAUNCH_HALF
      LAUNCH_CHECK("score_init_half");
      ia.only_if = d_inexact;
    } else if (rf == 8 && (ks || map->ncls == 7) && init_use_mfma()) {
      // matrix-core pass splitting the f32 records per sample (small filters, maps without the scratch); the vector
      // kernel below then runs only if a scan count did not fit f16
      const size_t lds16 = ((size_t)2 * nb + 1) * 16;
      const int variant = (us ? 4 : 0) | (unitw ? 2 : 0) | (ks ? 0 : 1);
      switch (variant) {
#define TDR_LAUNCH_MFMA(US, UW, SV) \
  hipLaunchKernelGGL((score_init_mfma_kernel<US, UW, SV>), grid, dim3(256), lds16, s, ia, d_inexact); break;
        case 0: TDR_LAUNCH_MFMA(false, false, false)
        case 1: TDR_LAUNCH_MFMA(false, false, true)
        case 2: TDR_LAUNCH_MFMA(false, true, false)
        case 3: TDR_LAUNCH_MFMA(false, true, true)
        case 4: TDR_LAUNCH_MFMA(true, false, false)
        case 5: TDR_LAUNCH_MFMA(true, false, true)
        case 6: TDR_LAUNCH_MFMA(true, true, false)
        default: TDR_LAUNCH_MFMA(true, true, true)
#undef TDR_LAUNCH_MFMA
      }
      LAUNCH_CHECK("score_init_mfma");
      ia.only_if = d_inexact;
    } else if ((rf == 12 || rf == 16) && init_use_mfma()) {
      // 8-15 classes: two groups of 8 slots per sample (score_init_mfma_wide_kernel)
      const size_t lds16 = ((size_t)2 * nb + 1) * 2 * 16;
      if (rf == 12) {
        if (us) hipLaunchKernelGGL((score_init_mfma_wide_kernel<3, true>), grid, dim3(256), lds16, s, ia, d_inexact);
        else hipLaunchKernelGGL((score_init_mfma_wide_kernel<3, false>), grid, dim3(256), lds16, s, ia, d_inexact);
      } else {
        if (us) hipLaunchKernelGGL((score_init_mfma_wide_kernel<4, true>), grid, dim3(256), lds16, s, ia, d_inexact);
        else hipLaunchKernelGGL((score_init_mfma_wide_kernel<4, false>), grid, dim3(256), lds16, s, ia, d_inexact);
      }
      LAUNCH_CHECK("score_init_mfma_wide");
      ia.only_if = d_inexact;
    }
#define TDR_LAUNCH_INIT(NV4)                                                                                \
  if (ks && us) hipLaunchKernelGGL((score_init_kernel<NV4, true, true>), grid, block, lds, s, ia);         \
  else if (ks) hipLaunchKernelGGL((score_init_kernel<NV4, true, false>), grid, block, lds, s, ia);         \
  else if (us) hipLaunchKernelGGL((score_init_kernel<NV4, false, true>), grid, block, lds, s, ia);         \
  else hipLaunchKernelGGL((score_init_kernel<NV4, false, false>), grid, block, lds, s, ia);
    switch (rf / 4) {
      case 1: TDR_LAUNCH_INIT(1) break;
      case 2: TDR_LAUNCH_INIT(2) break;
      case 3: TDR_LAUNCH_INIT(3) break;   // 8-11 classes
      case 4: TDR_LAUNCH_INIT(4) break;   // 12-15 classes (TDR_MAX_CLASSES)
      default: return fail(TDR_ERR_ARG, "score: unsupported record size %d", rf);
    }
#undef TDR_LAUNCH_INIT
    LAUNCH_CHECK("score_init");
    hipLaunchKernelGGL(init_apply_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, (const float*)res_theta,
                       (const float*)res_flag, n, st, cap);
    LAUNCH_CHECK("init_apply");
  }
  FinalizeArgs f;
  f.part = a.part; f.rf = rf; f.nchunks = a.nchunks; f.npad = a.npad; f.n = n; f.cap = cap;
  f.order = perm; f.count = nullptr; f.st = st; f.fp = *fp;
  f.gate = make_gate(fp, map);
  f.P = (int64_t)nb * nr; f.ncls = map->ncls; f.mode = 0; f.first = 0; f.theta_override = 0.f;
  f.raw_w = raw_w; f.best_cost = nullptr; f.best_theta = nullptr;
  f.gpart = nullptr; f.gnchunks = 0; f.gsum0 = f.gsum1 = 0.f; f.only_uninit = 0;
  if (W.su && map_has_compact(map, rf) && !map_is_wide(map, rf) && tdr_ray_map_ok(map)) {
    // The INTEGER form of the launch (tdr_score_su.h): dense particles by heading bin through the shift-uniform kernel,
    // scattered ones — behind the bins in the same slot list — one wave each through the ray-mapped kernel; both form exact
    // integer sums, so a particle's weight does not depend on which of the two scored it.  A scan or a map without
    // an integer form (fractional or non-finite counts; a dictionary finer than 2^-q) raises the device word `inexact`:
    // the integer kernels then return at once and the float kernel below does the launch — nothing is decided on the host.
    const int32_t* slots = nullptr;
    const int32_t* counts = nullptr;   // device words {slots of the dense share, scattered particles behind them, both}
    SuLaunch L;
    L.map = map; L.tab = a.utab ? a.utab : a.tab; L.uniform_scale = a.utab != nullptr; L.scan_pk = scan_pk;
    L.nb = nb; L.nr = nr; L.rf = rf; L.res = res; L.st = st; L.cap = cap; L.n = n; L.perm = perm;
    L.group = W.su_group; L.nchunks = W.su_nchunks; L.npad = W.npad_part; L.part = a.part;
    L.fac = ctx && ctx->fac && ctx->fac_nb == nb && ctx->fac_nr == nr ? ctx->fac : nullptr;
    L.uscale = uniform_scale;
    L.wave_span = tdr_su_wave_span();
    L.ray_split = tdr_ray_splits(nb, nr, n, tdr_ray_block_major(L));
    L.ws = reinterpret_cast<int32_t*>(workspace + W.off_su);
    TunerScope tuner_scope(ctx, s);   // (closes the tuner's measurement on every way out)
    L.span = tdr_su_span_begin(ctx ? &ctx->tuner : nullptr,
                               ((int64_t)n << 24) ^ ((int64_t)nb << 12) ^ nr ^ ((int64_t)map->rows << 40), s);
    if ((rc = tdr_su_prepare(L, W.suw, s, &slots, &counts))) return rc;
    if ((rc = tdr_ray_prepare(L, W.suw, s))) return rc;
    const int32_t* inexact = counts + 4;
    {
      ScoreProfScope prof(s);
      {
        ShareProfScope sp(0, s, true);
        if ((rc = tdr_ray_score(L, W.suw, s))) return rc;
      }
      {
        ShareProfScope sp(1, s, true);
        if ((rc = tdr_su_score(L, W.suw, s))) return rc;
      }
      if (g_prof_on) { g_share_valid = true; g_share_counts = counts; }
      // the float form, for the launches the integer form does not cover
      ScoreArgs r = a;
      r.run_if = inexact;
      if ((rc = launch_score(r, map, rf, map->ncls, s, false))) return rc;
    }
    FinalizeArgs fx = f;
    fx.npad = W.npad_part; fx.order = slots; fx.counts = counts; fx.inexact = inexact;
    fx.ipart = reinterpret_cast<const uint32_t*>(a.part);
    fx.dict_tail = reinterpret_cast<const uint32_t*>(map->dict) + 2 * TDR_CMAP_MAX_DICT;
    fx.nchunks = W.su_nchunks; fx.ray_split = L.ray_split;
    hipLaunchKernelGGL(score_finalize_exact_kernel, dim3((unsigned)cdiv(W.npad_part, 64)), dim3(256), 0, s, fx);
    LAUNCH_CHECK("score_finalize_exact");
    f.run_if = inexact;
    launch_finalize(f, n, s);
  } else {
    rc = launch_score(a, map, rf, map->ncls, s);
    if (rc) return rc;
    launch_finalize(f, n, s);
  }
  LAUNCH_CHECK("score_finalize");
  if (init_search) {
    hipLaunchKernelGGL(init_fixup_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, (const float*)res_flag, n,
                       fp->regularization, raw_w);
    LAUNCH_CHECK("init_fixup");
  }
  return TDR_OK;
}

// ---- scoring WITH the geometric term -------------------------------------------------------------------
// getCostForRot's geometric block — cost += (top_down_geo[i] . shifted geo_cls[i]).sum() * 0.01 for the two layers,
// normalization += top_down_geo[i].sum() — is commented out in the reference (src/state_particle.cpp:145-152) and its
// inputs are zero images there (src/top_down_render.cpp:533-540).  It is available here as an opt-in: a second scoring
// launch over the 2-layer geometric map (tdr_k_geo_map_from_map) against the packed geometric scan, combined in the
// finalize step.  The 40-rotation search then has to price the geometric term for every candidate too: it runs as one
// scoring pass per rotation over the workgroups that hold a particle without a heading (the window-gathered-once
// kernels only know the semantic term).
__global__ void init_from_best_kernel(const float* __restrict__ best_cost, const float* __restrict__ best_theta,
                                      const int32_t* __restrict__ order, int64_t n, float* __restrict__ st, int64_t cap,
                                      GateArgs gate, float* __restrict__ res_flag) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n) return;
  const int64_t p = order ? (int64_t)order[slot] : slot;
  res_flag[p] = 0.f;
  if (st[TDR_ST_HAVE_INIT * cap + p] != 0.f) return;
  const float scale = st[TDR_ST_SCALE * cap + p];
  const float cx = st[TDR_ST_DX * cap + p] * scale + st[TDR_ST_INIT_X * cap + p];
  const float cy = st[TDR_ST_DY * cap + p] * scale + st[TDR_ST_INIT_Y * cap + p];
  if (particle_gated(gate, cx, cy, scale)) return;          // computeWeight returns before the search (:163-176)
  const bool none = !(best_cost[slot] < 3.402823466e+38f);  // every candidate scored NaN: best_theta stays 0 (:193-194)
  st[TDR_ST_THETA * cap + p] = none ? 0.f : best_theta[slot];   // :205
  st[TDR_ST_HAVE_INIT * cap + p] = 1.f;                         // :206
  res_flag[p] = none ? 2.f : 1.f;
}
extern "C" size_t tdr_score_geo_workspace_floats(int ncls, int nb, int nr, int64_t n, int64_t n_total) {
  const int64_t npad = cdiv(std::max<int64_t>(n, 1), 64) * 64;
  const int ggroup = score_group_rings(nb, nr, 4, n_total > 0 ? n_total : n);
  return tdr_score_workspace_floats(ncls, nb, nr, n, n_total) + (size_t)(cdiv(nr, ggroup) * 5 * npad);
}
extern "C" int tdr_k_score_polar_geo(const tdr_map_desc* map, const tdr_map_desc* geo_map, const float* tab,
                                     const float* scan_pk, const float* geo_pk, float geo_sum0, float geo_sum1, int nb,
                                     int nr, float res, const tdr_filter_params* fp, float* st, int64_t cap, int64_t n,
                                     int64_t n_total, const int32_t* perm, float uniform_scale, int init_search,
                                     float* raw_w, float* workspace, void* stream) {
  if (!map || !map->rec || !geo_map || !geo_map->rec || !tab || !scan_pk || !geo_pk || !fp || !st || !raw_w || !workspace)
    return fail(TDR_ERR_ARG, "score_geo: null pointer");
  if (n < 0 || cap < n) return fail(TDR_ERR_ARG, "score_geo: n exceeds capacity");
  if (n == 0) return TDR_OK;
  if (n_total <= 0) n_total = n;
  if (nb < 1 || nr < 1) return fail(TDR_ERR_ARG, "score_geo: bad image shape");
  if (map->ncls < 1 || map->ncls > TDR_MAX_CLASSES || fp->num_classes != map->ncls)
    return fail(TDR_ERR_ARG, "score_geo: class count mismatch");
  if (geo_map->ncls != 2 || geo_map->rec_floats != 4 || geo_map->rows != map->rows || geo_map->cols != map->cols)
    return fail(TDR_ERR_ARG, "score_geo: the geometric map must be the 2-layer map of the same grid");
  const int rf = tdr_rec_floats(map->ncls);
  if (map->rec_floats != rf) return fail(TDR_ERR_ARG, "score_geo: map record size mismatch");
  if ((size_t)nb * rf * 4 > 60 * 1024) return fail(TDR_ERR_ARG, "score_geo: nb too large for the LDS scan ring");
  if (int rc0 = check_map_addressing(map, rf, "score_geo")) return rc0;
  hipStream_t s = (hipStream_t)stream;
  ScoreArgs a;
  a.rec = map->rec; a.rows = map->rows; a.cols = map->cols; a.resolution = map->resolution;
  a.tab = tab; a.scan_pk = scan_pk; a.nb = nb; a.nr = nr; a.res = res;
  a.st = st; a.cap = cap; a.n = n; a.order = perm; a.count = nullptr; a.slot_base = nullptr; a.kmask_off = 0; a.kmask_row = 0;
  a.use_theta_override = 0; a.theta_override = 0.f; a.only_uninit = 0;
  const ScoreWs W = score_ws(map->ncls, nb, nr, n, n_total);
  a.group = W.group;
  a.nchunks = W.nchunks;
  a.npad = W.npad;
  a.part = workspace;
  int rc = fill_utab(a, workspace, W, uniform_scale, s);
  if (rc) return rc;
  float* best_cost = workspace + W.off_aux;                               // npad floats (res_flag's place)
  float* best_theta = best_cost + a.npad;                                 // npad floats
  float* res_flag = best_theta + a.npad;                                  // npad floats ("list" region)
  ScoreArgs g = a;   // the geometric launch: same particles, same table, the 2-layer map and scan
  g.rec = geo_map->rec; g.scan_pk = geo_pk;
  g.group = score_group_rings(nb, nr, 4, n_total);
  g.nchunks = (int)cdiv(nr, g.group);
  g.part = workspace + tdr_score_workspace_floats(map->ncls, nb, nr, n, n_total);
  FinalizeArgs f;
  f.part = a.part; f.rf = rf; f.nchunks = a.nchunks; f.npad = a.npad; f.n = n; f.cap = cap;
  f.order = perm; f.count = nullptr; f.st = st; f.fp = *fp;
  f.gate = make_gate(fp, map);
  f.P = (int64_t)nb * nr; f.ncls = map->ncls;
  f.raw_w = raw_w; f.best_cost = best_cost; f.best_theta = best_theta;
  f.gpart = g.part; f.gnchunks = g.nchunks; f.gsum0 = geo_sum0; f.gsum1 = geo_sum1;
  const dim3 fgrid((unsigned)cdiv(n, 256)), fblock(256);
  if (init_search) {
    // state_particle.cpp:195-206 with the geometric term: one pass per candidate rotation, workgroups without an
    // un-initialised particle return at once
    a.only_uninit = g.only_uninit = 1;
    a.use_theta_override = g.use_theta_override = 1;
    f.mode = 1; f.only_uninit = 1;
    int k = 0;
    for (float t = 0; t < 2 * M_PI; t += 2 * M_PI / 40) {   // :197 (float t, double increment)
      a.theta_override = g.theta_override = t;
      if ((rc = launch_score(a, map, rf, map->ncls, s))) return rc;
      if ((rc = launch_score(g, geo_map, 4, 2, s))) return rc;
      f.first = k == 0; f.theta_override = t;
      launch_finalize(f, n, s);
      LAUNCH_CHECK("score_finalize(geo init)");
      k++;
    }
    hipLaunchKernelGGL(init_from_best_kernel, fgrid, fblock, 0, s, (const float*)best_cost, (const float*)best_theta, perm,
                       n, st, cap, f.gate, res_flag);
    LAUNCH_CHECK("init_from_best");
    a.only_uninit = g.only_uninit = 0;
    a.use_theta_override = g.use_theta_override = 0;
  }
  if ((rc = launch_score(a, map, rf, map->ncls, s))) return rc;
  if ((rc = launch_score(g, geo_map, 4, 2, s))) return rc;
  f.mode = 0; f.first = 0; f.theta_override = 0.f; f.only_uninit = 0;
  launch_finalize(f, n, s);
  LAUNCH_CHECK("score_finalize(geo)");
  if (init_search) {
    hipLaunchKernelGGL(init_fixup_kernel, fgrid, fblock, 0, s, (const float*)res_flag, n, fp->regularization, raw_w);
    LAUNCH_CHECK("init_fixup");
  }
  return TDR_OK;
}

static int64_t cart_part_floats(int ncls, int cols, int64_t n, int64_t n_total) {   // partial sums, 256-byte aligned
  int cpc, nchunks;
  choose_chunks(n_total > 0 ? n_total : n, cols, cpc, nchunks, TDR_CART_WAVE_MUL);
  // (the integer form's slot list pads the dense share to whole waves, its rows are words: 2 per class + 2, and a scattered
  // particle's window may be split over up to TDR_RAY_MAX_SPLIT chunk rows)
  const int64_t npad = su_npad(std::max<int64_t>(n, 1), 1);
  const int rowsmax = std::max(tdr_rec_floats(ncls) + 1, 2 * ncls + 2);
  return cdiv((int64_t)std::max(nchunks, TDR_RAY_MAX_SPLIT) * rowsmax * npad + 64, 64) * 64;
}
extern "C" size_t tdr_score_cart_workspace_floats(int ncls, int rows, int cols, int64_t n, int64_t n_total) {
  // partial sums + the float form's scan descriptors + the integer form's tables and ordering workspace
  return (size_t)(cart_part_floats(ncls, cols, n, n_total) + tdr_cart_desc_words(rows, cols) +
                  tdr_cart_int_words(rows, cols, std::max<int64_t>(n, 1)));
}

extern "C" int tdr_k_score_cart(const tdr_map_desc* map, const float* scan_pk, int rows, int cols, float res,
                                const tdr_filter_params* fp, float* st, int64_t cap, int64_t n, int64_t n_total,
                                const int32_t* perm, float* raw_w, float* workspace, void* stream) {
  if (!map || !map->rec || !scan_pk || !fp || !st || !raw_w || !workspace)
    return fail(TDR_ERR_ARG, "score_cart: null pointer");
  if (n_total <= 0) n_total = n;
  if (n < 0 || cap < n) return fail(TDR_ERR_ARG, "score_cart: n exceeds capacity");
  if (n == 0) return TDR_OK;
  if (rows < 1 || cols < 1) return fail(TDR_ERR_ARG, "score_cart: bad window shape");
  if (fp->num_classes != map->ncls) return fail(TDR_ERR_ARG, "score_cart: class count mismatch");
  const int rf = tdr_rec_floats(map->ncls);
  if (map->rec_floats != rf) return fail(TDR_ERR_ARG, "score_cart: map record size mismatch");
  if (int rc0 = check_map_addressing(map, rf, "score_cart")) return rc0;
  hipStream_t s = (hipStream_t)stream;
  CartArgs a;
  a.rec = map->rec; a.map_rows = map->rows; a.map_cols = map->cols; a.resolution = map->resolution;
  a.scan_pk = scan_pk; a.rows = rows; a.cols = cols; a.res = res;
  a.st = st; a.cap = cap; a.n = n; a.order = perm;
  // chunks of window columns from the filter's TOTAL particle count (the same on every rank of a sharded filter, like
  // score_group_rings): a particle's partial sums are then the same in an N-rank run as in the 1-rank run
  choose_chunks(n_total, cols, a.cpc, a.nchunks, TDR_CART_WAVE_MUL);
  a.npad = cdiv(n, 64) * 64;
  a.part = workspace;
  a.libm_fma = tdr_libm_fma();
  dim3 grid((unsigned)cdiv(n, 256), (unsigned)a.nchunks), block(256);
  const bool ks = tdr_has_kslot(map->ncls, rf);
  const bool cm = map_has_compact(map, rf), wide = cm && map_is_wide(map, rf);
  a.crec = nullptr; a.dict = nullptr; a.dict_n = 0; a.ctiles_r = 0;
  if (cm) {
    const int lc = map->cwords == 1 ? 3 : (map->cwords == 2 ? 2 : 1);
    a.crec = map->crec; a.dict = map->dict; a.dict_n = map->dict_n; a.ctiles_r = (map->rows >> lc) + 2;
  }
  CartIntOut io{};
  bool int_form = false;
  if (cm && !wide && tdr_cart_skip_ok(map, rf)) {
    ScoreProfScope prof(s);
    uint32_t* desc_ws = reinterpret_cast<uint32_t*>(workspace + cart_part_floats(map->ncls, cols, n, n_total));
    if (tdr_cart_int_ok(map, rf, rows, cols, n_total)) {
      // the integer form: dense particles through the skipping kernel with integer accumulators, scattered ones one wave
      // each through score_cart_ray_kernel; the float skipping kernel behind them for what has no integer form
      int32_t* iws = reinterpret_cast<int32_t*>(desc_ws + tdr_cart_desc_words(rows, cols));
      // (which particles count as dense: four times the polar launch's span — a Cartesian window is a rotated rectangle of
      // rows x cols cells and neighbours a few dozen cells apart still share most of their lines; measured on config 4, ms per
      // step at 8 / 16 / 32 / 64 cells: 32.6 / 29.2 / 28.8 / 28.5, the float kernel 36.9)
      if (int rc = tdr_cart_int_launch(a, map, rf, desc_ws, iws, 4.f * tdr_config_shift_uniform_span(-1.f), s, &io)) return rc;
      int_form = true;
    } else if (int rc = tdr_cart_skip_launch(a, map, rf, desc_ws, s)) return rc;
  } else {
    ScoreProfScope prof(s);
#define TDR_LAUNCH_CART2(NV4, CM)                                                                       \
  if (ks) hipLaunchKernelGGL((score_cart_kernel<NV4, TDR_SCORE_U, true, CM>), grid, block, 0, s, a);    \
  else hipLaunchKernelGGL((score_cart_kernel<NV4, TDR_SCORE_U, false, CM>), grid, block, 0, s, a);
#define TDR_LAUNCH_CART(NV4)               \
  if (cm) { TDR_LAUNCH_CART2(NV4, true) }  \
  else { TDR_LAUNCH_CART2(NV4, false) }
    if (wide) {   // more than 1024 distinct values: 16-bit fields (tdr_cmap.hip)
      if (ks) hipLaunchKernelGGL((score_cart_kernel<2, TDR_SCORE_U, true, true, true>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((score_cart_kernel<2, TDR_SCORE_U, false, true, true>), grid, block, 0, s, a);
    } else
    switch (rf / 4) {
      case 1: TDR_LAUNCH_CART(1) break;
      case 2: TDR_LAUNCH_CART(2) break;
      case 3: TDR_LAUNCH_CART(3) break;
      case 4: TDR_LAUNCH_CART2(4, false) break;
      default: return fail(TDR_ERR_ARG, "score_cart: unsupported record size %d", rf);
    }
#undef TDR_LAUNCH_CART
#undef TDR_LAUNCH_CART2
  }
  LAUNCH_CHECK("score_cart");
  FinalizeArgs f;
  f.part = a.part; f.rf = rf; f.nchunks = a.nchunks; f.npad = a.npad; f.n = n; f.cap = cap;
  f.order = perm; f.count = nullptr; f.st = st; f.fp = *fp;
  f.gate = make_gate(fp, map);
  f.gate.force_on_map = 0;   // the Cartesian definition has no gates (include/tdr.h)
  f.gate.scale_unknown = 0;
  f.P = (int64_t)rows * cols; f.ncls = map->ncls; f.mode = 0; f.first = 0; f.theta_override = 0.f;
  f.raw_w = raw_w; f.best_cost = nullptr; f.best_theta = nullptr;
  f.gpart = nullptr; f.gnchunks = 0; f.gsum0 = f.gsum1 = 0.f; f.only_uninit = 0;
  if (int_form) {
    FinalizeArgs fx = f;
    fx.npad = io.npad; fx.order = io.slots; fx.counts = io.counts; fx.inexact = io.flags;
    fx.ipart = reinterpret_cast<const uint32_t*>(a.part);
    fx.dict_tail = reinterpret_cast<const uint32_t*>(map->dict) + 2 * TDR_CMAP_MAX_DICT;
    fx.nchunks = io.nchunks_dense; fx.ray_split = io.ray_split;
    hipLaunchKernelGGL(score_finalize_exact_kernel, dim3((unsigned)cdiv(io.npad, 64)), dim3(256), 0, s, fx);
    f.run_if = io.flags;
  }
  launch_finalize(f, n, s);
  LAUNCH_CHECK("score_finalize(cart)");
  return TDR_OK;
}

// Self-test hook: the scoring loop's coordinate rounding applied to caller-supplied floats (clamped to [-1, limit]
// like the loop does), so the GPU tests can compare it with roundf over whole float ranges.
__global__ void selftest_round_kernel(const float* __restrict__ x, int64_t n, float limit, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = round_half_away_clamped(__builtin_amdgcn_fmed3f(x[i], -1.f, limit));
}
extern "C" int tdr_k_selftest_round(const float* x, int64_t n, float limit, int32_t* out, void* stream) {
  if (!x || !out || n < 1) return fail(TDR_ERR_ARG, "selftest_round: bad arguments");
  hipLaunchKernelGGL(selftest_round_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, limit,
                     out);
  LAUNCH_CHECK("selftest_round");
  return TDR_OK;
}
