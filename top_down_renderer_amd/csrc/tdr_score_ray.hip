// tdr_score_ray.hip — the RAY-MAPPED polar scoring kernel: one WAVE per particle, lanes = consecutive rings of a direction.
//
// For whom.  The scattered particles of a launch (su_key_kernel, tdr_score_su.hip: the uniform tenth of the bench mix,
// particle initialisation, the tails of a cluster).  With lane = particle (score_polar_kernel, score_polar_su_kernel) a
// wave of scattered particles gathers from 64 unrelated windows: 64 cache lines per gather instruction (160 CU cycles
// measured against 16 for <= 4 lines, tools/ta_cost.hip), every line used by ONE sample, and a window's lines are
// requested once per ring group, far apart in time.  Here a wave walks ONE window from its first sample to its last:
// 64 consecutive rings of one direction per step, i.e. 64 cells along a ray —
//   * a gather touches 8-11 lines instead of 64 (a plane tile holds 8 x 8 cells), 2-3 per 16 lanes: the address path's
//     minimum of 16 cycles per instruction,
//   * neighbouring directions follow each other in time: a line fetched for direction i is still in the L2 for i + 1,
//   * what is fetched is the CLASS PLANE of the one class the scan bin holds (2 bytes per cell, tdr_cmap.hip) — or, for an
//     empty bin, the 16 known bits around the cell from the coarse mask plane: 2 400 lines per window on the config-2 scene
//     where the 8-byte records of all classes take 6 700.
// Why it may: the product sums are EXACT integers (a scan count times a dictionary value that is an integer multiple of
// 2^-q, accumulated in 64 bits), so they do not depend on the order of the additions — lane-major here, sample-major in
// the lane = particle kernels — and a particle's weight is the same bits whichever kernel scored it
// (tests/test_shift_uniform.py, tests/test_ray.py).
//
// What bounds it: the L1 address path (every vector memory instruction of a wave costs it >= 16 cycles) and vector issue,
// about equally (the first version: three memory instructions and ~55 vector instructions per step, 53 CU cycles per step
// measured).  So a step of 64 samples is ONE gather and ~21 vector instructions:
//   * the scan descriptors of a lane's rings of one direction are 16 bits each — class code, count — packed side by side:
//     one load per direction brings up to four steps' worth; the sample offsets likewise (two 16-byte loads);
//   * everything that depends on the bin's class only — the byte offset of its plane (the coarse mask plane for an empty
//     bin), the column shift that turns a cell into a mask cell, the bit the cell's `known` flag sits at, the accumulator the
//     product goes to — comes out of a 16-entry table in LDS with one 16-byte read, so there is no branch and no select in
//     the loop: an empty bin multiplies a count of zero;
//   * mask plane and class planes share one offset formula (plane_offset: same tile shape, same tile-column stride);
//   * a lane adds its products into its own 64-bit accumulators in LDS, one per class (ds_add_u64).
// Bins that hold several classes, or a count beyond 12 bits, are "empty" to the loop and go through a list afterwards.
//
// Per launch: ray_prep_kernel (sample offsets and scan descriptors in ray order, the list, the `inexact` flag), then
// score_polar_ray_kernel over the sparse share of the slot list; score_finalize_exact_kernel (tdr_score.hip) turns the
// integer sums into weights.
#include "tdr_score_dev.h"
#include "tdr_score_su.h"

// ray order: a direction's rings in BLOCKS of GQ steps of 64 rings (GQ = 1, 2 or 4: ray_gq); "row" m = direction * blocks
// + block.  Lane l of step g of block b is ring (b * GQ + g) * 64 + l.
// BLOCK-MAJOR order (bm; launches whose table comes with its factors): one step per row, GQ = 1, and the rows of ring
// segment 0 (rings 0..63, every direction) first, then segment 1's, ...: row m = segment * nb + direction.  The eight
// gathers a wave has in flight are then neighbouring directions of ONE segment (config 2, uniform particles: 9.1 -> 8.4 ms;
// with the offsets read from tab_ray the narrow rows cost more than that gains, so those launches keep the first order).
// tdr_config_tuning("ray_block_major", 0) keeps the first order everywhere (A/B; same bits).
// PATCH order (round 5; block-major launches whose direction count is a multiple of 16 and at most RAY_PATCH_MAX_NB).
// Counters of the block-major ray order (100 000 uniform particles, profiles/r05_pmc_ray_patch_v1.txt): the L1 address path
// 0.87 busy at 38 cycles per step — for THREE vector memory instructions a step (the gather, the 2-byte descriptors, the
// radii), each at least 16 cycles there (four lanes a cycle, whatever a lane fetches).  So the step should cost ONE and a
// quarter:
//   * a UNIT is 16 scan rows x 16 rings = four steps; a lane's four 16-bit descriptors of a unit lie side by side: ONE 8-byte
//     load per unit.  That needs the unit aligned in the descriptor array whatever the particle's heading: units are groups of 16
//     SCAN rows, the window direction of scan row r is (r - shift) mod nb — the window is a full circle, any rotation of the
//     grouping walks all of it;
//   * a step's 64 lanes are 4 neighbouring directions x 16 consecutive rings — a PATCH of the window, not a ray (fewer tiles);
//     the four directions' {cos, sin} pairs are no longer uniform over the wave: they come out of LDS (the factors' direction
//     table, 8 bytes a direction), the lane's radius depends on the ring block only: four registers per 64-ring segment;
//   * order: segment of 64 rings -> group of 16 scan rows -> the segment's four ring blocks -> the unit's four steps, so that
//     what a wave has just fetched is the sector next to the one it fetches now.
// Lane l = direction (l >> 4) of the step's four, ring (l & 15) of the block.  tdr_config_tuning("ray_patch", 0): the ray order
// (A/B; same bits).
#define RAY_PR 16             // rings of a patch
#define RAY_PG 16             // scan rows of a unit (four steps of four directions)
#define RAY_PATCH_MAX_NB 256
static bool g_ray_borrow = true;   // empty bins borrow a neighbour's class plane for their known bit (ray_prep_kernel)
extern "C" int tdr_config_ray_borrow(int on) {   // < 0: query only
  if (on >= 0) g_ray_borrow = on != 0;
  return g_ray_borrow ? 1 : 0;
}
static bool g_ray_patch = true;
extern "C" int tdr_config_ray_patch(int on) {   // < 0: query only
  if (on >= 0) g_ray_patch = on != 0;
  return g_ray_patch ? 1 : 0;
}
static bool g_ray_bm = true;
extern "C" int tdr_config_ray_block_major(int on) {   // < 0: query only
  if (on >= 0) g_ray_bm = on != 0;
  return g_ray_bm ? 1 : 0;
}
static inline bool ray_bm(const SuLaunch& L) { return g_ray_bm && L.fac != nullptr; }
static inline bool ray_patch(const SuLaunch& L) { return ray_bm(L) && g_ray_patch && L.nb % RAY_PG == 0 && L.nb <= RAY_PATCH_MAX_NB; }
static inline int ray_gq(int nr, bool bm) { return bm ? 1 : (nr <= 64 ? 1 : (nr <= 128 ? 2 : 4)); }
static inline int ray_blocks(int nr, bool bm) { return (int)cdiv(nr, 64 * ray_gq(nr, bm)); }
// (the first order pads to whole blocks of GQ steps: never less than the block-major order needs)
int64_t tdr_ray_padded_samples(int nb, int nr) { return (int64_t)nb * ray_blocks(nr, false) * ray_gq(nr, false) * 64; }

// One thread per (scan row, padded ring).  tab_ray[((i * blocks + b) * 64 + l) * GQ + g] = sample offset of (direction i,
// ring j); desc_ray (16-bit) at the same index for scan row i, ring j: code << 12 | count — code 0: nothing for the loop
// (an empty bin, or one that went on the list), c + 1: class c alone.  Rings beyond nr: an offset far outside the map (their
// cell is the guard cell: unknown), descriptor 0.
// `list`: bins holding several classes or a count >= 4096, as row << 16 | ring.
// inexact[0] is raised when the scan has no integer form: a count that is negative, fractional, not finite or >= 2^24, or
// a dictionary without one (tdr_cmap.hip); inexact[1] collects the bound on the total count (int_form_off).
// fac (optional): the table's factors (tdr_polar_factors_host).  rad_ray[(b * 64 + l) * GQ + g] = ring j's radius (rings
// beyond nr: 1e30 — one of a direction's two products then leaves the map whatever the direction); inexact[2] is raised
// when an entry of `tab` is not the float product its factors give (with a uniform scale: that product, scaled like
// utab_kernel scales the table) — the scoring kernel then reads tab_ray instead of multiplying the factors itself.
__global__ __launch_bounds__(256) void ray_prep_kernel(const float* __restrict__ tab, const float* __restrict__ scan_pk, int nb,
                                                       int nr, int rf, int ncls, int gq, int blocks,
                                                       const uint32_t* __restrict__ dict_tail, float* __restrict__ tab_ray,
                                                       uint16_t* __restrict__ desc_ray, uint32_t* __restrict__ list,
                                                       int32_t* __restrict__ n_list, int32_t* __restrict__ inexact,
                                                       const float* __restrict__ fac, float uscale, float res,
                                                       float* __restrict__ rad_ray, int bm, int patch, int borrow) {
  const int rpad = blocks * gq * 64;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0 && dict_tail[1] != 1u) atomicOr(inexact, 1);
  const bool live = t < (int64_t)nb * rpad;
  const int i = live ? (int)(t / rpad) : 0, j = live ? (int)(t - (int64_t)i * rpad) : 0;
  const int64_t k = (int64_t)j * nb + i;
  const bool real = live && j < nr;
  uint32_t mass = 0;
  if (real) {
    const float sum = scan_pk[k * rf + rf - 1];
    if (sum >= 1.f && sum < 16777216.f) mass = ((uint32_t)sum >> 8) + 1u;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mass += __shfl_xor(mass, d, 64);
  if ((threadIdx.x & 63) == 0 && mass) atomicAdd(reinterpret_cast<unsigned*>(inexact) + 1, mass);
  if (!live) return;
  const int g = j >> 6, l = j & 63, b = g / gq;
  const int64_t at = bm ? ((int64_t)b * nb + i) * 64 + l : (((int64_t)i * blocks + b) * 64 + l) * gq + (g - b * gq);
  // patch order (descriptors only; the offsets keep the block-major order): unit (ring block j / 16, scan-row group i / 16),
  // lane (i & 3) * 16 + (j & 15), step (i & 15) >> 2 — a lane's four steps side by side
  int64_t at_d = at;
  if (patch) at_d = ((((int64_t)(j / RAY_PR) * (nb / RAY_PG) + i / RAY_PG) * 64 + (i & 3) * RAY_PR + (j % RAY_PR)) << 2) + ((i % RAY_PG) >> 2);
  float tx = -1.0e30f, ty = -1.0e30f;
  uint32_t d = 0;
  if (fac && i == 0) rad_ray[((int64_t)b * 64 + l) * gq + (g - b * gq)] = real ? fac[2 * nb + j] : 1.0e30f;
  if (real) {
    tx = tab[2 * k];
    ty = tab[2 * k + 1];
    if (fac) {
      float fx = fac[2 * i] * fac[2 * nb + j], fy = fac[2 * i + 1] * fac[2 * nb + j];
      if (uscale > 0.f) {
        fx = (fx * uscale) * res;
        fy = (fy * uscale) * res;
      }
      if (__float_as_uint(fx) != __float_as_uint(tx) || __float_as_uint(fy) != __float_as_uint(ty)) atomicOr(inexact + 2, 1);
    }
    const float* r = scan_pk + k * rf;
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
    if (!ok) atomicOr(inexact, 1);
    else if (nz == 1 && r[first] < 4096.f) d = (uint32_t)r[first] | ((uint32_t)(first + 1) << 12);
    else if (nz >= 1) list[atomicAdd(n_list, 1)] = ((uint32_t)i << 16) | (uint32_t)j;   // (any order: the sums are exact)
  }
  // An EMPTY bin (and one that went on the list) needs its cell's known bit and nothing else — and every class plane carries that
  // bit (bit 15 of a cell).  Four consecutive lanes of a gather — four consecutive rings of one scan row, in every order above —
  // are served together by the L1's address path, at a cost per distinct LINE among them: an empty bin between two bins of
  // class c that reads the coarse mask plane is a line of its own, one that reads class c's plane with a count of zero rides
  // along.  So an empty bin borrows the class of the nearest non-empty bin of its aligned group of four rings (none: code 0,
  // the mask plane, one line for the four); the product with a zero count adds nothing (round 5).
  {
    const uint32_t own = d >> 12;
    const int ql = threadIdx.x & 3;
    uint32_t c4[4];
#pragma unroll
    for (int q = 0; q < 4; q++) c4[q] = __shfl(own, (threadIdx.x & 60) + q, 64);   // (threads of a group: the same scan row i, rings 4 q' .. 4 q' + 3)
    if (borrow && own == 0) {
      uint32_t pick = 0;
#pragma unroll
      for (int dist = 3; dist >= 1; dist--) {   // the nearest wins (written last)
        if (ql + dist < 4 && c4[(ql + dist) & 3]) pick = c4[(ql + dist) & 3];
        if (ql - dist >= 0 && c4[(ql - dist) & 3]) pick = c4[(ql - dist) & 3];
      }
      d = pick << 12;
    }
  }
  tab_ray[2 * at] = tx;
  tab_ray[2 * at + 1] = ty;
  desc_ray[at_d] = (uint16_t)d;
}

struct RayArgs {
  const uint32_t* crec;     // compact map: tiles, known mask, class planes, coarse mask plane (byte offsets from here)
  unsigned planes_off;      // byte offset of class plane 0
  unsigned plane_bytes;     // bytes of a class plane
  unsigned cmask_off;       // byte offset of the coarse mask plane
  int pkcol;                // plane_offset: bytes of a tile column - 16 (the same for every plane)
  const uint32_t* dict_int;
  int dict_n;
  int rows, cols;
  float resolution;
  const float* tab;         // [P][2] in the table's own order (the list pass)
  const float* tab_ray;
  const float* fac;         // the table's factors, or NULL; rad_ray: the radii in ray order
  const float* rad_ray;
  float uscale;
  const uint16_t* desc_ray;
  const uint32_t* list;
  const int32_t* n_list;
  const float* scan_pk;
  int rf, ncls, nb, nr, blocks;
  float res;
  const float* st;
  int64_t cap;
  const int32_t* slots;     // the launch's slot list; the sparse share is slots [counts[0], counts[0] + counts[1])
  const int32_t* counts;
  const int32_t* inexact;
  int nsplit;               // waves per particle: each takes a contiguous share of the directions and writes one chunk row
  int64_t npad;
  uint32_t* part;           // [>= nsplit][2 ncls + 2][npad], like score_polar_su_kernel
};

// FAC: the sample offsets are multiplied out of the table's factors — a direction's pair (uniform over the wave: two scalar
// loads) times the lane's own radii (registers) — instead of read from tab_ray: the same float products the table holds
// (ray_prep_kernel checked that), and 16 bytes per lane and row less through the texture path.
// lacc [4 waves][ncls + 1][64 lanes]: a lane's sums per class (slot 0: no class); lut, per class code: {plane constant,
// column shift, known-bit index, accumulator}
// PATCH (with BM): the patch order (see the top of the file); ldir: the directions' pairs in LDS
template <int GQ, bool USCALE, bool FAC, bool BM, bool PATCH>
__device__ __forceinline__ void ray_body(const RayArgs& a, unsigned long long* lacc, uint32_t* ldict, uint4* lut, float2* ldir) {
  static_assert(!BM || GQ == 1, "block-major: one step per row");
  static_assert(!PATCH || BM, "the patch order is a block-major order");
  const int nsparse = a.counts[1];
  if ((int64_t)blockIdx.x * 4 >= (int64_t)nsparse * a.nsplit) return;   // whole workgroup idle (uniform)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int t = threadIdx.x; t < a.dict_n; t += 256) ldict[t] = a.dict_int[t];
  if constexpr (PATCH && FAC)
    for (int t = threadIdx.x; t < a.nb; t += 256) ldir[t] = reinterpret_cast<const float2*>(a.fac)[t];
  const unsigned pconst = (unsigned)(a.pkcol + 16) + 128u;   // plane_offset's constant without the plane's own offset
  if (threadIdx.x < 16) {
    const int code = threadIdx.x;
    uint4 e;
    if (code >= 1 && code <= a.ncls) {   // class code - 1 alone: its plane, cells as they are, `known` in bit 15
      e.x = a.planes_off + (unsigned)(code - 1) * a.plane_bytes + pconst;
      e.y = 0; e.z = 15; e.w = (unsigned)code * 512u;
    } else {                             // nothing to multiply: the coarse mask plane (4 x 4 map cells per cell)
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
  if (q >= nsparse) return;   // (wave-uniform; no barrier below)
  const int64_t slot = (int64_t)a.counts[0] + q;
  const int64_t p = a.slots[slot];
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float fscale = USCALE ? a.uscale : scale;   // (the caller's promise: the same number)
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];  // state_particle.cpp:161
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];  // :162
  const float off0 = cy / a.resolution;  // top_down_map_polar.cpp:29
  const float off1 = cx / a.resolution;  // :30
  const int shift = rot_shift_dev(a.st[TDR_ST_THETA * a.cap + p], a.nb);
  const float rmaxf = (float)a.rows, cmaxf = (float)a.cols;
  const char* __restrict__ crecb = reinterpret_cast<const char*>(a.crec);
  typedef float tdr_v2f __attribute__((ext_vector_type(2)));
  const tdr_v2f offv = {off0, off1};
  auto cell = [&](float tx, float ty, int& ri, int& ci) {
    tdr_v2f pv = {tx, ty};
    if constexpr (!USCALE) pv = (pv * scale) * a.res;  // top_down_map_polar.cpp:28
    pv = pv + offv;                                     // :29-30
    tdr_v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
    qv = qv + 0.49999997f;                              // round_half_away_clamped
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ri) : "v"(qv.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ci) : "v"(qv.y));
  };
  auto cell_fac = [&](tdr_v2f dir, float rad, int& ri, int& ci) {
    tdr_v2f pv = dir * rad;                             // top_down_map_polar.cpp:17-18
    pv = (pv * fscale) * a.res;                         // :28
    pv = pv + offv;
    tdr_v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
    qv = qv + 0.49999997f;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ri) : "v"(qv.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ci) : "v"(qv.y));
  };

  // this wave's share of the window: rows m = direction * blocks + block; a window row meets scan row (m + shift * blocks)
  // mod (nb * blocks)
  const int rows_all = a.nb * a.blocks, per = (rows_all + a.nsplit - 1) / a.nsplit;
  const int m0 = part_id * per, m1 = min(rows_all, m0 + per);
  const int rot = shift * a.blocks;
  int bm_b = m0 / a.nb, bm_i = m0 - bm_b * a.nb;   // BM: the row's block and direction, kept by counting
  uint32_t known = 0, norm = 0;
  constexpr int U = 8 / GQ;   // rows whose gathers are in flight together: 8 steps
  typedef uint16_t desc_t __attribute__((ext_vector_type(GQ)));
  typedef float tab_t __attribute__((ext_vector_type(2 * GQ)));
  typedef float rad_t __attribute__((ext_vector_type(GQ)));
  const desc_t* __restrict__ descv = reinterpret_cast<const desc_t*>(a.desc_ray);
  const tab_t* __restrict__ tabv = reinterpret_cast<const tab_t*>(a.tab_ray);
  const rad_t* __restrict__ radv = reinterpret_cast<const rad_t*>(a.rad_ray);
  const tdr_v2f* __restrict__ dirv = reinterpret_cast<const tdr_v2f*>(a.fac);
  const bool one_block = a.blocks == 1;
  rad_t rad0 = {};
  if constexpr (FAC) rad0 = radv[lane];   // (block 0's; the only block of a window of up to 256 rings)
  // one sample, first half: everything that depends on the bin's class out of the table, the cell's address, the gather
  auto issue = [&](uint32_t d, int ri, int ci, uint32_t& v, uint32_t& shb, uint32_t& cnt, uint32_t& acc_at) {
    cnt = d & 0xFFFu;
    // one 16-byte LDS read: everything that depends on the bin's class
    const uint4 e = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lut) + ((d >> 8) & 0xF0u));
    const int rr = ri >> (int)e.y, cc = ci >> (int)e.y;  // a mask cell spans 4 x 4 map cells
    const unsigned off = plane_offset(rr, cc, a.pkcol, (int)e.x);
    // bit of `known`: 15 in a class cell, (row & 3) * 4 + (column & 3) in a mask cell
    shb = ((((uint32_t)ri & 3u) << 2) | ((uint32_t)ci & 3u)) | e.z;
    acc_at = e.w;
    v = *reinterpret_cast<const uint16_t*>(crecb + off);
  };
  // ... second half: the known bit, the normalisation, the product into the lane's accumulator of the class
  auto consume = [&](uint32_t v, uint32_t shb, uint32_t cnt, uint32_t acc_at) {
    uint32_t kbit;
    asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(kbit) : "v"(v), "v"(shb));
    known += kbit;
    norm = __umul24(cnt, kbit) + norm;                  // state_particle.cpp:141-142
    const uint32_t D = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ldict) + (v & 0xFFCu));
    const unsigned long long prod = (unsigned long long)cnt * D;   // :136-139, as integers (0 for an empty bin)
    unsigned long long* const acc = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(my) + acc_at);
    __hip_atomic_fetch_add(acc, prod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_u64; the lane's own slot
  };
  auto rows_step = [&](auto cnt_c, int m) {
    constexpr int N = decltype(cnt_c)::value;
    desc_t dd[N];
    tab_t tt[N];
    rad_t rr_[N];
    tdr_v2f dir[N];
#pragma unroll
    for (int u = 0; u < N; u++) {
      int mr = m + u + rot;
      mr -= mr >= rows_all ? rows_all : 0;
      if constexpr (BM) {
        int r = bm_i + shift;
        r -= r >= a.nb ? a.nb : 0;
        mr = bm_b * a.nb + r;
      }
      dd[u] = descv[(int64_t)mr * 64 + lane];
      if constexpr (FAC && BM) {
        dir[u] = dirv[bm_i];
        rr_[u] = radv[(int64_t)bm_b * 64 + lane];
      } else if constexpr (FAC) {
        const int i = one_block ? m + u : (m + u) / a.blocks;
        dir[u] = dirv[i];
        rr_[u] = one_block ? rad0 : radv[(int64_t)(m + u - i * a.blocks) * 64 + lane];
      } else {
        tt[u] = tabv[(int64_t)(m + u) * 64 + lane];
      }
      if constexpr (BM) {
        bm_i++;
        if (bm_i == a.nb) { bm_i = 0; bm_b++; }
      }
    }
    uint32_t v[N * GQ], shb[N * GQ], cnt[N * GQ], acc_at[N * GQ];
#pragma unroll
    for (int u = 0; u < N; u++)
#pragma unroll
      for (int g = 0; g < GQ; g++) {
        const int s = u * GQ + g;
        int ri, ci;
        if constexpr (FAC) cell_fac(dir[u], rr_[u][g], ri, ci);
        else cell(tt[u][2 * g], tt[u][2 * g + 1], ri, ci);
        issue(dd[u][g], ri, ci, v[s], shb[s], cnt[s], acc_at[s]);
      }
#pragma unroll
    for (int s = 0; s < N * GQ; s++) consume(v[s], shb[s], cnt[s], acc_at[s]);
  };
  if constexpr (PATCH) {
    // the patch order (top of the file): a wave takes a contiguous share of the SECTORS (segment of 64 rings, group of 16 scan
    // rows); a sector = the segment's four ring blocks = four units of four steps; two units' gathers are in flight together
    const int ngroups = a.nb / RAY_PG, sectors_all = a.blocks * ngroups;
    const int sper = (sectors_all + a.nsplit - 1) / a.nsplit;
    const int s0 = part_id * sper, s1 = min(sectors_all, s0 + sper);
    const int pl_d = lane >> 4, pl_r = lane & (RAY_PR - 1);
    const uint2* __restrict__ descq = reinterpret_cast<const uint2*>(a.desc_ray);
    float rad4[4] = {0.f, 0.f, 0.f, 0.f};
    int cur_seg = -1;
    int seg = s0 / ngroups, pg = s0 - seg * ngroups;
    for (int sc = s0; sc < s1; sc++) {
      if (seg != cur_seg) {   // (wave-uniform) the lane's radii of the segment's four ring blocks
        cur_seg = seg;
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
          const int j = (seg * 4 + qq) * RAY_PR + pl_r;
          rad4[qq] = j < a.nr ? a.fac[2 * a.nb + j] : 1.0e30f;   // (a ring the image does not have: its cell leaves the map)
        }
      }
      // the window directions of the sector's four steps: scan row 16 pg + 4 k + (lane >> 4) meets direction (row - shift) mod nb
      int wi[4];
      tdr_v2f wdir[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int i = pg * RAY_PG + 4 * k + pl_d - shift;
        i += i < 0 ? a.nb : 0;
        wi[k] = i;
        if constexpr (FAC) {
          const float2 dv = ldir[i];
          wdir[k] = (tdr_v2f){dv.x, dv.y};
        }
      }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        uint2 dd[2];
#pragma unroll
        for (int u = 0; u < 2; u++) dd[u] = descq[((int64_t)(seg * 4 + 2 * h + u) * ngroups + pg) * 64 + lane];
        uint32_t v[8], shb[8], cnt[8], acc_at[8];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
          for (int k = 0; k < 4; k++) {
            int ri, ci;
            if constexpr (FAC) {
              cell_fac(wdir[k], rad4[2 * h + u], ri, ci);
            } else {   // (the table is not its factors' products: its own entries, in its own order)
              const int j = min((seg * 4 + 2 * h + u) * RAY_PR + pl_r, a.nr - 1);
              const int64_t kk = (int64_t)j * a.nb + wi[k];
              if ((seg * 4 + 2 * h + u) * RAY_PR + pl_r < a.nr) cell(a.tab[2 * kk], a.tab[2 * kk + 1], ri, ci);
              else cell(-1.0e30f, -1.0e30f, ri, ci);
            }
            const uint32_t w = k < 2 ? dd[u].x : dd[u].y;
            issue((k & 1) ? (w >> 16) : (w & 0xFFFFu), ri, ci, v[u * 4 + k], shb[u * 4 + k], cnt[u * 4 + k], acc_at[u * 4 + k]);
          }
#pragma unroll
        for (int t = 0; t < 8; t++) consume(v[t], shb[t], cnt[t], acc_at[t]);
      }
      pg++;
      if (pg == ngroups) { pg = 0; seg++; }
    }
  } else {
    int m = m0;
    for (; m + U <= m1; m += U) rows_step(std::integral_constant<int, U>{}, m);
    for (; m < m1; m++) rows_step(std::integral_constant<int, 1>{}, m);
  }

  // the list: bins that hold several classes (or one large count).  The loop above saw them as empty bins — their cell's
  // known bit is counted — ; here every class present meets its plane, and the bin's count enters the normalisation.
  {
    const int nm = *a.n_list, mper = (nm + a.nsplit - 1) / a.nsplit;
    const int e1 = min(nm, (part_id + 1) * mper);
    for (int e = part_id * mper + lane; e < e1; e += 64) {
      const uint32_t w = a.list[e];
      const int r = (int)(w >> 16), j = (int)(w & 0xFFFFu);
      int i = r - shift;   // the window row that meets scan row r
      i += i < 0 ? a.nb : 0;
      const int64_t k = (int64_t)j * a.nb + i, bin = (int64_t)j * a.nb + r;
      int ri, ci;
      cell(a.tab[2 * k], a.tab[2 * k + 1], ri, ci);
      const unsigned cell_off = plane_offset(ri, ci, a.pkcol, (int)(a.planes_off + pconst));
      uint32_t kbit = 0;
      for (int c = 0; c < a.ncls; c++) {
        const float s = a.scan_pk[bin * a.rf + c];
        if (s != 0.f) {
          const uint32_t vv = *reinterpret_cast<const uint16_t*>(crecb + cell_off + (unsigned)c * a.plane_bytes);
          kbit = vv >> 15;
          __hip_atomic_fetch_add(&my[(c + 1) * 64], (unsigned long long)(uint32_t)s * ldict[(vv & 0xFFCu) >> 2],
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      norm += (uint32_t)a.scan_pk[bin * a.rf + a.rf - 1] & (0u - kbit);
    }
  }
  // lanes -> one sum per class (the rings a block pads a direction with fell on the guard cell — unknown — and counted nothing)
  uint32_t* o = a.part + (int64_t)part_id * (2 * a.ncls + 2) * a.npad + slot;
  for (int c = 0; c < a.ncls; c++) {
    unsigned long long s = __hip_atomic_load(&my[(c + 1) * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int dlt = 32; dlt > 0; dlt >>= 1) s += __shfl_xor(s, dlt, 64);
    if (lane == 0) {
      o[(int64_t)(2 * c) * a.npad] = (uint32_t)s;
      o[(int64_t)(2 * c + 1) * a.npad] = (uint32_t)(s >> 32);
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
template <int GQ, bool USCALE, bool BM = false, bool PATCH = false>
__global__ __launch_bounds__(256) void score_polar_ray_kernel(RayArgs a) {
  extern __shared__ unsigned long long lacc[];
  __shared__ uint32_t ldict[TDR_CMAP_MAX_DICT];
  __shared__ uint4 lut[16];
  __shared__ float2 ldir[PATCH ? RAY_PATCH_MAX_NB : 1];
  if (int_form_off(a.inexact)) return;
  // with factors that ARE the table's (ray_prep_kernel's check; uniform over the launch) the offsets are multiplied out
  if (a.fac && a.inexact[2] == 0) ray_body<GQ, USCALE, true, BM, PATCH>(a, lacc, ldict, lut, ldir);
  else ray_body<GQ, USCALE, false, BM, PATCH>(a, lacc, ldict, lut, ldir);
}


// ---- host side ---------------------------------------------------------------------------------------------------------
extern "C" size_t tdr_cmap_plane_offset_words(int ncls, int rows, int cols);
extern "C" size_t tdr_cmap_plane_words(int ncls, int rows, int cols);

bool tdr_ray_map_ok(const tdr_map_desc* map) {
  return map->crec && map->dict && map->cwords == tdr_cmap_words(map->ncls) && map->dict_n > 0 &&
         map->dict_n <= TDR_CMAP_MAX_DICT && tdr_cmap_plane_words(map->ncls, map->rows, map->cols) != 0;
}
static int g_ray_split = 0;   // 0: chosen per launch
extern "C" int tdr_config_ray_split(int k) {   // >= 1: force; 0: per launch (default); < 0: query only
  if (k >= 0) g_ray_split = k > TDR_RAY_MAX_SPLIT ? TDR_RAY_MAX_SPLIT : k;
  return g_ray_split;
}
bool tdr_ray_block_major(const SuLaunch& L) { return ray_bm(L); }
int tdr_ray_splits(int nb, int nr, int64_t n, bool bm) {
  if (g_ray_split > 0) return g_ray_split;
  // waves per particle: a window is nb * blocks rows of up to four steps; small launches split it to fill the chip (the
  // sums are exact: any split gives the same bits)
  const int64_t rows = (int64_t)nb * ray_blocks(nr, bm);
  int s = 1;
  while (s < TDR_RAY_MAX_SPLIT && n * s < 32768 && rows / (2 * s) >= 8) s *= 2;
  if (s == 1 && rows >= 128) s = 2;
  // (config 2, 100 000 uniform particles at a span of 8, ms per call at 1 / 2 / 4 / 8 waves per particle — block-major:
  // 8.42 / 8.36 / 8.33 / 8.69; the first order: 10.35 / 10.42 / 10.35 / 9.66; the bench mix moves by 1 % at most)
  if (s == 2 && bm && rows >= 512) s = 4;
  if (s == 2 && !bm && rows >= 256) s = 8;
  return s;
}

int tdr_ray_prepare(const SuLaunch& L, const SuWs& W, hipStream_t s) {
  int32_t* base = L.ws;
  int* ints = base + W.ints + 3 * (L.nb + 1);   // [counts 3][n_list][inexact][mass bound]
  const bool bm = ray_bm(L);
  const int gq = ray_gq(L.nr, bm), blocks = ray_blocks(L.nr, bm);
  const int64_t T = tdr_ray_padded_samples(L.nb, L.nr);
  hipLaunchKernelGGL(ray_prep_kernel, dim3((unsigned)cdiv(T, 256)), dim3(256), 0, s, L.tab, L.scan_pk, L.nb, L.nr, L.rf,
                     L.map->ncls, gq, blocks, reinterpret_cast<const uint32_t*>(L.map->dict) + 2 * TDR_CMAP_MAX_DICT,
                     reinterpret_cast<float*>(base + W.ray_tab), reinterpret_cast<uint16_t*>(base + W.ray_desc),
                     reinterpret_cast<uint32_t*>(base + W.ray_multi), ints + 3, ints + 4, L.fac,
                     L.uniform_scale ? L.uscale : 0.f, L.res, reinterpret_cast<float*>(base + W.ray_rad), bm ? 1 : 0,
                     ray_patch(L) ? 1 : 0, g_ray_borrow ? 1 : 0);
  LAUNCH_CHECK("ray_prep");
  return TDR_OK;
}

int tdr_ray_score(const SuLaunch& L, const SuWs& W, hipStream_t s) {
  const tdr_map_desc* map = L.map;
  int32_t* base = L.ws;
  int* ints = base + W.ints + 3 * (L.nb + 1);
  RayArgs r;
  r.crec = map->crec;
  const size_t pw = tdr_cmap_plane_words(map->ncls, map->rows, map->cols);
  r.planes_off = (unsigned)(tdr_cmap_plane_offset_words(map->ncls, map->rows, map->cols) * 4);
  r.plane_bytes = (unsigned)(pw * 4);
  r.cmask_off = r.planes_off + (unsigned)map->ncls * r.plane_bytes;
  r.pkcol = plane_trows(map->rows) * 128 - 16;
  r.dict_int = reinterpret_cast<const uint32_t*>(map->dict) + TDR_CMAP_MAX_DICT;
  r.dict_n = map->dict_n;
  r.rows = map->rows; r.cols = map->cols; r.resolution = map->resolution;
  r.tab = L.tab;
  r.tab_ray = reinterpret_cast<const float*>(base + W.ray_tab);
  r.fac = L.fac;
  r.rad_ray = reinterpret_cast<const float*>(base + W.ray_rad);
  r.uscale = L.uscale;
  r.desc_ray = reinterpret_cast<const uint16_t*>(base + W.ray_desc);
  r.list = reinterpret_cast<const uint32_t*>(base + W.ray_multi);
  r.n_list = ints + 3;
  r.scan_pk = L.scan_pk;
  r.rf = L.rf; r.ncls = map->ncls; r.nb = L.nb; r.nr = L.nr; r.blocks = ray_blocks(L.nr, ray_bm(L)); r.res = L.res;
  r.st = L.st; r.cap = L.cap;
  r.slots = base + W.slots;
  r.counts = ints;
  r.inexact = ints + 4;
  r.nsplit = L.ray_split;
  r.npad = L.npad;
  r.part = reinterpret_cast<uint32_t*>(L.part);
  const dim3 grid((unsigned)cdiv(L.n * r.nsplit, 4)), block(256);
  const size_t lds = (size_t)4 * (map->ncls + 1) * 64 * sizeof(unsigned long long);
#define TDR_LAUNCH_RAY(GQ)                                                                            \
  if (L.uniform_scale) hipLaunchKernelGGL((score_polar_ray_kernel<GQ, true>), grid, block, lds, s, r); \
  else hipLaunchKernelGGL((score_polar_ray_kernel<GQ, false>), grid, block, lds, s, r);
  if (ray_patch(L)) {
    if (L.uniform_scale) hipLaunchKernelGGL((score_polar_ray_kernel<1, true, true, true>), grid, block, lds, s, r);
    else hipLaunchKernelGGL((score_polar_ray_kernel<1, false, true, true>), grid, block, lds, s, r);
  } else if (ray_bm(L)) {
    if (L.uniform_scale) hipLaunchKernelGGL((score_polar_ray_kernel<1, true, true>), grid, block, lds, s, r);
    else hipLaunchKernelGGL((score_polar_ray_kernel<1, false, true>), grid, block, lds, s, r);
  } else switch (ray_gq(L.nr, false)) {
    case 1: TDR_LAUNCH_RAY(1) break;
    case 2: TDR_LAUNCH_RAY(2) break;
    default: TDR_LAUNCH_RAY(4) break;
  }
#undef TDR_LAUNCH_RAY
  LAUNCH_CHECK("score_polar_ray");
  return TDR_OK;
}
