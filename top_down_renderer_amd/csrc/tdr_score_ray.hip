// tdr_score_ray.hip — the RAY-MAPPED polar scoring kernel: one WAVE per particle, lanes = consecutive samples along a ray.
//
// For whom.  The scattered particles of a launch (su_key_kernel, tdr_score_su.hip: the uniform tenth of the bench mix,
// particle initialisation, the tails of a cluster).  With lane = particle (score_polar_kernel, score_polar_su_kernel) a
// wave of scattered particles gathers from 64 unrelated windows: 64 cache lines per gather instruction (160 CU cycles
// measured against 16 for <= 4 lines, tools/ta_cost.hip), every line used by ONE sample, and a window's lines are
// requested once per ring group, far apart in time.  Here a wave walks ONE window from its first sample to its last:
// 64 consecutive rings of one direction per step, i.e. 64 cells along a ray —
//   * a gather touches 8-11 lines instead of 64 (a plane tile holds 8 x 8 cells),
//   * neighbouring directions follow each other in time: a line fetched for direction i is still in the L2 for i + 1,
//   * what is fetched is the CLASS PLANE of the one class the scan bin holds (2 bytes per cell, tdr_cmap.hip) — or, for an
//     empty bin, the word of the known mask: 2 400 lines per window on the config-2 scene where the 8-byte records of all
//     classes take 6 700.
// Why it may: the product sums are EXACT integers (a scan count times a dictionary value that is an integer multiple of
// 2^-q, accumulated in 64 bits), so they do not depend on the order of the additions — lane-major here, sample-major in
// the lane = particle kernels — and a particle's weight is the same bits whichever kernel scored it
// (tests/test_shift_uniform.py, tests/test_ray.py).
//
// Per launch: ray_prep_kernel (sample table and scan descriptors in ray order, the list of bins holding several classes,
// the `inexact` flag), then score_polar_ray_kernel over the sparse share of the slot list; score_finalize_exact_kernel
// (tdr_score.hip) turns the integer sums into weights.
#include "tdr_score_dev.h"
#include "tdr_score_su.h"

#define RAY_U 8   // steps (of 64 samples) whose gathers a wave keeps in flight

// One thread per window sample in RAY order k' = i * nr + j (direction i, ring j; the reference's images are column-major,
// k = i + nb * j).  tab_ray[k'] = the sample's offset; desc_ray[k'] = the scan bin (row i, ring j) as count | code << 24
// — code 0: empty, c + 1: class c alone (count = its count), 0xFF: several classes (count = their sum; the bin goes on
// the `multi` list as {k', j * nb + i}).  A window row i is paired with scan row (i + shift) mod nb: index
// (k' + shift * nr) mod P of desc_ray.
// `inexact` is raised when the scan has no integer form: a count that is negative, fractional, not finite or >= 2^24,
// or a dictionary without one (tdr_cmap.hip) — the integer kernels then return at once and the float kernel runs.
__global__ __launch_bounds__(256) void ray_prep_kernel(const float* __restrict__ tab, const float* __restrict__ scan_pk, int nb,
                                                       int nr, int rf, int ncls, const uint32_t* __restrict__ dict_tail,
                                                       float* __restrict__ tab_ray, uint32_t* __restrict__ desc_ray,
                                                       uint32_t* __restrict__ multi, int32_t* __restrict__ n_multi,
                                                       int32_t* __restrict__ inexact) {
  const int64_t P = (int64_t)nb * nr;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0 && dict_tail[1] != 1u) atomicOr(inexact, 1);
  // (inexact[1]: upper bound of the scan's total count / 256, see int_form_off)
  uint32_t mass = 0;
  if (t < P) {
    const float sum = scan_pk[((int64_t)(t % nr) * nb + t / nr) * rf + rf - 1];
    if (sum >= 1.f && sum < 16777216.f) mass = ((uint32_t)sum >> 8) + 1u;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) mass += __shfl_xor(mass, d, 64);
  if ((threadIdx.x & 63) == 0 && mass) atomicAdd(reinterpret_cast<unsigned*>(inexact) + 1, mass);
  if (t >= P) return;
  const int i = (int)(t / nr), j = (int)(t - (int64_t)i * nr);
  const int64_t k = (int64_t)j * nb + i;
  tab_ray[2 * t] = tab[2 * k];
  tab_ray[2 * t + 1] = tab[2 * k + 1];
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
  if (!ok) { atomicOr(inexact, 1); desc_ray[t] = 0; return; }
  uint32_t d = 0;
  if (nz == 1) d = (uint32_t)r[first] | ((uint32_t)(first + 1) << 24);
  else if (nz > 1) {
    d = (uint32_t)sum | 0xFF000000u;
    const int at = atomicAdd(n_multi, 1);   // (any order: the sums are exact)
    multi[2 * at] = (uint32_t)t;
    multi[2 * at + 1] = (uint32_t)k;
  }
  desc_ray[t] = d;
}

struct RayArgs {
  const uint32_t* crec;     // compact map: tiles, known mask, class planes (byte offsets from here)
  unsigned kmask_off;       // the mask's byte offset
  int kmask_col;            // bytes of one of its tile columns
  unsigned planes_off;      // byte offset of class plane 0
  unsigned plane_units;     // bytes of a plane / 128
  int pkcol;                // plane_offset: bytes of a tile column - 16
  const uint32_t* dict_int;
  int dict_n;
  int rows, cols;
  float resolution;
  const float* tab_ray;
  const uint32_t* desc_ray;
  const uint32_t* multi;
  const int32_t* n_multi;
  const float* scan_pk;
  int rf, ncls, nb, nr;
  float res;
  const float* st;
  int64_t cap;
  const int32_t* slots;     // the launch's slot list; the sparse share is slots [counts[0], counts[0] + counts[1])
  const int32_t* counts;
  const int32_t* inexact;
  int nsplit;               // waves per particle: each takes a contiguous share of the window and writes one chunk row
  int64_t npad;
  uint32_t* part;           // [>= nsplit][2 ncls + 2][npad], like score_polar_su_kernel
};

template <bool USCALE>
__global__ __launch_bounds__(256) void score_polar_ray_kernel(RayArgs a) {
  extern __shared__ unsigned long long lacc[];   // [4 waves][ncls + 1][64 lanes]: a lane's sums per class (slot 0: no class)
  __shared__ uint32_t ldict[TDR_CMAP_MAX_DICT];
  if (int_form_off(a.inexact)) return;
  const int nsparse = a.counts[1];
  if ((int64_t)blockIdx.x * 4 >= (int64_t)nsparse * a.nsplit) return;   // whole workgroup idle (uniform)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int t = threadIdx.x; t < a.dict_n; t += 256) ldict[t] = a.dict_int[t];
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
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];  // state_particle.cpp:161
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];  // :162
  const float off0 = cy / a.resolution;  // top_down_map_polar.cpp:29
  const float off1 = cx / a.resolution;  // :30
  const int shift = rot_shift_dev(a.st[TDR_ST_THETA * a.cap + p], a.nb);
  const int P = a.nb * a.nr;
  const int rot = shift * a.nr;   // window sample k' meets scan bin (k' + rot) mod P of desc_ray
  const float rmaxf = (float)a.rows, cmaxf = (float)a.cols;
  const char* __restrict__ crecb = reinterpret_cast<const char*>(a.crec);
  const int mconst = (int)a.kmask_off + a.kmask_col + 128;                          // kmask_offset
  const int pconst = (int)a.planes_off + (a.pkcol + 16) + 128;                      // plane_offset, plane 0
  const float2* __restrict__ tab2 = reinterpret_cast<const float2*>(a.tab_ray);
  typedef float tdr_v2f __attribute__((ext_vector_type(2)));
  const tdr_v2f offv = {off0, off1};
  auto cell = [&](float2 t, int& ri, int& ci) {
    tdr_v2f pv = {t.x, t.y};
    if constexpr (!USCALE) pv = (pv * scale) * a.res;  // top_down_map_polar.cpp:28
    pv = pv + offv;                                     // :29-30
    tdr_v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
    qv = qv + 0.49999997f;                              // round_half_away_clamped
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ri) : "v"(qv.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ci) : "v"(qv.y));
  };
  auto plane_at = [&](int ri, int ci, uint32_t cls) -> unsigned {   // byte offset of the cell in class `cls`'s plane
    return plane_offset(ri, ci, a.pkcol, pconst) + ((cls * a.plane_units) << 7);
  };

  // this wave's share of the window: whole steps of 64 samples
  const int steps = (P + 63) >> 6, per = (steps + a.nsplit - 1) / a.nsplit;
  const int k0 = part_id * per * 64, k1 = min(P, (part_id + 1) * per * 64);
  uint32_t known = 0, norm = 0;
  for (int kb = k0; kb < k1; kb += 64 * RAY_U) {
    float2 t[RAY_U];
    uint32_t d[RAY_U];
#pragma unroll
    for (int u = 0; u < RAY_U; u++) {
      const int k = kb + 64 * u + lane;
      const bool valid = k < k1;
      const int kk = valid ? k : k1 - 1;
      t[u] = tab2[kk];
      int dk = kk + rot;
      dk -= dk >= P ? P : 0;
      d[u] = a.desc_ray[dk];
      if (!valid) d[u] = 0xFE000000u;   // no sample: counts nothing (code 0xFE)
    }
    uint32_t v[RAY_U];
    uint32_t sh[RAY_U];
#pragma unroll
    for (int u = 0; u < RAY_U; u++) {
      int ri, ci;
      cell(t[u], ri, ci);
      const uint32_t code = d[u] >> 24;
      const bool single = code - 1u < (uint32_t)a.ncls;
      // one 2-byte gather per sample: the cell of the bin's class plane (its bit 15 = known), else the half of the known
      // mask's word that holds the cell's bit
      const unsigned moff = kmask_offset(ri, ci, a.kmask_col, mconst) + ((unsigned)(ci >> 3) & 2u);
      const unsigned poff = plane_at(ri, ci, code - 1u);
      const unsigned off = single ? poff : moff;
      sh[u] = single ? 15u : ((uint32_t)ci & 15u);
      v[u] = *reinterpret_cast<const uint16_t*>(crecb + off);
    }
#pragma unroll
    for (int u = 0; u < RAY_U; u++) {
      const uint32_t code = d[u] >> 24, cnt = d[u] & 0xFFFFFFu;
      const bool single = code - 1u < (uint32_t)a.ncls;
      const uint32_t kbit = code == 0xFEu ? 0u : (v[u] >> sh[u]) & 1u;
      known += kbit;
      norm += cnt & (0u - kbit);                                   // state_particle.cpp:141-142
      if (single) {
        const uint32_t D = ldict[v[u] & 0x3FFu];
        my[code * 64] += (unsigned long long)cnt * D;              // :136-139, as integers
      }
    }
  }
  // the bins that hold several classes: their known bit and count went in above, here every class present meets its plane
  {
    const int nm = *a.n_multi, mper = (nm + a.nsplit - 1) / a.nsplit;
    const int e1 = min(nm, (part_id + 1) * mper);
    for (int e = part_id * mper + lane; e < e1; e += 64) {
      int k = (int)a.multi[2 * e] - rot;   // the window sample that meets this scan bin
      k += k < 0 ? P : 0;
      const int64_t bin = a.multi[2 * e + 1];
      int ri, ci;
      cell(tab2[k], ri, ci);
      for (int c = 0; c < a.ncls; c++) {
        const float s = a.scan_pk[bin * a.rf + c];
        if (s != 0.f) {
          const uint32_t vv = *reinterpret_cast<const uint16_t*>(crecb + plane_at(ri, ci, (uint32_t)c));
          my[(c + 1) * 64] += (unsigned long long)(uint32_t)s * ldict[vv & 0x3FFu];
        }
      }
    }
  }
  // lanes -> one sum per class
  uint32_t* o = a.part + (int64_t)part_id * (2 * a.ncls + 2) * a.npad + slot;
  for (int c = 0; c < a.ncls; c++) {
    unsigned long long s = my[(c + 1) * 64];
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

// ---- host side ---------------------------------------------------------------------------------------------------------
extern "C" size_t tdr_cmap_tile_words(int ncls, int rows, int cols);
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
int tdr_ray_splits(int nb, int nr, int64_t n) {
  if (g_ray_split > 0) return g_ray_split;
  // waves per particle: a window of P samples is P / 64 steps; small launches split it to fill the chip (the sums are
  // exact: any split gives the same bits)
  const int64_t steps = cdiv((int64_t)nb * nr, 64);
  int s = 1;
  while (s < TDR_RAY_MAX_SPLIT && n * s < 32768 && steps / (2 * s) >= 8) s *= 2;
  if (s == 1 && steps >= 512) s = 2;
  return s;
}

int tdr_ray_prepare(const SuLaunch& L, const SuWs& W, hipStream_t s) {
  int32_t* base = L.ws;
  int* ints = base + W.ints + 3 * (L.nb + 1);   // [counts 3][n_multi][inexact]
  const int64_t P = (int64_t)L.nb * L.nr;
  hipLaunchKernelGGL(ray_prep_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, s, L.tab, L.scan_pk, L.nb, L.nr, L.rf,
                     L.map->ncls, reinterpret_cast<const uint32_t*>(L.map->dict) + 2 * TDR_CMAP_MAX_DICT,
                     reinterpret_cast<float*>(base + W.ray_tab), reinterpret_cast<uint32_t*>(base + W.ray_desc),
                     reinterpret_cast<uint32_t*>(base + W.ray_multi), ints + 3, ints + 4);
  LAUNCH_CHECK("ray_prep");
  return TDR_OK;
}

int tdr_ray_score(const SuLaunch& L, const SuWs& W, hipStream_t s) {
  const tdr_map_desc* map = L.map;
  int32_t* base = L.ws;
  int* ints = base + W.ints + 3 * (L.nb + 1);
  RayArgs r;
  r.crec = map->crec;
  r.kmask_off = (unsigned)(tdr_cmap_tile_words(map->ncls, map->rows, map->cols) * 4);
  r.kmask_col = kmask_trows(map->rows) * 128;
  r.planes_off = (unsigned)(tdr_cmap_plane_offset_words(map->ncls, map->rows, map->cols) * 4);
  r.plane_units = (unsigned)(tdr_cmap_plane_words(map->ncls, map->rows, map->cols) * 4 / 128);
  r.pkcol = plane_trows(map->rows) * 128 - 16;
  r.dict_int = reinterpret_cast<const uint32_t*>(map->dict) + TDR_CMAP_MAX_DICT;
  r.dict_n = map->dict_n;
  r.rows = map->rows; r.cols = map->cols; r.resolution = map->resolution;
  r.tab_ray = reinterpret_cast<const float*>(base + W.ray_tab);
  r.desc_ray = reinterpret_cast<const uint32_t*>(base + W.ray_desc);
  r.multi = reinterpret_cast<const uint32_t*>(base + W.ray_multi);
  r.n_multi = ints + 3;
  r.scan_pk = L.scan_pk;
  r.rf = L.rf; r.ncls = map->ncls; r.nb = L.nb; r.nr = L.nr; r.res = L.res;
  r.st = L.st; r.cap = L.cap;
  r.slots = base + W.slots;
  r.counts = ints;
  r.inexact = ints + 4;
  r.nsplit = L.ray_split;
  r.npad = L.npad;
  r.part = reinterpret_cast<uint32_t*>(L.part);
  const dim3 grid((unsigned)cdiv(L.n * r.nsplit, 4)), block(256);
  const size_t lds = (size_t)4 * (map->ncls + 1) * 64 * sizeof(unsigned long long);
  if (L.uniform_scale) hipLaunchKernelGGL((score_polar_ray_kernel<true>), grid, block, lds, s, r);
  else hipLaunchKernelGGL((score_polar_ray_kernel<false>), grid, block, lds, s, r);
  LAUNCH_CHECK("score_polar_ray");
  return TDR_OK;
}
