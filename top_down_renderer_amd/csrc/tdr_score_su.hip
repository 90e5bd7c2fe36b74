// tdr_score_su.hip — the SHIFT-UNIFORM polar scoring kernel and its ordering passes (host interface: tdr_score_su.h).
//
// score_polar_kernel (tdr_score.hip) pairs window row i with scan row (i + shift) mod nb, shift = the lane's own heading
// bin (state_particle.cpp:124-128): the scan record is a per-lane operand (LDS), and every sample pays the full record
// decode and every class's FMA although a LiDAR scan is sparse — in the config-2 scan 44 % of the (theta, r) bins are
// empty in every class and a non-empty bin holds ~1.0 class (90 % of the class-bins are zero).
//
// Here the particles are processed in (shift, Morton) order with every shift bucket padded to whole waves, so the shift
// is WAVE-uniform: the scan side of a sample becomes a scalar operand (a two-dword descriptor read through the scalar
// cache — no LDS scan image) and "this bin is empty" / "this bin holds class c only" are wave-uniform branches:
//   empty bin          coordinates + record load + the `known` bit                  (14 vector instructions)
//   one class present  ... + one dictionary decode + 2 FMAs (class, normalisation)  (~19)
//   several classes    the packed scan record through scalar loads, one decode + FMA per class present
// against ~30 for every sample in score_polar_kernel.  Skipping an FMA whose scan operand is zero leaves the accumulator
// unchanged bit for bit (fma(0, m, acc) == acc for finite m, acc never -0), and the samples of a particle are visited in
// the same order — direction ascending, ring ascending within the group — with the same partition into per-group partial
// sums, so both kernels produce IDENTICAL partial sums (tests/test_gpu_parity.py::test_shift_uniform_*).  A dictionary
// holding a non-finite value (0 * inf = NaN must not be skipped) turns the skipping off (flags[0], set by su_prep_kernel).
//
// Per launch (all on the caller's stream, nothing synchronises):
//   su_key_kernel      heading bin of every particle (in the caller's locality order) + histogram of the bins
//   rocPRIM            stable radix sort by bin: (shift, Morton) order
//   su_offsets_kernel  bucket starts in the sorted list and in the padded slot list, number of slots in use
//   su_scatter_kernel  slot -> particle (-1 = padding)
//   su_prep_kernel     per (direction, ring): sample offset and scan descriptor, group-major (a wave streams them in order)
//   score_polar_su_kernel, then score_finalize_kernel over the slots
//
// Compiled with -mllvm -structurizecfg-skip-uniform-regions (build.py): the per-sample dispatch on the descriptor is a
// tree of wave-uniform branches; left to the structuriser each leaf is followed by copies of all accumulators (phi
// merges of its flow blocks: 128 v_mov_b64 in the loop), with uniform regions skipped the leaves are 3-4 instructions.
#include <rocprim/device/device_radix_sort.hpp>

#include "tdr_score_dev.h"
#include "tdr_score_su.h"

#define SU_CODE_FULL 0xFFFFFFFFu

struct SuArgs {
  const uint32_t* crec;    // compact records (narrow form)
  const float* dict;
  int dict_n, ctiles_c;
  int rows, cols;          // map
  float resolution;
  const float* tab_su;     // [nchunks][nb][group][2]: (tab*scale)*res (USCALE) or tab
  const uint32_t* desc;    // [nchunks][nb][group][2]: {code, value bits} of scan bin (row, ring)
  const float* scan_pk;    // [nr][nb][rf]: read for bins holding several classes
  const int* flags;        // [0] != 0: never skip (non-finite dictionary value)
  int nb, nr;
  float res;
  const float* st;
  int64_t cap;
  const int32_t* slots;    // padded (shift, Morton) order, -1 = padding; every 64-slot batch holds one shift
  const int32_t* nslots;   // device word: slots in use (a multiple of 64)
  int group, nchunks, ncls;
  int64_t npad;            // stride of `part`
  float* part;             // [nchunks][rf+1][npad]
};

// descriptor code: 0 = every class zero; c + 1 = class c alone is non-zero (value = its count = the bin's sum);
// SU_CODE_FULL = several classes (value = the bin's sum, slot rf-1 of the packed record)
__global__ __launch_bounds__(256) void su_prep_kernel(const float* __restrict__ tab, const float* __restrict__ scan_pk,
                                                      int nb, int nr, int rf, int ncls, int group, int nchunks,
                                                      const float* __restrict__ dict, int dict_n, float* __restrict__ tab_su,
                                                      uint32_t* __restrict__ desc, int* __restrict__ flags) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0) {
    bool bad = false;
    for (int k = threadIdx.x; k < dict_n; k += blockDim.x) bad |= !(fabsf(dict[k]) <= 3.402823466e+38f);
    if (bad) atomicOr(flags, 1);
  }
  const int64_t total = (int64_t)nchunks * nb * group;
  if (t >= total) return;
  const int jj = (int)(t % group);
  const int64_t q = t / group;
  const int i = (int)(q % nb), chunk = (int)(q / nb);
  const int j = chunk * group + jj;
  float tx = 0.f, ty = 0.f, val = 0.f;
  uint32_t code = 0;
  if (j < nr) {
    const int64_t k = (int64_t)j * nb + i;
    tx = tab[2 * k];
    ty = tab[2 * k + 1];
    const float* r = scan_pk + k * rf;
    int nz = 0, first = 0;
    for (int c = 0; c < ncls; c++)
      if (r[c] != 0.f) {
        if (!nz) first = c;
        nz++;
      }
    if (nz == 1) { code = (uint32_t)first + 1u; val = r[first]; }
    else if (nz > 1) { code = SU_CODE_FULL; val = r[rf - 1]; }
  }
  tab_su[2 * t] = tx;
  tab_su[2 * t + 1] = ty;
  desc[2 * t] = code;
  desc[2 * t + 1] = __float_as_uint(val);
}

__global__ __launch_bounds__(256) void su_key_kernel(const float* __restrict__ st, int64_t cap, int64_t n,
                                                     const int32_t* __restrict__ perm, int nb,
                                                     uint32_t* __restrict__ keys, int32_t* __restrict__ vals,
                                                     int* __restrict__ cnt) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int32_t p = perm ? perm[t] : (int32_t)t;
  const int s = rot_shift_dev(st[TDR_ST_THETA * cap + p], nb);
  keys[t] = (uint32_t)s;
  vals[t] = p;
  atomicAdd(&cnt[s], 1);
}

// one workgroup: exclusive sums of the bucket sizes (start in the sorted list) and of the sizes rounded up to whole waves
// (start in the slot list); nslots = slots in use
__global__ __launch_bounds__(256) void su_offsets_kernel(const int* __restrict__ cnt, int nb, int* __restrict__ start,
                                                         int* __restrict__ slot_start, int* __restrict__ nslots) {
  __shared__ int sa[256], sb[256];
  int carry_a = 0, carry_b = 0;
  for (int base = 0; base < nb; base += 256) {
    const int k = base + threadIdx.x;
    const int c = k < nb ? cnt[k] : 0;
    const int cp = (c + 63) & ~63;
    sa[threadIdx.x] = c;
    sb[threadIdx.x] = cp;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {   // Hillis-Steele inclusive scan
      const int va = threadIdx.x >= d ? sa[threadIdx.x - d] : 0;
      const int vb = threadIdx.x >= d ? sb[threadIdx.x - d] : 0;
      __syncthreads();
      sa[threadIdx.x] += va;
      sb[threadIdx.x] += vb;
      __syncthreads();
    }
    if (k < nb) {
      start[k] = carry_a + sa[threadIdx.x] - c;
      slot_start[k] = carry_b + sb[threadIdx.x] - cp;
    }
    carry_a += sa[255];
    carry_b += sb[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) *nslots = carry_b;
}

__global__ __launch_bounds__(256) void su_scatter_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                                         int64_t n, const int* __restrict__ start,
                                                         const int* __restrict__ slot_start, int32_t* __restrict__ slots) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint32_t s = keys[t];
  slots[slot_start[s] + ((int)t - start[s])] = vals[t];
}

// lane = particle; every wave holds particles of ONE heading bin (see the file comment).  grid.y = group of a.group
// consecutive range rings (score_group_rings: a multiple of 4, nr a multiple of 4), samples visited ray-major like
// score_polar_kernel: direction i ascending, the group's rings in steps of 4 consecutive cells along the ray.
template <int NV4, bool KSLOT, bool USCALE>
__global__ __launch_bounds__(256) void score_polar_su_kernel(SuArgs a) {
  constexpr int RF = 4 * NV4;
  constexpr int ND = CmapShape<RF, KSLOT>::ND, CW = CmapShape<RF, KSLOT>::CW, LC = CmapShape<RF, KSLOT>::LC;
  __shared__ float ldict[TDR_CMAP_MAX_DICT];
  for (int t = threadIdx.x; t < a.dict_n; t += 256) ldict[t] = a.dict[t];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64;
  if (base >= (int64_t)*a.nslots) return;   // wave-uniform; no barrier follows
  const int32_t sp = a.slots[base + lane];
  const int32_t p0 = __builtin_amdgcn_readfirstlane(sp);   // a batch's first slot is never padding
  const int64_t p = sp >= 0 ? sp : p0;
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];  // state_particle.cpp:161
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];  // :162
  const float off0 = cy / a.resolution;  // top_down_map_polar.cpp:29
  const float off1 = cx / a.resolution;  // :30
  const int shift = __builtin_amdgcn_readfirstlane(rot_shift_dev(a.st[TDR_ST_THETA * a.cap + p], a.nb));
  const int nb = a.nb, G = a.group;
  const int j0 = blockIdx.y * G, gn = min(a.nr - j0, G);
  const float rmaxf = (float)a.rows, cmaxf = (float)a.cols;
  const char* __restrict__ crecb = reinterpret_cast<const char*>(a.crec);
  const int ckconst = (a.ctiles_c + 1) * 128;
  typedef const float __attribute__((address_space(4))) * tdr_const_f;
  typedef const uint32_t __attribute__((address_space(4))) * tdr_const_u;
  const tdr_const_f tbase = (tdr_const_f)a.tab_su + (int64_t)blockIdx.y * nb * G * 2;
  const tdr_const_u dbase = (tdr_const_u)a.desc + (int64_t)blockIdx.y * nb * G * 2;
  const tdr_const_f scanc = (tdr_const_f)a.scan_pk;
  const bool noskip = *(const int __attribute__((address_space(4)))*)a.flags != 0;

  typedef float tdr_v2f __attribute__((ext_vector_type(2)));
  const tdr_v2f offv = {off0, off1};
  auto cell_offset = [&](float tx, float ty) -> unsigned {
    tdr_v2f pv = {tx, ty};
    if constexpr (!USCALE) pv = (pv * scale) * a.res;  // top_down_map_polar.cpp:28
    pv = pv + offv;                                     // :29-30
    tdr_v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
    qv = qv + 0.49999997f;                              // round_half_away_clamped
    int ri, ci;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ri) : "v"(qv.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ci) : "v"(qv.y));
    return cmap_offset<CW, LC>(ri, ci, a.ctiles_c, ckconst);
  };
  // distance k of a compact record (cmap_decode, one field)
  auto field = [&](const uint32_t (&w)[CW], int k) -> float {
    const uint32_t ww = w[k / 3];
    const int sh = 10 * (k % 3);
    const uint32_t boff = sh ? ((ww >> sh) & 0xFFCu) : (ww & 0xFFCu);
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ldict) + boff);
  };

  float acc[ND];
#pragma unroll
  for (int k = 0; k < ND; k++) acc[k] = 0.f;
  float norm = 0.f;
  uint32_t known = 0;

  for (int i = 0; i < nb; i++) {
    int r = i + shift;
    r -= r >= nb ? nb : 0;
    const tdr_const_f T = tbase + (int64_t)i * G * 2;
    const tdr_const_u D = dbase + (int64_t)r * G * 2;
    for (int jj = 0; jj < gn; jj += 4) {
      float tx[4], ty[4], val[4];
      uint32_t code[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        tx[u] = T[2 * (jj + u)];
        ty[u] = T[2 * (jj + u) + 1];
        code[u] = D[2 * (jj + u)];
        val[u] = __uint_as_float(D[2 * (jj + u) + 1]);
      }
      uint32_t w[4][CW];
#pragma unroll
      for (int u = 0; u < 4; u++) cmap_load<CW>(crecb, cell_offset(tx[u], ty[u]), w[u]);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t kb = w[u][CW - 1] & 1u;
        known += kb;
        const uint32_t cd = noskip ? SU_CODE_FULL : code[u];
        if (cd != 0) {   // wave-uniform
          norm = __builtin_fmaf(val[u], (float)kb, norm);   // the bin's sum x known (state_particle.cpp:141-142)
          if (cd != SU_CODE_FULL) {
            switch (cd) {   // wave-uniform
#define SU_CASE(K)                                                            \
  case K + 1:                                                                 \
    if constexpr (K < ND) acc[K < ND ? K : 0] = __builtin_fmaf(val[u], field(w[u], K < ND ? K : 0), acc[K < ND ? K : 0]); \
    break;
              SU_CASE(0) SU_CASE(1) SU_CASE(2) SU_CASE(3) SU_CASE(4) SU_CASE(5)
              SU_CASE(6) SU_CASE(7) SU_CASE(8) SU_CASE(9) SU_CASE(10)
#undef SU_CASE
              default: break;
            }
          } else {
            const tdr_const_f S = scanc + ((int64_t)(j0 + jj + u) * nb + r) * RF;
#pragma unroll
            for (int k = 0; k < ND; k++) {
              const float sk = S[k];
              if (noskip || sk != 0.f) acc[k] = __builtin_fmaf(sk, field(w[u], k), acc[k]);
            }
          }
        }
      }
    }
  }
  const int64_t slot = base + lane;
  float* o = a.part + (int64_t)blockIdx.y * (RF + 1) * a.npad + slot;
#pragma unroll
  for (int k = 0; k < ND; k++)
    if (k < a.ncls) o[(int64_t)k * a.npad] = acc[k];
  o[(int64_t)(RF - 1) * a.npad] = norm;
  o[(int64_t)RF * a.npad] = (float)known;
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// 0 = never, 1 = when it pays (default), 2 = whenever the shapes allow (tests: small filters, heavy padding)
static int g_su_mode = [] {
  const char* e = getenv("TDR_SHIFT_UNIFORM");
  return e ? atoi(e) : 1;
}();
extern "C" int tdr_config_shift_uniform(int mode) {   // < 0: query only
  if (mode >= 0) g_su_mode = mode > 2 ? 2 : mode;
  return g_su_mode;
}
static int64_t g_su_launches = 0;
extern "C" int64_t tdr_shift_uniform_launches(void) { return g_su_launches; }
// Padding costs up to 63 idle lanes per heading bin: the order pays once a bin holds a few waves on average.
bool tdr_su_shape_ok(int nb, int nr, int group, int64_t n_total) {
  if (g_su_mode == 0) return false;
  if (group % 4 != 0 || nr % 4 != 0 || nb > 4096) return false;
  if (g_su_mode == 2) return true;
  return n_total >= (int64_t)192 * nb;
}
static size_t su_sort_tmp_bytes(int64_t n) {
  size_t bytes = 0;
  uint32_t* k = nullptr;
  int32_t* v = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)std::max<int64_t>(n, 1), 0u, 12u,
                                           (hipStream_t)0, false);
  if (e != hipSuccess || bytes == 0) bytes = (size_t)(n + 4096) * 16;   // no device to ask: a generous bound
  return bytes;
}
SuWs tdr_su_ws(int nb, int nr, int group, int64_t n) {
  SuWs w;
  const int64_t nchunks = cdiv(nr, group), desc_words = nchunks * nb * group * 2;
  int64_t o = 0;
  auto take = [&](int64_t words) { const int64_t at = o; o += (words + 63) / 64 * 64; return at; };   // 256-byte aligned
  w.tab_su = take(desc_words);
  w.desc = take(desc_words);
  w.keys_in = take(n);
  w.keys_out = take(n);
  w.vals_in = take(n);
  w.vals_out = take(n);
  w.ints = take(3 * (int64_t)nb + 64);
  w.slots = take(su_npad(n, nb));
  w.sort_tmp = take((int64_t)((su_sort_tmp_bytes(n) + 3) / 4));
  w.total = o;
  return w;
}

int tdr_su_prepare(const SuLaunch& L, const SuWs& W, hipStream_t s, const int32_t** slots_out, const int32_t** nslots_out) {
  int32_t* base = L.ws;
  float* tab_su = reinterpret_cast<float*>(base + W.tab_su);
  uint32_t* desc = reinterpret_cast<uint32_t*>(base + W.desc);
  uint32_t* keys_in = reinterpret_cast<uint32_t*>(base + W.keys_in);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(base + W.keys_out);
  int32_t* vals_in = base + W.vals_in;
  int32_t* vals_out = base + W.vals_out;
  int* cnt = base + W.ints;
  int* start = cnt + L.nb;
  int* slot_start = start + L.nb;
  int* nslots = slot_start + L.nb;
  int* flags = nslots + 1;
  int32_t* slots = base + W.slots;
  const int64_t n = L.n;
  HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)(3 * L.nb + 2), s));
  HIP_TRY(hipMemsetAsync(slots, 0xFF, sizeof(int32_t) * (size_t)L.npad, s));
  hipLaunchKernelGGL(su_key_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, L.st, L.cap, n, L.perm, L.nb, keys_in,
                     vals_in, cnt);
  LAUNCH_CHECK("su_key");
  unsigned bits = 1;
  while ((1u << bits) < (unsigned)L.nb) bits++;
  size_t tmp_bytes = su_sort_tmp_bytes(n);
  HIP_TRY(rocprim::radix_sort_pairs(base + W.sort_tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, bits,
                                    s, false));
  hipLaunchKernelGGL(su_offsets_kernel, dim3(1), dim3(256), 0, s, (const int*)cnt, L.nb, start, slot_start, nslots);
  LAUNCH_CHECK("su_offsets");
  hipLaunchKernelGGL(su_scatter_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, (const uint32_t*)keys_out,
                     (const int32_t*)vals_out, n, (const int*)start, (const int*)slot_start, slots);
  LAUNCH_CHECK("su_scatter");
  const int64_t ndesc = (int64_t)L.nchunks * L.nb * L.group;
  hipLaunchKernelGGL(su_prep_kernel, dim3((unsigned)cdiv(ndesc, 256)), dim3(256), 0, s, L.tab, L.scan_pk, L.nb, L.nr, L.rf,
                     L.map->ncls, L.group, L.nchunks, L.map->dict, L.map->dict_n, tab_su, desc, flags);
  LAUNCH_CHECK("su_prep");
  *slots_out = slots;
  *nslots_out = nslots;
  return TDR_OK;
}

int tdr_su_score(const SuLaunch& L, const SuWs& W, hipStream_t s) {
  const tdr_map_desc* map = L.map;
  int32_t* base = L.ws;
  int* nslots = base + W.ints + 3 * L.nb;
  SuArgs u;
  const int lc = map->cwords == 1 ? 3 : (map->cwords == 2 ? 2 : 1);
  u.crec = map->crec; u.dict = map->dict; u.dict_n = map->dict_n; u.ctiles_c = (map->cols >> lc) + 2;
  u.rows = map->rows; u.cols = map->cols; u.resolution = map->resolution;
  u.tab_su = reinterpret_cast<const float*>(base + W.tab_su);
  u.desc = reinterpret_cast<const uint32_t*>(base + W.desc);
  u.scan_pk = L.scan_pk; u.flags = nslots + 1;
  u.nb = L.nb; u.nr = L.nr; u.res = L.res; u.st = L.st; u.cap = L.cap;
  u.slots = base + W.slots; u.nslots = nslots;
  u.group = L.group; u.nchunks = L.nchunks; u.ncls = map->ncls; u.npad = L.npad; u.part = L.part;
  const dim3 grid((unsigned)cdiv(L.npad, 256), (unsigned)L.nchunks), block(256);
  const bool ks = tdr_has_kslot(map->ncls, L.rf), us = L.uniform_scale;
#define TDR_LAUNCH_SU(NV4)                                                                         \
  if (ks && us) hipLaunchKernelGGL((score_polar_su_kernel<NV4, true, true>), grid, block, 0, s, u);  \
  else if (ks) hipLaunchKernelGGL((score_polar_su_kernel<NV4, true, false>), grid, block, 0, s, u);  \
  else if (us) hipLaunchKernelGGL((score_polar_su_kernel<NV4, false, true>), grid, block, 0, s, u);  \
  else hipLaunchKernelGGL((score_polar_su_kernel<NV4, false, false>), grid, block, 0, s, u);
  switch (L.rf / 4) {
    case 1: TDR_LAUNCH_SU(1) break;
    case 2: TDR_LAUNCH_SU(2) break;
    case 3: TDR_LAUNCH_SU(3) break;
    default: return fail(TDR_ERR_ARG, "score: no shift-uniform kernel for record size %d", L.rf);
  }
#undef TDR_LAUNCH_SU
  LAUNCH_CHECK("score_polar_su");
  g_su_launches++;
  return TDR_OK;
}
