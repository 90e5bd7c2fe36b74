// tdr_filter.hip — propagate, weight statistics, resample / gather, pose statistics, state layout helpers, locality order.
#include <random>

#include <rocprim/device/device_radix_sort.hpp>
// rocPRIM sorts up to 2^20 pairs by merging sorted blocks; 2048-item blocks instead of its 1024 save a merge launch
// (20 000 keys: 40 -> 30 us; 8192-item blocks: 35 us, its radix sort proper — merge limit 0 — 90 us)
using TdrSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<512, 512, 4>, rocprim::default_config>;

#include "tdr_common.h"
#include "tdr_sincosf.h"

// ------------------------------------------------------------------------------------------------------------------
// K3: propagate (state_particle.cpp:57-78).  z*sigma+mu spelled without contraction like libstdc++'s
// normal_distribution (`__ret * stddev + mean`).
__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }
__device__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = mulhi32(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    uint32_t hi1 = mulhi32(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ void propagate_kernel(float* __restrict__ st, int64_t cap, int64_t n, float* __restrict__ last_dist,
                                 float tx, float ty, float omega, int scale_freeze, float pos_cov, float theta_cov,
                                 const float* __restrict__ z4, uint64_t seed, uint64_t step, int64_t index_base,
                                 int libm_fma) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float z[4];
  if (z4) {
    z[0] = z4[4 * p]; z[1] = z4[4 * p + 1]; z[2] = z4[4 * p + 2]; z[3] = z4[4 * p + 3];
  } else {
    uint64_t gi = (uint64_t)(index_base + p);
    uint32_t c[4] = {(uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    // Box-Muller on (0,1] uniforms
    float u0 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f), u1 = (float)(c[1] >> 8) * (1.0f / 16777216.0f);
    float u2 = ((float)(c[2] >> 8) + 1.0f) * (1.0f / 16777216.0f), u3 = (float)(c[3] >> 8) * (1.0f / 16777216.0f);
    float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    z[0] = r0 * cosf(6.283185307f * u1); z[1] = r0 * sinf(6.283185307f * u1);
    z[2] = r1 * cosf(6.283185307f * u3); z[3] = r1 * sinf(6.283185307f * u3);
  }
  float theta = st[TDR_ST_THETA * cap + p];
  float dx = st[TDR_ST_DX * cap + p], dy = st[TDR_ST_DY * cap + p];
  // Rotation2D<float>(theta) * trans (:58): std::cos / std::sin of a float = the host libm's cosf / sinf, restated
  // bit for bit (tdr_sincosf.h)
  const float c = tdr_libm::cosf_v(theta, libm_fma), s = tdr_libm::sinf_v(theta, libm_fma);
  const float gx = c * tx + (-s) * ty;
  const float gy = s * tx + c * ty;
  const float lx = dx, ly = dy;
  dx += gx;
  dy += gy;
  const float dist = sqrtf(gx * gx + gy * gy);
  const float sd_pos = pos_cov * dist, sd_th = theta_cov * dist;
  theta += (z[0] * sd_th + 0.f) + omega;
  dx += z[1] * sd_pos + 0.f;
  dy += z[2] * sd_pos + 0.f;
  if (!scale_freeze) {
    const float sd_s = (float)fmin(2. / (double)dist, 0.02);
    float scale = st[TDR_ST_SCALE * cap + p];
    scale *= z[3] * sd_s + 1.f;
    st[TDR_ST_SCALE * cap + p] = scale;
  }
  st[TDR_ST_THETA * cap + p] = theta;
  st[TDR_ST_DX * cap + p] = dx;
  st[TDR_ST_DY * cap + p] = dy;
  const float mx = lx - dx, my = ly - dy;
  last_dist[p] = sqrtf(mx * mx + my * my);
}

extern "C" int tdr_k_propagate(float* st, int64_t cap, int64_t n, float* last_dist, float tx, float ty, float omega,
                               int scale_freeze, float pos_cov, float theta_cov, const float* z4, uint64_t seed,
                               uint64_t step, int64_t index_base, void* stream) {
  if (!st || !last_dist) return fail(TDR_ERR_ARG, "propagate: null pointer");
  if (n < 0 || cap < n) return fail(TDR_ERR_ARG, "propagate: n exceeds capacity");
  if (n == 0) return TDR_OK;
  hipLaunchKernelGGL(propagate_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, st, cap, n,
                     last_dist, tx, ty, omega, scale_freeze, pos_cov, theta_cov, z4, seed, step, index_base,
                     tdr_libm_fma());
  LAUNCH_CHECK("propagate");
  return TDR_OK;
}

// host RNG: the reference's shared std::mt19937 + libstdc++ distributions (particle_filter.h:52, state_particle.cpp:64-73)
extern "C" void* tdr_rng_create(uint32_t seed) { return new std::mt19937(seed); }
extern "C" void tdr_rng_destroy(void* rng) { delete (std::mt19937*)rng; }
extern "C" float tdr_rng_uniform_host(void* rng) {
  std::uniform_real_distribution<float> d(0., 1.);
  return d(*(std::mt19937*)rng);
}
extern "C" int tdr_propagate_normals_host(void* rng, int64_t n, int scale_freeze, float* z4) {
  if (!rng || !z4 || n < 0) return fail(TDR_ERR_ARG, "propagate_normals: bad arguments");
  std::mt19937& gen = *(std::mt19937*)rng;
  for (int64_t p = 0; p < n; p++) {
    // fresh distribution objects per call and per use, like state_particle.cpp:64-65,72
    std::normal_distribution<float> disp{0, 1}, th{0, 1};
    z4[4 * p + 0] = th(gen);
    z4[4 * p + 1] = disp(gen);
    z4[4 * p + 2] = disp(gen);
    if (!scale_freeze) {
      std::normal_distribution<float> sc{0, 1};
      z4[4 * p + 3] = sc(gen);
    } else {
      z4[4 * p + 3] = 0.f;
    }
  }
  return TDR_OK;
}

// Host: particle initialisation.  Serial draws from the shared mt19937 with data-dependent rejection, exactly the
// consumption order of StateParticle::StateParticle (state_particle.cpp:3-49) inside
// ParticleFilter::initializeParticles (particle_filter.cpp:57-71) — including the draws the reference burns on the
// prototype particle and on its second buffer.  class_maps: HOST copy, the reference's column-major layout.
static bool on_road_host(const float* maps, int ncls, int rows, int cols, float resolution, int px, int py) {
  // TopDownMap::getClassesAtPoint (top_down_map.cpp:159-170) tested for class 1 (state_particle.cpp:29)
  const int c0 = (int)((float)px / resolution), c1 = (int)((float)py / resolution);
  if (ncls < 2) return false;
  if (!(c0 < cols && c1 < rows && c0 >= 0 && c1 >= 0)) return false;
  return maps[(size_t)1 * rows * cols + c1 + (size_t)rows * c0] < 1;
}
static tdr_state draw_particle(std::mt19937& gen, const float* maps, int ncls, int rows, int cols, float resolution,
                               const tdr_filter_params* fp) {
  std::uniform_real_distribution<float> uniform_dist(0., 1.);
  std::normal_distribution<float> normal_dist(0., 1.);
  tdr_state st;
  std::memset(&st, 0, sizeof(st));
  const float map_w = (float)cols * resolution, map_h = (float)rows * resolution;
  if (fp->fixed_scale < 0) st.scale = (float)std::pow(10, ((double)uniform_dist(gen) - 0.5) * 2);  // :15
  else st.scale = fp->fixed_scale;                                                                // :17
  while (true) {
    if (fp->init_pos_px_x > 0) {  // :21-23
      st.init_x_px = std::min(std::max(normal_dist(gen) * fp->init_pos_px_cov + fp->init_pos_px_x, 0.f), map_w);
      st.init_y_px = std::min(std::max(normal_dist(gen) * fp->init_pos_px_cov + fp->init_pos_px_y, 0.f), map_h);
    } else {                      // :25-26
      st.init_x_px = uniform_dist(gen) * map_w;
      st.init_y_px = uniform_dist(gen) * map_h;
    }
    if (on_road_host(maps, ncls, rows, cols, resolution, (int)st.init_x_px, (int)st.init_y_px)) break;  // :28-31
  }
  if (fp->init_pos_deg_theta != std::numeric_limits<float>::infinity()) {
    st.theta = normal_dist(gen) * fp->init_pos_deg_cov + fp->init_pos_deg_theta;  // :35
    st.theta = (float)((double)st.theta * (M_PI / 180));                          // :37
    st.have_init = 1;
  } else {
    st.theta = 0;
    st.have_init = 0;
  }
  return st;
}
// StateParticle's constructor with init == true (src/state_particle.cpp:3-49): one particle from the shared generator.
extern "C" int tdr_init_particle_host(void* rng, const float* class_maps, int ncls, int rows, int cols, float resolution,
                                      const tdr_filter_params* fp, tdr_state* out) {
  if (!rng || !class_maps || !fp || !out) return fail(TDR_ERR_ARG, "init_particle: bad arguments");
  if (ncls < 2) return fail(TDR_ERR_ARG, "init_particle: class 1 (road) is required for the on-road test");
  bool any_road = false;
  for (size_t k = 0; k < (size_t)rows * cols && !any_road; k++) any_road = class_maps[(size_t)rows * cols + k] < 1;
  if (!any_road) return fail(TDR_ERR_ARG, "init_particle: the map has no road cell, rejection sampling cannot end");
  *out = draw_particle(*(std::mt19937*)rng, class_maps, ncls, rows, cols, resolution, fp);
  return TDR_OK;
}
extern "C" int tdr_init_particles_host(void* rng, const float* class_maps, int ncls, int rows, int cols,
                                       float resolution, const tdr_filter_params* fp, int max_num, tdr_state* out,
                                       int64_t* n_out) {
  if (!rng || !class_maps || !fp || !out || !n_out || max_num < 0) return fail(TDR_ERR_ARG, "init_particles: bad arguments");
  if (ncls < 2) return fail(TDR_ERR_ARG, "init_particles: class 1 (road) is required for the on-road test");
  std::mt19937& gen = *(std::mt19937*)rng;
  bool any_road = false;
  for (size_t k = 0; k < (size_t)rows * cols && !any_road; k++) any_road = class_maps[(size_t)rows * cols + k] < 1;
  if (!any_road) return fail(TDR_ERR_ARG, "init_particles: the map has no road cell, rejection sampling cannot end");
  const size_t num_at_scale = (fp->fixed_scale < 0) ? 10 : 1;  // particle_filter.cpp:20-25
  int64_t count = 0;
  for (int i = 0; i < (int)((size_t)max_num / num_at_scale); i++) {                          // :57
    const tdr_state proto = draw_particle(gen, class_maps, ncls, rows, cols, resolution, fp);  // :58
    for (float scale = 0; scale < 1; scale += 1. / num_at_scale) {                            // :59
      tdr_state part = draw_particle(gen, class_maps, ncls, rows, cols, resolution, fp);       // :60
      if (fp->fixed_scale < 0) {
        part = proto;                                                                        // :62
        part.scale = (float)std::pow(10., (double)scale);                                    // :63
      }
      if (count < max_num + 16) out[count] = part;
      count++;
      (void)draw_particle(gen, class_maps, ncls, rows, cols, resolution, fp);                 // :68 second buffer
    }
  }
  *n_out = count;
  return TDR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// K4: weight statistics (particle_filter.cpp:107-147).  Sums that the reference leaves to Eigen (unspecified order) are
// taken in double with a fixed order, the reference's two serial float chains (`sum`, `bottom_stddev`) exactly, so the
// result depends only on (raw_w, last_dist, n) — identical on every rank that holds the all-gathered weights.  Up to
// 32768 particles everything is one launch of one workgroup (uw_small_kernel, tdr_prefix.hip); above, the passes below.
__device__ double block_sum_d(double v, double* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0;
  const int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; w++) t += sh[w];
  return t;
}
__device__ long long block_sum_ll(long long v, long long* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  long long t = 0;
  const int nw = blockDim.x >> 6;
  for (int w = 0; w < nw; w++) t += sh[w];
  return t;
}

// Multi-workgroup form for large n (> 32768): grid-wide passes, each workgroup reducing its own contiguous chunk in a fixed order
// and every workgroup re-reducing the G per-workgroup partials in index order, so the result is again a pure function
// of (raw_w, last_dist, n) — identical on every rank — without grid barriers.  Here the reference's two SERIAL float
// chains (`sum`, `bottom_stddev`, :108-126) are reproduced exactly (tdr_chain_total): at these sizes their own rounding,
// ~sqrt(n) 2^-24, is larger than the 1e-5 weight tolerance, and it ends up in the value written into NaN particles.
// Scratch lives behind the 8 info floats (TDR_UW_INFO_FLOATS in total).
#define UW_G 256
struct UwScratch {
  double a[UW_G];
  double b[UW_G];
};
__device__ __forceinline__ void uw_chunk(int64_t n, int64_t& lo, int64_t& hi) {
  const int64_t per = (n + UW_G - 1) / UW_G;
  lo = (int64_t)blockIdx.x * per;
  hi = lo + per < n ? lo + per : n;
  if (lo > n) lo = n;
}
// Sum of the UW_G per-workgroup partials in index order (same order in every workgroup -> same value everywhere).
// Staged through LDS so the dependent additions do not each wait on a global load.
__device__ double uw_total(const double* part) {
  __shared__ double stage[UW_G];
  __shared__ double result;
  __syncthreads();
  for (int g = threadIdx.x; g < UW_G; g += blockDim.x) stage[g] = part[g];
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int g = 0; g < UW_G; g++) t += stage[g];
    result = t;
  }
  __syncthreads();
  return result;
}
// pass 1: sum / count of the valid raw weights (:108-116)
__global__ __launch_bounds__(256) void uw_pass1(const float* __restrict__ raw, int64_t n, UwScratch* s1) {
  __shared__ double shd[4];
  __shared__ long long shl[4];
  int64_t lo, hi;
  uw_chunk(n, lo, hi);
  double s = 0;
  long long c = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = raw[i];
    if (!isnan(v)) { s += (double)v; c++; }
  }
  const double ts = block_sum_d(s, shd);
  const long long tc = block_sum_ll(c, shl);
  if (threadIdx.x == 0) { s1->a[blockIdx.x] = ts; s1->b[blockIdx.x] = (double)tc; }
}
// The serial float chains of :108-126 (`sum`, `bottom_stddev`) evaluated exactly by tdr_chain_total land here.
struct UwExact {
  float sum, mean, bsum, pad;
};
// pass 2: the mean (:117) from the exact `sum` chain; count of the weights below it (:118-125)
__global__ __launch_bounds__(256) void uw_pass2(const float* __restrict__ raw, int64_t n, const UwScratch* s1,
                                                UwScratch* s2, UwExact* ex) {
  __shared__ long long shl[4];
  const float mean = ex->sum / (float)(long long)uw_total(s1->b);
  int64_t lo, hi;
  uw_chunk(n, lo, hi);
  long long cu = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = raw[i];
    if (!isnan(v) && v < mean) cu++;
  }
  const long long tc = block_sum_ll(cu, shl);
  if (threadIdx.x == 0) {
    s2->b[blockIdx.x] = (double)tc;
    if (blockIdx.x == 0) ex->mean = mean;   // every workgroup computed the same value
  }
}
// pass 3: NaN fill / all-ones fallback (:129-134) and the first normalisation sum
__global__ __launch_bounds__(256) void uw_pass3(const float* __restrict__ raw, int64_t n, const UwExact* ex,
                                                const UwScratch* s2, UwScratch* s3, float* __restrict__ w) {
  __shared__ double shd[4];
  const float sum = ex->sum, mean = ex->mean;
  const long long num_under = (long long)uw_total(s2->b);
  const float bottom = sqrtf(ex->bsum / (float)num_under);   // :126
  const bool fallback = (sum == 0.f || num_under < 1);
  const float fill = mean - bottom;
  int64_t lo, hi;
  uw_chunk(n, lo, hi);
  double acc = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    float v = raw[i];
    v = fallback ? 1.f : (isnan(v) ? fill : v);
    w[i] = v;
    acc += (double)v;
  }
  const double t = block_sum_d(acc, shd);
  if (threadIdx.x == 0) s3->a[blockIdx.x] = t;
}
// pass 4: normalise (:135), motion regularisation (:138-141), second normalisation sum
__global__ __launch_bounds__(256) void uw_pass4(const float* __restrict__ last_dist, int64_t n, const UwScratch* s3,
                                                UwScratch* s4, float* __restrict__ w) {
  __shared__ double shd[4];
  const float fs1 = (float)uw_total(s3->a);
  const float fn = (float)n;
  int64_t lo, hi;
  uw_chunk(n, lo, hi);
  double acc = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    float v = w[i] / fs1;
    const float d = fminf(last_dist[i] * 5.f, 1.f);
    v = d * v + (1.f - d) / fn;
    w[i] = v;
    acc += (double)v;
  }
  const double t = block_sum_d(acc, shd);
  if (threadIdx.x == 0) s4->a[blockIdx.x] = t;
}
// pass 5: final normalisation (:142) and per-workgroup first maximum (:145-147)
__global__ __launch_bounds__(256) void uw_pass5(int64_t n, const UwScratch* s4, UwScratch* s5, float* __restrict__ w) {
  __shared__ float sb[4];
  __shared__ long long si[4];
  const float fs2 = (float)uw_total(s4->a);
  int64_t lo, hi;
  uw_chunk(n, lo, hi);
  float best = -INFINITY;
  long long besti = 0x7fffffffffffffffll;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = w[i] / fs2;
    w[i] = v;
    if (v > best || (v == best && i < besti)) { best = v; besti = i; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_down(best, o, 64);
    const long long oi = __shfl_down(besti, o, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = besti; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; k++)
      if (sb[k] > best || (sb[k] == best && si[k] < besti)) { best = sb[k]; besti = si[k]; }
    s5->a[blockIdx.x] = (double)best;
    s5->b[blockIdx.x] = (double)besti;  // exact: indices < 2^53
  }
}
__global__ __launch_bounds__(256) void uw_pass6(int64_t n, const UwScratch* s1, const UwScratch* s2,
                                                const UwScratch* s5, const UwExact* ex, float* info) {
  static_assert(UW_G == 256, "one candidate per thread");
  __shared__ float sb[4];
  __shared__ long long si[4];
  // first maximum over the workgroups' candidates (:145-147): largest value, smallest index among equals; a NaN never
  // wins (one thread per candidate and a tree instead of a serial loop over LDS: 23 -> 5 us)
  float best = (float)s5->a[threadIdx.x];
  long long besti = (long long)s5->b[threadIdx.x];
  if (best != best) { best = -INFINITY; besti = 0x7fffffffffffffffll; }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_down(best, o, 64);
    const long long oi = __shfl_down(besti, o, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sb[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = besti; }
  const float sum = ex->sum, mean = ex->mean;
  const long long nv = (long long)uw_total(s1->b), nu = (long long)uw_total(s2->b);   // (barriers inside)
  const float bottom = sqrtf(ex->bsum / (float)nu);
  if (threadIdx.x != 0) return;
  for (int k = 1; k < 4; k++)
    if (sb[k] > best || (sb[k] == best && si[k] < besti)) { best = sb[k]; besti = si[k]; }
  if (besti == 0x7fffffffffffffffll) besti = 0;
  info[0] = __int_as_float((int)besti);
  info[1] = sum; info[2] = mean; info[3] = bottom; info[4] = (sum == 0.f || nu < 1) ? 1.f : 0.f;
  info[5] = (float)nv; info[6] = (float)nu; info[7] = 0.f;
}

#define TDR_UW_SINGLE_MAX_N 32768
extern "C" int tdr_k_update_weights(const float* raw_w, const float* last_dist, int64_t n, float* w_out,
                                    float* info_out, void* stream) {
  if (!raw_w || !last_dist || !w_out || !info_out) return fail(TDR_ERR_ARG, "update_weights: null pointer");
  if (n < 1) return fail(TDR_ERR_ARG, "update_weights: n must be >= 1");
  hipStream_t s = (hipStream_t)stream;
  if (n <= TDR_UW_SINGLE_MAX_N) {
    const int rc = tdr_uw_small(raw_w, last_dist, n, w_out, info_out, s);
    if (rc) return rc;
    LAUNCH_CHECK("update_weights");
    return TDR_OK;
  }
  // scratch behind the 8 info floats: 5 x UwScratch, UwExact, then the chunk headers of the two exact chains
  constexpr size_t kFixed = 8 * sizeof(float) + 64 + 5 * sizeof(UwScratch) + sizeof(UwExact) + 64;
  static_assert(kFixed + 32 * 1024 <= TDR_UW_INFO_FLOATS * sizeof(float), "info scratch");
  UwScratch* sc = reinterpret_cast<UwScratch*>(
      (reinterpret_cast<uintptr_t>(info_out + 8) + 63) & ~(uintptr_t)63);
  UwExact* ex = reinterpret_cast<UwExact*>(sc + 5);
  void* chain_ws = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(ex + 1) + 63) & ~(uintptr_t)63);
  const size_t chain_room = TDR_UW_INFO_FLOATS * sizeof(float) - kFixed;
  if ((size_t)tdr_prefix_workspace_bytes(n) > chain_room)
    return fail(TDR_ERR_ARG, "update_weights: n = %lld needs more than TDR_UW_INFO_FLOATS of scratch", (long long)n);
  hipLaunchKernelGGL(uw_pass1, dim3(UW_G), dim3(256), 0, s, raw_w, n, sc + 0);
  int rc = tdr_chain_total(raw_w, nullptr, 0, n, &ex->sum, chain_ws, s);                     // `sum` (:108-116)
  if (rc) return rc;
  hipLaunchKernelGGL(uw_pass2, dim3(UW_G), dim3(256), 0, s, raw_w, n, (const UwScratch*)(sc + 0), sc + 1, ex);
  rc = tdr_chain_total(raw_w, &ex->mean, 1, n, &ex->bsum, chain_ws, s);                      // `bottom_stddev` (:118-125)
  if (rc) return rc;
  hipLaunchKernelGGL(uw_pass3, dim3(UW_G), dim3(256), 0, s, raw_w, n, (const UwExact*)ex, (const UwScratch*)(sc + 1),
                     sc + 2, w_out);
  hipLaunchKernelGGL(uw_pass4, dim3(UW_G), dim3(256), 0, s, last_dist, n, (const UwScratch*)(sc + 2), sc + 3, w_out);
  hipLaunchKernelGGL(uw_pass5, dim3(UW_G), dim3(256), 0, s, n, (const UwScratch*)(sc + 3), sc + 4, w_out);
  hipLaunchKernelGGL(uw_pass6, dim3(1), dim3(256), 0, s, n, (const UwScratch*)(sc + 0), (const UwScratch*)(sc + 1),
                     (const UwScratch*)(sc + 4), (const UwExact*)ex, info_out);
  LAUNCH_CHECK("update_weights(multi)");
  return TDR_OK;
}

__global__ void resample_kernel(const float* __restrict__ runmax, int64_t n, int64_t n_new, float shift,
                                int64_t i_begin, int64_t i_end, int32_t* __restrict__ idx) {
  const int64_t i = i_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= i_end) return;
  const float sample = ((float)i + shift) / (float)n_new;  // particle_filter.cpp:176
  int64_t lo = 0, hi = n - 1;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (runmax[mid] > sample) hi = mid; else lo = mid + 1;
  }
  idx[i - i_begin] = (int32_t)lo;
}

extern "C" int tdr_k_resample(const float* runmax, int64_t n, int64_t n_new, float shift, int64_t i_begin,
                              int64_t i_end, int32_t* idx_out, void* stream) {
  if (!runmax || !idx_out) return fail(TDR_ERR_ARG, "resample: null pointer");
  if (n < 1 || n_new < 1 || i_begin < 0 || i_end > n_new || i_begin > i_end)
    return fail(TDR_ERR_ARG, "resample: bad range");
  if (i_begin == i_end) return TDR_OK;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)cdiv(i_end - i_begin, 256)), dim3(256), 0, (hipStream_t)stream,
                     runmax, n, n_new, shift, i_begin, i_end, idx_out);
  LAUNCH_CHECK("resample");
  return TDR_OK;
}

__global__ void resample_dev_kernel(const float* __restrict__ runmax, int64_t n, int64_t n_new, const float* __restrict__ shift_dev,
                                    int64_t i_begin, int64_t i_end, int32_t* __restrict__ idx) {
  const int64_t i = i_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= i_end) return;
  const float sample = ((float)i + *shift_dev) / (float)n_new;  // particle_filter.cpp:176
  int64_t lo = 0, hi = n - 1;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (runmax[mid] > sample) hi = mid; else lo = mid + 1;
  }
  idx[i - i_begin] = (int32_t)lo;
}
extern "C" int tdr_k_resample_dev(const float* runmax, int64_t n, int64_t n_new, const float* shift_dev, int64_t i_begin,
                                  int64_t i_end, int32_t* idx_out, void* stream) {
  if (!runmax || !idx_out || !shift_dev) return fail(TDR_ERR_ARG, "resample: null pointer");
  if (n < 1 || n_new < 1 || i_begin < 0 || i_end > n_new || i_begin > i_end)
    return fail(TDR_ERR_ARG, "resample: bad range");
  if (i_begin == i_end) return TDR_OK;
  hipLaunchKernelGGL(resample_dev_kernel, dim3((unsigned)cdiv(i_end - i_begin, 256)), dim3(256), 0, (hipStream_t)stream,
                     runmax, n, n_new, shift_dev, i_begin, i_end, idx_out);
  LAUNCH_CHECK("resample");
  return TDR_OK;
}

__global__ void gather_states_kernel(const float* __restrict__ src, int64_t src_cap, int64_t src_shard,
                                     const int32_t* __restrict__ idx, int64_t n_new, float* __restrict__ dst,
                                     int64_t dst_cap) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const int64_t j = idx[i];
  if (src_shard > 0) {  // all-gathered source: [rank][field][src_shard], global particle j = rank*src_shard + local
    const int64_t r = j / src_shard, l = j - r * src_shard;
#pragma unroll
    for (int f = 0; f < TDR_ST_FIELDS; f++) dst[f * dst_cap + i] = src[(r * TDR_ST_FIELDS + f) * src_shard + l];
  } else {
#pragma unroll
    for (int f = 0; f < TDR_ST_FIELDS; f++) dst[f * dst_cap + i] = src[f * src_cap + j];
  }
}

extern "C" int tdr_k_gather_states(const float* src, int64_t src_cap, int64_t src_shard, const int32_t* idx,
                                   int64_t n_new, float* dst, int64_t dst_cap, void* stream) {
  if (!src || !idx || !dst) return fail(TDR_ERR_ARG, "gather_states: null pointer");
  if (n_new < 0 || dst_cap < n_new || src_shard < 0) return fail(TDR_ERR_ARG, "gather_states: n_new exceeds capacity");
  if (n_new == 0) return TDR_OK;
  hipLaunchKernelGGL(gather_states_kernel, dim3((unsigned)cdiv(n_new, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     src_cap, src_shard, idx, n_new, dst, dst_cap);
  LAUNCH_CHECK("gather_states");
  return TDR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// K6: pose statistics (particle_filter.cpp:191-236) + geometric-mean scale (:343-357).  Double accumulation.
__global__ __launch_bounds__(1024) void mean_cov_kernel(const float* __restrict__ st, int64_t cap, int64_t n,
                                                       const float* __restrict__ about, float* __restrict__ out,
                                                       int libm_fma) {
  __shared__ double shd[16];
  __shared__ float ref[4];
  const int tid = threadIdx.x, nt = blockDim.x;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int64_t p = tid; p < n; p += nt) {
    const float sc = st[TDR_ST_SCALE * cap + p];
    const float x = st[TDR_ST_DX * cap + p] * sc + st[TDR_ST_INIT_X * cap + p];  // mlState, state_particle.cpp:98-102
    const float y = st[TDR_ST_DY * cap + p] * sc + st[TDR_ST_INIT_Y * cap + p];
    const float th = st[TDR_ST_THETA * cap + p];
    acc[0] += x; acc[1] += y; acc[2] += th; acc[3] += sc;
    acc[4] += (double)tdr_libm::cosf_v(th, libm_fma); acc[5] += (double)tdr_libm::sinf_v(th, libm_fma);   // :198-199
    acc[6] += log((double)sc);
  }
  double tot[7];
  for (int k = 0; k < 7; k++) tot[k] = block_sum_d(acc[k], shd);
  if (tid == 0) {
    const float fn = (float)n;
    float mean[4];
    mean[0] = (float)tot[0] / fn; mean[1] = (float)tot[1] / fn; mean[3] = (float)tot[3] / fn;
    mean[2] = atan2f((float)tot[5] / fn, (float)tot[4] / fn);  // :202
    for (int k = 0; k < 4; k++) out[k] = mean[k];
    out[20] = (float)exp(tot[6] / (double)n);  // freezeScale geo-mean
    out[21] = out[22] = out[23] = 0.f;
    if (about) {  // computeCov: about the max-likelihood particle's mlState (particle_filter.cpp:226-236)
      for (int k = 0; k < 4; k++) ref[k] = about[k];
    } else {
      for (int k = 0; k < 4; k++) ref[k] = mean[k];
    }
  }
  __syncthreads();
  double c[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t p = tid; p < n; p += nt) {
    const float sc = st[TDR_ST_SCALE * cap + p];
    float d[4];
    d[0] = (st[TDR_ST_DX * cap + p] * sc + st[TDR_ST_INIT_X * cap + p]) - ref[0];
    d[1] = (st[TDR_ST_DY * cap + p] * sc + st[TDR_ST_INIT_Y * cap + p]) - ref[1];
    d[2] = st[TDR_ST_THETA * cap + p] - ref[2];
    d[3] = sc - ref[3];
    while (d[2] > M_PI) d[2] = (float)((double)d[2] - 2 * M_PI);    // :215
    while (d[2] < -M_PI) d[2] = (float)((double)d[2] + 2 * M_PI);   // :216
    int k = 0;
    for (int a = 0; a < 4; a++)
      for (int b = a; b < 4; b++) c[k++] += (double)(d[a] * d[b]);
  }
  double ct[10];
  for (int k = 0; k < 10; k++) ct[k] = block_sum_d(c[k], shd);
  if (tid == 0) {
    int k = 0;
    for (int a = 0; a < 4; a++)
      for (int b = a; b < 4; b++) {
        float v = (float)ct[k++] / (float)(n - 1);  // :219
        out[4 + 4 * a + b] = v;
        out[4 + 4 * b + a] = v;
      }
  }
}

// Larger particle sets: the same two reductions over MC_WGS workgroups.  Per-workgroup partial sums (double) go to the
// scratch part of `out`; they are combined in workgroup order, so the result is a pure function of the inputs.
//   mc_sums_kernel (MC_WGS)  -> partial sums of {x, y, theta, scale, cos, sin, log scale}
//   mc_cov_kernel  (MC_WGS)  -> every workgroup combines the partial sums (mean / reference), then its share of the
//                               10 second moments about it
//   mc_final_kernel (1)      -> combines both, writes the 24 result floats
#define MC_WGS 128
#define MC_THREADS 256
#define MC_SINGLE_MAX_N 4096   // up to here one workgroup does everything in one launch
struct McScratch {
  double sums[MC_WGS][8];
  double mom[MC_WGS][10];
};
static_assert(24 * 4 + sizeof(McScratch) <= TDR_MEAN_COV_FLOATS * 4, "TDR_MEAN_COV_FLOATS too small");
__device__ __forceinline__ McScratch* mc_scratch(float* out) { return reinterpret_cast<McScratch*>(out + 24); }

__global__ __launch_bounds__(MC_THREADS) void mc_sums_kernel(const float* __restrict__ st, int64_t cap, int64_t n,
                                                             float* __restrict__ out, int libm_fma) {
  __shared__ double shd[16];
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int64_t p = (int64_t)blockIdx.x * MC_THREADS + threadIdx.x; p < n; p += (int64_t)MC_WGS * MC_THREADS) {
    const float sc = st[TDR_ST_SCALE * cap + p];
    const float x = st[TDR_ST_DX * cap + p] * sc + st[TDR_ST_INIT_X * cap + p];  // mlState, state_particle.cpp:98-102
    const float y = st[TDR_ST_DY * cap + p] * sc + st[TDR_ST_INIT_Y * cap + p];
    const float th = st[TDR_ST_THETA * cap + p];
    acc[0] += x; acc[1] += y; acc[2] += th; acc[3] += sc;
    acc[4] += (double)tdr_libm::cosf_v(th, libm_fma); acc[5] += (double)tdr_libm::sinf_v(th, libm_fma);   // :198-199
    acc[6] += log((double)sc);
  }
  McScratch* sc = mc_scratch(out);
  for (int k = 0; k < 7; k++) {
    const double t = block_sum_d(acc[k], shd);
    if (threadIdx.x == 0) sc->sums[blockIdx.x][k] = t;
  }
}
// mean / reference point from the partial sums, identically in every caller (workgroup order); `stage` = MC_WGS*8 doubles
__device__ __forceinline__ void mc_means(const McScratch* sc, int64_t n, const float* about, double* stage,
                                         double* sh /*[8]*/, float mean[4], float ref[4], float& geo) {
  __syncthreads();
  for (int t = threadIdx.x; t < MC_WGS * 8; t += MC_THREADS) stage[t] = (&sc->sums[0][0])[t];   // coalesced
  __syncthreads();
  if (threadIdx.x < 7) {
    double t = 0;
    for (int g = 0; g < MC_WGS; g++) t += stage[g * 8 + threadIdx.x];
    sh[threadIdx.x] = t;
  }
  __syncthreads();
  const float fn = (float)n;
  mean[0] = (float)sh[0] / fn; mean[1] = (float)sh[1] / fn; mean[3] = (float)sh[3] / fn;
  mean[2] = atan2f((float)sh[5] / fn, (float)sh[4] / fn);  // :202
  geo = (float)exp(sh[6] / (double)n);                      // freezeScale geo-mean
  for (int k = 0; k < 4; k++) ref[k] = about ? about[k] : mean[k];
}
__global__ __launch_bounds__(MC_THREADS) void mc_cov_kernel(const float* __restrict__ st, int64_t cap, int64_t n,
                                                            const float* __restrict__ about, float* __restrict__ out) {
  __shared__ double shd[16];
  __shared__ double shm[8];
  __shared__ double stage[MC_WGS * 10];
  McScratch* sc = mc_scratch(out);
  float mean[4], ref[4], geo;
  mc_means(sc, n, about, stage, shm, mean, ref, geo);
  double c[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t p = (int64_t)blockIdx.x * MC_THREADS + threadIdx.x; p < n; p += (int64_t)MC_WGS * MC_THREADS) {
    const float s = st[TDR_ST_SCALE * cap + p];
    float d[4];
    d[0] = (st[TDR_ST_DX * cap + p] * s + st[TDR_ST_INIT_X * cap + p]) - ref[0];
    d[1] = (st[TDR_ST_DY * cap + p] * s + st[TDR_ST_INIT_Y * cap + p]) - ref[1];
    d[2] = st[TDR_ST_THETA * cap + p] - ref[2];
    d[3] = s - ref[3];
    while (d[2] > M_PI) d[2] = (float)((double)d[2] - 2 * M_PI);    // :215
    while (d[2] < -M_PI) d[2] = (float)((double)d[2] + 2 * M_PI);   // :216
    int k = 0;
    for (int a = 0; a < 4; a++)
      for (int b = a; b < 4; b++) c[k++] += (double)(d[a] * d[b]);
  }
  for (int k = 0; k < 10; k++) {
    const double t = block_sum_d(c[k], shd);
    if (threadIdx.x == 0) sc->mom[blockIdx.x][k] = t;
  }
}
__global__ __launch_bounds__(MC_THREADS) void mc_final_kernel(int64_t n, const float* __restrict__ about,
                                                              float* __restrict__ out) {
  __shared__ double shm[8];
  __shared__ double shc[10];
  __shared__ double stage[MC_WGS * 10];
  const McScratch* sc = mc_scratch(out);
  float mean[4], ref[4], geo;
  mc_means(sc, n, about, stage, shm, mean, ref, geo);
  __syncthreads();
  for (int t = threadIdx.x; t < MC_WGS * 10; t += MC_THREADS) stage[t] = (&sc->mom[0][0])[t];
  __syncthreads();
  if (threadIdx.x < 10) {
    double t = 0;
    for (int g = 0; g < MC_WGS; g++) t += stage[g * 10 + threadIdx.x];
    shc[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < 4; k++) out[k] = mean[k];
    out[20] = geo;
    out[21] = out[22] = out[23] = 0.f;
    int k = 0;
    for (int a = 0; a < 4; a++)
      for (int b = a; b < 4; b++) {
        const float v = (float)shc[k++] / (float)(n - 1);  // :219
        out[4 + 4 * a + b] = v;
        out[4 + 4 * b + a] = v;
      }
  }
}

extern "C" int tdr_k_mean_cov(const float* st, int64_t cap, int64_t n, const float* about, float* out, void* stream) {
  if (!st || !out || n < 1 || cap < n) return fail(TDR_ERR_ARG, "mean_cov: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (n <= MC_SINGLE_MAX_N) {
    hipLaunchKernelGGL(mean_cov_kernel, dim3(1), dim3(1024), 0, s, st, cap, n, about, out, tdr_libm_fma());
  } else {
    hipLaunchKernelGGL(mc_sums_kernel, dim3(MC_WGS), dim3(MC_THREADS), 0, s, st, cap, n, out, tdr_libm_fma());
    hipLaunchKernelGGL(mc_cov_kernel, dim3(MC_WGS), dim3(MC_THREADS), 0, s, st, cap, n, about, out);
    hipLaunchKernelGGL(mc_final_kernel, dim3(1), dim3(MC_THREADS), 0, s, n, about, out);
  }
  LAUNCH_CHECK("mean_cov");
  return TDR_OK;
}

__global__ void set_scale_kernel(float* __restrict__ st, int64_t cap, int64_t n, const float* __restrict__ scale) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) st[TDR_ST_SCALE * cap + p] = *scale;
}
extern "C" int tdr_k_set_scale(float* st, int64_t cap, int64_t n, const float* scale_dev, void* stream) {
  if (!st || !scale_dev || n < 0 || cap < n) return fail(TDR_ERR_ARG, "set_scale: bad arguments");
  if (n == 0) return TDR_OK;
  hipLaunchKernelGGL(set_scale_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, st, cap, n,
                     scale_dev);
  LAUNCH_CHECK("set_scale");
  return TDR_OK;
}

__global__ void shift_init_kernel(float* __restrict__ st, int64_t cap, int64_t n, float dx, float dy) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) {
    st[TDR_ST_INIT_X * cap + p] += dx;
    st[TDR_ST_INIT_Y * cap + p] += dy;
  }
}
extern "C" int tdr_k_shift_init(float* st, int64_t cap, int64_t n, float dx, float dy, void* stream) {
  if (!st || n < 0 || cap < n) return fail(TDR_ERR_ARG, "shift_init: bad arguments");
  if (n == 0) return TDR_OK;
  hipLaunchKernelGGL(shift_init_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, st, cap, n,
                     dx, dy);
  LAUNCH_CHECK("shift_init");
  return TDR_OK;
}

// max_likelihood_particle_ = particles_[argmax] (particle_filter.cpp:145-147) points at the PRE-resample particle:
// keep its fields and its mlState (state_particle.cpp:98-102) on the device, so the update needs no host round trip.
__global__ void save_ml_state_kernel(const float* __restrict__ info, const float* __restrict__ st, int64_t cap,
                                     int64_t src_shard, int64_t n, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int64_t best = (int64_t)__float_as_int(info[0]);
  if (best < 0 || best >= n) best = 0;
  float f[TDR_ST_FIELDS];
  if (src_shard > 0) {   // all-gathered source [rank][field][src_shard], `best` is a global particle index
    const int64_t r = best / src_shard, l = best - r * src_shard;
#pragma unroll
    for (int k = 0; k < TDR_ST_FIELDS; k++) { f[k] = st[(r * TDR_ST_FIELDS + k) * src_shard + l]; out[k] = f[k]; }
  } else {
#pragma unroll
    for (int k = 0; k < TDR_ST_FIELDS; k++) { f[k] = st[(int64_t)k * cap + best]; out[k] = f[k]; }
  }
  out[7] = 0.f;
  out[8] = f[TDR_ST_DX] * f[TDR_ST_SCALE] + f[TDR_ST_INIT_X];
  out[9] = f[TDR_ST_DY] * f[TDR_ST_SCALE] + f[TDR_ST_INIT_Y];
  out[10] = f[TDR_ST_THETA];
  out[11] = f[TDR_ST_SCALE];
}
extern "C" int tdr_k_save_ml_state(const float* info, const float* st, int64_t cap, int64_t src_shard, int64_t n,
                                   float* out12, void* stream) {
  if (!info || !st || !out12 || n < 1 || src_shard < 0 || (src_shard == 0 && cap < n))
    return fail(TDR_ERR_ARG, "save_ml_state: bad arguments");
  hipLaunchKernelGGL(save_ml_state_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, info, st, cap, src_shard, n,
                     out12);
  LAUNCH_CHECK("save_ml_state");
  return TDR_OK;
}

// computeGMM's sample set (src/particle_filter.cpp:262-272): mlState().head<3>() of every (n/num)-th particle.
__global__ void sample_ml_states_kernel(const float* __restrict__ st, int64_t cap, int64_t n, int num,
                                        float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num) return;
  const int64_t p = min(n - 1, (int64_t)i * n / num);   // :265-266
  const float sc = st[TDR_ST_SCALE * cap + p];
  out[3 * i + 0] = st[TDR_ST_DX * cap + p] * sc + st[TDR_ST_INIT_X * cap + p];
  out[3 * i + 1] = st[TDR_ST_DY * cap + p] * sc + st[TDR_ST_INIT_Y * cap + p];
  out[3 * i + 2] = st[TDR_ST_THETA * cap + p];
}
extern "C" int tdr_k_sample_ml_states(const float* st, int64_t cap, int64_t n, int num, float* out, void* stream) {
  if (!st || !out || n < 1 || cap < n || num < 1) return fail(TDR_ERR_ARG, "sample_ml_states: bad arguments");
  hipLaunchKernelGGL(sample_ml_states_kernel, dim3((unsigned)cdiv(num, 256)), dim3(256), 0, (hipStream_t)stream, st,
                     cap, n, num, out);
  LAUNCH_CHECK("sample_ml_states");
  return TDR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Sharded filter (one rank per GPU, tdr_host.cpp): the buffers the two all-gathers move.
//   pack2   : {a[nl], b[nl]} -> one contiguous send buffer [2][nl]
//   unpack2 : the gathered [world][2][nl] -> a_glob[world*nl], b_glob[world*nl] in global particle order
//   unshard : the gathered state planes [world][7][nl] -> a plain SoA [7][cap] (pose statistics run on that)
__global__ void shard_pack2_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t nl,
                                   float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nl) { out[i] = a[i]; out[nl + i] = b[i]; }
}
__global__ void shard_unpack2_kernel(const float* __restrict__ in, int world, int64_t nl, float* __restrict__ a,
                                     float* __restrict__ b) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)world * nl) return;
  const int64_t r = g / nl, l = g - r * nl;
  a[g] = in[(2 * r) * nl + l];
  b[g] = in[(2 * r + 1) * nl + l];
}
__global__ void unshard_states_kernel(const float* __restrict__ in, int world, int64_t nl, float* __restrict__ st,
                                      int64_t cap) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)world * nl) return;
  const int64_t r = g / nl, l = g - r * nl;
#pragma unroll
  for (int f = 0; f < TDR_ST_FIELDS; f++) st[f * cap + g] = in[(r * TDR_ST_FIELDS + f) * nl + l];
}
extern "C" int tdr_k_shard_pack2(const float* a, const float* b, int64_t nl, float* out, void* stream) {
  if (!a || !b || !out || nl < 0) return fail(TDR_ERR_ARG, "shard_pack2: bad arguments");
  if (nl == 0) return TDR_OK;
  hipLaunchKernelGGL(shard_pack2_kernel, dim3((unsigned)cdiv(nl, 256)), dim3(256), 0, (hipStream_t)stream, a, b, nl, out);
  LAUNCH_CHECK("shard_pack2");
  return TDR_OK;
}
extern "C" int tdr_k_shard_unpack2(const float* in, int world, int64_t nl, float* a_glob, float* b_glob, void* stream) {
  if (!in || !a_glob || !b_glob || world < 1 || nl < 0) return fail(TDR_ERR_ARG, "shard_unpack2: bad arguments");
  if (nl == 0) return TDR_OK;
  hipLaunchKernelGGL(shard_unpack2_kernel, dim3((unsigned)cdiv((int64_t)world * nl, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, world, nl, a_glob, b_glob);
  LAUNCH_CHECK("shard_unpack2");
  return TDR_OK;
}
extern "C" int tdr_k_unshard_states(const float* in, int world, int64_t nl, float* st, int64_t cap, void* stream) {
  if (!in || !st || world < 1 || nl < 0 || cap < (int64_t)world * nl) return fail(TDR_ERR_ARG, "unshard_states: bad arguments");
  if (nl == 0) return TDR_OK;
  hipLaunchKernelGGL(unshard_states_kernel, dim3((unsigned)cdiv((int64_t)world * nl, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, world, nl, st, cap);
  LAUNCH_CHECK("unshard_states");
  return TDR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// layout helpers
__global__ void aos_to_soa_kernel(const tdr_state* __restrict__ aos, int64_t n, float* __restrict__ st, int64_t cap) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const tdr_state s = aos[p];
  st[TDR_ST_INIT_X * cap + p] = s.init_x_px;
  st[TDR_ST_INIT_Y * cap + p] = s.init_y_px;
  st[TDR_ST_DX * cap + p] = s.dx_m;
  st[TDR_ST_DY * cap + p] = s.dy_m;
  st[TDR_ST_THETA * cap + p] = s.theta;
  st[TDR_ST_SCALE * cap + p] = s.scale;
  st[TDR_ST_HAVE_INIT * cap + p] = s.have_init ? 1.f : 0.f;
}
__global__ void soa_to_aos_kernel(const float* __restrict__ st, int64_t cap, int64_t n, tdr_state* __restrict__ aos) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  tdr_state s;
  s.init_x_px = st[TDR_ST_INIT_X * cap + p];
  s.init_y_px = st[TDR_ST_INIT_Y * cap + p];
  s.dx_m = st[TDR_ST_DX * cap + p];
  s.dy_m = st[TDR_ST_DY * cap + p];
  s.theta = st[TDR_ST_THETA * cap + p];
  s.scale = st[TDR_ST_SCALE * cap + p];
  s.have_init = st[TDR_ST_HAVE_INIT * cap + p] != 0.f;
  s.pad_[0] = s.pad_[1] = s.pad_[2] = 0;
  aos[p] = s;
}
extern "C" int tdr_k_states_aos_to_soa(const tdr_state* aos, int64_t n, float* st, int64_t cap, void* stream) {
  if (!aos || !st || n < 0 || cap < n) return fail(TDR_ERR_ARG, "aos_to_soa: bad arguments");
  if (n == 0) return TDR_OK;
  hipLaunchKernelGGL(aos_to_soa_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, aos, n, st,
                     cap);
  LAUNCH_CHECK("aos_to_soa");
  return TDR_OK;
}
extern "C" int tdr_k_states_soa_to_aos(const float* st, int64_t cap, int64_t n, tdr_state* aos, void* stream) {
  if (!aos || !st || n < 0 || cap < n) return fail(TDR_ERR_ARG, "soa_to_aos: bad arguments");
  if (n == 0) return TDR_OK;
  hipLaunchKernelGGL(soa_to_aos_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, st, cap, n,
                     aos);
  LAUNCH_CHECK("soa_to_aos");
  return TDR_OK;
}

// Locality order: particles sorted by the Morton (Z-order) code of their centre at half-pixel granularity, so that
// the 64 particles of a wave — and the 4 lanes of each TA quad — read the same or neighbouring map cells.
// The key kernel is ours; the sort itself is rocPRIM's device radix sort (a utility, not a hot op).
__device__ __forceinline__ uint32_t spread_bits16(uint32_t v) {
  v &= 0xFFFFu;
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}
__global__ void loc_key_kernel(const float* __restrict__ st, int64_t cap, int64_t n, float xmax, float ymax,
                               uint32_t* __restrict__ keys, int32_t* __restrict__ vals) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float sc = st[TDR_ST_SCALE * cap + p];
  float cx = st[TDR_ST_DX * cap + p] * sc + st[TDR_ST_INIT_X * cap + p];
  float cy = st[TDR_ST_DY * cap + p] * sc + st[TDR_ST_INIT_Y * cap + p];
  if (!(cx == cx)) cx = 0.f;
  if (!(cy == cy)) cy = 0.f;
  const uint32_t hx = (uint32_t)fminf(fmaxf(cx * 2.f, 0.f), xmax);
  const uint32_t hy = (uint32_t)fminf(fmaxf(cy * 2.f, 0.f), ymax);
  keys[p] = spread_bits16(hx) | (spread_bits16(hy) << 1);
  vals[p] = (int32_t)p;
}

// Cartesian windows rotate with the particle (top_down_map.cpp:367-389), so two particles read the same cells only if
// they agree in heading as well as in position: the key interleaves (x, y, theta), theta quantised so that one step
// moves a sample at `theta_radius` cells from the centre by half a cell.  (Polar windows do not rotate — theta only
// shifts the scan rows — and are ordered by position alone.)
__device__ __forceinline__ uint64_t spread_bits21_3(uint64_t v) {   // bit i -> bit 3i
  v &= 0x1FFFFFull;
  v = (v | (v << 32)) & 0x1F00000000FFFFull;
  v = (v | (v << 16)) & 0x1F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
__global__ void loc_key_pose_kernel(const float* __restrict__ st, int64_t cap, int64_t n, float xmax, float ymax,
                                    float theta_step, uint64_t* __restrict__ keys, int32_t* __restrict__ vals) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float sc = st[TDR_ST_SCALE * cap + p];
  float cx = st[TDR_ST_DX * cap + p] * sc + st[TDR_ST_INIT_X * cap + p];
  float cy = st[TDR_ST_DY * cap + p] * sc + st[TDR_ST_INIT_Y * cap + p];
  float th = st[TDR_ST_THETA * cap + p];
  if (!(cx == cx)) cx = 0.f;
  if (!(cy == cy)) cy = 0.f;
  if (!(fabsf(th) < 1e6f)) th = 0.f;
  th = th - 6.2831855f * floorf(th / 6.2831855f);   // [0, 2 pi): any consistent reduction will do for an ordering
  const uint64_t hx = (uint64_t)fminf(fmaxf(cx * 2.f, 0.f), xmax);
  const uint64_t hy = (uint64_t)fminf(fmaxf(cy * 2.f, 0.f), ymax);
  const uint64_t ht = (uint64_t)fminf(fmaxf(th / theta_step, 0.f), 2097151.f);
  keys[p] = spread_bits21_3(hx) | (spread_bits21_3(hy) << 1) | (spread_bits21_3(ht) << 2);
  vals[p] = (int32_t)p;
}
static size_t radix_tmp_bytes64(int64_t n) {
  size_t bytes = 0;
  uint64_t* k = nullptr;
  int32_t* v = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0u, 63u, (hipStream_t)0, false);
  if (e != hipSuccess || bytes == 0) bytes = (size_t)(n + 4096) * 32;
  return bytes;
}
extern "C" size_t tdr_locality_pose_tmp_ints(int64_t n) {
  if (n < 1) n = 1;
  return (size_t)(5 * n + 64) + (radix_tmp_bytes64(n) + 3) / 4 + 64;
}
extern "C" int tdr_k_locality_order_pose(const float* st, int64_t cap, int64_t n, int map_rows, int map_cols,
                                         float theta_radius, int32_t* perm_out, int32_t* keys_tmp, void* stream) {
  if (!st || !perm_out || !keys_tmp || n < 0 || cap < n || map_rows < 1 || map_cols < 1 || !(theta_radius > 0.f))
    return fail(TDR_ERR_ARG, "locality_order_pose: bad arguments");
  if (map_rows > 1000000 || map_cols > 1000000) return fail(TDR_ERR_ARG, "locality_order_pose: map too large");
  if (n == 0) return TDR_OK;
  hipStream_t s = (hipStream_t)stream;
  // [keys_in 2n][keys_out 2n][vals_in n][pad][sort scratch], 8-byte aligned (the caller's buffer is)
  uint64_t* keys_in = reinterpret_cast<uint64_t*>(keys_tmp);
  uint64_t* keys_out = keys_in + n;
  int32_t* vals_in = keys_tmp + 4 * n;
  void* tmp = keys_tmp + 5 * n + 64 - ((5 * n) % 64);
  size_t tmp_bytes = radix_tmp_bytes64(n);
  hipLaunchKernelGGL(loc_key_pose_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, st, cap, n,
                     (float)(2 * map_cols - 1), (float)(2 * map_rows - 1), 0.5f / theta_radius, keys_in, vals_in);
  LAUNCH_CHECK("loc_key_pose");
  HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, perm_out, (size_t)n, 0u, 63u, s, false));
  return TDR_OK;
}

static size_t radix_tmp_bytes(int64_t n) {
  size_t bytes = 0;
  uint32_t* k = nullptr;
  int32_t* v = nullptr;
  hipError_t e = rocprim::radix_sort_pairs<TdrSortConfig>(nullptr, bytes, k, k, v, v, (size_t)n, 0u, 32u, (hipStream_t)0, false);
  if (e != hipSuccess || bytes == 0) bytes = (size_t)(n + 4096) * 16;  // no device to ask: a generous bound
  return bytes;
}

extern "C" size_t tdr_locality_tmp_ints(int64_t n, int map_rows, int map_cols) {
  (void)map_rows; (void)map_cols;
  if (n < 1) n = 1;
  return (size_t)(3 * n + 64) + (radix_tmp_bytes(n) + 3) / 4 + 64;
}

extern "C" int tdr_k_locality_order(const float* st, int64_t cap, int64_t n, int map_rows, int map_cols,
                                    int32_t* perm_out, int32_t* keys_tmp, void* stream) {
  if (!st || !perm_out || !keys_tmp || n < 0 || cap < n || map_rows < 1 || map_cols < 1)
    return fail(TDR_ERR_ARG, "locality_order: bad arguments");
  if (map_rows > 32767 || map_cols > 32767) return fail(TDR_ERR_ARG, "locality_order: map larger than 32767 px");
  if (n == 0) return TDR_OK;
  hipStream_t s = (hipStream_t)stream;
  uint32_t* keys_in = reinterpret_cast<uint32_t*>(keys_tmp);
  uint32_t* keys_out = keys_in + n;
  int32_t* vals_in = keys_tmp + 2 * n;
  void* tmp = keys_tmp + 3 * n + 64 - ((3 * n) % 64);  // keep the sort's scratch 256-byte aligned
  size_t tmp_bytes = radix_tmp_bytes(n);
  hipLaunchKernelGGL(loc_key_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, st, cap, n,
                     (float)(2 * map_cols - 1), (float)(2 * map_rows - 1), keys_in, vals_in);
  LAUNCH_CHECK("loc_key");
  unsigned bits = 2;
  while ((1u << (bits / 2)) < (unsigned)(2 * std::max(map_rows, map_cols)) && bits < 32) bits += 2;
  HIP_TRY(rocprim::radix_sort_pairs<TdrSortConfig>(tmp, tmp_bytes, keys_in, keys_out, vals_in, perm_out, (size_t)n, 0u, bits, s,
                                    false));
  return TDR_OK;
}
