// tdr_geo.hip — renderGeometricTopDown: the ground / obstacle images of a scan.
//   polar:     src/scan_renderer_polar.cpp:6-81   per theta bin: points sorted by range descending, slope walk
//   Cartesian: src/scan_renderer.cpp:7-53         per scan line of the organised cloud: slope walk, ground cells filled
//                                                 along the line between consecutive returns
// The reference's node has the call commented out (src/top_down_render.cpp:540) and publishes zero images; the functions
// themselves are part of the class surface and are reproduced here exactly (integer counts; tests/test_geo.py checks
// them bit for bit).  Both walks are serial per bin / per scan line by definition (every step depends on the previous return), so
// the parallel axis is the bin / the line:
//   polar:     geo_keys_kernel (one thread per point: theta bin + range -> 64-bit key) -> rocPRIM radix sort (stable:
//              equal ranges keep their input order — the documented tie rule where std::sort leaves it open) ->
//              geo_walk_polar_kernel (one thread per theta bin walks its run of the sorted list; it owns its image row)
//   Cartesian: geo_walk_cart_kernel (one thread per scan line; lines may cross, so cells are counted with atomics —
//              the addends are all 1, the result is order-independent and exact below 2^24)
// Conventions where the reference leaves a choice: unqualified atan2 / sqrt / abs on
// floats are the float overloads; a point whose x or y is not finite is dropped (the reference would index a vector
// with (int)NaN); float -> int conversions follow x86 (cvttss2si: NaN / out of range -> INT_MIN).
#include <rocprim/device/device_radix_sort.hpp>

#include "tdr_common.h"
#include "tdr_atan2f.h"

__device__ __forceinline__ int cvt_x86(float v) {
  return (v >= -2147483648.f && v < 2147483648.f) ? (int)v : (int)0x80000000;
}
__device__ __forceinline__ bool finite_f(float v) { return fabsf(v) < INFINITY; }
#define GEO_MAX_IND (1 << 24)

struct GeoArgs {
  const float* pts;
  int stride;
  int64_t width, height, n;
  float res, ang_res;
  int rows, cols;      // image shape (polar: theta bins x range bins)
  float* img;          // [2][rows*cols]
};

// iteration position k = idx*height + idy (the reference's loop order, :27-28) -> element idy*width + idx
__device__ __forceinline__ const float* geo_point(const GeoArgs& a, int64_t k) {
  const int64_t idx = k / a.height, idy = k - idx * a.height;
  return a.pts + (idy * a.width + idx) * a.stride;
}

__global__ __launch_bounds__(256) void geo_keys_kernel(GeoArgs a, uint64_t* __restrict__ keys, int32_t* __restrict__ vals) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.n) return;
  const float* p = geo_point(a, k);
  const float x = p[0], y = p[1];
  uint64_t key = ~0ull;   // not binned: sorts behind every bin
  if (!(x == 0.f && y == 0.f) && finite_f(x) && finite_f(y)) {                 // :30
    const float theta = tdr_atan2f(x, y);                                       // :32
    const float r = sqrtf(x * x + y * y);                                       // :33
    float t = roundf(theta / a.ang_res) + (float)(a.rows / 2);                  // :36-37
    t = t < 0.f ? 0.f : (t > (float)(a.rows - 1) ? (float)(a.rows - 1) : t);    // std::clamp<float>
    key = ((uint64_t)(uint32_t)(int)t << 32) | (uint64_t)(~__float_as_uint(r)); // r >= 0: ~bits ascending = r descending
  }
  keys[k] = key;
  vals[k] = (int32_t)k;
}

__global__ __launch_bounds__(64) void geo_walk_polar_kernel(GeoArgs a, const uint64_t* __restrict__ keys,
                                                            const int32_t* __restrict__ vals) {
  const int theta_ind = blockIdx.x * blockDim.x + threadIdx.x;
  if (theta_ind >= a.rows) return;
  auto lower = [&](uint64_t want) {   // first position whose key >= want
    int64_t lo = 0, hi = a.n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int64_t b = lower((uint64_t)theta_ind << 32), e = lower((uint64_t)(theta_ind + 1) << 32);
  float* ground = a.img;
  float* obst = a.img + (int64_t)a.rows * a.cols;
  float lx = 0.f, ly = 0.f, lz = 0.f;                                           // :54
  bool last_high_grad = false;
  int last_r_ind = 0;
  for (int64_t s = b; s < e; s++) {
    const float* p = geo_point(a, vals[s]);
    const float x = p[0], y = p[1], z = p[2];
    const float r = sqrtf(x * x + y * y);
    const float dx = x - lx, dy = y - ly;
    const float dist = sqrtf(dx * dx + dy * dy);                                // :58
    const float slope = fabsf(z - lz) / dist;                                   // :59
    const int r_ind = cvt_x86(roundf(r / a.res));                               // :60
    if (slope > 1.f) {                                                          // :62-66
      if (r_ind >= 0 && r_ind < a.cols) obst[theta_ind + (int64_t)a.rows * r_ind] += 1.f;
      last_high_grad = true;
    } else if ((double)slope < 0.3 && !last_high_grad) {                        // :67-72
      const int hi = min(r_ind, a.cols - 1);                                    // `if (i < img_size[1])`
      // last_r_ind < 0 only after a return 2^31 range bins away (INT_MIN): the reference then writes in front of its
      // image (undefined); cells of negative range do not exist here
      for (int i = max(last_r_ind, 0); i <= hi; i++) ground[theta_ind + (int64_t)a.rows * i] += 1.f;
    } else {
      last_high_grad = false;                                                   // :73-75
    }
    lx = x; ly = y; lz = z;                                                     // :76-77
    last_r_ind = r_ind;
  }
}

__global__ __launch_bounds__(64) void geo_walk_cart_kernel(GeoArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.width) return;
  float* ground = a.img;
  float* obst = a.img + (int64_t)a.rows * a.cols;
  float lx = 0.f, ly = 0.f, lz = 0.f;                                           // :17
  int last_x = a.cols / 2, last_y = a.rows / 2;                                 // :19
  bool last_high_grad = false;
  for (int64_t idy = 0; idy < a.height; idy++) {                                // :23
    const float* p = a.pts + (idy * a.width + idx) * a.stride;
    const float x = p[0], y = p[1], z = p[2];
    if (x == 0.f && y == 0.f) continue;                                         // :26
    if (!finite_f(x) || !finite_f(y)) continue;
    const int x_ind = cvt_x86(roundf(x / a.res) + (float)(a.cols / 2));         // :27
    const int y_ind = cvt_x86(roundf(y / a.res) + (float)(a.rows / 2));         // :28
    // a return more than 2^24 cells away: the reference's line interpolation below overflows / does not terminate for
    // it; dropped like a non-finite point (and the loop count per return stays bounded)
    if (x_ind > GEO_MAX_IND || x_ind < -GEO_MAX_IND || y_ind > GEO_MAX_IND || y_ind < -GEO_MAX_IND) continue;
    const float dx = x - lx, dy = y - ly;
    const float dist = sqrtf(dx * dx + dy * dy);                                // :30
    const float slope = fabsf(z - lz) / dist;                                   // :31
    if (slope > 1.f) {                                                          // :32-36
      if (x_ind >= 0 && x_ind < a.cols && y_ind >= 0 && y_ind < a.rows)
        atomicAdd(&obst[y_ind + (int64_t)a.rows * x_ind], 1.f);
      last_high_grad = true;
    } else if ((double)slope < 0.3 && !last_high_grad) {                        // :37-45
      const long long ddx = (long long)x_ind - last_x, ddy = (long long)y_ind - last_y;
      const int nrm = (int)sqrt((double)(ddx * ddx + ddy * ddy));               // Vector2i::norm(): truncated
      const double inc = 1. / (double)nrm;                                      // nrm == 0: inf, one iteration
      for (float i = 0.f; i < 1.f; i = (float)((double)i + inc)) {              // :39
        const int ix = (int)roundf((float)last_x + i * (float)ddx);
        const int iy = (int)roundf((float)last_y + i * (float)ddy);
        if (ix >= 0 && ix < a.cols && iy >= 0 && iy < a.rows) atomicAdd(&ground[iy + (int64_t)a.rows * ix], 1.f);
      }
    } else {
      last_high_grad = false;                                                   // :46-48
    }
    lx = x; ly = y; lz = z;                                                     // :49
    last_x = x_ind; last_y = y_ind;                                             // :50
  }
}

static size_t geo_sort_tmp_bytes(int64_t n) {
  size_t bytes = 0;
  uint64_t* k = nullptr;
  int32_t* v = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0u, 64u, (hipStream_t)0, false);
  if (e != hipSuccess || bytes == 0) bytes = (size_t)(n + 4096) * 32;   // no device to ask: a generous bound
  return bytes;
}
// [keys_in 8n][keys_out 8n][vals_in 4n][vals_out 4n][sort scratch]
extern "C" int64_t tdr_raster_geo_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  return (int64_t)(24 * n + 512 + geo_sort_tmp_bytes(n));
}

static int geo_check(const float* pts, int stride, int64_t width, int64_t height, float res, int rows, int cols,
                     const float* img, const char* who) {
  if (!img) return fail(TDR_ERR_ARG, "%s: null output", who);
  if (width < 0 || height < 0 || (width * height > 0 && !pts)) return fail(TDR_ERR_ARG, "%s: null points", who);
  if (width * height > 0x7fffffffLL) return fail(TDR_ERR_ARG, "%s: more than 2^31 points", who);
  if (stride < 3) return fail(TDR_ERR_ARG, "%s: stride must cover x, y, z", who);
  if (rows < 1 || cols < 1) return fail(TDR_ERR_ARG, "%s: bad image shape", who);
  if (!(res > 0.f)) return fail(TDR_ERR_ARG, "%s: res must be > 0", who);
  return TDR_OK;
}

extern "C" int tdr_k_raster_geo_polar(const float* pts, int stride, int64_t width, int64_t height, float res,
                                      float ang_res, int nb, int nr, float* img_out, void* workspace, void* stream) {
  if (int rc = geo_check(pts, stride, width, height, res, nb, nr, img_out, "raster_geo_polar")) return rc;
  if (!workspace) return fail(TDR_ERR_ARG, "raster_geo_polar: workspace required (tdr_raster_geo_workspace_bytes)");
  if (!(ang_res > 0.f)) return fail(TDR_ERR_ARG, "raster_geo_polar: ang_res must be > 0");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(img_out, 0, sizeof(float) * (size_t)2 * nb * nr, s));                   // :11-13
  const int64_t n = width * height;
  if (n == 0) return TDR_OK;
  GeoArgs a{pts, stride, width, height, n, res, ang_res, nb, nr, img_out};
  char* w = reinterpret_cast<char*>(workspace);
  uint64_t* keys_in = reinterpret_cast<uint64_t*>(w);
  uint64_t* keys_out = keys_in + n;
  int32_t* vals_in = reinterpret_cast<int32_t*>(keys_out + n);
  int32_t* vals_out = vals_in + n;
  void* tmp = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(vals_out + n) + 255) & ~(uintptr_t)255);
  size_t tmp_bytes = geo_sort_tmp_bytes(n);
  hipLaunchKernelGGL(geo_keys_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, a, keys_in, vals_in);
  LAUNCH_CHECK("geo_keys");
  HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, 64u, s, false));
  hipLaunchKernelGGL(geo_walk_polar_kernel, dim3((unsigned)cdiv(nb, 64)), dim3(64), 0, s, a,
                     (const uint64_t*)keys_out, (const int32_t*)vals_out);
  LAUNCH_CHECK("geo_walk_polar");
  return TDR_OK;
}

extern "C" int tdr_k_raster_geo_cart(const float* pts, int stride, int64_t width, int64_t height, float res, int rows,
                                     int cols, float* img_out, void* stream) {
  if (int rc = geo_check(pts, stride, width, height, res, rows, cols, img_out, "raster_geo_cart")) return rc;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(img_out, 0, sizeof(float) * (size_t)2 * rows * cols, s));               // :12-14
  if (width * height == 0) return TDR_OK;
  GeoArgs a{pts, stride, width, height, width * height, res, 0.f, rows, cols, img_out};
  hipLaunchKernelGGL(geo_walk_cart_kernel, dim3((unsigned)cdiv(width, 64)), dim3(64), 0, s, a);
  LAUNCH_CHECK("geo_walk_cart");
  return TDR_OK;
}
