// tdr_kernels.hip — hand-written HIP kernels for gfx950 (MI355X) + the stateless C-ABI launchers of include/tdr.h.
//
// Compile with -ffp-contract=off: every float expression that decides a bin / cell index must round exactly like
// the reference's x86-64 code (no FMA contraction); FMAs are spelled out (__builtin_fmaf) where they are wanted.
//
// Kernel map (reference loop nest -> kernel), see DESIGN.md:
//   K0 pack_map_kernel        class_maps_/class_mask_ (top_down_map.h:77-79)          -> interleaved cell records
//   K1 raster_kernel          scan_renderer_polar.cpp:93-108 / scan_renderer.cpp:65-77 -> LDS-tile bin counters
//   K2 score_polar_kernel     top_down_map_polar.cpp:28-52 + state_particle.cpp:132-143 (lane = particle)
//      score_finalize_kernel  state_particle.cpp:117-120,136-139,154,161-176,212
//   K3 propagate_kernel       state_particle.cpp:57-78
//   K4 update_weights_kernel  particle_filter.cpp:107-147
//   K5 prefix_kernel / resample_kernel / gather_states_kernel   particle_filter.cpp:172-185
//   K6 mean_cov_kernel        particle_filter.cpp:191-236, 343-357
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <utility>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "tdr.h"

// tuning knobs (compile-time)
#ifndef TDR_SCORE_U
#define TDR_SCORE_U 4          // samples whose loads are kept in flight together in the scoring loop
#endif
#ifndef TDR_INIT_SCAN_LDS
#define TDR_INIT_SCAN_LDS 1   // init search: candidates' scan records via LDS broadcast (1) or the scalar cache (0)
#endif
#ifndef TDR_XCD_SWIZZLE
#define TDR_XCD_SWIZZLE 0
#endif
#ifndef TDR_OOB_ALIAS
#define TDR_OOB_ALIAS 1      // all out-of-bounds samples read one guard record (A/B on MI355X: -11 % on config 2)
#endif

// ------------------------------------------------------------------------------------------------------------------
// error plumbing
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(TDR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define LAUNCH_CHECK(name)                                                                   \
  do {                                                                                       \
    hipError_t e_ = hipGetLastError();                                                       \
    if (e_ != hipSuccess) return fail(TDR_ERR_HIP, "launch %s: %s", name, hipGetErrorString(e_)); \
  } while (0)

extern "C" const char* tdr_last_error(void) { return g_err; }
extern "C" int tdr_set_error(int code, const char* msg) { return fail(code, "%s", msg ? msg : ""); }
extern "C" int tdr_version(void) { return 100; }
extern "C" int tdr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" int tdr_rec_floats(int ncls) { return 4 * ((ncls + 1 + 3) / 4); }

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------------------------------
// K0: map packing.  Output is row-major over a GUARDED grid of (rows+2) x (cols+2) cell records: one ring of
// all-zero records around the map, so that a sample coordinate clamped to [-1, rows] x [-1, cols] always addresses a
// valid record and "out of bounds" needs no branch or select in the scoring loop — the guard record is exactly what
// the reference returns there: distance 0 (top_down_map_polar.cpp:39) and unknown (:51).
// Record slots: [0,ncls) distances, rf-1 = known (1 - mask); when a spare slot exists (ncls+2 <= rf) slot rf-2 also
// holds `known`, paired with a constant 1 in the scan record, so the known-cell count rides on the packed FMAs.
__host__ __device__ inline bool tdr_has_kslot(int ncls, int rf) { return ncls + 2 <= rf; }

__global__ void pack_map_kernel(const float* __restrict__ maps, const uint8_t* __restrict__ mask, int ncls, int rows,
                                int cols, int rf, float* __restrict__ rec) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t gcols = cols + 2, gcell = (int64_t)(rows + 2) * gcols;
  if (idx >= gcell) return;
  float* o = rec + idx * rf;
  for (int k = 0; k < rf; k++) o[k] = 0.f;
  const int r = (int)(idx / gcols) - 1, c = (int)(idx % gcols) - 1;
  if (r < 0 || r >= rows || c < 0 || c >= cols) return;  // guard record
  const int64_t ncell = (int64_t)rows * cols;
  const int64_t src = (int64_t)r + (int64_t)rows * c;    // the reference's column-major layout
  for (int k = 0; k < ncls; k++) o[k] = maps[(int64_t)k * ncell + src];
  const float known = 1.f - (float)mask[src];            // `1 - mask.cast<float>()` (state_particle.cpp:199,209)
  o[rf - 1] = known;
  if (tdr_has_kslot(ncls, rf)) o[rf - 2] = known;
}

extern "C" size_t tdr_map_rec_floats_total(int ncls, int rows, int cols) {
  return (size_t)(rows + 2) * (size_t)(cols + 2) * (size_t)tdr_rec_floats(ncls);
}

extern "C" int tdr_k_pack_map(const float* class_maps, const uint8_t* class_mask, int ncls, int rows, int cols,
                              float* rec_out, void* stream) {
  if (!class_maps || !class_mask || !rec_out) return fail(TDR_ERR_ARG, "pack_map: null pointer");
  if (ncls < 1 || ncls > TDR_MAX_CLASSES || rows < 1 || cols < 1) return fail(TDR_ERR_ARG, "pack_map: bad shape");
  if (rows > 8000000 || cols > 8000000) return fail(TDR_ERR_ARG, "pack_map: map side exceeds 2^23");
  int rf = tdr_rec_floats(ncls);
  int64_t gcell = (int64_t)(rows + 2) * (cols + 2);
  if (gcell * rf * 4 > (int64_t)0xFFFFFFF0ll) return fail(TDR_ERR_ARG, "pack_map: map exceeds 4 GiB of records");
  hipLaunchKernelGGL(pack_map_kernel, dim3((unsigned)cdiv(gcell, 256)), dim3(256), 0, (hipStream_t)stream,
                     class_maps, class_mask, ncls, rows, cols, rf, rec_out);
  LAUNCH_CHECK("pack_map");
  return TDR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// N1: map ingest on the device — TopDownMap::loadCompressedRasterMap (src/top_down_map.cpp:116-144) followed by
// computeDists (:289-326) for a label image (cv::Mat CV_8UC1 layout), i.e. what TopDownMap::updateMap (:146-157) does
// when a new aerial map arrives at run time.  The distance transform is exact (cv::distanceTransform DIST_L2 /
// DIST_MASK_PRECISE): squared Euclidean distances are integers, minimised exactly; because distances are truncated at
// 50 (:315) only cells within R = ceil(50/resolution) matter, so both separable passes are windowed brute force —
// every cell independent, integer arithmetic, one correctly rounded sqrtf at the end.
#define INGEST_MAXC 16
__global__ void ingest_labels_kernel(const uint8_t* __restrict__ img, int img_h, int img_w,
                                     const int32_t* __restrict__ lut, int lut_size, int ncls, int rows, int cols,
                                     float resolution, int8_t* __restrict__ cls_map) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  const int yi = (int)(idx / cols), xi = (int)(idx % cols);
  // :137-138 — row 0 of the map is the bottom row of the image
  int iy = (int)((float)img_h - (float)yi * resolution - 1.f);
  iy = iy > 0 ? iy : 0;
  int ix = (int)((float)xi * resolution);
  ix = ix < img_w - 1 ? ix : img_w - 1;
  const int label = img[(int64_t)iy * img_w + ix];
  int c = label < lut_size ? lut[label] : -1;
  if (c < 0 || c >= ncls) c = -1;  // :139
  cls_map[idx] = (int8_t)c;
}

// pass 1: per cell and class, distance (in cells, along the column) to the nearest cell of that class, capped at 255
__global__ void ingest_coldist_kernel(const int8_t* __restrict__ cls_map, int ncls, int rows, int cols, int R,
                                      uint8_t* __restrict__ g /* [rows*cols][INGEST_MAXC] */) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  const int y = (int)(idx / cols), x = (int)(idx % cols);
  int gd[INGEST_MAXC];
#pragma unroll
  for (int c = 0; c < INGEST_MAXC; c++) gd[c] = 255;
  for (int d = 0; d <= R; d++) {
    const int ya = y - d, yb = y + d;
    const int ca = ya >= 0 ? (int)cls_map[(int64_t)ya * cols + x] : -1;
    const int cb = yb < rows ? (int)cls_map[(int64_t)yb * cols + x] : -1;
#pragma unroll
    for (int c = 0; c < INGEST_MAXC; c++)
      if ((ca == c || cb == c) && gd[c] == 255) gd[c] = d;
  }
  uint8_t* o = g + idx * INGEST_MAXC;
#pragma unroll
  for (int c = 0; c < INGEST_MAXC; c++) o[c] = (uint8_t)gd[c];
  (void)ncls;
}

// pass 2: exact squared distance = min over the row window of dx^2 + g^2; then the reference's post-processing
__global__ void ingest_rowmin_kernel(const int8_t* __restrict__ cls_map, const uint8_t* __restrict__ g, int ncls,
                                     int rows, int cols, int R, float resolution, int rf, float* __restrict__ rec) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  const int y = (int)(idx / cols), x = (int)(idx % cols);
  int best[INGEST_MAXC];
#pragma unroll
  for (int c = 0; c < INGEST_MAXC; c++) best[c] = 0x7fffffff;
  const int x0 = x - R > 0 ? x - R : 0, x1 = x + R < cols - 1 ? x + R : cols - 1;
  for (int xx = x0; xx <= x1; xx++) {
    const int dx2 = (xx - x) * (xx - x);
    const uint4 gv = *reinterpret_cast<const uint4*>(g + ((int64_t)y * cols + xx) * INGEST_MAXC);
    const unsigned wv[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int c = 0; c < INGEST_MAXC; c++) {
      const int gd = (int)((wv[c >> 2] >> (8 * (c & 3))) & 0xFF);
      const int cand = gd == 255 ? 0x7fffffff : dx2 + gd * gd;
      best[c] = cand < best[c] ? cand : best[c];
    }
  }
  const bool unknown = cls_map[idx] < 0;  // no class at this cell (:294-299): mask = 1, distances zeroed (:317)
  float* o = rec + ((int64_t)(y + 1) * (cols + 2) + (x + 1)) * rf;
#pragma unroll
  for (int c = 0; c < INGEST_MAXC; c++) {
    if (c < ncls) {
      float d = best[c] == 0x7fffffff ? 3.0e38f : sqrtf((float)best[c]);  // cv::distanceTransform, precise L2
      d = d * resolution;                                                   // :314
      d = d > 50.f ? 50.f : d;                                              // :315 THRESH_TRUNC
      o[c] = unknown ? 0.f : d;
    }
  }
  const float known = unknown ? 0.f : 1.f;
  o[rf - 1] = known;
  if (tdr_has_kslot(ncls, rf)) o[rf - 2] = known;
}

__global__ void zero_floats_kernel(float* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

extern "C" size_t tdr_map_ingest_workspace_bytes(int ncls, int rows, int cols) {
  (void)ncls;
  return (size_t)rows * cols * (1 + INGEST_MAXC) + 256;
}
extern "C" int tdr_map_ingest_shape(int img_h, int img_w, float resolution, int* rows, int* cols) {
  if (!rows || !cols || !(resolution > 0) || img_h < 1 || img_w < 1) return fail(TDR_ERR_ARG, "ingest_shape: bad arguments");
  *rows = (int)((float)img_h / resolution);  // static_cast<int>(map.size().height/params_.resolution) (:121)
  *cols = (int)((float)img_w / resolution);
  return TDR_OK;
}

extern "C" int tdr_k_map_from_labels(const uint8_t* label_img, int img_h, int img_w, const int32_t* flatten_lut,
                                     int lut_size, int ncls, float resolution, float* rec_out, void* workspace,
                                     void* stream) {
  if (!label_img || !flatten_lut || !rec_out || !workspace) return fail(TDR_ERR_ARG, "map_from_labels: null pointer");
  if (ncls < 1 || ncls > TDR_MAX_CLASSES || lut_size < 1 || lut_size > 256)
    return fail(TDR_ERR_ARG, "map_from_labels: bad class count / lut size");
  int rows, cols;
  int rc = tdr_map_ingest_shape(img_h, img_w, resolution, &rows, &cols);
  if (rc) return rc;
  if (rows < 1 || cols < 1) return fail(TDR_ERR_ARG, "map_from_labels: empty map");
  const int R = (int)std::ceil(50.0 / (double)resolution);
  if (R > 250) return fail(TDR_ERR_ARG, "map_from_labels: resolution %g needs a %d-cell window (max 250)", resolution, R);
  const int rf = tdr_rec_floats(ncls);
  hipStream_t s = (hipStream_t)stream;
  int8_t* cls_map = reinterpret_cast<int8_t*>(workspace);
  uint8_t* g = reinterpret_cast<uint8_t*>(workspace) + (((size_t)rows * cols + 255) & ~(size_t)255);
  const int64_t ncell = (int64_t)rows * cols;
  const int64_t nrec = (int64_t)tdr_map_rec_floats_total(ncls, rows, cols);
  hipLaunchKernelGGL(zero_floats_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, rec_out, nrec);  // guard ring
  hipLaunchKernelGGL(ingest_labels_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, s, label_img, img_h, img_w,
                     flatten_lut, lut_size, ncls, rows, cols, resolution, cls_map);
  hipLaunchKernelGGL(ingest_coldist_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, s, (const int8_t*)cls_map,
                     ncls, rows, cols, R, g);
  hipLaunchKernelGGL(ingest_rowmin_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, s, (const int8_t*)cls_map,
                     (const uint8_t*)g, ncls, rows, cols, R, resolution, rf, rec_out);
  LAUNCH_CHECK("map_from_labels");
  return TDR_OK;
}

// Back to the reference's layout (class_maps_ / class_mask_: column-major per class), e.g. for the host copy that
// getClassesAtPoint and the particle initialisation read.
__global__ void unpack_map_kernel(const float* __restrict__ rec, int ncls, int rows, int cols, int rf,
                                  float* __restrict__ maps, uint8_t* __restrict__ mask) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ncell = (int64_t)rows * cols;
  if (idx >= ncell) return;
  const int r = (int)(idx % rows), c = (int)(idx / rows);  // idx walks the column-major output
  const float* o = rec + ((int64_t)(r + 1) * (cols + 2) + (c + 1)) * rf;
  for (int k = 0; k < ncls; k++) maps[(int64_t)k * ncell + idx] = o[k];
  mask[idx] = o[rf - 1] != 0.f ? 0 : 1;
}
extern "C" int tdr_k_unpack_map(const float* rec, int ncls, int rows, int cols, float* class_maps_out,
                                uint8_t* class_mask_out, void* stream) {
  if (!rec || !class_maps_out || !class_mask_out || ncls < 1 || ncls > TDR_MAX_CLASSES || rows < 1 || cols < 1)
    return fail(TDR_ERR_ARG, "unpack_map: bad arguments");
  const int64_t ncell = (int64_t)rows * cols;
  hipLaunchKernelGGL(unpack_map_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, (hipStream_t)stream, rec, ncls,
                     rows, cols, tdr_rec_floats(ncls), class_maps_out, class_mask_out);
  LAUNCH_CHECK("unpack_map");
  return TDR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Host: polar sampling table (top_down_map.cpp:367-389 + top_down_map_polar.cpp:7-19).  glibc cosf/sinf, like the
// reference's host code; the table is an input of the scoring kernel.
static inline float linspaced_f(int i, int n, float low, float high) {
  int size1 = (n == 1) ? 1 : n - 1;
  float step = (n == 1) ? 0.0f : (high - low) / (float)(n - 1);
  if (fabsf(high) < fabsf(low)) return (i == 0) ? low : (high - (float)(size1 - i) * step);
  return (i == size1) ? high : (low + (float)i * step);
}
extern "C" int tdr_polar_table_host(int nb, int nr, float ang_res, float resolution, float* tab) {
  if (nb < 1 || nr < 1 || !tab) return fail(TDR_ERR_ARG, "polar_table: bad arguments");
  // samplePts(0, 0, pts, cols=nr, rows=nb, res=1): row0 = L_nb[i], row1 = L_nr[j]; identity rotation
  float lo_r = (float)((double)(-1.f * (float)(nb - 1)) / 2.), hi_r = (float)((double)(1.f * (float)(nb - 1)) / 2.);
  float lo_c = (float)((double)(-1.f * (float)(nr - 1)) / 2.), hi_c = (float)((double)(1.f * (float)(nr - 1)) / 2.);
  float c = cosf(0.f), s = sinf(0.f);
  float inv_res = (float)(1. / (double)resolution);
  float first = 0.f;
  for (int j = 0; j < nr; j++) {
    for (int i = 0; i < nb; i++) {
      float p0 = linspaced_f(i, nb, lo_r, hi_r), p1 = linspaced_f(j, nr, lo_c, hi_c);
      float a = c * p0 + (-s) * p1 + 0.f;
      float r = s * p0 + c * p1 + 0.f;
      if (i == 0 && j == 0) first = r;
      r = r + (-first);
      a = a * ang_res;
      r = r * inv_res;
      size_t k = (size_t)i + (size_t)nb * j;
      tab[2 * k] = cosf(a) * r;
      tab[2 * k + 1] = sinf(a) * r;
    }
  }
  return TDR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// K1: scan raster.  Each workgroup owns a tile of `cpt` image columns (range bins) x all rows x all classes as u32
// counters in LDS, streams every point with coalesced loads and keeps those that fall into its tile.  Integer LDS
// atomics -> exact, order-independent counts; plain coalesced stores out (the tile is written whole, zeros
// included, so no memset pass is needed).
struct RasterArgs {
  const float* pts;
  int stride, ioff;
  int64_t n;
  float res, ang_res;
  const int32_t* lut;
  int ncls, rows, cols, rf, cpt, polar;
  float* img;
  float* pk;
  uint32_t* keys;   // optional [n]: bin of every point, computed once by raster_keys_kernel (col << 20 | class << 16 | row)
};
#define RASTER_NO_BIN 0xFFFFFFFFu
#define RASTER_KEY_MAX_COLS 4095
#define RASTER_KEY_MAX_ROWS 65535

// atan2f exactly as glibc computes it (the reference calls the host libm, src/scan_renderer_polar.cpp:97).
// glibc's float atan2f / atanf are the fdlibm algorithms (argument reduction to four intervals + an 11-term odd/even
// polynomial, all in float); restated here operation for operation — compiled with -ffp-contract=off, IEEE divide —
// so the device result is bit-identical to glibc 2.35's (checked against glibc on 8e7 inputs on the CPU, and by
// tests/test_gpu_parity.py::test_atan2f_bit_exact on the GPU).  The device math library's atan2f differs in the last
// ulp on ~1e-5 of the inputs, which would move a point into the neighbouring theta bin.
__device__ __forceinline__ float tdr_atanf(float x) {
  const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                        9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                        4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
  const int hx = __float_as_int(x), ix = hx & 0x7fffffff;
  float hi = 0.f, lo = 0.f;
  int id;
  if (ix >= 0x4c000000) {  // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {   // |x| < 0.4375
    if (ix < 0x31000000) return x;
    id = -1;
  } else {
    x = fabsf(x);
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) { id = 0; hi = atanhi[0]; lo = atanlo[0]; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else { id = 1; hi = atanhi[1]; lo = atanlo[1]; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { id = 2; hi = atanhi[2]; lo = atanlo[2]; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else { id = 3; hi = atanhi[3]; lo = atanlo[3]; x = -1.0f / x; }
    }
  }
  const float z = x * x, w = z * z;
  const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  const float r = hi - ((x * (s1 + s2) - lo) - x);
  return hx < 0 ? -r : r;
}
__device__ __forceinline__ float tdr_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
              pi_lo = -8.7422776573e-08f;
  const int hx = __float_as_int(x), ix = hx & 0x7fffffff, hy = __float_as_int(y), iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return tdr_atanf(y);
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : m == 1 ? -pi_o_4 - tiny : m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
    return m == 0 ? 0.0f : m == 1 ? -0.0f : m == 2 ? pi + tiny : -pi - tiny;
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = tdr_atanf(fabsf(y / x));
  switch (m) {
    case 0: return z;
    case 1: return __int_as_float(__float_as_int(z) ^ (int)0x80000000);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

__device__ __forceinline__ bool raster_bin(const RasterArgs& a, float x, float y, int& row, int& col) {
  if (x == 0.f && y == 0.f) return false;
  if (a.polar) {
    float theta = tdr_atan2f(x, y);  // glibc-exact, see above
    float r = sqrtf(x * x + y * y);
    row = (int)(roundf(theta / a.ang_res) + (float)(a.rows / 2));
    col = (int)roundf(r / a.res);
  } else {
    col = (int)(roundf(x / a.res) + (float)(a.cols / 2));
    row = (int)(roundf(y / a.res) + (float)(a.rows / 2));
  }
  return row >= 0 && row < a.rows && col >= 0 && col < a.cols;
}

// Phase 1 (when the caller gave a workspace): the bin of every point once — atan2f / sqrtf per point instead of per
// point and tile — as a 4-byte key the tiles then stream.
__global__ __launch_bounds__(256) void raster_keys_kernel(RasterArgs a) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.n) return;
  const float* p = a.pts + k * a.stride;
  float x, y, cf;
  if (a.stride == 4 && a.ioff == 3) {
    float4 v = *reinterpret_cast<const float4*>(p);
    x = v.x; y = v.y; cf = v.w;
  } else {
    x = p[0]; y = p[1]; cf = p[a.ioff];
  }
  uint32_t key = RASTER_NO_BIN;
  int row, col;
  if (raster_bin(a, x, y, row, col)) {
    const int pc = (int)cf;
    if (pc >= 0 && pc <= 255) {
      const int c = a.lut[pc];
      if (c >= 0 && c < a.ncls) key = ((uint32_t)col << 20) | ((uint32_t)c << 16) | (uint32_t)row;
    }
  }
  a.keys[k] = key;
}

__global__ __launch_bounds__(1024) void raster_kernel(RasterArgs a) {
  extern __shared__ unsigned int cnt[];  // [cpt][ncls][rows]
  const int col0 = blockIdx.x * a.cpt;
  const int ncol = min(a.cpt, a.cols - col0);
  const int tile = ncol * a.ncls * a.rows;
  for (int t = threadIdx.x; t < tile; t += blockDim.x) cnt[t] = 0;
  __shared__ int lut_s[256];
  if (threadIdx.x < 256) lut_s[threadIdx.x] = a.lut[threadIdx.x];
  __syncthreads();
  if (a.keys) {
    for (int64_t k = threadIdx.x; k < a.n; k += blockDim.x) {
      const uint32_t key = a.keys[k];
      const int col = (int)(key >> 20) - col0;
      if (key == RASTER_NO_BIN || col < 0 || col >= ncol) continue;
      atomicAdd(&cnt[(col * a.ncls + (int)((key >> 16) & 15u)) * a.rows + (int)(key & 0xFFFFu)], 1u);
    }
  } else
  for (int64_t k = threadIdx.x; k < a.n; k += blockDim.x) {
    const float* p = a.pts + k * a.stride;
    float x, y, cf;
    if (a.stride == 4 && a.ioff == 3) {
      float4 v = *reinterpret_cast<const float4*>(p);
      x = v.x; y = v.y; cf = v.w;
    } else {
      x = p[0]; y = p[1]; cf = p[a.ioff];
    }
    int row, col;
    if (!raster_bin(a, x, y, row, col)) continue;
    col -= col0;
    if (col < 0 || col >= ncol) continue;
    int pc = (int)cf;
    if (pc < 0 || pc > 255) continue;
    int c = lut_s[pc];
    if (c < 0 || c >= a.ncls) continue;
    atomicAdd(&cnt[(col * a.ncls + c) * a.rows + row], 1u);
  }
  __syncthreads();
  const int64_t P = (int64_t)a.rows * a.cols;
  if (a.img) {
    for (int t = threadIdx.x; t < tile; t += blockDim.x) {
      int row = t % a.rows, cc = t / a.rows;
      int c = cc % a.ncls, col = cc / a.ncls;
      a.img[(int64_t)c * P + row + (int64_t)a.rows * (col0 + col)] = (float)cnt[t];
    }
  }
  if (a.pk) {
    const int bins = ncol * a.rows;
    for (int t = threadIdx.x; t < bins; t += blockDim.x) {
      int row = t % a.rows, col = t / a.rows;
      float* o = a.pk + ((int64_t)(col0 + col) * a.rows + row) * a.rf;
      unsigned int tot = 0;
      for (int c = 0; c < a.ncls; c++) {
        unsigned int v = cnt[(col * a.ncls + c) * a.rows + row];
        o[c] = (float)v;
        tot += v;
      }
      for (int c = a.ncls; c < a.rf - 1; c++) o[c] = 0.f;
      if (tdr_has_kslot(a.ncls, a.rf)) o[a.rf - 2] = 1.f;
      o[a.rf - 1] = (float)tot;
    }
  }
}

extern "C" int64_t tdr_raster_workspace_bytes(int64_t n) { return n < 1 ? 0 : 4 * n; }
static int launch_raster(const float* pts, int stride, int ioff, int64_t n, float res, float ang_res,
                         const int32_t* lut, int ncls, int rows, int cols, int polar, float* img, float* pk,
                         void* workspace, void* stream) {
  if (ncls < 1 || ncls > TDR_MAX_CLASSES || rows < 1 || cols < 1) return fail(TDR_ERR_ARG, "raster: bad image shape");
  if (n < 0 || (n > 0 && !pts) || !lut) return fail(TDR_ERR_ARG, "raster: null points / lut");
  if (stride < 3 || ioff < 0 || ioff >= stride) return fail(TDR_ERR_ARG, "raster: bad point stride / offset");
  if (!(res > 0.f) || (polar && !(ang_res > 0.f))) return fail(TDR_ERR_ARG, "raster: resolution must be > 0");
  int64_t per_col = (int64_t)ncls * rows * 4;
  if (per_col > 64 * 1024) return fail(TDR_ERR_ARG, "raster: ncls*rows too large for one LDS tile");
  RasterArgs a;
  a.pts = pts; a.stride = stride; a.ioff = ioff; a.n = n; a.res = res; a.ang_res = ang_res; a.lut = lut;
  a.ncls = ncls; a.rows = rows; a.cols = cols; a.rf = tdr_rec_floats(ncls); a.polar = polar; a.img = img; a.pk = pk;
  a.cpt = (int)std::max<int64_t>(1, (64 * 1024) / per_col);
  a.cpt = std::min(a.cpt, cols);
  // enough workgroups to spread over the chip when the image is small
  while (a.cpt > 1 && cdiv(cols, a.cpt) < 32) a.cpt = (a.cpt + 1) / 2;
  size_t lds = (size_t)a.cpt * per_col;
  a.keys = nullptr;
  if (workspace && n > 0 && cols <= RASTER_KEY_MAX_COLS && rows <= RASTER_KEY_MAX_ROWS) {
    a.keys = reinterpret_cast<uint32_t*>(workspace);
    hipLaunchKernelGGL(raster_keys_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
  }
  hipLaunchKernelGGL(raster_kernel, dim3((unsigned)cdiv(cols, a.cpt)), dim3(1024), lds, (hipStream_t)stream, a);
  LAUNCH_CHECK("raster");
  return TDR_OK;
}

extern "C" int tdr_k_raster_polar(const float* pts, int stride, int ioff, int64_t n, float res, float ang_res,
                                  const int32_t* lut256, int ncls, int nb, int nr, float* img_out, float* pk_out,
                                  void* workspace, void* stream) {
  return launch_raster(pts, stride, ioff, n, res, ang_res, lut256, ncls, nb, nr, 1, img_out, pk_out, workspace, stream);
}
extern "C" int tdr_k_raster_cart(const float* pts, int stride, int ioff, int64_t n, float res, const int32_t* lut256,
                                 int ncls, int rows, int cols, float* img_out, float* pk_out, void* workspace,
                                 void* stream) {
  return launch_raster(pts, stride, ioff, n, res, 1.f, lut256, ncls, rows, cols, 0, img_out, pk_out, workspace, stream);
}

__global__ void pack_scan_kernel(const float* __restrict__ img, int ncls, int rows, int cols, int rf,
                                 float* __restrict__ pk) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t P = (int64_t)rows * cols;
  if (t >= P) return;
  float* o = pk + t * rf;  // t = row + rows*col == (col*rows + row)
  float tot = 0.f;
  for (int c = 0; c < ncls; c++) {
    float v = img[(int64_t)c * P + t];
    o[c] = v;
    tot += v;
  }
  for (int c = ncls; c < rf - 1; c++) o[c] = 0.f;
  if (tdr_has_kslot(ncls, rf)) o[rf - 2] = 1.f;
  o[rf - 1] = tot;
}
extern "C" int tdr_k_pack_scan(const float* img, int ncls, int nb, int nr, float* pk_out, void* stream) {
  if (!img || !pk_out || ncls < 1 || ncls > TDR_MAX_CLASSES || nb < 1 || nr < 1)
    return fail(TDR_ERR_ARG, "pack_scan: bad arguments");
  int64_t P = (int64_t)nb * nr;
  hipLaunchKernelGGL(pack_scan_kernel, dim3((unsigned)cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, img, ncls, nb,
                     nr, tdr_rec_floats(ncls), pk_out);
  LAUNCH_CHECK("pack_scan");
  return TDR_OK;
}

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
  const int32_t* count; // optional device count limiting the active slots (init search)
  int use_theta_override;
  float theta_override;
  int rpc, nchunks;     // rings per chunk
  int64_t npad;         // slots padded to a multiple of 64
  float* part;          // [nchunks][rf+1][npad]
};

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

template <int NV4, int U, bool KSLOT, bool USCALE>
__global__ __launch_bounds__(256) void score_polar_kernel(ScoreArgs a) {
  constexpr int RF = 4 * NV4;
  extern __shared__ float4 ring[];  // [NV4 planes][2*nb rows]: row r and r+nb hold scan row r (no wrap arithmetic)
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
  const int64_t nact = a.count ? (int64_t)*a.count : a.n;
  if ((int64_t)bx * 256 >= nact) return;  // whole workgroup idle (uniform)
  const bool valid = slot < nact;
  const int64_t p = a.order ? (int64_t)a.order[valid ? slot : 0] : (valid ? slot : 0);
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];  // state_particle.cpp:161
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];  // :162
  const float off0 = cy / a.resolution;  // top_down_map_polar.cpp:29
  const float off1 = cx / a.resolution;  // :30
  const float theta = a.use_theta_override ? a.theta_override : a.st[TDR_ST_THETA * a.cap + p];
  const int shift = rot_shift_dev(theta, a.nb);  // scan row paired with window row i is (i + shift) mod nb

  const int j0 = blockIdx.y * a.rpc, j1 = min(a.nr, j0 + a.rpc);
  const int rowstride = (a.cols + 2) * (RF * 4);            // bytes per guarded map row
  const int kbase = (a.cols + 3) * (RF * 4);                // byte offset of cell (0,0)
  const float rmaxf = (float)a.rows, cmaxf = (float)a.cols;
  const char* __restrict__ recb = reinterpret_cast<const char*>(a.rec);
  const float2* __restrict__ tab2 = reinterpret_cast<const float2*>(USCALE ? a.utab : a.tab);
  const float4* __restrict__ scan4 = reinterpret_cast<const float4*>(a.scan_pk);
  const int nb2 = 2 * a.nb;

  float acc2[RF];
#pragma unroll
  for (int k = 0; k < RF; k++) acc2[k] = 0.f;
  float known2 = 0.f;

  // USCALE: every particle has the same scale, so (tab*scale)*res was evaluated once per step into a.utab and is
  // wave-uniform here; otherwise it is evaluated per lane.  Identical float operations either way.
  auto cell_offset = [&](float2 t) -> unsigned {
    float p0, p1;
    if constexpr (USCALE) {
      p0 = t.x;
      p1 = t.y;
    } else {
      p0 = (t.x * scale) * a.res;  // top_down_map_polar.cpp:28
      p1 = (t.y * scale) * a.res;
    }
    p0 = p0 + off0;
    p1 = p1 + off1;
    // clamp into the guard ring, then round like `pts.round().cast<int>()` (:31)
    p0 = __builtin_amdgcn_fmed3f(p0, -1.f, rmaxf);
    p1 = __builtin_amdgcn_fmed3f(p1, -1.f, cmaxf);
    const int ri = round_half_away_clamped(p0), ci = round_half_away_clamped(p1);
#if TDR_OOB_ALIAS
    // every out-of-bounds sample reads the SAME guard record (always cache-resident) instead of a distinct one
    const bool inb = (unsigned)ri < (unsigned)a.rows && (unsigned)ci < (unsigned)a.cols;
    return inb ? (unsigned)(__mul24(ri, rowstride) + (ci * (RF * 4) + kbase)) : 0u;
#else
    return (unsigned)(__mul24(ri, rowstride) + (ci * (RF * 4) + kbase));  // v_mad_i32_i24 + v_lshl_add
#endif
  };

  for (int j = j0; j < j1; j++) {
    __syncthreads();
    for (int t = threadIdx.x; t < a.nb * NV4; t += 256) {
      const float4 v = scan4[(int64_t)j * a.nb * NV4 + t];
      const int row = t / NV4, pl = t - row * NV4;
      ring[pl * nb2 + row] = v;
      ring[pl * nb2 + row + a.nb] = v;
    }
    __syncthreads();
    float acc[RF];
#pragma unroll
    for (int k = 0; k < RF; k++) acc[k] = 0.f;
    float known = 0.f;
    const float2* trow = tab2 + (int64_t)j * a.nb;
    const float4* rl = ring + shift;
    int i = 0;
    // U samples per step: all addresses first, then all loads (map records + LDS scan records) in flight together,
    // then the FMAs — the wave keeps 2*U*NV4 16-byte loads outstanding instead of waiting per sample.
    for (; i + U <= a.nb; i += U) {
      unsigned boff[U];
#pragma unroll
      for (int u = 0; u < U; u++) boff[u] = cell_offset(trow[i + u]);
      float4 m[U][NV4], s[U][NV4];
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int v = 0; v < NV4; v++) m[u][v] = *reinterpret_cast<const float4*>(recb + boff[u] + 16 * v);
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int v = 0; v < NV4; v++) s[u][v] = rl[v * nb2 + i + u];
#pragma unroll
      for (int u = 0; u < U; u++) {
#pragma unroll
        for (int v = 0; v < NV4; v++) {
          acc[4 * v + 0] = __builtin_fmaf(s[u][v].x, m[u][v].x, acc[4 * v + 0]);
          acc[4 * v + 1] = __builtin_fmaf(s[u][v].y, m[u][v].y, acc[4 * v + 1]);
          acc[4 * v + 2] = __builtin_fmaf(s[u][v].z, m[u][v].z, acc[4 * v + 2]);
          acc[4 * v + 3] = __builtin_fmaf(s[u][v].w, m[u][v].w, acc[4 * v + 3]);
        }
        if (!KSLOT) known += m[u][NV4 - 1].w;
      }
    }
    for (; i < a.nb; i++) {  // remainder when nb is not a multiple of U
      const unsigned bo = cell_offset(trow[i]);
#pragma unroll
      for (int v = 0; v < NV4; v++) {
        const float4 m = *reinterpret_cast<const float4*>(recb + bo + 16 * v);
        const float4 s = rl[v * nb2 + i];
        acc[4 * v + 0] = __builtin_fmaf(s.x, m.x, acc[4 * v + 0]);
        acc[4 * v + 1] = __builtin_fmaf(s.y, m.y, acc[4 * v + 1]);
        acc[4 * v + 2] = __builtin_fmaf(s.z, m.z, acc[4 * v + 2]);
        acc[4 * v + 3] = __builtin_fmaf(s.w, m.w, acc[4 * v + 3]);
        if (!KSLOT && v == NV4 - 1) known += m.w;
      }
    }
#pragma unroll
    for (int k = 0; k < RF; k++) acc2[k] += acc[k];
    known2 += known;
  }
  if (slot < a.npad) {
    float* o = a.part + (int64_t)blockIdx.y * (RF + 1) * a.npad + slot;
#pragma unroll
    for (int k = 0; k < RF; k++) o[(int64_t)k * a.npad] = acc2[k];
    o[(int64_t)RF * a.npad] = KSLOT ? acc2[RF - 2] : known2;
  }
}

// K2c: Cartesian scoring (BASELINE config 4).  The reference's StateParticle never reaches the Cartesian
// TopDownMap::getLocalMap, so the Cartesian score is DEFINED as: window sampled by getLocalMap
// (src/top_down_map.cpp:429-459 via samplePts :367-389) at rot = theta, res = res*scale, scored by getCostForRot
// with shift 0 (src/state_particle.cpp:132-143) (definition recorded in include/tdr.h:tdr_k_score_cart and DESIGN.md).
// Same mapping as the polar kernel (lane = particle); the rotation now lives in the sampling, so the scan record of
// sample (i,j) is the same for every lane and comes through the scalar cache instead of LDS.
struct CartArgs {
  const float* rec;
  int map_rows, map_cols;
  float resolution;
  const float* scan_pk;  // [cols][rows][rf]
  int rows, cols;        // window (image) shape
  float res;
  const float* st;
  int64_t cap, n;
  const int32_t* order;
  int cpc, nchunks;      // window columns per chunk
  int64_t npad;
  float* part;
};

__device__ __forceinline__ float linspaced_dev(int i, int size1, float low, float high, float step) {
  // Eigen LinSpaced<float>, |high| == |low| here, so never the flipped branch of linspaced_op_impl
  return (i == size1) ? high : (low + (float)i * step);
}

template <int NV4, int U, bool KSLOT>
__global__ __launch_bounds__(256) void score_cart_kernel(CartArgs a) {
  constexpr int RF = 4 * NV4;
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
  // glibc's cosf/sinf are correctly rounded in practice; the device float versions are not -> evaluate in double
  const float c = (float)cos((double)theta), s = (float)sin((double)theta);
  const float ns = -s;
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
  const float4* __restrict__ scan4 = reinterpret_cast<const float4*>(a.scan_pk);

  float acc2[RF];
#pragma unroll
  for (int k = 0; k < RF; k++) acc2[k] = 0.f;
  float known2 = 0.f;

  for (int j = j0; j < j1; j++) {
    const float xj = linspaced_dev(j, c1, lo_c, hi_c, step_c);
    const float A = ns * xj, B = c * xj;  // rotm * pts (:383-385): q0 = c*y + (-s)*x, q1 = s*y + c*x
    auto cell_offset = [&](int i) -> unsigned {
      const float yi = linspaced_dev(i, r1, lo_r, hi_r, step_r);
      float p0 = c * yi + A;
      float p1 = s * yi + B;
      p0 = p0 + off0;
      p1 = p1 + off1;
      p0 = __builtin_amdgcn_fmed3f(p0, -1.f, rmaxf);
      p1 = __builtin_amdgcn_fmed3f(p1, -1.f, cmaxf);
      const int ri = round_half_away_clamped(p0), ci = round_half_away_clamped(p1);  // :437
      const bool inb = (unsigned)ri < (unsigned)a.map_rows && (unsigned)ci < (unsigned)a.map_cols;
      return inb ? (unsigned)(__mul24(ri, rowstride) + (ci * (RF * 4) + kbase)) : 0u;
    };
    float acc[RF];
#pragma unroll
    for (int k = 0; k < RF; k++) acc[k] = 0.f;
    float known = 0.f;
    const float4* srow = scan4 + (int64_t)j * a.rows * NV4;  // wave-uniform: scalar loads
    int i = 0;
    for (; i + U <= a.rows; i += U) {
      unsigned boff[U];
#pragma unroll
      for (int u = 0; u < U; u++) boff[u] = cell_offset(i + u);
      float4 m[U][NV4];
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int v = 0; v < NV4; v++) m[u][v] = *reinterpret_cast<const float4*>(recb + boff[u] + 16 * v);
#pragma unroll
      for (int u = 0; u < U; u++) {
#pragma unroll
        for (int v = 0; v < NV4; v++) {
          const float4 sv = srow[(i + u) * NV4 + v];
          acc[4 * v + 0] = __builtin_fmaf(sv.x, m[u][v].x, acc[4 * v + 0]);
          acc[4 * v + 1] = __builtin_fmaf(sv.y, m[u][v].y, acc[4 * v + 1]);
          acc[4 * v + 2] = __builtin_fmaf(sv.z, m[u][v].z, acc[4 * v + 2]);
          acc[4 * v + 3] = __builtin_fmaf(sv.w, m[u][v].w, acc[4 * v + 3]);
        }
        if (!KSLOT) known += m[u][NV4 - 1].w;
      }
    }
    for (; i < a.rows; i++) {
      const unsigned bo = cell_offset(i);
#pragma unroll
      for (int v = 0; v < NV4; v++) {
        const float4 m = *reinterpret_cast<const float4*>(recb + bo + 16 * v);
        const float4 sv = srow[i * NV4 + v];
        acc[4 * v + 0] = __builtin_fmaf(sv.x, m.x, acc[4 * v + 0]);
        acc[4 * v + 1] = __builtin_fmaf(sv.y, m.y, acc[4 * v + 1]);
        acc[4 * v + 2] = __builtin_fmaf(sv.z, m.z, acc[4 * v + 2]);
        acc[4 * v + 3] = __builtin_fmaf(sv.w, m.w, acc[4 * v + 3]);
        if (!KSLOT && v == NV4 - 1) known += m.w;
      }
    }
#pragma unroll
    for (int k = 0; k < RF; k++) acc2[k] += acc[k];
    known2 += known;
  }
  if (slot < a.npad) {
    float* o = a.part + (int64_t)blockIdx.y * (RF + 1) * a.npad + slot;
#pragma unroll
    for (int k = 0; k < RF; k++) o[(int64_t)k * a.npad] = acc2[k];
    o[(int64_t)RF * a.npad] = KSLOT ? acc2[RF - 2] : known2;
  }
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
};

__global__ __launch_bounds__(256) void score_finalize_kernel(FinalizeArgs a) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nact = a.count ? (int64_t)*a.count : a.n;
  if (slot >= nact) return;
  const int64_t p = a.order ? (int64_t)a.order[slot] : slot;
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];
  if (a.mode == 0 && particle_gated(a.gate, cx, cy, scale)) {
    a.raw_w[p] = 0.f;
    return;
  }
  // Per-chunk partial sums -> double totals, chunk order ascending for every slot.  The loads of FIN_B chunks x all
  // slots are issued together (independent addresses, coalesced over the particles) before the dependent additions.
  constexpr int FIN_B = 4, FIN_S = TDR_MAX_CLASSES + 2;   // slots: ncls class dots, normalisation, known count
  double tot[FIN_S];
#pragma unroll
  for (int k = 0; k < FIN_S; k++) tot[k] = 0;
  const int64_t cstride = (int64_t)(a.rf + 1) * a.npad;
  auto slot_row = [&](int k) { return k < a.ncls ? k : (k == a.ncls ? a.rf - 1 : a.rf); };
  int c0 = 0;
  for (; c0 + FIN_B <= a.nchunks; c0 += FIN_B) {
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
  for (; c0 < a.nchunks; c0++) {
#pragma unroll
    for (int k = 0; k < FIN_S; k++)
      if (k < a.ncls + 2) tot[k] += (double)a.part[(int64_t)c0 * cstride + (int64_t)slot_row(k) * a.npad + slot];
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
    cost = cost / (float)norm;  // :154
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
template <bool USCALE, bool UNITW>
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
  float wc[6];
#pragma unroll
  for (int c = 0; c < 6; c++) wc[c] = c < a.ncls ? (float)(0.01 * (double)a.fp.class_weights[c]) : 0.f;
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
             v1.w > 2048.f;
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
      float v[6] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y};
      if constexpr (!UNITW) {
#pragma unroll
        for (int c = 0; c < 6; c++) v[c] *= wc[c];
      }
      union { tdr_h2 h[4]; tdr_h8 v8; } bh, bl, bn;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const tdr_h2 hi = __builtin_amdgcn_cvt_pkrtz(v[2 * c], v[2 * c + 1]);
        bh.h[c] = hi;
        bl.h[c] = __builtin_amdgcn_cvt_pkrtz(v[2 * c] - (float)hi[0], v[2 * c + 1] - (float)hi[1]);
      }
      bh.h[3] = __builtin_amdgcn_cvt_pkrtz(0.f, 0.f);
      bl.h[3] = bh.h[3];
      bn.h[0] = bh.h[3]; bn.h[1] = bh.h[3]; bn.h[2] = bh.h[3];
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

static bool init_use_mfma() {
  static bool v = [] {
    const char* e = getenv("TDR_INIT_MFMA");   // 0 = vector-unit search only (A/B and debugging)
    return !(e && atoi(e) == 0);
  }();
  return v;
}
static int64_t score_wave_target() {
  static int64_t v = [] {
    // tuning knob.  Many short waves beat few long ones (A/B on MI355X, config 2: 16k waves 21.8 ms, 128k 15.2 ms):
    // workgroups of one ring chunk run together, so the concurrently touched part of the map is a thin annulus that
    // L2 can hold, and the slow (scattered) batches no longer leave a long tail.
    const char* e = getenv("TDR_SCORE_WAVES");
    long t = e ? atol(e) : 0;
    return (int64_t)(t > 0 ? t : 131072);
  }();
  return v;
}
static void choose_chunks(int64_t n, int nr, int& rpc, int& nchunks) {
  int64_t nbatches = cdiv(std::max<int64_t>(n, 1), 64);
  int64_t want = std::max<int64_t>(1, cdiv(score_wave_target(), nbatches));  // enough waves to fill the chip
  nchunks = (int)std::min<int64_t>(nr, want);
  rpc = (int)cdiv(nr, nchunks);
  nchunks = (int)cdiv(nr, rpc);
}

__global__ void utab_kernel(const float* __restrict__ tab, int64_t n2, float scale, float res,
                            float* __restrict__ utab) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n2) utab[k] = (tab[k] * scale) * res;  // `ang_sample_pts_*scale*res` (top_down_map_polar.cpp:28)
}

extern "C" size_t tdr_score_workspace_floats(int ncls, int nb, int nr, int64_t n) {
  int rpc, nchunks;
  choose_chunks(n, nr, rpc, nchunks);
  int64_t npad = cdiv(std::max<int64_t>(n, 1), 64) * 64;
  int rf = tdr_rec_floats(ncls);
  // partials + best_cost + best_theta + list + count(64) + uniform-scale table
  return (size_t)((int64_t)nchunks * (rf + 1) * npad + 3 * npad + 64 + 2 * (int64_t)nb * nr);
}
static float* ws_utab(float* workspace, int rf, int nchunks, int64_t npad) {
  return workspace + (int64_t)nchunks * (rf + 1) * npad + 3 * npad + 64;
}
static int fill_utab(ScoreArgs& a, float* workspace, int rf, float uniform_scale, hipStream_t s) {
  a.utab = nullptr;
  if (!(uniform_scale > 0.f)) return TDR_OK;
  float* ut = ws_utab(workspace, rf, a.nchunks, a.npad);
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
  explicit ScoreProfScope(hipStream_t s_) : s(s_) {
    if (!g_prof_on) return;
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
extern "C" int tdr_profile_enable(int on) {
  g_prof_on = on != 0;
  g_prof_used = 0;
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

static int launch_score(const ScoreArgs& a, int rf, int ncls, hipStream_t s) {
  dim3 grid((unsigned)cdiv(a.n, 256), (unsigned)a.nchunks), block(256);
  size_t lds = (size_t)2 * a.nb * rf * 4;
  const bool ks = tdr_has_kslot(ncls, rf);
  ScoreProfScope prof(s);
  const bool us = a.utab != nullptr;
#define TDR_LAUNCH_SCORE(NV4)                                                                                   \
  if (ks && us) hipLaunchKernelGGL((score_polar_kernel<NV4, TDR_SCORE_U, true, true>), grid, block, lds, s, a);   \
  else if (ks) hipLaunchKernelGGL((score_polar_kernel<NV4, TDR_SCORE_U, true, false>), grid, block, lds, s, a);   \
  else if (us) hipLaunchKernelGGL((score_polar_kernel<NV4, TDR_SCORE_U, false, true>), grid, block, lds, s, a);   \
  else hipLaunchKernelGGL((score_polar_kernel<NV4, TDR_SCORE_U, false, false>), grid, block, lds, s, a);
  switch (rf / 4) {
    case 1: TDR_LAUNCH_SCORE(1) break;
    case 2: TDR_LAUNCH_SCORE(2) break;
    case 3: TDR_LAUNCH_SCORE(3) break;
    case 4: TDR_LAUNCH_SCORE(4) break;
    default: return fail(TDR_ERR_ARG, "score: unsupported record size %d", rf);
  }
#undef TDR_LAUNCH_SCORE
  LAUNCH_CHECK("score_polar");
  return TDR_OK;
}

extern "C" int tdr_k_score_polar(const tdr_map_desc* map, const float* tab, const float* scan_pk, int nb, int nr,
                                 float res, const tdr_filter_params* fp, float* st, int64_t cap, int64_t n,
                                 const int32_t* perm, float uniform_scale, int init_search, float* raw_w,
                                 float* workspace, void* stream) {
  if (!map || !map->rec || !tab || !scan_pk || !fp || !st || !raw_w || !workspace)
    return fail(TDR_ERR_ARG, "score: null pointer");
  if (n < 0 || cap < n) return fail(TDR_ERR_ARG, "score: n=%lld exceeds capacity %lld", (long long)n, (long long)cap);
  if (n == 0) return TDR_OK;
  if (nb < 1 || nr < 1) return fail(TDR_ERR_ARG, "score: bad image shape");
  if (map->ncls < 1 || map->ncls > TDR_MAX_CLASSES || fp->num_classes != map->ncls)
    return fail(TDR_ERR_ARG, "score: class count mismatch (map %d, params %d)", map->ncls, fp->num_classes);
  const int rf = tdr_rec_floats(map->ncls);
  if (map->rec_floats != rf) return fail(TDR_ERR_ARG, "score: map record size %d != %d", map->rec_floats, rf);
  if ((size_t)2 * nb * rf * 4 > 64 * 1024) return fail(TDR_ERR_ARG, "score: nb too large for the LDS scan ring");
  if (!(map->resolution > 0.f)) return fail(TDR_ERR_ARG, "score: map resolution must be > 0");
  hipStream_t s = (hipStream_t)stream;

  ScoreArgs a;
  a.rec = map->rec; a.rows = map->rows; a.cols = map->cols; a.resolution = map->resolution;
  a.tab = tab; a.scan_pk = scan_pk; a.nb = nb; a.nr = nr; a.res = res;
  a.st = st; a.cap = cap; a.n = n; a.order = perm; a.count = nullptr;
  a.use_theta_override = 0; a.theta_override = 0.f;
  choose_chunks(n, nr, a.rpc, a.nchunks);
  a.npad = cdiv(n, 64) * 64;
  a.part = workspace;
  int rc = fill_utab(a, workspace, rf, uniform_scale, s);
  if (rc) return rc;
  float* res_flag = workspace + (int64_t)a.nchunks * (rf + 1) * a.npad;  // npad floats
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
    if (rf == 8 && ks && init_use_mfma()) {
      // matrix-core pass first; the vector kernel below then runs only if a scan count did not fit f16
      int* d_inexact = d_nrot + 1;
      const size_t lds16 = ((size_t)2 * nb + 1) * 16;
      bool unitw = true;
      for (int c = 1; c < map->ncls; c++) unitw &= fp->class_weights[c] == fp->class_weights[0];
      unitw &= fp->class_weights[0] > 0.f;
      if (us && unitw) hipLaunchKernelGGL((score_init_mfma_kernel<true, true>), grid, dim3(256), lds16, s, ia, d_inexact);
      else if (us) hipLaunchKernelGGL((score_init_mfma_kernel<true, false>), grid, dim3(256), lds16, s, ia, d_inexact);
      else if (unitw) hipLaunchKernelGGL((score_init_mfma_kernel<false, true>), grid, dim3(256), lds16, s, ia, d_inexact);
      else hipLaunchKernelGGL((score_init_mfma_kernel<false, false>), grid, dim3(256), lds16, s, ia, d_inexact);
      LAUNCH_CHECK("score_init_mfma");
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
      default: return fail(TDR_ERR_ARG, "score: init search supports up to 7 classes (record of %d floats)", rf);
    }
#undef TDR_LAUNCH_INIT
    LAUNCH_CHECK("score_init");
    hipLaunchKernelGGL(init_apply_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, (const float*)res_theta,
                       (const float*)res_flag, n, st, cap);
    LAUNCH_CHECK("init_apply");
  }
  rc = launch_score(a, rf, map->ncls, s);
  if (rc) return rc;

  FinalizeArgs f;
  f.part = a.part; f.rf = rf; f.nchunks = a.nchunks; f.npad = a.npad; f.n = n; f.cap = cap;
  f.order = perm; f.count = nullptr; f.st = st; f.fp = *fp;
  f.gate = make_gate(fp, map);
  f.P = (int64_t)nb * nr; f.ncls = map->ncls; f.mode = 0; f.first = 0; f.theta_override = 0.f;
  f.raw_w = raw_w; f.best_cost = nullptr; f.best_theta = nullptr;
  hipLaunchKernelGGL(score_finalize_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, f);
  LAUNCH_CHECK("score_finalize");
  if (init_search) {
    hipLaunchKernelGGL(init_fixup_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, (const float*)res_flag, n,
                       fp->regularization, raw_w);
    LAUNCH_CHECK("init_fixup");
  }
  return TDR_OK;
}

extern "C" size_t tdr_score_cart_workspace_floats(int ncls, int rows, int cols, int64_t n) {
  (void)rows;
  int cpc, nchunks;
  choose_chunks(n, cols, cpc, nchunks);
  int64_t npad = cdiv(std::max<int64_t>(n, 1), 64) * 64;
  return (size_t)((int64_t)nchunks * (tdr_rec_floats(ncls) + 1) * npad + 64);
}

extern "C" int tdr_k_score_cart(const tdr_map_desc* map, const float* scan_pk, int rows, int cols, float res,
                                const tdr_filter_params* fp, float* st, int64_t cap, int64_t n, const int32_t* perm,
                                float* raw_w, float* workspace, void* stream) {
  if (!map || !map->rec || !scan_pk || !fp || !st || !raw_w || !workspace)
    return fail(TDR_ERR_ARG, "score_cart: null pointer");
  if (n < 0 || cap < n) return fail(TDR_ERR_ARG, "score_cart: n exceeds capacity");
  if (n == 0) return TDR_OK;
  if (rows < 1 || cols < 1) return fail(TDR_ERR_ARG, "score_cart: bad window shape");
  if (fp->num_classes != map->ncls) return fail(TDR_ERR_ARG, "score_cart: class count mismatch");
  const int rf = tdr_rec_floats(map->ncls);
  if (map->rec_floats != rf) return fail(TDR_ERR_ARG, "score_cart: map record size mismatch");
  hipStream_t s = (hipStream_t)stream;
  CartArgs a;
  a.rec = map->rec; a.map_rows = map->rows; a.map_cols = map->cols; a.resolution = map->resolution;
  a.scan_pk = scan_pk; a.rows = rows; a.cols = cols; a.res = res;
  a.st = st; a.cap = cap; a.n = n; a.order = perm;
  choose_chunks(n, cols, a.cpc, a.nchunks);
  a.npad = cdiv(n, 64) * 64;
  a.part = workspace;
  dim3 grid((unsigned)cdiv(n, 256), (unsigned)a.nchunks), block(256);
  const bool ks = tdr_has_kslot(map->ncls, rf);
  {
    ScoreProfScope prof(s);
#define TDR_LAUNCH_CART(NV4)                                                                        \
  if (ks) hipLaunchKernelGGL((score_cart_kernel<NV4, TDR_SCORE_U, true>), grid, block, 0, s, a);    \
  else hipLaunchKernelGGL((score_cart_kernel<NV4, TDR_SCORE_U, false>), grid, block, 0, s, a);
    switch (rf / 4) {
      case 1: TDR_LAUNCH_CART(1) break;
      case 2: TDR_LAUNCH_CART(2) break;
      case 3: TDR_LAUNCH_CART(3) break;
      case 4: TDR_LAUNCH_CART(4) break;
      default: return fail(TDR_ERR_ARG, "score_cart: unsupported record size %d", rf);
    }
#undef TDR_LAUNCH_CART
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
  hipLaunchKernelGGL(score_finalize_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, f);
  LAUNCH_CHECK("score_finalize(cart)");
  return TDR_OK;
}

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
                                 const float* __restrict__ z4, uint64_t seed, uint64_t step, int64_t index_base) {
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
  // Rotation2D<float>(theta) * trans; sin/cos evaluated in double and rounded (glibc's sinf/cosf are correctly
  // rounded in practice, the device float versions are not)
  const float c = (float)cos((double)theta), s = (float)sin((double)theta);
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
                     last_dist, tx, ty, omega, scale_freeze, pos_cov, theta_cov, z4, seed, step, index_base);
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
// K4: weight statistics (particle_filter.cpp:107-147).  One 1024-thread workgroup; sums in double with a fixed
// strided/tree order, so the result depends only on (raw_w, last_dist, n) — identical on every rank that holds the
// all-gathered weights.
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

__global__ __launch_bounds__(1024) void update_weights_kernel(const float* __restrict__ raw,
                                                             const float* __restrict__ last_dist, int64_t n,
                                                             float* __restrict__ w, float* __restrict__ info) {
  __shared__ double shd[16];
  __shared__ long long shl[16];
  __shared__ float sh_best[16];
  __shared__ long long sh_besti[16];
  const int tid = threadIdx.x, nt = blockDim.x;
  // :108-116
  double s = 0;
  long long cnt = 0;
  for (int64_t i = tid; i < n; i += nt) {
    float v = raw[i];
    if (!isnan(v)) { s += (double)v; cnt++; }
  }
  const float sum = (float)block_sum_d(s, shd);
  const long long num_valid = block_sum_ll(cnt, shl);
  const float mean = sum / (float)num_valid;  // :117 (0/0 -> NaN like the reference)
  // :118-126
  double bs = 0;
  long long cu = 0;
  for (int64_t i = tid; i < n; i += nt) {
    float v = raw[i];
    if (!isnan(v) && v < mean) {
      double d = (double)(v - mean);
      bs += d * d;
      cu++;
    }
  }
  const float bsum = (float)block_sum_d(bs, shd);
  const long long num_under = block_sum_ll(cu, shl);
  const float bottom = sqrtf(bsum / (float)num_under);
  const bool fallback = (sum == 0.f || num_under < 1);  // :129
  const float fill = mean - bottom;                      // :133
  double s1 = 0;
  for (int64_t i = tid; i < n; i += nt) {
    float v = raw[i];
    v = fallback ? 1.f : (isnan(v) ? fill : v);
    w[i] = v;
    s1 += (double)v;
  }
  const float fs1 = (float)block_sum_d(s1, shd);
  const float invn_den = (float)n;
  double s2 = 0;
  for (int64_t i = tid; i < n; i += nt) {  // :135, :138-141
    float v = w[i] / fs1;
    float d = fminf(last_dist[i] * 5.f, 1.f);
    v = d * v + (1.f - d) / invn_den;
    w[i] = v;
    s2 += (double)v;
  }
  const float fs2 = (float)block_sum_d(s2, shd);
  float best = -INFINITY;
  long long besti = 0x7fffffffffffffffll;
  for (int64_t i = tid; i < n; i += nt) {  // :142, :145-147 (first maximum)
    float v = w[i] / fs2;
    w[i] = v;
    if (v > best || (v == best && i < besti)) { best = v; besti = i; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    float ob = __shfl_down(best, o, 64);
    long long oi = __shfl_down(besti, o, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  __syncthreads();
  if ((tid & 63) == 0) { sh_best[tid >> 6] = best; sh_besti[tid >> 6] = besti; }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < (nt >> 6); k++)
      if (sh_best[k] > best || (sh_best[k] == best && sh_besti[k] < besti)) { best = sh_best[k]; besti = sh_besti[k]; }
    if (besti == 0x7fffffffffffffffll) besti = 0;
    info[0] = __int_as_float((int)besti);
    info[1] = sum; info[2] = mean; info[3] = bottom; info[4] = fallback ? 1.f : 0.f;
    info[5] = (float)num_valid; info[6] = (float)num_under; info[7] = 0.f;
  }
}

// Multi-workgroup form of the same statistics for large n: five grid-wide passes, each workgroup reducing its own
// contiguous chunk in a fixed order and every workgroup re-reducing the G per-workgroup partials in index order, so
// the result is again a pure function of (raw_w, last_dist, n) — identical on every rank — without grid barriers.
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
// pass 2: squared deviations of the weights below the mean (:118-125)
__global__ __launch_bounds__(256) void uw_pass2(const float* __restrict__ raw, int64_t n, const UwScratch* s1,
                                                UwScratch* s2) {
  __shared__ double shd[4];
  __shared__ long long shl[4];
  const float sum = (float)uw_total(s1->a);
  const float mean = sum / (float)(long long)uw_total(s1->b);
  int64_t lo, hi;
  uw_chunk(n, lo, hi);
  double bs = 0;
  long long cu = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = raw[i];
    if (!isnan(v) && v < mean) {
      const double d = (double)(v - mean);
      bs += d * d;
      cu++;
    }
  }
  const double tb = block_sum_d(bs, shd);
  const long long tc = block_sum_ll(cu, shl);
  if (threadIdx.x == 0) { s2->a[blockIdx.x] = tb; s2->b[blockIdx.x] = (double)tc; }
}
// pass 3: NaN fill / all-ones fallback (:129-134) and the first normalisation sum
__global__ __launch_bounds__(256) void uw_pass3(const float* __restrict__ raw, int64_t n, const UwScratch* s1,
                                                const UwScratch* s2, UwScratch* s3, float* __restrict__ w) {
  __shared__ double shd[4];
  const float sum = (float)uw_total(s1->a);
  const float mean = sum / (float)(long long)uw_total(s1->b);
  const long long num_under = (long long)uw_total(s2->b);
  const float bottom = sqrtf((float)uw_total(s2->a) / (float)num_under);
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
                                                const UwScratch* s5, float* info) {
  __shared__ float sb[UW_G];
  __shared__ long long si[UW_G];
  for (int g = threadIdx.x; g < UW_G; g += blockDim.x) { sb[g] = (float)s5->a[g]; si[g] = (long long)s5->b[g]; }
  const float sum = (float)uw_total(s1->a);
  const long long nv = (long long)uw_total(s1->b), nu = (long long)uw_total(s2->b);
  const float mean = sum / (float)nv;
  const float bottom = sqrtf((float)uw_total(s2->a) / (float)nu);
  if (threadIdx.x != 0) return;
  float best = -INFINITY;
  long long besti = 0x7fffffffffffffffll;
  for (int g = 0; g < UW_G; g++)
    if (sb[g] > best || (sb[g] == best && si[g] < besti)) { best = sb[g]; besti = si[g]; }
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
    hipLaunchKernelGGL(update_weights_kernel, dim3(1), dim3(1024), 0, s, raw_w, last_dist, n, w_out, info_out);
    LAUNCH_CHECK("update_weights");
    return TDR_OK;
  }
  static_assert(8 * sizeof(float) + 5 * sizeof(UwScratch) + 64 <= TDR_UW_INFO_FLOATS * sizeof(float), "info scratch");
  UwScratch* sc = reinterpret_cast<UwScratch*>(
      (reinterpret_cast<uintptr_t>(info_out + 8) + 63) & ~(uintptr_t)63);
  hipLaunchKernelGGL(uw_pass1, dim3(UW_G), dim3(256), 0, s, raw_w, n, sc + 0);
  hipLaunchKernelGGL(uw_pass2, dim3(UW_G), dim3(256), 0, s, raw_w, n, (const UwScratch*)(sc + 0), sc + 1);
  hipLaunchKernelGGL(uw_pass3, dim3(UW_G), dim3(256), 0, s, raw_w, n, (const UwScratch*)(sc + 0),
                     (const UwScratch*)(sc + 1), sc + 2, w_out);
  hipLaunchKernelGGL(uw_pass4, dim3(UW_G), dim3(256), 0, s, last_dist, n, (const UwScratch*)(sc + 2), sc + 3, w_out);
  hipLaunchKernelGGL(uw_pass5, dim3(UW_G), dim3(256), 0, s, n, (const UwScratch*)(sc + 3), sc + 4, w_out);
  hipLaunchKernelGGL(uw_pass6, dim3(1), dim3(256), 0, s, n, (const UwScratch*)(sc + 0), (const UwScratch*)(sc + 1),
                     (const UwScratch*)(sc + 4), info_out);
  LAUNCH_CHECK("update_weights(multi)");
  return TDR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// K5: resample.  The running sum of particle_filter.cpp:179 is a serial float32 chain,
//     prefix_j = fl(prefix_{j-1} + w_j),
// and "bit-exact resample indices" needs exactly these values.  Two kernels produce them:
//
//  * prefix_serial_kernel — one wave, lane 0 performs the additions in index order (weights staged through LDS).
//    Simple, ~10 ns per element; kept as the reference implementation and for small n.
//  * prefix_exact_kernel — the same values computed in parallel.  While the running sum r stays inside one binade
//    [2^e, 2^(e+1)) it is an integer multiple R*u of u = 2^(e-23) and fl(r + w) = (R + q)*u, where q is w/u rounded
//    to nearest — a pure integer increment that depends on r only when w/u ends in exactly .5 (tie to even: the parity
//    of R + floor(w/u)).  So per tile of 4096 weights the workgroup (i) classifies every weight into an integer
//    increment / tie / "needs a real float add" (NaN, inf, larger than the binade), (ii) prefix-sums the increments,
//    (iii) resolves the (rare) ties in order on one thread, (iv) prefix-sums the tie corrections, (v) finds the first
//    element at which the sum leaves the binade or a real add is needed, commits everything before it, performs that
//    one addition in float arithmetic and restarts behind it in the new binade.  A running sum of 1 crosses ~24
//    binades, so a million weights take a few hundred workgroup passes instead of a million dependent additions.
//    tests/test_gpu_parity.py compares both kernels with the CPU chain bit for bit on random and adversarial inputs.
//
// The running maximum makes "first j with prefix_j > sample" searchable even when weights are negative (NaN fill, :133).
#define TDR_PFX_BLOCK 4096  // elements staged in LDS per pass (64 per lane)
__global__ __launch_bounds__(64) void prefix_serial_kernel(const float* __restrict__ w, int64_t n,
                                                           float* __restrict__ runmax) {
  __shared__ float4 buf4[TDR_PFX_BLOCK / 4];
  float* buf = reinterpret_cast<float*>(buf4);
  const int lane = threadIdx.x;
  float run = 0.f;        // the serial chain lives in lane 0
  float carry_max = -INFINITY;
  for (int64_t base = 0; base < n; base += TDR_PFX_BLOCK) {
    const int cnt = (int)min((int64_t)TDR_PFX_BLOCK, n - base);
    // coalesced stage-in (pad with zeros: x + 0 == x)
    for (int t = lane; t < TDR_PFX_BLOCK; t += 64) buf[t] = (t < cnt) ? w[base + t] : 0.f;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      const int q4 = (cnt + 3) / 4;
#pragma unroll 4
      for (int q = 0; q < q4; q++) {
        float4 v = buf4[q];
        run = run + v.x; v.x = run;   // particle_filter.cpp:179, one float add per weight, index order
        run = run + v.y; v.y = run;
        run = run + v.z; v.z = run;
        run = run + v.w; v.w = run;
        buf4[q] = v;
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // running maximum of the block's prefix values (max is associative: any order is exact)
    float m = -INFINITY;
    float loc[64];
#pragma unroll
    for (int t = 0; t < 64; t++) {
      float x = buf[lane * 64 + t];
      if (x != x) x = -INFINITY;  // a NaN prefix never exceeds a threshold (`running_sum > sample` is false)
      m = fmaxf(m, x);
      loc[t] = m;
    }
    float incl = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      float t = __shfl_up(incl, o, 64);
      if (lane >= o) incl = fmaxf(incl, t);
    }
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = -INFINITY;
    excl = fmaxf(excl, carry_max);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 64; t++) buf[lane * 64 + t] = fmaxf(loc[t], excl);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < cnt; t += 64) runmax[base + t] = buf[t];
    carry_max = fmaxf(__shfl(incl, 63, 64), carry_max);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- exact parallel prefix -------------------------------------------------------------------------------------------
#define PFX_THREADS 1024
#define PFX_K 8
#define PFX_TILE (PFX_THREADS * PFX_K)
#define PFX_TIE_CAP 2048   // ties resolved per pass; a (never observed) denser tile is simply cut at that tie
#define PFX_HEAD 2048      // leading elements added one by one: the running sum crosses most of its binades here

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global store
// (~1 us each time); in the prefix kernels the threads exchange data through LDS alone — global memory is read-only
// input (w, chunk headers of an earlier launch) or write-only output — so the store wait would be pure latency.
__device__ __forceinline__ void pfx_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// inclusive block scan (sum) of one value per thread; `sh` holds one slot per wave
template <int NT, class T>
__device__ __forceinline__ T pfx_block_scan(T v, T* sh, T& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  pfx_sync();
  if (lane == 63) sh[wave] = v;
  pfx_sync();
  T off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < NT / 64; k++) {
    const T x = sh[k];
    if (k < wave) off += x;
    tot += x;
  }
  total = tot;
  return v + off;
}
template <int NT>
__device__ __forceinline__ float pfx_block_scan_max(float v, float* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (lane >= o) v = fmaxf(v, t);
  }
  pfx_sync();
  if (lane == 63) sh[wave] = v;
  pfx_sync();
  float off = -INFINITY;
  for (int k = 0; k < wave; k++) off = fmaxf(off, sh[k]);
  return fmaxf(v, off);
}

// The running sum carried through [lo, hi) in index order by the whole workgroup (NT threads): r_io / carry_io
// are the (workgroup-uniform) running sum and running maximum before element lo on entry and after element hi-1 on
// exit.  With lo == 0 the first PFX_HEAD elements are added one by one.
template <int NT>
__device__ __forceinline__ void pfx_exact_range(const float* __restrict__ w, long long lo, long long hi,
                                             float* __restrict__ runmax, float* __restrict__ prefix_opt,
                                             float& r_io, float& carry_io, int head_len = PFX_HEAD) {
  const long long n = hi;
  __shared__ long long sh_ll[NT / 64];
  __shared__ int sh_i[NT / 64];
  __shared__ float sh_f[NT / 64];
  __shared__ int tie_pos[PFX_TIE_CAP];         // local element index of every tie in order, weight sign in bit 31
  __shared__ long long tie_sval[PFX_TIE_CAP];  // inclusive increment prefix at the tie
  __shared__ signed char tie_corr[PFX_TIE_CAP];
  __shared__ float head[PFX_HEAD];
  __shared__ float last_val[NT], last_max[NT];
  __shared__ int s_first_bad, s_first_cross, s_first_nz;
  __shared__ float s_r, s_carry;
  __shared__ long long s_base;
  const int tid = threadIdx.x;
  pfx_sync();
  if (tid == 0) { s_r = r_io; s_carry = carry_io; s_base = lo; }
  pfx_sync();
  if (lo == 0) {  // head: plain serial additions by one thread out of LDS
    const int hn = (int)min((long long)head_len, (long long)n);
    for (int t = tid; t < hn; t += NT) head[t] = w[t];
    pfx_sync();
    if (tid == 0) {
      float run = 0.f, mx = -INFINITY;
      for (int t = 0; t < hn; t++) {
        run = run + head[t];  // particle_filter.cpp:179
        if (prefix_opt) prefix_opt[t] = run;
        if (run == run) mx = fmaxf(mx, run);
        head[t] = mx;
      }
      s_r = run; s_carry = mx; s_base = hn;
    }
    pfx_sync();
    for (int t = tid; t < hn; t += NT) runmax[t] = head[t];
    pfx_sync();
  }

  while (true) {
    const long long base = s_base;
    if (base >= n) break;
    const float r = s_r;
    const float carry = s_carry;
    const int cnt = (int)min((long long)(NT * PFX_K), (long long)n - base);
    const unsigned rb = __float_as_uint(r);
    const int re = (rb >> 23) & 0xFF;
    float wv[PFX_K];
#pragma unroll
    for (int k = 0; k < PFX_K; k++) {
      const int li = tid * PFX_K + k;
      wv[k] = (li < cnt) ? w[base + li] : 0.f;
    }
    if (tid == 0) { s_first_bad = (NT * PFX_K); s_first_cross = (NT * PFX_K); s_first_nz = (NT * PFX_K); }
    pfx_sync();  // also: everyone has read s_r / s_carry / s_base

    if (r != r) {  // NaN running sum: every later prefix is NaN, the running maximum stays
#pragma unroll
      for (int k = 0; k < PFX_K; k++) {
        const int li = tid * PFX_K + k;
        if (li < cnt) {
          runmax[base + li] = carry;
          if (prefix_opt) prefix_opt[base + li] = r;
        }
      }
      pfx_sync();
      if (tid == 0) s_base = base + cnt;
      pfx_sync();
      continue;
    }
    const bool r_zero = (rb & 0x7FFFFFFFu) == 0;
    const bool r_slow = (rb >> 31) != 0 || re == 0 || re == 255;  // negative, zero / subnormal, inf: plain float steps
    if (r_slow) {
      int stop = 0;  // leading elements that leave r unchanged (r == +-0 only: skip the run of zero weights)
      if (r_zero) {
#pragma unroll
        for (int k = 0; k < PFX_K; k++) {
          const int li = tid * PFX_K + k;
          if (li < cnt && (__float_as_uint(wv[k]) & 0x7FFFFFFFu) != 0) atomicMin(&s_first_nz, li);
        }
        pfx_sync();
        stop = min(s_first_nz, cnt);
      }
      const float m0 = fmaxf(carry, r);
#pragma unroll
      for (int k = 0; k < PFX_K; k++) {
        const int li = tid * PFX_K + k;
        if (li < stop) {
          runmax[base + li] = m0;
          if (prefix_opt) prefix_opt[base + li] = r;
        }
      }
      pfx_sync();
      if (tid == 0) {
        float nr = r, nc = stop > 0 ? m0 : carry;
        long long nb = base + stop;
        if (stop < cnt) {  // one real float addition (particle_filter.cpp:179)
          nr = r + w[base + stop];
          if (nr == nr) nc = fmaxf(nc, nr);
          runmax[base + stop] = nc;
          if (prefix_opt) prefix_opt[base + stop] = nr;
          nb = base + stop + 1;
        }
        s_r = nr; s_carry = nc; s_base = nb;
      }
      pfx_sync();
      continue;
    }

    // ---- r is a positive normal float: r = R * 2^(e-23), R in [2^23, 2^24)
    const int e = re - 127;
    const long long R = (long long)((rb & 0x7FFFFFu) | 0x800000u);
    long long inc[PFX_K];
    bool tie[PFX_K], neg[PFX_K];
    long long tsum = 0;
    int ntie = 0;
#pragma unroll
    for (int k = 0; k < PFX_K; k++) {
      const int li = tid * PFX_K + k;
      const unsigned b = __float_as_uint(wv[k]);
      const int ew = (b >> 23) & 0xFF;
      unsigned mw = b & 0x7FFFFFu;
      const bool zero = (b & 0x7FFFFFFFu) == 0;
      const int E = ew == 0 ? -126 : ew - 127;
      if (ew != 0) mw |= 0x800000u;
      const int sft = e - E;
      const bool in = li < cnt;
      const bool bad = in && ((ew == 255) || (sft < 0));  // NaN / inf / at least as large as the binade: real add
      neg[k] = (b >> 31) != 0 && !zero;
      const int sc = sft < 0 ? 0 : (sft > 26 ? 26 : sft);
      const unsigned f = mw >> sc;
      const unsigned rem = mw & ((1u << sc) - 1u);
      const unsigned half = sc >= 1 ? (1u << (sc - 1)) : 0u;
      const bool up = sc >= 1 && rem > half;
      tie[k] = in && !bad && sc >= 1 && rem == half;
      long long q = (long long)f + (up ? 1 : 0);  // |w|/u rounded to nearest (ties toward zero, patched below)
      if (neg[k]) q = -q;
      if (!in || bad) q = 0;
      if (bad) atomicMin(&s_first_bad, li);
      inc[k] = q;
      tsum += q;
      ntie += tie[k] ? 1 : 0;
    }
    long long tot_ll;
    const long long sincl = pfx_block_scan<NT, long long>(tsum, sh_ll, tot_ll);
    long long S[PFX_K];  // inclusive prefix of the increments at this thread's elements
    {
      long long acc = sincl - tsum;
#pragma unroll
      for (int k = 0; k < PFX_K; k++) { acc += inc[k]; S[k] = acc; }
    }
    // ---- ties, in element order: the even neighbour wins, which depends on everything before the tie
    int tot_tie;
    const int tincl = pfx_block_scan<NT, int>(ntie, sh_i, tot_tie);
    const int tfirst = tincl - ntie;
    {
      int pos = tfirst;
#pragma unroll
      for (int k = 0; k < PFX_K; k++)
        if (tie[k]) {
          if (pos < PFX_TIE_CAP) {
            tie_pos[pos] = (tid * PFX_K + k) | (neg[k] ? (int)0x80000000 : 0);
            tie_sval[pos] = S[k];
          } else if (pos == PFX_TIE_CAP) {
            atomicMin(&s_first_bad, tid * PFX_K + k);  // more ties than slots: cut the tile here
          }
          pos++;
        }
    }
    pfx_sync();
    if (tid == 0 && tot_tie > 0) {
      long long c = 0;
      const int fb = s_first_bad;
      const int nt = min(tot_tie, PFX_TIE_CAP);
      for (int t = 0; t < nt; t++) {
        const int pk = tie_pos[t];
        const int li = pk & 0x7FFFFFFF;
        signed char corr = 0;
        if (li < fb) {
          const long long V = R + tie_sval[t] + c;  // mantissa if the tie is rounded toward zero
          if (V & 1) corr = (pk < 0) ? -1 : 1;      // exact value is V +- 1/2: move to the even neighbour
        }
        tie_corr[t] = corr;
        c += corr;
      }
    }
    pfx_sync();
    int csum = 0;
    int corr[PFX_K];
    {
      int pos = tfirst;
#pragma unroll
      for (int k = 0; k < PFX_K; k++) {
        corr[k] = 0;
        if (tie[k]) { corr[k] = pos < PFX_TIE_CAP ? tie_corr[pos] : 0; pos++; }
        csum += corr[k];
      }
    }
    int tot_c;
    const int cincl = pfx_block_scan<NT, int>(csum, sh_i, tot_c);
    // ---- mantissas, first element that leaves the binade
    long long state[PFX_K];
    {
      int cacc = cincl - csum;
#pragma unroll
      for (int k = 0; k < PFX_K; k++) {
        cacc += corr[k];
        state[k] = R + S[k] + cacc;
        const int li = tid * PFX_K + k;
        if (li < cnt && (state[k] >= (1ll << 24) || state[k] < (1ll << 23))) atomicMin(&s_first_cross, li);
      }
    }
    pfx_sync();
    const int stop = min(min(s_first_bad, s_first_cross), cnt);
    // ---- commit [0, stop): values and running maximum
    float val[PFX_K], lm[PFX_K];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < PFX_K; k++) {
      const int li = tid * PFX_K + k;
      val[k] = __uint_as_float(((unsigned)re << 23) | ((unsigned)state[k] & 0x7FFFFFu));
      if (li < stop) m = fmaxf(m, val[k]);
      lm[k] = m;
    }
    const float mincl = pfx_block_scan_max<NT>(m, sh_f);
    float mexcl = __shfl_up(mincl, 1, 64);
    {  // exclusive maximum over the preceding threads
      pfx_sync();
      last_max[tid] = mincl;
      pfx_sync();
      mexcl = tid > 0 ? last_max[tid - 1] : -INFINITY;
    }
    const float mbase = fmaxf(carry, mexcl);
    float lastv = r, lastm = carry;
#pragma unroll
    for (int k = 0; k < PFX_K; k++) {
      const int li = tid * PFX_K + k;
      if (li < stop) {
        const float rm = fmaxf(mbase, lm[k]);
        runmax[base + li] = rm;
        if (prefix_opt) prefix_opt[base + li] = val[k];
        lastv = val[k];
        lastm = rm;
      }
    }
    pfx_sync();
    last_val[tid] = lastv;   // value / running max at this thread's last committed element (if any)
    last_max[tid] = lastm;
    pfx_sync();
    if (tid == 0) {
      float pv = r, pm = carry;  // value / running max at element stop-1
      if (stop > 0) {
        const int ot = (stop - 1) / PFX_K;
        pv = last_val[ot];
        pm = last_max[ot];
      }
      long long nb = base + stop;
      if (stop < cnt) {  // one real float addition, then a new binade
        const float nr = pv + w[base + stop];
        if (nr == nr) pm = fmaxf(pm, nr);
        runmax[base + stop] = pm;
        if (prefix_opt) prefix_opt[base + stop] = nr;
        pv = nr;
        nb = base + stop + 1;
      }
      s_r = pv; s_carry = pm; s_base = nb;
    }
    pfx_sync();
  }
  r_io = s_r;
  carry_io = s_carry;
  pfx_sync();
}

__global__ __launch_bounds__(PFX_THREADS) void prefix_exact_kernel(const float* __restrict__ w, int64_t n,
                                                                   float* __restrict__ runmax,
                                                                   float* __restrict__ prefix_opt) {
  float r = 0.f, carry = -INFINITY;
  pfx_exact_range<PFX_THREADS>(w, 0, n, runmax, prefix_opt, r, carry);
}

// ---- exact parallel prefix over many workgroups ---------------------------------------------------------------------
// Inside one binade (ulp u) the chain r <- fl(r + w) with w >= 0 is R <- R + a(R & 1): the increment of one element is
// an integer that depends on the running mantissa only through its PARITY (round-half-even ties).  A run of elements is
// therefore summarised by two integers (D0, D1) — its total increment entered with an even / odd mantissa — and
// summaries compose associatively: (A then B)_p = A_p + B_{(p + A_p) & 1}.  That turns the chain into a scan:
//   1. pfx_chunk_sum_kernel      — per chunk of PFXM_CHUNK weights: the sum in double
//   2. pfx_chunk_summary_kernel  — per chunk: binade predicted from the double sum of everything before it; (D0, D1) in
//                                  that binade, or "irregular" (NaN / inf / negative / weight above the binade)
//   3. pfx_walk_kernel           — ONE workgroup walks the chunks in order with the exact running sum: a chunk whose
//                                  prediction holds (same binade, R + D_p stays below 2^24) is a single integer add;
//                                  any other chunk (the ~log2(n) binade crossings, irregular weights, the head) is
//                                  processed on the spot by pfx_exact_range
//   4. pfx_chunk_fill_kernel     — per accepted chunk: the same scan again, now with the exact starting mantissa,
//                                  writes every element's running sum / running maximum
// Every value written is the serial float32 chain's, whatever the prediction was: a wrong prediction only sends the
// chunk down the slower path.
#define PFXM_THREADS 256
#define PFXM_K 16
#define PFXM_CHUNK (PFXM_THREADS * PFXM_K)
#define PFXM_SAT (1u << 30)
struct PfxChunk {     // 32 bytes per chunk in the caller's workspace
  double sum;         // 1: double sum of the chunk
  int re;             // 2: predicted biased exponent of the running sum, or -1 = irregular
  unsigned d0, d1;    // 2: total increment entered with an even / odd mantissa
  float r0, carry0;   // 3: running sum / running maximum before the chunk's first element (accepted chunks)
  int accepted;       // 3: 1 = pfx_chunk_fill_kernel writes this chunk's outputs
};
static_assert(sizeof(PfxChunk) == 32, "PfxChunk");

struct PfxPair { unsigned a0, a1; };
__device__ __forceinline__ unsigned pfx_sat(unsigned x) { return x > PFXM_SAT ? PFXM_SAT : x; }
// first A, then B
__device__ __forceinline__ PfxPair pfx_compose(PfxPair A, PfxPair B) {
  PfxPair c;
  c.a0 = pfx_sat(A.a0 + ((A.a0 & 1u) ? B.a1 : B.a0));
  c.a1 = pfx_sat(A.a1 + ((A.a1 & 1u) ? B.a0 : B.a1));   // entered odd: the parity after A is (1 + A.a1) & 1
  return c;
}
// One weight against the binade with exponent e (unbiased): f = floor(w/u) (+1 when the remainder exceeds one half),
// tie = remainder exactly one half, bad = cannot be an integer increment (NaN, inf, negative, exponent above e).
__device__ __forceinline__ void pfx_classify(float wv, int e, unsigned& f, bool& tie, bool& bad) {
  const unsigned b = __float_as_uint(wv);
  const int ew = (b >> 23) & 0xFF;
  unsigned mw = b & 0x7FFFFFu;
  const bool zero = (b & 0x7FFFFFFFu) == 0;
  const int E = ew == 0 ? -126 : ew - 127;
  if (ew != 0) mw |= 0x800000u;
  const int sft = e - E;
  bad = ew == 255 || sft < 0 || ((b >> 31) != 0 && !zero);
  const int sc = sft < 0 ? 0 : (sft > 26 ? 26 : sft);
  const unsigned q = mw >> sc;
  const unsigned rem = mw & ((1u << sc) - 1u);
  const unsigned half = sc >= 1 ? (1u << (sc - 1)) : 0u;
  tie = sc >= 1 && rem == half;
  f = q + ((sc >= 1 && rem > half) ? 1u : 0u);
  if (bad) { f = 0; tie = false; }
}
__device__ __forceinline__ PfxPair pfx_element_pair(unsigned f, bool tie) {
  PfxPair p;
  p.a0 = f + (tie ? (f & 1u) : 0u);         // even mantissa + f + 1/2 -> the even neighbour
  p.a1 = f + (tie ? ((f & 1u) ^ 1u) : 0u);
  return p;
}
// inclusive scan of per-thread pairs over a workgroup of NT threads; returns the EXCLUSIVE pair of this thread and the
// workgroup total.  `sh` holds one slot per wave; safe to call repeatedly (leading barrier).
template <int NT>
__device__ __forceinline__ PfxPair pfx_pair_scan(PfxPair v, PfxPair* sh, PfxPair& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    PfxPair t;
    t.a0 = __shfl_up(v.a0, o, 64);
    t.a1 = __shfl_up(v.a1, o, 64);
    if (lane >= o) v = pfx_compose(t, v);
  }
  pfx_sync();
  if (lane == 63) sh[wave] = v;
  pfx_sync();
  PfxPair pre = {0u, 0u}, tot = {0u, 0u};
#pragma unroll
  for (int k = 0; k < NT / 64; k++) {
    const PfxPair x = sh[k];
    if (k < wave) pre = pfx_compose(pre, x);
    tot = pfx_compose(tot, x);
  }
  total = tot;
  PfxPair ex;
  ex.a0 = __shfl_up(v.a0, 1, 64);
  ex.a1 = __shfl_up(v.a1, 1, 64);
  if (lane == 0) { ex.a0 = 0u; ex.a1 = 0u; }
  return pfx_compose(pre, ex);
}
__device__ __forceinline__ void pfx_load_chunk(const float* __restrict__ w, long long lo, int cnt, float (&wv)[PFXM_K]) {
  const int t0 = threadIdx.x * PFXM_K;
  if (t0 + PFXM_K <= cnt && ((lo & 3) == 0)) {
    const float4* p = reinterpret_cast<const float4*>(w + lo + t0);
#pragma unroll
    for (int k = 0; k < PFXM_K / 4; k++) {
      const float4 v = p[k];
      wv[4 * k] = v.x; wv[4 * k + 1] = v.y; wv[4 * k + 2] = v.z; wv[4 * k + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < PFXM_K; k++) wv[k] = (t0 + k < cnt) ? w[lo + t0 + k] : 0.f;
  }
}

__global__ __launch_bounds__(PFXM_THREADS) void pfx_chunk_sum_kernel(const float* __restrict__ w, int64_t n,
                                                                     PfxChunk* __restrict__ ch) {
  __shared__ double shd[PFXM_THREADS / 64];
  const long long lo = (long long)blockIdx.x * PFXM_CHUNK;
  const int cnt = (int)min((long long)PFXM_CHUNK, (long long)n - lo);
  float wv[PFXM_K];
  pfx_load_chunk(w, lo, cnt, wv);
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < PFXM_K; k++) acc += (double)wv[k];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = acc;
  pfx_sync();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < PFXM_THREADS / 64; k++) t += shd[k];
    ch[blockIdx.x].sum = t;
  }
}

__global__ __launch_bounds__(PFXM_THREADS) void pfx_chunk_summary_kernel(const float* __restrict__ w, int64_t n,
                                                                         PfxChunk* __restrict__ ch) {
  __shared__ double shd[PFXM_THREADS / 64];
  __shared__ PfxPair shp[PFXM_THREADS / 64];
  __shared__ int s_bad;
  const int c = blockIdx.x;
  const long long lo = (long long)c * PFXM_CHUNK;
  const int cnt = (int)min((long long)PFXM_CHUNK, (long long)n - lo);
  // predicted running sum before the chunk: the double sums of the chunks before it, in chunk order per thread
  double acc = 0.0;
  for (int j = threadIdx.x; j < c; j += PFXM_THREADS) acc += ch[j].sum;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = acc;
  if (threadIdx.x == 0) s_bad = 0;
  pfx_sync();
  double before = 0.0;
  for (int k = 0; k < PFXM_THREADS / 64; k++) before += shd[k];
  const float r_pred = (float)before, r_end = (float)(before + ch[c].sum);
  const unsigned pb = __float_as_uint(r_pred), eb = __float_as_uint(r_end);
  const int re = (pb >> 23) & 0xFF;
  // a chunk that is predicted to start and end in one binade of a positive normal sum; everything else is irregular
  const bool plausible = (pb >> 31) == 0 && re != 0 && re != 255 && (int)((eb >> 23) & 0xFF) == re && (eb >> 31) == 0;
  if (!plausible) {   // uniform across the workgroup
    if (threadIdx.x == 0) { ch[c].re = -1; ch[c].d0 = 0u; ch[c].d1 = 0u; }
    return;
  }
  float wv[PFXM_K];
  pfx_load_chunk(w, lo, cnt, wv);
  PfxPair mine = {0u, 0u};
  bool anybad = false;
#pragma unroll
  for (int k = 0; k < PFXM_K; k++) {
    unsigned f; bool tie, bad;
    pfx_classify(wv[k], re - 127, f, tie, bad);
    anybad |= bad;
    mine = pfx_compose(mine, pfx_element_pair(f, tie));
  }
  if (anybad) s_bad = 1;   // benign race: every writer stores 1; ordered by the barrier inside the scan
  PfxPair total;
  (void)pfx_pair_scan<PFXM_THREADS>(mine, shp, total);
  if (threadIdx.x == 0) {
    const bool ok = s_bad == 0 && total.a0 < (1u << 24) && total.a1 < (1u << 24);
    ch[c].re = ok ? re : -1;
    ch[c].d0 = total.a0;
    ch[c].d1 = total.a1;
  }
}

// A chunk the walk cannot take as one integer add (it holds a binade crossing or an irregular weight, or was
// mispredicted), carried through in order by the walking workgroup: one scan per stretch between two real float
// additions.  The walking workgroup has PFXM_THREADS threads holding 16 weights each, as in the fill kernel (one wave
// per SIMD: a pass costs what ONE wave issues).  Anything that is not a positive normal running sum goes to
// pfx_exact_range.
#define PFXW_HEAD 64   // leading elements the walk adds one by one (tunable: TDR_PFX_HEAD)
__device__ __forceinline__ void pfx_walk_chunk(const float* __restrict__ w, long long lo, int cnt,
                                               float* __restrict__ runmax, float* __restrict__ prefix_opt,
                                               float& r, float& carry, int head_len) {
  __shared__ PfxPair shp[PFXM_THREADS / 64];
  __shared__ int s_bad, s_cross;
  __shared__ float s_last, s_wstop;
  const int tid = threadIdx.x, t0 = tid * PFXM_K;
  float wv[PFXM_K];
  pfx_load_chunk(w, lo, cnt, wv);
  int pos = 0;   // workgroup-uniform: elements before pos are done
  while (pos < cnt) {
    const unsigned rb = __float_as_uint(r);
    const unsigned re = rb >> 23;   // sign included
    if (!(re >= 1u && re <= 254u)) {
      // zero / subnormal / negative / inf / NaN running sum: the general path (with its serial head at the very start)
      const long long a = lo + pos;
      const long long b = a == 0 ? min((long long)head_len, (long long)cnt) : lo + cnt;
      pfx_exact_range<PFXM_THREADS>(w, a, b, runmax, prefix_opt, r, carry, head_len);
      pos = (int)(b - lo);
      continue;
    }
    const unsigned R = (rb & 0x7FFFFFu) | 0x800000u;
    pfx_sync();
    if (tid == 0) { s_bad = cnt; s_cross = cnt; s_last = r; s_wstop = 0.f; }
    pfx_sync();
    unsigned f[PFXM_K];
    unsigned tiebits = 0u;
    PfxPair mine = {0u, 0u};
    {
#pragma unroll
      for (int k = 0; k < PFXM_K; k++) {
        const int li = t0 + k;
        bool bad, tie;
        pfx_classify(wv[k], (int)re - 127, f[k], tie, bad);
        if (li < pos || li >= cnt) { f[k] = 0u; tie = false; bad = false; }
        if (bad) atomicMin(&s_bad, li);
        tiebits |= tie ? (1u << k) : 0u;
        mine = pfx_compose(mine, pfx_element_pair(f[k], tie));
      }
    }
    PfxPair total;
    const PfxPair ex = pfx_pair_scan<PFXM_THREADS>(mine, shp, total);
    unsigned st[PFXM_K];
    {
      unsigned state = R + ((R & 1u) ? ex.a1 : ex.a0);
      int first = cnt;
#pragma unroll
      for (int k = 0; k < PFXM_K; k++) {
        state += f[k] + (((tiebits >> k) & 1u) ? ((state + f[k]) & 1u) : 0u);
        st[k] = state;
        const int li = t0 + k;
        if (li >= pos && li < cnt && state >= (1u << 24)) first = min(first, li);
      }
      if (first < cnt) atomicMin(&s_cross, first);
    }
    pfx_sync();
    const int stop = min(s_bad, s_cross);   // first element that needs a real float addition (or cnt)
    {
#pragma unroll
      for (int k = 0; k < PFXM_K; k++) {
        const int li = t0 + k;
        if (li >= pos && li < stop) {
          const float val = __uint_as_float((re << 23) | (st[k] & 0x7FFFFFu));
          runmax[lo + li] = fmaxf(carry, val);
          if (prefix_opt) prefix_opt[lo + li] = val;
          if (li == stop - 1) s_last = val;
        }
        if (li == stop) s_wstop = wv[k];
      }
    }
    pfx_sync();
    r = s_last;                  // the sum after element stop-1 (unchanged when nothing was committed)
    carry = fmaxf(carry, r);     // increments are non-negative: the last committed value is the largest
    if (stop < cnt) {
      const float nr = r + s_wstop;   // particle_filter.cpp:179, one real addition
      if (nr == nr) carry = fmaxf(carry, nr);
      if (tid == 0) {
        runmax[lo + stop] = carry;
        if (prefix_opt) prefix_opt[lo + stop] = nr;
      }
      r = nr;
      pos = stop + 1;
    } else {
      pos = cnt;
    }
  }
}

#define PFXW_BLOCK 512   // chunk summaries / headers staged in LDS at a time
__global__ __launch_bounds__(PFXM_THREADS) void pfx_walk_kernel(const float* __restrict__ w, int64_t n,
                                                               PfxChunk* __restrict__ ch, int nch,
                                                               float* __restrict__ runmax,
                                                               float* __restrict__ prefix_opt, int head_len) {
  __shared__ int sm_re[PFXW_BLOCK], sm_acc[PFXW_BLOCK];
  __shared__ unsigned sm_d0[PFXW_BLOCK], sm_d1[PFXW_BLOCK];
  __shared__ float sm_r0[PFXW_BLOCK], sm_c0[PFXW_BLOCK];
  float r = 0.f, carry = -INFINITY;   // workgroup-uniform
  for (int cb = 0; cb < nch; cb += PFXW_BLOCK) {
    pfx_sync();
    for (int t = threadIdx.x; t < PFXW_BLOCK && cb + t < nch; t += PFXM_THREADS) {
      const PfxChunk x = ch[cb + t];
      sm_re[t] = x.re; sm_d0[t] = x.d0; sm_d1[t] = x.d1;
    }
    pfx_sync();
    const int ce = min(nch, cb + PFXW_BLOCK);
    for (int c = cb; c < ce; c++) {
      const unsigned rb = __float_as_uint(r);
      const int re = (int)(rb >> 23);                   // sign bit included: a negative sum never matches
      const unsigned R = (rb & 0x7FFFFFu) | 0x800000u;
      const unsigned D = (R & 1u) ? sm_d1[c - cb] : sm_d0[c - cb];
      const bool fast = sm_re[c - cb] == re && R + D < (1u << 24);   // sm_re is in [1, 254] or -1
      // the waves run through this loop unsynchronised and all store the same words
      if (fast) {
        sm_r0[c - cb] = r; sm_c0[c - cb] = carry; sm_acc[c - cb] = 1;
        r = __uint_as_float(((unsigned)re << 23) | ((R + D) & 0x7FFFFFu));
        carry = fmaxf(carry, r);
      } else {
        sm_acc[c - cb] = 0;
        const long long lo = (long long)c * PFXM_CHUNK;
        const int cnt = (int)min((long long)PFXM_CHUNK, (long long)n - lo);
        pfx_walk_chunk(w, lo, cnt, runmax, prefix_opt, r, carry, head_len);
      }
#ifdef TDR_PFX_TIMING   // diagnostic build: time stamp (100 MHz) after every chunk in the header's dead `sum` slot
      if (threadIdx.x == 0) *reinterpret_cast<long long*>(&ch[c].sum) = (long long)wall_clock64();
#endif
    }
    pfx_sync();
    for (int t = threadIdx.x; t < PFXW_BLOCK && cb + t < nch; t += PFXM_THREADS) {   // for pfx_chunk_fill_kernel
      PfxChunk* o = ch + cb + t;
      o->r0 = sm_r0[t];
      o->carry0 = sm_c0[t];
      o->accepted = sm_acc[t];
    }
  }
}

__global__ __launch_bounds__(PFXM_THREADS) void pfx_chunk_fill_kernel(const float* __restrict__ w, int64_t n,
                                                                      const PfxChunk* __restrict__ ch,
                                                                      float* __restrict__ runmax,
                                                                      float* __restrict__ prefix_opt) {
  __shared__ PfxPair shp[PFXM_THREADS / 64];
  const int c = blockIdx.x;
  const PfxChunk hdr = ch[c];
  if (!hdr.accepted) return;   // written by the walk
  const long long lo = (long long)c * PFXM_CHUNK;
  const int cnt = (int)min((long long)PFXM_CHUNK, (long long)n - lo);
  const unsigned rb = __float_as_uint(hdr.r0);
  const unsigned re = rb >> 23;
  const unsigned R = (rb & 0x7FFFFFu) | 0x800000u;
  float wv[PFXM_K];
  pfx_load_chunk(w, lo, cnt, wv);
  unsigned f[PFXM_K];
  bool tie[PFXM_K];
  PfxPair mine = {0u, 0u};
#pragma unroll
  for (int k = 0; k < PFXM_K; k++) {
    bool bad;
    pfx_classify(wv[k], (int)re - 127, f[k], tie[k], bad);
    mine = pfx_compose(mine, pfx_element_pair(f[k], tie[k]));
  }
  PfxPair total;
  const PfxPair ex = pfx_pair_scan<PFXM_THREADS>(mine, shp, total);
  unsigned state = R + ((R & 1u) ? ex.a1 : ex.a0);   // the exact mantissa before this thread's first element
  const int t0 = threadIdx.x * PFXM_K;
  float val[PFXM_K];
#pragma unroll
  for (int k = 0; k < PFXM_K; k++) {
    state += f[k] + (tie[k] ? ((state + f[k]) & 1u) : 0u);
    val[k] = __uint_as_float((re << 23) | (state & 0x7FFFFFu));
  }
  const float carry = hdr.carry0;
  if (t0 + PFXM_K <= cnt && ((lo & 3) == 0)) {
    float4* o = reinterpret_cast<float4*>(runmax + lo + t0);
#pragma unroll
    for (int k = 0; k < PFXM_K / 4; k++)
      o[k] = make_float4(fmaxf(carry, val[4 * k]), fmaxf(carry, val[4 * k + 1]), fmaxf(carry, val[4 * k + 2]),
                         fmaxf(carry, val[4 * k + 3]));
    if (prefix_opt) {
      float4* q = reinterpret_cast<float4*>(prefix_opt + lo + t0);
#pragma unroll
      for (int k = 0; k < PFXM_K / 4; k++) q[k] = make_float4(val[4 * k], val[4 * k + 1], val[4 * k + 2], val[4 * k + 3]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < PFXM_K; k++)
      if (t0 + k < cnt) {
        runmax[lo + t0 + k] = fmaxf(carry, val[k]);
        if (prefix_opt) prefix_opt[lo + t0 + k] = val[k];
      }
  }
}

// Dispatch (tools/bench_prefix_modes.py on MI355X; us at n = 1k / 4k / 8k / 20k / 100k: one wave 16 / 43 / 84 / 208 /
// 1036, one workgroup 60 / 129 / 156 / 235 / 417, multi-workgroup 38 / 59 / 66 / 94 / 129):
#define TDR_PFX_MULTI_MIN_N 6144    // with a workspace: the multi-workgroup scan from here on, one wave below
#define TDR_PFX_EXACT_MIN_N 24576   // without a workspace: one workgroup from here on, one wave below
extern "C" int64_t tdr_prefix_workspace_bytes(int64_t n) {
  return n < 1 ? 0 : (int64_t)sizeof(PfxChunk) * cdiv(n, (int64_t)PFXM_CHUNK);
}
static int prefix_multi(const float* w, int64_t n, float* runmax_out, float* prefix_out, void* workspace,
                        hipStream_t st) {
  const int64_t nch64 = cdiv(n, (int64_t)PFXM_CHUNK);
  if (nch64 > (1 << 24)) return fail(TDR_ERR_ARG, "prefix: n too large");
  const int nch = (int)nch64;
  PfxChunk* ch = reinterpret_cast<PfxChunk*>(workspace);
  hipLaunchKernelGGL(pfx_chunk_sum_kernel, dim3(nch), dim3(PFXM_THREADS), 0, st, w, n, ch);
  hipLaunchKernelGGL(pfx_chunk_summary_kernel, dim3(nch), dim3(PFXM_THREADS), 0, st, w, n, ch);
  static const int head_len = [] {
    const char* e = getenv("TDR_PFX_HEAD");
    const int v = e ? atoi(e) : PFXW_HEAD;
    return v < 1 ? 1 : (v > PFX_HEAD ? PFX_HEAD : v);
  }();
  hipLaunchKernelGGL(pfx_walk_kernel, dim3(1), dim3(PFXM_THREADS), 0, st, w, n, ch, nch, runmax_out, prefix_out,
                     head_len);
  hipLaunchKernelGGL(pfx_chunk_fill_kernel, dim3(nch), dim3(PFXM_THREADS), 0, st, w, n, (const PfxChunk*)ch,
                     runmax_out, prefix_out);
  return TDR_OK;
}
extern "C" int tdr_k_prefix(const float* w, int64_t n, float* runmax_out, void* workspace, void* stream) {
  if (!w || !runmax_out || n < 1) return fail(TDR_ERR_ARG, "prefix: bad arguments");
  if (workspace && n >= TDR_PFX_MULTI_MIN_N) {
    const int rc = prefix_multi(w, n, runmax_out, nullptr, workspace, (hipStream_t)stream);
    if (rc) return rc;
  } else if (n < TDR_PFX_EXACT_MIN_N)
    hipLaunchKernelGGL(prefix_serial_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, w, n, runmax_out);
  else
    hipLaunchKernelGGL(prefix_exact_kernel, dim3(1), dim3(PFX_THREADS), 0, (hipStream_t)stream, w, n, runmax_out,
                       (float*)nullptr);
  LAUNCH_CHECK("prefix");
  return TDR_OK;
}
// Test / diagnostic entry: mode 0 = serial kernel, 1 = exact parallel kernel in one workgroup, 2 = the multi-workgroup
// scan (needs a workspace of tdr_prefix_workspace_bytes(n)); prefix_out (optional, modes 1 and 2) receives the raw
// running sums.
extern "C" int tdr_k_prefix_mode(const float* w, int64_t n, int mode, float* runmax_out, float* prefix_out,
                                 void* workspace, void* stream) {
  if (!w || !runmax_out || n < 1) return fail(TDR_ERR_ARG, "prefix_mode: bad arguments");
  if (mode == 2 && !workspace) return fail(TDR_ERR_ARG, "prefix_mode: mode 2 needs a workspace");
  if (mode == 0)
    hipLaunchKernelGGL(prefix_serial_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, w, n, runmax_out);
  else if (mode == 2) {
    const int rc = prefix_multi(w, n, runmax_out, prefix_out, workspace, (hipStream_t)stream);
    if (rc) return rc;
  } else
    hipLaunchKernelGGL(prefix_exact_kernel, dim3(1), dim3(PFX_THREADS), 0, (hipStream_t)stream, w, n, runmax_out,
                       prefix_out);
  LAUNCH_CHECK("prefix_mode");
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
                                                       const float* __restrict__ about, float* __restrict__ out) {
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
    acc[4] += cos((double)th); acc[5] += sin((double)th);
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
                                                             float* __restrict__ out) {
  __shared__ double shd[16];
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int64_t p = (int64_t)blockIdx.x * MC_THREADS + threadIdx.x; p < n; p += (int64_t)MC_WGS * MC_THREADS) {
    const float sc = st[TDR_ST_SCALE * cap + p];
    const float x = st[TDR_ST_DX * cap + p] * sc + st[TDR_ST_INIT_X * cap + p];  // mlState, state_particle.cpp:98-102
    const float y = st[TDR_ST_DY * cap + p] * sc + st[TDR_ST_INIT_Y * cap + p];
    const float th = st[TDR_ST_THETA * cap + p];
    acc[0] += x; acc[1] += y; acc[2] += th; acc[3] += sc;
    acc[4] += cos((double)th); acc[5] += sin((double)th);
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
    hipLaunchKernelGGL(mean_cov_kernel, dim3(1), dim3(1024), 0, s, st, cap, n, about, out);
  } else {
    hipLaunchKernelGGL(mc_sums_kernel, dim3(MC_WGS), dim3(MC_THREADS), 0, s, st, cap, n, out);
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
                                     int64_t n, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int64_t best = (int64_t)__float_as_int(info[0]);
  if (best < 0 || best >= n) best = 0;
  float f[TDR_ST_FIELDS];
#pragma unroll
  for (int k = 0; k < TDR_ST_FIELDS; k++) { f[k] = st[(int64_t)k * cap + best]; out[k] = f[k]; }
  out[7] = 0.f;
  out[8] = f[TDR_ST_DX] * f[TDR_ST_SCALE] + f[TDR_ST_INIT_X];
  out[9] = f[TDR_ST_DY] * f[TDR_ST_SCALE] + f[TDR_ST_INIT_Y];
  out[10] = f[TDR_ST_THETA];
  out[11] = f[TDR_ST_SCALE];
}
extern "C" int tdr_k_save_ml_state(const float* info, const float* st, int64_t cap, int64_t n, float* out12,
                                   void* stream) {
  if (!info || !st || !out12 || n < 1 || cap < n) return fail(TDR_ERR_ARG, "save_ml_state: bad arguments");
  hipLaunchKernelGGL(save_ml_state_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, info, st, cap, n, out12);
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

__global__ void selftest_atan2_kernel(const float* __restrict__ y, const float* __restrict__ x, int64_t n,
                                      float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = tdr_atan2f(y[i], x[i]);
}
extern "C" int tdr_k_selftest_atan2(const float* y, const float* x, int64_t n, float* out, void* stream) {
  if (!y || !x || !out || n < 1) return fail(TDR_ERR_ARG, "selftest_atan2: bad arguments");
  hipLaunchKernelGGL(selftest_atan2_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, y, x, n, out);
  LAUNCH_CHECK("selftest_atan2");
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

static size_t radix_tmp_bytes(int64_t n) {
  size_t bytes = 0;
  uint32_t* k = nullptr;
  int32_t* v = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0u, 32u, (hipStream_t)0, false);
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
  HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, perm_out, (size_t)n, 0u, bits, s,
                                    false));
  return TDR_OK;
}
