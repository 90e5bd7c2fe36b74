// tdr_raster.hip — scan raster (polar and Cartesian) and the packed scan records.
#include "tdr_common.h"
#include "tdr_atan2f.h"

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

__device__ __forceinline__ bool raster_bin(const RasterArgs& a, float x, float y, int& row, int& col) {
  if (x == 0.f && y == 0.f) return false;
  // A non-finite coordinate never lands in the image: the reference's float -> int conversions of NaN / inf give INT_MIN
  // on x86-64, which fails `>= 0` (scan_renderer_polar.cpp:102, scan_renderer.cpp:71); the GPU's conversion of NaN gives
  // 0, so the point is dropped here (organised PCL clouds with is_dense == false carry NaN points).
  if (!(fabsf(x) < INFINITY) || !(fabsf(y) < INFINITY)) return false;
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
    const int pc = (cf == cf) ? (int)cf : -1;   // NaN label: x86 converts to INT_MIN, outside the LUT
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
    int pc = (cf == cf) ? (int)cf : -1;   // NaN label: x86 converts to INT_MIN, outside the LUT
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
  if (per_col > 152 * 1024) return fail(TDR_ERR_ARG, "raster: ncls*rows too large for one LDS tile (152 KB)");
  RasterArgs a;
  a.pts = pts; a.stride = stride; a.ioff = ioff; a.n = n; a.res = res; a.ang_res = ang_res; a.lut = lut;
  a.ncls = ncls; a.rows = rows; a.cols = cols; a.rf = tdr_rec_floats(ncls); a.polar = polar; a.img = img; a.pk = pk;
  a.cpt = (int)std::max<int64_t>(1, (64 * 1024) / per_col);
  a.cpt = std::min(a.cpt, cols);
  // enough workgroups to spread over the chip when the image is small
  while (a.cpt > 1 && cdiv(cols, a.cpt) < 32) a.cpt = (a.cpt + 1) / 2;
  size_t lds = (size_t)a.cpt * per_col;
  if (lds > 64 * 1024) {   // one image column of more than 64 KB (cpt = 1): most of a CU's 160 KB, allowed per device
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    if (dev >= 64 || !attr_set[dev]) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(raster_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  152 * 1024));
      if (dev < 64) attr_set[dev] = true;
    }
  }
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
