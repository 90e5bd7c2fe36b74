// tdr_map.hip — map cell records (pack / unpack), map ingest from a label image, the polar sample table.
#include "tdr_common.h"

// ------------------------------------------------------------------------------------------------------------------
// K0: map packing.  Output is row-major over a GUARDED grid of (rows+2) x (cols+2) cell records: one ring of
// all-zero records around the map, so that a sample coordinate clamped to [-1, rows] x [-1, cols] always addresses a
// valid record and "out of bounds" needs no branch or select in the scoring loop — the guard record is exactly what
// the reference returns there: distance 0 (top_down_map_polar.cpp:39) and unknown (:51).
// Record slots: [0,ncls) distances, rf-1 = known (1 - mask); when a spare slot exists (ncls+2 <= rf) slot rf-2 also
// holds `known`, paired with a constant 1 in the scan record, so the known-cell count rides on the packed FMAs.
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
// Per cell the ingest keeps a word of class bits: bit c = the cell lies inside class c; bit 31 = no known class (mask = 1).
// A label image gives one bit per cell; the per-class rasters of the raster cache may overlap.
#define INGEST_MAXC 16
#define INGEST_UNKNOWN 0x80000000u
__global__ void ingest_labels_kernel(const uint8_t* __restrict__ img, int img_h, int img_w,
                                     const int32_t* __restrict__ lut, int lut_size, int ncls, int rows, int cols,
                                     float resolution, uint32_t* __restrict__ cls_map) {
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
  cls_map[idx] = c < 0 ? INGEST_UNKNOWN : (1u << c);
}
// TopDownMap::loadRasterizedMaps (src/top_down_map.cpp:213-224) + the first lines of computeDists (:293-305): planes
// [ncls][h][w] are the class<i>.png images as stored (8-bit grey, row 0 = top); the loader flips them back (:217), scales by
// 1/255 (:218); computeDists binarises with convertTo(CV_8UC1) — round to nearest: p <= 127 -> 0 = inside the class — and
// marks a cell unknown where every class casts to 1 (:294-299: float -> uint8 truncation, so only p == 255 counts).
__global__ void ingest_rasters_kernel(const uint8_t* __restrict__ planes, int ncls, int rows, int cols,
                                      uint32_t* __restrict__ cls_map) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ncell = (int64_t)rows * cols;
  if (idx >= ncell) return;
  const int r = (int)(idx / cols), c = (int)(idx % cols);
  uint32_t bits = 0;
  int ones = 0;
  for (int k = 0; k < ncls; k++) {
    const int p = planes[(int64_t)k * ncell + (int64_t)(rows - 1 - r) * cols + c];
    if (p <= 127) bits |= 1u << k;
    ones += p == 255 ? 1 : 0;
  }
  cls_map[idx] = bits | (ones > ncls - 1 ? INGEST_UNKNOWN : 0u);
}

// pass 1: per cell and class, distance (in cells, along the column) to the nearest cell of that class, capped at 255
__global__ void ingest_coldist_kernel(const uint32_t* __restrict__ cls_map, int ncls, int rows, int cols, int R,
                                      uint8_t* __restrict__ g /* [rows*cols][INGEST_MAXC] */) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  const int y = (int)(idx / cols), x = (int)(idx % cols);
  int gd[INGEST_MAXC];
#pragma unroll
  for (int c = 0; c < INGEST_MAXC; c++) gd[c] = 255;
  for (int d = 0; d <= R; d++) {
    const int ya = y - d, yb = y + d;
    const uint32_t ca = ya >= 0 ? cls_map[(int64_t)ya * cols + x] : 0u;
    const uint32_t cb = yb < rows ? cls_map[(int64_t)yb * cols + x] : 0u;
    const uint32_t either = ca | cb;
#pragma unroll
    for (int c = 0; c < INGEST_MAXC; c++)
      if (((either >> c) & 1u) && gd[c] == 255) gd[c] = d;
  }
  uint8_t* o = g + idx * INGEST_MAXC;
#pragma unroll
  for (int c = 0; c < INGEST_MAXC; c++) o[c] = (uint8_t)gd[c];
  (void)ncls;
}

// pass 2: exact squared distance = min over the row window of dx^2 + g^2; then the reference's post-processing
__global__ void ingest_rowmin_kernel(const uint32_t* __restrict__ cls_map, const uint8_t* __restrict__ g, int ncls,
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
  const bool unknown = (cls_map[idx] & INGEST_UNKNOWN) != 0;  // no class at this cell (:294-299): mask = 1, distances zeroed (:317)
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
  return (size_t)rows * cols * (4 + INGEST_MAXC) + 256;
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
  uint32_t* cls_map = reinterpret_cast<uint32_t*>(workspace);
  uint8_t* g = reinterpret_cast<uint8_t*>(workspace) + (((size_t)rows * cols * 4 + 255) & ~(size_t)255);
  const int64_t ncell = (int64_t)rows * cols;
  const int64_t nrec = (int64_t)tdr_map_rec_floats_total(ncls, rows, cols);
  hipLaunchKernelGGL(zero_floats_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, rec_out, nrec);  // guard ring
  hipLaunchKernelGGL(ingest_labels_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, s, label_img, img_h, img_w,
                     flatten_lut, lut_size, ncls, rows, cols, resolution, cls_map);
  hipLaunchKernelGGL(ingest_coldist_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, s, (const uint32_t*)cls_map,
                     ncls, rows, cols, R, g);
  hipLaunchKernelGGL(ingest_rowmin_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, s, (const uint32_t*)cls_map,
                     (const uint8_t*)g, ncls, rows, cols, R, resolution, rf, rec_out);
  LAUNCH_CHECK("map_from_labels");
  return TDR_OK;
}

// The same from the per-class rasters of the raster cache (ingest_rasters_kernel): planes [ncls][rows][cols] device bytes,
// the PNG images as stored.  The map has the images' shape; rec_out / workspace as for tdr_k_map_from_labels.
extern "C" int tdr_k_map_from_rasters(const uint8_t* planes, int ncls, int rows, int cols, float resolution, float* rec_out,
                                      void* workspace, void* stream) {
  if (!planes || !rec_out || !workspace) return fail(TDR_ERR_ARG, "map_from_rasters: null pointer");
  if (ncls < 1 || ncls > TDR_MAX_CLASSES || rows < 1 || cols < 1 || !(resolution > 0.f))
    return fail(TDR_ERR_ARG, "map_from_rasters: bad class count / shape / resolution");
  const int R = (int)std::ceil(50.0 / (double)resolution);
  if (R > 250) return fail(TDR_ERR_ARG, "map_from_rasters: resolution %g needs a %d-cell window (max 250)", resolution, R);
  const int rf = tdr_rec_floats(ncls);
  hipStream_t s = (hipStream_t)stream;
  uint32_t* cls_map = reinterpret_cast<uint32_t*>(workspace);
  uint8_t* g = reinterpret_cast<uint8_t*>(workspace) + (((size_t)rows * cols * 4 + 255) & ~(size_t)255);
  const int64_t ncell = (int64_t)rows * cols;
  const int64_t nrec = (int64_t)tdr_map_rec_floats_total(ncls, rows, cols);
  hipLaunchKernelGGL(zero_floats_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, rec_out, nrec);  // guard ring
  hipLaunchKernelGGL(ingest_rasters_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, s, planes, ncls, rows, cols,
                     cls_map);
  hipLaunchKernelGGL(ingest_coldist_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, s, (const uint32_t*)cls_map,
                     ncls, rows, cols, R, g);
  hipLaunchKernelGGL(ingest_rowmin_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, s, (const uint32_t*)cls_map,
                     (const uint8_t*)g, ncls, rows, cols, R, resolution, rf, rec_out);
  LAUNCH_CHECK("map_from_rasters");
  return TDR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// N4: the geometric layers geo_maps_[0..1] (top_down_map.h:79).  The static-map constructor derives them from the class
// images — getGeoRasterMap (src/top_down_map.cpp:410-427): layer 1 marks the cells of any "geometric" class (flattened
// class >= 3), layer 0 its complement — and runs computeDists on them (:58): geo_maps_[1] = distance to the nearest cell
// WITH a geometric class, geo_maps_[0] = distance to the nearest cell WITHOUT one, times the resolution, truncated at 50;
// no cell is masked (the two binary layers sum to 1 everywhere, :294-299).  Here the class presence is read off the cell
// records (distance 0 on a known cell) and the two layers go through the same exact distance transform as the class
// maps, into records of their own: a 2-class map {d_without, d_with, 1, 1}.
__global__ void geo_labels_kernel(const float* __restrict__ rec, int ncls, int rows, int cols, int rf,
                                  uint32_t* __restrict__ cls_map) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  const int r = (int)(idx / cols), c = (int)(idx % cols);
  const float* o = rec + ((int64_t)(r + 1) * (cols + 2) + (c + 1)) * rf;
  bool geo = false;
  if (o[rf - 1] != 0.f)
    for (int k = 3; k < ncls; k++) geo |= o[k] == 0.f;                          // :417-419
  cls_map[idx] = geo ? 2u : 1u;   // class 1 = with a geometric class, class 0 = without; never unknown
}
// the dynamic-map path leaves both layers at their initial constant 1 (loadCompressedRasterMap :126-133, "Not actually
// used at the moment"; updateMap never recomputes them): records {1, 1, 1, 1} inside the map, the zero guard ring outside
__global__ void geo_ones_kernel(int rows, int cols, float* __restrict__ rec) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t gcols = cols + 2;
  if (idx >= (int64_t)(rows + 2) * gcols) return;
  const int r = (int)(idx / gcols) - 1, c = (int)(idx % gcols) - 1;
  const float v = (r >= 0 && r < rows && c >= 0 && c < cols) ? 1.f : 0.f;
  for (int k = 0; k < 4; k++) rec[idx * 4 + k] = v;
}
// geo_rec_out: tdr_map_rec_floats_total(2, rows, cols) floats; workspace: tdr_map_ingest_workspace_bytes(2, rows, cols)
// (unused when constant_one != 0).
extern "C" int tdr_k_geo_map_from_map(const tdr_map_desc* map, int constant_one, float* geo_rec_out, void* workspace,
                                      void* stream) {
  if (!map || !map->rec || !geo_rec_out) return fail(TDR_ERR_ARG, "geo_map_from_map: null pointer");
  const int rows = map->rows, cols = map->cols;
  hipStream_t s = (hipStream_t)stream;
  const int64_t ncell = (int64_t)rows * cols, nrec = (int64_t)tdr_map_rec_floats_total(2, rows, cols);
  if (constant_one) {
    hipLaunchKernelGGL(geo_ones_kernel, dim3((unsigned)cdiv(nrec / 4, 256)), dim3(256), 0, s, rows, cols, geo_rec_out);
    LAUNCH_CHECK("geo_ones");
    return TDR_OK;
  }
  if (!workspace) return fail(TDR_ERR_ARG, "geo_map_from_map: workspace required");
  const int R = (int)std::ceil(50.0 / (double)map->resolution);
  if (R > 250) return fail(TDR_ERR_ARG, "geo_map_from_map: resolution %g needs a %d-cell window (max 250)", map->resolution, R);
  uint32_t* cls_map = reinterpret_cast<uint32_t*>(workspace);
  uint8_t* g = reinterpret_cast<uint8_t*>(workspace) + (((size_t)ncell * 4 + 255) & ~(size_t)255);
  hipLaunchKernelGGL(zero_floats_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, geo_rec_out, nrec);
  hipLaunchKernelGGL(geo_labels_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, s, map->rec, map->ncls, rows, cols,
                     map->rec_floats, cls_map);
  hipLaunchKernelGGL(ingest_coldist_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, s, (const uint32_t*)cls_map, 2,
                     rows, cols, R, g);
  hipLaunchKernelGGL(ingest_rowmin_kernel, dim3((unsigned)cdiv(ncell, 256)), dim3(256), 0, s, (const uint32_t*)cls_map,
                     (const uint8_t*)g, 2, rows, cols, R, map->resolution, 4, geo_rec_out);
  LAUNCH_CHECK("geo_map_from_map");
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
// the angle and the radius of sample (direction i, ring j), as samplePtsPolar has them before the product (:10-14)
static void polar_angle_radius(int i, int j, int nb, int nr, float ang_res, float resolution, float& a, float& r) {
  // samplePts(0, 0, pts, cols=nr, rows=nb, res=1): row0 = L_nb[i], row1 = L_nr[j]; identity rotation
  const float lo_r = (float)((double)(-1.f * (float)(nb - 1)) / 2.), hi_r = (float)((double)(1.f * (float)(nb - 1)) / 2.);
  const float lo_c = (float)((double)(-1.f * (float)(nr - 1)) / 2.), hi_c = (float)((double)(1.f * (float)(nr - 1)) / 2.);
  const float c = cosf(0.f), s = sinf(0.f);
  const float inv_res = (float)(1. / (double)resolution);
  const float first = s * linspaced_f(0, nb, lo_r, hi_r) + c * linspaced_f(0, nr, lo_c, hi_c) + 0.f;
  const float p0 = linspaced_f(i, nb, lo_r, hi_r), p1 = linspaced_f(j, nr, lo_c, hi_c);
  a = c * p0 + (-s) * p1 + 0.f;
  r = s * p0 + c * p1 + 0.f;
  r = r + (-first);
  a = a * ang_res;
  r = r * inv_res;
}
extern "C" int tdr_polar_table_host(int nb, int nr, float ang_res, float resolution, float* tab) {
  if (nb < 1 || nr < 1 || !tab) return fail(TDR_ERR_ARG, "polar_table: bad arguments");
  for (int j = 0; j < nr; j++) {
    for (int i = 0; i < nb; i++) {
      float a, r;
      polar_angle_radius(i, j, nb, nr, ang_res, resolution, a, r);
      size_t k = (size_t)i + (size_t)nb * j;
      tab[2 * k] = cosf(a) * r;
      tab[2 * k + 1] = sinf(a) * r;
    }
  }
  return TDR_OK;
}
// The table's two factors: fac[2 i], fac[2 i + 1] = cos, sin of direction i; fac[2 nb + j] = radius of ring j — every table
// entry is ONE float product of a direction's and a ring's (the identity rotation leaves the angle to i and the radius to j).
// A scoring kernel given them (tdr_score_ctx_set_polar_factors) checks that on the device against the table it is given.
extern "C" int tdr_polar_factors_host(int nb, int nr, float ang_res, float resolution, float* fac) {
  if (nb < 1 || nr < 1 || !fac) return fail(TDR_ERR_ARG, "polar_factors: bad arguments");
  for (int i = 0; i < nb; i++) {
    float a, r;
    polar_angle_radius(i, 0, nb, nr, ang_res, resolution, a, r);
    fac[2 * i] = cosf(a);
    fac[2 * i + 1] = sinf(a);
  }
  for (int j = 0; j < nr; j++) {
    float a, r;
    polar_angle_radius(0, j, nb, nr, ang_res, resolution, a, r);
    fac[2 * nb + j] = r;
  }
  return TDR_OK;
}
