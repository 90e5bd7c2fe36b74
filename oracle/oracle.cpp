// oracle.cpp — TEST INFRASTRUCTURE ONLY (never shipped, never imported by the product path).
//
// CPU restatement of the per-scan particle-filter update of KumarRobotics/top_down_renderer,
// written from the reference's sources as text (the reference itself cannot be built here: it needs
// Eigen, PCL, OpenCV, ROS and TBB, none of which are in the image).  Each function cites the
// reference file:line it follows.  Plain C++17 + libstdc++ <random> (the reference's own RNG library),
// exported with a C ABI so tests/bench can drive it through ctypes.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
// (SURVEY.md §4, §8c).  This restatement is cross-checked by an independent NumPy statement of the
// same maths (oracle/np_oracle.py) and by the committed fixtures under tests/golden/, but nothing
// produced by the reference itself pins it.
//
// Arithmetic conventions (see DESIGN.md "Oracle arithmetic"):
//  * compile with -ffp-contract=off: the reference's CMake sets no -march/-O flags, so x86-64 baseline
//    code has no FMA contraction;
//  * Eigen float `.sum()` reductions have unspecified order -> accumulated here in double and rounded
//    to float once (1e-5 tolerance absorbs the difference);
//  * serial float loops of the reference (particle_filter.cpp:108-126, 175-183) are kept serial float;
//  * col-major images/maps: element (i,j) of an (rows x cols) array is at i + rows*j.
//  * the three `for (int i; ...)` loops of particle_filter.cpp:110,120,138 start at i = 0.

#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <limits>
#include <random>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" {

// Same layout as the reference's `State` (include/top_down_render/state_particle.h:9-17): 6 floats + bool, 28 B.
struct orc_state {
  float init_x_px, init_y_px, dx_m, dy_m, theta, scale;
  uint8_t have_init;
  uint8_t pad_[3];
};

// POD mirror of `FilterParams` (state_particle.h:19-38); class_weights inlined (max 16 classes).
struct orc_filter_params {
  float pos_cov, theta_cov, regularization;
  float init_pos_px_x, init_pos_px_y, init_pos_px_cov;
  float init_pos_m_x, init_pos_m_y, init_pos_deg_theta, init_pos_deg_cov;
  int32_t force_on_map;
  float fixed_scale, scale_log_min, scale_log_max;
  int32_t num_classes;
  float class_weights[16];
};

struct orc_map {
  const float* class_maps;   // [ncls][H*W], col-major (r + H*c)
  const uint8_t* class_mask; // [H*W] col-major, 1 = unknown
  int32_t ncls, rows, cols;  // rows = H (y), cols = W (x)
  float resolution;
};

// ---- the unpinned overload choices, as switches (CPU sensitivity study only: tools/overload_sensitivity.py) ----------
// The reference calls UNQUALIFIED atan2 / sqrt on floats (src/scan_renderer_polar.cpp:32-33, 97-98) and unqualified
// cos / sin / atan2 in meanLikelihood (src/particle_filter.cpp:198-202).  Which function that is depends on the include
// graph of its translation units, which is not in this image: with only <cmath> in scope the global names are the C
// library's double functions — (float)atan2((double)x, (double)y) — while libstdc++'s <math.h> wrapper pulls the float
// overloads into the global namespace.  Default here (and what the HIP path implements): the float overloads.
// bit 0: A1 / A3 polar raster through the double functions; bit 1: meanLikelihood through the double functions.
static int g_overload_mode = 0;
void orc_set_overload_mode(int mode) { g_overload_mode = mode; }
int orc_get_overload_mode() { return g_overload_mode; }
static inline float orc_atan2_raster(float x, float y) {
  return (g_overload_mode & 1) ? (float)atan2((double)x, (double)y) : atan2f(x, y);
}
static inline float orc_hypot_raster(float x, float y) {   // sqrt(pt.x*pt.x + pt.y*pt.y): the argument is a float either way
  return (g_overload_mode & 1) ? (float)sqrt((double)(x * x + y * y)) : sqrtf(x * x + y * y);
}

// ------------------------------------------------------------------------------------------------
// A1  ScanRendererPolar::renderSemanticTopDown   (src/scan_renderer_polar.cpp:83-109)
// pts: n points, `stride` floats apart, x,y,z at [0..2], class id ("intensity") at [ioff].
// imgs: [ncls][nb*nr] col-major, zero-filled here like :88-90.
// Deviation (documented): the LUT index is bounds-checked to 0..255 instead of UB.
void orc_raster_polar(const float* pts, int stride, int ioff, long n, float res, float ang_res,
                      const int32_t* lut256, int ncls, int nb, int nr, float* imgs) {
  if (ncls < 1) return;                                            // :85
  std::memset(imgs, 0, sizeof(float) * (size_t)ncls * nb * nr);    // :88-90
  for (long k = 0; k < n; k++) {
    const float* p = pts + (size_t)k * stride;
    float x = p[0], y = p[1];
    if (x == 0 && y == 0) continue;                                // :95
    float theta = orc_atan2_raster(x, y);                          // :97 (argument order x,y)
    float r = orc_hypot_raster(x, y);                              // :98
    // :100  std::round(float) + int -> float sum, truncated to int
    int theta_ind = (int)(roundf(theta / ang_res) + (float)(nb / 2));
    int r_ind = (int)roundf(r / res);                              // :101
    if (theta_ind >= 0 && theta_ind < nb && r_ind >= 0 && r_ind < nr) {
      int pt_class = (int)p[ioff];                                 // :103
      if (pt_class < 0 || pt_class > 255) continue;
      int c = lut256[pt_class];
      if (c >= 0 && c < ncls) imgs[(size_t)c * nb * nr + theta_ind + (size_t)nb * r_ind] += 1;  // :104-106
    }
  }
}

// A2  ScanRenderer::renderSemanticTopDown   (src/scan_renderer.cpp:55-78)
// imgs: [ncls][rows*cols] col-major; img_size = (cols, rows) (:58).
void orc_raster_cart(const float* pts, int stride, int ioff, long n, float res,
                     const int32_t* lut256, int ncls, int rows, int cols, float* imgs) {
  if (ncls < 1) return;
  std::memset(imgs, 0, sizeof(float) * (size_t)ncls * rows * cols);
  for (long k = 0; k < n; k++) {
    const float* p = pts + (size_t)k * stride;
    float x = p[0], y = p[1];
    if (x == 0 && y == 0) continue;                                // :67
    int x_ind = (int)(roundf(x / res) + (float)(cols / 2));        // :69
    int y_ind = (int)(roundf(y / res) + (float)(rows / 2));        // :70
    if (x_ind >= 0 && x_ind < cols && y_ind >= 0 && y_ind < rows) {
      int pt_class = (int)p[ioff];
      if (pt_class < 0 || pt_class > 255) continue;
      int c = lut256[pt_class];
      if (c >= 0 && c < ncls) imgs[(size_t)c * rows * cols + y_ind + (size_t)rows * x_ind] += 1;  // :73-75
    }
  }
}

// A3  ScanRendererPolar::renderGeometricTopDown   (src/scan_renderer_polar.cpp:6-81)
// The cloud is read as cloud->at(idx, idy) with idx < width outer, idy < height inner (:27-29): element idy*width + idx.
// imgs: [2][nb*nr] col-major: [0] ground, [1] obstacles.
// x86 float -> int conversion (cvttss2si): NaN and out-of-range give INT_MIN.
static inline int cvt_x86(float v) { return (v >= -2147483648.f && v < 2147483648.f) ? (int)v : INT32_MIN; }
// Conventions where the reference leaves a choice (documented in DESIGN.md):
//  * std::sort (:49-51) is not stable; ties in r keep their input order here (std::stable_sort);
//  * a point whose x or y is not finite would index ang_bins with (int)NaN (undefined behaviour): it is dropped;
//  * unqualified atan2 / sqrt / abs on floats are taken as the float overloads, like in A1.
void orc_raster_geo_polar(const float* pts, int stride, long width, long height, float res, float ang_res, int nb,
                          int nr, float* imgs) {
  std::memset(imgs, 0, sizeof(float) * (size_t)2 * nb * nr);       // :11-13
  struct P { float x, y, z, r; };
  std::vector<std::vector<P>> bins((size_t)nb);                    // :17-21
  for (long idx = 0; idx < width; idx++)
    for (long idy = 0; idy < height; idy++) {
      const float* p = pts + (size_t)(idy * width + idx) * stride;
      const float x = p[0], y = p[1], z = p[2];
      if (x == 0 && y == 0) continue;                              // :30
      if (!std::isfinite(x) || !std::isfinite(y)) continue;
      const float theta = orc_atan2_raster(x, y);                  // :32
      const float r = orc_hypot_raster(x, y);                      // :33
      float t = roundf(theta / ang_res) + (float)(nb / 2);         // :36-37 std::clamp<float>(.., 0, rows-1)
      t = t < 0.f ? 0.f : (t > (float)(nb - 1) ? (float)(nb - 1) : t);
      bins[(size_t)(int)t].push_back(P{x, y, z, r});               // :39
    }
  for (int theta_ind = 0; theta_ind < nb; theta_ind++) {
    std::vector<P>& bin = bins[(size_t)theta_ind];
    std::stable_sort(bin.begin(), bin.end(), [](const P& a, const P& b) { return a.r > b.r; });   // :49-51
    float lx = 0, ly = 0, lz = 0;                                  // :54
    bool last_high_grad = false;
    int last_r_ind = 0;
    for (const P& pt : bin) {
      const float dx = pt.x - lx, dy = pt.y - ly;
      const float dist = sqrtf(dx * dx + dy * dy);                 // :58
      const float slope = fabsf(pt.z - lz) / dist;                 // :59
      const int r_ind = cvt_x86(roundf(pt.r / res));               // :60
      if (slope > 1) {                                             // :62-66
        if (r_ind >= 0 && r_ind < nr) imgs[(size_t)nb * nr + theta_ind + (size_t)nb * r_ind] += 1;
        last_high_grad = true;
      } else if ((double)slope < 0.3 && last_high_grad == false) { // :67-72
        // (last_r_ind is negative only after a return 2^31 bins away, where the reference writes in front of its image)
        for (long i = last_r_ind < 0 ? 0 : last_r_ind; i <= r_ind && i < nr; i++) imgs[theta_ind + (size_t)nb * i] += 1;
      } else {
        last_high_grad = false;                                    // :73-75
      }
      lx = pt.x; ly = pt.y; lz = pt.z;                             // :76-77
      last_r_ind = r_ind;
    }
  }
}

// A3  ScanRenderer::renderGeometricTopDown   (src/scan_renderer.cpp:7-53): every column idx of the organised cloud is
// one vertical scan line walked upwards; ground cells are filled along the line between consecutive returns.
// imgs: [2][rows*cols] col-major, img_size = (cols, rows) (:10).  Non-finite x / y: dropped (their indices are INT_MIN
// in the reference and the line interpolation then overflows: undefined behaviour).
void orc_raster_geo_cart(const float* pts, int stride, long width, long height, float res, int rows, int cols,
                         float* imgs) {
  std::memset(imgs, 0, sizeof(float) * (size_t)2 * rows * cols);
  for (long idx = 0; idx < width; idx++) {
    float lx = 0, ly = 0, lz = 0;                                  // :17
    int last_x = cols / 2, last_y = rows / 2;                      // :19
    bool last_high_grad = false;
    for (long idy = 0; idy < height; idy++) {                      // :23
      const float* p = pts + (size_t)(idy * width + idx) * stride;
      const float x = p[0], y = p[1], z = p[2];
      if (x == 0 && y == 0) continue;                              // :26
      if (!std::isfinite(x) || !std::isfinite(y)) continue;
      const int x_ind = cvt_x86(roundf(x / res) + (float)(cols / 2));   // :27
      const int y_ind = cvt_x86(roundf(y / res) + (float)(rows / 2));   // :28
      // returns more than 2^24 cells away: the interpolation loop below overflows / does not terminate in the reference
      const int lim = 1 << 24;
      if (x_ind > lim || x_ind < -lim || y_ind > lim || y_ind < -lim) continue;
      const float dx = x - lx, dy = y - ly;
      const float dist = sqrtf(dx * dx + dy * dy);                 // :30
      const float slope = fabsf(z - lz) / dist;                    // :31
      if (slope > 1) {                                             // :32-36
        if (x_ind >= 0 && x_ind < cols && y_ind >= 0 && y_ind < rows) imgs[(size_t)rows * cols + y_ind + (size_t)rows * x_ind] += 1;
        last_high_grad = true;
      } else if ((double)slope < 0.3 && last_high_grad == false) { // :37-45
        const long long ddx = (long long)x_ind - last_x, ddy = (long long)y_ind - last_y;
        const int nrm = (int)std::sqrt((double)(ddx * ddx + ddy * ddy));   // Vector2i::norm(): integer sqrt, truncated
        for (float i = 0; i < 1; i = (float)((double)i + 1. / nrm)) {      // :39
          const int ix = (int)roundf((float)last_x + i * (float)ddx);
          const int iy = (int)roundf((float)last_y + i * (float)ddy);
          if (ix >= 0 && ix < cols && iy >= 0 && iy < rows) imgs[iy + (size_t)rows * ix] += 1;   // :41-44
        }
      } else {
        last_high_grad = false;                                    // :46-48
      }
      lx = x; ly = y; lz = z;                                      // :49
      last_x = x_ind; last_y = y_ind;                              // :50
    }
  }
}

// Eigen's LinSpaced<float>(n, low, high) coefficient i (linspaced_op_impl, non-integer branch).
static inline float linspaced(int i, int n, float low, float high) {
  int size1 = (n == 1) ? 1 : n - 1;
  float step = (n == 1) ? 0.0f : (high - low) / (float)(n - 1);
  bool flip = fabsf(high) < fabsf(low);
  if (flip) return (i == 0) ? low : (high - (float)(size1 - i) * step);
  return (i == size1) ? high : (low + (float)i * step);
}

// A4a  TopDownMap::samplePts   (src/top_down_map.cpp:367-389)
// pts: 2 x (rows*cols), stored interleaved like Eigen::Array2Xf (row0 at 2k, row1 at 2k+1), k = i + rows*j.
// Semantics derived from Eigen's coefficient-wise evaluation of the strided maps with assertions off:
//   pts(0,k) = L_rows[i], pts(1,k) = L_cols[j]  (:376-379); pts = R(rot)*pts (:382-385);
//   row0 += center[1], row1 += center[0] (:387-388).
void orc_sample_pts(float cx, float cy, float rot, float* pts, int cols, int rows, float res) {
  // `-res*(rows-1)/2.` : float*int -> float, /2. -> double, then narrowed to float by LinSpaced's Scalar args
  float lo_r = (float)((double)(-res * (float)(rows - 1)) / 2.), hi_r = (float)((double)(res * (float)(rows - 1)) / 2.);
  float lo_c = (float)((double)(-res * (float)(cols - 1)) / 2.), hi_c = (float)((double)(res * (float)(cols - 1)) / 2.);
  float c = cosf(rot), s = sinf(rot);
  for (int j = 0; j < cols; j++) {
    for (int i = 0; i < rows; i++) {
      size_t k = (size_t)i + (size_t)rows * j;
      float p0 = linspaced(i, rows, lo_r, hi_r);
      float p1 = linspaced(j, cols, lo_c, hi_c);
      float q0 = c * p0 + (-s) * p1;   // rotm*pts, :383-385
      float q1 = s * p0 + c * p1;
      pts[2 * k] = q0 + cy;            // x_vals += center[1]
      pts[2 * k + 1] = q1 + cx;        // y_vals += center[0]
    }
  }
}

// A4b  TopDownMapPolar::samplePtsPolar   (src/top_down_map_polar.cpp:7-19)
// tab: 2 x (nb*nr) interleaved; tab(0,k) = cos(theta_i)*r_j, tab(1,k) = sin(theta_i)*r_j, k = i + nb*j.
void orc_polar_table(int nb, int nr, float ang_res, float resolution, float* tab) {
  size_t P = (size_t)nb * nr;
  std::vector<float> ang(2 * P);
  orc_sample_pts(0.f, 0.f, 0.f, ang.data(), /*cols=*/nr, /*rows=*/nb, 1.f);  // :10
  float first = ang[1];
  float inv_res = (float)(1. / (double)resolution);                          // :14 (`*= 1./resolution`, Scalar=float)
  for (size_t k = 0; k < P; k++) {
    float a = ang[2 * k];
    float r = ang[2 * k + 1] + (-first);                                     // :11
    a = a * ang_res;                                                         // :13
    r = r * inv_res;                                                         // :14
    tab[2 * k] = cosf(a) * r;                                                // :17
    tab[2 * k + 1] = sinf(a) * r;                                            // :18
  }
}

// A5  TopDownMapPolar::getLocalMap(center, scale, res, dists, mask)   (src/top_down_map_polar.cpp:21-53)
// dists: [ncls][P]; maskout: [P] (1 = unknown / out of bounds).
void orc_local_map_polar(const orc_map* m, const float* tab, long P, float cx, float cy, float scale,
                         float res, float* dists, uint8_t* maskout) {
  float off0 = cy / m->resolution, off1 = cx / m->resolution;               // :29-30
  for (long k = 0; k < P; k++) {
    float p0 = (tab[2 * k] * scale) * res;                                    // :28
    float p1 = (tab[2 * k + 1] * scale) * res;
    p0 += off0;
    p1 += off1;
    int ri = (int)roundf(p0), ci = (int)roundf(p1);                          // :31
    bool in = ri >= 0 && ri < m->rows && ci >= 0 && ci < m->cols;
    for (int c = 0; c < m->ncls; c++)                                        // :33-42
      dists[(size_t)c * P + k] = in ? m->class_maps[(size_t)c * m->rows * m->cols + ri + (size_t)m->rows * ci] : 0.f;
    maskout[k] = in ? m->class_mask[ri + (size_t)m->rows * ci] : 1;          // :44-52
  }
}

// A7  TopDownMap::getLocalMap(center, rot, res, dists, mask)   (src/top_down_map.cpp:429-459)
void orc_local_map_cart(const orc_map* m, float cx, float cy, float rot, float res, int rows, int cols,
                        float* dists, uint8_t* maskout) {
  long P = (long)rows * cols;
  std::vector<float> pts(2 * (size_t)P);
  orc_sample_pts(cx / m->resolution, cy / m->resolution, rot, pts.data(), cols, rows, res / m->resolution);  // :433-434
  for (long k = 0; k < P; k++) {
    int ri = (int)roundf(pts[2 * k]), ci = (int)roundf(pts[2 * k + 1]);     // :437
    bool in = ri >= 0 && ri < m->rows && ci >= 0 && ci < m->cols;
    for (int c = 0; c < m->ncls; c++)
      dists[(size_t)c * P + k] = in ? m->class_maps[(size_t)c * m->rows * m->cols + ri + (size_t)m->rows * ci] : 0.f;
    maskout[k] = in ? m->class_mask[ri + (size_t)m->rows * ci] : 1;
  }
}

// TopDownMap::getClassesAtPoint(Vector2i)   (src/top_down_map.cpp:159-170); returns a bitmask of classes.
uint32_t orc_classes_at_point(const orc_map* m, int px, int py) {
  int c0 = (int)((float)px / m->resolution), c1 = (int)((float)py / m->resolution);   // :160
  uint32_t bits = 0;
  for (int cls = 0; cls < m->ncls; cls++)
    if (c0 < m->cols && c1 < m->rows && c0 >= 0 && c1 >= 0)
      if (m->class_maps[(size_t)cls * m->rows * m->cols + c1 + (size_t)m->rows * c0] < 1) bits |= 1u << cls;
  return bits;
}

// rotation -> circular bin shift   (src/state_particle.cpp:123-128)
int orc_rot_shift(float rot, int num_bins) {
  // rot*num_bins/2/M_PI : (float*int -> float)/2 -> float, /M_PI -> double
  int rot_shift = (int)std::round((double)(rot * (float)num_bins / 2) / M_PI);
  while (rot_shift >= num_bins) rot_shift -= num_bins;
  while (rot_shift < 0) rot_shift += num_bins;
  return rot_shift;
}

// A9  StateParticle::getCostForRot   (src/state_particle.cpp:112-155)
// scan, classes: [ncls][nb*nr] col-major; maskf: [nb*nr] = 1 - mask.
float orc_cost_for_rot(const float* scan, const float* classes, const float* maskf, int ncls, int nb, int nr,
                       const float* class_weights, float rot) {
  long P = (long)nb * nr;
  double msum = 0;
  for (long k = 0; k < P; k++) msum += maskf[k];
  if ((float)msum / (float)P < 0.5) return std::numeric_limits<float>::quiet_NaN();   // :117-120
  int s = orc_rot_shift(rot, nb);
  float cost = 0, normalization = 0;
  for (int c = 0; c < ncls; c++) {
    const float* sc = scan + (size_t)c * P;
    const float* cl = classes + (size_t)c * P;
    // scan.topRows(s) pairs with window.bottomRows(s): scan row a<s  <-> window row nb-s+a
    double top = 0, bot = 0, ntop = 0, nbot = 0;
    for (int j = 0; j < nr; j++) {
      for (int a = 0; a < s; a++) {
        top += (double)(sc[a + (size_t)nb * j] * cl[(nb - s + a) + (size_t)nb * j]);
        ntop += (double)(sc[a + (size_t)nb * j] * maskf[(nb - s + a) + (size_t)nb * j]);
      }
      // scan.bottomRows(nb-s) pairs with window.topRows(nb-s): scan row a>=s <-> window row a-s
      for (int a = s; a < nb; a++) {
        bot += (double)(sc[a + (size_t)nb * j] * cl[(a - s) + (size_t)nb * j]);
        nbot += (double)(sc[a + (size_t)nb * j] * maskf[(a - s) + (size_t)nb * j]);
      }
    }
    // :136-139  float sum * 0.01 (double) * float weight, accumulated into float `cost`
    cost = (float)((double)cost + (double)(float)top * 0.01 * (double)class_weights[c]);
    cost = (float)((double)cost + (double)(float)bot * 0.01 * (double)class_weights[c]);
    normalization += (float)ntop;                                            // :141
    normalization += (float)nbot;                                            // :142
  }
  return cost / normalization;                                               // :154
}

// A8  StateParticle::computeWeight for every particle   (src/state_particle.cpp:157-219, driven by
//     particle_filter.cpp:104-105; parallel over particles exactly where the reference is).
// scan: [ncls][nb*nr] col-major.  Mutates theta/have_init of un-initialised particles (:205-206).
// Particles gated out at :163-176 keep weight 0.
void orc_compute_weights(const orc_map* m, const float* tab, int nb, int nr, const float* scan, float res,
                         const orc_filter_params* fp, orc_state* states, long n, float* weights, int nthreads) {
  long P = (long)nb * nr;
  int ncls = m->ncls;
  float width = (float)m->cols * m->resolution, height = (float)m->rows * m->resolution;   // state_particle.cpp:11,46-47
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    std::vector<float> classes((size_t)ncls * P), maskf(P);
    std::vector<uint8_t> mask(P);
#pragma omp for schedule(dynamic, 16)
    for (long p = 0; p < n; p++) {
      orc_state& st = states[p];
      float cx = st.dx_m * st.scale + st.init_x_px;                          // :161
      float cy = st.dy_m * st.scale + st.init_y_px;                          // :162
      if (fp->force_on_map) {
        if (cx < 0 || cy < 0 || cx > width || cy > height) { weights[p] = 0; continue; }   // :163-168
      }
      if (fp->fixed_scale < 0) {
        if ((double)st.scale < std::pow(10., (double)fp->scale_log_min) ||
            (double)st.scale > std::pow(10., (double)fp->scale_log_max)) { weights[p] = 0; continue; }  // :169-176
      }
      orc_local_map_polar(m, tab, P, cx, cy, st.scale, res, classes.data(), mask.data());   // :188
      for (long k = 0; k < P; k++) maskf[k] = 1.f - (float)mask[k];          // :199,209
      float best_cost = std::numeric_limits<float>::max();                   // :193
      float best_theta = 0;
      if (!st.have_init) {
        for (float t = 0; t < 2 * M_PI; t += 2 * M_PI / 40) {                // :197
          float cost = orc_cost_for_rot(scan, classes.data(), maskf.data(), ncls, nb, nr, fp->class_weights, t);
          if (cost < best_cost) { best_cost = cost; best_theta = t; }        // :200-203
        }
        st.theta = best_theta;                                               // :205
        st.have_init = 1;                                                    // :206
      } else {
        best_cost = orc_cost_for_rot(scan, classes.data(), maskf.data(), ncls, nb, nr, fp->class_weights, st.theta);
      }
      weights[p] = (float)(1. / (double)(best_cost + fp->regularization));   // :212
    }
  }
}

// N4  getCostForRot WITH its geometric block (src/state_particle.cpp:145-152, commented out in the reference) and
// computeWeight around it: geo_scan [2][nb*nr] are the images of renderGeometricTopDown, geo (an orc_map with 2 "classes"
// = geo_maps_[0..1], mask all zero) is gathered by getLocalGeoMap (src/top_down_map_polar.cpp:55-76) — the same addressing
// as getLocalMap.
float orc_cost_for_rot_geo(const float* scan, const float* geo_scan, const float* classes, const float* geo_cls,
                           const float* maskf, int ncls, int nb, int nr, const float* class_weights, float rot) {
  long P = (long)nb * nr;
  double msum = 0;
  for (long k = 0; k < P; k++) msum += maskf[k];
  if ((float)msum / (float)P < 0.5) return std::numeric_limits<float>::quiet_NaN();   // :117-120
  int s = orc_rot_shift(rot, nb);
  float cost = 0, normalization = 0;
  for (int c = 0; c < ncls; c++) {                                             // :132-143
    const float* sc = scan + (size_t)c * P;
    const float* cl = classes + (size_t)c * P;
    double top = 0, bot = 0, ntop = 0, nbot = 0;
    for (int j = 0; j < nr; j++) {
      for (int a = 0; a < s; a++) {
        top += (double)(sc[a + (size_t)nb * j] * cl[(nb - s + a) + (size_t)nb * j]);
        ntop += (double)(sc[a + (size_t)nb * j] * maskf[(nb - s + a) + (size_t)nb * j]);
      }
      for (int a = s; a < nb; a++) {
        bot += (double)(sc[a + (size_t)nb * j] * cl[(a - s) + (size_t)nb * j]);
        nbot += (double)(sc[a + (size_t)nb * j] * maskf[(a - s) + (size_t)nb * j]);
      }
    }
    cost = (float)((double)cost + (double)(float)top * 0.01 * (double)class_weights[c]);
    cost = (float)((double)cost + (double)(float)bot * 0.01 * (double)class_weights[c]);
    normalization += (float)ntop;
    normalization += (float)nbot;
  }
  for (int i = 0; i < 2; i++) {                                                // :146-151
    const float* sc = geo_scan + (size_t)i * P;
    const float* cl = geo_cls + (size_t)i * P;
    double top = 0, bot = 0, total = 0;
    for (int j = 0; j < nr; j++) {
      for (int a = 0; a < s; a++) top += (double)(sc[a + (size_t)nb * j] * cl[(nb - s + a) + (size_t)nb * j]);
      for (int a = s; a < nb; a++) bot += (double)(sc[a + (size_t)nb * j] * cl[(a - s) + (size_t)nb * j]);
    }
    for (long k = 0; k < P; k++) total += sc[k];
    cost = (float)((double)cost + (double)(float)top * 0.01);                 // :148
    cost = (float)((double)cost + (double)(float)bot * 0.01);                 // :149
    normalization += (float)total;                                             // :150
  }
  return cost / normalization;
}
void orc_compute_weights_geo(const orc_map* m, const orc_map* geo, const float* tab, int nb, int nr, const float* scan,
                             const float* geo_scan, float res, const orc_filter_params* fp, orc_state* states, long n,
                             float* weights, int nthreads) {
  long P = (long)nb * nr;
  int ncls = m->ncls;
  float width = (float)m->cols * m->resolution, height = (float)m->rows * m->resolution;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    std::vector<float> classes((size_t)ncls * P), geo_cls(2 * (size_t)P), maskf(P);
    std::vector<uint8_t> mask(P), gmask(P);
#pragma omp for schedule(dynamic, 16)
    for (long p = 0; p < n; p++) {
      orc_state& st = states[p];
      float cx = st.dx_m * st.scale + st.init_x_px;
      float cy = st.dy_m * st.scale + st.init_y_px;
      if (fp->force_on_map) {
        if (cx < 0 || cy < 0 || cx > width || cy > height) { weights[p] = 0; continue; }
      }
      if (fp->fixed_scale < 0) {
        if ((double)st.scale < std::pow(10., (double)fp->scale_log_min) ||
            (double)st.scale > std::pow(10., (double)fp->scale_log_max)) { weights[p] = 0; continue; }
      }
      orc_local_map_polar(m, tab, P, cx, cy, st.scale, res, classes.data(), mask.data());      // :188
      orc_local_map_polar(geo, tab, P, cx, cy, st.scale, res, geo_cls.data(), gmask.data());   // :189 getLocalGeoMap
      for (long k = 0; k < P; k++) maskf[k] = 1.f - (float)mask[k];
      float best_cost = std::numeric_limits<float>::max();
      float best_theta = 0;
      if (!st.have_init) {
        for (float t = 0; t < 2 * M_PI; t += 2 * M_PI / 40) {
          float cost = orc_cost_for_rot_geo(scan, geo_scan, classes.data(), geo_cls.data(), maskf.data(), ncls, nb, nr,
                                            fp->class_weights, t);
          if (cost < best_cost) { best_cost = cost; best_theta = t; }
        }
        st.theta = best_theta;
        st.have_init = 1;
      } else {
        best_cost = orc_cost_for_rot_geo(scan, geo_scan, classes.data(), geo_cls.data(), maskf.data(), ncls, nb, nr,
                                         fp->class_weights, st.theta);
      }
      weights[p] = (float)(1. / (double)(best_cost + fp->regularization));
    }
  }
}

// Cartesian score used for BASELINE config 4: the reference has no Cartesian score function;
// defined as A9 with shift 0 on a window sampled by A7 with rot = theta, res as given (scale folded by caller).
void orc_compute_weights_cart(const orc_map* m, int rows, int cols, const float* scan, float res,
                              const orc_filter_params* fp, const orc_state* states, long n, float* weights, int nthreads) {
  long P = (long)rows * cols;
  int ncls = m->ncls;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    std::vector<float> classes((size_t)ncls * P), maskf(P);
    std::vector<uint8_t> mask(P);
#pragma omp for schedule(dynamic, 4)
    for (long p = 0; p < n; p++) {
      const orc_state& st = states[p];
      float cx = st.dx_m * st.scale + st.init_x_px, cy = st.dy_m * st.scale + st.init_y_px;
      orc_local_map_cart(m, cx, cy, st.theta, res * st.scale, rows, cols, classes.data(), mask.data());
      for (long k = 0; k < P; k++) maskf[k] = 1.f - (float)mask[k];
      float cost = orc_cost_for_rot(scan, classes.data(), maskf.data(), ncls, rows, cols, fp->class_weights, 0.f);
      weights[p] = (float)(1. / (double)(cost + fp->regularization));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// RNG: one shared std::mt19937 consumed serially (particle_filter.h:52, state_particle.h:63).
// The reference seeds from std::random_device (particle_filter.cpp:4-5); here the seed is explicit.
void* orc_rng_create(uint32_t seed) { return new std::mt19937(seed); }
void orc_rng_destroy(void* h) { delete (std::mt19937*)h; }
float orc_rng_uniform(void* h) {                                             // particle_filter.cpp:172-173
  std::uniform_real_distribution<float> d(0., 1.);
  return d(*(std::mt19937*)h);
}

// A10/A11  ParticleFilter::propagate -> StateParticle::propagate, in index order
//          (src/particle_filter.cpp:86-92, src/state_particle.cpp:57-78)
void orc_propagate(orc_state* states, float* last_dist, long n, float tx, float ty, float omega, int scale_freeze,
                   const orc_filter_params* fp, void* rng) {
  std::mt19937& gen = *(std::mt19937*)rng;
  for (long p = 0; p < n; p++) {
    orc_state& st = states[p];
    float c = cosf(st.theta), s = sinf(st.theta);                            // Rotation2D<float>(theta) :58
    float gx = c * tx + (-s) * ty;
    float gy = s * tx + c * ty;
    float lx = st.dx_m, ly = st.dy_m;                                        // :59
    st.dx_m += gx;                                                           // :60-61
    st.dy_m += gy;
    float dist = sqrtf(gx * gx + gy * gy);                                   // :63
    std::normal_distribution<float> disp_dist{0, fp->pos_cov * dist};        // :64
    std::normal_distribution<float> theta_dist{0, fp->theta_cov * dist};     // :65
    st.theta += theta_dist(gen) + omega;                                     // :67
    st.dx_m += disp_dist(gen);                                               // :68
    st.dy_m += disp_dist(gen);                                               // :69
    if (!scale_freeze) {
      std::normal_distribution<float> scale_dist{1, static_cast<float>(std::min(2. / dist, 0.02))};  // :72
      st.scale *= scale_dist(gen);                                           // :73
    }
    float mx = lx - st.dx_m, my = ly - st.dy_m;                              // :76
    last_dist[p] = sqrtf(mx * mx + my * my);                                 // :77
  }
}

// Standard normals in the exact order A10 consumes them: z[p] = {z_theta, z_dx, z_dy, z_scale}
// (a N(mu,sigma) draw of libstdc++ is z*sigma+mu in float).  Used to drive the GPU propagate in parity mode.
void orc_propagate_normals(long n, int scale_freeze, float* z4, void* rng) {
  std::mt19937& gen = *(std::mt19937*)rng;
  for (long p = 0; p < n; p++) {
    std::normal_distribution<float> disp{0, 1}, th{0, 1};
    z4[4 * p + 0] = th(gen);
    z4[4 * p + 1] = disp(gen);
    z4[4 * p + 2] = disp(gen);
    if (!scale_freeze) {
      std::normal_distribution<float> sc{0, 1};
      z4[4 * p + 3] = sc(gen);
    } else {
      z4[4 * p + 3] = 0;
    }
  }
}

// A12  ParticleFilter::update, weights & statistics   (src/particle_filter.cpp:107-147)
// raw: per-particle StateParticle::weight(); w_out: final normalised weights; returns argmax index.
// stats_out (optional, 4 floats): sum, mean, bottom_stddev, fallback_flag.
long orc_update_weights(const float* raw, const float* last_dist, long n, float* w_out, float* stats_out) {
  float sum = 0;
  int num_valid = 0;
  for (long i = 0; i < n; i++) {                                             // :110-116 (i from 0)
    w_out[i] = raw[i];
    if (!std::isnan(w_out[i])) { sum += w_out[i]; ++num_valid; }
  }
  float mean = sum / num_valid;                                              // :117
  float bottom_stddev = 0;
  int num_under = 0;
  for (long i = 0; i < n; i++) {                                             // :120-125
    if (!std::isnan(w_out[i]) && w_out[i] < mean) {
      bottom_stddev += std::pow(w_out[i] - mean, 2);                         // float - float, pow in double, += into float
      ++num_under;
    }
  }
  bottom_stddev = std::sqrt(bottom_stddev / num_under);                      // :126
  bool fallback = (sum == 0 || num_under < 1);                               // :129
  if (fallback) {
    for (long i = 0; i < n; i++) w_out[i] = 1;                               // :130
  } else {
    float fill = mean - bottom_stddev;                                       // :133
    for (long i = 0; i < n; i++) if (std::isnan(w_out[i])) w_out[i] = fill;
  }
  double s1 = 0;
  for (long i = 0; i < n; i++) s1 += w_out[i];
  float fs1 = (float)s1;
  for (long i = 0; i < n; i++) w_out[i] = w_out[i] / fs1;                    // :135
  for (long i = 0; i < n; i++) {                                             // :138-141
    float d = std::min<float>(last_dist[i] * 5, 1);
    w_out[i] = d * w_out[i] + (1 - d) / (float)n;
  }
  double s2 = 0;
  for (long i = 0; i < n; i++) s2 += w_out[i];
  float fs2 = (float)s2;
  for (long i = 0; i < n; i++) w_out[i] = w_out[i] / fs2;                    // :142
  long best = 0;                                                             // :145-147 (first maximum)
  for (long i = 1; i < n; i++) if (w_out[i] > w_out[best]) best = i;
  if (stats_out) { stats_out[0] = sum; stats_out[1] = mean; stats_out[2] = bottom_stddev; stats_out[3] = fallback ? 1.f : 0.f; }
  return best;
}

// A13  adaptive particle count   (src/particle_filter.cpp:151-157); covs: ngauss x (2x2 row-major position cov).
int orc_adaptive_count(const float* covs2x2, int ngauss, int last_num, int max_num) {
  int num = 0;
  for (int g = 0; g < ngauss; g++) {
    float a = covs2x2[4 * g], b = covs2x2[4 * g + 1], c = covs2x2[4 * g + 2], d = covs2x2[4 * g + 3];
    // eigenvalues of a general 2x2 (real parts), Eigen's .eigenvalues() :154
    float tr = a + d, det = a * d - b * c;
    float disc = tr * tr / 4 - det;
    float e0, e1;
    if (disc >= 0) { float q = sqrtf(disc); e0 = tr / 2 - q; e1 = tr / 2 + q; } else { e0 = e1 = tr / 2; }
    num += (int)(sqrtf(e0) * sqrtf(e1));                                     // :155
  }
  return std::min(std::max(num, 3 * last_num / 4 + 10), max_num);            // :157
}

// A14  systematic resample, literal O(N*N') form   (src/particle_filter.cpp:172-185)
void orc_resample_literal(const float* w, long n, long n_new, float shift, int32_t* idx) {
  for (long i = 0; i < n_new; i++) {
    float running_sum = 0;
    float sample = ((float)i + shift) / (float)n_new;                        // :176
    long j = 0;
    for (; j < n; j++) {
      running_sum += w[j];                                                   // :179
      if (running_sum > sample || j == n - 1) break;                         // :180
    }
    idx[i] = (int32_t)j;
  }
}

// Same indices in O(N + N' log N): one serial float prefix (identical additions, identical order), then the first
// exceedance per sample via a running maximum (weights may be negative after the NaN fill, :133).
void orc_resample_prefix(const float* w, long n, long n_new, float shift, int32_t* idx) {
  std::vector<float> runmax(n);
  float running_sum = 0, mx = -std::numeric_limits<float>::infinity();
  for (long j = 0; j < n; j++) {
    running_sum += w[j];
    if (running_sum > mx) mx = running_sum;   // NaN never raises the max (NaN > x is false), like `running_sum > sample`
    runmax[j] = mx;
  }
  for (long i = 0; i < n_new; i++) {
    float sample = ((float)i + shift) / (float)n_new;
    // first j with runmax[j] > sample, else n-1
    long lo = 0, hi = n - 1;
    while (lo < hi) {
      long mid = (lo + hi) / 2;
      if (runmax[mid] > sample) hi = mid; else lo = mid + 1;
    }
    idx[i] = (int32_t)lo;
  }
}

// `new_particles_[i]->setState(particles_[j]->state())`   (src/particle_filter.cpp:184)
void orc_gather_states(const orc_state* src, const int32_t* idx, long n_new, orc_state* dst) {
  for (long i = 0; i < n_new; i++) dst[i] = src[idx[i]];
}

// StateParticle::mlState   (src/state_particle.cpp:98-102)
static inline void ml_state(const orc_state& s, float out[4]) {
  out[0] = s.dx_m * s.scale + s.init_x_px;
  out[1] = s.dy_m * s.scale + s.init_y_px;
  out[2] = s.theta;
  out[3] = s.scale;
}

// A16  ParticleFilter::meanLikelihood   (src/particle_filter.cpp:191-203) — serial float accumulation as written.
void orc_mean_likelihood(const orc_state* st, long n, float mean[4]) {
  float acc[4] = {0, 0, 0, 0};
  float cos_sum = 0, sin_sum = 0;
  for (long p = 0; p < n; p++) {
    float s[4];
    ml_state(st[p], s);
    for (int k = 0; k < 4; k++) acc[k] += s[k];
    if (g_overload_mode & 2) {   // float += double: (float)((double)sum + cos((double)theta))
      cos_sum = (float)((double)cos_sum + cos((double)s[2]));
      sin_sum = (float)((double)sin_sum + sin((double)s[2]));
    } else {
      cos_sum += cosf(s[2]);
      sin_sum += sinf(s[2]);
    }
  }
  for (int k = 0; k < 4; k++) mean[k] = acc[k] / (float)n;
  mean[2] = (g_overload_mode & 2) ? (float)atan2((double)(sin_sum / (float)n), (double)(cos_sum / (float)n))
                                  : atan2f(sin_sum / (float)n, cos_sum / (float)n);
}

// A16  ParticleFilter::computeMeanCov / computeCov   (src/particle_filter.cpp:205-236); cov row-major 4x4.
void orc_cov_about(const orc_state* st, long n, const float ref[4], float cov[16]) {
  for (int k = 0; k < 16; k++) cov[k] = 0;
  for (long p = 0; p < n; p++) {
    float s[4];
    ml_state(st[p], s);
    for (int k = 0; k < 4; k++) s[k] -= ref[k];
    while (s[2] > M_PI) s[2] = (float)((double)s[2] - 2 * M_PI);             // :215
    while (s[2] < -M_PI) s[2] = (float)((double)s[2] + 2 * M_PI);            // :216
    for (int a = 0; a < 4; a++)
      for (int b = 0; b < 4; b++) cov[4 * a + b] += s[a] * s[b];             // :217
  }
  for (int k = 0; k < 16; k++) cov[k] /= (float)(n - 1);                     // :219
}
void orc_mean_cov(const orc_state* st, long n, float mean[4], float cov[16]) {
  if (n < 1) { for (int k = 0; k < 16; k++) cov[k] = 0; return; }            // :207-209
  orc_mean_likelihood(st, n, mean);
  orc_cov_about(st, n, mean, cov);
}

// A16  ParticleFilter::freezeScale   (src/particle_filter.cpp:343-357); returns the geometric mean.
float orc_freeze_scale(orc_state* st, long n) {
  float geo_mean = 1;
  for (long p = 0; p < n; p++) geo_mean = (float)((double)geo_mean * std::pow((double)st[p].scale, 1. / (double)n));   // :347
  for (long p = 0; p < n; p++) st[p].scale = geo_mean;
  return geo_mean;
}

// ParticleFilter::updateMap state shift   (src/particle_filter.cpp:325-334)
void orc_shift_init(orc_state* st, long n, int dx, int dy) {
  for (long p = 0; p < n; p++) { st[p].init_x_px += (float)dx; st[p].init_y_px += (float)dy; }
}

// StateParticle::StateParticle(gen, map, params, init=true)   (src/state_particle.cpp:3-49)
void orc_init_particle(const orc_map* m, const orc_filter_params* fp, void* rng, orc_state* out) {
  std::mt19937& gen = *(std::mt19937*)rng;
  std::uniform_real_distribution<float> uniform_dist(0., 1.);
  std::normal_distribution<float> normal_dist(0., 1.);
  orc_state st{};
  st.scale = 1;
  float map_w = (float)m->cols * m->resolution, map_h = (float)m->rows * m->resolution;   // :11
  if (fp->fixed_scale < 0) st.scale = (float)std::pow(10, ((double)uniform_dist(gen) - 0.5) * 2);  // :15
  else st.scale = fp->fixed_scale;
  while (true) {
    if (fp->init_pos_px_x > 0) {
      st.init_x_px = std::clamp<float>(normal_dist(gen) * fp->init_pos_px_cov + fp->init_pos_px_x, 0, map_w);  // :22
      st.init_y_px = std::clamp<float>(normal_dist(gen) * fp->init_pos_px_cov + fp->init_pos_px_y, 0, map_h);  // :23
    } else {
      st.init_x_px = uniform_dist(gen) * map_w;                              // :25-26
      st.init_y_px = uniform_dist(gen) * map_h;
    }
    if (orc_classes_at_point(m, (int)st.init_x_px, (int)st.init_y_px) & 2u) break;   // :28-31, class 1 = road
  }
  if (fp->init_pos_deg_theta != std::numeric_limits<float>::infinity()) {
    st.theta = normal_dist(gen) * fp->init_pos_deg_cov + fp->init_pos_deg_theta;      // :35
    st.theta = (float)((double)st.theta * (M_PI / 180));                     // :37
    st.have_init = 1;
  } else {
    st.theta = 0;
    st.have_init = 0;
  }
  *out = st;
}

// ParticleFilter::initializeParticles particle loop   (src/particle_filter.cpp:57-71), including the RNG draws
// burnt on the prototype and on the second buffer.  Returns the number of particles created.
long orc_initialize_particles(const orc_map* m, const orc_filter_params* fp, int max_num, void* rng, orc_state* out) {
  size_t num_at_scale = (fp->fixed_scale < 0) ? 10 : 1;                      // :20-25
  long count = 0;
  for (int i = 0; i < (int)((size_t)max_num / num_at_scale); i++) {           // :57
    orc_state proto;
    orc_init_particle(m, fp, rng, &proto);                                   // :58
    for (float scale = 0; scale < 1; scale += 1. / num_at_scale) {           // :59
      orc_state part;
      orc_init_particle(m, fp, rng, &part);                                  // :60
      if (fp->fixed_scale < 0) {
        part = proto;                                                        // :62
        part.scale = (float)std::pow(10., (double)scale);                    // :63
      }
      out[count++] = part;
      orc_state burn;
      orc_init_particle(m, fp, rng, &burn);                                  // :68 (second buffer)
    }
  }
  return count;
}

// N4  ActiveLocalizer   (src/active_localizer.cpp; dead at its call sites src/particle_filter.cpp:77-78,316, built for
// completeness).  Local maps are nb x nr images (the reference hard-codes 100 x 25, the shape its node gives the table,
// src/top_down_render.cpp:115); `tab` is the map's sample table of that shape.
// getLocalMap (:22-42): the window at state.head<2>() with res = 2 (scale 1, top_down_map_polar.cpp:78-82), rows rotated by
// rot_shift = round(state[2] * nb / 2 / pi) (:32-36): out row a = window row (a - shift) mod nb (:38-41).
static void orc_active_local_map(const orc_map* m, const float* tab, int nb, int nr, const float state[3], float* out) {
  const long P = (long)nb * nr;
  std::vector<float> orig((size_t)m->ncls * P);
  std::vector<uint8_t> mask(P);
  orc_local_map_polar(m, tab, P, state[0], state[1], 1.f, 2.f, orig.data(), mask.data());   // :30
  int rot_shift = (int)std::round((double)(state[2] * (float)nb / 2) / M_PI);                 // :33
  while (rot_shift >= nb) rot_shift -= nb;                                                    // :35-36
  while (rot_shift < 0) rot_shift += nb;
  for (int c = 0; c < m->ncls; c++)
    for (int j = 0; j < nr; j++)
      for (int a = 0; a < nb; a++) {
        const int src = a < rot_shift ? nb - rot_shift + a : a - rot_shift;                   // :39-40
        out[(size_t)c * P + a + (size_t)nb * j] = orig[(size_t)c * P + src + (size_t)nb * j];
      }
}
// computeTotalDifference (:7-20): mean over pairs i > j and classes of sum |L_i - L_j| (Eigen float sums, order
// unspecified: accumulated in double per block here, the running total in float like the reference's).
static float orc_active_total_difference(const float* maps, int K, int ncls, long P) {
  float total = 0;
  int cnt = 0;
  for (int i = 0; i < K; i++)
    for (int j = 0; j < i; j++)
      for (int c = 0; c < ncls; c++) {
        const float* a = maps + ((size_t)i * ncls + c) * P;
        const float* b = maps + ((size_t)j * ncls + c) * P;
        double s = 0;
        for (long k = 0; k < P; k++) s += (double)fabsf(a[k] - b[k]);
        total += (float)s;                                                                    // :14
        cnt += 1;
      }
  return total / (float)cnt;                                                                  // :19 (0 / 0 = NaN for one mode)
}
// getBestRelPos (:44-82).  preds: [K][3] = {x, y, theta}.  out = {dist, theta} of the best candidate, *best_diff its mean
// difference; diffs_out (optional, [4][17]): every candidate's difference, NaN where the loops did not go.
void orc_active_best_rel_pos(const orc_map* m, const float* tab, int nb, int nr, const float* preds, int K, float out[2],
                             float* best_diff_out, float* diffs_out) {
  const long P = (long)nb * nr;
  std::vector<float> maps((size_t)K * m->ncls * P);
  if (diffs_out)
    for (int q = 0; q < 4 * 17; q++) diffs_out[q] = std::numeric_limits<float>::quiet_NaN();
  float dist = 50, best_diff = 0;                                                             // :55-56
  float best[2] = {0, 0};
  int di = 0;
  while (best_diff < 6000 && dist < 150) {                                                    // :58
    int ti = 0;
    for (float theta = 0; theta < 2 * M_PI; theta += M_PI / 8) {                              // :59
      for (int i = 0; i < K; i++) {
        float pos[3] = {preds[3 * i], preds[3 * i + 1], preds[3 * i + 2]};
        const float ang = theta + pos[2];
        pos[0] += dist * cosf(ang);                                                           // :63 (float overloads)
        pos[1] += dist * sinf(ang);
        orc_active_local_map(m, tab, nb, nr, pos, maps.data() + (size_t)i * m->ncls * P);
      }
      const float diff = orc_active_total_difference(maps.data(), K, m->ncls, P);             // :69
      if (diffs_out && di < 4 && ti < 17) diffs_out[di * 17 + ti] = diff;
      if (diff > best_diff) {                                                                 // :70-73
        best_diff = diff;
        best[0] = dist;
        best[1] = theta;
      }
      ti++;
    }
    dist += 25;                                                                               // :76
    di++;
  }
  out[0] = best[0];
  out[1] = best[1];
  if (best_diff_out) *best_diff_out = best_diff;
}

// N2  TopDownRender::publishPoseEst — the per-step control logic   (src/top_down_render.cpp:331-365)
// The node's own members travel in `orc_node_state`; the filter's answers (computeMeanCov, scale(), numParticles(),
// meanLikelihood()[3], isScaleFrozen()) are arguments.  Returns bit 0: freezeScale() is to be called now (:356-359).
// Arithmetic as written there: `current_range_scale_ += 0.05` is float += double; `std::pow(float, int)` and
// `0.003 * ml_state[3]` are double, the comparisons against them promote the float side.
struct orc_node_state {
  float current_range_scale, range_scale_min, range_scale_max, target_uncertainty_m;   // top_down_render.h:78-82
  int32_t is_converged;                                                                 // :83
};
int orc_publish_pose_est(orc_node_state* ns, const float cov[16], float filter_scale, int num_particles, float ml_scale,
                         int scale_frozen) {
  const float scale = filter_scale;                                                     // :335
  const float scale_2 = scale * scale;                                                  // :336
  const float big = std::max(cov[0], cov[5]) / scale_2;                                 // :337 max(cov(0,0), cov(1,1))/scale_2
  if ((double)big > std::pow((double)ns->target_uncertainty_m, 2) && ns->current_range_scale < ns->range_scale_max) {
    ns->current_range_scale = (float)((double)ns->current_range_scale + 0.05);          // :341
  } else if (ns->current_range_scale > ns->range_scale_min) {
    ns->current_range_scale = (float)((double)ns->current_range_scale - 0.02);          // :344
  }
  if (num_particles < 1) return 0;                                                      // :347-350
  int freeze = 0;
  if ((double)cov[15] < 0.003 * (double)ml_scale && !scale_frozen) freeze = 1;          // :356-359 cov(3,3)
  // :362 reads filter_->scale() AFTER the freeze: frozen -> particles_[0]->state().scale, which the caller passes back in
  // through orc_publish_pose_est_gate once it has frozen; kept in one function for the common case of a fixed scale
  return freeze;
}
// :362-364, evaluated with the scale the filter reports after a possible freezeScale()
void orc_publish_pose_est_gate(orc_node_state* ns, const float cov[16], float scale_2_before, float filter_scale_now) {
  if (cov[0] / scale_2_before < 40 && cov[5] / scale_2_before < 40 && cov[10] < 0.5 && filter_scale_now > 0)
    ns->is_converged = 1;
}
// ParticleFilter::scale()   (src/particle_filter.cpp:359-367)
float orc_filter_scale(const orc_filter_params* fp, int scale_frozen, const orc_state* st, long n) {
  if (fp->fixed_scale > 0) return fp->fixed_scale;
  if (scale_frozen && n > 0) return st[0].scale;
  return -1;
}

int orc_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
