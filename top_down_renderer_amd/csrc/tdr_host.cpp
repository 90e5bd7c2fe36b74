// tdr_host.cpp — the handle layer of the C ABI (include/tdr.h, "tdr_map_* / tdr_renderer_* / tdr_filter_*"):
// C++ host code that owns device memory and sequences the hand-written HIP kernels (tdr_*.hip) exactly the way
// the reference's classes sequence their Eigen loops.  One handle = one reference object:
//     tdr_map       TopDownMapPolar   (include/top_down_render/top_down_map_polar.h:6-22)
//     tdr_renderer  ScanRendererPolar (include/top_down_render/scan_renderer_polar.h:15-22)
//     tdr_filter    ParticleFilter    (include/top_down_render/particle_filter.h:22-73)
// One caller thread per handle (the reference calls everything from the ROS spinner thread).  A filter lives on one GPU
// or is sharded over the ranks of a tdr_comm (one process per GPU, tdr_comm.cpp: RCCL or caller-supplied transport).
// No CPU fallback: every entry point fails with TDR_ERR_HIP when no device is present.
#include <sys/stat.h>
#include <cerrno>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "tdr.h"

extern "C" int tdr_set_error(int code, const char* msg);  // tdr_core.hip

namespace {

int failh(int code, const char* fmt, ...) {
  char buf[400];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return tdr_set_error(code, buf);
}
#define HTRY(expr)                                                                        \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) return failh(TDR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define TTRY(expr)            \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != TDR_OK) return rc_; \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;             // owns device memory
  DevBuf& operator=(const DevBuf&) = delete;
  int resize(size_t count) {
    if (count <= n) return TDR_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e != hipSuccess) return failh(TDR_ERR_NOMEM, "hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e));
    n = count;
    return TDR_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { release(); }
};

}  // namespace

struct tdr_map {
  DevBuf<float> rec;
  DevBuf<float> tab;
  DevBuf<float> fac;   // the table's factors (tdr_polar_factors_host), handed to the filter's tdr_score_ctx
  DevBuf<uint32_t> crec;   // compact form of `rec` (tdr_k_compact_map), when the map has one
  DevBuf<float> cdict;
  DevBuf<uint8_t> cws;
  DevBuf<uint8_t> rec16;   // scratch of the 40-rotation search (tdr_map_desc.rec16), allocated by the first large search
  std::vector<float> maps_host;  // class_maps_ (column-major), kept for getClassesAtPoint / particle initialisation
  std::vector<uint8_t> mask_host;  // class_mask_ (column-major), kept for the map cache
  tdr_map_desc desc{};
  DevBuf<float> geo_rec;         // geo_maps_[0..1] as a 2-class record map (tdr_k_geo_map_from_map), built on first use
  int geo_pending = 0;           // 0: geo_rec is current (or there is no map); 1: derive from the classes; 2: constant 1
  tdr_map_desc geo_desc{};
  int nb = 0, nr = 0;
  float ang_res = 0;
  int center_x = 0, center_y = 0;
  bool have_map = false;
  // staging of the run-time map replacement (tdr_map_set_labels: aerial maps keep arriving, top_down_render.cpp:574-600),
  // kept between calls: allocating and freeing ~1 GB per map costs more than the ingest itself
  DevBuf<uint8_t> ing_img, ing_ws, ing_mask;
  DevBuf<int32_t> ing_lut;
  DevBuf<float> ing_maps;
};

struct tdr_renderer {
  DevBuf<int32_t> lut;
  DevBuf<float> pts, img, pk, geo;
  DevBuf<uint8_t> keys;  // per-point bin keys of the two-phase raster
  DevBuf<uint8_t> geo_ws;  // sort keys / scratch of the geometric render
  int ncls = 0, rows = 0, cols = 0;  // shape of the last render
  bool have_scan = false;
};

struct tdr_filter {
  tdr_map* map = nullptr;
  tdr_filter_params fp{};
  int64_t n_max = 0, n = 0;
  DevBuf<float> st, st_new, last_dist, raw_w, w, runmax, info, ws, z4, scan_img, scan_pk, stats;
  DevBuf<uint8_t> pfx_ws;  // chunk headers of the multi-workgroup running sum
  DevBuf<int32_t> idx, perm, loc_tmp;
  DevBuf<tdr_state> aos;
  void* rng = nullptr;
  uint64_t seed = 0, step = 0;
  uint64_t prop_calls = 0;    // device RNG: every propagate call draws fresh noise (counter = calls so far)
  bool scale_frozen = false, maybe_uninit = true, parity_rng = true;
  int locality_every = 1;
  float uniform_scale = 0.f;
  bool rng_owned = true;      // false after tdr_filter_share_rng: the generator belongs to the caller
  DevBuf<float> gmm_samples;  // [num][3] device staging for computeGMM
  int num_gaussians = 1;      // particle_filter.cpp:7
  std::vector<float> gmm_means, gmm_covs;
  DevBuf<float> ml_dev;  // fields + mlState of the max-likelihood particle of the last update (tdr_k_save_ml_state)
  bool have_ml = false;
  // meanLikelihood + computeMeanCov of the CURRENT particle set, as last read back: publishPoseEst asks for both in a row
  // (src/top_down_render.cpp:333, 354), which is one kernel and one read-back here.  Everything that changes the set
  // clears the flag (states_changed).
  float mean_cov_host[24] = {0};
  bool mean_cov_valid = false;
  void states_changed() { mean_cov_valid = false; }
  hipStream_t stream = nullptr;
  tdr_score_ctx* score_ctx = nullptr;   // this filter's own span tuner (and the table's factors) for its scoring launches (tdr.h)
  // The reference's generator in parity mode: the host std::mt19937 `rng` and its continuation on the device, a
  // tdr_rng_pipe (csrc/tdr_rng.hip).  Exactly one of them is current: propagate and the resample's uniform draw continue
  // the stream on the device (drawn ahead, beside the scoring launch), the host engine takes it back when host code
  // draws (particle initialisation).  A generator shared with the caller (tdr_filter_share_rng) stays on the host.
  tdr_rng_pipe* pipe = nullptr;
  // Sharded over the ranks of `comm` (one process per GPU; NULL = the whole filter lives here).  n / n_max stay the
  // GLOBAL counts; this rank holds particles [rank * nl, (rank + 1) * nl), nl = n / world, in st[7][cap] with
  // cap = n_max / world.  raw_glob / ld_glob / w / runmax are global arrays, identical on every rank.
  tdr_comm* comm = nullptr;
  int world = 1, rank = 0;
  int64_t cap = 0;
  DevBuf<float> xchg_in, xchg_out, raw_glob, ld_glob, st_send, st_all, st_glob, pk_recv;
  DevBuf<float> geo_pk;   // packed geometric scan (tdr_filter_update_geo)
  int64_t nl() const { return n / world; }
};

// .eig files of the reference's map cache (top_down_map.h:29-50)
static std::string cache_dir_or_default(const char* cache_dir) {
  if (cache_dir && *cache_dir) return cache_dir;
  const char* home = getenv("HOME");
  return std::string(home ? home : ".") + "/.ros/xview_cache";
}
int tdr_png_read_gray8(const char* path, std::vector<uint8_t>& px, int& w, int& h);   // tdr_png.cpp
int tdr_png_write_gray8(const char* path, const uint8_t* px, int w, int h);

template <class T>
static int read_eig(const std::string& path, std::vector<T>& out, int64_t& rows, int64_t& cols) {
  FILE* fh = fopen(path.c_str(), "rb");
  if (!fh) return failh(TDR_ERR_ARG, "map cache: cannot open %s", path.c_str());
  int64_t hdr[2] = {0, 0};
  bool ok = fread(hdr, sizeof(int64_t), 2, fh) == 2 && hdr[0] > 0 && hdr[1] > 0 && hdr[0] < (1 << 24) && hdr[1] < (1 << 24);
  if (ok) {   // the payload the header promises must be what the file holds (a damaged header must not size the buffer)
    const long at = ftell(fh);
    ok = at >= 0 && fseek(fh, 0, SEEK_END) == 0;
    const long end = ok ? ftell(fh) : -1;
    ok = ok && end >= at && (uint64_t)(end - at) == (uint64_t)hdr[0] * (uint64_t)hdr[1] * sizeof(T) &&
         fseek(fh, at, SEEK_SET) == 0;
  }
  if (ok) {
    out.resize((size_t)hdr[0] * hdr[1]);
    ok = fread(out.data(), sizeof(T), out.size(), fh) == out.size() && fgetc(fh) == EOF;
  }
  fclose(fh);
  if (!ok) return failh(TDR_ERR_ARG, "map cache: %s is not a well-formed .eig file of this scalar type", path.c_str());
  rows = hdr[0];
  cols = hdr[1];
  return TDR_OK;
}
template <class T>
static int write_eig(const std::string& path, const T* data, int64_t rows, int64_t cols) {
  FILE* fh = fopen(path.c_str(), "wb");
  if (!fh) return failh(TDR_ERR_ARG, "map cache: cannot write %s", path.c_str());
  const int64_t hdr[2] = {rows, cols};
  const bool ok = fwrite(hdr, sizeof(int64_t), 2, fh) == 2 && fwrite(data, sizeof(T), (size_t)rows * cols, fh) == (size_t)rows * cols;
  fclose(fh);
  return ok ? TDR_OK : failh(TDR_ERR_ARG, "map cache: short write to %s", path.c_str());
}
// geo_maps_ for a freshly packed map: computed from the class maps like the static-map constructor does
// (src/top_down_map.cpp:48-58), or the constant 1 the dynamic-map path leaves them at (:126-133)
// Nothing on the hot path reads them (the reference's score ignores top_down_geo, state_particle.cpp:145-152): they are
// built — two more distance transforms over the whole map — when something first asks for them (map_ensure_geo).
static int map_make_geo(tdr_map* m, bool constant_one) {
  m->geo_pending = constant_one ? 2 : 1;
  m->geo_rec.release();
  m->geo_desc = tdr_map_desc{};
  return TDR_OK;
}
static int map_ensure_geo(tdr_map* m) {
  if (!m->geo_pending) return TDR_OK;
  const bool constant_one = m->geo_pending == 2;
  const int rows = m->desc.rows, cols = m->desc.cols;
  TTRY(m->geo_rec.resize(tdr_map_rec_floats_total(2, rows, cols)));
  DevBuf<uint8_t> ws;
  if (!constant_one) TTRY(ws.resize(tdr_map_ingest_workspace_bytes(2, rows, cols)));
  TTRY(tdr_k_geo_map_from_map(&m->desc, constant_one ? 1 : 0, m->geo_rec.p, ws.p, nullptr));
  HTRY(hipDeviceSynchronize());
  m->geo_desc = tdr_map_desc{};
  m->geo_desc.rec = m->geo_rec.p;
  m->geo_desc.ncls = 2;
  m->geo_desc.rows = rows;
  m->geo_desc.cols = cols;
  m->geo_desc.rec_floats = tdr_rec_floats(2);
  m->geo_desc.resolution = m->desc.resolution;
  m->geo_pending = 0;
  return TDR_OK;
}

// the compact records of a freshly packed map (desc.rec etc. already set)
static int map_compact(tdr_map* m) {
  m->desc.crec = nullptr; m->desc.dict = nullptr; m->desc.dict_n = 0; m->desc.cwords = 0;
  m->desc.rec16 = nullptr;   // sized for the previous grid: the next large init search allocates it again
  const size_t nw = tdr_cmap_words_total(m->desc.ncls, m->desc.rows, m->desc.cols);
  if (nw == 0) return TDR_OK;
  TTRY(m->crec.resize(nw));
  TTRY(m->cdict.resize(TDR_CMAP_WIDE_MAX_DICT));
  TTRY(m->cws.resize(TDR_CMAP_WORKSPACE_BYTES));
  TTRY(tdr_k_compact_map(&m->desc, m->crec.p, m->cdict.p, m->cws.p, nullptr));
  if (m->desc.cwords == 0 && m->desc.dict_n < 0) {   // too many distinct values for 10-bit fields: the wide form
    const size_t nww = tdr_cmap_wide_words_total(m->desc.ncls, m->desc.rows, m->desc.cols);
    TTRY(m->crec.resize(nww));
    TTRY(tdr_k_compact_map_wide(&m->desc, m->crec.p, m->cdict.p, m->cws.p, nullptr));
  }
  if (m->desc.cwords == 0) m->desc.dict_n = 0;
  return TDR_OK;
}

extern "C" {

// ---- TopDownMap(Polar) ------------------------------------------------------------------------------------------------
int tdr_map_create(tdr_map** out) {
  if (!out) return failh(TDR_ERR_ARG, "map_create: null out");
  if (tdr_device_count() < 1) return failh(TDR_ERR_HIP, "map_create: no HIP device (there is no CPU fallback)");
  *out = new tdr_map();
  return TDR_OK;
}
void tdr_map_destroy(tdr_map* m) { delete m; }

// Storage of class_maps_ / class_mask_ (top_down_map.h:77-79) in the form computeDists leaves them
// (top_down_map.cpp:289-326); also the body of TopDownMap::updateMap once the distance transform is done (:146-157).
int tdr_map_set(tdr_map* m, const float* class_maps, const uint8_t* class_mask, int ncls, int rows, int cols,
                float resolution, int center_x, int center_y) {
  if (!m || !class_maps || !class_mask) return failh(TDR_ERR_ARG, "map_set: null pointer");
  if (ncls < 1 || ncls > TDR_MAX_CLASSES || rows < 1 || cols < 1 || !(resolution > 0))
    return failh(TDR_ERR_ARG, "map_set: bad shape / resolution");
  const size_t ncell = (size_t)rows * cols;
  DevBuf<float> d_maps;
  DevBuf<uint8_t> d_mask;
  TTRY(d_maps.resize(ncell * ncls));
  TTRY(d_mask.resize(ncell));
  HTRY(hipMemcpy(d_maps.p, class_maps, ncell * ncls * sizeof(float), hipMemcpyHostToDevice));
  HTRY(hipMemcpy(d_mask.p, class_mask, ncell, hipMemcpyHostToDevice));
  TTRY(m->rec.resize(tdr_map_rec_floats_total(ncls, rows, cols)));
  TTRY(tdr_k_pack_map(d_maps.p, d_mask.p, ncls, rows, cols, m->rec.p, nullptr));
  HTRY(hipDeviceSynchronize());
  m->maps_host.assign(class_maps, class_maps + ncell * ncls);
  m->mask_host.assign(class_mask, class_mask + ncell);
  m->desc.rec = m->rec.p;
  m->desc.ncls = ncls;
  m->desc.rows = rows;
  m->desc.cols = cols;
  m->desc.rec_floats = tdr_rec_floats(ncls);
  m->desc.resolution = resolution;
  m->center_x = center_x;
  m->center_y = center_y;
  TTRY(map_compact(m));
  TTRY(map_make_geo(m, false));
  m->have_map = true;
  if (m->nb > 0) return tdr_map_sample_pts_polar(m, m->nb, m->nr, m->ang_res);
  return TDR_OK;
}

// TopDownMap::updateMap(const cv::Mat&, map_center) (top_down_map.cpp:146-157): loadCompressedRasterMap (:116-144) +
// computeDists (:289-326) for a HOST class-index image (cv::Mat CV_8UC1 layout), all on the device.
int tdr_map_set_labels(tdr_map* m, const uint8_t* label_img, int img_h, int img_w, const int32_t* flatten_lut,
                       int lut_size, int ncls, float resolution, int center_x, int center_y) {
  if (!m || !label_img || !flatten_lut) return failh(TDR_ERR_ARG, "map_set_labels: null pointer");
  int rows = 0, cols = 0;
  TTRY(tdr_map_ingest_shape(img_h, img_w, resolution, &rows, &cols));
  if (ncls < 1 || ncls > TDR_MAX_CLASSES || rows < 1 || cols < 1) return failh(TDR_ERR_ARG, "map_set_labels: bad shape");
  DevBuf<uint8_t>&d_img = m->ing_img, &d_ws = m->ing_ws, &d_mask = m->ing_mask;
  DevBuf<int32_t>& d_lut = m->ing_lut;
  DevBuf<float>& d_maps = m->ing_maps;
  const size_t ncell = (size_t)rows * cols;
  TTRY(d_img.resize((size_t)img_h * img_w));
  TTRY(d_lut.resize((size_t)lut_size));
  TTRY(d_ws.resize(tdr_map_ingest_workspace_bytes(ncls, rows, cols)));
  HTRY(hipMemcpy(d_img.p, label_img, (size_t)img_h * img_w, hipMemcpyHostToDevice));
  HTRY(hipMemcpy(d_lut.p, flatten_lut, (size_t)lut_size * sizeof(int32_t), hipMemcpyHostToDevice));
  TTRY(m->rec.resize(tdr_map_rec_floats_total(ncls, rows, cols)));
  TTRY(tdr_k_map_from_labels(d_img.p, img_h, img_w, d_lut.p, lut_size, ncls, resolution, m->rec.p, d_ws.p, nullptr));
  // host copy of class_maps_ for getClassesAtPoint / particle initialisation
  TTRY(d_maps.resize(ncell * ncls));
  TTRY(d_mask.resize(ncell));
  TTRY(tdr_k_unpack_map(m->rec.p, ncls, rows, cols, d_maps.p, d_mask.p, nullptr));
  m->maps_host.resize(ncell * ncls);
  m->mask_host.resize(ncell);
  HTRY(hipMemcpy(m->maps_host.data(), d_maps.p, ncell * ncls * sizeof(float), hipMemcpyDeviceToHost));
  HTRY(hipMemcpy(m->mask_host.data(), d_mask.p, ncell, hipMemcpyDeviceToHost));
  m->desc.rec = m->rec.p;
  m->desc.ncls = ncls;
  m->desc.rows = rows;
  m->desc.cols = cols;
  m->desc.rec_floats = tdr_rec_floats(ncls);
  m->desc.resolution = resolution;
  m->center_x = center_x;
  m->center_y = center_y;
  TTRY(map_compact(m));
  TTRY(map_make_geo(m, true));   // updateMap leaves geo_maps_ at their constant 1 (:126-133)
  // `if (!class_maps_[1].isZero(0)) have_map_ = true; else "Received map with no road"` (:150-154)
  bool road = false;
  if (ncls > 1)
    for (size_t k = 0; k < ncell && !road; k++) road = m->maps_host[ncell + k] != 0.f;
  if (road) m->have_map = true;
  if (m->nb > 0 && m->have_map) return tdr_map_sample_pts_polar(m, m->nb, m->nr, m->ang_res);
  return TDR_OK;
}

// TopDownMapPolar::samplePtsPolar (top_down_map_polar.cpp:7-19)
int tdr_map_sample_pts_polar(tdr_map* m, int nb, int nr, float ang_res) {
  if (!m || nb < 1 || nr < 1) return failh(TDR_ERR_ARG, "sample_pts_polar: bad arguments");
  m->nb = nb;
  m->nr = nr;
  m->ang_res = ang_res;
  if (!m->have_map) return TDR_OK;  // table needs params_.resolution; built when the map arrives
  std::vector<float> tab((size_t)2 * nb * nr);
  TTRY(tdr_polar_table_host(nb, nr, ang_res, m->desc.resolution, tab.data()));
  TTRY(m->tab.resize(tab.size()));
  HTRY(hipMemcpy(m->tab.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
  std::vector<float> fac((size_t)2 * nb + nr);
  TTRY(tdr_polar_factors_host(nb, nr, ang_res, m->desc.resolution, fac.data()));
  TTRY(m->fac.resize(fac.size()));
  HTRY(hipMemcpy(m->fac.p, fac.data(), fac.size() * sizeof(float), hipMemcpyHostToDevice));
  return TDR_OK;
}

// the (theta bins, range bins) given to samplePtsPolar last: the shape ParticleFilter::update's images must have
int tdr_map_polar_shape(const tdr_map* m, int* nb, int* nr) {
  if (!m || !nb || !nr) return failh(TDR_ERR_ARG, "map_polar_shape: bad arguments");
  *nb = m->nb;
  *nr = m->nr;
  return TDR_OK;
}

int tdr_map_info(const tdr_map* m, int* ncls, int* rows, int* cols, float* resolution, int* have_map) {
  if (!m) return failh(TDR_ERR_ARG, "map_info: null map");
  if (ncls) *ncls = m->desc.ncls;
  if (rows) *rows = m->desc.rows;
  if (cols) *cols = m->desc.cols;
  if (resolution) *resolution = m->desc.resolution;
  if (have_map) *have_map = m->have_map ? 1 : 0;
  return TDR_OK;
}

// TopDownMap::mapCenter() (top_down_map.h:72): the centre given with the last map, whoever set it (the map's own
// updateMap or ParticleFilter::updateMap)
int tdr_map_center(const tdr_map* m, int* center_x, int* center_y) {
  if (!m || !center_x || !center_y) return failh(TDR_ERR_ARG, "map_center: bad arguments");
  *center_x = m->center_x;
  *center_y = m->center_y;
  return TDR_OK;
}

// getLocalMap through the handle: one window, host arrays out
int tdr_map_local_map(tdr_map* m, int polar, float cx, float cy, float scale_or_rot, float res, int rows, int cols,
                      float* dists_out, uint8_t* mask_out) {
  if (!m || !m->have_map || !dists_out || !mask_out) return failh(TDR_ERR_ARG, "map_local_map: no map / null output");
  if (polar) {
    if (m->nb < 1 || !m->tab.p) return failh(TDR_ERR_ARG, "map_local_map: samplePtsPolar was never called");
    rows = m->nb;
    cols = m->nr;
  }
  if (rows < 1 || cols < 1) return failh(TDR_ERR_ARG, "map_local_map: bad window shape");
  const size_t P = (size_t)rows * cols;
  DevBuf<float> d;
  DevBuf<uint8_t> k;
  TTRY(d.resize(P * m->desc.ncls));
  TTRY(k.resize(P));
  if (polar) TTRY(tdr_k_local_map_polar(&m->desc, m->tab.p, rows, cols, cx, cy, scale_or_rot, res, d.p, k.p, nullptr));
  else TTRY(tdr_k_local_map_cart(&m->desc, rows, cols, cx, cy, scale_or_rot, res, d.p, k.p, nullptr));
  HTRY(hipMemcpy(dists_out, d.p, P * m->desc.ncls * sizeof(float), hipMemcpyDeviceToHost));
  HTRY(hipMemcpy(mask_out, k.p, P, hipMemcpyDeviceToHost));
  return TDR_OK;
}

// ActiveLocalizer::getBestRelPos (src/active_localizer.cpp:44-82): every candidate's difference in one launch, then the
// reference's sequential choice — strict `>` over the candidates in loop order, the next distance only while the best
// difference is below 6000 (:58, 70-73).
int tdr_map_best_rel_pos(tdr_map* m, const float* preds, int K, float best_rel_pos[2], float* best_diff) {
  if (!m || !m->have_map || !preds || !best_rel_pos) return failh(TDR_ERR_ARG, "best_rel_pos: no map / null pointer");
  if (m->nb < 1 || !m->tab.p) return failh(TDR_ERR_ARG, "best_rel_pos: samplePtsPolar was never called");
  if (K < 1 || K > TDR_GMM_MAX_K) return failh(TDR_ERR_ARG, "best_rel_pos: %d hypotheses (1 .. %d)", K, TDR_GMM_MAX_K);
  const int ncand_max = 4 * 17;
  std::vector<float> centres((size_t)ncand_max * K * 2), dists(ncand_max), thetas(ncand_max);
  std::vector<int32_t> shifts(K);
  int nt = 0, nd = 0;
  TTRY(tdr_active_candidates_host(preds, K, m->nb, centres.data(), dists.data(), thetas.data(), shifts.data(), &nt, &nd));
  DevBuf<float> d_c;
  DevBuf<int32_t> d_s;
  DevBuf<double> d_out;
  TTRY(d_c.resize(centres.size()));
  TTRY(d_s.resize(K));
  TTRY(d_out.resize(ncand_max));
  HTRY(hipMemcpy(d_c.p, centres.data(), centres.size() * sizeof(float), hipMemcpyHostToDevice));
  HTRY(hipMemcpy(d_s.p, shifts.data(), K * sizeof(int32_t), hipMemcpyHostToDevice));
  HTRY(hipMemset(d_out.p, 0, ncand_max * sizeof(double)));
  TTRY(tdr_k_active_diffs(&m->desc, m->tab.p, m->nb, m->nr, 2.f, d_c.p, d_s.p, K, ncand_max, d_out.p, nullptr));
  std::vector<double> sums(ncand_max);
  HTRY(hipMemcpy(sums.data(), d_out.p, ncand_max * sizeof(double), hipMemcpyDeviceToHost));
  const int cnt = K * (K - 1) / 2 * m->desc.ncls;   // :15
  float best = 0.f, bd = 0.f, bt = 0.f;
  for (int di = 0; di < nd && best < 6000.f; di++)   // :58
    for (int t = 0; t < nt; t++) {
      const float diff = (float)sums[di * 17 + t] / (float)cnt;   // :19 (0 / 0 = NaN for one hypothesis: never wins)
      if (diff > best) { best = diff; bd = dists[di * 17 + t]; bt = thetas[di * 17 + t]; }   // :70-73
    }
  best_rel_pos[0] = bd;
  best_rel_pos[1] = bt;
  if (best_diff) *best_diff = best;
  return TDR_OK;
}

// getLocalGeoMap (top_down_map_polar.cpp:55-76, top_down_map.cpp:461-481): the window of one pose gathered from the two
// geometric layers; dists_out HOST [2][rows*cols]
int tdr_map_local_geo_map(tdr_map* m, int polar, float cx, float cy, float scale_or_rot, float res, int rows, int cols,
                          float* dists_out) {
  if (!m || !m->have_map || !dists_out) return failh(TDR_ERR_ARG, "map_local_geo_map: no map / null output");
  TTRY(map_ensure_geo(m));
  if (polar) {
    if (m->nb < 1 || !m->tab.p) return failh(TDR_ERR_ARG, "map_local_geo_map: samplePtsPolar was never called");
    rows = m->nb;
    cols = m->nr;
  }
  if (rows < 1 || cols < 1) return failh(TDR_ERR_ARG, "map_local_geo_map: bad window shape");
  const size_t P = (size_t)rows * cols;
  DevBuf<float> d;
  DevBuf<uint8_t> k;
  TTRY(d.resize(P * 2));
  TTRY(k.resize(P));
  if (polar) TTRY(tdr_k_local_map_polar(&m->geo_desc, m->tab.p, rows, cols, cx, cy, scale_or_rot, res, d.p, k.p, nullptr));
  else TTRY(tdr_k_local_map_cart(&m->geo_desc, rows, cols, cx, cy, scale_or_rot, res, d.p, k.p, nullptr));
  HTRY(hipMemcpy(dists_out, d.p, P * 2 * sizeof(float), hipMemcpyDeviceToHost));
  return TDR_OK;
}

// ---- the reference's on-disk map cache (src/top_down_map.cpp:226-286) ------------------------------------------------
// ~/.ros/xview_cache/{cached_data.txt, class_map<i>.eig, geo_map<i>.eig, class_mask.eig}; an .eig file is
// `Index rows, Index cols` (2 x int64) followed by the column-major scalars (top_down_map.h:29-50).
// loadCacheMetaData + loadCachedMaps (:226-261).  *loaded = 0 (and TDR_OK) when no cache matches (map_path, num_classes,
// resolution) — the caller then builds the map by other means; a matching but damaged cache is an error.
int tdr_map_load_cache(tdr_map* m, const char* cache_dir, const char* map_path, int num_classes, float resolution,
                       int center_x, int center_y, int* loaded) {
  if (!m || !map_path || !loaded) return failh(TDR_ERR_ARG, "map_load_cache: bad arguments");
  *loaded = 0;
  const std::string dir = cache_dir_or_default(cache_dir);
  FILE* fh = fopen((dir + "/cached_data.txt").c_str(), "r");
  if (!fh) return TDR_OK;
  char line[4096];
  bool match = fgets(line, sizeof(line), fh) != nullptr;
  if (match) {
    line[strcspn(line, "\r\n")] = 0;
    match = std::string(line) == map_path;                                        // :234-235
  }
  if (match) match = fgets(line, sizeof(line), fh) && atoi(line) == num_classes;  // :236-237
  if (match) match = fgets(line, sizeof(line), fh) && std::fabs((float)atof(line) - resolution) <= 0.01f;   // :238-239
  fclose(fh);
  if (!match) return TDR_OK;
  if (num_classes < 1 || num_classes > TDR_MAX_CLASSES) return failh(TDR_ERR_ARG, "map_load_cache: bad class count");
  std::vector<float> maps, one;
  std::vector<uint8_t> mask;
  int64_t rows = 0, cols = 0, r2 = 0, c2 = 0;
  for (int c = 0; c < num_classes; c++) {
    TTRY(read_eig(dir + "/class_map" + std::to_string(c) + ".eig", one, r2, c2));
    if (c == 0) { rows = r2; cols = c2; }
    if (r2 != rows || c2 != cols) return failh(TDR_ERR_ARG, "map_load_cache: class maps differ in shape");
    maps.insert(maps.end(), one.begin(), one.end());
  }
  TTRY(read_eig(dir + "/class_mask.eig", mask, r2, c2));
  if (r2 != rows || c2 != cols) return failh(TDR_ERR_ARG, "map_load_cache: mask shape differs from the class maps");
  TTRY(tdr_map_set(m, maps.data(), mask.data(), num_classes, (int)rows, (int)cols, resolution, center_x, center_y));
  // the cached geometric layers replace the ones tdr_map_set derived (they are the same for a cache this library wrote)
  std::vector<float> g0, g1;
  if (read_eig(dir + "/geo_map0.eig", g0, r2, c2) == TDR_OK && r2 == rows && c2 == cols &&
      read_eig(dir + "/geo_map1.eig", g1, r2, c2) == TDR_OK && r2 == rows && c2 == cols) {
    g0.insert(g0.end(), g1.begin(), g1.end());
    m->geo_pending = 2;        // (cheapest fill: sizes geo_rec and sets geo_desc; the records are overwritten below)
    TTRY(map_ensure_geo(m));
    std::vector<uint8_t> zero((size_t)rows * cols, 0);
    DevBuf<float> d_maps;
    DevBuf<uint8_t> d_mask;
    TTRY(d_maps.resize(g0.size()));
    TTRY(d_mask.resize(zero.size()));
    HTRY(hipMemcpy(d_maps.p, g0.data(), g0.size() * sizeof(float), hipMemcpyHostToDevice));
    HTRY(hipMemcpy(d_mask.p, zero.data(), zero.size(), hipMemcpyHostToDevice));
    TTRY(tdr_k_pack_map(d_maps.p, d_mask.p, 2, (int)rows, (int)cols, m->geo_rec.p, nullptr));
    HTRY(hipDeviceSynchronize());
  }
  *loaded = 1;
  return TDR_OK;
}
// saveCachedMaps (:263-286)
int tdr_map_save_cache(tdr_map* m, const char* cache_dir, const char* map_path) {
  if (!m || !m->have_map || !map_path) return failh(TDR_ERR_ARG, "map_save_cache: no map");
  const std::string dir = cache_dir_or_default(cache_dir);
  const int ncls = m->desc.ncls, rows = m->desc.rows, cols = m->desc.cols;
  const size_t ncell = (size_t)rows * cols;
  // the reference creates the directory (boost::filesystem::create_directory, src/top_down_map.cpp:228-232): one level
  if (mkdir(dir.c_str(), 0777) != 0 && errno != EEXIST)
    return failh(TDR_ERR_ARG, "map_save_cache: cannot create %s (its parent must exist)", dir.c_str());
  FILE* fh = fopen((dir + "/cached_data.txt").c_str(), "w");
  if (!fh) return failh(TDR_ERR_ARG, "map_save_cache: cannot write into %s", dir.c_str());
  fprintf(fh, "%s\n%d\n%g\n", map_path, ncls, (double)m->desc.resolution);
  fclose(fh);
  for (int c = 0; c < ncls; c++)
    TTRY(write_eig(dir + "/class_map" + std::to_string(c) + ".eig", m->maps_host.data() + ncell * c, rows, cols));
  TTRY(write_eig(dir + "/class_mask.eig", m->mask_host.data(), rows, cols));
  DevBuf<float> d_maps;
  DevBuf<uint8_t> d_mask;
  TTRY(d_maps.resize(ncell * 2));
  TTRY(d_mask.resize(ncell));
  TTRY(map_ensure_geo(m));
  TTRY(tdr_k_unpack_map(m->geo_rec.p, 2, rows, cols, d_maps.p, d_mask.p, nullptr));
  std::vector<float> g(ncell * 2);
  HTRY(hipMemcpy(g.data(), d_maps.p, g.size() * sizeof(float), hipMemcpyDeviceToHost));
  TTRY(write_eig(dir + "/geo_map0.eig", g.data(), rows, cols));
  TTRY(write_eig(dir + "/geo_map1.eig", g.data() + ncell, rows, cols));
  return TDR_OK;
}

// TopDownMap::saveRasterizedMaps (top_down_map.cpp:197-211): class<i>.png, 8-bit grey, 0 inside the class and 255
// elsewhere, flipped to look like the input map (:208).  The reference writes its binary rasters before computeDists turns
// them into distances; from the distance maps held here the raster of a class is "known cell at distance 0".
int tdr_map_save_rasters(tdr_map* m, const char* dir) {
  if (!m || !m->have_map || !dir) return failh(TDR_ERR_ARG, "map_save_rasters: no map");
  const int ncls = m->desc.ncls, rows = m->desc.rows, cols = m->desc.cols;
  const size_t ncell = (size_t)rows * cols;
  if (mkdir(dir, 0700) != 0 && errno != EEXIST) return failh(TDR_ERR_ARG, "map_save_rasters: cannot create %s", dir);   // :198
  std::vector<uint8_t> img(ncell);
  for (int c = 0; c < ncls; c++) {
    const float* d = m->maps_host.data() + ncell * c;   // column-major like class_maps_
    for (int r = 0; r < rows; r++)
      for (int x = 0; x < cols; x++) {
        const size_t k = (size_t)x * rows + r;
        img[(size_t)(rows - 1 - r) * cols + x] = (m->mask_host[k] == 0 && d[k] == 0.f) ? 0 : 255;
      }
    TTRY(tdr_png_write_gray8((std::string(dir) + "/class" + std::to_string(c) + ".png").c_str(), img.data(), cols, rows));
  }
  return TDR_OK;
}
// TopDownMap::loadRasterizedMaps (:213-224) followed by what the constructor does with the rasters (:48-58): the geometric
// layers derived from them and computeDists on both — on the device (tdr_k_map_from_rasters).
static int map_load_rasters(tdr_map* m, const char* dir, int num_classes, float resolution, int center_x, int center_y);
int tdr_map_load_rasters(tdr_map* m, const char* dir, int num_classes, float resolution, int center_x, int center_y) {
  try {   // (no exception crosses the C ABI: a file that makes an allocation fail is an error code)
    return map_load_rasters(m, dir, num_classes, resolution, center_x, center_y);
  } catch (const std::exception& e) {
    return failh(TDR_ERR_NOMEM, "map_load_rasters: %s", e.what());
  }
}
static int map_load_rasters(tdr_map* m, const char* dir, int num_classes, float resolution, int center_x, int center_y) {
  if (!m || !dir) return failh(TDR_ERR_ARG, "map_load_rasters: null pointer");
  if (num_classes < 1 || num_classes > TDR_MAX_CLASSES || !(resolution > 0.f))
    return failh(TDR_ERR_ARG, "map_load_rasters: bad class count / resolution");
  std::vector<uint8_t> planes, one;
  int w = 0, h = 0;
  for (int c = 0; c < num_classes; c++) {
    int w2 = 0, h2 = 0;
    TTRY(tdr_png_read_gray8((std::string(dir) + "/class" + std::to_string(c) + ".png").c_str(), one, w2, h2));
    if (c == 0) { w = w2; h = h2; }
    if (w2 != w || h2 != h) return failh(TDR_ERR_ARG, "map_load_rasters: class%d.png differs in size from class0.png", c);
    planes.insert(planes.end(), one.begin(), one.end());
  }
  const int rows = h, cols = w;
  const size_t ncell = (size_t)rows * cols;
  DevBuf<uint8_t> d_planes, d_ws, d_mask;
  DevBuf<float> d_maps;
  TTRY(d_planes.resize(planes.size()));
  TTRY(d_ws.resize(tdr_map_ingest_workspace_bytes(num_classes, rows, cols)));
  HTRY(hipMemcpy(d_planes.p, planes.data(), planes.size(), hipMemcpyHostToDevice));
  TTRY(m->rec.resize(tdr_map_rec_floats_total(num_classes, rows, cols)));
  TTRY(tdr_k_map_from_rasters(d_planes.p, num_classes, rows, cols, resolution, m->rec.p, d_ws.p, nullptr));
  TTRY(d_maps.resize(ncell * num_classes));
  TTRY(d_mask.resize(ncell));
  TTRY(tdr_k_unpack_map(m->rec.p, num_classes, rows, cols, d_maps.p, d_mask.p, nullptr));
  m->maps_host.resize(ncell * num_classes);
  m->mask_host.resize(ncell);
  HTRY(hipMemcpy(m->maps_host.data(), d_maps.p, ncell * num_classes * sizeof(float), hipMemcpyDeviceToHost));
  HTRY(hipMemcpy(m->mask_host.data(), d_mask.p, ncell, hipMemcpyDeviceToHost));
  m->desc.rec = m->rec.p;
  m->desc.ncls = num_classes;
  m->desc.rows = rows;
  m->desc.cols = cols;
  m->desc.rec_floats = tdr_rec_floats(num_classes);
  m->desc.resolution = resolution;
  m->center_x = center_x;
  m->center_y = center_y;
  TTRY(map_compact(m));
  TTRY(map_make_geo(m, false));   // getGeoRasterMap + computeDists (:48-58)
  m->have_map = true;             // :63
  if (m->nb > 0) return tdr_map_sample_pts_polar(m, m->nb, m->nr, m->ang_res);
  return TDR_OK;
}

// TopDownMap::getClassesAtPoint(Vector2i) (top_down_map.cpp:159-170): bit c set = class c present (< 1 px away)
int tdr_map_classes_at_point(const tdr_map* m, int px, int py, uint32_t* class_bits) {
  if (!m || !class_bits || !m->have_map) return failh(TDR_ERR_ARG, "classes_at_point: no map");
  const int rows = m->desc.rows, cols = m->desc.cols;
  const int c0 = (int)((float)px / m->desc.resolution), c1 = (int)((float)py / m->desc.resolution);
  uint32_t bits = 0;
  if (c0 < cols && c1 < rows && c0 >= 0 && c1 >= 0)
    for (int c = 0; c < m->desc.ncls; c++)
      if (m->maps_host[(size_t)c * rows * cols + c1 + (size_t)rows * c0] < 1) bits |= 1u << c;
  *class_bits = bits;
  return TDR_OK;
}

// ---- ScanRenderer(Polar) ------------------------------------------------------------------------------------------------
int tdr_renderer_create(const int32_t* flatten_lut256, tdr_renderer** out) {
  if (!flatten_lut256 || !out) return failh(TDR_ERR_ARG, "renderer_create: null pointer");
  if (tdr_device_count() < 1) return failh(TDR_ERR_HIP, "renderer_create: no HIP device (there is no CPU fallback)");
  tdr_renderer* r = new tdr_renderer();
  int rc = r->lut.resize(256);
  if (rc == TDR_OK && hipMemcpy(r->lut.p, flatten_lut256, 256 * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
    rc = failh(TDR_ERR_HIP, "renderer_create: lut upload failed");
  if (rc != TDR_OK) {
    delete r;
    return rc;
  }
  *out = r;
  return TDR_OK;
}
void tdr_renderer_destroy(tdr_renderer* r) { delete r; }

// renderSemanticTopDown (scan_renderer_polar.cpp:83-109 when polar != 0, scan_renderer.cpp:55-78 otherwise).
// pts: HOST points; imgs_out: HOST [ncls][rows*cols] column-major images, written in place (may be NULL: the render
// then only stays on the device for tdr_filter_update).
int tdr_renderer_render(tdr_renderer* r, int polar, const float* pts, int stride, int ioff, int64_t n, float res,
                        float ang_res, int ncls, int rows, int cols, float* imgs_out) {
  if (!r) return failh(TDR_ERR_ARG, "render: null renderer");
  if (n < 0 || (n > 0 && !pts)) return failh(TDR_ERR_ARG, "render: null points");
  if (ncls < 1 || rows < 1 || cols < 1) return TDR_OK;  // `if (imgs.size() < 1) return;` (:85)
  const size_t P = (size_t)rows * cols;
  TTRY(r->pts.resize((size_t)std::max<int64_t>(n, 1) * stride));
  TTRY(r->img.resize(P * ncls));
  TTRY(r->pk.resize(P * tdr_rec_floats(ncls)));
  if (n > 0) HTRY(hipMemcpy(r->pts.p, pts, (size_t)n * stride * sizeof(float), hipMemcpyHostToDevice));
  TTRY(r->keys.resize((size_t)tdr_raster_workspace_bytes(std::max<int64_t>(n, 1))));
  if (polar)
    TTRY(tdr_k_raster_polar(r->pts.p, stride, ioff, n, res, ang_res, r->lut.p, ncls, rows, cols, r->img.p, r->pk.p,
                            r->keys.p, nullptr));
  else
    TTRY(tdr_k_raster_cart(r->pts.p, stride, ioff, n, res, r->lut.p, ncls, rows, cols, r->img.p, r->pk.p, r->keys.p,
                           nullptr));
  if (imgs_out) HTRY(hipMemcpy(imgs_out, r->img.p, P * ncls * sizeof(float), hipMemcpyDeviceToHost));
  else HTRY(hipDeviceSynchronize());
  r->ncls = ncls;
  r->rows = rows;
  r->cols = cols;
  r->have_scan = true;
  return TDR_OK;
}

// renderGeometricTopDown (scan_renderer_polar.cpp:6-81 when polar != 0, scan_renderer.cpp:7-53 otherwise).  pts: HOST
// organised cloud (element idy*width + idx); imgs_out: HOST [2][rows*cols] column-major (ground, obstacles).
int tdr_renderer_render_geo(tdr_renderer* r, int polar, const float* pts, int stride, int64_t width, int64_t height,
                            float res, float ang_res, int rows, int cols, float* imgs_out) {
  if (!r || !imgs_out) return failh(TDR_ERR_ARG, "render_geo: null pointer");
  const int64_t n = width * height;
  if (width < 0 || height < 0 || (n > 0 && !pts)) return failh(TDR_ERR_ARG, "render_geo: bad cloud");
  if (rows < 1 || cols < 1) return TDR_OK;
  const size_t P = (size_t)rows * cols;
  TTRY(r->pts.resize((size_t)std::max<int64_t>(n, 1) * stride));
  TTRY(r->geo.resize(2 * P));
  if (n > 0) HTRY(hipMemcpy(r->pts.p, pts, (size_t)n * stride * sizeof(float), hipMemcpyHostToDevice));
  if (polar) {
    TTRY(r->geo_ws.resize((size_t)tdr_raster_geo_workspace_bytes(std::max<int64_t>(n, 1))));
    TTRY(tdr_k_raster_geo_polar(r->pts.p, stride, width, height, res, ang_res, rows, cols, r->geo.p, r->geo_ws.p, nullptr));
  } else {
    TTRY(tdr_k_raster_geo_cart(r->pts.p, stride, width, height, res, rows, cols, r->geo.p, nullptr));
  }
  HTRY(hipMemcpy(imgs_out, r->geo.p, 2 * P * sizeof(float), hipMemcpyDeviceToHost));
  return TDR_OK;
}

// ---- ParticleFilter ------------------------------------------------------------------------------------------------------
static int filter_create(tdr_map* map, int n_max, const tdr_filter_params* fp, uint32_t seed, tdr_comm* comm,
                         tdr_filter** out) {
  if (!map || !fp || !out || n_max < 1) return failh(TDR_ERR_ARG, "filter_create: bad arguments");
  const int world = comm ? tdr_comm_world(comm) : 1;
  if (n_max % world) return failh(TDR_ERR_ARG, "filter_create: n_max = %d is not a multiple of the %d ranks", n_max, world);
  tdr_filter* f = new tdr_filter();
  f->map = map;
  f->fp = *fp;
  f->n_max = n_max;
  f->comm = comm;
  f->world = world;
  f->rank = comm ? tdr_comm_rank(comm) : 0;
  f->cap = n_max / world;
  f->seed = seed;
  f->rng = tdr_rng_create(seed);  // explicit seed instead of std::random_device (particle_filter.cpp:4-5)
  // seed 0 = "unseeded", like the reference's std::random_device: nothing to reproduce, so propagate draws its noise on
  // the device; a non-zero seed asks for the reference-ordered std::mt19937 stream (tdr_filter_configure overrides)
  f->parity_rng = seed != 0;
  int rc = TDR_OK;
  const size_t cap = (size_t)f->cap, N = (size_t)n_max;
  if (rc == TDR_OK) rc = f->st.resize(TDR_ST_FIELDS * cap);
  if (rc == TDR_OK) rc = f->st_new.resize(TDR_ST_FIELDS * cap);
  if (rc == TDR_OK) rc = f->last_dist.resize(cap);
  if (rc == TDR_OK) rc = f->raw_w.resize(cap);
  if (rc == TDR_OK) rc = f->w.resize(N);
  if (rc == TDR_OK) rc = f->runmax.resize(N);
  if (rc == TDR_OK) rc = f->pfx_ws.resize((size_t)tdr_prefix_workspace_bytes((int64_t)N));
  if (rc == TDR_OK) rc = f->idx.resize(cap);
  if (rc == TDR_OK) rc = f->perm.resize(cap);
  if (rc == TDR_OK) rc = f->info.resize(TDR_UW_INFO_FLOATS);
  if (rc == TDR_OK) rc = f->stats.resize(TDR_MEAN_COV_FLOATS);
  if (rc == TDR_OK) rc = f->aos.resize(N);
  if (rc == TDR_OK) rc = f->z4.resize(4 * cap);
  if (rc == TDR_OK && comm) {
    rc = f->xchg_in.resize(2 * cap);
    if (rc == TDR_OK) rc = f->xchg_out.resize(2 * N);
    if (rc == TDR_OK) rc = f->raw_glob.resize(N);
    if (rc == TDR_OK) rc = f->ld_glob.resize(N);
    if (rc == TDR_OK) rc = f->st_send.resize(TDR_ST_FIELDS * cap);
    if (rc == TDR_OK) rc = f->st_all.resize(TDR_ST_FIELDS * N);
    if (rc == TDR_OK) rc = f->st_glob.resize(TDR_ST_FIELDS * N);
  }
  if (rc == TDR_OK && hipMemset(f->last_dist.p, 0, cap * sizeof(float)) != hipSuccess) rc = failh(TDR_ERR_HIP, "memset");
  if (rc == TDR_OK) rc = tdr_score_ctx_create(&f->score_ctx);
  if (rc != TDR_OK) {
    tdr_filter_destroy(f);
    return rc;
  }
  *out = f;
  return TDR_OK;
}
int tdr_filter_create(tdr_map* map, int n_max, const tdr_filter_params* fp, uint32_t seed, tdr_filter** out) {
  return filter_create(map, n_max, fp, seed, nullptr, out);
}
// particles sharded over the ranks of `comm` (not owned; must outlive the filter)
int tdr_filter_create_sharded(tdr_map* map, int n_max, const tdr_filter_params* fp, uint32_t seed, tdr_comm* comm,
                              tdr_filter** out) {
  if (!comm) return failh(TDR_ERR_ARG, "filter_create_sharded: null comm");
  return filter_create(map, n_max, fp, seed, comm, out);
}
int64_t tdr_filter_num_local(const tdr_filter* f) { return f ? f->nl() : 0; }
void tdr_filter_destroy(tdr_filter* f) {
  if (!f) return;
  if (f->rng && f->rng_owned) tdr_rng_destroy(f->rng);
  tdr_score_ctx_destroy(f->score_ctx);
  tdr_rng_pipe_destroy(f->pipe);
  delete f;
}

static int rng_to_host(tdr_filter* f);   // (below, with the propagate step)
int tdr_filter_configure(tdr_filter* f, int parity_rng, int locality_every) {
  if (!f) return failh(TDR_ERR_ARG, "filter_configure: null filter");
  if (!parity_rng) TTRY(rng_to_host(f));
  f->parity_rng = parity_rng != 0;
  f->locality_every = locality_every;
  return TDR_OK;
}

static void note_uniform_scale(tdr_filter* f, const tdr_state* s, int64_t n) {
  f->uniform_scale = 0.f;
  if (!f->scale_frozen || n < 1 || !(s[0].scale > 0)) return;
  for (int64_t i = 1; i < n; i++)
    if (s[i].scale != s[0].scale) return;
  f->uniform_scale = s[0].scale;
}

// states: the GLOBAL particle array; a sharded filter keeps this rank's slice
int tdr_filter_set_states(tdr_filter* f, const tdr_state* states, int64_t n) {
  if (!f || (n > 0 && !states) || n < 0 || n > f->n_max) return failh(TDR_ERR_ARG, "filter_set_states: bad arguments");
  if (n % f->world) return failh(TDR_ERR_ARG, "filter_set_states: %lld particles over %d ranks", (long long)n, f->world);
  const int64_t nl = n / f->world;
  if (nl > 0) {
    HTRY(hipMemcpy(f->aos.p, states + (size_t)f->rank * nl, (size_t)nl * sizeof(tdr_state), hipMemcpyHostToDevice));
    TTRY(tdr_k_states_aos_to_soa(f->aos.p, nl, f->st.p, f->cap, f->stream));
    HTRY(hipDeviceSynchronize());
  }
  f->n = n;
  f->states_changed();
  f->maybe_uninit = false;
  for (int64_t i = 0; i < n; i++) f->maybe_uninit |= states[i].have_init == 0;
  if (f->fp.fixed_scale > 0) f->scale_frozen = true;
  note_uniform_scale(f, states, n);
  return TDR_OK;
}

// this rank's particles (all of them when the filter is not sharded): n <= tdr_filter_num_local
int tdr_filter_get_states(tdr_filter* f, tdr_state* out, int64_t n) {
  if (!f || !out || n < 0 || n > f->nl()) return failh(TDR_ERR_ARG, "filter_get_states: bad arguments");
  if (n == 0) return TDR_OK;
  TTRY(tdr_k_states_soa_to_aos(f->st.p, f->cap, n, f->aos.p, f->stream));
  HTRY(hipMemcpy(out, f->aos.p, (size_t)n * sizeof(tdr_state), hipMemcpyDeviceToHost));
  return TDR_OK;
}

// ParticleFilter::initializeParticles (particle_filter.cpp:19-84)
int tdr_filter_initialize_particles(tdr_filter* f) {
  if (!f || !f->map || !f->map->have_map) return failh(TDR_ERR_ARG, "initialize_particles: no map");
  tdr_map* m = f->map;
  tdr_filter_params& p = f->fp;
  if (p.fixed_scale >= 0) f->scale_frozen = true;  // :23-25
  const float inf = std::numeric_limits<float>::infinity();
  if (f->scale_frozen && p.init_pos_m_x != inf) {   // :27-53
    p.init_pos_px_x = (p.init_pos_m_x * p.fixed_scale) + (float)m->center_x;
    p.init_pos_px_y = (p.init_pos_m_y * p.fixed_scale) + (float)m->center_y;
    if (p.init_pos_px_x < 0 || p.init_pos_px_x >= (float)m->desc.cols || p.init_pos_px_y < 0 ||
        p.init_pos_px_y >= (float)m->desc.rows)
      return TDR_OK;  // "No map received for input loc"
    bool good = false;
    for (int dx = -4; dx <= 4 && !good; dx++)
      for (int dy = -4; dy <= 4 && !good; dy++) {
        uint32_t bits = 0;
        TTRY(tdr_map_classes_at_point(m, (int)(p.init_pos_px_x + dx), (int)(p.init_pos_px_y + dy), &bits));
        good = (bits & 2u) != 0;
      }
    if (!good) return TDR_OK;  // "No road in map at init location"
  }
  std::vector<tdr_state> states((size_t)f->n_max + 16);
  int64_t n = 0;
  TTRY(rng_to_host(f));
  TTRY(tdr_init_particles_host(f->rng, m->maps_host.data(), m->desc.ncls, m->desc.rows, m->desc.cols,
                               m->desc.resolution, &p, (int)f->n_max, states.data(), &n));
  n = std::min<int64_t>(n, f->n_max);
  n -= n % f->world;
  return tdr_filter_set_states(f, states.data(), n);
}

// the generator's stream continues on the device / on the host (see tdr_filter::rng_dev)
static bool rng_on_device(const tdr_filter* f) { return f->pipe && tdr_rng_pipe_on_device(f->pipe); }
static int rng_to_device(tdr_filter* f) {
  if (rng_on_device(f)) return TDR_OK;
  if (!f->pipe) TTRY(tdr_rng_pipe_create(f->n_max, &f->pipe));
  return tdr_rng_pipe_from_host(f->pipe, f->rng, f->stream);
}
static int rng_to_host(tdr_filter* f) {
  if (!rng_on_device(f)) return TDR_OK;
  return tdr_rng_pipe_to_host(f->pipe, f->rng, f->stream);
}
static bool rng_device_capable(const tdr_filter* f) { return f->rng_owned && f->parity_rng; }

// ParticleFilter::propagate (particle_filter.cpp:86-92)
static int filter_propagate(tdr_filter* f, float tx, float ty, float omega, bool scale_freeze) {
  if (f->n == 0) return TDR_OK;
  f->states_changed();
  const int64_t nl = f->nl();
  const float* z = nullptr;
  if (rng_device_capable(f)) {
    // the reference draws serially in GLOBAL particle order from one generator: every rank continues the same stream on
    // its device (same state everywhere) and keeps the normals of its own particles — nothing is drawn on the host
    TTRY(rng_to_device(f));
    TTRY(tdr_rng_pipe_normals(f->pipe, f->n, (int64_t)f->rank * nl, (int64_t)(f->rank + 1) * nl, scale_freeze ? 1 : 0, &z,
                              f->stream));
  } else if (f->parity_rng) {
    // a generator shared with the caller (StateParticle's surface): the host draws, serially
    std::vector<float> zh((size_t)4 * f->n);
    TTRY(tdr_propagate_normals_host(f->rng, f->n, scale_freeze ? 1 : 0, zh.data()));
    HTRY(hipMemcpyAsync(f->z4.p, zh.data() + (size_t)4 * f->rank * nl, (size_t)4 * nl * sizeof(float),
                        hipMemcpyHostToDevice, f->stream));
    HTRY(hipStreamSynchronize(f->stream));
    z = f->z4.p;
  }
  return tdr_k_propagate(f->st.p, f->cap, nl, f->last_dist.p, tx, ty, omega, scale_freeze ? 1 : 0, f->fp.pos_cov,
                         f->fp.theta_cov, z, f->seed, f->prop_calls++, (int64_t)f->rank * nl, f->stream);
}
int tdr_filter_propagate(tdr_filter* f, float tx, float ty, float omega) {
  if (!f) return failh(TDR_ERR_ARG, "filter_propagate: null filter");
  return filter_propagate(f, tx, ty, omega, f->scale_frozen);
}
// StateParticle::propagate(trans, omega, scale_freeze) (state_particle.cpp:57-78): the freeze flag is the caller's
int tdr_filter_propagate_freeze(tdr_filter* f, float tx, float ty, float omega, int scale_freeze) {
  if (!f) return failh(TDR_ERR_ARG, "filter_propagate_freeze: null filter");
  return filter_propagate(f, tx, ty, omega, scale_freeze != 0);
}
// The filter draws from the caller's std::mt19937 from now on (the reference's particles share ONE generator with
// their filter, state_particle.h:61-64).  `mt19937` must point to a std::mt19937 of the libstdc++ this library was
// built with; it is not owned.
int tdr_filter_share_rng(tdr_filter* f, void* mt19937) {
  if (!f || !mt19937) return failh(TDR_ERR_ARG, "filter_share_rng: bad arguments");
  if (f->pipe) { tdr_rng_pipe_destroy(f->pipe); f->pipe = nullptr; }   // (the filter's own stream ends here)
  if (f->rng && f->rng_owned) tdr_rng_destroy(f->rng);
  f->rng = mt19937;
  f->rng_owned = false;
  f->parity_rng = true;
  return TDR_OK;
}
// StateParticle's constructor with init == true (state_particle.cpp:3-49) for particle 0 of the filter
int tdr_filter_init_one(tdr_filter* f) {
  if (!f || !f->map || !f->map->have_map) return failh(TDR_ERR_ARG, "filter_init_one: no map");
  tdr_map* m = f->map;
  tdr_state st;
  TTRY(rng_to_host(f));
  TTRY(tdr_init_particle_host(f->rng, m->maps_host.data(), m->desc.ncls, m->desc.rows, m->desc.cols, m->desc.resolution,
                              &f->fp, &st));
  return tdr_filter_set_states(f, &st, 1);
}

// ParticleFilter::update (particle_filter.cpp:94-189).  scan_imgs: HOST [ncls][nb*nr] column-major images, or NULL to
// score against `renderer`'s last render without a host round trip.  n_target < 0 keeps the particle count
// (the adaptive count of :151-157 is an explicit input; the reference feeds it from an OpenCV EM thread).
static int filter_score(tdr_filter* f, const float* scan_imgs, const tdr_renderer* renderer, float res);
static int filter_resample(tdr_filter* f, int64_t n_target);
int tdr_filter_update(tdr_filter* f, const float* scan_imgs, const tdr_renderer* renderer, float res, int64_t n_target) {
  if (!f || !f->map || !f->map->have_map) return failh(TDR_ERR_ARG, "filter_update: no map");
  if (f->n == 0) return TDR_OK;  // :96-99
  TTRY(filter_score(f, scan_imgs, renderer, res));
  const int64_t n = f->n, nl = f->nl();
  const float *raw = f->raw_w.p, *ld = f->last_dist.p;
  if (f->comm) {
    // ONE all-gather of {raw weight, last_dist}: afterwards every rank computes the same statistics and the same
    // order-exact running sum on the same global arrays (replaces the north star's all-reduce, whose result
    // would depend on the reduction tree)
    TTRY(tdr_k_shard_pack2(f->raw_w.p, f->last_dist.p, nl, f->xchg_in.p, f->stream));
    TTRY(tdr_comm_all_gather(f->comm, f->xchg_in.p, f->xchg_out.p, (size_t)2 * nl * sizeof(float), f->stream));
    TTRY(tdr_k_shard_unpack2(f->xchg_out.p, f->world, nl, f->raw_glob.p, f->ld_glob.p, f->stream));
    raw = f->raw_glob.p;
    ld = f->ld_glob.p;
  }
  TTRY(tdr_k_update_weights(raw, ld, n, f->w.p, f->info.p, f->stream));
  return filter_resample(f, n_target);
}
// ParticleFilter::update with the geometric images entering the score (state_particle.cpp:145-152, opt-in)
int tdr_filter_update_geo(tdr_filter* f, const float* scan_imgs, const float* geo_imgs, float res, int64_t n_target) {
  if (!f || !f->map || !f->map->have_map) return failh(TDR_ERR_ARG, "filter_update_geo: no map");
  if (!scan_imgs || !geo_imgs) return failh(TDR_ERR_ARG, "filter_update_geo: null images");
  if (f->comm) return failh(TDR_ERR_ARG, "filter_update_geo: not available on a sharded filter");
  if (f->n == 0) return TDR_OK;
  tdr_map* m = f->map;
  if (m->nb < 1 || !m->tab.p) return failh(TDR_ERR_ARG, "filter_update_geo: samplePtsPolar was never called");
  TTRY(map_ensure_geo(m));
  f->fp.num_classes = m->desc.ncls;
  const int ncls = m->desc.ncls, nb = m->nb, nr = m->nr;
  const size_t P = (size_t)nb * nr;
  TTRY(f->scan_img.resize(P * std::max(ncls, 2)));
  TTRY(f->scan_pk.resize(P * tdr_rec_floats(ncls)));
  TTRY(f->geo_pk.resize(P * 4));
  HTRY(hipMemcpyAsync(f->scan_img.p, scan_imgs, P * ncls * sizeof(float), hipMemcpyHostToDevice, f->stream));
  TTRY(tdr_k_pack_scan(f->scan_img.p, ncls, nb, nr, f->scan_pk.p, f->stream));
  HTRY(hipMemcpyAsync(f->scan_img.p, geo_imgs, P * 2 * sizeof(float), hipMemcpyHostToDevice, f->stream));
  TTRY(tdr_k_pack_scan(f->scan_img.p, 2, nb, nr, f->geo_pk.p, f->stream));
  double gs[2] = {0, 0};   // top_down_geo[i].sum(): Eigen's order is unspecified; counts are small integers, any order is exact
  for (int i = 0; i < 2; i++)
    for (size_t k = 0; k < P; k++) gs[i] += (double)geo_imgs[P * i + k];
  const int64_t n = f->n;
  const int32_t* perm = nullptr;
  if (f->locality_every > 0) {
    TTRY(f->loc_tmp.resize(tdr_locality_tmp_ints(n, m->desc.rows, m->desc.cols)));
    TTRY(tdr_k_locality_order(f->st.p, f->cap, n, m->desc.rows, m->desc.cols, f->perm.p, f->loc_tmp.p, f->stream));
    perm = f->perm.p;
  }
  f->states_changed();   // (the init search writes headings)
  TTRY(f->ws.resize(tdr_score_geo_workspace_floats(ncls, nb, nr, n, n)));
  TTRY(tdr_k_score_polar_geo(&m->desc, &m->geo_desc, m->tab.p, f->scan_pk.p, f->geo_pk.p, (float)gs[0], (float)gs[1], nb, nr,
                             res, &f->fp, f->st.p, f->cap, n, n, perm, f->uniform_scale, f->maybe_uninit ? 1 : 0,
                             f->raw_w.p, f->ws.p, f->stream));
  if (f->maybe_uninit && !(f->fp.force_on_map || f->fp.fixed_scale < 0)) f->maybe_uninit = false;
  TTRY(tdr_k_update_weights(f->raw_w.p, f->last_dist.p, n, f->w.p, f->info.p, f->stream));
  return filter_resample(f, n_target);
}
// StateParticle::computeWeight for every particle (state_particle.cpp:157-219): raw weights only, no statistics, no
// resampling; read them with tdr_filter_get_raw_weights
int tdr_filter_compute_weights(tdr_filter* f, const float* scan_imgs, const tdr_renderer* renderer, float res) {
  if (!f || !f->map || !f->map->have_map) return failh(TDR_ERR_ARG, "filter_compute_weights: no map");
  if (f->n == 0) return TDR_OK;
  return filter_score(f, scan_imgs, renderer, res);
}
int tdr_filter_get_raw_weights(tdr_filter* f, float* out, int64_t n) {
  if (!f || !out || n < 0 || n > f->cap) return failh(TDR_ERR_ARG, "filter_get_raw_weights: bad arguments");
  HTRY(hipMemcpy(out, f->raw_w.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return TDR_OK;
}
int tdr_filter_get_last_dist(tdr_filter* f, float* out, int64_t n) {
  if (!f || !out || n < 0 || n > f->cap) return failh(TDR_ERR_ARG, "filter_get_last_dist: bad arguments");
  HTRY(hipMemcpy(out, f->last_dist.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return TDR_OK;
}
static int filter_score(tdr_filter* f, const float* scan_imgs, const tdr_renderer* renderer, float res) {
  tdr_map* m = f->map;
  if (m->nb < 1 || !m->tab.p) return failh(TDR_ERR_ARG, "filter_update: samplePtsPolar was never called");
  f->fp.num_classes = m->desc.ncls;
  const int ncls = m->desc.ncls, nb = m->nb, nr = m->nr;
  const size_t P = (size_t)nb * nr;
  const size_t pk_floats = P * tdr_rec_floats(ncls);
  const float* pk = nullptr;
  if (scan_imgs) {
    TTRY(f->scan_img.resize(P * ncls));
    TTRY(f->scan_pk.resize(pk_floats));
    HTRY(hipMemcpyAsync(f->scan_img.p, scan_imgs, P * ncls * sizeof(float), hipMemcpyHostToDevice, f->stream));
    TTRY(tdr_k_pack_scan(f->scan_img.p, ncls, nb, nr, f->scan_pk.p, f->stream));
    pk = f->scan_pk.p;
  } else if (renderer) {
    if (!renderer->have_scan) return failh(TDR_ERR_ARG, "filter_update: no scan");
    if (renderer->ncls != ncls || renderer->rows != nb || renderer->cols != nr)
      return failh(TDR_ERR_ARG, "filter_update: render shape %dx%dx%d does not match the map's %dx%dx%d",
                   renderer->ncls, renderer->rows, renderer->cols, ncls, nb, nr);
    pk = renderer->pk.p;
  } else if (!(f->comm && f->rank != 0)) {
    return failh(TDR_ERR_ARG, "filter_update: no scan");
  }
  if (f->comm) {
    // the rasterised scan is produced once, on rank 0, and broadcast (north star): 2 MB at config 2
    TTRY(f->pk_recv.resize(pk_floats));
    if (f->rank == 0) HTRY(hipMemcpyAsync(f->pk_recv.p, pk, pk_floats * sizeof(float), hipMemcpyDeviceToDevice, f->stream));
    TTRY(tdr_comm_broadcast(f->comm, f->pk_recv.p, pk_floats * sizeof(float), 0, f->stream));
    pk = f->pk_recv.p;
  }
  const int64_t n = f->nl();   // this rank's particles
  const int32_t* perm = nullptr;
  if (f->locality_every > 0) {
    TTRY(f->loc_tmp.resize(tdr_locality_tmp_ints(n, m->desc.rows, m->desc.cols)));
    TTRY(tdr_k_locality_order(f->st.p, f->cap, n, m->desc.rows, m->desc.cols, f->perm.p, f->loc_tmp.p, f->stream));
    perm = f->perm.p;
  }
  f->states_changed();   // (the init search writes headings)
  TTRY(f->ws.resize(tdr_score_workspace_floats(ncls, nb, nr, n, f->n)));
  if (f->maybe_uninit && !m->desc.rec16 && f->n >= tdr_config_rec16_min_particles(-1)) {
    // the search over this many particles pays for pre-split half records (filters on one map share the scratch: their
    // searches must not overlap in time)
    const size_t b16 = tdr_map_rec16_bytes(ncls, m->desc.rows, m->desc.cols);
    if (b16) {
      TTRY(m->rec16.resize(b16));
      m->desc.rec16 = m->rec16.p;
    }
  }
  TTRY(tdr_score_ctx_set_polar_factors(f->score_ctx, m->fac.p, nb, nr));
  TTRY(tdr_k_score_polar_ctx(&m->desc, m->tab.p, pk, nb, nr, res, &f->fp, f->st.p, f->cap, n, f->n, perm, f->uniform_scale,
                             f->maybe_uninit ? 1 : 0, f->raw_w.p, f->ws.p, f->score_ctx, f->stream));
  // the search initialises every un-gated particle; only gated ones (state_particle.cpp:163-176) can stay un-initialised
  if (f->maybe_uninit && !(f->fp.force_on_map || f->fp.fixed_scale < 0)) f->maybe_uninit = false;
  return TDR_OK;
}
// statistics are done: running sum, resample, gather, bookkeeping (particle_filter.cpp:151-188)
static int filter_resample(tdr_filter* f, int64_t n_target) {
  const int64_t n = f->n, nl = f->nl();
  int64_t n_new = n;
  if (n_target >= 0) n_new = std::max<int64_t>(1, std::min<int64_t>(n_target, f->n_max));
  n_new = std::max<int64_t>(f->world, n_new - n_new % f->world);   // whole shards
  const int64_t nl_new = n_new / f->world, i0 = (int64_t)f->rank * nl_new;
  TTRY(f->ml_dev.resize(12));
  TTRY(tdr_k_prefix(f->w.p, n, f->runmax.p, f->pfx_ws.p, f->stream));
  // :172-173 (every rank owns an identically seeded generator); each rank draws its own slice [i0, i0 + nl_new) of the new
  // set, idx holds GLOBAL source indices
  if (rng_on_device(f)) {   // the stream is on the device: so is the draw
    const float* shift_dev = nullptr;
    TTRY(tdr_rng_pipe_uniform(f->pipe, &shift_dev, f->stream));
    TTRY(tdr_k_resample_dev(f->runmax.p, n, n_new, shift_dev, i0, i0 + nl_new, f->idx.p, f->stream));
  } else {
    const float shift = tdr_rng_uniform_host(f->rng);
    TTRY(tdr_k_resample(f->runmax.p, n, n_new, shift, i0, i0 + nl_new, f->idx.p, f->stream));
  }
  if (f->comm) {
    // the second all-gather: the pre-resample state planes, [rank][7][nl] (28 B x N)
    for (int k = 0; k < TDR_ST_FIELDS; k++)
      HTRY(hipMemcpyAsync(f->st_send.p + (size_t)k * nl, f->st.p + (size_t)k * f->cap, (size_t)nl * sizeof(float),
                          hipMemcpyDeviceToDevice, f->stream));
    TTRY(tdr_comm_all_gather(f->comm, f->st_send.p, f->st_all.p, (size_t)TDR_ST_FIELDS * nl * sizeof(float), f->stream));
    TTRY(tdr_k_gather_states(f->st_all.p, 0, nl, f->idx.p, nl_new, f->st_new.p, f->cap, f->stream));
    TTRY(tdr_k_save_ml_state(f->info.p, f->st_all.p, 0, nl, n, f->ml_dev.p, f->stream));
  } else {
    TTRY(tdr_k_gather_states(f->st.p, f->cap, 0, f->idx.p, n_new, f->st_new.p, f->cap, f->stream));
    // max_likelihood_particle_ = particles_[argmax] (:145-147): keep that particle's pre-resample state (on the device:
    // the update returns without waiting for the GPU)
    TTRY(tdr_k_save_ml_state(f->info.p, f->st.p, f->cap, 0, n, f->ml_dev.p, f->stream));
  }
  f->have_ml = true;
  f->states_changed();
  std::swap(f->st.p, f->st_new.p);  // :187
  f->n = n_new;
  f->step++;
  return TDR_OK;
}
// The current particle set of ALL ranks as a plain SoA (pose statistics, the mixture fit): the filter's own arrays when
// it is not sharded, else one all-gather of the state planes.
static int filter_global_states(tdr_filter* f, const float** st, int64_t* cap) {
  if (!f->comm) {
    *st = f->st.p;
    *cap = f->cap;
    return TDR_OK;
  }
  const int64_t nl = f->nl();
  for (int k = 0; k < TDR_ST_FIELDS; k++)
    HTRY(hipMemcpyAsync(f->st_send.p + (size_t)k * nl, f->st.p + (size_t)k * f->cap, (size_t)nl * sizeof(float),
                        hipMemcpyDeviceToDevice, f->stream));
  TTRY(tdr_comm_all_gather(f->comm, f->st_send.p, f->st_all.p, (size_t)TDR_ST_FIELDS * nl * sizeof(float), f->stream));
  TTRY(tdr_k_unshard_states(f->st_all.p, f->world, nl, f->st_glob.p, f->n_max, f->stream));
  *st = f->st_glob.p;
  *cap = f->n_max;
  return TDR_OK;
}

int tdr_filter_get_weights(tdr_filter* f, float* out, int64_t n) {
  if (!f || !out || n < 0 || n > f->n_max) return failh(TDR_ERR_ARG, "filter_get_weights: bad arguments");
  HTRY(hipMemcpy(out, f->w.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return TDR_OK;
}
int tdr_filter_get_resample_indices(tdr_filter* f, int32_t* out, int64_t n) {
  if (!f || !out || n < 0 || n > f->nl()) return failh(TDR_ERR_ARG, "filter_get_resample_indices: bad arguments");
  HTRY(hipMemcpy(out, f->idx.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return TDR_OK;
}

// meanLikelihood + computeMeanCov (particle_filter.cpp:191-220); about_max != 0: maxLikelihood + computeCov (:222-236)
int tdr_filter_mean_cov(tdr_filter* f, int about_max, float state[4], float cov[16]) {
  if (!f) return failh(TDR_ERR_ARG, "filter_mean_cov: null filter");
  if (cov) std::memset(cov, 0, 16 * sizeof(float));
  if (state) std::memset(state, 0, 4 * sizeof(float));
  if (f->n < 1) return TDR_OK;  // :207-209
  float out[24];
  const float* gst = nullptr;
  int64_t gcap = 0;
  // (a sharded filter's ranks make the same calls in the same order, so the cache is valid on all of them or on none:
  // the all-gather inside filter_global_states stays collective)
  if (about_max || !f->mean_cov_valid) TTRY(filter_global_states(f, &gst, &gcap));
  if (!about_max) {
    if (!f->mean_cov_valid) {
      TTRY(tdr_k_mean_cov(gst, gcap, f->n, nullptr, f->stats.p, f->stream));
      HTRY(hipMemcpy(f->mean_cov_host, f->stats.p, sizeof(out), hipMemcpyDeviceToHost));
      f->mean_cov_valid = true;
    }
    std::memcpy(out, f->mean_cov_host, sizeof(out));
    if (state) std::memcpy(state, out, 4 * sizeof(float));
  } else {
    float ref[4] = {0, 0, 0, 0};
    if (!f->have_ml) return failh(TDR_ERR_ARG, "filter_mean_cov: no update yet, there is no max-likelihood particle");
    TTRY(tdr_k_mean_cov(gst, gcap, f->n, f->ml_dev.p + 8, f->stats.p, f->stream));
    HTRY(hipMemcpyAsync(out, f->stats.p, sizeof(out), hipMemcpyDeviceToHost, f->stream));
    HTRY(hipMemcpyAsync(ref, f->ml_dev.p + 8, sizeof(ref), hipMemcpyDeviceToHost, f->stream));
    HTRY(hipStreamSynchronize(f->stream));
    if (state) std::memcpy(state, ref, sizeof(ref));
  }
  if (cov) std::memcpy(cov, out + 4, 16 * sizeof(float));
  return TDR_OK;
}

// freezeScale (particle_filter.cpp:343-357)
int tdr_filter_freeze_scale(tdr_filter* f) {
  if (!f) return failh(TDR_ERR_ARG, "filter_freeze_scale: null filter");
  if (f->scale_frozen || f->n < 1) return TDR_OK;
  const float* gst = nullptr;
  int64_t gcap = 0;
  TTRY(filter_global_states(f, &gst, &gcap));
  TTRY(tdr_k_mean_cov(gst, gcap, f->n, nullptr, f->stats.p, f->stream));
  f->states_changed();
  TTRY(tdr_k_set_scale(f->st.p, f->cap, f->nl(), f->stats.p + 20, f->stream));
  float gm = 0;
  HTRY(hipMemcpy(&gm, f->stats.p + 20, sizeof(float), hipMemcpyDeviceToHost));
  f->scale_frozen = true;
  f->uniform_scale = gm;
  return TDR_OK;
}
int tdr_filter_is_scale_frozen(const tdr_filter* f) { return f && f->scale_frozen; }
// scale() (particle_filter.cpp:359-367)
float tdr_filter_scale(tdr_filter* f) {
  if (!f) return -1.f;
  if (f->fp.fixed_scale > 0) return f->fp.fixed_scale;
  if (f->scale_frozen && f->n > 0) {
    float s = -1.f;
    if (hipMemcpy(&s, f->st.p + (size_t)TDR_ST_SCALE * f->cap, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -1.f;
    return s;
  }
  return -1.f;
}
int64_t tdr_filter_num_particles(const tdr_filter* f) { return f ? f->n : 0; }

// computeGMM (particle_filter.cpp:252-318): <= 1000 strided samples {x, y, 50 cos theta, 50 sin theta} -> mixture
int tdr_filter_compute_gmm(tdr_filter* f) {
  if (!f) return failh(TDR_ERR_ARG, "filter_compute_gmm: null filter");
  if (f->n < 1) return TDR_OK;
  const int num = (int)std::min<int64_t>(1000, f->n);   // :262
  TTRY(f->gmm_samples.resize((size_t)3 * num));
  const float* gst = nullptr;
  int64_t gcap = 0;
  TTRY(filter_global_states(f, &gst, &gcap));
  TTRY(tdr_k_sample_ml_states(gst, gcap, f->n, num, f->gmm_samples.p, f->stream));
  std::vector<float> h((size_t)3 * num);
  HTRY(hipMemcpyAsync(h.data(), f->gmm_samples.p, h.size() * sizeof(float), hipMemcpyDeviceToHost, f->stream));
  HTRY(hipStreamSynchronize(f->stream));
  std::vector<double> x((size_t)4 * num);
  for (int i = 0; i < num; i++) {
    x[4 * i + 0] = h[3 * i + 0];
    x[4 * i + 1] = h[3 * i + 1];
    x[4 * i + 2] = 50 * std::cos(h[3 * i + 2]);   // :269-270 (float argument: the float overload)
    x[4 * i + 3] = 50 * std::sin(h[3 * i + 2]);
  }
  int k = f->num_gaussians;
  std::vector<float> means((size_t)3 * TDR_GMM_MAX_K), covs((size_t)9 * TDR_GMM_MAX_K);
  TTRY(tdr_gmm_select_host(x.data(), num, f->n, &k, TDR_GMM_MAX_K, means.data(), covs.data()));
  f->num_gaussians = k;
  f->gmm_means.assign(means.begin(), means.begin() + 3 * k);
  f->gmm_covs.assign(covs.begin(), covs.begin() + 9 * k);
  return TDR_OK;
}
int tdr_filter_get_gmm(tdr_filter* f, int max_k, int* k_out, float* means, float* covs) {
  if (!f || !k_out) return failh(TDR_ERR_ARG, "filter_get_gmm: bad arguments");
  const int k = (int)(f->gmm_means.size() / 3);
  *k_out = k;
  if (k > max_k) return failh(TDR_ERR_ARG, "filter_get_gmm: %d clusters, room for %d", k, max_k);
  if (means && k) std::memcpy(means, f->gmm_means.data(), f->gmm_means.size() * sizeof(float));
  if (covs && k) std::memcpy(covs, f->gmm_covs.data(), f->gmm_covs.size() * sizeof(float));
  return TDR_OK;
}
int64_t tdr_filter_adaptive_count(tdr_filter* f) {
  if (!f) return -1;
  const int k = (int)(f->gmm_means.size() / 3);
  if (k == 0) return f->n;
  return tdr_adaptive_count_host(f->gmm_covs.data(), k, f->n, f->n_max);
}

// ParticleFilter::updateMap(const cv::Mat& map, map_center) (particle_filter.cpp:320-341) for a class-index image
int tdr_filter_update_map_labels(tdr_filter* f, const uint8_t* label_img, int img_h, int img_w,
                                 const int32_t* flatten_lut, int lut_size, int ncls, float resolution, int center_x,
                                 int center_y) {
  if (!f || !f->map) return failh(TDR_ERR_ARG, "filter_update_map_labels: null filter");
  const int ox = f->map->center_x, oy = f->map->center_y;
  TTRY(tdr_map_set_labels(f->map, label_img, img_h, img_w, flatten_lut, lut_size, ncls, resolution, center_x, center_y));
  f->states_changed();
  if (f->n > 0) return tdr_k_shift_init(f->st.p, f->cap, f->nl(), (float)(center_x - ox), (float)(center_y - oy), f->stream);
  if (f->map->have_map) return tdr_filter_initialize_particles(f);  // :337-340
  return TDR_OK;
}

// ParticleFilter::updateMap (particle_filter.cpp:320-341), with the map already in distance-map form
int tdr_filter_update_map(tdr_filter* f, const float* class_maps, const uint8_t* class_mask, int ncls, int rows, int cols,
                          float resolution, int center_x, int center_y) {
  if (!f || !f->map) return failh(TDR_ERR_ARG, "filter_update_map: null filter");
  const int ox = f->map->center_x, oy = f->map->center_y;
  TTRY(tdr_map_set(f->map, class_maps, class_mask, ncls, rows, cols, resolution, center_x, center_y));
  f->states_changed();
  if (f->n > 0) return tdr_k_shift_init(f->st.p, f->cap, f->nl(), (float)(center_x - ox), (float)(center_y - oy), f->stream);
  return tdr_filter_initialize_particles(f);  // :337-340
}


// ---- device self-test of the scoring kernels -------------------------------------------------------------------------------
// The integer-form kernels run hand-scheduled, generated assembly (tdr_score_su_asm.h, tdr_score_cart_asm.h): a toolchain
// change must fail LOUDLY, not shift weights.  A tiny fixed problem, scored every way the library can score it:
//   polar      integer form, dense share through score_polar_su_kernel  ==  all particles through score_polar_ray_kernel
//              (bit for bit: the sums are exact), and both against the float kernel score_polar_kernel (rounding: 3e-6)
//   Cartesian  score_cart_su_kernel (generated loop)  ==  score_cart_skip_kernel<INT> (plain C++)  ==  score_cart_ray_kernel,
//              and against the float kernel
// Process-wide switches are set for the duration of the call and restored (one caller at a time, like the profile switch).
namespace {
struct SelftestRestore {
  int mode;
  float span;
  bool span_fixed_before;
  int64_t seg;
  SelftestRestore() : mode(tdr_config_shift_uniform(-1)), span(tdr_config_shift_uniform_span(-1.f)), seg(tdr_config_tuning("cart_seg_rows", -1)) {}
  ~SelftestRestore() {
    tdr_config_shift_uniform(mode);
    tdr_config_shift_uniform_span(-2.f);   // back to tuning (the default); a caller that had fixed a span sets it again
    tdr_config_tuning("cart_seg_rows", seg);
  }
};
}  // namespace
int tdr_selftest_score(void) {
  constexpr int NCLS = 6, SIZE = 160, NB = 64, NR = 32, CR = 32, CC = 24, N = 512;
  SelftestRestore restore;
  // a label image: bands of classes, a road grid (class 1), an unlabelled hole
  std::vector<uint8_t> lab((size_t)SIZE * SIZE);
  for (int y = 0; y < SIZE; y++)
    for (int x = 0; x < SIZE; x++) {
      int c = ((x / 13) + 2 * (y / 17)) % NCLS;
      if (c == 1) c = 2;
      if (x % 40 < 3 || y % 40 < 3) c = 1;
      if (x >= 100 && x < 120 && y >= 30 && y < 52) c = 255;   // unknown
      lab[(size_t)y * SIZE + x] = (uint8_t)c;
    }
  int32_t lut[256];
  for (int i = 0; i < 256; i++) lut[i] = i < NCLS ? i : -1;
  tdr_map* m = nullptr;
  TTRY(tdr_map_create(&m));
  struct MapGuard { tdr_map* m; ~MapGuard() { tdr_map_destroy(m); } } mg{m};
  TTRY(tdr_map_set_labels(m, lab.data(), SIZE, SIZE, lut, 256, NCLS, 1.f, 0, 0));
  TTRY(tdr_map_sample_pts_polar(m, NB, NR, (float)(2 * M_PI / NB)));
  // scans: integer counts, mostly one class per bin, some bins with two, many empty
  auto make_scan = [&](int rows, int cols, std::vector<float>& img) {
    img.assign((size_t)NCLS * rows * cols, 0.f);
    uint32_t h = 12345u;
    for (int k = 0; k < rows * cols; k++) {
      h = h * 1664525u + 1013904223u;
      const uint32_t r = h >> 8;
      if (r % 100 < 55) continue;
      const int c = (int)((r >> 7) % NCLS);
      img[(size_t)c * rows * cols + k] = (float)(1 + (r >> 11) % 5);
      if (r % 100 > 96) img[(size_t)((c + 2) % NCLS) * rows * cols + k] = (float)(1 + (r >> 15) % 3);
    }
  };
  std::vector<float> scan_p, scan_c;
  make_scan(NB, NR, scan_p);
  make_scan(CR, CC, scan_c);
  // particles: a cluster inside the map (different headings), a few at the border and outside
  std::vector<tdr_state> st(N);
  {
    uint32_t h = 777u;
    auto u01 = [&]() { h = h * 1664525u + 1013904223u; return (float)(h >> 8) / 16777216.f; };
    for (int p = 0; p < N; p++) {
      tdr_state s{};
      s.init_x_px = 70.f + 14.f * (u01() - 0.5f);
      s.init_y_px = 85.f + 14.f * (u01() - 0.5f);
      s.theta = 0.4f + 0.5f * (u01() - 0.5f);
      s.scale = 1.f;
      s.have_init = 1;
      if (p % 61 == 0) { s.init_x_px = 2.f + 150.f * u01(); s.init_y_px = (p % 2) ? 1.f : 158.f; }
      if (p == 3) { s.init_x_px = -40.f; s.init_y_px = 80.f; }
      st[(size_t)p] = s;
    }
  }
  tdr_filter_params fp{};
  fp.pos_cov = 0.3f; fp.theta_cov = 0.03f; fp.regularization = 0.15f;
  fp.fixed_scale = 1.f; fp.scale_log_min = -0.1f; fp.scale_log_max = 1.f;
  fp.num_classes = NCLS;
  for (int c = 0; c < 16; c++) fp.class_weights[c] = c < NCLS ? 1.f : 0.f;
  tdr_filter* f = nullptr;
  TTRY(tdr_filter_create(m, N, &fp, 1, &f));
  struct FilterGuard { tdr_filter* f; ~FilterGuard() { tdr_filter_destroy(f); } } fg{f};
  TTRY(tdr_filter_set_states(f, st.data(), N));
  std::vector<float> got[6];
  auto rel_ok = [](const std::vector<float>& a, const std::vector<float>& b, double tol, double* worst) {
    *worst = 0;
    for (size_t i = 0; i < a.size(); i++) {
      if (std::isnan(a[i]) != std::isnan(b[i])) return false;
      if (std::isnan(a[i])) continue;
      const double d = std::fabs((double)a[i] - (double)b[i]) / std::max(std::fabs((double)b[i]), 1e-30);
      *worst = std::max(*worst, d);
    }
    return *worst <= tol;
  };
  auto same_bits = [](const std::vector<float>& a, const std::vector<float>& b) {
    return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0;
  };
  // ---- polar: float kernel, integer form (dense share), integer form with every particle ray-mapped
  const struct { int mode; float span; } polar_runs[3] = {{0, 16.f}, {2, 16.f}, {2, 1e-6f}};
  for (int r = 0; r < 3; r++) {
    tdr_config_shift_uniform(polar_runs[r].mode);
    tdr_config_shift_uniform_span(polar_runs[r].span);
    TTRY(tdr_filter_compute_weights(f, scan_p.data(), nullptr, 1.f));
    got[r].resize(N);
    TTRY(tdr_filter_get_raw_weights(f, got[r].data(), N));
  }
  double worst = 0;
  if (!same_bits(got[1], got[2]))
    return failh(TDR_ERR_HIP, "selftest: score_polar_su_kernel and score_polar_ray_kernel disagree (exact integer sums must "
                              "be identical): the generated loop does not survive this toolchain");
  if (!rel_ok(got[1], got[0], 3e-6, &worst))
    return failh(TDR_ERR_HIP, "selftest: the polar integer form is %.3g away from the float kernel", worst);
  int finite = 0;
  for (float w : got[1]) finite += std::isfinite(w) && w > 0.f;
  if (finite < N / 2) return failh(TDR_ERR_HIP, "selftest: only %d of %d polar weights are finite", finite, N);
  // ---- Cartesian: float kernel, plain integer kernel, generated loop (two segment lengths), ray-mapped
  {
    const int rf = tdr_rec_floats(NCLS);
    DevBuf<float> img, pk, raw, ws;
    TTRY(img.resize((size_t)NCLS * CR * CC));
    TTRY(pk.resize((size_t)CR * CC * rf));
    TTRY(raw.resize(N));
    TTRY(ws.resize(tdr_score_cart_workspace_floats(NCLS, CR, CC, N, N)));
    HTRY(hipMemcpyAsync(img.p, scan_c.data(), scan_c.size() * sizeof(float), hipMemcpyHostToDevice, f->stream));
    TTRY(tdr_k_pack_scan(img.p, NCLS, CR, CC, pk.p, f->stream));
    const struct { int mode; float span; int seg; } cart_runs[5] = {{0, 16.f, 32}, {2, 0.f, 0}, {2, 0.f, 8}, {2, 0.f, 32}, {2, 1e-6f, 32}};
    std::vector<float> c[5];
    for (int r = 0; r < 5; r++) {
      tdr_config_shift_uniform(cart_runs[r].mode);
      tdr_config_shift_uniform_span(cart_runs[r].span);
      tdr_config_tuning("cart_seg_rows", cart_runs[r].seg);
      TTRY(tdr_k_score_cart(&m->desc, pk.p, CR, CC, 0.75f, &fp, f->st.p, f->cap, N, N, nullptr, raw.p, ws.p, f->stream));
      c[r].resize(N);
      HTRY(hipMemcpyAsync(c[r].data(), raw.p, N * sizeof(float), hipMemcpyDeviceToHost, f->stream));
      HTRY(hipStreamSynchronize(f->stream));
    }
    if (!same_bits(c[1], c[2]) || !same_bits(c[1], c[3]))
      return failh(TDR_ERR_HIP, "selftest: score_cart_su_kernel (generated loop) and score_cart_skip_kernel disagree (exact "
                                "integer sums must be identical): the generated loop does not survive this toolchain");
    if (!same_bits(c[1], c[4])) return failh(TDR_ERR_HIP, "selftest: score_cart_ray_kernel disagrees with the dense kernels");
    if (!rel_ok(c[1], c[0], 3e-6, &worst))
      return failh(TDR_ERR_HIP, "selftest: the Cartesian integer form is %.3g away from the float kernel", worst);
    finite = 0;
    for (float w : c[1]) finite += std::isfinite(w) && w > 0.f;
    if (finite < N / 2) return failh(TDR_ERR_HIP, "selftest: only %d of %d Cartesian weights are finite", finite, N);
  }
  return TDR_OK;
}

}  // extern "C"
