/* tdr.h — C ABI of libtdr_hip.so: the MI355X (gfx950) implementation of top_down_renderer's per-scan
 * particle-filter update (scan raster -> per-particle map gather + class-wise score -> weight statistics ->
 * systematic resample, plus propagate).
 *
 * The reference (KumarRobotics/top_down_renderer) has no FFI layer: its boundary is the public C++ surface of
 * library target `top_down_render` (CMakeLists.txt:145-165).  Every entry point below names the reference method
 * (file:line, relative to the reference root) whose work it performs; the headers under include/top_down_render/ rebuild the
 * reference's class names on top of these calls.
 *
 * Conventions
 *  - plain C, no torch / Eigen / PCL types; every function returns an int status (TDR_OK == 0), never throws;
 *  - "tdr_k_*" entry points are stateless launchers: every pointer is a DEVICE pointer unless the name says
 *    `_host`, `stream` is a hipStream_t (NULL = default stream), nothing is allocated, nothing synchronises, and no
 *    state is kept between calls.  What a launcher may be GIVEN is an object the caller owns (tdr_score_ctx: the tuner of
 *    the polar launch; tdr_rng_pipe: the generator's state and what it drew ahead, with a stream of its own).  Load-time entry points (tdr_k_compact_map,
 *    tdr_k_map_from_*) say so where they synchronise.  The tdr_config_* calls and the TDR_* environment variables they
 *    mirror are PROCESS-WIDE switches for A/B measurements and tests (set them before the first launch, not while
 *    another thread launches); results never depend on them unless a comment says so;
 *  - images follow the reference's Eigen::ArrayXXf layout: column-major float32, element (i,j) at i + rows*j,
 *    one image per class, images of one scan contiguous: [ncls][rows*cols];
 *  - particle state on the device is a structure of arrays `float st[TDR_ST_FIELDS][cap]` (plane stride `cap`),
 *    planes in the field order of the reference's `State` (include/top_down_render/state_particle.h:9-17),
 *    have_init stored as 0.0f / 1.0f;
 *  - no CPU fallback exists: without a HIP device every launcher fails with TDR_ERR_HIP.
 */
#ifndef TDR_H_
#define TDR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDR_OK 0
#define TDR_ERR_ARG (-1)      /* bad shape / null pointer / unsupported size (message in tdr_last_error) */
#define TDR_ERR_HIP (-2)      /* HIP runtime error (message in tdr_last_error) */
#define TDR_ERR_NOMEM (-3)

#define TDR_MAX_CLASSES 15
#define TDR_ST_FIELDS 7       /* init_x_px, init_y_px, dx_m, dy_m, theta, scale, have_init */
enum { TDR_ST_INIT_X = 0, TDR_ST_INIT_Y = 1, TDR_ST_DX = 2, TDR_ST_DY = 3, TDR_ST_THETA = 4, TDR_ST_SCALE = 5,
       TDR_ST_HAVE_INIT = 6 };

/* Same bytes as the reference's `State` (state_particle.h:9-17): 6 floats + bool, sizeof == 28. */
typedef struct tdr_state {
  float init_x_px, init_y_px, dx_m, dy_m, theta, scale;
  uint8_t have_init;
  uint8_t pad_[3];
} tdr_state;

/* POD mirror of the reference's `FilterParams` (state_particle.h:19-38). */
typedef struct tdr_filter_params {
  float pos_cov, theta_cov, regularization;
  float init_pos_px_x, init_pos_px_y, init_pos_px_cov;
  float init_pos_m_x, init_pos_m_y, init_pos_deg_theta, init_pos_deg_cov;
  int32_t force_on_map;
  float fixed_scale, scale_log_min, scale_log_max;
  int32_t num_classes;
  float class_weights[16];
} tdr_filter_params;

/* Geometry of the device-resident map produced by tdr_k_pack_map. */
typedef struct tdr_map_desc {
  const float* rec;     /* [(rows+2)*(cols+2)][rec_floats], row-major over the map plus a one-cell guard ring of zero
                           records; per cell {dist_0..dist_{ncls-1}, 0.., [known,] known} (see tdr_k_pack_map) */
  int32_t ncls, rows, cols, rec_floats;   /* rec_floats = 4*ceil((ncls+1)/4) */
  float resolution;     /* TopDownMap::Params::resolution (top_down_map.h:61) */
  /* Optional compact form of the same records (tdr_k_compact_map; cwords == 0: absent).  A cell is cwords dwords of
   * 10-bit indices into `dict` (three per dword, class k in dword k/3 at bit 2 + 10*(k%3)), `known` in bit 0 of every
   * dword; records are tiled 32/(4*cwords) rows x 4 columns per 128-byte line, row-major inside a tile, the tiles column
   * by column over the map (csrc/tdr_cmap.hip); the map's known mask follows the tiles (tdr_cmap_words_total).  Decoding reproduces `rec` bit for bit;
   * the scoring kernels read it instead of `rec` whenever it is present (tdr_config_compact(0) forces `rec`).
   * The WIDE form (tdr_k_compact_map_wide: rec_floats == 8, cwords == 4, dict_n > TDR_CMAP_MAX_DICT) holds 16-bit
   * indices, two per dword (class k in dword k/2 at bit 2 + 16*(k%2)), into a dictionary of up to
   * TDR_CMAP_WIDE_MAX_DICT values. */
  int32_t cwords, dict_n;
  const uint32_t* crec;
  const float* dict;    /* [TDR_CMAP_WIDE_MAX_DICT], entry 0 = +0.0f */
  /* Optional SCRATCH of tdr_map_rec16_bytes(ncls, rows, cols) bytes of device memory (NULL: none).  The 40-rotation
   * search of tdr_k_score_polar(init_search != 0) writes the map's records there as pre-split f16 pairs with the
   * filter's class weights folded in and gathers those (twice as fast as splitting the f32 records per sample; same
   * bits).  Its contents mean nothing between calls; two searches that share one map on different streams need a
   * scratch each. */
  void* rec16;
} tdr_map_desc;

const char* tdr_last_error(void);
int tdr_version(void);
int tdr_device_count(void);
/* floats per map / scan record for `ncls` classes */
int tdr_rec_floats(int ncls);

/* ---- map (storage of TopDownMap, include/top_down_render/top_down_map.h:77-79) ------------------------------ */
/* Interleaves the reference's per-class column-major distance maps `class_maps_` ([ncls][rows*cols], element (r,c)
 * at r + rows*c) and the unknown mask `class_mask_` (u8, 1 = unknown) into cell records (tdr_map_desc.rec).
 * rec_out must hold tdr_map_rec_floats_total(ncls, rows, cols) floats. */
size_t tdr_map_rec_floats_total(int ncls, int rows, int cols);
/* bytes behind tdr_map_desc.rec16; 0 when the class count has no matrix-core search (more than 7 classes) */
size_t tdr_map_rec16_bytes(int ncls, int rows, int cols);
/* Filters of fewer particles than this (n_total of tdr_k_score_polar) ignore rec16 (rebuilding it costs one pass over
 * the map); default 8192.  n >= 0 sets the threshold, n < 0 only returns it. */
int64_t tdr_config_rec16_min_particles(int64_t n);
int tdr_k_pack_map(const float* class_maps, const uint8_t* class_mask, int ncls, int rows, int cols, float* rec_out,
                   void* stream);

/* Compact form of the cell records (csrc/tdr_cmap.hip).  Fills map->crec / dict / dict_n / cwords from map->rec;
 * crec_out: tdr_cmap_words_total(ncls, rows, cols) dwords, dict_out: TDR_CMAP_WIDE_MAX_DICT floats, workspace:
 * TDR_CMAP_WORKSPACE_BYTES, all device memory.  Load-time work: synchronises with `stream`.  Maps with more than
 * TDR_CMAP_MAX_DICT distinct distance values or more than 11 classes have no (narrow) compact form: cwords stays 0,
 * TDR_OK — and dict_n = -(distinct values) when the WIDE form exists for the map (4-7 classes, up to
 * TDR_CMAP_WIDE_MAX_DICT values: 16 bytes per cell instead of 8): tdr_k_compact_map_wide then builds it into
 * tdr_cmap_wide_words_total(ncls, rows, cols) dwords.
 * tdr_k_unpack_compact_map decodes either form back into dense records (rec_out like tdr_k_pack_map's output). */
#define TDR_CMAP_MAX_DICT 1024        /* dictionary entries of the narrow form (10-bit fields) */
#define TDR_CMAP_WIDE_MAX_DICT 4096   /* ... of the wide form (16-bit fields; tdr_k_compact_map_wide) */
#define TDR_CMAP_WORKSPACE_BYTES (16384 * 4 + 16384 * 2 + 256)
int tdr_cmap_words(int ncls);
/* dwords behind crec, three parts:
 *  - the record tiles (tdr_cmap_tile_words);
 *  - the map's KNOWN MASK, one bit per cell in 32 x 32-cell tiles of 32 words (one 128-byte line), the tiles stored tile
 *    column by tile column with a guard band of 32 unknown cells around the map: with r' = r + 32, c' = c + 32 cell (r, c)
 *    is bit c & 31 of word (c' >> 5) * 32 * ((rows >> 5) + 2) + r' (csrc/tdr_score_dev.h: kmask_offset);
 *  - from dword tdr_cmap_plane_offset_words on, the CLASS PLANES (tdr_cmap_plane_words dwords each, 0 = none: the map is
 *    too large for 32-bit offsets): per class one 16-bit value per cell — dictionary index * 4 in bits 2-11, known in bit
 *    15 — in tiles of 8 x 8 cells, tile column by tile column, a guard band of 8 cells (csrc/tdr_score_dev.h:
 *    plane_offset); behind them the COARSE MASK PLANE (tdr_cmap_cmask_words dwords), the same shape and tile-column stride
 *    with a 16-bit cell holding the known bits of a 4 x 4 block of map cells: cell (r >> 2, c >> 2), bit (r & 3) * 4 + (c & 3).
 * Behind the float dictionary, `dict` also carries the dictionary as integers (narrow form): entries [1024, 2048) =
 * value * 2^q as uint32, [2048] = q, [2049] = 1 when every value has that form (csrc/tdr_cmap.hip). */
size_t tdr_cmap_words_total(int ncls, int rows, int cols);
size_t tdr_cmap_tile_words(int ncls, int rows, int cols);
size_t tdr_cmap_plane_offset_words(int ncls, int rows, int cols);
size_t tdr_cmap_plane_words(int ncls, int rows, int cols);
size_t tdr_cmap_cmask_words(int ncls, int rows, int cols);
int tdr_k_compact_map(tdr_map_desc* map, uint32_t* crec_out, float* dict_out, void* workspace, void* stream);
size_t tdr_cmap_wide_words_total(int ncls, int rows, int cols);   /* 0: no wide form for this class count */
int tdr_k_compact_map_wide(tdr_map_desc* map, uint32_t* wrec_out, float* dict_out, void* workspace, void* stream);
int tdr_k_unpack_compact_map(const tdr_map_desc* map, float* rec_out, void* stream);

/* Map ingest on the device: TopDownMap::loadCompressedRasterMap (src/top_down_map.cpp:116-144) +
 * computeDists (:289-326) for a class-index image, the work of TopDownMap::updateMap (:146-157) when a new aerial
 * map arrives.  label_img: DEVICE image, img_h x img_w, row-major u8 (cv::Mat CV_8UC1); flatten_lut: DEVICE int32
 * [lut_size] raw label -> flattened class (params_.flatten_lut).  Writes the cell records of the
 * rows = int(img_h/resolution), cols = int(img_w/resolution) map (tdr_map_ingest_shape) into rec_out
 * (tdr_map_rec_floats_total floats).  The Euclidean distance transform is exact. */
size_t tdr_map_ingest_workspace_bytes(int ncls, int rows, int cols);
int tdr_map_ingest_shape(int img_h, int img_w, float resolution, int* rows, int* cols);
/* The same from the per-class rasters of the raster cache: planes [ncls][rows][cols] device bytes = the class<i>.png
 * images as stored (8-bit grey, row 0 = top; src/top_down_map.cpp:213-224 flips and scales them, computeDists :289-326
 * binarises: p <= 127 = inside the class, a cell is unknown where every class holds 255).  Classes may overlap.  The map
 * has the images' shape; rec_out / workspace as for tdr_k_map_from_labels (tdr_map_ingest_workspace_bytes). */
int tdr_k_map_from_rasters(const uint8_t* planes, int ncls, int rows, int cols, float resolution, float* rec_out,
                           void* workspace, void* stream);
int tdr_k_map_from_labels(const uint8_t* label_img, int img_h, int img_w, const int32_t* flatten_lut, int lut_size,
                          int ncls, float resolution, float* rec_out, void* workspace, void* stream);
/* geo_maps_ (top_down_map.h:79) as a 2-class record map {d_without, d_with, 1, 1}, derived from the class records
 * (getGeoRasterMap top_down_map.cpp:410-427 + computeDists :58; exact distance transform), or — constant_one != 0 — the
 * constant 1 of the dynamic-map path (:126-133).  geo_rec_out: tdr_map_rec_floats_total(2, rows, cols) floats;
 * workspace: tdr_map_ingest_workspace_bytes(2, rows, cols) bytes (may be NULL with constant_one). */
int tdr_k_geo_map_from_map(const tdr_map_desc* map, int constant_one, float* geo_rec_out, void* workspace, void* stream);
/* Cell records back to the reference's layout: class_maps_out [ncls][rows*cols] column-major, class_mask_out u8. */
int tdr_k_unpack_map(const float* rec, int ncls, int rows, int cols, float* class_maps_out, uint8_t* class_mask_out,
                     void* stream);

/* TopDownMapPolar::samplePtsPolar (src/top_down_map_polar.cpp:7-19, via TopDownMap::samplePts
 * src/top_down_map.cpp:367-389): fills the HOST table tab[nb*nr][2] = {cos(theta_i)*r_j, sin(theta_i)*r_j}. */
int tdr_polar_table_host(int nb, int nr, float ang_res, float resolution, float* tab_out);
/* The same table as its two factors, fac_out[2 nb + nr] (HOST): {cos, sin}(theta_i) at [2 i], [2 i + 1], r_j at
 * [2 nb + j] — every table entry is one float product of the two (top_down_map_polar.cpp:17-18).  A DEVICE copy handed
 * to a tdr_score_ctx (tdr_score_ctx_set_polar_factors) lets the ray-mapped kernel multiply the offsets itself instead of
 * reading the table; it checks on the device, every call, that the table it was given is those products, and reads the
 * table when it is not. */
int tdr_polar_factors_host(int nb, int nr, float ang_res, float resolution, float* fac_out);

/* ---- scan raster -------------------------------------------------------------------------------------------- */
/* ScanRendererPolar::renderSemanticTopDown (src/scan_renderer_polar.cpp:83-109).
 * pts: n points, `stride` floats apart (pcl::PointXYZI: stride 8, ioff 4; packed xyzi: stride 4, ioff 3);
 * lut256: flatten_lut_ (int32[256], -1 = ignore).  Outputs (either may be NULL):
 *   img_out  [ncls][nb*nr]   the reference's images (zero-filled, counts as float);
 *   pk_out   [nr][nb][rf]    the same counts interleaved per bin, slot rf-1 = sum over classes (scoring input).
 * workspace: device scratch of tdr_raster_workspace_bytes(n) bytes, or NULL.  With it every point's bin is computed
 * once (one atan2f / sqrtf per point) and the LDS tiles stream 4-byte keys; without it every tile bins every point. */
int64_t tdr_raster_workspace_bytes(int64_t n);
int tdr_k_raster_polar(const float* pts, int stride, int ioff, int64_t n, float res, float ang_res,
                       const int32_t* lut256, int ncls, int nb, int nr, float* img_out, float* pk_out, void* workspace,
                       void* stream);
/* ScanRenderer::renderSemanticTopDown (src/scan_renderer.cpp:55-78); img_out [ncls][rows*cols]. */
int tdr_k_raster_cart(const float* pts, int stride, int ioff, int64_t n, float res, const int32_t* lut256, int ncls,
                      int rows, int cols, float* img_out, float* pk_out, void* workspace, void* stream);
/* renderGeometricTopDown: the ground / obstacle images (dead at the reference's call site,
 * src/top_down_render.cpp:540, but part of the class surface).  pts: the ORGANISED cloud, element idy*width + idx
 * (cloud->at(idx, idy)), `stride` floats apart, x y z first; unorganised clouds have height 1.
 *   polar     (src/scan_renderer_polar.cpp:6-81): img_out [2][nb*nr], [0] ground, [1] obstacles; per theta bin the
 *             returns are walked by range descending — equal ranges in input order, the tie rule where the reference's
 *             std::sort leaves the order open; workspace of tdr_raster_geo_workspace_bytes(width*height) bytes;
 *   Cartesian (src/scan_renderer.cpp:7-53): img_out [2][rows*cols]; every column idx is one scan line walked upwards.
 * Counts are exact.  Points with a non-finite x or y are dropped (the reference indexes with (int)NaN there). */
int64_t tdr_raster_geo_workspace_bytes(int64_t n);
int tdr_k_raster_geo_polar(const float* pts, int stride, int64_t width, int64_t height, float res, float ang_res, int nb,
                           int nr, float* img_out, void* workspace, void* stream);
int tdr_k_raster_geo_cart(const float* pts, int stride, int64_t width, int64_t height, float res, int rows, int cols,
                          float* img_out, void* stream);
/* Builds pk_out from caller-supplied images (ParticleFilter::update is handed images, particle_filter.cpp:94-95). */
int tdr_k_pack_scan(const float* img, int ncls, int nb, int nr, float* pk_out, void* stream);

/* ---- scoring: StateParticle::computeWeight for all particles (src/state_particle.cpp:157-219 =
 *      TopDownMapPolar::getLocalMap src/top_down_map_polar.cpp:21-53 + getCostForRot src/state_particle.cpp:112-155,
 *      driven by ParticleFilter::update src/particle_filter.cpp:104-105) ------------------------------------- */
/* workspace floats needed by tdr_k_score_polar for n particles */
size_t tdr_score_workspace_floats(int ncls, int nb, int nr, int64_t n, int64_t n_total);
/* Scores particles [0,n) of st (plane stride cap) against the packed scan at each particle's own theta; writes raw
 * weights raw_w[n] (NaN = "too much unknown", 0 = gated by force_on_map / scale range, state_particle.cpp:163-176).
 * perm (optional, int32[n]): processing order (a permutation of 0..n-1, e.g. from tdr_k_locality_order) for cache
 * locality; results are independent of it.  tab: DEVICE copy of the table from tdr_polar_table_host.
 * uniform_scale: > 0 is the caller's promise that every particle's scale equals it (fixed or frozen scale,
 * particle_filter.cpp:24,343-357): the sample offsets (tab*scale)*res are then evaluated once per call instead of
 * once per particle and sample — same float operations, same results; <= 0 reads each particle's own scale.
 * init_search != 0: particles whose have_init is 0 (and that are not gated) first run the 40-rotation search of
 * state_particle.cpp:195-206 — one pass over their window scoring all candidate rotations — which sets their theta
 * and have_init in st; they are then scored at that rotation like everyone else (all-NaN searches keep the
 * reference's 1/(FLT_MAX + regularization)).  Batches without such particles exit immediately; nothing synchronises
 * with the host.  Pass 0 when the caller knows every particle is initialised.
 * n_total: the particle count of the WHOLE filter (all ranks of a sharded filter; <= 0 means n).  The kernel splits a
 * particle's score into partial sums over groups of range rings, sized from the image shape and n_total — a small filter
 * gets many short workgroups — so a shard scored on its own gives the same bits as the same particles inside the
 * one-rank filter. */
int tdr_k_score_polar(const tdr_map_desc* map, const float* tab, const float* scan_pk, int nb, int nr, float res,
                      const tdr_filter_params* fp, float* st, int64_t cap, int64_t n, int64_t n_total,
                      const int32_t* perm, float uniform_scale, int init_search, float* raw_w, float* workspace,
                      void* stream);
/* The same with a caller-owned context.  A large launch runs two kernels one after the other on `stream` — dense particles
 * through the shift-uniform kernel, scattered ones through the ray-mapped kernel (tdr_config_shift_uniform below) — and
 * chooses the split between them by timing; the context keeps the tuner's state from call to call.  One context per filter
 * (or per caller thread); calls that share a context must not overlap.  ctx == NULL: tdr_k_score_polar — the configured
 * span, no state.  Results never depend on the context.  Nothing waits on the host either way (the tuner polls its events).
 * tdr_score_ctx_span: the split the context's tuner has settled on so far (map cells). */
typedef struct tdr_score_ctx tdr_score_ctx;
int tdr_score_ctx_create(tdr_score_ctx** out);
void tdr_score_ctx_destroy(tdr_score_ctx* ctx);
float tdr_score_ctx_span(const tdr_score_ctx* ctx);
/* launches of this context that ran at a TRIAL span so far (a benchmark reports how many fell into its timed region) */
int64_t tdr_score_ctx_trial_calls(const tdr_score_ctx* ctx);
/* fac_dev: device copy of tdr_polar_factors_host's output for the (nb, nr) table the context's calls score with; the
 * caller's memory, alive until replaced (NULL: none).  Calls with another shape ignore it. */
int tdr_score_ctx_set_polar_factors(tdr_score_ctx* ctx, const float* fac_dev, int nb, int nr);
int tdr_k_score_polar_ctx(const tdr_map_desc* map, const float* tab, const float* scan_pk, int nb, int nr, float res,
                          const tdr_filter_params* fp, float* st, int64_t cap, int64_t n, int64_t n_total,
                          const int32_t* perm, float uniform_scale, int init_search, float* raw_w, float* workspace,
                          tdr_score_ctx* ctx, void* stream);

/* Scoring WITH the geometric term of getCostForRot (src/state_particle.cpp:145-152: commented out in the reference, whose
 * geometric images are all-zero anyway; opt-in here): cost += (geo_i . shifted geo_cls_i).sum() * 0.01 and
 * normalization += geo_i.sum() for the two geometric layers.  geo_map: the 2-layer map of tdr_k_geo_map_from_map;
 * geo_pk: the two geometric scan images packed by tdr_k_pack_scan(img, 2, nb, nr); geo_sum0/1: the sums of those images.
 * Everything else as tdr_k_score_polar; the init search prices the geometric term for every candidate rotation (one
 * scoring pass per rotation over the batches that hold an un-initialised particle).
 * workspace: tdr_score_geo_workspace_floats. */
size_t tdr_score_geo_workspace_floats(int ncls, int nb, int nr, int64_t n, int64_t n_total);
int tdr_k_score_polar_geo(const tdr_map_desc* map, const tdr_map_desc* geo_map, const float* tab, const float* scan_pk,
                          const float* geo_pk, float geo_sum0, float geo_sum1, int nb, int nr, float res,
                          const tdr_filter_params* fp, float* st, int64_t cap, int64_t n, int64_t n_total,
                          const int32_t* perm, float uniform_scale, int init_search, float* raw_w, float* workspace,
                          void* stream);

/* Cartesian scoring (BASELINE config 4).  The reference's StateParticle only reaches the polar overloads
 * (state_particle.h:61), so there is no reference function to match; the score is DEFINED as the window of
 * TopDownMap::getLocalMap(center, rot = theta, res = res*scale) (src/top_down_map.cpp:429-459) scored by
 * getCostForRot with shift 0 (src/state_particle.cpp:132-143), weight = 1/(cost + regularization), no gates.
 * scan_pk: packed [cols][rows][rf] Cartesian render (tdr_k_raster_cart / tdr_k_pack_scan with nb=rows, nr=cols).
 * n_total: particle count of the whole (possibly sharded) filter, as in tdr_k_score_polar (0 = n): it fixes the split
 * of the window into partial sums, so that a shard scores its particles to the same bits as the one-rank filter. */
size_t tdr_score_cart_workspace_floats(int ncls, int rows, int cols, int64_t n, int64_t n_total);
int tdr_k_score_cart(const tdr_map_desc* map, const float* scan_pk, int rows, int cols, float res,
                     const tdr_filter_params* fp, float* st, int64_t cap, int64_t n, int64_t n_total,
                     const int32_t* perm, float* raw_w, float* workspace, void* stream);

/* ---- StateParticle::propagate for all particles (src/state_particle.cpp:57-78 via particle_filter.cpp:86-92) -- */
/* z4: optional DEVICE array [n][4] of standard normals {theta, dx, dy, scale} in the reference's consumption
 * order (parity mode, from tdr_propagate_normals_host); NULL = counter-based device RNG keyed by (seed, step). */
int tdr_k_propagate(float* st, int64_t cap, int64_t n, float* last_dist, float tx, float ty, float omega,
                    int scale_freeze, float pos_cov, float theta_cov, const float* z4, uint64_t seed, uint64_t step,
                    int64_t index_base, void* stream);
/* The shared std::mt19937 of the reference (particle_filter.h:52): host-side handle, explicit seed. */
void* tdr_rng_create(uint32_t seed);
void tdr_rng_destroy(void* rng);
float tdr_rng_uniform_host(void* rng);                      /* particle_filter.cpp:172-173 */
int tdr_propagate_normals_host(void* rng, int64_t n, int scale_freeze, float* z4_out);
/* The same stream on the DEVICE (csrc/tdr_rng.hip).  `state`: TDR_RNG_STATE_WORDS device words — a std::mt19937 in
 * libstdc++'s own representation: [0, 624) the engine's state array, [624] the index of the next word (624: a twist comes
 * first), [625] an error flag the kernels raise if a call runs out of its attempt budget (probability ~1e-23; then the state
 * is left alone and tdr_rng_set_state_host refuses it).  tdr_rng_get_state_host / _set_state_host move a host engine's state
 * into and out of that form (HOST arrays), so either side can continue the other's stream.
 * tdr_k_rng_propagate_normals: z4_out (device) [hi - lo][4] = the standard normals {theta, dx, dy, scale} of particles
 * [lo, hi) of a propagate call over n particles — bit for bit what tdr_propagate_normals_host draws for them — and the
 * state moves on by the words the WHOLE call consumes (every rank of a sharded filter passes its own [lo, hi) and keeps
 * the same state).  Marsaglia attempts are independent given the words, so only the generator's state recurrence is
 * serial (one wave); acceptance, ranking and the normals themselves are data-parallel.  workspace: device scratch of
 * tdr_rng_dev_workspace_bytes(n) bytes.  tdr_k_rng_uniform: *out_dev = std::uniform_real_distribution<float>(0, 1)(gen),
 * the draw of the systematic resample (src/particle_filter.cpp:172-173).  Nothing synchronises. */
#define TDR_RNG_STATE_WORDS 640
size_t tdr_rng_dev_workspace_bytes(int64_t n);
int tdr_k_rng_propagate_normals(uint32_t* state, int64_t n, int64_t lo, int64_t hi, int scale_freeze, float* z4_out,
                                void* workspace, void* stream);
int tdr_k_rng_uniform(uint32_t* state, float* out_dev, void* stream);
int tdr_rng_get_state_host(void* rng, uint32_t* words);
int tdr_rng_set_state_host(void* rng, const uint32_t* words);
/* tdr_rng_pipe: the generator of ONE filter on the device, drawing ahead (csrc/tdr_rng.hip).  A filter's step draws in a
 * fixed order — propagate's normals, the resample's uniform, the next step's normals — and none of it depends on the
 * particles: when a propagate call has been served, the pipe draws what the step after it will most likely ask for on a
 * stream of its own, beside the scoring launch.  A call that asks for something else (another particle count, another
 * order, the host engine taking the stream back) makes it drop what it drew ahead and continue from the state the stream
 * really is in: the values handed out are the stream's own in every case.  One pipe per filter, one caller thread; n_max:
 * the largest particle count of a call.  from_host / to_host move the stream between a host std::mt19937 and the device
 * (they synchronise `stream`); normals / uniform are ordered on `stream` and wait for nothing on the host; the device
 * pointers they return stay valid until the next call of the same function. */
typedef struct tdr_rng_pipe tdr_rng_pipe;
int tdr_rng_pipe_create(int64_t n_max, tdr_rng_pipe** out);
void tdr_rng_pipe_destroy(tdr_rng_pipe* p);
int tdr_rng_pipe_on_device(const tdr_rng_pipe* p);
int tdr_rng_pipe_from_host(tdr_rng_pipe* p, void* host_rng, void* stream);
int tdr_rng_pipe_to_host(tdr_rng_pipe* p, void* host_rng, void* stream);
int tdr_rng_pipe_normals(tdr_rng_pipe* p, int64_t n, int64_t lo, int64_t hi, int scale_freeze, const float** z4_out,
                         void* stream);
int tdr_rng_pipe_uniform(tdr_rng_pipe* p, const float** shift_out, void* stream);
/* ParticleFilter::initializeParticles particle loop (src/particle_filter.cpp:57-71) with the StateParticle
 * constructor (src/state_particle.cpp:3-49): host-side, serial mt19937 draws with on-road rejection.
 * class_maps: HOST copy of class_maps_ (column-major [ncls][rows*cols]); out must hold max_num+16 states. */
/* One particle drawn like StateParticle's constructor (src/state_particle.cpp:3-49). */
int tdr_init_particle_host(void* rng, const float* class_maps, int ncls, int rows, int cols, float resolution,
                           const tdr_filter_params* fp, tdr_state* out);
int tdr_init_particles_host(void* rng, const float* class_maps, int ncls, int rows, int cols, float resolution,
                            const tdr_filter_params* fp, int max_num, tdr_state* out, int64_t* n_out);

/* ---- ParticleFilter::update, weight statistics (src/particle_filter.cpp:107-147) ---------------------------- */
/* raw_w, last_dist: [n] -> w_out [n] final normalised weights; info_out (device, TDR_UW_INFO_FLOATS floats: the first 8
 * are {argmax (as int bits), sum, mean, bottom_stddev, fallback, num_valid, num_under, 0}, the rest is scratch for
 * the multi-workgroup reductions and the chunk headers of the exact chains).  The result is a pure function of
 * (raw_w, last_dist, n).  `sum`, `mean` and `bottom_stddev` are the reference's serial float32 accumulations bit for
 * bit at every n (csrc/tdr_prefix.hip: uw_small_kernel up to 32768 particles, tdr_chain_total above).  n is limited
 * to what the scratch holds (~7 million). */
#define TDR_UW_INFO_FLOATS 65536
int tdr_k_update_weights(const float* raw_w, const float* last_dist, int64_t n, float* w_out, float* info_out,
                         void* stream);

/* ---- systematic resample (src/particle_filter.cpp:171-185) -------------------------------------------------- */
/* Serial-order float32 running sum of w (the additions of :179 in the same order) and its running maximum.
 * workspace: device scratch of tdr_prefix_workspace_bytes(n) bytes, or NULL.  From 256 to 32 768 weights — the
 * reference's operating point — everything is one launch of one workgroup with the weights in LDS; above, with a
 * workspace, the chain is evaluated by many workgroups (per-chunk parity summaries, see csrc/tdr_prefix.hip); without
 * one, by one workgroup.  The bits written are the serial chain's either way.  tdr_config_prefix_small(0) takes the
 * one-launch kernel out of the choice (A/B measurements, tests), 1 restores the default, < 0 only queries; env
 * TDR_PFX_SMALL. */
int tdr_config_prefix_small(int on);
int64_t tdr_prefix_workspace_bytes(int64_t n);
int tdr_k_prefix(const float* w, int64_t n, float* runmax_out, void* workspace, void* stream);
/* Same, choosing the implementation: mode 0 = one wave adding in index order, 1 = the exact parallel kernel in one
 * workgroup (integer increments per binade), 2 = the multi-workgroup scan (workspace required), 3 = the one-launch
 * kernel (n <= 32 768); all give identical bits.  prefix_out (optional, modes 1 to 3) = raw sums. */
int tdr_k_prefix_mode(const float* w, int64_t n, int mode, float* runmax_out, float* prefix_out, void* workspace,
                      void* stream);
/* idx_out[i - i_begin] = first j with prefix_j > (float(i)+shift)/n_new, else n-1, for i in [i_begin, i_end). */
int tdr_k_resample(const float* runmax, int64_t n, int64_t n_new, float shift, int64_t i_begin, int64_t i_end,
                   int32_t* idx_out, void* stream);
/* The same with the shift read from device memory (tdr_k_rng_uniform): no host value in the step. */
int tdr_k_resample_dev(const float* runmax, int64_t n, int64_t n_new, const float* shift_dev, int64_t i_begin, int64_t i_end,
                       int32_t* idx_out, void* stream);
/* new_particles_[i]->setState(particles_[j]->state()) (:184): dst[f][i] = src[f][idx[i]].
 * src_shard == 0: src is a plain [7][src_cap] SoA.  src_shard > 0: src is the all-gathered [rank][7][src_shard]
 * buffer of a sharded filter and idx holds global particle indices (rank*src_shard + local). */
int tdr_k_gather_states(const float* src, int64_t src_cap, int64_t src_shard, const int32_t* idx, int64_t n_new,
                        float* dst, int64_t dst_cap, void* stream);

/* Buffers of a sharded filter (one rank per GPU): {a, b}[nl] -> one send buffer [2][nl]; the gathered
 * [world][2][nl] -> a_glob / b_glob [world*nl] in global particle order; the gathered state planes [world][7][nl] ->
 * a plain SoA [7][cap]. */
int tdr_k_shard_pack2(const float* a, const float* b, int64_t nl, float* out, void* stream);
int tdr_k_shard_unpack2(const float* in, int world, int64_t nl, float* a_glob, float* b_glob, void* stream);
int tdr_k_unshard_states(const float* in, int world, int64_t nl, float* st, int64_t cap, void* stream);

/* ---- getLocalMap materialised (src/top_down_map_polar.cpp:21-53, src/top_down_map.cpp:429-459) ----------------- */
/* The window of ONE pose as the reference's arrays: dists_out (device) [ncls][rows*cols] column-major images,
 * mask_out (device) [rows*cols], 1 = unknown or outside the map.  Polar: rows = nb, cols = nr, `tab` from
 * tdr_polar_table_host, centre (cx, cy) in metres like StateParticle's (cx / resolution is the cell).  Cartesian:
 * samplePts(center / resolution, rot, cols, rows, res / resolution) (top_down_map.cpp:433-434). */
int tdr_k_local_map_polar(const tdr_map_desc* map, const float* tab, int nb, int nr, float cx, float cy, float scale,
                          float res, float* dists_out, uint8_t* mask_out, void* stream);
int tdr_k_local_map_cart(const tdr_map_desc* map, int rows, int cols, float cx, float cy, float rot, float res,
                         float* dists_out, uint8_t* mask_out, void* stream);

/* ---- ActiveLocalizer (src/active_localizer.cpp; dead at its call sites src/particle_filter.cpp:77-78,316) -------------- */
/* computeTotalDifference (:7-20) over getLocalMap (:22-42) for many candidates at once.  centres (device) [ncand][K][2]:
 * position {x, y} of hypothesis i displaced by candidate q (:62-63); shifts (device) [K]: rot_shift of hypothesis i
 * (:33-36); res: getLocalMap's resolution (2 at :30); (nb, nr) the shape of `tab` (100 x 25 in the reference).
 * sums_out (device) [ncand] doubles: sum over pairs i > j, classes and samples of |L_i - L_j|; the reference's figure is
 * sum / (K (K-1) / 2 * ncls).  K <= 32. */
int tdr_k_active_diffs(const tdr_map_desc* map, const float* tab, int nb, int nr, float res, const float* centres,
                       const int32_t* shifts, int K, int ncand, double* sums_out, void* stream);
/* The candidates of getBestRelPos (:55-77) — distances 50, 75, 100, 125; directions by the float loop `theta += pi / 8`
 * as written — for the HOST arrays preds [K][3] = {x, y, theta}: centres [4 * 17][K][2], dists / thetas [4 * 17] (candidate
 * d * 17 + t), shifts [K]; *ntheta_out directions per distance, *ndist_out distances. */
int tdr_active_candidates_host(const float* preds, int K, int nb, float* centres, float* dists, float* thetas,
                               int32_t* shifts, int* ntheta_out, int* ndist_out);

/* ---- per-step consumers (src/particle_filter.cpp:191-236, 325-334, 343-357) --------------------------------- */
/* out (device, TDR_MEAN_COV_FLOATS floats; the first 24 are the result, the rest is scratch for the multi-workgroup
 * reductions): mean[4] (meanLikelihood :191-203), cov[16] row-major about the mean (computeMeanCov, about == NULL) or
 * about the 4 device floats `about` (computeCov about the max-likelihood mlState :226-236), geometric-mean scale
 * (freezeScale :345-348), 3 spare.  A pure function of (st, n, about). */
#define TDR_MEAN_COV_FLOATS 4800
int tdr_k_mean_cov(const float* st, int64_t cap, int64_t n, const float* about, float* out, void* stream);
int tdr_k_set_scale(float* st, int64_t cap, int64_t n, const float* scale_dev, void* stream);   /* freezeScale :350-352 */
/* max_likelihood_particle_ (:145-147): out12 (device) = the 7 SoA fields of particle argmax (info[0] of
 * tdr_k_update_weights), one spare, then its mlState {x, y, theta, scale} (state_particle.cpp:98-102).  Call before the
 * resampled set replaces `st`. */
int tdr_k_save_ml_state(const float* info, const float* st, int64_t cap, int64_t src_shard, int64_t n, float* out12,
                        void* stream);   /* src_shard > 0: st is the all-gathered [rank][7][src_shard] buffer */
int tdr_k_shift_init(float* st, int64_t cap, int64_t n, float dx, float dy, void* stream);      /* updateMap :325-334 */

/* The scoring kernels read the compact records whenever the map has them (tdr_k_compact_map); tdr_config_compact(0)
 * forces the dense records (A/B measurements, tests), 1 restores the default, < 0 only queries.  Results never depend on
 * it: both forms decode to the same operands.  Returns the value in force. */
int tdr_config_compact(int on);
/* Large polar launches take the INTEGER form of the score.  A scan count is an integer and a distance value of the map
 * an integer multiple of 2^-q (tdr_cmap_words_total above), so a class's product sum is accumulated as a 64-bit integer:
 * exact, hence independent of the order of the additions and of the kernel that forms it.  Two kernels share a launch:
 *  - DENSE particles — those whose 64 neighbours in the locality order lie within the SPAN, a number of map cells (0 =
 *    every particle counts as dense, a very large span = none) — go in (heading bin, Morton) order, every bin padded to
 *    whole waves, lane = particle: the scan side of a sample is a scalar operand and empty scan bins / absent classes are
 *    skipped wave-wide (csrc/tdr_score_su.hip);
 *  - the others one WAVE per particle, lanes = consecutive samples along a ray, gathering 2-byte cells of per-class
 *    planes (csrc/tdr_score_ray.hip).
 * Whichever kernel scores a particle, its weight is the same bits; the float kernel (score_polar_kernel: small filters,
 * shapes and maps the integer form does not cover, scans with fractional or non-finite counts — detected on the device)
 * agrees with it to rounding (<= 1e-6 relative).
 * mode 0 = never, 1 = when the filter holds enough particles per heading bin for the padding to pay (default: 64 x the
 * polar image's rows), 2 = whenever the shapes allow (ring groups and ring count multiples of 4, a map with narrow
 * compact records and class planes); < 0 only returns the mode.  Env TDR_SHIFT_UNIFORM sets the initial mode.
 * The span: a launch with a tdr_score_ctx TUNES it while the filter runs — from the 31st call of a shape on, 2, 8, 16, 24
 * and 40 cells are timed over two scoring calls each (HIP events on the caller's stream, polled, never waited for), the fastest is kept and the trial is
 * repeated every 4000 calls.  Results never depend on it.  tdr_config_shift_uniform_span(cells >= 0) fixes
 * it for every caller; -1 only returns the configured span (16 by default: what a launch without a context uses); -2 goes
 * back to tuning. */
int tdr_config_shift_uniform(int mode);
float tdr_config_shift_uniform_span(float cells);
/* Waves a scattered particle's window is split over in the ray-mapped kernel (1 .. 8; 0 = chosen per launch from its size,
 * the default; < 0 only queries).  The sums are exact integers: results never depend on it (tests). */
int tdr_config_ray_split(int k);
/* Large Cartesian launches (a filter of >= 4096 particles on a map with narrow compact records and class planes) take the
 * INTEGER form too, with the same guarantee as the polar one (tdr_config_shift_uniform above, whose mode 0 switches both
 * off): dense particles through the skipping kernel below with 64-bit integer accumulators, scattered ones one wave each,
 * lanes = consecutive window columns (csrc/tdr_score_cart.hip: score_cart_ray_kernel); the span is four times the polar
 * launch's.  The float kernels score what has no integer form and agree to rounding.
 * The Cartesian scoring has a second kernel that reads the scan side of a sample as a scalar descriptor and gives an empty
 * scan bin one 4-byte gather from the map's known mask instead of the record gather, decode and FMAs
 * (csrc/tdr_score_cart.hip); same partial sums, bit for bit.  It is used whenever the map has narrow compact records;
 * 0 forces the general kernel (A/B measurements, tests), 1 restores the default, < 0 only queries.  Env TDR_CART_SKIP. */
int tdr_config_cart_skip(int on);
/* The 40-rotation search of a particle without a heading (src/state_particle.cpp:195-206) runs on the matrix cores
 * (v_mfma_f32_16x16x32_f16; every record size: 4 / 8 floats through pre-split half records or split on the fly, 12 / 16
 * floats in two groups of 8 slots); 0 forces the vector-unit search (A/B measurements, tests), 1 restores the default,
 * < 0 only queries.  Only the choice among rotations whose costs tie to rounding can differ.  Env TDR_INIT_MFMA. */
int tdr_config_init_mfma(int on);
/* The weight statistics of at most 32 768 particles (src/particle_filter.cpp:107-147) evaluate their two serial float
 * chains wave by wave: predicted wave-chunks in parallel, the rest carried through by one wave without workgroup barriers
 * (csrc/tdr_prefix.hip); 0 forces the chunk-by-chunk evaluation on the whole workgroup (A/B measurements, tests), 1
 * restores the default, < 0 only queries.  Same bits either way.  Env TDR_UW_WAVES. */
int tdr_config_uw_waves(int on);
/* diagnostics: scoring launches of this process that took the shift-uniform kernel */
int64_t tdr_shift_uniform_launches(void);

/* ---- measurement ---------------------------------------------------------------------------------------------- */
/* While enabled, every launch of the scoring kernel is bracketed by HIP events on its launch stream;
 * tdr_profile_score_ms synchronises on them, returns the summed duration and the launch count, and resets. */
int tdr_profile_enable(int on);
int tdr_profile_score_ms(double* total_ms, int64_t* launches);
/* While enabled, a polar launch of the integer form also brackets each of its two kernels: the last such launch's durations — shift-uniform kernel over the dense particles,
 * ray-mapped kernel over the scattered ones — and the number of scattered particles (synchronises).  Measurement state is
 * process-wide like the switch itself: one measuring thread. */
int tdr_profile_shares(double* dense_ms, double* scattered_ms, int64_t* scattered_particles);
/* With tdr_profile_enable(2) — never inside a timed region: an atomic per wave-sector slows the kernels — the dense kernels count which loop variant each wave-sector (polar) / wave-segment (Cartesian) ran; reads
 * and resets the 16 counters (synchronises): [0..2] score_polar_su_kernel with the workgroup's staged box — every cell known /
 * every cell inside the map / general; [3..5] the same with the wave's own box; [6] its far path; [8..10]
 * score_cart_su_kernel — all known / inside / general; [11] its plain steps (a box that did not fit). */
int tdr_profile_variants(int64_t out[16]);

/* The library reads NOTHING from the process environment: behaviour switches are these calls (and the tdr_config_* calls
 * above), process-wide, meant for A/B measurements, tests and debugging — results never depend on them unless stated.
 * tdr_config_tuning(name, value): value < 0 queries; returns the value in force, -1 for an unknown name.
 *   "score_waves"      waves the float / Cartesian scoring kernels aim for (default 131072)
 *   "score_group"      rings per workgroup of the float and shift-uniform kernels, 0 = from the shapes (changes the
 *                      partition of the FLOAT kernel's sums: its results move in the last bits)
 *   "su_group"         rings per workgroup of the shift-uniform kernel alone (a multiple of 4; integer sums: same bits)
 *   "init_ahead"       record loads in flight in the init search (1-3)
 *   "prefix_head"      leading addends the long running sum's walk adds one by one
 *   "ray_block_major"  0: the ray-mapped kernel keeps its first row order (direction-major) also when the caller's context
 *                      holds the table's factors (same bits)
 *   "ray_patch"        0: the ray-mapped kernel walks 64 rings of ONE direction per step also where the patch order applies (a
 *                      context with the table's factors, a direction count that is a multiple of 16, at most 256): 1 (default) =
 *                      4 directions x 16 rings per step, a lane's four descriptors of a unit in one load (same bits)
 *   "ray_borrow"       0: an empty scan bin of the ray-mapped kernel reads its known bit from the coarse mask plane; 1 (default):
 *                      from the class plane its nearest non-empty neighbour of four rings reads anyway (same bits)
 *   "su_wave_span"     map cells a wave's own 64 same-heading particles may spread over before the wave is re-routed from the
 *                      shift-uniform kernel to the ray-mapped kernel (0, the default: never — measured, it does not pay) — same bits either way
 *   "mt_stretches"     0: the reference's random stream is always generated by one wave; 1 (default): calls of more than 128
 *                      state blocks fill stretches side by side, reached by jump-ahead (csrc/tdr_rng.hip) — the same words
 *   "cart_seg_rows"    window rows per segment of score_cart_su_kernel (a multiple of 4; 0: the Cartesian integer form's dense
 *                      share goes through the plain kernel instead — same bits)
 *   "su_lds_pad"       bytes of dynamic LDS added to a workgroup of score_polar_su_kernel: fewer workgroups fit a CU — an
 *                      occupancy sweep without touching the code object (0, the default; same bits) */
int64_t tdr_config_tuning(const char* name, int64_t value);
/* Device self-test of the scoring kernels: a tiny fixed problem (160 x 160 map, 6 classes, 512 particles) scored by every
 * kernel the library has for it.  The integer-form kernels run generated, hand-scheduled assembly; their sums are exact, so
 * score_polar_su_kernel == score_polar_ray_kernel and score_cart_su_kernel == score_cart_skip_kernel == score_cart_ray_kernel
 * BIT FOR BIT, and both agree with the float kernels to rounding (3e-6).  TDR_OK, or an error whose message says which
 * pair disagrees: a toolchain that miscompiles around the generated loops fails loudly here (smoke() calls it). */
int tdr_selftest_score(void);
/* Self-test hook: out[i] = the scoring loop's coordinate rounding of x[i] clamped to [-1, limit] (== roundf). */
int tdr_k_selftest_round(const float* x, int64_t n, float limit, int32_t* out, void* stream);
/* sinf / cosf on the device are the HOST libm's, bit for bit (csrc/tdr_sincosf.h: glibc >= 2.28's double-precision
 * algorithm restated).  glibc ships two builds of it, plain and FMA-contracted, and picks one per CPU at load time;
 * tdr_libm_variant() probes the host's sinf / cosf where the two differ: 1 = fused, 0 = plain, -1 = neither (an
 * unknown libm: the kernels then use the fused form and indices derived from sin / cos are unpinned).
 * tdr_libm_force_variant(0 | 1) overrides the probe (tests), -1 returns to it.  tdr_sincosf_host evaluates the
 * restatement on the host (variant 0 | 1), tdr_k_selftest_sincos on the device with the variant in force. */
int tdr_libm_variant(void);
int tdr_libm_force_variant(int variant);
int tdr_sincosf_host(const float* x, int64_t n, int variant, float* sin_out, float* cos_out);
int tdr_k_selftest_sincos(const float* x, int64_t n, float* sin_out, float* cos_out, void* stream);
/* logf on the device is the host libm's too (csrc/tdr_logf.h; glibc's two builds agree on every argument): the
 * restatement on the host / on the device (libstdc++'s normal_distribution<float> calls it, csrc/tdr_rng.hip). */
int tdr_logf_host(const float* x, int64_t n, float* out);
int tdr_k_selftest_logf(const float* x, int64_t n, float* out, void* stream);
/* Self-test hook: out[i] = the raster kernel's atan2f(y[i], x[i]) (bit-identical to glibc's atan2f). */
int tdr_k_selftest_atan2(const float* y, const float* x, int64_t n, float* out, void* stream);

/* ---- layout helpers ------------------------------------------------------------------------------------------ */
int tdr_k_states_aos_to_soa(const tdr_state* aos, int64_t n, float* st, int64_t cap, void* stream);
int tdr_k_states_soa_to_aos(const float* st, int64_t cap, int64_t n, tdr_state* aos, void* stream);
/* Locality order: perm_out = particle indices grouped by the 4x4 px map tile of their centre (row-major tiles).
 * keys_tmp: int32[tdr_locality_tmp_ints(n, map_rows, map_cols)] scratch. */
size_t tdr_locality_tmp_ints(int64_t n, int map_rows, int map_cols);
int tdr_k_locality_order(const float* st, int64_t cap, int64_t n, int map_rows, int map_cols, int32_t* perm_out,
                         int32_t* keys_tmp, void* stream);
/* The same for windows that rotate with the particle (Cartesian scoring, top_down_map.cpp:367-389): Morton order of
 * (x, y, theta), theta quantised so that one step moves a sample `theta_radius` cells from the centre by half a cell.
 * keys_tmp: tdr_locality_pose_tmp_ints(n) int32 of device scratch, 8-byte aligned. */
size_t tdr_locality_pose_tmp_ints(int64_t n);
int tdr_k_locality_order_pose(const float* st, int64_t cap, int64_t n, int map_rows, int map_cols, float theta_radius,
                              int32_t* perm_out, int32_t* keys_tmp, void* stream);

/* =================================================================================================================
 * Handle layer: C++ host code (csrc/tdr_host.cpp) that owns the device memory and sequences the kernels above the way
 * the reference's classes sequence their loops.  HOST pointers in and out; one handle per reference object; one caller
 * thread per handle.  This is what the headers under include/top_down_render/ (the reference's class surface) are
 * written against.
 * ================================================================================================================= */
typedef struct tdr_map tdr_map;            /* TopDownMapPolar   (top_down_map_polar.h:6-22)  */
typedef struct tdr_renderer tdr_renderer;  /* ScanRenderer[Polar] (scan_renderer_polar.h:15-22) */
typedef struct tdr_filter tdr_filter;      /* ParticleFilter    (particle_filter.h:22-73)    */

int tdr_map_create(tdr_map** out);
void tdr_map_destroy(tdr_map* m);
/* class_maps_/class_mask_ as computeDists leaves them (top_down_map.cpp:289-326), column-major HOST arrays; also the
 * map side of TopDownMap::updateMap (:146-157).  center = map_center_. */
int tdr_map_set(tdr_map* m, const float* class_maps, const uint8_t* class_mask, int ncls, int rows, int cols,
                float resolution, int center_x, int center_y);
/* TopDownMap::updateMap(const cv::Mat&, map_center) (top_down_map.cpp:146-157) for a HOST class-index image
 * (CV_8UC1 layout): loadCompressedRasterMap + the exact distance transform of computeDists run on the device.
 * haveMap() turns true only if the map contains road (class 1), like :150-154. */
int tdr_map_set_labels(tdr_map* m, const uint8_t* label_img, int img_h, int img_w, const int32_t* flatten_lut,
                       int lut_size, int ncls, float resolution, int center_x, int center_y);
int tdr_map_sample_pts_polar(tdr_map* m, int nb, int nr, float ang_res);                 /* top_down_map_polar.cpp:7-19 */
int tdr_map_polar_shape(const tdr_map* m, int* nb, int* nr);   /* the shape of the last samplePtsPolar (0, 0 before) */
int tdr_map_info(const tdr_map* m, int* ncls, int* rows, int* cols, float* resolution, int* have_map);
int tdr_map_center(const tdr_map* m, int* center_x, int* center_y);                      /* mapCenter(), top_down_map.h:72 */
/* getLocalMap (polar != 0: top_down_map_polar.cpp:21-53 with scale_or_rot = scale and the table of
 * tdr_map_sample_pts_polar, rows/cols ignored; else top_down_map.cpp:429-459 with scale_or_rot = rot).
 * dists_out HOST [ncls][rows*cols], mask_out HOST [rows*cols]. */
int tdr_map_local_map(tdr_map* m, int polar, float cx, float cy, float scale_or_rot, float res, int rows, int cols,
                      float* dists_out, uint8_t* mask_out);
int tdr_map_classes_at_point(const tdr_map* m, int px, int py, uint32_t* class_bits);    /* top_down_map.cpp:159-170 */
/* ActiveLocalizer::getBestRelPos (src/active_localizer.cpp:44-82) on the map's table (tdr_map_sample_pts_polar; the
 * reference's local maps are 100 x 25): preds HOST [K][3] = {x, y, theta} of the pose hypotheses (the mixture's means).
 * best_rel_pos = {distance, direction} of the displacement whose local maps differ most, *best_diff that mean difference
 * (0 and {0, 0} when nothing beats 0 — e.g. one hypothesis: the reference's 0 / 0 never wins). */
int tdr_map_best_rel_pos(tdr_map* m, const float* preds, int K, float best_rel_pos[2], float* best_diff);
/* getLocalGeoMap (top_down_map_polar.cpp:55-76 / top_down_map.cpp:461-481): the same window gathered from the two
 * geometric layers geo_maps_ — [0] distance to the nearest cell WITHOUT a geometric class (flattened class >= 3), [1]
 * to the nearest cell WITH one (getGeoRasterMap :410-427 + computeDists), as tdr_map_set derives them; after
 * tdr_map_set_labels both layers are the constant 1 the reference's updateMap path leaves them at (:126-133).
 * dists_out HOST [2][rows*cols]. */
int tdr_map_local_geo_map(tdr_map* m, int polar, float cx, float cy, float scale_or_rot, float res, int rows, int cols,
                          float* dists_out);
/* The reference's on-disk map cache (src/top_down_map.cpp:226-286; .eig = 2 x int64 shape + column-major scalars,
 * top_down_map.h:29-50): cached_data.txt, class_map<i>.eig, geo_map<i>.eig, class_mask.eig under cache_dir (NULL =
 * $HOME/.ros/xview_cache).  load: *loaded = 1 and the map is set when the cache's (map_path, num_classes, resolution)
 * match like loadCacheMetaData, else *loaded = 0.  save writes the files of the map the handle holds. */
int tdr_map_load_cache(tdr_map* m, const char* cache_dir, const char* map_path, int num_classes, float resolution,
                       int center_x, int center_y, int* loaded);
int tdr_map_save_cache(tdr_map* m, const char* cache_dir, const char* map_path);
/* The raster cache (TopDownMap::saveRasterizedMaps / loadRasterizedMaps, src/top_down_map.cpp:197-224): a directory of
 * class<i>.png — 8-bit greyscale, 0 inside class i and 255 elsewhere, stored flipped like the reference stores them — read
 * and written over zlib (no OpenCV).  load = the constructor's path for such a directory (:44-58): rasters -> geometric
 * layers -> distance transforms, on the device; the map takes the images' shape. */
int tdr_map_save_rasters(tdr_map* m, const char* dir);
/* the codec on its own (host only): 8-bit greyscale, non-interlaced PNG; px row-major, row 0 = top.  read: px_out holds
 * `capacity` bytes, *w / *h are set even when the image does not fit (then TDR_ERR_ARG). */
int tdr_png_read_gray8_host(const char* path, uint8_t* px_out, int64_t capacity, int* w, int* h);
int tdr_png_write_gray8_host(const char* path, const uint8_t* px, int w, int h);
int tdr_map_load_rasters(tdr_map* m, const char* dir, int num_classes, float resolution, int center_x, int center_y);

int tdr_renderer_create(const int32_t* flatten_lut256, tdr_renderer** out);              /* scan_renderer.cpp:3-5 */
void tdr_renderer_destroy(tdr_renderer* r);
/* renderSemanticTopDown: polar != 0 -> scan_renderer_polar.cpp:83-109 (rows = theta bins, cols = range bins),
 * else scan_renderer.cpp:55-78.  imgs_out: HOST [ncls][rows*cols] column-major, or NULL to keep the render on the
 * device for tdr_filter_update. */
int tdr_renderer_render(tdr_renderer* r, int polar, const float* pts, int stride, int ioff, int64_t n, float res,
                        float ang_res, int ncls, int rows, int cols, float* imgs_out);

/* renderGeometricTopDown (scan_renderer_polar.cpp:6-81 / scan_renderer.cpp:7-53): HOST organised cloud (element
 * idy*width + idx; pcl clouds: width = cloud->width, height = cloud->height), imgs_out HOST [2][rows*cols]. */
int tdr_renderer_render_geo(tdr_renderer* r, int polar, const float* pts, int stride, int64_t width, int64_t height,
                            float res, float ang_res, int rows, int cols, float* imgs_out);

/* ---- several GPUs: one process (rank) per GPU, particles partitioned contiguously by rank -------------
 * A tdr_comm carries the three exchange steps of a sharded filter — broadcast of the rasterised scan from rank 0,
 * all-gather of {raw weight, last_dist}, all-gather of the state planes — over RCCL (xGMI), called directly on the
 * filter's stream, or over functions the caller supplies (MPI, a test double).  Every rank then computes the weight
 * statistics and the order-exact running sum on the same gathered arrays, so an N-rank filter equals the 1-rank filter
 * bit for bit (tests/test_sharded_handle.py).  The reference has no counterpart: it is one CPU process. */
#define TDR_COMM_ID_BYTES 128
typedef struct tdr_comm tdr_comm;
typedef struct tdr_comm_ops {   /* device pointers; enqueue on `stream` (a hipStream_t) or complete before returning */
  void* ctx;
  int (*all_gather)(void* ctx, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
  int (*broadcast)(void* ctx, void* buf_dev, size_t bytes, int root, void* stream);
} tdr_comm_ops;
/* RCCL transport: rank 0 makes the id (ncclGetUniqueId) and hands its 128 bytes to the other ranks by any means; every
 * rank then calls create_rccl with its HIP device current (ncclCommInitRank).  librccl.so is loaded on first use. */
int tdr_comm_rccl_unique_id(void* id_out128);
int tdr_comm_create_rccl(int world_size, int rank, const void* unique_id128, tdr_comm** out);
int tdr_comm_create(int world_size, int rank, const tdr_comm_ops* ops, tdr_comm** out);
void tdr_comm_destroy(tdr_comm* c);
int tdr_comm_world(const tdr_comm* c);
int tdr_comm_rank(const tdr_comm* c);
int tdr_comm_all_gather(tdr_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
int tdr_comm_broadcast(tdr_comm* c, void* buf_dev, size_t bytes, int root, void* stream);
/* A filter whose particles are sharded over comm's ranks: n_max (the GLOBAL maximum) and every particle count must be
 * multiples of the world size; every rank makes the same calls with the same arguments (same seed: the host generator
 * is replicated), each on its own map handle holding the same map.  set_states takes the GLOBAL array and keeps this
 * rank's slice; get_states / get_raw_weights / get_last_dist / get_resample_indices return this rank's slice
 * (tdr_filter_num_local entries, indices global); get_weights, mean_cov, scale, num_particles are global and identical
 * on every rank.  tdr_filter_update: ranks other than 0 may pass scan_imgs == NULL and renderer == NULL — the packed scan
 * of rank 0 is broadcast. */
int tdr_filter_create_sharded(tdr_map* map, int n_max, const tdr_filter_params* fp, uint32_t seed, tdr_comm* comm,
                              tdr_filter** out);
int64_t tdr_filter_num_local(const tdr_filter* f);

/* seed: the reference seeds its std::mt19937 from std::random_device (src/particle_filter.cpp:4-5), i.e. not
 * reproducibly.  seed == 0 stands for that case: propagate's noise is then drawn on the device (counter-based, 4 us).
 * seed != 0: every draw comes from std::mt19937(seed) in the reference's order, propagate's 4N normals included (host
 * code, ~35 ns per draw) — the mode the parity tests use.  tdr_filter_configure changes the mode afterwards. */
int tdr_filter_create(tdr_map* map, int n_max, const tdr_filter_params* fp, uint32_t seed, tdr_filter** out);
void tdr_filter_destroy(tdr_filter* f);
/* parity_rng: propagate consumes host std::mt19937 normals in the reference's order; 0 = device RNG.
 * locality_every: > 0 processes particles in Morton order of their map position (default 1; results unchanged). */
int tdr_filter_configure(tdr_filter* f, int parity_rng, int locality_every);
int tdr_filter_initialize_particles(tdr_filter* f);                                      /* particle_filter.cpp:19-84 */
int tdr_filter_set_states(tdr_filter* f, const tdr_state* states, int64_t n);
int tdr_filter_get_states(tdr_filter* f, tdr_state* out, int64_t n);
int tdr_filter_propagate(tdr_filter* f, float tx, float ty, float omega);                /* particle_filter.cpp:86-92 */
/* particle_filter.cpp:94-189.  scan_imgs: HOST [ncls][nb*nr] images or NULL (= renderer's last render, no host round
 * trip), (nb, nr) = tdr_map_polar_shape — the caller checks its images against it, ncls*nb*nr floats are read;
 * n_target < 0 keeps the particle count (explicit input of the adaptive count :151-157).  A filter sharded over W ranks
 * keeps the same number of particles on every rank: n_target is rounded DOWN to a multiple of W (at least W), so a W-rank
 * filter asked for 70 particles at W = 8 resamples 64 and equals, bit for bit, the one-rank filter asked for 64. */
int tdr_filter_update(tdr_filter* f, const float* scan_imgs, const tdr_renderer* renderer, float res, int64_t n_target);
/* The same with the geometric images top_down_geo (HOST [2][nb*nr]) entering the score (tdr_k_score_polar_geo); the
 * reference passes them to update() too but its score ignores them.  Not available on a sharded filter. */
int tdr_filter_update_geo(tdr_filter* f, const float* scan_imgs, const float* geo_imgs, float res, int64_t n_target);
int tdr_filter_get_weights(tdr_filter* f, float* out, int64_t n);
/* The per-particle surface of StateParticle (include/top_down_render/state_particle.h:42-53) on a filter:
 * computeWeight for every particle without statistics / resampling (state_particle.cpp:157-219) and its result
 * (weight(), :55), lastDist(), propagate with the caller's scale_freeze flag (:57-78), the constructor's draw of one
 * particle (:3-49), and the generator shared with the caller (state_particle.h:61-64: `mt19937` is a std::mt19937* of
 * this library's libstdc++, not owned; the filter then consumes it in the reference's draw order). */
int tdr_filter_compute_weights(tdr_filter* f, const float* scan_imgs, const tdr_renderer* renderer, float res);
int tdr_filter_get_raw_weights(tdr_filter* f, float* out, int64_t n);
int tdr_filter_get_last_dist(tdr_filter* f, float* out, int64_t n);
int tdr_filter_propagate_freeze(tdr_filter* f, float tx, float ty, float omega, int scale_freeze);
int tdr_filter_init_one(tdr_filter* f);
int tdr_filter_share_rng(tdr_filter* f, void* mt19937);
int tdr_filter_get_resample_indices(tdr_filter* f, int32_t* out, int64_t n);
/* about_max == 0: meanLikelihood + computeMeanCov (:191-220); != 0: maxLikelihood + computeCov (:222-236). */
int tdr_filter_mean_cov(tdr_filter* f, int about_max, float state[4], float cov[16]);
int tdr_filter_freeze_scale(tdr_filter* f);                                              /* :343-357 */
int tdr_filter_is_scale_frozen(const tdr_filter* f);
float tdr_filter_scale(tdr_filter* f);                                                   /* :359-367 */
int64_t tdr_filter_num_particles(const tdr_filter* f);                                   /* :369-371 */
int tdr_filter_update_map(tdr_filter* f, const float* class_maps, const uint8_t* class_mask, int ncls, int rows, int cols,
                          float resolution, int center_x, int center_y);                 /* :320-341 */
int tdr_filter_update_map_labels(tdr_filter* f, const uint8_t* label_img, int img_h, int img_w,
                                 const int32_t* flatten_lut, int lut_size, int ncls, float resolution, int center_x,
                                 int center_y);                                          /* :320-341, cv::Mat form */
/* ---- adaptive particle count from a Gaussian mixture (src/particle_filter.cpp:151-157, 245-318) ---- */
/* out (device, [num][3] floats): mlState().head<3>() = {x, y, theta} of particle min(n-1, i*n/num) (:262-266). */
int tdr_k_sample_ml_states(const float* st, int64_t cap, int64_t n, int num, float* out, void* stream);
/* Deterministic EM fit of k full-covariance Gaussians to m samples of 4 doubles (host code, tdr_gmm.cpp; the
 * reference uses cv::ml::EM, randomly initialised: parity unpinned).  weights [k], means [k][4], covs [k][4][4]. */
int tdr_gmm_fit_host(const double* samples, int m, int k, int max_iter, double* weights, double* means, double* covs,
                     double* mean_loglik);
/* computeGMM's search for the cluster count around *num_gaussians_io (:259, 276-297) and the conversion of the clusters
 * to means [k][3] = {x, y, atan2} / covs [k][9] (:303-312).  samples [m][4] = {x, y, 50 cos theta, 50 sin theta}. */
int tdr_gmm_select_host(const double* samples, int m, int64_t num_particles, int* num_gaussians_io, int max_k,
                        float* means_out, float* covs_out);
/* num_particles_ of :151-157 from the clusters' covariances. */
int64_t tdr_adaptive_count_host(const float* covs, int k, int64_t last_count, int64_t max_count);
/* Handle layer: computeGMM (:252-318) on the filter's current particles (synchronous; the reference runs it in a
 * detached thread once per second), getGMM (:238-243), and the count of :151-157 from the stored clusters
 * (n_target for tdr_filter_update).  max_k bounds the arrays passed to get_gmm. */
#define TDR_GMM_MAX_K 32
int tdr_filter_compute_gmm(tdr_filter* f);
int tdr_filter_get_gmm(tdr_filter* f, int max_k, int* k_out, float* means, float* covs);
int64_t tdr_filter_adaptive_count(tdr_filter* f);

/* internal: lets tdr_host.cpp report through tdr_last_error() */
int tdr_set_error(int code, const char* msg);

#ifdef __cplusplus
}
#endif
#endif /* TDR_H_ */
