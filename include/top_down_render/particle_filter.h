// ParticleFilter — reference surface: include/top_down_render/particle_filter.h:22-73, src/particle_filter.cpp.
// Same constructor and method signatures as the reference, so the call sites of TopDownRender
// (src/top_down_render.cpp:116, 333-359, 423-425, 591) compile against it unchanged; the work runs on the MI355X
// through tdr_filter (include/tdr.h).  Documented differences:
//   * an explicit seed replaces std::random_device (src/particle_filter.cpp:4-5): the default 0 means "unseeded" —
//     propagate then draws its noise on the device; a non-zero seed reproduces the reference's std::mt19937 draw order
//     (host code, 80 000 serial draws per step at 20 000 particles); configure() switches explicitly;
//   * the mixture behind the adaptive particle count (:151-157, 245-318) is fitted on demand by computeGMM() with a
//     deterministic EM (csrc/tdr_gmm.cpp) instead of cv::ml::EM in a detached thread; getGMM() returns it;
//     setAdaptiveCount(true) feeds it into update() like :151-157, setTargetCount(n) overrides;
//   * visualize(cv::Mat&) (:373-423) works on a host snapshot of the particle set: a hook the host sets (setVisualizer)
//     gets it; without a hook and with OpenCV present (TDR_HAVE_OPENCV) the default drawing of particle_viz.h — particles
//     as red arrows, the mixture as blue ellipses, the best particle as a blue arrow, like the reference — runs; without
//     either the call is a no-op (there is nothing to draw with);
//   * top_down_geo is accepted and ignored like in the reference's score, whose geometric block is commented out
//     (src/state_particle.cpp:145-152); setGeometricCost(true) switches that block on.
#ifndef PARTICLE_FILTER_H_
#define PARTICLE_FILTER_H_

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "top_down_render/active_localizer.h"   // like the reference's header (particle_filter.h:5)
#include "top_down_render/scan_renderer.h"
#include "top_down_render/state_particle.h"

class ParticleFilter {
 public:
  ParticleFilter(int N, TopDownMapPolar* map, FilterParams& params, uint32_t seed = 0) : map_(map), params_(params) {
    if (!map) throw std::invalid_argument("ParticleFilter: null map");
    max_num_particles_ = N;
    tdr_filter_params c = to_tdr_params(params_, map_->numClasses());
    if (tdr_filter_create(map_->handle(), N, &c, seed, &f_) != TDR_OK) fail("ParticleFilter");
    if (map_->haveMap() && tdr_filter_initialize_particles(f_) != TDR_OK) {   // particle_filter.cpp:14-16
      const std::string msg = std::string("initializeParticles: ") + tdr_last_error();
      tdr_filter_destroy(f_);   // the destructor does not run for a constructor that throws
      throw std::runtime_error(msg);
    }
  }
  // Several GPUs, one process per GPU: the particles are sharded over the ranks of `comm` (tdr_comm_create_rccl: RCCL over
  // xGMI; or tdr_comm_create with the host's own transport).  N is the GLOBAL particle count, a multiple of the world
  // size; every rank constructs the filter with the same arguments and makes the same calls; update() on ranks other
  // than 0 may be given empty images — rank 0's rasterised scan is broadcast.  Results equal the one-GPU filter bit for
  // bit; statistics (computeMeanCov, meanLikelihood, maxLikelihood, scale, numParticles) are global on every rank,
  // states() returns this rank's slice.
  ParticleFilter(int N, TopDownMapPolar* map, FilterParams& params, uint32_t seed, tdr_comm* comm) : map_(map), params_(params) {
    if (!map || !comm) throw std::invalid_argument("ParticleFilter: null map / comm");
    max_num_particles_ = N;
    tdr_filter_params c = to_tdr_params(params_, map_->numClasses());
    if (tdr_filter_create_sharded(map_->handle(), N, &c, seed, comm, &f_) != TDR_OK) fail("ParticleFilter");
    sharded_rank_ = tdr_comm_rank(comm);
    if (map_->haveMap() && tdr_filter_initialize_particles(f_) != TDR_OK) {
      const std::string msg = std::string("initializeParticles: ") + tdr_last_error();
      tdr_filter_destroy(f_);
      throw std::runtime_error(msg);
    }
  }
  ~ParticleFilter() { tdr_filter_destroy(f_); }
  ParticleFilter(const ParticleFilter&) = delete;
  ParticleFilter& operator=(const ParticleFilter&) = delete;

  void propagate(Eigen::Vector2f& trans, float omega) {                                   // :86-92
    check(tdr_filter_propagate(f_, trans[0], trans[1], omega), "propagate");
  }
  void update(std::vector<Eigen::ArrayXXf>& top_down_scan, std::vector<Eigen::ArrayXXf>& top_down_geo, float res) {  // :94-189
    if (sharded_rank_ > 0 && top_down_scan.empty()) {   // not rank 0 of a sharded filter: the scan arrives by broadcast
      if (numParticles() > 0) check(tdr_filter_update(f_, nullptr, nullptr, res, next_count()), "update");
      return;
    }
    if (top_down_scan.empty() || numParticles() == 0) return;
    const int ncls = map_->numClasses();
    // Wrong-sized input: the reference's library never throws — its per-scan calls return silently on a size they cannot
    // use (scan_renderer_polar.cpp:85, top_down_map_polar.cpp:25, particle_filter.cpp:96-99) — and the node's only handler
    // is main()'s catch-and-exit (top_down_render_node.cpp:8-14).  Same here: nothing is scored, the particle set stays as
    // it is, and the reason is left in tdr_last_error().
    if ((int)top_down_scan.size() < ncls) return soft_fail("update: fewer scan images than map classes");
    // tdr_filter_update reads ncls * nb * nr floats, (nb, nr) = the shape given to samplePtsPolar: every image must
    // have exactly that shape (the reference indexes the images with the table's size too, state_particle.cpp:178-188)
    const Eigen::Vector2i shape = map_->polarShape();
    const size_t P = (size_t)shape[0] * shape[1];
    for (int c = 0; c < ncls; c++)
      if (top_down_scan[c].rows() != shape[0] || top_down_scan[c].cols() != shape[1])
        return soft_fail("update: scan image " + std::to_string(c) + " is " + std::to_string(top_down_scan[c].rows()) +
                         "x" + std::to_string(top_down_scan[c].cols()) + ", samplePtsPolar was given " +
                         std::to_string(shape[0]) + "x" + std::to_string(shape[1]));
    std::vector<float> buf(P * ncls);
    for (int c = 0; c < ncls; c++) std::memcpy(buf.data() + P * c, top_down_scan[c].data(), P * sizeof(float));
    if (geometric_cost_ && top_down_geo.size() >= 2 && (size_t)top_down_geo[0].size() == P && (size_t)top_down_geo[1].size() == P) {
      std::vector<float> geo(2 * P);
      for (int i = 0; i < 2; i++) std::memcpy(geo.data() + P * i, top_down_geo[i].data(), P * sizeof(float));
      check(tdr_filter_update_geo(f_, buf.data(), geo.data(), res, next_count()), "update");
      return;
    }
    check(tdr_filter_update(f_, buf.data(), nullptr, res, next_count()), "update");
  }
  // Extension: let top_down_geo enter the score — getCostForRot's geometric block (src/state_particle.cpp:145-152), which
  // the reference has commented out.  Off by default = the reference's behaviour.
  void setGeometricCost(bool on) { geometric_cost_ = on; }
  // Extension: score against the renderer's last render without copying the images through the host.
  void update(const ScanRenderer& renderer, float res) {
    check(tdr_filter_update(f_, nullptr, renderer.handle(), res, next_count()), "update");
  }
  // getGMM (:238-243) / computeGMM (:252-318)
  void computeGMM() { check(tdr_filter_compute_gmm(f_), "computeGMM"); }
  void getGMM(std::vector<Eigen::Vector3f>& means, std::vector<Eigen::Matrix3f>& covs) {
    float m[3 * TDR_GMM_MAX_K], c[9 * TDR_GMM_MAX_K];
    int k = 0;
    check(tdr_filter_get_gmm(f_, TDR_GMM_MAX_K, &k, m, c), "getGMM");
    means.clear();
    covs.clear();
    for (int g = 0; g < k; g++) {
      means.push_back(Eigen::Vector3f(m[3 * g], m[3 * g + 1], m[3 * g + 2]));
      Eigen::Matrix3f cv;
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) cv(i, j) = c[9 * g + 3 * i + j];
      covs.push_back(cv);
    }
  }
  void setAdaptiveCount(bool on) { adaptive_ = on; }   // :151-157 from the clusters of the last computeGMM
  void computeCov(Eigen::Matrix4f& cov) { stat(1, nullptr, &cov); }                        // :226-236
  void maxLikelihood(Eigen::Vector4f& state) { stat(1, &state, nullptr); }                 // :222-224
  void computeMeanCov(Eigen::Matrix4f& cov) { stat(0, nullptr, &cov); }                    // :205-220
  void meanLikelihood(Eigen::Vector4f& state) { stat(0, &state, nullptr); }                // :191-203
  void freezeScale() { check(tdr_filter_freeze_scale(f_), "freezeScale"); }                // :343-357
  bool isScaleFrozen() { return tdr_filter_is_scale_frozen(f_) != 0; }
  float scale() const { return tdr_filter_scale(f_); }                                     // :359-367
  int numParticles() const { return (int)tdr_filter_num_particles(f_); }                   // :369-371
  // updateMap (:320-341) with the map already in distance-map form (see TopDownMap::setDistanceMaps)
  void updateMap(const std::vector<Eigen::ArrayXXf>& class_maps, const Eigen::ArrayXXc& class_mask,
                 const Eigen::Vector2i& map_center) {
    if (class_maps.empty()) return;
    const int rows = (int)class_maps[0].rows(), cols = (int)class_maps[0].cols(), ncls = (int)class_maps.size();
    std::vector<float> buf((size_t)rows * cols * ncls);
    for (int c = 0; c < ncls; c++)
      std::memcpy(buf.data() + (size_t)c * rows * cols, class_maps[c].data(), (size_t)rows * cols * sizeof(float));
    check(tdr_filter_update_map(f_, buf.data(), class_mask.data(), ncls, rows, cols, map_->resolution(), map_center[0],
                                map_center[1]), "updateMap");
  }

  // updateMap (:320-341) for a class-index image in cv::Mat CV_8UC1 layout with an explicit flatten LUT
  void updateMap(const uint8_t* label_img, int img_h, int img_w, const std::vector<int>& flatten_lut,
                 const Eigen::Vector2i& map_center) {
    std::vector<int32_t> lut(flatten_lut.begin(), flatten_lut.end());
    check(tdr_filter_update_map_labels(f_, label_img, img_h, img_w, lut.data(), (int)lut.size(), map_->numClasses(),
                                       map_->resolution(), map_center[0], map_center[1]), "updateMap");
  }
  // The reference's signature (particle_filter.h:41, call site src/top_down_render.cpp:591): a class-index image, the
  // flatten LUT comes from the map's Params like in TopDownMap::updateMap (src/top_down_map.cpp:146-148).
  void updateMap(const cv::Mat& map, const Eigen::Vector2i& map_center) {
    if (map.empty()) throw std::invalid_argument("updateMap: empty image");
    const std::vector<int>& lut = map_->params().flatten_lut;
    if (lut.empty()) throw std::invalid_argument("updateMap: the map's Params::flatten_lut is not set");
    if (map.isContinuous()) return updateMap(map.ptr<uint8_t>(), map.rows, map.cols, lut, map_center);
    std::vector<uint8_t> packed((size_t)map.rows * map.cols);
    for (int r = 0; r < map.rows; r++) std::memcpy(packed.data() + (size_t)r * map.cols, map.ptr<uint8_t>(r), (size_t)map.cols);
    updateMap(packed.data(), map.rows, map.cols, lut, map_center);
  }
  // visualize (call site src/top_down_render.cpp:431).  Drawing is out of scope here and this library
  // links no image library: the call hands the host a copy of the particle states, the mixture of the last computeGMM
  // and the max-likelihood state through the hook set with setVisualizer(), and does nothing when none is set.
  struct Snapshot {
    std::vector<State> particles;
    std::vector<Eigen::Vector3f> gmm_means;
    std::vector<Eigen::Matrix3f> gmm_covs;
    bool have_best = false;
    Eigen::Vector4f best;   // mlState of the max-likelihood particle (valid after the first update)
  };
  typedef void (*Visualizer)(cv::Mat& img, const Snapshot& snap, void* user);
  void setVisualizer(Visualizer fn, void* user = nullptr) { visualizer_ = fn; visualizer_user_ = user; }
  void visualize(cv::Mat& img) {
#ifndef TDR_HAVE_OPENCV
    if (!visualizer_) return;
#endif
    Snapshot snap;
    snap.particles = states();
    getGMM(snap.gmm_means, snap.gmm_covs);
    float s[4], c[16];
    if (tdr_filter_mean_cov(f_, 1, s, c) == TDR_OK) {   // fails before the first update: no best particle yet
      snap.have_best = true;
      for (int i = 0; i < 4; i++) snap.best[i] = s[i];
    }
    if (visualizer_) visualizer_(img, snap, visualizer_user_);
#ifdef TDR_HAVE_OPENCV
    else drawSnapshot(img, snap);   // include/top_down_render/particle_viz.h
#endif
  }
#ifdef TDR_HAVE_OPENCV
  static void drawSnapshot(cv::Mat& img, const Snapshot& snap);   // the default drawing (particle_viz.h)
#endif

  // --- beyond the reference's surface -------------------------------------------------------------------------------
  void setTargetCount(int n) { target_count_ = n; }  // explicit adaptive particle count; < 0 keeps N
  void configure(bool parity_rng, int locality_every) { check(tdr_filter_configure(f_, parity_rng, locality_every), "configure"); }
  void setStates(const std::vector<State>& s) {
    check(tdr_filter_set_states(f_, reinterpret_cast<const tdr_state*>(s.data()), (int64_t)s.size()), "setStates");
  }
  std::vector<State> states() {   // this rank's particles (all of them unless the filter is sharded)
    std::vector<State> s((size_t)tdr_filter_num_local(f_));
    if (!s.empty()) check(tdr_filter_get_states(f_, reinterpret_cast<tdr_state*>(s.data()), (int64_t)s.size()), "states");
    return s;
  }
  std::vector<float> weights(int n) {
    std::vector<float> w((size_t)n);
    if (n > 0) check(tdr_filter_get_weights(f_, w.data(), n), "weights");
    return w;
  }
  std::vector<float> rawWeights(int n) {   // StateParticle::weight() of every particle, as scored by the last update
    std::vector<float> w((size_t)n);
    if (n > 0) check(tdr_filter_get_raw_weights(f_, w.data(), n), "rawWeights");
    return w;
  }
  std::vector<int32_t> resampleIndices() {
    std::vector<int32_t> idx((size_t)tdr_filter_num_local(f_));
    if (!idx.empty()) check(tdr_filter_get_resample_indices(f_, idx.data(), (int64_t)idx.size()), "resampleIndices");
    return idx;
  }

 private:
  void stat(int about_max, Eigen::Vector4f* state, Eigen::Matrix4f* cov) {
    float s[4], c[16];
    check(tdr_filter_mean_cov(f_, about_max, s, c), "statistics");
    if (state) for (int i = 0; i < 4; i++) (*state)[i] = s[i];
    if (cov) for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) (*cov)(i, j) = c[4 * i + j];
  }
  int64_t next_count() {
    if (target_count_ >= 0) return target_count_;
    return adaptive_ ? tdr_filter_adaptive_count(f_) : -1;
  }
  void check(int rc, const char* what) { if (rc != TDR_OK) fail(what); }
  // a per-scan call given something it cannot use: no exception (see update()), the message goes to tdr_last_error()
  static void soft_fail(const std::string& msg) { tdr_set_error(TDR_ERR_ARG, msg.c_str()); }
  [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string(what) + ": " + tdr_last_error()); }

  int max_num_particles_ = 0;
  int sharded_rank_ = -1;   // >= 0: rank of this process in a sharded filter
  int target_count_ = -1;
  bool adaptive_ = false;
  bool geometric_cost_ = false;
  Visualizer visualizer_ = nullptr;
  void* visualizer_user_ = nullptr;
  TopDownMapPolar* map_;
  FilterParams params_;
  tdr_filter* f_ = nullptr;
};

#ifdef TDR_HAVE_OPENCV
#include "top_down_render/particle_viz.h"
#endif

#endif  // PARTICLE_FILTER_H_
