// ParticleFilter — reference surface: include/top_down_render/particle_filter.h:22-73, src/particle_filter.cpp.
// Same constructor and method signatures as the reference, so the call sites of TopDownRender
// (src/top_down_render.cpp:116, 333-359, 423-425, 591) compile against it unchanged; the work runs on the MI355X
// through tdr_filter (include/tdr.h).  Documented differences:
//   * an explicit seed replaces std::random_device (src/particle_filter.cpp:4-5): the default 0 means "unseeded" —
//     propagate then draws its noise on the device; a non-zero seed reproduces the reference's std::mt19937 draw order
//     (host code, 80 000 serial draws per step at 20 000 particles); configure() switches explicitly;
//   * the mixture behind the adaptive particle count (:151-157, 245-318) is fitted on demand by computeGMM() with a
//     deterministic EM (csrc/tdr_gmm.cpp) instead of cv::ml::EM in a detached thread; getGMM() returns it;
//     setAdaptiveCount(true) feeds it into update() like :151-157, setTargetCount(n) overrides; visualize() (OpenCV
//     drawing) is not provided;
//   * top_down_geo is accepted and ignored like in the reference's score (src/state_particle.cpp:145-152).
#ifndef PARTICLE_FILTER_H_
#define PARTICLE_FILTER_H_

#include <stdexcept>
#include <string>
#include <vector>

#include "top_down_render/scan_renderer.h"
#include "top_down_render/state_particle.h"

class ParticleFilter {
 public:
  ParticleFilter(int N, TopDownMapPolar* map, FilterParams& params, uint32_t seed = 0) : map_(map), params_(params) {
    if (!map) throw std::invalid_argument("ParticleFilter: null map");
    max_num_particles_ = N;
    tdr_filter_params c = to_tdr_params(params_, map_->numClasses());
    if (tdr_filter_create(map_->handle(), N, &c, seed, &f_) != TDR_OK) fail("ParticleFilter");
    if (map_->haveMap()) check(tdr_filter_initialize_particles(f_), "initializeParticles");  // particle_filter.cpp:14-16
  }
  ~ParticleFilter() { tdr_filter_destroy(f_); }
  ParticleFilter(const ParticleFilter&) = delete;
  ParticleFilter& operator=(const ParticleFilter&) = delete;

  void propagate(Eigen::Vector2f& trans, float omega) {                                   // :86-92
    check(tdr_filter_propagate(f_, trans[0], trans[1], omega), "propagate");
  }
  void update(std::vector<Eigen::ArrayXXf>& top_down_scan, std::vector<Eigen::ArrayXXf>& /*top_down_geo*/, float res) {  // :94-189
    if (top_down_scan.empty() || numParticles() == 0) return;
    const size_t P = (size_t)top_down_scan[0].size();
    const int ncls = map_->numClasses();
    if ((int)top_down_scan.size() < ncls) throw std::invalid_argument("update: fewer scan images than map classes");
    std::vector<float> buf(P * ncls);
    for (int c = 0; c < ncls; c++) std::memcpy(buf.data() + P * c, top_down_scan[c].data(), P * sizeof(float));
    check(tdr_filter_update(f_, buf.data(), nullptr, res, next_count()), "update");
  }
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

  // updateMap (:320-341) for a class-index image in cv::Mat CV_8UC1 layout; the map's Params carry the flatten LUT
  void updateMap(const uint8_t* label_img, int img_h, int img_w, const std::vector<int>& flatten_lut,
                 const Eigen::Vector2i& map_center) {
    std::vector<int32_t> lut(flatten_lut.begin(), flatten_lut.end());
    check(tdr_filter_update_map_labels(f_, label_img, img_h, img_w, lut.data(), (int)lut.size(), map_->numClasses(),
                                       map_->resolution(), map_center[0], map_center[1]), "updateMap");
  }
#ifdef CV_VERSION
  void updateMap(const cv::Mat& map, const std::vector<int>& flatten_lut, const Eigen::Vector2i& map_center) {
    cv::Mat m = map.isContinuous() ? map : map.clone();
    updateMap(m.ptr<uint8_t>(), m.rows, m.cols, flatten_lut, map_center);
  }
#endif

  // --- beyond the reference's surface -------------------------------------------------------------------------------
  void setTargetCount(int n) { target_count_ = n; }  // explicit adaptive particle count; < 0 keeps N
  void configure(bool parity_rng, int locality_every) { check(tdr_filter_configure(f_, parity_rng, locality_every), "configure"); }
  void setStates(const std::vector<State>& s) {
    check(tdr_filter_set_states(f_, reinterpret_cast<const tdr_state*>(s.data()), (int64_t)s.size()), "setStates");
  }
  std::vector<State> states() {
    std::vector<State> s((size_t)numParticles());
    if (!s.empty()) check(tdr_filter_get_states(f_, reinterpret_cast<tdr_state*>(s.data()), (int64_t)s.size()), "states");
    return s;
  }
  std::vector<float> weights(int n) {
    std::vector<float> w((size_t)n);
    if (n > 0) check(tdr_filter_get_weights(f_, w.data(), n), "weights");
    return w;
  }
  std::vector<int32_t> resampleIndices() {
    std::vector<int32_t> idx((size_t)numParticles());
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
  [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string(what) + ": " + tdr_last_error()); }

  int max_num_particles_ = 0;
  int target_count_ = -1;
  bool adaptive_ = false;
  TopDownMapPolar* map_;
  FilterParams params_;
  tdr_filter* f_ = nullptr;
};

#endif  // PARTICLE_FILTER_H_
