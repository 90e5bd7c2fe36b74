// TopDownRenderCore — the ROS-free part of the reference's orchestrator `TopDownRender`
// (include/top_down_render/top_down_render.h:36-108, src/top_down_render.cpp): what the node does with the hot-path
// classes once a cloud and a motion prior have arrived, and nothing of what it does with ROS.
//
//   initialize   :81, 115-117    new TopDownMapPolar / samplePtsPolar(100 x 25) / new ParticleFilter / new ScanRendererPolar
//   takeStep     :505-560        render at current_range_scale_, updateFilter, publishPoseEst
//   updateFilter :413-425        propagate(trans, yaw), update(top_down, top_down_geo, res)
//   publishPoseEst :331-365      range-scale stepping (+0.05 / -0.02 inside [range_scale_min_, range_scale_max_]), the
//                                freezeScale trigger (cov(3,3) < 0.003 scale), the convergence gate
//
// The node changes `res` on EVERY step (publishPoseEst moves current_range_scale_ each time it runs), so a filter is
// scored with a different range scale scan after scan: tests/test_takestep_loop.py drives this class for that reason.
// Message conversion (pcl::fromROSMsg, tf::poseMsgToEigen), publishers, TF and the visualisation image stay with the
// host; `PoseEst` is what publishPoseEst would have put on the wire.
#ifndef TOP_DOWN_RENDER_CORE_H_
#define TOP_DOWN_RENDER_CORE_H_

#include <algorithm>
#include <cmath>

#include "top_down_render/particle_filter.h"
#include "top_down_render/scan_renderer_polar.h"
#include "top_down_render/top_down_map_polar.h"

class TopDownRenderCore {
 public:
  struct Config {                       // the node's parameters that reach the step (src/top_down_render.cpp:45-53)
    int particle_count = 20000;         // :53
    float range_scale_min = 0.5f;       // :45
    float range_scale_max = 4.f;        // :46
    float target_uncertainty_m = 2.5f;  // top_down_render.h:80
    int theta_bins = 100, range_bins = 25;   // hard-coded 100 x 25 in the node (:115, 530, 534); a parameter here
    uint32_t seed = 0;                  // ParticleFilter's seed (0: device noise; see particle_filter.h)
  };
  struct PoseEst {                      // what publishPoseEst computed this step
    Eigen::Matrix4f cov;                // computeMeanCov (:333)
    Eigen::Vector4f ml_state;           // meanLikelihood (:354); valid when have_ml
    bool have_ml = false;
    float scale = -1.f;                 // filter_->scale() at :335
    float range_scale = 0.f;            // current_range_scale_ AFTER the step's adjustment: the next scan's res
    bool froze_scale = false;           // freezeScale() was called in this step (:356-359)
    bool converged = false;             // is_converged_ (:362-364; sticky)
  };

  TopDownRenderCore() { current_range_scale_ = cfg_.range_scale_max; }
  explicit TopDownRenderCore(const Config& cfg) : cfg_(cfg) {
    current_range_scale_ = cfg_.range_scale_max;   // :47
  }
  ~TopDownRenderCore() {
    delete renderer_;
    delete filter_;
    if (own_map_) delete map_;
  }
  TopDownRenderCore(const TopDownRenderCore&) = delete;
  TopDownRenderCore& operator=(const TopDownRenderCore&) = delete;

  // :81, 115-117 with the map constructed here from its Params (the static-map path)
  void initialize(const TopDownMap::Params& map_params, FilterParams& filter_params, const Eigen::VectorXi& flatten_lut) {
    map_ = new TopDownMapPolar(map_params);
    own_map_ = true;
    finishInit(filter_params, flatten_lut);
  }
  // ... with a map the host built (and keeps): dynamic maps arrive later through aerialMap()
  void initialize(TopDownMapPolar* map, FilterParams& filter_params, const Eigen::VectorXi& flatten_lut) {
    map_ = map;
    own_map_ = false;
    finishInit(filter_params, flatten_lut);
  }

  // takeStep (:505-560).  trans / yaw: the motion prior's delta already projected to the plane like updateFilter does
  // (:418-420, see projectPrior).  Returns false when the step was skipped (no map yet, :508-511).
  bool takeStep(const pcl::PointCloud<PointType>::ConstPtr& cloud_ptr, Eigen::Vector2f trans, float yaw, PoseEst* est = nullptr) {
    if (!map_->haveMap()) return false;
    const float ang_res = (float)(2 * M_PI / cfg_.theta_bins);
    if ((int)top_down_.size() != map_->numClasses()) {                                    // :528-536
      top_down_.clear();
      for (int i = 0; i < map_->numClasses(); i++) top_down_.push_back(Eigen::ArrayXXf(cfg_.theta_bins, cfg_.range_bins));
      top_down_geo_.clear();
      for (int i = 0; i < 2; i++) top_down_geo_.push_back(Eigen::ArrayXXf(cfg_.theta_bins, cfg_.range_bins));
    }
    last_res_ = current_range_scale_;
    renderer_->renderSemanticTopDown(cloud_ptr, current_range_scale_, ang_res, top_down_);    // :539
    updateFilter(top_down_, top_down_geo_, current_range_scale_, trans, yaw);                 // :559
    PoseEst e = publishPoseEst();                                                            // :560
    if (est) *est = e;
    return true;
  }
  void updateFilter(std::vector<Eigen::ArrayXXf>& top_down, std::vector<Eigen::ArrayXXf>& top_down_geo, float res,
                    Eigen::Vector2f trans, float yaw) {
    filter_->propagate(trans, yaw);                                                          // :423
    if (device_scan_) filter_->update(*renderer_, res);   // the images the renderer just produced, without the host copy
    else filter_->update(top_down, top_down_geo, res);                                       // :425
  }
  // publishPoseEst (:331-365) without the publishing
  PoseEst publishPoseEst() {
    PoseEst e;
    filter_->computeMeanCov(e.cov);                                                          // :333
    const float scale = filter_->scale();                                                    // :335
    const float scale_2 = scale * scale;
    e.scale = scale;
    const float spread = std::max(e.cov(0, 0), e.cov(1, 1)) / scale_2;
    // (the node compares against std::pow(float, int), a double, and steps its float member by double constants)
    if ((double)spread > std::pow((double)cfg_.target_uncertainty_m, 2) && current_range_scale_ < cfg_.range_scale_max) {
      current_range_scale_ = (float)((double)current_range_scale_ + 0.05);                    // :341 widen the local region
    } else if (current_range_scale_ > cfg_.range_scale_min) {
      current_range_scale_ = (float)((double)current_range_scale_ - 0.02);                    // :344 shrink to refine
    }
    e.range_scale = current_range_scale_;
    e.converged = is_converged_;
    if (filter_->numParticles() < 1) return e;                                               // :347-350
    filter_->meanLikelihood(e.ml_state);                                                     // :354
    e.have_ml = true;
    if ((double)e.cov(3, 3) < 0.003 * (double)e.ml_state[3] && !filter_->isScaleFrozen()) {   // :356
      filter_->freezeScale();                                                                // :359
      e.froze_scale = true;
    }
    if (e.cov(0, 0) / scale_2 < 40 && e.cov(1, 1) / scale_2 < 40 && e.cov(2, 2) < 0.5 && filter_->scale() > 0)   // :363
      is_converged_ = true;
    e.converged = is_converged_;
    return e;
  }
  // aerialMapCallback's effect on the hot-path classes (:574-593)
  void aerialMap(const cv::Mat& map_img, const Eigen::Vector2i& map_center) { filter_->updateMap(map_img, map_center); }

  // the plane projection of a 3-D motion prior (updateFilter, :418-420): R row-major 3 x 3, t the translation
  static void projectPrior(const float R[9], const float t[3], Eigen::Vector2f& trans, float& yaw) {
    trans = Eigen::Vector2f(t[0], t[1]);
    yaw = std::atan2(R[3], R[0]);   // (rotation * UnitX) = first column; atan2(y, x) of it
  }

  TopDownMapPolar* map() { return map_; }
  ParticleFilter* filter() { return filter_; }
  ScanRendererPolar* renderer() { return renderer_; }
  float currentRangeScale() const { return current_range_scale_; }
  float lastRes() const { return last_res_; }              // the res the last takeStep rendered and scored with
  bool isConverged() const { return is_converged_; }
  std::vector<Eigen::ArrayXXf>& topDown() { return top_down_; }
  // Extension: score the renderer's device images directly (ParticleFilter::update(const ScanRenderer&, float))
  void setDeviceScan(bool on) { device_scan_ = on; }

 private:
  void finishInit(FilterParams& filter_params, const Eigen::VectorXi& flatten_lut) {
    map_->samplePtsPolar(Eigen::Vector2i(cfg_.theta_bins, cfg_.range_bins), (float)(2 * M_PI / cfg_.theta_bins));   // :115
    filter_ = new ParticleFilter(cfg_.particle_count, map_, filter_params, cfg_.seed);                              // :116
    renderer_ = new ScanRendererPolar(flatten_lut);                                                                 // :117
  }

  Config cfg_;
  TopDownMapPolar* map_ = nullptr;
  bool own_map_ = false;
  ParticleFilter* filter_ = nullptr;
  ScanRendererPolar* renderer_ = nullptr;
  std::vector<Eigen::ArrayXXf> top_down_, top_down_geo_;
  float current_range_scale_ = 4.f;    // top_down_render.h:82
  float last_res_ = 0.f;
  bool is_converged_ = false;          // :83
  bool device_scan_ = false;
};

#endif  // TOP_DOWN_RENDER_CORE_H_
