// call_sites.cpp — compile-only check of the drop-in boundary: every way TopDownRender (the reference's ROS node,
// src/top_down_render.cpp) touches the hot-path classes, written out with the argument types the node uses, against
// include/top_down_render/*.h.  Nothing here runs on a GPU; `main` only proves the translation unit links.  The
// statements are this repository's own — they restate the SHAPE of each call (reference line in the comment), not the
// node's code; ROS / semantics_manager types are reduced to the members those calls read.
#include <cmath>
#include <cstdio>
#include <vector>

#include "top_down_render/particle_filter.h"
#include "top_down_render/scan_renderer_polar.h"
#include "top_down_render/top_down_map_polar.h"

namespace node_side {                       // what the node holds (top_down_render.h:36-108), reduced
struct ClassConfig {                        // semantics_manager::ClassConfig: the members getTopDownMapParams reads
  SemanticColorLut color_lut;
  std::vector<int> class_to_flattened;
  std::vector<int> flattened_to_class;
};

struct Node {
  TopDownMapPolar* map_ = nullptr;
  ParticleFilter* filter_ = nullptr;
  ScanRendererPolar* renderer_ = nullptr;
  Eigen::VectorXi flatten_lut_;
  float current_range_scale_ = 1.f;

  TopDownMap::Params mapParams(const ClassConfig& class_params) {            // :161-190
    TopDownMap::Params params;
    params.map_path = "";
    params.color_lut = class_params.color_lut;                                // :173
    params.flatten_lut = class_params.class_to_flattened;                     // :174
    params.num_classes = (int)class_params.flattened_to_class.size();         // :175
    params.exclusive_classes.resize(params.num_classes);                      // :177
    params.resolution = 1;                                                    // :186
    params.out_of_bounds_const = 5;                                           // :187
    return params;
  }
  void initialize(const ClassConfig& cc, FilterParams& filter_params, int particle_count) {
    map_ = new TopDownMapPolar(mapParams(cc));                                // :81
    map_->samplePtsPolar(Eigen::Vector2i(100, 25), 2 * M_PI / 100);           // :115
    filter_ = new ParticleFilter(particle_count, map_, filter_params);        // :116
    renderer_ = new ScanRendererPolar(flatten_lut_);                          // :117
  }
  void takeStep(const pcl::PointCloud<pcl::PointXYZI>::ConstPtr& cloud_ptr, cv::Mat& background_copy) {
    if (!map_->haveMap()) return;                                             // :508
    std::vector<Eigen::ArrayXXf> top_down, top_down_geo;
    for (int i = 0; i < map_->numClasses(); i++) top_down.push_back(Eigen::ArrayXXf(100, 25));   // :529-532
    for (int i = 0; i < 2; i++) top_down_geo.push_back(Eigen::ArrayXXf(100, 25));                // :533-536
    renderer_->renderSemanticTopDown(cloud_ptr, current_range_scale_, 2 * M_PI / 100, top_down);  // :539
    renderer_->renderGeometricTopDown(cloud_ptr, current_range_scale_, 2 * M_PI / 100, top_down_geo);  // :540 (commented out there)
    Eigen::Vector2f motion_priort(1.f, 0.f);
    float motion_priora = 0.01f;
    filter_->propagate(motion_priort, motion_priora);                         // :423
    filter_->update(top_down, top_down_geo, current_range_scale_);            // :425
    filter_->visualize(background_copy);                                      // :431
  }
  void publishPoseEst() {                                                     // :331-365
    Eigen::Matrix4f cov;
    filter_->computeMeanCov(cov);                                             // :333
    const float scale = filter_->scale();                                     // :335
    if (cov(0, 0) / (scale * scale) > 1.f) current_range_scale_ += 0.05f;
    if (filter_->numParticles() < 1) return;                                  // :347
    Eigen::Vector4f ml_state;
    filter_->meanLikelihood(ml_state);                                        // :354
    if (cov(3, 3) < 0.003f * ml_state[3] && !filter_->isScaleFrozen()) filter_->freezeScale();   // :356-359
    std::vector<Eigen::Vector3f> means;
    std::vector<Eigen::Matrix3f> covs;
    filter_->getGMM(means, covs);
    Eigen::Vector4f best;
    filter_->maxLikelihood(best);
    filter_->computeCov(cov);
    // the two lines the reference keeps commented out (src/particle_filter.cpp:77-78, 316)
    ActiveLocalizer* active_loc_ = new ActiveLocalizer(map_);
    Eigen::Vector2f best_rel_pos_ = active_loc_->getBestRelPos(means);
    (void)best_rel_pos_;
    delete active_loc_;
  }
  void aerialMap(const cv::Mat& map_img) {                                    // :574-593
    Eigen::Vector2i map_loc_eig(10, 20);
    filter_->updateMap(map_img, map_loc_eig);                                 // :591
    map_->updateMap(map_img, map_loc_eig);                                    // TopDownMap::updateMap, top_down_map.h:65
  }
};

// StateParticle's const accessors (state_particle.h:45-51) must be callable through a const reference
inline float read_particle(const StateParticle& p) { return p.state().scale + p.weight() + p.lastDist(); }
}  // namespace node_side

int main() {
  std::printf("call sites compile: %zu\n", sizeof(node_side::Node));
  return 0;
}
