// TopDownMapPolar — reference surface: include/top_down_render/top_down_map_polar.h:6-22.
//
// Inside ParticleFilter::update the per-pose window gather (getLocalMap, src/top_down_map_polar.cpp:21-53) is fused
// into the scoring kernel and never materialised; the methods below materialise it for one pose on request
// (tdr_map_local_map), like the reference's public method.
#ifndef TOP_DOWN_MAP_POLAR_H_
#define TOP_DOWN_MAP_POLAR_H_

#include "top_down_render/top_down_map.h"

class TopDownMapPolar : public TopDownMap {
 public:
  explicit TopDownMapPolar(const Params& params, const char* cache_dir = nullptr) : TopDownMap(params, cache_dir) {
    samplePtsPolar(Eigen::Vector2i(100, 50), (float)(2 * 3.14159265358979323846 / 100));  // top_down_map_polar.cpp:3-5
  }
  void samplePtsPolar(Eigen::Vector2i shape, float ang_res) {                              // :7-19
    if (tdr_map_sample_pts_polar(m_, shape[0], shape[1], ang_res) != TDR_OK)
      throw std::runtime_error(std::string("samplePtsPolar: ") + tdr_last_error());
    shape_ = shape;
  }
  // :21-53.  dists[c] / mask are (theta bins x range bins) as set by samplePtsPolar; mask: 1 = unknown or outside.
  void getLocalMap(Eigen::Vector2f center, float scale, float res, std::vector<Eigen::ArrayXXf>& dists,
                   Eigen::ArrayXXc& mask) {
    if (dists.size() < 1) return;   // :25
    if (dists[0].rows() * dists[0].cols() != (Eigen::Index)shape_[0] * shape_[1] || mask.rows() * mask.cols() != dists[0].rows() * dists[0].cols()) {
      // (the reference would index past the arrays here; a silent return, the reason in tdr_last_error())
      tdr_set_error(TDR_ERR_ARG, "getLocalMap: output arrays do not have the shape given to samplePtsPolar");
      return;
    }
    local_map(1, center, scale, res, shape_[0], shape_[1], dists, mask);
  }
  void getLocalMap(Eigen::Vector2f center, float res, std::vector<Eigen::ArrayXXf>& dists, Eigen::ArrayXXc& mask) {
    getLocalMap(center, 1.f, res, dists, mask);                                           // :78-81
  }
  // :55-76: the polar window gathered from the two geometric layers geo_maps_ (see TopDownMap::getLocalGeoMap)
  void getLocalGeoMap(Eigen::Vector2f center, float scale, float res, std::vector<Eigen::ArrayXXf>& dists) {
    if (dists.size() < 1) return;   // :58
    if (dists[0].rows() * dists[0].cols() != (Eigen::Index)shape_[0] * shape_[1]) {
      tdr_set_error(TDR_ERR_ARG, "getLocalGeoMap: output arrays do not have the shape given to samplePtsPolar");
      return;
    }
    local_geo_map(1, center, scale, res, shape_[0], shape_[1], dists);
  }
  void getLocalGeoMap(Eigen::Vector2f center, float res, std::vector<Eigen::ArrayXXf>& dists) {
    getLocalGeoMap(center, 1.f, res, dists);
  }

  Eigen::Vector2i polarShape() const { return shape_; }   // (theta bins, range bins) of the last samplePtsPolar

 private:
  Eigen::Vector2i shape_{100, 50};
};

#endif  // TOP_DOWN_MAP_POLAR_H_
