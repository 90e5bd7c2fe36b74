// TopDownMapPolar — reference surface: include/top_down_render/top_down_map_polar.h:6-22.  The per-pose window gather
// (getLocalMap, src/top_down_map_polar.cpp:21-53) is fused into the scoring kernel and never materialised.
#ifndef TOP_DOWN_MAP_POLAR_H_
#define TOP_DOWN_MAP_POLAR_H_

#include "top_down_render/top_down_map.h"

class TopDownMapPolar : public TopDownMap {
 public:
  explicit TopDownMapPolar(const Params& params) : TopDownMap(params) {
    samplePtsPolar(Eigen::Vector2i(100, 50), (float)(2 * 3.14159265358979323846 / 100));  // top_down_map_polar.cpp:3-5
  }
  void samplePtsPolar(Eigen::Vector2i shape, float ang_res) {                              // :7-19
    if (tdr_map_sample_pts_polar(m_, shape[0], shape[1], ang_res) != TDR_OK)
      throw std::runtime_error(std::string("samplePtsPolar: ") + tdr_last_error());
  }
};

#endif  // TOP_DOWN_MAP_POLAR_H_
