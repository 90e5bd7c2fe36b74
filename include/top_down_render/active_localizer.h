// ActiveLocalizer — reference surface: include/top_down_render/active_localizer.h:6-17, src/active_localizer.cpp.
// Same constructor and public method as the reference.  The reference keeps
// the class alive but never calls it (src/particle_filter.cpp:77-78, 316 are commented out); it is rebuilt for
// completeness of the surface ParticleFilter's header pulls in.  Every candidate displacement of getBestRelPos is one
// workgroup of ONE launch on the MI355X (csrc/tdr_active.hip); the local maps are gathered in place, never materialised.
// Like the reference, local maps are the shape of the map's sample table: call it after
// map->samplePtsPolar(Eigen::Vector2i(100, 25), ...) as the node does (src/top_down_render.cpp:115).
#ifndef ACTIVE_LOCALIZER_H_
#define ACTIVE_LOCALIZER_H_

#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "top_down_render/top_down_map_polar.h"

class ActiveLocalizer {
 public:
  explicit ActiveLocalizer(TopDownMapPolar* map) : map_(map) {                                 // :3-5
    if (!map) throw std::invalid_argument("ActiveLocalizer: null map");
  }
  // :44-82.  preds: {x, y, theta} of the pose hypotheses (the mixture's means, particle_filter.cpp:316);
  // returns {distance, direction} of the relative move that tells them apart best ({0, 0} when nothing beats 0).
  Eigen::Vector2f getBestRelPos(std::vector<Eigen::Vector3f>& preds) {
    last_best_diff_ = 0.f;
    if (preds.empty()) return Eigen::Vector2f(0.f, 0.f);   // (the reference's loops then compare NaN: nothing wins)
    std::vector<float> p(3 * preds.size());
    for (size_t i = 0; i < preds.size(); i++)
      for (int d = 0; d < 3; d++) p[3 * i + d] = preds[i][d];
    float out[2] = {0.f, 0.f};
    if (tdr_map_best_rel_pos(map_->handle(), p.data(), (int)preds.size(), out, &last_best_diff_) != TDR_OK)
      throw std::runtime_error(std::string("getBestRelPos: ") + tdr_last_error());
    return Eigen::Vector2f(out[0], out[1]);
  }
  float lastBestDiff() const { return last_best_diff_; }   // the "Max diff" the reference logs (:79)

  // (computeTotalDifference and getLocalMap, :7-42, are private helpers of the reference's CPU loop; the launch above gathers
  // the hypotheses' cells in place and has no host images to hand them)

 private:
  TopDownMapPolar* map_;
  float last_best_diff_ = 0.f;
};

#endif  // ACTIVE_LOCALIZER_H_
