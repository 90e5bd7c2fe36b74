// ActiveLocalizer — reference surface: include/top_down_render/active_localizer.h:6-17, src/active_localizer.cpp.
// Same constructor and methods as the reference (its two private helpers are public here, for tests).  The reference keeps
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

  // :7-20 on host images (the device path never builds them)
  float computeTotalDifference(std::vector<std::vector<Eigen::ArrayXXf>>& local_maps) {
    float total_difference = 0;
    int cnt = 0;
    for (size_t i = 0; i < local_maps.size(); i++)
      for (size_t j = 0; j < i; j++)
        for (size_t cls = 0; cls < local_maps[0].size(); cls++) {
          const Eigen::ArrayXXf& a = local_maps[i][cls];
          const Eigen::ArrayXXf& b = local_maps[j][cls];
          double s = 0;
          for (Eigen::Index k = 0; k < a.size(); k++) s += std::fabs(a.data()[k] - b.data()[k]);
          total_difference += (float)s;
          cnt += 1;
        }
    return total_difference / (float)cnt;
  }
  // :22-42: the window at state.head<2>() (res 2, scale 1) with its rows rotated by the heading
  void getLocalMap(Eigen::Vector3f& state, std::vector<Eigen::ArrayXXf>& local_map) {
    const Eigen::Vector2i shape = map_->polarShape();
    std::vector<Eigen::ArrayXXf> orig;
    for (int n = 0; n < map_->numClasses(); n++) orig.push_back(Eigen::ArrayXXf(shape[0], shape[1]));
    Eigen::ArrayXXc mask(shape[0], shape[1]);
    map_->getLocalMap(Eigen::Vector2f(state[0], state[1]), 2.f, orig, mask);                    // :30
    const int num_bins = (int)local_map[0].rows();
    int rot_shift = (int)std::round((double)(state[2] * (float)num_bins / 2) / 3.14159265358979323846);   // :33
    while (rot_shift >= num_bins) rot_shift -= num_bins;
    while (rot_shift < 0) rot_shift += num_bins;
    for (int n = 0; n < map_->numClasses(); n++)
      for (Eigen::Index j = 0; j < local_map[n].cols(); j++)
        for (int a = 0; a < num_bins; a++)
          local_map[n](a, j) = orig[n](a < rot_shift ? num_bins - rot_shift + a : a - rot_shift, j);   // :38-41
  }

 private:
  TopDownMapPolar* map_;
  float last_best_diff_ = 0.f;
};

#endif  // ACTIVE_LOCALIZER_H_
