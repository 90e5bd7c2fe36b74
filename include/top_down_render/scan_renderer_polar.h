// ScanRendererPolar — reference surface: include/top_down_render/scan_renderer_polar.h:15-22.
#ifndef SCAN_RENDERER_POLAR_H_
#define SCAN_RENDERER_POLAR_H_

#include "top_down_render/scan_renderer.h"

class ScanRendererPolar : public ScanRenderer {
 public:
  explicit ScanRendererPolar(const Eigen::VectorXi& flatten_lut) : ScanRenderer(flatten_lut) {}
  // src/scan_renderer_polar.cpp:83-109; imgs[c] is (theta bins x range bins)
  void renderSemanticTopDown(const pcl::PointCloud<pcl::PointXYZI>::ConstPtr& cloud, float res, float ang_res,
                             std::vector<Eigen::ArrayXXf>& imgs) {
    render(1, cloud, res, ang_res, imgs);
  }
  void renderGeometricTopDown(const pcl::PointCloud<PointType>::ConstPtr&, float, float,
                              std::vector<Eigen::ArrayXXf>& imgs) {
    for (auto& im : imgs) im.setZero();  // src/scan_renderer_polar.cpp:11-13; see ScanRenderer::renderGeometricTopDown
  }
};

#endif  // SCAN_RENDERER_POLAR_H_
