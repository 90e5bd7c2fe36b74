// ScanRendererPolar over libtdr_hip — the class surface TopDownRender uses (reference:
// include/top_down_render/scan_renderer_polar.h:15-22, call sites src/top_down_render.cpp:117,539).
//
// Header-only: all work happens in ScanRenderer::render (scan_renderer.h), which copies the cloud to the GPU, runs the
// two raster kernels of csrc/tdr_raster.hip and copies the class images back into the caller's Eigen arrays.  The
// packed copy of the render stays on the device, so ParticleFilter::update(const ScanRenderer&, res) can score it
// without another round trip.
//
// Image convention (src/scan_renderer_polar.cpp:93-108): imgs[c] has one ROW per theta bin and one COLUMN per range
// bin, column-major like every Eigen::ArrayXXf; a point (x, y) lands in row round(atan2(x, y) / ang_res) + rows / 2 and
// column round(hypot(x, y) / res); points at the origin and points whose class the flatten LUT maps to -1 are skipped;
// fewer images than one per class -> the call returns without touching anything, like the reference (:85).
#ifndef TDR_FACADE_SCAN_RENDERER_POLAR_H_
#define TDR_FACADE_SCAN_RENDERER_POLAR_H_

#include "top_down_render/scan_renderer.h"

class ScanRendererPolar : public ScanRenderer {
 public:
  explicit ScanRendererPolar(const Eigen::VectorXi& flatten_lut) : ScanRenderer(flatten_lut) {}

  // Semantic render, polar: see the convention above.
  void renderSemanticTopDown(const pcl::PointCloud<pcl::PointXYZI>::ConstPtr& cloud, float res, float ang_res,
                             std::vector<Eigen::ArrayXXf>& imgs) {
    constexpr int kPolar = 1;
    render(kPolar, cloud, res, ang_res, imgs);
  }

  // Geometric render, polar (src/scan_renderer_polar.cpp:6-81): per theta bin the returns are sorted by range
  // descending (equal ranges keep their input order) and walked: slope > 1 against the previous return counts an
  // obstacle at its range bin, slope < 0.3 (and no obstacle just before) counts ground from the previous range bin up
  // to this one.  imgs[0] ground, imgs[1] obstacles.  (Commented out at the reference's call site,
  // src/top_down_render.cpp:540; part of the class surface.)
  void renderGeometricTopDown(const pcl::PointCloud<PointType>::ConstPtr& cloud, float res, float ang_res,
                              std::vector<Eigen::ArrayXXf>& imgs) {
    render_geo(1, cloud, res, ang_res, imgs);
  }
};

#endif  // TDR_FACADE_SCAN_RENDERER_POLAR_H_
