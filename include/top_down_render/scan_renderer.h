// ScanRenderer — the reference's class surface (include/top_down_render/scan_renderer.h:14-23 in the reference)
// over the MI355X raster kernel (tdr_renderer, include/tdr.h).  Same constructor and method signatures, same
// ownership rule (the caller sizes the images, the call zero-fills and writes them in place), same silent early
// return on an empty image list.
#ifndef SCAN_RENDERER_H_
#define SCAN_RENDERER_H_

#include <stdexcept>
#include <string>
#include <vector>

#include "tdr.h"
#include "top_down_render/tdr_compat.h"

typedef pcl::PointXYZI PointType;

class ScanRenderer {
 public:
  explicit ScanRenderer(const Eigen::VectorXi& flatten_lut) {  // src/scan_renderer.cpp:3-5
    if (flatten_lut.size() != 256) throw std::invalid_argument("flatten_lut must have 256 entries");
    int32_t lut[256];
    for (int i = 0; i < 256; i++) lut[i] = flatten_lut[i];
    if (tdr_renderer_create(lut, &r_) != TDR_OK) throw std::runtime_error(std::string("ScanRenderer: ") + tdr_last_error());
  }
  virtual ~ScanRenderer() { tdr_renderer_destroy(r_); }
  ScanRenderer(const ScanRenderer&) = delete;
  ScanRenderer& operator=(const ScanRenderer&) = delete;

  // src/scan_renderer.cpp:55-78
  void renderSemanticTopDown(const pcl::PointCloud<pcl::PointXYZI>::ConstPtr& cloud, float res,
                             std::vector<Eigen::ArrayXXf>& imgs) {
    render(0, cloud, res, 1.f, imgs);
  }
  // src/scan_renderer.cpp:7-53: ground (imgs[0]) / obstacle (imgs[1]) counts along the scan lines of the organised
  // cloud.  (Commented out at the reference's call site, src/top_down_render.cpp:540; part of the class surface.)
  void renderGeometricTopDown(const pcl::PointCloud<PointType>::ConstPtr& cloud, float res, std::vector<Eigen::ArrayXXf>& imgs) {
    render_geo(0, cloud, res, 1.f, imgs);
  }
  const tdr_renderer* handle() const { return r_; }  // device-resident last render, for ParticleFilter::update

 protected:
  void render(int polar, const pcl::PointCloud<pcl::PointXYZI>::ConstPtr& cloud, float res, float ang_res,
              std::vector<Eigen::ArrayXXf>& imgs) {
    if (imgs.size() < 1) return;
    const int rows = (int)imgs[0].rows(), cols = (int)imgs[0].cols(), ncls = (int)imgs.size();
    const size_t P = (size_t)rows * cols;
    // images of different sizes: the reference would write past the smaller ones; a silent return, the reason in tdr_last_error()
    for (int c = 1; c < ncls; c++)
      if (imgs[c].rows() != imgs[0].rows() || imgs[c].cols() != imgs[0].cols()) {
        tdr_set_error(TDR_ERR_ARG, "renderSemanticTopDown: images of different sizes");
        return;
      }
    std::vector<float> buf(P * ncls);
    const float* pts = cloud && !cloud->points.empty() ? reinterpret_cast<const float*>(cloud->points.data()) : nullptr;
    const int64_t n = cloud ? (int64_t)cloud->points.size() : 0;
    if (tdr_renderer_render(r_, polar, pts, 8, 4, n, res, ang_res, ncls, rows, cols, buf.data()) != TDR_OK)
      throw std::runtime_error(std::string("renderSemanticTopDown: ") + tdr_last_error());
    for (int c = 0; c < ncls; c++) std::memcpy(imgs[c].data(), buf.data() + P * c, P * sizeof(float));
  }
  void render_geo(int polar, const pcl::PointCloud<PointType>::ConstPtr& cloud, float res, float ang_res,
                  std::vector<Eigen::ArrayXXf>& imgs) {
    if (imgs.size() < 2) return;                                   // scan_renderer.cpp:8
    for (auto& im : imgs) im.setZero();                            // :12-14
    const int rows = (int)imgs[0].rows(), cols = (int)imgs[0].cols();
    const size_t P = (size_t)rows * cols;
    if (P == 0 || (size_t)imgs[1].rows() * imgs[1].cols() != P) return;
    const int64_t n = cloud ? (int64_t)cloud->points.size() : 0;
    int64_t width = cloud ? (int64_t)cloud->width : 0, height = cloud ? (int64_t)cloud->height : 0;
    if (width * height != n) { width = n; height = 1; }            // not organised: one line of returns
    std::vector<float> buf(2 * P);
    const float* pts = n ? reinterpret_cast<const float*>(cloud->points.data()) : nullptr;
    if (tdr_renderer_render_geo(r_, polar, pts, 8, width, height, res, ang_res, rows, cols, buf.data()) != TDR_OK)
      throw std::runtime_error(std::string("renderGeometricTopDown: ") + tdr_last_error());
    std::memcpy(imgs[0].data(), buf.data(), P * sizeof(float));
    std::memcpy(imgs[1].data(), buf.data() + P, P * sizeof(float));
  }
  tdr_renderer* r_ = nullptr;
};

#endif  // SCAN_RENDERER_H_
