// tdr_compat.h — stand-ins for the few Eigen / PCL types the reference's hot-path class surface mentions, used ONLY
// when the real headers are not installed (this image has neither).  With Eigen and PCL present the real types are
// used and nothing here is compiled.  Layouts match: ArrayXXf is column-major float (element (i,j) at i + rows*j),
// pcl::PointXYZI is 32 bytes with intensity at float offset 4.
#ifndef TOP_DOWN_RENDER_TDR_COMPAT_H_
#define TOP_DOWN_RENDER_TDR_COMPAT_H_

#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#define TDR_HAVE_EIGEN 1
#endif
#if __has_include(<pcl/point_cloud.h>) && __has_include(<pcl/point_types.h>)
#define TDR_HAVE_PCL 1
#endif
#if __has_include(<opencv2/core.hpp>) && __has_include(<opencv2/imgproc.hpp>)
#define TDR_HAVE_OPENCV 1
#endif
#if __has_include(<semantics_manager/semantic_color_lut.h>)
#define TDR_HAVE_SEMANTICS_MANAGER 1
#endif
#endif

#ifdef TDR_HAVE_EIGEN
#include <Eigen/Dense>
namespace Eigen {
typedef Array<uint8_t, Dynamic, Dynamic> ArrayXXc;  // include/top_down_render/top_down_map.h:23
}
#else
namespace Eigen {
typedef std::ptrdiff_t Index;
template <class T>
class DenseShim {
 public:
  DenseShim() : rows_(0), cols_(0) {}
  DenseShim(Index r, Index c) : rows_(r), cols_(c), d_((size_t)r * c) {}
  Index rows() const { return rows_; }
  Index cols() const { return cols_; }
  Index size() const { return rows_ * cols_; }
  T* data() { return d_.data(); }
  const T* data() const { return d_.data(); }
  T& operator()(Index i, Index j) { return d_[(size_t)i + (size_t)rows_ * j]; }
  const T& operator()(Index i, Index j) const { return d_[(size_t)i + (size_t)rows_ * j]; }
  T& operator()(Index k) { return d_[(size_t)k]; }
  const T& operator()(Index k) const { return d_[(size_t)k]; }
  void resize(Index r, Index c) { rows_ = r; cols_ = c; d_.assign((size_t)r * c, T()); }
  void setZero() { std::fill(d_.begin(), d_.end(), T()); }
 private:
  Index rows_, cols_;
  std::vector<T> d_;
};
typedef DenseShim<float> ArrayXXf;
typedef DenseShim<uint8_t> ArrayXXc;
template <class T, int N>
class FixedVecShim {
 public:
  FixedVecShim() { for (int i = 0; i < N; i++) v_[i] = T(); }
  FixedVecShim(T a, T b) { static_assert(N == 2, "2 components"); v_[0] = a; v_[1] = b; }
  FixedVecShim(T a, T b, T c) { static_assert(N == 3, "3 components"); v_[0] = a; v_[1] = b; v_[2] = c; }
  FixedVecShim(T a, T b, T c, T d) { static_assert(N == 4, "4 components"); v_[0] = a; v_[1] = b; v_[2] = c; v_[3] = d; }
  T& operator[](int i) { return v_[i]; }
  const T& operator[](int i) const { return v_[i]; }
  T& operator()(int i) { return v_[i]; }
  const T& operator()(int i) const { return v_[i]; }
  T x() const { return v_[0]; }
  T y() const { return v_[1]; }
  T* data() { return v_; }
 private:
  T v_[N];
};
typedef FixedVecShim<float, 2> Vector2f;
typedef FixedVecShim<int, 2> Vector2i;
typedef FixedVecShim<float, 3> Vector3f;
typedef FixedVecShim<float, 4> Vector4f;
template <int N>
class SquareMatShim {  // column-major like Eigen's default
 public:
  SquareMatShim() { setZero(); }
  float& operator()(int i, int j) { return v_[i + N * j]; }
  const float& operator()(int i, int j) const { return v_[i + N * j]; }
  void setZero() { for (float& x : v_) x = 0; }
  float* data() { return v_; }
 private:
  float v_[N * N];
};
typedef SquareMatShim<3> Matrix3f;
typedef SquareMatShim<4> Matrix4f;
class VectorXi {
 public:
  VectorXi() {}
  explicit VectorXi(Index n) : d_((size_t)n) {}
  static VectorXi Constant(Index n, int v) { VectorXi r(n); std::fill(r.d_.begin(), r.d_.end(), v); return r; }
  Index size() const { return (Index)d_.size(); }
  int& operator[](Index i) { return d_[(size_t)i]; }
  const int& operator[](Index i) const { return d_[(size_t)i]; }
  const int* data() const { return d_.data(); }
 private:
  std::vector<int> d_;
};
}  // namespace Eigen
#endif

#ifdef TDR_HAVE_PCL
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#else
namespace pcl {
struct alignas(16) PointXYZI {
  float x, y, z, pad_;
  float intensity, pad2_[3];
};
template <class PointT>
class PointCloud {
 public:
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  std::vector<PointT> points;
  uint32_t width = 0, height = 0;
  void push_back(const PointT& p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
  size_t size() const { return points.size(); }
};
}  // namespace pcl
#endif
// cv::Mat appears in two signatures of the reference's surface (updateMap, visualize).  Without OpenCV a minimal
// stand-in carries what those methods need: an 8-bit single-channel image (rows x cols, row 0 = top, `step` bytes per row).
#ifdef TDR_HAVE_OPENCV
#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>
#else
namespace cv {
class Mat {
 public:
  Mat() {}
  Mat(int r, int c, uint8_t* d, size_t row_bytes = 0) : rows(r), cols(c), data(d), step(row_bytes ? row_bytes : (size_t)c) {}
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  size_t step = 0;
  bool isContinuous() const { return step == (size_t)cols; }
  bool empty() const { return !data || rows < 1 || cols < 1; }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
};
}  // namespace cv
#endif

// TopDownMap::Params::color_lut (top_down_map.h:56) is a semantics_manager type used only by the static-map loader
// (SVG / colour PNG -> class image), which is outside the per-scan path.  The member exists so that
// `params.color_lut = class_params.color_lut` (src/top_down_render.cpp:173) compiles unchanged.
#ifdef TDR_HAVE_SEMANTICS_MANAGER
#include <semantics_manager/semantic_color_lut.h>
#else
class SemanticColorLut {};
#endif

static_assert(sizeof(pcl::PointXYZI) == 32, "pcl::PointXYZI is 32 bytes (x,y,z,pad,intensity,pad,pad,pad)");

#endif  // TOP_DOWN_RENDER_TDR_COMPAT_H_
