// MINIMAL STAND-IN for <opencv2/core.hpp> — test infrastructure only (tests/test_facade.py compiles
// include/top_down_render/particle_viz.h against it: this image has no OpenCV).  It declares what the drawing code and the
// class surface use — cv::Mat as an 8-bit single-channel image, Point / Size / Scalar — and RECORDS drawing calls instead of
// rasterising them.  Never shipped, never on the product's include path.
#ifndef TDR_TEST_OPENCV_STUB_CORE_HPP_
#define TDR_TEST_OPENCV_STUB_CORE_HPP_
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
namespace cv {
struct Point {
  int x = 0, y = 0;
  Point() {}
  Point(int x_, int y_) : x(x_), y(y_) {}
};
inline Point operator+(const Point& a, const Point& b) { return Point(a.x + b.x, a.y + b.y); }
inline Point operator-(const Point& a, const Point& b) { return Point(a.x - b.x, a.y - b.y); }
struct Size {
  int width = 0, height = 0;
  Size() {}
  Size(int w, int h) : width(w), height(h) {}
};
inline Size operator*(const Size& s, int k) { return Size(s.width * k, s.height * k); }
struct Scalar {
  double v[4];
  Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {}
};
enum { LINE_AA = 16 };
struct DrawCall {
  std::string what;
  Point a, b;
  double c0, c1, c2;
};
class Mat {
 public:
  Mat() {}
  Mat(int r, int c, uint8_t* d, size_t row_bytes = 0) : rows(r), cols(c), data(d), step(row_bytes ? row_bytes : (size_t)c) {}
  int rows = 0, cols = 0;
  uint8_t* data = nullptr;
  size_t step = 0;
  std::vector<DrawCall> drawn;   // (the stand-in's record of what was drawn)
  Size size() const { return Size(cols, rows); }
  bool isContinuous() const { return step == (size_t)cols; }
  bool empty() const { return !data || rows < 1 || cols < 1; }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
};
}  // namespace cv
#endif
