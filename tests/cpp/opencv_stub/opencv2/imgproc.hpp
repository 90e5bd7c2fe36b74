// MINIMAL STAND-IN for <opencv2/imgproc.hpp> — see core.hpp beside it: the three drawing calls, recorded.
#ifndef TDR_TEST_OPENCV_STUB_IMGPROC_HPP_
#define TDR_TEST_OPENCV_STUB_IMGPROC_HPP_
#include "opencv2/core.hpp"
namespace cv {
inline void circle(Mat& img, Point c, int, const Scalar& col, int = 1) { img.drawn.push_back({"circle", c, c, col.v[0], col.v[1], col.v[2]}); }
inline void arrowedLine(Mat& img, Point a, Point b, const Scalar& col, int = 1, int = 8, int = 0, double = 0.1) {
  img.drawn.push_back({"arrow", a, b, col.v[0], col.v[1], col.v[2]});
}
inline void ellipse(Mat& img, Point c, Size axes, double, double, double, const Scalar& col, int = 1) {
  img.drawn.push_back({"ellipse", c, Point(axes.width, axes.height), col.v[0], col.v[1], col.v[2]});
}
}  // namespace cv
#endif
