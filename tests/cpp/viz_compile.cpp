// viz_compile.cpp — the default drawing of ParticleFilter::visualize (include/top_down_render/particle_viz.h, reference
// src/particle_filter.cpp:373-423) compiled against the minimal OpenCV stand-in under tests/cpp/opencv_stub and run on a
// hand-made snapshot: no GPU involved (drawSnapshot is a static function of host data).
#include <cstdio>

#include "top_down_render/particle_filter.h"

int main() {
#ifndef TDR_HAVE_OPENCV
  std::puts("built without the OpenCV stand-in");
  return 2;
#else
  uint8_t px[100 * 80] = {0};
  cv::Mat img(80, 100, px);
  ParticleFilter::Snapshot snap;
  State a;                      // inside the image: an arrow
  a.init_x_px = 50; a.init_y_px = 30; a.theta = 0.f; a.scale = 1.f;
  State b = a;                  // outside: a green dot on the border
  b.init_x_px = -20;
  snap.particles = {a, b};
  snap.gmm_means.push_back(Eigen::Vector3f(40.f, 20.f, 1.5707964f));
  Eigen::Matrix3f cov;
  cov(0, 0) = 16.f; cov(1, 1) = 4.f;            // axis-aligned: eigenvalues 4 (along y) and 16 (along x)
  snap.gmm_covs.push_back(cov);
  snap.have_best = true;
  snap.best = Eigen::Vector4f(10.f, 70.f, 3.1415927f, 1.f);
  ParticleFilter::drawSnapshot(img, snap);
  // arrow of particle a: centre (50, 80 - 30), +-5 px along x, red (BGR 0, 0, 255)
  const auto& d = img.drawn;
  bool ok = d.size() == 5;
  ok = ok && d[0].what == "arrow" && d[0].a.x == 45 && d[0].b.x == 55 && d[0].a.y == 50 && d[0].c2 == 255;
  ok = ok && d[1].what == "circle" && d[1].a.x == 5 && d[1].c1 == 255;
  ok = ok && d[2].what == "ellipse" && d[2].a.x == 40 && d[2].a.y == 60 && d[2].b.x == 4 && d[2].b.y == 8;   // 2 sqrt(4), 2 sqrt(16)
  ok = ok && d[3].what == "arrow" && d[3].a.x == 40 && d[3].a.y == 65 && d[3].b.y == 55 && d[3].c0 == 255;   // heading +90 deg: up the image
  ok = ok && d[4].what == "arrow" && d[4].a.x == 15 && d[4].b.x == 5 && d[4].a.y == 10;                       // the best particle, heading 180 deg
  std::printf("drawn %zu primitives: %s\n", d.size(), ok ? "as expected" : "UNEXPECTED");
  return ok ? 0 : 1;
#endif
}
