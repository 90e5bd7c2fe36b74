// particle_viz.h — the default drawing behind ParticleFilter::visualize(cv::Mat&) when OpenCV is present and the host set
// no hook of its own (reference: src/particle_filter.cpp:373-423, call site src/top_down_render.cpp:431).  Included by
// particle_filter.h under TDR_HAVE_OPENCV only; it draws from the host SNAPSHOT the filter hands over (states, the
// mixture of the last computeGMM, the max-likelihood state) and touches no device memory.
//   particles        red arrows of +-5 px along the heading at (x, H - y); a particle outside the image: a green dot on
//                    the border
//   mixture          per component a blue ellipse of the position covariance (axes 2 sqrt(eigenvalue), turned by the first
//                    eigenvector) and a blue arrow along its mean heading
//   best particle    a blue arrow
// The eigen-decomposition of the symmetric 2 x 2 block is written out in closed form (ascending eigenvalues, like
// Eigen::SelfAdjointEigenSolver orders them).  tests/test_facade.py compiles this file against a minimal OpenCV
// stand-in and checks what it draws.
#ifndef TOP_DOWN_RENDER_PARTICLE_VIZ_H_
#define TOP_DOWN_RENDER_PARTICLE_VIZ_H_
#ifdef TDR_HAVE_OPENCV

#include <algorithm>
#include <cmath>

inline void ParticleFilter::drawSnapshot(cv::Mat& img, const Snapshot& snap) {
  const int H = img.size().height, W = img.size().width;
  auto arrow = [&](const cv::Point& at, float theta, const cv::Scalar& colour) {
    const cv::Point dir((int)(std::cos(theta) * 5), (int)(-std::sin(theta) * 5));
    cv::arrowedLine(img, at - dir, at + dir, colour, 2, cv::LINE_AA, 0, 0.3);
  };
  for (const State& p : snap.particles) {
    const float x = p.dx_m * p.scale + p.init_x_px, y = p.dy_m * p.scale + p.init_y_px;   // mlState (state_particle.cpp:98-102)
    cv::Point pt((int)x, (int)((float)H - y));
    if (pt.x < 0 || pt.x > W || pt.y < 0 || pt.y > H) {
      pt.x = std::min(std::max(pt.x, 5), W - 5);
      pt.y = std::min(std::max(pt.y, 5), H - 5);
      cv::circle(img, pt, 2, cv::Scalar(0, 255, 0), -1);
    } else {
      arrow(pt, p.theta, cv::Scalar(0, 0, 255));
    }
  }
  for (size_t i = 0; i < snap.gmm_means.size() && i < snap.gmm_covs.size(); i++) {
    const float a = snap.gmm_covs[i](0, 0), b = snap.gmm_covs[i](0, 1), d = snap.gmm_covs[i](1, 1);
    const float tr = a + d, disc = std::sqrt(std::max(0.f, (a - d) * (a - d) / 4 + b * b));
    const float l0 = tr / 2 - disc, l1 = tr / 2 + disc;
    if (l0 < 0 || l1 < 0) break;                       // (the reference stops at the first component that is not PSD)
    float vx = b, vy = l0 - a;                         // eigenvector of the smaller eigenvalue
    if (std::fabs(vx) + std::fabs(vy) < 1e-12f) { vx = 1; vy = 0; }
    const float angle = std::atan2(-vy, vx);
    const cv::Point center((int)snap.gmm_means[i][0], (int)((float)H - snap.gmm_means[i][1]));
    cv::ellipse(img, center, cv::Size((int)std::sqrt(l0), (int)std::sqrt(l1)) * 2, angle * 180 / M_PI, 0, 360,
                cv::Scalar(255, 0, 0), 2);
    arrow(center, snap.gmm_means[i][2], cv::Scalar(255, 0, 0));
  }
  if (snap.have_best) arrow(cv::Point((int)snap.best[0], (int)((float)H - snap.best[1])), snap.best[2], cv::Scalar(255, 0, 0));
}

#endif  // TDR_HAVE_OPENCV
#endif  // TOP_DOWN_RENDER_PARTICLE_VIZ_H_
