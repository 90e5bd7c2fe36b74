// tdr_gmm.cpp — Gaussian-mixture fit for the adaptive particle count.
//
// The reference fits cv::ml::EM (full covariances) to <= 1000 strided particle samples (x, y, 50 cos θ, 50 sin θ) once
// per second in a detached thread, moves the cluster count by ±1 when the mean log-likelihood changes by more than 0.3,
// and turns the clusters' xy covariances into the next particle count (src/particle_filter.cpp:151-157, 245-318).
// OpenCV is not available here and cv::ml::EM starts from a randomised k-means, so its numbers cannot be matched
// ("parity unpinned" for this row).  This is a DETERMINISTIC fit with the same model and the same selection rule:
//   * seeding: the sample nearest to the overall mean, then farthest-first traversal (lowest index wins ties);
//   * 10 Lloyd iterations; cluster weights / means / covariances from the partition;
//   * EM, full 4x4 covariances + 1e-6 I, at most 100 iterations, stops when the mean log-likelihood moves < 1e-6.
// A few thousand flops on <= 1000 x 4 doubles: host code, like the reference's (tests/test_gmm.py checks it against
// an independent NumPy statement of the same algorithm).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "tdr.h"

namespace {

constexpr int D = 4;
constexpr double REG = 1e-6;

struct Gauss {
  double w;
  double mu[D];
  double cov[D][D];
  double L[D][D];   // Cholesky factor of cov
  double logdet;
  bool ok;
};

bool cholesky(Gauss& g) {
  std::memset(g.L, 0, sizeof(g.L));
  double ld = 0;
  for (int i = 0; i < D; i++) {
    for (int j = 0; j <= i; j++) {
      double s = g.cov[i][j];
      for (int k = 0; k < j; k++) s -= g.L[i][k] * g.L[j][k];
      if (i == j) {
        if (!(s > 0)) return false;
        g.L[i][i] = std::sqrt(s);
        ld += std::log(s);
      } else {
        g.L[i][j] = s / g.L[j][j];
      }
    }
  }
  g.logdet = ld;
  return true;
}

double log_pdf(const Gauss& g, const double* x) {
  double y[D];
  double maha = 0;
  for (int i = 0; i < D; i++) {   // forward substitution L y = x - mu
    double s = x[i] - g.mu[i];
    for (int k = 0; k < i; k++) s -= g.L[i][k] * y[k];
    y[i] = s / g.L[i][i];
    maha += y[i] * y[i];
  }
  return std::log(g.w) - 0.5 * (D * std::log(2 * M_PI) + g.logdet + maha);
}

void params_from_resp(const double* X, int m, const std::vector<double>& resp, int k, std::vector<Gauss>& gs) {
  for (int c = 0; c < k; c++) {
    double nk = 0;
    for (int i = 0; i < m; i++) nk += resp[(size_t)i * k + c];
    if (!(nk > 1e-10)) continue;   // empty cluster: keep its previous parameters
    Gauss g = gs[c];
    g.w = nk / m;
    for (int d = 0; d < D; d++) {
      double s = 0;
      for (int i = 0; i < m; i++) s += resp[(size_t)i * k + c] * X[(size_t)i * D + d];
      g.mu[d] = s / nk;
    }
    for (int a = 0; a < D; a++)
      for (int b = 0; b <= a; b++) {
        double s = 0;
        for (int i = 0; i < m; i++)
          s += resp[(size_t)i * k + c] * (X[(size_t)i * D + a] - g.mu[a]) * (X[(size_t)i * D + b] - g.mu[b]);
        g.cov[a][b] = g.cov[b][a] = s / nk + (a == b ? REG : 0.0);
      }
    if (cholesky(g)) { g.ok = true; gs[c] = g; }
  }
}

}  // namespace

extern "C" int tdr_gmm_fit_host(const double* samples, int m, int k, int max_iter, double* weights, double* means,
                                double* covs, double* mean_loglik) {
  if (!samples || m < 1 || k < 1 || k > m || !weights || !means || !covs)
    return tdr_set_error(TDR_ERR_ARG, "gmm_fit: bad arguments");
  const double* X = samples;
  // ---- seeding: nearest to the overall mean, then farthest-first
  double mean[D] = {0, 0, 0, 0};
  for (int i = 0; i < m; i++)
    for (int d = 0; d < D; d++) mean[d] += X[(size_t)i * D + d];
  for (int d = 0; d < D; d++) mean[d] /= m;
  auto dist2 = [&](const double* a, const double* b) {
    double s = 0;
    for (int d = 0; d < D; d++) s += (a[d] - b[d]) * (a[d] - b[d]);
    return s;
  };
  std::vector<int> centre_idx;
  {
    int best = 0;
    double bd = std::numeric_limits<double>::infinity();
    for (int i = 0; i < m; i++) {
      const double d2 = dist2(X + (size_t)i * D, mean);
      if (d2 < bd) { bd = d2; best = i; }
    }
    centre_idx.push_back(best);
  }
  std::vector<double> mind(m, std::numeric_limits<double>::infinity());
  while ((int)centre_idx.size() < k) {
    const double* c = X + (size_t)centre_idx.back() * D;
    int best = 0;
    double bd = -1;
    for (int i = 0; i < m; i++) {
      mind[i] = std::min(mind[i], dist2(X + (size_t)i * D, c));
      if (mind[i] > bd) { bd = mind[i]; best = i; }
    }
    centre_idx.push_back(best);
  }
  std::vector<double> cen((size_t)k * D);
  for (int c = 0; c < k; c++) std::memcpy(&cen[(size_t)c * D], X + (size_t)centre_idx[c] * D, D * sizeof(double));
  // ---- Lloyd
  std::vector<int> label(m, 0);
  for (int it = 0; it < 10; it++) {
    for (int i = 0; i < m; i++) {
      int best = 0;
      double bd = std::numeric_limits<double>::infinity();
      for (int c = 0; c < k; c++) {
        const double d2 = dist2(X + (size_t)i * D, &cen[(size_t)c * D]);
        if (d2 < bd) { bd = d2; best = c; }
      }
      label[i] = best;
    }
    for (int c = 0; c < k; c++) {
      double s[D] = {0, 0, 0, 0};
      int cnt = 0;
      for (int i = 0; i < m; i++)
        if (label[i] == c) {
          for (int d = 0; d < D; d++) s[d] += X[(size_t)i * D + d];
          cnt++;
        }
      if (cnt > 0)
        for (int d = 0; d < D; d++) cen[(size_t)c * D + d] = s[d] / cnt;
    }
  }
  // ---- initial mixture from the hard partition
  std::vector<Gauss> gs(k);
  for (int c = 0; c < k; c++) {
    Gauss& g = gs[c];
    g.w = 1.0 / k;
    std::memcpy(g.mu, &cen[(size_t)c * D], sizeof(g.mu));
    std::memset(g.cov, 0, sizeof(g.cov));
    for (int d = 0; d < D; d++) g.cov[d][d] = 1.0;
    g.ok = cholesky(g);
  }
  std::vector<double> resp((size_t)m * k, 0.0);
  for (int i = 0; i < m; i++) resp[(size_t)i * k + label[i]] = 1.0;
  params_from_resp(X, m, resp, k, gs);
  // ---- EM
  std::vector<double> lp(k);
  double prev = -std::numeric_limits<double>::infinity(), ll = prev;
  for (int it = 0; it < std::max(1, max_iter); it++) {
    double tot = 0;
    for (int i = 0; i < m; i++) {
      double mx = -std::numeric_limits<double>::infinity();
      for (int c = 0; c < k; c++) {
        lp[c] = log_pdf(gs[c], X + (size_t)i * D);
        mx = std::max(mx, lp[c]);
      }
      double se = 0;
      for (int c = 0; c < k; c++) se += std::exp(lp[c] - mx);
      const double lse = mx + std::log(se);
      for (int c = 0; c < k; c++) resp[(size_t)i * k + c] = std::exp(lp[c] - lse);
      tot += lse;
    }
    ll = tot / m;
    if (std::fabs(ll - prev) < 1e-6) break;
    prev = ll;
    params_from_resp(X, m, resp, k, gs);
  }
  for (int c = 0; c < k; c++) {
    weights[c] = gs[c].w;
    std::memcpy(means + (size_t)c * D, gs[c].mu, sizeof(gs[c].mu));
    std::memcpy(covs + (size_t)c * D * D, gs[c].cov, sizeof(gs[c].cov));
  }
  if (mean_loglik) *mean_loglik = ll;
  return TDR_OK;
}

// computeGMM's cluster-count search (src/particle_filter.cpp:259,276-297) around the current count, on the given samples.
// num_gaussians_io: in = current count, out = chosen count; means_out [k][3] = {x, y, atan2(m3, m2)} (:305-306),
// covs_out [k][9] row-major = xy block of the cluster covariance, 1 at (2,2) (:307-311).
extern "C" int tdr_gmm_select_host(const double* samples, int m, int64_t num_particles, int* num_gaussians_io,
                                   int max_k, float* means_out, float* covs_out) {
  if (!samples || m < 1 || !num_gaussians_io || max_k < 1 || !means_out || !covs_out)
    return tdr_set_error(TDR_ERR_ARG, "gmm_select: bad arguments");
  int k = std::max(1, std::min<int>((int)(num_particles / 20) + 1, *num_gaussians_io));  // :259
  k = std::min(k, std::min(max_k, m));
  std::vector<double> w((size_t)(k + 1)), mu((size_t)(k + 1) * D), cv((size_t)(k + 1) * D * D);
  double ll = 0, ll2 = 0;
  int rc = tdr_gmm_fit_host(samples, m, k, 100, w.data(), mu.data(), cv.data(), &ll);
  if (rc) return rc;
  int dir = 0;
  if ((int64_t)k * 50 < num_particles && k + 1 <= std::min(max_k, m)) {  // :280-286
    rc = tdr_gmm_fit_host(samples, m, k + 1, 100, w.data(), mu.data(), cv.data(), &ll2);
    if (rc) return rc;
    if (ll + 0.3 < ll2) dir = 1;
  }
  if (k > 1) {                                                            // :288-294
    rc = tdr_gmm_fit_host(samples, m, k - 1, 100, w.data(), mu.data(), cv.data(), &ll2);
    if (rc) return rc;
    if (ll - 0.3 < ll2) dir = -1;
  }
  k += dir;
  rc = tdr_gmm_fit_host(samples, m, k, 100, w.data(), mu.data(), cv.data(), &ll);
  if (rc) return rc;
  for (int c = 0; c < k; c++) {
    means_out[3 * c + 0] = (float)mu[(size_t)c * D + 0];
    means_out[3 * c + 1] = (float)mu[(size_t)c * D + 1];
    means_out[3 * c + 2] = (float)std::atan2(mu[(size_t)c * D + 3], mu[(size_t)c * D + 2]);
    float* o = covs_out + 9 * c;
    const double* s = &cv[(size_t)c * D * D];
    o[0] = (float)s[0]; o[1] = (float)s[1]; o[2] = 0.f;
    o[3] = (float)s[D]; o[4] = (float)s[D + 1]; o[5] = 0.f;
    o[6] = 0.f; o[7] = 0.f; o[8] = 1.f;
  }
  *num_gaussians_io = k;
  return TDR_OK;
}

// The adaptive particle count of src/particle_filter.cpp:151-157 from the clusters' covariances.
extern "C" int64_t tdr_adaptive_count_host(const float* covs, int k, int64_t last_count, int64_t max_count) {
  int64_t acc = 0;
  for (int c = 0; c < k; c++) {
    // eigenvalues of the symmetric 2x2 block (Eigen's .eigenvalues() of cov.block<2,2>(0,0), real parts)
    const float a = covs[9 * c + 0], b = covs[9 * c + 1], cc = covs[9 * c + 3], d = covs[9 * c + 4];
    const float tr = a + d, det = a * d - b * cc;
    const float disc = std::sqrt(std::max(0.f, tr * tr * 0.25f - det));
    const float e0 = tr * 0.5f - disc, e1 = tr * 0.5f + disc;
    acc += (int64_t)(int)(std::sqrt(std::max(0.f, e0)) * std::sqrt(std::max(0.f, e1)));  // area of the cov ellipse
  }
  return std::min<int64_t>(std::max<int64_t>(acc, 3 * last_count / 4 + 10), max_count);
}
