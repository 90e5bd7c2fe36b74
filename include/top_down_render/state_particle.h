// State / FilterParams — the reference's PODs (include/top_down_render/state_particle.h:9-38), unchanged — and
// StateParticle.
// Inside ParticleFilter the per-particle methods (propagate, computeWeight, src/state_particle.cpp:57-78,157-219) run
// batched on the GPU over a structure of arrays; ParticleFilter::states() returns the particles as `State`s.  The
// StateParticle class below keeps the reference's per-particle surface for code that holds single particles: it is a
// one-particle filter on the device (a launch per call — use ParticleFilter for throughput).
#ifndef STATE_PARTICLE_H_
#define STATE_PARTICLE_H_

#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "tdr.h"
#include "top_down_render/top_down_map_polar.h"

typedef struct State {
  float init_x_px = 0;
  float init_y_px = 0;
  float dx_m = 0;
  float dy_m = 0;
  float theta = 0;
  float scale = 1;  // px/m
  bool have_init = false;
} State;
static_assert(sizeof(State) == sizeof(tdr_state), "State must stay byte-compatible with tdr_state (28 bytes)");

typedef struct FilterParams {
  float pos_cov;
  float theta_cov;
  float regularization;
  float init_pos_px_x = -1;
  float init_pos_px_y = -1;
  float init_pos_px_cov = -1;

  float init_pos_m_x = -1;
  float init_pos_m_y = -1;
  float init_pos_deg_theta = -1;
  float init_pos_deg_cov = -1;

  bool force_on_map = false;
  float fixed_scale = -1;
  float scale_log_min = -0.1;
  float scale_log_max = 1;

  std::vector<float> class_weights;
} FilterParams;

inline tdr_filter_params to_tdr_params(const FilterParams& p, int num_classes) {
  tdr_filter_params c{};
  c.pos_cov = p.pos_cov; c.theta_cov = p.theta_cov; c.regularization = p.regularization;
  c.init_pos_px_x = p.init_pos_px_x; c.init_pos_px_y = p.init_pos_px_y; c.init_pos_px_cov = p.init_pos_px_cov;
  c.init_pos_m_x = p.init_pos_m_x; c.init_pos_m_y = p.init_pos_m_y;
  c.init_pos_deg_theta = p.init_pos_deg_theta; c.init_pos_deg_cov = p.init_pos_deg_cov;
  c.force_on_map = p.force_on_map ? 1 : 0;
  c.fixed_scale = p.fixed_scale; c.scale_log_min = p.scale_log_min; c.scale_log_max = p.scale_log_max;
  c.num_classes = num_classes;
  for (int i = 0; i < 16; i++) c.class_weights[i] = i < (int)p.class_weights.size() ? p.class_weights[i] : (i < num_classes ? 1.f : 0.f);
  return c;
}

// include/top_down_render/state_particle.h:40-66 in the reference.
class StateParticle {
 public:
  // src/state_particle.cpp:3-49: init == true draws the pose from the shared generator (rejection-sampled onto a
  // road cell), init == false leaves the default State.
  StateParticle(std::mt19937* gen, TopDownMapPolar* map, FilterParams* params, bool init = true)
      : map_(map), params_(params) {
    if (!gen || !map || !params) throw std::invalid_argument("StateParticle: null argument");
    tdr_filter_params c = to_tdr_params(*params_, map_->numClasses());
    if (tdr_filter_create(map_->handle(), 1, &c, /*seed (unused: the generator is shared)*/ 1, &f_) != TDR_OK) fail("StateParticle");
    try {
      check(tdr_filter_share_rng(f_, gen), "share_rng");
      check(tdr_filter_configure(f_, /*parity_rng=*/1, /*locality_every=*/0), "configure");
      if (init && map_->haveMap()) check(tdr_filter_init_one(f_), "init");
      else setState(State());
    } catch (...) {
      tdr_filter_destroy(f_);   // the destructor does not run for a constructor that throws
      throw;
    }
  }
  ~StateParticle() { tdr_filter_destroy(f_); }
  StateParticle(const StateParticle&) = delete;
  StateParticle& operator=(const StateParticle&) = delete;

  void propagate(Eigen::Vector2f& trans, float omega, bool scale_freeze = false) {   // :57-78
    check(tdr_filter_propagate_freeze(f_, trans[0], trans[1], omega, scale_freeze ? 1 : 0), "propagate");
  }
  State state() const {
    State s;
    check(tdr_filter_get_states(f_, reinterpret_cast<tdr_state*>(&s), 1), "state");
    return s;
  }
  Eigen::Vector4f mlState() const {                                                   // :98-102
    const State s = state();
    return Eigen::Vector4f(s.dx_m * s.scale + s.init_x_px, s.dy_m * s.scale + s.init_y_px, s.theta, s.scale);
  }
  void setState(const State& s) { check(tdr_filter_set_states(f_, reinterpret_cast<const tdr_state*>(&s), 1), "setState"); }
  // :157-219.  top_down_geo is accepted and ignored like in the reference's score (:145-152).
  void computeWeight(std::vector<Eigen::ArrayXXf>& top_down_scan, std::vector<Eigen::ArrayXXf>& /*top_down_geo*/, float res) {
    const int ncls = map_->numClasses();
    // (sizes it cannot use: a silent return like the reference's per-scan calls, the reason in tdr_last_error())
    if ((int)top_down_scan.size() < ncls) { tdr_set_error(TDR_ERR_ARG, "computeWeight: fewer scan images than map classes"); return; }
    const Eigen::Vector2i shape = map_->polarShape();
    const size_t P = (size_t)shape[0] * shape[1];
    for (int c = 0; c < ncls; c++)
      if (top_down_scan[c].rows() != shape[0] || top_down_scan[c].cols() != shape[1])
        { tdr_set_error(TDR_ERR_ARG, "computeWeight: scan image shape differs from the shape given to samplePtsPolar"); return; }
    std::vector<float> buf(P * ncls);
    for (int c = 0; c < ncls; c++) std::memcpy(buf.data() + P * c, top_down_scan[c].data(), P * sizeof(float));
    check(tdr_filter_compute_weights(f_, buf.data(), nullptr, res), "computeWeight");
  }
  float weight() const {                                                              // :55
    float w = 0;
    check(tdr_filter_get_raw_weights(f_, &w, 1), "weight");
    return w;
  }
  float lastDist() const {
    float d = 0;
    check(tdr_filter_get_last_dist(f_, &d, 1), "lastDist");
    return d;
  }
  void setScale(float scale) {                                                        // :104-106
    State s = state();
    s.scale = scale;
    setState(s);
  }
  void updateSize() {}   // :108-110 caches the map size in metres; here the gate reads the map at score time

 private:
  void check(int rc, const char* what) const { if (rc != TDR_OK) fail(what); }
  [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("StateParticle::") + what + ": " + tdr_last_error()); }
  TopDownMapPolar* map_;
  FilterParams* params_;
  tdr_filter* f_ = nullptr;
};

#endif  // STATE_PARTICLE_H_
