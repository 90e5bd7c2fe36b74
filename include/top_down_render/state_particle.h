// State / FilterParams — the reference's PODs (include/top_down_render/state_particle.h:9-38), unchanged.
// StateParticle's per-particle methods (propagate, computeWeight, src/state_particle.cpp:57-78,157-219) run batched on
// the GPU inside ParticleFilter::propagate / ::update; particles are a structure of arrays on the device, so there is
// no per-particle object to hand out.  ParticleFilter::states() returns them as `State`s.
#ifndef STATE_PARTICLE_H_
#define STATE_PARTICLE_H_

#include <random>
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

#endif  // STATE_PARTICLE_H_
