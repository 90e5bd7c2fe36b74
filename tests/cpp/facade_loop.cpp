// facade_loop.cpp — the node's per-scan loop through the drop-in classes: TopDownRenderCore::takeStep
// (include/top_down_render/top_down_render_core.h) = TopDownRender::takeStep + updateFilter + publishPoseEst of the
// reference (src/top_down_render.cpp:505-560, 413-425, 331-365), `steps` scans in a row.  The range scale — the `res` of
// the render and of the score — moves on every step like in the node.  Inputs / outputs: raw little-endian files in
// argv[1] (tests/test_takestep_loop.py writes them and compares every step with the CPU oracle's loop).
//   TDR_FACADE_BENCH=<steps>: times takeStep with the node's range-scale stepping and with a fixed range scale.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "top_down_render/top_down_render_core.h"

template <class T>
static std::vector<T> slurp(const std::string& path) {
  std::ifstream in(path, std::ios::binary | std::ios::ate);
  if (!in) throw std::runtime_error("cannot open " + path);
  std::vector<T> v((size_t)in.tellg() / sizeof(T));
  in.seekg(0);
  in.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
  return v;
}
template <class T>
static void dump(const std::string& path, const std::vector<T>& v) {
  std::ofstream out(path, std::ios::binary | std::ios::trunc);
  out.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s <dir>\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  try {
    int ncls, rows, cols, nb, nr, npts, nclouds, npart, steps, device_scan;
    unsigned seed;
    float rs_min, rs_max, target_unc, fixed_scale, map_resolution;
    {
      std::ifstream meta(dir + "/meta.txt");
      meta >> ncls >> rows >> cols >> nb >> nr >> npts >> nclouds >> npart >> seed >> steps >> rs_min >> rs_max >>
          target_unc >> fixed_scale >> map_resolution >> device_scan;
      if (!meta) throw std::runtime_error("bad meta.txt");
    }
    auto maps = slurp<float>(dir + "/maps.bin");
    auto mask = slurp<uint8_t>(dir + "/mask.bin");
    auto pts = slurp<float>(dir + "/pts.bin");        // [nclouds][npts][8]: pcl::PointXYZI layout
    auto st_in = slurp<State>(dir + "/states.bin");
    auto motion = slurp<float>(dir + "/motion.bin");  // [steps][3]: trans x, trans y, yaw of the projected prior

    TopDownMap::Params map_params;
    map_params.num_classes = ncls;
    map_params.resolution = map_resolution;
    for (int c = 0; c < ncls; c++) map_params.flatten_lut.push_back(c);
    TopDownMapPolar map(map_params);
    {
      std::vector<Eigen::ArrayXXf> class_maps;
      for (int c = 0; c < ncls; c++) {
        Eigen::ArrayXXf m(rows, cols);
        std::memcpy(m.data(), maps.data() + (size_t)c * rows * cols, (size_t)rows * cols * sizeof(float));
        class_maps.push_back(m);
      }
      Eigen::ArrayXXc class_mask(rows, cols);
      std::memcpy(class_mask.data(), mask.data(), (size_t)rows * cols);
      map.setDistanceMaps(class_maps, class_mask);
    }
    FilterParams filter_params;
    filter_params.pos_cov = 0.3f;
    filter_params.theta_cov = (float)(M_PI / 100);
    filter_params.regularization = 0.15f;
    filter_params.fixed_scale = fixed_scale;
    for (int c = 0; c < ncls; c++) filter_params.class_weights.push_back(1.f);
    filter_params.init_pos_m_x = 1e9f;   // initializeParticles returns early: the test brings its own particle set
    filter_params.init_pos_m_y = 1e9f;
    Eigen::VectorXi flatten_lut = Eigen::VectorXi::Constant(256, -1);
    for (int c = 0; c < ncls; c++) flatten_lut[c] = c;

    std::vector<pcl::PointCloud<PointType>::Ptr> clouds;
    for (int k = 0; k < nclouds; k++) {
      pcl::PointCloud<PointType>::Ptr cloud_ptr(new pcl::PointCloud<PointType>());
      for (int i = 0; i < npts; i++) {
        const float* q = pts.data() + ((size_t)k * npts + i) * 8;
        PointType p{};
        p.x = q[0]; p.y = q[1]; p.z = q[2]; p.intensity = q[4];
        cloud_ptr->push_back(p);
      }
      clouds.push_back(cloud_ptr);
    }

    auto make_core = [&](float lo, float hi) {
      TopDownRenderCore::Config cfg;
      cfg.particle_count = npart;
      cfg.range_scale_min = lo;
      cfg.range_scale_max = hi;
      cfg.target_uncertainty_m = target_unc;
      cfg.theta_bins = nb;
      cfg.range_bins = nr;
      cfg.seed = seed;
      auto* core = new TopDownRenderCore(cfg);
      core->initialize(&map, filter_params, flatten_lut);
      core->setDeviceScan(device_scan != 0);
      core->filter()->setStates(st_in);
      return core;
    };

    {
      TopDownRenderCore* core = make_core(rs_min, rs_max);
      std::vector<float> raw_all, est_all;
      std::vector<int32_t> idx_all;
      std::vector<State> st_all;
      for (int k = 0; k < steps; k++) {
        TopDownRenderCore::PoseEst e;
        const bool ran = core->takeStep(clouds[(size_t)k % clouds.size()], Eigen::Vector2f(motion[3 * k], motion[3 * k + 1]),
                                        motion[3 * k + 2], &e);
        if (!ran) throw std::runtime_error("takeStep skipped: no map");
        auto raw = core->filter()->rawWeights(npart);
        raw_all.insert(raw_all.end(), raw.begin(), raw.end());
        auto idx = core->filter()->resampleIndices();
        idx_all.insert(idx_all.end(), idx.begin(), idx.end());
        auto st = core->filter()->states();
        st_all.insert(st_all.end(), st.begin(), st.end());
        // [res used, range scale after, froze, converged, scale at :335, frozen now, cov 16, ml 4]
        est_all.push_back(core->lastRes());
        est_all.push_back(e.range_scale);
        est_all.push_back(e.froze_scale ? 1.f : 0.f);
        est_all.push_back(e.converged ? 1.f : 0.f);
        est_all.push_back(e.scale);
        est_all.push_back(core->filter()->isScaleFrozen() ? 1.f : 0.f);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) est_all.push_back(e.cov(i, j));
        for (int i = 0; i < 4; i++) est_all.push_back(e.ml_state[i]);
      }
      dump(dir + "/out_raw.bin", raw_all);
      dump(dir + "/out_idx.bin", idx_all);
      dump(dir + "/out_states.bin", st_all);
      dump(dir + "/out_est.bin", est_all);
      delete core;
    }

    if (const char* env = std::getenv("TDR_FACADE_BENCH")) {
      const int bsteps = std::atoi(env);
      auto time_loop = [&](float lo, float hi, const char* what) {
        TopDownRenderCore* core = make_core(lo, hi);
        if (const char* dr = std::getenv("TDR_FACADE_DEVICE_RNG")) core->filter()->configure(std::atoi(dr) == 0, 1);
        Eigen::Vector2f t(motion[0], motion[1]);
        int distinct = 0;
        float last = -1.f;
        for (int i = 0; i < 10; i++) core->takeStep(clouds[0], t, motion[2]);
        // takeStep alone is timed (it ends with publishPoseEst's read-back, i.e. synchronised); resetting the particle set —
        // every step scores the same set, a converging one gets cheaper — stays outside
        double ms = 0;
        for (int i = 0; i < bsteps; i++) {
          core->filter()->setStates(st_in);
          const auto t0 = std::chrono::steady_clock::now();
          core->takeStep(clouds[(size_t)i % clouds.size()], t, motion[2]);
          ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
          if (core->lastRes() != last) { distinct++; last = core->lastRes(); }
        }
        std::printf("facade_loop_bench %s steps=%d particles=%d distinct_res=%d ms_per_step=%.4f\n", what, bsteps, npart,
                    distinct, ms / bsteps);
        delete core;
      };
      time_loop(rs_min, rs_max, "varying_res");
      time_loop(rs_max, rs_max, "fixed_res");
    }
    std::puts("facade_loop ok");
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "facade_loop failed: %s\n", e.what());
    return 1;
  }
}
