// facade_step.cpp — drives the reference's class surface (include/top_down_render/*.h) through one step exactly like
// TopDownRender::initialize + takeStep do (src/top_down_render.cpp:81,115-117,423-425,505-560 in the reference):
//   new TopDownMapPolar(params); map->samplePtsPolar(shape, ang_res); new ParticleFilter(N, map, filter_params);
//   new ScanRendererPolar(flatten_lut); renderer->renderSemanticTopDown(cloud, res, ang_res, top_down);
//   filter->propagate(t, a); filter->update(top_down, top_down_geo, res); computeMeanCov / meanLikelihood / ...
// Inputs and outputs are raw little-endian files in the directory given as argv[1] (written / read by
// tests/test_gpu_facade.py, which compares them with the CPU oracle).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "top_down_render/particle_filter.h"
#include "top_down_render/scan_renderer_polar.h"

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
static void dump(const std::string& path, const T* p, size_t n) {
  std::ofstream out(path, std::ios::binary | std::ios::trunc);
  out.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s <dir>\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  try {
    int ncls, rows, cols, nb, nr, npts, npart, use_device_scan, labels_mode = 0;
    float res, ang_res, tx, ty, omega;
    unsigned seed;
    {
      std::ifstream meta(dir + "/meta.txt");
      meta >> ncls >> rows >> cols >> nb >> nr >> npts >> npart >> res >> ang_res >> seed >> tx >> ty >> omega >> use_device_scan;
      if (!meta) throw std::runtime_error("bad meta.txt");
      meta >> labels_mode;  // optional: 1 = the map arrives as a class-index image (aerialMapCallback path)
    }
    auto maps = slurp<float>(dir + "/maps.bin");
    auto mask = slurp<uint8_t>(dir + "/mask.bin");
    auto pts = slurp<float>(dir + "/pts.bin");     // pcl::PointXYZI layout, 8 floats per point
    auto st_in = slurp<State>(dir + "/states.bin");

    // --- TopDownRender::initialize --------------------------------------------------------------------------------
    TopDownMap::Params map_params;
    map_params.num_classes = ncls;
    map_params.resolution = 1;
    for (int c = 0; c < ncls; c++) map_params.flatten_lut.push_back(c);
    TopDownMapPolar* map_ = new TopDownMapPolar(map_params);
    if (labels_mode) {
      // TopDownRender::aerialMapCallback -> updateMap(map_img, centre) (src/top_down_render.cpp:574-593)
      auto labels = slurp<uint8_t>(dir + "/labels.bin");
      map_->updateMap(labels.data(), rows, cols, Eigen::Vector2i(0, 0));
    }
    std::vector<Eigen::ArrayXXf> class_maps;
    for (int c = 0; c < ncls; c++) {
      Eigen::ArrayXXf m(rows, cols);
      std::memcpy(m.data(), maps.data() + (size_t)c * rows * cols, (size_t)rows * cols * sizeof(float));
      class_maps.push_back(m);
    }
    Eigen::ArrayXXc class_mask(rows, cols);
    std::memcpy(class_mask.data(), mask.data(), (size_t)rows * cols);
    if (!labels_mode) map_->setDistanceMaps(class_maps, class_mask);
    map_->samplePtsPolar(Eigen::Vector2i(nb, nr), ang_res);

    FilterParams filter_params;
    filter_params.pos_cov = 0.3f;
    filter_params.theta_cov = (float)(M_PI / 100);
    filter_params.regularization = 0.15f;
    filter_params.fixed_scale = 1.f;
    for (int c = 0; c < ncls; c++) filter_params.class_weights.push_back(1.f);
    // an initial pose far outside the map makes the constructor's initializeParticles() return early ("No map received
    // for input loc", src/particle_filter.cpp:32-36): the test supplies its own particle set below
    filter_params.init_pos_m_x = 1e9f;
    filter_params.init_pos_m_y = 1e9f;
    ParticleFilter* filter_ = new ParticleFilter(npart, map_, filter_params, seed);
    filter_->setStates(st_in);

    Eigen::VectorXi flatten_lut = Eigen::VectorXi::Constant(256, -1);
    for (int c = 0; c < ncls; c++) flatten_lut[c] = c;
    ScanRendererPolar* renderer_ = new ScanRendererPolar(flatten_lut);

    // --- TopDownRender::takeStep -------------------------------------------------------------------------------------
    pcl::PointCloud<PointType>::Ptr cloud_ptr(new pcl::PointCloud<PointType>());
    for (int i = 0; i < npts; i++) {
      PointType p{};
      p.x = pts[8 * i]; p.y = pts[8 * i + 1]; p.z = pts[8 * i + 2]; p.intensity = pts[8 * i + 4];
      cloud_ptr->push_back(p);
    }
    std::vector<Eigen::ArrayXXf> top_down, top_down_geo;
    for (int i = 0; i < map_->numClasses(); i++) top_down.push_back(Eigen::ArrayXXf(nb, nr));
    for (int i = 0; i < 2; i++) top_down_geo.push_back(Eigen::ArrayXXf(nb, nr));
    renderer_->renderSemanticTopDown(cloud_ptr, res, ang_res, top_down);

    Eigen::Vector2f motion_priort(tx, ty);
    filter_->propagate(motion_priort, omega);
    if (use_device_scan) filter_->update(*renderer_, res);
    else filter_->update(top_down, top_down_geo, res);

    Eigen::Matrix4f cov, cov_ml;
    Eigen::Vector4f mean, ml;
    filter_->computeMeanCov(cov);
    filter_->meanLikelihood(mean);
    filter_->maxLikelihood(ml);
    filter_->computeCov(cov_ml);

    // getGMM / adaptive particle count (src/particle_filter.cpp:151-157, 238-318)
    std::vector<Eigen::Vector3f> gmm_means;
    std::vector<Eigen::Matrix3f> gmm_covs;
    filter_->computeGMM();
    filter_->getGMM(gmm_means, gmm_covs);
    {
      std::vector<float> g;
      for (size_t c = 0; c < gmm_means.size(); c++) {
        for (int i = 0; i < 3; i++) g.push_back(gmm_means[c][i]);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) g.push_back(gmm_covs[c](i, j));
      }
      dump(dir + "/out_gmm.bin", g.data(), g.size());
    }

    // --- outputs -----------------------------------------------------------------------------------------------------
    std::vector<float> scan((size_t)ncls * nb * nr);
    for (int c = 0; c < ncls; c++) std::memcpy(scan.data() + (size_t)c * nb * nr, top_down[c].data(), (size_t)nb * nr * sizeof(float));
    dump(dir + "/out_scan.bin", scan.data(), scan.size());
    auto w = filter_->weights(npart);
    dump(dir + "/out_weights.bin", w.data(), w.size());
    auto idx = filter_->resampleIndices();
    dump(dir + "/out_idx.bin", idx.data(), idx.size());
    auto st = filter_->states();
    dump(dir + "/out_states.bin", st.data(), st.size());
    float stats[4 + 16 + 4 + 16 + 3];
    for (int i = 0; i < 4; i++) stats[i] = mean[i];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) stats[4 + 4 * i + j] = cov(i, j);
    for (int i = 0; i < 4; i++) stats[20 + i] = ml[i];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) stats[24 + 4 * i + j] = cov_ml(i, j);
    stats[40] = filter_->scale();
    stats[41] = (float)filter_->numParticles();
    stats[42] = filter_->isScaleFrozen() ? 1.f : 0.f;
    dump(dir + "/out_stats.bin", stats, 43);
    // optional: latency of takeStep through the C++ classes (TDR_FACADE_BENCH = number of timed steps)
    if (const char* e = std::getenv("TDR_FACADE_BENCH")) {
      const int steps = std::atoi(e);
      auto step = [&]() {
        renderer_->renderSemanticTopDown(cloud_ptr, res, ang_res, top_down);
        filter_->propagate(motion_priort, omega);
        if (use_device_scan) filter_->update(*renderer_, res);
        else filter_->update(top_down, top_down_geo, res);
        filter_->computeMeanCov(cov);   // publishPoseEst reads this every step (src/top_down_render.cpp:331-333)
      };
      if (const char* dr = std::getenv("TDR_FACADE_DEVICE_RNG")) filter_->configure(std::atoi(dr) == 0, 1);
      for (int i = 0; i < 5; i++) step();
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < steps; i++) step();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      std::printf("facade_bench steps=%d particles=%d ms_per_step=%.4f\n", steps, npart, ms / steps);
      // the same calls one by one
      double part[4] = {0, 0, 0, 0};
      auto lap = [&](int k, auto&& fn) {
        const auto a = std::chrono::steady_clock::now();
        fn();
        part[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
      };
      for (int i = 0; i < steps; i++) {
        lap(0, [&] { renderer_->renderSemanticTopDown(cloud_ptr, res, ang_res, top_down); });
        lap(1, [&] { filter_->propagate(motion_priort, omega); });
        lap(2, [&] { if (use_device_scan) filter_->update(*renderer_, res); else filter_->update(top_down, top_down_geo, res); });
        lap(3, [&] { filter_->computeMeanCov(cov); });
      }
      std::printf("facade_bench render=%.4f propagate=%.4f update=%.4f meancov=%.4f ms\n", part[0] / steps,
                  part[1] / steps, part[2] / steps, part[3] / steps);
    }
    delete renderer_;
    delete filter_;
    delete map_;
    std::puts("facade_step ok");
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "facade_step failed: %s\n", e.what());
    return 1;
  }
}
