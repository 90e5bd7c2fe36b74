// facade_session.cpp — the parts of the class surface facade_step.cpp does not touch, in the order a localisation
// session uses them (reference: src/top_down_render.cpp:81,115-117 initialize; :331-359 publishPoseEst; :574-593
// aerialMapCallback; src/particle_filter.cpp:19-84 initializeParticles):
//   map from a label image -> ParticleFilter constructor draws the initial particles around the configured pose with
//   an unknown scale -> getClassesAtPoint -> two steps -> freezeScale -> computeGMM + adaptive particle count ->
//   updateMap with a moved centre.
// Inputs / outputs: raw files in argv[1], like facade_step.cpp (tests/test_facade.py compares with the CPU oracle).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <stdexcept>
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
    int ncls, rows, cols, nb, nr, npts, npart;
    float res, ang_res, init_x, init_y, init_cov, init_deg, init_deg_cov;
    unsigned seed;
    {
      std::ifstream meta(dir + "/meta.txt");
      meta >> ncls >> rows >> cols >> nb >> nr >> npts >> npart >> res >> ang_res >> seed >> init_x >> init_y >> init_cov >>
          init_deg >> init_deg_cov;
      if (!meta) throw std::runtime_error("bad meta.txt");
    }
    auto labels = slurp<uint8_t>(dir + "/labels.bin");
    auto pts = slurp<float>(dir + "/pts.bin");

    TopDownMap::Params map_params;
    map_params.num_classes = ncls;
    map_params.resolution = 1;
    for (int c = 0; c < ncls; c++) map_params.flatten_lut.push_back(c);
    TopDownMapPolar* map_ = new TopDownMapPolar(map_params);
    if (map_->haveMap()) throw std::runtime_error("haveMap() before any map arrived");
    map_->updateMap(labels.data(), rows, cols, Eigen::Vector2i(0, 0));
    map_->samplePtsPolar(Eigen::Vector2i(nb, nr), ang_res);

    FilterParams fp;
    fp.pos_cov = 0.3f;
    fp.theta_cov = (float)(M_PI / 100);
    fp.regularization = 0.15f;
    fp.fixed_scale = -1.f;                      // unknown scale: log-uniform draw per particle (:40-58), scale gate on
    fp.init_pos_px_x = init_x; fp.init_pos_px_y = init_y; fp.init_pos_px_cov = init_cov;
    fp.init_pos_deg_theta = init_deg; fp.init_pos_deg_cov = init_deg_cov;
    for (int c = 0; c < ncls; c++) fp.class_weights.push_back(1.f);
    ParticleFilter* filter_ = new ParticleFilter(npart, map_, fp, seed);   // initializeParticles() (:14-16)
    auto st0 = filter_->states();
    dump(dir + "/out_init_states.bin", st0.data(), st0.size());

    // getClassesAtPoint (src/top_down_map.cpp:159-175) at the first particles' cells
    std::vector<int> cls_out;
    for (int i = 0; i < 16 && i < (int)st0.size(); i++) {
      std::vector<int> classes;
      map_->getClassesAtPoint(Eigen::Vector2i((int)st0[i].init_x_px, (int)st0[i].init_y_px), classes);
      for (int c = 0; c < ncls; c++) cls_out.push_back(c < (int)classes.size() ? classes[c] : -1);
    }
    dump(dir + "/out_classes.bin", cls_out.data(), cls_out.size());

    // getLocalMap (src/top_down_map_polar.cpp:21-53): the window of the first particle, materialised
    {
      std::vector<Eigen::ArrayXXf> dists;
      for (int c = 0; c < ncls; c++) dists.push_back(Eigen::ArrayXXf(nb, nr));
      Eigen::ArrayXXc wmask(nb, nr);
      const State& s0 = st0[0];
      map_->getLocalMap(Eigen::Vector2f(s0.dx_m * s0.scale + s0.init_x_px, s0.dy_m * s0.scale + s0.init_y_px), s0.scale,
                        res, dists, wmask);
      std::vector<float> w((size_t)(ncls + 1) * nb * nr);
      for (int c = 0; c < ncls; c++) std::memcpy(w.data() + (size_t)c * nb * nr, dists[c].data(), (size_t)nb * nr * sizeof(float));
      for (int k = 0; k < nb * nr; k++) w[(size_t)ncls * nb * nr + k] = (float)wmask(k);
      dump(dir + "/out_window.bin", w.data(), w.size());
    }

    // ActiveLocalizer (src/active_localizer.cpp): the best relative move for three pose hypotheses taken from the first
    // particles (checked against the oracle's restatement by tests/test_facade.py)
    {
      ActiveLocalizer al(map_);
      std::vector<Eigen::Vector3f> preds;
      for (int i = 0; i < 3; i++) {
        const State& s = st0[(size_t)i * 7 % st0.size()];
        preds.push_back(Eigen::Vector3f(s.dx_m * s.scale + s.init_x_px, s.dy_m * s.scale + s.init_y_px, s.theta + 0.4f * (float)i));
      }
      const Eigen::Vector2f best = al.getBestRelPos(preds);
      float act[12] = {best[0], best[1], al.lastBestDiff(), 0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < 3; i++)
        for (int d = 0; d < 3; d++) act[3 + 3 * i + d] = preds[i][d];
      const float host_diff = al.lastBestDiff();
      float act2[13];
      std::memcpy(act2, act, sizeof(act));
      act2[12] = host_diff;
      dump(dir + "/out_active.bin", act2, 13);
    }

    // the raster cache (src/top_down_map.cpp:197-224): saveRasterizedMaps, then a SECOND map constructed from that
    // directory (map_path without .svg / .png / .jpg, :42-46) must hold the same map: the same window, value for value
    {
      const std::string rdir = dir + "/site_raster_cache";
      map_->saveRasterizedMaps(rdir);
      TopDownMap::Params p2 = map_params;
      p2.map_path = rdir;
      const std::string cdir = dir + "/xview_cache";
      TopDownMapPolar m2(p2, cdir.c_str());
      if (!m2.haveMap()) throw std::runtime_error("the raster-cache directory did not load");
      m2.samplePtsPolar(Eigen::Vector2i(nb, nr), ang_res);
      std::vector<Eigen::ArrayXXf> da, db;
      for (int c = 0; c < ncls; c++) { da.push_back(Eigen::ArrayXXf(nb, nr)); db.push_back(Eigen::ArrayXXf(nb, nr)); }
      Eigen::ArrayXXc ma(nb, nr), mb(nb, nr);
      const Eigen::Vector2f ctr(st0[0].init_x_px, st0[0].init_y_px);
      map_->getLocalMap(ctr, 1.3f, res, da, ma);
      m2.getLocalMap(ctr, 1.3f, res, db, mb);
      for (int c = 0; c < ncls; c++)
        if (std::memcmp(da[c].data(), db[c].data(), (size_t)nb * nr * sizeof(float)) != 0)
          throw std::runtime_error("raster cache: distance maps differ");
      if (std::memcmp(ma.data(), mb.data(), (size_t)nb * nr) != 0) throw std::runtime_error("raster cache: masks differ");
      std::ifstream cached(cdir + "/cached_data.txt");   // (the constructor wrote the .eig cache behind it, :61)
      if (!cached) throw std::runtime_error("raster cache: no cached_data.txt written");
    }

    // the Cartesian pair: ScanRenderer::renderSemanticTopDown (src/scan_renderer.cpp:55-78) and
    // TopDownMap::getLocalMap(center, rot, res, ...) (src/top_down_map.cpp:429-459) on a 40 x 56 window
    {
      const int wr = 40, wc = 56;
      Eigen::VectorXi lut = Eigen::VectorXi::Constant(256, -1);
      for (int c = 0; c < ncls; c++) lut[c] = c;
      ScanRenderer cart(lut);
      pcl::PointCloud<PointType>::Ptr cl(new pcl::PointCloud<PointType>());
      for (int i = 0; i < npts; i++) {
        PointType p{};
        p.x = pts[8 * i]; p.y = pts[8 * i + 1]; p.z = pts[8 * i + 2]; p.intensity = pts[8 * i + 4];
        cl->push_back(p);
      }
      std::vector<Eigen::ArrayXXf> imgs, dists;
      for (int c = 0; c < ncls; c++) { imgs.push_back(Eigen::ArrayXXf(wr, wc)); dists.push_back(Eigen::ArrayXXf(wr, wc)); }
      cart.renderSemanticTopDown(cl, res, imgs);
      Eigen::ArrayXXc wmask(wr, wc);
      TopDownMap* base = map_;   // the Cartesian overload lives in the base class
      base->getLocalMap(Eigen::Vector2f(st0[0].init_x_px, st0[0].init_y_px), 0.6f, 1.5f, dists, wmask);
      std::vector<float> out((size_t)(2 * ncls + 1) * wr * wc);
      for (int c = 0; c < ncls; c++) {
        std::memcpy(out.data() + (size_t)c * wr * wc, imgs[c].data(), (size_t)wr * wc * sizeof(float));
        std::memcpy(out.data() + (size_t)(ncls + c) * wr * wc, dists[c].data(), (size_t)wr * wc * sizeof(float));
      }
      for (int k = 0; k < wr * wc; k++) out[(size_t)2 * ncls * wr * wc + k] = (float)wmask(k);
      dump(dir + "/out_cartesian.bin", out.data(), out.size());
    }

    // StateParticle (include/top_down_render/state_particle.h:40-66): two particles sharing ONE generator, as in the
    // reference's filter; constructor draw, computeWeight, propagate with and without scale freeze
    std::vector<float> sp_out;
    {
      std::mt19937 gen(seed + 1);
      StateParticle a(&gen, map_, &fp), b(&gen, map_, &fp);
      std::vector<Eigen::ArrayXXf> scan_imgs, geo;
      for (int c = 0; c < ncls; c++) scan_imgs.push_back(Eigen::ArrayXXf(nb, nr));
      {
        pcl::PointCloud<PointType>::Ptr cl(new pcl::PointCloud<PointType>());
        for (int i = 0; i < npts; i++) {
          PointType p{};
          p.x = pts[8 * i]; p.y = pts[8 * i + 1]; p.z = pts[8 * i + 2]; p.intensity = pts[8 * i + 4];
          cl->push_back(p);
        }
        Eigen::VectorXi lut = Eigen::VectorXi::Constant(256, -1);
        for (int c = 0; c < ncls; c++) lut[c] = c;
        ScanRendererPolar rr(lut);
        rr.renderSemanticTopDown(cl, res, ang_res, scan_imgs);
      }
      StateParticle* ps[2] = {&a, &b};
      for (StateParticle* p : ps) {
        State s = p->state();
        const float* f = reinterpret_cast<const float*>(&s);
        for (int i = 0; i < 6; i++) sp_out.push_back(f[i]);
        sp_out.push_back(s.have_init ? 1.f : 0.f);
      }
      Eigen::Vector2f tr(1.f, 0.25f);
      a.propagate(tr, 0.01f, false);
      b.propagate(tr, 0.01f, true);
      for (StateParticle* p : ps) {
        p->computeWeight(scan_imgs, geo, res);
        State s = p->state();
        const float* f = reinterpret_cast<const float*>(&s);
        for (int i = 0; i < 6; i++) sp_out.push_back(f[i]);
        sp_out.push_back(s.have_init ? 1.f : 0.f);
        sp_out.push_back(p->weight());
        sp_out.push_back(p->lastDist());
        const Eigen::Vector4f ml = p->mlState();
        for (int i = 0; i < 4; i++) sp_out.push_back(ml[i]);
      }
      a.setScale(2.5f);
      sp_out.push_back(a.state().scale);
    }
    dump(dir + "/out_state_particles.bin", sp_out.data(), sp_out.size());

    Eigen::VectorXi flatten_lut = Eigen::VectorXi::Constant(256, -1);
    for (int c = 0; c < ncls; c++) flatten_lut[c] = c;
    ScanRendererPolar* renderer_ = new ScanRendererPolar(flatten_lut);
    pcl::PointCloud<PointType>::Ptr cloud_ptr(new pcl::PointCloud<PointType>());
    for (int i = 0; i < npts; i++) {
      PointType p{};
      p.x = pts[8 * i]; p.y = pts[8 * i + 1]; p.z = pts[8 * i + 2]; p.intensity = pts[8 * i + 4];
      cloud_ptr->push_back(p);
    }
    std::vector<Eigen::ArrayXXf> top_down, top_down_geo;
    for (int i = 0; i < map_->numClasses(); i++) top_down.push_back(Eigen::ArrayXXf(nb, nr));
    for (int i = 0; i < 2; i++) top_down_geo.push_back(Eigen::ArrayXXf(nb, nr));

    Eigen::Vector2f motion(1.f, 0.25f);
    for (int step = 0; step < 2; step++) {
      renderer_->renderSemanticTopDown(cloud_ptr, res, ang_res, top_down);
      filter_->propagate(motion, 0.01f);
      filter_->update(top_down, top_down_geo, res);
      if (step == 0) {
        auto st1 = filter_->states();
        dump(dir + "/out_states_step1.bin", st1.data(), st1.size());
        auto w1 = filter_->weights(npart);
        dump(dir + "/out_weights_step1.bin", w1.data(), w1.size());
        auto i1 = filter_->resampleIndices();
        dump(dir + "/out_idx_step1.bin", i1.data(), i1.size());
      }
    }
    auto st2 = filter_->states();
    dump(dir + "/out_states_step2.bin", st2.data(), st2.size());
    float misc[8] = {filter_->scale(), filter_->isScaleFrozen() ? 1.f : 0.f, 0, 0, 0, 0, 0, 0};

    filter_->freezeScale();                       // :343-357
    auto st3 = filter_->states();
    dump(dir + "/out_states_frozen.bin", st3.data(), st3.size());
    misc[2] = filter_->scale();
    misc[3] = filter_->isScaleFrozen() ? 1.f : 0.f;

    // one more step with the scale frozen (propagate stops drawing the fourth normal, state_particle.cpp:70-73)
    renderer_->renderSemanticTopDown(cloud_ptr, res, ang_res, top_down);
    filter_->propagate(motion, 0.01f);
    filter_->update(top_down, top_down_geo, res);
    auto st4 = filter_->states();
    dump(dir + "/out_states_step3.bin", st4.data(), st4.size());

    // adaptive particle count from the mixture (:151-157, 252-318)
    filter_->computeGMM();
    filter_->setAdaptiveCount(true);
    renderer_->renderSemanticTopDown(cloud_ptr, res, ang_res, top_down);
    filter_->propagate(motion, 0.01f);
    filter_->update(top_down, top_down_geo, res);
    misc[4] = (float)filter_->numParticles();
    filter_->setAdaptiveCount(false);

    // a new aerial map with a moved centre shifts every particle's init position (:325-334)
    auto before = filter_->states();
    filter_->updateMap(labels.data(), rows, cols, map_params.flatten_lut, Eigen::Vector2i(7, -3));
    auto after = filter_->states();
    misc[5] = after[0].init_x_px - before[0].init_x_px;
    misc[6] = after[0].init_y_px - before[0].init_y_px;
    misc[7] = (float)map_->mapCenter()[0] * 1000.f + (float)map_->mapCenter()[1];
    dump(dir + "/out_misc.bin", misc, 8);

    // ---- the rest of the class surface the node touches (src/top_down_render.cpp:431, 540, 591) -------------------------
    float extra[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    {   // the reference's two-argument updateMap(cv::Mat, centre): the LUT comes from the map's Params
      cv::Mat img(rows, cols, labels.data());
      before = filter_->states();
      filter_->updateMap(img, Eigen::Vector2i(9, -1));
      after = filter_->states();
      extra[0] = after[0].init_x_px - before[0].init_x_px;      // +2 against the centre (7, -3) set above
      extra[1] = after[0].init_y_px - before[0].init_y_px;      // +2
      cv::Mat canvas(rows, cols, labels.data());
      filter_->visualize(canvas);                                // no hook set: a no-op, must not throw
      // with a hook the host gets a snapshot of what the reference's drawing code reads: states, mixture, best state
      filter_->setVisualizer([](cv::Mat&, const ParticleFilter::Snapshot& s, void* user) {
        *static_cast<float*>(user) = (float)s.particles.size() + (s.have_best ? 0.5f : 0.f);
      }, &extra[7]);
      filter_->visualize(canvas);
      extra[7] -= (float)filter_->numParticles();                // 0.5 left: every particle + a best state arrived
    }
    {   // Wrong-sized images: the reference's per-scan calls return silently (scan_renderer_polar.cpp:85,
        // top_down_map_polar.cpp:25, particle_filter.cpp:96-99) and so do these — no exception reaches the node's single
        // catch-and-exit (top_down_render_node.cpp:8-14), the particle set is untouched, tdr_last_error() says why
      std::vector<Eigen::ArrayXXf> wrong, geo2;
      for (int c = 0; c < ncls; c++) wrong.push_back(Eigen::ArrayXXf(nb, nr + 1));
      const std::vector<State> st0 = filter_->states();
      bool threw = false;
      try {
        filter_->update(wrong, geo2, res);                               // every image the wrong shape
        std::string why = tdr_last_error();
        bool said = why.find("update: scan image") != std::string::npos;
        std::vector<Eigen::ArrayXXf> few(1, Eigen::ArrayXXf(nb, nr));
        if (ncls > 1) {
          filter_->update(few, geo2, res);                               // fewer images than classes
          why = tdr_last_error();
          said = said && why.find("fewer scan images") != std::string::npos;
        }
        std::vector<Eigen::ArrayXXf> ragged;
        for (int c = 0; c < ncls; c++) ragged.push_back(Eigen::ArrayXXf(nb, c == ncls - 1 ? nr - 1 : nr));
        renderer_->renderSemanticTopDown(cloud_ptr, res, ang_res, ragged);   // images of different sizes
        why = tdr_last_error();
        said = said && (ncls < 2 || why.find("different sizes") != std::string::npos);
        std::vector<Eigen::ArrayXXf> win;
        for (int c = 0; c < ncls; c++) win.push_back(Eigen::ArrayXXf(nb, nr + 2));
        Eigen::ArrayXXc wmask(nb, nr + 2);
        map_->getLocalMap(Eigen::Vector2f(60.f, 70.f), 1.f, res, win, wmask);   // not the shape given to samplePtsPolar
        why = tdr_last_error();
        said = said && why.find("getLocalMap") != std::string::npos;
        const std::vector<State> st1 = filter_->states();
        const bool same = st0.size() == st1.size() && std::memcmp(st0.data(), st1.data(), st0.size() * sizeof(State)) == 0;
        extra[2] = (said && same) ? 1.f : 0.f;
      } catch (const std::exception&) {
        threw = true;
      }
      if (threw) extra[2] = -1.f;
    }
    {   // renderGeometricTopDown + getLocalGeoMap (dead at the node's call site, part of the surface)
      std::vector<Eigen::ArrayXXf> geo_imgs;
      for (int i = 0; i < 2; i++) geo_imgs.push_back(Eigen::ArrayXXf(nb, nr));
      renderer_->renderGeometricTopDown(cloud_ptr, res, ang_res, geo_imgs);
      dump(dir + "/out_geo_render.bin", geo_imgs[0].data(), (size_t)nb * nr);
      dump(dir + "/out_geo_render1.bin", geo_imgs[1].data(), (size_t)nb * nr);
      std::vector<Eigen::ArrayXXf> gwin;
      for (int i = 0; i < 2; i++) gwin.push_back(Eigen::ArrayXXf(nb, nr));
      map_->getLocalGeoMap(Eigen::Vector2f(60.f, 70.f), 1.f, res, gwin);
      float mx = 0.f;
      for (int k = 0; k < nb * nr; k++) mx = std::max(mx, std::max(gwin[0](k), gwin[1](k)));
      extra[3] = mx;   // the updateMap path leaves both geometric layers at 1 (src/top_down_map.cpp:126-133)
    }
    {   // several GPUs behind the class surface: the sharded constructor on a one-rank RCCL communicator (every exchange
        // really goes through RCCL) must reproduce the plain filter bit for bit; and the opt-in geometric cost
        // (state_particle.cpp:145-152) with all-zero geometric images adds exact zeros
      unsigned char id[TDR_COMM_ID_BYTES];
      tdr_comm* comm = nullptr;
      if (tdr_comm_rccl_unique_id(id) != TDR_OK || tdr_comm_create_rccl(1, 0, id, &comm) != TDR_OK)
        throw std::runtime_error(std::string("rccl communicator: ") + tdr_last_error());
      const int n2 = 768;
      FilterParams fp2 = fp;
      fp2.fixed_scale = 1.f;
      ParticleFilter plain(n2, map_, fp2, 77), shard(n2, map_, fp2, 77, comm), geo(n2, map_, fp2, 77);
      geo.setGeometricCost(true);
      for (int i = 0; i < 2; i++) top_down_geo[i].setZero();
      renderer_->renderSemanticTopDown(cloud_ptr, res, ang_res, top_down);
      for (ParticleFilter* f : {&plain, &shard, &geo}) {
        f->propagate(motion, 0.01f);
        f->update(top_down, top_down_geo, res);
      }
      const auto wp = plain.weights(n2), ws = shard.weights(n2), wg = geo.weights(n2);
      const auto sp2 = plain.states(), ss = shard.states();
      bool same = wp.size() == ws.size() && sp2.size() == ss.size() &&
                  std::memcmp(wp.data(), ws.data(), wp.size() * sizeof(float)) == 0 &&
                  std::memcmp(sp2.data(), ss.data(), sp2.size() * sizeof(State)) == 0;
      extra[4] = same ? 1.f : 0.f;
      float worst = 0.f;
      for (size_t i = 0; i < wp.size(); i++) worst = std::max(worst, std::fabs(wg[i] - wp[i]) / std::max(std::fabs(wp[i]), 1e-30f));
      extra[5] = worst;
      extra[6] = (float)wp.size();
      tdr_comm_destroy(comm);
    }
    dump(dir + "/out_extra.bin", extra, 8);
    delete renderer_;
    delete filter_;
    delete map_;
    std::puts("facade_session ok");
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "facade_session failed: %s\n", e.what());
    return 1;
  }
}
