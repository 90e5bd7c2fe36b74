// TopDownMap — reference surface: include/top_down_render/top_down_map.h:52-102.  Holds the per-class truncated
// distance maps + unknown mask on the GPU (tdr_map, include/tdr.h).  The reference builds these at load time from an
// SVG / PNG / cache file (src/top_down_map.cpp:9-64, OpenCV + nanosvg): that ingest is outside the per-scan path
// (SURVEY §8f N1), so here the distance maps are handed over with setDistanceMaps() in the same layout as the
// reference's class_maps_ / class_mask_ members.
#ifndef TOP_DOWN_MAP_H_
#define TOP_DOWN_MAP_H_

#include <stdexcept>
#include <string>
#include <vector>

#include "tdr.h"
#include "top_down_render/tdr_compat.h"

class TopDownMap {
 public:
  struct Params {  // top_down_map.h:54-62, member for member
    std::string map_path = "";
    SemanticColorLut color_lut;   // static-map loader only (outside the per-scan path); carried, never read
    std::vector<int> flatten_lut;
    int num_classes = 0;
    std::vector<int> exclusive_classes;
    float resolution = 1;
    float out_of_bounds_const = 5;  // unused by the reference too: every out-of-bounds write is a literal 0
  };
  // src/top_down_map.cpp:9-64.  An empty map_path is the dynamic-map case (the map arrives through updateMap).  A static
  // map is taken from the reference's own cache — ~/.ros/xview_cache, written by the reference or by saveCachedMaps() —
  // when its (map_path, num_classes, resolution) match (:18-20, :226-261), else from a raster-cache directory of class<i>.png
  // (:42-46).  Parsing SVG / decoding colour images (nanosvg, OpenCV: load-time work outside the per-scan path) is not done
  // here: without either the map stays empty until setDistanceMaps() / updateMap() provide it.
  explicit TopDownMap(const Params& params, const char* cache_dir = nullptr) : params_(params) {
    if (tdr_map_create(&m_) != TDR_OK) throw std::runtime_error(std::string("TopDownMap: ") + tdr_last_error());
    if (!params_.map_path.empty() && params_.num_classes > 0) {
      int loaded = 0;
      if (tdr_map_load_cache(m_, cache_dir, params_.map_path.c_str(), params_.num_classes, params_.resolution, 0, 0,
                             &loaded) != TDR_OK) {
        const std::string msg = std::string("TopDownMap: ") + tdr_last_error();
        tdr_map_destroy(m_);
        throw std::runtime_error(msg);
      }
      // a map_path that is neither .svg nor .png / .jpg names a raster-cache directory (:42-46): class<i>.png, read
      // over zlib, distance transforms on the GPU; then the cache is written like the reference does (:61)
      const std::string& mp = params_.map_path;
      const std::string ext = mp.size() >= 4 ? mp.substr(mp.size() - 4) : std::string();
      if (!loaded && ext != ".svg" && ext != ".png" && ext != ".jpg" &&
          tdr_map_load_rasters(m_, mp.c_str(), params_.num_classes, params_.resolution, 0, 0) == TDR_OK)
        (void)tdr_map_save_cache(m_, cache_dir, mp.c_str());
    }
  }
  // saveRasterizedMaps / loadRasterizedMaps (:197-224): a directory of class<i>.png (8-bit grey, 0 inside the class,
  // flipped like the reference stores them); load = rasters -> geometric layers -> distance maps, on the GPU
  void saveRasterizedMaps(const std::string& path) {
    if (tdr_map_save_rasters(m_, path.c_str()) != TDR_OK)
      throw std::runtime_error(std::string("saveRasterizedMaps: ") + tdr_last_error());
  }
  void loadRasterizedMaps(const std::string& map_path) {
    if (tdr_map_load_rasters(m_, map_path.c_str(), params_.num_classes, params_.resolution, map_center_[0], map_center_[1]) != TDR_OK)
      throw std::runtime_error(std::string("loadRasterizedMaps: ") + tdr_last_error());
  }
  // saveCachedMaps (:263-286): the cache the constructor above (and the reference) reads
  void saveCachedMaps(const std::string& map_path, const char* cache_dir = nullptr) {
    if (tdr_map_save_cache(m_, cache_dir, map_path.c_str()) != TDR_OK)
      throw std::runtime_error(std::string("saveCachedMaps: ") + tdr_last_error());
  }
  virtual ~TopDownMap() { tdr_map_destroy(m_); }
  TopDownMap(const TopDownMap&) = delete;
  TopDownMap& operator=(const TopDownMap&) = delete;

  // class_maps_[c] (rows = height/resolution, cols = width/resolution, column-major) and class_mask_ (1 = unknown),
  // as computeDists leaves them (src/top_down_map.cpp:289-326).  Also the map side of updateMap (:146-157).
  void setDistanceMaps(const std::vector<Eigen::ArrayXXf>& class_maps, const Eigen::ArrayXXc& class_mask,
                       const Eigen::Vector2i& map_center = Eigen::Vector2i(0, 0)) {
    if (class_maps.empty()) return;
    const int rows = (int)class_maps[0].rows(), cols = (int)class_maps[0].cols(), ncls = (int)class_maps.size();
    std::vector<float> buf((size_t)rows * cols * ncls);
    for (int c = 0; c < ncls; c++)
      std::memcpy(buf.data() + (size_t)c * rows * cols, class_maps[c].data(), (size_t)rows * cols * sizeof(float));
    if (tdr_map_set(m_, buf.data(), class_mask.data(), ncls, rows, cols, params_.resolution, map_center[0],
                    map_center[1]) != TDR_OK)
      throw std::runtime_error(std::string("TopDownMap::setDistanceMaps: ") + tdr_last_error());
    params_.num_classes = ncls;
    map_center_ = map_center;
  }
  // updateMap (src/top_down_map.cpp:146-157) for a class-index image in cv::Mat CV_8UC1 layout (row 0 = top):
  // loadCompressedRasterMap + computeDists run on the GPU.  `flatten_lut` is Params::flatten_lut.
  void updateMap(const uint8_t* label_img, int img_h, int img_w, const Eigen::Vector2i& map_center) {
    std::vector<int32_t> lut(params_.flatten_lut.begin(), params_.flatten_lut.end());
    if (lut.empty() || params_.num_classes < 1) throw std::invalid_argument("updateMap: Params::flatten_lut / num_classes not set");
    if (tdr_map_set_labels(m_, label_img, img_h, img_w, lut.data(), (int)lut.size(), params_.num_classes,
                           params_.resolution, map_center[0], map_center[1]) != TDR_OK)
      throw std::runtime_error(std::string("TopDownMap::updateMap: ") + tdr_last_error());
    map_center_ = map_center;
  }
  void updateMap(const cv::Mat& map, const Eigen::Vector2i& map_center) {  // the reference's signature (:146-157)
    if (map.empty()) throw std::invalid_argument("updateMap: empty image");
    if (map.isContinuous()) return updateMap(map.ptr<uint8_t>(), map.rows, map.cols, map_center);
    std::vector<uint8_t> packed((size_t)map.rows * map.cols);   // a view with row padding: pack the rows
    for (int r = 0; r < map.rows; r++) std::memcpy(packed.data() + (size_t)r * map.cols, map.ptr<uint8_t>(r), (size_t)map.cols);
    updateMap(packed.data(), map.rows, map.cols, map_center);
  }
  const Params& params() const { return params_; }
  void getClassesAtPoint(const Eigen::Vector2i& center_ind, std::vector<int>& classes) {  // top_down_map.cpp:159-170
    classes.clear();
    uint32_t bits = 0;
    if (tdr_map_classes_at_point(m_, center_ind[0], center_ind[1], &bits) != TDR_OK) return;
    for (int c = 0; c < params_.num_classes; c++)
      if (bits & (1u << c)) classes.push_back(c);
  }
  // getLocalMap (src/top_down_map.cpp:429-459): the Cartesian window of one pose; sizes are read off dists[0] like the
  // reference does; dists.size() < 1 -> nothing happens (:431).  mask: 1 = unknown or outside the map.
  void getLocalMap(Eigen::Vector2f center, float rot, float res, std::vector<Eigen::ArrayXXf>& dists,
                   Eigen::ArrayXXc& mask) {
    if (dists.size() < 1) return;
    local_map(0, center, rot, res, (int)dists[0].rows(), (int)dists[0].cols(), dists, mask);
  }
  // getLocalGeoMap (:461-481): the same window gathered from the two geometric layers geo_maps_ ([0] distance to the
  // nearest cell without a geometric class, [1] with one); dists.size() < 1 -> nothing happens (:464).
  void getLocalGeoMap(Eigen::Vector2f center, float rot, float res, std::vector<Eigen::ArrayXXf>& dists) {
    if (dists.size() < 1) return;
    local_geo_map(0, center, rot, res, (int)dists[0].rows(), (int)dists[0].cols(), dists);
  }
  void getClassesAtPoint(const Eigen::Vector2f& center, std::vector<int>& classes) {      // :172-175
    getClassesAtPoint(Eigen::Vector2i((int)(center[0] / params_.resolution), (int)(center[1] / params_.resolution)), classes);
  }
  Eigen::Vector2i size() const {
    int rows = 0, cols = 0;
    tdr_map_info(m_, nullptr, &rows, &cols, nullptr, nullptr);
    return Eigen::Vector2i(cols, rows);
  }
  Eigen::Vector2i mapCenter() const {   // asks the handle: ParticleFilter::updateMap moves the centre too
    int cx = map_center_[0], cy = map_center_[1];
    tdr_map_center(m_, &cx, &cy);
    return Eigen::Vector2i(cx, cy);
  }
  int numClasses() const { return params_.num_classes; }
  float resolution() const { return params_.resolution; }
  bool haveMap() const {
    int have = 0;
    tdr_map_info(m_, nullptr, nullptr, nullptr, nullptr, &have);
    return have != 0;
  }
  tdr_map* handle() const { return m_; }

 protected:
  void local_map(int polar, const Eigen::Vector2f& center, float scale_or_rot, float res, int rows, int cols,
                 std::vector<Eigen::ArrayXXf>& dists, Eigen::ArrayXXc& mask) {
    const int ncls = params_.num_classes;
    // sizes the call cannot use: a silent return like the reference's (top_down_map.cpp:431, top_down_map_polar.cpp:25),
    // the reason in tdr_last_error()
    if ((int)dists.size() < ncls) { tdr_set_error(TDR_ERR_ARG, "getLocalMap: fewer output arrays than map classes"); return; }
    const size_t P = (size_t)rows * cols;
    for (int c = 0; c < ncls; c++)
      if ((size_t)dists[c].rows() * dists[c].cols() != P) { tdr_set_error(TDR_ERR_ARG, "getLocalMap: output arrays of different sizes"); return; }
    if ((size_t)mask.rows() * mask.cols() != P) { tdr_set_error(TDR_ERR_ARG, "getLocalMap: mask size differs from the windows'"); return; }
    std::vector<float> d(P * ncls);
    std::vector<uint8_t> k(P);
    if (tdr_map_local_map(m_, polar, center[0], center[1], scale_or_rot, res, rows, cols, d.data(), k.data()) != TDR_OK)
      throw std::runtime_error(std::string("getLocalMap: ") + tdr_last_error());
    for (int c = 0; c < ncls; c++) std::memcpy(dists[c].data(), d.data() + P * c, P * sizeof(float));
    std::memcpy(mask.data(), k.data(), P);
  }
  void local_geo_map(int polar, const Eigen::Vector2f& center, float scale_or_rot, float res, int rows, int cols,
                     std::vector<Eigen::ArrayXXf>& dists) {
    const size_t P = (size_t)rows * cols;
    for (size_t c = 0; c < dists.size() && c < 2; c++)
      if ((size_t)dists[c].rows() * dists[c].cols() != P) { tdr_set_error(TDR_ERR_ARG, "getLocalGeoMap: output arrays of different sizes"); return; }
    std::vector<float> d(P * 2);
    if (tdr_map_local_geo_map(m_, polar, center[0], center[1], scale_or_rot, res, rows, cols, d.data()) != TDR_OK)
      throw std::runtime_error(std::string("getLocalGeoMap: ") + tdr_last_error());
    for (size_t c = 0; c < dists.size() && c < 2; c++) std::memcpy(dists[c].data(), d.data() + P * c, P * sizeof(float));
  }
  Params params_;
  Eigen::Vector2i map_center_;
  tdr_map* m_ = nullptr;
};

#endif  // TOP_DOWN_MAP_H_
