// tdr_cmap.hip — the COMPACT map: the same cell records as tdr_map.hip at a quarter of the bytes, exact by construction.
//
// Why.  A scoring wave whose 64 particles are spread over the map (particle initialisation, the uniform tenth of the
// bench workload, multi-hypothesis clusters) touches one 128-byte line per lane and sample and uses 32 bytes of it: the
// kernel then runs at the HBM rate and only fewer bytes make it faster.  Distance maps are
// `min(50, resolution * sqrt(d2))` with integer d2 (src/top_down_map.cpp:312-317, cv::distanceTransform DIST_L2 /
// MASK_PRECISE): a 4000 x 4000 six-class map holds < 800 DISTINCT float values.  So a cell is stored as 10-bit indices
// into a dictionary of the map's own float values — three per dword at bits [2 + 10k, 12 + 10k), `known` in bit 0 of every
// dword (the wide form: of the last dword only) —
// and decoded through an LDS copy of the dictionary: bit-identical operands, 8 bytes instead of 32 for six classes.
// Records are tiled (32 / record bytes) rows x 4 columns per 128-byte line, so a ray crosses about (|sin| + |cos|) / 4
// lines per cell step instead of one (row direction) or a quarter (column direction) of a row-major layout.  Inside a
// tile the records run row-major and the tiles themselves column by column over the map: the row's share of a byte
// offset is then plainly linear — row * 4 * record bytes, tile and all — and an offset is four integer instructions
// (cmap_offset, tdr_score_dev.h).
//
// Built from the dense records, whoever produced them (tdr_k_pack_map, tdr_k_map_from_labels), once per map:
//   cmap_collect_kernel  every distance value -> a hash set in device memory (distinct count capped at 1024)
//   host                 the <= 1024 values, sorted, become the dictionary; slot -> index table for the hash set
//   cmap_pack_kernel     one thread per compact record: look the cell's values up, pack, store tile-major (coalesced)
// A map with more than 1024 distinct values (fine resolutions) takes the WIDE form further down (4-7 classes, up to 4096
// values); beyond that, or with more than 11 classes, there is no compact form and the scoring kernels read the dense
// records.
#include <algorithm>

#include "tdr_common.h"
#include "tdr_score_dev.h"   // the known mask's geometry

#define CMAP_HASH_BITS 14
#define CMAP_HASH_SLOTS (1 << CMAP_HASH_BITS)
static_assert(TDR_CMAP_WORKSPACE_BYTES >= CMAP_HASH_SLOTS * 6 + 64, "compact-map workspace");
#define CMAP_EMPTY 0xFFFFFFFFu

__device__ __forceinline__ unsigned cmap_hash(unsigned v) { return (v * 2654435761u) >> (32 - CMAP_HASH_BITS); }

// count[0] = distinct values inserted, count[1] = overflow flag
__global__ __launch_bounds__(256) void cmap_collect_kernel(const float* __restrict__ rec, int64_t gcell, int rf, int ncls,
                                                           unsigned* __restrict__ hash, int* __restrict__ count) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= gcell) return;
  unsigned last = CMAP_EMPTY;
  for (int k = 0; k < ncls; k++) {
    const unsigned v = __float_as_uint(rec[idx * rf + k]);
    if (v == last) continue;   // runs of equal values (the truncation distance) cost one lookup
    last = v;
    if (v == CMAP_EMPTY) { atomicExch(&count[1], 1); return; }   // this NaN payload is the empty marker: not compactable
    unsigned h = cmap_hash(v);
    for (int probe = 0; probe < CMAP_HASH_SLOTS; probe++) {
      const unsigned cur = hash[h];
      if (cur == v) break;
      if (cur == CMAP_EMPTY) {
        if (count[1]) return;                                     // already overflowed: stop filling the table
        const unsigned old = atomicCAS(&hash[h], CMAP_EMPTY, v);
        if (old == CMAP_EMPTY) {
          if (atomicAdd(&count[0], 1) >= TDR_CMAP_WIDE_MAX_DICT) atomicExch(&count[1], 1);
          break;
        }
        if (old == v) break;
      }
      h = (h + 1) & (CMAP_HASH_SLOTS - 1);
    }
  }
}

struct CmapGeom {
  int cw;            // dwords per record: 1, 2 or 4
  int lc;            // log2 of the tile's row count: tile = (1 << lc) rows x 4 columns = 128 bytes
  int tiles_r, tiles_c;
};
static CmapGeom cmap_geom(int cw, int rows, int cols) {
  CmapGeom g;
  g.cw = cw;
  g.lc = cw == 1 ? 3 : (cw == 2 ? 2 : 1);
  // cell (r, c), r in [-1, rows], lives in tile ((r >> lc) + 1, (c >> 2) + 1): tile row / column 0 hold the guard ring
  g.tiles_r = (rows >> g.lc) + 2;
  g.tiles_c = (cols >> 2) + 2;
  return g;
}
extern "C" int tdr_cmap_words(int ncls) {
  const int rf = tdr_rec_floats(ncls);
  const int nd = tdr_has_kslot(ncls, rf) ? rf - 2 : rf - 1;   // distance slots the scoring loop multiplies
  const int dw = (nd + 2) / 3;                                // three 10-bit fields per dword
  if (dw > 4) return 0;
  return dw <= 1 ? 1 : (dw == 2 ? 2 : 4);
}
// Behind the tiles of a narrow compact map sits its KNOWN MASK: one bit per cell in 32 x 32-cell tiles of 32 words (layout:
// tdr_score_dev.h, kmask_offset); cells outside the map are 0 = unknown, like their guard records.  2 MB for a 4000 x 4000
// map.  The shift-uniform scoring kernel stages the part a workgroup's windows cover in LDS and reads the `known` bit of
// every sample there (tdr_score_su.hip); the kernels that skip empty scan bins gather it (tdr_score_cart.hip,
// score_polar_kernel<SKIP>).
extern "C" size_t tdr_cmap_tile_words(int ncls, int rows, int cols) {
  const int cw = tdr_cmap_words(ncls);
  if (!cw) return 0;
  const CmapGeom g = cmap_geom(cw, rows, cols);
  return (size_t)g.tiles_r * g.tiles_c * 32;   // 128 bytes per tile
}
// dwords of the tiles + the known mask (+ 4 words: a lane of score_polar_kernel<SKIP> reads a whole record's worth of
// dwords at a mask word), rounded up to whole 128-byte lines: where the class planes start
extern "C" size_t tdr_cmap_plane_offset_words(int ncls, int rows, int cols) {
  const size_t tiles = tdr_cmap_tile_words(ncls, rows, cols);
  if (!tiles) return 0;
  return (tiles + (size_t)kmask_trows(rows) * kmask_tcols(cols) * 32 + 4 + 31) / 32 * 32;
}
// Behind the mask sit the CLASS PLANES (layout: plane_offset, tdr_score_dev.h): per class one 16-bit value per cell —
// the class's dictionary index times 4 in bits 2-11, `known` in bit 15 — in tiles of 8 x 8 cells = one 128-byte line, the tiles
// column by column with a guard band like the mask's.  A sample whose scan bin holds ONE class needs 2 bytes of the map,
// and a ray of the polar window crosses 8 cells of a plane's line where it crosses 4 of a record tile's: the kernel that
// scores scattered particles one wave per particle (tdr_score_ray.hip) pulls a third of the lines through the fabric
// (measured on the config-2 scene: 2 400 lines per window against 6 700).  0: the map is too large for the kernels'
// 32-bit byte offsets — it then has no planes and the scattered particles stay with score_polar_kernel.
extern "C" size_t tdr_cmap_plane_words(int ncls, int rows, int cols) {
  const size_t off = tdr_cmap_plane_offset_words(ncls, rows, cols);
  if (!off) return 0;
  const size_t per = (size_t)plane_trows(rows) * plane_tcols(cols) * 32;
  if ((off + per * (size_t)(ncls + 1)) * 4 > 0xFFFFFF00ull || (size_t)plane_trows(rows) * 128 >= (1u << 23) || per * 4 / 128 >= (1u << 24))
    return 0;
  return per;
}
// ... and behind the class planes the COARSE MASK PLANE: the known mask once more in the planes' own shape — a 16-bit cell
// holds the known bits of a 4 x 4 block of map cells (cell (r >> 2, c >> 2), bit (r & 3) * 4 + (c & 3)), tiles of 8 x 8 such
// cells: a 128-byte line covers 32 x 32 map cells, like the known mask's, and a ray of 64 cells crosses 2-3 of them (16
// columns of one row per cell, the first layout, made that 1-8) — so that the ray-mapped kernel addresses "the cell's known
// bit" and "the cell's class value" with one formula.  A tile column has the class planes' stride (plane_trows tile rows, of
// which the first quarter is used): the formula's column constant is then the same for every plane.
extern "C" size_t tdr_cmap_cmask_words(int ncls, int rows, int cols) {
  if (!tdr_cmap_plane_words(ncls, rows, cols)) return 0;
  return (size_t)plane_trows(rows) * plane_tcols(cols >> 2) * 32;
}
extern "C" size_t tdr_cmap_words_total(int ncls, int rows, int cols) {
  const size_t off = tdr_cmap_plane_offset_words(ncls, rows, cols);
  return off ? off + tdr_cmap_plane_words(ncls, rows, cols) * (size_t)ncls + tdr_cmap_cmask_words(ncls, rows, cols) : 0;
}
__global__ __launch_bounds__(256) void cmap_kmask_kernel(const float* __restrict__ rec, int rows, int cols, int rf,
                                                         uint32_t* __restrict__ kmask) {
  const int colw = kmask_trows(rows) * 32;                            // words of a tile column
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // word t = tile column t / colw, row t % colw of it
  if (t >= (int64_t)colw * kmask_tcols(cols)) return;
  const int r = (int)(t % colw) - 32, c0 = (int)(t / colw) * 32 - 32;
  uint32_t bits = 0;
  if (r >= 0 && r < rows)
    for (int b = 0; b < 32; b++) {
      const int c = c0 + b;
      if (c >= 0 && c < cols && rec[((int64_t)(r + 1) * (cols + 2) + (c + 1)) * rf + rf - 1] != 0.f) bits |= 1u << b;
    }
  kmask[t] = bits;
}

// one thread per cell of one class plane (tile-major: coalesced stores); blockIdx.y = class
__global__ __launch_bounds__(256) void cmap_plane_kernel(const float* __restrict__ rec, int rows, int cols, int rf,
                                                         const unsigned* __restrict__ hash, const uint16_t* __restrict__ hidx,
                                                         size_t plane_words, uint16_t* __restrict__ planes) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y, tr_n = plane_trows(rows);
  if (t >= (int64_t)plane_words * 2) return;
  const int64_t tile = t >> 6;
  const int within = (int)(t & 63);
  const int tc = (int)(tile / tr_n), tr = (int)(tile - (int64_t)tc * tr_n);
  const int r = ((tr - 1) << 3) + (within >> 3), c = ((tc - 1) << 3) + (within & 7);
  uint16_t v = 0;   // outside the map: index 0 (distance 0), unknown — the guard record
  if (r >= 0 && r < rows && c >= 0 && c < cols) {
    const float* src = rec + ((int64_t)(r + 1) * (cols + 2) + (c + 1)) * rf;
    const unsigned bits = __float_as_uint(src[k]);
    unsigned h = cmap_hash(bits);
    while (hash[h] != bits) h = (h + 1) & (CMAP_HASH_SLOTS - 1);
    v = (uint16_t)(((unsigned)hidx[h] << 2) | (src[rf - 1] != 0.f ? 0x8000u : 0u));   // index * 4: the dictionary's byte offset
  }
  planes[(size_t)k * plane_words * 2 + t] = v;
}
// one thread per cell of the coarse mask plane
__global__ __launch_bounds__(256) void cmap_cmask_kernel(const float* __restrict__ rec, int rows, int cols, int rf,
                                                         size_t words, uint16_t* __restrict__ cmask) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)words * 2) return;
  const int tr_n = plane_trows(rows);
  const int64_t tile = t >> 6;
  const int within = (int)(t & 63);
  const int tc = (int)(tile / tr_n), tr = (int)(tile - (int64_t)tc * tr_n);
  const int R = ((tr - 1) << 3) + (within >> 3), C = ((tc - 1) << 3) + (within & 7);   // the 4 x 4 block (R, C)
  uint16_t v = 0;
  if (R >= 0 && C >= 0)
    for (int b = 0; b < 16; b++) {
      const int r = R * 4 + (b >> 2), c = C * 4 + (b & 3);
      if (r < rows && c < cols && rec[((int64_t)(r + 1) * (cols + 2) + (c + 1)) * rf + rf - 1] != 0.f) v |= (uint16_t)(1u << b);
    }
  cmask[t] = v;
}

// wide: 16-bit fields, two per dword (the wide form, below) instead of 10-bit fields, three per dword
__global__ __launch_bounds__(256) void cmap_pack_kernel(const float* __restrict__ rec, int rows, int cols, int rf,
                                                        int ncls, const unsigned* __restrict__ hash,
                                                        const uint16_t* __restrict__ hidx, CmapGeom g, int wide,
                                                        uint32_t* __restrict__ crec) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_tile = 4 << g.lc;
  const int64_t ntile = (int64_t)g.tiles_r * g.tiles_c;
  if (t >= ntile * per_tile) return;
  const int64_t tile = t / per_tile;
  const int within = (int)(t - tile * per_tile);
  const int tc = (int)(tile / g.tiles_r), tr = (int)(tile - (int64_t)tc * g.tiles_r);   // tiles run column by column
  const int r = ((tr - 1) << g.lc) + (within >> 2), c = ((tc - 1) << 2) + (within & 3);  // row-major inside the tile
  uint32_t w[4] = {0u, 0u, 0u, 0u};
  if (r >= 0 && r < rows && c >= 0 && c < cols) {
    const float* src = rec + ((int64_t)(r + 1) * (cols + 2) + (c + 1)) * rf;
    for (int k = 0; k < ncls; k++) {
      const unsigned v = __float_as_uint(src[k]);
      unsigned h = cmap_hash(v);
      while (hash[h] != v) h = (h + 1) & (CMAP_HASH_SLOTS - 1);   // every map value is in the set
      if (wide) w[k / 2] |= ((uint32_t)hidx[h] << 2) << (16 * (k & 1));
      else w[k / 3] |= (uint32_t)hidx[h] << (2 + 10 * (k % 3));
    }
    if (src[rf - 1] != 0.f) {                                      // known: bit 0 of the last dword — and of every
      w[g.cw - 1] |= 1u;                                           // other dword of a narrow record (a reader that needs
      if (!wide)                                                   // one class reads one dword, tdr_score_su.hip)
        for (int d = 0; d < g.cw - 1; d++) w[d] |= 1u;
    }
  }
  for (int d = 0; d < g.cw; d++) crec[t * g.cw + d] = w[d];
}

// The WIDE form: maps with more than 1024 distinct distance values (a `resolution` below ~0.8: min(50, resolution *
// sqrt(d2)) then takes more values) keep 16-bit fields, two per dword — 16 bytes per cell for 4-7 classes, tiles of 4 rows
// x 2 columns, a dictionary of up to TDR_CMAP_WIDE_MAX_DICT values (16 KB of LDS in the scoring kernel).  Half the bytes
// of the dense records instead of a quarter, still bit-identical operands.
extern "C" size_t tdr_cmap_wide_words_total(int ncls, int rows, int cols) {
  if (tdr_rec_floats(ncls) != 8) return 0;   // 4-7 classes
  const CmapGeom g = cmap_geom(4, rows, cols);
  return (size_t)g.tiles_r * g.tiles_c * 32;
}

// Shared by both forms: the map's distinct values -> dictionary (dict_out, TDR_CMAP_WIDE_MAX_DICT floats) and the hash
// slot -> index table; returns the number of dictionary entries in *nvals (0: not compactable at all).
static int cmap_dictionary(const tdr_map_desc* map, float* dict_out, void* workspace, hipStream_t s, int* nvals) {
  *nvals = 0;
  const int ncls = map->ncls, rf = map->rec_floats, rows = map->rows, cols = map->cols;
  unsigned* hash = reinterpret_cast<unsigned*>(workspace);
  uint16_t* hidx = reinterpret_cast<uint16_t*>(hash + CMAP_HASH_SLOTS);
  int* count = reinterpret_cast<int*>(hidx + CMAP_HASH_SLOTS);
  HIP_TRY(hipMemsetAsync(hash, 0xFF, sizeof(unsigned) * CMAP_HASH_SLOTS, s));
  HIP_TRY(hipMemsetAsync(count, 0, 2 * sizeof(int), s));
  const int64_t gcell = (int64_t)(rows + 2) * (cols + 2);
  hipLaunchKernelGGL(cmap_collect_kernel, dim3((unsigned)cdiv(gcell, 256)), dim3(256), 0, s, map->rec, gcell, rf, ncls,
                     hash, count);
  LAUNCH_CHECK("cmap_collect");
  int hcount[2] = {0, 0};
  std::vector<unsigned> hh(CMAP_HASH_SLOTS);
  HIP_TRY(hipMemcpyAsync(hcount, count, sizeof(hcount), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(hh.data(), hash, sizeof(unsigned) * CMAP_HASH_SLOTS, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (hcount[1] || hcount[0] > TDR_CMAP_WIDE_MAX_DICT) return TDR_OK;   // too many distinct values: dense records only
  // dictionary: +0.0f first (the guard record and unused fields are index 0), the rest ascending by bit pattern
  std::vector<unsigned> vals;
  vals.push_back(0u);
  for (unsigned v : hh)
    if (v != CMAP_EMPTY && v != 0u) vals.push_back(v);
  std::sort(vals.begin() + 1, vals.end());
  if ((int)vals.size() > TDR_CMAP_WIDE_MAX_DICT) return TDR_OK;
  std::vector<float> dict(TDR_CMAP_WIDE_MAX_DICT, 0.f);
  std::memcpy(dict.data(), vals.data(), vals.size() * sizeof(unsigned));
  std::vector<uint16_t> hx(CMAP_HASH_SLOTS, 0);
  for (int h = 0; h < CMAP_HASH_SLOTS; h++)
    if (hh[h] != CMAP_EMPTY && hh[h] != 0u)
      hx[h] = (uint16_t)(std::lower_bound(vals.begin() + 1, vals.end(), hh[h]) - vals.begin());
  // The dictionary as INTEGERS (narrow form): every value a non-negative multiple of 2^-q below 2^32 - then a product
  // sum of integer scan counts with these values is an integer sum: exact, whatever the order it is added up in
  // (tdr_score_su.hip, tdr_score_ray.hip).  Entries [1024, 2048) = value * 2^q as u32, [2048] = q, [2049] = 1 when
  // the map qualifies (distance maps of resolution >= ~0.2: min(50, resolution * sqrt(d2)) >= 2^-3 ... 50).
  if ((int)vals.size() <= TDR_CMAP_MAX_DICT) {
    int q = 0;
    bool ok = true;
    for (size_t i = 1; i < vals.size() && ok; i++) {
      float v;
      std::memcpy(&v, &vals[i], 4);
      if (!(v > 0.f) || !(v <= 3.402823466e+38f)) { ok = false; break; }
      int e;
      const double m = std::frexp((double)v, &e);                     // v = m 2^e, m in [0.5, 1): m 2^24 is an integer
      uint32_t mi = (uint32_t)std::ldexp(m, 24);
      int tz = 0;
      while (!(mi & 1u)) { mi >>= 1; tz++; }
      q = std::max(q, 24 - tz - e);                                    // v = mi 2^(e - 24 + tz)
    }
    std::vector<uint32_t> di(TDR_CMAP_MAX_DICT, 0u);
    for (size_t i = 1; i < vals.size() && ok; i++) {
      float v;
      std::memcpy(&v, &vals[i], 4);
      const double x = std::ldexp((double)v, q);
      if (!(x < 4294967296.0)) { ok = false; break; }
      di[i] = (uint32_t)x;
    }
    std::memcpy(dict.data() + TDR_CMAP_MAX_DICT, di.data(), sizeof(uint32_t) * TDR_CMAP_MAX_DICT);
    const uint32_t tail[2] = {(uint32_t)q, ok ? 1u : 0u};
    std::memcpy(dict.data() + 2 * TDR_CMAP_MAX_DICT, tail, sizeof(tail));
  }
  HIP_TRY(hipMemcpyAsync(dict_out, dict.data(), sizeof(float) * TDR_CMAP_WIDE_MAX_DICT, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(hidx, hx.data(), sizeof(uint16_t) * CMAP_HASH_SLOTS, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));   // dict / hx live on this stack frame until the copies are done
  *nvals = (int)vals.size();
  return TDR_OK;
}
static int cmap_pack(tdr_map_desc* map, uint32_t* crec_out, float* dict_out, void* workspace, hipStream_t s, int cw,
                     int wide, int nvals) {
  const CmapGeom g = cmap_geom(cw, map->rows, map->cols);
  if ((uint64_t)g.tiles_r * g.tiles_c * 128 > 0xFFFFFFF0ull || g.tiles_r >= (1 << 16)) return TDR_OK;   // 32-bit offsets
  unsigned* hash = reinterpret_cast<unsigned*>(workspace);
  uint16_t* hidx = reinterpret_cast<uint16_t*>(hash + CMAP_HASH_SLOTS);
  const int64_t nrec = (int64_t)g.tiles_r * g.tiles_c * (4 << g.lc);
  hipLaunchKernelGGL(cmap_pack_kernel, dim3((unsigned)cdiv(nrec, 256)), dim3(256), 0, s, map->rec, map->rows, map->cols,
                     map->rec_floats, map->ncls, (const unsigned*)hash, (const uint16_t*)hidx, g, wide, crec_out);
  LAUNCH_CHECK("cmap_pack");
  if (!wide) {   // the known mask behind the tiles
    const int64_t nwords = (int64_t)kmask_trows(map->rows) * kmask_tcols(map->cols) * 32;
    hipLaunchKernelGGL(cmap_kmask_kernel, dim3((unsigned)cdiv(nwords, 256)), dim3(256), 0, s, map->rec, map->rows, map->cols,
                       map->rec_floats, crec_out + (size_t)g.tiles_r * g.tiles_c * 32);
    LAUNCH_CHECK("cmap_kmask");
    const size_t pw = tdr_cmap_plane_words(map->ncls, map->rows, map->cols);
    if (pw) {   // the class planes behind the mask
      uint16_t* planes = reinterpret_cast<uint16_t*>(crec_out + tdr_cmap_plane_offset_words(map->ncls, map->rows, map->cols));
      hipLaunchKernelGGL(cmap_plane_kernel, dim3((unsigned)cdiv((int64_t)pw * 2, 256), (unsigned)map->ncls), dim3(256), 0, s,
                         map->rec, map->rows, map->cols, map->rec_floats, (const unsigned*)hash, (const uint16_t*)hidx, pw,
                         planes);
      LAUNCH_CHECK("cmap_plane");
      const size_t cw16 = tdr_cmap_cmask_words(map->ncls, map->rows, map->cols);
      hipLaunchKernelGGL(cmap_cmask_kernel, dim3((unsigned)cdiv((int64_t)cw16 * 2, 256)), dim3(256), 0, s, map->rec, map->rows,
                         map->cols, map->rec_floats, cw16, planes + (size_t)map->ncls * pw * 2);
      LAUNCH_CHECK("cmap_cmask");
    }
  }
  HIP_TRY(hipStreamSynchronize(s));
  map->crec = crec_out;
  map->dict = dict_out;
  map->dict_n = nvals;
  map->cwords = cw;
  return TDR_OK;
}

// Builds the compact form of map->rec into crec_out (tdr_cmap_words_total dwords) / dict_out (TDR_CMAP_WIDE_MAX_DICT
// floats), workspace = TDR_CMAP_WORKSPACE_BYTES of device scratch, and fills map->crec / dict / dict_n / cwords.
// Load-time work: synchronises with `stream` (the dictionary is sorted on the host).  A map that has no compact form
// leaves map->cwords = 0 and returns TDR_OK; if it has too many distinct values for this (narrow) form but a wide form
// exists, map->dict_n = -(number of distinct values): the caller then provides tdr_cmap_wide_words_total dwords to
// tdr_k_compact_map_wide.
extern "C" int tdr_k_compact_map(tdr_map_desc* map, uint32_t* crec_out, float* dict_out, void* workspace, void* stream) {
  if (!map || !map->rec || !crec_out || !dict_out || !workspace) return fail(TDR_ERR_ARG, "compact_map: null pointer");
  map->crec = nullptr; map->dict = nullptr; map->dict_n = 0; map->cwords = 0;
  const int cw = tdr_cmap_words(map->ncls);
  if (!cw) return TDR_OK;
  int nvals = 0;
  if (int rc = cmap_dictionary(map, dict_out, workspace, (hipStream_t)stream, &nvals)) return rc;
  if (nvals == 0) return TDR_OK;
  if (nvals > TDR_CMAP_MAX_DICT) {
    if (tdr_cmap_wide_words_total(map->ncls, map->rows, map->cols)) map->dict_n = -nvals;
    return TDR_OK;
  }
  return cmap_pack(map, crec_out, dict_out, workspace, (hipStream_t)stream, cw, 0, nvals);
}
// The wide form (see above) into wrec_out (tdr_cmap_wide_words_total dwords); everything else as tdr_k_compact_map.
// For maps tdr_k_compact_map answered with a negative dict_n (more than TDR_CMAP_MAX_DICT distinct values); others are
// left without (cwords = 0).  A wide record is told from a narrow one by map->dict_n > TDR_CMAP_MAX_DICT.
extern "C" int tdr_k_compact_map_wide(tdr_map_desc* map, uint32_t* wrec_out, float* dict_out, void* workspace,
                                      void* stream) {
  if (!map || !map->rec || !wrec_out || !dict_out || !workspace) return fail(TDR_ERR_ARG, "compact_map_wide: null pointer");
  map->crec = nullptr; map->dict = nullptr; map->dict_n = 0; map->cwords = 0;
  if (!tdr_cmap_wide_words_total(map->ncls, map->rows, map->cols)) return TDR_OK;
  int nvals = 0;
  if (int rc = cmap_dictionary(map, dict_out, workspace, (hipStream_t)stream, &nvals)) return rc;
  if (nvals <= TDR_CMAP_MAX_DICT) return TDR_OK;   // the narrow form's case: a wide record is told by dict_n > 1024
  return cmap_pack(map, wrec_out, dict_out, workspace, (hipStream_t)stream, 4, 1, nvals);
}

// Compact records back to dense ones: rec_out [(rows+2)*(cols+2)][rf] exactly as tdr_k_pack_map writes them — the
// round-trip check of the encoding (tests/test_gpu_parity.py::test_compact_map_round_trip).
__global__ __launch_bounds__(256) void cmap_unpack_kernel(const uint32_t* __restrict__ crec, const float* __restrict__ dict,
                                                          int rows, int cols, int rf, int ncls, CmapGeom g, int wide,
                                                          float* __restrict__ rec) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t gcols = cols + 2, gcell = (int64_t)(rows + 2) * gcols;
  if (idx >= gcell) return;
  const int r = (int)(idx / gcols) - 1, c = (int)(idx % gcols) - 1;
  const int64_t tile = (int64_t)((c >> 2) + 1) * g.tiles_r + ((r >> g.lc) + 1);
  const int within = ((r & ((1 << g.lc) - 1)) << 2) | (c & 3);
  const uint32_t* w = crec + (tile * (4 << g.lc) + within) * g.cw;
  float* o = rec + idx * rf;
  for (int k = 0; k < rf; k++) o[k] = 0.f;
  for (int k = 0; k < ncls; k++)
    o[k] = wide ? dict[(w[k / 2] >> (2 + 16 * (k & 1))) & 0x3FFFu] : dict[(w[k / 3] >> (2 + 10 * (k % 3))) & 1023u];
  const float known = (w[g.cw - 1] & 1u) ? 1.f : 0.f;
  o[rf - 1] = known;
  if (tdr_has_kslot(ncls, rf)) o[rf - 2] = known;
}
extern "C" int tdr_k_unpack_compact_map(const tdr_map_desc* map, float* rec_out, void* stream) {
  if (!map || !map->crec || !map->dict || !rec_out || !map->cwords) return fail(TDR_ERR_ARG, "unpack_compact_map: no compact map");
  const CmapGeom g = cmap_geom(map->cwords, map->rows, map->cols);
  const int64_t gcell = (int64_t)(map->rows + 2) * (map->cols + 2);
  const int wide = (map->cwords == 4 && map->rec_floats == 8 && map->dict_n > TDR_CMAP_MAX_DICT) ? 1 : 0;
  hipLaunchKernelGGL(cmap_unpack_kernel, dim3((unsigned)cdiv(gcell, 256)), dim3(256), 0, (hipStream_t)stream, map->crec,
                     map->dict, map->rows, map->cols, map->rec_floats, map->ncls, g, wide, rec_out);
  LAUNCH_CHECK("cmap_unpack");
  return TDR_OK;
}
