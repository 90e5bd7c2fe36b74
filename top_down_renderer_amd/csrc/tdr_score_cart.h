// tdr_score_cart.h — argument block of the Cartesian scoring kernels (tdr_score.hip: the general kernel; tdr_score_cart.hip:
// the kernel that skips empty scan bins) and the host interface of the latter.
#ifndef TDR_SCORE_CART_H_
#define TDR_SCORE_CART_H_
#include "tdr_common.h"

struct CartArgs {
  const float* rec;
  int map_rows, map_cols;
  float resolution;
  const float* scan_pk;  // [cols][rows][rf]
  int rows, cols;        // window (image) shape
  float res;
  const float* st;
  int64_t cap, n;
  const int32_t* order;
  int cpc, nchunks;      // window columns per chunk
  int64_t npad;
  float* part;
  int libm_fma;          // which build of sinf / cosf the host's libm runs (tdr_sincosf.h)
  const uint32_t* crec;  // compact form of the records (COMPACT instantiations)
  const float* dict;
  int dict_n;
  int ctiles_r;
  // tdr_score_cart.hip only
  const uint32_t* desc;  // [cols][rows][4]: scan descriptor of every bin (cart_prep_kernel)
  unsigned kmask_off;    // byte offset of the known mask from crec
  int kmask_row;         // bytes of one tile column of it (kmask_offset, tdr_score_dev.h)
  // the integer form (tdr_score_cart.hip: score_cart_skip_kernel<.., INT>, score_cart_ray_kernel)
  const uint32_t* dict_int = nullptr;   // the dictionary as integers (tdr_cmap.hip)
  const int32_t* flags = nullptr;       // device words of int_form_off (tdr_score_dev.h); NULL: no gate
  int run_if_int = 0;                   // with flags: 1 = run only while the integer form is on, 0 = only while it is off
  const int32_t* count = nullptr;       // device word: slots of `order` this launch covers (the dense share); NULL: n
  int ncls = 0;
  int pkcol = 0;                        // != 0: a bin with one class fetches the cell of the class's PLANE (tdr_cmap.hip; plane_offset:
                                        // bytes of a tile column - 16), its constant in descriptor word [2] — the integer form's dense kernel
};

// dwords of workspace the descriptors take (behind the partial sums of tdr_score_cart_workspace_floats)
static inline int64_t tdr_cart_desc_words(int rows, int cols) { return (int64_t)rows * cols * 4 + 64; }
// whether the skipping kernel applies (tdr_config_cart_skip, a map with narrow compact records of at most 11 classes)
bool tdr_cart_skip_ok(const tdr_map_desc* map, int rf);
// descriptors + the scoring kernel; `a` complete but for desc / kmask_*; desc_ws: tdr_cart_desc_words dwords
int tdr_cart_skip_launch(CartArgs a, const tdr_map_desc* map, int rf, uint32_t* desc_ws, hipStream_t s);
// The INTEGER form of a Cartesian launch (exact class sums, like the polar one's: tdr_score_su.h): dense particles through
// score_cart_skip_kernel with integer accumulators, scattered ones one wave each through score_cart_ray_kernel; the float
// skipping kernel behind them scores the launch when the scan or the map has no integer form.  ws_words: 4-byte words of
// workspace behind the partial sums; the functions below carve it.
bool tdr_cart_int_ok(const tdr_map_desc* map, int rf, int rows, int cols, int64_t n_total);
int64_t tdr_cart_int_words(int rows, int cols, int64_t n);
struct CartIntOut {
  const int32_t* slots;    // slot list: dense particles (padded to whole waves), then the scattered ones
  const int32_t* counts;   // {dense slots, scattered particles, both}
  const int32_t* flags;    // int_form_off
  int ray_split;
  int nchunks_dense;       // chunk rows of partial sums a DENSE slot has (score_cart_su_kernel walks wider chunks)
  int64_t npad;            // slot capacity = stride of the integer partial sums
};
int tdr_cart_int_launch(CartArgs a, const tdr_map_desc* map, int rf, uint32_t* desc_ws, int32_t* ws, float span,
                        hipStream_t s, CartIntOut* out);
#endif  // TDR_SCORE_CART_H_
