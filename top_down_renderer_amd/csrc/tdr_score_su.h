// tdr_score_su.h — host interface of the integer polar scoring path: the shift-uniform kernel for dense particles
// (tdr_score_su.hip) and the ray-mapped kernel for scattered ones (tdr_score_ray.hip), used by tdr_score.hip.
#ifndef TDR_SCORE_SU_H_
#define TDR_SCORE_SU_H_
#include "tdr_common.h"

// slots a launch of n particles can need: every non-empty heading bin rounded up to whole waves
static inline int64_t su_npad(int64_t n, int nb) { return cdiv(n + 63 * std::min<int64_t>(nb, n), 64) * 64; }
// whether the path applies to a launch shape (tdr_config_shift_uniform, padding economics, kernel constraints)
bool tdr_su_shape_ok(int nb, int nr, int group, int64_t n_total);
// The path's share of the scoring workspace, in 4-byte words, carved in this order (each part 256-byte aligned):
struct SuWs {
  int64_t tab_su, desc, bbox, keys_in, keys_out, vals_in, vals_out, ints, slots, sort_tmp, ray_tab, ray_desc, ray_multi, ray_rad, total;
  int64_t slots2, wave_tmp;   // the re-routing pass (su_wave_far_kernel ...): a second slot list, four ints per wave
  // ints: [cnt nb + 1][start nb + 1][slot_start nb + 1][counts 3][n_multi][inexact][mass bound][table is not its factors]
  //       [the counts before the re-routing pass 3]  (int_form_off)
};
#define TDR_SU_TAIL_INTS 10       // the words of `ints` behind the three per-key tables
#define TDR_RAY_MAX_SPLIT 8       // waves a particle's window may be split over (tdr_ray_splits): chunk rows of `part`
SuWs tdr_su_ws(int nb, int nr, int group, int64_t n);

struct SuLaunch {
  const tdr_map_desc* map;   // with a narrow compact form
  const float* tab;          // [P][2]: (tab*scale)*res when uniform_scale, else the table itself
  bool uniform_scale;
  const float* scan_pk;
  int nb, nr, rf;
  float res;
  const float* st;
  int64_t cap, n;
  const int32_t* perm;       // caller's locality order (NULL = identity)
  int group, nchunks;
  int64_t npad;              // slot capacity = stride of part (su_npad)
  float* part;               // integer partial sums, [chunks][2 ncls + 2][npad] words (tdr_score_su.hip)
  int ray_split;             // waves per scattered particle (tdr_ray_splits)
  const float* fac;          // the table's factors (tdr_polar_factors_host) or NULL
  float uscale;              // the caller's uniform scale (<= 0: none)
  int32_t* ws;               // tdr_su_ws(...).total words
  float span;                // map cells the 64 locality neighbours of a dense particle may span (tdr_su_span_begin)
  float wave_span = 0.f;     // > 0: a wave of the heading-bin order whose OWN 64 particles spread over more cells than this goes
                             // to the ray-mapped kernel after all (su_wave_far_kernel; tdr_config_tuning("su_wave_span"))
};
// ordering passes + descriptors (everything but the scoring kernels).  slots_out: the slot list — the dense particles
// by heading bin, every bin padded to whole waves (-1), then the sparse particles in the caller's order (su_key_kernel);
// counts_out: device words {slots of the heading bins, sparse particles behind them, both together (what finalize walks)}
int tdr_su_prepare(const SuLaunch& L, const SuWs& W, hipStream_t s, const int32_t** slots_out, const int32_t** counts_out);
// the ordering passes alone (L.st, cap, n, perm, nb, span, ws; nb == 1: no heading bins — the Cartesian score)
int tdr_su_order(const SuLaunch& L, const SuWs& W, hipStream_t s, const int32_t** slots_out, const int32_t** counts_out);
// The span for this launch.  With a fixed span (tdr_config_shift_uniform_span, TDR_SU_SPAN) that one; otherwise it is tuned
// while the filter runs: a few candidates are timed over one launch each (events on `s` around the whole scoring call,
// tdr_su_span_end closes the measurement), the fastest is kept, and the trial is repeated every few thousand launches
// (`shape`: anything that identifies the launch's sizes; a change restarts the trial).  The span only routes particles
// between two kernels that produce identical partial sums: results do not depend on it.
// The tuner's state belongs to the caller's tdr_score_ctx (tdr.h); without one the span is the configured / default one.
// Nothing here waits on the host: a trial whose events have not completed when the next call arrives is repeated.
struct SpanTuner {
  int64_t shape = -1;
  int phase = -2;                     // < 0: skipping; < number of candidates: timing that candidate; else settled
  int settled_launches = 0;
  int trial = 0;                      // timed calls of the current candidate so far
  float best = 16.f;                  // the span in use: what the last COMPLETED round of trials found fastest
  float round_best = 16.f, best_ms = 3.0e38f;   // the fastest candidate of the round in progress, and its time
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool open = false, pending = false;
  int64_t trial_calls = 0;            // launches that ran at a candidate span so far (diagnostics: tdr_score_ctx_trial_calls)
};
float tdr_su_span_begin(SpanTuner* t, int64_t shape, hipStream_t s);
void tdr_su_span_end(SpanTuner* t, hipStream_t s);
float tdr_su_wave_span();   // tdr_config_tuning("su_wave_span")
// the shift-uniform kernel over the heading bins' slots
int tdr_su_score(const SuLaunch& L, const SuWs& W, hipStream_t s);
// the ray-mapped kernel (tdr_score_ray.hip): whether the map carries what it reads (class planes, integer dictionary
// space), the split of a window over waves, its tables / the `inexact` flag (before either scoring kernel), the launch
bool tdr_ray_map_ok(const tdr_map_desc* map);
int64_t tdr_ray_padded_samples(int nb, int nr);   // window samples with every direction padded to whole blocks of steps
int tdr_ray_splits(int nb, int nr, int64_t n, bool block_major = false);
bool tdr_ray_block_major(const SuLaunch& L);   // (L.fac set)
int tdr_ray_prepare(const SuLaunch& L, const SuWs& W, hipStream_t s);
int tdr_ray_score(const SuLaunch& L, const SuWs& W, hipStream_t s);
#endif  // TDR_SCORE_SU_H_
