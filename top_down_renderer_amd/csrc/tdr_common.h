// tdr_common.h — shared by the HIP translation units of libtdr_hip.so (not installed; the public header is include/tdr.h).
//
// Compile with -ffp-contract=off: every float expression that decides a bin / cell index must round exactly like
// the reference's x86-64 code (no FMA contraction); FMAs are spelled out (__builtin_fmaf) where they are wanted.
//
// Kernel map (reference loop nest -> kernel -> file), see DESIGN.md:
//   K0 pack_map_kernel, ingest_*   class_maps_/class_mask_ (top_down_map.h:77-79), top_down_map.cpp:116-144,289-326  tdr_map.hip
//   K1 raster_keys/raster_kernel   scan_renderer_polar.cpp:93-108 / scan_renderer.cpp:65-77                          tdr_raster.hip
//   K2 score_polar_kernel          top_down_map_polar.cpp:28-52 + state_particle.cpp:132-143 (lane = particle)       tdr_score.hip
//      score_cart_kernel           top_down_map.cpp:429-459 + state_particle.cpp:112-155
//      score_finalize_kernel       state_particle.cpp:117-120,136-139,154,161-176,212
//      score_init(_mfma)_kernel    state_particle.cpp:195-206
//   K3 propagate_kernel            state_particle.cpp:57-78                                                          tdr_filter.hip
//   K4 update_weights / uw_pass*   particle_filter.cpp:107-147
//   K5 prefix kernels              particle_filter.cpp:175-183 (the serial float32 running sum, bit-exact)            tdr_prefix.hip
//      resample / gather_states    particle_filter.cpp:172-185                                                       tdr_filter.hip
//   K6 mean_cov / mc_* kernels     particle_filter.cpp:191-236, 343-357
#ifndef TDR_COMMON_H_
#define TDR_COMMON_H_
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

#include "tdr.h"

// tuning knobs (compile-time)
#ifndef TDR_SCORE_U
#define TDR_SCORE_U 4          // samples whose loads are kept in flight together in the scoring loop
#endif
#ifndef TDR_INIT_SCAN_LDS
#define TDR_INIT_SCAN_LDS 1   // init search: candidates' scan records via LDS broadcast (1) or the scalar cache (0)
#endif
#ifndef TDR_XCD_SWIZZLE
#define TDR_XCD_SWIZZLE 0
#endif
#ifndef TDR_OOB_ALIAS
#define TDR_OOB_ALIAS 1      // all out-of-bounds samples read one guard record (A/B on MI355X: -11 % on config 2)
#endif

// error plumbing (tdr_core.hip): the message behind tdr_last_error(), per host thread
int tdr_fail(int code, const char* fmt, ...);
#define fail tdr_fail
#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(TDR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define LAUNCH_CHECK(name)                                                                   \
  do {                                                                                       \
    hipError_t e_ = hipGetLastError();                                                       \
    if (e_ != hipSuccess) return fail(TDR_ERR_HIP, "launch %s: %s", name, hipGetErrorString(e_)); \
  } while (0)

// tdr_core.hip: 1 if the device should evaluate sinf / cosf like glibc's FMA-contracted build, 0 like its plain build
// (whichever the host's libm runs; tdr_sincosf.h)
int tdr_libm_fma();

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// tdr_prefix.hip: final value of a serial float32 chain over the raw weights (kind 0: sum of the non-NaN weights;
// kind 1: float-accumulated squared deviations of the weights below *mean_dev), see there
int tdr_chain_total(const float* raw, const float* mean_dev, int kind, int64_t n, float* total_out, void* workspace,
                    hipStream_t st);
// tdr_prefix.hip: particle_filter.cpp:107-147 for n <= 32768 in one launch, both serial chains exact
int tdr_uw_small(const float* raw, const float* last_dist, int64_t n, float* w, float* info, hipStream_t st);
// a record with a spare slot (ncls + 2 <= rf) carries `known` twice: slot rf-2 pairs with a constant 1 of the scan record
__host__ __device__ inline bool tdr_has_kslot(int ncls, int rf) { return ncls + 2 <= rf; }
// diagnostics: 16 device counters while tdr_profile_enable(1) is in force, else NULL (tdr_score.hip, tdr_profile_variants)
uint32_t* tdr_profile_stats_ptr();
#endif  // TDR_COMMON_H_
