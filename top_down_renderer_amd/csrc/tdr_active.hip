// tdr_active.hip — ActiveLocalizer (src/active_localizer.cpp): where should the robot go so that the hypotheses of its pose
// become distinguishable?  For every candidate displacement (distance, direction) each hypothesis' local map is taken at
// its displaced pose (getLocalMap :22-42) and the candidates are ranked by the mean pairwise L1 difference of those maps
// (computeTotalDifference :7-20).  The reference evaluates candidates one after the other on the CPU; here all of them
// are one launch: workgroup = candidate, thread = window sample, and the local maps are never materialised — a thread
// gathers the K hypotheses' cells of its sample and adds up |v_i - v_j| over the pairs, class by class.
// (Dead at the reference's call sites, src/particle_filter.cpp:77-78,316)
#include "tdr_score_dev.h"

#define ACTIVE_MAX_K 32   // = TDR_GMM_MAX_K

struct ActiveArgs {
  const float* rec;
  int rows, cols, rf, ncls;
  float resolution;
  const float* tab;        // [nb * nr][2]
  int nb, nr;
  float res;               // getLocalMap's res (2 in the reference, :30)
  const float* centres;    // [ncand][K][2]: x, y of hypothesis i displaced by candidate q (:62-63, evaluated on the host)
  const int32_t* shifts;   // [K]: rot_shift of hypothesis i (:33-36)
  int K;
  double* sums;            // [ncand]: sum over pairs i > j, classes, samples of |L_i - L_j|
};

__global__ __launch_bounds__(256) void active_diff_kernel(ActiveArgs a) {
  const int q = blockIdx.x;
  const int P = a.nb * a.nr;
  __shared__ float s_off[ACTIVE_MAX_K][2];
  __shared__ int s_shift[ACTIVE_MAX_K];
  __shared__ double s_red[256];
  if ((int)threadIdx.x < a.K) {
    const float* c = a.centres + ((int64_t)q * a.K + threadIdx.x) * 2;
    s_off[threadIdx.x][0] = c[1] / a.resolution;   // top_down_map_polar.cpp:29: row offset from y
    s_off[threadIdx.x][1] = c[0] / a.resolution;   // :30
    s_shift[threadIdx.x] = a.shifts[threadIdx.x];
  }
  __syncthreads();
  double acc = 0;
  for (int k = threadIdx.x; k < P; k += 256) {
    const int j = k / a.nb, row = k - j * a.nb;
    const float* cellp[ACTIVE_MAX_K];
#pragma unroll 1
    for (int i = 0; i < a.K; i++) {
      // local_map row `row` = window row (row - shift) mod nb (active_localizer.cpp:38-41)
      int src = row - s_shift[i];
      src += src < 0 ? a.nb : 0;
      const int64_t ks = (int64_t)src + (int64_t)a.nb * j;
      float p0 = (a.tab[2 * ks] * 1.f) * a.res + s_off[i][0];       // top_down_map_polar.cpp:28-30 with scale = 1 (:78-82)
      float p1 = (a.tab[2 * ks + 1] * 1.f) * a.res + s_off[i][1];
      p0 = __builtin_amdgcn_fmed3f(p0, -1.f, (float)a.rows);
      p1 = __builtin_amdgcn_fmed3f(p1, -1.f, (float)a.cols);
      const int ri = round_half_away_clamped(p0), ci = round_half_away_clamped(p1);   // :31
      // the guard ring of the record grid is all zero: the reference's out-of-bounds value (:41)
      cellp[i] = a.rec + ((int64_t)(ri + 1) * (a.cols + 2) + (ci + 1)) * a.rf;
    }
    for (int c = 0; c < a.ncls; c++) {
      float v[ACTIVE_MAX_K];
#pragma unroll 1
      for (int i = 0; i < a.K; i++) v[i] = cellp[i][c];
      float s = 0.f;
#pragma unroll 1
      for (int i = 1; i < a.K; i++)
        for (int jj = 0; jj < i; jj++) s += fabsf(v[i] - v[jj]);   // :14
      acc += (double)s;
    }
  }
  s_red[threadIdx.x] = acc;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.sums[q] = s_red[0];
}

extern "C" int tdr_k_active_diffs(const tdr_map_desc* map, const float* tab, int nb, int nr, float res, const float* centres,
                                  const int32_t* shifts, int K, int ncand, double* sums_out, void* stream) {
  if (!map || !map->rec || !tab || !centres || !shifts || !sums_out) return fail(TDR_ERR_ARG, "active_diffs: null pointer");
  if (K < 1 || K > ACTIVE_MAX_K) return fail(TDR_ERR_ARG, "active_diffs: %d hypotheses (1 .. %d)", K, ACTIVE_MAX_K);
  if (nb < 1 || nr < 1 || ncand < 1) return fail(TDR_ERR_ARG, "active_diffs: bad shape");
  if (!(map->resolution > 0.f)) return fail(TDR_ERR_ARG, "active_diffs: map resolution must be > 0");
  ActiveArgs a;
  a.rec = map->rec; a.rows = map->rows; a.cols = map->cols; a.rf = map->rec_floats; a.ncls = map->ncls;
  a.resolution = map->resolution; a.tab = tab; a.nb = nb; a.nr = nr; a.res = res;
  a.centres = centres; a.shifts = shifts; a.K = K; a.sums = sums_out;
  hipLaunchKernelGGL(active_diff_kernel, dim3((unsigned)ncand), dim3(256), 0, (hipStream_t)stream, a);
  LAUNCH_CHECK("active_diff");
  return TDR_OK;
}

// The candidate loop of getBestRelPos (:55-77) on the host: distances 50, 75, ... while the best difference stays below
// 6000 and dist < 150; directions `for (float theta = 0; theta < 2 pi; theta += pi / 8)` — the float loop as written (a
// 17th direction when 16 steps round below 2 pi).  Fills centres [<= 4 * 17][K][2], thetas / dists per candidate, and
// shifts [K]; returns the number of directions per distance.
extern "C" int tdr_active_candidates_host(const float* preds, int K, int nb, float* centres, float* dists, float* thetas,
                                          int32_t* shifts, int* ntheta_out, int* ndist_out) {
  if (!preds || !centres || !dists || !thetas || !shifts || !ntheta_out || !ndist_out || K < 1)
    return fail(TDR_ERR_ARG, "active_candidates: bad arguments");
  for (int i = 0; i < K; i++) {
    int s = (int)std::round((double)(preds[3 * i + 2] * (float)nb / 2) / M_PI);   // :33
    while (s >= nb) s -= nb;                                                       // :35-36
    while (s < 0) s += nb;
    shifts[i] = s;
  }
  int nd = 0, nt = 0;
  for (float dist = 50; dist < 150; dist += 25) {   // :58, 76 (the early exit on best_diff is the caller's)
    int t = 0;
    for (float theta = 0; theta < 2 * M_PI; theta += M_PI / 8) {   // :59
      if (t >= 17) break;
      for (int i = 0; i < K; i++) {
        const float ang = theta + preds[3 * i + 2];
        float* c = centres + ((size_t)(nd * 17 + t) * K + i) * 2;
        c[0] = preds[3 * i] + dist * cosf(ang);       // :62-63
        c[1] = preds[3 * i + 1] + dist * sinf(ang);
      }
      dists[nd * 17 + t] = dist;
      thetas[nd * 17 + t] = theta;
      t++;
    }
    nt = t;
    nd++;
    if (nd >= 4) break;
  }
  *ntheta_out = nt;
  *ndist_out = nd;
  return TDR_OK;
}
