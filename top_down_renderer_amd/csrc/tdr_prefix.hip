// tdr_prefix.hip — the serial float32 running sum of the resampling step, reproduced bit-exactly in parallel.
#include "tdr_common.h"

// ------------------------------------------------------------------------------------------------------------------
// K5: resample.  The running sum of particle_filter.cpp:179 is a serial float32 chain,
//     prefix_j = fl(prefix_{j-1} + w_j),
// and "bit-exact resample indices" needs exactly these values.  Two kernels produce them:
//
//  * prefix_serial_kernel — one wave, lane 0 performs the additions in index order (weights staged through LDS).
//    Simple, ~10 ns per element; kept as the reference implementation and for small n.
//  * prefix_exact_kernel — the same values computed in parallel.  While the running sum r stays inside one binade
//    [2^e, 2^(e+1)) it is an integer multiple R*u of u = 2^(e-23) and fl(r + w) = (R + q)*u, where q is w/u rounded
//    to nearest — a pure integer increment that depends on r only when w/u ends in exactly .5 (tie to even: the parity
//    of R + floor(w/u)).  So per tile of 4096 weights the workgroup (i) classifies every weight into an integer
//    increment / tie / "needs a real float add" (NaN, inf, larger than the binade), (ii) prefix-sums the increments,
//    (iii) resolves the (rare) ties in order on one thread, (iv) prefix-sums the tie corrections, (v) finds the first
//    element at which the sum leaves the binade or a real add is needed, commits everything before it, performs that
//    one addition in float arithmetic and restarts behind it in the new binade.  A running sum of 1 crosses ~24
//    binades, so a million weights take a few hundred workgroup passes instead of a million dependent additions.
//    tests/test_gpu_parity.py compares both kernels with the CPU chain bit for bit on random and adversarial inputs.
//
// The running maximum makes "first j with prefix_j > sample" searchable even when weights are negative (NaN fill, :133).
#define TDR_PFX_BLOCK 4096  // elements staged in LDS per pass (64 per lane)
__global__ __launch_bounds__(64) void prefix_serial_kernel(const float* __restrict__ w, int64_t n,
                                                           float* __restrict__ runmax) {
  __shared__ float4 buf4[TDR_PFX_BLOCK / 4];
  float* buf = reinterpret_cast<float*>(buf4);
  const int lane = threadIdx.x;
  float run = 0.f;        // the serial chain lives in lane 0
  float carry_max = -INFINITY;
  for (int64_t base = 0; base < n; base += TDR_PFX_BLOCK) {
    const int cnt = (int)min((int64_t)TDR_PFX_BLOCK, n - base);
    // coalesced stage-in (pad with zeros: x + 0 == x)
    for (int t = lane; t < TDR_PFX_BLOCK; t += 64) buf[t] = (t < cnt) ? w[base + t] : 0.f;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      const int q4 = (cnt + 3) / 4;
#pragma unroll 4
      for (int q = 0; q < q4; q++) {
        float4 v = buf4[q];
        run = run + v.x; v.x = run;   // particle_filter.cpp:179, one float add per weight, index order
        run = run + v.y; v.y = run;
        run = run + v.z; v.z = run;
        run = run + v.w; v.w = run;
        buf4[q] = v;
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // running maximum of the block's prefix values (max is associative: any order is exact)
    float m = -INFINITY;
    float loc[64];
#pragma unroll
    for (int t = 0; t < 64; t++) {
      float x = buf[lane * 64 + t];
      if (x != x) x = -INFINITY;  // a NaN prefix never exceeds a threshold (`running_sum > sample` is false)
      m = fmaxf(m, x);
      loc[t] = m;
    }
    float incl = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      float t = __shfl_up(incl, o, 64);
      if (lane >= o) incl = fmaxf(incl, t);
    }
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = -INFINITY;
    excl = fmaxf(excl, carry_max);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 64; t++) buf[lane * 64 + t] = fmaxf(loc[t], excl);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < cnt; t += 64) runmax[base + t] = buf[t];
    carry_max = fmaxf(__shfl(incl, 63, 64), carry_max);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- exact parallel prefix -------------------------------------------------------------------------------------------
#define PFX_THREADS 1024
#define PFX_K 8
#define PFX_TILE (PFX_THREADS * PFX_K)
#define PFX_TIE_CAP 2048   // ties resolved per pass; a (never observed) denser tile is simply cut at that tie
#define PFX_HEAD 2048      // leading elements added one by one: the running sum crosses most of its binades here

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global store
// (~1 us each time); in the prefix kernels the threads exchange data through LDS alone — global memory is read-only
// input (w, chunk headers of an earlier launch) or write-only output — so the store wait would be pure latency.
__device__ __forceinline__ void pfx_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// inclusive block scan (sum) of one value per thread; `sh` holds one slot per wave
template <int NT, class T>
__device__ __forceinline__ T pfx_block_scan(T v, T* sh, T& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  pfx_sync();
  if (lane == 63) sh[wave] = v;
  pfx_sync();
  T off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < NT / 64; k++) {
    const T x = sh[k];
    if (k < wave) off += x;
    tot += x;
  }
  total = tot;
  return v + off;
}
template <int NT>
__device__ __forceinline__ float pfx_block_scan_max(float v, float* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (lane >= o) v = fmaxf(v, t);
  }
  pfx_sync();
  if (lane == 63) sh[wave] = v;
  pfx_sync();
  float off = -INFINITY;
  for (int k = 0; k < wave; k++) off = fmaxf(off, sh[k]);
  return fmaxf(v, off);
}

// The running sum carried through [lo, hi) in index order by the whole workgroup (NT threads): r_io / carry_io
// are the (workgroup-uniform) running sum and running maximum before element lo on entry and after element hi-1 on
// exit.  With lo == 0 the first PFX_HEAD elements are added one by one.
template <int NT>
__device__ __forceinline__ void pfx_exact_range(const float* __restrict__ w, long long lo, long long hi,
                                             float* __restrict__ runmax, float* __restrict__ prefix_opt,
                                             float& r_io, float& carry_io, int head_len = PFX_HEAD) {
  const long long n = hi;
  __shared__ long long sh_ll[NT / 64];
  __shared__ int sh_i[NT / 64];
  __shared__ float sh_f[NT / 64];
  __shared__ int tie_pos[PFX_TIE_CAP];         // local element index of every tie in order, weight sign in bit 31
  __shared__ long long tie_sval[PFX_TIE_CAP];  // inclusive increment prefix at the tie
  __shared__ signed char tie_corr[PFX_TIE_CAP];
  __shared__ float head[PFX_HEAD];
  __shared__ float last_val[NT], last_max[NT];
  __shared__ int s_first_bad, s_first_cross, s_first_nz;
  __shared__ float s_r, s_carry;
  __shared__ long long s_base;
  const int tid = threadIdx.x;
  pfx_sync();
  if (tid == 0) { s_r = r_io; s_carry = carry_io; s_base = lo; }
  pfx_sync();
  if (lo == 0) {  // head: plain serial additions by one thread out of LDS
    const int hn = (int)min((long long)head_len, (long long)n);
    for (int t = tid; t < hn; t += NT) head[t] = w[t];
    pfx_sync();
    if (tid == 0) {
      float run = 0.f, mx = -INFINITY;
      for (int t = 0; t < hn; t++) {
        run = run + head[t];  // particle_filter.cpp:179
        if (prefix_opt) prefix_opt[t] = run;
        if (run == run) mx = fmaxf(mx, run);
        head[t] = mx;
      }
      s_r = run; s_carry = mx; s_base = hn;
    }
    pfx_sync();
    for (int t = tid; t < hn; t += NT) runmax[t] = head[t];
    pfx_sync();
  }

  while (true) {
    const long long base = s_base;
    if (base >= n) break;
    const float r = s_r;
    const float carry = s_carry;
    const int cnt = (int)min((long long)(NT * PFX_K), (long long)n - base);
    const unsigned rb = __float_as_uint(r);
    const int re = (rb >> 23) & 0xFF;
    float wv[PFX_K];
#pragma unroll
    for (int k = 0; k < PFX_K; k++) {
      const int li = tid * PFX_K + k;
      wv[k] = (li < cnt) ? w[base + li] : 0.f;
    }
    if (tid == 0) { s_first_bad = (NT * PFX_K); s_first_cross = (NT * PFX_K); s_first_nz = (NT * PFX_K); }
    pfx_sync();  // also: everyone has read s_r / s_carry / s_base

    if (r != r) {  // NaN running sum: every later prefix is NaN, the running maximum stays
#pragma unroll
      for (int k = 0; k < PFX_K; k++) {
        const int li = tid * PFX_K + k;
        if (li < cnt) {
          runmax[base + li] = carry;
          if (prefix_opt) prefix_opt[base + li] = r;
        }
      }
      pfx_sync();
      if (tid == 0) s_base = base + cnt;
      pfx_sync();
      continue;
    }
    const bool r_zero = (rb & 0x7FFFFFFFu) == 0;
    const bool r_slow = (rb >> 31) != 0 || re == 0 || re == 255;  // negative, zero / subnormal, inf: plain float steps
    if (r_slow) {
      int stop = 0;  // leading elements that leave r unchanged (r == +-0 only: skip the run of zero weights)
      if (r_zero) {
#pragma unroll
        for (int k = 0; k < PFX_K; k++) {
          const int li = tid * PFX_K + k;
          if (li < cnt && (__float_as_uint(wv[k]) & 0x7FFFFFFFu) != 0) atomicMin(&s_first_nz, li);
        }
        pfx_sync();
        stop = min(s_first_nz, cnt);
      }
      const float m0 = fmaxf(carry, r);
#pragma unroll
      for (int k = 0; k < PFX_K; k++) {
        const int li = tid * PFX_K + k;
        if (li < stop) {
          runmax[base + li] = m0;
          if (prefix_opt) prefix_opt[base + li] = r;
        }
      }
      pfx_sync();
      if (tid == 0) {
        float nr = r, nc = stop > 0 ? m0 : carry;
        long long nb = base + stop;
        if (stop < cnt) {  // one real float addition (particle_filter.cpp:179)
          nr = r + w[base + stop];
          if (nr == nr) nc = fmaxf(nc, nr);
          runmax[base + stop] = nc;
          if (prefix_opt) prefix_opt[base + stop] = nr;
          nb = base + stop + 1;
        }
        s_r = nr; s_carry = nc; s_base = nb;
      }
      pfx_sync();
      continue;
    }

    // ---- r is a positive normal float: r = R * 2^(e-23), R in [2^23, 2^24)
    const int e = re - 127;
    const long long R = (long long)((rb & 0x7FFFFFu) | 0x800000u);
    long long inc[PFX_K];
    bool tie[PFX_K], neg[PFX_K];
    long long tsum = 0;
    int ntie = 0;
#pragma unroll
    for (int k = 0; k < PFX_K; k++) {
      const int li = tid * PFX_K + k;
      const unsigned b = __float_as_uint(wv[k]);
      const int ew = (b >> 23) & 0xFF;
      unsigned mw = b & 0x7FFFFFu;
      const bool zero = (b & 0x7FFFFFFFu) == 0;
      const int E = ew == 0 ? -126 : ew - 127;
      if (ew != 0) mw |= 0x800000u;
      const int sft = e - E;
      const bool in = li < cnt;
      const bool bad = in && ((ew == 255) || (sft < 0));  // NaN / inf / at least as large as the binade: real add
      neg[k] = (b >> 31) != 0 && !zero;
      const int sc = sft < 0 ? 0 : (sft > 26 ? 26 : sft);
      const unsigned f = mw >> sc;
      const unsigned rem = mw & ((1u << sc) - 1u);
      const unsigned half = sc >= 1 ? (1u << (sc - 1)) : 0u;
      const bool up = sc >= 1 && rem > half;
      tie[k] = in && !bad && sc >= 1 && rem == half;
      long long q = (long long)f + (up ? 1 : 0);  // |w|/u rounded to nearest (ties toward zero, patched below)
      if (neg[k]) q = -q;
      if (!in || bad) q = 0;
      if (bad) atomicMin(&s_first_bad, li);
      inc[k] = q;
      tsum += q;
      ntie += tie[k] ? 1 : 0;
    }
    long long tot_ll;
    const long long sincl = pfx_block_scan<NT, long long>(tsum, sh_ll, tot_ll);
    long long S[PFX_K];  // inclusive prefix of the increments at this thread's elements
    {
      long long acc = sincl - tsum;
#pragma unroll
      for (int k = 0; k < PFX_K; k++) { acc += inc[k]; S[k] = acc; }
    }
    // ---- ties, in element order: the even neighbour wins, which depends on everything before the tie
    int tot_tie;
    const int tincl = pfx_block_scan<NT, int>(ntie, sh_i, tot_tie);
    const int tfirst = tincl - ntie;
    {
      int pos = tfirst;
#pragma unroll
      for (int k = 0; k < PFX_K; k++)
        if (tie[k]) {
          if (pos < PFX_TIE_CAP) {
            tie_pos[pos] = (tid * PFX_K + k) | (neg[k] ? (int)0x80000000 : 0);
            tie_sval[pos] = S[k];
          } else if (pos == PFX_TIE_CAP) {
            atomicMin(&s_first_bad, tid * PFX_K + k);  // more ties than slots: cut the tile here
          }
          pos++;
        }
    }
    pfx_sync();
    if (tid == 0 && tot_tie > 0) {
      long long c = 0;
      const int fb = s_first_bad;
      const int nt = min(tot_tie, PFX_TIE_CAP);
      for (int t = 0; t < nt; t++) {
        const int pk = tie_pos[t];
        const int li = pk & 0x7FFFFFFF;
        signed char corr = 0;
        if (li < fb) {
          const long long V = R + tie_sval[t] + c;  // mantissa if the tie is rounded toward zero
          if (V & 1) corr = (pk < 0) ? -1 : 1;      // exact value is V +- 1/2: move to the even neighbour
        }
        tie_corr[t] = corr;
        c += corr;
      }
    }
    pfx_sync();
    int csum = 0;
    int corr[PFX_K];
    {
      int pos = tfirst;
#pragma unroll
      for (int k = 0; k < PFX_K; k++) {
        corr[k] = 0;
        if (tie[k]) { corr[k] = pos < PFX_TIE_CAP ? tie_corr[pos] : 0; pos++; }
        csum += corr[k];
      }
    }
    int tot_c;
    const int cincl = pfx_block_scan<NT, int>(csum, sh_i, tot_c);
    // ---- mantissas, first element that leaves the binade
    long long state[PFX_K];
    {
      int cacc = cincl - csum;
#pragma unroll
      for (int k = 0; k < PFX_K; k++) {
        cacc += corr[k];
        state[k] = R + S[k] + cacc;
        const int li = tid * PFX_K + k;
        if (li < cnt && (state[k] >= (1ll << 24) || state[k] < (1ll << 23))) atomicMin(&s_first_cross, li);
      }
    }
    pfx_sync();
    const int stop = min(min(s_first_bad, s_first_cross), cnt);
    // ---- commit [0, stop): values and running maximum
    float val[PFX_K], lm[PFX_K];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < PFX_K; k++) {
      const int li = tid * PFX_K + k;
      val[k] = __uint_as_float(((unsigned)re << 23) | ((unsigned)state[k] & 0x7FFFFFu));
      if (li < stop) m = fmaxf(m, val[k]);
      lm[k] = m;
    }
    const float mincl = pfx_block_scan_max<NT>(m, sh_f);
    float mexcl = __shfl_up(mincl, 1, 64);
    {  // exclusive maximum over the preceding threads
      pfx_sync();
      last_max[tid] = mincl;
      pfx_sync();
      mexcl = tid > 0 ? last_max[tid - 1] : -INFINITY;
    }
    const float mbase = fmaxf(carry, mexcl);
    float lastv = r, lastm = carry;
#pragma unroll
    for (int k = 0; k < PFX_K; k++) {
      const int li = tid * PFX_K + k;
      if (li < stop) {
        const float rm = fmaxf(mbase, lm[k]);
        runmax[base + li] = rm;
        if (prefix_opt) prefix_opt[base + li] = val[k];
        lastv = val[k];
        lastm = rm;
      }
    }
    pfx_sync();
    last_val[tid] = lastv;   // value / running max at this thread's last committed element (if any)
    last_max[tid] = lastm;
    pfx_sync();
    if (tid == 0) {
      float pv = r, pm = carry;  // value / running max at element stop-1
      if (stop > 0) {
        const int ot = (stop - 1) / PFX_K;
        pv = last_val[ot];
        pm = last_max[ot];
      }
      long long nb = base + stop;
      if (stop < cnt) {  // one real float addition, then a new binade
        const float nr = pv + w[base + stop];
        if (nr == nr) pm = fmaxf(pm, nr);
        runmax[base + stop] = pm;
        if (prefix_opt) prefix_opt[base + stop] = nr;
        pv = nr;
        nb = base + stop + 1;
      }
      s_r = pv; s_carry = pm; s_base = nb;
    }
    pfx_sync();
  }
  r_io = s_r;
  carry_io = s_carry;
  pfx_sync();
}

__global__ __launch_bounds__(PFX_THREADS) void prefix_exact_kernel(const float* __restrict__ w, int64_t n,
                                                                   float* __restrict__ runmax,
                                                                   float* __restrict__ prefix_opt) {
  float r = 0.f, carry = -INFINITY;
  pfx_exact_range<PFX_THREADS>(w, 0, n, runmax, prefix_opt, r, carry);
}

// ---- exact parallel prefix over many workgroups ---------------------------------------------------------------------
// Inside one binade (ulp u) the chain r <- fl(r + w) with w >= 0 is R <- R + a(R & 1): the increment of one element is
// an integer that depends on the running mantissa only through its PARITY (round-half-even ties).  A run of elements is
// therefore summarised by two integers (D0, D1) — its total increment entered with an even / odd mantissa — and
// summaries compose associatively: (A then B)_p = A_p + B_{(p + A_p) & 1}.  That turns the chain into a scan:
//   1. pfx_chunk_sum_kernel      — per chunk of PFXM_CHUNK weights: the sum in double
//   2. pfx_chunk_summary_kernel  — per chunk: binade predicted from the double sum of everything before it; (D0, D1) in
//                                  that binade, or "irregular" (NaN / inf / negative / weight above the binade)
//   3. pfx_walk_kernel           — ONE workgroup walks the chunks in order with the exact running sum: a chunk whose
//                                  prediction holds (same binade, R + D_p stays below 2^24) is a single integer add;
//                                  any other chunk (the ~log2(n) binade crossings, irregular weights, the head) is
//                                  processed on the spot by pfx_exact_range
//   4. pfx_chunk_fill_kernel     — per accepted chunk: the same scan again, now with the exact starting mantissa,
//                                  writes every element's running sum / running maximum
// Every value written is the serial float32 chain's, whatever the prediction was: a wrong prediction only sends the
// chunk down the slower path.
#define PFXM_THREADS 256
#define PFXM_K 16
#define PFXM_CHUNK (PFXM_THREADS * PFXM_K)
#define PFXM_SAT (1u << 30)
struct PfxChunk {     // 32 bytes per chunk in the caller's workspace
  double sum;         // 1: double sum of the chunk
  int re;             // 2: predicted biased exponent of the running sum, or -1 = irregular
  unsigned d0, d1;    // 2: total increment entered with an even / odd mantissa
  float r0, carry0;   // 3: running sum / running maximum before the chunk's first element (accepted chunks)
  int accepted;       // 3: 1 = pfx_chunk_fill_kernel writes this chunk's outputs
};
static_assert(sizeof(PfxChunk) == 32, "PfxChunk");

struct PfxPair { unsigned a0, a1; };
__device__ __forceinline__ unsigned pfx_sat(unsigned x) { return x > PFXM_SAT ? PFXM_SAT : x; }
// first A, then B
__device__ __forceinline__ PfxPair pfx_compose(PfxPair A, PfxPair B) {
  PfxPair c;
  c.a0 = pfx_sat(A.a0 + ((A.a0 & 1u) ? B.a1 : B.a0));
  c.a1 = pfx_sat(A.a1 + ((A.a1 & 1u) ? B.a0 : B.a1));   // entered odd: the parity after A is (1 + A.a1) & 1
  return c;
}
// One weight against the binade of a running sum with biased exponent `re` (PFXM_RE_MIN <= re <= PFXM_RE_MAX), ulp
// u = 2^(re-150): f = round-half-down(w/u), tie = the remainder is exactly one half, bad = cannot be an integer increment
// (NaN, inf, negative, or at least twice the binade's lower end).  w * 2^(150-re) is exact (a power-of-two scaling
// inside the normal range; a product that underflows is far below one half anyway), and so are floor and the
// remainder, so this is the integer classification of pfx_exact_range in a handful of float operations.
#define PFXM_RE_MIN 24
#define PFXM_RE_MAX 254
__device__ __forceinline__ void pfx_classify(float wv, unsigned re, unsigned& f, bool& tie, bool& bad) {
  const float scale = __uint_as_float((277u - re) << 23);   // 2^(150 - re) = 1/u
  const float t = wv * scale;
  bad = !(wv >= 0.f) || !(t < 16777216.f);
  const float fl = floorf(t);
  const float fr = t - fl;
  tie = fr == 0.5f;
  f = (unsigned)fl + (fr > 0.5f ? 1u : 0u);
  if (bad) { f = 0; tie = false; }
}
__device__ __forceinline__ PfxPair pfx_element_pair(unsigned f, bool tie) {
  PfxPair p;
  p.a0 = f + (tie ? (f & 1u) : 0u);         // even mantissa + f + 1/2 -> the even neighbour
  p.a1 = f + (tie ? ((f & 1u) ^ 1u) : 0u);
  return p;
}
// inclusive scan of per-thread pairs over a workgroup of NT threads; returns the EXCLUSIVE pair of this thread and the
// workgroup total.  `sh` holds one slot per wave; safe to call repeatedly (leading barrier).
// Inclusive wave scan of pairs with data-parallel-primitive moves instead of LDS permutes (each __shfl_up is a
// ds_bpermute: ~100 clocks of latency, twelve of them in a row per scan).  Lanes without a source lane receive the
// identity pair {0, 0} (compose({0,0}, v) == v), so no lane predicate is needed.  Steps: shift right by 1, 2, 4, 8
// within rows of 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into rows 2 and 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ PfxPair pfx_pair_dpp(PfxPair v) {
  PfxPair t;
  t.a0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.a0, CTRL, ROW_MASK, 0xF, false);
  t.a1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.a1, CTRL, ROW_MASK, 0xF, false);
  return t;
}
__device__ __forceinline__ PfxPair pfx_pair_wave_scan(PfxPair v) {
  v = pfx_compose(pfx_pair_dpp<0x111, 0xF>(v), v);   // row_shr:1
  v = pfx_compose(pfx_pair_dpp<0x112, 0xF>(v), v);   // row_shr:2
  v = pfx_compose(pfx_pair_dpp<0x114, 0xF>(v), v);   // row_shr:4
  v = pfx_compose(pfx_pair_dpp<0x118, 0xF>(v), v);   // row_shr:8
  v = pfx_compose(pfx_pair_dpp<0x142, 0xA>(v), v);   // row_bcast:15 into rows 1 and 3
  v = pfx_compose(pfx_pair_dpp<0x143, 0xC>(v), v);   // row_bcast:31 into rows 2 and 3
  return v;
}
template <int NT>
__device__ __forceinline__ PfxPair pfx_pair_scan(PfxPair v, PfxPair* sh, PfxPair& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = pfx_pair_wave_scan(v);
  pfx_sync();
  if (lane == 63) sh[wave] = v;
  pfx_sync();
  PfxPair pre = {0u, 0u}, tot = {0u, 0u};
#pragma unroll
  for (int k = 0; k < NT / 64; k++) {
    const PfxPair x = sh[k];
    if (k < wave) pre = pfx_compose(pre, x);
    tot = pfx_compose(tot, x);
  }
  total = tot;
  // exclusive: the inclusive value of the lane before (wave_shr:1; lane 0 gets the identity)
  PfxPair ex = pfx_pair_dpp<0x138, 0xF>(v);
  return pfx_compose(pre, ex);
}
template <int K>
__device__ __forceinline__ void pfx_load_chunk(const float* __restrict__ w, long long lo, int cnt, float (&wv)[K]) {
  const int t0 = threadIdx.x * K;
  if (t0 + K <= cnt && ((lo & 3) == 0)) {
    const float4* p = reinterpret_cast<const float4*>(w + lo + t0);
#pragma unroll
    for (int k = 0; k < K / 4; k++) {
      const float4 v = p[k];
      wv[4 * k] = v.x; wv[4 * k + 1] = v.y; wv[4 * k + 2] = v.z; wv[4 * k + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; k++) wv[k] = (t0 + k < cnt) ? w[lo + t0 + k] : 0.f;
  }
}

__global__ __launch_bounds__(PFXM_THREADS) void pfx_chunk_sum_kernel(const float* __restrict__ w, int64_t n,
                                                                     PfxChunk* __restrict__ ch) {
  __shared__ double shd[PFXM_THREADS / 64];
  const long long lo = (long long)blockIdx.x * PFXM_CHUNK;
  const int cnt = (int)min((long long)PFXM_CHUNK, (long long)n - lo);
  float wv[PFXM_K];
  pfx_load_chunk(w, lo, cnt, wv);
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < PFXM_K; k++) acc += (double)wv[k];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = acc;
  pfx_sync();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < PFXM_THREADS / 64; k++) t += shd[k];
    ch[blockIdx.x].sum = t;
  }
}

__global__ __launch_bounds__(PFXM_THREADS) void pfx_chunk_summary_kernel(const float* __restrict__ w, int64_t n,
                                                                         PfxChunk* __restrict__ ch, int c_first) {
  __shared__ double shd[PFXM_THREADS / 64];
  __shared__ PfxPair shp[PFXM_THREADS / 64];
  __shared__ int s_bad;
  const int c = blockIdx.x + c_first;   // (the chunks before c_first are not walked, see prefix_multi)
  const long long lo = (long long)c * PFXM_CHUNK;
  const int cnt = (int)min((long long)PFXM_CHUNK, (long long)n - lo);
  // predicted running sum before the chunk: the double sums of the chunks before it, in chunk order per thread
  double acc = 0.0;
  for (int j = threadIdx.x; j < c; j += PFXM_THREADS) acc += ch[j].sum;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = acc;
  if (threadIdx.x == 0) s_bad = 0;
  pfx_sync();
  double before = 0.0;
  for (int k = 0; k < PFXM_THREADS / 64; k++) before += shd[k];
  const float r_pred = (float)before, r_end = (float)(before + ch[c].sum);
  const unsigned pb = __float_as_uint(r_pred), eb = __float_as_uint(r_end);
  const int re = (pb >> 23) & 0xFF;
  // a chunk that is predicted to start and end in one binade of a positive normal sum; everything else is irregular
  const bool plausible = (pb >> 31) == 0 && re >= PFXM_RE_MIN && re <= PFXM_RE_MAX && (int)((eb >> 23) & 0xFF) == re && (eb >> 31) == 0;
  if (!plausible) {   // uniform across the workgroup
    if (threadIdx.x == 0) { ch[c].re = -1; ch[c].d0 = 0u; ch[c].d1 = 0u; }
    return;
  }
  float wv[PFXM_K];
  pfx_load_chunk(w, lo, cnt, wv);
  PfxPair mine = {0u, 0u};
  bool anybad = false;
#pragma unroll
  for (int k = 0; k < PFXM_K; k++) {
    unsigned f; bool tie, bad;
    pfx_classify(wv[k], (unsigned)re, f, tie, bad);
    anybad |= bad;
    mine = pfx_compose(mine, pfx_element_pair(f, tie));
  }
  if (anybad) s_bad = 1;   // benign race: every writer stores 1; ordered by the barrier inside the scan
  PfxPair total;
  (void)pfx_pair_scan<PFXM_THREADS>(mine, shp, total);
  if (threadIdx.x == 0) {
    const bool ok = s_bad == 0 && total.a0 < (1u << 24) && total.a1 < (1u << 24);
    ch[c].re = ok ? re : -1;
    ch[c].d0 = total.a0;
    ch[c].d1 = total.a1;
  }
}

// A chunk the walk cannot take as one integer add (it holds a binade crossing or an irregular weight, or was
// mispredicted), carried through in order by the walking workgroup: one scan per stretch between two real float
// additions.  The walking workgroup has PFXW_THREADS threads holding PFXW_K weights each (two waves per SIMD: the
// dependent instruction chains of one wave hide behind the other's).  Anything that is not a positive normal running sum goes to
// pfx_exact_range.
#ifndef PFXW_THREADS
#define PFXW_THREADS 512   // the walking workgroup: PFXW_THREADS x PFXW_K = one chunk
#endif
#define PFXW_K (PFXM_CHUNK / PFXW_THREADS)
#define PFXW_HEAD 64   // leading elements the walk adds one by one (tunable: tdr_config_tuning("prefix_head", n))
static int g_pfx_head = PFXW_HEAD;
extern "C" int tdr_config_prefix_head(int n) {   // < 0: query only
  if (n >= 0) g_pfx_head = n < 1 ? 1 : (n > PFX_HEAD ? PFX_HEAD : n);
  return g_pfx_head;
}
__device__ __forceinline__ void pfx_walk_chunk(const float* __restrict__ w, long long lo, int cnt,
                                               float* __restrict__ runmax, float* __restrict__ prefix_opt,
                                               float& r, float& carry, int head_len) {
  __shared__ PfxPair shp[PFXW_THREADS / 64];
  __shared__ int s_bad, s_cross;
  __shared__ float s_last, s_wstop;
  const int tid = threadIdx.x, t0 = tid * PFXW_K;
  float wv[PFXW_K];
  pfx_load_chunk(w, lo, cnt, wv);
  int pos = 0;   // workgroup-uniform: elements before pos are done
  while (pos < cnt) {
    const unsigned rb = __float_as_uint(r);
    const unsigned re = rb >> 23;   // sign included
    if (!(re >= PFXM_RE_MIN && re <= PFXM_RE_MAX)) {
      // zero / tiny / huge / negative / inf / NaN running sum: the general path (with its serial head at the very start)
      const long long a = lo + pos;
      const long long b = a == 0 ? min((long long)head_len, (long long)cnt) : lo + cnt;
      pfx_exact_range<PFXW_THREADS>(w, a, b, runmax, prefix_opt, r, carry, head_len);
      pos = (int)(b - lo);
      continue;
    }
    const unsigned R = (rb & 0x7FFFFFu) | 0x800000u;
    pfx_sync();
    if (tid == 0) { s_bad = cnt; s_cross = cnt; s_last = r; s_wstop = 0.f; }
    pfx_sync();
    unsigned f[PFXW_K];
    unsigned tiebits = 0u;
    PfxPair mine = {0u, 0u};
    {
#pragma unroll
      for (int k = 0; k < PFXW_K; k++) {
        const int li = t0 + k;
        bool bad, tie;
        pfx_classify(wv[k], re, f[k], tie, bad);
        if (li < pos || li >= cnt) { f[k] = 0u; tie = false; bad = false; }
        if (bad) atomicMin(&s_bad, li);
        tiebits |= tie ? (1u << k) : 0u;
        mine = pfx_compose(mine, pfx_element_pair(f[k], tie));
      }
    }
    PfxPair total;
    const PfxPair ex = pfx_pair_scan<PFXW_THREADS>(mine, shp, total);
    unsigned st[PFXW_K];
    {
      unsigned state = R + ((R & 1u) ? ex.a1 : ex.a0);
      int first = cnt;
#pragma unroll
      for (int k = 0; k < PFXW_K; k++) {
        state += f[k] + (((tiebits >> k) & 1u) ? ((state + f[k]) & 1u) : 0u);
        st[k] = state;
        const int li = t0 + k;
        if (li >= pos && li < cnt && state >= (1u << 24)) first = min(first, li);
      }
      if (first < cnt) atomicMin(&s_cross, first);
    }
    pfx_sync();
    const int stop = min(s_bad, s_cross);   // first element that needs a real float addition (or cnt)
    {
#pragma unroll
      for (int k = 0; k < PFXW_K; k++) {
        const int li = t0 + k;
        if (li >= pos && li < stop) {
          const float val = __uint_as_float((re << 23) | (st[k] & 0x7FFFFFu));
          runmax[lo + li] = fmaxf(carry, val);
          if (prefix_opt) prefix_opt[lo + li] = val;
          if (li == stop - 1) s_last = val;
        }
        if (li == stop) s_wstop = wv[k];
      }
    }
    pfx_sync();
    r = s_last;                  // the sum after element stop-1 (unchanged when nothing was committed)
    carry = fmaxf(carry, r);     // increments are non-negative: the last committed value is the largest
    if (stop < cnt) {
      const float nr = r + s_wstop;   // particle_filter.cpp:179, one real addition
      if (nr == nr) carry = fmaxf(carry, nr);
      if (tid == 0) {
        runmax[lo + stop] = carry;
        if (prefix_opt) prefix_opt[lo + stop] = nr;
      }
      r = nr;
      pos = stop + 1;
    } else {
      pos = cnt;
    }
  }
}

#define PFXW_BLOCK 512   // chunk summaries / headers staged in LDS at a time
__global__ __launch_bounds__(PFXW_THREADS) void pfx_walk_kernel(const float* __restrict__ w, int64_t n,
                                                               PfxChunk* __restrict__ ch, int nch,
                                                               float* __restrict__ runmax,
                                                               float* __restrict__ prefix_opt, int head_len,
                                                               int c_first, const float* __restrict__ tail) {
  __shared__ int sm_re[PFXW_BLOCK], sm_acc[PFXW_BLOCK];
  __shared__ unsigned sm_d0[PFXW_BLOCK], sm_d1[PFXW_BLOCK];
  __shared__ float sm_r0[PFXW_BLOCK], sm_c0[PFXW_BLOCK];
  // workgroup-uniform; chunks [0, c_first) were done by pfx_small_kernel, which left the sum and the maximum behind them
  float r = c_first > 0 ? tail[0] : 0.f, carry = c_first > 0 ? tail[1] : -INFINITY;
  for (int cb = 0; cb < nch; cb += PFXW_BLOCK) {
    pfx_sync();
    for (int t = threadIdx.x; t < PFXW_BLOCK && cb + t < nch; t += PFXW_THREADS) {
      const PfxChunk x = ch[cb + t];
      sm_re[t] = x.re; sm_d0[t] = x.d0; sm_d1[t] = x.d1;
    }
    pfx_sync();
    const int ce = min(nch, cb + PFXW_BLOCK);
    for (int c = cb; c < ce; c++) {
      if (c < c_first) { sm_acc[c - cb] = 0; continue; }
      const unsigned rb = __float_as_uint(r);
      const int re = (int)(rb >> 23);                   // sign bit included: a negative sum never matches
      const unsigned R = (rb & 0x7FFFFFu) | 0x800000u;
      const unsigned D = (R & 1u) ? sm_d1[c - cb] : sm_d0[c - cb];
      const bool fast = sm_re[c - cb] == re && R + D < (1u << 24);   // sm_re is in [PFXM_RE_MIN, PFXM_RE_MAX] or -1
      // the waves run through this loop unsynchronised and all store the same words
      if (fast) {
        sm_r0[c - cb] = r; sm_c0[c - cb] = carry; sm_acc[c - cb] = 1;
        r = __uint_as_float(((unsigned)re << 23) | ((R + D) & 0x7FFFFFu));
        carry = fmaxf(carry, r);
      } else {
        sm_acc[c - cb] = 0;
        const long long lo = (long long)c * PFXM_CHUNK;
        const int cnt = (int)min((long long)PFXM_CHUNK, (long long)n - lo);
        pfx_walk_chunk(w, lo, cnt, runmax, prefix_opt, r, carry, head_len);
      }
#ifdef TDR_PFX_TIMING   // diagnostic build: time stamp (100 MHz) after every chunk in the header's dead `sum` slot
      if (threadIdx.x == 0) *reinterpret_cast<long long*>(&ch[c].sum) = (long long)wall_clock64();
#endif
    }
    pfx_sync();
    for (int t = threadIdx.x; t < PFXW_BLOCK && cb + t < nch; t += PFXW_THREADS) {   // for pfx_chunk_fill_kernel
      PfxChunk* o = ch + cb + t;
      o->r0 = sm_r0[t];
      o->carry0 = sm_c0[t];
      o->accepted = sm_acc[t];
    }
  }
}

__global__ __launch_bounds__(PFXM_THREADS) void pfx_chunk_fill_kernel(const float* __restrict__ w, int64_t n,
                                                                      const PfxChunk* __restrict__ ch,
                                                                      float* __restrict__ runmax,
                                                                      float* __restrict__ prefix_opt) {
  __shared__ PfxPair shp[PFXM_THREADS / 64];
  const int c = blockIdx.x;
  const PfxChunk hdr = ch[c];
  if (!hdr.accepted) return;   // written by the walk
  const long long lo = (long long)c * PFXM_CHUNK;
  const int cnt = (int)min((long long)PFXM_CHUNK, (long long)n - lo);
  const unsigned rb = __float_as_uint(hdr.r0);
  const unsigned re = rb >> 23;
  const unsigned R = (rb & 0x7FFFFFu) | 0x800000u;
  float wv[PFXM_K];
  pfx_load_chunk(w, lo, cnt, wv);
  unsigned f[PFXM_K];
  bool tie[PFXM_K];
  PfxPair mine = {0u, 0u};
#pragma unroll
  for (int k = 0; k < PFXM_K; k++) {
    bool bad;
    pfx_classify(wv[k], re, f[k], tie[k], bad);
    mine = pfx_compose(mine, pfx_element_pair(f[k], tie[k]));
  }
  PfxPair total;
  const PfxPair ex = pfx_pair_scan<PFXM_THREADS>(mine, shp, total);
  unsigned state = R + ((R & 1u) ? ex.a1 : ex.a0);   // the exact mantissa before this thread's first element
  const int t0 = threadIdx.x * PFXM_K;
  float val[PFXM_K];
#pragma unroll
  for (int k = 0; k < PFXM_K; k++) {
    state += f[k] + (tie[k] ? ((state + f[k]) & 1u) : 0u);
    val[k] = __uint_as_float((re << 23) | (state & 0x7FFFFFu));
  }
  const float carry = hdr.carry0;
  if (t0 + PFXM_K <= cnt && ((lo & 3) == 0)) {
    float4* o = reinterpret_cast<float4*>(runmax + lo + t0);
#pragma unroll
    for (int k = 0; k < PFXM_K / 4; k++)
      o[k] = make_float4(fmaxf(carry, val[4 * k]), fmaxf(carry, val[4 * k + 1]), fmaxf(carry, val[4 * k + 2]),
                         fmaxf(carry, val[4 * k + 3]));
    if (prefix_opt) {
      float4* q = reinterpret_cast<float4*>(prefix_opt + lo + t0);
#pragma unroll
      for (int k = 0; k < PFXM_K / 4; k++) q[k] = make_float4(val[4 * k], val[4 * k + 1], val[4 * k + 2], val[4 * k + 3]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < PFXM_K; k++)
      if (t0 + k < cnt) {
        runmax[lo + t0 + k] = fmaxf(carry, val[k]);
        if (prefix_opt) prefix_opt[lo + t0 + k] = val[k];
      }
  }
}

// ---- totals of serial float32 chains with double addends ---------------------------------------------------------------
// ParticleFilter::update's statistics are serial chains too (src/particle_filter.cpp:108-126):
//     sum           : float += float            over the valid (non-NaN) raw weights
//     bottom_stddev : float += pow(w - mean, 2)  = (float)((double)acc + x), x an exact double, over the weights below the mean
// Only their FINAL values are used.  Both are evaluated by the scan of this file with the addend as a double made on the
// fly from (raw, mean): for a float chain r <- (float)((double)r + x) the step is x first rounded to the double grid of
// the sum's binade — 2^-29 ulp(r), and because R is an integer that rounding does not depend on R — and then to the
// float grid with the parity rule of pfx_classify.  (A float addend lies on the double grid or is far below half an
// ulp, so the `sum` chain is the plain float chain.)  Chunks as above: double sum -> predicted binade + parity summary
// -> one walking workgroup; no fill pass.
struct ChainSrc {
  const float* raw;    // raw weights
  const float* mean;   // device scalar (kind 1)
  int kind;            // 0: valid raw weights; 1: squared deviations of the weights below the mean
};
__device__ __forceinline__ double chain_addend(const ChainSrc& s, long long i, float mean) {
  const float v = s.raw[i];
  if (s.kind == 0) return (v != v) ? 0.0 : (double)v;                       // :111-115
  if (v != v || !(v < mean)) return 0.0;                                    // :120
  const double d = (double)(v - mean);                                      // float subtraction, then pow(double, 2)
  return d * d;
}
__device__ __forceinline__ void chain_classify(double x, unsigned re, unsigned& f, bool& tie, bool& bad) {
  const double scale = __longlong_as_double((long long)(1023 + 150 - (int)re) << 52);   // 2^(150 - re) = 1/u
  const double t = x * scale;
  bad = !(x >= 0.0) || !(t < 4194304.0);                 // negative / NaN / inf, or not small against the sum: real add
  const double tg = (t + 8388608.0) - 8388608.0;         // t on the double grid of [2^23, 2^24): multiples of 2^-29
  const double fl = floor(tg);
  const double fr = tg - fl;
  tie = fr == 0.5;
  f = (unsigned)fl + (fr > 0.5 ? 1u : 0u);
  if (bad) { f = 0; tie = false; }
}
#define CHAIN_K (PFXM_CHUNK / PFXW_THREADS)
#ifndef CHAIN_HEAD
#define CHAIN_HEAD 1024   // leading addends added one by one (<= PFXM_CHUNK)
#endif
__device__ __forceinline__ void chain_load(const ChainSrc& s, long long lo, int cnt, float mean, double (&xv)[CHAIN_K]) {
  const int t0 = threadIdx.x * CHAIN_K;
#pragma unroll
  for (int k = 0; k < CHAIN_K; k++) xv[k] = (t0 + k < cnt) ? chain_addend(s, lo + t0 + k, mean) : 0.0;
}
__global__ __launch_bounds__(PFXW_THREADS) void chain_sum_kernel(ChainSrc s, int64_t n, PfxChunk* __restrict__ ch) {
  __shared__ double shd[PFXW_THREADS / 64];
  const long long lo = (long long)blockIdx.x * PFXM_CHUNK;
  const int cnt = (int)min((long long)PFXM_CHUNK, (long long)n - lo);
  const float mean = s.kind ? *s.mean : 0.f;
  double xv[CHAIN_K];
  chain_load(s, lo, cnt, mean, xv);
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < CHAIN_K; k++) acc += xv[k];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < PFXW_THREADS / 64; k++) t += shd[k];
    ch[blockIdx.x].sum = t;
  }
}
__global__ __launch_bounds__(PFXW_THREADS) void chain_summary_kernel(ChainSrc s, int64_t n, PfxChunk* __restrict__ ch,
                                                                    int c_first) {
  __shared__ double shd[PFXW_THREADS / 64];
  __shared__ PfxPair shp[PFXW_THREADS / 64];
  __shared__ int s_bad;
  const int c = blockIdx.x + c_first;   // (the chunks before c_first are not walked, see tdr_chain_total)
  const long long lo = (long long)c * PFXM_CHUNK;
  const int cnt = (int)min((long long)PFXM_CHUNK, (long long)n - lo);
  double acc = 0.0;
  for (int j = threadIdx.x; j < c; j += PFXW_THREADS) acc += ch[j].sum;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = acc;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  double before = 0.0;
  for (int k = 0; k < PFXW_THREADS / 64; k++) before += shd[k];
  const float r_pred = (float)before, r_end = (float)(before + ch[c].sum);
  const unsigned pb = __float_as_uint(r_pred), eb = __float_as_uint(r_end);
  const int re = (pb >> 23) & 0xFF;
  const bool plausible = (pb >> 31) == 0 && re >= PFXM_RE_MIN && re <= PFXM_RE_MAX && (int)((eb >> 23) & 0xFF) == re && (eb >> 31) == 0;
  if (!plausible) {
    if (threadIdx.x == 0) { ch[c].re = -1; ch[c].d0 = 0u; ch[c].d1 = 0u; }
    return;
  }
  const float mean = s.kind ? *s.mean : 0.f;
  double xv[CHAIN_K];
  chain_load(s, lo, cnt, mean, xv);
  PfxPair mine = {0u, 0u};
  bool anybad = false;
#pragma unroll
  for (int k = 0; k < CHAIN_K; k++) {
    unsigned f; bool tie, bad;
    chain_classify(xv[k], (unsigned)re, f, tie, bad);
    anybad |= bad;
    mine = pfx_compose(mine, pfx_element_pair(f, tie));
  }
  if (anybad) s_bad = 1;
  PfxPair total;
  (void)pfx_pair_scan<PFXW_THREADS>(mine, shp, total);
  if (threadIdx.x == 0) {
    const bool ok = s_bad == 0 && total.a0 < (1u << 24) && total.a1 < (1u << 24);
    ch[c].re = ok ? re : -1;
    ch[c].d0 = total.a0;
    ch[c].d1 = total.a1;
  }
}
// one chunk on the spot: like pfx_walk_chunk, without outputs and with the real additions done in double
#ifdef TDR_UW_TIMELINE   // diagnostic build: 100 MHz time stamps at the phase boundaries of uw_small_kernel
__device__ unsigned long long g_uw_tl[16];
extern "C" int tdr_debug_read_uw_timeline(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_uw_tl), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#define UW_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0) g_uw_tl[k] = wall_clock64(); } while (0)
#define UW_COUNT(k) do { if (threadIdx.x == 0) g_uw_tl[k]++; } while (0)
#else
#define UW_STAMP(k) do { } while (0)
#define UW_COUNT(k) do { } while (0)
#endif
// FROM_LDS: the raw weights sit in the kernel's dynamic LDS array (uw_small_kernel) instead of global memory
__device__ __forceinline__ int uws_idx(int e) { return e + (e >> 5); }   // one pad word per 32: bank spread for strided runs
__device__ __forceinline__ double uws_addend(int kind, int e, float mean) {
  extern __shared__ float uws_lraw[];
  const float v = uws_lraw[uws_idx(e)];
  if (kind == 2) return (double)v;                                          // :179 (the running sum of the resample)
  if (kind == 0) return (v != v) ? 0.0 : (double)v;                         // :111-115
  if (v != v || !(v < mean)) return 0.0;                                    // :120
  const double d = (double)(v - mean);                                      // float subtraction, then pow(double, 2)
  return d * d;
}
template <bool FROM_LDS = false>
__device__ __forceinline__ void chain_walk_chunk(const ChainSrc& s, long long lo, int cnt, float mean, float& r,
                                                 int pos_start) {
  __shared__ PfxPair shp[PFXW_THREADS / 64];
  __shared__ int s_bad, s_cross, s_nz;
  __shared__ unsigned s_state;
  __shared__ double s_xstop;
  const int tid = threadIdx.x, t0 = tid * CHAIN_K;
  double xv[CHAIN_K];
  if constexpr (FROM_LDS) {
#pragma unroll
    for (int k = 0; k < CHAIN_K; k++) xv[k] = (t0 + k < cnt) ? uws_addend(s.kind, (int)lo + t0 + k, mean) : 0.0;
  } else {
    chain_load(s, lo, cnt, mean, xv);
  }
  int pos = pos_start;
  while (pos < cnt) {
#ifdef TDR_UW_TIMELINE
    if (FROM_LDS && tid == 0) g_uw_tl[10 + s.kind]++;
#endif
    const unsigned rb = __float_as_uint(r);
    const unsigned re = rb >> 23;   // sign included
    pfx_sync();
    if (tid == 0) { s_bad = cnt; s_cross = cnt; s_nz = cnt; s_xstop = 0.0; s_state = 0u; }
    pfx_sync();

    if (!(re >= PFXM_RE_MIN && re <= PFXM_RE_MAX)) {
      // zero / tiny / huge / inf / NaN running sum: zero addends change nothing, the next other one is really added
      int first = cnt;
#pragma unroll
      for (int k = 0; k < CHAIN_K; k++) {
        const int li = t0 + k;
        if (li >= pos && li < cnt && xv[k] != 0.0) first = min(first, li);
      }
      if (first < cnt) atomicMin(&s_nz, first);
      pfx_sync();
      const int stop = s_nz;
#pragma unroll
      for (int k = 0; k < CHAIN_K; k++)
        if (t0 + k == stop) s_xstop = xv[k];
      pfx_sync();
      if (stop < cnt) r = (float)((double)r + s_xstop);
      pos = stop < cnt ? stop + 1 : cnt;
      continue;
    }
    const unsigned R = (rb & 0x7FFFFFu) | 0x800000u;
    unsigned f[CHAIN_K];
    unsigned tiebits = 0u;
    PfxPair mine = {0u, 0u};
#pragma unroll
    for (int k = 0; k < CHAIN_K; k++) {
      const int li = t0 + k;
      bool bad, tie;
      chain_classify(xv[k], re, f[k], tie, bad);
      if (li < pos || li >= cnt) { f[k] = 0u; tie = false; bad = false; }
      if (bad) atomicMin(&s_bad, li);
      tiebits |= tie ? (1u << k) : 0u;
      mine = pfx_compose(mine, pfx_element_pair(f[k], tie));
    }

    PfxPair total;
    const PfxPair ex = pfx_pair_scan<PFXW_THREADS>(mine, shp, total);
    unsigned st[CHAIN_K];
    {
      unsigned state = R + ((R & 1u) ? ex.a1 : ex.a0);
      int first = cnt;
#pragma unroll
      for (int k = 0; k < CHAIN_K; k++) {
        state += f[k] + (((tiebits >> k) & 1u) ? ((state + f[k]) & 1u) : 0u);
        st[k] = state;
        const int li = t0 + k;
        if (li >= pos && li < cnt && state >= (1u << 24)) first = min(first, li);
      }
      if (first < cnt) atomicMin(&s_cross, first);
    }
    pfx_sync();
    const int stop = min(s_bad, s_cross);
#pragma unroll
    for (int k = 0; k < CHAIN_K; k++) {
      const int li = t0 + k;
      if (li >= pos && li == stop - 1) s_state = st[k];
      if (li == stop) s_xstop = xv[k];
    }
    pfx_sync();

    if (stop > pos) r = __uint_as_float((re << 23) | (s_state & 0x7FFFFFu));
    if (stop < cnt) {
      r = (float)((double)r + s_xstop);   // one real addition, in the reference's types
      pos = stop + 1;
    } else {
      pos = cnt;
    }
  }
}
__global__ __launch_bounds__(PFXW_THREADS) void chain_walk_kernel(ChainSrc s, int64_t n, const PfxChunk* __restrict__ ch,
                                                                 int nch, float* __restrict__ total_out, int c_first,
                                                                 const float* __restrict__ r_first) {
  __shared__ int sm_re[PFXW_BLOCK];
  __shared__ unsigned sm_d0[PFXW_BLOCK], sm_d1[PFXW_BLOCK];
  const float mean = s.kind ? *s.mean : 0.f;
  // head: the sum of unnormalised weights crosses a binade every time it doubles — about ten times within the first
  // thousand addends — so those are added one by one by a single thread out of LDS
  __shared__ double head[CHAIN_HEAD];
  __shared__ float s_head_r;
  // (c_first > 0: chunks [0, c_first) were summed by chain_head_kernel, which left the sum behind them in *r_first)
  const int hn = c_first > 0 ? 0 : (int)min((long long)CHAIN_HEAD, (long long)n);
  for (int t = threadIdx.x; t < hn; t += PFXW_THREADS) head[t] = chain_addend(s, t, mean);
  pfx_sync();
  if (threadIdx.x == 0) {
    float run = c_first > 0 ? *r_first : 0.f;
    for (int t = 0; t < hn; t++) run = (float)((double)run + head[t]);
    s_head_r = run;
  }
  pfx_sync();
  float r = s_head_r;   // workgroup-uniform
  for (int cb = (c_first / PFXW_BLOCK) * PFXW_BLOCK; cb < nch; cb += PFXW_BLOCK) {
    pfx_sync();
    for (int t = threadIdx.x; t < PFXW_BLOCK && cb + t < nch; t += PFXW_THREADS) {
      const PfxChunk x = ch[cb + t];
      sm_re[t] = x.re; sm_d0[t] = x.d0; sm_d1[t] = x.d1;
    }
    pfx_sync();
    const int ce = min(nch, cb + PFXW_BLOCK);
    for (int c = max(cb, c_first); c < ce; c++) {
      const unsigned rb = __float_as_uint(r);
      const int re = (int)(rb >> 23);
      const unsigned R = (rb & 0x7FFFFFu) | 0x800000u;
      const unsigned D = (R & 1u) ? sm_d1[c - cb] : sm_d0[c - cb];
      if (c > 0 && sm_re[c - cb] == re && R + D < (1u << 24)) {
        r = __uint_as_float(((unsigned)re << 23) | ((R + D) & 0x7FFFFFu));
      } else {
        const long long lo = (long long)c * PFXM_CHUNK;
        chain_walk_chunk(s, lo, (int)min((long long)PFXM_CHUNK, (long long)n - lo), mean, r, c == 0 ? hn : 0);
      }
    }
  }
  if (threadIdx.x == 0) *total_out = r;
}
static int g_pfx_small = 1;   // 0 = without the one-launch kernel (A/B and debugging)
extern "C" int tdr_config_prefix_small(int on) {   // < 0: query only
  if (on >= 0) g_pfx_small = on ? 1 : 0;
  return g_pfx_small;
}
#define CHAIN_HEAD_N 32768
// whole chunks at the start that the one-workgroup machinery takes (0: none; tdr_config_prefix_small(0) switches it off
// for the running sum AND the statistics chains: the chunk walk from the first addend on, for A/B and debugging)
static int chain_head_chunks(int64_t n) { return (g_pfx_small && n >= CHAIN_HEAD_N) ? CHAIN_HEAD_N / PFXM_CHUNK : 0; }
static int chain_head_launch(const float* raw, const float* mean_dev, int kind, int n_head, float* r_out, hipStream_t st);
// raw: [n] raw weights; kind 0: total = serial float sum of the non-NaN weights; kind 1: total = serial
// float-accumulated sum of pow(w - *mean_dev, 2) over the non-NaN weights below *mean_dev.  workspace: chunk headers,
// tdr_prefix_workspace_bytes(n).  total_out: one device float.
int tdr_chain_total(const float* raw, const float* mean_dev, int kind, int64_t n, float* total_out, void* workspace,
                    hipStream_t st) {
  const int64_t nch64 = cdiv(n, (int64_t)PFXM_CHUNK);
  if (nch64 > (1 << 24)) return fail(TDR_ERR_ARG, "chain_total: n too large");
  const int nch = (int)nch64;
  PfxChunk* ch = reinterpret_cast<PfxChunk*>(workspace);
  ChainSrc s{raw, mean_dev, kind};
  // The first 32 768 addends — where the sum crosses most of its binades — go through the one-workgroup machinery of
  // uw_small_kernel (chain_head_kernel, below); the chunk walk starts behind them.
  const int c_first = chain_head_chunks(n);
  hipLaunchKernelGGL(chain_sum_kernel, dim3(nch), dim3(PFXW_THREADS), 0, st, s, n, ch);
  if (nch > c_first)
    hipLaunchKernelGGL(chain_summary_kernel, dim3(nch - c_first), dim3(PFXW_THREADS), 0, st, s, n, ch, c_first);
  float* r_first = reinterpret_cast<float*>(&ch[0].r0);   // (a header slot the chains do not use)
  if (c_first > 0) {
    const int rc = chain_head_launch(raw, mean_dev, kind, c_first * PFXM_CHUNK, r_first, st);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(chain_walk_kernel, dim3(1), dim3(PFXW_THREADS), 0, st, s, n, (const PfxChunk*)ch, nch, total_out,
                     c_first, (const float*)r_first);
  return TDR_OK;
}

// ---- ParticleFilter::update's statistics for small particle sets, in ONE launch ------------------------------------
// src/particle_filter.cpp:107-147 for n <= TDR_UW_SMALL_MAX_N — the reference's own operating point (20 000 particles,
// src/top_down_render.cpp:53).  One workgroup; the raw weights are staged into LDS once (n floats, at most 128 KB) and
// every pass — the two exact serial chains, the counts, fill / normalise / argmax — runs out of LDS; the weights are
// written to memory once, at the end.  Same results as the multi-workgroup path of tdr_filter.hip, bit for bit in
// `sum`, `mean`, `bottom_stddev` (tests/test_gpu_parity.py::test_update_weights_serial_chains_bit_exact).
//
// The chains: head one by one, then chunk by chunk with chain_walk_chunk (no prediction pass: at most 8 chunks).  (One
// stretch over the whole array instead of chunks re-classifies every remaining addend at each binade crossing: 80 us
// against 48 at 20 000 weights.)
__device__ __forceinline__ float uws_chain_total(int kind, int n, float mean) {
  __shared__ double head[CHAIN_HEAD];
  __shared__ float s_head_r;
  const int hn = min(CHAIN_HEAD, n);
  pfx_sync();
  for (int t = threadIdx.x; t < hn; t += PFXW_THREADS) head[t] = uws_addend(kind, t, mean);
  pfx_sync();
  if (threadIdx.x == 0) {
    float run = 0.f;
    if (kind == 0) {
      // float + float: (float)((double)a + (double)b) == a + b for every pair of floats (53 >= 2 * 24 + 2 bits: the
      // double sum rounds to the same float), and a float add is a third of the dependent latency
      for (int t = 0; t < hn; t++) run += (float)head[t];
    } else {
      for (int t = 0; t < hn; t++) run = (float)((double)run + head[t]);
    }
    s_head_r = run;
  }
  pfx_sync();
  UW_STAMP(kind ? 5 : 2);
  float r = s_head_r;   // workgroup-uniform
  const ChainSrc s{nullptr, nullptr, kind};
  const int nch = (n + PFXM_CHUNK - 1) / PFXM_CHUNK;
  for (int c = 0; c < nch; c++) {
    const int lo = c * PFXM_CHUNK;
    const int cnt = min((int)PFXM_CHUNK, n - lo);
    if (c == 0 && hn >= cnt) continue;
    chain_walk_chunk<true>(s, lo, cnt, mean, r, c == 0 ? hn : 0);
  }
  return r;
}
// ---- the same chains, wave by wave (default) --------------------------------------------------------------------------
// The weights are cut into wave-chunks of 512 (one wave, 8 consecutive addends a lane).  The pass before the chain (the
// count of valid weights / of weights below the mean) leaves the double sum of every wave-chunk's addends behind; from the
// sums before a chunk every wave predicts the binade the chain is in when it enters and leaves it.  Then, side by side:
//   * wave 0 adds the first wave-chunk one by one (the sum starts at zero and doubles every few addends);
//   * the other waves summarise their chunks: a chunk predicted to stay in one binade as its parity pair (D0, D1) there;
//     a chunk predicted to cross into the next binade as the parity pairs of every LANE (8 addends) in both binades.
// One barrier later wave 0 walks the chunk list with the exact running sum, with register-level moves only — no workgroup
// barrier inside the serial part: a one-binade chunk whose prediction holds is one integer add; in a crossing chunk a
// scan of the lane pairs of the first binade finds the lane the sum crosses in, that lane's 8 addends are really added,
// and a scan of the lane pairs of the second binade carries the sum to the chunk's end.  Whatever fits neither (a wrong
// prediction, irregular addends, two crossings in one chunk) is carried through element by element by uws_wave_walk.
// The bits are the serial chain's whatever was predicted, as in the multi-workgroup path.
#define UWS_WC (64 * CHAIN_K)   // addends per wave-chunk
#define UWS_THREADS 1024
#define UWS_DUAL_SLOTS 8        // crossing chunks summarised lane by lane (the sum doubles log2(n / 512) times behind the head)
#define UWS_DUAL_FLAG 0x100
struct UwsShared {
  double csum[64];                       // double sum of every wave-chunk's addends
  int code[64];                          // -1: carried through; binade; binade | UWS_DUAL_FLAG | slot << 16
  unsigned d0[64], d1[64];               // parity pair of a one-binade chunk
  uint4 dual[UWS_DUAL_SLOTS][64];        // crossing chunks: scans of the lane pairs in the binade entered (x, y) / the next (z, w)
  double shd[UWS_THREADS / 64];
  float total;
};
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int uws_min_dpp(int v) {
  return min(v, __builtin_amdgcn_update_dpp(0x7FFFFFFF, v, CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ int uws_wave_min(int v) {   // minimum over the wave, in every lane (uniform)
  v = uws_min_dpp<0x111, 0xF>(v);
  v = uws_min_dpp<0x112, 0xF>(v);
  v = uws_min_dpp<0x114, 0xF>(v);
  v = uws_min_dpp<0x118, 0xF>(v);
  v = uws_min_dpp<0x142, 0xA>(v);
  v = uws_min_dpp<0x143, 0xC>(v);
  return __builtin_amdgcn_readlane(v, 63);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double uws_add_dpp(double v) {   // lanes without a source add 0.0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double uws_wave_scan_d(double v) {   // inclusive sum over the lanes before and this one
  v = uws_add_dpp<0x111, 0xF>(v);
  v = uws_add_dpp<0x112, 0xF>(v);
  v = uws_add_dpp<0x114, 0xF>(v);
  v = uws_add_dpp<0x118, 0xF>(v);
  v = uws_add_dpp<0x142, 0xA>(v);
  v = uws_add_dpp<0x143, 0xC>(v);
  return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned uws_addu_dpp(unsigned v) {
  return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ unsigned uws_wave_scan_u(unsigned v) {
  v = uws_addu_dpp<0x111, 0xF>(v);
  v = uws_addu_dpp<0x112, 0xF>(v);
  v = uws_addu_dpp<0x114, 0xF>(v);
  v = uws_addu_dpp<0x118, 0xF>(v);
  v = uws_addu_dpp<0x142, 0xA>(v);
  v = uws_addu_dpp<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ double uws_readlane_d(double v, int lane) {   // lane: wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// addend e of the chain if `take`, else 0.0 — the LDS read is unconditional (n: number of weights staged)
// (the staged array ends in UWS_PAD unused words, so a read up to that far behind the last weight stays inside it)
#define UWS_PAD 16
__device__ __forceinline__ double uws_addend_if(int kind, int e, bool take, float mean) {
  const double x = uws_addend(kind, e, mean);
  return take ? x : 0.0;
}
// the `sum` chain's addends are floats: pfx_classify's float arithmetic (exact, see there) with chain_classify's bound
__device__ __forceinline__ float uws_addend_f(int kind, int e, bool take) {
  extern __shared__ float uws_lraw[];
  const float v = uws_lraw[uws_idx(e)];
  return (take && (kind == 2 || v == v)) ? v : 0.f;   // :111-115 (`sum` skips NaN weights; the running sum, kind 2, does not)
}
__device__ __forceinline__ void uws_classify_f(float wv, unsigned re, unsigned& f, bool& tie, bool& bad) {
  const float t = wv * __uint_as_float((277u - re) << 23);   // w / u, u = 2^(re - 150)
  bad = !(wv >= 0.f) || !(t < 4194304.f);
  const float fl = floorf(t);
  const float fr = t - fl;
  tie = fr == 0.5f;
  f = (unsigned)fl + (fr > 0.5f ? 1u : 0u);
  if (bad) { f = 0; tie = false; }
}
// an addend of chain `kind` (a compile-time constant where it matters): a float in the `sum` chain, else a double
struct UwsX { float f; double d; };
__device__ __forceinline__ UwsX uws_load_x(int kind, int e, bool take, float mean) {
  UwsX x;
  x.f = 0.f; x.d = 0.0;
  if (kind != 1) x.f = uws_addend_f(kind, e, take); else x.d = uws_addend_if(kind, e, take, mean);
  return x;
}
__device__ __forceinline__ void uws_classify_x(int kind, const UwsX& x, unsigned re, unsigned& f, bool& tie, bool& bad) {
  if (kind != 1) uws_classify_f(x.f, re, f, tie, bad); else chain_classify(x.d, re, f, tie, bad);
}
__device__ __forceinline__ float uws_mant(unsigned re, unsigned state) { return __uint_as_float((re << 23) | (state & 0x7FFFFFu)); }
// wave-chunk [lo, lo + cnt) carried through by the calling wave from its element `pos` on, entered with the running sum r
// (wave-uniform): chain_walk_chunk on one wave
__device__ __forceinline__ float uws_wave_walk(int kind, int lo, int cnt, float mean, float r, int pos) {
  const int t0 = (threadIdx.x & 63) * CHAIN_K;
  double xv[CHAIN_K];
#pragma unroll
  for (int k = 0; k < CHAIN_K; k++) xv[k] = uws_addend_if(kind, lo + t0 + k, t0 + k < cnt, mean);
  while (pos < cnt) {
#ifdef TDR_UW_TIMELINE
    if ((threadIdx.x & 63) == 0 && kind < 2) g_uw_tl[10 + kind]++;
#endif
    const unsigned rb = __float_as_uint(r);
    const unsigned re = rb >> 23;   // sign included
    int first = cnt;
    unsigned st[CHAIN_K] = {};
    const bool regular = re >= PFXM_RE_MIN && re <= PFXM_RE_MAX;
    if (r != r) return r;   // NaN stays NaN
    if (!regular) {
      // zero / tiny / huge / inf running sum: zero addends change nothing (an infinite sum: finite ones), the next other
      // one is really added
      const bool isinf = (rb & 0x7FFFFFFFu) == 0x7F800000u;
#pragma unroll
      for (int k = 0; k < CHAIN_K; k++) {
        const int li = t0 + k;
        const bool acts = isinf ? !(fabs(xv[k]) < INFINITY) : (xv[k] != 0.0);
        if (li >= pos && li < cnt && acts) first = min(first, li);
      }
    } else {
      const unsigned R = (rb & 0x7FFFFFu) | 0x800000u;
      unsigned f[CHAIN_K];
      unsigned tiebits = 0u;
      PfxPair mine = {0u, 0u};
#pragma unroll
      for (int k = 0; k < CHAIN_K; k++) {
        const int li = t0 + k;
        bool bad, tie;
        chain_classify(xv[k], re, f[k], tie, bad);
        if (li < pos || li >= cnt) { f[k] = 0u; tie = false; bad = false; }
        if (bad) first = min(first, li);
        tiebits |= tie ? (1u << k) : 0u;
        mine = pfx_compose(mine, pfx_element_pair(f[k], tie));
      }
      const PfxPair ex = pfx_pair_dpp<0x138, 0xF>(pfx_pair_wave_scan(mine));   // exclusive: wave_shr:1
      unsigned state = R + ((R & 1u) ? ex.a1 : ex.a0);
#pragma unroll
      for (int k = 0; k < CHAIN_K; k++) {
        state += f[k] + (((tiebits >> k) & 1u) ? ((state + f[k]) & 1u) : 0u);
        st[k] = state;
        const int li = t0 + k;
        if (li >= pos && li < cnt && state >= (1u << 24)) first = min(first, li);
      }
    }
    const int stop = uws_wave_min(first);   // first addend that is really added (cnt: none)
    unsigned sv = 0u;   // in the lane that holds it: the mantissa before the stop, the addend at the stop
    double xs = 0.0;
#pragma unroll
    for (int k = 0; k < CHAIN_K; k++) {
      if (t0 + k == stop - 1) sv = st[k];
      if (t0 + k == stop) xs = xv[k];
    }
    if (regular && stop > pos) r = uws_mant(re, (unsigned)__builtin_amdgcn_readlane((int)sv, (stop - 1) / CHAIN_K));
    if (stop < cnt) {
      const double x = uws_readlane_d(xs, stop / CHAIN_K);
      r = (float)((double)r + x);   // one real addition, in the reference's types
      pos = stop + 1;
    } else {
      pos = cnt;
    }
  }
  return r;
}
// a chunk summarised lane by lane — the inclusive scans of the lanes' parity pairs in the binade predicted at its entry
// (x, y) and in the next one (z, w) — entered with r; pos: the element of the chunk the returned sum stands before (cnt: the
// chunk is done, else uws_wave_walk takes over there)
__device__ __forceinline__ float uws_wave_dual(int kind, int lo, int cnt, float mean, float r, const uint4* pq, int& pos) {
  const int lane = threadIdx.x & 63, t0 = lane * CHAIN_K;
  const unsigned rb = __float_as_uint(r), re = rb >> 23, R = (rb & 0x7FFFFFu) | 0x800000u;
  const uint4 v = pq[lane];
  const unsigned after = R + ((R & 1u) ? v.y : v.x);   // mantissa behind this lane's addends, while below 2^24
  const unsigned long long cross = __ballot(after >= (1u << 24));
  pos = cnt;
  if (cross == 0ull) return uws_mant(re, (unsigned)__builtin_amdgcn_readlane((int)after, 63));
  const int L = __ffsll((long long)cross) - 1;   // the lane the sum leaves the binade in: its addends are really added
  float rl = uws_mant(re, L > 0 ? (unsigned)__builtin_amdgcn_readlane((int)after, L - 1) : R);   // exact before lane L
  UwsX x[CHAIN_K];
#pragma unroll
  for (int k = 0; k < CHAIN_K; k++) x[k] = uws_load_x(kind, lo + t0 + k, t0 + k < cnt, mean);
#pragma unroll
  for (int k = 0; k < CHAIN_K; k++) {
    if (kind != 1) rl += x[k].f;   // float + float: the same sum as through double
    else rl = (float)((double)rl + x[k].d);
  }
  const float r2 = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(rl), L));
  const unsigned rb2 = __float_as_uint(r2), R2 = (rb2 & 0x7FFFFFu) | 0x800000u;
  const int next = (L + 1) * CHAIN_K;
  if (next >= cnt) return r2;
  pos = next;
  if ((rb2 >> 23) != re + 1u) return r2;
  // the lanes behind L in the next binade: scan(63) = scan(L) + rest((p + scan(L)) & 1) for an entering parity p, so a
  // scan(L) that is the same for both parities gives the rest for both
  const unsigned qL = (unsigned)__builtin_amdgcn_readlane((int)v.z, L);
  if (qL != (unsigned)__builtin_amdgcn_readlane((int)v.w, L)) return r2;
  const unsigned q63 = (unsigned)__builtin_amdgcn_readlane((int)(((R2 + qL) & 1u) ? v.w : v.z), 63);
  if (q63 >= PFXM_SAT || R2 + (q63 - qL) >= (1u << 24)) return r2;
  pos = cnt;
  return uws_mant(re + 1u, R2 + (q63 - qL));
}
// the first hn addends one by one: 64 at a time into the lanes, then every lane runs the same chain over them
__device__ __forceinline__ float uws_head(int kind, int hn, float mean) {
  const int lane = threadIdx.x & 63;
  float run = 0.f;
  if (kind == 0) {
    // float + float: (float)((double)a + (double)b) == a + b for every pair of floats (53 >= 2 * 24 + 2 bits)
    float nxt = (float)uws_addend_if(0, min(lane, hn - 1), lane < hn, 0.f);
    for (int b = 0; b < hn; b += 64) {
      const int cur = (int)__float_as_uint(nxt);
      nxt = (float)uws_addend_if(0, min(b + 64 + lane, hn - 1), b + 64 + lane < hn, 0.f);
#pragma unroll
      for (int j = 0; j < 64; j++) run += __uint_as_float((unsigned)__builtin_amdgcn_readlane(cur, j));
    }
  } else {
    double nxt = uws_addend_if(1, min(lane, hn - 1), lane < hn, mean);
    for (int b = 0; b < hn; b += 64) {
      const double cur = nxt;
      nxt = uws_addend_if(1, min(b + 64 + lane, hn - 1), b + 64 + lane < hn, mean);
#pragma unroll
      for (int j = 0; j < 64; j++) run = (float)((double)run + uws_readlane_d(cur, j));
    }
  }
  return run;
}
// the running sum of the resample: the same head, and every lane keeps the running sum behind its own addend, which
// replaces the weight in the staged array
__device__ __forceinline__ float pfs_head(int hn) {
  extern __shared__ float uws_lraw[];
  const int lane = threadIdx.x & 63;
  float run = 0.f;
  float nxt = uws_addend_f(2, min(lane, hn - 1), lane < hn);
  for (int b = 0; b < hn; b += 64) {
    const int cur = (int)__float_as_uint(nxt);
    nxt = uws_addend_f(2, min(b + 64 + lane, hn - 1), b + 64 + lane < hn);
    float mine = 0.f;
#pragma unroll
    for (int j = 0; j < 64; j++) {
      run += __uint_as_float((unsigned)__builtin_amdgcn_readlane(cur, j));
      mine = (lane == j) ? run : mine;
    }
    if (b + lane < hn) uws_lraw[uws_idx(b + lane)] = mine;
  }
  return run;
}
// sh.csum holds the chunks' double sums and a barrier has passed since they were written; rin (kind 2): the running sum
// in front of every wave-chunk
__device__ __forceinline__ float uws_chain_total_waves(int kind, int n, float mean, UwsShared& sh, float* rin = nullptr) {
  constexpr int NW = UWS_THREADS / 64;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), t0 = lane * CHAIN_K;
  const int nwc = (n + UWS_WC - 1) / UWS_WC;   // <= 64
  float r = 0.f;
  if (wave == 0) {
    r = kind == 2 ? pfs_head(min(UWS_WC, n)) : uws_head(kind, min(UWS_WC, n), mean);
  } else {
    // the running double sum before / after each chunk -> what is predicted for it (lane c: chunk c)
    int my_code;
    {
      const double own = lane < nwc ? sh.csum[lane] : 0.0;
      const double after = uws_wave_scan_d(own), before = after - own;
      const int re_b = (int)(__float_as_uint((float)before) >> 23), re_a = (int)(__float_as_uint((float)after) >> 23);   // sign included
      const bool inr = lane >= 1 && lane < nwc && re_b >= PFXM_RE_MIN && re_a <= PFXM_RE_MAX;
      const bool dualc = inr && re_a == re_b + 1;
      const unsigned long long dm = __ballot(dualc);
      const int slot = __popcll(dm & ((1ull << lane) - 1ull));
      my_code = (inr && re_a == re_b) ? re_b : (dualc && slot < UWS_DUAL_SLOTS) ? (re_b | UWS_DUAL_FLAG | (slot << 16)) : -1;
    }
    for (int c = wave; c < nwc; c += NW - 1) {   // chunks 1 .. nwc - 1 over waves 1 .. NW - 1
      const int code = __builtin_amdgcn_readlane(my_code, c);
      const int lo = c * UWS_WC, cnt = min(UWS_WC, n - lo);
      PfxPair P = {0u, 0u}, Q = {0u, 0u};
      bool anybad = false;
      if (code >= 0) {
        const unsigned re = (unsigned)(code & 0xFF);
        const bool dual = (code & UWS_DUAL_FLAG) != 0;
        UwsX x[CHAIN_K];   // (all eight reads in flight before the first use)
#pragma unroll
        for (int k = 0; k < CHAIN_K; k++) x[k] = uws_load_x(kind, lo + t0 + k, t0 + k < cnt, mean);
        // without a rounding tie in the chunk a lane's pair is (s, s), s the plain sum of its increments
        bool anytie = false;
        unsigned f; bool tie, bad;
#pragma unroll
        for (int k = 0; k < CHAIN_K; k++) {
          uws_classify_x(kind, x[k], re, f, tie, bad);
          anybad |= bad; anytie |= tie;
          P.a0 += f;   // (f < 2^22: no overflow in a lane, none below 2^31 in a wave)
        }
        if (dual) {
#pragma unroll
          for (int k = 0; k < CHAIN_K; k++) {
            uws_classify_x(kind, x[k], re + 1u, f, tie, bad);
            anybad |= bad; anytie |= tie;
            Q.a0 += f;
          }
        }
        if (__ballot(anytie) == 0ull) {
          P.a0 = P.a1 = uws_wave_scan_u(P.a0);
          if (dual) Q.a0 = Q.a1 = uws_wave_scan_u(Q.a0);
        } else {
          P = PfxPair{0u, 0u}; Q = PfxPair{0u, 0u};
#pragma unroll
          for (int k = 0; k < CHAIN_K; k++) {
            uws_classify_x(kind, x[k], re, f, tie, bad);
            P = pfx_compose(P, pfx_element_pair(f, tie));
          }
          P = pfx_pair_wave_scan(P);
          if (dual) {
#pragma unroll
            for (int k = 0; k < CHAIN_K; k++) {
              uws_classify_x(kind, x[k], re + 1u, f, tie, bad);
              Q = pfx_compose(Q, pfx_element_pair(f, tie));
            }
            Q = pfx_pair_wave_scan(Q);
          }
        }
        if (dual) sh.dual[code >> 16][lane] = make_uint4(P.a0, P.a1, Q.a0, Q.a1);
      }
      const bool ok = code >= 0 && __ballot(anybad) == 0ull;
      if (lane == 63) {
        sh.code[c] = (ok && ((code & UWS_DUAL_FLAG) || (P.a0 < (1u << 24) && P.a1 < (1u << 24)))) ? code : -1;
        sh.d0[c] = P.a0; sh.d1[c] = P.a1;
      }
    }
  }
  pfx_sync();
  UW_STAMP(kind ? 5 : 2);
  if (wave == 0) {   // the walk
    const int v_code = (lane >= 1 && lane < nwc) ? sh.code[lane] : -1;
    const unsigned v_d0 = (lane >= 1 && lane < nwc) ? sh.d0[lane] : 0u, v_d1 = (lane >= 1 && lane < nwc) ? sh.d1[lane] : 0u;
    float v_rin = 0.f;   // lane c: the running sum in front of chunk c
    int c = 1;
    while (c < nwc) {
      const unsigned rb = __float_as_uint(r);
      const int re = (int)(rb >> 23);
      const unsigned R = (rb & 0x7FFFFFu) | 0x800000u;
      bool overflow = false;   // chunk c was predicted to stay in this binade and does not
      // the run of chunks from c on that were summarised in the binade the sum is in: one scan of their pairs (plain
      // sums when none of them holds a rounding tie) carries the sum through all of them
      const unsigned long long other = ~__ballot(lane < nwc && v_code == re) & (~0ull << c);
      const int cend = other ? __ffsll((long long)other) - 1 : 64;   // the run is [c, cend)
      if (cend > c) {
        const bool in_run = lane >= c && lane < cend;
        unsigned after;   // lane l: mantissa behind chunk l, while below 2^24
        if (__ballot(in_run && v_d0 != v_d1) == 0ull) {
          after = R + uws_wave_scan_u(in_run ? v_d0 : 0u);
        } else {
          const PfxPair in = pfx_pair_wave_scan(in_run ? PfxPair{v_d0, v_d1} : PfxPair{0u, 0u});
          after = R + ((R & 1u) ? in.a1 : in.a0);
        }
        const unsigned long long cross = __ballot(in_run && after >= (1u << 24));
        const int c1 = cross ? __ffsll((long long)cross) - 1 : cend;   // chunks [c, c1) are taken
        if (kind == 2) {
          const unsigned prev = (unsigned)__builtin_amdgcn_update_dpp(0, (int)after, 0x138, 0xF, 0xF, false);   // wave_shr:1
          if (lane >= c && lane < c1) v_rin = lane == c ? r : uws_mant((unsigned)re, prev);
        }
        if (c1 > c) r = uws_mant((unsigned)re, (unsigned)__builtin_amdgcn_readlane((int)after, c1 - 1));
#ifdef TDR_UW_TIMELINE
        if (threadIdx.x == 0) g_uw_tl[12] += c1 - c;
#endif
        c = c1;
        if (c1 == cend) continue;
        overflow = true;
      }
      const int lo = c * UWS_WC, cnt = min(UWS_WC, n - lo);
      const int code = __builtin_amdgcn_readlane(v_code, c);
      if (kind == 2) v_rin = (lane == c) ? r : v_rin;
      int pos = 0;
      const int re_now = (int)(__float_as_uint(r) >> 23);
      if (!overflow && code >= 0 && re_now <= PFXM_RE_MAX && (code & 0x1FF) == (re_now | UWS_DUAL_FLAG)) {
        r = uws_wave_dual(kind, lo, cnt, mean, r, sh.dual[code >> 16], pos);
        UW_COUNT(13);
      }
      if (pos < cnt) {
        UW_COUNT(14);
        r = uws_wave_walk(kind, lo, cnt, mean, r, pos);
      }
      c++;
    }
    if (kind == 2) rin[lane] = v_rin;
    if (lane == 0) sh.total = r;
  }
  pfx_sync();
  return sh.total;
}
template <int NT>
__device__ __forceinline__ double uws_sum_d(double v, double* sh) {   // block sum, fixed order: a pure function of the inputs
  v = uws_wave_scan_d(v);   // lane 63: the wave's sum
  pfx_sync();
  if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = v;
  pfx_sync();
  double t = 0;
  for (int w = 0; w < NT / 64; w++) t += sh[w];
  return t;
}
template <bool WAVES>
__global__ __launch_bounds__(WAVES ? UWS_THREADS : PFXW_THREADS) void uw_small_kernel(const float* __restrict__ raw,
                                                                const float* __restrict__ last_dist, int n,
                                                                float* __restrict__ w, float* __restrict__ info) {
  extern __shared__ float uws_lraw[];   // [uws_idx(n)]: the raw weights, later the weights being normalised
  float* const lraw = uws_lraw;
  constexpr int nt = WAVES ? UWS_THREADS : PFXW_THREADS;
  __shared__ UwsShared ush;
  double* const shd = ush.shd;
  __shared__ float sh_best[nt / 64];
  __shared__ int sh_besti[nt / 64];
  const int tid = threadIdx.x;
#ifdef TDR_UW_TIMELINE
  if (tid == 0) { for (int k = 10; k < 16; k++) g_uw_tl[k] = 0; }
#endif
  UW_STAMP(0);
  float sum = 0.f, mean = 0.f, bsum = 0.f;
  long long num_valid = 0, num_under = 0;
  double valid_sum = 0.0;   // WAVES: the valid weights summed in double (the sum of the wave-chunks' sums)
  if constexpr (WAVES) {
    // chain 0: stage the weights and count the valid ones (:108-116), `sum`; chain 1: count the weights below the mean,
    // `bottom_stddev` (:118-126).  Either pass works wave-chunk by wave-chunk and leaves the chunks' double sums behind
    // for the chain's predictions.  (Unrolled: `kind` is a constant in each copy — 47 us against 53 with one copy.)
    const int lane = tid & 63, nwc = (n + UWS_WC - 1) / UWS_WC;
#pragma unroll
    for (int kind = 0; kind < 2; kind++) {
      int cnt = 0;
      for (int c = tid >> 6; c < nwc; c += nt / 64) {
        const int base = c * UWS_WC + lane;
        float v[CHAIN_K];
        if (kind == 0) {   // (eight loads of 64 consecutive weights in flight per wave)
#pragma unroll
          for (int m = 0; m < CHAIN_K; m++) v[m] = base + 64 * m < n ? raw[base + 64 * m] : __uint_as_float(0x7FC00000u);
#pragma unroll
          for (int m = 0; m < CHAIN_K; m++)
            if (base + 64 * m < n) lraw[uws_idx(base + 64 * m)] = v[m];
        } else {
#pragma unroll
          for (int m = 0; m < CHAIN_K; m++) {
            const float x = lraw[uws_idx(min(base + 64 * m, n - 1))];
            v[m] = base + 64 * m < n ? x : __uint_as_float(0x7FC00000u);
          }
        }
        double acc = 0;
#pragma unroll
        for (int m = 0; m < CHAIN_K; m++) {
          const bool take = kind == 0 ? v[m] == v[m] : (v[m] == v[m] && v[m] < mean);
          double x = (double)(kind == 0 ? v[m] : v[m] - mean);
          if (kind) x = x * x;
          cnt += take ? 1 : 0;
          acc += take ? x : 0.0;
        }
        acc = uws_wave_scan_d(acc);
        if (lane == 63) ush.csum[c] = acc;
      }
      const long long count = (long long)uws_sum_d<nt>((double)cnt, shd);   // (its barriers also publish the staged weights)
      if (kind == 0) valid_sum = uws_readlane_d(uws_wave_scan_d(lane < nwc ? ush.csum[lane] : 0.0), 63);
      UW_STAMP(kind ? 4 : 1);
      const float total = uws_chain_total_waves(kind, n, mean, ush);   // serial float chain, exact
      UW_STAMP(kind ? 6 : 3);
      if (kind == 0) {
        sum = total; num_valid = count;
        mean = sum / (float)num_valid;  // :117 (0/0 -> NaN like the reference)
      } else {
        bsum = total; num_under = count;
      }
    }
  } else {
    // stage (four loads in flight per thread), count the valid weights on the way (:108-116)
    double cnt = 0;
    int i = tid;
    for (; i + 3 * nt < n; i += 4 * nt) {
      const float v0 = raw[i], v1 = raw[i + nt], v2 = raw[i + 2 * nt], v3 = raw[i + 3 * nt];
      lraw[uws_idx(i)] = v0; lraw[uws_idx(i + nt)] = v1; lraw[uws_idx(i + 2 * nt)] = v2; lraw[uws_idx(i + 3 * nt)] = v3;
      cnt += (v0 == v0 ? 1.0 : 0.0) + (v1 == v1 ? 1.0 : 0.0) + (v2 == v2 ? 1.0 : 0.0) + (v3 == v3 ? 1.0 : 0.0);
    }
    for (; i < n; i += nt) {
      const float v = raw[i];
      lraw[uws_idx(i)] = v;
      cnt += (v == v) ? 1.0 : 0.0;
    }
    num_valid = (long long)uws_sum_d<nt>(cnt, shd);   // (its barriers also publish the staged weights)
    UW_STAMP(1);
    sum = uws_chain_total(0, n, 0.f);   // serial float chain, exact
    UW_STAMP(3);
    mean = sum / (float)num_valid;  // :117 (0/0 -> NaN like the reference)
    // :118-126  bottom_stddev (serial float chain with double addends, exact) and the count below the mean
    double cu = 0;
    for (int i = tid; i < n; i += nt) {
      const float v = lraw[uws_idx(i)];
      cu += (v == v && v < mean) ? 1.0 : 0.0;
    }
    num_under = (long long)uws_sum_d<nt>(cu, shd);
    UW_STAMP(4);
    bsum = uws_chain_total(1, n, mean);
    UW_STAMP(6);
  }
  const float bottom = sqrtf(bsum / (float)num_under);
  const bool fallback = (sum == 0.f || num_under < 1);  // :129
  const float fill = mean - bottom;                      // :133
  const float fn = (float)n;
  float fs1;
  double s2 = 0;
  if constexpr (WAVES) {
    // :130-135 without a pass of its own: the filled weights sum to the valid ones plus `fill` for every NaN (all ones in
    // the fallback); the fill itself happens where the weight is read next.  The last travel distances are requested
    // sixteen per thread at a time, ahead of the divisions.
    fs1 = (float)(fallback ? (double)n : valid_sum + (double)(n - num_valid) * (double)fill);
    UW_STAMP(7);
    for (int i0 = tid; i0 < n; i0 += 16 * nt) {
      float ld[16];
#pragma unroll
      for (int j = 0; j < 16; j++) ld[j] = i0 + j * nt < n ? last_dist[i0 + j * nt] : 0.f;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int i = i0 + j * nt;
        if (i < n) {   // :135, :138-141
          float v = lraw[uws_idx(i)];
          v = (fallback ? 1.f : (v != v ? fill : v)) / fs1;
          const float d = fminf(ld[j] * 5.f, 1.f);
          v = d * v + (1.f - d) / fn;
          lraw[uws_idx(i)] = v;
          s2 += (double)v;
        }
      }
    }
  } else {
    pfx_sync();                                            // every thread is done reading the raw weights
    double s1a = 0;
    for (int i = tid; i < n; i += nt) {
      float v = lraw[uws_idx(i)];
      v = fallback ? 1.f : (v != v ? fill : v);
      lraw[uws_idx(i)] = v;
      s1a += (double)v;
    }
    fs1 = (float)uws_sum_d<nt>(s1a, shd);
    UW_STAMP(7);
    auto one = [&](int i, float ld) {   // :135, :138-141
      float v = lraw[uws_idx(i)] / fs1;
      const float d = fminf(ld * 5.f, 1.f);
      v = d * v + (1.f - d) / fn;
      lraw[uws_idx(i)] = v;
      s2 += (double)v;
    };
    int i = tid;
    for (; i + 3 * nt < n; i += 4 * nt) {
      const float l0 = last_dist[i], l1 = last_dist[i + nt], l2 = last_dist[i + 2 * nt], l3 = last_dist[i + 3 * nt];
      one(i, l0); one(i + nt, l1); one(i + 2 * nt, l2); one(i + 3 * nt, l3);   // same order as one by one
    }
    for (; i < n; i += nt) one(i, last_dist[i]);
  }
  const float fs2 = (float)uws_sum_d<nt>(s2, shd);
  UW_STAMP(8);
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int i = tid; i < n; i += nt) {  // :142, :145-147 (first maximum)
    const float v = lraw[uws_idx(i)] / fs2;
    w[i] = v;
    if (v > best || (v == best && i < besti)) { best = v; besti = i; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_down(best, o, 64);
    const int oi = __shfl_down(besti, o, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  pfx_sync();
  if ((tid & 63) == 0) { sh_best[tid >> 6] = best; sh_besti[tid >> 6] = besti; }
  pfx_sync();
  if (tid == 0) {
    for (int k = 1; k < nt / 64; k++)
      if (sh_best[k] > best || (sh_best[k] == best && sh_besti[k] < besti)) { best = sh_best[k]; besti = sh_besti[k]; }
    if (besti == 0x7fffffff) besti = 0;
    info[0] = __int_as_float(besti);
    info[1] = sum; info[2] = mean; info[3] = bottom; info[4] = fallback ? 1.f : 0.f;
    info[5] = (float)num_valid; info[6] = (float)num_under; info[7] = 0.f;
  }
  UW_STAMP(9);
}
static int g_uw_waves = 1;   // 0 = the chains chunk by chunk on the whole workgroup (A/B and debugging)
extern "C" int tdr_config_uw_waves(int on) {   // < 0: query only
  if (on >= 0) g_uw_waves = on ? 1 : 0;
  return g_uw_waves;
}
int tdr_uw_small(const float* raw, const float* last_dist, int64_t n, float* w, float* info, hipStream_t st) {
  if (n < 1 || n > 32768) return fail(TDR_ERR_ARG, "uw_small: n out of range");
  const size_t lds = ((size_t)n + (size_t)(n >> 5) + 1 + UWS_PAD) * sizeof(float);
  static bool attr_set[64] = {false};   // per device: the attribute lives with the device's copy of the code object
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  if (dev >= 64 || !attr_set[dev]) {   // more than the default 64 KB of dynamic LDS
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(uw_small_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 133 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(uw_small_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 133 * 1024) != hipSuccess)
      return fail(TDR_ERR_HIP, "uw_small: cannot raise the dynamic LDS limit");
    if (dev < 64) attr_set[dev] = true;
  }
  if (g_uw_waves)
    hipLaunchKernelGGL(uw_small_kernel<true>, dim3(1), dim3(UWS_THREADS), lds, st, raw, last_dist, (int)n, w, info);
  else
    hipLaunchKernelGGL(uw_small_kernel<false>, dim3(1), dim3(PFXW_THREADS), lds, st, raw, last_dist, (int)n, w, info);
  return TDR_OK;
}

// ---- the running sum of the resample for small particle sets, in ONE launch ------------------------------------------
// `running_sum += weights_[j]` (src/particle_filter.cpp:179) for n <= 32 768 with the machinery of the statistics chains:
// the weights staged in LDS, wave-chunks predicted and summarised side by side with the one-by-one head, one wave walking
// the chunk list with the exact sum (it notes the sum in front of every chunk), then every wave redoes its chunks from
// their exact starting sums and leaves each addend's running sum in place of the weight; a last pass takes the running
// maximum (the running sum itself when no weight is negative or NaN) and writes the outputs.
struct PfsShared {
  UwsShared u;
  float rin[64];    // running sum in front of every wave-chunk
  float cmax[64];   // largest running sum inside it (NaN skipped)
  int irregular;    // a negative or NaN weight exists: the running sum is not its own running maximum
};
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float pfs_max_dpp(float v) {
  return fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)0xFF800000u, (int)__float_as_uint(v), CTRL,
                                                                        ROW_MASK, 0xF, false)));
}
__device__ __forceinline__ float pfs_wave_scan_max(float v) {   // inclusive maximum over the lanes before and this one
  v = pfs_max_dpp<0x111, 0xF>(v);
  v = pfs_max_dpp<0x112, 0xF>(v);
  v = pfs_max_dpp<0x114, 0xF>(v);
  v = pfs_max_dpp<0x118, 0xF>(v);
  v = pfs_max_dpp<0x142, 0xA>(v);
  v = pfs_max_dpp<0x143, 0xC>(v);
  return v;
}
// wave-chunk [lo, lo + cnt) from its exact starting sum r: every addend's running sum, in place of the weight
__device__ __forceinline__ void pfs_wave_fill(int lo, int cnt, float r) {
  extern __shared__ float uws_lraw[];
  const int t0 = (threadIdx.x & 63) * CHAIN_K;
  float x[CHAIN_K], pv[CHAIN_K];
#pragma unroll
  for (int k = 0; k < CHAIN_K; k++) { x[k] = uws_addend_f(2, lo + t0 + k, t0 + k < cnt); pv[k] = 0.f; }
  int pos = 0;
  while (pos < cnt) {
    const unsigned rb = __float_as_uint(r);
    const unsigned re = rb >> 23;   // sign included
    if (r != r) {                   // NaN stays NaN
#pragma unroll
      for (int k = 0; k < CHAIN_K; k++)
        if (t0 + k >= pos) pv[k] = r;
      break;
    }
    int first = cnt;
    unsigned st[CHAIN_K] = {};
    const bool regular = re >= PFXM_RE_MIN && re <= PFXM_RE_MAX;
    if (!regular) {
      const bool isinf = (rb & 0x7FFFFFFFu) == 0x7F800000u;
#pragma unroll
      for (int k = 0; k < CHAIN_K; k++) {
        const int li = t0 + k;
        const bool acts = isinf ? !(fabsf(x[k]) < INFINITY) : (x[k] != 0.f);
        if (li >= pos && li < cnt && acts) first = min(first, li);
      }
    } else {
      const unsigned R = (rb & 0x7FFFFFu) | 0x800000u;
      unsigned f[CHAIN_K];
      unsigned tiebits = 0u, lane_sum = 0u;
#pragma unroll
      for (int k = 0; k < CHAIN_K; k++) {
        const int li = t0 + k;
        bool bad, tie;
        uws_classify_f(x[k], re, f[k], tie, bad);
        if (li < pos || li >= cnt) { f[k] = 0u; tie = false; bad = false; }
        if (bad) first = min(first, li);
        tiebits |= tie ? (1u << k) : 0u;
        lane_sum += f[k];
      }
      unsigned state;
      if (__ballot(tiebits != 0u) == 0ull) {   // no rounding tie: plain sums
        state = R + (uws_wave_scan_u(lane_sum) - lane_sum);
      } else {
        PfxPair mine = {0u, 0u};
#pragma unroll
        for (int k = 0; k < CHAIN_K; k++) mine = pfx_compose(mine, pfx_element_pair(f[k], ((tiebits >> k) & 1u) != 0u));
        const PfxPair ex = pfx_pair_dpp<0x138, 0xF>(pfx_pair_wave_scan(mine));   // exclusive: wave_shr:1
        state = R + ((R & 1u) ? ex.a1 : ex.a0);
      }
#pragma unroll
      for (int k = 0; k < CHAIN_K; k++) {
        state += f[k] + (((tiebits >> k) & 1u) ? ((state + f[k]) & 1u) : 0u);
        st[k] = state;
        const int li = t0 + k;
        if (li >= pos && li < cnt && state >= (1u << 24)) first = min(first, li);
      }
    }
    const int stop = uws_wave_min(first);   // first addend that is really added (cnt: none)
    unsigned sv = 0u;
    float xs = 0.f;
#pragma unroll
    for (int k = 0; k < CHAIN_K; k++) {
      const int li = t0 + k;
      if (li >= pos && li < stop) pv[k] = regular ? uws_mant(re, st[k]) : r;
      if (li == stop - 1) sv = st[k];
      if (li == stop) xs = x[k];
    }
    if (regular && stop > pos) r = uws_mant(re, (unsigned)__builtin_amdgcn_readlane((int)sv, (stop - 1) / CHAIN_K));
    if (stop < cnt) {
      r += __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(xs), stop / CHAIN_K));   // one real addition
#pragma unroll
      for (int k = 0; k < CHAIN_K; k++)
        if (t0 + k == stop) pv[k] = r;
      pos = stop + 1;
    } else {
      pos = cnt;
    }
  }
#pragma unroll
  for (int k = 0; k < CHAIN_K; k++)
    if (t0 + k < cnt) uws_lraw[uws_idx(lo + t0 + k)] = pv[k];
}
__global__ __launch_bounds__(UWS_THREADS) void pfx_small_kernel(const float* __restrict__ w, int n,
                                                                float* __restrict__ runmax, float* __restrict__ prefix_opt,
                                                                float* __restrict__ tail) {   // tail (optional): sum, maximum
  extern __shared__ float uws_lraw[];
  float* const lraw = uws_lraw;
  __shared__ PfsShared sh;
  constexpr int nt = UWS_THREADS, NW = UWS_THREADS / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwc = (n + UWS_WC - 1) / UWS_WC;
  if (tid == 0) sh.irregular = 0;
  pfx_sync();
  // stage wave-chunk by wave-chunk; their double sums for the predictions
  bool irr = false;
  for (int c = wave; c < nwc; c += NW) {
    const int base = c * UWS_WC + lane;
    float v[CHAIN_K];
#pragma unroll
    for (int m = 0; m < CHAIN_K; m++) v[m] = base + 64 * m < n ? w[base + 64 * m] : 0.f;
    double acc = 0;
#pragma unroll
    for (int m = 0; m < CHAIN_K; m++) {
      if (base + 64 * m < n) lraw[uws_idx(base + 64 * m)] = v[m];
      irr |= !(v[m] >= 0.f);
      acc += (double)v[m];
    }
    acc = uws_wave_scan_d(acc);
    if (lane == 63) sh.u.csum[c] = acc;
  }
  if (__ballot(irr) != 0ull && lane == 0) atomicOr(&sh.irregular, 1);
  pfx_sync();
  const float total = uws_chain_total_waves(2, n, 0.f, sh.u, sh.rin);   // (ends in a barrier)
  // every chunk again from its exact starting sum (the first one was filled by the head)
  for (int c = 1 + wave; c < nwc; c += NW) pfs_wave_fill(c * UWS_WC, min(UWS_WC, n - c * UWS_WC), sh.rin[c]);
  pfx_sync();
  if (sh.irregular == 0) {   // the running sum never falls: it is its own running maximum
    if (tail && tid == 0) { tail[0] = total; tail[1] = total; }
    for (int i = tid; i < n; i += nt) {
      const float v = lraw[uws_idx(i)];
      runmax[i] = v;
      if (prefix_opt) prefix_opt[i] = v;
    }
    return;
  }
  const int t0 = lane * CHAIN_K;
  for (int c = wave; c < nwc; c += NW) {
    const int lo = c * UWS_WC, cnt = min(UWS_WC, n - lo);
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < CHAIN_K; k++) {
      const float v = lraw[uws_idx(lo + t0 + k)];
      if (t0 + k < cnt && v == v) m = fmaxf(m, v);
    }
    m = pfs_wave_scan_max(m);
    if (lane == 63) sh.cmax[c] = m;
  }
  pfx_sync();
  const float before = pfs_wave_scan_max(lane < nwc ? sh.cmax[lane] : -INFINITY);   // lane c: maximum up to chunk c's end
  if (tail && tid == 63) { tail[0] = total; tail[1] = before; }
  for (int c = wave; c < nwc; c += NW) {
    const int lo = c * UWS_WC, cnt = min(UWS_WC, n - lo);
    const float carry = c > 0 ? __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(before), c - 1)) : -INFINITY;
    float pv[CHAIN_K], lm[CHAIN_K];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < CHAIN_K; k++) {
      pv[k] = lraw[uws_idx(lo + t0 + k)];
      if (t0 + k < cnt && pv[k] == pv[k]) m = fmaxf(m, pv[k]);
      lm[k] = m;
    }
    const float incl = pfs_wave_scan_max(m);
    float ex = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)0xFF800000u, (int)__float_as_uint(incl), 0x138, 0xF, 0xF, false));
    ex = fmaxf(ex, carry);
#pragma unroll
    for (int k = 0; k < CHAIN_K; k++)
      if (t0 + k < cnt) {
        runmax[lo + t0 + k] = fmaxf(lm[k], ex);
        if (prefix_opt) prefix_opt[lo + t0 + k] = pv[k];
      }
  }
}
#define TDR_PFX_SMALL_MAX_N 32768
static int pfx_small(const float* w, int64_t n, float* runmax, float* prefix_opt, hipStream_t st, float* tail = nullptr) {
  if (n < 1 || n > TDR_PFX_SMALL_MAX_N) return fail(TDR_ERR_ARG, "pfx_small: n out of range");
  const size_t lds = ((size_t)n + (size_t)(n >> 5) + 1 + UWS_PAD) * sizeof(float);
  static bool attr_set[64] = {false};   // per device: the attribute lives with the device's copy of the code object
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  if (dev >= 64 || !attr_set[dev]) {   // more than the default 64 KB of dynamic LDS
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(pfx_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            133 * 1024) != hipSuccess)
      return fail(TDR_ERR_HIP, "pfx_small: cannot raise the dynamic LDS limit");
    if (dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(pfx_small_kernel, dim3(1), dim3(UWS_THREADS), lds, st, w, (int)n, runmax, prefix_opt, tail);
  return TDR_OK;
}

// The head of a statistics chain over more than 32 768 weights: its first n_head addends (whole 4096-chunks, at most
// 32 768) staged into LDS and summed by uws_chain_total_waves; the chunk walk of chain_walk_kernel starts behind them.
template <int KIND>
__global__ __launch_bounds__(UWS_THREADS) void chain_head_kernel(const float* __restrict__ raw, const float* __restrict__ mean_dev,
                                                                 int n, float* __restrict__ r_out) {
  extern __shared__ float uws_lraw[];
  __shared__ UwsShared ush;
  const int tid = threadIdx.x, lane = tid & 63, nwc = (n + UWS_WC - 1) / UWS_WC;
  const float mean = KIND ? *mean_dev : 0.f;
  for (int c = tid >> 6; c < nwc; c += UWS_THREADS / 64) {
    const int base = c * UWS_WC + lane;
    float v[CHAIN_K];
#pragma unroll
    for (int m = 0; m < CHAIN_K; m++) v[m] = base + 64 * m < n ? raw[base + 64 * m] : __uint_as_float(0x7FC00000u);
    double acc = 0;
#pragma unroll
    for (int m = 0; m < CHAIN_K; m++) {
      if (base + 64 * m < n) uws_lraw[uws_idx(base + 64 * m)] = v[m];
      const bool take = KIND == 0 ? v[m] == v[m] : (v[m] == v[m] && v[m] < mean);
      double x = (double)(KIND == 0 ? v[m] : v[m] - mean);
      if (KIND) x = x * x;
      acc += take ? x : 0.0;
    }
    acc = uws_wave_scan_d(acc);
    if (lane == 63) ush.csum[c] = acc;
  }
  pfx_sync();
  const float r = uws_chain_total_waves(KIND, n, mean, ush);
  if (tid == 0) *r_out = r;
}
static int chain_head_launch(const float* raw, const float* mean_dev, int kind, int n_head, float* r_out, hipStream_t st) {
  const size_t lds = ((size_t)n_head + (size_t)(n_head >> 5) + 1 + UWS_PAD) * sizeof(float);
  static bool attr_set[64] = {false};   // per device: the attribute lives with the device's copy of the code object
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  if (dev >= 64 || !attr_set[dev]) {   // more than the default 64 KB of dynamic LDS
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(chain_head_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            133 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(chain_head_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            133 * 1024) != hipSuccess)
      return fail(TDR_ERR_HIP, "chain_head: cannot raise the dynamic LDS limit");
    if (dev < 64) attr_set[dev] = true;
  }
  if (kind == 0)
    hipLaunchKernelGGL(chain_head_kernel<0>, dim3(1), dim3(UWS_THREADS), lds, st, raw, mean_dev, n_head, r_out);
  else
    hipLaunchKernelGGL(chain_head_kernel<1>, dim3(1), dim3(UWS_THREADS), lds, st, raw, mean_dev, n_head, r_out);
  return TDR_OK;
}

// Dispatch (tools/bench_prefix_modes.py on MI355X; us at n = 1k / 4k / 8k / 20k / 100k: one wave 16 / 43 / 84 / 208 /
// 1036, one workgroup 60 / 129 / 156 / 235 / 417, multi-workgroup 38 / 59 / 53 / 74 / 101):
#define TDR_PFX_MULTI_MIN_N 6144    // with a workspace: the multi-workgroup scan from here on, one wave below
#define TDR_PFX_EXACT_MIN_N 24576   // without a workspace: one workgroup from here on, one wave below
#define TDR_PFX_SMALL_MIN_N 256     // the one-launch kernel from here up to TDR_PFX_SMALL_MAX_N
extern "C" int64_t tdr_prefix_workspace_bytes(int64_t n) {
  return n < 1 ? 0 : (int64_t)sizeof(PfxChunk) * cdiv(n, (int64_t)PFXM_CHUNK);
}
static int prefix_multi(const float* w, int64_t n, float* runmax_out, float* prefix_out, void* workspace,
                        hipStream_t st) {
  const int64_t nch64 = cdiv(n, (int64_t)PFXM_CHUNK);
  if (nch64 > (1 << 24)) return fail(TDR_ERR_ARG, "prefix: n too large");
  const int nch = (int)nch64;
  PfxChunk* ch = reinterpret_cast<PfxChunk*>(workspace);
  // the first 32 768 weights through pfx_small_kernel, the chunk walk behind them (see tdr_chain_total)
  const int c_first = chain_head_chunks(n);
  float* tail = reinterpret_cast<float*>(&ch[0].sum);   // (chunk 0's sum is not read once the summaries are made)
  hipLaunchKernelGGL(pfx_chunk_sum_kernel, dim3(nch), dim3(PFXM_THREADS), 0, st, w, n, ch);
  if (nch > c_first)
    hipLaunchKernelGGL(pfx_chunk_summary_kernel, dim3(nch - c_first), dim3(PFXM_THREADS), 0, st, w, n, ch, c_first);
  if (c_first > 0) {
    const int rc = pfx_small(w, (int64_t)c_first * PFXM_CHUNK, runmax_out, prefix_out, st, tail);
    if (rc) return rc;
  }
  const int head_len = tdr_config_prefix_head(-1);
  hipLaunchKernelGGL(pfx_walk_kernel, dim3(1), dim3(PFXW_THREADS), 0, st, w, n, ch, nch, runmax_out, prefix_out,
                     head_len, c_first, (const float*)tail);
  hipLaunchKernelGGL(pfx_chunk_fill_kernel, dim3(nch), dim3(PFXM_THREADS), 0, st, w, n, (const PfxChunk*)ch,
                     runmax_out, prefix_out);
  return TDR_OK;
}
extern "C" int tdr_k_prefix(const float* w, int64_t n, float* runmax_out, void* workspace, void* stream) {
  if (!w || !runmax_out || n < 1) return fail(TDR_ERR_ARG, "prefix: bad arguments");
  if (g_pfx_small && n >= TDR_PFX_SMALL_MIN_N && n <= TDR_PFX_SMALL_MAX_N) {
    const int rc = pfx_small(w, n, runmax_out, nullptr, (hipStream_t)stream);
    if (rc) return rc;
  } else if (workspace && n >= TDR_PFX_MULTI_MIN_N) {
    const int rc = prefix_multi(w, n, runmax_out, nullptr, workspace, (hipStream_t)stream);
    if (rc) return rc;
  } else if (n < TDR_PFX_EXACT_MIN_N)
    hipLaunchKernelGGL(prefix_serial_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, w, n, runmax_out);
  else
    hipLaunchKernelGGL(prefix_exact_kernel, dim3(1), dim3(PFX_THREADS), 0, (hipStream_t)stream, w, n, runmax_out,
                       (float*)nullptr);
  LAUNCH_CHECK("prefix");
  return TDR_OK;
}
// Test / diagnostic entry: mode 0 = serial kernel, 1 = exact parallel kernel in one workgroup, 2 = the multi-workgroup
// scan (needs a workspace of tdr_prefix_workspace_bytes(n)), 3 = the one-launch kernel for n <= 32 768; prefix_out
// (optional, modes 1 to 3) receives the raw running sums.
extern "C" int tdr_k_prefix_mode(const float* w, int64_t n, int mode, float* runmax_out, float* prefix_out,
                                 void* workspace, void* stream) {
  if (!w || !runmax_out || n < 1) return fail(TDR_ERR_ARG, "prefix_mode: bad arguments");
  if (mode == 2 && !workspace) return fail(TDR_ERR_ARG, "prefix_mode: mode 2 needs a workspace");
  if (mode == 3) {
    const int rc = pfx_small(w, n, runmax_out, prefix_out, (hipStream_t)stream);
    if (rc) return rc;
  } else if (mode == 0)
    hipLaunchKernelGGL(prefix_serial_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, w, n, runmax_out);
  else if (mode == 2) {
    const int rc = prefix_multi(w, n, runmax_out, prefix_out, workspace, (hipStream_t)stream);
    if (rc) return rc;
  } else
    hipLaunchKernelGGL(prefix_exact_kernel, dim3(1), dim3(PFX_THREADS), 0, (hipStream_t)stream, w, n, runmax_out,
                       prefix_out);
  LAUNCH_CHECK("prefix_mode");
  return TDR_OK;
}
