// tdr_rng.hip — the reference's random stream on the device: std::mt19937 + libstdc++'s std::normal_distribution<float> /
// std::uniform_real_distribution<float>, word for word.
//
// ParticleFilter::propagate (src/particle_filter.cpp:86-92) walks the particles in index order and every
// StateParticle::propagate (src/state_particle.cpp:64-73) draws from ONE shared std::mt19937 through fresh distribution
// objects: N(0, theta_cov dist) once, N(0, pos_cov dist) twice, and — unless the scale is frozen — N(1, ...) once.
// libstdc++'s normal_distribution is Marsaglia's polar method (bits/random.tcc): an ATTEMPT takes two 32-bit words
// (generate_canonical<float, 24> = one word each), x = 2 c0 - 1, y = 2 c1 - 1, and is accepted when 0 < x^2 + y^2 <= 1;
// an accepted attempt yields y * mult and keeps x * mult for the object's next call, mult = sqrt(-2 log(r2) / r2).  A fresh
// object has nothing saved, so the stream of a propagate call is nothing but attempts, and particle p owns the accepted
// attempts 3p (theta: y), 3p + 1 (dx: y, dy: the saved x), 3p + 2 (scale: y) — 2p, 2p + 1 with the scale frozen.  That
// makes the stream parallel:
//   mt_fill_kernel      the generator's untempered state blocks, one after the other (one wave per STRETCH of blocks: a
//                       block of 624 words in four LDS round trips — 192 elements of a block depend on nothing younger
//                       than 227 elements);
//   mt_jump_kernel      a long call (more than MT_JUMP_STRIDE blocks) is cut into stretches whose first blocks are reached
//                       by JUMPING AHEAD — the recurrence is linear over GF(2): the state J words on is g(A) s with
//                       g = t^J mod the characteristic polynomial, constants of the generator (tdr_mt_jump.h, computed by
//                       tools/gen_mt_jump.py) — in ceil(log2 stretches) rounds of doubling; then the stretches fill side
//                       by side.  A million particles' 13 500 blocks: 4.7 ms on one wave, ~0.4 ms this way;
//   mt_attempt_kernel   every attempt of the budget: tempering, the two canonical floats, accepted or not;
//   rocPRIM             exclusive scan of the accepted flags = the rank of every accepted attempt;
//   mt_normal_kernel    accepted attempt of rank a -> particle a / 3 (or a / 2), with glibc's logf restated (tdr_logf.h) and
//                       correctly rounded division / square root; the attempt that completes the last particle records
//                       how many words the call consumed;
//   mt_advance_kernel   the generator state behind exactly those words.
// The uniform draw of the systematic resample (src/particle_filter.cpp:172-173) is mt_uniform_kernel: one word.
// The state lives on the device in libstdc++'s own representation (624 words + the index of the next word), so it moves
// to and from a host std::mt19937 through the engine's stream operators (tdr_rng_get_state_host / _set_state_host) and the
// host object can take over at any time (particle initialisation, a generator shared with the caller).
// tests/test_rng.py: normals, consumed words and the state afterwards against the host's distributions for many sizes.
#include <rocprim/device/device_scan.hpp>

#include <random>
#include <sstream>

#include "tdr_common.h"
#include "tdr_logf.h"
#include "tdr_mt_jump.h"

#define MT_N 624
#define MT_M 397

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);                    // (d = 0xffffffff)
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
// generate_canonical<float, 24>(mt19937) (bits/random.tcc): one word, float(u) / 2^32, clamped below 1
__device__ __forceinline__ float mt_canonical(uint32_t u) {
  float c = (float)u * 0x1p-32f;     // u32 -> float rounds to nearest; the scaling is exact
  return c >= 1.f ? 0x1.fffffep-1f : c;
}
// One twist of the state in LDS (mersenne_twister_engine::_M_gen_rand), by ONE wave: elements in ascending order, 192 at
// a time — element k needs the OLD x[k], x[k + 1] and, from 227 on, the NEW x[k - 227], written at least one batch earlier;
// element 623 reads the NEW x[0] (it is the last of the engine's loop), long written when its batch comes.  So a batch is
// nine LDS reads in flight together, then three writes.  One wave, and the LDS serves a wave's instructions in order: no
// hardware barrier is needed — only the compiler must keep a batch's loads in front of its stores and the stores in front
// of the next batch's loads (with workgroup barriers a block took 0.9 us).  x holds MT_X words: the lanes behind element
// 623 work on padding instead of being predicated off.
#define MT_X 1040
template <int S0, int NS>
__device__ __forceinline__ void mt_twist_batch(uint32_t* x, int lane, uint32_t* __restrict__ out) {
  uint32_t nv[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const int k = 64 * (S0 + s) + lane;
    const int k1 = k == MT_N - 1 ? 0 : k + 1;
    const int si = k < MT_N - MT_M ? k + MT_M : k - (MT_N - MT_M);
    const uint32_t y = (x[k] & 0x80000000u) | (x[k1] & 0x7fffffffu);
    nv[s] = x[si] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const int k = 64 * (S0 + s) + lane;
    x[k] = nv[s];
    if (out && k < MT_N) out[k] = nv[s];   // the new block to global memory, from the registers
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void mt_twist(uint32_t* x, int lane, uint32_t* __restrict__ out) {
  mt_twist_batch<0, 3>(x, lane, out);
  mt_twist_batch<3, 3>(x, lane, out);
  mt_twist_batch<6, 3>(x, lane, out);
  mt_twist_batch<9, 1>(x, lane, out);   // elements 576 .. 623 (and 16 lanes of padding)
}

// raw [nblocks][624]: block 0 = the state as it is, block b = the state b twists later (untempered words).
// One wave per stretch of `stride` blocks: stretch w starts from `state` (w == 0) or from the block a jump left at
// raw[w * stride] (mt_jump_kernel) and fills the blocks behind it up to the next stretch's first / to nblocks.
__global__ __launch_bounds__(64) void mt_fill_kernel(const uint32_t* __restrict__ state, int nblocks, int stride,
                                                     uint32_t* __restrict__ raw) {
  __shared__ uint32_t x[MT_X];
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * stride;
  const uint32_t* __restrict__ from = blockIdx.x == 0 ? state : raw + b0 * MT_N;
  for (int k = lane; k < MT_N; k += 64) {
    const uint32_t v = from[k];
    x[k] = v;
    if (blockIdx.x == 0) raw[k] = v;
  }
  __syncthreads();
  const int64_t b1 = min((int64_t)nblocks, b0 + stride);
  for (int64_t b = b0 + 1; b < b1; b++) mt_twist(x, lane, raw + b * MT_N);
}

// One jump: the block at the start of stretch k -> the block at the start of stretch k + 2^level, i.e. MT_JUMP_STRIDE * 2^level
// blocks further down the stream, without walking there.  With A the generator's one-word step and g = t^J mod its
// characteristic polynomial split as g(t) = sum_j t^(624 j) r_j(t) (tdr_mt_jump.h), the target is sum_j B^j (r_j(A) s), B = A^624
// = the block step: r_j(A) s is the XOR of the 624-word windows [i, i + 624) of (block || next block) over the set bits i of
// r_j — 16 waves, two chunks j each, a lane ten output words — and the sum over j is Horner's rule in B, 31 block steps by one
// wave.  A window's first word carries garbage in its low 31 bits (the generator's state has 19937 bits: only the top bit of
// its first word belongs to it); they are restored at the end from x[J + 623] = x[J + 396] ^ T(top(x[J - 1]) | low(x[J])).
// Stretch 0's first block is the state itself.
__global__ __launch_bounds__(1024) void mt_jump_kernel(const uint32_t* __restrict__ state, uint32_t* __restrict__ raw, int level) {
  __shared__ uint32_t stream[2 * MT_N + 64];   // block || next block (+ padding read by lanes without an output word)
  __shared__ uint32_t xs[MT_X];
  __shared__ uint32_t R[MT_JUMP_CHUNKS][MT_N];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = blockIdx.x;
  const uint32_t* __restrict__ src = k == 0 ? state : raw + (int64_t)k * MT_JUMP_STRIDE * MT_N;
  uint32_t* __restrict__ dst = raw + (int64_t)(k + (1 << level)) * MT_JUMP_STRIDE * MT_N;
  for (int i = tid; i < MT_X; i += 1024) xs[i] = i < MT_N ? src[i] : 0u;
  for (int i = tid; i < 2 * MT_N + 64; i += 1024) stream[i] = i < MT_N ? src[i] : 0u;
  __syncthreads();
  if (wave == 0) mt_twist(xs, lane, nullptr);
  __syncthreads();
  for (int i = tid; i < MT_N; i += 1024) stream[MT_N + i] = xs[i];
  __syncthreads();
  for (int j = wave; j < MT_JUMP_CHUNKS; j += 16) {
    uint32_t acc[10];
#pragma unroll
    for (int t = 0; t < 10; t++) acc[t] = 0;
    for (int wd = 0; wd < 20; wd++) {
      uint32_t bits = __builtin_amdgcn_readfirstlane(MT_JUMP[level][j][wd]);
      while (bits) {   // (wave-uniform)
        const int i = 32 * wd + __builtin_ctz(bits);
        bits &= bits - 1;
#pragma unroll
        for (int t = 0; t < 10; t++) acc[t] ^= stream[i + lane + 64 * t];
      }
    }
#pragma unroll
    for (int t = 0; t < 10; t++)
      if (lane + 64 * t < MT_N) R[j][lane + 64 * t] = acc[t];
  }
  __syncthreads();
  if (wave != 0) return;
  for (int p = lane; p < MT_X; p += 64) xs[p] = p < MT_N ? R[MT_JUMP_CHUNKS - 1][p] : 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int j = MT_JUMP_CHUNKS - 2; j >= 0; j--) {
    mt_twist(xs, lane, nullptr);
    for (int p = lane; p < MT_N; p += 64) xs[p] ^= R[j][p];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    const uint32_t v = xs[MT_N - 1] ^ xs[MT_M - 1];
    const uint32_t lsb = v >> 31;
    const uint32_t y = ((v ^ (lsb ? 0x9908b0dfu : 0u)) << 1) | lsb;
    xs[0] = (xs[0] & 0x80000000u) | (y & 0x7fffffffu);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int p = lane; p < MT_N; p += 64) dst[p] = xs[p];
}

struct MtAttempt {
  float x, y, r2;
  bool ok;
};
__device__ __forceinline__ MtAttempt mt_attempt(const uint32_t* __restrict__ raw, int64_t g) {
  MtAttempt a;
  const float c0 = mt_canonical(mt_temper(raw[g])), c1 = mt_canonical(mt_temper(raw[g + 1]));
  a.x = (float)((double)(2.0f * c0) - 1.0);   // result_type(2.0) * aurng() - 1.0: float product, double difference
  a.y = (float)((double)(2.0f * c1) - 1.0);
  a.r2 = a.x * a.x + a.y * a.y;               // (compiled with -ffp-contract=off: two roundings, like the host's)
  a.ok = !((double)a.r2 > 1.0 || (double)a.r2 == 0.0);
  return a;
}
// attempt t uses words p0 + 2t, p0 + 2t + 1 of the raw stream
__global__ __launch_bounds__(256) void mt_attempt_kernel(const uint32_t* __restrict__ raw, const uint32_t* __restrict__ state,
                                                         int64_t nattempts, uint32_t* __restrict__ flags) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nattempts) return;
  flags[t] = mt_attempt(raw, (int64_t)state[MT_N] + 2 * t).ok ? 1u : 0u;
}
// per: accepted attempts a particle owns (3, or 2 with the scale frozen); particles [lo, hi) of n are written, to z4[p - lo]
__global__ __launch_bounds__(256) void mt_normal_kernel(const uint32_t* __restrict__ raw, const uint32_t* __restrict__ state,
                                                        const uint32_t* __restrict__ flags, const uint32_t* __restrict__ rank,
                                                        int64_t nattempts, int64_t n, int per, int64_t lo, int64_t hi,
                                                        float* __restrict__ z4, uint32_t* __restrict__ consumed) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nattempts || !flags[t]) return;
  const int64_t a = rank[t];
  if (a >= n * per) return;
  if (a == n * per - 1) *consumed = (uint32_t)(2 * (t + 1));
  const int64_t p = a / per;
  const int slot = (int)(a - p * per);
  if (p < lo || p >= hi) return;
  const MtAttempt at = mt_attempt(raw, (int64_t)state[MT_N] + 2 * t);
  // std::sqrt(-2 * std::log(r2) / r2): the float overloads (glibc logf, IEEE division and square root)
  const float mult = sqrtf(-2.f * tdr_libm::logf_t<true>(at.r2) / at.r2);
  float* z = z4 + 4 * (p - lo);
  const float vy = at.y * mult * 1.f + 0.f, vx = at.x * mult * 1.f + 0.f;   // ret * stddev + mean of the {0, 1} objects
  if (slot == 0) z[0] = vy;                       // theta: a fresh object's first value; its saved one is dropped
  else if (slot == 1) { z[1] = vy; z[2] = vx; }   // dx, dy: one object, two calls
  else z[3] = vy;                                 // scale
  if (per == 2 && slot == 1) z[3] = 0.f;
}
// the state behind `consumed` words of the raw stream; error word: 1 when the budget of attempts did not suffice
__global__ __launch_bounds__(64) void mt_advance_kernel(const uint32_t* __restrict__ raw, int nblocks,
                                                        const uint32_t* __restrict__ consumed, uint32_t* __restrict__ state) {
  const uint32_t c = *consumed;
  if (c == 0xFFFFFFFFu) {   // not enough accepted attempts inside the budget: leave the state, raise the flag
    if (threadIdx.x == 0) state[MT_N + 1] = 1u;
    return;
  }
  const int64_t q = (int64_t)state[MT_N] + c;
  int b = (int)(q / MT_N), p = (int)(q % MT_N);
  if (p == 0 && b > 0) { b--; p = MT_N; }   // "all of block b - 1 is used" is the same state as "none of block b"
  if (b >= nblocks) {
    if (threadIdx.x == 0) state[MT_N + 1] = 1u;
    return;
  }
  uint32_t v[(MT_N + 63) / 64];
  for (int k = threadIdx.x, i = 0; k < MT_N; k += 64, i++) v[i] = raw[(int64_t)b * MT_N + k];
  for (int k = threadIdx.x, i = 0; k < MT_N; k += 64, i++) state[k] = v[i];
  if (threadIdx.x == 0) state[MT_N] = (uint32_t)p;
}
// std::uniform_real_distribution<float>(0, 1)(gen): one word
__global__ __launch_bounds__(64) void mt_uniform_kernel(uint32_t* __restrict__ state, float* __restrict__ out) {
  __shared__ uint32_t x[MT_X];
  const int lane = threadIdx.x;
  uint32_t p = state[MT_N];
  uint32_t word;
  if (p >= MT_N) {   // (uniform) the block is used up: twist first
    for (int k = lane; k < MT_N; k += 64) x[k] = state[k];
    __syncthreads();
    mt_twist(x, lane, state);
    p = 0;
    word = x[0];
  } else {
    word = state[p];
  }
  if (lane == 0) {
    *out = mt_canonical(mt_temper(word)) * (1.f - 0.f) + 0.f;   // (b - a) * canonical + a
    state[MT_N] = p + 1;
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// 0: every call's raw stream on one wave (A/B, tests); 1 (default): long calls in stretches (tdr_config_tuning("mt_stretches"))
static int g_mt_stretches = 1;
extern "C" int tdr_config_mt_stretches(int on) {   // < 0: query only
  if (on >= 0) g_mt_stretches = on ? 1 : 0;
  return g_mt_stretches;
}
// attempts a call may look at: the expected 3.82 per particle-normal... (acceptance pi / 4) plus 5 % and ten standard
// deviations: running out has probability ~1e-23 (and is reported, never silent)
static int64_t mt_attempt_budget(int64_t need) {
  return (int64_t)std::ceil((double)need / 0.7853981633974483 * 1.05 + 12.0 * std::sqrt((double)need + 1.0) + 64.0);
}
static size_t mt_scan_bytes(int64_t nattempts) {
  size_t bytes = 0;
  uint32_t* p = nullptr;
  hipError_t e = rocprim::exclusive_scan(nullptr, bytes, p, p, 0u, (size_t)nattempts, rocprim::plus<uint32_t>(), (hipStream_t)0, false);
  if (e != hipSuccess || bytes == 0) bytes = (size_t)nattempts / 16 + 65536;
  return (bytes + 255) / 256 * 256;
}
struct MtWs {
  int64_t nattempts, nblocks;
  size_t off_flags, off_rank, off_scan, off_consumed, total;
};
static MtWs mt_ws(int64_t n, int per) {
  MtWs w;
  w.nattempts = mt_attempt_budget(n * per);
  w.nblocks = (2 * w.nattempts + MT_N) / MT_N + 2;   // from anywhere inside block 0
  size_t o = (size_t)w.nblocks * MT_N * 4;
  auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
  o = (o + 255) / 256 * 256;
  w.off_flags = take((size_t)w.nattempts * 4);
  w.off_rank = take((size_t)w.nattempts * 4);
  w.off_scan = take(mt_scan_bytes(w.nattempts));
  w.off_consumed = take(256);
  w.total = o;
  return w;
}
extern "C" size_t tdr_rng_dev_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  return mt_ws(n, 3).total;
}
// z4_out [hi - lo][4] = the standard normals {theta, dx, dy, scale} of particles [lo, hi) of n, exactly what
// tdr_propagate_normals_host draws for them; `state` (device, TDR_RNG_STATE_WORDS words) moves on by the words the WHOLE
// call of n particles consumes (every rank of a sharded filter keeps the same state).
extern "C" int tdr_k_rng_propagate_normals(uint32_t* state, int64_t n, int64_t lo, int64_t hi, int scale_freeze, float* z4_out,
                                           void* workspace, void* stream) {
  if (!state || !z4_out || !workspace) return fail(TDR_ERR_ARG, "rng_propagate_normals: null pointer");
  if (n < 1 || lo < 0 || hi > n || lo > hi) return fail(TDR_ERR_ARG, "rng_propagate_normals: bad range");
  hipStream_t s = (hipStream_t)stream;
  const int per = scale_freeze ? 2 : 3;
  const MtWs W = mt_ws(n, per);
  char* base = reinterpret_cast<char*>(workspace);
  uint32_t* raw = reinterpret_cast<uint32_t*>(base);
  uint32_t* flags = reinterpret_cast<uint32_t*>(base + W.off_flags);
  uint32_t* rank = reinterpret_cast<uint32_t*>(base + W.off_rank);
  uint32_t* consumed = reinterpret_cast<uint32_t*>(base + W.off_consumed);
  HIP_TRY(hipMemsetAsync(consumed, 0xFF, 4, s));
  // the raw stream: one wave walks it, or — a long call — stretches of MT_JUMP_STRIDE blocks side by side, their first blocks
  // reached by jumping ahead in rounds of doubling (round m: the 2^m stretch starts there are, each 2^m stretches further)
  const int64_t nstretch = cdiv(W.nblocks, (int64_t)MT_JUMP_STRIDE);
  if (g_mt_stretches && nstretch > 1 && nstretch <= ((int64_t)1 << MT_JUMP_LEVELS)) {
    for (int m = 0; ((int64_t)1 << m) < nstretch; m++) {
      const int64_t have = (int64_t)1 << m, jumps = std::min(have, nstretch - have);
      hipLaunchKernelGGL(mt_jump_kernel, dim3((unsigned)jumps), dim3(1024), 0, s, (const uint32_t*)state, raw, m);
      LAUNCH_CHECK("mt_jump");
    }
    hipLaunchKernelGGL(mt_fill_kernel, dim3((unsigned)nstretch), dim3(64), 0, s, (const uint32_t*)state, (int)W.nblocks,
                       MT_JUMP_STRIDE, raw);
  } else {
    hipLaunchKernelGGL(mt_fill_kernel, dim3(1), dim3(64), 0, s, (const uint32_t*)state, (int)W.nblocks, (int)W.nblocks, raw);
  }
  LAUNCH_CHECK("mt_fill");
  const unsigned blocks = (unsigned)cdiv(W.nattempts, 256);
  hipLaunchKernelGGL(mt_attempt_kernel, dim3(blocks), dim3(256), 0, s, (const uint32_t*)raw, (const uint32_t*)state,
                     W.nattempts, flags);
  LAUNCH_CHECK("mt_attempt");
  size_t scan_bytes = mt_scan_bytes(W.nattempts);
  HIP_TRY(rocprim::exclusive_scan(base + W.off_scan, scan_bytes, flags, rank, 0u, (size_t)W.nattempts,
                                  rocprim::plus<uint32_t>(), s, false));
  hipLaunchKernelGGL(mt_normal_kernel, dim3(blocks), dim3(256), 0, s, (const uint32_t*)raw, (const uint32_t*)state,
                     (const uint32_t*)flags, (const uint32_t*)rank, W.nattempts, n, per, lo, hi, z4_out, consumed);
  LAUNCH_CHECK("mt_normal");
  hipLaunchKernelGGL(mt_advance_kernel, dim3(1), dim3(64), 0, s, (const uint32_t*)raw, (int)W.nblocks,
                     (const uint32_t*)consumed, state);
  LAUNCH_CHECK("mt_advance");
  return TDR_OK;
}
extern "C" int tdr_k_rng_uniform(uint32_t* state, float* out, void* stream) {
  if (!state || !out) return fail(TDR_ERR_ARG, "rng_uniform: null pointer");
  hipLaunchKernelGGL(mt_uniform_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, out);
  LAUNCH_CHECK("mt_uniform");
  return TDR_OK;
}

// A host std::mt19937 <-> the device representation: words[0 .. 623] the engine's state array, words[624] the index of the
// next word (624: a twist comes first), words[625] the error flag of the device kernels (0).  Through the engine's own
// stream operators — the textual form libstdc++ defines: the 624 words, then the index.
extern "C" int tdr_rng_get_state_host(void* rng, uint32_t* words) {
  if (!rng || !words) return fail(TDR_ERR_ARG, "rng_get_state: null pointer");
  std::ostringstream os;
  os << *(std::mt19937*)rng;
  std::istringstream is(os.str());
  for (int k = 0; k <= MT_N; k++) {
    unsigned long long v = 0;
    if (!(is >> v)) return fail(TDR_ERR_ARG, "rng_get_state: unexpected engine format");
    words[k] = (uint32_t)v;
  }
  for (int k = MT_N + 1; k < TDR_RNG_STATE_WORDS; k++) words[k] = 0;
  return TDR_OK;
}
extern "C" int tdr_rng_set_state_host(void* rng, const uint32_t* words) {
  if (!rng || !words) return fail(TDR_ERR_ARG, "rng_set_state: null pointer");
  if (words[MT_N] > MT_N) return fail(TDR_ERR_ARG, "rng_set_state: index %u", words[MT_N]);
  if (words[MT_N + 1] != 0) return fail(TDR_ERR_ARG, "rng_set_state: the device generator ran out of its attempt budget");
  std::ostringstream os;
  for (int k = 0; k <= MT_N; k++) os << words[k] << (k < MT_N ? " " : "");
  std::istringstream is(os.str());
  is >> *(std::mt19937*)rng;
  if (!is) return fail(TDR_ERR_ARG, "rng_set_state: the engine refused the state");
  return TDR_OK;
}

// self tests of the logf restatement (tests/test_libm.py): on the host, and on the device
extern "C" int tdr_logf_host(const float* x, int64_t n, float* out) {
  if (!x || !out || n < 0) return fail(TDR_ERR_ARG, "logf_host: bad arguments");
  for (int64_t i = 0; i < n; i++) out[i] = tdr_libm::logf_t<true>(x[i]);
  return TDR_OK;
}
__global__ void selftest_logf_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = tdr_libm::logf_t<true>(x[i]);
}
extern "C" int tdr_k_selftest_logf(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out || n < 0) return fail(TDR_ERR_ARG, "selftest_logf: bad arguments");
  if (n == 0) return TDR_OK;
  hipLaunchKernelGGL(selftest_logf_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, out);
  LAUNCH_CHECK("selftest_logf");
  return TDR_OK;
}

// ---- tdr_rng_pipe: the generator of ONE filter on the device, drawing AHEAD ----------------------------------------------
// A filter's step draws in a fixed order — the normals of propagate, then the uniform of the resample, then the next
// step's normals — and none of it depends on the particles.  So when a propagate call has been served, the pipe at once
// draws, on a stream of its own and from a COPY of the state, what the step after it will most likely ask for: the uniform,
// then the normals of a propagate call with the same particle count and freeze flag.  It runs beside the scoring launch;
// the next calls find their values ready.  Nothing is assumed: a call that asks for something else (another count, another
// order, the host taking the stream back) makes the pipe drop what it drew ahead and continue from the state the stream
// really is in — the states behind each speculative step are kept for exactly that.  Results are the stream's own either
// way (tests/test_rng.py replays call sequences against the host engine).
// (device-to-device moves of a state are kernels, not runtime copies: a runtime copy between event waits of two streams
// cost the caller's stream ~0.3 ms per step whenever it also did host-to-device copies of its own)
// err (optional): a word of pinned HOST memory the pipe looks at before every call — a state whose error word
// (state[MT_N + 1]: mt_advance_kernel ran out of its attempt budget) is set raises it, so a filter that never hands its
// stream back to the host still hears about it at its next call instead of continuing with stale normals
__global__ void mt_copy_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int n, uint32_t* __restrict__ err) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) {
    const uint32_t v = src[k];
    if (dst) dst[k] = v;
    if (err && k == MT_N + 1 && v != 0) *err = v;
  }
}
static int mt_copy(const void* src, void* dst, int words, hipStream_t s, uint32_t* err = nullptr) {
  hipLaunchKernelGGL(mt_copy_kernel, dim3((unsigned)cdiv(words, 256)), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, words, err);
  LAUNCH_CHECK("mt_copy");
  return TDR_OK;
}
struct tdr_rng_pipe {
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_uniform = nullptr, ev_all = nullptr;
  uint32_t* state = nullptr;        // the stream's real state (device, TDR_RNG_STATE_WORDS)
  uint32_t* spec_state = nullptr;   // working state of the draw-ahead; behind it: [1] snapshot after the uniform
  float* shift = nullptr;           // [0] served value, [16] drawn ahead
  float* z[2] = {nullptr, nullptr};
  void* ws[2] = {nullptr, nullptr};
  int64_t n_max = 0;
  int cur = 0;                      // z[cur] / ws[cur]: what a call on the caller's stream uses; the other: draw-ahead
  bool on_device = false;
  // the draw-ahead in flight: 0 none, 1 uniform not yet taken, 2 uniform taken
  int spec = 0;
  int64_t spec_n = 0, spec_lo = 0, spec_hi = 0;
  int spec_freeze = 0;
  int misses = 0;                   // consecutive calls the draw-ahead did not fit: it pauses after two
  uint32_t* err_host = nullptr;     // pinned, device-visible: raised by the device when a call ran out of its attempt budget
};
static int pipe_check(const tdr_rng_pipe* p, const char* who) {
  if (p->err_host && *(volatile const uint32_t*)p->err_host)
    return fail(TDR_ERR_HIP, "%s: an earlier call of the device generator ran out of its attempt budget — the stream is "
                             "stopped where it stood (probability ~1e-23 per call: more likely a damaged state)", who);
  return TDR_OK;
}
extern "C" int tdr_rng_pipe_create(int64_t n_max, tdr_rng_pipe** out) {
  if (!out || n_max < 1) return fail(TDR_ERR_ARG, "rng_pipe_create: bad arguments");
  *out = nullptr;
  tdr_rng_pipe* p = new (std::nothrow) tdr_rng_pipe;
  if (!p) return fail(TDR_ERR_NOMEM, "rng_pipe_create: out of memory");
  p->n_max = n_max;
  const size_t wsb = tdr_rng_dev_workspace_bytes(n_max);
  hipError_t e = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_uniform, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_all, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&p->state, sizeof(uint32_t) * TDR_RNG_STATE_WORDS);
  if (e == hipSuccess) e = hipMalloc((void**)&p->spec_state, sizeof(uint32_t) * TDR_RNG_STATE_WORDS * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&p->shift, sizeof(float) * 32);
  if (e == hipSuccess) e = hipHostMalloc((void**)&p->err_host, sizeof(uint32_t), hipHostMallocMapped);
  if (e == hipSuccess) *p->err_host = 0;
  for (int k = 0; k < 2 && e == hipSuccess; k++) {
    e = hipMalloc((void**)&p->z[k], sizeof(float) * 4 * (size_t)n_max);
    if (e == hipSuccess) e = hipMalloc(&p->ws[k], wsb);
  }
  if (e != hipSuccess) {
    tdr_rng_pipe_destroy(p);
    return fail(TDR_ERR_HIP, "rng_pipe_create: %s", hipGetErrorString(e));
  }
  *out = p;
  return TDR_OK;
}
extern "C" void tdr_rng_pipe_destroy(tdr_rng_pipe* p) {
  if (!p) return;
  if (p->side) { (void)hipStreamSynchronize(p->side); (void)hipStreamDestroy(p->side); }
  for (hipEvent_t ev : {p->ev_fork, p->ev_uniform, p->ev_all})
    if (ev) (void)hipEventDestroy(ev);
  (void)hipFree(p->state); (void)hipFree(p->spec_state); (void)hipFree(p->shift);
  if (p->err_host) (void)hipHostFree(p->err_host);
  for (int k = 0; k < 2; k++) { (void)hipFree(p->z[k]); (void)hipFree(p->ws[k]); }
  delete p;
}
extern "C" int tdr_rng_pipe_on_device(const tdr_rng_pipe* p) { return p && p->on_device ? 1 : 0; }
// what was drawn ahead is dropped; `s` continues from the state the stream really is in
static int pipe_drop(tdr_rng_pipe* p, hipStream_t s) {
  if (!p->spec) return TDR_OK;
  HIP_TRY(hipStreamWaitEvent(s, p->ev_all, 0));   // (its kernels still use the other workspace and the snapshot)
  if (p->spec == 2)   // the uniform was handed out: the stream stands behind it
    if (int rc = mt_copy(p->spec_state + TDR_RNG_STATE_WORDS, p->state, TDR_RNG_STATE_WORDS, s)) return rc;
  p->spec = 0;
  return TDR_OK;
}
static int pipe_draw_ahead(tdr_rng_pipe* p, int64_t n, int64_t lo, int64_t hi, int freeze, hipStream_t s) {
  if (p->misses >= 2) return TDR_OK;   // the caller's pattern is not the step's: wait until it is again
  HIP_TRY(hipEventRecord(p->ev_fork, s));
  HIP_TRY(hipStreamWaitEvent(p->side, p->ev_fork, 0));
  if (int rc = mt_copy(p->state, p->spec_state, TDR_RNG_STATE_WORDS, p->side)) return rc;
  if (int rc = tdr_k_rng_uniform(p->spec_state, p->shift + 16, p->side)) return rc;
  if (int rc = mt_copy(p->spec_state, p->spec_state + TDR_RNG_STATE_WORDS, TDR_RNG_STATE_WORDS, p->side)) return rc;
  HIP_TRY(hipEventRecord(p->ev_uniform, p->side));
  if (int rc = tdr_k_rng_propagate_normals(p->spec_state, n, lo, hi, freeze, p->z[1 - p->cur], p->ws[1 - p->cur], p->side)) return rc;
  HIP_TRY(hipEventRecord(p->ev_all, p->side));
  p->spec = 1;
  p->spec_n = n; p->spec_lo = lo; p->spec_hi = hi; p->spec_freeze = freeze;
  return TDR_OK;
}
// the stream continues on the device from where the host engine stands (synchronises: the state travels through the host)
extern "C" int tdr_rng_pipe_from_host(tdr_rng_pipe* p, void* host_rng, void* stream) {
  if (!p || !host_rng) return fail(TDR_ERR_ARG, "rng_pipe_from_host: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (int rc = pipe_drop(p, s)) return rc;
  uint32_t words[TDR_RNG_STATE_WORDS];
  if (int rc = tdr_rng_get_state_host(host_rng, words)) return rc;
  HIP_TRY(hipMemcpyAsync(p->state, words, sizeof(words), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  p->on_device = true;
  p->misses = 0;
  if (p->err_host) *p->err_host = 0;   // a fresh state from the host engine
  return TDR_OK;
}
// ... and back on the host engine (synchronises)
extern "C" int tdr_rng_pipe_to_host(tdr_rng_pipe* p, void* host_rng, void* stream) {
  if (!p || !host_rng) return fail(TDR_ERR_ARG, "rng_pipe_to_host: null pointer");
  if (!p->on_device) return TDR_OK;
  hipStream_t s = (hipStream_t)stream;
  if (int rc = pipe_drop(p, s)) return rc;
  uint32_t words[TDR_RNG_STATE_WORDS];
  HIP_TRY(hipMemcpyAsync(words, p->state, sizeof(words), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (int rc = tdr_rng_set_state_host(host_rng, words)) return rc;
  p->on_device = false;
  return TDR_OK;
}
// The normals of a propagate call (tdr_k_rng_propagate_normals): *z4_out = device array [hi - lo][4], valid until the next
// call of this function.  Ordered on `stream`.
extern "C" int tdr_rng_pipe_normals(tdr_rng_pipe* p, int64_t n, int64_t lo, int64_t hi, int scale_freeze, const float** z4_out,
                                    void* stream) {
  if (!p || !z4_out) return fail(TDR_ERR_ARG, "rng_pipe_normals: null pointer");
  if (!p->on_device) return fail(TDR_ERR_ARG, "rng_pipe_normals: the stream is on the host (tdr_rng_pipe_from_host)");
  if (n < 1 || n > p->n_max || lo < 0 || hi > n || lo > hi) return fail(TDR_ERR_ARG, "rng_pipe_normals: bad range");
  if (int rc = pipe_check(p, "rng_pipe_normals")) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int fr = scale_freeze ? 1 : 0;
  if (p->spec == 2 && p->spec_n == n && p->spec_lo == lo && p->spec_hi == hi && p->spec_freeze == fr) {
    // drawn ahead: adopt its values and the state behind them
    HIP_TRY(hipStreamWaitEvent(s, p->ev_all, 0));
    if (int rc = mt_copy(p->spec_state, p->state, TDR_RNG_STATE_WORDS, s, p->err_host)) return rc;
    p->cur = 1 - p->cur;
    p->spec = 0;
    p->misses = 0;
  } else {
    if (p->spec) p->misses++;
    else if (p->misses) p->misses--;   // (a call that had nothing to compare with: the pattern gets another chance)
    if (int rc = pipe_drop(p, s)) return rc;
    if (int rc = tdr_k_rng_propagate_normals(p->state, n, lo, hi, fr, p->z[p->cur], p->ws[p->cur], s)) return rc;
    if (int rc = mt_copy(p->state, nullptr, TDR_RNG_STATE_WORDS, s, p->err_host)) return rc;   // (only looks at the error word)
  }
  *z4_out = p->z[p->cur];
  return pipe_draw_ahead(p, n, lo, hi, fr, s);
}
// The uniform draw of the resample: *shift_out = device float, valid until the next call of this function.
extern "C" int tdr_rng_pipe_uniform(tdr_rng_pipe* p, const float** shift_out, void* stream) {
  if (!p || !shift_out) return fail(TDR_ERR_ARG, "rng_pipe_uniform: null pointer");
  if (!p->on_device) return fail(TDR_ERR_ARG, "rng_pipe_uniform: the stream is on the host (tdr_rng_pipe_from_host)");
  if (int rc = pipe_check(p, "rng_pipe_uniform")) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (p->spec == 1) {   // drawn ahead
    HIP_TRY(hipStreamWaitEvent(s, p->ev_uniform, 0));
    if (int rc = mt_copy(p->shift + 16, p->shift, 1, s)) return rc;
    p->spec = 2;
  } else {
    if (p->spec) p->misses++;
    if (int rc = pipe_drop(p, s)) return rc;
    if (int rc = tdr_k_rng_uniform(p->state, p->shift, s)) return rc;
  }
  *shift_out = p->shift;
  return TDR_OK;
}
