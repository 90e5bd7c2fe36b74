// tdr_score_su.hip — the SHIFT-UNIFORM polar scoring kernel and its ordering passes (host interface: tdr_score_su.h).
//
// score_polar_kernel (tdr_score.hip) pairs window row i with scan row (i + shift) mod nb, shift = the lane's own heading
// bin (state_particle.cpp:124-128): the scan record is a per-lane operand (LDS), and every sample pays the full record
// decode and every class's FMA although a LiDAR scan is sparse — in the config-2 scan 44 % of the (theta, r) bins are
// empty in every class and a non-empty bin holds ~1.0 class (90 % of the class-bins are zero).
//
// Here the particles are processed in (shift, Morton) order with every shift bucket padded to whole waves, so the shift
// is WAVE-uniform: the scan side of a sample is a scalar operand (a four-dword descriptor read through the scalar cache —
// no LDS scan image) and "this bin is empty" / "this bin holds class c only" are wave-uniform branches:
//   every sample        coordinates + the `known` bit from the KNOWN MASK (tdr_cmap.hip), which the workgroup stages in LDS
//                       for the part of the map the windows of its 256 particles cover in the current sector of directions
//   one class present   + ONE dword of the compact record (4-byte gather), one dictionary decode, 2 FMAs (class, norm)
//   several classes     the packed scan record through scalar loads, one decode + FMA per class present
// An empty bin — 44 % of the samples — touches no map record at all; a 64-lane gather is what the L1 address path prices
// highest (>= 16 cycles per CU, tools/ta_cost.hip).  Skipping an FMA whose scan operand is zero leaves the accumulator
// unchanged bit for bit (fma(0, m, acc) == acc for finite m, acc never -0), and the samples of a particle are visited in
// the same order — direction ascending, ring ascending within the group — with the same partition into per-group partial
// sums, so both kernels produce IDENTICAL partial sums (tests/test_shift_uniform.py).  Non-finite dictionary or scan
// values (0 * inf = NaN must not be skipped) turn the skipping off bin by bin (descriptor code SU_CODE_FULL_ALL).
//
// Per launch (all on the caller's stream, nothing synchronises):
//   su_key_kernel      heading bin of every particle (in the caller's locality order) + histogram of the bins
//   rocPRIM            stable radix sort by bin: (shift, Morton) order
//   su_offsets_kernel  bucket starts in the sorted list and in the padded slot list, number of slots in use
//   su_scatter_kernel  slot -> particle (-1 = padding)
//   su_prep_kernel     per (direction, ring): sample offset and scan descriptor, group-major (a wave streams them in order)
//   su_bbox_kernel     bounding box of the sample offsets of every (ring group, sector of directions)
//   score_polar_su_kernel, then score_finalize_kernel over the slots
//
// Compiled with -mllvm -structurizecfg-skip-uniform-regions (build.py): the per-sample dispatch on the descriptor is a
// tree of wave-uniform branches; left to the structuriser each leaf is followed by copies of all accumulators (phi
// merges of its flow blocks: 128 v_mov_b64 in the loop), with uniform regions skipped the leaves are 3-4 instructions.
// Memory operations: the hand-scheduled inner loop (tdr_score_su_asm.h, generated and statically checked by
// tools/gen_su_asm.py) issues its loads and the waits for them inside ONE assembly text with a planned register file — the
// compiler sees a single statement with declared outputs and clobbers.  Everything else in this file loads through plain
// C++ (the compiler tracks the destination registers and places the waits): rounds 3-4 issued the C++ steps' loads through
// separate inline-assembly statements with hand-counted s_waitcnt, which twice let the compiler move a copy between a load
// and its wait ("fragile") — that pattern is gone.
#include <rocprim/device/device_radix_sort.hpp>

#include <atomic>

#include "tdr_score_dev.h"
#include "tdr_score_su.h"
#include "tdr_score_su_asm.h"

#define SU_CODE_FULL 0xFFu       // several classes present: packed scan record, classes with a zero count skipped
#define SU_CODE_FULL_ALL 0xFEu   // a non-finite value is in play: every class multiplied like score_polar_kernel does
#define SU_CODE_PAD 0xFDu        // no such ring: the last group of an image whose ring count is no multiple of the group
#define SU_NSECT 8               // sectors of directions per known-mask staging (16: 2 % slower on config 2)
#define SU_BOX_WORDS 4096        // LDS words of the staged known mask: 16 KB (config 2's dense share 2.97 ms at 12 KB, 2.81 at 16
                                 // and at 20 KB, 4.18 at 24 KB, where a sixth wave per SIMD no longer fits; a wave that
                                 // stages a box of its own — the far-apart waves of config 5 — has a quarter of it)

struct SuArgs {
  const uint32_t* crec;    // compact records (narrow form)
  const uint32_t* kmask;   // the map's known mask (behind the tiles of crec)
  int kcolw;               // words of one of its tile columns (32 kmask_trows)
  unsigned kmask_off;      // its byte offset from crec
  const uint32_t* dict_int;   // the dictionary as integers: value * 2^q (tdr_cmap.hip)
  int dict_n, ctiles_r;
  int pkcol;               // class planes (tdr_cmap.hip): bytes of a tile column - 16 (plane_offset); their constants come with the descriptors
  const int32_t* inexact;  // device words (int_form_off): this scan / map has no exact integer form, the launch does nothing
  int rows, cols;          // map
  float resolution;
  const float* tab_su;     // [nchunks][nb][group][2]: (tab*scale)*res (USCALE) or tab
  const uint32_t* desc;    // [nchunks][nb][group][4]: scan descriptor of bin (row, ring), see su_prep_kernel
  const float* bbox;       // [nchunks][SU_NSECT][4]: min / max of tab_su's two coordinates over the sector
  const float* scan_pk;    // [nr][nb][rf]: read for bins holding several classes
  int nb, nr;
  float res;
  const float* st;
  int64_t cap;
  const int32_t* slots;    // padded (shift, Morton) order, -1 = padding; every 64-slot batch holds one shift
  const int32_t* nslots;   // device word: slots in use (a multiple of 64)
  int group, nchunks, ncls;
  int64_t npad;            // stride of `part`
  uint32_t* part;          // [nchunks][2 ncls + 2][npad]: class k's integer sum as {low, high} words, normalisation, known count
  uint32_t* stats;         // NULL, or (profiling) counters of the variants the wave-sectors ran: tdr_profile_variants
};

// Scan descriptor of a bin, four dwords:
//   [0] code: 0 = every class zero; c + 1 = class c alone is non-zero; SU_CODE_FULL = several classes;
//       SU_CODE_FULL_ALL = a non-finite dictionary / scan value: no skipping in this bin
//   [1] the bin's count summed over the classes, as an integer — for a single class: its count
//   [2] a single class: the constant of plane_offset for the class's PLANE (byte offset from crec: pbase + c * plane_bytes) —
//       the 2-byte cell of the one class is what such a bin fetches: tiles of 8 x 8 cells, a third of the lines the 4 x 4-cell
//       record tiles cost a wave whose particles lie a few cells apart (the gathers' lines bound this kernel: DESIGN.md
//       5.1); several classes: the constant of the record offset (cmap_offset)
//   [3] bit 31, on the first bin of a step (4 consecutive rings) only: one of the step's bins is SU_CODE_FULL / SU_CODE_FULL_ALL
//   (the steps that read RECORDS — the C++ steps, for bins with several classes and their neighbours, and the far path —
//    work the record constants of a single class out of its code: su_rec_const / single_class)
__global__ __launch_bounds__(256) void su_prep_kernel(const float* __restrict__ tab, const float* __restrict__ scan_pk,
                                                      int nb, int nr, int rf, int ncls, int ckconst, unsigned pbase,
                                                      unsigned plane_bytes, int group, int nchunks,
                                                      const float* __restrict__ dict, int dict_n, float* __restrict__ tab_su,
                                                      uint32_t* __restrict__ desc) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;   // the dictionary is small: every workgroup checks it for itself
  for (int k = threadIdx.x; k < dict_n; k += blockDim.x) bad |= !(fabsf(dict[k]) <= 3.402823466e+38f);
  const bool dict_bad = __syncthreads_or(bad);
  const int64_t total = (int64_t)nchunks * nb * group;
  const bool live = t < total;
  const int64_t tt = live ? t : 0;
  const int jj = (int)(tt % group);
  const int64_t q = tt / group;
  const int i = (int)(q % nb), chunk = (int)(q / nb);
  const int j = chunk * group + jj;
  float tx = 0.f, ty = 0.f, val = 0.f;
  uint32_t code = SU_CODE_PAD, ckc = (uint32_t)ckconst, sh = 0;
  if (live && j >= nr) {   // no such ring: the offset of the direction's last real one (inside every box the real ones span)
    const int64_t k = (int64_t)(nr - 1) * nb + i;
    tx = tab[2 * k];
    ty = tab[2 * k + 1];
  }
  if (live && j < nr) {
    code = 0;
    const int64_t k = (int64_t)j * nb + i;
    tx = tab[2 * k];
    ty = tab[2 * k + 1];
    const float* r = scan_pk + k * rf;
    int nz = 0, first = 0;
    bool finite = true;
    for (int c = 0; c < ncls; c++) {
      finite &= fabsf(r[c]) <= 3.402823466e+38f;
      if (r[c] != 0.f) {
        if (!nz) first = c;
        nz++;
      }
    }
    // (a non-finite or fractional count: the launch runs in its float form instead — ray_prep_kernel raises `inexact`)
    if (dict_bad || !finite) { code = SU_CODE_FULL_ALL; val = r[rf - 1]; }
    else if (nz == 1) { code = (uint32_t)first + 1u; val = r[first]; ckc = pbase + (uint32_t)first * plane_bytes; }
    else if (nz > 1) { code = SU_CODE_FULL; val = r[rf - 1]; }
  }
  // steps are 4 consecutive bins (group is a multiple of 4, so they are 4 consecutive threads of a wave)
  uint32_t anyfull = code >= SU_CODE_FULL_ALL ? 1u : 0u;
  anyfull |= __shfl_xor(anyfull, 1);
  anyfull |= __shfl_xor(anyfull, 2);
  if (!live) return;
  tab_su[2 * t] = tx;
  tab_su[2 * t + 1] = ty;
  desc[4 * t] = code;
  desc[4 * t + 1] = (uint32_t)val;
  desc[4 * t + 2] = ckc;
  desc[4 * t + 3] = sh | (((jj & 3) == 0 && anyfull) ? 0x80000000u : 0u);
}

// bounding box of the sample offsets of ring group blockIdx.x, sector blockIdx.y (directions [sect nb / NSECT, ...))
__global__ __launch_bounds__(256) void su_bbox_kernel(const float* __restrict__ tab_su, int nb, int nr, int group,
                                                      float* __restrict__ bbox) {
  const int chunk = blockIdx.x, sect = blockIdx.y;
  const int i0 = (int)((int64_t)sect * nb / SU_NSECT), i1 = (int)((int64_t)(sect + 1) * nb / SU_NSECT);
  const int gn = min(nr - chunk * group, group);
  float lo0 = 3.402823466e+38f, hi0 = -3.402823466e+38f, lo1 = lo0, hi1 = hi0;
  const int cnt = (i1 - i0) * gn;
  for (int t = threadIdx.x; t < cnt; t += 256) {
    const int i = i0 + t / gn, jj = t % gn;
    const float* e = tab_su + (((int64_t)chunk * nb + i) * group + jj) * 2;
    lo0 = fminf(lo0, e[0]); hi0 = fmaxf(hi0, e[0]);
    lo1 = fminf(lo1, e[1]); hi1 = fmaxf(hi1, e[1]);
  }
  __shared__ float red[4][256];
  red[0][threadIdx.x] = lo0; red[1][threadIdx.x] = hi0; red[2][threadIdx.x] = lo1; red[3][threadIdx.x] = hi1;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) {
      red[0][threadIdx.x] = fminf(red[0][threadIdx.x], red[0][threadIdx.x + d]);
      red[1][threadIdx.x] = fmaxf(red[1][threadIdx.x], red[1][threadIdx.x + d]);
      red[2][threadIdx.x] = fminf(red[2][threadIdx.x], red[2][threadIdx.x + d]);
      red[3][threadIdx.x] = fmaxf(red[3][threadIdx.x], red[3][threadIdx.x + d]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) bbox[((int64_t)chunk * SU_NSECT + sect) * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// Sort key of every particle (in the caller's locality order): its heading bin when its neighbourhood is DENSE, nb when
// it is SPARSE — the 64 particles around it in the locality (Morton) order are more than `span` map cells apart.  A
// workgroup of the shift-uniform kernel stages the known mask of everything its 256 particles' windows cover: dense
// particles sorted together keep that box small and share the cache lines of their record gathers.  Sparse ones share
// nothing whatever the order and are bound by the memory system, not by instruction issue: they keep their locality order
// and go through score_polar_kernel (tdr_score.hip), behind the dense ones in the same slot list.
// Histogram of the nb + 1 keys: per workgroup in LDS first (a converged filter fills a few bins).
__global__ __launch_bounds__(256) void su_key_kernel(const float* __restrict__ st, int64_t cap, int64_t n,
                                                     const int32_t* __restrict__ perm, int nb, float span,
                                                     uint32_t* __restrict__ keys, int32_t* __restrict__ vals,
                                                     int* __restrict__ cnt) {
  extern __shared__ int hist[];
  for (int k = threadIdx.x; k <= nb; k += 256) hist[k] = 0;
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    const int32_t p = perm ? perm[t] : (int32_t)t;
    uint32_t key = (uint32_t)rot_shift_dev(st[TDR_ST_THETA * cap + p], nb);
    if (span > 0.f) {
      auto centre = [&](int64_t q, float& x, float& y) {
        const float sc = st[TDR_ST_SCALE * cap + q];
        x = st[TDR_ST_DX * cap + q] * sc + st[TDR_ST_INIT_X * cap + q];
        y = st[TDR_ST_DY * cap + q] * sc + st[TDR_ST_INIT_Y * cap + q];
      };
      const int64_t ta = t >= 32 ? t - 32 : 0, tb = t + 32 < n ? t + 32 : n - 1;
      float x0, y0, x1, y1;
      centre(perm ? perm[ta] : ta, x0, y0);
      centre(perm ? perm[tb] : tb, x1, y1);
      if (!(fabsf(x1 - x0) <= span && fabsf(y1 - y0) <= span)) key = (uint32_t)nb;   // (NaN positions: sparse)
    }
    keys[t] = key;
    vals[t] = p;
    atomicAdd(&hist[key], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= nb; k += 256)
    if (hist[k]) atomicAdd(&cnt[k], hist[k]);
}

// One workgroup.  The particles of key k start at start[k] in the sorted list and take slots [slot_start[k], + count) of
// the slot list, the count of a heading bin (k < nkeys - 1) rounded up to whole waves.
// counts = {slots of the heading bins (a multiple of 64), sparse particles behind them, both together}
__global__ __launch_bounds__(256) void su_offsets_kernel(const int* __restrict__ cnt, int nkeys, int* __restrict__ start,
                                                         int* __restrict__ slot_start, int* __restrict__ counts) {
  __shared__ int sa[256], sb[256];
  int carry_a = 0, carry_b = 0;
  for (int base = 0; base < nkeys; base += 256) {
    const int k = base + threadIdx.x;
    const int c = k < nkeys ? cnt[k] : 0;
    const int cp = k < nkeys - 1 ? (c + 63) & ~63 : c;
    sa[threadIdx.x] = c;
    sb[threadIdx.x] = cp;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {   // Hillis-Steele inclusive scan
      const int va = threadIdx.x >= d ? sa[threadIdx.x - d] : 0;
      const int vb = threadIdx.x >= d ? sb[threadIdx.x - d] : 0;
      __syncthreads();
      sa[threadIdx.x] += va;
      sb[threadIdx.x] += vb;
      __syncthreads();
    }
    if (k < nkeys) {
      start[k] = carry_a + sa[threadIdx.x] - c;
      slot_start[k] = carry_b + sb[threadIdx.x] - cp;
    }
    carry_a += sa[255];
    carry_b += sb[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int sparse = cnt[nkeys - 1];
    counts[0] = carry_b - sparse;
    counts[1] = sparse;
    counts[2] = carry_b;
  }
}

__global__ __launch_bounds__(256) void su_scatter_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                                         int64_t n, const int* __restrict__ start,
                                                         const int* __restrict__ slot_start, int32_t* __restrict__ slots) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const uint32_t k = keys[t];
  slots[slot_start[k] + ((int)t - start[k])] = vals[t];
}

// ---- re-routing by the wave's own box ---------------------------------------------------------------------------------------
// su_key_kernel calls a particle dense when its 64 neighbours in the ALL-heading locality order lie close together.  The
// shift-uniform kernel's waves, though, hold 64 neighbours of ONE heading bin: where the headings are many and the cloud is
// wide (config 5: 8 clusters x 40 headings, 780 particles per cluster and heading over ~25 000 cells) those lie tens of cells
// apart, the wave's windows do not fit its quarter of the mask staging area, and the wave takes the kernel's far path — one
// gather per SAMPLE, a quarter of config 5's wave-sectors in round 4.  Such a wave is what the ray-mapped kernel is for: this
// pass, behind the heading-bin order, moves every wave whose own particles spread over more than `wave_span` cells to the
// scattered share (three small kernels; the sums are exact integers: whichever kernel scores a particle, the bits agree).
__global__ __launch_bounds__(256) void su_wave_far_kernel(const float* __restrict__ st, int64_t cap, const int32_t* __restrict__ slots,
                                                          const int32_t* __restrict__ counts, float wave_span,
                                                          int32_t* __restrict__ keep, int32_t* __restrict__ moved) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), base = w * 64;
  if (base >= (int64_t)counts[0]) return;
  const int32_t p = slots[base + lane];
  float x0 = 3.0e38f, x1 = -3.0e38f, y0 = 3.0e38f, y1 = -3.0e38f;
  if (p >= 0) {
    const float sc = st[TDR_ST_SCALE * cap + p];
    const float x = st[TDR_ST_DX * cap + p] * sc + st[TDR_ST_INIT_X * cap + p];
    const float y = st[TDR_ST_DY * cap + p] * sc + st[TDR_ST_INIT_Y * cap + p];
    x0 = x1 = x;
    y0 = y1 = y;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    x0 = fminf(x0, __shfl_xor(x0, d)); x1 = fmaxf(x1, __shfl_xor(x1, d));
    y0 = fminf(y0, __shfl_xor(y0, d)); y1 = fmaxf(y1, __shfl_xor(y1, d));
  }
  const int nvalid = __popcll(__ballot(p >= 0));
  // (a NaN centre makes the comparison false: the wave stays — the kernel's own checks deal with it)
  const bool far = (x1 - x0 > wave_span) || (y1 - y0 > wave_span);
  if (lane == 0) {
    keep[w] = far ? 0 : 1;
    moved[w] = far ? nvalid : 0;
  }
}
// One workgroup: exclusive sums of the kept waves (x 64 slots) and of the moved particles; the new counts
__global__ __launch_bounds__(256) void su_compact_offsets_kernel(int32_t* __restrict__ keep, int32_t* __restrict__ moved,
                                                                 int32_t* __restrict__ counts, int32_t* __restrict__ old_counts) {
  __shared__ int sa[256], sb[256];
  const int c0 = counts[0], c1 = counts[1], c2 = counts[2];
  const int nw = c0 >> 6;
  int carry_a = 0, carry_b = 0;
  for (int base = 0; base < nw; base += 256) {
    const int w = base + threadIdx.x;
    const int a = w < nw ? keep[w] : 0, b = w < nw ? moved[w] : 0;
    sa[threadIdx.x] = a;
    sb[threadIdx.x] = b;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {   // Hillis-Steele inclusive scan
      const int va = threadIdx.x >= d ? sa[threadIdx.x - d] : 0;
      const int vb = threadIdx.x >= d ? sb[threadIdx.x - d] : 0;
      __syncthreads();
      sa[threadIdx.x] += va;
      sb[threadIdx.x] += vb;
      __syncthreads();
    }
    if (w < nw) {
      // keep[w]: the wave's new first slot, or -1; moved[w]: the rank of its first particle among the moved ones
      keep[w] = a ? (carry_a + sa[threadIdx.x] - a) * 64 : -1;
      moved[w] = carry_b + sb[threadIdx.x] - b;
    }
    carry_a += sa[255];
    carry_b += sb[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    old_counts[0] = c0; old_counts[1] = c1; old_counts[2] = c2;
    counts[0] = carry_a * 64;
    counts[1] = c1 + carry_b;
    counts[2] = carry_a * 64 + c1 + carry_b;
  }
}
// slot t of the old list -> its place in the new one: kept waves close ranks, then the old scattered share, then the moved
__global__ __launch_bounds__(256) void su_compact_scatter_kernel(const int32_t* __restrict__ slots, const int32_t* __restrict__ keep,
                                                                 const int32_t* __restrict__ moved,
                                                                 const int32_t* __restrict__ counts,
                                                                 const int32_t* __restrict__ old_counts, int32_t* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int oc0 = old_counts[0], oc1 = old_counts[1];
  if (t >= (int64_t)oc0 + oc1) return;
  const int32_t p = slots[t];
  if (t >= oc0) {   // the old scattered share, in its order
    out[(int64_t)counts[0] + (t - oc0)] = p;
    return;
  }
  const int64_t w = t >> 6;   // (a wave of this kernel is a wave of the list: blocks of 256 slots)
  const int32_t k = keep[w];
  const uint64_t valid = __ballot(p >= 0);
  if (k >= 0) {
    out[(int64_t)k + (t & 63)] = p;
  } else if (p >= 0) {
    const int rank = __popcll(valid & (((uint64_t)1 << (t & 63)) - 1));
    out[(int64_t)counts[0] + oc1 + moved[w] + rank] = p;
  }
}
__global__ void su_copy_slots_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ counts, int32_t* __restrict__ dst) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (int64_t)counts[2]) dst[t] = src[t];
}

// LDS of the scoring kernel, ONE object so that the dictionary sits at LDS address 0 (the assembly loop reads it there)
struct SuLds {
  uint32_t dict[TDR_CMAP_MAX_DICT];   // integer dictionary
  uint32_t bits[SU_BOX_WORDS];   // the staged known mask: rows rlo..rhi of words wlo..whi of the map's mask
  int box[4];
};

// lane = particle; every wave holds particles of ONE heading bin (see the file comment).  grid.y = group of a.group
// consecutive range rings (score_group_rings: a multiple of 4, nr a multiple of 4), samples visited ray-major like
// score_polar_kernel: direction i ascending, the group's rings in steps of 4 consecutive cells along the ray.  The
// directions are walked in SU_NSECT sectors; for each the workgroup stages the known mask of the cells its windows can reach.
template <int NV4, bool KSLOT, bool USCALE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void score_polar_su_kernel(SuArgs a) {
  constexpr int RF = 4 * NV4;
  constexpr int ND = CmapShape<RF, KSLOT>::ND, CW = CmapShape<RF, KSLOT>::CW;
  constexpr bool ASM_LOOP = CW == 2 && ND == 6;   // tdr_score_su_asm.h: two-dword records
  __shared__ SuLds lds;
  if (int_form_off(a.inexact)) return;   // (uniform) no integer form of this scan / map: score_polar_kernel does the launch in floats
  for (int t = threadIdx.x; t < a.dict_n; t += 256) lds.dict[t] = a.dict_int[t];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform
  const int64_t nsl = (int64_t)__builtin_amdgcn_readfirstlane(*a.nslots);
  if ((int64_t)blockIdx.x * 256 >= nsl) return;   // the whole workgroup is beyond the slots in use (uniform)
  const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64;
  const bool active = base < nsl;                 // wave-uniform; an idle wave still keeps the barriers below
  const int32_t sp = active ? a.slots[base + lane] : -1;
  const int32_t p0 = a.slots[active ? base : (int64_t)blockIdx.x * 256];   // a batch's first slot is never padding
  const int64_t p = sp >= 0 ? sp : p0;
  const float scale = a.st[TDR_ST_SCALE * a.cap + p];
  const float cx = a.st[TDR_ST_DX * a.cap + p] * scale + a.st[TDR_ST_INIT_X * a.cap + p];  // state_particle.cpp:161
  const float cy = a.st[TDR_ST_DY * a.cap + p] * scale + a.st[TDR_ST_INIT_Y * a.cap + p];  // :162
  const float off0 = cy / a.resolution;  // top_down_map_polar.cpp:29
  const float off1 = cx / a.resolution;  // :30
  const int shift = __builtin_amdgcn_readfirstlane(rot_shift_dev(a.st[TDR_ST_THETA * a.cap + p], a.nb));
  const int nb = a.nb, G = a.group;
  const int j0 = blockIdx.y * G, gn = min(a.nr - j0, G);
  const float rmaxf = (float)a.rows, cmaxf = (float)a.cols;
  const int ckcol = a.ctiles_r * 128 - 16 * CW;   // cmap_offset
  const int pkcol = a.pkcol;                      // plane_offset; its constant comes with the descriptor
  // the record constant of a bin for the steps that read records: dword (c / 3) of class c = code - 1 alone (c < 11: c / 3
  // == c * 11 >> 5), dword 0 otherwise (wave-uniform: scalar arithmetic)
  const uint32_t ckconst = (uint32_t)(a.ctiles_r * 128 + 128);
  auto su_rec_const = [&](uint32_t cd) -> uint32_t {
    return cd != 0 && cd < SU_CODE_PAD ? ckconst + 4u * (((cd - 1u) * 11u) >> 5) : ckconst;
  };
  typedef const float __attribute__((address_space(4))) * tdr_const_f;
  typedef const uint32_t __attribute__((address_space(4))) * tdr_const_u;
  const tdr_const_f tbase = (tdr_const_f)a.tab_su + (int64_t)blockIdx.y * nb * G * 2;
  const tdr_const_u dbase = (tdr_const_u)a.desc + (int64_t)blockIdx.y * nb * G * 4;
  const tdr_const_f bbase = (tdr_const_f)a.bbox + (int64_t)blockIdx.y * SU_NSECT * 4;
  const tdr_const_f scanc = (tdr_const_f)a.scan_pk;
  const uint32_t* __restrict__ crec = a.crec;
  const uint32_t* __restrict__ kmask = a.kmask;
  const unsigned lds_base = (unsigned)(uintptr_t)&lds;                // LDS byte address of the dictionary ...
  const unsigned lbits_lds = (unsigned)(uintptr_t)&lds.bits[0];      // ... and of the staged mask

  typedef float tdr_v2f __attribute__((ext_vector_type(2)));
  const tdr_v2f offv = {off0, off1};
  // plain, compiler-tracked loads: a word of the staged mask at an LDS byte address, bytes of the compact map at a byte offset
  typedef const uint32_t __attribute__((address_space(3))) * tdr_lds_u;
  auto lds_word = [](unsigned byte_addr) -> uint32_t { return *reinterpret_cast<tdr_lds_u>((uintptr_t)byte_addr); };
  const char* __restrict__ crecb = reinterpret_cast<const char*>(a.crec);
  auto map_dword = [&](unsigned off) -> uint32_t { return *reinterpret_cast<const uint32_t*>(crecb + off); };
  auto map_ushort = [&](unsigned off) -> uint32_t { return *reinterpret_cast<const uint16_t*>(crecb + off); };
  const bool weird = !(fabsf(off0) <= 1e9f) || !(fabsf(off1) <= 1e9f) || (!USCALE && !(fabsf(scale * a.res) <= 1e9f));
  auto field = [&](const uint32_t (&w)[CW], int k) -> uint32_t {   // distance k of a compact record (cmap_decode, one field)
    const uint32_t ww = w[k / 3];
    const int sh = 10 * (k % 3);
    const uint32_t boff = sh ? ((ww >> sh) & 0xFFCu) : (ww & 0xFFCu);
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds.dict) + boff);
  };
  auto field1 = [&](uint32_t ww, int k) -> uint32_t {   // ... when the sample loaded only the dword class k lives in
    const int sh = 10 * (k % 3);
    const uint32_t boff = sh ? ((ww >> sh) & 0xFFCu) : (ww & 0xFFCu);
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds.dict) + boff);
  };

  // integer sums (see the file comment: exact, so independent of the order and of the kernel that forms them)
  uint64_t acc[ND];
#pragma unroll
  for (int k = 0; k < ND; k++) acc[k] = 0;
  uint32_t norm = 0;
  uint32_t known = 0;

  // per-class accumulate of a bin that holds class cd - 1 only: a switch over a wave-uniform value
  auto single_class = [&](uint32_t cd, uint32_t v, uint32_t ww) {
    switch (cd) {
#define SU_CASE(K)                                                                                 \
  case K + 1:                                                                                      \
    if constexpr (K < ND) {                                                                        \
      const uint32_t m = field1(ww, K < ND ? K : 0);                                               \
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[K < ND ? K : 0]) : "s"(v), "v"(m) : "vcc"); \
    }                                                                                              \
    break;
      SU_CASE(0) SU_CASE(1) SU_CASE(2) SU_CASE(3) SU_CASE(4) SU_CASE(5)
      SU_CASE(6) SU_CASE(7) SU_CASE(8) SU_CASE(9) SU_CASE(10)
#undef SU_CASE
      default: break;
    }
  };
  // ... when the sample loaded the 2-byte cell of the class's plane (dictionary index * 4 in bits 2..11)
  auto single_class_plane = [&](uint32_t cd, uint32_t v, uint32_t cell) {
    const uint32_t m = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds.dict) + (cell & 0xFFCu));
    switch (cd) {
#define SU_CASE(K)                                                                                 \
  case K + 1:                                                                                      \
    if constexpr (K < ND) {                                                                        \
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[K < ND ? K : 0]) : "s"(v), "v"(m) : "vcc"); \
    }                                                                                              \
    break;
      SU_CASE(0) SU_CASE(1) SU_CASE(2) SU_CASE(3) SU_CASE(4) SU_CASE(5)
      SU_CASE(6) SU_CASE(7) SU_CASE(8) SU_CASE(9) SU_CASE(10)
#undef SU_CASE
      default: break;
    }
  };
  // a bin with several classes (or a non-finite value in play): the whole record against the packed scan record
  auto full_bin = [&](uint32_t cd, uint32_t v, const uint32_t (&wr)[CW], int kbit, int64_t bin) {
    const tdr_const_f S = scanc + bin * RF;
    norm += v & (0u - (uint32_t)kbit);
#pragma unroll
    for (int k = 0; k < ND; k++) {
      const float sk = S[k];
      if (sk != 0.f) acc[k] += (uint64_t)(uint32_t)sk * (uint64_t)field(wr, k);
    }
  };
  // cell of one sample (top_down_map_polar.cpp:28-31)
  auto cell = [&](float tx, float ty, int& ri, int& ci) {
    tdr_v2f pv = {tx, ty};
    if constexpr (!USCALE) pv = (pv * scale) * a.res;  // top_down_map_polar.cpp:28
    pv = pv + offv;                                     // :29-30
    tdr_v2f qv = {__builtin_amdgcn_fmed3f(pv.x, -1.f, rmaxf), __builtin_amdgcn_fmed3f(pv.y, -1.f, cmaxf)};
    qv = qv + 0.49999997f;                              // round_half_away_clamped
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ri) : "v"(qv.x));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ci) : "v"(qv.y));
  };

  // One step with the known mask staged in LDS: the 4 samples (i, j0 + jj .. jj + 3), paired with scan row r.  Known bits
  // come from word ri * krow4 + (ci >> 5) * 4 + kconst of the staged mask.
  auto cpp_step = [&](int i, int r, int jj, int krow4, int kconst) {
    const tdr_const_f T = tbase + ((int64_t)i * G + jj) * 2;
    const tdr_const_u D = dbase + ((int64_t)r * G + jj) * 4;
    uint32_t val[4];
    uint32_t code[4], pad[4];
    uint32_t w[4], bits[4];
    int cis[4];
    unsigned offs[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      code[u] = D[4 * u];
      // A ring the image does not have (SU_CODE_PAD) goes through the step like an EMPTY bin — su_prep gives it the offset of
      // the direction's last real ring, so its mask lookup stays inside the staged box — and is masked out of the known
      // count below.
      pad[u] = code[u] == SU_CODE_PAD ? 0u : 0xFFFFFFFFu;
      code[u] = code[u] == SU_CODE_PAD ? 0u : code[u];
      val[u] = D[4 * u + 1];
      const uint32_t ckc = su_rec_const(code[u]);
      int ri, ci;
      cell(T[2 * u], T[2 * u + 1], ri, ci);
      cis[u] = ci;
      int wa;
      asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(wa) : "v"(ri), "v"(krow4), "s"(kconst));
      const int cw5 = ci >> 5;
      unsigned la;
      asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(la) : "v"(cw5), "v"(wa));
      bits[u] = lds_word(la);
      w[u] = 0;
      offs[u] = 0;
      if (code[u] != 0) {   // wave-uniform: only a non-empty bin needs its record — one dword of it
        int t1, t2;
        const int cq = ci >> 2;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(cq), "v"(ckcol), "s"(ckc));
        asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(t2) : "v"(ci), "n"(CW == 1 ? 2 : (CW == 2 ? 3 : 4)), "v"(t1));
        asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(offs[u]) : "v"(ri), "n"(CW == 1 ? 4 : (CW == 2 ? 5 : 6)), "v"(t2));
        w[u] = map_dword(offs[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t cd = code[u];
      int kmsk;   // 0 / -1: the cell's known bit
      asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(kmsk) : "v"(bits[u]), "v"(cis[u]));
      known -= (uint32_t)kmsk & pad[u];
      if (cd != 0) {   // wave-uniform
        if (cd < SU_CODE_FULL_ALL) {
          // the bin's count x known (state_particle.cpp:141-142)
          norm += (uint32_t)kmsk & val[u];
          single_class(cd, val[u], w[u]);
        } else {
          uint32_t wr[CW];
          wr[0] = w[u];
#pragma unroll
          for (int d = 1; d < CW; d++)
            wr[d] = map_dword(offs[u] + 4u * d);
          full_bin(cd, val[u], wr, kmsk & 1, (int64_t)(j0 + jj + u) * nb + r);
        }
      }
    }
  };
  // The same step when none of its bins holds several classes (the descriptor's flag, wave-uniform): a bin with a class
  // fetches the 2-byte cell of that class's PLANE (8 x 8-cell tiles) instead of a record dword (4 x 4-cell tiles) — a third
  // of the lines for a wave whose particles lie a few cells apart.  The assembly loop does the same.
  auto cpp_step_plane = [&](int i, int r, int jj, int krow4, int kconst) {
    const tdr_const_f T = tbase + ((int64_t)i * G + jj) * 2;
    const tdr_const_u D = dbase + ((int64_t)r * G + jj) * 4;
    uint32_t val[4];
    uint32_t code[4], pad[4];
    uint32_t w[4], bits[4];
    int cis[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      code[u] = D[4 * u];
      pad[u] = code[u] == SU_CODE_PAD ? 0u : 0xFFFFFFFFu;   // (see cpp_step)
      code[u] = code[u] == SU_CODE_PAD ? 0u : code[u];
      val[u] = D[4 * u + 1];
      const uint32_t pkc = D[4 * u + 2];
      int ri, ci;
      cell(T[2 * u], T[2 * u + 1], ri, ci);
      cis[u] = ci;
      int wa;
      asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(wa) : "v"(ri), "v"(krow4), "s"(kconst));
      const int cw5 = ci >> 5;
      unsigned la;
      asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(la) : "v"(cw5), "v"(wa));
      bits[u] = lds_word(la);
      w[u] = 0;
      if (code[u] != 0) {   // wave-uniform
        int t1, t2;
        unsigned off;
        const int cq = ci >> 3;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(cq), "v"(pkcol), "s"(pkc));
        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(t2) : "v"(ci), "v"(t1));
        asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(off) : "v"(ri), "v"(t2));
        w[u] = map_ushort(off);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int kmsk;   // 0 / -1: the cell's known bit
      asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(kmsk) : "v"(bits[u]), "v"(cis[u]));
      known -= (uint32_t)kmsk & pad[u];
      if (code[u] != 0) {   // wave-uniform
        norm += (uint32_t)kmsk & val[u];   // the bin's count x known (state_particle.cpp:141-142)
        single_class_plane(code[u], val[u], w[u]);
      }
    }
  };
  // NS consecutive steps of a sector (step k = direction i0 + k / spd, rings (k % spd) * 4 ..) WITHOUT the staged mask:
  // the windows of the workgroup's particles are too far apart to stage (scattered particles: su_key_kernel gives them
  // waves of their own).  Every gather of such a wave misses the caches, so the wave is bound by how many it keeps in
  // flight: the gathers of all 4 NS samples are requested before the first is used — ONE per sample: the mask word of an
  // empty bin (the global mask: 32 x 32-cell tiles, a cache line each), the record dword of a bin with a class (its bit 0 is
  // the known bit: every dword of a compact record carries it, tdr_cmap.hip).
  auto far_steps = [&](auto nsteps_c, int i, int jj) {
    constexpr int NS = decltype(nsteps_c)::value;
    uint32_t w[4 * NS];
    uint32_t cbits[(4 * NS + 5) / 6];   // the column's low 5 bits of every sample (the bit of its mask word), six per word
#pragma unroll
    for (int q = 0; q < (4 * NS + 5) / 6; q++) cbits[q] = 0;
    const int mtrb = a.kcolw * 4, mconst = (int)a.kmask_off + a.kcolw * 4 + 128;   // kmask_offset
    int ii = i, jx = jj;
#pragma unroll
    for (int d = 0; d < NS; d++) {
      int r = ii + shift;
      r -= r >= nb ? nb : 0;
      const tdr_const_f T = tbase + ((int64_t)ii * G + jx) * 2;
      const tdr_const_u D = dbase + ((int64_t)r * G + jx) * 4;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        int ri, ci;
        cell(T[2 * u], T[2 * u + 1], ri, ci);
        unsigned off;
        const int sidx = 4 * d + u;   // (d, u are unrolled: constants after unrolling)
        if (D[4 * u] == 0 || D[4 * u] == SU_CODE_PAD) {   // wave-uniform: an empty bin (or no ring at all) reads the cell's mask word
          off = kmask_offset(ri, ci, mtrb, mconst);
          cbits[sidx / 6] |= (uint32_t)(ci & 31) << (5 * (sidx % 6));
        } else {               // a single class: the dword it lives in; several classes: dword 0
          const uint32_t ckc = su_rec_const(D[4 * u]);
          int t1, t2;
          const int cq = ci >> 2;
          asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(t1) : "v"(cq), "v"(ckcol), "s"(ckc));
          asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(t2) : "v"(ci), "n"(CW == 1 ? 2 : (CW == 2 ? 3 : 4)), "v"(t1));
          asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(off) : "v"(ri), "n"(CW == 1 ? 4 : (CW == 2 ? 5 : 6)), "v"(t2));
        }
        w[4 * d + u] = map_dword(off);   // (all 4 NS requests are issued before the first value is used below)
      }
      jx += 4;
      if (jx >= ((gn + 3) & ~3)) { jx = 0; ii++; }
    }
    ii = i; jx = jj;
#pragma unroll
    for (int d = 0; d < NS; d++) {
      int r = ii + shift;
      r -= r >= nb ? nb : 0;
      const tdr_const_u D = dbase + ((int64_t)r * G + jx) * 4;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t ww = w[4 * d + u];
        const uint32_t cdr = D[4 * u];
        const uint32_t cd = cdr == SU_CODE_PAD ? 0u : cdr;   // a ring the image does not have: an empty bin that counts nothing
        const int sidx = 4 * d + u;
        const uint32_t kb = cd == 0 ? (ww >> ((cbits[sidx / 6] >> (5 * (sidx % 6))) & 31u)) & 1u : (ww & 1u);
        known += cdr == SU_CODE_PAD ? 0u : kb;
        if (cd != 0) {   // wave-uniform
          const uint32_t v = D[4 * u + 1];
          if (cd < SU_CODE_FULL_ALL) {
            norm += (0u - kb) & v;
            single_class(cd, v, ww);
          } else {
            // the other dwords of the record: its offset again (rare: ~1 % of the bins hold several classes)
            // (behind a barrier the optimiser cannot see through: it would keep the coordinates of all 4 NS samples
            // alive from the request pass for the sake of this branch — 8 NS registers, a wave per SIMD less)
            int iq = ii;
            asm volatile("" : "+s"(iq));
            const tdr_const_f T = tbase + ((int64_t)iq * G + jx) * 2;
            int ri, ci;
            cell(T[2 * u], T[2 * u + 1], ri, ci);
            const unsigned off = cmap_offset<CW, (CW == 1 ? 3 : (CW == 2 ? 2 : 1))>(ri, ci, ckcol, a.ctiles_r * 128 + 128);
            uint32_t wr[CW];
            wr[0] = ww;
#pragma unroll
            for (int q = 1; q < CW; q++)
              wr[q] = map_dword(off + 4u * q);
            full_bin(cd, v, wr, (int)kb, (int64_t)(j0 + jx + u) * nb + r);
          }
        }
      }
      jx += 4;
      if (jx >= ((gn + 3) & ~3)) { jx = 0; ii++; }
    }
  };
  constexpr int FAR_DEPTH = 4;   // steps (of 4 samples) a wave of the memory-bound path keeps in flight
  auto far_sector = [&](int i0, int i1) {
    const int gn4 = (gn + 3) & ~3;   // steps cover whole fours of rings; what lies behind the last ring is SU_CODE_PAD
    const int spd = gn4 >> 2, total = (i1 - i0) * spd;
    int i = i0, jj = 0, k = 0;
    auto advance = [&](int steps) {
      jj += 4 * steps;
      while (jj >= gn4) { jj -= gn4; i++; }
    };
    for (; k + FAR_DEPTH <= total; k += FAR_DEPTH) {
      far_steps(std::integral_constant<int, FAR_DEPTH>{}, i, jj);
      advance(FAR_DEPTH);
    }
    for (; k < total; k++) {
      far_steps(std::integral_constant<int, 1>{}, i, jj);
      advance(1);
    }
  };
  // One sector of directions [i0, i1) with the known mask staged in LDS
  // (inside: every cell the workgroup's windows can reach in this sector lies inside the map — the clamp into the guard
  // ring is the identity and the loop without it runs; allknown: every staged mask word is all ones — every sample is a
  // known cell and the loop without clamp and without mask lookups runs)
  auto run_sector = [&](int i0, int i1, int krow4, int kconst, bool inside, bool allknown) {
    if constexpr (ASM_LOOP) {
      if (gn == G && lds_base == 0 && (G == 4 || G == 8 || G == 16)) {
        // the steps of the sector as one stream: step k reads T at byte k * 32 from its start and D at byte k * 64 from the
        // start of scan row (i0 + shift) mod nb, wrapping to row 0; the loop hands a step that holds a bin with several
        // classes back (nleft >= 0 on exit), cpp_step does that one, and the loop goes on behind it
        const int spd = G / 4;                                // steps per direction
        int r0 = i0 + shift;
        r0 -= r0 >= nb ? nb : 0;
        // (readfirstlane: values the compiler cannot prove wave-uniform must not reach an "s" operand)
        uint32_t toff = __builtin_amdgcn_readfirstlane((uint32_t)i0 * (uint32_t)G * 8u);
        uint32_t doff = __builtin_amdgcn_readfirstlane((uint32_t)r0 * (uint32_t)G * 16u);
        uint32_t nleft = __builtin_amdgcn_readfirstlane((uint32_t)((i1 - i0) * spd - 1));
        uint32_t wleft = __builtin_amdgcn_readfirstlane((uint32_t)((nb - r0) * spd - 1));
        const uint32_t wrapm1 = __builtin_amdgcn_readfirstlane((uint32_t)(nb * spd - 1));
        const int kconst_s = __builtin_amdgcn_readfirstlane(kconst);
        const uint64_t half2 = 0x3EFFFFFF3EFFFFFFull;         // {0.49999997f, 0.49999997f}
        // int -> float is a vector instruction: bring the map's limits back to scalar registers
        const float rmax_s = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rmaxf)));
        const float cmax_s = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(cmaxf)));
        const tdr_v2f scale2 = {scale, scale};
        const uint64_t res2 = (uint64_t)__float_as_uint(a.res) * 0x100000001ull;   // {res, res} in an SGPR pair
        while (nleft != 0xFFFFFFFFu) {
#define SU_ASM_OPERANDS                                                                                               \
          /* the accumulators (64-bit) are TIED to v[32:43]: the loop reaches them by VGPR-relative indexing */                \
          : [a0] "+{v[32:33]}"(acc[0]), [a1] "+{v[34:35]}"(acc[1]), [a2] "+{v[36:37]}"(acc[2]), [a3] "+{v[38:39]}"(acc[3]),    \
            [a4] "+{v[40:41]}"(acc[4]), [a5] "+{v[42:43]}"(acc[ND > 5 ? 5 : 0]), [norm] "+v"(norm), [known] "+v"(known),       \
            [toff] "+v"(toff), [doff] "+v"(doff),                                                                             \
            [nleft] "+v"(nleft), [wleft] "+v"(wleft)                                                                   \
          : [offv] "v"(offv), [krow4] "v"(krow4), [pkcol] "v"(pkcol), [tb] "s"(tbase), [db] "s"(dbase),              \
            [rmax] "s"(rmax_s), [cmax] "s"(cmax_s), [half] "s"(half2), [kconst] "s"(kconst_s), [crec] "s"(crec),           \
            [wrapm1] "s"(wrapm1), [scale2] "v"(scale2), [res2] "s"(res2)                                               \
          : SU_ASM_CLOBBERS
          if (allknown) {   // wave-uniform
            if constexpr (USCALE) asm volatile(SU_ASM_US_ALLKNOWN SU_ASM_OPERANDS);
            else asm volatile(SU_ASM_PS_ALLKNOWN SU_ASM_OPERANDS);
          } else if (inside) {
            if constexpr (USCALE) asm volatile(SU_ASM_US_NOCLAMP SU_ASM_OPERANDS);
            else asm volatile(SU_ASM_PS_NOCLAMP SU_ASM_OPERANDS);
          } else {
            if constexpr (USCALE) asm volatile(SU_ASM_US SU_ASM_OPERANDS);
            else asm volatile(SU_ASM_PS SU_ASM_OPERANDS);
          }
#undef SU_ASM_OPERANDS
          // (the compiler takes the outputs of an asm statement for divergent)
          toff = __builtin_amdgcn_readfirstlane(toff); doff = __builtin_amdgcn_readfirstlane(doff);
          nleft = __builtin_amdgcn_readfirstlane(nleft); wleft = __builtin_amdgcn_readfirstlane(wleft);
          if (nleft == 0xFFFFFFFFu) break;
          // the step the loop stopped in front of
          // (no division here: it would run on the vector unit and drag the loop's scalar state there with it)
          const int k = (i1 - i0) * spd - 1 - (int)nleft, lsp = G == 4 ? 0 : (G == 8 ? 1 : 2);
          const int i = i0 + (k >> lsp), jj = (k & (spd - 1)) * 4;
          int r = i + shift;
          r -= r >= nb ? nb : 0;
          cpp_step(i, r, jj, krow4, kconst);
          toff += 32u;
          doff += 64u;
          if (wleft == 0) { doff = 0; wleft = wrapm1; } else wleft--;
          nleft--;   // 0 -> 0xFFFFFFFF: the sector is done
          toff = __builtin_amdgcn_readfirstlane(toff); doff = __builtin_amdgcn_readfirstlane(doff);
          nleft = __builtin_amdgcn_readfirstlane(nleft); wleft = __builtin_amdgcn_readfirstlane(wleft);
        }
        return;
      }
    }
    for (int i = i0; i < i1; i++) {
      int r = i + shift;
      r -= r >= nb ? nb : 0;
      for (int jj = 0; jj < gn; jj += 4) {
        // (the flag sits on the step's first bin: su_prep_kernel)
        if (dbase[((int64_t)r * G + jj) * 4 + 3] >> 31) cpp_step(i, r, jj, krow4, kconst);
        else cpp_step_plane(i, r, jj, krow4, kconst);
      }
    }
  };

  for (int sect = 0; sect < SU_NSECT; sect++) {
    const int i0 = (int)((int64_t)sect * nb / SU_NSECT), i1 = (int)((int64_t)(sect + 1) * nb / SU_NSECT);
    // cells this lane's samples of the sector can fall on: rounding is monotone, so the box of the offsets carries over
    float a0 = bbase[4 * sect], b0 = bbase[4 * sect + 1], a1 = bbase[4 * sect + 2], b1 = bbase[4 * sect + 3];
    if constexpr (!USCALE) {
      const float x0 = (a0 * scale) * a.res, y0 = (b0 * scale) * a.res, x1 = (a1 * scale) * a.res, y1 = (b1 * scale) * a.res;
      a0 = fminf(x0, y0); b0 = fmaxf(x0, y0); a1 = fminf(x1, y1); b1 = fmaxf(x1, y1);
    }
    int rl = (int)fminf(fmaxf(floorf(a0 + off0) - 1.f, -1.f), rmaxf), rh = (int)fminf(fmaxf(ceilf(b0 + off0) + 1.f, -1.f), rmaxf);
    int cl = (int)fminf(fmaxf(floorf(a1 + off1) - 1.f, -1.f), cmaxf), ch = (int)fminf(fmaxf(ceilf(b1 + off1) + 1.f, -1.f), cmaxf);
    if (weird) { rl = -1; rh = a.rows; cl = -1; ch = a.cols; }
    if (!active) { rl = 0x7FFFFFFF; rh = -0x7FFFFFFF; cl = 0x7FFFFFFF; ch = -0x7FFFFFFF; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      rl = min(rl, __shfl_xor(rl, d)); rh = max(rh, __shfl_xor(rh, d));
      cl = min(cl, __shfl_xor(cl, d)); ch = max(ch, __shfl_xor(ch, d));
    }
    __syncthreads();   // the previous sector's lookups are done (and, the first time, the dictionary is staged)
    if (threadIdx.x == 0) { lds.box[0] = 0x7FFFFFFF; lds.box[1] = -0x7FFFFFFF; lds.box[2] = 0x7FFFFFFF; lds.box[3] = -0x7FFFFFFF; }
    __syncthreads();
    if (lane == 0 && active) {
      atomicMin(&lds.box[0], rl); atomicMax(&lds.box[1], rh); atomicMin(&lds.box[2], cl); atomicMax(&lds.box[3], ch);
    }
    __syncthreads();
    const int rlo = __builtin_amdgcn_readfirstlane(lds.box[0]), rhi = __builtin_amdgcn_readfirstlane(lds.box[1]);   // mask rows
    const int wlo = (__builtin_amdgcn_readfirstlane(lds.box[2]) >> 5) + 1;                                            // mask words
    const int whi = (__builtin_amdgcn_readfirstlane(lds.box[3]) >> 5) + 1;
    const int H = rhi - rlo + 1, Wb = whi - wlo + 1;
    const bool fits = (int64_t)H * Wb <= SU_BOX_WORDS;   // uniform over the workgroup
    uint32_t every = 0xFFFFFFFFu;   // AND of the words this thread staged
    if (fits) {
      const int total = H * Wb;
      // (row fastest: consecutive threads read consecutive words of one tile column of the mask, kmask_offset)
      for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int wc = idx / H, row = idx - wc * H;
        const uint32_t wv = kmask[(int64_t)(wlo + wc) * a.kcolw + (rlo + row + 32)];
        lds.bits[row * Wb + wc] = wv;
        every &= wv;
      }
    }
    const bool allknown = __syncthreads_and(fits && every == 0xFFFFFFFFu);   // (cells outside the map are unknown)
    if (active) {
      const bool inside = rlo >= 0 && rhi < a.rows && __builtin_amdgcn_readfirstlane(lds.box[2]) >= 0 &&
                          __builtin_amdgcn_readfirstlane(lds.box[3]) < a.cols;
      // one call of the sector's loop (two would double the assembly text in the kernel), its box chosen here
      int box_krow4 = Wb * 4, box_kconst = (int)lbits_lds + (1 - wlo - rlo * Wb) * 4;
      int box_ok = fits ? 1 : 0, box_inside = inside ? 1 : 0, box_allknown = allknown ? 1 : 0;
      if (!fits) {
        // The workgroup's windows do not fit one box — its waves belong to different clusters or heading bins (config 5:
        // 8 clusters x 40 headings, more than half of the sectors).  A wave's own 64 particles still lie together: a box
        // per wave in a quarter of the staging area each (rl .. ch: the wave's own bounds after the shuffles above;
        // nothing here crosses waves, so no workgroup barrier: a wave's LDS operations execute in order).
        const int wrl = __builtin_amdgcn_readfirstlane(rl), wrh = __builtin_amdgcn_readfirstlane(rh);
        const int wcl = __builtin_amdgcn_readfirstlane(cl), wch = __builtin_amdgcn_readfirstlane(ch);
        const int wl = (wcl >> 5) + 1, wh = (wch >> 5) + 1;
        const int Hw = wrh - wrl + 1, Wbw = wh - wl + 1;
        if ((int64_t)Hw * Wbw <= SU_BOX_WORDS / 4) {
          uint32_t* const mine = lds.bits + wave * (SU_BOX_WORDS / 4);
          uint32_t ev = 0xFFFFFFFFu;
          const int total = Hw * Wbw;
          for (int idx = lane; idx < total; idx += 64) {
            const int wc = idx / Hw, row = idx - wc * Hw;
            const uint32_t wv = kmask[(int64_t)(wl + wc) * a.kcolw + (wrl + row + 32)];
            mine[row * Wbw + wc] = wv;
            ev &= wv;
          }
          // the wave reads what its lanes just wrote — through the assembly loop's ds_read as well: order the stores in front
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          box_ok = 1;
          box_krow4 = Wbw * 4;
          box_kconst = (int)lbits_lds + wave * (SU_BOX_WORDS / 4) * 4 + (1 - wl - wrl * Wbw) * 4;
          box_inside = (wrl >= 0 && wrh < a.rows && wcl >= 0 && wch < a.cols) ? 1 : 0;
          box_allknown = __all(ev == 0xFFFFFFFFu) ? 1 : 0;
        }
      }
      // (readfirstlane: the loop's variants are chosen by these, and the compiler must see them wave-uniform)
      box_ok = __builtin_amdgcn_readfirstlane(box_ok);
      box_inside = __builtin_amdgcn_readfirstlane(box_inside);
      box_allknown = __builtin_amdgcn_readfirstlane(box_allknown);
      box_krow4 = __builtin_amdgcn_readfirstlane(box_krow4);
      box_kconst = __builtin_amdgcn_readfirstlane(box_kconst);
      if (a.stats && lane == 0)   // [0..2] the workgroup's box: all known / inside / general; [3..5] the wave's own; [6] far
        atomicAdd(&a.stats[!box_ok ? 6 : ((fits ? 0 : 3) + (box_allknown ? 0 : (box_inside ? 1 : 2)))], 1u);
      if (box_ok) run_sector(i0, i1, box_krow4, box_kconst, box_inside != 0, box_allknown != 0);
      else far_sector(i0, i1);
    }
  }
  if (active) {
    const int64_t slot = base + lane;
    uint32_t* o = a.part + (int64_t)blockIdx.y * (2 * a.ncls + 2) * a.npad + slot;
#pragma unroll
    for (int k = 0; k < ND; k++)
      if (k < a.ncls) {
        o[(int64_t)(2 * k) * a.npad] = (uint32_t)acc[k];
        o[(int64_t)(2 * k + 1) * a.npad] = (uint32_t)(acc[k] >> 32);
      }
    o[(int64_t)(2 * a.ncls) * a.npad] = norm;
    o[(int64_t)(2 * a.ncls + 1) * a.npad] = known;
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// 0 = never, 1 = when it pays (default), 2 = whenever the shapes allow (tests: small filters, heavy padding)
static int g_su_mode = 1;
extern "C" int tdr_config_shift_uniform(int mode) {   // < 0: query only
  if (mode >= 0) g_su_mode = mode > 2 ? 2 : mode;
  return g_su_mode;
}
extern "C" size_t tdr_cmap_tile_words(int ncls, int rows, int cols);   // tdr_cmap.hip
extern "C" size_t tdr_cmap_plane_offset_words(int ncls, int rows, int cols);
extern "C" size_t tdr_cmap_plane_words(int ncls, int rows, int cols);
// map cells the 64 locality neighbours of a "dense" particle may span: fixed (the config call), or — the default — tuned
// while running, starting from SU_SPAN_START
#define SU_SPAN_START 16.f
static bool g_su_span_fixed = false;
static float g_su_span = SU_SPAN_START;
extern "C" float tdr_config_shift_uniform_span(float cells) {   // >= 0: fix it (0: every particle counts as dense);
  if (cells >= 0.f) { g_su_span = cells; g_su_span_fixed = true; }   // -1: query only; below -1.5: back to tuning
  else if (cells < -1.5f) { g_su_span = SU_SPAN_START; g_su_span_fixed = false; }
  return g_su_span;
}
// Which span is fastest depends on the particle set (how far the same-heading neighbours of a moderately dense particle
// lie apart): measured on MI355X, config 2 wants 8 (7.00 against 7.29 ms at 24), config 5 wants 16 (15.9 against 17.7 at 8),
// a cluster with a single heading wants 24 or more (4.6 against 6.9 ms at 8).
namespace {
// (measured on MI355X with the ray-mapped kernel taking the scattered share, ms per scoring call at 8 / 16 / 24 / 40 cells:
// config 2's mix 5.88 / 5.55 / 5.51 / 5.67, uniform particles 10.7 / 11.0 / - / 11.7, a converged filter 3.1 / 3.05 / - / 3.0)
// (round 4, later: with the table's factors the ray-mapped kernel scores 100 000 uniform particles in 7.6 ms, the mixed
// launch at 8 cells in 8.4 — hence a candidate that sends all but the densest cores to it;
// profiles/r04_time_int_form_c2_v2.txt)
constexpr float kSpanCand[] = {2.f, 8.f, 16.f, 24.f, 40.f};
constexpr int kSpanCands = (int)(sizeof(kSpanCand) / sizeof(kSpanCand[0]));
constexpr int kSpanTrials = 2;        // timed calls per candidate: the faster one counts (a single call is noisy)
constexpr int kSpanSkip = 30;         // calls of a new shape before the first trial (first-use allocations, cold caches, and
                                      // a short run — a benchmark of a few dozen steps — is not worth ten trial calls)
constexpr int kSpanRetune = 4000;     // launches between two trials
}  // namespace
float tdr_su_span_begin(SpanTuner* t, int64_t shape, hipStream_t s) {
  if (g_su_span_fixed || !t) return g_su_span;
  if (!t->e0 && (hipEventCreate(&t->e0) != hipSuccess || hipEventCreate(&t->e1) != hipSuccess)) return g_su_span;
  if (shape != t->shape) { t->shape = shape; t->phase = -kSpanSkip; t->trial = 0; t->best_ms = 3.0e38f; t->pending = false; t->best = t->round_best = g_su_span; }
  if (t->pending) {   // the candidate timed by an earlier launch — if its events are not through yet, ask again next time
    if (hipEventQuery(t->e1) != hipSuccess) return t->best;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t->e0, t->e1) == hipSuccess && ms > 0.f && ms < t->best_ms) {
      t->best_ms = ms;
      t->round_best = kSpanCand[t->phase];
    }
    t->pending = false;
    if (++t->trial >= kSpanTrials) { t->trial = 0; t->phase++; }
    // Only a finished round changes the span in use.  (A caller that runs ahead of the device — a benchmark loop — makes
    // hundreds of calls while one trial's events are pending: they must not run at the first candidate's span just because
    // it is the only one timed so far.  Seen with the candidate of 2 cells in front: 5.3 -> 5.95 ms per config-2 step.)
    if (t->phase >= kSpanCands) { t->settled_launches = 0; t->best = t->round_best; }
  }
  if (t->phase < 0) { t->phase++; return t->best; }
  if (t->phase >= kSpanCands) {
    if (++t->settled_launches < kSpanRetune) return t->best;
    t->phase = 0;   // try them again: the particle set changes as the filter converges
    t->best_ms = 3.0e38f;
    t->round_best = t->best;
  }
  t->open = hipEventRecord(t->e0, s) == hipSuccess;
  t->trial_calls++;
  return kSpanCand[t->phase];
}
void tdr_su_span_end(SpanTuner* t, hipStream_t s) {
  if (g_su_span_fixed || !t || !t->open) return;
  t->open = false;
  t->pending = hipEventRecord(t->e1, s) == hipSuccess;
}
// cells a wave's own 64 particles may spread over before the wave is re-routed to the ray-mapped kernel (0: never — the
// default: MEASURED, AND IT DOES NOT PAY.  MI355X, ms per step at 0 / 24 / 32 / 40 / 56 cells: config 5 14.40 / 16.16 / 15.19 /
// 14.87 / 14.31, config 2 4.67 / 6.58 / 5.69 / 5.67 / 5.34 — a particle on the shift-uniform kernel's far path costs ~68 ns, on
// the ray-mapped kernel ~80 ns: the waves this moves are cheaper where they are.  DESIGN.md 5.1.)
static float g_su_wave_span = 0.f;
extern "C" int tdr_config_su_wave_span(int cells) {   // < 0: query only
  if (cells >= 0) g_su_wave_span = (float)cells;
  return (int)g_su_wave_span;
}
float tdr_su_wave_span() { return g_su_wave_span; }
static int g_su_lds_pad = 0;   // EXPERIMENT: dynamic LDS bytes per workgroup (limits the waves per SIMD without touching the code)
extern "C" int tdr_config_su_lds_pad(int bytes) {
  if (bytes >= 0) g_su_lds_pad = bytes;
  return g_su_lds_pad;
}
static std::atomic<int64_t> g_su_launches{0};   // diagnostics only
extern "C" int64_t tdr_shift_uniform_launches(void) { return g_su_launches.load(); }
// Padding costs up to 63 idle lanes per heading bin: the order pays once a bin holds a few waves on average.
bool tdr_su_shape_ok(int nb, int nr, int group, int64_t n_total) {
  if (g_su_mode == 0) return false;
  if (group % 4 != 0 || nb > 4095) return false;
  if (g_su_mode == 2) return true;   // tests: small filters and small windows too
  // A small window does not pay for the per-sector set-up of the shift-uniform kernel (bounding box, mask staging, three
  // barriers) nor for a wave per particle: at the reference node's own 100 x 25 bins and 20 000 particles the integer form
  // takes 0.23 ms (0.18 + 0.10, side by side) where the float kernel takes 0.11 (profiles/r04_bench_ref_integer_form_v1.json).
  if ((int64_t)nb * nr < 8192) return false;
  return n_total >= (int64_t)64 * nb;
}
static size_t su_sort_tmp_bytes(int64_t n) {
  size_t bytes = 0;
  uint32_t* k = nullptr;
  int32_t* v = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)std::max<int64_t>(n, 1), 0u, 12u,
                                           (hipStream_t)0, false);
  if (e != hipSuccess || bytes == 0) bytes = (size_t)(n + 4096) * 16;   // no device to ask: a generous bound
  return bytes;
}
SuWs tdr_su_ws(int nb, int nr, int group, int64_t n) {
  SuWs w;
  const int64_t nchunks = cdiv(nr, group), nbins = nchunks * nb * group;
  int64_t o = 0;
  auto take = [&](int64_t words) { const int64_t at = o; o += (words + 63) / 64 * 64; return at; };   // 256-byte aligned
  w.tab_su = take(2 * nbins);
  w.desc = take(4 * nbins);
  w.bbox = take(nchunks * SU_NSECT * 4);
  w.keys_in = take(n);
  w.keys_out = take(n);
  w.vals_in = take(n);
  w.vals_out = take(n);
  w.ints = take(3 * ((int64_t)nb + 1) + TDR_SU_TAIL_INTS);   // [cnt][start][slot_start] nb + 1 each, [counts 3][n_multi][inexact][mass bound][table is not its factors]
  w.slots = take(su_npad(n, nb));
  w.slots2 = take(su_npad(n, nb));
  w.wave_tmp = take(2 * (su_npad(n, nb) / 64 + 1));
  w.sort_tmp = take((int64_t)((su_sort_tmp_bytes(n) + 3) / 4));
  const int64_t T = tdr_ray_padded_samples(nb, nr);
  w.ray_tab = take(2 * T);                     // tdr_score_ray.hip: sample offsets and 16-bit scan descriptors in ray order,
  w.ray_desc = take((T + 1) / 2);              // the list of bins that hold several classes
  w.ray_multi = take((int64_t)nb * nr);
  w.ray_rad = take(T / nb);                    // the rings' radii in ray order (a table given as factors)
  w.total = o;
  return w;
}

// The ordering passes alone: the slot list — dense particles by heading bin (one bin when L.nb == 1: the Cartesian score has
// no heading bins), every bin padded to whole waves (-1), then the sparse particles in the caller's order — and the counts.
int tdr_su_order(const SuLaunch& L, const SuWs& W, hipStream_t s, const int32_t** slots_out, const int32_t** counts_out) {
  int32_t* base = L.ws;
  uint32_t* keys_in = reinterpret_cast<uint32_t*>(base + W.keys_in);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(base + W.keys_out);
  int32_t* vals_in = base + W.vals_in;
  int32_t* vals_out = base + W.vals_out;
  int* cnt = base + W.ints;
  const int nkeys = L.nb + 1;
  int* start = cnt + nkeys;
  int* slot_start = start + nkeys;
  int* counts = slot_start + nkeys;
  int32_t* slots = base + W.slots;
  const int64_t n = L.n;
  HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)(3 * nkeys + TDR_SU_TAIL_INTS), s));
  HIP_TRY(hipMemsetAsync(slots, 0xFF, sizeof(int32_t) * (size_t)L.npad, s));
  hipLaunchKernelGGL(su_key_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), sizeof(int) * (size_t)nkeys, s, L.st, L.cap, n,
                     L.perm, L.nb, L.span, keys_in, vals_in, cnt);
  LAUNCH_CHECK("su_key");
  hipLaunchKernelGGL(su_offsets_kernel, dim3(1), dim3(256), 0, s, (const int*)cnt, nkeys, start, slot_start, counts);
  LAUNCH_CHECK("su_offsets");
  unsigned bits = 1;
  while ((1u << bits) < (unsigned)nkeys) bits++;
  size_t tmp_bytes = su_sort_tmp_bytes(n);
  HIP_TRY(rocprim::radix_sort_pairs(base + W.sort_tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, bits,
                                    s, false));
  hipLaunchKernelGGL(su_scatter_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, (const uint32_t*)keys_out,
                     (const int32_t*)vals_out, n, (const int*)start, (const int*)slot_start, slots);
  LAUNCH_CHECK("su_scatter");
  if (L.wave_span > 0.f && L.nb > 1) {   // waves whose own particles lie far apart: to the scattered share after all
    int32_t* slots2 = base + W.slots2;
    int32_t* keep = base + W.wave_tmp;
    int32_t* moved = keep + (L.npad / 64 + 1);
    int32_t* old_counts = counts + 7;   // (ints: TDR_SU_TAIL_INTS)
    const int64_t nwaves = L.npad / 64;
    hipLaunchKernelGGL(su_wave_far_kernel, dim3((unsigned)cdiv(nwaves, 4)), dim3(256), 0, s, L.st, L.cap, (const int32_t*)slots,
                       (const int32_t*)counts, L.wave_span, keep, moved);
    hipLaunchKernelGGL(su_compact_offsets_kernel, dim3(1), dim3(256), 0, s, keep, moved, counts, old_counts);
    hipLaunchKernelGGL(su_compact_scatter_kernel, dim3((unsigned)cdiv(L.npad, 256)), dim3(256), 0, s, (const int32_t*)slots,
                       (const int32_t*)keep, (const int32_t*)moved, (const int32_t*)counts, (const int32_t*)old_counts, slots2);
    // padding slots of kept waves travel with them; nothing reads behind counts[2]
    hipLaunchKernelGGL(su_copy_slots_kernel, dim3((unsigned)cdiv(L.npad, 256)), dim3(256), 0, s, (const int32_t*)slots2,
                       (const int32_t*)counts, slots);
    LAUNCH_CHECK("su_reroute");
  }
  *slots_out = slots;
  *counts_out = counts;
  return TDR_OK;
}
int tdr_su_prepare(const SuLaunch& L, const SuWs& W, hipStream_t s, const int32_t** slots_out, const int32_t** counts_out) {
  if (int rc = tdr_su_order(L, W, s, slots_out, counts_out)) return rc;
  int32_t* base = L.ws;
  float* tab_su = reinterpret_cast<float*>(base + W.tab_su);
  uint32_t* desc = reinterpret_cast<uint32_t*>(base + W.desc);
  const int64_t ndesc = (int64_t)L.nchunks * L.nb * L.group;
  const int lc = L.map->cwords == 1 ? 3 : (L.map->cwords == 2 ? 2 : 1);
  const int ckconst = ((L.map->rows >> lc) + 2) * 128 + 128;   // cmap_offset (tdr_score_dev.h)
  // plane_offset's constant for class 0's plane, as a byte offset from crec (tdr_score_ray.hip uses the same)
  const unsigned plane_bytes = (unsigned)(tdr_cmap_plane_words(L.map->ncls, L.map->rows, L.map->cols) * 4);
  const unsigned pbase = (unsigned)(tdr_cmap_plane_offset_words(L.map->ncls, L.map->rows, L.map->cols) * 4) +
                         (unsigned)(plane_trows(L.map->rows) * 128) + 128u;
  hipLaunchKernelGGL(su_prep_kernel, dim3((unsigned)cdiv(ndesc, 256)), dim3(256), 0, s, L.tab, L.scan_pk, L.nb, L.nr, L.rf,
                     L.map->ncls, ckconst, pbase, plane_bytes, L.group, L.nchunks, L.map->dict, L.map->dict_n, tab_su, desc);
  LAUNCH_CHECK("su_prep");
  hipLaunchKernelGGL(su_bbox_kernel, dim3((unsigned)L.nchunks, SU_NSECT), dim3(256), 0, s, (const float*)tab_su, L.nb, L.nr,
                     L.group, reinterpret_cast<float*>(base + W.bbox));
  LAUNCH_CHECK("su_bbox");
  return TDR_OK;
}

int tdr_su_score(const SuLaunch& L, const SuWs& W, hipStream_t s) {
  const tdr_map_desc* map = L.map;
  int32_t* base = L.ws;
  int* nslots = base + W.ints + 3 * (L.nb + 1);   // counts[0]
  SuArgs u;
  const int lc = map->cwords == 1 ? 3 : (map->cwords == 2 ? 2 : 1);
  u.crec = map->crec; u.dict_int = reinterpret_cast<const uint32_t*>(map->dict) + TDR_CMAP_MAX_DICT;
  u.dict_n = map->dict_n; u.ctiles_r = (map->rows >> lc) + 2;
  u.pkcol = plane_trows(map->rows) * 128 - 16;
  u.inexact = nslots + 4;
  u.rows = map->rows; u.cols = map->cols; u.resolution = map->resolution;
  u.tab_su = reinterpret_cast<const float*>(base + W.tab_su);
  u.desc = reinterpret_cast<const uint32_t*>(base + W.desc);
  u.bbox = reinterpret_cast<const float*>(base + W.bbox);
  u.kmask = map->crec + tdr_cmap_tile_words(map->ncls, map->rows, map->cols);
  u.kcolw = kmask_trows(map->rows) * 32;
  u.kmask_off = (unsigned)(tdr_cmap_tile_words(map->ncls, map->rows, map->cols) * 4);
  u.scan_pk = L.scan_pk;
  u.nb = L.nb; u.nr = L.nr; u.res = L.res; u.st = L.st; u.cap = L.cap;
  u.slots = base + W.slots; u.nslots = nslots;
  u.group = L.group; u.nchunks = L.nchunks; u.ncls = map->ncls; u.npad = L.npad; u.part = reinterpret_cast<uint32_t*>(L.part);
  u.stats = tdr_profile_stats_ptr();
  const dim3 grid((unsigned)cdiv(L.npad, 256), (unsigned)L.nchunks), block(256);
  const bool ks = tdr_has_kslot(map->ncls, L.rf), us = L.uniform_scale;
#define TDR_LAUNCH_SU(NV4)                                                                         \
  if (ks && us) hipLaunchKernelGGL((score_polar_su_kernel<NV4, true, true>), grid, block, g_su_lds_pad, s, u);  \
  else if (ks) hipLaunchKernelGGL((score_polar_su_kernel<NV4, true, false>), grid, block, g_su_lds_pad, s, u);  \
  else if (us) hipLaunchKernelGGL((score_polar_su_kernel<NV4, false, true>), grid, block, g_su_lds_pad, s, u);  \
  else hipLaunchKernelGGL((score_polar_su_kernel<NV4, false, false>), grid, block, g_su_lds_pad, s, u);
  switch (L.rf / 4) {
    case 1: TDR_LAUNCH_SU(1) break;
    case 2: TDR_LAUNCH_SU(2) break;
    case 3: TDR_LAUNCH_SU(3) break;
    default: return fail(TDR_ERR_ARG, "score: no shift-uniform kernel for record size %d", L.rf);
  }
#undef TDR_LAUNCH_SU
  LAUNCH_CHECK("score_polar_su");
  g_su_launches.fetch_add(1);
  return TDR_OK;
}
