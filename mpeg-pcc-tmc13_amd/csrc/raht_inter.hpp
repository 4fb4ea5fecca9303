// raht_inter.hpp -- attribute inter prediction in RAHT (tmc3/RAHT.cpp:849-972, 1165-1198, 1256-1347,
// 1504-1549, 1810-1829): what the level pass without sub-node prediction (raht_tile.hpp) needs on top of
// the intra coder when a reference frame is given.
//
//   the frame's blocks   The reference descends a SECOND tree in lock step, each from its own top, and
//                        matches the block of a parent against the frame's nodes with the same key.  The
//                        frame's points are in Morton order, so that tree is never built here: the node
//                        with key k at bit level lr is the run of points with pos >> lr == k -- two
//                        bisections per child position, weight = length of the run, attribute sum = a
//                        difference of the frame's modular prefix sums (the reference sums in `int` too).
//                        The frame's block is transformed IN ITS OWN WEIGHTS by the same eight lanes and,
//                        filtered by the level's tap, predicts every coefficient of the block
//                        (inter_block below).
//   two candidates       With raht_enable_inter_intra_layer_RDO the encoder codes a level twice -- with
//                        the frame and without -- and keeps the cheaper one by an adaptive rate estimate
//                        (PCCRAHTACCoefficientEntropyEstimate, RAHT.h:71-94): the analyze pass writes both
//                        candidates (coefficients, RDOQ descriptors, transformed predictions), both zero-run
//                        chains are resolved (raht_rdoq.hpp, twice), then the estimate:
//                          rate_pack / rate_p1 / rate_p0_bits   the estimate's probabilities are recurrences
//                                       over ALL coefficients in coding order (p += (2^20 - p) >> 6 or
//                                       p -= p >> 6: the shifts do not compose).  probResGt1 moves only at
//                                       non-zero coefficients: one wavefront walks those; probResGt0 moves at
//                                       every coefficient: threads take chunks and find their starting state
//                                       by running the (monotone) recurrence from its two extreme states
//                                       until they meet; the cost of every coefficient follows (log2 from a
//                                       table the HOST's libm filled -- the comparison of the two sums must
//                                       come out as it does in the reference);
//                          rate_sum     the reference adds the costs into a double in coding order: one
//                                       wavefront per candidate does exactly that;
//                          decide       the cheaper candidate's coefficients, prediction record, zero-run
//                                       state and probabilities go on (inter_commit).
//                        The sum is what a level costs (a dependent double addition per coefficient and
//                        component); everything else is the intra pass twice.
//   estimated taps       (enableFilterEstimation) the level's tap is 128 * crosscorr / autocorr of the
//                        first component's coefficients over every block that lines up (inter_tap_kernel:
//                        integer sums, any order), quantised like a coefficient (inter_tap_finish_kernel).
#pragma once

#include "raht_levels.hpp"

namespace gpcc {

constexpr int kAcRateScaleLog = 20;
constexpr uint32_t kAcRateScale = 1u << kAcRateScaleLog;
constexpr int kAcRateTable = 1 << 20;  // log2 of 1 .. 2^20 (probabilities, and magnitudes the table covers)

// first frame point whose node at bit level `sh` has a key >= k
__device__ __forceinline__ int
inter_lower_bound(const int64_t* __restrict__ pos, int n, int sh, uint64_t k)
{
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (((uint64_t)pos[mid] >> sh) < k)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// The frame's block under the parent with key `pkey`, by the eight lanes of a group (lane t = child
// position t; every lane of the wavefront calls).  *node: the frame has a node there (:1322-1347; the
// caller excludes single-child blocks under the extension).  out[k]: the frame's coefficient at position
// t -- its attribute sums normalised, transformed in the frame's own weights (:1533-1545) and filtered --
// or 0 where the frame's block has none (fwdTransformBlock222 swaps values into place, :680-690, so a
// position without a coefficient ends up holding what an empty position held: 0).
template<int C>
__device__ __forceinline__ void
inter_block(const InterRef& ir, int64_t pkey, int t, bool on, const SharedLut& lut, bool* node, int64_t out[C])
{
  int32_t wr = 0;
  int64_t v[C];
#pragma unroll
  for (int k = 0; k < C; k++)
    v[k] = 0;
  if (on) {
    const uint64_t k0 = ((uint64_t)pkey << 3) + (uint64_t)t;
    const int lo = inter_lower_bound(ir.pos, ir.n_ref, ir.lr, k0);
    const int hi = inter_lower_bound(ir.pos, ir.n_ref, ir.lr, k0 + 1);
    wr = hi - lo;
    if (wr > 0) {
#pragma unroll
      for (int k = 0; k < C; k++)
        v[k] = fp_from_int((int32_t)((uint32_t)ir.prefix[(size_t)hi * C + k] - (uint32_t)ir.prefix[(size_t)lo * C + k]));
    }
  }
  *node = group8_bits(wr > 0) != 0;
  if (wr > 1) {
#pragma unroll
    for (int k = 0; k < C; k++)
      v[k] = scale_rsqrt(v[k], wr, lut);
  }
  int32_t cw = wr;
#pragma unroll
  for (int st = 0; st < 3; st++) {
    const int bit = 1 << st;
    const int32_t pw = lane_xor8(cw, bit);
    const bool left = !(t & bit);
    const int32_t wl = left ? cw : pw, wrr = left ? pw : cw;
    const bool both = wl && wrr;
    const bool swap = !wl && wrr;
    int64_t ca = 0, cb = 0;
    if (both)
      raht_coeffs(wl, wrr, lut, &ca, &cb);
#pragma unroll
    for (int k = 0; k < C; k++) {
      const int64_t own = v[k];
      const int64_t oth = shfl_xor_i64(own, bit);
      if (both)
        v[k] = left ? fp_mul_c(oth, cb) + fp_mul_c(own, ca) : fp_mul_c(own, ca) - fp_mul_c(oth, cb);
      else if (swap)
        v[k] = oth;
    }
    cw = both ? wl + wrr : (left ? wl + wrr : 0);
  }
  const int64_t tap = ir.filtered ? (int64_t)*ir.tap : 128;
#pragma unroll
  for (int k = 0; k < C; k++)
    out[k] = ir.filtered ? (v[k] * tap) >> 7 : v[k];
}

// The same under the integer Haar kernel (HaarKernel, RAHT.cpp:642-668): the frame's node values come from
// its own level arrays (InterRef::hkey ..), there is no normalisation and no filter tap ("the integer Haar
// kernel takes the frame's coefficients as they are", :1520-1526).
template<int C>
__device__ __forceinline__ void
inter_block_haar(const InterRef& ir, int64_t pkey, int t, bool on, bool* node, int64_t out[C])
{
  int32_t wr = 0;
  int64_t v[C];
#pragma unroll
  for (int k = 0; k < C; k++)
    v[k] = 0;
  if (on) {
    const int64_t want = (int64_t)(((uint64_t)pkey << 3) + (uint64_t)t);
    int lo = 0, hi = ir.hsoff[1];
    const int m = hi;
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (ir.hkey[mid] < want)
        lo = mid + 1;
      else
        hi = mid;
    }
    if (lo < m && ir.hkey[lo] == want) {
      wr = ir.hfp[lo + 1] - ir.hfp[lo];
#pragma unroll
      for (int k = 0; k < C; k++)
        v[k] = fp_from_int(ir.hlf[(size_t)lo * C + k]);
    }
  }
  *node = group8_bits(wr > 0) != 0;
  int32_t cw = wr;
#pragma unroll
  for (int st = 0; st < 3; st++) {
    const int bit = 1 << st;
    const int32_t pw = lane_xor8(cw, bit);
    const bool left = !(t & bit);
    const int32_t wl = left ? cw : pw, wrr = left ? pw : cw;
    const bool both = wl && wrr;
    const bool swap = !wl && wrr;
#pragma unroll
    for (int k = 0; k < C; k++) {
      const int64_t own = v[k];
      const int64_t oth = shfl_xor_i64(own, bit);
      if (both) {
        const int64_t hf = left ? oth - own : own - oth;
        v[k] = left ? own + ((hf >> (1 + kFpFrac)) << kFpFrac) : hf;
      } else if (swap) {
        v[k] = oth;
      }
    }
    cw = both ? wl + wrr : (left ? wl + wrr : 0);
  }
#pragma unroll
  for (int k = 0; k < C; k++)
    out[k] = v[k];
}

// ---- the rate estimate of the per-level decision ---------------------------------------------------
// est 0 = the candidate with the frame ("cur"), est 1 = the intra candidate
struct RateState {
  int32_t p0[2][3], p1[2][3];  // the estimates' probabilities before the level
  int32_t q0[2][3], q1[2][3];  // ... after it (rate_p0_bits, rate_p1)
  double bits[2];              // the level's cost per candidate (rate_sum)
  int32_t itz;                 // the intra candidate's zero run behind its last dual level (:1252-1254)
  int32_t intra_wins;          // the level's decision
  int32_t num_modes;           // decisions taken so far
  int32_t pad;
};

struct RateCtx {
  TreeView tv;
  const int32_t* plane[2];  // coefficients of the two candidates (planar, stride n)
  int32_t n;
  int32_t a, b;             // the level's coefficients
  int32_t c;
  int32_t* pb;              // [2 est][C][n] probResGt1 in front of every NON-ZERO coefficient
  unsigned long long* nzw;  // [2 est][C][wstride] flags of the level's coefficients, 64 per word: non-zero ...
  unsigned long long* bigw; // ... magnitude above 1
  int32_t wstride;
  double* term;             // [2 est][n * C] cost of every (coefficient, component) in coding order
  RateState* rs;
  const double* log2tab;    // [kAcRateTable + 1] log2((double)x)
  int32_t* slice_l;         // [1] the zero-run state of the candidate with the frame (raht_rdoq.hpp) ...
  int32_t* islice_l;        // [1] ... and of the intra candidate
  int32_t* islice_pair;     // [2] both entries of the intra candidate's state (the dependency kernels keep one per level parity)
  int32_t* modes;           // [32] attr_layer_code_mode, in order
  // commit
  int32_t* coeffs;
  const int32_t* icoeffs;
  int64_t* ptrans;
  const int64_t* iptrans;
  int32_t rows;             // children of the level (rows of the prediction record)
};

__device__ __forceinline__ unsigned long long
rate_readlane_u64(unsigned long long v, int src)
{
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void
rate_init_kernel(RateState* rs)
{
  if (threadIdx.x == 0) {
    for (int e = 0; e < 2; e++) {
      for (int k = 0; k < 3; k++)
        rs->p0[e][k] = rs->p1[e][k] = rs->q0[e][k] = rs->q1[e][k] = (int32_t)(kAcRateScale >> 1);
      rs->bits[e] = 0.0;
    }
    rs->itz = 0;
    rs->intra_wins = 0;
    rs->num_modes = 0;
    rs->pad = 0;
  }
}

// in front of a dual level: the intra candidate's zero run is what ITS last level left (it is not updated
// by the levels coded once in between, :1618-1669), as an index of the last reset (raht_rdoq.hpp)
__global__ __launch_bounds__(64) void
rate_level_begin_kernel(RateCtx cx)
{
  if (tree_failed(cx.tv))
    return;
  if (threadIdx.x == 0) {
    cx.islice_pair[0] = cx.islice_pair[1] = cx.a - 1 - cx.rs->itz;
    cx.rs->bits[0] = cx.rs->bits[1] = 0.0;
  }
}

// flags of the level's coefficients, 64 per word: non-zero, magnitude above 1
__global__ __launch_bounds__(256) void
rate_pack_kernel(RateCtx cx)
{
  if (tree_failed(cx.tv))
    return;
  const int lane = threadIdx.x & 63;
  const int C = cx.c;
  const int count = cx.b - cx.a;
  const int words = (count + 63) >> 6;
  const int64_t total = (int64_t)2 * C * words;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t x = wave; x < total; x += nwaves) {
    const int est = (int)(x / ((int64_t)C * words));
    const int k = (int)((x / words) % C);
    const int wd = (int)(x % words);
    const int i = wd * 64 + lane;
    const int32_t val = i < count ? cx.plane[est][(size_t)k * cx.n + cx.a + i] : 0;
    const unsigned long long nz = __ballot(val != 0);
    const unsigned long long big = __ballot(val > 1 || val < -1);
    if (lane == 0) {
      cx.nzw[((size_t)est * C + k) * cx.wstride + wd] = nz;
      cx.bigw[((size_t)est * C + k) * cx.wstride + wd] = big;
    }
  }
}

// probResGt1 steps only at the non-zero coefficients (resStatUpdate, RAHT.cpp:80-91): one wavefront per
// (estimate, component) walks the flag words -- 64 words per load, the set bits of a word one after the other
// in scalar registers -- and records the state in front of every non-zero coefficient.  Cost: words + events.
__global__ __launch_bounds__(64) void
rate_p1_kernel(RateCtx cx)
{
  if (tree_failed(cx.tv))
    return;
  const int lane = threadIdx.x & 63;
  const int C = cx.c;
  const int est = (int)blockIdx.x / C;
  const int k = (int)blockIdx.x % C;
  const int count = cx.b - cx.a;
  const int words = (count + 63) >> 6;
  const unsigned long long* __restrict__ nzw = cx.nzw + ((size_t)est * C + k) * cx.wstride;
  const unsigned long long* __restrict__ bigw = cx.bigw + ((size_t)est * C + k) * cx.wstride;
  int32_t* __restrict__ out = cx.pb + ((size_t)est * C + k) * cx.n;
  int p = cx.rs->p1[est][k];
  for (int wb = 0; wb < words; wb += 64) {
    const int wd = wb + lane;
    const unsigned long long mynz = wd < words ? nzw[wd] : 0ull;
    const unsigned long long mybig = wd < words ? bigw[wd] : 0ull;
    unsigned long long busy = __ballot(mynz != 0);
    while (busy) {
      const int j = __ffsll((long long)busy) - 1;
      busy &= busy - 1;
      const unsigned long long nz = rate_readlane_u64(mynz, j);
      const unsigned long long big = rate_readlane_u64(mybig, j);
      unsigned long long m = nz;
      int rec = 0;
      while (m) {
        const int u = __ffsll((long long)m) - 1;
        m &= m - 1;
        rec = lane == u ? p : rec;
        p += ((big >> u) & 1) ? (int)((kAcRateScale - (uint32_t)p) >> 6) : -(p >> 6);
      }
      if ((nz >> lane) & 1)
        out[(size_t)(wb + j) * 64 + lane] = rec;
    }
  }
  if (lane == 0)
    cx.rs->q1[est][k] = p;
}

// probResGt0 steps at EVERY coefficient, up (p += (2^20 - p) >> 6) where it is non-zero, down (p -= p >> 6)
// where it is zero.  Both steps are monotone in p and every reachable state lies between their fixed points
// 63 and 2^20 - 63, so a thread that takes a chunk of the level does not need the state at the chunk's start
// from its predecessor: it runs the recurrence from the two extremes over a window in front of the chunk, the
// true state stays between the two runs, and where they have met (about 900 steps) it is known exactly; a
// window that was too short is quadrupled, at worst back to the level's first coefficient, where the state is
// the one the previous level left.  (The predicting encoder's rate model is resolved the same way,
// pred_kernels.hpp.)  With the state in hand the thread writes the cost of its coefficients
// (PCCRAHTACCoefficientEntropyEstimate::updateCostBits, RAHT.cpp:54-78: the same additions in the same order).
// (64 coefficients per thread: the chunk's table look-ups are a chain of round trips per thread, so what counts is
// how many threads share the level -- with 256 per thread a 1 M-coefficient level kept 64 wavefronts busy: 1.4 ms)
constexpr int kAcRateChunk = 64;

__device__ __forceinline__ int
rate_p0_step(int p, bool nz)
{
  return nz ? p + (int)((kAcRateScale - (uint32_t)p) >> 6) : p - (p >> 6);
}

__global__ __launch_bounds__(256) void
rate_p0_bits_kernel(RateCtx cx)
{
#pragma clang fp contract(off)
  if (tree_failed(cx.tv))
    return;
  const int C = cx.c;
  const int count = cx.b - cx.a;
  const int chunks = (count + kAcRateChunk - 1) / kAcRateChunk;
  const int64_t total = (int64_t)2 * C * chunks;
  const double lg = (double)kAcRateScaleLog;
  const double* __restrict__ T = cx.log2tab;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (int64_t)gridDim.x * blockDim.x) {
    const int est = (int)(x / ((int64_t)C * chunks));
    const int k = (int)((x / chunks) % C);
    const int chunk = (int)(x % chunks);
    const unsigned long long* __restrict__ nzw = cx.nzw + ((size_t)est * C + k) * cx.wstride;
    const unsigned long long* __restrict__ bigw = cx.bigw + ((size_t)est * C + k) * cx.wstride;
    const int start = chunk * kAcRateChunk;
    const int end = start + kAcRateChunk < count ? start + kAcRateChunk : count;
    int p = cx.rs->p0[est][k];
    if (start > 0) {
      // (the flags a WORD at a time: one load per 64 steps -- read bit by bit the window was ~1 000 dependent
      // loads per thread, 0.6 ms for a 500 k-coefficient level with every step's arithmetic waiting for its load)
      int win = 1024;
      for (;;) {
        const int from = start - win > 0 ? start - win : 0;
        int lo = from ? 63 : p, hi = from ? (int)(kAcRateScale - 63) : p;
        for (int i = from; i < start;) {
          const int wi = i >> 6;
          const int b1 = start - (wi << 6) < 64 ? start - (wi << 6) : 64;
          unsigned long long w = nzw[wi] >> (i & 63);
          for (int b = i & 63; b < b1; b++, w >>= 1) {
            const bool nz = w & 1;
            lo = rate_p0_step(lo, nz);
            hi = rate_p0_step(hi, nz);
          }
          i = (wi << 6) + b1;
        }
        if (lo == hi) {
          p = lo;
          break;
        }
        win *= 4;
      }
    }
    const int32_t* __restrict__ plane = cx.plane[est] + (size_t)k * cx.n + cx.a;
    const int32_t* __restrict__ pb1 = cx.pb + ((size_t)est * C + k) * cx.n;
    double* __restrict__ term = cx.term + (size_t)est * cx.n * C;
    static_assert(kAcRateChunk == 64, "a thread's coefficients are one word of flags");
    const unsigned long long my_nz = nzw[chunk], my_big = bigw[chunk];
    for (int i = start; i < end; i++) {
      const bool nz = (my_nz >> (i & 63)) & 1;
      double bits = 0;
      bits += nz ? lg - T[p] : lg - T[kAcRateScale - (uint32_t)p];
      if (nz) {
        const bool big = (my_big >> (i & 63)) & 1;
        const int p1 = pb1[i];
        bits += big ? lg - T[p1] : lg - T[kAcRateScale - (uint32_t)p1];
        bits += 1;
        if (big) {
          const int32_t value = plane[i];
          const int64_t mag = value < 0 ? -(int64_t)value : (int64_t)value;
          if (mag - 1 > kAcRateTable) {
            // a magnitude the table does not hold: the caller keeps the slice on the CPU
            atomicCAS(cx.tv.error, 0, 4);
          } else {
            bits += 2.0 * T[mag - 1] + 1.0;
          }
        }
      }
      term[(size_t)i * C + k] = bits;
      p = rate_p0_step(p, nz);
    }
    if (end == count)
      cx.rs->q0[est][k] = p;
  }
}

// e->bits += bits, coefficient after coefficient (RAHT.cpp:77): a chain of double additions that no other ORDER
// reproduces -- but most of it can be done without the chain.  While the running sum s stays inside one binade
// [2^e, 2^(e+1)) it is a multiple M of g = 2^(e-52), and adding a term t >= 0 rounds the exact s + t to that grid:
//     RN(M g + t) = (M + rn(t / g)) g      with rn = round to nearest,
// which does not depend on M unless t / g lies exactly half-way between two integers (the tie goes to the EVEN
// neighbour of M + t / g).  So for a chunk of 512 terms with no tie, no negative or oversized term, and
// M + sum rn(t_i / g) < 2^53 (the sum never leaves the binade: the terms are >= 0, so the last prefix is the
// largest) the chain's result is (M + sum_i rn(t_i / g)) g -- an INTEGER sum, any order, one wavefront
// reduction; t / g, rn and the final product are exact (powers of two, integers below 2^53).  A chunk that fails
// one of the tests (the first one, where s starts at 0; the ~25 chunks in which the sum crosses into the next
// binade; a tie: about one term in 2^20) takes the chain as before: the terms go to LDS and are read back with the
// same address in every lane (a broadcast read, its latency off the chain), one dependent v_add_f64 per term.
// One wavefront per estimate, every lane alike.  (Until round 5 every chunk took the chain: 9 cycles per term,
// 4.3 of the 22.6 ms of a 1 M-point inter forward.)
constexpr int kAcSumChunk = 512;

// s += the 512 doubles at `cur` (LDS), in order
__device__ __forceinline__ double
rate_sum_chain(double s, const double* cur)
{
#pragma clang fp contract(off)
#if defined(__HIP_DEVICE_COMPILE__)
  // 32 terms at a time: their sixteen LDS reads are issued together, the additions follow as the reads
  // arrive.  Written out as machine code: left to the compiler every read ends up in front of its own two
  // additions -- whatever the source order or the scheduler hints -- and the chain waits an LDS round trip
  // per pair (measured: 27 cycles per term against 9 here; the addition itself has 8 cycles of latency).
  typedef __attribute__((address_space(3))) const double LdsDouble;
  uint32_t addr = (uint32_t)(uintptr_t)(LdsDouble*)cur;
  for (int u0 = 0; u0 < kAcSumChunk; u0 += 32, addr += 32 * 8) {
    asm volatile(
      "ds_read_b128 v[64:67], %1\n\t"
      "ds_read_b128 v[68:71], %1 offset:16\n\t"
      "ds_read_b128 v[72:75], %1 offset:32\n\t"
      "ds_read_b128 v[76:79], %1 offset:48\n\t"
      "ds_read_b128 v[80:83], %1 offset:64\n\t"
      "ds_read_b128 v[84:87], %1 offset:80\n\t"
      "ds_read_b128 v[88:91], %1 offset:96\n\t"
      "ds_read_b128 v[92:95], %1 offset:112\n\t"
      "ds_read_b128 v[96:99], %1 offset:128\n\t"
      "ds_read_b128 v[100:103], %1 offset:144\n\t"
      "ds_read_b128 v[104:107], %1 offset:160\n\t"
      "ds_read_b128 v[108:111], %1 offset:176\n\t"
      "ds_read_b128 v[112:115], %1 offset:192\n\t"
      "ds_read_b128 v[116:119], %1 offset:208\n\t"
      "ds_read_b128 v[120:123], %1 offset:224\n\t"
      "ds_read_b128 v[124:127], %1 offset:240\n\t"
      "s_waitcnt lgkmcnt(15)\n\t"
      "v_add_f64 %0, %0, v[64:65]\n\t"
      "v_add_f64 %0, %0, v[66:67]\n\t"
      "s_waitcnt lgkmcnt(14)\n\t"
      "v_add_f64 %0, %0, v[68:69]\n\t"
      "v_add_f64 %0, %0, v[70:71]\n\t"
      "s_waitcnt lgkmcnt(13)\n\t"
      "v_add_f64 %0, %0, v[72:73]\n\t"
      "v_add_f64 %0, %0, v[74:75]\n\t"
      "s_waitcnt lgkmcnt(12)\n\t"
      "v_add_f64 %0, %0, v[76:77]\n\t"
      "v_add_f64 %0, %0, v[78:79]\n\t"
      "s_waitcnt lgkmcnt(11)\n\t"
      "v_add_f64 %0, %0, v[80:81]\n\t"
      "v_add_f64 %0, %0, v[82:83]\n\t"
      "s_waitcnt lgkmcnt(10)\n\t"
      "v_add_f64 %0, %0, v[84:85]\n\t"
      "v_add_f64 %0, %0, v[86:87]\n\t"
      "s_waitcnt lgkmcnt(9)\n\t"
      "v_add_f64 %0, %0, v[88:89]\n\t"
      "v_add_f64 %0, %0, v[90:91]\n\t"
      "s_waitcnt lgkmcnt(8)\n\t"
      "v_add_f64 %0, %0, v[92:93]\n\t"
      "v_add_f64 %0, %0, v[94:95]\n\t"
      "s_waitcnt lgkmcnt(7)\n\t"
      "v_add_f64 %0, %0, v[96:97]\n\t"
      "v_add_f64 %0, %0, v[98:99]\n\t"
      "s_waitcnt lgkmcnt(6)\n\t"
      "v_add_f64 %0, %0, v[100:101]\n\t"
      "v_add_f64 %0, %0, v[102:103]\n\t"
      "s_waitcnt lgkmcnt(5)\n\t"
      "v_add_f64 %0, %0, v[104:105]\n\t"
      "v_add_f64 %0, %0, v[106:107]\n\t"
      "s_waitcnt lgkmcnt(4)\n\t"
      "v_add_f64 %0, %0, v[108:109]\n\t"
      "v_add_f64 %0, %0, v[110:111]\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "v_add_f64 %0, %0, v[112:113]\n\t"
      "v_add_f64 %0, %0, v[114:115]\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "v_add_f64 %0, %0, v[116:117]\n\t"
      "v_add_f64 %0, %0, v[118:119]\n\t"
      "s_waitcnt lgkmcnt(1)\n\t"
      "v_add_f64 %0, %0, v[120:121]\n\t"
      "v_add_f64 %0, %0, v[122:123]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_add_f64 %0, %0, v[124:125]\n\t"
      "v_add_f64 %0, %0, v[126:127]\n\t"
      : "+v"(s)
      : "v"(addr)
      : "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
  }
#else
  for (int u = 0; u < kAcSumChunk; u++)
    s += cur[u];
#endif
  return s;
}

__device__ __forceinline__ double
rate_pow2(int k)  // 2^k, -1022 <= k <= 1023
{
  const unsigned long long bits = (unsigned long long)(1023 + k) << 52;
  double v;
  __builtin_memcpy(&v, &bits, 8);
  return v;
}

// Sixteen wavefronts, a group of sixteen chunks at a time, wavefront w owning chunk w of the group (its next chunk's
// loads in flight meanwhile).  A round: every wavefront whose chunk is still open evaluates it under the binade of
// the current sum and publishes (integer sum, usable); all threads then walk the summaries in order -- a few
// integer additions per chunk -- and take chunks as long as they are usable and the sum stays in the binade.  A
// chunk that is not (the level's first, a crossing, a tie) is summed by its owner with the chain, the new sum is
// handed round, and the chunks behind it are evaluated again under the new binade: the next round.  One barrier per
// group when nothing fails.  (One wavefront doing all of this alone is bound by its own instruction stream: ~200
// double / conversion instructions and a six-step shuffle reduction per chunk -- 2.4 ms per 1 M terms against 4.3
// for the chain alone.)
constexpr int kAcSumWaves = 16;
constexpr int kAcSumThreads = kAcSumWaves * 64;

__global__ __launch_bounds__(kAcSumThreads) void
rate_sum_kernel(RateCtx cx)
{
#pragma clang fp contract(off)
  __shared__ double buf[kAcSumChunk];
  __shared__ long long part_sum[2][kAcSumWaves];
  __shared__ int part_bad[2][kAcSumWaves];
  __shared__ double s_pub;
  if (tree_failed(cx.tv))
    return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int est = blockIdx.x;
  const int64_t total = (int64_t)(cx.b - cx.a) * cx.c;
  const double* __restrict__ term = cx.term + (size_t)est * cx.n * cx.c;
  const int64_t nchunks = (total + kAcSumChunk - 1) / kAcSumChunk;
  const int64_t ngroups = (nchunks + kAcSumWaves - 1) / kAcSumWaves;
  constexpr int J = kAcSumChunk / 64, W = kAcSumWaves;
  double cur[J], nxt[J] = {};
  auto load = [&](int64_t grp, double* r) {
#pragma unroll
    for (int j = 0; j < J; j++) {
      const int64_t i = (grp * W + wave) * kAcSumChunk + j * 64 + lane;
      r[j] = i < total ? term[i] : 0.0;
    }
  };
  double s = 0.0;
  if (ngroups > 0)
    load(0, cur);
  int par = 0;  // which half of part_* the next round writes
  for (int64_t grp = 0; grp < ngroups; grp++) {
    if (grp + 1 < ngroups)
      load(grp + 1, nxt);
    int k = 0;  // first open chunk of the group
    while (k < W) {
      // ---- a round: the open chunks under the binade of s ----
      const bool usable_s = s >= 1.0 && s < 1.0e60;
      int e = 0;
      if (usable_s) {
        unsigned long long sb;
        __builtin_memcpy(&sb, &s, 8);
        e = (int)((sb >> 52) & 0x7ff) - 1023;
      }
      const double ginv = rate_pow2(52 - e), g = rate_pow2(e - 52);
      if (wave >= k) {
        // (a lane's eight rn(t / g) are added as doubles: each is an integer below 2^49, so the sum is exact, and
        // there is ONE conversion to int64 per lane -- the conversion is a dozen instructions)
        double lane_sum = 0.0;
        bool bad = !usable_s;
#pragma unroll
        for (int j = 0; j < J; j++) {
          const double x = cur[j] * ginv;
          bad |= !(x >= 0.0 && x < 562949953421312.0);  // negative / NaN, or 2^49 grid steps and more
          const double r = __builtin_rint(x);
          bad |= __builtin_fabs(x - r) == 0.5;
          lane_sum += r;
        }
        long long sum = bad ? 0 : (long long)lane_sum;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1)
          sum += __shfl_xor(sum, m);
        const bool any_bad = __any(bad);
        if (lane == 0) {
          part_sum[par][wave] = sum;
          part_bad[par][wave] = any_bad;
        }
      }
      __syncthreads();
      if (usable_s) {
        const long long M = (long long)(s * ginv);  // 2^52 <= M < 2^53, exact
        long long acc = M;
        long long ps[W];
        int pb[W];
#pragma unroll
        for (int w = 0; w < W; w++) {  // (all summaries with one LDS round trip; the walk below runs on registers)
          ps[w] = part_sum[par][w];
          pb[w] = part_bad[par][w];
        }
        bool open = true;
#pragma unroll
        for (int w = 0; w < W; w++) {
          if (w >= k && open) {
            if (!pb[w] && acc + ps[w] < (1ll << 53))
              acc += ps[w];
            else
              open = false;
          }
          if (w >= k && open)
            k = w + 1;
        }
        s = (double)acc * g;
      }
      par ^= 1;
      if (k == W)
        break;
      // ---- chunk k by the chain, by its owner ----
      if (wave == k) {
#pragma unroll
        for (int j = 0; j < J; j++)
          buf[j * 64 + lane] = cur[j];
        // (the owner's own stores, read back by the same wavefront: no workgroup barrier needed)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        __builtin_amdgcn_wave_barrier();
        const int64_t left = total - (grp * W + k) * kAcSumChunk;
        double t = s;
        if (left >= kAcSumChunk) {
          t = rate_sum_chain(t, buf);
        } else {
          for (int u = 0; u < (int)left; u++)
            t += buf[u];
        }
        if (lane == 0)
          s_pub = t;
      }
      __syncthreads();
      s = s_pub;
      k++;
      // (s_pub and buf are written again only behind the next round's barrier)
    }
#pragma unroll
    for (int j = 0; j < J; j++)
      cur[j] = nxt[j];
  }
  if (threadIdx.x == 0)
    cx.rs->bits[est] = s;
}

// the level's decision (RAHT.cpp:1810-1829)
__global__ __launch_bounds__(64) void
rate_decide_kernel(RateCtx cx)
{
  if (tree_failed(cx.tv))
    return;
  if (threadIdx.x != 0)
    return;
  RateState* rs = cx.rs;
  const int wins = rs->bits[1] < rs->bits[0];
  rs->intra_wins = wins;
  if (rs->num_modes < 32)
    cx.modes[rs->num_modes++] = !wins;
  for (int k = 0; k < 3; k++) {
    const int w0 = rs->q0[wins][k], w1 = rs->q1[wins][k];
    rs->p0[0][k] = rs->p0[1][k] = w0;
    rs->p1[0][k] = rs->p1[1][k] = w1;
  }
  const int lw = wins ? cx.islice_l[0] : cx.slice_l[0];
  cx.slice_l[0] = lw;
  rs->itz = cx.b - 1 - lw;
}

// ... and the intra candidate's coefficients and prediction record in place of the other's when it won
__global__ __launch_bounds__(256) void
inter_commit_kernel(RateCtx cx)
{
  if (tree_failed(cx.tv))
    return;
  if (!cx.rs->intra_wins)
    return;
  const int count = cx.b - cx.a;
  const int64_t nco = (int64_t)count * cx.c;
  const int64_t npt = (int64_t)cx.rows * cx.c;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < nco + npt; x += (int64_t)gridDim.x * blockDim.x) {
    if (x < nco) {
      const int k = (int)(x / count), i = (int)(x % count);
      const size_t at = (size_t)k * cx.n + cx.a + i;
      cx.coeffs[at] = cx.icoeffs[at];
    } else {
      cx.ptrans[x - nco] = cx.iptrans[x - nco];
    }
  }
}

__device__ __forceinline__ int64_t
inter_wave_xor_i64(int64_t v, int mask)
{
  const int lo = __shfl_xor((int)(uint32_t)v, mask);
  const int hi = __shfl_xor((int)(v >> 32), mask);
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

// ---- a filter tap of its own for the level (estimate_layer_filter, RAHT.cpp:849-972) ------------------
// Eight lanes per parent of level li + 1, every parent (no staging: the pass runs once per level and is
// bound by the bisections in the frame).  The current block and the frame's block are transformed each in
// its own weights, FIRST COMPONENT only, without any prediction; over the coefficient positions of the
// current block: autocorr += r * r, crosscorr += r * b (Q15 products in int64, as the reference's).
// The reference walks the frame's nodes with a cursor that never rests on the LAST node of the level: a
// block whose first frame node is that one is not seen (:872-884).
struct TapCtx {
  LevelCtx lc;
  unsigned long long* acc;  // [2] autocorr, crosscorr (cleared by the caller)
  int32_t* taps;            // [32] FilterTaps, quantised, in order
  int32_t* num_taps;
  int32_t* tap_out;         // the level's tap
  int32_t qp_layer;
};

template<int C>
__global__ __launch_bounds__(256) void
inter_tap_kernel(TapCtx cx)
{
  // (50 registers: the allocation class in which finish_kernel's LDS tables miscompared, gpcc_primitives.hpp)
  GPCC_VGPR_FLOOR_64();
  __shared__ SharedLut lut;
  const LevelCtx& ctx = cx.lc;
  const TreeView& tv = ctx.tv;
  if (tree_failed(tv))
    return;
  load_lut(&lut, ctx.lut);
  const int li = ctx.li;
  const ParamsConst prm = (ParamsConst)ctx.params;
  const bool ext = prm->raht_extension != 0;
  const bool inherit_dc = !ctx.sched[0].lvl[li].is_root;
  const int np = tv.soff[li + 1][1];
  const int t = threadIdx.x & 7;
  const InterRef& ir = ctx.inter;
  int64_t autoc = 0, crossc = 0;
  const int groups = (int)(gridDim.x * blockDim.x) >> 3;
  const int rounds = (np + groups - 1) / groups;
  for (int r = 0; r < rounds; r++) {
    const int j = r * groups + (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 3);
    const bool on = j < np;
    const int c0 = on ? tv.fc[li + 1][j] : 0;
    const int nchild = on ? tv.fc[li + 1][j + 1] - c0 : 0;
    const bool take = on && !(ext && nchild == 1);
    const uint32_t occ = group8_or(t < nchild ? 1u << (int)(tv.key[li][c0 + t] & 7) : 0u);
    const bool has = take && ((occ >> t) & 1);
    const int child = c0 + popc32(occ & ((1u << t) - 1));
    int32_t w = 0;
    int64_t b = 0;
    if (has) {
      const int fa = tv.fp[li][child], fb = tv.fp[li][child + 1];
      w = fb - fa;
      // (under the integer Haar kernel the nodes hold low-pass values, and the estimate still runs RAHT
      // butterflies over them: estimate_layer_filter knows no Haar kernel)
      b = ir.hkey ? fp_from_int(ctx.haar_lf[li][(size_t)child * C])
                  : fp_from_int((int32_t)((uint32_t)ctx.attr_prefix[(size_t)fb * C] - (uint32_t)ctx.attr_prefix[(size_t)fa * C]));
    }
    // the frame's block, first component
    int32_t wr = 0;
    int64_t rv = 0;
    int hi = 0;        // what follows the node: a point index (or, integer Haar, a node index) ...
    int hi_end = ir.n_ref;  // ... and where the level ends
    if (take && !ir.hkey) {
      const uint64_t k0 = ((uint64_t)tv.key[li + 1][j] << 3) + (uint64_t)t;
      const int lo = inter_lower_bound(ir.pos, ir.n_ref, ir.lr, k0);
      hi = inter_lower_bound(ir.pos, ir.n_ref, ir.lr, k0 + 1);
      wr = hi - lo;
      if (wr > 0)
        rv = fp_from_int((int32_t)((uint32_t)ir.prefix[(size_t)hi * C] - (uint32_t)ir.prefix[(size_t)lo * C]));
    }
    if (take && ir.hkey) {
      const int64_t want = (int64_t)(((uint64_t)tv.key[li + 1][j] << 3) + (uint64_t)t);
      int lo = 0, up = ir.hsoff[1];
      hi_end = up;
      while (lo < up) {
        const int mid = lo + ((up - lo) >> 1);
        if (ir.hkey[mid] < want)
          lo = mid + 1;
        else
          up = mid;
      }
      if (lo < hi_end && ir.hkey[lo] == want) {
        wr = ir.hfp[lo + 1] - ir.hfp[lo];
        rv = fp_from_int(ir.hlf[(size_t)lo * C]);
        hi = lo + 1;
      }
    }
    const uint32_t rocc = group8_bits(wr > 0);
    // (the cursor's rule: the block's first frame node is not the last node of the level)
    const int first_t = rocc ? __ffs((int)rocc) - 1 : 0;
    const int hi_first = __shfl(hi, (int)((threadIdx.x & 56) | first_t));
    const bool match = take && rocc != 0 && hi_first < hi_end;
    if (w > 1)
      b = scale_rsqrt(b, w, lut);
    if (wr > 1)
      rv = scale_rsqrt(rv, wr, lut);
    int32_t cw = w, cwr = wr;
#pragma unroll
    for (int st = 0; st < 3; st++) {
      const int bit = 1 << st;
      const bool left = !(t & bit);
      {
        const int32_t pw = lane_xor8(cw, bit);
        const int32_t wl = left ? cw : pw, wrr = left ? pw : cw;
        const bool both = wl && wrr, swap = !wl && wrr;
        int64_t ca = 0, cb = 0;
        if (both)
          raht_coeffs(wl, wrr, lut, &ca, &cb);
        const int64_t own = b;
        const int64_t oth = shfl_xor_i64(own, bit);
        if (both)
          b = left ? fp_mul_c(oth, cb) + fp_mul_c(own, ca) : fp_mul_c(own, ca) - fp_mul_c(oth, cb);
        else if (swap)
          b = oth;
        cw = both ? wl + wrr : (left ? wl + wrr : 0);
      }
      {
        const int32_t pw = lane_xor8(cwr, bit);
        const int32_t wl = left ? cwr : pw, wrr = left ? pw : cwr;
        const bool both = wl && wrr, swap = !wl && wrr;
        int64_t ca = 0, cb = 0;
        if (both)
          raht_coeffs(wl, wrr, lut, &ca, &cb);
        const int64_t own = rv;
        const int64_t oth = shfl_xor_i64(own, bit);
        if (both)
          rv = left ? fp_mul_c(oth, cb) + fp_mul_c(own, ca) : fp_mul_c(own, ca) - fp_mul_c(oth, cb);
        else if (swap)
          rv = oth;
        cwr = both ? wl + wrr : (left ? wl + wrr : 0);
      }
    }
    const bool counted = match && (t == 0 ? !inherit_dc : cw != 0);
    if (counted && rv) {
      autoc += (int64_t)((uint64_t)rv * (uint64_t)rv) >> kFpFrac;
      crossc += (int64_t)((uint64_t)rv * (uint64_t)b) >> kFpFrac;
    }
  }
  // one pair of atomics per wavefront
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    autoc += inter_wave_xor_i64(autoc, d);
    crossc += inter_wave_xor_i64(crossc, d);
  }
  if ((threadIdx.x & 63) == 0 && (autoc || crossc)) {
    atomicAdd(&cx.acc[0], (unsigned long long)autoc);
    atomicAdd(&cx.acc[1], (unsigned long long)crossc);
  }
}

// getFilterTap (RAHT.cpp:805-846): 128 * crosscorr / autocorr, the fraction by bisection of (mid * autocorr) >> 7
__device__ __forceinline__ int
filter_tap_of(int64_t autocorr, int64_t crosscorr)
{
  if (crosscorr == 0)
    return 0;
  const bool neg = crosscorr < 0;
  crosscorr = neg ? -crosscorr : crosscorr;
  if (crosscorr == autocorr)
    return neg ? -128 : 128;
  int64_t tapint = 0;
  if (crosscorr >= autocorr) {
    // (the reference subtracts in a loop)
    tapint = 128 * (crosscorr / autocorr);
    crosscorr %= autocorr;
  }
  if (crosscorr == 0)
    return (int)(neg ? -tapint : tapint);
  int lo = 0, hi = 128;
  while (lo < hi - 1) {
    const int mid = (lo + hi) >> 1;
    const int64_t midval = (mid * autocorr) >> 7;
    if (crosscorr == midval)
      return (int)(neg ? -(tapint + mid) : (tapint + mid));
    if (crosscorr < midval)
      hi = mid;
    else
      lo = mid;
  }
  return (int)(neg ? -(tapint + lo) : (tapint + lo));
}

// the tap quantised like a coefficient of the level (:1287-1300)
__global__ __launch_bounds__(64) void
inter_tap_finish_kernel(TapCtx cx)
{
  if (tree_failed(cx.lc.tv))
    return;
  if (threadIdx.x != 0)
    return;
  const ParamsConst prm = (ParamsConst)cx.lc.params;
  Quantizer tq[2];
  qpset_quantizers(prm, cx.qp_layer, 0, 0, tq);
  const int64_t autocorr = (int64_t)cx.acc[0], crosscorr = (int64_t)cx.acc[1];
  const int orig = autocorr > 0 ? filter_tap_of(autocorr, crosscorr) : 128;
  const int64_t qtap = quantize(tq[0], (int64_t)(128 - orig) * 256);
  if (*cx.num_taps < 32)
    cx.taps[(*cx.num_taps)++] = (int32_t)qtap;
  *cx.tap_out = (int32_t)(128 - dequantize(tq[0], qtap));
  cx.acc[0] = cx.acc[1] = 0;
}

// the decoder's tap of a level from the signalled value (:1301-1304); qtap < 0x7fffffff
__global__ __launch_bounds__(64) void
inter_tap_decode_kernel(const gpcc_raht_params* params, int qp_layer, int32_t qtap, int32_t* tap_out)
{
  if (threadIdx.x != 0)
    return;
  Quantizer tq[2];
  qpset_quantizers((ParamsConst)params, qp_layer, 0, 0, tq);
  *tap_out = (int32_t)(128 - dequantize(tq[0], (int64_t)qtap));
}

__global__ __launch_bounds__(64) void
inter_set_word_kernel(int32_t* p, int32_t v)
{
  if (threadIdx.x == 0)
    *p = v;
}

// modular prefix sums of the frame's attributes, [n + 1][C] (three launches: tile sums, their scan, emit)
template<int C>
__global__ __launch_bounds__(256) void
frame_sum_kernel(const int32_t* __restrict__ a, int n, int32_t* __restrict__ tile)
{
  const int lane = lane_id();
  const int wave = (int)(blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int nwaves = (int)(gridDim.x * blockDim.x) / kWave;
  const int ntiles = (n + kTilePoints - 1) / kTilePoints;
  for (int tl = wave; tl < ntiles; tl += nwaves) {
    uint32_t acc[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      acc[k] = 0;
    for (int r = 0; r < kTilePoints / kWave; r++) {
      const int i = tl * kTilePoints + r * kWave + lane;
      if (i < n) {
#pragma unroll
        for (int k = 0; k < C; k++)
          acc[k] += (uint32_t)a[(size_t)i * C + k];
      }
    }
#pragma unroll
    for (int k = 0; k < C; k++) {
      uint32_t v = acc[k];
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1)
        v += (uint32_t)__shfl_xor((int)v, d);
      if (lane == 0)
        tile[(size_t)tl * C + k] = (int32_t)v;
    }
  }
}

template<int C>
__global__ __launch_bounds__(64) void
frame_scan_kernel(int32_t* tile, int ntiles)
{
  const int lane = lane_id();
  for (int k = 0; k < C; k++) {
    uint32_t running = 0;
    for (int t0 = 0; t0 < ntiles; t0 += kWave) {
      const int tl = t0 + lane;
      const uint32_t v = tl < ntiles ? (uint32_t)tile[(size_t)tl * C + k] : 0u;
      const uint32_t inc = wave_incl_scan_u32(v);
      if (tl < ntiles)
        tile[(size_t)tl * C + k] = (int32_t)(running + inc - v);
      running += (uint32_t)__shfl((int)inc, kWave - 1);
    }
  }
}

template<int C>
__global__ __launch_bounds__(256) void
frame_prefix_kernel(const int32_t* __restrict__ a, int n, const int32_t* __restrict__ tile, int32_t* __restrict__ prefix)
{
  const int lane = lane_id();
  const int wave = (int)(blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int nwaves = (int)(gridDim.x * blockDim.x) / kWave;
  const int ntiles = (n + kTilePoints - 1) / kTilePoints;
  for (int tl = wave; tl < ntiles; tl += nwaves) {
    uint32_t run[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      run[k] = (uint32_t)tile[(size_t)tl * C + k];
    for (int r = 0; r < kTilePoints / kWave; r++) {
      const int i = tl * kTilePoints + r * kWave + lane;
      const bool in = i < n;
#pragma unroll
      for (int k = 0; k < C; k++) {
        const uint32_t v = in ? (uint32_t)a[(size_t)i * C + k] : 0u;
        const uint32_t inc = wave_incl_scan_u32(v);
        if (in)
          prefix[(size_t)i * C + k] = (int32_t)(run[k] + inc - v);
        if (i == n - 1)
          prefix[(size_t)n * C + k] = (int32_t)(run[k] + inc);
        run[k] += (uint32_t)__shfl((int)inc, kWave - 1);
      }
    }
  }
}

}  // namespace gpcc
