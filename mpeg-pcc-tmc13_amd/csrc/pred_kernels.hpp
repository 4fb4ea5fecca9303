// pred_kernels.hpp -- the predicting transform on gfx950, given the LoD
// structure (predictors in coding order).
//
// The reference walks the predictors one by one
// (encode/decode{Colors,Reflectances}Pred, tmc3/AttributeEncoder.cpp:749-853,
// 1075-1210, tmc3/AttributeDecoder.cpp:328-523): a point is predicted from the
// RECONSTRUCTED values of up to three neighbours that precede it in coding
// order -- of coarser levels of detail and, unless
// intra_lod_prediction_skip_layers excludes it, of its own.  That is a
// dependency DAG over the points, not a level-synchronous structure, so:
//   * quantisation weights (computeQuantizationWeights, PCCTMC3Common.h:
//     895-922): the DAG walked from the last predictor to the first; a point
//     is final once every later point that references it has added its share
//     (in-degree counted up front, 64-bit atomic adds commute);
//   * reconstruction: the DAG walked forward; wavefronts claim 64 consecutive
//     predictors in coding order (a ticket, so everything a claim waits for is
//     owned by a wavefront that is already running), a lane publishes its
//     reconstruction as ONE write-through granule {values, tag} that doubles
//     as the done flag, neighbours inside the claim are handed over lane to
//     lane (ds_bpermute) without touching memory.
// The decoder is complete (prediction modes hidden in the coefficient
// parities, inter-component prediction, QP layers, region offsets).  The
// encoder's choice among direct predictors reads a running rate model that
// every EARLIER point has updated (PCCResidualsEncoder::resStatUpdate*,
// AttributeEncoder.cpp:136-159): that is one serial scan over the slice with a
// double-precision log2 per candidate, so the device encoder covers
// max_num_direct_predictors == 0 and returns GPCC_ERR_UNSUPPORTED otherwise
// (the caller keeps the reference's loop).
#pragma once

#include "lift_kernels.hpp"

namespace gpcc {

struct PredCtx {
  int32_t n, c;
  int32_t num_lods;
  int32_t npl[GPCC_MAX_LODS];
  // the reference's running counters per range between distinct LoD
  // boundaries (replayed on the host): quantLayer, the `lod` of the coding
  // loop (icpCoeffs[lod]) and the `lod` of the coefficient estimation
  int32_t num_ranges;
  int32_t range_start[kMaxLodRanges];
  int32_t range_qlayer[kMaxLodRanges];
  int32_t range_lod[kMaxLodRanges];
  int32_t range_est[kMaxLodRanges];
  int32_t est_resolved;  // estimation: LoDs [0, est_resolved) reach their boundary
  int32_t max_levels;
  int32_t bitdepth;
  int32_t num_qp_layers;
  int32_t layer_qp[GPCC_MAX_QP_LAYERS][2];
  int32_t max_qp;
  int32_t max_direct, avg_disabled, threshold, icp_enabled;
  int32_t qnw[3];
  const int32_t* nc;
  const int32_t* ni;
  const int32_t* nw;
  const int32_t* indexes;
  const int32_t* qp_off;
  int32_t* attrs;   // [n][c] point order
  int32_t* values;  // [n][c] coding order
  int8_t* icp;      // [GPCC_MAX_LODS][3]
  int32_t* indeg;   // [n]
  int32_t* recv;    // [n]
  unsigned long long* acc;  // [n]
  unsigned long long* qw;   // [n]
  // [n][4] {r, g, b, tag}: one 16-byte granule per predictor.  HARDWARE ASSUMPTION
  // (as for the granules of lod_kernels.hpp and raht_subnode.hpp): a naturally
  // aligned 16-byte sc1 store / load is observed whole -- a reader that sees the tag
  // sees the three values of the same store (MI355X_MICROARCH.md, hand-off section);
  // there is no release / acquire pair around it.
  uint32_t* rec;
  uint32_t tag;       // tag of this pass's granules (the encoder's mode decision iterates passes)
  // encoder with direct predictors (pred_rate_* below)
  const int32_t* src;       // [n][c] source attributes, point order (attrs then only takes the reconstruction)
  const int32_t* rm;        // [n][6] rate model before every predictor: probResGt0[3], probResGt1[3]
  const double* log2tab;    // [2^20 + 1] log2 of every integer, computed by the HOST's libm
  int32_t* ticket;  // [2]
  int32_t* wide;    // set by pred_indegree_kernel: an in-degree >= 2^20
  int32_t packed_ok;  // quant_neigh_weight >= 0 and their sum < 256
  int32_t* error;
  unsigned long long* icp_sums;  // [GPCC_MAX_LODS][18]: 8 weights x {k=1,2}, orig x {1,2}
  // attribute inter prediction (pred_dag_kernel<.., true>, one component): a neighbour index
  // >= n names entry (index - n) of the reference frame; its value is always there
  // (predictReflectance PCCTMC3Common.h:555-585, predModeEligibleRefl
  // AttributeCommon.cpp:176-210, decidePredModeRefl AttributeEncoder.cpp:663-717 all read the
  // frame's reflectance for such a neighbour).  indeg / recv / acc / qw have n_frame spare
  // entries behind the n predictors: the shares such a neighbour would be skipped for
  // (computeQuantizationWeights :913-914) land there.
  const int32_t* frame_attr;  // [n_frame]
};

__device__ __forceinline__ int
pred_range_of(const PredCtx& cx, int i)
{
  int r = 0;
  for (int k = 1; k < cx.num_ranges; k++)
    r += i >= cx.range_start[k];
  return r;
}

constexpr int kPredCountShift = 44;  // packed share word: (count << 44) | sum

__global__ __launch_bounds__(256) void
pred_indegree_kernel(PredCtx cx)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cx.n; i += gridDim.x * blockDim.x) {
    const int cnt = cx.nc[i];
    for (int j = 0; j < cnt; j++)
      if (atomicAdd(&cx.indeg[cx.ni[3 * (size_t)i + j]], 1) + 1 >= (1 << (64 - kPredCountShift)))
        atomicExch(cx.wide, 1);  // a count that does not fit the packed word
  }
}

// computeQuantizationWeights: predictors from the last to the first.  A
// wavefront claims 64 consecutive predictors (descending).  A point's shares
// and the number of referrers that have delivered travel in ONE 64-bit word,
// (count << 44) | sum, so a delivery is a single fire-and-forget atomic add
// and the word a consumer polls is complete when its count is (no fence, no
// wait on the producer's side).  44 bits hold every sum when the three
// quant_neigh_weights are non-negative and add up to < 256: the shares a point
// hands on are then less than its own weight, so all weights together stay
// below 2^16 n.  Shares for a point of the same claim go through the
// wavefront's LDS slots, the rest through memory; a lane polls memory only
// until the referrers OUTSIDE its claim have delivered (the in-degree minus the
// referrers counted inside the claim), after that the claim iterates on LDS
// alone.  In-degrees of 2^20 or more, or other weights: kWide, separate sum and
// count words with release / acquire ordering.
template<bool kWide>
__device__ __forceinline__ void
pred_quant_weights_body(const PredCtx& cx)
{
  __shared__ unsigned long long lacc[4][64];
  __shared__ int lrecv[4][64];
  __shared__ int lneed[4][64];
  constexpr unsigned long long kOne = 1ull << kPredCountShift, kSumMask = kOne - 1;
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  for (;;) {
    int tk = 0;
    if (lane == 0)
      tk = atomicAdd(&cx.ticket[0], 1);
    tk = __shfl(tk, 0);
    const int64_t base = (int64_t)tk * 64;
    if (base >= cx.n)
      break;
    if (__hip_atomic_load(cx.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      break;
    const int i = cx.n - 1 - (int)(base + lane);
    bool pending = i >= 0;
    int cnt = 0, need = 0;
    int nb[3] = {0, 0, 0};
    int tl[3] = {64, 64, 64};  // lane of a neighbour inside the claim (> own lane), 64 = outside
    __hip_atomic_store(&lacc[wv][lane], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&lrecv[wv][lane], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&lneed[wv][lane], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (pending) {
      cnt = cx.nc[i];
      need = cx.indeg[i];
      for (int j = 0; j < 3; j++)
        if (j < cnt) {
          nb[j] = cx.ni[3 * (size_t)i + j];
          const int64_t t = (int64_t)(cx.n - 1 - nb[j]) - base;
          // (t < 0: a neighbour in the reference frame, addressed behind the n predictors --
          // its share goes the memory way, into an entry nobody reads)
          tl[j] = t >= 0 && t < 64 ? (int)t : 64;
        }
    }
    // (the wavefront is in lock step here: every slot is cleared before a lane counts into
    // another lane's, every count is in before a lane reads its own -- said to the compiler,
    // and to the CPU emulator of tests/emu, with wave barriers; no instruction on the device)
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < 3; j++)
      if (tl[j] < 64)
        __hip_atomic_fetch_add(&lneed[wv][tl[j]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_wave_barrier();
    const int need_l = __hip_atomic_load(&lneed[wv][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int need_g = need - need_l;
    bool gdone = !pending || need_g == 0;
    unsigned long long gsum = 0;
    unsigned spins = 0;
    while (__any(pending)) {
      if (pending && !gdone) {
        if (kWide) {
          gdone = __hip_atomic_load(&cx.recv[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == need_g;
          if (gdone)
            gsum = __hip_atomic_load(&cx.acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          const unsigned long long v =
            __hip_atomic_load(&cx.acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gdone = (int)(v >> kPredCountShift) == need_g;
          gsum = v & kSumMask;
        }
      }
      bool ready = false;
      unsigned long long lsum = 0;
      if (pending && gdone) {
        if (kWide) {
          ready = __hip_atomic_load(&lrecv[wv][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == need_l;
          if (ready)
            lsum = __hip_atomic_load(&lacc[wv][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          const unsigned long long v =
            __hip_atomic_load(&lacc[wv][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          ready = (int)(v >> kPredCountShift) == need_l;
          lsum = v & kSumMask;
        }
      }
      if (ready) {
        const uint64_t w = 256 + gsum + lsum;
        cx.qw[i] = w;
        for (int j = 0; j < 3; j++) {
          if (j >= cnt)
            continue;
          // int32 * uint64 -> uint64 in the reference: the unsigned overload of
          // divExp2RoundHalfInf (PCCMath.h:678-685), modular product, logical shift
          unsigned long long share = ((unsigned long long)(long long)cx.qnw[j] * w + 128ull) >> 8;
          if (!kWide)
            share += kOne;
          if (tl[j] < 64)
            __hip_atomic_fetch_add(&lacc[wv][tl[j]], share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else
            atomicAdd(&cx.acc[nb[j]], share);
        }
        if (kWide)
          for (int j = 0; j < 3; j++) {
            if (j >= cnt)
              continue;
            if (tl[j] < 64)
              __hip_atomic_fetch_add(&lrecv[wv][tl[j]], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
              __hip_atomic_fetch_add(&cx.recv[nb[j]], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          }
        pending = false;
      }
      if (++spins > (1u << 24)) {
        if (lane == 0)
          atomicExch(cx.error, 1);
        break;
      }
    }
  }
}

__global__ __launch_bounds__(256) void
pred_quant_weights_kernel(PredCtx cx)
{
  // uniform: the flag was written by the kernel before this one
  if (cx.packed_ok && !*cx.wide)
    pred_quant_weights_body<false>(cx);
  else
    pred_quant_weights_body<true>(cx);
}

// ---- computeInterComponentPredictionCoeffs (encoder) -----------------------
__global__ __launch_bounds__(256) void
pred_icp_sums_kernel(PredCtx cx)
{
  __shared__ unsigned long long s[GPCC_MAX_LODS][18];
  for (int t = threadIdx.x; t < GPCC_MAX_LODS * 18; t += blockDim.x)
    (&s[0][0])[t] = 0;
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cx.n; i += gridDim.x * blockDim.x) {
    const int lod = cx.range_est[pred_range_of(cx, i)];
    int32_t resid[3];
    const int32_t* me = &cx.attrs[3 * (size_t)cx.indexes[i]];
    const bool has = cx.nc[i] >= 1;
    const int32_t* nbp = has ? &cx.attrs[3 * (size_t)cx.indexes[cx.ni[3 * (size_t)i]]] : me;
    for (int k = 0; k < 3; k++)
      resid[k] = me[k] - (has ? nbp[k] : 0);
    for (int w = 0; w < 8; w++)
      for (int k = 1; k < 3; k++)
        atomicAdd(
          &s[lod][2 * w + k - 1],
          (unsigned long long)abs(resid[k] - (((w + 1) * resid[0] + 2) >> 2)));
    for (int k = 1; k < 3; k++)
      atomicAdd(&s[lod][16 + k - 1], (unsigned long long)abs(resid[k]));
  }
  __syncthreads();
  for (int t = threadIdx.x; t < GPCC_MAX_LODS * 18; t += blockDim.x)
    if ((&s[0][0])[t])
      atomicAdd(&cx.icp_sums[t], (&s[0][0])[t]);
}

__global__ void
pred_icp_resolve_kernel(PredCtx cx)
{
  const int lod = threadIdx.x;
  if (lod >= GPCC_MAX_LODS)
    return;
  int8_t out[3] = {0, 0, 0};
  if (lod < cx.est_resolved && lod < cx.max_levels) {
    const unsigned long long* s = &cx.icp_sums[18 * lod];
    for (int k = 1; k < 3; k++) {
      int best = 0;
      for (int w = 1; w < 8; w++)
        if (s[2 * w + k - 1] < s[2 * best + k - 1])
          best = w;
      out[k] = s[2 * best + k - 1] > s[16 + k - 1] ? 0 : (int8_t)(1 + best);
    }
  }
  for (int k = 0; k < 3; k++)
    cx.icp[3 * lod + k] = out[k];
}

// ---- the encoder's rate model (PCCResidualsEncoder, AttributeEncoder.cpp:81-222) -------
// The choice among direct predictors (decidePredModeRefl :663-745, decidePredModeColor
// :896-985) scores every candidate with an estimate of its coded size from two running
// probabilities per component, updated after every predictor in coding order
// (resStatUpdate :137-165): the one loop-carried state of the encoder that is not a
// neighbour value.  Here the state BEFORE every predictor is an input of the DAG pass
// (cx.rm), recomputed from the pass's values by pred_rate_scan_kernel, and the two are
// iterated: a pass whose values equal the previous pass's has used exactly the states the
// sequential coder would have had (induction over the coding order), so the fixed point IS
// the reference's result.  Each pass is exact up to its first wrong decision, and a wrong
// state decays by 1/64 per predictor, so a handful of passes settle a slice.
// The estimate takes log2 of integers below 2^20: the table is filled by the host's libm
// (the reference's own log2), the sums are evaluated in the reference's order without
// contraction -- identical doubles, identical decisions.
constexpr int kRateScale = 1 << 20, kRateWindowLog2 = 6;

__device__ __forceinline__ double
pred_log2i(const PredCtx& cx, int64_t v)
{
  return v >= 0 && v <= kRateScale ? cx.log2tab[v] : log2((double)v);
}

__device__ __forceinline__ double
pred_rate_component(const PredCtx& cx, const int32_t* rm, int k, int32_t value)
{
#pragma clang fp contract(off)
  const int l2 = 20;  // ilog2(scaleRes)
  double bits = 0;
  bits += value ? l2 - pred_log2i(cx, rm[k]) : l2 - pred_log2i(cx, kRateScale - rm[k]);
  const int mag = value < 0 ? -value : value;
  if (mag) {
    bits += mag > 1 ? l2 - pred_log2i(cx, rm[3 + k]) : l2 - pred_log2i(cx, kRateScale - rm[3 + k]);
    bits += 1;
    if (mag > 1)
      bits += 2.0 * pred_log2i(cx, (int64_t)mag - 1) + 1.0;
  }
  return bits;
}

// bitsPtRefl (:203-222)
__device__ __forceinline__ double
pred_rate_refl(const PredCtx& cx, const int32_t* rm, int avail, int32_t value, int mode)
{
#pragma clang fp contract(off)
  const int a = value < 0 ? -value : value;
  if (avail == 4) {
    value = (a << 2) + mode;
  } else if (avail == 3) {
    int v = a;
    if (mode > 0)
      v = (v << 1) + (mode - 1);
    value = (v << 1) + (mode > 0);
  } else if (avail == 2) {
    value = (a << 1) + (mode & 1);
  }
  double bits = 0;
  bits += pred_rate_component(cx, rm, 0, value);
  return bits;
}

// bitsPtColor (:168-199)
__device__ __forceinline__ double
pred_rate_colour(const PredCtx& cx, const int32_t* rm, int avail, const int64_t r[3], int mode)
{
#pragma clang fp contract(off)
  int32_t v[3] = {(int32_t)r[0], (int32_t)r[1], (int32_t)r[2]};
  const int a1 = v[1] < 0 ? -v[1] : v[1], a2 = v[2] < 0 ? -v[2] : v[2];
  if (avail == 4) {
    v[1] = 2 * a1 + (mode >> 1);
    v[2] = 2 * a2 + (mode & 1);
  } else if (avail == 3) {
    v[1] = 2 * a1 + (mode > 0);
    if (mode > 0)
      v[2] = 2 * a2 + (mode - 1);
  } else if (avail == 2) {
    v[1] = 2 * a1 + (mode & 1);
  }
  double bits = 0;
  for (int k = 0; k < 3; k++)
    bits += pred_rate_component(cx, rm, k, v[k]);
  return bits;
}

__device__ __forceinline__ int64_t
pred_half_up8(int64_t x)
{
  return (x + 128) >> 8;
}

// computeColorResiduals (:858-890)
__device__ __forceinline__ void
pred_colour_residuals(
  bool icp_on, const int32_t col[3], const int64_t pred[3], const int8_t icp[3], const Quantizer q[2],
  int64_t r[3])
{
  r[0] = quantize(q[0], ((int64_t)col[0] - pred[0]) << 8);
  const int64_t residual0 = pred_half_up8(mul_i64_u32(r[0], q[0].step));
  for (int k = 1; k < 3; k++) {
    int64_t err = (int64_t)col[k] - pred[k];
    if (icp_on)
      err -= ((int64_t)icp[k] * residual0 + 2) >> 2;
    r[k] = quantize(q[1], err << 8);
  }
}

// computeColorDistortions (:792-823)
__device__ __forceinline__ int
pred_colour_distortion(int64_t clip_max, const int32_t col[3], const int64_t pred[3], const Quantizer q[2])
{
  int d = 0;
  for (int k = 0; k < 3; k++) {
    const Quantizer qq = q[k ? 1 : 0];
    const int64_t rq = quantize(qq, ((int64_t)col[k] - pred[k]) << 8);
    int64_t rec = pred[k] + pred_half_up8(mul_i64_u32(rq, qq.step));
    rec = rec < 0 ? 0 : (rec > clip_max ? clip_max : rec);
    const int e = (int)((int64_t)col[k] - (int64_t)(uint16_t)rec);
    d += e < 0 ? -e : e;
  }
  return d;
}

__device__ __forceinline__ int
pred_rate_step(int x, bool up)
{
  return up ? x + ((kRateScale - x) >> kRateWindowLog2) : x - (x >> kRateWindowLog2);
}

__global__ __launch_bounds__(256) void
pred_rate_init_kernel(int32_t* rm, int n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)n * 6; i += (size_t)gridDim.x * blockDim.x)
    rm[i] = kRateScale >> 1;
}

// The states before every predictor from the values of a pass.  probResGt0 steps at
// every predictor (up when the value is non-zero); probResGt1 only at the non-zero
// values (up when the magnitude exceeds 1), so its recurrence runs over the COMPACTED
// list of those events (flags -> ranks by a scan -> events) and a predictor reads the
// state in front of the first event at or behind it.  A recurrence is cut into chunks of
// kRateChunk events, one thread each: a thread does not know the state at its chunk's
// start, so it runs the recurrence from the two EXTREME reachable states (63 and
// 2^20 - 63: the fixed points of the two steps) over a warm-up window in front of the
// chunk -- every step is monotone in the state, the true state lies between the two
// runs, and where they have met it is known exactly.  A warm-up that has not met is
// quadrupled (at worst back to the first event, where the state is the initial one).
constexpr int kRateChunk = 256;
constexpr int kRateMin = 63, kRateMax = kRateScale - 63;

// rank[i + 1] = (value of component k at predictor i is non-zero); rank[0] = 0
__global__ __launch_bounds__(256) void
pred_rate_flags_kernel(const int32_t* __restrict__ values, int n, int c, int k, int32_t* __restrict__ rank)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    rank[i + 1] = values[(size_t)i * c + k] != 0;
    if (i == 0)
      rank[0] = 0;
  }
}

// after the inclusive scan rank[i] = non-zero values before predictor i
__global__ __launch_bounds__(256) void
pred_rate_events_kernel(
  const int32_t* __restrict__ values, int n, int c, int k, const int32_t* __restrict__ rank,
  uint8_t* __restrict__ ev)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int32_t v = values[(size_t)i * c + k];
    if (v)
      ev[rank[i]] = (v < 0 ? -v : v) > 1;
  }
}

// One recurrence over `m` events (up[e] from the values for probResGt0: stride c,
// non-zero test; from the event bytes for probResGt1); state[e] = the state in front of
// event e, state[m] behind the last (m is read from *m_ptr when given).
// One wavefront per 64 chunks.  The wavefront first packs the flags of ITS events -- the 1024 of the first
// chunk's warm-up and the 64 x 256 of the chunks -- into bit words in LDS with coalesced loads and ballots; the
// threads then run their recurrences on the words, stage the states in LDS and the wavefront writes them out
// 64 consecutive events per store.  (Until round 5 a thread read its 1 280 flags itself, sixteen loads at a
// time, every lane in a different cache line, and stored its 256 states the same way: 1.07 ms per call for
// 1 M events, 8.6 ms of a predicting encode.)  A warm-up that has not met after 1024 events is continued from
// global memory with quadrupled windows, as before.
constexpr int kRateWarm = 1024;
constexpr int kRateHalf = kRateChunk / 2;

__global__ __launch_bounds__(64) void
pred_rate_scan_kernel(
  const int32_t* __restrict__ values, int stride, const uint8_t* __restrict__ ev, int m_max,
  const int32_t* __restrict__ m_ptr, int32_t* __restrict__ state, int state_stride, int write_final)
{
  __shared__ unsigned long long wbits[(kRateWarm + 64 * kRateChunk) / 64];
  __shared__ int32_t stage[64 * (kRateHalf + 1)];
  const int m = m_ptr ? *m_ptr : m_max;
  const int lane = threadIdx.x;
  const long long wave_start = (long long)blockIdx.x * 64 * kRateChunk;
  if (wave_start > m || (wave_start == m && m > 0))
    return;  // (no chunk of this wavefront holds an event, and state[m] is another wavefront's)
  auto up = [&](long long e) -> bool { return values ? values[(size_t)e * stride] != 0 : ev[e] != 0; };
  // ---- the flags of events [base, last) as bit words: bit (e - base) ----
  const long long base = wave_start - kRateWarm;
  const long long last = wave_start + 64 * kRateChunk < m ? wave_start + 64 * kRateChunk : m;
  const int nwords = (int)((last - base + 63) >> 6);
  // (sixteen words per batch: the sixteen loads are unconditional -- the index clamped into the range, the range
  // test applied to the loaded flag -- so that they are in flight together; behind a test per event every load
  // waited for its predecessor's ballot: 272 round trips to memory per wavefront)
  if (last > 0) {
    auto pack = [&](auto flag_at) {
      for (int w0 = 0; w0 < nwords; w0 += 16) {
        int f[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const long long e = base + (long long)(w0 + j) * 64 + lane;
          const long long ec = e < 0 ? 0 : (e >= last ? last - 1 : e);
          f[j] = flag_at(ec);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const long long e = base + (long long)(w0 + j) * 64 + lane;
          const unsigned long long b = __ballot((f[j] != 0) & (e >= 0) & (e < last));
          if (lane == 0 && w0 + j < nwords)
            wbits[w0 + j] = b;
        }
      }
    };
    if (values)
      pack([&](long long e) -> int { return values[(size_t)e * stride]; });
    else
      pack([&](long long e) -> int { return ev[e]; });
  } else {
    for (int w = lane; w < nwords; w += 64)
      wbits[w] = 0;
  }
  __syncthreads();
  const long long start = wave_start + (long long)lane * kRateChunk;
  const bool active = !(start > m || (start == m && m > 0));
  const int end = !active ? 0 : (start + kRateChunk < m ? (int)(start + kRateChunk) : m);
  int x = kRateScale >> 1;
  if (active && start > 0) {
    // warm-up over [b, start): b = start - 1024 or 0
    const int b = start - kRateWarm > 0 ? (int)(start - kRateWarm) : 0;
    int lo = b == 0 ? kRateScale >> 1 : kRateMin, hi = b == 0 ? kRateScale >> 1 : kRateMax;
    for (long long e = b; e < start;) {
      const int bit = (int)(e - base);
      const int b1 = start - (e - (bit & 63)) < 64 ? (int)(start - (e - (bit & 63))) : 64;
      unsigned long long w = wbits[bit >> 6] >> (bit & 63);
      for (int q = bit & 63; q < b1; q++, w >>= 1) {
        lo = pred_rate_step(lo, w & 1);
        hi = pred_rate_step(hi, w & 1);
      }
      e += b1 - (bit & 63);
    }
    x = lo;
    if (lo != hi) {
      for (long long w = 4 * kRateWarm;; w *= 4) {
        const int bb = start - w > 0 ? (int)(start - w) : 0;
        lo = bb == 0 ? kRateScale >> 1 : kRateMin;
        hi = bb == 0 ? kRateScale >> 1 : kRateMax;
        for (int e = bb; e < (int)start; e++) {
          const bool u = up(e);
          lo = pred_rate_step(lo, u);
          hi = pred_rate_step(hi, u);
        }
        x = lo;
        if (lo == hi)
          break;
      }
    }
  }
  // ---- the chunk, in two halves: states staged in LDS (row of a lane: kRateHalf + 1 words, so that the lanes'
  // stores fall into different banks), written out 64 consecutive events at a time ----
  for (int h = 0; h < 2; h++) {
    const long long h0 = start + (long long)h * kRateHalf;
    if (active) {
      for (int q = 0; q < kRateHalf; q += 64) {
        const long long e0 = h0 + q;
        if (e0 >= end)
          break;
        const int bit = (int)(e0 - base);  // (a multiple of 64)
        unsigned long long w = wbits[bit >> 6];
        const int cnt = end - e0 < 64 ? (int)(end - e0) : 64;
        for (int t = 0; t < cnt; t++, w >>= 1) {
          stage[lane * (kRateHalf + 1) + q + t] = x;
          x = pred_rate_step(x, w & 1);
        }
      }
    }
    __syncthreads();
    for (int i = 0; i < kRateHalf; i++) {
      const int idx = i * 64 + lane;
      const int c = idx / kRateHalf, t = idx % kRateHalf;
      const long long e = wave_start + (long long)c * kRateChunk + (long long)h * kRateHalf + t;
      if (e < m)
        state[(size_t)e * state_stride] = stage[c * (kRateHalf + 1) + t];
    }
    __syncthreads();
  }
  if (active && write_final && end == m)
    state[(size_t)m * state_stride] = x;
}

// probResGt1 of every predictor: the state in front of the first event at or behind it
__global__ __launch_bounds__(256) void
pred_rate_gather_kernel(
  const int32_t* __restrict__ rank, const int32_t* __restrict__ evstate, int n, int k, int32_t* __restrict__ rm)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    rm[(size_t)i * 6 + 3 + k] = evstate[rank[i]];
}

// did this pass change a value?  (and keep the values for the next comparison)
__global__ __launch_bounds__(256) void
pred_values_diff_kernel(const int32_t* __restrict__ values, int32_t* __restrict__ prev, size_t count, int32_t* flag)
{
  bool diff = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const int32_t v = values[i];
    diff |= v != prev[i];
    prev[i] = v;
  }
  if (__any(diff) && (threadIdx.x & 63) == 0)
    atomicOr(flag, 1);
}

// ---- reconstruction: the DAG walked forward --------------------------------
template<int C, bool ENC, bool INTER = false>
__global__ __launch_bounds__(256) void
pred_dag_kernel(PredCtx cx)
{
  GPCC_VGPR_FLOOR_64();
  const int lane = lane_id();
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
    cx.rec, 0, (int)((size_t)cx.n * 16), 0x00020000);
  const int maxcand = cx.max_direct + !cx.avg_disabled;
  const int64_t clip_max = ((int64_t)1 << cx.bitdepth) - 1;
  for (;;) {
    int tk = 0;
    if (lane == 0)
      tk = atomicAdd(&cx.ticket[1], 1);
    tk = __shfl(tk, 0);
    const int64_t base64 = (int64_t)tk * 64;
    if (base64 >= cx.n)
      break;
    if (__hip_atomic_load(cx.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      break;
    const int base = (int)base64;
    const int i = base + lane;
    const bool live = i < cx.n;
    int cnt = 0, pt = 0;
    int nidx[3] = {0, 0, 0};
    int32_t nwt[3] = {0, 0, 0};
    int32_t val[3] = {0, 0, 0};
    int32_t col[3] = {0, 0, 0};
    int64_t wgt[2] = {1, 1};
    Quantizer q[2] = {make_quantizer(4), make_quantizer(4)};
    int8_t icpc[3] = {0, 0, 0};
    if (live) {
      cnt = cx.nc[i];
      pt = cx.indexes[i];
      for (int j = 0; j < 3; j++) {
        nidx[j] = j < cnt ? cx.ni[3 * (size_t)i + j] : 0;
        nwt[j] = j < cnt ? cx.nw[3 * (size_t)i + j] : 0;
      }
      const int r = pred_range_of(cx, i);
      const int layer = cx.range_qlayer[r];
      const int o0 = cx.qp_off ? cx.qp_off[2 * (size_t)pt] : 0;
      const int o1 = cx.qp_off ? cx.qp_off[2 * (size_t)pt + 1] : 0;
      const int qp0 = clip(cx.layer_qp[layer][0] + o0, 4, cx.max_qp);
      const int qp1 = clip(cx.layer_qp[layer][1] + o1 + qp0, 4, cx.max_qp);
      q[0] = make_quantizer(qp0);
      q[1] = make_quantizer(qp1);
      const int64_t w = (int64_t)cx.qw[i];
      wgt[0] = (w < q[0].step ? w : (int64_t)q[0].step) >> 8;
      wgt[1] = (w < q[1].step ? w : (int64_t)q[1].step) >> 8;
      if (C == 3 && cx.icp_enabled) {
        const int l = cx.range_lod[r];
        for (int k = 0; k < 3; k++)
          icpc[k] = cx.icp[3 * l + k];
      }
      for (int k = 0; k < C; k++) {
        if (ENC)
          col[k] = (cx.src ? cx.src : cx.attrs)[(size_t)pt * C + k];
        else
          val[k] = cx.values[(size_t)i * C + k];
      }
    }
    int32_t nbv[3][C];
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < C; k++)
        nbv[j][k] = 0;
    uint32_t have = 0;  // neighbours received
    const uint32_t want = live ? (1u << cnt) - 1 : 0;
    if (INTER) {
      // neighbours in the reference frame: their values do not wait for anybody
      for (int j = 0; j < 3; j++)
        if (j < cnt && nidx[j] >= cx.n) {
          nbv[j][0] = cx.frame_attr[nidx[j] - cx.n];
          have |= 1u << j;
        }
    }
    int32_t myrec[C];
    for (int k = 0; k < C; k++)
      myrec[k] = 0;
    int mydone = 0;
    bool pending = live;
    unsigned spins = 0;
    while (__any(pending)) {
      // neighbours inside the claim: lane to lane
      for (int j = 0; j < 3; j++) {
        const int src = nidx[j] - base;
        const bool inw = j < cnt && src >= 0 && (!INTER || nidx[j] < cx.n);
        const int sl = inw ? src : lane;
        const int d = __shfl(mydone, sl);
        int32_t v[C];
        for (int k = 0; k < C; k++)
          v[k] = __shfl(myrec[k], sl);
        if (inw && d && !((have >> j) & 1)) {
          for (int k = 0; k < C; k++)
            nbv[j][k] = v[k];
          have |= 1u << j;
        }
      }
      // neighbours of earlier claims: poll their granules, loads first
      {
        const uint32_t todo = pending ? want & ~have : 0;
        u32x4 g[3];
        for (int j = 0; j < 3; j++) {
          g[j] = u32x4{0, 0, 0, 0};
          if (((todo >> j) & 1) && nidx[j] < base)
            g[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, nidx[j] * 16, 0, /*sc1*/ 16);
        }
        for (int j = 0; j < 3; j++)
          if (((todo >> j) & 1) && nidx[j] < base && g[j].w == cx.tag) {
            nbv[j][0] = (int32_t)g[j].x;
            nbv[j][1 % C] = C > 1 ? (int32_t)g[j].y : nbv[j][1 % C];
            nbv[j][2 % C] = C > 2 ? (int32_t)g[j].z : nbv[j][2 % C];
            have |= 1u << j;
          }
      }
      const bool ready = pending && have == want;
      if (ready) {
        // predModeEligibleColor / ...Refl
        bool elig = false;
        if (cnt > 1 && cx.max_direct) {
          int32_t best = 0;
          for (int k = 0; k < C; k++) {
            int32_t lo = nbv[0][k], hi = nbv[0][k];
            for (int j = 1; j < 3; j++)
              if (j < cnt) {
                lo = min(lo, nbv[j][k]);
                hi = max(hi, nbv[j][k]);
              }
            best = k == 0 ? hi - lo : max(best, hi - lo);
          }
          elig = best >= cx.threshold;
        }
        int mode = 0;
        if (!ENC && elig) {
          if (C == 1) {
            // decodePredModeRefl
            int a = abs(val[0]);
            const int sg = val[0] < 0 ? -1 : 1;
            if (maxcand == 4) {
              mode = a & 3;
              a >>= 2;
            } else if (maxcand == 3) {
              mode = a & 1;
              a >>= 1;
              if (mode > 0) {
                mode += a & 1;
                a >>= 1;
              }
            } else if (maxcand == 2) {
              mode = a & 1;
              a >>= 1;
            }
            val[0] = sg * a;
          } else {
            // decodePredModeColor
            const int s1 = val[1 % C] < 0 ? -1 : 1, s2 = val[2 % C] < 0 ? -1 : 1;
            const int a1 = abs(val[1 % C]), a2 = abs(val[2 % C]);
            if (maxcand == 4) {
              val[1 % C] = s1 * (a1 >> 1);
              val[2 % C] = s2 * (a2 >> 1);
              mode = ((a1 & 1) << 1) + (a2 & 1);
            } else if (maxcand == 3) {
              val[1 % C] = s1 * (a1 >> 1);
              mode = a1 & 1;
              if (a1 & 1) {
                val[2 % C] = s2 * (a2 >> 1);
                mode += a2 & 1;
              }
            } else if (maxcand == 2) {
              val[1 % C] = s1 * (a1 >> 1);
              mode = a1 & 1;
            }
          }
          mode += cx.avg_disabled;
        }
        if (ENC && elig) {
          // decidePredModeRefl / decidePredModeColor with the rate model as it is
          // before this predictor
#pragma clang fp contract(off)
          const int32_t* rm = cx.rm + (size_t)i * 6;
          const int dis = cx.avg_disabled;
          mode = dis;
          int64_t p0[3] = {0, 0, 0};
          if (dis) {
            for (int k = 0; k < C; k++)
              p0[k] = nbv[0][k];
          } else {
            for (int k = 0; k < C; k++) {
              int64_t s = 0;
              for (int j = 0; j < 3; j++)
                if (j < cnt)
                  s += (int64_t)(uint32_t)nwt[j] * nbv[j][k];
              p0[k] = (uint16_t)div_exp2_round_half_inf(s, 8);
            }
          }
          if (C == 1) {
            int64_t rq = quantize(q[0], ((int64_t)col[0] - p0[0]) << 8);
            int64_t best = (int64_t)pred_rate_refl(cx, rm, maxcand, (int32_t)rq, mode - dis);
            for (int j = 0; j < 3; j++) {
              if (j < dis || j >= cnt || j >= cx.max_direct)
                continue;
              rq = quantize(q[0], ((int64_t)col[0] - (int64_t)nbv[j][0]) << 8);
              const int64_t score = (int64_t)pred_rate_refl(cx, rm, maxcand, (int32_t)rq, j + !dis);
              if (score < best) {
                best = score;
                mode = j + 1;
              }
            }
          } else {
            int32_t c3[3] = {col[0], col[1 % C], col[2 % C]};
            int64_t r[3];
            pred_colour_residuals(cx.icp_enabled != 0, c3, p0, icpc, q, r);
            int dist = pred_colour_distortion(clip_max, c3, p0, q);
            double rate = pred_rate_colour(cx, rm, maxcand, r, 0);
            double best = dist + rate * 0.14 * (q[0].step >> 8);
            for (int j = 0; j < 3; j++) {
              if (j < dis || j >= cnt || j >= cx.max_direct)
                continue;
              const int64_t np[3] = {nbv[j][0], nbv[j][1 % C], nbv[j][2 % C]};
              pred_colour_residuals(cx.icp_enabled != 0, c3, np, icpc, q, r);
              dist = pred_colour_distortion(clip_max, c3, np, q);
              rate = pred_rate_colour(cx, rm, maxcand, r, j + !dis);
              const double score = dist + rate * 0.14 * (q[0].step >> 8);
              if (score < best) {
                best = score;
                mode = j + 1;
              }
            }
          }
        }
        // PCCPredictor::predictColor / predictReflectance
        int64_t pr[C];
        for (int k = 0; k < C; k++)
          pr[k] = 0;
        if (mode > cnt) {
        } else if (mode > 0) {
          for (int k = 0; k < C; k++)
            pr[k] = mode == 1 ? nbv[0][k] : (mode == 2 ? nbv[1][k] : nbv[2][k]);
        } else {
          for (int k = 0; k < C; k++) {
            int64_t s = 0;
            for (int j = 0; j < 3; j++)
              if (j < cnt)
                s += (int64_t)(uint32_t)nwt[j] * nbv[j][k];
            pr[k] = (uint16_t)div_exp2_round_half_inf(s, 8);
          }
        }
        int64_t residual0 = 0;
        for (int k = 0; k < C; k++) {
          const Quantizer qq = q[k ? 1 : 0];
          const int64_t weight = wgt[k ? 1 : 0];
          const int64_t icpterm = C == 3 ? ((int64_t)icpc[k] * residual0 + 2) >> 2 : 0;
          int64_t rr;
          if (ENC) {
            int64_t residual = col[k] - pr[k];
            int64_t rq = quantize(qq, (residual * weight) << 8);
            rr = ((mul_i64_u32(rq, qq.step) + 128) >> 8) / weight;
            if (C == 3 && cx.icp_enabled && k > 0) {
              residual -= icpterm;
              rq = quantize(qq, (residual * weight) << 8);
              rr = ((mul_i64_u32(rq, qq.step) + 128) >> 8) / weight;
              rr += icpterm;
            }
            val[k] = (int32_t)rq;
            if (k == 0)
              residual0 = rr;
          } else {
            rr = ((mul_i64_u32((int64_t)val[k], qq.step) + 128) >> 8) / weight;
            const int64_t residual = rr;
            rr += icpterm;
            if (!k && cx.icp_enabled)
              residual0 = residual;
          }
          int64_t v = pr[k] + rr;
          v = v < 0 ? 0 : (v > clip_max ? clip_max : v);
          myrec[k] = (int32_t)(uint16_t)v;
        }
        if (ENC && elig) {
          // encodePredModeRefl (:749-772) / encodePredModeColor (:988-1016): the mode
          // travels in the low bits of the coded magnitudes
          const int m = mode - cx.avg_disabled;
          if (C == 1) {
            const int sg = val[0] < 0 ? -1 : 1;
            int a = val[0] < 0 ? -val[0] : val[0];
            if (maxcand == 4) {
              a = (a << 2) + m;
            } else if (maxcand == 3) {
              if (m > 0)
                a = (a << 1) + (m - 1);
              a = (a << 1) + (m > 0);
            } else if (maxcand == 2) {
              a = (a << 1) + m;
            }
            val[0] = sg * a;
          } else {
            const int s1 = val[1 % C] < 0 ? -1 : 1, s2 = val[2 % C] < 0 ? -1 : 1;
            const int a1 = val[1 % C] < 0 ? -val[1 % C] : val[1 % C];
            const int a2 = val[2 % C] < 0 ? -val[2 % C] : val[2 % C];
            if (maxcand == 4) {
              val[1 % C] = s1 * ((a1 << 1) + (m >> 1));
              val[2 % C] = s2 * ((a2 << 1) + (m & 1));
            } else if (maxcand == 3) {
              const int p1 = m ? 1 : 0;
              val[1 % C] = s1 * ((a1 << 1) + p1);
              if (p1)
                val[2 % C] = s2 * ((a2 << 1) + (m - p1));
            } else if (maxcand == 2) {
              val[1 % C] = s1 * ((a1 << 1) + m);
            }
          }
        }
        {
          const u32x4 st = {(uint32_t)myrec[0], (uint32_t)myrec[1 % C], (uint32_t)myrec[2 % C], cx.tag};
          __builtin_amdgcn_raw_buffer_store_b128(st, rsrc, i * 16, 0, /*sc1*/ 16);
        }
        for (int k = 0; k < C; k++) {
          cx.attrs[(size_t)pt * C + k] = myrec[k];
          if (ENC)
            cx.values[(size_t)i * C + k] = val[k];
        }
        mydone = 1;
        pending = false;
      }
      if (!__any(ready)) {
        if (++spins > (1u << 22)) {
          if (lane == 0)
            atomicExch(cx.error, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
}

}  // namespace gpcc
