// raht_levels.hpp -- the per-level RAHT block kernels.
//
// One 2x2x2 sibling block (a parent node and its <= 8 children) is mapped
// onto EIGHT ADJACENT LANES of a wavefront: lane t of the group owns child
// position t (and, after the forward butterflies, coefficient position t).
// The three butterfly stages of fwd/invTransformBlock222
// (tmc3/RAHT.cpp:671-737) pair positions that differ in bit 0, 1, 2, i.e.
// lane^1, lane^2, lane^4: each stage is one cross-lane exchange, eight
// blocks per wavefront advance in lock step, and no LDS or scratch array
// holds the block.
//
// Modes (template parameter):
//   kAnalyze : lossy encoder, first pass.  Forward-transforms source and
//              prediction, writes the TENTATIVE quantised coefficients and
//              one RDOQ descriptor per coefficient (the zero-run state of
//              tmc3/RAHT.cpp:1576-1670 is resolved by rdoq.hpp afterwards).
//   kSynth   : decoder, and encoder last pass.  Transforms the prediction,
//              adds the de-quantised coefficients, inherits the DC, inverse
//              transforms and stores the children's reconstruction.
//   kFused   : integer-Haar encoder (no RDOQ): both in one pass.
#pragma once

#include "raht_common.hpp"

namespace gpcc {

enum LevelMode { kAnalyze = 0, kSynth = 1, kFused = 2, kLossySub = 3 };

constexpr uint32_t kDescNever = 0x7fffffffu;  // threshold that never passes
constexpr uint32_t kDescZero = 0x80000000u;   // all components quantise to 0

// The parameter block is read-only for the whole call: addressed through the
// constant address space its fields are scalar loads (s_load, scalar cache,
// lgkmcnt).  Through a generic pointer they are VECTOR loads of a uniform
// value whose s_waitcnt vmcnt(0) also waits for every store in flight -- on
// the sub-node kernel's dependency chain, behind its write-through stores.
typedef const __attribute__((address_space(4))) gpcc_raht_params* ParamsConst;

struct LevelCtx {
  TreeView tv;
  const gpcc_raht_params* params;  // device copy
  const SliceSched* sched;
  const int32_t* attr_prefix;      // P[N+1][C]        (sum mode, encoder)
  const int32_t* const* haar_lf;   // [nlev] -> [M][C] (Haar encoder)
  const int32_t* const* asc_qp;    // [nlev] -> [M][2] (region qp), or null
  int64_t* rec[2];                 // scaled reconstruction   [N][C]
  int64_t* rec_us[2];              // unscaled reconstruction [N][C]
  int32_t* nneigh[2];              // numParentNeigh          [N]
  int32_t* dqp[2];                 // descent-time node qp    [N][2]
  int32_t* coeffs;                 // planar per slice
  uint32_t* desc;                  // RDOQ descriptors, one per coefficient
  int64_t* ptrans;                 // [N][C] transformed prediction per coefficient position
                                   // (analyze -> synthesis record, raht_tile.hpp)
  int32_t li;                      // children level of this launch
  const struct SharedLut* lut;     // tables built by lut_init_kernel
  int32_t* worklist;               // [cap] parents with >= 2 children (or non-ext)
  int32_t* work_count;             // [nlev] entries of the worklist per level
  unsigned long long* scan_state;  // [<=1024] prepass look-back words
  // sub-node prediction (raht_subnode.hpp)
  uint8_t* pocc;                   // [cap] child occupancy of every parent of this level (prepass)
  uint32_t* mbox;                  // [N*C][4] 16-byte granules {value lo, hi, tag, 0}: the
                                   // children reconstructed by THIS launch (data is the flag)
  uint32_t mtag;                   // tag of this launch, never reused while mbox lives
  int32_t* ticket;                 // [nlev][8] wave-round tickets
  int32_t* error;                  // set when a bounded spin expires
  unsigned long long* rdoq_state;  // [cap] per worklist block: RDOQ hand-off word
  int32_t* slice_l;                // [2][S] last RDOQ reset carried between levels (sub-node path: by level parity)
};

// ctx.X[parity] with a per-lane parity, as a select between the two kernel
// arguments.  Indexing the argument array with a vector value makes the
// compiler fetch the pointer from the kernarg segment with a vector load in
// front of every access: a dependent memory round trip, whose s_waitcnt
// vmcnt(0) on gfx9 also waits for every store still in flight (10 us per
// commit of the sub-node kernel, behind its write-through granule stores).
template<typename T>
__device__ __forceinline__ T*
par2(T* const (&a)[2], int p)
{
  // base + offset keeps it an address computation on a kernel-argument
  // pointer (a plain select is folded back into one vector load of the
  // selected argument slot); both buffers of a pair come from one arena
  const ptrdiff_t d = a[1] - a[0];
  return a[0] + (p ? d : (ptrdiff_t)0);
}

// Small-weight tables.  Near the leaves almost every node weight is a
// small integer, so the butterfly coefficients (a, b) of a (wl, wr) pair
// and the 1/sqrt(w), sqrt(w) normalisers are looked up in LDS instead of
// running the fixed-point Newton iteration three times per butterfly; the
// tables are filled by the same functions, so the values are identical.
constexpr int kSmallW = 16;   // butterfly pairs with wl, wr < 16
constexpr int kSmallN = 64;   // normalisers with w < 64

struct SharedLut {
  RsqrtLut rsqrt;
  int32_t bfly_a[kSmallW * kSmallW];
  int32_t bfly_b[kSmallW * kSmallW];
  int32_t norm_rs[kSmallN];   // irsqrt(w) >> 25
  int32_t norm_sq[kSmallN];   // isqrt(w << 30)
};

// RahtKernel ctor (tmc3/RAHT.cpp:596-604)
__device__ __forceinline__ void
raht_coeffs_slow(int32_t wl, int32_t wr, const RsqrtLut& lut, int64_t* a, int64_t* b)
{
  const uint64_t rs = irsqrt((uint64_t)wl + (uint64_t)wr, lut);
  *a = (int64_t)(((uint64_t)isqrt((uint64_t)wl << 30, lut) * rs) >> 40);
  *b = (int64_t)(((uint64_t)isqrt((uint64_t)wr << 30, lut) * rs) >> 40);
}

// Fill the tables once per context (lut_init_kernel) ...
__global__ __launch_bounds__(256) void
lut_init_kernel(SharedLut* g)
{
  __shared__ RsqrtLut rs;
  constexpr uint16_t r3[96] = {GPCC_RSQRT_R3};
  constexpr uint32_t rc[96] = {GPCC_RSQRT_RC};
  for (int i = threadIdx.x; i < 96; i += blockDim.x) {
    rs.r3[i] = r3[i];
    rs.rc[i] = rc[i];
    g->rsqrt.r3[i] = r3[i];
    g->rsqrt.rc[i] = rc[i];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kSmallW * kSmallW; i += blockDim.x) {
    const int wl = i / kSmallW, wr = i % kSmallW;
    int64_t a = 0, b = 0;
    if (wl && wr)
      raht_coeffs_slow(wl, wr, rs, &a, &b);
    g->bfly_a[i] = (int32_t)a;
    g->bfly_b[i] = (int32_t)b;
  }
  for (int i = threadIdx.x; i < kSmallN; i += blockDim.x) {
    g->norm_rs[i] = i ? (int32_t)(irsqrt((uint64_t)i, rs) >> (40 - kFpFrac)) : 0;
    g->norm_sq[i] = i ? (int32_t)isqrt((uint64_t)i << (2 * kFpFrac), rs) : 0;
  }
}

// ... and stage them into LDS at the start of every workgroup (3.4 KB,
// L2 resident).
__device__ __forceinline__ void
load_lut(SharedLut* s, const SharedLut* __restrict__ g)
{
  static_assert(sizeof(SharedLut) % 4 == 0, "word copy");
  const uint32_t* src = reinterpret_cast<const uint32_t*>(g);
  uint32_t* dst = reinterpret_cast<uint32_t*>(s);
  for (int i = threadIdx.x; i < (int)(sizeof(SharedLut) / 4); i += blockDim.x)
    dst[i] = src[i];
  __syncthreads();
}

__device__ __forceinline__ void
raht_coeffs(int32_t wl, int32_t wr, const SharedLut& L, int64_t* a, int64_t* b)
{
  if (wl < kSmallW && wr < kSmallW) {
    *a = L.bfly_a[wl * kSmallW + wr];
    *b = L.bfly_b[wl * kSmallW + wr];
    return;
  }
  raht_coeffs_slow(wl, wr, L.rsqrt, a, b);
}

// value / sqrt(weight) (tmc3/RAHT.cpp:1474-1481, 1780-1787)
__device__ __forceinline__ int64_t
scale_rsqrt(int64_t v, int32_t weight, const SharedLut& L)
{
  if (weight < kSmallN)
    return fp_mul_c(v, L.norm_rs[weight]);
  const uint64_t w = (uint64_t)weight;
  const int shift = w > 1024 ? ilog2_u64(w - 1) >> 1 : 0;
  const int64_t rs = (int64_t)(irsqrt(w, L.rsqrt) >> (40 - shift - kFpFrac));
  return fp_mul_c(v >> shift, rs);
}

// sqrt(weight) in Q15 (tmc3/RAHT.cpp:1487-1488)
__device__ __forceinline__ int64_t
sqrt_weight(int32_t weight, const SharedLut& L)
{
  if (weight < kSmallN)
    return L.norm_sq[weight];
  return (int64_t)isqrt((uint64_t)weight << (2 * kFpFrac), L.rsqrt);
}

// QpSet::quantizers (tmc3/quantization.cpp:165-174)
template<typename ParamsP>
__device__ __forceinline__ void
qpset_quantizers(
  ParamsP p, int layer, int off0, int off1, Quantizer q[2])
{
  const int qp0 = clip(p->layer_qp[layer][0] + off0, 4, p->max_qp);
  const int qp1 = clip(p->layer_qp[layer][1] + off1 + qp0, 4, p->max_qp);
  q[0] = make_quantizer(qp0 + p->fixed_point_qp_offset);
  q[1] = make_quantizer(qp1 + p->fixed_point_qp_offset);
}

// findNeighbour (tmc3/RAHT.cpp:272-293): lower_bound inside a window of
// |d| entries before / after `from`, clamped to the slice's node range.
// (A per-level hash table was measured instead of this search: the level
// kernels are VALU-bound, not latency-bound, so the look-ups bought nothing
// while the 3.3 M atomic inserts cost 1 ms -- see DESIGN.md.)
__device__ __forceinline__ int
find_in_window(
  const int64_t* __restrict__ key, int first, int last, int from,
  int64_t value, int64_t d)
{
  int lo, end;
  if (d >= 0) {
    lo = from;
    end = (d + 1 < (int64_t)(last - from)) ? from + (int)(d + 1) : last;
  } else {
    end = from;
    lo = (-d < (int64_t)(from - first)) ? from - (int)(-d) : first;
  }
  int hi = end;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (key[mid] < value)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo == end)
    return -1;
  return key[lo] == value ? lo : -1;
}

// neighbour tables of findNeighbours / intraDcPred
// (tmc3/RAHT.cpp:314-326, 438-440), packed for constant indexing
__device__ __forceinline__ uint32_t
neigh_mask(int i)
{
  constexpr uint8_t m[19] = {255, 240, 204, 170, 192, 160, 136, 3,  5, 15,
                             17,  51,  85,  10,  34,  12,  68,  48, 80};
  return m[i];
}
__device__ __forceinline__ uint32_t
neigh_offset(int i)
{
  constexpr uint8_t o[19] = {0, 35, 21, 14, 49, 42, 28, 1,  2, 3,
                             4, 5,  6,  10, 12, 17, 20, 33, 34};
  return o[i];
}

// LUT_LOG[min(|q|, 15)] of the rate estimate (tmc3/RAHT.cpp:1601-1606) as
// selects.  The estimate is only consulted when the magnitudes of a
// coefficient sum to < 3, i.e. for |q| <= 2; a table indexed by a lane value
// would be a vector load from constant memory on the dependency chain.
__device__ __forceinline__ int
rate_log_small(int64_t aq)
{
  return aq == 0 ? 0 : (aq == 1 ? 256 : 406);
}

// Smallest zero-run length for which RDOQ zeroes a coefficient
// (tmc3/RAHT.cpp:1617-1637).  The rate term is a non-decreasing step
// function of trainZeros: LUTbins for 0..10, then 12 + 2*bitlen(tz - 10).
__device__ __forceinline__ uint32_t
rdoq_threshold(int64_t dist2, int64_t lambda, int rate_coeff, uint32_t limit)
{
  const int64_t d = (int64_t)((uint64_t)dist2 << 26);
  const int rc = (rate_coeff + 128) >> 8;
  // (rate of the class, first trainZeros of the class) -- literal, no table
#define GPCC_RDOQ_CLASS(rate, tz) \
  if (d < lambda * ((rate) + rc))  \
    return (tz);
  GPCC_RDOQ_CLASS(1, 0)
  GPCC_RDOQ_CLASS(2, 1)
  GPCC_RDOQ_CLASS(3, 2)
  GPCC_RDOQ_CLASS(5, 3)
  GPCC_RDOQ_CLASS(7, 5)
  GPCC_RDOQ_CLASS(9, 7)
  GPCC_RDOQ_CLASS(11, 9)
#undef GPCC_RDOQ_CLASS
  for (int b = 1; b < 31; b++) {
    const uint32_t tz = 10u + (1u << (b - 1));
    if (tz > limit)
      break;
    if (d < lambda * (12 + 2 * b + rc))
      return tz;
  }
  return kDescNever;
}

// Pre-pass of a level, ONE THREAD PER PARENT (64 blocks in flight per
// wavefront instead of 8): single-child blocks are finished here -- no
// prediction, no coefficient, the three butterfly stages only move the
// inherited DC to the child's position and the child has its parent's
// weight, so its reconstruction IS the parent's (tmc3/RAHT.cpp:1382-1401
// with the extension).  Sparse (lidar) clouds are mostly such chains.
// Every other processed block is appended to the level's worklist, which
// the 8-lanes-per-block kernels then walk densely.
template<int C>
__global__ __launch_bounds__(256) void
raht_level_prepass_kernel(LevelCtx ctx)
{
  __shared__ int wave_cnt[4];
  __shared__ int base_s;
  if (tree_failed(ctx.tv))
    return;
  const TreeView& tv = ctx.tv;
  const int li = ctx.li;
  const bool ext = ctx.params->raht_extension != 0;
  const int num_parents = tv.soff[li + 1][tv.num_slices];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  // sub-node lossy encoder: this level accumulates its last reset on top of
  // the value the previous level left (raht_subnode.hpp reads [prev], writes [cur])
  if (ctx.rdoq_state && ctx.slice_l && blockIdx.x == 0)
    for (int s = threadIdx.x; s < tv.num_slices; s += blockDim.x)
      ctx.slice_l[(li & 1) * tv.num_slices + s] = ctx.slice_l[((li + 1) & 1) * tv.num_slices + s];
  // workgroup b owns the b-th contiguous range of 256-parent chunks, so the
  // worklist comes out in ascending block order (the dependency order of
  // sub-node prediction, and the coefficient order)
  const int64_t chunks = ((int64_t)num_parents + 255) >> 8;
  const int64_t per = (chunks + gridDim.x - 1) / gridDim.x;
  const int64_t gbeg = min((int64_t)blockIdx.x * per, chunks);
  const int64_t gend = min(gbeg + per, chunks);

  // pass 1: how many blocks does this workgroup append?
  int mine = 0;
  for (int64_t chunk = gbeg; chunk < gend; chunk++) {
    const int j = (int)(chunk * 256) + threadIdx.x;
    if (j < num_parents) {
      const int s = find_slice(tv.soff[li + 1], tv.num_slices, j);
      if (ctx.sched[s].lvl[li].processed) {
        const int nchild = tv.fc[li + 1][j + 1] - tv.fc[li + 1][j];
        mine += !(ext && nchild == 1);
      }
    }
  }
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1)
    mine += __shfl_xor(mine, d);
  if (lane == 0)
    wave_cnt[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    // exclusive prefix over workgroups by decoupled look-back.  One 8-byte
    // word per workgroup: {epoch:16 | kind:16 | count:32}; kind 1 = this
    // workgroup's own count, 2 = inclusive prefix.  The grid (<= 1024
    // workgroups of 256 threads) is resident, so every predecessor publishes.
    const unsigned long long ep = (unsigned long long)(li + 1) << 48;
    const unsigned total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    unsigned excl = 0;
    if (blockIdx.x > 0) {
      __hip_atomic_store(
        &ctx.scan_state[blockIdx.x], ep | (1ull << 32) | total, __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
      int k = (int)blockIdx.x - 1;
      for (;;) {
        const unsigned long long v = __hip_atomic_load(
          &ctx.scan_state[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 48) != (unsigned long long)(li + 1)) {
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        excl += (unsigned)v;
        if (((v >> 32) & 0xffff) == 2)
          break;
        k--;
      }
    }
    __hip_atomic_store(
      &ctx.scan_state[blockIdx.x], ep | (2ull << 32) | (excl + total),
      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == gridDim.x - 1)
      ctx.work_count[li] = (int)(excl + total);
    base_s = (int)excl;
  }
  __syncthreads();
  int running = base_s;

  // pass 2: finish the single-child blocks, append the others in order
  for (int64_t chunk = gbeg; chunk < gend; chunk++) {
    const int j = (int)(chunk * 256) + threadIdx.x;
    bool real = false;
    if (j < num_parents) {
      const int s = find_slice(tv.soff[li + 1], tv.num_slices, j);
      const LevelSched e = ctx.sched[s].lvl[li];
      if (e.processed) {
        const int c0 = tv.fc[li + 1][j];
        const int nchild = tv.fc[li + 1][j + 1] - c0;
        if (ctx.pocc) {
          uint32_t o = 0;
          for (int u = 0; u < nchild; u++)
            o |= 1u << (int)(tv.key[li][c0 + u] & 7);
          ctx.pocc[j] = (uint8_t)o;
        }
        if (ext && nchild == 1) {
          const int pt0 = tv.pt_off[s];
          const int64_t prow = (int64_t)pt0 + (j - tv.soff[li + 1][s]);
          const int64_t crow = (int64_t)pt0 + (c0 - tv.soff[li][s]);
          const int pp = e.parity ^ 1, cp = e.parity;
#pragma unroll
          for (int k = 0; k < C; k++) {
            par2(ctx.rec_us, cp)[crow * C + k] = par2(ctx.rec_us, pp)[prow * C + k];
            par2(ctx.rec, cp)[crow * C + k] = par2(ctx.rec, pp)[prow * C + k];
          }
          par2(ctx.nneigh, cp)[crow] = 19;
          if (ctx.asc_qp) {
            par2(ctx.dqp, cp)[crow * 2] = par2(ctx.dqp, pp)[prow * 2];
            par2(ctx.dqp, cp)[crow * 2 + 1] = par2(ctx.dqp, pp)[prow * 2 + 1];
          }
        } else {
          real = true;
        }
      }
    }
    const unsigned long long m = __ballot(real);
    __syncthreads();  // previous iteration's readers are done
    if (lane == 0)
      wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = running;
    for (int w = 0; w < wave; w++)
      off += wave_cnt[w];
    if (real)
      ctx.worklist[off + __popcll(m & ((1ull << lane) - 1))] = j;
    running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
  }
}

// The block kernels are bound by dependent-load latency (rocprofv3: VALU
// active 12 % of wave cycles), so residency is bought with registers:
// GPCC_LEVEL_WAVES waves per SIMD (see DESIGN.md for the measured sweep).
// (measured on 1M lidar / dense, forward+inverse: 4 waves 4.90 / 3.33 ms,
// 5 waves 4.59 / 3.20 ms -- the C=1 kernels need 98-104 registers -- 6 and 8
// waves spill and lose)
#ifndef GPCC_LEVEL_WAVES
#define GPCC_LEVEL_WAVES 5
#endif
template<int C, int MODE>
__global__ __launch_bounds__(256, GPCC_LEVEL_WAVES) void
raht_level_kernel(LevelCtx ctx)
{
  __shared__ SharedLut lut_s;
  if (tree_failed(ctx.tv))
    return;
  {
    // the grid is sized from a host-side bound; workgroups beyond the
    // level's real work leave before touching anything
    int64_t b0, b1;
    xcd_chunk(((int64_t)ctx.work_count[ctx.li] + 31) >> 5, &b0, &b1);
    if (b0 >= b1)
      return;
  }
  load_lut(&lut_s, ctx.lut);
  const SharedLut& lut = lut_s;

  constexpr bool kEnc = MODE != kSynth;
  constexpr bool kRecon = MODE != kAnalyze;
  const TreeView& tv = ctx.tv;
  const ParamsConst prm = (ParamsConst)ctx.params;
  const int li = ctx.li;
  const int t = threadIdx.x & 7;
  const bool haar = prm->integer_haar_enable_flag != 0;
  const bool ext = prm->raht_extension != 0;

  const int num_work = ctx.work_count[li];
  int64_t gbeg, gend;
  {
    // blocks are dealt out in XCD-contiguous chunks of 32 (one workgroup
    // iteration), see xcd_chunk()
    const int64_t rounds = ((int64_t)num_work + 31) >> 5;
    xcd_chunk(rounds, &gbeg, &gend);
  }
  for (int64_t round = gbeg; round < gend; round++) {
    const int wi = (int)(round * 32) + (threadIdx.x >> 3);
    const bool live = wi < num_work;
    const int j = live ? ctx.worklist[wi] : 0;
    // ---- locate the block -------------------------------------------
    int s = 0;
    LevelSched e;
    e.processed = 0;
    if (live) {
      s = find_slice(tv.soff[li + 1], tv.num_slices, j);
      e = ctx.sched[s].lvl[li];
    }
    // all shuffles below run in wave-uniform control flow; lanes of dead
    // groups carry zeros and store nothing
    const bool on = live && e.processed;
    const int sp0 = on ? tv.soff[li + 1][s] : 0;      // slice's parents
    const int sp1 = on ? tv.soff[li + 1][s + 1] : 0;
    const int sc0 = on ? tv.soff[li][s] : 0;          // slice's children
    const int pt0 = on ? tv.pt_off[s] : 0;
    const int n_s = on ? tv.pt_off[s + 1] - pt0 : 0;
    const int c0 = on ? tv.fc[li + 1][j] : 0;
    const int nchild = on ? tv.fc[li + 1][j + 1] - c0 : 0;
    const int pj = j - sp0;
    const int par_par = e.parity ^ 1, cur_par = e.parity;
    const int64_t prow = (int64_t)pt0 + pj;  // parent row in rec buffers

    // ---- children -> positions ---------------------------------------
    const int64_t ckey = t < nchild ? tv.key[li][c0 + t] : 0;
    const uint32_t occ = group8_or(t < nchild ? 1u << (int)(ckey & 7) : 0u);
    const bool has = (occ >> t) & 1;
    const int child = c0 + popc32(occ & ((1u << t) - 1));
    const int64_t crow = (int64_t)pt0 + (child - sc0);
    int32_t w = 0;
    int64_t src[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      src[k] = 0;
    if (has) {
      const int f0 = tv.fp[li][child], f1 = tv.fp[li][child + 1];
      w = f1 - f0;
      if (kEnc) {
        if (haar) {
          const int32_t* lf = ctx.haar_lf[li];
#pragma unroll
          for (int k = 0; k < C; k++)
            src[k] = fp_from_int(lf[(size_t)child * C + k]);
        } else {
#pragma unroll
          for (int k = 0; k < C; k++)
            src[k] = fp_from_int((int32_t)(
              (uint32_t)ctx.attr_prefix[(size_t)f1 * C + k]
              - (uint32_t)ctx.attr_prefix[(size_t)f0 * C + k]));
        }
      }
    }

    // ---- node qp on the way down (see oracle/raht_oracle.c,
    //      descend_block_qp; tmc3/RAHT.cpp:185-189 vs :246-253) ----------
    int32_t nq0 = 0, nq1 = 0;
    if (ctx.asc_qp) {
      int32_t a0 = 0, a1 = 0;
      if (has) {
        a0 = ctx.asc_qp[li][(size_t)child * 2];
        a1 = ctx.asc_qp[li][(size_t)child * 2 + 1];
      }
      // ascent averages of the pair / quad this position belongs to
      int32_t wa = w, b0 = a0, b1 = a1;   // current sub-tree weight, avg
      int32_t st_w[3], st_a0[3], st_a1[3], st_pw[3];
#pragma unroll
      for (int st = 0; st < 3; st++) {
        const int bit = 1 << st;
        const int32_t pw = lane_xor8(wa, bit);
        const int32_t p0 = lane_xor8(b0, bit), p1 = lane_xor8(b1, bit);
        st_w[st] = wa;
        st_a0[st] = b0;
        st_a1[st] = b1;
        st_pw[st] = pw;
        if (wa && pw) {
          b0 = (b0 + p0) >> 1;
          b1 = (b1 + p1) >> 1;
        } else if (pw) {
          b0 = p0;
          b1 = p1;
        }
        wa += pw;
      }
      // descend: the sub-tree containing this position is the RIGHT one
      // of a real pair -> its own ascent average, otherwise inherit
      int32_t d0 = on ? par2(ctx.dqp, par_par)[prow * 2] : 0;
      int32_t d1 = on ? par2(ctx.dqp, par_par)[prow * 2 + 1] : 0;
#pragma unroll
      for (int st = 2; st >= 0; st--) {
        const int bit = 1 << st;
        if ((t & bit) && st_w[st] && st_pw[st]) {
          d0 = st_a0[st];
          d1 = st_a1[st];
        }
      }
      if (has) {
        nq0 = d0 >> 4;
        nq1 = d1 >> 4;
        if (kRecon) {
          par2(ctx.dqp, cur_par)[crow * 2] = d0;
          par2(ctx.dqp, cur_par)[crow * 2 + 1] = d1;
        }
      }
    }

    // ---- butterfly weights + coefficients (mkWeightTree :742) ----------
    int32_t wl[3], wr[3];
    int64_t ca[3], cb[3];
    int32_t cw = w;
#pragma unroll
    for (int st = 0; st < 3; st++) {
      const int bit = 1 << st;
      const int32_t pw = lane_xor8(cw, bit);
      const bool left = !(t & bit);
      wl[st] = left ? cw : pw;
      wr[st] = left ? pw : cw;
      ca[st] = cb[st] = 0;
      if (wl[st] && wr[st]) {
        if (!haar)
          raht_coeffs(wl[st], wr[st], lut, &ca[st], &cb[st]);
        cw = wl[st] + wr[st];
      } else {
        cw = left ? wl[st] + wr[st] : 0;
      }
    }

    // ---- inter-level prediction (tmc3/RAHT.cpp:1391-1432) --------------
    const bool inherit_dc = !e.is_root;
    const bool pred_in_level =
      on && inherit_dc && prm->raht_prediction_enabled_flag != 0;
    bool enable_pred = pred_in_level;
    int neigh_count = 0;
    int64_t pred[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      pred[k] = 0;

    bool do_search = false;
    if (pred_in_level) {
      if (ext && nchild == 1) {
        enable_pred = false;
        neigh_count = 19;
      } else if (par2(ctx.nneigh, par_par)[prow] < prm->raht_prediction_threshold0) {
        enable_pred = false;
      } else {
        do_search = true;
      }
    }
    // (group-uniform; other groups of the wave idle through the shuffles)
    int pn[3] = {-1, -1, -1};  // neighbour i = 1 + t + 8*slot
    {
      // the three lower_bound searches of a lane advance in lock step, so
      // their probes are in flight together (12 dependent steps, not 36)
      int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, end[3] = {0, 0, 0};
      int64_t want[3] = {0, 0, 0};
      if (do_search) {
        const int64_t cur_pos = tv.key[li + 1][j];
        const uint64_t base = morton3d_add((uint64_t)cur_pos, ~0ull);
        const int64_t range = prm->raht_prediction_search_range;
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
          const int i = 1 + t + 8 * slot;
          if (i < 19 && (occ & neigh_mask(i))) {
            const int64_t np = (int64_t)morton3d_add(base, neigh_offset(i));
            int64_t d = np - cur_pos;
            if (d >= 0) {
              d = d >= range ? range : d;
              lo[slot] = j;
              end[slot] = (d + 1 < (int64_t)(sp1 - j)) ? j + (int)(d + 1) : sp1;
            } else {
              d = (-d) >= range ? range : -d;
              end[slot] = j;
              lo[slot] = (d < (int64_t)(j - sp0)) ? j - (int)d : sp0;
            }
            hi[slot] = end[slot];
            want[slot] = np;
          }
        }
      }
      const int64_t* __restrict__ pkey = tv.key[li + 1];
      while (__any((lo[0] < hi[0]) | (lo[1] < hi[1]) | (lo[2] < hi[2]))) {
        int mid[3];
        int64_t kv[3];
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
          mid[slot] = lo[slot] + ((hi[slot] - lo[slot]) >> 1);
          kv[slot] = lo[slot] < hi[slot] ? pkey[mid[slot]] : 0;
        }
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
          if (lo[slot] < hi[slot]) {
            if (kv[slot] < want[slot])
              lo[slot] = mid[slot] + 1;
            else
              hi[slot] = mid[slot];
          }
        }
      }
#pragma unroll
      for (int slot = 0; slot < 3; slot++) {
        if (lo[slot] < end[slot] && pkey[lo[slot]] == want[slot])
          pn[slot] = lo[slot];
      }
    }
    {
      int found = (pn[0] >= 0) + (pn[1] >= 0) + (pn[2] >= 0);
      found = group8_sum(found);
      if (do_search) {
        neigh_count = found + 1;
        if (neigh_count < prm->raht_prediction_threshold1)
          enable_pred = false;
      }
    }
    // intraDcPred (tmc3/RAHT.cpp:421-589), parent-level neighbours
    {
      const bool run = do_search && enable_pred;
      int wsum = 0;
      if (__any(run)) {
      int64_t lim_lo = 0, lim_hi = 0;
      const int64_t* __restrict__ prec = par2(ctx.rec, par_par);
      const int64_t rbase = (int64_t)pt0 - sp0;
      // every lane fetches the values of the neighbours it searched (and of
      // the parent itself) in ONE round trip; the 19-step loop below then
      // runs on registers -- with the loads inside it, each step's outlier
      // test serialised the next step's load
      int64_t nb_v[3][C], own_v[C];
#pragma unroll
      for (int k = 0; k < C; k++)
        own_v[k] = run ? prec[(rbase + j) * C + k] : 0;
#pragma unroll
      for (int slot = 0; slot < 3; slot++)
#pragma unroll
        for (int k = 0; k < C; k++)
          nb_v[slot][k] = (run && pn[slot] >= 0) ? prec[(rbase + pn[slot]) * C + k] : 0;
#pragma unroll
      for (int i = 0; i < 19; i++) {
        int q;
        int64_t v[C];
        if (i == 0) {
          q = j;
#pragma unroll
          for (int k = 0; k < C; k++)
            v[k] = own_v[k];
        } else {
          const int owner = (threadIdx.x & 56) | ((i - 1) & 7);
          q = __shfl(pn[(i - 1) >> 3], owner);
#pragma unroll
          for (int k = 0; k < C; k++)
            v[k] = shfl_i64(nb_v[(i - 1) >> 3][k], owner);
        }
        if (!run || q < 0)
          continue;
        if (i) {
          if (10 * v[0] <= lim_lo || 10 * v[0] >= lim_hi)
            continue;
        } else {
          lim_lo = 2 * v[0];
          lim_hi = 25 * v[0];
        }
        if (has && ((neigh_mask(i) >> t) & 1)) {
          const int64_t pw = prm->pred_weight_parent[i];
          wsum += (int)pw;
          const int64_t mul = ext ? pw : (pw << kFpFrac);
#pragma unroll
          for (int k = 0; k < C; k++)
            pred[k] += v[k] * mul;
        }
      }
      }
      if (run && has) {
        const int64_t div = pred_divisor(wsum);
#pragma unroll
        for (int k = 0; k < C; k++) {
          pred[k] = fp_mul_c(pred[k], div);
          if (haar)
            pred[k] = (pred[k] >> kFpFrac) << kFpFrac;
        }
      }
    }

    // ---- normalise (tmc3/RAHT.cpp:1445-1499) ---------------------------
    if (!haar && w > 1) {
      if (kEnc) {
#pragma unroll
        for (int k = 0; k < C; k++)
          src[k] = scale_rsqrt(src[k], w, lut);
      }
      if (enable_pred) {
        const int64_t sq = sqrt_weight(w, lut);
#pragma unroll
        for (int k = 0; k < C; k++)
          pred[k] = fp_mul_c(pred[k], sq);
      }
    }

    // ---- forward butterflies (tmc3/RAHT.cpp:671-701) -------------------
    // (group-uniform decision which buffers to transform, :1504-1533)
#pragma unroll
    for (int st = 0; st < 3; st++) {
      const int bit = 1 << st;
      const bool left = !(t & bit);
      const bool both = wl[st] && wr[st];
      const bool swap = !wl[st] && wr[st];
#pragma unroll
      for (int k = 0; k < C; k++) {
        if (kEnc) {
          const int64_t own = src[k], oth = shfl_xor_i64(own, bit);
          if (both) {
            if (haar) {
              const int64_t hf = left ? oth - own : own - oth;
              src[k] = left ? own + ((hf >> (1 + kFpFrac)) << kFpFrac) : hf;
            } else {
              src[k] = left ? fp_mul_c(oth, cb[st]) + fp_mul_c(own, ca[st])
                            : fp_mul_c(own, ca[st]) - fp_mul_c(oth, cb[st]);
            }
          } else if (swap) {
            src[k] = oth;
          }
        }
        {
          const int64_t own = pred[k], oth = shfl_xor_i64(own, bit);
          if (enable_pred) {
            if (both) {
              if (haar) {
                const int64_t hf = left ? oth - own : own - oth;
                pred[k] = left ? own + ((hf >> (1 + kFpFrac)) << kFpFrac) : hf;
              } else {
                pred[k] = left ? fp_mul_c(oth, cb[st]) + fp_mul_c(own, ca[st])
                               : fp_mul_c(own, ca[st]) - fp_mul_c(oth, cb[st]);
              }
            } else if (swap) {
              pred[k] = oth;
            }
          }
        }
      }
    }

    // ---- coefficient slot of this position (scanBlock :776-791) --------
    const uint32_t present = group8_bits(on && cw != 0) | (on ? 1u : 0u);
    // scan order 0,4,2,1,6,5,3,7 -> scan position of t
    const int spos = (0x74516230u >> (4 * t)) & 7;
    const uint32_t pscan = ((present >> 0) & 1) | (((present >> 4) & 1) << 1)
      | (((present >> 2) & 1) << 2) | (((present >> 1) & 1) << 3)
      | (((present >> 6) & 1) << 4) | (((present >> 5) & 1) << 5)
      | (((present >> 3) & 1) << 6) | (((present >> 7) & 1) << 7);
    const int rank = popc32(pscan & ((1u << spos) - 1));
    const bool coded = on && ((present >> t) & 1) && (t != 0 || !inherit_dc);
    // slice-relative coefficient index
    const int cidx = e.coeff_base
      + (inherit_dc ? (c0 - sc0) - pj + rank - 1 : rank);
    int32_t* __restrict__ cplane = ctx.coeffs + (size_t)pt0 * C + cidx;

    if (coded) {
      int ac0 = 0, ac1 = 0;
      if (e.ac_layer < prm->num_ac_qp_layers && t) {
        ac0 = prm->ac_qp_offset[e.ac_layer][t - 1][0];
        ac1 = prm->ac_qp_offset[e.ac_layer][t - 1][1];
      }
      Quantizer qa[2];
      qpset_quantizers(prm, e.qp_layer, nq0 + ac0, nq1 + ac1, qa);

      if (kEnc) {
        if (enable_pred) {
#pragma unroll
          for (int k = 0; k < C; k++)
            src[k] -= pred[k];
        }
        if (MODE == kAnalyze) {
          // RDOQ statistics (tmc3/RAHT.cpp:1584-1616)
          Quantizer qr[2];
          qpset_quantizers(prm, e.qp_layer, nq0, nq1, qr);
          int64_t sum_coeff = 0, dist2 = 0;
          int rate_coeff = 0;
#pragma unroll
          for (int k = 0; k < C; k++) {
            const int64_t co = fp_round(src[k]);
            dist2 += co * co;
            int64_t aq = quantize(qr[k ? 1 : 0], co * 256);
            aq = aq < 0 ? -aq : aq;
            sum_coeff += aq;
            rate_coeff += rate_log_small(aq);
          }
          uint32_t d = kDescNever;
          if (sum_coeff < 3) {
            const int64_t l0 = qr[0].step;
            const int64_t lambda = l0 * l0 * (C == 1 ? 25 : 35);
            d = rdoq_threshold(dist2, lambda, rate_coeff, (uint32_t)n_s);
            if (sum_coeff == 0)
              d |= kDescZero;
          }
          ctx.desc[(size_t)pt0 + cidx] = d;
        }
#pragma unroll
        for (int k = 0; k < C; k++) {
          const int64_t co = quantize(qa[k ? 1 : 0], fp_round(src[k]) * 256);
          cplane[(size_t)k * n_s] = (int32_t)co;
          if (MODE == kFused)
            pred[k] += fp_from_int(dequantize(qa[k ? 1 : 0], co));
        }
      } else {
#pragma unroll
        for (int k = 0; k < C; k++) {
          const int64_t co = cplane[(size_t)k * n_s];
          pred[k] += fp_from_int(dequantize(qa[k ? 1 : 0], co));
        }
      }
    }

    if (!kRecon)
      continue;

    // ---- DC inheritance (tmc3/RAHT.cpp:1727-1742) ----------------------
    if (on && inherit_dc && t == 0) {
#pragma unroll
      for (int k = 0; k < C; k++) {
        const int64_t val = par2(ctx.rec_us, par_par)[prow * C + k];
        if (ext)
          pred[k] = val;
        else
          pred[k] = val > 0 ? val << (kFpFrac - 2) : -((-val) << (kFpFrac - 2));
      }
    }

    // ---- inverse butterflies (tmc3/RAHT.cpp:707-737) -------------------
#pragma unroll
    for (int st = 2; st >= 0; st--) {
      const int bit = 1 << st;
      const bool left = !(t & bit);
      const bool both = wl[st] && wr[st];
      const bool swap = !wl[st] && wr[st];
#pragma unroll
      for (int k = 0; k < C; k++) {
        const int64_t own = pred[k], oth = shfl_xor_i64(own, bit);
        if (both) {
          if (haar) {
            // left lane holds lf, right lane hf
            const int64_t lf = left ? own : oth, hf = left ? oth : own;
            const int64_t lv = lf - ((hf >> (1 + kFpFrac)) << kFpFrac);
            pred[k] = left ? lv : hf + lv;
          } else {
            pred[k] = left ? fp_mul_c(own, ca[st]) - fp_mul_c(oth, cb[st])
                           : fp_mul_c(oth, cb[st]) + fp_mul_c(own, ca[st]);
          }
        } else if (swap) {
          pred[k] = oth;
        }
      }
    }

    // ---- store the children's reconstruction (:1754-1806) -------------
    if (has) {
#pragma unroll
      for (int k = 0; k < C; k++) {
        int64_t v = pred[k];
        par2(ctx.rec_us, cur_par)[crow * C + k] = ext ? v : fp_round(v * 4);
        if (!haar && w > 1)
          v = scale_rsqrt(v, w, lut);
        par2(ctx.rec, cur_par)[crow * C + k] = ext ? v : fp_round(v);
      }
      par2(ctx.nneigh, cur_par)[crow] = inherit_dc ? neigh_count : 19;
    }
  }
}

}  // namespace gpcc
