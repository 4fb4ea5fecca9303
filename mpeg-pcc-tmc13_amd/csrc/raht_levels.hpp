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
// This header holds what the two families of level kernels share -- the
// context, the small-weight tables, the neighbour tables, the quantiser set-up,
// the RDOQ threshold -- and the prepass of the sub-node family.  The kernels
// themselves: raht_tile.hpp (blocks of a level independent) and
// raht_subnode.hpp (sub-node prediction: blocks depend on earlier ones).
//
// Modes (template parameter of those kernels):
//   kAnalyze : lossy encoder, first pass.  Forward-transforms source and
//              prediction, writes the TENTATIVE quantised coefficients and
//              one RDOQ descriptor per coefficient (the zero-run state of
//              tmc3/RAHT.cpp:1576-1670 is resolved by raht_rdoq.hpp afterwards).
//   kSynth   : decoder.  Transforms the prediction, adds the de-quantised
//              coefficients, inherits the DC, inverse transforms and stores
//              the children's reconstruction.
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

// Attribute inter prediction (raht_inter.hpp; tmc3/RAHT.cpp:1165-1198, 1283-1347, 1504-1549): what a level
// launch needs of the reference frame.  The frame's tree is never built: its points are in Morton order, so
// the node at bit level `lr` with key k is the run of points with pos >> lr == k (two bisections), its weight
// the length of the run and its attribute sum a difference of the frame's modular prefix sums.
struct InterRef {
  const int64_t* pos;     // [n_ref] the frame's Morton codes, ascending (null: no inter prediction)
  const int32_t* prefix;  // [n_ref + 1][C] modular prefix sums of its attributes
  int32_t n_ref;
  int32_t lr;             // bit level of the frame's nodes that line up with the CHILDREN of this launch
  int32_t blocks;         // blocks are matched against the frame in this launch
  int32_t dual;           // encoder: the level is coded twice, the second candidate without the frame
  int32_t filtered;       // the level's filter tap applies (tree depth >= skipInitLayersForFiltering)
  const int32_t* tap;     // the tap (a device word: fixed taps are written by the host, estimated ones by
                          // inter_tap_finish_kernel)
  // integer Haar kernel: the frame's nodes cannot be sums -- its tree is reduced pass by pass with half
  // differences (tmc3/RAHT.cpp:1065-1120 with HaarKernel) -- so the frame gets level arrays of its own
  // (raht_tree.hpp + ascend_*_kernel) and a launch sees the level that lines up (bit level lr = 3 x level)
  const int64_t* hkey;    // [hsoff[1]] keys of the frame's nodes at that level, ascending (null: not Haar)
  const int32_t* hfp;     // [.. + 1] first point (weights are differences)
  const int32_t* hlf;     // [..][C] low-pass values
  const int32_t* hsoff;   // {0, number of nodes}
  uint32_t* idesc;        // the second candidate's RDOQ descriptors, ...
  int64_t* iptrans;       // ... transformed prediction ...
  int32_t* icoeffs;       // ... and coefficients (layouts of desc / ptrans / coeffs)
};

// ---- block records (raht_sweep.hpp: raht_sweep_record_kernel writes them) --------------------------------
enum SweepField {
  kSfW = 0, kSfCa = 1, kSfCb = 4, kSfNsq = 7, kSfNrs = 8, kSfPn = 9, kSfNbc0 = 12, kSfPk = 15, kSfPk2 = 16,
  kSfC0 = 17, kSfSlice = 18, kSweepFields = 19
};
// pk : occupancies of the three neighbours this lane owns (8 bits each) | their single-child bits << 24
//      | the normaliser's shift << 27
// pk2: occupancy of the block (8) | coded-position mask `present` << 8 | neighbours found << 16
//      | butterfly stages with both sides << 21 | stages that only move << 24 | block takes a round << 27

struct SweepRec {
  int32_t* f32;    // [kSweepFields][lanes]
  int64_t* src;    // [C][lanes] forward-transformed source, bit pattern of the launch's arithmetic (encoder)
  uint8_t* occ;    // [lanes / 8] child occupancy of every parent (sweep_occ_kernel)
  int32_t lanes;   // 8 x (parents of all levels of the sweep + 8: a slice's last round reads whole)
  int32_t rbase[kMaxLevels];  // first record of children level li (in parents)
  // records of ONE level by worklist index (raht_level_sub_kernel<.., REC>): record wi belongs to block worklist[wi];
  // null: records of several levels by parent index (raht_sub_sweep_kernel)
  const int32_t* worklist;
  const int32_t* work_count;  // [nlev]
};


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
  InterRef inter;                  // attribute inter prediction (tile kernels instantiated with INTER)
  // neighbour links of the parents' level (raht_links.hpp, sub-node kernels); null: bisection
  const int32_t* link_rec;
  const int32_t* link_lrec;
  // rounds of 8 blocks a wavefront of the sub-node kernels takes per claim: 0 / 1 = one round from one of
  // eight tickets, R > 1 = R consecutive rounds from a single ticket (raht_subnode.hpp)
  int32_t claim_rounds;
  // block records of this level by worklist index (raht_level_sub_kernel<.., REC = true>)
  SweepRec brec;
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
// while the 3.3 M atomic inserts cost 1 ms -- see profiles/HISTORY.md.)
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
  // Every class test is  d < lambda * (rate + rc)  with integers, i.e.
  // floor(d / lambda) < rate + rc: ONE exact quotient q decides them all (the
  // literal form walked up to 37 classes, a 64-bit multiply each, and sat on
  // the dependency chain of the sub-node encoder).  q is capped where no
  // class is left; below the cap d < 2^57, the double quotient is within one
  // of the truth and two integer checks make it exact.
  const uint64_t d = (uint64_t)dist2 << 26;
  const uint64_t lam = (uint64_t)lambda;
  const int rc = (rate_coeff + 128) >> 8;
  constexpr uint64_t kCap = 128;  // > 12 + 2 * 30 + rc for every rc the LUT can give (<= 12)
  uint64_t q = kCap;
  if (d < lam * kCap) {
    q = (uint64_t)((double)d / (double)lam);
    q -= q * lam > d;
    q += (q + 1) * lam <= d;
  }
  const int m = (int)q - rc + 1;  // smallest rate that passes
  if (m <= 1)
    return 0;
  if (m <= 2)
    return 1;
  if (m <= 3)
    return 2;
  if (m <= 5)
    return 3;
  if (m <= 7)
    return 5;
  if (m <= 9)
    return 7;
  if (m <= 11)
    return 9;
  // rate 12 + 2 b for trainZeros in [10 + 2^(b-1), 10 + 2^b), b = 1 .. 30
  int bb = (m - 12 + 1) >> 1;
  bb = bb < 1 ? 1 : bb;
  if (bb > 30)
    return kDescNever;
  const uint32_t tz = 10u + (1u << (bb - 1));
  return tz > limit ? kDescNever : tz;
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

}  // namespace gpcc
