// cx_level.hpp -- the COMPACT level pass of RAHT without sub-node prediction
// (tmc3/RAHT.cpp:1306-1808 with raht_subnode_prediction_enabled_flag = 0 and the
// RAHT extension): ONE LANE PER CHILD of a branching block, one launch per level.
//
// The tile pass (raht_tile.hpp) visits every parent of a level -- three of four
// are single-child links of a chain in a lidar frame -- and gives every block
// eight lanes, of which two or three hold a child.  Here
//   * only blocks with >= 2 children exist (lists of cx_tree.hpp); a chain of
//     single-child nodes is one value slot, never copied level by level;
//   * a wavefront takes ~56 consecutive real children (whole blocks), LANE =
//     CHILD: every lane searches the 6 neighbours of ITS octant (3 faces, 3 edges;
//     the parent is the seventh), sums its own prediction -- no 19-step broadcast --
//     and the butterflies pair lanes through ds_bpermute with the partner derived
//     from the block's occupancy;
//   * the lossy encoder resolves the RDOQ zero-run state (tmc3/RAHT.cpp:1576-1670)
//     inside the pass: a wavefront's coefficients are one 64-wide rdoq_chunk, the
//     state is handed from wavefront to wavefront by decoupled look-back (words
//     as in raht_rdoq.hpp) -- one launch per level instead of analyze / resolve /
//     synthesis, and no transformed prediction written to memory and read back.
// Wavefronts do not synchronise with each other (no barrier after the table load).
//
// Round 4: the arithmetic is a template parameter (raht_arith.hpp).  The pass is bound by
// VALU issue, and half of that was 64 x 32-bit fixed-point products on the quarter-rate
// 32-bit multiplier; with ArithF64 a product is fma + trunc on doubles that hold the same
// integers (exact below 2^53; the kernels check the magnitudes and a slice that leaves the
// range is redone in int64).  The value slots then hold the doubles' bit patterns -- every
// reader is a launch of the same call with the same back end (finish converts).
#pragma once

#include "cx_tree.hpp"
#include "raht_arith.hpp"
#include "raht_levels.hpp"
#include "raht_links.hpp"
#include "raht_rdoq.hpp"

// pivots + 1 of a step of the neighbour search (2: bisection)
#ifndef GPCC_CX_SEARCH_ARY
#define GPCC_CX_SEARCH_ARY 2
#endif
#ifndef GPCC_CX_SEARCH_MASKED
#define GPCC_CX_SEARCH_MASKED 0
#endif

namespace gpcc {

constexpr int kCxG = 57;        // ranks per wavefront; whole blocks: <= 57 + 7 lanes
constexpr int kCxSlices = 64;   // slices whose plan is staged in LDS (more: read from memory)
constexpr int kCxTopTiles = 8;  // a level of at most this many tiles goes into the top levels' single launch

struct CxCtx {
  TreeView tv;
  CxLists cl;
  const gpcc_raht_params* params;
  const SliceSched* sched;
  const int32_t* attr_prefix;  // P[N+1][C] (encoder)
  int64_t* val;                // [2N][C] unscaled reconstruction of every value slot (the back end's bits)
  int64_t* rec;                // [2N][C] scaled reconstruction
  int32_t* nn;                 // [2N]    numParentNeigh of the node that was reconstructed there
  int32_t* coeffs;             // planar per slice
  const SharedLut* lut;
  unsigned long long* tstate;  // [tiles] RDOQ look-back words of this level
  int32_t* slice_l;            // [2][S] last RDOQ reset, by the level's parity in the slice's plan
  int32_t li;                  // children level of this launch
  // neighbour links of the parents' level (raht_links.hpp); null: the lanes search by bisection
  const int32_t* link_rec;
  const int32_t* link_lrec;
};

struct CxSlice {
  int32_t sp0, sp1;   // the slice's nodes in level li + 1
  int32_t sc0;        // first node of the slice in level li
  int32_t pt0, n_s;
  int32_t coeff_end;  // one past the level's last coefficient (slice relative)
  uint32_t procmask;  // bit l: level l is processed in this slice
  LevelSched e;
};

// quantisers of a (slice, level): [0] the RDOQ statistics (no AC offset), [1 + p]
// coefficient position p (tmc3/RAHT.cpp:1584-1616, quantization.cpp:165-174)
struct CxQuant {
  Quantizer q[9][2];
  double inv_lambda;
  int64_t lambda;
};

struct CxSmem {
  SharedLut lut;
  int32_t pw[19];
  uint8_t nid[8][8];  // the 6 neighbours of every octant: x, y, z face, xy, xz, yz edge
  CxSlice sl[kCxSlices];
  CxQuant qt[kCxSlices];
  uint32_t occ[4][32];
  uint32_t found[4][32];
  uint32_t desc[4][64];
};

__device__ __forceinline__ CxSlice
cx_load_slice(const CxCtx& cx, int li, int s)
{
  const TreeView& tv = cx.tv;
  CxSlice r;
  r.sp0 = tv.soff[li + 1][s];
  r.sp1 = tv.soff[li + 1][s + 1];
  r.sc0 = tv.soff[li][s];
  r.pt0 = tv.pt_off[s];
  r.n_s = tv.pt_off[s + 1] - r.pt0;
  const SliceSched* sc = &cx.sched[s];
  r.e = sc->lvl[li];
  const int m = tv.soff[li][s + 1] - r.sc0;
  r.coeff_end = r.e.coeff_base + (r.e.is_root ? m : m - (r.sp1 - r.sp0));
  uint32_t pm = 0;
  for (int l = 0; l < tv.nlev; l++)
    pm |= (uint32_t)(sc->lvl[l].processed != 0) << l;
  r.procmask = pm;
  return r;
}

// slice of parent j (level li + 1) and its plan
__device__ __forceinline__ CxSlice
cx_slice_of(const CxCtx& cx, const CxSmem& sm, int li, int j, int* s_out)
{
  const int S = cx.tv.num_slices;
  if (S <= kCxSlices) {
    int lo = 0, hi = S - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sm.sl[mid].sp1 <= j)
        lo = mid + 1;
      else
        hi = mid;
    }
    *s_out = lo;
    return sm.sl[lo];
  }
  const int s = find_slice(cx.tv.soff[li + 1], S, j);
  *s_out = s;
  return cx_load_slice(cx, li, s);
}

// 8 presence bits -> a mask of 8 nibbles
__device__ __forceinline__ uint32_t
cx_nibbles(uint32_t x)
{
  x = (x | (x << 12)) & 0x000F000Fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x * 15u;
}

__device__ __forceinline__ int64_t
cx_bperm_i64(int src_lane, int64_t v)
{
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)v);
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(v >> 32));
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

// the neighbours child octant o predicts from (tmc3/RAHT.cpp:314-326): the face
// and edge neighbours of the parent on the child's side.  t = 0..2 faces across x,
// y, z; 3..5 edges xy, xz, yz.
__device__ __forceinline__ int
cx_neigh_id(int o, int t)
{
  const int x = (o >> 2) & 1, y = (o >> 1) & 1, z = o & 1;
  switch (t) {
  case 0: return x ? 1 : 9;
  case 1: return y ? 2 : 11;
  case 2: return z ? 3 : 12;
  case 3: return x ? (y ? 4 : 17) : (y ? 15 : 7);
  case 4: return x ? (z ? 5 : 18) : (z ? 13 : 8);
  default: return y ? (z ? 6 : 16) : (z ? 14 : 10);
  }
}

// Morton key with one axis moved by +1 (up) or -1: the axis' bit field (mask m) is
// incremented / decremented modulo its width, as morton3dAdd does (PCCMisc.h:245)
__device__ __forceinline__ uint64_t
cx_axis_step(uint64_t k, uint64_t m, bool up)
{
  const uint64_t t = up ? (k | ~m) + 1 : (k & m) - 1;
  return (t & m) | (k & ~m);
}

// sqrt(w) in Q15 and the 1/sqrt(w) normaliser of a node weight from ONE irsqrt:
// irsqrt(w << 30) == irsqrt(w) >> 15 (the normalisation shifts in bit pairs), so
// isqrt(w << 30) follows from irsqrt(w) for w <= 2^16 (isqrt's first branch).
struct CxNorm {
  int64_t sq;   // isqrt(w << 30)                  (sqrt_weight)
  int64_t rs;   // irsqrt(w) >> (25 - shift)       (scale_rsqrt)
  int shift;
};

__device__ __forceinline__ int64_t
cx_sqrt_from_rsqrt(uint64_t w, uint64_t rs40, const SharedLut& L)
{
  if (w <= 65536)
    return (int64_t)(1 + (((w << 30) * (rs40 >> 15)) >> 40));
  return (int64_t)isqrt(w << (2 * kFpFrac), L.rsqrt);
}

__device__ __forceinline__ CxNorm
cx_norm(int32_t w, const SharedLut& L)
{
  CxNorm r;
  if (w < kSmallN) {
    r.sq = L.norm_sq[w];
    r.rs = L.norm_rs[w];
    r.shift = 0;
    return r;
  }
  const uint64_t rs40 = irsqrt((uint64_t)w, L.rsqrt);
  r.shift = w > 1024 ? ilog2_u64((uint64_t)w - 1) >> 1 : 0;
  r.rs = (int64_t)(rs40 >> (40 - r.shift - kFpFrac));
  r.sq = cx_sqrt_from_rsqrt((uint64_t)w, rs40, L);
  return r;
}

template<class A>
__device__ __forceinline__ typename A::T
cx_scale(typename A::T v, const CxNorm& nm, typename A::Coef rs)
{
  return A::mulc(A::shr(v, nm.shift), rs);
}

template<class VT>
__device__ __forceinline__ VT
cx_bperm_v(int src_lane, VT v)
{
  return __builtin_bit_cast(VT, cx_bperm_i64(src_lane, __builtin_bit_cast(int64_t, v)));
}

// butterfly coefficients of (wl, wr) (RahtKernel, tmc3/RAHT.cpp:596-604) given the
// square roots of the two weights, and the square root of their sum for the next stage
__device__ __forceinline__ void
cx_coeffs(
  int32_t wl, int32_t wr, int64_t sql, int64_t sqr, const SharedLut& L, int64_t* a, int64_t* b,
  int64_t* sqw)
{
  const uint64_t w = (uint64_t)wl + (uint64_t)wr;
  if (wl < kSmallW && wr < kSmallW) {
    *a = L.bfly_a[wl * kSmallW + wr];
    *b = L.bfly_b[wl * kSmallW + wr];
    *sqw = L.norm_sq[w];
    return;
  }
  const uint64_t rs = irsqrt(w, L.rsqrt);
  *a = (int64_t)(((uint64_t)sql * rs) >> 40);
  *b = (int64_t)(((uint64_t)sqr * rs) >> 40);
  *sqw = cx_sqrt_from_rsqrt(w, rs, L);
}

// rdoq_threshold() of raht_levels.hpp with the reciprocal of lambda at hand
__device__ __forceinline__ uint32_t
cx_rdoq_threshold(int64_t dist2, int64_t lambda, double inv_lambda, int rate_coeff, uint32_t limit)
{
  const uint64_t d = (uint64_t)dist2 << 26;
  const uint64_t lam = (uint64_t)lambda;
  const int rc = (rate_coeff + 128) >> 8;
  constexpr uint64_t kCap = 128;
  uint64_t q = kCap;
  if (d < lam * kCap) {
    q = (uint64_t)((double)d * inv_lambda);
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
  int bb = (m - 12 + 1) >> 1;
  bb = bb < 1 ? 1 : bb;
  if (bb > 30)
    return kDescNever;
  const uint32_t tz = 10u + (1u << (bb - 1));
  return tz > limit ? kDescNever : tz;
}

__device__ __forceinline__ void
cx_fill_quant(ParamsConst prm, const LevelSched& e, int c, CxQuant* qt, int entry)
{
  int a0 = 0, a1 = 0;
  const int pos = entry - 1;
  if (pos >= 1 && e.ac_layer < prm->num_ac_qp_layers) {
    a0 = prm->ac_qp_offset[e.ac_layer][pos - 1][0];
    a1 = prm->ac_qp_offset[e.ac_layer][pos - 1][1];
  }
  Quantizer q[2];
  qpset_quantizers(prm, e.qp_layer, a0, a1, q);
  qt->q[entry][0] = q[0];
  qt->q[entry][1] = q[1];
  if (entry == 0) {
    const int64_t l0 = q[0].step;
    qt->lambda = l0 * l0 * (c == 1 ? 25 : 35);
    qt->inv_lambda = 1.0 / (double)qt->lambda;
  }
}

// Where a wavefront's time goes (experiment builds only, -DGPCC_CX_PROF: s_memtime
// at the phase boundaries, lane 0, summed over all tiles; gpcc_debug_cx_prof).
#ifdef GPCC_CX_PROF
__device__ unsigned long long g_cx_prof[16];
__device__ unsigned long long g_cx_prof_n[4];
struct CxProf {
  unsigned long long last;
  __device__ CxProf() { last = __builtin_amdgcn_s_memtime(); }
  __device__ void mark(int phase)
  {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    if (lane_id() == 0)
      atomicAdd(&g_cx_prof[phase], t - last);
    last = t;
  }
  __device__ void count(int slot, unsigned n)
  {
    if (lane_id() == 0)
      atomicAdd(&g_cx_prof[slot], (unsigned long long)n);
  }
};
#else
struct CxProf {
  __device__ void mark(int) {}
  __device__ void count(int, unsigned) {}
};
#endif

// ---- tables shared by the workgroup (tables of level li: ends with a barrier) ---------
template<int C>
__device__ __forceinline__ void
cx_level_setup(const CxCtx& cx, CxSmem& sm, int li)
{
  const ParamsConst prm = (ParamsConst)cx.params;
  const int S = cx.tv.num_slices;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(cx.lut);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&sm.lut);
    for (int i = threadIdx.x; i < (int)(sizeof(SharedLut) / 4); i += blockDim.x)
      dst[i] = src[i];
    if (threadIdx.x < 19)
      sm.pw[threadIdx.x] = prm->pred_weight_parent[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 48) {
      const int o = (threadIdx.x - 64) / 6, t = (threadIdx.x - 64) % 6;
      sm.nid[o][t] = (uint8_t)cx_neigh_id(o, t);
    }
    if (S <= kCxSlices) {
      if ((int)threadIdx.x >= 128 && (int)threadIdx.x < 128 + S)
        sm.sl[threadIdx.x - 128] = cx_load_slice(cx, li, threadIdx.x - 128);
      for (int i = threadIdx.x; i < S * 9; i += blockDim.x)
        cx_fill_quant(prm, cx.sched[i / 9].lvl[li], C, &sm.qt[i / 9], i % 9);
    }
    __syncthreads();
  }
}

// ---- one tile of level li: the wavefront's work -------------------------------------------
template<int C, bool ENC, class A>
__device__ __forceinline__ void
cx_level_tile(const CxCtx& cx, CxSmem& sm, int li, int tile)
{
  typedef typename A::T VT;
  typedef typename A::Coef VC;
  bool in_range = true;  // (ArithF64: the magnitudes that bound every product, raht_arith.hpp)
  const TreeView& tv = cx.tv;
  const int L = li + 1;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const ParamsConst prm = (ParamsConst)cx.params;
  const int S = tv.num_slices;
  const int n = tv.n_total;
  const SharedLut& lut = sm.lut;

  const CxLevelTab* __restrict__ tab = cx.cl.tab;
  const int R = tab->nr[li];
  const int nb = tab->nb[li];
  const int ntiles = (R + kCxG - 1) / kCxG;
  if (tile >= ntiles)
    return;
  CxProf prof;
  const int32_t* __restrict__ bp = cx.cl.bp + tab->boff[li];
  const int32_t* __restrict__ bc = cx.cl.bc + tab->boff[li];
  const int32_t* __restrict__ bq = cx.cl.bq + tab->boff[li] + li;
  const int32_t* __restrict__ rb = cx.cl.rb + tab->roff[li];

  // ---- the tile: blocks whose first child has a rank in [tile G, (tile + 1) G) -----
  int b0, b1;
  {
    const int r_lo = tile * kCxG;
    const int r_hi = r_lo + kCxG;
    b0 = rb[r_lo];
    if (bq[b0] < r_lo)
      b0++;
    if (r_hi >= R) {
      b1 = nb;
    } else {
      b1 = rb[r_hi];
      if (bq[b1] < r_hi)
        b1++;
    }
  }
  const int r0 = bq[b0];
  const int nch = bq[b1] - r0;
  if (nch <= 0) {
    // (the last tile of a level when the block before it took its ranks): the
    // zero-run state passes through
    if (ENC && lane == 0)
      __hip_atomic_store(
        &cx.tstate[tile], ((unsigned long long)(li + 1) << 48) | (1ull << 32), __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const bool on = lane < nch;
  uint32_t* wocc = sm.occ[wave];
  uint32_t* wfound = sm.found[wave];
  uint32_t* wdesc = sm.desc[wave];
  if (lane < 32) {
    wocc[lane] = 0;
    wfound[lane] = 0;
  }
  __builtin_amdgcn_wave_barrier();  // (LDS words of the wavefront: written and read in program order)

  // ---- lane = child: its block, its parent, itself ----------------------------------
  const int r = r0 + (on ? lane : 0);
  const int b = rb[r];
  const int bl = b - b0;                 // block of the tile
  const int j = bp[b];                   // parent (level li + 1)
  const int q0 = bq[b];
  const int u = r - q0;                  // ordinal among the block's children
  const int base_lane = q0 - r0;         // lane of the block's first child
  const int c0 = bc[b];
  const int64_t pkey = tv.key[L][j];
  const uint32_t phold = cx.cl.hold[L][j];
  int s;
  const CxSlice sl = cx_slice_of(cx, sm, li, j, &s);
  const LevelSched e = sl.e;
  const int child = c0 + u;
  const int fa = tv.fp[li][child], fb = tv.fp[li][child + 1];
  const int oct = (int)(tv.key[li][child] & 7);
  const int32_t w = fb - fa;
  const int cslot = u ? fa : n + fb;
  const int pslot = (int)(phold & kCxSlotMask);
  const int ptop = (int)(phold >> kCxSlotBits);
  const bool inherit_dc = !e.is_root;

  atomicOr(&wocc[bl], on ? 1u << oct : 0u);
#ifdef CX_DEBUG
  if (lane < 4 || !(fa >= 0 && fb <= n)) fprintf(stderr, "li %d tile %d lane %d on %d r %d b %d j %d q0 %d u %d c0 %d child %d fa %d fb %d b0 %d b1 %d r0 %d nch %d R %d nb %d\n", li, tile, lane, (int)on, r, b, j, q0, u, c0, child, fa, fb, b0, b1, r0, nch, R, nb);
#endif

  // source sum of the child: difference of the modular prefix sums (the reference
  // accumulates these sums in `int` as well, tmc3/RAHT.cpp:131,196)
  VT src[C];
#pragma unroll
  for (int k = 0; k < C; k++)
    src[k] = A::zero();
  if (ENC) {
#pragma unroll
    for (int k = 0; k < C; k++)
      src[k] = A::from_int((int32_t)(
        (uint32_t)cx.attr_prefix[(size_t)fb * C + k] - (uint32_t)cx.attr_prefix[(size_t)fa * C + k]));
  }

  // the parent's values.  numParentNeigh: 19 when the parent came down a chain
  // through a level its slice processes (the single-child copy sets it,
  // tmc3/RAHT.cpp:1382-1401), else what its own block left
  VT pval[C], prec[C];
  int pneigh = 0;
#pragma unroll
  for (int k = 0; k < C; k++)
    pval[k] = prec[k] = A::zero();
  if (inherit_dc) {
#pragma unroll
    for (int k = 0; k < C; k++) {
      pval[k] = __builtin_bit_cast(VT, cx.val[(size_t)pslot * C + k]);
      prec[k] = __builtin_bit_cast(VT, cx.rec[(size_t)pslot * C + k]);
      in_range = in_range && A::below(pval[k], A::kInvLimit) && A::below(prec[k], A::kRecLimit);
    }
    const uint32_t through = ptop > L ? (sl.procmask >> L) & ((1u << (ptop - L)) - 1u) : 0u;
    pneigh = through ? 19 : cx.nn[pslot];
  }

  __builtin_amdgcn_wave_barrier();
  const uint32_t occ = wocc[bl];

  // ---- inter-level prediction gating (tmc3/RAHT.cpp:1391-1432) ----------------------
  const bool pred_in_level = on && inherit_dc && prm->raht_prediction_enabled_flag != 0;
#ifdef CX_EXP_NOSEARCH  // (experiment builds: what the neighbour search costs)
  const bool do_search = false;
#else
  const bool do_search = pred_in_level && pneigh >= prm->raht_prediction_threshold0;
#endif

  prof.mark(0);  // tile and child set-up
  // ---- neighbour search: the parents at the 6 positions this octant predicts from
  //      (findNeighbours, tmc3/RAHT.cpp:299-368; findNeighbour :272-293 is a
  //      lower_bound limited to raht_prediction_search_range entries either side) -----
  int nq[6];
  if (GPCC_EXPERIMENTS && cx.link_rec) {
    // round 5: the parent's record holds its 18 neighbours (raht_links.hpp) -- one load per neighbour
    // instead of a 12-step bisection; the search window is an index distance
    const int64_t range = prm->raht_prediction_search_range;
    const int rj = do_search ? cx.link_lrec[j] : 0;
    uint32_t fmask = 0;
#pragma unroll
    for (int t = 0; t < 6; t++) {
      const int id = sm.nid[oct][t];
      nq[t] = do_search ? link_lookup(cx.link_rec, rj, id, j, range) : -1;
      fmask |= nq[t] >= 0 ? 1u << id : 0u;
    }
    atomicOr(&wfound[bl], fmask);
    __builtin_amdgcn_wave_barrier();
  } else {
    const int64_t* __restrict__ pk = tv.key[L];
    constexpr uint64_t mz = 0x9249249249249249ull, my = mz << 1, mx = mz << 2;
    const int64_t range = prm->raht_prediction_search_range;
    int lo[6], hi[6], end[6];
    int64_t want[6];
    {
      const uint64_t k = (uint64_t)pkey;
      const uint64_t kx = cx_axis_step(k, mx, (oct & 4) != 0);
      const uint64_t ky = cx_axis_step(k, my, (oct & 2) != 0);
      const uint64_t kz = cx_axis_step(k, mz, (oct & 1) != 0);
      want[0] = (int64_t)kx;
      want[1] = (int64_t)ky;
      want[2] = (int64_t)kz;
      want[3] = (int64_t)((kx & ~my) | (ky & my));
      want[4] = (int64_t)((kx & ~mz) | (kz & mz));
      want[5] = (int64_t)((ky & ~mz) | (kz & mz));
    }
#pragma unroll
    for (int t = 0; t < 6; t++) {
      int64_t d = want[t] - pkey;
      int ga, gb;
      if (d >= 0) {
        d = d >= range ? range : d;
        ga = j;
        gb = (d + 1 < (int64_t)(sl.sp1 - j)) ? j + (int)(d + 1) : sl.sp1;
      } else {
        d = (-d) >= range ? range : -d;
        gb = j;
        ga = (d < (int64_t)(j - sl.sp0)) ? j - (int)d : sl.sp0;
      }
      lo[t] = do_search ? ga : 0;
      hi[t] = end[t] = do_search ? gb : 0;
    }
    // six lower_bounds side by side, no branch inside
    // (a window of the parent keys staged in LDS in front of this search was
    // measured: 19 of 20 look-ups of a lidar frame end there, but its instructions
    // and registers -- 4 waves per SIMD instead of 5 -- cost more than the loads)
#if GPCC_CX_SEARCH_ARY == 1
    // Six lower_bounds side by side as UNIFORM halving: every search adds the same power of two per step when the key in
    // front of the new position is still smaller (and the position stays inside its window) -- an add, a bound test, a
    // clamp, the load, the compare and a select per search and step, half the instructions of the (lo, hi) form below,
    // whose lock-step loop ran to the longest window of the wavefront anyway.  (Experiment: same time -- the pass is bound by
    // the NUMBER of its vector loads, profiles/r06_cx_ta_counters.txt.)  Positions in bytes; a window's end is
    // readable (sentinel entry), so a step that would leave the window reads the end instead.
    {
      const char* __restrict__ pkb = (const char*)pk;
      const uint32_t wmax = (uint32_t)(range < (1 << 28) ? range + 1 : (1 << 28));
      uint32_t pos8[6], gb8[6];
#pragma unroll
      for (int t = 0; t < 6; t++) {
        pos8[t] = (uint32_t)lo[t] << 3;
        gb8[t] = (uint32_t)hi[t] << 3;
      }
      for (uint32_t half8 = (1u << (31 - __builtin_clz(wmax))) << 3; half8 >= 8; half8 >>= 1) {
        uint32_t t8[6];
        int64_t kv[6];
#pragma unroll
        for (int t = 0; t < 6; t++) {
          t8[t] = pos8[t] + half8;
          const uint32_t at = t8[t] < gb8[t] + 8 ? t8[t] : gb8[t] + 8;
          kv[t] = *(const int64_t*)(pkb + (at - 8));
        }
#pragma unroll
        for (int t = 0; t < 6; t++)
          pos8[t] = (t8[t] <= gb8[t] && kv[t] < want[t]) ? t8[t] : pos8[t];
      }
#pragma unroll
      for (int t = 0; t < 6; t++)
        lo[t] = (int)(pos8[t] >> 3);
    }
#elif GPCC_CX_SEARCH_ARY == 3
    // (experiment) two pivots per step: a third of the window left instead of half -- 8 memory round trips for a
    // 2 500-entry window instead of 12, twelve loads in flight per lane: 96 loads instead of 72, a third slower
    while (__any((lo[0] < hi[0]) | (lo[1] < hi[1]) | (lo[2] < hi[2]) | (lo[3] < hi[3])
                 | (lo[4] < hi[4]) | (lo[5] < hi[5]))) {
      int m1[6], m2[6];
      int64_t k1[6], k2[6];
#pragma unroll
      for (int t = 0; t < 6; t++) {
        const uint32_t len = (uint32_t)(hi[t] - lo[t]);
        m1[t] = lo[t] + (int)__umulhi(len, 0x55555556u);      // lo + len / 3
        m2[t] = lo[t] + (int)__umulhi(2 * len, 0x55555556u);  // lo + 2 len / 3 (>= m1; == m1 only for len 1)
        k1[t] = pk[m1[t]];
        k2[t] = pk[m2[t]];
      }
#pragma unroll
      for (int t = 0; t < 6; t++) {
        const bool act = lo[t] < hi[t];
        const bool ge1 = !(k1[t] < want[t]);
        const bool ge2 = !(k2[t] < want[t]);
        const int nlo = ge1 ? lo[t] : (ge2 ? m1[t] + 1 : m2[t] + 1);
        const int nhi = ge1 ? m1[t] : (ge2 ? m2[t] : hi[t]);
        lo[t] = act ? nlo : lo[t];
        hi[t] = act ? nhi : hi[t];
      }
    }
#else
    while (__any((lo[0] < hi[0]) | (lo[1] < hi[1]) | (lo[2] < hi[2]) | (lo[3] < hi[3])
                 | (lo[4] < hi[4]) | (lo[5] < hi[5]))) {
      int mid[6];
      int64_t kv[6];
#pragma unroll
      for (int t = 0; t < 6; t++) {
        mid[t] = lo[t] + ((hi[t] - lo[t]) >> 1);
#if GPCC_CX_SEARCH_MASKED
        // (experiment) a finished search asks for nothing -- 7 % fewer load instructions (most windows are the full
        // search range), same time
        kv[t] = 0;
        if (lo[t] < hi[t])
          kv[t] = pk[mid[t]];
#else
        kv[t] = pk[mid[t]];  // (a finished search reloads key[lo]: the arrays have a sentinel entry)
#endif
      }
#pragma unroll
      for (int t = 0; t < 6; t++) {
        const bool act = lo[t] < hi[t];
        const bool less = kv[t] < want[t];
        lo[t] = (act && less) ? mid[t] + 1 : lo[t];
        hi[t] = (act && !less) ? mid[t] : hi[t];
      }
    }
#endif
    int64_t kf[6];
#pragma unroll
    for (int t = 0; t < 6; t++)
      kf[t] = pk[lo[t]];
    uint32_t fmask = 0;
#pragma unroll
    for (int t = 0; t < 6; t++) {
      const bool hit = lo[t] < end[t] && kf[t] == want[t];
      nq[t] = hit ? lo[t] : -1;
      fmask |= hit ? 1u << sm.nid[oct][t] : 0u;
    }
    atomicOr(&wfound[bl], fmask);
    __builtin_amdgcn_wave_barrier();
  }
  prof.mark(1);  // search
  int neigh_count = 0;
  bool enable_pred = false;
  if (do_search) {
    neigh_count = popc32(wfound[bl]) + 1;
    enable_pred = neigh_count >= prm->raht_prediction_threshold1;
  }

  // ---- intraDcPred for this child (tmc3/RAHT.cpp:421-589, parent-level part) ---------
  VT pred[C];
#pragma unroll
  for (int k = 0; k < C; k++)
    pred[k] = A::zero();
  {
    const bool run = enable_pred;
    uint32_t nh[6];
#pragma unroll
    for (int t = 0; t < 6; t++)
      nh[t] = (run && nq[t] >= 0) ? cx.cl.hold[L][nq[t]] : 0u;
    VT nv[6][C];
#pragma unroll
    for (int t = 0; t < 6; t++)
#pragma unroll
      for (int k = 0; k < C; k++) {
        nv[t][k] = (run && nq[t] >= 0) ? __builtin_bit_cast(VT, cx.rec[(size_t)(nh[t] & kCxSlotMask) * C + k])
                                       : A::zero();
        in_range = in_range && A::below(nv[t][k], A::kRecLimit);
      }
    if (run) {
      const VT lim_lo = A::muli(prec[0], 2), lim_hi = A::muli(prec[0], 25);
      int wsum = sm.pw[0];
#pragma unroll
      for (int k = 0; k < C; k++)
        pred[k] = A::muli(prec[k], sm.pw[0]);
#pragma unroll
      for (int t = 0; t < 6; t++) {
        if (nq[t] >= 0 && !(A::muli(nv[t][0], 10) <= lim_lo || A::muli(nv[t][0], 10) >= lim_hi)) {
          const int pwt = sm.pw[sm.nid[oct][t]];
          wsum += pwt;
#pragma unroll
          for (int k = 0; k < C; k++)
            pred[k] += A::muli(nv[t][k], pwt);
        }
      }
      const VC div = A::coef(pred_divisor(wsum));
#pragma unroll
      for (int k = 0; k < C; k++)
        pred[k] = A::mulc(pred[k], div);
    }
  }

  prof.mark(2);  // prediction gathers + sums
  // ---- normalise (tmc3/RAHT.cpp:1445-1499) -------------------------------------------
  const CxNorm nm = cx_norm(on ? w : 1, lut);
  const VC nm_rs = A::coef(nm.rs);
  if (on && w > 1) {
    if (ENC) {
#pragma unroll
      for (int k = 0; k < C; k++)
        src[k] = cx_scale<A>(src[k], nm, nm_rs);
    }
    if (enable_pred) {
      const VC nm_sq = A::coef(nm.sq);
#pragma unroll
      for (int k = 0; k < C; k++)
        pred[k] = A::mulc(pred[k], nm_sq);
    }
  }
#pragma unroll
  for (int k = 0; k < C; k++)
    in_range = in_range && (!on || (A::below(src[k], A::kFwdLimit) && A::below(pred[k], A::kFwdLimit)));

  // ---- forward butterflies (fwdTransformBlock222, tmc3/RAHT.cpp:671-701; weights as
  //      mkWeightTree :742).  A value stays in its lane; `pos` is where it sits in
  //      the 2x2x2 block, P the occupied positions, M the lane (nibble) of every
  //      position.  Stage s pairs pos and pos ^ (1 << s): both present -> butterfly
  //      (low-pass left, high-pass right); only the right one -> it moves left. -------
  uint32_t P = occ;
  uint32_t M = 0;
  {
    // the u-th child of the block sits at the u-th occupied octant
#pragma unroll
    for (int p = 0; p < 8; p++)
      M |= (uint32_t)popc32(occ & ((1u << p) - 1u)) << (4 * p);
  }
  int pos = oct;
  int32_t cw = on ? w : 0;
  int32_t sqw = (int32_t)nm.sq;  // isqrt(cw << 30), < 2^30
  bool st_both[3], st_left[3];
  int st_lane[3];
  int32_t st_a[3], st_b[3];
#pragma unroll
  for (int st = 0; st < 3; st++) {
    const int bit = 1 << st;
    const uint32_t lm = st == 0 ? 0x55u : (st == 1 ? 0x33u : 0x0Fu);
    const int pp = pos ^ bit;
    const bool partner = on && ((P >> pp) & 1);
    const bool left = !(pos & bit);
    const int plane = base_lane + (int)((M >> (4 * pp)) & 15u);
    const int from = partner ? plane : lane;
    const int32_t ow = __builtin_amdgcn_ds_bpermute(from << 2, cw);
    const int32_t osq = __builtin_amdgcn_ds_bpermute(from << 2, sqw);
    const int32_t wl = left ? cw : ow, wr = left ? ow : cw;
    int64_t ca = 0, cb = 0, nsq = sqw;
    if (partner)
      cx_coeffs(wl, wr, left ? sqw : osq, left ? osq : sqw, lut, &ca, &cb, &nsq);
    st_both[st] = partner;
    st_left[st] = left;
    st_lane[st] = from;
    st_a[st] = (int32_t)ca;
    st_b[st] = (int32_t)cb;
    const VC fa_ = A::coef(ca), fb_ = A::coef(cb);
#pragma unroll
    for (int k = 0; k < C; k++) {
      if (ENC) {
        const VT own = src[k], oth = cx_bperm_v(from, own);
        if (partner)
          src[k] = left ? A::mulc(oth, fb_) + A::mulc(own, fa_) : A::mulc(own, fa_) - A::mulc(oth, fb_);
      }
      {
        const VT own = pred[k], oth = cx_bperm_v(from, own);
        // (a block's lanes share enable_pred)
        if (partner && enable_pred)
          pred[k] = left ? A::mulc(oth, fb_) + A::mulc(own, fa_) : A::mulc(own, fa_) - A::mulc(oth, fb_);
      }
    }
    if (partner) {
      cw = wl + wr;
      sqw = (int32_t)nsq;
    }
    // positions after the stage
    const uint32_t Pl = P & lm, Pr = (P >> bit) & lm;
    const uint32_t moved = cx_nibbles(Pr & ~Pl);
    M = (M & ~moved) | ((M >> (4 * bit)) & moved);
    if (!left && !partner)
      pos = pp;
    P = (Pl | Pr) | ((Pl & Pr) << bit);
  }

  prof.mark(3);  // normalise + forward butterflies
  // ---- coefficient slot of this lane's position (scanBlock :776-791) -------------------
  const uint32_t pscan = ((P >> 0) & 1) | (((P >> 4) & 1) << 1) | (((P >> 2) & 1) << 2)
    | (((P >> 1) & 1) << 3) | (((P >> 6) & 1) << 4) | (((P >> 5) & 1) << 5)
    | (((P >> 3) & 1) << 6) | (((P >> 7) & 1) << 7);
  const int spos = (0x74516230u >> (4 * pos)) & 7;  // scan order 0,4,2,1,6,5,3,7
  const int rank = popc32(pscan & ((1u << spos) - 1u));
  const bool coded = on && (pos != 0 || !inherit_dc);
  const int pj = j - sl.sp0;
  const int cidx = e.coeff_base + (inherit_dc ? (c0 - sl.sc0) - pj + rank - 1 : rank);
  int32_t* __restrict__ cplane = cx.coeffs + (size_t)sl.pt0 * C + cidx;
  const int cblock = cidx - (inherit_dc ? rank - 1 : rank);  // the block's first coefficient

  Quantizer qa[2], qr[2];
  int64_t lambda;
  double inv_lambda;
  if (S <= kCxSlices) {
    const CxQuant& qt = sm.qt[s];
    qa[0] = qt.q[1 + pos][0];
    qa[1] = qt.q[1 + pos][1];
    qr[0] = qt.q[0][0];
    qr[1] = qt.q[0][1];
    lambda = qt.lambda;
    inv_lambda = qt.inv_lambda;
  } else {
    int ac0 = 0, ac1 = 0;
    if (e.ac_layer < prm->num_ac_qp_layers && pos) {
      ac0 = prm->ac_qp_offset[e.ac_layer][pos - 1][0];
      ac1 = prm->ac_qp_offset[e.ac_layer][pos - 1][1];
    }
    qpset_quantizers(prm, e.qp_layer, ac0, ac1, qa);
    qpset_quantizers(prm, e.qp_layer, 0, 0, qr);
    const int64_t l0 = qr[0].step;
    lambda = l0 * l0 * (C == 1 ? 25 : 35);
    inv_lambda = 1.0 / (double)lambda;
  }

  const typename A::Quant qaa[2] = {A::quant(qa[0]), A::quant(qa[1])};
  int32_t qco[C];
#pragma unroll
  for (int k = 0; k < C; k++)
    qco[k] = 0;
  if (ENC) {
    // ---- residual, RDOQ statistics (tmc3/RAHT.cpp:1584-1616), quantisation -----------
    uint32_t d = kDescZero;
    if (coded) {
      if (enable_pred) {
#pragma unroll
        for (int k = 0; k < C; k++)
          src[k] -= pred[k];
      }
      const typename A::Quant qra[2] = {A::quant(qr[0]), A::quant(qr[1])};
      int64_t sum_coeff = 0, dist2 = 0;
      int rate_coeff = 0;
#pragma unroll
      for (int k = 0; k < C; k++) {
        const VT co = A::round_int(src[k]);
        const int64_t coi = A::to_small(co);  // (|co| < 2^20 inside the checked range)
        dist2 += coi * coi;
        int64_t aq = A::quantize(qra[k ? 1 : 0], co);
        aq = aq < 0 ? -aq : aq;
        sum_coeff += aq;
        rate_coeff += rate_log_small(aq);
        qco[k] = A::quantize(qaa[k ? 1 : 0], co);
      }
      d = kDescNever;
      if (sum_coeff < 3) {
        d = cx_rdoq_threshold(dist2, lambda, inv_lambda, rate_coeff, (uint32_t)sl.n_s);
        if (sum_coeff == 0)
          d |= kDescZero;
      }
    }

    // ---- the zero-run state (tmc3/RAHT.cpp:1618-1669): the tile's coefficients in
    //      coding order are one chunk per slice; the incoming last reset comes from
    //      the wavefront before (look-back) or from the level before (slice_l) -----------
    prof.mark(4);  // quantisers, residual, RDOQ statistics
    const unsigned long long ep = (unsigned long long)(li + 1) << 48;
    unsigned long long todo = __ballot(coded);
    bool zero_me = false;
    bool published = false;
    while (todo) {
      const int first = __ffsll((long long)todo) - 1;
      const int s_cur = __shfl(s, first);
      // (lanes are in child order, coefficients in scan order: the segment's first
      // coefficient is the first of its first block)
      const int c_first = __shfl(cblock, first);
      const bool mine = coded && s == s_cur;
      const unsigned long long seg = __ballot(mine);
      todo &= ~seg;
      const bool last_seg = todo == 0;
      const int ncoef = __popcll(seg);
      const int e_base = __shfl(e.coeff_base, first);
      const int e_par = __shfl((int)e.parity, first);
      const int c_end = __shfl(sl.coeff_end, first);
      if (mine)
        wdesc[cidx - c_first] = d;
      __builtin_amdgcn_wave_barrier();
      const bool valid = lane < ncoef;
      const uint32_t dd = valid ? wdesc[lane] : kDescZero;
      const int ci = c_first + lane;
      const bool starts = c_first == e_base;      // the slice's first coefficient of the level
      const bool ends = c_first + ncoef == c_end;  // ... and its last
      int l_in = 0, l_out = 0;
      int tz;
      if (starts) {
        l_in = cx.slice_l[(e_par ^ 1) * S + s_cur];
        l_out = rdoq_chunk(dd, ci, valid, l_in, c_first, &tz);
        if (last_seg && lane == 0)
          __hip_atomic_store(
            &cx.tstate[tile], ep | (3ull << 32) | (uint32_t)l_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        published = published || last_seg;
      } else {
        // (only the tile's first segment continues a slice from the tile before)
        // The incoming last reset lies in [-1, c_first - 1] and zeroing is monotone in
        // it: when the two extremes zero the same coefficients, every value does, and
        // the tile neither waits for its predecessor nor can it be open.  Only a tile
        // with an undecided coefficient (1 in 80 coefficients at qp 34), or the one
        // that hands a transparent state on to the next level, looks back.
        const int la0 = -1, lb0 = c_first - 1;
        int tza, tzb;
        const int la = rdoq_chunk(dd, ci, valid, la0, c_first, &tza);
        const int lb = rdoq_chunk(dd, ci, valid, lb0, c_first, &tzb);
        const int status = lb == lb0 ? kTileTransparent : (la == lb ? kTileClosed : kTileOpen);
        if (last_seg && status != kTileOpen) {
          if (lane == 0)
            __hip_atomic_store(
              &cx.tstate[tile],
              ep | ((unsigned long long)(status == kTileClosed ? 2 : 1) << 32)
                | (uint32_t)(status == kTileClosed ? la : ncoef),
              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          published = true;
        }
        const uint32_t thr0 = dd & kDescNever;
        const bool za = valid && thr0 != kDescNever && (uint32_t)tza >= thr0;
        const bool zb = valid && thr0 != kDescNever && (uint32_t)tzb >= thr0;
        const bool depends = __any(za != zb);
        prof.mark(5);  // hypotheses
        tz = tza;
        l_out = la;  // (closed: the state this tile leaves)
        bool have = !depends && status != kTileOpen && !(ends && status == kTileTransparent);
        prof.count(13, have ? 0 : 1);
        const bool waited = !have;
        bool exact = false;   // the incoming state itself is known (not only bounded)
        int k0 = tile - 1;    // lane v looks at tile k0 - v
        int run = 0;          // coefficients of the transparent tiles right before this one
        unsigned spins = 0;
        while (!have) {
          const int kt = k0 - lane;
          const bool in = kt >= 0;
          unsigned long long wv = 0;
          if (in)
            wv = __hip_atomic_load(&cx.tstate[kt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool ready = in && (wv >> 48) == (unsigned long long)(li + 1);
          const int kind = ready ? (int)((wv >> 32) & 0xffff) : 0;
          const unsigned long long decides = __ballot(ready && kind >= 2);
          const unsigned long long notready = __ballot(in && !ready);
          const unsigned long long stop = decides | notready;
          const int firstw = stop ? __ffsll((long long)stop) - 1 : kWave;
          // a transparent word carries the tile's coefficient count: the zero run in
          // front of this tile is at least as long as the transparent tiles before it
          int add = (lane < firstw && kind == 1) ? (int)(uint32_t)wv : 0;
#pragma unroll
          for (int dd2 = 1; dd2 < kWave; dd2 <<= 1)
            add += __shfl_xor(add, dd2);
          run += add;
          if (!stop) {
            k0 -= kWave;  // 64 transparent tiles: further back (the slice's first tile decides)
            if (k0 < 0) {
              // cannot happen: the walk ends at the tile that starts the slice's level
              if (lane == 0)
                atomicExch(tv.error, 1);
              return;
            }
            continue;
          }
          k0 -= firstw;
          if ((decides >> firstw) & 1) {
            l_in = (int)(uint32_t)__shfl((int)(uint32_t)wv, firstw);
            have = true;
            exact = true;
          } else {
            // the nearest tile that is not transparent has not published yet.  The last
            // reset is at least `run` coefficients back: if that bound already settles
            // every coefficient of this tile, the exact state is not needed (a smooth
            // attribute leaves long zero runs, and the rate thresholds are short)
            const int lbx = c_first - 1 - run;
            int tzx;
            const int lx = rdoq_chunk(dd, ci, valid, lbx, c_first, &tzx);
            const bool zx = valid && thr0 != kDescNever && (uint32_t)tzx >= thr0;
            const bool settled = !__any(za != zx);
            const bool resets = lx != lbx;  // (then la == lx: the same coefficients reset)
            if (settled && (resets || !ends)) {
              have = true;
              l_out = la;
              if (last_seg && status == kTileOpen) {
                if (lane == 0)
                  __hip_atomic_store(
                    &cx.tstate[tile],
                    ep | ((unsigned long long)(resets ? 2 : 1) << 32) | (uint32_t)(resets ? la : ncoef),
                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                published = true;
              }
            } else {
              if (++spins > (1u << 22)) {
                if (lane == 0)
                  atomicExch(tv.error, 1);
                return;
              }
              __builtin_amdgcn_s_sleep(1);
            }
          }
        }
        prof.mark(6);  // look-back wait
        prof.count(10, spins);
        prof.count(11, 1);
        if (exact)
          l_out = rdoq_chunk(dd, ci, valid, l_in, c_first, &tz);
        // A tile that has walked back knows the state it leaves for good: publishing
        // it ends every later walk here (most tiles of a smooth attribute hold no
        // reset at all, and a walk over transparent words costs a round trip per 64).
        if (last_seg && exact) {
          if (lane == 0)
            __hip_atomic_store(
              &cx.tstate[tile], ep | (3ull << 32) | (uint32_t)l_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          published = true;
        }
      }
      if (ends && lane == 0)
        cx.slice_l[e_par * S + s_cur] = l_out;
      // decisions back to the lanes that own the coefficients
      const uint32_t thr = dd & kDescNever;
      const bool zero = valid && thr != kDescNever && (uint32_t)tz >= thr;
      __builtin_amdgcn_wave_barrier();
      wdesc[lane] = zero ? 1u : 0u;
      __builtin_amdgcn_wave_barrier();
      if (mine)
        zero_me = wdesc[cidx - c_first] != 0;
      __builtin_amdgcn_wave_barrier();
    }
    if (!published && lane == 0)  // a tile without a coefficient lets the state pass
      __hip_atomic_store(&cx.tstate[tile], ep | (1ull << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (coded) {
#pragma unroll
      for (int k = 0; k < C; k++) {
        if (zero_me)
          qco[k] = 0;
        cplane[(size_t)k * sl.n_s] = qco[k];
      }
    }
  } else if (coded) {
#pragma unroll
    for (int k = 0; k < C; k++)
      qco[k] = cplane[(size_t)k * sl.n_s];
  }

  prof.mark(7);  // decisions, coefficient stores
  // ---- reconstruction: prediction + de-quantised residual, inherited DC ---------------
  if (!enable_pred) {
#pragma unroll
    for (int k = 0; k < C; k++)
      pred[k] = A::zero();
  }
  if (coded) {
#pragma unroll
    for (int k = 0; k < C; k++)
      pred[k] += A::dequant_fp(qaa[k ? 1 : 0], qco[k]);
  }
  if (on && inherit_dc && pos == 0) {
#pragma unroll
    for (int k = 0; k < C; k++)
      pred[k] = pval[k];  // (tmc3/RAHT.cpp:1727-1742, extension)
  }

#pragma unroll
  for (int k = 0; k < C; k++)
    in_range = in_range && (!on || A::below(pred[k], A::kInvLimit));
  // ---- inverse butterflies (tmc3/RAHT.cpp:707-737) -------------------------------------
#pragma unroll
  for (int st = 2; st >= 0; st--) {
    const VC ca = A::coef(st_a[st]), cb = A::coef(st_b[st]);
#pragma unroll
    for (int k = 0; k < C; k++) {
      const VT own = pred[k], oth = cx_bperm_v(st_lane[st], own);
      if (st_both[st])
        pred[k] = st_left[st] ? A::mulc(own, ca) - A::mulc(oth, cb) : A::mulc(oth, cb) + A::mulc(own, ca);
    }
  }

  // ---- the child's values (:1754-1806) ---------------------------------------------------
  if (on) {
#pragma unroll
    for (int k = 0; k < C; k++) {
      VT v = pred[k];
      cx.val[(size_t)cslot * C + k] = __builtin_bit_cast(int64_t, v);
      if (w > 1)
        v = cx_scale<A>(v, nm, nm_rs);
      cx.rec[(size_t)cslot * C + k] = __builtin_bit_cast(int64_t, v);
    }
    cx.nn[cslot] = inherit_dc ? neigh_count : 19;
  }
  prof.mark(8);  // inverse butterflies, stores
  prof.count(12, 1);
  // ArithF64: a value left the range in which doubles are exact -- the sticky word stops every later
  // kernel of the call (the source attributes stay intact) and the call is redone with ArithI64
  // (host tier) or reports GPCC_ERR_RANGE (device tier)
  if (A::kF64 && __any(!in_range) && lane == 0)
    atomicCAS(tv.error, 0, 3);
}

// One launch per level: a wavefront per tile.  Wavefronts per SIMD, measured on ten 1 M-point
// slices (forward / inverse, ms): left to the compiler (5 / 7: it squeezes the decoder into 66
// registers) 3.38 / 3.24; at most 4: 3.50 / 3.17; at most 5: 3.36 / 2.93; 5-6: 3.30 / 2.91; exactly
// 6: 3.37 / 2.80 -- so the encoder is asked for 5-6 and the one-component decoder for 6 (with more components six
// wavefronts spill).
template<int C, bool ENC, class A = ArithI64>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((ENC || C > 1) ? 5 : 6, 6))) void
cx_level_kernel(CxCtx cx)
{
  __shared__ CxSmem sm;
  if (tree_failed(cx.tv))
    return;
  cx_level_setup<C>(cx, sm, cx.li);
  cx_level_tile<C, ENC, A>(cx, sm, cx.li, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
}

// The top levels -- a few tiles each, and a level needs the one above it -- in ONE launch of
// one workgroup: levels li_hi .. li_lo one after the other, the workgroup's four wavefronts
// take the tiles of a level in turn (a tile that looks back for the RDOQ state finds its
// predecessors in the same pass or an earlier one).  Between two levels the values the first
// wrote are made visible to the whole workgroup (fence: write back, invalidate the L1) and
// the level's tables are rebuilt.  Saves a launch and its latency per level: 7 of the 17
// launches of a 10 x 1 M-point batch, 10 of a single frame's.
template<int C, bool ENC, class A = ArithI64>
__global__ __launch_bounds__(256) void
cx_top_kernel(CxCtx cx, int li_hi, int li_lo)
{
  __shared__ CxSmem sm;
  if (tree_failed(cx.tv))
    return;
  const int wave = threadIdx.x >> 6;
  for (int li = li_hi; li >= li_lo; li--) {
    cx_level_setup<C>(cx, sm, li);
    const int ntiles = (cx.cl.tab->nr[li] + kCxG - 1) / kCxG;
    for (int tile = wave; tile < ntiles; tile += 4)
      cx_level_tile<C, ENC, A>(cx, sm, li, tile);
    __threadfence();
    __syncthreads();
  }
}

}  // namespace gpcc
